// Dense-stack (discriminator / reward-net) engine on top of the fp32 MFMA GEMM, plus the
// HBM-bound helpers of the discriminator update: RunningNorm (Chan merge), gather+concat
// batch assembly, BCE-with-logits + train statistics, split-K partial reduction, Adam.
#include "common.h"
#include "rn_common.h"
#include "../../include/imitation_hip.h"

namespace {

struct LayerOff {
  long long w, b;
};

inline void layer_offsets(const ia_mlp_desc* d, LayerOff* off, long long* total) {
  long long o = 0;
  for (int l = 0; l < d->n_layers; ++l) {
    off[l].w = o;
    o += (long long)d->dims[l + 1] * d->dims[l];
    off[l].b = o;
    o += d->dims[l + 1];
  }
  *total = o;
}

inline bool desc_ok(const ia_mlp_desc* d) {
  if (!d || d->n_layers < 1 || d->n_layers > IA_MAX_LAYERS) return false;
  for (int l = 0; l <= d->n_layers; ++l)
    if (d->dims[l] <= 0) return false;
  return true;
}

// ---------------------------------------------------------------- reductions / elementwise

__global__ void reduce_partials_kernel(const float* __restrict__ partials, int splits, long long n, float scale,
                                       int accumulate, float* __restrict__ grads, long long stride) {
  // (`stride`: elements between consecutive slabs -- n for whole slabs, more when a PIECE of every slab is reduced)
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  int k = 0;
  for (; k + 8 <= splits; k += 8) {  // 8 independent loads in flight, then a fixed-order sum
    float t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) t[u] = partials[(long long)(k + u) * stride + i];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += t[u];
  }
  for (; k < splits; ++k) s += partials[(long long)k * stride + i];  // fixed order: deterministic
  s *= scale;
  grads[i] = accumulate ? grads[i] + s : s;
}

// split-K slab reduction fused with the Adam step (single minibatch, single GPU): the reduced
// gradient is also written out (it is the flat gradient the API exposes).
// (`partials2`: a second set of slabs -- the gradient penalty's -- whose fixed-order sum is ADDED to the first set's scaled sum:
// the arithmetic of `reduce_partials_kernel` (accumulate = 0), `reduce_partials_kernel` (accumulate = 1, scale 1) and
// `adam_kernel` in one launch, bit for bit)
__device__ __forceinline__ float slab_sum(const float* __restrict__ partials, int splits, long long n, long long i) {
  float s = 0.f;
  int k = 0;
  for (; k + 8 <= splits; k += 8) {
    float t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) t[u] = partials[(long long)(k + u) * n + i];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += t[u];
  }
  for (; k < splits; ++k) s += partials[(long long)k * n + i];
  return s;
}

__global__ void reduce_adam_kernel(const float* __restrict__ partials, int splits, long long n, float scale,
                                   float* __restrict__ grads, float* __restrict__ p, float* __restrict__ m,
                                   float* __restrict__ v, float beta1, float beta2, float eps, float wd,
                                   float step_size, float bc2_sqrt, const float* __restrict__ partials2, int splits2) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float grad = slab_sum(partials, splits, n, i) * scale;
  if (partials2 != nullptr) grad = grad + slab_sum(partials2, splits2, n, i) * 1.0f;
  grads[i] = grad;
  const float pi = p[i];
  if (wd != 0.f) grad = grad + wd * pi;
  float mi = m[i];
  mi = mi + (grad - mi) * (1.f - beta1);
  const float vi = v[i] * beta2 + (1.f - beta2) * grad * grad;
  const float denom = sqrtf(vi) / bc2_sqrt + eps;
  p[i] = pi - step_size * (mi / denom);
  m[i] = mi;
  v[i] = vi;
}

// Step-dependent scalars of torch.optim.Adam from a DEVICE-resident step count, so that a captured launch sequence
// (hipGraph replay of a whole PPO update, `GeneralTowers.ppo_update`) advances them without host arguments:
// t = ++*step; scal = {lr / (1 - b1^t), sqrt(1 - b2^t)} in double like the host side of `ia_adam_step`'s callers.
__global__ void adam_scalars_kernel(long long* __restrict__ step, double lr, const double* __restrict__ lr_dev,
                                    double beta1, double beta2, float* __restrict__ scal) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const long long t = *step + 1;
  *step = t;
  if (lr_dev != nullptr) lr = *lr_dev;   // a learning-rate schedule reaches a captured sequence through device memory
  scal[0] = (float)(lr / (1.0 - pow(beta1, (double)t)));
  scal[1] = (float)sqrt(1.0 - pow(beta2, (double)t));
}

// torch/optim/adam.py _single_tensor_adam for one element: lerp, mul+addcmul, sqrt/bc2_sqrt + eps, addcdiv
__device__ __forceinline__ void adam_element(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                             float* __restrict__ v, long long i, float beta1, float beta2, float eps,
                                             float wd, float step_size, float bc2_sqrt) {
  float grad = g[i];
  const float pi = p[i];
  if (wd != 0.f) grad = grad + wd * pi;
  float mi = m[i];
  mi = mi + (grad - mi) * (1.f - beta1);
  float vi = v[i] * beta2 + (1.f - beta2) * grad * grad;
  const float denom = sqrtf(vi) / bc2_sqrt + eps;
  p[i] = pi - step_size * (mi / denom);
  m[i] = mi;
  v[i] = vi;
}

__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, long long n, float beta1, float beta2, float eps, float wd,
                            float step_size, float bc2_sqrt) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  adam_element(p, g, m, v, i, beta1, beta2, eps, wd, step_size, bc2_sqrt);
}

// the same step with {step_size, bc2_sqrt} read from device memory (adam_scalars_kernel)
__global__ void adam_dev_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                float* __restrict__ v, long long n, float beta1, float beta2, float eps, float wd,
                                const float* __restrict__ scal) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  adam_element(p, g, m, v, i, beta1, beta2, eps, wd, scal[0], scal[1]);
}

// RunningNorm statistics. Stage 1: each block owns a contiguous slab of rows and produces, per
// column, (mean_b, M2_b) by a two-pass over its slab (rows are re-read from L2). Stage 2: one
// block Chan-merges the slabs in slab order and applies the reference's update formula.
__global__ __launch_bounds__(256) void rn_partial_kernel(const float* __restrict__ X, int ldx, int R, int D,
                                                         float* __restrict__ ws) {
  // threads: 256 = 8 row-lanes x 32 column-lanes; loops over columns in steps of 32
  __shared__ float red[8][33];
  const int r0 = blockIdx.x * RN_ROWS_PER_BLOCK;
  const int rows = min(RN_ROWS_PER_BLOCK, R - r0);
  const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
  constexpr int RPT = RN_ROWS_PER_BLOCK / 8;   // rows per thread
  for (int c0 = 0; c0 < D; c0 += 32) {
    const int c = c0 + cl;
    // the thread's RPT values of this column in one batch of unconditional loads (clamped addresses), kept in
    // registers for both passes: the row loop with a load per iteration was a chain of ~64 serial L2 round
    // trips (12 us for a 1.5 MB input); sums run in the same order as before
    float v[RPT];
#pragma unroll
    for (int k = 0; k < RPT; ++k)
      v[k] = X[(long long)(r0 + min(rl + 8 * k, rows - 1)) * ldx + min(c, D - 1)];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < RPT; ++k) s += (c < D && rl + 8 * k < rows) ? v[k] : 0.f;
    red[rl][cl] = s;
    __syncthreads();
    float mean = 0.f;
    if (rl == 0) {
      float t = 0.f;
      for (int k = 0; k < 8; ++k) t += red[k][cl];
      mean = t / (float)rows;
      red[0][cl] = mean;
    }
    __syncthreads();
    mean = red[0][cl];
    __syncthreads();
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
      const float dlt = v[k] - mean;
      q += (c < D && rl + 8 * k < rows) ? dlt * dlt : 0.f;
    }
    red[rl][cl] = q;
    __syncthreads();
    if (rl == 0 && c < D) {
      float t = 0.f;
      for (int k = 0; k < 8; ++k) t += red[k][cl];
      ws[((long long)blockIdx.x * 2 + 0) * D + c] = mean;
      ws[((long long)blockIdx.x * 2 + 1) * D + c] = t;  // M2 of the slab
    }
    __syncthreads();
  }
}

// `nblocks` slabs in groups of `bpg` (one group per data-parallel rank, each covering `rpg` rows);
// R = total rows. One WAVE per column (rn_common.h). `ws_ld` = column count the slab moments were
// written with.
__global__ __launch_bounds__(64) void rn_merge_kernel(const float* __restrict__ ws, int nblocks, int bpg, int rpg,
                                                      int R, int D, int ws_ld, float* __restrict__ mean,
                                                      float* __restrict__ var, const int32_t* __restrict__ count) {
  const int c = blockIdx.x, lane = threadIdx.x;
  const int cnt = *count;
  float b_mean, b_M2;
  rn_wave_batch_moments(ws, nblocks, bpg, rpg, ws_ld, c, lane, b_mean, b_M2);
  if (lane == 0) {
    float mc = mean[c], vc = var[c];
    rn_absorb(mc, vc, cnt, R, b_mean, b_M2 / (float)R);
    mean[c] = mc;
    var[c] = vc;
  }
  // the count is bumped by a tiny follow-up kernel: columns run in different blocks and all read `cnt`
}

// `n_seq` CONSECUTIVE updates (each the merge above of one batch's slab moments, `seq_stride` floats
// apart) applied in order by one launch: the replay of the deferred policy feature-norm updates of a
// round. Same arithmetic per update as rn_merge_kernel; the count seen by update k is cnt0 + k*R.
// The batch moments of the n_seq updates do not depend on one another: wave w of the block reduces update
// k0 + w's slabs (one load round trip + the butterfly for ALL updates at once instead of one per update: the
// sequential form took 35-50 us for 16 updates), then one lane applies the n_seq running updates in order.
constexpr int RN_SEQ_WAVES = 16;
__global__ __launch_bounds__(64 * RN_SEQ_WAVES) void rn_merge_seq_kernel(
    const float* __restrict__ ws_seq, int n_seq, long long seq_stride, int nblocks, int bpg, int rpg, int R, int D,
    int ws_ld, float* __restrict__ mean, float* __restrict__ var, const int32_t* __restrict__ count,
    float* __restrict__ snapshots) {
  __shared__ float s_bm[RN_SEQ_WAVES], s_bq[RN_SEQ_WAVES];
  const int c = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int cnt = *count;
  float mc = mean[c], vc = var[c];
  for (int k0 = 0; k0 < n_seq; k0 += RN_SEQ_WAVES) {
    const int k = k0 + wave;
    if (k < n_seq) {
      float b_mean, b_M2;
      rn_wave_batch_moments(ws_seq + (long long)k * seq_stride, nblocks, bpg, rpg, ws_ld, c, lane, b_mean, b_M2);
      if (lane == 0) { s_bm[wave] = b_mean; s_bq[wave] = b_M2; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      for (int w = 0; w < RN_SEQ_WAVES && k0 + w < n_seq; ++w) {
        rn_absorb(mc, vc, cnt, R, s_bm[w], s_bq[w] / (float)R);
        cnt = rn_count_add(cnt, R);
        if (snapshots != nullptr) {  // statistics as update k's own forward pass sees them
          snapshots[((long long)(k0 + w) * 2 + 0) * D + c] = mc;
          snapshots[((long long)(k0 + w) * 2 + 1) * D + c] = vc;
        }
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    mean[c] = mc;
    var[c] = vc;
  }
}

// EMANorm (util/networks.py:137-201): exponentially weighted statistics from the same slab moments. One wave per
// column; every column reads the OLD inverse learning rate and batch counter, the follow-up kernel bumps them.
//   inv_lr' = inv_lr + decay^num_batches;  lr = 1 / inv_lr';  d = b_mean - mean;  mean += lr d;
//   var += lr (b_var + (1 - lr) d^2 - var)
__global__ __launch_bounds__(64) void ema_merge_kernel(const float* __restrict__ ws, int nblocks, int bpg, int rpg, int R,
                                                       int D, int ws_ld, float* __restrict__ mean,
                                                       float* __restrict__ var, const float* __restrict__ inv_lr,
                                                       const int32_t* __restrict__ num_batches, float decay) {
  const int c = blockIdx.x, lane = threadIdx.x;
  float b_mean, b_M2;
  rn_wave_batch_moments(ws, nblocks, bpg, rpg, ws_ld, c, lane, b_mean, b_M2);
  if (lane == 0) {
    const float ilr = *inv_lr + powf(decay, (float)*num_batches);
    const float lr = 1.f / ilr;
    const float b_var = b_M2 / (float)R;
    const float mc = mean[c], vc = var[c];
    const float dm = b_mean - mc;
    mean[c] = mc + lr * dm;
    const float dv = b_var + (1.f - lr) * (dm * dm) - vc;
    var[c] = vc + lr * dv;
  }
}

__global__ void ema_count_kernel(int32_t* __restrict__ count, long long R, float* __restrict__ inv_lr,
                                 int32_t* __restrict__ num_batches, float decay) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    *inv_lr = *inv_lr + powf(decay, (float)*num_batches);
    *num_batches = *num_batches + 1;
    *count = rn_count_add(*count, R);
  }
}

__global__ void rn_count_kernel(int32_t* __restrict__ count, long long R) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *count = rn_count_add(*count, R);
}

__global__ void rn_apply_kernel(const float* __restrict__ X, int ldx, int R, int D, const float* __restrict__ mean,
                                const float* __restrict__ var, float eps, float* __restrict__ Y, int ldy) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)R * ldy) return;
  const int c = (int)(i % ldy);
  const long long r = i / ldy;
  Y[i] = (c < D) ? (X[r * ldx + c] - mean[c]) / sqrtf(var[c] + eps) : 0.f;
}

__global__ void gather_concat_kernel(const float* __restrict__ obs, const float* __restrict__ act_f32,
                                     const int64_t* __restrict__ act_i64, const float* __restrict__ next_obs,
                                     const uint8_t* __restrict__ dones, const int64_t* __restrict__ idx, int n,
                                     int obs_dim, int act_dim, int use_state, int use_action, int use_next,
                                     int use_done, float* __restrict__ X, int ldx, int row0) {
  // one thread per output element: coalesced row-major writes, gathered reads
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long long)n * ldx) return;
  const int c = (int)(e % ldx);
  const int i = (int)(e / ldx);
  const long long src = idx ? idx[i] : i;
  int o = c;
  float v = 0.f;
  bool done_ = false;
  if (use_state) {
    if (o < obs_dim) { v = obs[src * obs_dim + o]; done_ = true; }
    o -= obs_dim;
  }
  if (!done_ && use_action) {
    if (o >= 0 && o < act_dim) {
      v = act_i64 ? (act_i64[src] == o ? 1.f : 0.f) : act_f32[src * act_dim + o];
      done_ = true;
    }
    o -= act_dim;
  }
  if (!done_ && use_next) {
    if (o >= 0 && o < obs_dim) { v = next_obs[src * obs_dim + o]; done_ = true; }
    o -= obs_dim;
  }
  if (!done_ && use_done) {
    if (o == 0) v = dones[src] ? 1.f : 0.f;
  }
  X[(long long)(row0 + i) * ldx + c] = v;
}

__global__ void gather_rows_kernel(const float* __restrict__ src, const int64_t* __restrict__ idx, int n,
                                   int width, float* __restrict__ dst) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long long)n * width) return;
  const int c = (int)(e % width);
  const long long i = e / width;
  dst[e] = src[idx[i] * width + c];
}

// BCE-with-logits, its gradient and the discriminator statistics. Blocks of 256 threads own 1024
// rows each and write 6 partial sums; the block that draws the last ticket folds the partials in
// block order (deterministic) and resets the ticket. Hand-off per the gfx950 rules: plain stores ->
// __syncthreads -> one-lane agent-scope release (+ explicit vmcnt(0)) -> relaxed ticket; last block:
// one-lane agent-scope acquire -> __syncthreads -> plain loads.
constexpr int BCE_ROWS_PER_BLOCK = 1024;

__global__ __launch_bounds__(256) void bce_kernel(const float* __restrict__ logits, int R, int n_expert, float scale,
                                                  float* __restrict__ dlogits, float* __restrict__ stats,
                                                  float* __restrict__ part /*[nblk][8]*/,
                                                  unsigned int* __restrict__ ticket) {
  __shared__ float red[6][4];
  __shared__ int is_last;
  float vals[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};  // loss, correct, correct_e, correct_g, pred_gen, entropy
  const float inv = scale / (float)R;
  const int r0 = blockIdx.x * BCE_ROWS_PER_BLOCK;
#pragma unroll
  for (int u = 0; u < BCE_ROWS_PER_BLOCK / 256; ++u) {
    const int i = r0 + u * 256 + threadIdx.x;
    if (i < R) {
      const float x = logits[i];
      const float y = i < n_expert ? 1.f : 0.f;
      const float lse = log1pf(expf(-fabsf(x)));
      // (1-y)*x - logsigmoid(x), logsigmoid(x) = min(x,0) - log1p(exp(-|x|))
      vals[0] += (1.f - y) * x - (fminf(x, 0.f) - lse);
      const float p = 1.f / (1.f + expf(-x));
      if (dlogits) dlogits[i] = (p - y) * inv;
      const bool is_gen_pred = x < 0.f, is_gen_true = y == 0.f;
      const bool ok = is_gen_pred == is_gen_true;
      vals[1] += ok ? 1.f : 0.f;
      vals[2] += (ok && !is_gen_true) ? 1.f : 0.f;
      vals[3] += (ok && is_gen_true) ? 1.f : 0.f;
      vals[4] += is_gen_pred ? 1.f : 0.f;
      // Bernoulli(logits=x).entropy() = BCEWithLogits(x, target=sigmoid(x))
      vals[5] += (1.f - p) * x - (fminf(x, 0.f) - lse);
    }
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    float v = vals[k];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if (lane == 0) red[k][wv] = v;
  }
  __syncthreads();
  if (threadIdx.x < 6)
    part[blockIdx.x * 8 + threadIdx.x] = red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] +
                                         red[threadIdx.x][3];
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned int t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    is_last = (t == gridDim.x - 1);
    if (is_last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
  if (!is_last) return;
  if (threadIdx.x < 6) {
    float t = 0.f;
    for (unsigned int b = 0; b < gridDim.x; ++b) t += part[b * 8 + threadIdx.x];  // fixed block order
    if (threadIdx.x == 0) t = t / (float)R * scale;
    stats[threadIdx.x] = t;
  }
  if (threadIdx.x == 6) stats[6] = (float)n_expert;
  if (threadIdx.x == 7) stats[7] = (float)(R - n_expert);
  if (threadIdx.x == 0) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ void airl_logits_kernel(const float* __restrict__ g, const float* __restrict__ h_cur,
                                   const float* __restrict__ h_next, const float* __restrict__ dones,
                                   const float* __restrict__ logp, float gamma, int R,
                                   float* __restrict__ logits) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= R) return;
  // reward_nets.py:727-733 order: base + gamma*((1-done)*new) - old ; then airl.py:118: - logp
  const float new_shaping = (1.f - dones[i]) * h_next[i];
  float f = g[i] + gamma * new_shaping;
  f = f - h_cur[i];
  logits[i] = logp ? f - logp[i] : f;
}

__global__ void airl_route_kernel(const float* __restrict__ dlogits, const float* __restrict__ dones,
                                  float gamma, int R, float* __restrict__ dg, float* __restrict__ dh_cur,
                                  float* __restrict__ dh_next) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= R) return;
  const float d = dlogits[i];
  dg[i] = d;
  dh_cur[i] = -d;
  dh_next[i] = gamma * (1.f - dones[i]) * d;
}

// rewards/reward_nets.py:660-669 applied once per env step t: normalise step t's n rewards with the
// statistics accumulated over steps < t (eval-mode forward), THEN Chan-update them with those n
// raw rewards. One block walks the T steps in order (the update is inherently sequential in t).
__global__ __launch_bounds__(256) void reward_norm_seq_kernel(const float* __restrict__ raw, int T, int n, float eps,
                                                               int update, float* __restrict__ mean,
                                                               float* __restrict__ var, int32_t* __restrict__ count,
                                                               float* __restrict__ out) {
  __shared__ float red[256];
  __shared__ float s_mean, s_var;
  __shared__ int s_cnt;
  const int tid = threadIdx.x;
  if (tid == 0) { s_mean = *mean; s_var = *var; s_cnt = *count; }
  __syncthreads();
  for (int t = 0; t < T; ++t) {
    const float m = s_mean, inv = 1.f / sqrtf(s_var + eps);
    float sum = 0.f;
    for (int i = tid; i < n; i += 256) {
      const float r = raw[(long long)t * n + i];
      out[(long long)t * n + i] = (r - m) * inv;
      sum += r;
    }
    if (!update) continue;
    red[tid] = sum;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
    const float bmean = red[0] / (float)n;
    __syncthreads();
    float q = 0.f;
    for (int i = tid; i < n; i += 256) {
      const float dl = raw[(long long)t * n + i] - bmean;
      q += dl * dl;
    }
    red[tid] = q;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
    if (tid == 0) {
      const float bvar = red[0] / (float)n;
      float mc = s_mean, vc = s_var;
      rn_absorb(mc, vc, s_cnt, n, bmean, bvar);
      s_mean = mc;
      s_var = vc;
      s_cnt = rn_count_add(s_cnt, n);
    }
    __syncthreads();
  }
  if (tid == 0 && update) { *mean = s_mean; *var = s_var; *count = s_cnt; }
}

// Data-parallel form (every rank relabels its own env batch, the statistics must be those of ONE process on
// the env batches side by side). Pass 1, one block per step: (mean, M2) of the step's n raw rewards -- the two-pass
// arithmetic of the kernel above.
__global__ __launch_bounds__(256) void reward_step_moments_kernel(const float* __restrict__ raw, int n,
                                                                   float* __restrict__ mom /*[T][2]*/) {
  __shared__ float red[256];
  const int tid = threadIdx.x, t = blockIdx.x;
  float sum = 0.f;
  for (int i = tid; i < n; i += 256) sum += raw[(long long)t * n + i];
  red[tid] = sum;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
  const float bmean = red[0] / (float)n;
  __syncthreads();
  float q = 0.f;
  for (int i = tid; i < n; i += 256) {
    const float dl = raw[(long long)t * n + i] - bmean;
    q += dl * dl;
  }
  red[tid] = q;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
  if (tid == 0) { mom[2 * t] = bmean; mom[2 * t + 1] = red[0]; }
}

// Pass 2 (after the all-gather of the ranks' [T][2] moments): walk the T steps in order -- normalise this rank's
// rewards of step t with the statistics over steps < t, then absorb the step's batch over ALL ranks (Chan combination of
// the `groups` per-rank moments in rank order: groups * n samples).
__global__ __launch_bounds__(256) void reward_norm_seq_groups_kernel(const float* __restrict__ raw, int T, int n, float eps,
                                                                      int update, const float* __restrict__ mom_all,
                                                                      int groups, float* __restrict__ mean,
                                                                      float* __restrict__ var, int32_t* __restrict__ count,
                                                                      float* __restrict__ out) {
  __shared__ float s_mean, s_var;
  __shared__ int s_cnt;
  const int tid = threadIdx.x;
  if (tid == 0) { s_mean = *mean; s_var = *var; s_cnt = *count; }
  __syncthreads();
  for (int t = 0; t < T; ++t) {
    const float m = s_mean, inv = 1.f / sqrtf(s_var + eps);
    for (int i = tid; i < n; i += 256) out[(long long)t * n + i] = (raw[(long long)t * n + i] - m) * inv;
    __syncthreads();
    if (update && tid == 0) {
      float na = 0.f, ma = 0.f, qa = 0.f;
      for (int g = 0; g < groups; ++g) {
        const float* p = mom_all + ((long long)g * T + t) * 2;
        chan_combine(na, ma, qa, (float)n, p[0], p[1]);
      }
      const int R = groups * n;
      float mc = s_mean, vc = s_var;
      rn_absorb(mc, vc, s_cnt, R, ma, qa / (float)R);
      s_mean = mc;
      s_var = vc;
      s_cnt = rn_count_add(s_cnt, R);
    }
    __syncthreads();
  }
  if (tid == 0 && update) { *mean = s_mean; *var = s_var; *count = s_cnt; }
}

// ---- gradient penalty (opt-in extension; BASELINE.json config 3 names it, the reference has none: SURVEY M1) ----
// x_hat = e * x_expert + (1 - e) * x_gen per row pair, then the input normalisation with FROZEN statistics:
// Xn[r, c] = (x_hat - mean[c]) / sqrt(var[c] + eps)  (mean == nullptr: x_hat itself); columns [D, ld) zero.
__global__ void gp_interpolate_kernel(const float* __restrict__ X, int ldx, int B, int D, const float* __restrict__ e,
                                      const float* __restrict__ mean, const float* __restrict__ var, float eps,
                                      float* __restrict__ Xn, int ld) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)B * ld) return;
  const int c = (int)(i % ld);
  const long long r = i / ld;
  float v = 0.f;
  if (c < D) {
    const float w = e[r];
    v = w * X[r * ldx + c] + (1.f - w) * X[(r + B) * ldx + c];
    if (mean != nullptr) v = (v - mean[c]) / sqrtf(var[c] + eps);
  }
  Xn[i] = v;
}

// Per row: g = gn / sigma (gradient w.r.t. the un-normalised input), n = |g|_2, penalty (n - target)^2,
// Cn = d(coef / B * sum penalty) / d gn = coef / B * 2 (n - target) / n * g / sigma. One wave per row (lanes stride
// the columns); pen[r] = (n - target)^2.
__global__ __launch_bounds__(256) void gp_row_coeffs_kernel(const float* __restrict__ gn, int ld, int B, int D,
                                                            const float* __restrict__ var, float eps, float coef,
                                                            float target, float* __restrict__ Cn, float* __restrict__ pen) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= B) return;
  float sq = 0.f;
  for (int c = lane; c < D; c += 64) {
    const float inv = var ? 1.f / sqrtf(var[c] + eps) : 1.f;
    const float g = gn[(long long)r * ld + c] * inv;
    sq += g * g;
  }
  for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 64);
  const float n = sqrtf(sq);
  const float k = n > 0.f ? coef / (float)B * 2.f * (n - target) / n : 0.f;
  for (int c = lane; c < ld; c += 64) {
    float v = 0.f;
    if (c < D) {
      const float inv = var ? 1.f / sqrtf(var[c] + eps) : 1.f;
      v = k * gn[(long long)r * ld + c] * inv * inv;
    }
    Cn[(long long)r * ld + c] = v;
  }
  if (lane == 0) pen[r] = (n - target) * (n - target);
}

// The same for AIRL's shaped reward f(s, a, s') = g([s | a | s' | d]) + gamma (1 - d) h(s') - h(s) (reward_nets.py:727-733;
// d: the interpolated done flag, a constant of the row): the input gradient T over the blocks [s | a | s' | d] combines
// the three stacks' input gradients, T_s = G_b[s] - G_c, T_a = G_b[a], T_s' = G_b[s'] + c G_n (c = gamma (1 - d)),
// T_d = G_b[d] (base blocks as flagged; G = gn / sigma per stack). n = |T|_2, pen = (n - target)^2 and, with
// Q = coef / B * 2 (n - target) / n * T, the coefficients each stack's second pass starts from:
// Cn_b = Q[block] / sigma_b, Cn_n = c Q_s' / sigma_p, Cn_c = -Q_s / sigma_p. One wave per row.
struct GpShaped {
  const float *gn_b, *gn_n, *gn_c, *dhat, *var_b, *var_p;
  int ldb, ldp, B, od, ad, use_state, use_action, use_next, use_done;
  float eps_b, eps_p, gamma, coef, target;
  float *Cn_b, *Cn_n, *Cn_c, *pen;
};

__global__ __launch_bounds__(256) void gp_shaped_coeffs_kernel(GpShaped a) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= a.B) return;
  const int od = a.od, ad = a.ad;
  const int o_s = 0, o_a = o_s + (a.use_state ? od : 0), o_n = o_a + (a.use_action ? ad : 0), o_d = o_n + (a.use_next ? od : 0);
  const float c = a.gamma * (1.f - a.dhat[r]);
  const float* gb = a.gn_b + (long long)r * a.ldb;
  const float* gnx = a.gn_n + (long long)r * a.ldp;
  const float* gc = a.gn_c + (long long)r * a.ldp;
  auto inv_b = [&](int col) { return a.var_b ? 1.f / sqrtf(a.var_b[col] + a.eps_b) : 1.f; };
  auto inv_p = [&](int j) { return a.var_p ? 1.f / sqrtf(a.var_p[j] + a.eps_p) : 1.f; };
  auto T_s = [&](int j) { return (a.use_state ? gb[o_s + j] * inv_b(o_s + j) : 0.f) - gc[j] * inv_p(j); };
  auto T_a = [&](int j) { return a.use_action ? gb[o_a + j] * inv_b(o_a + j) : 0.f; };
  auto T_n = [&](int j) { return (a.use_next ? gb[o_n + j] * inv_b(o_n + j) : 0.f) + c * (gnx[j] * inv_p(j)); };
  float sq = 0.f;
  for (int j = lane; j < od; j += 64) {
    const float ts = T_s(j), tn = T_n(j);
    sq += ts * ts + tn * tn;
  }
  for (int j = lane; j < ad; j += 64) {
    const float ta = T_a(j);
    sq += ta * ta;
  }
  if (a.use_done && lane == 0) {
    const float td = gb[o_d] * inv_b(o_d);
    sq += td * td;
  }
  for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 64);
  const float n = sqrtf(sq);
  const float k = n > 0.f ? a.coef / (float)a.B * 2.f * (n - a.target) / n : 0.f;
  float* cb = a.Cn_b + (long long)r * a.ldb;
  float* cn = a.Cn_n + (long long)r * a.ldp;
  float* cc = a.Cn_c + (long long)r * a.ldp;
  for (int j = lane; j < a.ldp; j += 64) {
    const bool on = j < od;
    const float qs = on ? k * T_s(j) : 0.f, qn = on ? k * T_n(j) : 0.f;
    cn[j] = on ? c * qn * inv_p(j) : 0.f;
    cc[j] = on ? -qs * inv_p(j) : 0.f;
    if (on && a.use_state) cb[o_s + j] = qs * inv_b(o_s + j);
    if (on && a.use_next) cb[o_n + j] = qn * inv_b(o_n + j);
  }
  if (a.use_action)
    for (int j = lane; j < ad; j += 64) cb[o_a + j] = k * T_a(j) * inv_b(o_a + j);
  const int Db = o_d + (a.use_done ? 1 : 0);
  if (lane == 0) {
    if (a.use_done) cb[o_d] = k * gb[o_d] * inv_b(o_d) * inv_b(o_d);
    for (int col = Db; col < a.ldb; ++col) cb[col] = 0.f;
    a.pen[r] = (n - a.target) * (n - a.target);
  }
}

inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

}  // namespace

int ia_disc_step_fused(const ia_disc_step_args* a, void* stream);  // disc_fused.hip

extern "C" {

int ia_version(void) { return 100; }

int64_t ia_mlp_param_count(const ia_mlp_desc* d) {
  if (!desc_ok(d)) return IA_ERR_ARG;
  LayerOff off[IA_MAX_LAYERS];
  long long tot;
  layer_offsets(d, off, &tot);
  return tot;
}

int64_t ia_mlp_hidden_floats_per_row(const ia_mlp_desc* d) {
  if (!desc_ok(d)) return IA_ERR_ARG;
  long long s = 0;
  for (int l = 1; l < d->n_layers; ++l) s += d->dims[l];
  return s;
}

}  // extern "C" (kernels of the single-output layer follow)

namespace {

// Single-output Linear (the logit / reward / potential head, out_size = 1): a GEMM tile would leave
// 127 of 128 MFMA columns idle and re-launch for an elementwise amount of work, so it runs as a
// streaming row-dot: one wave per row, 16-byte loads, butterfly reduction.
//   out[r] = act(dot(in[r, :K], w) + b)
__global__ __launch_bounds__(256) void rowdot_kernel(const float* __restrict__ in, int ldin, const float* __restrict__ w,
                                                     const float* __restrict__ b, int R, int K, int act,
                                                     float* __restrict__ out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int waves = gridDim.x * 4;
  const bool vec = (K % 4 == 0) && (ldin % 4 == 0) && ((reinterpret_cast<uintptr_t>(in) & 15) == 0) &&
                   ((reinterpret_cast<uintptr_t>(w) & 15) == 0);
  for (int r = blockIdx.x * 4 + wave; r < R; r += waves) {
    const float* row = in + (long long)r * ldin;
    float acc = 0.f;
    if (vec) {
      for (int k = lane * 4; k < K; k += 256) {
        const float4 x = *reinterpret_cast<const float4*>(row + k);
        const float4 ww = *reinterpret_cast<const float4*>(w + k);
        acc += x.x * ww.x + x.y * ww.y + x.z * ww.z + x.w * ww.w;
      }
    } else {
      for (int k = lane; k < K; k += 64) acc += row[k] * w[k];
    }
    for (int s = 32; s > 0; s >>= 1) acc += __shfl_xor(acc, s, 64);
    if (lane == 0) out[r] = ia_apply_act(acc + (b ? b[0] : 0.f), act);
  }
}

// Backward of that layer in ONE pass over the previous activations H[R, N] (post-activation):
//   dIn[r, c]   = dY[r] * w[c] * act'(H[r, c])                      (input gradient, written to dhidden)
//   dW_s[c]     = sum_{r in slab s} dY[r] * H[r, c]                 (weight-gradient partial of K-slab s)
//   db_s        = sum_{r in slab s} dY[r]
// grid = (column chunks of 64, splits); block = 16 column quads x 16 row groups.
__global__ __launch_bounds__(256) void single_out_backward_kernel(const float* __restrict__ dY, const float* __restrict__ H,
                                                                  int ldh, const float* __restrict__ w, int R, int N,
                                                                  int rows_per_split, int act, float* __restrict__ dIn,
                                                                  int ldd, float* __restrict__ dW, float* __restrict__ db,
                                                                  long long split_stride) {
  __shared__ float red[16][65];
  __shared__ float redb[16];
  const int cq = threadIdx.x & 15, rg = threadIdx.x >> 4;
  const int c0 = blockIdx.x * 64 + cq * 4;
  const int split = blockIdx.y;
  const int r0 = split * rows_per_split, r1 = min(R, r0 + rows_per_split);
  const bool vec = (N % 4 == 0) && (ldh % 4 == 0) && (ldd % 4 == 0) && c0 + 3 < N;
  float wv[4], acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 4; ++j) wv[j] = c0 + j < N ? w[c0 + j] : 0.f;
  float bacc = 0.f;
  for (int r = r0 + rg; r < r1; r += 16) {
    const float g = dY[r];
    bacc += g;
    float h[4];
    if (vec) {
      const float4 t = *reinterpret_cast<const float4*>(H + (long long)r * ldh + c0);
      h[0] = t.x; h[1] = t.y; h[2] = t.z; h[3] = t.w;
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) h[j] = c0 + j < N ? H[(long long)r * ldh + c0 + j] : 0.f;
    }
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      acc[j] += g * h[j];
      o[j] = g * wv[j] * ia_act_grad_from_post(h[j], act);
    }
    if (vec) {
      *reinterpret_cast<float4*>(dIn + (long long)r * ldd + c0) = make_float4(o[0], o[1], o[2], o[3]);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (c0 + j < N) dIn[(long long)r * ldd + c0 + j] = o[j];
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) red[rg][cq * 4 + j] = acc[j];
  if (cq == 0) redb[rg] = bacc;
  __syncthreads();
  if (threadIdx.x < 64) {
    const int c = blockIdx.x * 64 + threadIdx.x;
    float t = 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) t += red[g][threadIdx.x];
    if (c < N) dW[(long long)split * split_stride + c] = t;
  }
  if (blockIdx.x == 0 && threadIdx.x == 64 && db != nullptr) {
    float t = 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) t += redb[g];
    db[(long long)split * split_stride] = t;
  }
}

}  // namespace

extern "C" {

int ia_mlp_forward(const ia_mlp_desc* d, const float* params, const float* X, int ldx, int R, float* hidden,
                   float* out, int out_act, void* stream) {
  if (!desc_ok(d) || R <= 0) return IA_ERR_ARG;
  LayerOff off[IA_MAX_LAYERS];
  long long tot;
  layer_offsets(d, off, &tot);
  const float* in = X;
  int ldin = ldx;
  float* hp = hidden;
  for (int l = 0; l < d->n_layers; ++l) {
    const bool last = (l == d->n_layers - 1);
    IaGemm g{};
    g.A = in; g.lda = ldin;
    g.B = params + off[l].w; g.ldb = d->dims[l];
    g.bias = params + off[l].b;
    g.M = R; g.N = d->dims[l + 1]; g.K = d->dims[l];
    g.C = last ? out : hp; g.ldc = d->dims[l + 1];
    g.act = last ? out_act : d->hidden_act;
    int rc = IA_OK;
    if (g.N == 1 && l > 0) {  // single-output head: streaming row-dot instead of a GEMM tile
      const int blocks = R / 4 < 2048 ? (R + 3) / 4 : 2048;
      hipLaunchKernelGGL(rowdot_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, g.A, g.lda, g.B, g.bias, R,
                         g.K, g.act, g.C);
      IA_CHECK_LAUNCH();
    } else {
      rc = ia_launch_gemm(IA_GEMM_NT, g, (hipStream_t)stream);
    }
    if (rc) return rc;
    if (!last) {
      in = hp; ldin = d->dims[l + 1];
      hp += (long long)R * d->dims[l + 1];
    }
  }
  return IA_OK;
}

int ia_mlp_backward(const ia_mlp_desc* d, const float* params, const float* X, int ldx, int R,
                    const float* hidden, const float* dOut, float* dhidden, float* partials, int splits,
                    float* dX, void* stream) {
  if (!desc_ok(d) || R <= 0 || splits < 1) return IA_ERR_ARG;
  LayerOff off[IA_MAX_LAYERS];
  long long tot;
  layer_offsets(d, off, &tot);
  // hidden layer l (output of Linear l, l < n_layers-1) starts at hoff[l]
  long long hoff[IA_MAX_LAYERS];
  long long o = 0;
  for (int l = 0; l + 1 < d->n_layers; ++l) { hoff[l] = o; o += (long long)R * d->dims[l + 1]; }
  const int kps = (((R + splits - 1) / splits) + 31) / 32 * 32;
  for (int l = d->n_layers - 1; l >= 0; --l) {
    const float* dY = (l == d->n_layers - 1) ? dOut : dhidden + hoff[l];
    const float* in = (l == 0) ? X : hidden + hoff[l - 1];
    const int ldin = (l == 0) ? ldx : d->dims[l];
    if (d->dims[l + 1] == 1 && l > 0) {  // single-output head: dW/db partials and dIn in one pass
      const int N = d->dims[l];
      hipLaunchKernelGGL(single_out_backward_kernel, dim3((N + 63) / 64, splits), dim3(256), 0, (hipStream_t)stream, dY,
                         in, ldin, params + off[l].w, R, N, kps, d->hidden_act, dhidden + hoff[l - 1], N,
                         partials + off[l].w, partials + off[l].b, tot);
      IA_CHECK_LAUNCH();
      continue;
    }
    IaGemm w{};  // dW_l[dims(l+1), dims(l)] = dY^T . in   (+ db_l = column sums of dY)
    w.A = dY; w.lda = d->dims[l + 1];
    w.B = in; w.ldb = ldin;
    w.M = d->dims[l + 1]; w.N = d->dims[l]; w.K = R;
    w.C = partials + off[l].w; w.ldc = d->dims[l];
    w.splits = splits; w.k_per_split = kps; w.c_split_stride = tot;
    w.dbias = partials + off[l].b; w.dbias_split_stride = tot;
    int rc = ia_launch_gemm(IA_GEMM_TN, w, (hipStream_t)stream);
    if (rc) return rc;
    if (l > 0 || dX != nullptr) {
      IaGemm g{};  // dIn[R, dims(l)] = dY . W_l  (* act'(in) for hidden inputs)
      g.A = dY; g.lda = d->dims[l + 1];
      g.B = params + off[l].w; g.ldb = d->dims[l];
      g.M = R; g.N = d->dims[l]; g.K = d->dims[l + 1];
      if (l > 0) {
        g.C = dhidden + hoff[l - 1]; g.ldc = d->dims[l];
        g.P = hidden + hoff[l - 1]; g.ldp = d->dims[l];
        g.act = d->hidden_act;
      } else {
        g.C = dX; g.ldc = ldx; g.P = nullptr; g.act = IA_ACT_NONE;
      }
      rc = ia_launch_gemm(IA_GEMM_NN, g, (hipStream_t)stream);
      if (rc) return rc;
    }
  }
  return IA_OK;
}

int ia_reduce_partials(const float* partials, int splits, int64_t n, float scale, int accumulate, float* grads,
                       void* stream) {
  if (n <= 0 || splits < 1) return IA_ERR_ARG;
  hipLaunchKernelGGL(reduce_partials_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, partials,
                     splits, (long long)n, scale, accumulate, grads, (long long)n);
  IA_CHECK_LAUNCH();
  return IA_OK;
}

int ia_reduce_partials_strided(const float* partials, int splits, int64_t n, int64_t stride, float scale, int accumulate,
                               float* grads, void* stream) {
  if (n <= 0 || splits < 1 || stride < n) return IA_ERR_ARG;
  hipLaunchKernelGGL(reduce_partials_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, partials,
                     splits, (long long)n, scale, accumulate, grads, (long long)stride);
  IA_CHECK_LAUNCH();
  return IA_OK;
}

namespace {
// up to 16 slab reductions in ONE launch: segment g's element i = scale-free sum over its `splits[g]` slabs (stride n[g]) in
// slab order, ADDED to dst[g][i] -- `reduce_partials_kernel(accumulate = 1, scale = 1)`'s arithmetic per element
struct ReduceSegs { const float* part[16]; float* dst[16]; long long n[16]; long long start[17]; int splits[16]; int count; };
__global__ void reduce_segments_kernel(ReduceSegs c) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= c.start[c.count]) return;
  int g = 0;
  while (g + 1 < c.count && i >= c.start[g + 1]) ++g;
  const long long j = i - c.start[g], n = c.n[g];
  const float* __restrict__ partials = c.part[g];
  const int splits = c.splits[g];
  float s = 0.f;
  int k = 0;
  for (; k + 8 <= splits; k += 8) {
    float t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) t[u] = partials[(long long)(k + u) * n + j];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += t[u];
  }
  for (; k < splits; ++k) s += partials[(long long)k * n + j];
  s *= 1.0f;   // (the single reduction's `scale`)
  c.dst[g][j] = c.dst[g][j] + s;
}
}  // namespace

int ia_reduce_partials_multi(int n_segs, const float* const* partials, const int* splits, const int64_t* n, float* const* dst,
                             void* stream) {
  if (n_segs < 1 || n_segs > 16) return IA_ERR_ARG;
  ReduceSegs c{};
  long long total = 0;
  for (int g = 0; g < n_segs; ++g) {
    if (n[g] <= 0 || splits[g] < 1) return IA_ERR_ARG;
    c.part[g] = partials[g]; c.dst[g] = dst[g]; c.n[g] = n[g]; c.splits[g] = splits[g];
    c.start[g] = total;
    total += n[g];
  }
  c.start[n_segs] = total;
  c.count = n_segs;
  hipLaunchKernelGGL(reduce_segments_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, c);
  IA_CHECK_LAUNCH();
  return IA_OK;
}

namespace {
struct CopyPieces { const float* src[4]; float* dst[4]; long long n[4]; long long start[5]; };
__global__ void copy_pieces_kernel(CopyPieces c) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= c.start[4]) return;
  const int p = i >= c.start[2] ? (i >= c.start[3] ? 3 : 2) : (i >= c.start[1] ? 1 : 0);
  const long long j = i - c.start[p];
  c.dst[p][j] = c.src[p][j];
}
}  // namespace

int ia_copy_pieces(int n_pieces, const float* const* src, float* const* dst, const int64_t* n, void* stream) {
  if (n_pieces < 1 || n_pieces > 4) return IA_ERR_ARG;
  CopyPieces c{};
  long long total = 0;
  for (int p = 0; p < 4; ++p) {
    c.start[p] = total;
    if (p < n_pieces) {
      if (n[p] < 0) return IA_ERR_ARG;
      c.src[p] = src[p]; c.dst[p] = dst[p]; c.n[p] = n[p];
      total += n[p];
    }
  }
  c.start[4] = total;
  if (total == 0) return IA_OK;
  hipLaunchKernelGGL(copy_pieces_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, c);
  IA_CHECK_LAUNCH();
  return IA_OK;
}

int ia_reduce_partials_adam(const float* partials, int splits, int64_t n, float scale, float* grads, float* params,
                            float* exp_avg, float* exp_avg_sq, float beta1, float beta2, float eps,
                            float weight_decay, float step_size, float bc2_sqrt, void* stream) {
  if (n <= 0 || splits < 1) return IA_ERR_ARG;
  hipLaunchKernelGGL(reduce_adam_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, partials, splits,
                     (long long)n, scale, grads, params, exp_avg, exp_avg_sq, beta1, beta2, eps, weight_decay, step_size,
                     bc2_sqrt, (const float*)nullptr, 0);
  IA_CHECK_LAUNCH();
  return IA_OK;
}

int ia_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float beta1,
                 float beta2, float eps, float weight_decay, float step_size, float bc2_sqrt, void* stream) {
  if (n <= 0) return IA_ERR_ARG;
  hipLaunchKernelGGL(adam_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, params, grads, exp_avg,
                     exp_avg_sq, (long long)n, beta1, beta2, eps, weight_decay, step_size, bc2_sqrt);
  IA_CHECK_LAUNCH();
  return IA_OK;
}

int ia_adam_step_dev(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float beta1,
                     float beta2, float eps, float weight_decay, const float* scalars, void* stream) {
  if (n <= 0 || !scalars) return IA_ERR_ARG;
  hipLaunchKernelGGL(adam_dev_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, params, grads, exp_avg,
                     exp_avg_sq, (long long)n, beta1, beta2, eps, weight_decay, scalars);
  IA_CHECK_LAUNCH();
  return IA_OK;
}

int ia_adam_step_scalars(int64_t* step, double lr, const double* lr_dev, double beta1, double beta2, float* scalars,
                         void* stream) {
  if (!step || !scalars) return IA_ERR_ARG;
  hipLaunchKernelGGL(adam_scalars_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, reinterpret_cast<long long*>(step),
                     lr, lr_dev, beta1, beta2, scalars);
  IA_CHECK_LAUNCH();
  return IA_OK;
}

int64_t ia_running_norm_ws_floats(int R, int D) {
  return (int64_t)cdiv(R, RN_ROWS_PER_BLOCK) * 2 * D;
}

int ia_running_norm_update(const float* X, int ldx, int R, int D, float* mean, float* var, int32_t* count,
                           float* ws, void* stream) {
  if (R <= 0 || D <= 0) return IA_ERR_ARG;
  const int nb = cdiv(R, RN_ROWS_PER_BLOCK);
  hipLaunchKernelGGL(rn_partial_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, X, ldx, R, D, ws);
  IA_CHECK_LAUNCH();
  hipLaunchKernelGGL(rn_merge_kernel, dim3(D), dim3(64), 0, (hipStream_t)stream, ws, nb, nb, R, R, D, D, mean, var,
                     count);
  IA_CHECK_LAUNCH();
  hipLaunchKernelGGL(rn_count_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, count, (long long)R);
  IA_CHECK_LAUNCH();
  return IA_OK;
}

int ia_running_norm_partial(const float* X, int ldx, int R, int D, float* ws, void* stream) {
  if (R <= 0 || D <= 0) return IA_ERR_ARG;
  hipLaunchKernelGGL(rn_partial_kernel, dim3(cdiv(R, RN_ROWS_PER_BLOCK)), dim3(256), 0, (hipStream_t)stream, X, ldx, R,
                     D, ws);
  IA_CHECK_LAUNCH();
  return IA_OK;
}

int ia_running_norm_merge(const float* ws_all, int groups, int rows_per_group, int D, int ws_ld, float* mean,
                          float* var, int32_t* count, void* stream) {
  if (groups <= 0 || rows_per_group <= 0 || D <= 0 || ws_ld < D) return IA_ERR_ARG;
  const int bpg = cdiv(rows_per_group, RN_ROWS_PER_BLOCK);
  hipLaunchKernelGGL(rn_merge_kernel, dim3(D), dim3(64), 0, (hipStream_t)stream, ws_all, groups * bpg, bpg,
                     rows_per_group, groups * rows_per_group, D, ws_ld, mean, var, count);
  IA_CHECK_LAUNCH();
  hipLaunchKernelGGL(rn_count_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, count, (long long)groups * rows_per_group);
  IA_CHECK_LAUNCH();
  return IA_OK;
}

int ia_ema_norm_merge(const float* ws_all, int groups, int rows_per_group, int D, int ws_ld, float* mean, float* var,
                      int32_t* count, float* inv_learning_rate, int32_t* num_batches, float decay, void* stream) {
  if (groups <= 0 || rows_per_group <= 0 || D <= 0 || ws_ld < D || !inv_learning_rate || !num_batches) return IA_ERR_ARG;
  const int bpg = cdiv(rows_per_group, RN_ROWS_PER_BLOCK);
  hipLaunchKernelGGL(ema_merge_kernel, dim3(D), dim3(64), 0, (hipStream_t)stream, ws_all, groups * bpg, bpg, rows_per_group,
                     groups * rows_per_group, D, ws_ld, mean, var, inv_learning_rate, num_batches, decay);
  IA_CHECK_LAUNCH();
  hipLaunchKernelGGL(ema_count_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, count,
                     (long long)groups * rows_per_group, inv_learning_rate, num_batches, decay);
  IA_CHECK_LAUNCH();
  return IA_OK;
}

int ia_running_norm_merge_seq(const float* ws_seq, int n_seq, int64_t seq_stride, int groups, int rows_per_group,
                              int D, int ws_ld, float* mean, float* var, int32_t* count, float* snapshots,
                              void* stream) {
  if (n_seq <= 0 || groups <= 0 || rows_per_group <= 0 || D <= 0 || ws_ld < D) return IA_ERR_ARG;
  const int bpg = cdiv(rows_per_group, RN_ROWS_PER_BLOCK);
  const int rows = groups * rows_per_group;
  hipLaunchKernelGGL(rn_merge_seq_kernel, dim3(D), dim3(64 * RN_SEQ_WAVES), 0, (hipStream_t)stream, ws_seq, n_seq,
                     (long long)seq_stride, groups * bpg, bpg, rows_per_group, rows, D, ws_ld, mean, var, count,
                     snapshots);
  IA_CHECK_LAUNCH();
  hipLaunchKernelGGL(rn_count_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, count, (long long)n_seq * rows);
  IA_CHECK_LAUNCH();
  return IA_OK;
}

int ia_running_norm_apply(const float* X, int ldx, int R, int D, const float* mean, const float* var, float eps,
                          float* Y, int ldy, void* stream) {
  if (R <= 0 || D <= 0 || ldy < D) return IA_ERR_ARG;
  hipLaunchKernelGGL(rn_apply_kernel, dim3(cdiv((long long)R * ldy, 256)), dim3(256), 0, (hipStream_t)stream, X, ldx,
                     R, D, mean, var, eps, Y, ldy);
  IA_CHECK_LAUNCH();
  return IA_OK;
}

int ia_gather_concat(const float* obs, const float* act_f32, const int64_t* act_i64, const float* next_obs,
                     const uint8_t* dones, const int64_t* idx, int n, int obs_dim, int act_dim, int use_state,
                     int use_action, int use_next_state, int use_done, float* X, int ldx, int row0, void* stream) {
  if (n <= 0) return IA_ERR_ARG;
  const int width = (use_state ? obs_dim : 0) + (use_action ? act_dim : 0) + (use_next_state ? obs_dim : 0) +
                    (use_done ? 1 : 0);
  if (width > ldx) return IA_ERR_ARG;
  hipLaunchKernelGGL(gather_concat_kernel, dim3(cdiv((long long)n * ldx, 256)), dim3(256), 0, (hipStream_t)stream,
                     obs, act_f32, act_i64, next_obs, dones, idx, n, obs_dim, act_dim, use_state, use_action,
                     use_next_state, use_done, X, ldx, row0);
  IA_CHECK_LAUNCH();
  return IA_OK;
}

int64_t ia_bce_ws_floats(int R) { return (int64_t)cdiv(R, BCE_ROWS_PER_BLOCK) * 8 + 8; }

int ia_bce_logits(const float* logits, int R, int n_expert, float scale, float* dlogits, float* stats, float* ws,
                  void* stream) {
  if (R <= 0 || n_expert < 0 || n_expert > R || !ws) return IA_ERR_ARG;
  const int nblk = cdiv(R, BCE_ROWS_PER_BLOCK);
  // ws: [nblk][8] partials followed by the ticket word (must be zero before the first call; the
  // kernel leaves it at zero)
  hipLaunchKernelGGL(bce_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, logits, R, n_expert, scale, dlogits,
                     stats, ws, reinterpret_cast<unsigned int*>(ws + (long long)nblk * 8));
  IA_CHECK_LAUNCH();
  return IA_OK;
}

int ia_disc_step_basic(const ia_disc_step_args* a, void* stream) {
  if (!a || !desc_ok(a->desc) || a->n0 < 0 || a->n1 < 0 || a->n0 + a->n1 <= 0) return IA_ERR_ARG;
  // D -> H -> H -> 1 ReLU stacks with a fused workspace take the five-launch path (disc_fused.hip)
  if (a->fused_ws != nullptr && ia_disc_fused_ws_floats(a->desc, a->n0 + a->n1, a->ldx) > 0)
    return ia_disc_step_fused(a, stream);
  if (a->gp_e != nullptr) return IA_ERR_UNSUPPORTED;   // the fused penalty exists on the 128 / 256-wide tile path only
  const int R = a->n0 + a->n1;
  const int D = a->desc->dims[0];
  int rc;
  if (a->n0 > 0 &&
      (rc = ia_gather_concat(a->obs0, a->act0_f32, a->act0_i64, a->next0, a->done0, a->idx0, a->n0, a->obs_dim,
                             a->act_dim, a->use_state, a->use_action, a->use_next_state, a->use_done, a->X, a->ldx, 0,
                             stream)))
    return rc;
  if (a->n1 > 0 &&
      (rc = ia_gather_concat(a->obs1, a->act1_f32, a->act1_i64, a->next1, a->done1, a->idx1, a->n1, a->obs_dim,
                             a->act_dim, a->use_state, a->use_action, a->use_next_state, a->use_done, a->X, a->ldx,
                             a->n0, stream)))
    return rc;
  const float* in = a->X;
  if (a->norm_mean) {
    if (a->update_norm) {
      if ((rc = ia_running_norm_partial(a->X, a->ldx, R, D, a->rn_ws, stream))) return rc;
      if ((rc = ia_running_norm_merge(a->rn_ws, 1, R, D, D, a->norm_mean, a->norm_var, a->norm_count, stream)))
        return rc;
      if (a->pnorm_mean && a->pnorm_dim > 0 && a->pnorm_dim <= D &&
          (rc = ia_running_norm_merge(a->rn_ws, 1, R, a->pnorm_dim, D, a->pnorm_mean, a->pnorm_var, a->pnorm_count,
                                      stream)))
        return rc;
    }
    if ((rc = ia_running_norm_apply(a->X, a->ldx, R, D, a->norm_mean, a->norm_var, a->norm_eps, a->Xn, a->ldx, stream)))
      return rc;
    in = a->Xn;
  }
  if ((rc = ia_mlp_forward(a->desc, a->params, in, a->ldx, R, a->hidden, a->logits, IA_ACT_NONE, stream))) return rc;
  if ((rc = ia_bce_logits(a->logits, R, a->n_expert, a->loss_scale, a->dlogits, a->stats, a->bce_ws, stream))) return rc;
  if ((rc = ia_mlp_backward(a->desc, a->params, in, a->ldx, R, a->hidden, a->dlogits, a->dhidden, a->partials, a->splits,
                            nullptr, stream)))
    return rc;
  const int64_t P = ia_mlp_param_count(a->desc);
  if (a->adam && !a->accumulate)
    return ia_reduce_partials_adam(a->partials, a->splits, P, 1.0f, a->grads, a->params, a->exp_avg, a->exp_avg_sq,
                                   a->beta1, a->beta2, a->adam_eps, a->weight_decay, a->step_size, a->bc2_sqrt, stream);
  if ((rc = ia_reduce_partials(a->partials, a->splits, P, 1.0f, a->accumulate, a->grads, stream))) return rc;
  if (a->adam)
    return ia_adam_step(a->params, a->grads, a->exp_avg, a->exp_avg_sq, P, a->beta1, a->beta2, a->adam_eps,
                        a->weight_decay, a->step_size, a->bc2_sqrt, stream);
  return IA_OK;
}

// A round's updates in ONE host call: update k runs with a[k] (the caller has filled every struct: batch rows, statistics
// snapshots, Adam scalars of step k, statistics row k). The launches are exactly those of n calls of ia_disc_step_basic
// in order; what this entry removes is the per-update host work of the binding above it (the reference's
// `for _ in range(n_disc_updates_per_round): train_disc()`, adversarial/common.py:454-458, costs ~110 us of Python per
// update on the fused path -- more than the update's kernels take to enqueue).
int ia_disc_round_basic(const ia_disc_step_args* a, int n, void* stream) {
  if (!a || n <= 0) return IA_ERR_ARG;
  for (int k = 0; k < n; ++k) {
    const int rc = ia_disc_step_basic(a + k, stream);
    if (rc) return rc;
  }
  return IA_OK;
}

int ia_gp_interpolate(const float* X, int ldx, int B, int D, const float* e, const float* mean, const float* var,
                      float eps, float* Xn, int ld, void* stream) {
  if (!X || !e || !Xn || B <= 0 || D <= 0 || ld < D) return IA_ERR_ARG;
  hipLaunchKernelGGL(gp_interpolate_kernel, dim3(cdiv((long long)B * ld, 256)), dim3(256), 0, (hipStream_t)stream, X,
                     ldx, B, D, e, mean, var, eps, Xn, ld);
  IA_CHECK_LAUNCH();
  return IA_OK;
}

int ia_gp_row_coeffs(const float* gn, int ld, int B, int D, const float* var, float eps, float coef, float target,
                     float* Cn, float* pen, void* stream) {
  if (!gn || !Cn || !pen || B <= 0 || D <= 0 || ld < D) return IA_ERR_ARG;
  hipLaunchKernelGGL(gp_row_coeffs_kernel, dim3(cdiv(B, 4)), dim3(256), 0, (hipStream_t)stream, gn, ld, B, D, var, eps,
                     coef, target, Cn, pen);
  IA_CHECK_LAUNCH();
  return IA_OK;
}

int ia_gp_shaped_coeffs(const float* gn_b, int ldb, const float* gn_n, const float* gn_c, int ldp, const float* dhat, int B,
                        int obs_dim, int act_dim, int use_state, int use_action, int use_next_state, int use_done,
                        const float* var_b, float eps_b, const float* var_p, float eps_p, float gamma, float coef,
                        float target, float* Cn_b, float* Cn_n, float* Cn_c, float* pen, void* stream) {
  const int Db = (use_state ? obs_dim : 0) + (use_action ? act_dim : 0) + (use_next_state ? obs_dim : 0) + (use_done ? 1 : 0);
  if (!gn_b || !gn_n || !gn_c || !dhat || !Cn_b || !Cn_n || !Cn_c || !pen || B <= 0 || obs_dim <= 0 || act_dim <= 0 ||
      Db <= 0 || ldb < Db || ldp < obs_dim)
    return IA_ERR_ARG;
  GpShaped a{gn_b, gn_n, gn_c, dhat, var_b, var_p, ldb, ldp, B, obs_dim, act_dim, use_state, use_action, use_next_state,
             use_done, eps_b, eps_p, gamma, coef, target, Cn_b, Cn_n, Cn_c, pen};
  hipLaunchKernelGGL(gp_shaped_coeffs_kernel, dim3(cdiv(B, 4)), dim3(256), 0, (hipStream_t)stream, a);
  IA_CHECK_LAUNCH();
  return IA_OK;
}

int ia_airl_logits(const float* g, const float* h_cur, const float* h_next, const float* dones,
                   const float* logp, float gamma, int R, float* logits, void* stream) {
  if (R <= 0) return IA_ERR_ARG;
  hipLaunchKernelGGL(airl_logits_kernel, dim3(cdiv(R, 256)), dim3(256), 0, (hipStream_t)stream, g, h_cur, h_next,
                     dones, logp, gamma, R, logits);
  IA_CHECK_LAUNCH();
  return IA_OK;
}

int ia_airl_route_grad(const float* dlogits, const float* dones, float gamma, int R, float* dg, float* dh_cur,
                       float* dh_next, void* stream) {
  if (R <= 0) return IA_ERR_ARG;
  hipLaunchKernelGGL(airl_route_kernel, dim3(cdiv(R, 256)), dim3(256), 0, (hipStream_t)stream, dlogits, dones,
                     gamma, R, dg, dh_cur, dh_next);
  IA_CHECK_LAUNCH();
  return IA_OK;
}

int ia_reward_norm_sequential(const float* raw, int T, int n, float eps, int update_stats, float* mean, float* var,
                              int32_t* count, float* out, void* stream) {
  if (T <= 0 || n <= 0) return IA_ERR_ARG;
  hipLaunchKernelGGL(reward_norm_seq_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, raw, T, n, eps, update_stats,
                     mean, var, count, out);
  IA_CHECK_LAUNCH();
  return IA_OK;
}

int ia_reward_step_moments(const float* raw, int T, int n, float* moments, void* stream) {
  if (T <= 0 || n <= 0 || !raw || !moments) return IA_ERR_ARG;
  hipLaunchKernelGGL(reward_step_moments_kernel, dim3(T), dim3(256), 0, (hipStream_t)stream, raw, n, moments);
  IA_CHECK_LAUNCH();
  return IA_OK;
}

int ia_reward_norm_sequential_groups(const float* raw, int T, int n, float eps, int update_stats, const float* moments_all,
                                     int groups, float* mean, float* var, int32_t* count, float* out, void* stream) {
  if (T <= 0 || n <= 0 || groups <= 0 || !moments_all) return IA_ERR_ARG;
  hipLaunchKernelGGL(reward_norm_seq_groups_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, raw, T, n, eps,
                     update_stats, moments_all, groups, mean, var, count, out);
  IA_CHECK_LAUNCH();
  return IA_OK;
}

int ia_gather_rows(const float* src, const int64_t* idx, int n, int width, float* dst, void* stream) {
  if (n <= 0 || width <= 0) return IA_ERR_ARG;
  hipLaunchKernelGGL(gather_rows_kernel, dim3(cdiv((long long)n * width, 256)), dim3(256), 0, (hipStream_t)stream,
                     src, idx, n, width, dst);
  IA_CHECK_LAUNCH();
  return IA_OK;
}

}  // extern "C"

// reduce (first slabs, scaled) + reduce (second slabs, added) + Adam in one launch (internal: airl_fused.hip)
int ia_reduce2_partials_adam(const float* partials, int splits, const float* partials2, int splits2, int64_t n, float scale,
                             float* grads, float* params, float* exp_avg, float* exp_avg_sq, float beta1, float beta2,
                             float eps, float weight_decay, float step_size, float bc2_sqrt, hipStream_t stream) {
  if (n <= 0 || splits < 1 || splits2 < 1 || !partials || !partials2) return IA_ERR_ARG;
  hipLaunchKernelGGL(reduce_adam_kernel, dim3(cdiv(n, 256)), dim3(256), 0, stream, partials, splits, (long long)n, scale,
                     grads, params, exp_avg, exp_avg_sq, beta1, beta2, eps, weight_decay, step_size, bc2_sqrt, partials2,
                     splits2);
  IA_CHECK_LAUNCH();
  return IA_OK;
}
