// Dense-stack (discriminator / reward-net) engine on top of the fp32 MFMA GEMM, plus the
// HBM-bound helpers of the discriminator update: RunningNorm (Chan merge), gather+concat
// batch assembly, BCE-with-logits + train statistics, split-K partial reduction, Adam.
#include "common.h"
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
                                       int accumulate, float* __restrict__ grads) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  int k = 0;
  for (; k + 8 <= splits; k += 8) {  // 8 independent loads in flight, then a fixed-order sum
    float t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) t[u] = partials[(long long)(k + u) * n + i];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += t[u];
  }
  for (; k < splits; ++k) s += partials[(long long)k * n + i];  // fixed order: deterministic
  s *= scale;
  grads[i] = accumulate ? grads[i] + s : s;
}

__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, long long n, float beta1, float beta2, float eps, float wd,
                            float step_size, float bc2_sqrt) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float grad = g[i];
  const float pi = p[i];
  if (wd != 0.f) grad = grad + wd * pi;
  // torch/optim/adam.py _single_tensor_adam: lerp, mul+addcmul, sqrt/bc2_sqrt + eps, addcdiv
  float mi = m[i];
  mi = mi + (grad - mi) * (1.f - beta1);
  float vi = v[i] * beta2 + (1.f - beta2) * grad * grad;
  const float denom = sqrtf(vi) / bc2_sqrt + eps;
  p[i] = pi - step_size * (mi / denom);
  m[i] = mi;
  v[i] = vi;
}

// RunningNorm statistics. Stage 1: each block owns a contiguous slab of rows and produces, per
// column, (mean_b, M2_b) by a two-pass over its slab (rows are re-read from L2). Stage 2: one
// block Chan-merges the slabs in slab order and applies the reference's update formula.
constexpr int RN_ROWS_PER_BLOCK = 256;

__global__ __launch_bounds__(256) void rn_partial_kernel(const float* __restrict__ X, int ldx, int R, int D,
                                                         float* __restrict__ ws) {
  // threads: 256 = 8 row-lanes x 32 column-lanes; loops over columns in steps of 32
  __shared__ float red[8][33];
  const int r0 = blockIdx.x * RN_ROWS_PER_BLOCK;
  const int rows = min(RN_ROWS_PER_BLOCK, R - r0);
  const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
  for (int c0 = 0; c0 < D; c0 += 32) {
    const int c = c0 + cl;
    float s = 0.f;
    if (c < D)
      for (int r = rl; r < rows; r += 8) s += X[(long long)(r0 + r) * ldx + c];
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
    if (c < D)
      for (int r = rl; r < rows; r += 8) {
        const float dlt = X[(long long)(r0 + r) * ldx + c] - mean;
        q += dlt * dlt;
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
// R = total rows = groups * rpg.
__global__ void rn_merge_kernel(const float* __restrict__ ws, int nblocks, int bpg, int rpg, int R, int D,
                                float* __restrict__ mean, float* __restrict__ var, int32_t* __restrict__ count) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  const int cnt = *count;
  if (c < D) {
    // Chan merge of slab moments -> batch mean / biased variance
    float n_acc = 0.f, m_acc = 0.f, M2 = 0.f;
    for (int b0 = 0; b0 < nblocks; b0 += 8) {  // slab moments are fetched 8 at a time (independent loads)
      float mbv[8], qbv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int b = min(b0 + u, nblocks - 1);
        mbv[u] = ws[((long long)b * 2 + 0) * D + c];
        qbv[u] = ws[((long long)b * 2 + 1) * D + c];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int b = b0 + u;
        if (b < nblocks) {
          const float nb = (float)min(RN_ROWS_PER_BLOCK, rpg - (b % bpg) * RN_ROWS_PER_BLOCK);
          const float tot = n_acc + nb;
          const float dlt = mbv[u] - m_acc;
          M2 = M2 + qbv[u] + dlt * dlt * n_acc * nb / tot;
          m_acc = m_acc + dlt * nb / tot;
          n_acc = tot;
        }
      }
    }
    const float b_mean = m_acc, b_var = M2 / (float)R;
    // util/networks.py:123-134, same operation order
    const float fcount = (float)cnt, fn = (float)R;
    const float tot = (float)(cnt + R);
    const float delta = b_mean - mean[c];
    mean[c] = mean[c] + delta * fn / tot;
    float rv = var[c] * fcount;
    rv = rv + b_var * fn;
    rv = rv + delta * delta * fcount * fn / tot;
    var[c] = rv / tot;
  }
  __syncthreads();
  if (blockIdx.x == 0 && threadIdx.x == 0) *count = cnt + R;
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

// BCE-with-logits, its gradient and the discriminator statistics in one pass by one block.
__global__ __launch_bounds__(1024) void bce_kernel(const float* __restrict__ logits, int R, int n_expert,
                                                   float scale, float* __restrict__ dlogits,
                                                   float* __restrict__ stats) {
  __shared__ float red[6][16];
  float loss = 0.f, correct = 0.f, correct_e = 0.f, correct_g = 0.f, pred_gen = 0.f, ent = 0.f;
  const float inv = scale / (float)R;
  for (int i = threadIdx.x; i < R; i += blockDim.x) {
    const float x = logits[i];
    const float y = i < n_expert ? 1.f : 0.f;
    const float lse = log1pf(expf(-fabsf(x)));
    // (1-y)*x - logsigmoid(x), logsigmoid(x) = min(x,0) - log1p(exp(-|x|))
    loss += (1.f - y) * x - (fminf(x, 0.f) - lse);
    const float p = 1.f / (1.f + expf(-x));
    if (dlogits) dlogits[i] = (p - y) * inv;
    const bool is_gen_pred = x < 0.f;
    const bool is_gen_true = y == 0.f;
    const bool ok = is_gen_pred == is_gen_true;
    correct += ok ? 1.f : 0.f;
    correct_e += (ok && !is_gen_true) ? 1.f : 0.f;
    correct_g += (ok && is_gen_true) ? 1.f : 0.f;
    pred_gen += is_gen_pred ? 1.f : 0.f;
    // Bernoulli(logits=x).entropy() = BCEWithLogits(x, target=sigmoid(x))
    ent += (1.f - p) * x - (fminf(x, 0.f) - lse);
  }
  float vals[6] = {loss, correct, correct_e, correct_g, pred_gen, ent};
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    float v = vals[k];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if (lane == 0) red[k][wv] = v;
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    float t = 0.f;
    const int nw = blockDim.x >> 6;
    for (int w = 0; w < nw; ++w) t += red[threadIdx.x][w];
    if (threadIdx.x == 0) t = t / (float)R * scale;
    stats[threadIdx.x] = t;
  }
  if (threadIdx.x == 6) stats[6] = (float)n_expert;
  if (threadIdx.x == 7) stats[7] = (float)(R - n_expert);
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
      const float fcount = (float)s_cnt, fn = (float)n, tot = (float)(s_cnt + n);
      const float delta = bmean - s_mean;
      s_mean = s_mean + delta * fn / tot;
      float rv = s_var * fcount;
      rv = rv + bvar * fn;
      rv = rv + delta * delta * fcount * fn / tot;
      s_var = rv / tot;
      s_cnt += n;
    }
    __syncthreads();
  }
  if (tid == 0 && update) { *mean = s_mean; *var = s_var; *count = s_cnt; }
}

inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

}  // namespace

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
    int rc = ia_launch_gemm(IA_GEMM_NT, g, (hipStream_t)stream);
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
                     splits, (long long)n, scale, accumulate, grads);
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

int64_t ia_running_norm_ws_floats(int R, int D) {
  return (int64_t)cdiv(R, RN_ROWS_PER_BLOCK) * 2 * D;
}

int ia_running_norm_update(const float* X, int ldx, int R, int D, float* mean, float* var, int32_t* count,
                           float* ws, void* stream) {
  if (R <= 0 || D <= 0) return IA_ERR_ARG;
  const int nb = cdiv(R, RN_ROWS_PER_BLOCK);
  hipLaunchKernelGGL(rn_partial_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, X, ldx, R, D, ws);
  IA_CHECK_LAUNCH();
  hipLaunchKernelGGL(rn_merge_kernel, dim3(cdiv(D, 64)), dim3(64), 0, (hipStream_t)stream, ws, nb, nb, R, R, D, mean,
                     var, count);
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

int ia_running_norm_merge(const float* ws_all, int groups, int rows_per_group, int D, float* mean, float* var,
                          int32_t* count, void* stream) {
  if (groups <= 0 || rows_per_group <= 0 || D <= 0) return IA_ERR_ARG;
  const int bpg = cdiv(rows_per_group, RN_ROWS_PER_BLOCK);
  hipLaunchKernelGGL(rn_merge_kernel, dim3(cdiv(D, 64)), dim3(64), 0, (hipStream_t)stream, ws_all, groups * bpg, bpg,
                     rows_per_group, groups * rows_per_group, D, mean, var, count);
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

int ia_bce_logits(const float* logits, int R, int n_expert, float scale, float* dlogits, float* stats,
                  void* stream) {
  if (R <= 0 || n_expert < 0 || n_expert > R) return IA_ERR_ARG;
  hipLaunchKernelGGL(bce_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, logits, R, n_expert, scale, dlogits,
                     stats);
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

int ia_gather_rows(const float* src, const int64_t* idx, int n, int width, float* dst, void* stream) {
  if (n <= 0 || width <= 0) return IA_ERR_ARG;
  hipLaunchKernelGGL(gather_rows_kernel, dim3(cdiv((long long)n * width, 256)), dim3(256), 0, (hipStream_t)stream,
                     src, idx, n, width, dst);
  IA_CHECK_LAUNCH();
  return IA_OK;
}

}  // extern "C"
