// Generator-side kernels of the GAIL/AIRL round for gfx950: batched policy inference over the
// VecEnv observation tensor (SB3 ActorCriticPolicy.forward / evaluate_actions / predict_values),
// GAE, and the PPO minibatch update (gather -> forward -> clipped-surrogate/value/entropy loss
// -> backward -> global-norm clip -> Adam).
//
// Shapes are tiny (obs<=64, hidden 32/64, act<=16; 3.5k-10k parameters) so the design target is
// launch count and latency, not FLOPs:
//  * one lane owns one row; the two tanh towers run out of VGPR accumulators with the weights
//    broadcast from scalar loads (weights are wave-uniform: transposed copies [in][out] are kept
//    next to the torch-layout parameters so a whole output row is one s_load_dwordx16);
//  * activations live in per-row LDS lines with an odd stride (bank-conflict free both for the
//    owner lane and for the MFMA fragment reads below);
//  * weight gradients are the only cross-row contractions: they are taken straight from those LDS
//    tiles with v_mfma_f32_32x32x2_f32 (A[i][k=row]=dZ, B[k=row][j]=activation), 64 rows per wave;
//  * per-wave partial gradients go to a slab; a single-block kernel reduces the slabs in fixed
//    order (deterministic), applies clip_grad_norm_ + Adam and refreshes the transposed copies.
#include "common.h"
#include "rn_common.h"
#include <algorithm>
#include <type_traits>
#include <cstring>

#include "../../include/imitation_hip.h"

namespace {

constexpr int MAXD = 64;   // max observation width
constexpr int MAXA = 16;   // max action width / number of discrete actions
constexpr int ROWS = 64;   // rows per block (one wave)
constexpr float LOG_SQRT_2PI = 0.9189385332046727f;  // math.log(math.sqrt(2*math.pi))

struct PolOff {
  int log_std, pW1, pb1, pW2, pb2, vW1, vb1, vW2, vb2, aW, ab, cW, cb, total;
};

__host__ __device__ inline PolOff pol_offsets(int D, int A, int H, int discrete) {
  PolOff o;
  int p = 0;
  o.log_std = discrete ? -1 : 0;
  if (!discrete) p += A;
  o.pW1 = p; p += H * D;
  o.pb1 = p; p += H;
  o.pW2 = p; p += H * H;
  o.pb2 = p; p += H;
  o.vW1 = p; p += H * D;
  o.vb1 = p; p += H;
  o.vW2 = p; p += H * H;
  o.vb2 = p; p += H;
  o.aW = p; p += A * H;
  o.ab = p; p += A;
  o.cW = p; p += H;
  o.cb = p; p += 1;
  o.total = p;
  return o;
}

inline bool pol_ok(const ia_policy_desc* d) {
  return d && d->obs_dim >= 1 && d->obs_dim <= MAXD && d->act_dim >= 1 && d->act_dim <= MAXA &&
         (d->hidden == 32 || d->hidden == 64);
}

// Per-block LDS carve-up (floats). Strides are odd => lane r touching column c hits bank (r+c)%32.
template <int H>
struct Lds {
  static constexpr int XS = MAXD + 1, HS = H + 1, AS = MAXA + 1;
  static constexpr int x = 0;
  static constexpr int a1 = x + ROWS * XS;
  static constexpr int a2 = a1 + ROWS * HS;
  static constexpr int dz = a2 + ROWS * HS;
  static constexpr int out = dz + ROWS * HS;
  static constexpr int dout = out + ROWS * AS;
  static constexpr int aux = dout + ROWS * AS;
  static constexpr int total = aux + ROWS * AS + 64;  // +64: MFMA fragment reads may overrun a 17-wide tile
};

// Branch-free tanh: odd polynomial for |x| <= 0.1 (rel. error < 1e-9), 1 - 2/(exp(2|x|)+1) otherwise
// (v_exp_f32 / v_rcp_f32: ~1 ulp each). libm's tanhf costs ~45 instructions and divergent branches.
// Written with explicit fused multiply-adds (the file is built with -ffp-contract=off: as plain expressions the
// polynomial was 3 mul + 3 add, the exponent (|x| + |x|) * log2(e) an add + a mul, 1 - 2 r a mul + a sub -- 16 VALU
// instructions + the two transcendentals per value, 8 values per lane and layer on the PPO chain): 11 + 2. The exponent
// and 1 - 2 r are the same bits as before (scaling by two is exact); the polynomial rounds three times less.
#ifndef IA_TANH_FMA
#define IA_TANH_FMA 1
#endif
__device__ __forceinline__ float fast_tanh(float x) {
  const float ax = fabsf(x);
  const float x2 = x * x;
#if IA_TANH_FMA
  const float poly = x * __builtin_fmaf(x2, __builtin_fmaf(x2, __builtin_fmaf(x2, -0.05396825f, 0.13333334f), -0.33333334f), 1.f);
  const float e = __builtin_amdgcn_exp2f(ax * 2.8853900817779268f);   // = exp2((|x| + |x|) * log2(e)), bit for bit
  const float big = copysignf(__builtin_fmaf(-2.f, __builtin_amdgcn_rcpf(e + 1.f), 1.f), x);
#else
  const float poly = x * (1.f + x2 * (-0.33333334f + x2 * (0.13333334f + x2 * -0.05396825f)));
  const float e = __expf(2.f * ax);
  const float big = copysignf(1.f - 2.f * __builtin_amdgcn_rcpf(e + 1.f), x);
#endif
  return ax <= 0.1f ? poly : big;
}

// One tanh tower: a1 = tanh(W1 x + b1) -> LDS line, a2 = tanh(W2 a1 + b2) -> registers (+LDS).
template <int H>
__device__ __forceinline__ void tower_forward(const float* __restrict__ W1t, const float* __restrict__ b1,
                                              const float* __restrict__ W2t, const float* __restrict__ b2, int D,
                                              const float* xrow, float* a1row, float* a2row, float (&a2)[H]) {
  float acc[H];
#pragma unroll
  for (int j = 0; j < H; ++j) acc[j] = b1[j];
  for (int k = 0; k < D; ++k) {
    const float xk = xrow[k];
#pragma unroll
    for (int j = 0; j < H; ++j) acc[j] = fmaf(W1t[k * H + j], xk, acc[j]);
  }
#pragma unroll
  for (int j = 0; j < H; ++j) a1row[j] = fast_tanh(acc[j]);
#pragma unroll
  for (int j = 0; j < H; ++j) acc[j] = b2[j];
  for (int k = 0; k < H; ++k) {
    const float ak = a1row[k];
#pragma unroll
    for (int j = 0; j < H; ++j) acc[j] = fmaf(W2t[k * H + j], ak, acc[j]);
  }
#pragma unroll
  for (int j = 0; j < H; ++j) {
    a2[j] = fast_tanh(acc[j]);
    if (a2row) a2row[j] = a2[j];
  }
}

// out[a] = b[a] + W[a,:] . a2   for a < A (W in torch layout [A][H]); results to an LDS line.
template <int H>
__device__ __forceinline__ void head_forward(const float* __restrict__ W, const float* __restrict__ b, int A,
                                             const float (&a2)[H], float* outrow) {
  for (int a = 0; a < A; ++a) {
    float s = b[a];
#pragma unroll
    for (int k = 0; k < H; ++k) s = fmaf(W[a * H + k], a2[k], s);
    outrow[a] = s;
  }
}

__device__ __forceinline__ void load_features(const ia_policy_desc& d, const float* __restrict__ obs_row,
                                              const float* __restrict__ nm, const float* __restrict__ nv,
                                              bool valid, float* xrow) {
  for (int k = 0; k < d.obs_dim; ++k) {
    float v = valid ? obs_row[k] : 0.f;
    if (d.has_norm) v = (v - nm[k]) / sqrtf(nv[k] + d.norm_eps);  // util/networks.py:91
    xrow[k] = valid ? v : 0.f;
  }
}

// Diagonal-Gaussian log-prob / entropy exactly as torch.distributions.Normal composes them.
__device__ __forceinline__ float gauss_logp_term(float a, float mu, float log_std) {
  const float sd = expf(log_std);
  const float var = sd * sd;
  const float diff = a - mu;
  return -(diff * diff) / (2.f * var) - logf(sd) - LOG_SQRT_2PI;
}

// ------------------------------------------------------------------------------- inference

template <int H>
__global__ __launch_bounds__(ROWS) void policy_eval_kernel(ia_policy_desc d, const float* __restrict__ P,
                                                           const float* __restrict__ Pt, const float* __restrict__ nm,
                                                           const float* __restrict__ nv, const float* __restrict__ obs,
                                                           const float* __restrict__ act_in, int n,
                                                           float* __restrict__ logp, float* __restrict__ values,
                                                           float* __restrict__ entropy) {
  using L = Lds<H>;
  extern __shared__ float lds[];
  const int tid = threadIdx.x, row = blockIdx.x * ROWS + tid;
  const bool valid = row < n;
  const int D = d.obs_dim, A = d.act_dim;
  const PolOff o = pol_offsets(D, A, H, d.discrete);
  float* xrow = lds + L::x + tid * L::XS;
  float* a1row = lds + L::a1 + tid * L::HS;
  float* outrow = lds + L::out + tid * L::AS;
  load_features(d, obs + (long long)(valid ? row : 0) * D, nm, nv, valid, xrow);
  float a2[H];
  if (logp || entropy) {
    tower_forward<H>(Pt + o.pW1, P + o.pb1, Pt + o.pW2, P + o.pb2, D, xrow, a1row, nullptr, a2);
    head_forward<H>(P + o.aW, P + o.ab, A, a2, outrow);
  }
  if (values) {
    tower_forward<H>(Pt + o.vW1, P + o.vb1, Pt + o.vW2, P + o.vb2, D, xrow, a1row, nullptr, a2);
    float v = P[o.cb];
#pragma unroll
    for (int k = 0; k < H; ++k) v = fmaf(P[o.cW + k], a2[k], v);
    if (valid) values[row] = v;
  }
  if (!valid || !(logp || entropy)) return;
  if (!d.discrete) {
    float lp = 0.f, en = 0.f;
    for (int a = 0; a < A; ++a) {
      const float ls = P[o.log_std + a];
      if (act_in) lp += gauss_logp_term(act_in[(long long)row * A + a], outrow[a], ls);
      en += 0.5f + LOG_SQRT_2PI + logf(expf(ls));  // Normal.entropy: 0.5 + 0.5*log(2*pi) + log(scale)
    }
    if (logp) logp[row] = lp;
    if (entropy) entropy[row] = en;
  } else {
    float mx = outrow[0];
    for (int a = 1; a < A; ++a) mx = fmaxf(mx, outrow[a]);
    float se = 0.f;
    for (int a = 0; a < A; ++a) se += expf(outrow[a] - mx);
    const float lse = mx + logf(se);
    float en = 0.f;
    for (int a = 0; a < A; ++a) {
      const float l = outrow[a] - lse;
      en -= expf(l) * l;
    }
    if (logp && act_in) logp[row] = outrow[(int)act_in[row]] - lse;
    if (entropy) entropy[row] = en;
  }
}

// Raw head outputs: action_net(latent_pi) (Categorical logits / Gaussian means) and the value head, for
// callers that sample on the host with the reference's own RNG call ([SB3 CategoricalDistribution.sample] =
// torch.multinomial on torch's global CPU generator). Thread per row.
// Host <-> resident-kernel hand-off of the rollout mailboxes (policy_rollout_mailbox_kernel, policy_logits_mailbox_kernel):
// `ready` is one int in pinned host memory (step t may run once it exceeds t; negative: abort), `done[workgroup]` the
// acknowledgement. mailbox_wait: lane 0 polls with system-scope loads (bounded by `timeout_ticks` of the 100 MHz clock),
// the verdict goes round the workgroup, and nothing of the step is read ahead of a system-scope acquire. mailbox_ack:
// every thread has drained its stores, block barrier, one lane's system-scope release (the step's outputs in host memory
// are ordinary L2-cached stores) and flag store.
__device__ __forceinline__ bool mailbox_wait(const int* ready, int t, long long timeout_ticks, int* s_go) {
  if (threadIdx.x == 0) {
    const long long t0 = wall_clock64();
    int go = 0;
    for (;;) {
      const int v = __hip_atomic_load(ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      if (v > t) { go = 1; break; }
      if (v < 0 || wall_clock64() - t0 > timeout_ticks) break;
      __builtin_amdgcn_s_sleep(8);
    }
    *s_go = go;
  }
  __syncthreads();
  const bool go = *s_go != 0;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
  return go;
}
__device__ __forceinline__ void mailbox_ack(int* done, int value) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    __hip_atomic_store(done + blockIdx.x, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// The host-sampled Discrete rollout step (`ia_policy_logits` per step + torch.multinomial on the host) with the launches
// folded into one resident kernel: step t's logits land in the SAME pinned [n, A] tile every step (the host has sampled
// from it before it posts the next step), the values in the device tile.
struct LogitsMailbox {
  const float* obs; long long s_obs;
  float* logits;
  float* values; long long s_val;
  int T; const int* ready; int* done; long long timeout_ticks;
};

__global__ void transpose_params_kernel(ia_policy_desc d, const float* __restrict__ P, float* __restrict__ Pt) {
  const int H = d.hidden, D = d.obs_dim;
  const PolOff o = pol_offsets(D, d.act_dim, H, d.discrete);
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < o.total; i += gridDim.x * blockDim.x) {
    float v = P[i];
    int dst = i;
    auto tr = [&](int base, int rows, int cols) {  // [rows][cols] -> [cols][rows]
      if (i >= base && i < base + rows * cols) {
        const int r = (i - base) / cols, c = (i - base) % cols;
        dst = base + c * rows + r;
      }
    };
    tr(o.pW1, H, D); tr(o.pW2, H, H); tr(o.vW1, H, D); tr(o.vW2, H, H);
    Pt[dst] = v;
  }
}

// ---------------------------------------------------------------------------------- GAE

__global__ void gae_kernel(const float* __restrict__ rewards, const float* __restrict__ values,
                           const float* __restrict__ starts, const float* __restrict__ last_values,
                           const float* __restrict__ last_dones, int T, int n, float gamma, float gl,
                           float* __restrict__ adv, float* __restrict__ ret) {
  // one lane per environment, reverse scan over T; every operation rounded separately in the
  // order NumPy evaluates `r + gamma*nv*nnt - v` and `delta + gamma*lambda*nnt*last` (bit-exact: the recurrence stays
  // sequential in t). The loads do not depend on the recurrence, so they are requested GAE_U time steps at a time --
  // the scan was one memory round trip per time step (472 us for 1 000 steps x 1 024 environments, 43 GB/s).
  constexpr int GAE_U = 16;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  float last = 0.f;
  float nv = last_values[e];                       // V of the step behind t (t = T - 1: the bootstrap value)
  float nnt = __fsub_rn(1.0f, last_dones[e]);      // 1 - start flag of the step behind t
  for (int t1 = T - 1; t1 >= 0; t1 -= GAE_U) {     // steps t1, t1 - 1, ..., t1 - GAE_U + 1 (clamped at 0)
    float r[GAE_U], v[GAE_U], st[GAE_U];
#pragma unroll
    for (int u = 0; u < GAE_U; ++u) {
      const long long o = (long long)max(t1 - u, 0) * n + e;   // clamped: unconditional loads, all in flight together
      r[u] = rewards[o];
      v[u] = values[o];
      st[u] = starts[o];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < GAE_U; ++u) {
      const int t = t1 - u;
      if (t >= 0) {
        const float delta = __fsub_rn(__fadd_rn(r[u], __fmul_rn(__fmul_rn(gamma, nv), nnt)), v[u]);
        last = __fadd_rn(delta, __fmul_rn(__fmul_rn(gl, nnt), last));
        adv[(long long)t * n + e] = last;
        ret[(long long)t * n + e] = __fadd_rn(last, v[u]);
        nv = v[u];                               // step t - 1 looks at values[t] and starts[t]
        nnt = __fsub_rn(1.0f, st[u]);
      }
    }
  }
}

// rewards[i] += gamma * V(terminal_obs_i) where the episode ended by time limit
__global__ void bootstrap_kernel(float* __restrict__ rewards, const float* __restrict__ term_values,
                                 const uint8_t* __restrict__ truncated, float gamma, long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (truncated[i]) rewards[i] = __fadd_rn(rewards[i], __fmul_rn(gamma, term_values[i]));
}

// --------------------------------------------------------------------------- PPO minibatch
//
// Two launches per minibatch:
//   ppo_grad_kernel   : 64 rows per block, 8 waves: waves 0-3 run the policy tower, waves 4-7 the
//                       value tower, each wave owning a quarter of every layer's outputs (so the
//                       serial chain per wave is 4x shorter and each SIMD holds two waves);
//                       weight gradients come from MFMA over the LDS activation tiles.
//   ppo_apply_kernel  : one 1024-thread block: fixed-order slab reduction, clip_grad_norm_, Adam,
//                       transposed-copy refresh, THEN the statistics of the NEXT minibatch
//                       (advantage mean/std, feature RunningNorm update), so no separate
//                       "prepare" launch is needed except for the first minibatch of an epoch.

__host__ __device__ inline long long epoch_ll_words(int nrb, int P);   // (ppo_epoch_ll_kernel's word areas)
__host__ __device__ inline int epoch_ll_row_blocks(int nrb, int P);
// ws layout (floats): [0..7] adv stats {mean, std}; [8..8+nblk*8) loss-stat partials;
// then gradient slabs [nblk][P]; then reduced gradient [P].
struct PpoWs {
  float* advstat;
  float* statpart;
  float* slabs;
  float* grad;
};
__host__ __device__ inline PpoWs ppo_ws(float* ws, int nblk, int P) {
  PpoWs w;
  w.advstat = ws;
  w.statpart = ws + 8;
  w.slabs = w.statpart + (long long)nblk * 8;
  w.grad = w.slabs + (long long)nblk * P;
  return w;
}

__device__ __forceinline__ long long rollout_offset(long long flat, int T, int n_envs) {
  // SB3 swap_and_flatten: flat = env*T + t  ->  time-major storage offset t*n_envs + env
  const long long env = flat / T, t = flat % T;
  return t * n_envs + env;
}
// Row i of a minibatch: through the permutation (idx != null) or already gathered (contiguous).
__device__ __forceinline__ long long mb_row(const int64_t* __restrict__ idx, long long i, int T, int n_envs) {
  return idx ? rollout_offset(idx[i], T, n_envs) : i;
}

// One coalesced pass per epoch: rows of the rollout tile in permuted order -> contiguous arrays,
// so every minibatch kernel afterwards streams its rows without dependent index loads.
__global__ void ppo_epoch_gather_kernel(const float* __restrict__ obs, const float* __restrict__ act,
                                        const float* __restrict__ logp, const float* __restrict__ adv,
                                        const float* __restrict__ ret, const int64_t* __restrict__ perm,
                                        long long total, int T, int n_envs, int D, int aw, float* __restrict__ gobs,
                                        float* __restrict__ gact, float* __restrict__ glogp, float* __restrict__ gadv,
                                        float* __restrict__ gret, unsigned long long* __restrict__ zero_words,
                                        long long n_zero, unsigned* __restrict__ zero_ctl) {
  const int W = D + aw + 3;
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  // (the word-exchange epoch kernel's areas and control words are cleared here instead of by three fill launches per epoch)
  for (long long z = e; z < n_zero; z += (long long)gridDim.x * blockDim.x) zero_words[z] = 0ull;
  if (zero_ctl != nullptr && e < 3) zero_ctl[e] = 0u;
  if (e >= total * W) return;
  const long long p = e / W;
  const int c = (int)(e - p * W);
  const long long src = rollout_offset(perm[p], T, n_envs);
  if (c < D) gobs[p * D + c] = obs[src * D + c];
  else if (c < D + aw) gact[p * aw + (c - D)] = act[src * aw + (c - D)];
  else if (c == D + aw) glogp[p] = logp[src];
  else if (c == D + aw + 1) gadv[p] = adv[src];
  else gret[p] = ret[src];
}

constexpr int PREP_THREADS = 1024;
constexpr int PREP_STAGE_FLOATS = 24576;            // 96 KB of staged observation rows
constexpr int PREP_LDS_FLOATS = PREP_STAGE_FLOATS + 16 * 65 + 65 + 64;

template <int NT>
__device__ __forceinline__ float block_sum(float v, float* red /*>=17 floats*/) {
  for (int s = 32; s > 0; s >>= 1) v += __shfl_down(v, s, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = 0.f;  // every thread adds the wave sums in the same order: same value everywhere, no third barrier
#pragma unroll
  for (int k = 0; k < NT / 64; ++k) t += red[k];
  return t;
}
__device__ __forceinline__ float block_sum_1024(float v, float* red) { return block_sum<1024>(v, red); }

// Wave sum on the DPP path (cross-lane operands of the VALU itself: ~10 clocks a step) instead of six ds_bpermute round
// trips through the LDS pipe: quad swaps, row rotations, then row_bcast:15 / row_bcast:31 (GFX9). Lanes 48..63 end
// with the total. Fixed order: deterministic, identical in every workgroup.
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float dpp_take(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_sum_dpp(float v) {
  v += dpp_take<0xb1>(v);          // quad_perm:[1,0,3,2]
  v += dpp_take<0x4e>(v);          // quad_perm:[2,3,0,1]
  v += dpp_take<0x124>(v);         // row_ror:4
  v += dpp_take<0x128>(v);         // row_ror:8   -> every lane holds its row's sum
  v += dpp_take<0x142, 0xa>(v);    // row_bcast:15 into rows 1 and 3
  v += dpp_take<0x143, 0xc>(v);    // row_bcast:31 into rows 2 and 3 -> row 3 holds the wave's sum
  return v;
}
// Sum over a 512-thread workgroup with ONE barrier: the wave sums go to the half of `red` (2 x 8 floats) selected by
// `parity`, which the caller alternates between consecutive calls (the previous call's readers may still be reading).
__device__ __forceinline__ float block_sum512_dpp(float v, float* red, int parity) {
  v = wave_sum_dpp(v);
  if ((threadIdx.x & 63) == 63) red[parity * 8 + (threadIdx.x >> 6)] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) t += red[parity * 8 + k];
  return t;
}

// Minibatch statistics (SB3 PPO.train: advantage mean / unbiased std; train-mode RunningNorm
// update of the feature extractor with the minibatch observations, util/networks.py:111-134).
// Element-parallel gathers staged through LDS; all reductions in fixed order.
// NT = threads of the calling block (1024: prepare / apply kernels, 512: the persistent update).
template <int NT>
__device__ void prepare_stats_t(const ia_policy_desc& d, const float* __restrict__ obs, const float* __restrict__ adv,
                              const int64_t* __restrict__ idx, int batch, int T, int n_envs, int update_norm,
                              float* __restrict__ nm, float* __restrict__ nv, int32_t* __restrict__ ncount,
                              float* __restrict__ advstat, float* lds, int* __restrict__ rowoff = nullptr,
                              float* __restrict__ partial = nullptr) {
  // `partial` (slices of a large minibatch, see the persistent kernel): instead of updating the running
  // statistics, leave this slice's raw moments -- column means [MAXD], column sums of squared deviations
  // [MAXD], advantage mean, advantage sum of squared deviations, row count -- for a later ordered merge.
  const int tid = threadIdx.x;
  float* stage = lds;
  float* red = lds + PREP_STAGE_FLOATS;          // [16][65]
  float* cmean = red + 16 * 65;                  // [65]
  float* misc = cmean + 65;                      // [64]
  // `rowoff` (LDS, >= batch ints; T*n_envs < 2^31): every row's permutation entry is resolved to its
  // tile offset ONCE (one 32-bit divide per row) instead of once per gathered element with 64-bit
  // divides -- the per-element form made the statistics block slower than the gradient chain.
  const bool pre = rowoff != nullptr && idx != nullptr;
  if (pre) {
    for (int i = tid; i < batch; i += NT) {
      const unsigned f = (unsigned)idx[i], env = f / (unsigned)T, t = f - env * (unsigned)T;
      rowoff[i] = (int)(t * (unsigned)n_envs + env);
    }
    __syncthreads();
  }
  auto src_row = [&](int i) -> long long { return pre ? (long long)rowoff[i] : mb_row(idx, i, T, n_envs); };
  // advantages: the first value of each thread stays in a register for the second pass
  const float a0 = tid < batch ? adv[src_row(tid)] : 0.f;
  float s = a0;
  for (int i = tid + NT; i < batch; i += NT) s += adv[src_row(i)];
  const float mean = block_sum<NT>(s, misc) / (float)batch;
  float q = tid < batch ? (a0 - mean) * (a0 - mean) : 0.f;
  for (int i = tid + NT; i < batch; i += NT) {
    const float dl = adv[src_row(i)] - mean;
    q += dl * dl;
  }
  const float qq = block_sum<NT>(q, misc);
  if (tid == 0) {
    if (partial != nullptr) {
      partial[2 * MAXD + 0] = mean;
      partial[2 * MAXD + 1] = qq;
      partial[2 * MAXD + 2] = (float)batch;
    } else {
      advstat[0] = mean;
      advstat[1] = batch > 1 ? sqrtf(qq / (float)(batch - 1)) : 0.f;
    }
  }
  if (!(d.has_norm && update_norm)) return;
  const int D = d.obs_dim, DP = idx ? (D | 1) : D;  // column reads are conflict-free for any row stride
  const int chunk_rows = min(batch, PREP_STAGE_FLOATS / DP);
  const int col = tid & 63, rg = tid >> 6;
  float n_acc = 0.f, m_acc = 0.f, M2 = 0.f;  // Chan accumulators, live in threads tid < D
  for (int c0 = 0; c0 < batch; c0 += chunk_rows) {
    const int rows = min(chunk_rows, batch - c0);
    __syncthreads();
    if (idx == nullptr && NT == 1024) {
      // contiguous rows: a straight linear copy (no index math), all loads of a thread issued
      // before its first LDS store so they are in flight together
      constexpr int MAXIT = PREP_STAGE_FLOATS / NT;
      const float* __restrict__ srcp = obs + (long long)c0 * D;
      const int nel = rows * D;
      float v[MAXIT];
#pragma unroll
      for (int it = 0; it < MAXIT; ++it) {
        const int e = tid + it * NT;
        v[it] = e < nel ? srcp[e] : 0.f;
      }
#pragma unroll
      for (int it = 0; it < MAXIT; ++it) {
        const int e = tid + it * NT;
        if (e < nel) stage[e] = v[it];
      }
    } else {
      for (int e = tid; e < rows * D; e += NT) {
        const int r = e / D, k = e - r * D;
        stage[r * DP + k] = obs[src_row(c0 + r) * D + k];
      }
    }
    __syncthreads();
    float cs = 0.f;
    if (col < D)
      for (int r = rg; r < rows; r += NT / 64) cs += stage[r * DP + col];
    red[rg * 65 + col] = cs;
    __syncthreads();
    if (rg == 0 && col < D) {
      float t = 0.f;
      for (int g = 0; g < NT / 64; ++g) t += red[g * 65 + col];
      cmean[col] = t / (float)rows;
    }
    __syncthreads();
    float cq = 0.f;
    if (col < D) {
      const float cm = cmean[col];
      for (int r = rg; r < rows; r += NT / 64) {
        const float dl = stage[r * DP + col] - cm;
        cq += dl * dl;
      }
    }
    red[rg * 65 + col] = cq;
    __syncthreads();
    if (rg == 0 && col < D) {
      float t = 0.f;
      for (int g = 0; g < NT / 64; ++g) t += red[g * 65 + col];
      const float nb = (float)rows, mb = cmean[col];
      const float tot = n_acc + nb, dlt = mb - m_acc;
      M2 = M2 + t + dlt * dlt * n_acc * nb / tot;
      m_acc = m_acc + dlt * nb / tot;
      n_acc = tot;
    }
  }
  if (partial != nullptr) {
    if (rg == 0 && col < D) {
      partial[col] = m_acc;
      partial[MAXD + col] = M2;
    }
    return;
  }
  const int cnt = *ncount;
  __syncthreads();
  if (rg == 0 && col < D) {
    const float bmean = m_acc, bvar = M2 / (float)batch;
    const float fcount = (float)cnt, fn = (float)batch, tot = (float)((long long)cnt + batch);
    const float delta = bmean - nm[col];
    nm[col] = nm[col] + delta * fn / tot;
    float rv = nv[col] * fcount;
    rv = rv + bvar * fn;
    rv = rv + delta * delta * fcount * fn / tot;
    nv[col] = rv / tot;
  }
  if (tid == 0) *ncount = rn_count_add(cnt, batch);  // saturates at INT32_MAX (rn_common.h)
}

__device__ void prepare_stats(const ia_policy_desc& d, const float* __restrict__ obs, const float* __restrict__ adv,
                              const int64_t* __restrict__ idx, int batch, int T, int n_envs, int update_norm,
                              float* __restrict__ nm, float* __restrict__ nv, int32_t* __restrict__ ncount,
                              float* __restrict__ advstat, float* lds) {
  prepare_stats_t<PREP_THREADS>(d, obs, adv, idx, batch, T, n_envs, update_norm, nm, nv, ncount, advstat, lds);
}

// Statistics of EVERY minibatch of an epoch in one launch, ahead of the per-minibatch kernels (`ia_ppo_epoch`): the
// advantage mean / std and the train-mode feature RunningNorm update of a minibatch depend on its rows only, and the
// running statistics are a sequential merge of per-minibatch moments. Workgroup b takes the raw moments of minibatch
// b's (gathered, contiguous) rows (`prepare_stats_t`, partial form); the workgroup that draws the last ticket applies
// the updates in order (util/networks.py:111-134, the expressions of `prepare_stats_t`) and leaves, per minibatch,
// seq[b] = {adv mean, adv std, -, -, -, -, -, -, mean[MAXD], var[MAXD]} -- the statistics that minibatch's forward
// normalises with. Before, the next minibatch's statistics were one 23 us workgroup inside every apply launch.
constexpr int EPS_PART = 2 * MAXD + 8;       // raw moments of one minibatch (prepare_stats_t's partial layout)
constexpr int EPS_SEQ = 8 + 2 * MAXD;        // published statistics of one minibatch
__global__ __launch_bounds__(PREP_THREADS) void ppo_epoch_stats_kernel(ia_policy_desc d, const float* __restrict__ obs,
                                                                       const float* __restrict__ adv, long long total,
                                                                       int batch, int update_norm, float* __restrict__ nm,
                                                                       float* __restrict__ nv, int32_t* __restrict__ ncount,
                                                                       float* __restrict__ seq, float* __restrict__ part,
                                                                       unsigned* __restrict__ ticket) {
  extern __shared__ float lds[];
  __shared__ int is_last;
  const int tid = threadIdx.x, b = blockIdx.x, n_mb = gridDim.x;
  const long long start = (long long)b * batch;
  const int n = (int)((total - start) < batch ? (total - start) : batch);
  prepare_stats_t<PREP_THREADS>(d, obs + start * d.obs_dim, adv + start, nullptr, n, 1, 1, update_norm, nm, nv, ncount,
                                nullptr, lds, nullptr, part + (long long)b * EPS_PART);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned tk = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    is_last = tk == (unsigned)n_mb - 1u;
    if (is_last) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  __syncthreads();
  if (!is_last) return;
  auto rows_of = [&](int k) { const long long s0 = (long long)k * batch; return (int)((total - s0) < batch ? (total - s0) : batch); };
  for (int k = tid; k < n_mb; k += PREP_THREADS) {
    const float* pk = part + (long long)k * EPS_PART;
    const int nk = rows_of(k);
    seq[(long long)k * EPS_SEQ + 0] = pk[2 * MAXD + 0];
    seq[(long long)k * EPS_SEQ + 1] = nk > 1 ? sqrtf(pk[2 * MAXD + 1] / (float)(nk - 1)) : 0.f;
  }
  if (!(d.has_norm && update_norm)) return;
  const int cnt0 = *ncount;
  __syncthreads();
  if (tid < d.obs_dim) {
    float mean = nm[tid], var = nv[tid];
    int cnt = cnt0;
    for (int k = 0; k < n_mb; ++k) {
      const float* pk = part + (long long)k * EPS_PART;
      const int nk = rows_of(k);
      const float bmean = pk[tid], bvar = pk[MAXD + tid] / (float)nk;
      const float fcount = (float)cnt, fn = (float)nk, tot = (float)((long long)cnt + nk);
      const float delta = bmean - mean;
      mean = mean + delta * fn / tot;
      float rv = var * fcount;
      rv = rv + bvar * fn;
      rv = rv + delta * delta * fcount * fn / tot;
      var = rv / tot;
      cnt = rn_count_add(cnt, nk);
      seq[(long long)k * EPS_SEQ + 8 + tid] = mean;
      seq[(long long)k * EPS_SEQ + 8 + MAXD + tid] = var;
    }
    nm[tid] = mean;
    nv[tid] = var;
    if (tid == 0) *ncount = cnt;
  }
}

__global__ __launch_bounds__(PREP_THREADS) void ppo_prepare_kernel(ia_policy_desc d, const float* __restrict__ obs,
                                                                   const float* __restrict__ adv,
                                                                   const int64_t* __restrict__ idx, int batch, int T,
                                                                   int n_envs, int update_norm, float* __restrict__ nm,
                                                                   float* __restrict__ nv, int32_t* __restrict__ ncount,
                                                                   float* __restrict__ advstat) {
  extern __shared__ float lds[];
  prepare_stats(d, obs, adv, idx, batch, T, n_envs, update_norm, nm, nv, ncount, advstat, lds);
}

// G[J][K] = sum over the block's 64 rows of U[r][j] * V[r][k]; U/V are LDS tiles with odd strides
// (fragment reads are bank-conflict free); result written (not accumulated) to `dst` with leading
// dimension ldk, clipped to j<jmax, k<kmax. One wave, 32 MFMAs.
__device__ __forceinline__ void mfma_outer_store(const float* U, int us, const float* V, int vs, int j0, int k0,
                                                 int jmax, int kmax, float* __restrict__ dst, int ldk, int lane) {
  const int li = lane & 31, lh = lane >> 5;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll 8
  for (int r = 0; r < ROWS; r += 2) {
    const float a = U[(r + lh) * us + j0 + li];
    const float b = V[(r + lh) * vs + k0 + li];
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
  }
  const int k = k0 + li;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int j = j0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
    if (j < jmax && k < kmax) dst[j * ldk + k] = acc[r];
  }
}

__device__ __forceinline__ void column_sum_store(const float* tile, int stride, int ncols, float* __restrict__ dst,
                                                 int lane) {
  if (lane < ncols) {
    float s = 0.f;
#pragma unroll 8
    for (int r = 0; r < ROWS; ++r) s += tile[r * stride + lane];
    dst[lane] = s;
  }
}

// Gradient-slab store of the persistent kernel: write-through (system-scope relaxed store = global_store sc0 sc1), so
// the agent-scope release fence before the grid barrier finds no dirty L2 lines of the slab left to write back.
__device__ __forceinline__ void slab_store(float* p, float v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// (value, sequence) words: one aligned 8-byte store / load each, past the caches of the other compute units (agent scope:
// within the GPU; system scope: peer-mapped memory of another GPU)
__device__ __forceinline__ unsigned long long ll_pack(float v, unsigned seq) {
  return ((unsigned long long)seq << 32) | (unsigned long long)__float_as_uint(v);
}
__device__ __forceinline__ void ll_store_agent(unsigned long long* p, float v, unsigned seq) {
  __hip_atomic_store(p, ll_pack(v, seq), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// ... for readers on the SAME XCD only (workgroup-scope store: the line stays in that XCD's L2, which is the point of
// coherence for its compute units -- no fabric write per word; agent-scope polls bypass L1 and are served by that L2)
__device__ __forceinline__ void ll_store_xcd(unsigned long long* p, float v, unsigned seq) {
  __hip_atomic_store(p, ll_pack(v, seq), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void ll_store_system(unsigned long long* p, float v, unsigned seq) {
  __hip_atomic_store(p, ll_pack(v, seq), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// The same sums (rows added in order 0 .. ROWS-1) with the reads issued sixteen at a time.
__device__ __forceinline__ float column_sum_b(const float* col, int stride) {
  float s = 0.f;
#pragma unroll
  for (int r0 = 0; r0 < ROWS; r0 += 16) {
    float t[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) t[u] = col[(r0 + u) * stride];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 16; ++u) s += t[u];
  }
  return s;
}
__device__ __forceinline__ void column_sum_store_b(const float* tile, int stride, int ncols, float* __restrict__ dst,
                                                   int lane) {
  if (lane < ncols) dst[lane] = column_sum_b(tile + lane, stride);
}

template <int H, int NTOW = 2>
struct GLds {  // LDS carve-up of the 8-wave gradient kernel (floats); NTOW = 1: one tower per workgroup (4 waves)
  static constexpr int XS = MAXD + 1, HS = H + 1, AS = MAXA + 1, MS = 9;
  static constexpr int x = 0;
  static constexpr int a1 = x + ROWS * XS;             // [NTOW towers][ROWS][HS]
  static constexpr int a2 = a1 + NTOW * ROWS * HS;
  static constexpr int dz = a2 + NTOW * ROWS * HS;
  static constexpr int out = dz + NTOW * ROWS * HS;
  static constexpr int dout = out + ROWS * AS;
  static constexpr int aux = dout + ROWS * AS;
  static constexpr int misc = aux + ROWS * AS;         // [ROWS][MS]: 0 = value, 1 = dvalue
  static constexpr int total = misc + ROWS * MS + 64;
};

// ---------------------------------------------------------------------------------------------
// H = 32 specialisation built on v_mfma_f32_16x16x4_f32 (lane l: li = l&15, lk = l>>4; A[i=li][k=lk],
// B[k=lk][j=li], C: col = li, rows = 4*lk + reg). 64 rows per block, 8 waves: waves 0-3 = policy
// tower, 4-7 = value tower; wave q owns row-tile q (16 rows) of every layer output and one 16x16
// tile of every weight gradient. All weight (B) fragments are fetched into VGPRs at kernel start, so
// the layer chain only touches LDS (activations) and the matrix pipe: no dependent global or scalar
// loads inside the phases.
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// Rows of one minibatch as the kernels address them.
struct MbRows {
  const float *obs, *actions, *old_logp, *adv, *ret;
  const int64_t* idx;   // permutation slice (null: rows already gathered, contiguous)
  int batch, T, n_envs;
};

// LDS staging area of the persistent kernel: the block's raw feature rows and per-row scalars of the
// NEXT minibatch (prefetched), plus a copy of its statistics-ring slot.
struct UpdStage {
  static constexpr int x = 0;                                   // [ROWS][D] raw observations, rows packed (room for D = MAXD + 1)
  static constexpr int oldlp = ROWS * (MAXD + 1);               // [ROWS]
  static constexpr int adv = oldlp + ROWS, ret = adv + ROWS;    // [ROWS] each
  static constexpr int src = ret + ROWS;                        // [ROWS] row offset into the rollout tile (as float bits)
  static constexpr int nxt = src + ROWS;                        // [ROWS] row offsets of the minibatch being prefetched
  static constexpr int ring = nxt + ROWS;                       // [UPD_RS_] copy of the minibatch's statistics-ring slot
                                                                // (+ 56 floats: its 8-float tail arrives as one 64-lane LDS-direct load)
  static constexpr int act = ring + 2 * MAXD + 64;              // [ROWS][aw] the rows' actions, row-major, aw <= MAXA
  __host__ __device__ static constexpr int total(int aw) { return act + ROWS * aw; }
};
// One minibatch of block `vblk` (64 rows), one launch per minibatch (`ia_ppo_minibatch*`, `ia_ppo_epoch`:
// hidden = 32 without the persistent kernel, i.e. data-parallel runs): forward, losses, backward;
// gradient partials -> `slab`, loss-statistic partials -> `statpart[0..4]`. LOAD_PARAMS: copy the flat
// parameter vectors into LDS first.
// SPLIT (H = 64, ppo_epoch_persistent_kernel<64, true>): the workgroup is FOUR waves and runs ONE tower (`tower`: 0 policy,
// 1 value) of the block's 64 rows; the other tower of the same rows is another workgroup. One tower's tiles are 82 KB, so
// that tower's parameters (both layouts, <= 71 KB) are resident in LDS like the H = 32 kernel's: every weight fragment is
// an LDS read instead of a load from the memory-side cache, one wave per SIMD has the matrix pipe to itself, and the block
// barriers are among four waves. The two workgroups of a row block write disjoint parts of the block's slab.
// NLL > 0 (SPLIT only; ppo_epoch_ll_kernel): the word-exchange form. The tower's parameters arrive as 8-byte (value,
// sequence) words `ll.par` -- every thread polls its NLL words of the tower's layers (+ head, log_std) until all carry
// `ll.par_seq` -- and every gradient / statistics entry LEAVES as such a word with `ll.out_seq`: `slab` / `statpart` are
// then word arrays (pointers to 8-byte words in float* clothing: only used for offsets). A poll that times out raises
// `*ll.fail` (LDS) and `*ll.err`; the caller returns behind this function's closing barrier.
struct MbLl {
  const unsigned long long* par; unsigned par_seq, out_seq; int* fail; unsigned* err;
};
template <int H, bool LOAD_PARAMS, bool SPLIT = false, int NLL = 0>
__device__ __forceinline__ void mfma_minibatch(
    const ia_policy_desc& d, const float* __restrict__ P, const float* __restrict__ Pt, const float* __restrict__ nm,
    const float* __restrict__ nv, const float adv_mean, const float adv_std, const MbRows rows, const int vblk,
    const int normalize_adv, const float clip, const float ent_coef, const float vf_coef, float* __restrict__ slab,
    float* __restrict__ statpart, float* __restrict__ lds_in, long long* __restrict__ tstamp, const int oz = 0,
    const int tower = 0, const MbLl ll = MbLl{}, const int* __restrict__ tdst = nullptr) {
  // (`tdst`, word-exchange form: where in the transposed image each of the thread's NLL parameter words goes, -1: nowhere
  //  -- the same every step, so the caller works it out once per launch and keeps it in registers)
  static_assert(!SPLIT || (H == 64 && LOAD_PARAMS), "the one-tower form is the 64-wide epoch kernel's");
  static_assert(NLL == 0 || SPLIT, "the word-exchange form is the one-tower kernel's");
  constexpr bool LLX = NLL > 0;
  constexpr int NT = SPLIT ? 256 : 512;
  // `oz`: an opaque zero when the function sits inside a step loop (ppo_epoch_persistent_kernel): every per-lane offset is
  // then re-derived per step instead of being hoisted out of the loop into (spilled) registers
  float* __restrict__ lds = lds_in + oz;
  const float* __restrict__ obs = rows.obs;
  const float* __restrict__ actions = rows.actions;
  const float* __restrict__ old_logp = rows.old_logp;
  const float* __restrict__ adv = rows.adv;
  const float* __restrict__ ret = rows.ret;
  const int64_t* __restrict__ idx = rows.idx;
  const int batch = rows.batch, T = rows.T, n_envs = rows.n_envs;
  constexpr int NC = H / 16, KS = H / 4;   // 16-column tiles of a layer output, MFMA steps over a hidden layer
  using L = GLds<H, SPLIT ? 1 : 2>;
#define IA_TS(slot) do { if (tstamp && vblk == 0 && threadIdx.x == 0) tstamp[slot] = clock64(); } while (0)
  const int tid = threadIdx.x + oz, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tw = SPLIT ? tower : wv >> 2, q = SPLIT ? wv : wv & 3;
  const int li = lane & 15, lk = lane >> 4;
  const int D = d.obs_dim, A = d.act_dim;
  const PolOff o = pol_offsets(D, A, H, d.discrete);
  const int i0 = vblk * ROWS;
  const int aw = d.discrete ? 1 : A;
  const float invB = 1.f / (float)batch;
  const int S1 = (D + 3) >> 2, SA = (A + 3) >> 2;

  const int oW1 = tw ? o.vW1 : o.pW1, ob1 = tw ? o.vb1 : o.pb1, oW2 = tw ? o.vW2 : o.pW2, ob2 = tw ? o.vb2 : o.pb2;
  IA_TS(0);

  // ---- phase 0a: issue the feature-row loads FIRST (VMEM returns in order: the LDS stores below then
  // only wait for these, while the weight fragments requested next keep streaming in behind them)
  // (the block's ROWS x D real elements, element e = row * D + column: one division per thread, then (row, column) advance
  //  by NT elements per trip; trips past ROWS * D are skipped wave-uniformly)
  // (the word-exchange form is instantiated per observation-width class -- NLL = 20: D <= 14, 24: D <= 30 -- and takes
  //  its row-load trips from the class: at D = 11 three of MAXD's sixteen trips carry elements)
  constexpr int DMAX = NLL == 20 ? 14 : (NLL == 24 ? 30 : MAXD);
  constexpr int NIT = (ROWS * DMAX + NT - 1) / NT;
  const int nel = ROWS * D;
  float xv[NIT], xm[NIT], xs_[NIT];
  int xi[NIT];   // LDS offset of the element inside the x tile (-1: none)
  {
    // every load is UNCONDITIONAL at a clamped address (values selected afterwards): behind a divergent branch the
    // compiler cannot count a load in vmcnt and each trip would wait for the one before -- a memory round trip per trip
    int r = tid / D, k = tid - r * D;
    const int dr = NT / D, dk = NT - dr * D;
    const float* nmp = d.has_norm ? nm : obs;   // (no statistics: any readable address, the value is discarded)
    const float* nvp = d.has_norm ? nv : obs;
    unsigned okbits = 0u;   // bit it: the trip's element is a real row of the minibatch
    // (no branch around a trip either, not even a uniform one: the merge of "loaded" and "not loaded" values is a copy
    //  that waits for the load; trips past ROWS * D re-read the last row at clamped addresses and are discarded)
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const bool on = tid + it * NT < nel;
      okbits |= (on && i0 + r < batch) ? 1u << it : 0u;
      xv[it] = obs[mb_row(idx, min(i0 + min(r, ROWS - 1), batch - 1), T, n_envs) * D + k];
      xm[it] = nmp[k];
      xs_[it] = nvp[k];
      xi[it] = on ? r * L::XS + k : -1;
      r += dr; k += dk;
      if (k >= D) { k -= D; ++r; }
    }
    __builtin_amdgcn_sched_barrier(0);   // (the selects below consume the values: all the loads are issued before the first)
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const bool ok = (okbits >> it) & 1u;
      xv[it] = ok ? xv[it] : 0.f;
      xm[it] = (ok && d.has_norm) ? xm[it] : 0.f;
      xs_[it] = (ok && d.has_norm) ? xs_[it] : 1.f - d.norm_eps;
    }
  }
  // per-row scalars of the loss phase: every wave takes the loss terms of ITS 16 rows (the rows whose head outputs it
  // wrote) -- policy waves with four lanes per row (lane = 4 * row + part, actions part, part + 4, ...: the action loops are
  // MAXA / 4 trips and the row sums two cross-lane adds), value waves with lanes 0 .. 15 (the form of the H = 32 chain)
  const int lrow = tw == 0 ? q * 16 + (lane >> 2) : q * 16 + (lane & 15);
  const int part = lane & 3;
  const bool loss_lane = tw == 0 || lane < 16;
  const int i = i0 + lrow;
  const bool valid = i < batch;
  const long long src = mb_row(idx, min(i, batch - 1), T, n_envs);
  float r_oldlp = 0.f, r_adv = 0.f, r_ret = 0.f, r_act[4];
  if (tw == 0) {   // (wave-uniform; consumed in phase 4: the latency hides behind phases 1-3)
    r_oldlp = old_logp[src];
    r_adv = adv[src];
#pragma unroll
    for (int j = 0; j < 4; ++j) r_act[j] = actions[src * aw + min(part + 4 * j, aw - 1)];
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) r_act[j] = 0.f;
    r_ret = ret[src];
  }

  IA_TS(9);
  // ---- parameters: ONE cooperative, coalesced 16-byte copy of both flat vectors (torch layout P and
  // the transposed shadow copy Pt) into LDS; every weight fragment below is then an LDS read.
  // (H = 64: the 64-wide activation tiles leave no room for parameter copies -- fragments come straight from the
  // L2-resident flat vectors)
  const float* sP = H == 32 ? lds + L::total : P;
  const float* sPt = H == 32 ? sP + ((o.total + 3) & ~3) : Pt;
  // SPLIT: LDS images of what the tower reads -- A = its layers W1 b1 W2 b2 in torch layout, T = the same piece of the
  // transposed copy, B = its head (aW, ab | cW, cb), C = log_std -- each image starting AT the piece's first element, so
  // that every fragment row is 16-byte aligned in LDS (the pieces' sizes are multiples of H) and a lane's four values of
  // a weight row are one ds_read_b128 (from the flat vector's own alignment they were four conflicted ds_read_b32). The
  // global side of the copy reads 16 bytes at 4-byte alignment. sPA / sPt / sPB / sLS are biased so that the flat offsets
  // of PolOff address the images. The loads are issued here, the LDS stores come behind the row staging below: the
  // staging arithmetic runs while the parameters are on their way.
  const float* sPA = sP;   // reads of the tower's layers / of its head / of log_std (the same vector unless SPLIT)
  const float* sPB = sP;
  const float* sLS = sP;
  typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));
  constexpr int NA = (SPLIT && !LLX) ? (H * MAXD + H + H * H + H + 4 * NT - 1) / (4 * NT) : 1;   // 16-byte pieces per thread, images A / T
  constexpr int NB = SPLIT ? (MAXA * H + MAXA + NT - 1) / NT : 1;                      // elements per thread, image B
  constexpr int NLA = LLX ? NLL : 1;                                                   // words per thread, image A
  float4 va[NA], vt[NA];
  float vb[NB];
  float wa[NLA];
  float lsv = 0.f;
  const int tlen = H * D + H + H * H + H, t0 = tower ? o.vW1 : o.pW1;
  const int segB0 = tower ? o.cW : o.aW, nB = (tower ? o.total : o.cW) - segB0, lenB = (nB + 3) & ~3;
  // (word-exchange form: image T holds W1^T and W2^T only, rows H + 4 floats apart -- the polled words go straight to
  //  their transposed places, and with that row distance the 64 columns of a W2 row land in 16 banks instead of one)
  constexpr int TS = LLX ? H + 4 : H;
  float* imgA = lds + ((L::total + 3) & ~3);
  float* imgT = imgA + tlen;
  float* imgB = imgT + (LLX ? (D + H) * TS : tlen);
  float* imgC = imgB + lenB;
  auto stage_x = [&]() {
#pragma unroll
    for (int it = 0; it < NIT; ++it)
      if (xi[it] >= 0) lds[L::x + xi[it]] = (xv[it] - xm[it]) / sqrtf(xs_[it] + d.norm_eps);
    {
      // columns D .. 4 * S1 - 1 feed layer 1's last MFMA step (against zero weight rows): they must be finite
      const int padw = 4 * S1 - D;
      for (int e = tid; e < ROWS * padw; e += NT) {
        const int r = e / padw;
        lds[L::x + r * L::XS + D + e - r * padw] = 0.f;
      }
    }
    for (int e = tid; e < ROWS * L::AS; e += NT) { lds[L::dout + e] = 0.f; lds[L::aux + e] = 0.f; lds[L::out + e] = 0.f; }
    for (int e = tid; e < ROWS * L::MS; e += NT) lds[L::misc + e] = 0.f;
  };
  if constexpr (LLX) {
    // (the rows are normalised into the x tile FIRST: the owners publish the parameters at about the same moment everywhere,
    //  and their flight -- plus the row loads' -- passes under this work instead of under a spin)
    stage_x();
    // the parameters as the chunk owners published them (this step's sequence number): all words requested together, the
    // whole set again until every one has arrived (the row loads above are in flight meanwhile)
    typedef unsigned long long u64;
    const u64* pa = ll.par + t0;
    const u64* pb = ll.par + segB0;
    const u64* pc = ll.par + (d.discrete ? 0 : o.log_std + min(tid, A - 1));
    u64 ta[NLA], tb[NB], tc;
    unsigned it = 0;
    for (;;) {
#pragma unroll
      for (int i = 0; i < NLA; ++i)
        ta[i] = __hip_atomic_load(pa + min(tid + i * NT, tlen - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
      for (int i = 0; i < NB; ++i)
        tb[i] = __hip_atomic_load(pb + min(tid + i * NT, nB - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      tc = __hip_atomic_load(pc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      bool ok = (unsigned)(tc >> 32) == ll.par_seq;
#pragma unroll
      for (int i = 0; i < NLA; ++i) ok = ok && (unsigned)(ta[i] >> 32) == ll.par_seq;
#pragma unroll
      for (int i = 0; i < NB; ++i) ok = ok && (unsigned)(tb[i] >> 32) == ll.par_seq;
      if (__all(ok)) break;
      __builtin_amdgcn_s_sleep(1);
      if (++it > (1u << 22) || ((it & 255u) == 0 && __hip_atomic_load(ll.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
        if (lane == 0) {
          __hip_atomic_store(ll.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          *ll.fail = 1;
        }
        break;
      }
    }
#pragma unroll
    for (int i = 0; i < NLA; ++i) wa[i] = __uint_as_float((unsigned)ta[i]);
#pragma unroll
    for (int i = 0; i < NB; ++i) vb[i] = __uint_as_float((unsigned)tb[i]);
    lsv = __uint_as_float((unsigned)tc);
    sPA = imgA - t0;
    sPt = imgT - t0;
    sPB = imgB - segB0;
    sLS = imgC - (d.discrete ? 0 : o.log_std);
    // image A (torch layout) and, for the two weight matrices, the transposed places of image T; the head and log_std
#pragma unroll
    for (int i = 0; i < NLA; ++i) {
      const int e = tid + i * NT;
      if (e < tlen) imgA[e] = wa[i];
      if (tdst[i] >= 0) imgT[tdst[i]] = wa[i];
    }
#pragma unroll
    for (int i = 0; i < NB; ++i)
      if (tid + i * NT < nB) imgB[tid + i * NT] = vb[i];
    if (tid < MAXA) imgC[tid] = lsv;
  } else if constexpr (SPLIT) {
    // every load UNCONDITIONAL at a clamped index (see the row loads above); images A / T lie inside the flat vector whole,
    // the head goes element by element (its last 16-byte piece could run past the vector's end)
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int e = min(tid + i * NT, (tlen >> 2) - 1);
      const f32x4_u t = *reinterpret_cast<const f32x4_u*>(P + t0 + 4 * e);
      va[i] = make_float4(t[0], t[1], t[2], t[3]);
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) vb[i] = P[segB0 + min(tid + i * NT, nB - 1)];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int e = min(tid + i * NT, (tlen >> 2) - 1);
      const f32x4_u t = *reinterpret_cast<const f32x4_u*>(Pt + t0 + 4 * e);
      vt[i] = make_float4(t[0], t[1], t[2], t[3]);
    }
    lsv = P[d.discrete ? 0 : o.log_std + min(tid, A - 1)];
    sPA = imgA - t0;
    sPt = imgT - t0;
    sPB = imgB - segB0;
    sLS = imgC - (d.discrete ? 0 : o.log_std);
  }
  auto store_images = [&]() {
    if constexpr (LLX) {
      // (nothing left: the poll wrote every image)
    } else if constexpr (SPLIT) {
#pragma unroll
      for (int i = 0; i < NA; ++i)
        if (tid + i * NT < (tlen >> 2)) {
          reinterpret_cast<float4*>(imgA)[tid + i * NT] = va[i];
          reinterpret_cast<float4*>(imgT)[tid + i * NT] = vt[i];
        }
#pragma unroll
      for (int i = 0; i < NB; ++i)
        if (tid + i * NT < nB) imgB[tid + i * NT] = vb[i];
      if (tid < MAXA) imgC[tid] = lsv;
    }
  };
  if (LOAD_PARAMS && H == 32) {
    float* wP = lds + L::total;
    float* wPt = wP + ((o.total + 3) & ~3);
    const int n4 = (o.total + 3) >> 2;
    const bool vec = ((reinterpret_cast<uintptr_t>(P) | reinterpret_cast<uintptr_t>(Pt)) & 15) == 0;
    for (int e = tid; e < n4; e += NT) {
      float4 v0, v1;
      if (vec && 4 * e + 3 < o.total) {
        v0 = reinterpret_cast<const float4*>(P)[e];
        v1 = reinterpret_cast<const float4*>(Pt)[e];
      } else {
        float t0[4], t1[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int ii = min(4 * e + u, o.total - 1);
          t0[u] = P[ii];
          t1[u] = Pt[ii];
        }
        v0 = make_float4(t0[0], t0[1], t0[2], t0[3]);
        v1 = make_float4(t1[0], t1[1], t1[2], t1[3]);
      }
      reinterpret_cast<float4*>(wP)[e] = v0;
      reinterpret_cast<float4*>(wPt)[e] = v1;
    }
  }
  IA_TS(10);
  // ---- phase 0b: normalise + stage the feature rows in LDS; clear the small tiles
  auto stage_rows = [&]() {
    if constexpr (!LLX) stage_x();   // (word-exchange form: done ahead of the parameter poll, see there)
    store_images();   // (SPLIT: the parameter images, whose loads were issued ahead of this staging)
    __syncthreads();
  };
  // With the parameters already resident in LDS the fragment reads below do not depend on this
  // barrier, so the staging (which waits for the normaliser statistics, a global load) goes after them.
  if (LOAD_PARAMS) stage_rows();

  IA_TS(11);
  // weight fragments -> VGPR: B[k = 4s+lk][j = CJ(c)]. H = 32: column tile c is columns c*16 + li (LDS reads). H = 64:
  // tile c is columns li*NC + c, so the NC values a lane needs from one weight row are 16 contiguous bytes -- one
  // global_load_dwordx4 instead of four scalar loads (144 -> 40 load instructions per wave; the fragments come from L2)
  auto CJ = [&](int c) { return H == 32 ? c * 16 + li : li * NC + c; };
  auto ldc = [&](const float* __restrict__ rowbase, float (&out)[NC]) {
    if constexpr (H == 32) {
#pragma unroll
      for (int c = 0; c < NC; ++c) out[c] = rowbase[c * 16 + li];
    } else if constexpr (LLX) {
      // (word-exchange form: every image starts on a 16-byte boundary of LDS -- the caller rounds the dynamic area's base --
      //  and every fragment row is a multiple of four floats into its image: one ds_read_b128 per fragment)
      const f32x4 t = *reinterpret_cast<const f32x4*>(rowbase + li * NC);
#pragma unroll
      for (int c = 0; c < NC; ++c) out[c] = t[c];
    } else {
      const f32x4_u t = *reinterpret_cast<const f32x4_u*>(rowbase + li * NC);
#pragma unroll
      for (int c = 0; c < NC; ++c) out[c] = t[c];
    }
  };
  // (Every batch of fragment / operand reads below is issued as a block, then a scheduling fence, then its consumers: in
  //  source order "read, use, read, use" the compiler keeps that order in a kernel of this size and every use waits for its
  //  own read -- ~100 cycles per MFMA. With the block form the waits are in-order counters behind one latency.)
#define IA_FENCE() __builtin_amdgcn_sched_barrier(0)
  // slab / statistics stores: write-through in the one-tower form (the grid barrier's release fence then finds no dirty
  // lines of the 44 KB slab to write back, as in the H = 32 persistent kernel), plain stores otherwise
  auto sst = [&](float* __restrict__ p_, float v_) {
    if constexpr (LLX) ll_store_agent(reinterpret_cast<unsigned long long*>(slab) + (p_ - slab), v_, ll.out_seq);
    else if constexpr (SPLIT) slab_store(p_, v_);
    else *p_ = v_;
  };
  const float* tW1 = LLX ? imgT : sPt + oW1;          // rows of W1^T / W2^T, TS floats apart
  const float* tW2 = LLX ? imgT + D * TS : sPt + oW2;
  float bW1[16][NC], bW2[KS][NC], bW2o[KS][NC], bHead[KS], bDa2[4][NC], b1v[NC], b2v[NC], cwv[NC];
#pragma unroll
  for (int s = 0; s < 16; ++s) {
#pragma unroll
    for (int c = 0; c < NC; ++c) bW1[s][c] = 0.f;
    if (s < S1) ldc(tW1 + min(4 * s + lk, D - 1) * TS, bW1[s]);  // wave-uniform: steps beyond the observation width issue no reads
  }
  const int head_row = tw == 0 ? o.aW + min(li, A - 1) * H : o.cW;
  const bool head_on = tw == 0 ? li < A : li == 0;
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const int kk = 4 * s + lk;
    ldc(tW2 + kk * TS, bW2[s]);   // W2^T[k][j]
    bHead[s] = sPB[head_row + kk];
  }
  IA_FENCE();
#pragma unroll
  for (int s = 0; s < 16; ++s)
    if (s < S1) {
#pragma unroll
      for (int c = 0; c < NC; ++c) bW1[s][c] = 4 * s + lk < D ? bW1[s][c] : 0.f;
    }
#pragma unroll
  for (int s = 0; s < KS; ++s) bHead[s] = head_on ? bHead[s] : 0.f;
  IA_TS(12);
  // fragments of the backward phases: H = 32 takes them here (LDS reads, resident all along); H = 64 requests them
  // from L2 after the forward chain, when the forward fragments' registers are free again
  auto load_backward_fragments = [&]() {
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int kk = 4 * s + lk;
      ldc(sPA + oW2 + kk * H, bW2o[s]);   // W2[j=k][k'] (row j contiguous)
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int aa = 4 * s + lk;
#pragma unroll
      for (int c = 0; c < NC; ++c) bDa2[s][c] = 0.f;
      if (!SPLIT || tw == 0) ldc(sPB + o.aW + min(aa, A - 1) * H, bDa2[s]);   // (SPLIT: the value tower holds no aW)
    }
    IA_FENCE();
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int c = 0; c < NC; ++c) bDa2[s][c] = (tw == 0 && 4 * s + lk < A) ? bDa2[s][c] : 0.f;
  };
  if (H == 32) load_backward_fragments();
  ldc(sPA + ob1, b1v);
  ldc(sPA + ob2, b2v);
#pragma unroll
  for (int c = 0; c < NC; ++c) cwv[c] = 0.f;
  if (!SPLIT || tw == 1) ldc(sPB + o.cW, cwv);   // (used by the value tower only)
  const float head_bias = tw == 0 ? (li < A ? sPB[o.ab + li] : 0.f) : sPB[o.cb];
  IA_TS(13);
  // per-action Gaussian constants of the lane's actions (policy waves): 1 / sd^2, log sd with sd = exp(log_std)
  float c_ivar[4], c_logsd[4];
  const bool need_sd = tw == 0 && !d.discrete;
  {
    // lane l works out the pair of action (l & 15) -- one exp, one division, one log per lane instead of four -- and the
    // lanes pick theirs up by shuffle (the form of the H = 32 chain; same expressions, same values)
    float my_ivar = 1.f, my_logsd = need_sd ? sLS[o.log_std + min(lane & 15, A - 1)] : 0.f;
    IA_FENCE();
    if (need_sd) {
      const float sd = expf(my_logsd);
      my_ivar = 1.f / (sd * sd);
      my_logsd = logf(sd);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      c_ivar[j] = __shfl(my_ivar, part + 4 * j, 64);
      c_logsd[j] = __shfl(my_logsd, part + 4 * j, 64);
    }
  }

  if (!LOAD_PARAMS) stage_rows();
  float* a1t = lds + L::a1 + (SPLIT ? 0 : tw) * ROWS * L::HS;
  float* a2t = lds + L::a2 + (SPLIT ? 0 : tw) * ROWS * L::HS;
  float* dzt = lds + L::dz + (SPLIT ? 0 : tw) * ROWS * L::HS;
  const int arow = q * 16 + li;  // row whose A fragment this lane feeds
  IA_TS(1);
  // ---- phase 1: a1 = tanh(x W1^T + b1)
  {
    f32x4 acc[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    float av[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) av[s] = s < S1 ? lds[L::x + arow * L::XS + 4 * s + lk] : 0.f;
    IA_FENCE();
#pragma unroll
    for (int s = 0; s < 16; ++s)
      if (s < S1) {
#pragma unroll
        for (int c = 0; c < NC; ++c) acc[c] = mfma16(av[s], bW1[s][c], acc[c]);
      }
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) a1t[(q * 16 + lk * 4 + r) * L::HS + CJ(c)] = fast_tanh(acc[c][r] + b1v[c]);
  }
  __syncthreads();
  IA_TS(2);
  // ---- phase 2: a2 = tanh(a1 W2^T + b2)
  {
    f32x4 acc[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    float av[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) av[s] = a1t[arow * L::HS + 4 * s + lk];
    IA_FENCE();
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int c = 0; c < NC; ++c) acc[c] = mfma16(av[s], bW2[s][c], acc[c]);
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) a2t[(q * 16 + lk * 4 + r) * L::HS + CJ(c)] = fast_tanh(acc[c][r] + b2v[c]);
  }
  __syncthreads();
  IA_TS(3);
  // ---- phase 3: heads (policy: action_net -> out[row][a]; value: value_net -> misc[row][0])
  {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    float av[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) av[s] = a2t[arow * L::HS + 4 * s + lk];
    IA_FENCE();
#pragma unroll
    for (int s = 0; s < KS; ++s) acc = mfma16(av[s], bHead[s], acc);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = q * 16 + lk * 4 + r;
      if (tw == 0) { if (li < A) lds[L::out + row * L::AS + li] = acc[r] + head_bias; }
      else if (li == 0) lds[L::misc + row * L::MS + 0] = acc[r] + head_bias;
    }
  }
  if (H != 32) load_backward_fragments();   // (in flight during the loss phase)
  __syncthreads();
  IA_TS(4);
  // ---- phase 4: per-row losses of the wave's 16 rows (see the loads above)
  if (loss_lane) {
    if (tw == 0) {
      const float* outrow = lds + L::out + lrow * L::AS;
      float* doutrow = lds + L::dout + lrow * L::AS;
      float* auxrow = lds + L::aux + lrow * L::AS;
      float logp = 0.f, entropy = 0.f, lse = 0.f;
      int act_i = 0;
      float o_[4];   // the head outputs of this lane's actions, read in one block (columns >= A hold zeros)
#pragma unroll
      for (int j = 0; j < 4; ++j) o_[j] = outrow[part + 4 * j];
      if (d.discrete) act_i = (int)r_act[0];
      const float o_act = outrow[act_i];
      IA_FENCE();
      auto quad_sum = [](float v) {
        v += __shfl_xor(v, 1, 64);
        v += __shfl_xor(v, 2, 64);
        return v;
      };
      if (!d.discrete) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (part + 4 * j < A) {
            const float diff = r_act[j] - o_[j];
            logp += -(diff * diff) * (0.5f * c_ivar[j]) - c_logsd[j] - LOG_SQRT_2PI;
            entropy += 0.5f + LOG_SQRT_2PI + c_logsd[j];
          }
        logp = quad_sum(logp);
        entropy = quad_sum(entropy);
      } else {
        float mx = -3.0e38f;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (part + 4 * j < A) mx = fmaxf(mx, o_[j]);
        mx = fmaxf(mx, __shfl_xor(mx, 1, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 2, 64));
        float se = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (part + 4 * j < A) se += expf(o_[j] - mx);
        lse = mx + logf(quad_sum(se));
        logp = o_act - lse;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (part + 4 * j < A) {
            const float l = o_[j] - lse;
            entropy -= expf(l) * l;
          }
        entropy = quad_sum(entropy);
      }
      float advn = r_adv;
      if (normalize_adv && batch > 1) advn = (advn - adv_mean) / (adv_std + 1e-8f);
      const float log_ratio = logp - r_oldlp;
      const float ratio = expf(log_ratio);
      const float lo = 1.f - clip, hi = 1.f + clip;
      const float pl1 = advn * ratio;
      const float pl2 = advn * fminf(fmaxf(ratio, lo), hi);
      const float g1 = pl1 < pl2 ? 1.f : (pl1 == pl2 ? 0.5f : 0.f);
      const float g2 = pl2 < pl1 ? 1.f : (pl1 == pl2 ? 0.5f : 0.f);
      const float inrange = (ratio >= lo && ratio <= hi) ? 1.f : 0.f;
      const float dlogp = valid ? -invB * advn * (g1 + g2 * inrange) * ratio : 0.f;
      if (!d.discrete) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (part + 4 * j < A) {
            const float diff = r_act[j] - o_[j];
            doutrow[part + 4 * j] = dlogp * diff * c_ivar[j];
            auxrow[part + 4 * j] = valid ? dlogp * (diff * diff * c_ivar[j] - 1.f) - ent_coef * invB : 0.f;
          }
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (part + 4 * j < A) {
            const float l = o_[j] - lse, p = expf(l);
            const float dH = -p * (l + entropy);
            float g = dlogp * ((part + 4 * j == act_i ? 1.f : 0.f) - p);
            g += valid ? -ent_coef * invB * dH : 0.f;
            doutrow[part + 4 * j] = g;
          }
      }
      if (part == 0) {   // loss statistics: staged per row in the misc tile, summed in phase 5
        float* mrow = lds + L::misc + lrow * L::MS;
        mrow[2] = valid ? -fminf(pl1, pl2) : 0.f;                           // policy_gradient_loss
        mrow[3] = valid ? -entropy : 0.f;                                    // entropy_loss
        mrow[4] = valid ? (ratio - 1.f) - log_ratio : 0.f;                   // approx_kl
        mrow[5] = valid ? (fabsf(ratio - 1.f) > clip ? 1.f : 0.f) : 0.f;     // clip_fraction
      }
    } else {
      const float v = lds[L::misc + lrow * L::MS + 0];
      const float verr = r_ret - v;
      lds[L::misc + lrow * L::MS + 1] = valid ? vf_coef * 2.f * (v - r_ret) * invB : 0.f;
      lds[L::misc + lrow * L::MS + 6] = valid ? verr * verr : 0.f;        // value_loss
    }
  }
  __syncthreads();
  IA_TS(5);
  // ---- phase 5: dz2 = d(a2) * (1 - a2^2); head weight / bias gradients
  if (tw == 0) {
    f32x4 acc[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    float av[4], ae[NC][4], ua[16], ub[16];
#pragma unroll
    for (int s = 0; s < 4; ++s) av[s] = s < SA ? lds[L::dout + arow * L::AS + 4 * s + lk] : 0.f;   // columns >= A are zero
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) ae[c][r] = a2t[(q * 16 + lk * 4 + r) * L::HS + CJ(c)];
    if (q < NC) {
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        ua[s] = lds[L::dout + (4 * s + lk) * L::AS + li];
        ub[s] = a2t[(4 * s + lk) * L::HS + q * 16 + li];
      }
    }
    IA_FENCE();
#pragma unroll
    for (int s = 0; s < 4; ++s)
      if (s < SA) {
#pragma unroll
        for (int c = 0; c < NC; ++c) acc[c] = mfma16(av[s], bDa2[s][c], acc[c]);
      }
    if (q < NC) {  // dWa[a][h] = sum_r dout[r][a] a2[r][h], tile of 16 h-columns per wave
      f32x4 g = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 16; ++s) g = mfma16(ua[s], ub[s], g);
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (lk * 4 + r < A) sst(slab + o.aW + (lk * 4 + r) * H + q * 16 + li, g[r]);
    }
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) dzt[(q * 16 + lk * 4 + r) * L::HS + CJ(c)] = acc[c][r] * (1.f - ae[c][r] * ae[c][r]);
    if (q == 2 && lane < A) sst(slab + o.ab + lane, column_sum_b(lds + L::dout + lane, L::AS));
    if (q == 3 && !d.discrete && lane < A) sst(slab + o.log_std + lane, column_sum_b(lds + L::aux + lane, L::AS));
    if (SPLIT && q == 1 && lane < 4)   // this workgroup's loss statistics: misc columns 2..5 -> slots {0 pg, 2 ent, 3 kl, 4 clip}
      sst(statpart + (lane == 0 ? 0 : lane + 1), column_sum_b(lds + L::misc + 2 + lane, L::MS));
  } else {
    float ae[NC][4], dv[4], ua[16], ub[16];
#pragma unroll
    for (int r = 0; r < 4; ++r) dv[r] = lds[L::misc + (q * 16 + lk * 4 + r) * L::MS + 1];
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) ae[c][r] = a2t[(q * 16 + lk * 4 + r) * L::HS + CJ(c)];
    if (q < NC) {
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        ua[s] = lds[L::misc + (4 * s + lk) * L::MS + 1];
        ub[s] = a2t[(4 * s + lk) * L::HS + q * 16 + li];
      }
    }
    IA_FENCE();
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        dzt[(q * 16 + lk * 4 + r) * L::HS + CJ(c)] = cwv[c] * dv[r] * (1.f - ae[c][r] * ae[c][r]);
    if (q < NC) {  // dcW[h] = sum_r dv[r] a2[r][h]  (only output row 0 is meaningful)
      f32x4 g = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 16; ++s) g = mfma16(li == 0 ? ua[s] : 0.f, ub[s], g);
      if (lk == 0) sst(slab + o.cW + q * 16 + li, g[0]);
    }
    if (q == 2 && lane == 0) sst(slab + o.cb, column_sum_b(lds + L::misc + 1, L::MS));
    if (q == 3 && lane < 5 && (!SPLIT || lane == 4)) {  // statpart slots {0 pg, 2 ent, 3 kl, 4 clip, 1 value} <- misc columns 2..6
      const int slot = lane == 0 ? 0 : (lane == 4 ? 1 : lane + 1);   // (SPLIT: the policy workgroup sums its own four columns)
      sst(statpart + slot, column_sum_b(lds + L::misc + 2 + lane, L::MS));
    }
  }
  __syncthreads();
  IA_TS(6);
  // ---- phase 6: dW2 (one 16x16 tile per wave), db2, and dz1 = (dz2 W2) * (1 - a1^2) -> a2 tile
  {
    // NC x NC tiles of 16 x 16 over the tower's four waves, TWO at a time where a wave has several: their operand reads
    // are one block and their (dependent) MFMA chains interleave
    constexpr int NTW = NC * NC / 4, TP = NTW >= 2 ? 2 : 1;
#pragma unroll
    for (int t0 = 0; t0 < NTW; t0 += TP) {
      float ua[TP][16], ub[TP][16];
#pragma unroll
      for (int u = 0; u < TP; ++u) {
        const int ti = q + 4 * (t0 + u), jt = ti / NC, kt = ti % NC;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
          ua[u][s] = dzt[(4 * s + lk) * L::HS + jt * 16 + li];
          ub[u][s] = a1t[(4 * s + lk) * L::HS + kt * 16 + li];
        }
      }
      IA_FENCE();
      f32x4 g[TP];
#pragma unroll
      for (int u = 0; u < TP; ++u) g[u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 16; ++s)
#pragma unroll
        for (int u = 0; u < TP; ++u) g[u] = mfma16(ua[u][s], ub[u][s], g[u]);
#pragma unroll
      for (int u = 0; u < TP; ++u) {
        const int ti = q + 4 * (t0 + u), jt = ti / NC, kt = ti % NC;
#pragma unroll
        for (int r = 0; r < 4; ++r) sst(slab + oW2 + (jt * 16 + lk * 4 + r) * H + kt * 16 + li, g[u][r]);
      }
    }
    f32x4 acc[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    float av[KS], ae[NC][4];
#pragma unroll
    for (int s = 0; s < KS; ++s) av[s] = dzt[arow * L::HS + 4 * s + lk];
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) ae[c][r] = a1t[(q * 16 + lk * 4 + r) * L::HS + CJ(c)];
    IA_FENCE();
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int c = 0; c < NC; ++c) acc[c] = mfma16(av[s], bW2o[s][c], acc[c]);
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r)   // dz1 (the a2 tile is free from here on)
        a2t[(q * 16 + lk * 4 + r) * L::HS + CJ(c)] = acc[c][r] * (1.f - ae[c][r] * ae[c][r]);
    if (q == 3 && lane < H) sst(slab + ob2 + lane, column_sum_b(dzt + lane, L::HS));
  }
  __syncthreads();
  IA_TS(7);
  // ---- phase 7: dW1 tiles (dz1^T x), db1
  {
    const int KT = (D + 15) >> 4;
    for (int ti = q; ti < NC * KT; ti += 4) {
      const int jt = ti / KT, kt = ti - jt * KT;
      f32x4 g = {0.f, 0.f, 0.f, 0.f};
      float ua[16], ub[16];
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        ua[s] = a2t[(4 * s + lk) * L::HS + jt * 16 + li];
        ub[s] = lds[L::x + (4 * s + lk) * L::XS + kt * 16 + li];
      }
      IA_FENCE();
#pragma unroll
      for (int s = 0; s < 16; ++s) g = mfma16(ua[s], ub[s], g);
      const int col = kt * 16 + li;
      if (col < D)
#pragma unroll
        for (int r = 0; r < 4; ++r) sst(slab + oW1 + (jt * 16 + lk * 4 + r) * D + col, g[r]);
    }
    if (q == 3 && lane < H) sst(slab + ob1 + lane, column_sum_b(a2t + lane, L::HS));
  }
  __syncthreads();
  IA_TS(8);
#undef IA_TS
#undef IA_FENCE
}


template <int H>
__global__ __launch_bounds__(512) void ppo_grad_mfma_kernel(
    ia_policy_desc d, const float* __restrict__ P, const float* __restrict__ Pt, const float* __restrict__ nm,
    const float* __restrict__ nv, const float* __restrict__ obs, const float* __restrict__ actions,
    const float* __restrict__ old_logp, const float* __restrict__ adv, const float* __restrict__ ret,
    const int64_t* __restrict__ idx, int batch, int T, int n_envs, int normalize_adv, float clip, float ent_coef,
    float vf_coef, float* __restrict__ ws, int nblk, long long* __restrict__ tstamp, const float* __restrict__ advstat) {
  extern __shared__ float lds[];
  const PolOff o = pol_offsets(d.obs_dim, d.act_dim, H, d.discrete);
  const PpoWs w = ppo_ws(ws, nblk, o.total);
  const MbRows rows{obs, actions, old_logp, adv, ret, idx, batch, T, n_envs};
  if (blockIdx.x == 0 && threadIdx.x == 0) reinterpret_cast<unsigned*>(ws)[7] = 0u;   // ticket of ppo_apply_split_kernel
  const float* as = advstat ? advstat : w.advstat;
  mfma_minibatch<H, true>(d, P, Pt, nm, nv, as[0], as[1], rows, blockIdx.x, normalize_adv, clip, ent_coef,
                         vf_coef, w.slabs + (long long)blockIdx.x * o.total, w.statpart + blockIdx.x * 8, lds, tstamp);
}

__global__ __launch_bounds__(PREP_THREADS) void ppo_apply_kernel(
    ia_policy_desc d, float* __restrict__ P, float* __restrict__ Pt, float* __restrict__ m, float* __restrict__ v,
    float* __restrict__ ws, int nblk, int batch, float max_norm, float ent_coef, float vf_coef, float beta1,
    float beta2, float eps, float step_size, float bc2_sqrt, float* __restrict__ stats,
    // statistics of the NEXT minibatch (next_batch == 0: none)
    const float* __restrict__ obs, const float* __restrict__ adv, const int64_t* __restrict__ next_idx, int next_batch,
    int T, int n_envs, int update_norm, float* __restrict__ nm, float* __restrict__ nv,
    int32_t* __restrict__ ncount, int phases /* bit0: reduce slabs -> ws.grad, bit1: clip + Adam */) {
  extern __shared__ float lds[];
  __shared__ float coef;
  const int H = d.hidden, D = d.obs_dim;
  const PolOff o = pol_offsets(D, d.act_dim, H, d.discrete);
  const PpoWs w = ppo_ws(ws, nblk, o.total);
  const int tid = threadIdx.x;
  // Block 1 (when launched) computes the NEXT minibatch's statistics concurrently with block 0's
  // reduce / clip / Adam: the two halves touch disjoint state (advstat + RunningNorm vs parameters).
  if (blockIdx.x == 1) {
    if (next_batch > 0)
      prepare_stats(d, obs, adv, next_idx, next_batch, T, n_envs, update_norm, nm, nv, ncount, w.advstat, lds);
    return;
  }
  float sq = 0.f;
  for (int i = tid; i < o.total; i += PREP_THREADS) {
    float g;
    if (phases & 1) {
      g = 0.f;
      int b = 0;
      for (; b + 8 <= nblk; b += 8) {  // 8 independent loads in flight, then a fixed-order sum
        float t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = w.slabs[(long long)(b + u) * o.total + i];
#pragma unroll
        for (int u = 0; u < 8; ++u) g += t[u];
      }
      for (; b < nblk; ++b) g += w.slabs[(long long)b * o.total + i];
      w.grad[i] = g;
    } else {
      g = w.grad[i];  // already reduced (and all-reduced across ranks) by the caller
    }
    sq += g * g;
  }
  if (!(phases & 2)) return;
  const float total_sq = block_sum_1024(sq, lds);
  if (tid == 0) {
    const float total_norm = sqrtf(total_sq);
    // torch.nn.utils.clip_grad_norm_: coef = max_norm/(norm+1e-6), clamped to 1
    coef = fminf(max_norm / (total_norm + 1e-6f), 1.0f);
    if (stats) {
      float st[5] = {0, 0, 0, 0, 0};
      for (int b = 0; b < nblk; ++b)
        for (int k = 0; k < 5; ++k) st[k] += w.statpart[b * 8 + k];
      const float invB = 1.f / (float)batch;
      for (int k = 0; k < 5; ++k) st[k] *= invB;
      stats[0] = st[0]; stats[1] = st[1]; stats[2] = st[2]; stats[3] = st[3]; stats[4] = st[4];
      stats[5] = st[0] + ent_coef * st[2] + vf_coef * st[1];  // loss
      stats[6] = total_norm;
      stats[7] = coef;
    }
  }
  __syncthreads();
  const float c = coef;
  for (int i = tid; i < o.total; i += PREP_THREADS) {
    const float g = w.grad[i] * c;
    float mi = m[i];
    mi = mi + (g - mi) * (1.f - beta1);
    const float vi = v[i] * beta2 + (1.f - beta2) * g * g;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    const float pn = P[i] - step_size * (mi / denom);
    P[i] = pn;
    m[i] = mi;
    v[i] = vi;
    int dst = i;
    auto tr = [&](int base, int rows, int cols) {
      if (i >= base && i < base + rows * cols) {
        const int r = (i - base) / cols, cc = (i - base) % cols;
        dst = base + cc * rows + r;
      }
    };
    tr(o.pW1, H, D); tr(o.pW2, H, H); tr(o.vW1, H, D); tr(o.vW2, H, H);
    Pt[dst] = pn;
  }
  if (next_batch > 0 && gridDim.x == 1) {
    __syncthreads();
    prepare_stats(d, obs, adv, next_idx, next_batch, T, n_envs, update_norm, nm, nv, ncount, w.advstat, lds);
  }
}

// The same step (slab reduction, clip_grad_norm_, Adam, transposed shadow copy, loss statistics) spread over `G`
// workgroups: workgroup g reduces its contiguous chunk of the gradient and leaves the chunk's sum of squares; the one
// that draws the last ticket folds the G partial sums in order, then applies clip + Adam to the whole vector (11 k
// parameters at H = 64: a few per thread). The single-workgroup form above walks the slabs of ALL parameters in one
// thread block -- 24 us at H = 64, as long as the gradient kernel itself. Workgroup G (launched when `stats_block`)
// computes the next minibatch's statistics beside it. The ticket word is ws[7] (zeroed by the gradient kernels, reset
// here); partial sums go to the unused slot 5 of the loss-statistic partials.
__global__ __launch_bounds__(PREP_THREADS) void ppo_apply_split_kernel(
    ia_policy_desc d, float* __restrict__ P, float* __restrict__ Pt, float* __restrict__ m, float* __restrict__ v,
    float* __restrict__ ws, int nblk, int batch, float max_norm, float ent_coef, float vf_coef, float beta1,
    float beta2, float eps, float step_size, float bc2_sqrt, float* __restrict__ stats, const float* __restrict__ obs,
    const float* __restrict__ adv, const int64_t* __restrict__ next_idx, int next_batch, int T, int n_envs,
    int update_norm, float* __restrict__ nm, float* __restrict__ nv, int32_t* __restrict__ ncount, int G,
    int stats_block) {
  extern __shared__ float lds[];
  __shared__ float coef;
  __shared__ int is_last;
  const int H = d.hidden, D = d.obs_dim;
  const PolOff o = pol_offsets(D, d.act_dim, H, d.discrete);
  const PpoWs w = ppo_ws(ws, nblk, o.total);
  const int tid = threadIdx.x;
  if ((int)blockIdx.x == G) {
    if (next_batch > 0)
      prepare_stats(d, obs, adv, next_idx, next_batch, T, n_envs, update_norm, nm, nv, ncount, w.advstat, lds);
    return;
  }
  const int chunk = (o.total + G - 1) / G;
  const int i0 = blockIdx.x * chunk, i1 = min(o.total, i0 + chunk);
  float sq = 0.f;
  for (int i = i0 + tid; i < i1; i += PREP_THREADS) {
    float g = 0.f;
    int b = 0;
    for (; b + 8 <= nblk; b += 8) {  // 8 independent loads in flight, then a fixed-order sum (as ppo_apply_kernel)
      float t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = w.slabs[(long long)(b + u) * o.total + i];
#pragma unroll
      for (int u = 0; u < 8; ++u) g += t[u];
    }
    for (; b < nblk; ++b) g += w.slabs[(long long)b * o.total + i];
    w.grad[i] = g;
    sq += g * g;
  }
  // hand-off per the gfx950 rules: every thread's gradient stores acknowledged, block barrier (inside the sum), then
  // one lane's agent-scope release + relaxed ticket; the last workgroup acquires before reading the others' chunks
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const float part = block_sum_1024(sq, lds);
  unsigned* ticket = reinterpret_cast<unsigned*>(ws) + 7;
  if (tid == 0) {
    w.statpart[blockIdx.x * 8 + 5] = part;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned tk = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    is_last = tk == (unsigned)G - 1u;
    if (is_last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
  if (!is_last) return;
  __shared__ float s_norm, s_st[5];
  if (tid == 0) {
    float total_sq = 0.f;
    for (int g = 0; g < G; ++g) total_sq += w.statpart[g * 8 + 5];
    const float total_norm = sqrtf(total_sq);
    s_norm = total_norm;
    coef = fminf(max_norm / (total_norm + 1e-6f), 1.0f);   // torch.nn.utils.clip_grad_norm_
    __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (stats && tid >= 64 && tid < 69) {   // loss statistics: one lane of the second wave per statistic, beside the norm
    const int k = tid - 64;
    float st = 0.f;
    for (int b = 0; b < nblk; ++b) st += w.statpart[b * 8 + k];
    st *= 1.f / (float)batch;
    stats[k] = st;
    s_st[k] = st;
  }
  __syncthreads();
  if (stats && tid == 0) {
    stats[5] = s_st[0] + ent_coef * s_st[2] + vf_coef * s_st[1];  // loss
    stats[6] = s_norm;
    stats[7] = coef;
  }
  const float c = coef;
  // Adam over the whole vector by this one workgroup: the loads of NPT parameters per thread are issued together (a
  // loop with one parameter per trip is a chain of ~11 memory round trips at H = 64: 20 us)
  constexpr int NPT = 12;
  for (int base = 0; base < o.total; base += NPT * PREP_THREADS) {
    float g_[NPT], m_[NPT], v_[NPT], p_[NPT];
#pragma unroll
    for (int k = 0; k < NPT; ++k) {
      const int i = min(base + tid + k * PREP_THREADS, o.total - 1);
      g_[k] = w.grad[i];
      m_[k] = m[i];
      v_[k] = v[i];
      p_[k] = P[i];
    }
#pragma unroll
    for (int k = 0; k < NPT; ++k) {
      const int i = base + tid + k * PREP_THREADS;
      if (i < o.total) {
        const float g = g_[k] * c;
        const float mi = m_[k] + (g - m_[k]) * (1.f - beta1);
        const float vi = v_[k] * beta2 + (1.f - beta2) * g * g;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        const float pn = p_[k] - step_size * (mi / denom);
        P[i] = pn;
        m[i] = mi;
        v[i] = vi;
        int dst = i;
        auto tr = [&](int b0, int rows, int cols) {
          if (i >= b0 && i < b0 + rows * cols) {
            const int r = (i - b0) / cols, cc = (i - b0) % cols;
            dst = b0 + cc * rows + r;
          }
        };
        tr(o.pW1, H, D); tr(o.pW2, H, H); tr(o.vW1, H, D); tr(o.vW2, H, H);
        Pt[dst] = pn;
      }
    }
  }
  if (next_batch > 0 && !stats_block) {
    __syncthreads();
    prepare_stats(d, obs, adv, next_idx, next_batch, T, n_envs, update_norm, nm, nv, ncount, w.advstat, lds);
  }
}

// ---------------------------------------------------------------------------------------------
// Persistent PPO update (H = 32): ALL minibatch steps of a `PPO.train` call in one launch.
//
//   grid = nblk gradient blocks (64 rows each) + 1 statistics block, 512 threads, all co-resident.
//   Parameters live in LDS (sP torch layout, sPt transposed towers) and Adam's m, v in registers of
//   EVERY gradient block: after one grid barrier per step each block reduces the nblk gradient
//   slabs in the same fixed order and applies the same clip + Adam update to its own copy, so no
//   block ever waits for a parameter broadcast. Slabs are double-buffered by step parity: a block
//   can only reach the writes of step s+1 after its reads of step s-1's slabs (program order), and
//   readers of step s finish before anyone passes barrier s+1.
//   The statistics block runs AHEAD of the gradient chain -- advantage mean/std and the train-mode
//   feature RunningNorm update of every minibatch depend on the data only -- and publishes them
//   through a small ring (flag = steps published; back-pressure from the barrier counter).
// Cross-block visibility: writers fence (agent scope) before the counter increment, readers after
// the spin (same pattern as the BCE last-block reduction). Spins are bounded: on timeout an error
// word is set and every block leaves the kernel.
// ---------------------------------------------------------------------------------------------
// Minibatch body of the persistent kernel, re-phased around data locality: wave (tower, q) owns rows
// q*16 .. q*16+15 of the block for the WHOLE forward / loss / backward-activation chain
//     x -> a1 -> a2 -> head -> per-row loss -> d(head) -> dz2 -> dz1
// (every hand-off is an LDS write read back by the same wave: wave-scope ordering, no block barrier),
// then ONE block barrier, then all weight-gradient tiles, bias column sums and loss-statistic sums --
// the only parts that contract over all 64 rows -- run independently per wave. Three block barriers
// per minibatch instead of eight. Parameters are resident in LDS (sP / sPt), rows are staged (stg).
struct CLds {  // LDS carve-up (floats) of the persistent update's minibatch tiles
  // Every tile is TRANSPOSED: [feature / column][row], rows contiguous, row stride RS = 68 floats (272 bytes: 16-byte
  // aligned, = 4 mod 32 banks). The chain's lanes (lk, li) -- features 16 t + 4 lk + r of row li -- then write a tile with
  // 16 consecutive rows per lane group 16 banks apart (conflict-free), and the weight-gradient tiles, which contract over
  // ROWS, read four consecutive rows of a feature in ONE ds_read_b128 (row steps permuted: step s of lane group lk is row
  // 16 (s >> 2) + 4 lk + (s & 3)) -- 8 wide reads per 16 x 16 x 64 tile instead of 32 two-way conflicted ds_read_b32.
  static constexpr int RS = 68;
  static constexpr int x = 0;                         // [MAXD][RS] normalised observations
  static constexpr int a1 = x + MAXD * RS;            // [2 towers][32][RS]
  static constexpr int a2 = a1 + 2 * 32 * RS;
  static constexpr int dz2 = a2 + 2 * 32 * RS;
  static constexpr int dz1 = dz2 + 2 * 32 * RS;
  static constexpr int dout = dz1 + 2 * 32 * RS;      // [MAXA][RS] d loss / d head output
  static constexpr int aux = dout + MAXA * RS;        // [MAXA][RS] log_std gradient terms
  static constexpr int misc = aux + MAXA * RS;        // [8][RS]: 1 dvalue, 2..6 loss statistics
  static constexpr int scratch = misc + 8 * RS;       // 64 floats: block reductions
  static constexpr int total = scratch + 64;
};

__device__ __forceinline__ void wave_sync_lds() {
  // same-wave LDS hand-off: DS operations of one wave execute in issue order; this only stops the
  // compiler from moving the reads above the writes
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// Rows of the minibatch (raw, packed, prefetched into the staging area) -> normalised rows in the x tile; this wave's rows
// of the small tiles cleared. Wave (tower tw, quarter q) and its twin of the other tower share the quarter's 16 rows.
// Depends on the minibatch and its statistics only -- NOT on the parameters -- so the persistent kernel runs it for the
// next minibatch while this minibatch's sum vector is still on its way (`STAGED` chain).
// The transposed towers' LDS image of the persistent update (`sPt`): rows TSTR = 36 floats apart instead of H = 32, every
// matrix at `upd_tbase(its offset in the parameter vector)` (36/32 of it: a matrix of n = 32 c floats grows to 36 c, so the
// scaled intervals stay disjoint). Why: Adam writes parameter i of thread tid + 512 k to the transposed copy as well; with
// rows 32 apart consecutive lanes (consecutive input columns of one output row) hit ONE bank -- a 32-way conflict on W2,
// 17-way on W1 at 17 observation columns: ~2 600 LDS cycles per step = 1.1 of the Adam phase's 1.3 us in every gradient
// workgroup. 36 = 4 (mod 32): the writes spread over 8 banks (4-way: 2x a conflict-free ds_write_b32, guide's LDS table),
// and the chain's fragment reads (lane group lk = rows 4 lk + r: 4 x 36 = 16 mod 32) become conflict-free (2-way before).
#ifndef IA_UPD_TSTR
#define IA_UPD_TSTR 36   // (32: the unpadded image, for same-box A/Bs -- tools/ab_libs.sh)
#endif
constexpr int UPD_TSTR = IA_UPD_TSTR;
__host__ __device__ inline int upd_tbase(int off) { return (off * UPD_TSTR + 31) >> 5; }
__host__ __device__ inline int upd_pt4(int P4) { return (upd_tbase(P4) + 3) & ~3; }   // floats of the padded image

template <bool SMALL>
__device__ __forceinline__ void chain_stage_rows(const ia_policy_desc& d, const float* __restrict__ nm,
                                                 const float* __restrict__ nv, const int i0, const int row_lim,
                                                 float* __restrict__ lds, const float* __restrict__ stg, const int tw,
                                                 const int q, const int lane) {
  using L = CLds;
  const int D = d.obs_dim;
  const int S1 = (D + 3) >> 2;
#ifndef IA_SMALL_STAGE_ASIDE
#define IA_SMALL_STAGE_ASIDE 1   // (0: the chain waves stage their rows themselves, for same-box A/Bs -- tools/ab_libs.sh)
#endif
  // SMALL (minibatches of <= 16 rows: only the waves of row quarter 0 run the chain): the rows are staged by the waves of row
  // quarter 1 -- idle otherwise -- WHILE the chain waves request their weight fragments; both meet at the caller's barrier.
  // Same values into the same places (~1 100 clocks of the chain waves' path per step on the tuned AIRL file).
  constexpr int QS = (SMALL && IA_SMALL_STAGE_ASIDE) ? 1 : 0;
  const bool idle = SMALL && q != QS;   // wave-uniform; compiled out of the general instantiations
  if (!idle) {
    const int rbase = SMALL ? 0 : q * 16;
    // the two waves that share q (tower 0 / tower 1) stage the 16 rows together, four consecutive columns per lane
    // and only the 4 * S1 columns the first layer reads with non-zero weights (the caller cleared the tile once: the
    // columns beyond are never written). Group g = row * S1 + column group; the row comes from a multiply-shift
    // (S1 <= 16, g < 256). All 12 loads of a group (raw value, mean, 1 / std) are in flight together.
    const int l128 = lane + 64 * tw;
    const int s1r = (65536 + S1 - 1) / S1;   // wave-uniform
    for (int g = l128; g < 16 * S1; g += 128) {
      const int r = (g * s1r) >> 16, k0 = (g - r * S1) * 4;
      const bool rok = (i0 + rbase + r) < row_lim;
      float raw[4], mu[4], vr[4];
      {   // rows staged 4 S1 floats apart (16-byte pieces, as the LDS-direct loads left them): group g's piece is slot
          // rbase S1 + g of the area -- ONE ds_read_b128 each for the raw values, the means and 1 / std
        const f32x4 rq = *reinterpret_cast<const f32x4*>(stg + UpdStage::x + (rbase * S1 + g) * 4);
        const f32x4 mq = *reinterpret_cast<const f32x4*>(nm + k0);
        const f32x4 vq = *reinterpret_cast<const f32x4*>(nv + k0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          raw[j] = rq[j];   // (columns >= D hold the next row's first values: masked below)
          mu[j] = mq[j];
          vr[j] = vq[j];
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bool ok = rok && k0 + j < D;
        const bool nrm = ok && d.has_norm;          // (`nv` carries 1/sqrt(var + eps), see the statistics block)
        lds[L::x + (k0 + j) * L::RS + rbase + r] = ok ? (nrm ? (raw[j] - mu[j]) * vr[j] : raw[j]) : 0.f;
      }
    }
    if (tw == 0) {
      for (int e = lane; e < 16 * MAXA; e += 64) {   // (e >> 4: action, e & 15: row of this wave)
        lds[L::dout + (e >> 4) * L::RS + rbase + (e & 15)] = 0.f;
        lds[L::aux + (e >> 4) * L::RS + rbase + (e & 15)] = 0.f;
      }
    } else {
      for (int e = lane; e < 16 * 8; e += 64) lds[L::misc + (e >> 4) * L::RS + rbase + (e & 15)] = 0.f;
    }
  }
}

// (`nv` = 1/sqrt(running_var + eps) per column, as the statistics block publishes it)
// KS1 = k-steps (of 4 input columns) the first layer's fragments are sized for: 16 covers MAXD = 64 columns, 8 covers
// observation widths <= 32 -- every reference environment of the path -- with 16 fragment registers less across the chain.
// LOCAL (one gradient workgroup: minibatches <= 64 rows, both tuned configurations of the reference): the gradient
// "slab" is an LDS image -- the part of the row staging area this minibatch has consumed by the time the gradient tiles
// are formed -- that the caller reads straight back: no write-through stores to acknowledge, no release fence, no
// grid wait, no re-read through L2. The loss-statistic partials (for the statistics workgroup) are stored write-through;
// the caller drains them (`vmcnt(0)` in every wave) at the END of the step, where it waits for the prefetched rows
// anyway, and arrives after that -- no release fence (cdna_hip_programming.md G16 R1), no acknowledgement on the chain.
template <int KS1, bool LOCAL, bool SMALL = false, bool STAGED = false>
__device__ __forceinline__ void mfma32_minibatch_chain(
    const ia_policy_desc& d, const float* __restrict__ nm, const float* __restrict__ nv, const float adv_mean,
    const float adv_std, const MbRows rows, const int i0_in, const int row_lim, const int normalize_adv, const float clip,
    const float ent_coef, const float vf_coef, float* __restrict__ slab_g, float* __restrict__ statpart,
    float* __restrict__ lds_in, const float* __restrict__ sP_in, float* __restrict__ stg_in,
    const int opaque_zero, long long* __restrict__ tstamp, const unsigned ll_seq, const bool ll_xcd = false) {
  // `ll_xcd` (wave-uniform): every gradient workgroup of the launch sits on this XCD (the kernel has checked): the slab
  // words are stored at workgroup scope -- they stay in the XCD's L2 instead of costing a fabric write each.
  // `i0_in`: first minibatch row of this workgroup; `row_lim`: rows at or beyond it are not this workgroup's (the
  // minibatch size, or -- row-sharded data parallelism -- the end of this rank's row range inside the global minibatch);
  // means are over `rows.batch`, the whole (global) minibatch, either way
  float* __restrict__ lds = lds_in + opaque_zero;
  const float* __restrict__ stg = stg_in + opaque_zero;
  // Several gradient workgroups: `slab_g` is this workgroup's slab of 8-byte (value, sequence) words -- a naturally aligned
  // 8-byte store arrives whole, so a reader needs no flag, the writer no acknowledgement, fence or barrier: the words are
  // fire-and-forget and the consumers spin on the sequence number they carry (`ll_seq`: this step's).
  float* __restrict__ slab = LOCAL ? stg_in + opaque_zero + UpdStage::x : slab_g;
  unsigned long long* __restrict__ slab64 = reinterpret_cast<unsigned long long*>(slab_g);
  auto put = [&](float* p, float v) {
    if constexpr (LOCAL) *p = v;
    else if (ll_xcd) ll_store_xcd(slab64 + (p - slab), v, ll_seq);
    else ll_store_agent(slab64 + (p - slab), v, ll_seq);
  };
  const int batch = rows.batch;
  constexpr int H = 32;
  using L = CLds;
#define IA_TS(slot) do { if (tstamp && i0_in == 0 && threadIdx.x == 0) tstamp[slot] = clock64(); } while (0)
  const int tid = threadIdx.x + opaque_zero, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  // wave w: tower w & 1, row quarter w >> 1 -- waves 0 and 1 (the two towers of rows 0..15) sit on different SIMDs, so a
  // minibatch of <= 16 rows (the reference's tuned AIRL configuration) has a SIMD per tower to itself once the other six
  // waves skip the layer chain (`idle` below); this numbering only then:
  // (SMALL: an instantiation of its own, chosen by the host for minibatches of <= 16 rows. In the general kernel the
  // uniform branch around the chain alone cost ~1 us per step at config P -- measured --, and full workgroups want the
  // OTHER numbering: SIMD k then hosts (policy, k) and (value, k), complementary work, instead of two waves of one tower
  // in lockstep)
  const int tw = SMALL ? (wv & 1) : (wv >> 2), q = SMALL ? (wv >> 1) : (wv & 3);
  const int li = lane & 15, lk = lane >> 4;
  const int D = d.obs_dim, A = d.act_dim;
  const PolOff o = pol_offsets(D, A, H, d.discrete);
  const int i0 = i0_in;
  const int aw = d.discrete ? 1 : A;
  const float invB = 1.f / (float)batch;
  const int S1 = (D + 3) >> 2;
  const int oW1 = tw ? o.vW1 : o.pW1, ob1 = tw ? o.vb1 : o.pb1, oW2 = tw ? o.vW2 : o.pW2, ob2 = tw ? o.vb2 : o.pb2;
  const float* __restrict__ sP = sP_in + opaque_zero;
  const float* __restrict__ sPt = sP + ((o.total + 3) & ~3);   // padded image: rows UPD_TSTR apart, bases upd_tbase(.)
  constexpr int TS_ = UPD_TSTR;
  const int tW1 = upd_tbase(oW1), tW2 = upd_tbase(oW2);
  IA_TS(0);

  // ---- per-row scalars of the loss. The chain below runs TRANSPOSED -- features along the MFMA's M index, the wave's 16
  // rows along N -- so a lane (lk, li) = (lane >> 4, lane & 15) ends every layer holding features 16 t + 4 lk + r
  // (r = 0..3) of row li: four lanes per row, lane group lk owning actions 4 lk .. 4 lk + 3 in the loss phase (sums over
  // actions finish with two cross-group shuffles). Value waves: every lane group carries the row's value; group 0 writes.
  const int lrow = q * 16 + (lane & 15);   // local row of this lane
  const bool valid = (i0 + lrow) < row_lim;
  float r_oldlp = 0.f, r_adv = 0.f, r_ret = 0.f, r_act[4] = {0.f, 0.f, 0.f, 0.f};
  // Round 6, minibatches of <= 16 rows (SMALL: the value wave has a SIMD to itself and reaches the barrier below early): what
  // the policy wave's loss needs but the towers do not -- the rows' NORMALISED advantages (one IEEE division per row) and the
  // per-action Gaussian constants (exp, IEEE division, log of `log_std`) -- is formed by the VALUE wave and handed over through
  // LDS (the staged advantages in place; the constants in the upper half of the block-reduction scratch, free during the
  // chain): same operations on the same inputs, bit for bit (`tools/ppo_bits.py`), ~70 instructions and their dependent
  // latencies off the policy wave's path: 7.53 -> 7.40 us per step on the tuned AIRL file. With full workgroups the two towers
  // of a row quarter SHARE a SIMD and the hand-over measures 2-4 % slower (`profiles/r06_ppo_ab.md`): they keep the old split.
  constexpr bool HANDOVER = SMALL;
  if (tw == 0) {   // staged by the prefetch (LDS-direct loads); unconditional, clamped
    r_oldlp = stg[UpdStage::oldlp + lrow];
    if constexpr (!HANDOVER) r_adv = stg[UpdStage::adv + lrow];
#pragma unroll
    for (int j = 0; j < 4; ++j) r_act[j] = stg[UpdStage::act + lrow * aw + min(4 * (lane >> 4) + j, aw - 1)];
  } else {
    r_ret = stg[UpdStage::ret + lrow];
    if constexpr (HANDOVER) r_adv = stg[UpdStage::adv + lrow];
  }
  float* gconst = lds + L::scratch + 32;   // (HANDOVER) [16] 1 / sd^2, [16] log sd, of actions 0..15

  IA_TS(9);
  // (SMALL: rows 16.. do not exist in any minibatch of the launch; the caller zeroed every tile once, their waves idle)
  const bool idle = SMALL && q > 0;   // wave-uniform; compiled out of the general instantiations
  // ---- stage this wave's 16 feature rows (normalised) into the x tile; clear its rows of the small tiles (STAGED: the
  // caller did that already -- several gradient workgroups stage the NEXT minibatch while the sum vector is in flight)
  if constexpr (!STAGED) chain_stage_rows<SMALL>(d, nm, nv, i0, row_lim, lds, stg, tw, q, lane);
  IA_TS(10);
  // Weight fragments (LDS -> VGPR) as the MFMAs' A operand: A[m = li][k = lk] of k-step (kt, r) is W[out 16 t + li][in
  // 16 kt + 4 lk + r] -- the k index of a step is PERMUTED (a step takes inputs 4 lk + r, lk = 0..3, of K tile kt) so
  // that the accumulator of one layer (lane (lk, li) holds outputs 4 lk + r of row li) IS the B operand of the next:
  // B[k = lk][n = li] of step (t, r) = output 16 t + 4 lk + r. Activations never leave the registers on the way
  // x -> a1 -> a2 -> head -> loss -> d head -> dz2 -> dz1; the LDS tiles ([row][feature], as before) are written on the
  // side for the weight-gradient tiles, which contract over rows. (Before: rows along M, one LDS write + wave sync + read
  // between every two layers -- seven hand-offs of ~250 clocks on the chain.)
  // ALL reads first -- unconditional, at clamped addresses -- then the masks: a guarded read (or a select right behind
  // its read) costs a branch and a full `lgkmcnt(0)` wait each.
  constexpr int KT1 = KS1 / 4;   // K tiles (16 input columns each) the first layer's fragments are sized for
  float fW1[KT1][2][4], fW2[2][2][4], fW2T[2][2][4], fHead[2][4], fHeadT[2][4], b1c[2][4], b2c[2][4], hb[4];
  // (biases: initial values of the layers' accumulators -- C layout, output 16 t + 4 lk + r -- dead once the layer starts)
#pragma unroll
  for (int kt = 0; kt < KT1; ++kt) {
    if (4 * kt < S1) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) fW1[kt][t][r] = sPt[tW1 + min(16 * kt + 4 * lk + r, D - 1) * TS_ + 16 * t + li];
    } else {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) fW1[kt][t][r] = 0.f;
    }
  }
  const int head_base = tw == 0 ? o.aW + min(li, A - 1) * H : o.cW;
  // (the fragments of layer 2 and of the head are requested behind the previous layer's MFMAs -- their LDS latency
  //  passes under that layer's tanh -- instead of here: 40 registers less across the first layer)
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) b1c[t][r] = sP[ob1 + 16 * t + 4 * lk + r];
  float my_sd = sP[o.log_std + min(lane, A - 1)];
  __builtin_amdgcn_sched_barrier(0);   // every read above is issued before the first value is touched
#pragma unroll
  for (int kt = 0; kt < KT1; ++kt)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) fW1[kt][t][r] = (16 * kt + 4 * lk + r < D) ? fW1[kt][t][r] : 0.f;
  float c_ivar[4] = {1.f, 1.f, 1.f, 1.f}, c_logsd[4] = {0.f, 0.f, 0.f, 0.f};   // of this lane's actions 4 lk + j (policy waves; read behind the barrier)
  if constexpr (HANDOVER) {
    if (tw != 0) {
      if (!idle) {   // this wave's 16 rows: the advantage as the loss uses it
        float advn = r_adv;
        if (normalize_adv && batch > 1) advn = (advn - adv_mean) / (adv_std + 1e-8f);
        if (lk == 0) stg_in[opaque_zero + UpdStage::adv + lrow] = advn;
      }
      if (q == 0 && !d.discrete) {
        const float sd = expf(my_sd);
        const float my_ivar = lane < A ? 1.f / (sd * sd) : 1.f;
        const float my_logsd = lane < A ? logf(sd) : 0.f;
        if (lane < 16) {
          gconst[lane] = my_ivar;
          gconst[16 + lane] = my_logsd;
        }
      }
    }
  } else {
    // per-action Gaussian constants; the reciprocal variance turns the ~3 IEEE divisions per action and row of the loss into
    // multiplications (<= 1 ulp away from dividing). Lane a computes action a's pair once (exp, division, log); every lane
    // then picks its four actions' pairs up from the wave.
    float my_ivar = 1.f, my_logsd = 0.f;
    if (tw == 0 && !d.discrete) {
      const float sd = expf(my_sd);
      my_ivar = lane < A ? 1.f / (sd * sd) : 1.f;
      my_logsd = lane < A ? logf(sd) : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      c_ivar[j] = __shfl(my_ivar, 4 * lk + j, 64);
      c_logsd[j] = __shfl(my_logsd, 4 * lk + j, 64);
    }
  }
  IA_TS(11);
  // the x rows of this wave were written by the two waves (tower 0 / tower 1) that share q
  __syncthreads();
  IA_TS(1);

  float* a1t = lds + L::a1 + tw * 32 * L::RS;
  float* a2t = lds + L::a2 + tw * 32 * L::RS;
  float* dz2t = lds + L::dz2 + tw * 32 * L::RS;
  float* dz1t = lds + L::dz1 + tw * 32 * L::RS;
  const int trow = 4 * lk * L::RS + q * 16 + li;   // feature 16 t + 4 lk + r of this lane's row: trow + (16 t + r) * RS
  // sum / max over the four lane groups of a row (the same bits in all four lanes): v_permlane16_swap / v_permlane32_swap
  // with both operands the same register leave (even rows | odd rows) resp. (lower half | upper half) of it in both
  // halves of the pair -- one VALU exchange per level instead of a ds_bpermute round trip
  auto xchg16 = [](float v, float& a, float& b) {
    const auto p2 = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    a = __uint_as_float(p2[0]);
    b = __uint_as_float(p2[1]);
  };
  auto xchg32 = [](float v, float& a, float& b) {
    const auto p2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    a = __uint_as_float(p2[0]);
    b = __uint_as_float(p2[1]);
  };
  auto group_sum = [&](float v) {
    float a, b;
    xchg16(v, a, b);
    v = a + b;
    xchg32(v, a, b);
    return a + b;
  };
  auto group_max = [&](float v) {
    float a, b;
    xchg16(v, a, b);
    v = fmaxf(a, b);
    xchg32(v, a, b);
    return fmaxf(a, b);
  };
  // ---- a1^T = tanh(W1 x^T + b1): B operand = the wave's x rows, column 16 kt + 4 lk + r of row li
  // (<= 16 rows in every minibatch of the launch: the waves of rows 16.. have nothing to run -- the weight-gradient tiles
  //  below contract over rows 0..15 only, and the column sums over 64 rows find the zeros the caller left in the tiles)
  if (!idle) {
  f32x4 a1[2], a2[2];
  {
    f32x4 acc[2][2] = {{{b1c[0][0], b1c[0][1], b1c[0][2], b1c[0][3]}, {0.f, 0.f, 0.f, 0.f}},
                       {{b1c[1][0], b1c[1][1], b1c[1][2], b1c[1][3]}, {0.f, 0.f, 0.f, 0.f}}};
    float xb[KT1][4];
#pragma unroll
    for (int kt = 0; kt < KT1; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) xb[kt][r] = lds[L::x + min(16 * kt + 4 * lk + r, MAXD - 1) * L::RS + q * 16 + li];
    if (HANDOVER && tw == 0) {   // the value wave's hand-over (landed long before the loss phase reads the registers)
      r_adv = stg_in[opaque_zero + UpdStage::adv + lrow];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        c_ivar[j] = gconst[4 * lk + j];
        c_logsd[j] = gconst[16 + 4 * lk + j];
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kt = 0; kt < KT1; ++kt)
      if (4 * kt < S1) {   // (wave-uniform; K tiles accumulate alternately into two chains per output tile)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          acc[0][kt & 1] = mfma16(fW1[kt][0][r], xb[kt][r], acc[0][kt & 1]);
          acc[1][kt & 1] = mfma16(fW1[kt][1][r], xb[kt][r], acc[1][kt & 1]);
        }
      }
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)   // layer 2's fragments and bias: in flight under the tanh below
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int t = 0; t < 2; ++t) fW2[kt][t][r] = sPt[tW2 + (16 * kt + 4 * lk + r) * TS_ + 16 * t + li];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) b2c[t][r] = sP[ob2 + 16 * t + 4 * lk + r];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        a1[t][r] = fast_tanh(acc[t][0][r] + acc[t][1][r]);
        a1t[trow + (16 * t + r) * L::RS] = a1[t][r];
      }
  }
  IA_TS(2);
  // ---- a2^T = tanh(W2 a1^T + b2)
  {
    f32x4 acc[2][2] = {{{b2c[0][0], b2c[0][1], b2c[0][2], b2c[0][3]}, {0.f, 0.f, 0.f, 0.f}},
                       {{b2c[1][0], b2c[1][1], b2c[1][2], b2c[1][3]}, {0.f, 0.f, 0.f, 0.f}}};
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        acc[0][kt] = mfma16(fW2[kt][0][r], a1[kt][r], acc[0][kt]);
        acc[1][kt] = mfma16(fW2[kt][1][r], a1[kt][r], acc[1][kt]);
      }
    {   // the head's fragments and bias: in flight under the tanh below
      const bool head_on = tw == 0 ? li < A : li == 0;
      float hraw[2][4], braw[4];
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) hraw[kt][r] = sP[head_base + 16 * kt + 4 * lk + r];
#pragma unroll
      for (int r = 0; r < 4; ++r) braw[r] = sP[tw == 0 ? o.ab + min(4 * lk + r, A - 1) : o.cb];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) fHead[kt][r] = head_on ? hraw[kt][r] : 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) hb[r] = (tw == 0 && 4 * lk + r >= A) ? 0.f : braw[r];
    }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        a2[t][r] = fast_tanh(acc[t][0][r] + acc[t][1][r]);
        a2t[trow + (16 * t + r) * L::RS] = a2[t][r];
      }
  }
  IA_TS(3);
  // ---- heads (policy: hout[r] = action_net output 4 lk + r of row li; value: hout[0] of lane group 0 = value_net output)
  float hout[4];
  {
    f32x4 acc[2] = {{hb[0], hb[1], hb[2], hb[3]}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[kt] = mfma16(fHead[kt][r], a2[kt][r], acc[kt]);
#pragma unroll
    for (int r = 0; r < 4; ++r) hout[r] = acc[0][r] + acc[1][r];
  }
  IA_TS(4);
  // backward weight fragments (W2 in torch orientation, action_net rows as the A operand's k index): requested here,
  // behind the forward pass (24 registers less across it), and landed by the time the loss phase below is through
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) fW2T[kt][t][r] = sP[oW2 + (16 * kt + 4 * lk + r) * H + 16 * t + li];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r)   // (value waves: value_net's weights of outputs 16 t + 4 lk + r in the same registers)
      fHeadT[t][r] = sP[tw == 0 ? o.aW + min(4 * lk + r, A - 1) * H + 16 * t + li : o.cW + 16 * t + 4 * lk + r];
  // ---- per-row losses of this wave's 16 rows; dout[r] = d loss / d head output 4 lk + r of row li (registers: the B
  // operand of the backward pass; the LDS copy is for the head's weight gradient)
  float dout[4] = {0.f, 0.f, 0.f, 0.f}, dvb = 0.f;
  if (tw == 0) {
    float logp = 0.f, entropy = 0.f, lse = 0.f;
    int act_i = 0;
    if (d.discrete) act_i = (int)r_act[0];
    IA_TS(12);
    if (!d.discrete) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (4 * lk + j < A) {
          const float diff = r_act[j] - hout[j];
          logp += -(diff * diff) * (0.5f * c_ivar[j]) - c_logsd[j] - LOG_SQRT_2PI;
          entropy += 0.5f + LOG_SQRT_2PI + c_logsd[j];
        }
      logp = group_sum(logp);
      entropy = group_sum(entropy);   // (the same for every row; two VALU exchanges)
    } else {
      float mx = -3.0e38f, o_act = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (4 * lk + j < A) {
          mx = fmaxf(mx, hout[j]);
          o_act += (4 * lk + j == act_i) ? hout[j] : 0.f;
        }
      mx = group_max(mx);
      o_act = group_sum(o_act);
      float se = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (4 * lk + j < A) se += expf(hout[j] - mx);
      lse = mx + logf(group_sum(se));
      logp = o_act - lse;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (4 * lk + j < A) {
          const float l = hout[j] - lse;
          entropy -= expf(l) * l;
        }
      entropy = group_sum(entropy);
    }
    IA_TS(13);
    float advn = r_adv;   // (HANDOVER: normalised by the value wave of these rows, see the top of the chain)
    if constexpr (!HANDOVER)
      if (normalize_adv && batch > 1) advn = (advn - adv_mean) / (adv_std + 1e-8f);
    const float log_ratio = logp - r_oldlp;
    const float ratio = expf(log_ratio);
    const float lo = 1.f - clip, hi = 1.f + clip;
    const float pl1 = advn * ratio;
    const float pl2 = advn * fminf(fmaxf(ratio, lo), hi);
    const float g1 = pl1 < pl2 ? 1.f : (pl1 == pl2 ? 0.5f : 0.f);
    const float g2 = pl2 < pl1 ? 1.f : (pl1 == pl2 ? 0.5f : 0.f);
    const float inrange = (ratio >= lo && ratio <= hi) ? 1.f : 0.f;
    const float dlogp = valid ? -invB * advn * (g1 + g2 * inrange) * ratio : 0.f;
    IA_TS(14);
    float* doutrow = lds + L::dout + lrow;   // (column a of this lane's row: [a * RS])
    float* auxrow = lds + L::aux + lrow;
    if (!d.discrete) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (4 * lk + j < A) {
          const float diff = r_act[j] - hout[j];
          dout[j] = dlogp * diff * c_ivar[j];
          doutrow[(4 * lk + j) * L::RS] = dout[j];
          auxrow[(4 * lk + j) * L::RS] = valid ? dlogp * (diff * diff * c_ivar[j] - 1.f) - ent_coef * invB : 0.f;
        }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (4 * lk + j < A) {
          const float l = hout[j] - lse, p = expf(l);
          const float dH = -p * (l + entropy);
          float g = dlogp * ((4 * lk + j == act_i ? 1.f : 0.f) - p);
          g += valid ? -ent_coef * invB * dH : 0.f;
          dout[j] = g;
          doutrow[(4 * lk + j) * L::RS] = g;
        }
    }
    if (lk == 0) {
      float* mrow = lds + L::misc + lrow;
      mrow[2 * L::RS] = valid ? -fminf(pl1, pl2) : 0.f;                           // policy_gradient_loss
      mrow[3 * L::RS] = valid ? -entropy : 0.f;                                    // entropy_loss
      mrow[4 * L::RS] = valid ? (ratio - 1.f) - log_ratio : 0.f;                   // approx_kl
      mrow[5 * L::RS] = valid ? (fabsf(ratio - 1.f) > clip ? 1.f : 0.f) : 0.f;     // clip_fraction
    }
    IA_TS(15);
  } else {
    // value_net's only output row is m = 0: lane group 0, register 0 holds V(row li); every group needs its gradient
    const float v = __shfl(hout[0], li, 64);
    const float verr = r_ret - v;
    dvb = valid ? vf_coef * 2.f * (v - r_ret) * invB : 0.f;
    if (lk == 0) {
      lds[L::misc + 1 * L::RS + lrow] = dvb;
      lds[L::misc + 6 * L::RS + lrow] = valid ? verr * verr : 0.f;        // value_loss
    }
  }
  IA_TS(5);
  // ---- dz2^T = (W_head^T d head^T) * (1 - a2^2) for this wave's rows
  f32x4 dz2[2];
  if (tw == 0) {
    f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (r < A) {   // (k-step r carries actions r, 4 + r, 8 + r, 12 + r; rows of actions >= A are masked)
        const bool on = 4 * lk + r < A;
        acc[0] = mfma16(on ? fHeadT[0][r] : 0.f, dout[r], acc[0]);
        acc[1] = mfma16(on ? fHeadT[1][r] : 0.f, dout[r], acc[1]);
      }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) dz2[t][r] = acc[t][r] * (1.f - a2[t][r] * a2[t][r]);
  } else {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) dz2[t][r] = fHeadT[t][r] * dvb * (1.f - a2[t][r] * a2[t][r]);
  }
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) dz2t[trow + (16 * t + r) * L::RS] = dz2[t][r];
  // ---- dz1^T = (W2^T dz2^T) * (1 - a1^2) for this wave's rows
  {
    f32x4 acc[2][2] = {{{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}};
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        acc[0][kt] = mfma16(fW2T[kt][0][r], dz2[kt][r], acc[0][kt]);
        acc[1][kt] = mfma16(fW2T[kt][1][r], dz2[kt][r], acc[1][kt]);
      }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        dz1t[trow + (16 * t + r) * L::RS] = (acc[t][0][r] + acc[t][1][r]) * (1.f - a1[t][r] * a1[t][r]);
  }
  }   // (!idle)
  __syncthreads();   // every row's activations and activation gradients are in LDS
  IA_TS(6);

  // ---- gradient tiles: contractions over all 64 rows, independent per wave. Every tile requests its
  // 32 LDS operands first and then runs its 16 dependent MFMAs (the compiler otherwise pairs each
  // MFMA with its two reads and exposes an LDS round trip per step).
  // (minibatches of <= 16 rows -- the reference's tuned AIRL configuration -- contract over the first four row steps
  //  only: the gradient rows past the minibatch are exact zeros, so the remaining twelve steps add nothing)
  const bool few_rows = row_lim - i0 <= 16;   // wave-uniform
  // A tile = 16 features of U (the MFMA's M index) x 16 features of V (N), contracted over the rows (K). A lane reads
  // four consecutive rows of its feature per ds_read_b128: row steps 4 sq .. 4 sq + 3 of lane group lk are rows
  // 16 sq + 4 lk .. + 3 -- the same permutation on both operands. Two accumulator chains per tile (a chain of sixteen
  // dependent MFMAs is 16 x 40 clocks of latency against 16 x 32 of issue); `few_rows`: rows 0..15 only.
  auto rd4 = [&](const float* p) { return *reinterpret_cast<const f32x4*>(p); };
  auto outer16 = [&](const float* __restrict__ U, int ufeat, const float* __restrict__ V, int vfeat) {
    const float* up = U + ufeat * L::RS + 4 * lk;
    const float* vp = V + vfeat * L::RS + 4 * lk;
    f32x4 g = {0.f, 0.f, 0.f, 0.f}, g2 = {0.f, 0.f, 0.f, 0.f};
    if (few_rows) {
      const f32x4 u = rd4(up), v = rd4(vp);
      __builtin_amdgcn_sched_barrier(0);
      g = mfma16(u[0], v[0], g);
      g2 = mfma16(u[1], v[1], g2);
      g = mfma16(u[2], v[2], g);
      g2 = mfma16(u[3], v[3], g2);
      return g + g2;
    }
    f32x4 u[4], v[4];
#pragma unroll
    for (int sq = 0; sq < 4; ++sq) {
      u[sq] = rd4(up + 16 * sq);
      v[sq] = rd4(vp + 16 * sq);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int sq = 0; sq < 4; ++sq) {
      g = mfma16(u[sq][0], v[sq][0], g);
      g2 = mfma16(u[sq][1], v[sq][1], g2);
      g = mfma16(u[sq][2], v[sq][2], g);
      g2 = mfma16(u[sq][3], v[sq][3], g2);
    }
    return g + g2;
  };
  // two tiles at once: all operands requested first, then four interleaved chains
  auto outer16_pair = [&](const float* __restrict__ U0, int uf0, const float* __restrict__ V0, int vf0,
                          const float* __restrict__ U1, int uf1, const float* __restrict__ V1, int vf1, f32x4& r0, f32x4& r1) {
    const float* up0 = U0 + uf0 * L::RS + 4 * lk;
    const float* vp0 = V0 + vf0 * L::RS + 4 * lk;
    const float* up1 = U1 + uf1 * L::RS + 4 * lk;
    const float* vp1 = V1 + vf1 * L::RS + 4 * lk;
    f32x4 g0 = {0.f, 0.f, 0.f, 0.f}, g0b = {0.f, 0.f, 0.f, 0.f}, g1 = {0.f, 0.f, 0.f, 0.f}, g1b = {0.f, 0.f, 0.f, 0.f};
    if (few_rows) {
      const f32x4 u0 = rd4(up0), v0 = rd4(vp0), u1 = rd4(up1), v1 = rd4(vp1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 4; i += 2) {
        g0 = mfma16(u0[i], v0[i], g0);
        g1 = mfma16(u1[i], v1[i], g1);
        g0b = mfma16(u0[i + 1], v0[i + 1], g0b);
        g1b = mfma16(u1[i + 1], v1[i + 1], g1b);
      }
    } else {
      f32x4 u0[4], v0[4], u1[4], v1[4];
#pragma unroll
      for (int sq = 0; sq < 4; ++sq) {
        u0[sq] = rd4(up0 + 16 * sq);
        v0[sq] = rd4(vp0 + 16 * sq);
        u1[sq] = rd4(up1 + 16 * sq);
        v1[sq] = rd4(vp1 + 16 * sq);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int sq = 0; sq < 4; ++sq)
#pragma unroll
        for (int i = 0; i < 4; i += 2) {
          g0 = mfma16(u0[sq][i], v0[sq][i], g0);
          g1 = mfma16(u1[sq][i], v1[sq][i], g1);
          g0b = mfma16(u0[sq][i + 1], v0[sq][i + 1], g0b);
          g1b = mfma16(u1[sq][i + 1], v1[sq][i + 1], g1b);
        }
    }
    r0 = g0 + g0b;
    r1 = g1 + g1b;
  };
  // a column's sum over the 64 rows: four lanes per column (16 rows each: four ds_read_b128) and a cross-lane add
  auto colsum64 = [&](const float* __restrict__ tile, int ncols, float* __restrict__ dst) {
    const int c = lane & 15, part = lane >> 4;
    const float* cp = tile + min(c, MAXA - 1) * L::RS + part * 16;
    f32x4 t[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) t[i] = rd4(cp + 4 * i);
    __builtin_amdgcn_sched_barrier(0);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) s += (t[i][0] + t[i][1]) + (t[i][2] + t[i][3]);
    s = c < ncols ? s : 0.f;
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    if (lane < ncols && lane < 16) put(dst + lane, s);
  };
  auto colsum64_wide = [&](const float* __restrict__ tile, float* __restrict__ dst) {  // 32 columns, two lanes each
    const int c = lane & 31, part = lane >> 5;
    const float* cp = tile + c * L::RS + part * 32;
    f32x4 t[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) t[i] = rd4(cp + 4 * i);
    __builtin_amdgcn_sched_barrier(0);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += (t[i][0] + t[i][1]) + (t[i][2] + t[i][3]);
    s += __shfl_xor(s, 32, 64);
    if (lane < 32) put(dst + lane, s);
  };
  // Round 5: six of the twenty 16 x 16 x 64 tiles of a step produce ONE useful row or column (the value head's weight
  // gradient: two tiles for 32 numbers; at obs 17 the second K tile of both towers' first layer: four tiles for 2 x 32
  // numbers). With several gradient workgroups (TILES2) those products are VALU dots -- a lane per (feature, row half):
  // sixteen ds_read_b128 and 32 fused multiply-adds -- and the policy head's two tiles sit on the waves whose first-layer
  // tile went away (q = 1, 3): <= 64 MFMAs on every SIMD instead of 96 on two of them. Same-box bisect over the round's
  // versions (`profiles/r05_ppo_ab.md`): together with the 16-byte prefetch and the fast Adam 15.61 -> 15.13 us per step at
  // config P, those two alone 15.55; the one-workgroup forms measure 1-2 % SLOWER with the dots and keep the tiles.
  constexpr bool TILES2 = !LOCAL;
  const int KTg = (D + 15) >> 4;
  const bool narrow = TILES2 && KTg == 2 && D - 16 <= 4;   // (launch-constant) the first layer's second K tile holds <= 4 columns
  auto dot32 = [&](const float* __restrict__ U /* [32 features][RS] */, const float* __restrict__ vrow /* [RS] */) {
    const float* up = U + (lane & 31) * L::RS + (lane >> 5) * 32;
    const float* vp = vrow + (lane >> 5) * 32;
    f32x4 u[8], w[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      u[i] = rd4(up + 4 * i);
      w[i] = rd4(vp + 4 * i);
    }
    __builtin_amdgcn_sched_barrier(0);
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      s0 = __builtin_fmaf(u[i][0], w[i][0], s0);
      s1 = __builtin_fmaf(u[i][1], w[i][1], s1);
      s0 = __builtin_fmaf(u[i][2], w[i][2], s0);
      s1 = __builtin_fmaf(u[i][3], w[i][3], s1);
    }
    float sm = s0 + s1;
    sm += __shfl_xor(sm, 32, 64);
    return sm;   // (lanes j and j + 32: feature j's sum over the 64 rows)
  };
  if (tw == 0) {
    const bool head_tile = TILES2 ? (q & 1) != 0 : q < 2;   // dWa[a][h] = sum_r dout[r][a] a2[r][h], 16 h-columns per wave
    if (head_tile) {
      const int ht = TILES2 ? q >> 1 : q;
      const f32x4 g = outer16(lds + L::dout, li, a2t, ht * 16 + li);
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (lk * 4 + r < A) put(slab + (o.aW + (lk * 4 + r) * H + ht * 16 + li), g[r]);
    }
    if (q == 2) colsum64(lds + L::dout, A, slab + o.ab);
    if (q == (TILES2 ? 0 : 3) && !d.discrete) colsum64(lds + L::aux, A, slab + o.log_std);
  } else {
    if constexpr (TILES2) {
      if (q == 1) {  // dcW[h] = sum_r dv[r] a2[r][h]: 32 numbers
        const float sm = dot32(a2t, lds + L::misc + L::RS);
        if (lane < 32) put(slab + (o.cW + lane), sm);
      }
    } else {
      if (q < 2) {  // dcW[h] = sum_r dv[r] a2[r][h]  (only output row 0 is meaningful: the dvalue column feeds M index 0)
        const f32x4 g = outer16(lds + L::misc + L::RS, 0, a2t, q * 16 + li);   // (every lane reads column 1; rows m > 0 unused)
        if (lk == 0) put(slab + (o.cW + q * 16 + li), g[0]);
      }
    }
    if (q == 2) {  // cb = sum_r dv[r]; statpart slots {0 pg, 2 ent, 3 kl, 4 clip, 1 value} <- misc columns 2..6
      // columns 1..6 of the misc tile summed together: lane c < 6 handles column 1 + c
      const int c = lane & 15, part = lane >> 4;
      const float* cp = lds + L::misc + (1 + min(c, 5)) * L::RS + part * 16;
      f32x4 t[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) t[i] = rd4(cp + 4 * i);
      __builtin_amdgcn_sched_barrier(0);
      float sm = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) sm += (t[i][0] + t[i][1]) + (t[i][2] + t[i][3]);
      sm = c < 6 ? sm : 0.f;
      sm += __shfl_xor(sm, 16, 64);
      sm += __shfl_xor(sm, 32, 64);
      if (lane == 0) put(slab + (o.cb), sm);
      if (lane >= 1 && lane < 6) {
        const int m = lane - 1;                      // misc column 2 + m
        const int slot = m == 0 ? 0 : (m == 4 ? 1 : m + 1);
        if constexpr (LOCAL) slab_store(statpart + slot, sm);   // (write-through; the caller drains it)
        else if (ll_xcd) ll_store_xcd(slab64 + (((o.total + 3) & ~3) + slot), sm, ll_seq);   // the slab's tail
        else ll_store_agent(slab64 + (((o.total + 3) & ~3) + slot), sm, ll_seq);
      }
    }
  }
  IA_TS(7);
  {  // dW2 (one 16x16 tile per wave) together with the wave's first dW1 tile (dz1^T x); db2, db1
    const int KT = (D + 15) >> 4;
    const int jt2 = q >> 1, kt2 = q & 1;
    auto store_w1 = [&](int ti, const f32x4& g) {
      const int jt = ti / KT, kt = ti - jt * KT;
      const int col = kt * 16 + li;
      if (col < D)
#pragma unroll
        for (int r = 0; r < 4; ++r) put(slab + (oW1 + (jt * 16 + lk * 4 + r) * D + col), g[r]);
    };
    f32x4 g2;
    if (q < 2 * KT && !(narrow && (q & 1))) {   // (wave-uniform; narrow: tiles (jt, kt = 1) are the VALU dots below)
      const int jt = q / KT, kt = q - jt * KT;
      f32x4 g1;
      outer16_pair(dz2t, jt2 * 16 + li, a1t, kt2 * 16 + li, dz1t, jt * 16 + li, lds + L::x, kt * 16 + li, g2, g1);
      store_w1(q, g1);
    } else {
      g2 = outer16(dz2t, jt2 * 16 + li, a1t, kt2 * 16 + li);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) put(slab + (oW2 + (jt2 * 16 + lk * 4 + r) * H + kt2 * 16 + li), g2[r]);
    if (narrow && q == 1)   // dW1[j][16 + c] = sum_r dz1[r][j] x[r][16 + c], c < D - 16 <= 4: all 32 rows j of the tower at once
      for (int c = 16; c < D; ++c) {
        const float sm = dot32(dz1t, lds + L::x + c * L::RS);
        if (lane < 32) put(slab + (oW1 + lane * D + c), sm);
      }
    if (q == 3) colsum64_wide(dz2t, slab + ob2);
    for (int ti = q + 4; ti < 2 * KT; ti += 4) {   // observation widths beyond 32 columns: further dW1 tiles
      const int jt = ti / KT, kt = ti - jt * KT;
      store_w1(ti, outer16(dz1t, jt * 16 + li, lds + L::x, kt * 16 + li));
    }
    if (q == 2) colsum64_wide(dz1t, slab + ob1);
  }
  __syncthreads();
  IA_TS(8);
#undef IA_TS
}

// ---------------------------------------------------------------------------------------------
// Rollout step on the matrix pipe (hidden = 32): the forward half of the chain above for 64 rows per
// block -- wave (tower, q) takes rows q*16 .. q*16+15 through both layers and its head without block
// barriers after the feature staging -- then lanes 0..15 of the policy waves sample / clip / score
// their row exactly as the thread-per-row kernel does (same expression order), value waves store V.
template <int H>
struct ALds {
  static constexpr int XS = MAXD + 1, HS = H + 1, AS = MAXA + 1;
  static constexpr int x = 0;
  static constexpr int a1 = x + ROWS * XS;       // [2][ROWS][HS]
  static constexpr int a2 = a1 + 2 * ROWS * HS;
  static constexpr int out = a2 + 2 * ROWS * HS; // [ROWS][AS]
  static constexpr int total = out + ROWS * AS;
};

// EVAL = false: the rollout step (sample from `noise`, clip, log-prob of the sample).
// EVAL = true: [SB3 evaluate_actions] -- `noise` holds the GIVEN actions, `clipped` receives the entropy;
// `logp`, `values` and the entropy output may each be NULL (`ia_policy_evaluate`, hidden = 32).
template <int H, bool EVAL>
__device__ __forceinline__ void policy_act_body(
    const ia_policy_desc& d, const float* __restrict__ P, const float* __restrict__ Pt, const float* __restrict__ nm,
    const float* __restrict__ nv, const float* __restrict__ obs, int n, const float* __restrict__ noise,
    const float* __restrict__ low, const float* __restrict__ high, float* __restrict__ actions,
    float* __restrict__ clipped, float* __restrict__ values, float* __restrict__ logp, const int blk,
    float* __restrict__ lds, const int oz = 0, float* __restrict__ logits_out = nullptr) {
  // `logits_out` (host-sampled Discrete rollout step, `ia_policy_logits*`): the head outputs [n, A] and the values are all
  // that is wanted -- nothing is sampled.
  // `oz`: an opaque zero when the body sits inside the mailbox kernel's step loop -- its per-lane offsets and the weight
  // fragments are then re-derived per step instead of being hoisted out of the loop (71 spilled registers at H = 64)
  constexpr int NC = H / 16, KS = H / 4;
  using L = ALds<H>;
  const int tid = threadIdx.x + oz, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tw = wv >> 2, q = wv & 3;
  const int li = lane & 15, lk = lane >> 4;
  const int D = d.obs_dim, A = d.act_dim;
  const PolOff o = pol_offsets(D, A, H, d.discrete);
  const int i0 = blk * ROWS;
  const int S1 = (D + 3) >> 2;
  const int oW1 = tw ? o.vW1 : o.pW1, ob1 = tw ? o.vb1 : o.pb1, oW2 = tw ? o.vW2 : o.pW2, ob2 = tw ? o.vb2 : o.pb2;
  // The observations and the noise may live in device-mapped HOST memory (the rollout step of `PPO`): a load
  // is then a PCIe round trip of ~2 us, so every one of them is issued up front -- the block's ROWS x D
  // observation elements (contiguous rows, <= NOBS loads per thread) and, for the lanes that sample, the
  // row's noise -- and consumed later.
  constexpr int NOBS = (ROWS * MAXD + 511) / 512;
  float raw[NOBS], nrm_m[NOBS], nrm_v[NOBS];
  const int n_real = ROWS * D;
#pragma unroll
  for (int j = 0; j < NOBS; ++j) {
    const int e = min(tid + 512 * j, n_real - 1);
    const int r = e / D, k = e - r * D;
    raw[j] = obs[(long long)min(i0 + r, n - 1) * D + k];
    nrm_m[j] = d.has_norm ? nm[k] : 0.f;
    nrm_v[j] = d.has_norm ? nv[k] : 1.f - d.norm_eps;
  }
  const int srow = i0 + q * 16 + (lane & 15);        // row this lane samples for (tower 0, lanes 0..15)
  float r_noise[MAXA];
#pragma unroll
  for (int a2 = 0; a2 < MAXA; ++a2) r_noise[a2] = 0.f;
  if (tw == 0 && lane < 16 && noise != nullptr) {
    const long long nrow = (long long)min(srow, n - 1) * (d.discrete ? 1 : A);
#pragma unroll
    for (int a2 = 0; a2 < MAXA; ++a2) r_noise[a2] = noise[nrow + (d.discrete ? 0 : min(a2, A - 1))];
  }
  for (int e = tid; e < ROWS * L::XS; e += 512) lds[L::x + e] = 0.f;   // padding columns / rows
  __syncthreads();
#pragma unroll
  for (int j = 0; j < NOBS; ++j) {
    const int e = tid + 512 * j;
    if (e < n_real) {
      const int r = e / D, k = e - r * D;
      const float v = d.has_norm ? (raw[j] - nrm_m[j]) / sqrtf(nrm_v[j] + d.norm_eps) : raw[j];
      lds[L::x + r * L::XS + k] = (i0 + r) < n ? v : 0.f;
    }
  }
  // weight fragments straight from global memory (L2-resident, 14 KB): B[k = 4s+lk][j = c*16+li]
  float bW1[16][NC], bW2[KS][NC], bHead[KS], b1v[NC], b2v[NC];
  // (all loads first, unconditional at clamped addresses, behind at most four wave-uniform branches; masks after)
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    if (4 * g < S1) {
#pragma unroll
      for (int s = 4 * g; s < 4 * g + 4; ++s)
#pragma unroll
        for (int c = 0; c < NC; ++c) bW1[s][c] = Pt[oW1 + min(4 * s + lk, D - 1) * H + c * 16 + li];
    } else {
#pragma unroll
      for (int s = 4 * g; s < 4 * g + 4; ++s)
#pragma unroll
        for (int c = 0; c < NC; ++c) bW1[s][c] = 0.f;
    }
  }
  const int head_base = tw == 0 ? o.aW + min(li, A - 1) * H : o.cW;
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const int kk = 4 * s + lk;
#pragma unroll
    for (int c = 0; c < NC; ++c) bW2[s][c] = Pt[oW2 + kk * H + c * 16 + li];
    bHead[s] = P[head_base + kk];
  }
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    b1v[c] = P[ob1 + c * 16 + li];
    b2v[c] = P[ob2 + c * 16 + li];
  }
  const float head_bias = P[tw == 0 ? o.ab + min(li, A - 1) : o.cb];
  float r_ls[MAXA], r_low[MAXA], r_high[MAXA];   // per-action constants of the sampling lanes
#pragma unroll
  for (int a2 = 0; a2 < MAXA; ++a2) {
    const int ac = min(a2, A - 1);
    r_ls[a2] = d.discrete ? 0.f : P[o.log_std + ac];
    r_low[a2] = (EVAL || d.discrete) ? 0.f : low[ac];
    r_high[a2] = (EVAL || d.discrete) ? 0.f : high[ac];
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int s = 0; s < 16; ++s)
#pragma unroll
    for (int c = 0; c < NC; ++c) bW1[s][c] = (4 * s + lk < D) ? bW1[s][c] : 0.f;
  {
    const bool head_on = tw == 0 ? li < A : li == 0;
#pragma unroll
    for (int s = 0; s < KS; ++s) bHead[s] = head_on ? bHead[s] : 0.f;
  }
  __syncthreads();

  float* a1t = lds + L::a1 + tw * ROWS * L::HS;
  float* a2t = lds + L::a2 + tw * ROWS * L::HS;
  const int arow = q * 16 + li;
  {
    f32x4 acc[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 16; ++s)
      if (s < S1) {
        const float a = lds[L::x + arow * L::XS + 4 * s + lk];
#pragma unroll
        for (int c = 0; c < NC; ++c) acc[c] = mfma16(a, bW1[s][c], acc[c]);
      }
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) a1t[(q * 16 + lk * 4 + r) * L::HS + c * 16 + li] = fast_tanh(acc[c][r] + b1v[c]);
  }
  wave_sync_lds();
  {
    f32x4 acc[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const float a = a1t[arow * L::HS + 4 * s + lk];
#pragma unroll
      for (int c = 0; c < NC; ++c) acc[c] = mfma16(a, bW2[s][c], acc[c]);
    }
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) a2t[(q * 16 + lk * 4 + r) * L::HS + c * 16 + li] = fast_tanh(acc[c][r] + b2v[c]);
  }
  wave_sync_lds();
  {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < KS; ++s) acc = mfma16(a2t[arow * L::HS + 4 * s + lk], bHead[s], acc);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int rr = q * 16 + lk * 4 + r;
      if (tw == 0) { if (li < A) lds[L::out + rr * L::AS + li] = acc[r] + head_bias; }
      else if (li == 0 && i0 + rr < n && values != nullptr) values[i0 + rr] = acc[r] + head_bias;
    }
  }
  if (tw != 0) return;
  wave_sync_lds();
  const int row = i0 + q * 16 + lane;
  if (lane >= 16 || row >= n) return;
  const float* outrow = lds + L::out + (q * 16 + lane) * L::AS;
  if (logits_out != nullptr) {
    for (int a = 0; a < A; ++a) logits_out[(long long)row * A + a] = outrow[a];
    return;
  }
  if constexpr (EVAL) {
    float* entropy = clipped;
    if (!d.discrete) {
      float lp = 0.f, en = 0.f;
#pragma unroll
      for (int a = 0; a < MAXA; ++a)
        if (a < A) {
          lp += gauss_logp_term(r_noise[a], outrow[a], r_ls[a]);
          en += 0.5f + LOG_SQRT_2PI + logf(expf(r_ls[a]));  // Normal.entropy: 0.5 + 0.5*log(2*pi) + log(scale)
        }
      if (logp != nullptr) logp[row] = lp;
      if (entropy != nullptr) entropy[row] = en;
    } else {
      float mx = outrow[0];
      for (int a = 1; a < A; ++a) mx = fmaxf(mx, outrow[a]);
      float se = 0.f;
      for (int a = 0; a < A; ++a) se += expf(outrow[a] - mx);
      const float lse = mx + logf(se);
      float en = 0.f;
      for (int a = 0; a < A; ++a) {
        const float l = outrow[a] - lse;
        en -= expf(l) * l;
      }
      if (logp != nullptr) logp[row] = outrow[(int)r_noise[0]] - lse;
      if (entropy != nullptr) entropy[row] = en;
    }
    return;
  }
  if (!d.discrete) {
    float lp = 0.f;
#pragma unroll
    for (int a = 0; a < MAXA; ++a)
      if (a < A) {
        const float ls = r_ls[a];
        const float mu = outrow[a];
        const float act = __fadd_rn(mu, __fmul_rn(r_noise[a], expf(ls)));  // Normal.rsample
        actions[(long long)row * A + a] = act;
        clipped[(long long)row * A + a] = fminf(fmaxf(act, r_low[a]), r_high[a]);
        lp += gauss_logp_term(act, mu, ls);
      }
    logp[row] = lp;
  } else {
    float mx = outrow[0];
    for (int a = 1; a < A; ++a) mx = fmaxf(mx, outrow[a]);
    float se = 0.f;
    for (int a = 0; a < A; ++a) se += expf(outrow[a] - mx);
    const float lse = mx + logf(se);
    const float u = r_noise[0];
    float c = 0.f;
    int pick = A - 1;
    if (u < 0.f) {  // mode of the Categorical (argmax, first index on ties)
      pick = 0;
      for (int a = 1; a < A; ++a)
        if (outrow[a] > outrow[pick]) pick = a;
    } else {
      for (int a = 0; a < A; ++a) {
        c += expf(outrow[a] - lse);
        if (u < c) { pick = a; break; }
      }
    }
    actions[row] = (float)pick;
    clipped[row] = (float)pick;
    logp[row] = outrow[pick] - lse;
  }
}

template <int H, bool EVAL>
__global__ __launch_bounds__(512) void policy_act_mfma_kernel(
    ia_policy_desc d, const float* __restrict__ P, const float* __restrict__ Pt, const float* __restrict__ nm,
    const float* __restrict__ nv, const float* __restrict__ obs, int n, const float* __restrict__ noise,
    const float* __restrict__ low, const float* __restrict__ high, float* __restrict__ actions,
    float* __restrict__ clipped, float* __restrict__ values, float* __restrict__ logp) {
  extern __shared__ float lds[];
  policy_act_body<H, EVAL>(d, P, Pt, nm, nv, obs, n, noise, low, high, actions, clipped, values, logp, blockIdx.x, lds);
}

// A whole rollout's act steps in ONE launch: the workgroups stay resident and take step t when the host has posted it --
// `ready` (one int in pinned, device-mapped host memory) reaches t + 1 after the step's observations (and noise) are in
// their pinned tiles -- run the same body, and acknowledge in `done[workgroup]` (pinned host memory) once the step's
// outputs have left: the clipped actions the host env workers read next are ordinary stores to host memory, every
// thread drains its stores (`vmcnt(0)`), block barrier, then one lane's system-scope release (L2 write-back) and flag
// store; the step's loads sit behind a system-scope acquire (a tile may share a cache line with the previous step's). A step then costs the host one flag write and one poll instead of a launch and a stream
// synchronisation (~30 -> ~12 us per env step at config P). Bounded: a workgroup leaves when `ready` turns negative (the
// host's abort / error path) or after `timeout_ticks` (100 MHz) without a new step; `done` then holds -(t + 1).
struct ActMailbox {
  const float* obs; long long s_obs;     // element strides between consecutive steps
  const float* noise; long long s_noise;
  float* actions; long long s_act;
  float* clipped; long long s_clip;
  float* values; long long s_val;
  float* logp; long long s_lp;
  float* last_val;                       // non-null: one more step, T, that only evaluates V(obs[T]) (the GAE bootstrap)
  int T; const int* ready; int* done; long long timeout_ticks;
};

template <int H>
__global__ __launch_bounds__(512) void policy_rollout_mailbox_kernel(
    ia_policy_desc d, const float* __restrict__ P, const float* __restrict__ Pt, const float* __restrict__ nm,
    const float* __restrict__ nv, int n, const float* __restrict__ low, const float* __restrict__ high, ActMailbox mb) {
  extern __shared__ float lds[];
  __shared__ int s_go;
  for (int t = 0; t < mb.T; ++t) {
    if (!mailbox_wait(mb.ready, t, mb.timeout_ticks, &s_go)) {
      if (threadIdx.x == 0) __hip_atomic_store(mb.done + blockIdx.x, -(t + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      return;
    }
    // H = 32: the weight fragments are loop-invariant and the compiler keeps them in registers across the steps (236
    // VGPRs, no spills: 6 us per step less than re-reading them). H = 64: hoisted, they spill 71 registers -- an opaque
    // zero makes every step re-derive its offsets and fragments there.
    int oz = 0;
    if constexpr (H == 64) asm volatile("s_mov_b32 %0, 0" : "=s"(oz));
    policy_act_body<H, false>(d, P + oz, Pt + oz, nm, nv, mb.obs + t * mb.s_obs, n,
                              mb.noise ? mb.noise + t * mb.s_noise : nullptr, low, high, mb.actions + t * mb.s_act,
                              mb.clipped + t * mb.s_clip, mb.values + t * mb.s_val, mb.logp + t * mb.s_lp,
                              blockIdx.x + oz, lds, oz);
    mailbox_ack(mb.done, t + 1);
  }
  if (mb.last_val != nullptr) {
    // the value of the observation behind the last step ([SB3 collect_rollouts]: `predict_values(new_obs)` for the GAE
    // bootstrap), posted by the host as step T: one launch and ~30 us less between the last env step and the update
    if (!mailbox_wait(mb.ready, mb.T, mb.timeout_ticks, &s_go)) return;
    policy_act_body<H, true>(d, P, Pt, nm, nv, mb.obs + mb.T * mb.s_obs, n, nullptr, nullptr, nullptr, nullptr, nullptr,
                             mb.last_val, nullptr, blockIdx.x, lds);
    mailbox_ack(mb.done, mb.T + 1);
  }
}

// The host-sampled Discrete step on the MFMA body (one 512-thread workgroup per 64 rows, both towers at once): logits
// [n, A] + values. (The thread-per-row kernels this replaced took ~30 us for a 64-wide policy -- 2 x 4 500 dependent FMAs
// per row with every weight a load --, HALF of an 8-environment rollout step of BASELINE config 1; they and the other
// thread-per-row forms behind `ia_ppo_force_valu` were retired in round 5: no production path reached them.)
template <int H>
__global__ __launch_bounds__(512) void policy_logits_mfma_kernel(
    ia_policy_desc d, const float* __restrict__ P, const float* __restrict__ Pt, const float* __restrict__ nm,
    const float* __restrict__ nv, const float* __restrict__ obs, int n, float* __restrict__ logits,
    float* __restrict__ values) {
  extern __shared__ float lds[];
  policy_act_body<H, false>(d, P, Pt, nm, nv, obs, n, nullptr, nullptr, nullptr, nullptr, nullptr, values, nullptr,
                            blockIdx.x, lds, 0, logits);
}

template <int H>
__global__ __launch_bounds__(512) void policy_logits_mailbox_mfma_kernel(
    ia_policy_desc d, const float* __restrict__ P, const float* __restrict__ Pt, const float* __restrict__ nm,
    const float* __restrict__ nv, int n, LogitsMailbox mb) {
  extern __shared__ float lds[];
  __shared__ int s_go;
  for (int t = 0; t < mb.T; ++t) {
    if (!mailbox_wait(mb.ready, t, mb.timeout_ticks, &s_go)) {
      if (threadIdx.x == 0) __hip_atomic_store(mb.done + blockIdx.x, -(t + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      return;
    }
    int oz = 0;   // (H = 64: see policy_rollout_mailbox_kernel)
    if constexpr (H == 64) asm volatile("s_mov_b32 %0, 0" : "=s"(oz));
    policy_act_body<H, false>(d, P + oz, Pt + oz, nm, nv, mb.obs + t * mb.s_obs, n, nullptr, nullptr, nullptr, nullptr, nullptr,
                              mb.values + t * mb.s_val, nullptr, blockIdx.x + oz, lds, oz, mb.logits);
    mailbox_ack(mb.done, t + 1);
  }
}

constexpr int UPD_RING = 4;
constexpr int UPD_MAX_STEPS = 2048;                // optimiser steps per launch (Adam scalar tables in ws)
constexpr int UPD_NPT = 8;                        // parameters per thread: 8 (<= 4096 parameters) or, for wider
constexpr int UPD_NPT_WIDE = 9;                   // observation/action spaces (Ant-shaped: 4209), 9 -- as many as the
                                                  // LDS-resident parameter copies allow next to the minibatch tiles
constexpr int UPD_RS = 2 * MAXD + 8;              // ring slot: mean[MAXD], var[MAXD], adv mean, adv std
constexpr int UPD_CTRL = 192;                     // control words: 1 steps published, 2 second-level arrivals,
                                                  // 8 error (sticky), 16..47 steps sliced per slicer,
                                                  // 64 + 16 i (i < 8): arrival counters of the grid barrier
constexpr int UPD_ARR = 8;                        // workgroup v arrives on counter v % 8 (one cache line each): 128
                                                  // workgroups adding to ONE word serialise at its memory channel
constexpr int UPD_SLICE = 512;                    // rows per statistics slice (minibatches > 1024 rows)
constexpr int UPD_SLICES_MAX = 32;                // => minibatches up to 16 384 rows
constexpr int UPD_PRS = 2 * MAXD + 4;             // slice partial: mean[MAXD], M2[MAXD], adv mean, adv M2, rows
constexpr int UPD_SD = 8;                         // depth of the loss-statistic partial ring (> UPD_RING + 2)
constexpr int UPD_NBLK_MAX = 192;                 // co-residency: nblk + 1 workgroups on 256 CUs
struct UpdSched {
  int n_steps, first, n_mb, batch_size;
  long long total;
};
// Row-sharded data parallelism (template parameter SHARD of the persistent kernel): every rank runs
// the single-GPU kernel on ITS rows [rank * rows_per_rank, ...) of each GLOBAL minibatch (the rollout tile, permutations
// and minibatch statistics are the global ones on every rank), reduces its workgroups' slabs in one level exactly like the
// single-GPU update, and exchanges one record per optimiser step -- its partial gradient (P4 floats) + its loss-statistic
// sums (8 floats) -- by writing it straight into every rank's receive area (peer-mapped device memory: xGMI between GPUs,
// hipIpc between processes) and raising a per-(source, piece) flag there; every workgroup then sums the `world` records
// in rank order (identical on all ranks: replicas stay bit-identical), clips by the GLOBAL norm and applies Adam. All
// hand-offs are system-scope: write-through stores, `vmcnt(0)`, block barrier, release + relaxed flag store | relaxed
// polling, acquire, loads that bypass the caches. Flags carry a sequence number that only grows over the exchange
// context's life (never reset: a peer may already be writing the next launch's first step), records are double-buffered
// by step parity (a sender cannot be two steps ahead of a receiver: it needs that receiver's record of the step between).
constexpr int SHARD_WORLD_MAX = 8;
constexpr int SHARD_PIECES_MAX = 8;
constexpr int SHARD_RB = 2;             // records read per batch (registers: SHARD_RB x parameters per thread x 2)     // a record travels to a peer in <= 8 pieces, each sent by another workgroup
struct ShardArgs {
  int world, rank, rows_per_rank, pieces;
  int loopback;                         // cost model on one process: this rank stands in for every source rank in turn
  unsigned seq_base;                    // step s of this launch carries sequence number seq_base + s + 1
  long long timeout_ticks;              // 100 MHz ticks a workgroup waits for the peers' records of one step
  unsigned long long* recv;                        // this rank's receive area: [2][world][P4 + 8] (value, sequence) words
  unsigned long long* peer_recv[SHARD_WORLD_MAX];  // every rank's receive area as mapped into this process (own included)
};
struct UpdWs {
  unsigned* ctrl;
  float *tab;   // [2][UPD_MAX_STEPS]: Adam step size lr/(1-b1^t) and sqrt(1-b2^t) per step (host doubles)
  float *ring, *normcoef, *statpart;           // normcoef: [UPD_SD][2] gradient norm, clip coefficient
  unsigned long long *slabs64;                 // [2][nblk][P4 + 8] (value, sequence) words: the workgroups' gradient slabs,
                                               // tail = their loss-statistic partials (several gradient workgroups only)
  unsigned long long *sums64;                  // [2][P4 + 8]: the reduced vector, published slice by slice
  float *pring;                                // [UPD_RING][UPD_SLICES_MAX][UPD_PRS]: statistics slice partials
  int P4;
};
__host__ __device__ inline UpdWs upd_ws(float* ws, int nblk, int P) {
  UpdWs w;
  w.P4 = (P + 3) & ~3;
  w.ctrl = reinterpret_cast<unsigned*>(ws);
  w.tab = ws + UPD_CTRL;
  w.ring = w.tab + 2 * UPD_MAX_STEPS;
  w.normcoef = w.ring + UPD_RING * UPD_RS;
  w.statpart = w.normcoef + UPD_SD * 2;
  float* slabs = w.statpart + UPD_SD * nblk * 8;
  w.slabs64 = reinterpret_cast<unsigned long long*>(slabs);
  float* sums = slabs + 4 * (long long)nblk * (w.P4 + 8);
  w.sums64 = reinterpret_cast<unsigned long long*>(sums);
  w.pring = sums + 4 * (long long)(w.P4 + 8);
  return w;
}

// Grid-barrier wait on the UPD_ARR arrival counters, called by ALL lanes of one wave: lane i < UPD_ARR polls counter i
// until it has seen `steps` arrivals of each of its workgroups (those with index % UPD_ARR == i). Bounded like spin_until.
__device__ __forceinline__ bool spin_arrivals(unsigned* arr, unsigned steps, int nblk, unsigned* err) {
  const int lane = threadIdx.x & 63;
  const unsigned mine = lane < UPD_ARR ? (unsigned)((nblk - lane + UPD_ARR - 1) / UPD_ARR) : 0u;
  const unsigned target = steps * mine;
  unsigned* p = arr + 16 * (lane < UPD_ARR ? lane : 0);
  unsigned it = 0;
  for (;;) {
    const bool ok = lane >= UPD_ARR || __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target;
    if (__all(ok)) return true;
    __builtin_amdgcn_s_sleep(2);
    if (++it > (1u << 24)) {
      if (lane == 0) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return false;
    }
    if ((it & 255u) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return false;
  }
}

__device__ __forceinline__ bool spin_until(unsigned* p, unsigned target, unsigned* err) {
  unsigned it = 0;
  while (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
    __builtin_amdgcn_s_sleep(2);
    if (++it > (1u << 24)) {
      __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return false;
    }
    if ((it & 255u) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return false;
  }
  return true;
}

// ---------------------------------------------------------------------------------------------
// One launch per EPOCH for the towers the H = 32 persistent kernel does not cover ([64, 64]: SB3's default `MlpPolicy`):
// the minibatch steps of `ia_ppo_epoch` -- gradient kernel + split apply kernel, two launches and ~6 us of dispatch
// boundary each -- as phases of one co-resident grid of `nwg` workgroups separated by grid barriers:
//   A  gradient of the minibatch (mfma_minibatch<H>, the device function the per-minibatch kernel runs; parameter
//      fragments from L2), slabs + loss-statistic partials                                      | barrier
//   B1 workgroup g sums chunk g of the slabs in slab order (registers) and leaves the chunk's sum of squares | barrier
//   B2 every workgroup folds the G partial sums in order -> norm, clip coefficient; Adam on its own chunk (torch layout
//      + transposed shadow copy); workgroup 0 writes the step's loss statistics                 | barrier
// Same arithmetic as the two kernels (the chunking of the sum of squares differs: G = nwg chunks here). Barrier: every
// thread's stores acknowledged, block barrier, one lane's agent-scope release + relaxed arrive on a monotonic counter,
// bounded spin, acquire (the gfx950 hand-off rules of disc_fused.hip / the persistent H = 32 kernel).
struct EpochSteps {
  static constexpr int MAX = 192;     // (1.5 KB of kernel arguments: a round's 160 steps in one launch)
  int n, first;                       // minibatches [first, first + n) of the sequence
  float step_size[MAX], bc2_sqrt[MAX];
};

__device__ __forceinline__ bool epoch_grid_sync(unsigned* ctr, unsigned target, unsigned* err, int* s_flag) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool ok = spin_until(ctr, target, err);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    *s_flag = ok ? 1 : 0;
  }
  __syncthreads();
  return *s_flag != 0;
}

// SPLIT: two workgroups of four waves per row block, one per tower (mfma_minibatch<H, true, true>); the apply phases are
// the same with 256 threads per workgroup and twice as many chunks.
template <int H, bool SPLIT = false>
__global__ __launch_bounds__(SPLIT ? 256 : 512) void ppo_epoch_persistent_kernel(
    ia_policy_desc d, float* __restrict__ P, float* __restrict__ Pt, float* __restrict__ m, float* __restrict__ v,
    const float* __restrict__ nm_in, const float* __restrict__ nv_in, const float* __restrict__ obs,
    const float* __restrict__ actions, const float* __restrict__ old_logp, const float* __restrict__ adv,
    const float* __restrict__ ret, long long total_rows, int batch_size, int T, int n_envs, int normalize_adv,
    float clip, float ent_coef, float vf_coef, float max_norm, float beta1, float beta2, float eps,
    float* __restrict__ ws, const float* __restrict__ seq, int snap, float* __restrict__ stats, EpochSteps st,
    long long* __restrict__ dbg /* measurement: [0..5] += 100 MHz ticks of workgroup 0 in {A, barrier, B1, barrier, B2, barrier} */) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  __shared__ int s_flag;
  long long tprev = 0;
#define EP_TS(slot)                                                       \
  do {                                                                    \
    if (dbg != nullptr && blockIdx.x == 0 && threadIdx.x == 0) {          \
      const long long tn = wall_clock64();                                \
      dbg[slot] += tn - tprev;                                            \
      tprev = tn;                                                         \
    }                                                                     \
  } while (0)
  if (dbg != nullptr && blockIdx.x == 0 && threadIdx.x == 0) tprev = wall_clock64();
  __shared__ float s_part[64];
  constexpr int NT = SPLIT ? 256 : 512;
  const int bid = blockIdx.x, nwg = gridDim.x;
  const int nrb = SPLIT ? nwg >> 1 : nwg;   // row blocks of a full minibatch: the workspace layout of every step
  const int D = d.obs_dim, aw = d.discrete ? 1 : d.act_dim;
  const PolOff o = pol_offsets(D, d.act_dim, H, d.discrete);
  unsigned* ctr = reinterpret_cast<unsigned*>(ws) + 4;
  unsigned* err = reinterpret_cast<unsigned*>(ws) + 5;
  unsigned bar = 0;
  const int chunk = (o.total + nwg - 1) / nwg;
  constexpr int NPC = 4;   // parameters of the chunk per thread (chunk <= 2048)
  // SPLIT: the workgroup's chunk of the parameters and of Adam's moments lives in registers across the launch's steps (the
  // chunk is the same every step and nobody else writes it): no loads in the apply phase, m / v go back to memory once at
  // the end. (The eight-wave form has no registers to spare: it loads and stores them every step.)
  float m_[NPC], v_[NPC], p_[NPC];
#pragma unroll
  for (int j = 0; j < NPC; ++j) {
    m_[j] = 0.f; v_[j] = 0.f; p_[j] = 0.f;
    if constexpr (SPLIT) {
      const int i = min(bid * chunk + (int)threadIdx.x + j * NT, o.total - 1);
      m_[j] = m[i];
      v_[j] = v[i];
      p_[j] = P[i];
    }
  }
#pragma nounroll
  for (int k = 0; k < st.n; ++k) {
    const int mb = st.first + k;
    const long long start = (long long)mb * batch_size;
    const int b = (int)min((long long)batch_size, total_rows - start);
    const int nblk = (b + ROWS - 1) / ROWS;
    // (the layout of a FULL minibatch also for a short last one: the partial sums of squares of workgroups past its row
    //  blocks must not land inside its slabs)
    const PpoWs w = ppo_ws(ws, nrb, o.total);
    const float* sq = seq + (long long)mb * EPS_SEQ;
    // ---- A: gradient of this minibatch
    if ((SPLIT ? bid >> 1 : bid) < nblk) {
      const MbRows rows{obs + start * D, actions + start * aw, old_logp + start, adv + start, ret + start, nullptr, b, T,
                        n_envs};
      // (opaque zero: inlined into the step loop, the chain's loop-invariant per-lane offsets are otherwise hoisted out of
      //  it -- 278 spilled registers; 9 remain. As a real call (`noinline`) the ABI's saves cost 43.)
      int oz;
      asm volatile("s_mov_b32 %0, 0" : "=s"(oz));
      const int rb = SPLIT ? bid >> 1 : bid;
      mfma_minibatch<H, true, SPLIT>(d, P + oz, Pt + oz, snap ? sq + 8 : nm_in, snap ? sq + 8 + MAXD : nv_in, sq[0], sq[1],
                                     rows, rb + oz, normalize_adv, clip, ent_coef, vf_coef,
                                     w.slabs + (long long)rb * o.total, w.statpart + rb * 8, lds,
                                     dbg != nullptr ? dbg + 16 + (bid & 1) * 16 : nullptr, oz, bid & 1);
    }
    EP_TS(0);
    if (!epoch_grid_sync(ctr, (++bar) * nwg, err, &s_flag)) return;
    EP_TS(1);
    // ---- B1: chunk `bid` of the slab sum (fixed slab order), the chunk's sum of squares
    int oz2;   // (the apply phases' per-thread offsets are re-derived per step as well: nothing of theirs lives across A)
    asm volatile("s_mov_b32 %0, 0" : "=s"(oz2));
    const int tid = threadIdx.x + oz2;
    const int i0 = bid * chunk + oz2, i1 = min(o.total, i0 + chunk);
    float g[NPC];
    float sqs = 0.f;
#pragma unroll
    for (int j = 0; j < NPC; ++j) {
      const int i = i0 + tid + j * NT;
      float acc = 0.f;
      if (i < i1) {
        int sb = 0;
        for (; sb + 8 <= nblk; sb += 8) {   // 8 independent loads in flight, then a fixed-order sum (ppo_apply_split_kernel)
          float t[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) t[u] = w.slabs[(long long)(sb + u) * o.total + i];
#pragma unroll
          for (int u = 0; u < 8; ++u) acc += t[u];
        }
        for (; sb < nblk; ++sb) acc += w.slabs[(long long)sb * o.total + i];
        sqs += acc * acc;
      }
      g[j] = acc;
    }
    {
      const float part = block_sum<NT>(sqs, lds);
      // (slots 5 / 6 of the loss-statistic partials are unused by the gradient; SPLIT: one per tower workgroup)
      if (tid == 0) slab_store(w.statpart + (SPLIT ? (bid >> 1) * 8 + 5 + (bid & 1) : bid * 8 + 5), part);
    }
    EP_TS(2);
    if (!epoch_grid_sync(ctr, (++bar) * nwg, err, &s_flag)) return;
    EP_TS(3);
    // ---- B2: norm, clip, Adam on the own chunk; loss statistics
    if (tid < 64) s_part[tid] = tid < nwg ? w.statpart[SPLIT ? (tid >> 1) * 8 + 5 + (tid & 1) : tid * 8 + 5] : 0.f;
    __syncthreads();
    float total_sq = 0.f;
    for (int q = 0; q < nwg; ++q) total_sq += s_part[q];
    const float total_norm = sqrtf(total_sq);
    const float coef = fminf(max_norm / (total_norm + 1e-6f), 1.0f);   // torch.nn.utils.clip_grad_norm_
    if (bid == 0 && stats != nullptr) {
      __shared__ float s_st[5];
      if (tid >= 64 && tid < 69) {   // one lane of the second wave per statistic (the order of ppo_apply_split_kernel)
        const int kk = tid - 64;
        float sv = 0.f;
        for (int q = 0; q < nblk; ++q) sv += w.statpart[q * 8 + kk];
        sv *= 1.f / (float)b;
        stats[(long long)mb * 8 + kk] = sv;
        s_st[kk] = sv;
      }
      __syncthreads();
      if (tid == 0) {
        stats[(long long)mb * 8 + 5] = s_st[0] + ent_coef * s_st[2] + vf_coef * s_st[1];  // loss
        stats[(long long)mb * 8 + 6] = total_norm;
        stats[(long long)mb * 8 + 7] = coef;
      }
    }
    {
      const float step_size = st.step_size[k], bc2_sqrt = st.bc2_sqrt[k];
      if constexpr (!SPLIT) {
#pragma unroll
        for (int j = 0; j < NPC; ++j) {
          const int i = min(i0 + tid + j * NT, o.total - 1);
          m_[j] = m[i];
          v_[j] = v[i];
          p_[j] = P[i];
        }
      }
#pragma unroll
      for (int j = 0; j < NPC; ++j) {
        const int i = i0 + tid + j * NT;
        if (i < i1) {
          const float gi = g[j] * coef;
          const float mi = m_[j] + (gi - m_[j]) * (1.f - beta1);
          const float vi = v_[j] * beta2 + (1.f - beta2) * gi * gi;
          const float denom = sqrtf(vi) / bc2_sqrt + eps;
          const float pn = p_[j] - step_size * (mi / denom);
          if constexpr (SPLIT) {
            slab_store(P + i, pn);   // (write-through: read by every workgroup behind the barrier)
            p_[j] = pn;
            m_[j] = mi;
            v_[j] = vi;
          } else {
            P[i] = pn;
            m[i] = mi;
            v[i] = vi;
          }
          int dst = i;
          auto tr = [&](int b0, int rows_, int cols) {
            if (i >= b0 && i < b0 + rows_ * cols) {
              const int r = (i - b0) / cols, cc = (i - b0) % cols;
              dst = b0 + cc * rows_ + r;
            }
          };
          tr(o.pW1, H, D); tr(o.pW2, H, H); tr(o.vW1, H, D); tr(o.vW2, H, H);
          if constexpr (SPLIT) slab_store(Pt + dst, pn);
          else Pt[dst] = pn;
        }
      }
    }
    EP_TS(4);
    if (!epoch_grid_sync(ctr, (++bar) * nwg, err, &s_flag)) return;
    EP_TS(5);
  }
  if constexpr (SPLIT) {
#pragma unroll
    for (int j = 0; j < NPC; ++j) {
      const int i = bid * chunk + (int)threadIdx.x + j * NT;
      if (i < min(o.total, (bid + 1) * chunk)) {
        m[i] = m_[j];
        v[i] = v_[j];
      }
    }
  }
#undef EP_TS
}

// ---------------------------------------------------------------------------------------------
// The one-tower epoch kernel WITHOUT grid barriers: the three hand-offs of a step travel as 8-byte (value, sequence) words
// (the exchange of the H = 32 persistent kernel: no flags, no fences, nothing to reset -- a word is valid when its upper
// half carries the step's sequence number; two buffers by sequence parity):
//   A   gradient of the minibatch (mfma_minibatch<H, true, true, NLL>): the tower's parameters are POLLED from the
//       parameter words (`par`, published by the chunk owners at the end of the previous step), the gradient entries and
//       the loss-statistic partials leave as words of the row block's slab
//   B1  workgroup g polls chunk g of every slab and sums it in slab order; its sum of squares leaves as ONE word (`sq`)
//   B2  every workgroup polls the G partial sums of squares and folds them in order -> norm, clip coefficient; Adam on
//       its own chunk (parameters and moments in registers across the launch); the new parameters leave as words.
// Same arithmetic in the same order as ppo_epoch_persistent_kernel<H, true>: bit-identical results
// (`ia_ppo_epoch_split(3)` keeps that kernel; tests compare). Who may overwrite what: a workgroup writes step s + 2's slab
// words (same buffer as step s's) only behind its poll of step s + 1's parameters, which every chunk owner publishes
// behind ITS poll of all sums of squares of step s + 1, which every workgroup publishes behind its slab poll of step
// s + 1 -- so every reader of step s's slab words is long done; the same chain covers `sq` and `par`.
// Sequence numbers: `seq0` + k for "the parameters step k of this launch reads", `seq0` + k + 1 for what it produces;
// the host passes 1 + Adam steps done (never 0) and clears the word areas once per call (the workspace arrives
// uninitialised and a caller may restart its step count).
struct EpochLl {
  unsigned long long* base;   // slabs [2][nrb][P + 8] | sq [2][64] | par [2][P]   (8-byte words)
  unsigned seq0;
};
__host__ __device__ inline long long epoch_ll_words(int nrb, int P) { return 2LL * nrb * (P + 8) + 2 * 64 + 2LL * P; }
// row blocks the launch is laid out for: the minibatch's own, or as many as it takes for chunks of <= 1 024 parameters
__host__ __device__ inline int epoch_ll_row_blocks(int nrb, int P) {
  const int owners = (P + 1023) / 1024;
  return nrb > (owners + 1) / 2 ? nrb : (owners + 1) / 2;
}

template <int H, int NLL>
__global__ __launch_bounds__(256) void ppo_epoch_ll_kernel(
    ia_policy_desc d, float* __restrict__ P, float* __restrict__ Pt, float* __restrict__ m, float* __restrict__ v,
    const float* __restrict__ nm_in, const float* __restrict__ nv_in, const float* __restrict__ obs,
    const float* __restrict__ actions, const float* __restrict__ old_logp, const float* __restrict__ adv,
    const float* __restrict__ ret, long long total_rows, int batch_size, int T, int n_envs, int normalize_adv,
    float clip, float ent_coef, float vf_coef, float max_norm, float beta1, float beta2, float eps,
    float* __restrict__ ws, EpochLl ll, const float* __restrict__ seq, int snap, float* __restrict__ stats, EpochSteps st,
    long long* __restrict__ dbg /* measurement: [0..5] += 100 MHz ticks of workgroup 0 in {A, slab poll + sum, sum of squares
                                   published, poll of the sums of squares, Adam + publish, -} */) {
  typedef unsigned long long u64;
  extern __shared__ __attribute__((aligned(16))) float lds_raw[];
  // (the dynamic area starts where the static one ends -- not a 16-byte boundary here: rounded up, the launch asks for 16
  //  bytes more; the images' ds_read_b128 fragment reads would otherwise be split)
  float* lds = lds_raw + (((16 - (__builtin_amdgcn_groupstaticsize() & 15)) & 15) >> 2);
  __shared__ int s_fail;
  __shared__ float s_part[64];
  __shared__ float s_stat[32 * 8];
  long long tprev = 0;
#define EP_TS(slot)                                                       \
  do {                                                                    \
    if (dbg != nullptr && blockIdx.x == 0 && threadIdx.x == 0) {          \
      const long long tn = wall_clock64();                                \
      dbg[slot] += tn - tprev;                                            \
      tprev = tn;                                                         \
    }                                                                     \
  } while (0)
  if (dbg != nullptr && blockIdx.x == 0 && threadIdx.x == 0) tprev = wall_clock64();
  constexpr int NT = 256;
  const int bid = blockIdx.x, nwg = gridDim.x, nrb = nwg >> 1;
  const int D = d.obs_dim, aw = d.discrete ? 1 : d.act_dim;
  const PolOff o = pol_offsets(D, d.act_dim, H, d.discrete);
  const int SW = o.total + 8;
  u64* slabs64 = ll.base;
  u64* sq64 = slabs64 + 2LL * nrb * SW;
  u64* par64 = sq64 + 2 * 64;
  unsigned* err = reinterpret_cast<unsigned*>(ws) + 5;
  const int chunk = (o.total + nwg - 1) / nwg;
  constexpr int NPC = 4;   // parameters of the chunk per thread (chunk <= 1024: the host launches enough workgroups -- those
                           // beyond the minibatch's row blocks have no gradient phase, they only own a chunk)
  if (threadIdx.x == 0) s_fail = 0;
  // where each of the thread's polled parameter words goes in the transposed LDS image (mfma_minibatch: rows of W1^T, then
  // of W2^T, H + 4 floats apart); the same every step
  int tdst[NLL];
  {
    constexpr int TS = H + 4;
    const int nW1 = H * D, oW2i = nW1 + H;
#pragma unroll
    for (int i = 0; i < NLL; ++i) {
      const int e = (int)threadIdx.x + i * NT;
      int t = -1;
      if (e < nW1) {
        const int r = e / D;
        t = (e - r * D) * TS + r;
      } else if (e >= oW2i && e < oW2i + H * H) {
        const int j = e - oW2i;
        t = (D + (j & (H - 1))) * TS + (j >> 6);
      }
      tdst[i] = t;
    }
  }
  // the workgroup's chunk of the parameters and of Adam's moments: registers across the launch (nobody else writes it)
  float m_[NPC], v_[NPC], p_[NPC];
#pragma unroll
  for (int j = 0; j < NPC; ++j) {
    const int i = min(bid * chunk + (int)threadIdx.x + j * NT, o.total - 1);
    m_[j] = m[i];
    v_[j] = v[i];
    p_[j] = P[i];
  }
  {   // the parameters the first step reads
    u64* dst = par64 + (long long)(ll.seq0 & 1u) * o.total;
#pragma unroll
    for (int j = 0; j < NPC; ++j) {
      const int i = bid * chunk + (int)threadIdx.x + j * NT;
      if (i < min(o.total, (bid + 1) * chunk)) ll_store_agent(dst + i, p_[j], ll.seq0);
    }
  }
  __syncthreads();
#pragma nounroll
  for (int k = 0; k < st.n; ++k) {
    const unsigned seq_par = ll.seq0 + (unsigned)k, seq_out = seq_par + 1u;
    const int mb = st.first + k;
    const long long start = (long long)mb * batch_size;
    const int b = (int)min((long long)batch_size, total_rows - start);
    const int nblk = (b + ROWS - 1) / ROWS;
    const float* sq = seq + (long long)mb * EPS_SEQ;
    u64* slabs_s = slabs64 + (long long)(seq_out & 1u) * nrb * SW;
    // ---- A: gradient of this minibatch
    if ((bid >> 1) < nblk) {
      const MbRows rows{obs + start * D, actions + start * aw, old_logp + start, adv + start, ret + start, nullptr, b, T,
                        n_envs};
      int oz;   // (opaque zero: see ppo_epoch_persistent_kernel. Without it the loop-invariant offsets are hoisted into 495-512
                //  registers -- spills in two of the three width classes -- for 0.4 us per step: measured, not taken)
      asm volatile("s_mov_b32 %0, 0" : "=s"(oz));
      const int rb = bid >> 1;
      float* slab = reinterpret_cast<float*>(slabs_s + (long long)rb * SW);   // (a word array: offsets only)
      const MbLl mll{par64 + (long long)(seq_par & 1u) * o.total, seq_par, seq_out, &s_fail, err};
      mfma_minibatch<H, true, true, NLL>(d, P + oz, Pt + oz, snap ? sq + 8 : nm_in, snap ? sq + 8 + MAXD : nv_in, sq[0], sq[1],
                                         rows, rb + oz, normalize_adv, clip, ent_coef, vf_coef, slab, slab + o.total, lds,
                                         dbg != nullptr ? dbg + 16 + (bid & 1) * 16 : nullptr, oz, bid & 1, mll, tdst);
    }
    if (s_fail) return;   // (behind the function's closing barrier; workgroups without rows never set it)
    EP_TS(0);
    // ---- B1: chunk `bid` of every slab, summed in slab order; workgroup 0 also collects the loss-statistic partials
    int oz2;
    asm volatile("s_mov_b32 %0, 0" : "=s"(oz2));
    const int tid = threadIdx.x + oz2, lane = tid & 63;
    const int i0 = bid * chunk + oz2, i1 = min(o.total, i0 + chunk);
    bool fail = false;
    auto timed_out = [&](unsigned& it) {
      __builtin_amdgcn_s_sleep(1);
      if (++it > (1u << 22) || ((it & 255u) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
        if (lane == 0) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return true;
      }
      return false;
    };
    float g[NPC];
    float sqs = 0.f;
#pragma unroll
    for (int j = 0; j < NPC; ++j) {
      const int i = i0 + tid + j * NT;
      float acc = 0.f;
      if (i0 + (tid & ~63) + j * NT < i1) {   // (wave-uniform: some lane of the wave has an element)
        const u64* col = slabs_s + min(i, o.total - 1);
        for (int sb = 0; sb < nblk && !fail; sb += 8) {
          u64 t[8];
          unsigned it = 0;
          for (;;) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
              t[u] = __hip_atomic_load(col + (long long)min(sb + u, nblk - 1) * SW, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            bool ok = true;
#pragma unroll
            for (int u = 0; u < 8; ++u) ok = ok && (unsigned)(t[u] >> 32) == seq_out;
            if (__all(ok)) break;
            if (timed_out(it)) { fail = true; break; }
          }
#pragma unroll
          for (int u = 0; u < 8; ++u)
            if (sb + u < nblk) acc += __uint_as_float((unsigned)t[u]);
        }
        if (i < i1) sqs += acc * acc;
      }
      g[j] = acc;
    }
    if (bid == 0 && stats != nullptr && (tid & ~63) < nblk * 8) {   // (wave-uniform) slab q's statistics slot: word P + slot
      const int q = min(tid >> 3, nblk - 1), slot = tid & 7;
      const bool mine = tid < nblk * 8 && slot < 5;
      const u64* wp = slabs_s + (long long)q * SW + o.total + (mine ? slot : 0);
      u64 t;
      unsigned it = 0;
      for (;;) {
        t = __hip_atomic_load(wp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__all(!mine || (unsigned)(t >> 32) == seq_out)) break;
        if (timed_out(it)) { fail = true; break; }
      }
      if (tid < nblk * 8) s_stat[tid] = __uint_as_float((unsigned)t);
    }
    EP_TS(1);
    {
      const float part = block_sum<NT>(sqs, lds);
      if (tid == 0) ll_store_agent(sq64 + (seq_out & 1u) * 64 + bid, part, seq_out);
    }
    EP_TS(2);
    // ---- B2: the G partial sums of squares -> norm, clip coefficient; Adam on the own chunk; loss statistics
    if (tid < 64) {
      const u64* wp = sq64 + (seq_out & 1u) * 64 + min(tid, nwg - 1);
      u64 t;
      unsigned it = 0;
      for (;;) {
        t = __hip_atomic_load(wp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__all((unsigned)(t >> 32) == seq_out)) break;
        if (timed_out(it)) { fail = true; break; }
      }
      s_part[tid] = tid < nwg ? __uint_as_float((unsigned)t) : 0.f;
    }
    if (__any(fail) && lane == 0) s_fail = 1;
    __syncthreads();
    if (s_fail) return;
    EP_TS(3);
    float total_sq = 0.f;
    for (int q = 0; q < nwg; ++q) total_sq += s_part[q];
    const float total_norm = sqrtf(total_sq);
    const float coef = fminf(max_norm / (total_norm + 1e-6f), 1.0f);   // torch.nn.utils.clip_grad_norm_
    {
      const float step_size = st.step_size[k], bc2_sqrt = st.bc2_sqrt[k];
      u64* dst = par64 + (long long)(seq_out & 1u) * o.total;
#pragma unroll
      for (int j = 0; j < NPC; ++j) {
        const int i = i0 + tid + j * NT;
        if (i < i1) {
          const float gi = g[j] * coef;
          const float mi = m_[j] + (gi - m_[j]) * (1.f - beta1);
          const float vi = v_[j] * beta2 + (1.f - beta2) * gi * gi;
          const float denom = sqrtf(vi) / bc2_sqrt + eps;
          const float pn = p_[j] - step_size * (mi / denom);
          ll_store_agent(dst + i, pn, seq_out);
          p_[j] = pn;
          m_[j] = mi;
          v_[j] = vi;
        }
      }
    }
    if (bid == 0 && stats != nullptr) {
      __shared__ float s_st[5];
      if (tid >= 64 && tid < 69) {   // one lane of the second wave per statistic, partials in row-block order
        const int kk = tid - 64;
        float sv = 0.f;
        for (int q = 0; q < nblk; ++q) sv += s_stat[q * 8 + kk];
        sv *= 1.f / (float)b;
        stats[(long long)mb * 8 + kk] = sv;
        s_st[kk] = sv;
      }
      __syncthreads();
      if (tid == 0) {
        stats[(long long)mb * 8 + 5] = s_st[0] + ent_coef * s_st[2] + vf_coef * s_st[1];  // loss
        stats[(long long)mb * 8 + 6] = total_norm;
        stats[(long long)mb * 8 + 7] = coef;
      }
    }
    EP_TS(4);
  }
  // the launch's last parameters and moments back to memory (torch layout + the transposed shadow copy)
#pragma unroll
  for (int j = 0; j < NPC; ++j) {
    const int i = bid * chunk + (int)threadIdx.x + j * NT;
    if (i < min(o.total, (bid + 1) * chunk)) {
      m[i] = m_[j];
      v[i] = v_[j];
      P[i] = p_[j];
      int dst = i;
      auto tr = [&](int b0, int rows_, int cols) {
        if (i >= b0 && i < b0 + rows_ * cols) {
          const int r = (i - b0) / cols, cc = (i - b0) % cols;
          dst = b0 + cc * rows_ + r;
        }
      };
      tr(o.pW1, H, D); tr(o.pW2, H, H); tr(o.vW1, H, D); tr(o.vW2, H, H);
      Pt[dst] = p_[j];
    }
  }
#undef EP_TS
}

// ---------------------------------------------------------------------------------------------
// 64-wide towers (SB3's default `MlpPolicy`): the transposed chain of the 32-wide persistent kernel for ONE tower per workgroup
// (`ppo_epoch_ll2_kernel` below). A group of 16 rows runs the activation chain
//       x -> a1 -> a2 -> head -> per-row loss -> d head -> dz2 -> dz1
// with features along the MFMA's M index, the rows along N and the k index of a step permuted so that the accumulator of one
// layer has the B operand's layout of the next (`mfma32_minibatch_chain`); the `[feature][row]` LDS tiles serve the
// weight-gradient tiles (ds_read_b128 of four consecutive rows) and hand a layer's outputs from the wave that formed them to
// the others of its row group. The tower's parameters are read from LDS images: W1 / W2 in torch layout with padded rows (a
// forward fragment = the four consecutive INPUTS of an output row: one ds_read_b128), W2 transposed likewise for the backward
// pass, the head in both orientations. (Round 5's form -- one wave per SIMD, every wave all 64 output features of its rows,
// activations in registers from x to dz1: 20.6 us per step -- was retired in round 6 once the eight-wave form below, bit-identical
// to it, had replaced it everywhere: `profiles/r06_mlp64.md`, DESIGN_HISTORY.md.)
template <int KT1, int RB = 64>
struct T64Geo {   // LDS carve-up (floats; every offset a multiple of 4) of one tower workgroup: compile-time but for the
                  // staging area's pieces, whose sizes follow the observation / action widths. RB = rows of the block
  static constexpr int RS = 68;    // row stride of the 64-wide weight images
  static constexpr int TS = RB + 4;   // row stride of the [feature][row] tiles (68 at 64 rows: the images' stride)
  static constexpr int HT = 20;    // row stride of the transposed head image [64 hidden][16 actions + 4]
  static constexpr int DP = 16 * KT1 + 4;   // first-layer image row: a whole number of K tiles + one quad -- an odd number of
                                            // quads (eight lanes' b128 reads hit 32 banks); columns >= D stay zero
  // tiles
  static constexpr int x = 0;
  static constexpr int a1 = x + 16 * KT1 * TS;
  static constexpr int a2 = a1 + 64 * TS;
  static constexpr int dz2 = a2 + 64 * TS;
  static constexpr int dz1 = dz2 + 64 * TS;
  static constexpr int dout = dz1 + 64 * TS;
  static constexpr int aux = dout + 16 * TS;
  static constexpr int misc = aux + 16 * TS;
  static constexpr int scratch = misc + 8 * TS;
  // images
  static constexpr int W1 = scratch + 64;
  static constexpr int W2 = W1 + 64 * DP;
  static constexpr int W2T = W2 + 64 * RS;
  static constexpr int b1 = W2T + 64 * RS;
  static constexpr int b2 = b1 + 64;
  static constexpr int HW = b2 + 64;
  static constexpr int HWT = HW + 16 * RS;
  static constexpr int hb = HWT + 64 * HT;
  static constexpr int ls = hb + 16;       // log_std [16] | 1 / sd^2 [16] | log sd [16] (policy tower of a Box space)
  // staging of the next minibatch's rows; its statistics slot
  static constexpr int ring = ls + 48;
  static constexpr int soldlp = ring + 2 * MAXD + 8;   // old log-prob | advantage | return: three consecutive 64-float areas
  static constexpr int sadv = soldlp + 64;
  static constexpr int sret = sadv + 64;
  static constexpr int sx = sret + 64;                 // [RB][D] raw rows, packed
  int sact, total;                                     // [RB][aw]
  __host__ __device__ T64Geo(int D, int aw) {
    sact = sx + ((RB * D + 3) & ~3);
    total = sact + ((RB * aw + 3) & ~3);
  }
};
struct T64Out {   // word index (inside the workgroup's slab) of each piece of the tower's gradient; the loss-statistic tail
  int W1, b1, W2, b2, HW, Hb, LS, tail;
  bool zero_tail;   // the slab is this tower's alone: the tail slots of the other tower's statistics are written as zeros
};
// ---------------------------------------------------------------------------------------------
// The chain: every layer's four output tiles are split between the NH = 2 (or 4) waves of a group of 16 rows. Wave
// w = (q, h) = (w % NQ, w / NQ), NQ = RB / 16 row groups, owns rows 16 q .. of the block and the output tiles 2 h, 2 h + 1 of
// every layer (features 32 h .. 32 h + 31; quarters: tile h). The k index of
// a layer runs over all 64 features of the layer below: a wave's own half is in its registers, its partner's half comes from
// the `[feature][row]` tile the partner writes anyway (one workgroup barrier per layer; the next layer's weight fragments are
// requested ahead of it). Head and per-row losses (16 MFMAs, no weights to split) are computed by both waves of a pair. Two
// forms:
//   * RB = 64, eight waves (two per SIMD, the second one issuing under the first one's latencies). Every tile is accumulated
//     by the same MFMAs in the same order as in round 5's four-wave form: the gradients were bit-identical. Weight-gradient tiles:
//     dW2's four input tiles go two per wave; h = 0 takes dW1's first K tile and its bias, h = 1 dW1's second K tile
//     (observation widths > 16), the second layer's bias and the head's tile.
//   * RB = 32, four waves (one per SIMD): twice the workgroups per minibatch, each with half the chain AND half the
//     weight-gradient MFMAs (contractions over 32 rows) per compute unit -- at the price of twice the slabs in phase B1 and
//     twice the readers of the parameter words. Wave w takes output tile w of every weight-gradient product.
//   * RB = 32, EIGHT waves = two row groups x four feature QUARTERS (NH = 4; the default up to 1 024-row minibatches): one
//     output tile per wave and layer, all four k tiles of the layer below read from the `[feature][row]` tile behind the
//     barrier; head and per-row losses by the row group's first wave alone, d head / dv handed to the other three through LDS
//     behind one more barrier; weight-gradient tiles by the wave's number as in the 64-row eight-wave form.
// The row tiles' stride is TS = RB + 4 floats (an odd number of quads: sixteen lanes' ds_read_b128 of consecutive features hit
// distinct bank quads), the 64-wide weight images keep RS = 68.
template <int KT1, int RB, int NW, class Mid>
__device__ __forceinline__ void t64h_tower_minibatch(
    const ia_policy_desc& d, const T64Geo<KT1, RB>& G, const T64Out& Lc, const int tw, float* __restrict__ lds_in, const int row0,
    const int b, const float adv_mean, const float adv_std, const int normalize_adv, const float clip, const float ent_coef,
    const float vf_coef, unsigned long long* __restrict__ slab, const unsigned seq, Mid&& mid,
    long long* __restrict__ ts /* measurement (nullable): shader clocks of the phases, thread 0 */, const int oz) {
  constexpr int RS = T64Geo<KT1, RB>::RS, TS = T64Geo<KT1, RB>::TS;   // strides of the weight images / of the row tiles
  constexpr int NQ = RB / 16, SQ = RB / 16;                           // row groups of the block; 16-row steps of a contraction
  constexpr int NH = NW / NQ, TPW = 4 / NH;                           // waves per row group (feature parts); output tiles per wave
  static_assert(NW == NH * NQ && (NH == 2 || NH == 4), "two or four waves (feature halves / quarters) per group of 16 rows");
#define T64C_TS(slot) do { if (ts != nullptr && threadIdx.x == 0) ts[slot] = clock64(); } while (0)
  T64C_TS(0);
  float* __restrict__ lds = lds_in + oz;
  const int tid = threadIdx.x + oz, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = w % NQ, h = w / NQ;
  const bool hi = h != 0;
  const int li = lane & 15, lk = lane >> 4;
  const int D = d.obs_dim, A = d.act_dim;
  const int aw = d.discrete ? 1 : A;
  const float invB = 1.f / (float)b;
  auto rd4 = [&](const float* p) { return *reinterpret_cast<const f32x4*>(p); };
  auto put = [&](int idx, float v) { ll_store_agent(slab + idx, v, seq); };
  auto sel4 = [&](const f32x4& own, const f32x4& other, bool own_first) {
    f32x4 o;
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = own_first ? own[r] : other[r];
    return o;
  };
  const int lrow = q * 16 + li;
  const bool valid = row0 + lrow < b;
  const int T0 = 16 * TPW * h;         // first feature of the wave's tiles
  const int P0 = 32 - T0;              // ... of its partner's (feature halves)
  const bool headw = NH == 2 || h == 0;   // (wave-uniform) this wave runs the head and the per-row losses
  float* const colp = lds + q * 16 + li;   // column (row of the minibatch) of the lane in every [feature][row] tile
  // per-row scalars of the loss (staged by the prefetch)
  float r_oldlp = 0.f, r_adv = 0.f, r_ret = 0.f, r_act[4] = {0.f, 0.f, 0.f, 0.f};
  if (tw == 0) {
    r_oldlp = lds[G.soldlp + lrow];
    r_adv = lds[G.sadv + lrow];
#pragma unroll
    for (int j = 0; j < 4; ++j) r_act[j] = lds[G.sact + lrow * aw + min(4 * lk + j, aw - 1)];
  } else {
    r_ret = lds[G.sret + lrow];
  }
  // the partner's half of a [feature][row] tile (features P0 + 16 j + 4 lk + r of the lane's row), and the four k tiles of a
  // layer in feature order from the two halves
  auto partner = [&](int tile, f32x4 (&o)[2]) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) o[j][r] = colp[tile + (P0 + 16 * j + 4 * lk + r) * TS];
  };
  auto whole = [&](const f32x4 (&own)[2], const f32x4 (&oth)[2], f32x4 (&o)[4]) {
    o[0] = sel4(own[0], oth[0], !hi);
    o[1] = sel4(own[1], oth[1], !hi);
    o[2] = sel4(own[0], oth[0], hi);
    o[3] = sel4(own[1], oth[1], hi);
  };
  // all four k tiles of a layer below (behind the workgroup barrier that follows its tiles' stores). Halves: the own two from
  // the registers, the partner's two from the tile; quarters: all four from the tile (16 conflict-free ds_read_b32)
  auto gather4 = [&](int tile, const f32x4 (&own)[TPW], f32x4 (&o)[4]) {
    if constexpr (NH == 2) {
      f32x4 oth[2];
      partner(tile, oth);
      whole(own, oth, o);
    } else {
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) o[kt][r] = colp[tile + (16 * kt + 4 * lk + r) * TS];
    }
  };
  // ---- layer 1: tiles 2 h, 2 h + 1 of a1^T = tanh(W1 x^T + b1)
  f32x4 a1o[TPW], a2o[TPW], a1[4], a2[4];
  f32x4 fW2[4][TPW], b2c[TPW];
  {
    f32x4 fW1[KT1][TPW], b1c[TPW];
    float xb[KT1][4];
#pragma unroll
    for (int kt = 0; kt < KT1; ++kt)
#pragma unroll
      for (int j = 0; j < TPW; ++j) fW1[kt][j] = rd4(lds + G.W1 + (T0 + 16 * j + li) * G.DP + 16 * kt + 4 * lk);
#pragma unroll
    for (int j = 0; j < TPW; ++j) b1c[j] = rd4(lds + G.b1 + T0 + 16 * j + 4 * lk);
#pragma unroll
    for (int kt = 0; kt < KT1; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) xb[kt][r] = colp[G.x + (16 * kt + 4 * lk + r) * TS];
    __builtin_amdgcn_sched_barrier(0);
    f32x4 acc[TPW][2];
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
      acc[j][0] = b1c[j];
      acc[j][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    auto l1_tile = [&](const int j) {
#pragma unroll
      for (int kt = 0; kt < KT1; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[j][(kt * 4 + r) & 1] = mfma16(fW1[kt][j][r], xb[kt][r], acc[j][(kt * 4 + r) & 1]);
    };
    auto l1_tanh = [&](const int j) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        a1o[j][r] = fast_tanh(acc[j][0][r] + acc[j][1][r]);
        colp[G.a1 + (T0 + 16 * j + 4 * lk + r) * TS] = a1o[j][r];
      }
    };
#pragma unroll
    for (int j = 0; j < TPW; ++j) l1_tile(j);
    // (the second layer's fragments of the wave's output rows: in flight under the tanh and the barrier)
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int j = 0; j < TPW; ++j) fW2[kt][j] = rd4(lds + G.W2 + (T0 + 16 * j + li) * RS + 16 * kt + 4 * lk);
#pragma unroll
    for (int j = 0; j < TPW; ++j) b2c[j] = rd4(lds + G.b2 + T0 + 16 * j + 4 * lk);
#pragma unroll
    for (int j = 0; j < TPW; ++j) l1_tanh(j);
  }
  __syncthreads();
  T64C_TS(1);
  gather4(G.a1, a1o, a1);
  // ---- layer 2: tiles 2 h, 2 h + 1 of a2^T = tanh(W2 a1^T + b2)
  f32x4 fHead[4], hbias;
  {
    f32x4 acc[TPW][2];
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
      acc[j][0] = b2c[j];
      acc[j][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    auto l2_tile = [&](const int j) {
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[j][r & 1] = mfma16(fW2[kt][j][r], a1[kt][r], acc[j][r & 1]);
    };
    auto l2_tanh = [&](const int j) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        a2o[j][r] = fast_tanh(acc[j][0][r] + acc[j][1][r]);
        colp[G.a2 + (T0 + 16 * j + 4 * lk + r) * TS] = a2o[j][r];
      }
    };
#pragma unroll
    for (int j = 0; j < TPW; ++j) l2_tile(j);
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) fHead[kt] = rd4(lds + G.HW + li * RS + 16 * kt + 4 * lk);
    hbias = rd4(lds + G.hb + 4 * lk);
#pragma unroll
    for (int j = 0; j < TPW; ++j) l2_tanh(j);
  }
  __syncthreads();
  T64C_TS(2);
  if (headw) gather4(G.a2, a2o, a2);
  // ---- head (both waves of a pair): hout[r] = output 4 lk + r of row li (value: lane group 0, register 0)
  float hout[4] = {0.f, 0.f, 0.f, 0.f};
  if (headw) {
    f32x4 acc[2] = {hbias, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r & 1] = mfma16(fHead[kt][r], a2[kt][r], acc[r & 1]);
#pragma unroll
    for (int r = 0; r < 4; ++r) hout[r] = acc[0][r] + acc[1][r];
  }
  T64C_TS(3);
  // backward fragments of the wave's tiles: the head's weights of the lane's outputs T0 + 16 j + 4 lk + r; W2 transposed
  f32x4 fHeadT[TPW];
#pragma unroll
  for (int j = 0; j < TPW; ++j)
    fHeadT[j] = tw == 0 ? rd4(lds + G.HWT + (T0 + 16 * j + li) * T64Geo<KT1, RB>::HT + 4 * lk) : rd4(lds + G.HW + T0 + 16 * j + 4 * lk);
  f32x4 fW2T[4][TPW];
#pragma unroll
  for (int kt = 0; kt < 4; ++kt)
#pragma unroll
    for (int j = 0; j < TPW; ++j) fW2T[kt][j] = rd4(lds + G.W2T + (T0 + 16 * j + li) * RS + 16 * kt + 4 * lk);
  // ---- per-row losses (the expressions of `mfma32_minibatch_chain`; the group's first wave leaves the rows' pieces in LDS)
  auto xchg16 = [](float v, float& a, float& bq) {
    const auto p2 = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    a = __uint_as_float(p2[0]);
    bq = __uint_as_float(p2[1]);
  };
  auto xchg32 = [](float v, float& a, float& bq) {
    const auto p2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    a = __uint_as_float(p2[0]);
    bq = __uint_as_float(p2[1]);
  };
  auto group_sum = [&](float v) {
    float a, bq;
    xchg16(v, a, bq);
    v = a + bq;
    xchg32(v, a, bq);
    return a + bq;
  };
  auto group_max = [&](float v) {
    float a, bq;
    xchg16(v, a, bq);
    v = fmaxf(a, bq);
    xchg32(v, a, bq);
    return fmaxf(a, bq);
  };
  float dout[4] = {0.f, 0.f, 0.f, 0.f}, dvb = 0.f;
  if (!headw) {
    // (feature quarters: the row group's first wave runs head and losses alone and leaves d head / dv in LDS, see below)
  } else if (tw == 0) {
    const f32x4 c_ivar = rd4(lds + G.ls + 16 + 4 * lk), c_logsd = rd4(lds + G.ls + 32 + 4 * lk);
    float logp = 0.f, entropy = 0.f, lse = 0.f;
    int act_i = 0;
    if (d.discrete) act_i = (int)r_act[0];
    if (!d.discrete) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (4 * lk + j < A) {
          const float diff = r_act[j] - hout[j];
          logp += -(diff * diff) * (0.5f * c_ivar[j]) - c_logsd[j] - LOG_SQRT_2PI;
          entropy += 0.5f + LOG_SQRT_2PI + c_logsd[j];
        }
      logp = group_sum(logp);
      entropy = group_sum(entropy);
    } else {
      float mx = -3.0e38f, o_act = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (4 * lk + j < A) {
          mx = fmaxf(mx, hout[j]);
          o_act += (4 * lk + j == act_i) ? hout[j] : 0.f;
        }
      mx = group_max(mx);
      o_act = group_sum(o_act);
      float se = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (4 * lk + j < A) se += expf(hout[j] - mx);
      lse = mx + logf(group_sum(se));
      logp = o_act - lse;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (4 * lk + j < A) {
          const float l = hout[j] - lse;
          entropy -= expf(l) * l;
        }
      entropy = group_sum(entropy);
    }
    float advn = r_adv;
    if (normalize_adv && b > 1) advn = (advn - adv_mean) / (adv_std + 1e-8f);
    const float log_ratio = logp - r_oldlp;
    const float ratio = expf(log_ratio);
    const float lo = 1.f - clip, hi_ = 1.f + clip;
    const float pl1 = advn * ratio;
    const float pl2 = advn * fminf(fmaxf(ratio, lo), hi_);
    const float g1 = pl1 < pl2 ? 1.f : (pl1 == pl2 ? 0.5f : 0.f);
    const float g2 = pl2 < pl1 ? 1.f : (pl1 == pl2 ? 0.5f : 0.f);
    const float inrange = (ratio >= lo && ratio <= hi_) ? 1.f : 0.f;
    const float dlogp = valid ? -invB * advn * (g1 + g2 * inrange) * ratio : 0.f;
    float* doutrow = lds + G.dout + lrow;   // (column a of this lane's row: [a * RS])
    float* auxrow = lds + G.aux + lrow;
    if (!d.discrete) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (4 * lk + j < A) {
          const float diff = r_act[j] - hout[j];
          dout[j] = dlogp * diff * c_ivar[j];
          if (!hi) {
            doutrow[(4 * lk + j) * TS] = dout[j];
            auxrow[(4 * lk + j) * TS] = valid ? dlogp * (diff * diff * c_ivar[j] - 1.f) - ent_coef * invB : 0.f;
          }
        }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (4 * lk + j < A) {
          const float l = hout[j] - lse, p = expf(l);
          const float dH = -p * (l + entropy);
          float g = dlogp * ((4 * lk + j == act_i ? 1.f : 0.f) - p);
          g += valid ? -ent_coef * invB * dH : 0.f;
          dout[j] = g;
          if (!hi) doutrow[(4 * lk + j) * TS] = g;
        }
    }
    if (lk == 0 && !hi) {
      float* mrow = lds + G.misc + lrow;
      mrow[2 * TS] = valid ? -fminf(pl1, pl2) : 0.f;                           // policy_gradient_loss
      mrow[3 * TS] = valid ? -entropy : 0.f;                                    // entropy_loss
      mrow[4 * TS] = valid ? (ratio - 1.f) - log_ratio : 0.f;                   // approx_kl
      mrow[5 * TS] = valid ? (fabsf(ratio - 1.f) > clip ? 1.f : 0.f) : 0.f;     // clip_fraction
    }
  } else {
    const float v = __shfl(hout[0], li, 64);   // (lane group 0, register 0 holds V(row li))
    const float verr = r_ret - v;
    dvb = valid ? vf_coef * 2.f * (v - r_ret) * invB : 0.f;
    if (lk == 0 && !hi) {
      lds[G.misc + 1 * TS + lrow] = dvb;
      lds[G.misc + 6 * TS + lrow] = valid ? verr * verr : 0.f;        // value_loss
    }
  }
  if constexpr (NH == 4) {   // d head^T (policy: the [action][row] tile, zero beyond the actions) / dv of the lane's row
    __syncthreads();
    if (tw == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) dout[r] = colp[G.dout + (4 * lk + r) * TS];
    } else {
      dvb = lds[G.misc + 1 * TS + lrow];
    }
  }
  T64C_TS(4);
  // ---- tiles 2 h, 2 h + 1 of dz2^T = (W_head^T d head^T) * (1 - a2^2), then of dz1^T = (W2^T dz2^T) * (1 - a1^2)
  f32x4 dz2o[TPW];
  if (tw == 0) {
    f32x4 acc[TPW];
#pragma unroll
    for (int j = 0; j < TPW; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 4; ++r)   // (k-step r carries actions r, 4 + r, 8 + r, 12 + r; the image's columns >= A are zero)
#pragma unroll
      for (int j = 0; j < TPW; ++j) acc[j] = mfma16(fHeadT[j][r], dout[r], acc[j]);
#pragma unroll
    for (int j = 0; j < TPW; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) dz2o[j][r] = acc[j][r] * (1.f - a2o[j][r] * a2o[j][r]);
  } else {
#pragma unroll
    for (int j = 0; j < TPW; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) dz2o[j][r] = fHeadT[j][r] * dvb * (1.f - a2o[j][r] * a2o[j][r]);
  }
#pragma unroll
  for (int j = 0; j < TPW; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) colp[G.dz2 + (T0 + 16 * j + 4 * lk + r) * TS] = dz2o[j][r];
  __syncthreads();
  {
    f32x4 dz2[4];
    gather4(G.dz2, dz2o, dz2);
    f32x4 acc[TPW][2];
#pragma unroll
    for (int j = 0; j < TPW; ++j) acc[j][0] = acc[j][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto d1_tile = [&](const int j) {
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[j][r & 1] = mfma16(fW2T[kt][j][r], dz2[kt][r], acc[j][r & 1]);
    };
    auto d1_out = [&](const int j) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        colp[G.dz1 + (T0 + 16 * j + 4 * lk + r) * TS] = (acc[j][0][r] + acc[j][1][r]) * (1.f - a1o[j][r] * a1o[j][r]);
    };
#pragma unroll
    for (int j = 0; j < TPW; ++j) d1_tile(j);
#pragma unroll
    for (int j = 0; j < TPW; ++j) d1_out(j);
  }
  T64C_TS(5);
  __syncthreads();   // every row's activations and activation gradients are in LDS
  mid();
  T64C_TS(6);
  // ---- weight-gradient tiles: contractions over the block's RB rows (SQ steps of 16), independent per wave. Tile = 16 features
  // of U (the MFMA's M index) x 16 features of V (N); a lane reads four consecutive rows of its feature per ds_read_b128 (row
  // steps 4 sq .. + 3 of lane group lk are rows 16 sq + 4 lk ..: the same permutation on both operands); two accumulator
  // chains per tile
  auto tile2 = [&](const f32x4 (&u)[SQ], const float* __restrict__ V0, const float* __restrict__ V1, f32x4& r0, f32x4& r1) {
    const float* vp0 = V0 + li * TS + 4 * lk;
    const float* vp1 = V1 + li * TS + 4 * lk;
    f32x4 v0[SQ], v1[SQ];
#pragma unroll
    for (int sq = 0; sq < SQ; ++sq) {
      v0[sq] = rd4(vp0 + 16 * sq);
      v1[sq] = rd4(vp1 + 16 * sq);
    }
    __builtin_amdgcn_sched_barrier(0);
    f32x4 g0 = {0.f, 0.f, 0.f, 0.f}, g0b = g0, g1 = g0, g1b = g0;
#pragma unroll
    for (int sq = 0; sq < SQ; ++sq)
#pragma unroll
      for (int i = 0; i < 4; i += 2) {
        g0 = mfma16(u[sq][i], v0[sq][i], g0);
        g1 = mfma16(u[sq][i], v1[sq][i], g1);
        g0b = mfma16(u[sq][i + 1], v0[sq][i + 1], g0b);
        g1b = mfma16(u[sq][i + 1], v1[sq][i + 1], g1b);
      }
    r0 = g0 + g0b;
    r1 = g1 + g1b;
  };
  auto tile1 = [&](const f32x4 (&u)[SQ], const float* __restrict__ V0, f32x4& r0) {
    const float* vp0 = V0 + li * TS + 4 * lk;
    f32x4 v0[SQ];
#pragma unroll
    for (int sq = 0; sq < SQ; ++sq) v0[sq] = rd4(vp0 + 16 * sq);
    __builtin_amdgcn_sched_barrier(0);
    f32x4 g0 = {0.f, 0.f, 0.f, 0.f}, g0b = g0;
#pragma unroll
    for (int sq = 0; sq < SQ; ++sq)
#pragma unroll
      for (int i = 0; i < 4; i += 2) {
        g0 = mfma16(u[sq][i], v0[sq][i], g0);
        g0b = mfma16(u[sq][i + 1], v0[sq][i + 1], g0b);
      }
    r0 = g0 + g0b;
  };
  auto load_u = [&](const float* __restrict__ U, f32x4 (&u)[SQ]) {
    const float* up = U + li * TS + 4 * lk;
#pragma unroll
    for (int sq = 0; sq < SQ; ++sq) u[sq] = rd4(up + 16 * sq);
  };
  // a feature's sum over the block's rows (optionally weighted by a row vector): lane (j, quarter) = (lane & 15, lane >> 4)
  auto colsum16 = [&](const float* __restrict__ tile /* 16 features */, const float* __restrict__ wrow /* nullable */) {
    const float* cp = tile + (lane & 15) * TS + (lane >> 4) * (RB / 4);
    f32x4 t[SQ], wv[SQ];
#pragma unroll
    for (int i = 0; i < SQ; ++i) {
      t[i] = rd4(cp + 4 * i);
      wv[i] = wrow ? rd4(wrow + (lane >> 4) * (RB / 4) + 4 * i) : f32x4{1.f, 1.f, 1.f, 1.f};
    }
    __builtin_amdgcn_sched_barrier(0);
    float s = 0.f;
    if (wrow) {
#pragma unroll
      for (int i = 0; i < SQ; ++i) s += (t[i][0] * wv[i][0] + t[i][1] * wv[i][1]) + (t[i][2] * wv[i][2] + t[i][3] * wv[i][3]);
    } else {
#pragma unroll
      for (int i = 0; i < SQ; ++i) s += (t[i][0] + t[i][1]) + (t[i][2] + t[i][3]);
    }
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    return s;   // (all four lanes of feature j hold the sum)
  };
  if constexpr (NW == 8) {
    const int jw = w & 3, iw = w >> 2, I0 = 32 * iw;   // output tile and pair of input tiles of this wave's dW2 products
    const bool second = iw != 0;
    // eight waves: dW2's four input tiles go two per wave; h = 0 takes dW1's first K tile and its bias, h = 1 dW1's second K
    // tile (observation widths > 16), the second layer's bias, the head's tile; the small column sums ride with h = 0
    {   // dW2[j][i] = sum_r dz2[r][j] a1[r][i]: wave w takes output rows j = 16 (w & 3) .., input tiles 2 (w >> 2) and 2 (w >> 2) + 1
      f32x4 u[SQ];
      load_u(lds + G.dz2 + 16 * jw * TS, u);
      f32x4 g0, g1;
      tile2(u, lds + G.a1 + I0 * TS, lds + G.a1 + (I0 + 16) * TS, g0, g1);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        put(Lc.W2 + (16 * jw + 4 * lk + r) * 64 + I0 + li, g0[r]);
        put(Lc.W2 + (16 * jw + 4 * lk + r) * 64 + I0 + 16 + li, g1[r]);
      }
    }
    T64C_TS(7);
    if (!second) {   // dW1[j][c] = sum_r dz1[r][j] x[r][c], first K tile; the first layer's bias
      f32x4 u[SQ], g0;
      load_u(lds + G.dz1 + 16 * jw * TS, u);
      tile1(u, lds + G.x, g0);
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (li < D) put(Lc.W1 + (16 * jw + 4 * lk + r) * D + li, g0[r]);
      const float sb1 = colsum16(lds + G.dz1 + 16 * jw * TS, nullptr);
      if (lane < 16) put(Lc.b1 + 16 * jw + lane, sb1);
    } else {     // second K tile; the second layer's bias
      if (KT1 == 2) {
        f32x4 u[SQ], g1;
        load_u(lds + G.dz1 + 16 * jw * TS, u);
        tile1(u, lds + G.x + 16 * TS, g1);
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (16 + li < D) put(Lc.W1 + (16 * jw + 4 * lk + r) * D + 16 + li, g1[r]);
      }
      const float sb2 = colsum16(lds + G.dz2 + 16 * jw * TS, nullptr);
      if (lane < 16) put(Lc.b2 + 16 * jw + lane, sb2);
    }
    T64C_TS(8);
    if (tw == 0) {
      if (second) {   // head weights: dWa[a][h] = sum_r dout[r][a] a2[r][h], hidden tile q
        f32x4 u[SQ], g0;
        load_u(lds + G.dout, u);
        tile1(u, lds + G.a2 + 16 * jw * TS, g0);
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (4 * lk + r < A) put(Lc.HW + (4 * lk + r) * 64 + 16 * jw + li, g0[r]);
      } else if (jw == 0) {   // action_net bias
        const float s = colsum16(lds + G.dout, nullptr);
        if (lane < A) put(Lc.Hb + lane, s);
      } else if (jw == 1) {   // log_std
        if (!d.discrete) {
          const float s = colsum16(lds + G.aux, nullptr);
          if (lane < A) put(Lc.LS + lane, s);
        }
      } else if (jw == 2) {   // loss statistics: misc columns 2..5 -> tail slots {0 pg, 2 ent, 3 kl, 4 clip}
        const float s = colsum16(lds + G.misc, nullptr);   // (features 0..7 of the misc tile; 8..15 read the next tile: unused)
        if (lane >= 2 && lane < 6) put(Lc.tail + (lane == 2 ? 0 : lane - 1), s);
      } else {
        if (Lc.zero_tail && lane < 4) put(Lc.tail + (lane == 0 ? 1 : lane + 4), 0.f);   // (slots 1, 5, 6, 7: not this tower's)
      }
    } else {
      if (second) {   // value_net: dcW[h] = sum_r dv[r] a2[r][h] -- one useful row of a tile: a weighted column sum instead
        const float s = colsum16(lds + G.a2 + 16 * jw * TS, lds + G.misc + 1 * TS);
        if (lane < 16) put(Lc.HW + 16 * jw + lane, s);
      } else if (jw == 0) {   // value_net bias = sum_r dv[r]; value_loss -> tail slot 1
        const float sm = colsum16(lds + G.misc, nullptr);
        if (lane == 1) put(Lc.Hb, sm);
        if (lane == 6) put(Lc.tail + 1, sm);
      } else if (jw == 1) {
        if (Lc.zero_tail && lane < 7) put(Lc.tail + (lane == 0 ? 0 : lane + 1), 0.f);   // (slots 0, 2..7: not this tower's)
      }
    }
  } else {
    // four waves (32-row blocks): wave w takes output tile w of every product
    {
      f32x4 u[SQ];
      load_u(lds + G.dz2 + 16 * w * TS, u);
#pragma unroll
      for (int it = 0; it < 4; it += 2) {
        f32x4 g0, g1;
        tile2(u, lds + G.a1 + 16 * it * TS, lds + G.a1 + 16 * (it + 1) * TS, g0, g1);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          put(Lc.W2 + (16 * w + 4 * lk + r) * 64 + 16 * it + li, g0[r]);
          put(Lc.W2 + (16 * w + 4 * lk + r) * 64 + 16 * (it + 1) + li, g1[r]);
        }
      }
      const float sb2 = colsum16(lds + G.dz2 + 16 * w * TS, nullptr);
      if (lane < 16) put(Lc.b2 + 16 * w + lane, sb2);
    }
    T64C_TS(7);
    {
      f32x4 u[SQ], g0;
      load_u(lds + G.dz1 + 16 * w * TS, u);
      tile1(u, lds + G.x, g0);
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (li < D) put(Lc.W1 + (16 * w + 4 * lk + r) * D + li, g0[r]);
      if (KT1 == 2) {
        f32x4 g1;
        tile1(u, lds + G.x + 16 * TS, g1);
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (16 + li < D) put(Lc.W1 + (16 * w + 4 * lk + r) * D + 16 + li, g1[r]);
      }
      const float sb1 = colsum16(lds + G.dz1 + 16 * w * TS, nullptr);
      if (lane < 16) put(Lc.b1 + 16 * w + lane, sb1);
    }
    T64C_TS(8);
    if (tw == 0) {
      f32x4 u[SQ], g0;
      load_u(lds + G.dout, u);
      tile1(u, lds + G.a2 + 16 * w * TS, g0);
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (4 * lk + r < A) put(Lc.HW + (4 * lk + r) * 64 + 16 * w + li, g0[r]);
      if (w == 0) {   // action_net bias
        const float s = colsum16(lds + G.dout, nullptr);
        if (lane < A) put(Lc.Hb + lane, s);
      } else if (w == 1) {   // log_std
        if (!d.discrete) {
          const float s = colsum16(lds + G.aux, nullptr);
          if (lane < A) put(Lc.LS + lane, s);
        }
      } else if (w == 2) {   // loss statistics: misc columns 2..5 -> tail slots {0 pg, 2 ent, 3 kl, 4 clip}
        const float s = colsum16(lds + G.misc, nullptr);
        if (lane >= 2 && lane < 6) put(Lc.tail + (lane == 2 ? 0 : lane - 1), s);
      } else {
        if (Lc.zero_tail && lane < 4) put(Lc.tail + (lane == 0 ? 1 : lane + 4), 0.f);
      }
    } else {
      const float s = colsum16(lds + G.a2 + 16 * w * TS, lds + G.misc + 1 * TS);
      if (lane < 16) put(Lc.HW + 16 * w + lane, s);
      if (w == 0) {   // value_net bias = sum_r dv[r]; value_loss -> tail slot 1
        const float sm = colsum16(lds + G.misc, nullptr);
        if (lane == 1) put(Lc.Hb, sm);
        if (lane == 6) put(Lc.tail + 1, sm);
      } else if (w == 1) {
        if (Lc.zero_tail && lane < 7) put(Lc.tail + (lane == 0 ? 0 : lane + 1), 0.f);
      }
    }
  }
  T64C_TS(9);
  __syncthreads();
  T64C_TS(10);
#undef T64C_TS
}

// ---------------------------------------------------------------------------------------------
// Round 5: the word-exchange epoch kernel with the TRANSPOSED chain (`t64h_tower_minibatch`) as its
// gradient phase -- observation widths up to 32. Exchange, sequence numbers, chunk owners and Adam are those of
// `ppo_epoch_ll_kernel` (same word areas, same sums in the same order in phases B1 / B2); what changed is phase A:
//   * the tower's polled parameter words go straight to their places in the chain's LDS images (W1 / W2 in torch layout
//     with padded rows, W2 transposed, the head in both orientations: every weight fragment one ds_read_b128), places
//     worked out once per launch;
//   * the minibatch's rows are plain loads of the gathered, contiguous rows issued a step AHEAD (parked in LDS behind the
//     chain's barrier, normalised into the `[feature][row]` x tile while the partial sums of squares travel);
//   * a wave's own output tiles stay in its registers through x -> a1 -> a2 -> head -> loss -> dz2 -> dz1 (the other tiles of a
//     layer come from the row tiles); the tiles of a wave's weight gradient share their first operand; one-row products
//     (value head) are weighted column sums;
//   * round 6: NW waves per tower workgroup on RB-row blocks (`t64h_tower_minibatch`'s three forms); phase B1 requests ALL slabs'
//     words of an element together when there are more than sixteen (32-row blocks of a 1 024-row minibatch).
// A tower workgroup that was tried in between -- parameters resident in LDS for the launch, every workgroup stepping its
// whole tower, two hops -- lost: Adam on 5.7 k parameters by 256 threads and 22-30 words per thread in both hops cost more
// than the parameter hand-off saves (28.2 against 26.5 us per step; `profiles/r05_mlp64.md`).
template <int KT1, int NW, int RB>
__global__ __launch_bounds__(64 * NW) void ppo_epoch_ll2_kernel(
    ia_policy_desc d, float* __restrict__ P, float* __restrict__ Pt, float* __restrict__ m, float* __restrict__ v,
    const float* __restrict__ nm_in, const float* __restrict__ nv_in, const float* __restrict__ obs,
    const float* __restrict__ actions, const float* __restrict__ old_logp, const float* __restrict__ adv,
    const float* __restrict__ ret, long long total_rows, int batch_size, int T, int n_envs, int normalize_adv,
    float clip, float ent_coef, float vf_coef, float max_norm, float beta1, float beta2, float eps,
    float* __restrict__ ws, EpochLl ll, const float* __restrict__ seq, int snap, float* __restrict__ stats, EpochSteps st,
    long long* __restrict__ dbg /* measurement: [0..5] += 100 MHz ticks of workgroup 0 in {A, slab poll + sum, sum of squares
                                   published, poll of the sums of squares, Adam + publish, -}; [16..], [32..]: shader clocks of
                                   the chain's phases (policy / value tower of row block 0, last step) */) {
  typedef unsigned long long u64;
  constexpr int H = 64, NT = 64 * NW;                        // NW = 4: round 5's chain, one wave per SIMD; 8: two (`t64h_...`)
  constexpr int RS = T64Geo<KT1, RB>::RS, TS = T64Geo<KT1, RB>::TS;
  constexpr int NLL = (64 * 16 * KT1 + 64 + 4096 + 64 + NT - 1) / NT;   // words of the tower's layers per thread (64 D + 64 + 4096 + 64)
  constexpr int NB = (MAXA * H + MAXA + NT - 1) / NT;        // words of its head per thread
  extern __shared__ __attribute__((aligned(16))) float lds_raw[];
  float* const lds0 = lds_raw + (((16 - (__builtin_amdgcn_groupstaticsize() & 15)) & 15) >> 2);
  float* lds = lds0;
  __shared__ int s_fail;
  __shared__ float s_part[64];
  __shared__ float s_stat[32 * 8];
  long long tprev = 0;
#define EP_TS(slot)                                                       \
  do {                                                                    \
    if (dbg != nullptr && blockIdx.x == 0 && threadIdx.x == 0) {          \
      const long long tn = wall_clock64();                                \
      dbg[slot] += tn - tprev;                                            \
      tprev = tn;                                                         \
    }                                                                     \
  } while (0)
  if (dbg != nullptr && blockIdx.x == 0 && threadIdx.x == 0) tprev = wall_clock64();
  const int bid = blockIdx.x, nwg = gridDim.x, nrb = nwg >> 1, rb = bid >> 1, tower = bid & 1;
  const int D = d.obs_dim, A = d.act_dim, aw = d.discrete ? 1 : A;
  const PolOff o = pol_offsets(D, A, H, d.discrete);
  const T64Geo<KT1, RB> G(D, aw);
  const int SW = o.total + 8;
  u64* slabs64 = ll.base;
  u64* sq64 = slabs64 + 2LL * nrb * SW;
  u64* par64 = sq64 + 2 * 64;
  unsigned* err = reinterpret_cast<unsigned*>(ws) + 5;
  const int chunk = (o.total + nwg - 1) / nwg;
  constexpr int NPC = 1024 / NT;   // parameters of the chunk per thread (chunk <= 1024)
  int tid = threadIdx.x, lane = tid & 63;   // (re-formed every step behind an opaque zero: see the step loop)
  if (tid == 0) s_fail = 0;
  for (int e = tid; e < G.total; e += NT) lds[e] = 0.f;
  // ---- the tower's pieces of the parameter vector and where each of the thread's polled words goes in the LDS images
  // (first place | second place << 16, 0xffff: none): the same every step
  const int tlen = H * D + H + H * H + H, t0 = tower ? o.vW1 : o.pW1;
  const int segB0 = tower ? o.cW : o.aW, nB = (tower ? o.total : o.cW) - segB0;
  unsigned dA[NLL], dB[NB];
#pragma unroll
  for (int i = 0; i < NLL; ++i) {
    const int e = tid + i * NT;
    unsigned da = 0xffffu, db = 0xffffu;
    if (e < H * D) {
      const int r = e / D;
      da = G.W1 + r * G.DP + (e - r * D);
    } else if (e < H * D + H) {
      da = G.b1 + (e - H * D);
    } else if (e < H * D + H + H * H) {
      const int j = e - (H * D + H), r = j >> 6, c = j & 63;
      da = G.W2 + r * RS + c;
      db = G.W2T + c * RS + r;
    } else if (e < tlen) {
      da = G.b2 + (e - (H * D + H + H * H));
    }
    dA[i] = da | (db << 16);
  }
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int j = tid + i * NT;
    unsigned da = 0xffffu, db = 0xffffu;
    if (tower == 0) {
      if (j < A * H) {
        da = G.HW + (j >> 6) * RS + (j & 63);
        db = G.HWT + (j & 63) * T64Geo<KT1, RB>::HT + (j >> 6);
      } else if (j < nB) {
        da = G.hb + (j - A * H);
      }
    } else if (j < H) {
      da = G.HW + j;
    } else if (j < nB) {
      da = G.hb;
    }
    dB[i] = da | (db << 16);
  }
  const T64Out out{tower ? o.vW1 : o.pW1, tower ? o.vb1 : o.pb1, tower ? o.vW2 : o.pW2, tower ? o.vb2 : o.pb2,
                   tower ? o.cW : o.aW,   tower ? o.cb : o.ab,   d.discrete ? 0 : o.log_std, o.total, false};
  // the workgroup's chunk of the parameters and of Adam's moments: registers across the launch (nobody else writes it)
  float m_[NPC], v_[NPC], p_[NPC];
#pragma unroll
  for (int j = 0; j < NPC; ++j) {
    const int i = min(bid * chunk + tid + j * NT, o.total - 1);
    m_[j] = m[i];
    v_[j] = v[i];
    p_[j] = P[i];
  }
  {   // the parameters the first step reads
    u64* dst = par64 + (long long)(ll.seq0 & 1u) * o.total;
#pragma unroll
    for (int j = 0; j < NPC; ++j) {
      const int i = bid * chunk + tid + j * NT;
      if (i < min(o.total, (bid + 1) * chunk)) ll_store_agent(dst + i, p_[j], ll.seq0);
    }
  }
  // ---- rows of a minibatch: gathered and contiguous. `load_rows`: plain loads of block rb's rows of minibatch `mbi`, every
  // one UNCONDITIONAL at a clamped 32-bit offset from a uniform base (a load behind a branch is merged with the "not
  // loaded" value by a copy that waits for it; a per-thread choice of pointer makes the load a flat one behind 64-bit
  // address arithmetic); `park_rows` leaves them in the staging area; `stage_rows` normalises them into the x tile
  constexpr int NXR = (RB * 16 * KT1 + NT - 1) / NT;
  constexpr int NAR = (RB * MAXA + NT - 1) / NT;
  float pf_x[NXR], pf_a[NAR], pf_s[3], pf_r[3];
  auto load_rows = [&](int mbi) {
    long long start = (long long)mbi * batch_size + RB * rb;
    start = start < total_rows - 1 ? start : total_rows - 1;
    const float* xb0 = obs + start * D;
    const float* ab0 = actions + start * aw;
    const int rrem = (int)(total_rows - 1 - start);            // rows behind the block's first
    const int xrem = rrem * D + D - 1, arem = rrem * aw + aw - 1;
#pragma unroll
    for (int it = 0; it < NXR; ++it) pf_x[it] = xb0[min(tid + it * NT, xrem)];
#pragma unroll
    for (int it = 0; it < NAR; ++it) pf_a[it] = ab0[min(tid + it * NT, arem)];
    const int r = min(tid & (RB - 1), rrem);
    pf_s[0] = (old_logp + start)[r];
    pf_s[1] = (adv + start)[r];
    pf_s[2] = (ret + start)[r];
    // statistics of the minibatch: [0] adv mean, [1] adv std, [8 + c] feature mean, [8 + MAXD + c] feature variance (the
    // snapshot of the running statistics AFTER this minibatch's update when the call updates them, else the running ones)
    const float* sqp = seq + (long long)mbi * EPS_SEQ;
    const int e = min(tid, 2 * MAXD + 7), c = (max(e, 8) - 8) & (MAXD - 1);
    const float* nmp = d.has_norm ? nm_in : sqp;
    const float* nvp = d.has_norm ? nv_in : sqp;
    pf_r[0] = sqp[e];
    pf_r[1] = nmp[min(c, D - 1)];
    pf_r[2] = nvp[min(c, D - 1)];
  };
  auto park_rows = [&]() {
    const int nx = RB * D, na = RB * aw;
#pragma unroll
    for (int it = 0; it < NXR; ++it)
      if (tid + it * NT < nx) lds[G.sx + tid + it * NT] = pf_x[it];
#pragma unroll
    for (int it = 0; it < NAR; ++it)
      if (tid + it * NT < na) lds[G.sact + tid + it * NT] = pf_a[it];
    if (tid < 3 * RB)   // (three consecutive 64-float areas)
      lds[G.soldlp + 64 * (tid / RB) + (tid & (RB - 1))] = tid < RB ? pf_s[0] : (tid < 2 * RB ? pf_s[1] : pf_s[2]);
    if (tid < 2 * MAXD + 8) {
      const bool var = tid >= 8 + MAXD;
      float val = (snap || tid < 8) ? pf_r[0] : (var ? pf_r[2] : pf_r[1]);
      // (variance columns: 1 / sqrt(var + eps) once per column here -- the product below is within 1 ulp of the quotient
      //  `util/networks.py:91` forms, as in the 32-wide kernel -- instead of a square root and a division per element)
      if (var) val = 1.f / sqrtf(val + d.norm_eps);
      if (tid >= 8 && (!d.has_norm || ((tid - 8) & (MAXD - 1)) >= D)) val = 0.f;
      lds[G.ring + tid] = val;
    }
  };
  auto stage_rows = [&](int bn) {   // `bn`: rows of the staged minibatch
    const int S1 = (D + 3) >> 2;
    const int s1r = (65536 + S1 - 1) / S1;
    for (int g = tid; g < RB * S1; g += NT) {
      const int r = (g * s1r) >> 16, k0 = (g - r * S1) * 4;
      const bool rok = RB * rb + r < bn;
      float raw[4], mu[4], vr[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c = min(k0 + j, D - 1);
        raw[j] = lds[G.sx + r * D + c];
        mu[j] = lds[G.ring + 8 + c];
        vr[j] = lds[G.ring + 8 + MAXD + c];
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bool ok = rok && k0 + j < D;
        float val = raw[j];
        if (d.has_norm) val = (raw[j] - mu[j]) * vr[j];   // (`vr`: 1 / sqrt(var + eps), see park_rows)
        lds[G.x + (k0 + j) * TS + r] = ok ? val : 0.f;
      }
    }
  };
  auto rows_of = [&](int mbi) { return (int)min((long long)batch_size, total_rows - (long long)mbi * batch_size); };
  __syncthreads();
  if (st.n > 0 && rb < (rows_of(st.first) + RB - 1) / RB) {   // (block-uniform)
    load_rows(st.first);
    park_rows();
    __syncthreads();
    stage_rows(rows_of(st.first));
  }
  __syncthreads();
#pragma nounroll
  for (int k = 0; k < st.n; ++k) {
    int oz;   // opaque zero, refreshed every step: per-lane offsets are re-derived per step instead of being hoisted out of the
              // loop into spilled registers
    asm volatile("s_mov_b32 %0, 0" : "=s"(oz));
    tid = (int)threadIdx.x + oz;
    lane = tid & 63;
    lds = lds0 + oz;
    const unsigned seq_par = ll.seq0 + (unsigned)k, seq_out = seq_par + 1u;
    const int mb = st.first + k;
    const int b = rows_of(mb);
    const int nblk = (b + RB - 1) / RB;
    const bool more = k + 1 < st.n;
    const int bnext = more ? rows_of(mb + 1) : 0;
    const bool have = rb < nblk, have_next = more && rb < (bnext + RB - 1) / RB;   // (block-uniform)
    u64* slabs_s = slabs64 + (long long)(seq_out & 1u) * nrb * SW;
    bool fail = false;
    auto timed_out = [&](unsigned& it) {
      __builtin_amdgcn_s_sleep(1);
      if (++it > (1u << 22) || ((it & 255u) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
        if (lane == 0) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return true;
      }
      return false;
    };
    // ---- A: gradient of this minibatch
    if (have_next) load_rows(mb + 1);   // (in flight under the poll and the chain; parked behind the chain's barrier)
    if (have) {
      const float adv_mean = lds[G.ring + 0], adv_std = lds[G.ring + 1];
      {   // the tower's parameters as the chunk owners published them: all words requested together, the whole set again
          // until every one carries this step's number; then straight to their places in the images
        const u64* par = par64 + (long long)(seq_par & 1u) * o.total;
        const u64* pa = par + t0;
        const u64* pb = par + segB0;
        const u64* pc = par + (d.discrete ? 0 : o.log_std + min(tid, A - 1));
        u64 ta[NLL], tb[NB], tc;
        unsigned it = 0;
        for (;;) {
#pragma unroll
          for (int i = 0; i < NLL; ++i)
            ta[i] = __hip_atomic_load(pa + min(tid + i * NT, tlen - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
          for (int i = 0; i < NB; ++i)
            tb[i] = __hip_atomic_load(pb + min(tid + i * NT, nB - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          tc = __hip_atomic_load(pc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __builtin_amdgcn_sched_barrier(0);
          bool ok = (unsigned)(tc >> 32) == seq_par;
#pragma unroll
          for (int i = 0; i < NLL; ++i) ok = ok && (unsigned)(ta[i] >> 32) == seq_par;
#pragma unroll
          for (int i = 0; i < NB; ++i) ok = ok && (unsigned)(tb[i] >> 32) == seq_par;
          if (__all(ok)) break;
          if (timed_out(it)) { fail = true; break; }
        }
#pragma unroll
        for (int i = 0; i < NLL; ++i) {
          const float w = __uint_as_float((unsigned)ta[i]);
          const unsigned da = dA[i] & 0xffffu, db = dA[i] >> 16;
          if (da != 0xffffu) lds[da] = w;
          if (db != 0xffffu) lds[db] = w;
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
          const float w = __uint_as_float((unsigned)tb[i]);
          const unsigned da = dB[i] & 0xffffu, db = dB[i] >> 16;
          if (da != 0xffffu) lds[da] = w;
          if (db != 0xffffu) lds[db] = w;
        }
        if (tower == 0 && !d.discrete && tid < A) {   // log_std[a] and the loss phase's constants 1 / sd^2, log sd
          const float lsv = __uint_as_float((unsigned)tc);
          const float sd = expf(lsv);
          lds[G.ls + tid] = lsv;
          lds[G.ls + 16 + tid] = 1.f / (sd * sd);
          lds[G.ls + 32 + tid] = logf(sd);
        }
      }
      if (fail) s_fail = 1;
      __syncthreads();
      u64* slab = slabs_s + (long long)rb * SW;
      t64h_tower_minibatch<KT1, RB, NW>(d, G, out, tower, lds, RB * rb, b, adv_mean, adv_std, normalize_adv, clip, ent_coef, vf_coef,
                                        slab, seq_out, [&]() { if (have_next) park_rows(); },
                                        (dbg != nullptr && bid < 2) ? dbg + 16 + 16 * bid : nullptr, oz);
    } else if (have_next) {
      __syncthreads();
      park_rows();
      __syncthreads();
    }
    if (s_fail) return;   // (behind a block barrier; workgroups without rows never set it)
    EP_TS(0);
    // ---- B1: chunk `bid` of every slab, summed in slab order; workgroup 0 also collects the loss-statistic partials
    const int i0 = bid * chunk + oz, i1 = min(o.total, i0 + chunk);
    float g[NPC];
    float sqs = 0.f;
    // (the slabs' words of an element are requested TOGETHER, two elements at a time: up to 32 loads in flight and one trip
    //  through the fabric where eight-slab batches per element took four -- 3.6 of the step's 21 us; same sums, same order)
    if ((NW == 4 || RB == 32) && nblk > 16) {
      // more than sixteen slabs (32-row blocks of a 1 024-row minibatch): ALL words of one element requested together -- one
      // trip through the fabric where two batches of sixteen took two (same sums in the same slab order)
#pragma unroll
      for (int j = 0; j < NPC; ++j) {
        float acc = 0.f;
        if (i0 + (tid & ~63) + j * NT < i1) {   // (wave-uniform)
          const u64* col = slabs_s + min(i0 + tid + j * NT, o.total - 1);
          const unsigned sw = (unsigned)SW;
          u64 t[32];
          unsigned it = 0;
          for (;;) {
#pragma unroll
            for (int u = 0; u < 32; ++u)
              t[u] = __hip_atomic_load(col + (unsigned)min(u, nblk - 1) * sw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            bool ok = true;
#pragma unroll
            for (int u = 0; u < 32; ++u) ok = ok && (unsigned)(t[u] >> 32) == seq_out;
            if (__all(ok)) break;
            if (timed_out(it)) { fail = true; break; }
          }
#pragma unroll
          for (int u = 0; u < 32; ++u)
            if (u < nblk) acc += __uint_as_float((unsigned)t[u]);
          if (i0 + tid + j * NT < i1) sqs += acc * acc;
        }
        g[j] = acc;
      }
    } else
#pragma unroll
    for (int j0 = 0; j0 < NPC; j0 += 2) {
      float acc[2] = {0.f, 0.f};
      if (i0 + (tid & ~63) + j0 * NT < i1) {   // (wave-uniform: some lane of the wave has an element of this pair)
        const bool two = i0 + (tid & ~63) + (j0 + 1) * NT < i1;   // (wave-uniform)
        const u64* col0 = slabs_s + min(i0 + tid + j0 * NT, o.total - 1);
        const u64* col1 = slabs_s + min(i0 + tid + (j0 + 1) * NT, o.total - 1);
        for (int sb = 0; sb < nblk && !fail; sb += 16) {
          u64 t0[16], t1[16];
          unsigned it = 0;
          const bool full = nblk - sb >= 16;   // (block-uniform: sixteen real slabs -- no clamps, no conditional sums)
          const unsigned sw = (unsigned)SW;
          const u64* c0 = col0 + (long long)sb * SW;
          const u64* c1 = col1 + (long long)sb * SW;
          for (;;) {
#pragma unroll
            for (int u = 0; u < 16; ++u)
              t0[u] = __hip_atomic_load(c0 + (full ? (unsigned)u : (unsigned)min(u, nblk - 1 - sb)) * sw, __ATOMIC_RELAXED,
                                        __HIP_MEMORY_SCOPE_AGENT);
            if (two) {
#pragma unroll
              for (int u = 0; u < 16; ++u)
                t1[u] = __hip_atomic_load(c1 + (full ? (unsigned)u : (unsigned)min(u, nblk - 1 - sb)) * sw, __ATOMIC_RELAXED,
                                          __HIP_MEMORY_SCOPE_AGENT);
            }
            bool ok = true;
#pragma unroll
            for (int u = 0; u < 16; ++u) ok = ok && (unsigned)(t0[u] >> 32) == seq_out && (!two || (unsigned)(t1[u] >> 32) == seq_out);
            if (__all(ok)) break;
            if (timed_out(it)) { fail = true; break; }
          }
          if (full) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
              acc[0] += __uint_as_float((unsigned)t0[u]);
              acc[1] += two ? __uint_as_float((unsigned)t1[u]) : 0.f;
            }
          } else {
#pragma unroll
            for (int u = 0; u < 16; ++u)
              if (sb + u < nblk) {
                acc[0] += __uint_as_float((unsigned)t0[u]);
                if (two) acc[1] += __uint_as_float((unsigned)t1[u]);
              }
          }
        }
        if (i0 + tid + j0 * NT < i1) sqs += acc[0] * acc[0];
        if (two && i0 + tid + (j0 + 1) * NT < i1) sqs += acc[1] * acc[1];
      }
      g[j0] = acc[0];
      g[j0 + 1] = acc[1];
    }
    if (bid == 0 && stats != nullptr && (tid & ~63) < nblk * 8) {   // (wave-uniform) slab q's statistics slot: word P + slot
      const int q = min(tid >> 3, nblk - 1), slot = tid & 7;
      const bool mine = tid < nblk * 8 && slot < 5;
      const u64* wp = slabs_s + (long long)q * SW + o.total + (mine ? slot : 0);
      u64 t;
      unsigned it = 0;
      for (;;) {
        t = __hip_atomic_load(wp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__all(!mine || (unsigned)(t >> 32) == seq_out)) break;
        if (timed_out(it)) { fail = true; break; }
      }
      if (tid < nblk * 8) s_stat[tid] = __uint_as_float((unsigned)t);
    }
    EP_TS(1);
    {
      const float part = block_sum<NT>(sqs, lds + G.scratch);
      if (tid == 0) ll_store_agent(sq64 + (seq_out & 1u) * 64 + bid, part, seq_out);
    }
    EP_TS(2);
    // ---- while the partial sums of squares travel: the next minibatch's rows are normalised into the x tile
    if (have_next) stage_rows(bnext);
    // ---- B2: the G partial sums of squares -> norm, clip coefficient; Adam on the own chunk; loss statistics
    if (tid < 64) {
      const u64* wp = sq64 + (seq_out & 1u) * 64 + min(tid, nwg - 1);
      u64 t;
      unsigned it = 0;
      for (;;) {
        t = __hip_atomic_load(wp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__all((unsigned)(t >> 32) == seq_out)) break;
        if (timed_out(it)) { fail = true; break; }
      }
      s_part[tid] = tid < nwg ? __uint_as_float((unsigned)t) : 0.f;
    }
    if (__any(fail) && lane == 0) s_fail = 1;
    __syncthreads();
    if (s_fail) return;
    EP_TS(3);
    float total_sq = 0.f;
    for (int q = 0; q < nwg; ++q) total_sq += s_part[q];
    const float total_norm = sqrtf(total_sq);
    const float coef = fminf(max_norm / (total_norm + 1e-6f), 1.0f);   // torch.nn.utils.clip_grad_norm_
    {
      const float step_size = st.step_size[k], bc2_sqrt = st.bc2_sqrt[k];
      u64* dst = par64 + (long long)(seq_out & 1u) * o.total;
#pragma unroll
      for (int j = 0; j < NPC; ++j) {
        const int i = i0 + tid + j * NT;
        if (i < i1) {
          const float gi = g[j] * coef;
          const float mi = m_[j] + (gi - m_[j]) * (1.f - beta1);
          const float vi = v_[j] * beta2 + (1.f - beta2) * gi * gi;
          const float denom = sqrtf(vi) / bc2_sqrt + eps;
          const float pn = p_[j] - step_size * (mi / denom);
          ll_store_agent(dst + i, pn, seq_out);
          p_[j] = pn;
          m_[j] = mi;
          v_[j] = vi;
        }
      }
    }
    if (bid == 0 && stats != nullptr) {
      __shared__ float s_st[5];
      if (tid >= 64 && tid < 69) {   // one lane of the second wave per statistic, partials in row-block order
        const int kk = tid - 64;
        float sv = 0.f;
        for (int q = 0; q < nblk; ++q) sv += s_stat[q * 8 + kk];
        sv *= 1.f / (float)b;
        stats[(long long)mb * 8 + kk] = sv;
        s_st[kk] = sv;
      }
      __syncthreads();
      if (tid == 0) {
        stats[(long long)mb * 8 + 5] = s_st[0] + ent_coef * s_st[2] + vf_coef * s_st[1];  // loss
        stats[(long long)mb * 8 + 6] = total_norm;
        stats[(long long)mb * 8 + 7] = coef;
      }
    }
    EP_TS(4);
  }
  // the launch's last parameters and moments back to memory (torch layout + the transposed shadow copy)
  tid = threadIdx.x;
#pragma unroll
  for (int j = 0; j < NPC; ++j) {
    const int i = bid * chunk + tid + j * NT;
    if (i < min(o.total, (bid + 1) * chunk)) {
      m[i] = m_[j];
      v[i] = v_[j];
      P[i] = p_[j];
      int dst = i;
      auto tr = [&](int b0, int rows_, int cols) {
        if (i >= b0 && i < b0 + rows_ * cols) {
          const int r = (i - b0) / cols, cc = (i - b0) % cols;
          dst = b0 + cc * rows_ + r;
        }
      };
      tr(o.pW1, H, D); tr(o.pW2, H, H); tr(o.vW1, H, D); tr(o.vW2, H, H);
      Pt[dst] = p_[j];
    }
  }
#undef EP_TS
}

// TIMING = false (production): the phase-clock accumulators (24 VGPRs of `tacc` alone) and every stamp are
// compiled out -- the measurement build is a separate instantiation picked only while ia_ppo_debug_timing is on.
template <int NPT, bool TIMING, int KS1, bool LOCAL, bool SHARD = false, bool SMALL = false>
__global__ __launch_bounds__(512) void ppo_update_persistent_kernel(
    ia_policy_desc d, float* __restrict__ P, float* __restrict__ Pt, float* __restrict__ m, float* __restrict__ v,
    float* __restrict__ nm, float* __restrict__ nv, int32_t* __restrict__ ncount, int update_norm,
    const float* __restrict__ obs, const float* __restrict__ actions, const float* __restrict__ old_logp,
    const float* __restrict__ adv, const float* __restrict__ ret, const int64_t* __restrict__ perm, int T, int n_envs,
    int normalize_adv, float clip, float ent_coef, float vf_coef, float max_norm, float beta1, float beta2, float eps,
    float* __restrict__ ws, int nblk, int n_slices, float* __restrict__ stats, UpdSched sch, int xcd_pack,
    long long* __restrict__ tstamp /* debug: [0..3] += 100 MHz ticks in {stat wait, minibatch, barrier, update} */,
    ShardArgs sh) {
  constexpr int H = 32;
  using L = CLds;
  // The tiles are read with ds_read_b128: their base must be 16-byte aligned. The dynamic part of the LDS starts right
  // behind the static part -- 8 or 28 bytes here, depending on the instantiation -- and every wide read was split (2.4x
  // the gradient-tile phase) until the base was rounded up (the host allocates 16 bytes more).
  extern __shared__ float lds_raw[];
  float* lds = lds_raw + (((16u - (__builtin_amdgcn_groupstaticsize() & 15u)) & 15u) >> 2);
  __shared__ int s_ok, s_pub, s_fail, s_pubw[4];
  if (!TIMING) tstamp = nullptr;
  long long tacc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  long long tprev = 0;
#define UPD_TS(k) do { if (tstamp && tid == 0) { const long long tn = wall_clock64(); tacc[k] += tn - tprev; tprev = tn; } } while (0)
  // xcd_pack: the launch has 8x the blocks and only every 8th works, so that all working blocks sit
  // on one XCD (hardware deals consecutive block ids round-robin over the 8 XCDs) and share its L2.
  if (xcd_pack && (blockIdx.x & 7)) return;
  const int vb = xcd_pack ? blockIdx.x >> 3 : blockIdx.x;
  const int tid = threadIdx.x;
  const int D = d.obs_dim;
  const PolOff o = pol_offsets(D, d.act_dim, H, d.discrete);
  const UpdWs w = upd_ws(ws, nblk, o.total);
  unsigned* arrivals = w.ctrl + 64;   // UPD_ARR counters, 16 words apart
  unsigned* published = w.ctrl + 1;
  unsigned* sliced = w.ctrl + 16;     // [n_slices] steps whose partial moments slicer j has written
  unsigned* err = w.ctrl + 8;
  // sequence numbers of the (value, sequence) words: they only ever grow over the workspace's life (word 9 of it, advanced
  // by workgroup 0 when a launch ends) -- a word left by an earlier launch can never pass for this one's
  const unsigned lseq_base = __hip_atomic_load(w.ctrl + 9, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (threadIdx.x == 0) s_fail = 0;
  // Packed launches: do ALL gradient workgroups sit on one XCD? Each publishes the id of its XCC (an ordinary word of the
  // exchange, in the unused last tail word of its slab, under a sequence number no step uses) and reads everybody's; if
  // they agree -- the same answer in every workgroup --, the step's words are stored at workgroup scope (`ll_store_xcd`):
  // they stay in the XCD's L2, the coherence point of its compute units, and the polls (agent scope: past L1) find them
  // there. One exchange per launch; any other placement keeps the agent-scope stores.
  bool ll_xcd = false;
  if constexpr (!LOCAL) {
    if (xcd_pack && vb < nblk && nblk <= 64) {
      __shared__ int s_xcd;
      unsigned xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      xcc &= 0xfu;
      const unsigned aseq = ~lseq_base;
      unsigned long long* aw = w.slabs64 + w.P4 + 7;
      const long long P8a = w.P4 + 8;
      if (tid == 0) ll_store_agent(aw + (long long)vb * P8a, __uint_as_float(xcc), aseq);
      if (tid < 64) {
        const unsigned long long* wp = aw + (long long)min(tid, nblk - 1) * P8a;
        unsigned long long t;
        bool bad = false;
        const long long t0a = wall_clock64();
        for (unsigned it = 0;;) {
          t = __hip_atomic_load(wp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (__all((unsigned)(t >> 32) == aseq)) break;
          __builtin_amdgcn_s_sleep(1);
          if ((++it & 255u) == 0 &&
              (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) || wall_clock64() - t0a > 200000000ll)) {
            bad = true;
            break;
          }
        }
        if (tid == 0) {
          s_xcd = 0;
          if (bad) s_fail = 1;
        }
        if (!bad && __all((unsigned)t == xcc) && tid == 0) s_xcd = 1;
      }
      __syncthreads();
      if (s_fail) {
        if (tid == 0) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
      }
      ll_xcd = s_xcd != 0;
      if (tstamp && tid == 0 && vb < 8) tstamp[42 + vb] = (long long)xcc * 2 + (ll_xcd ? 1 : 0);   // (measurement build)
    }
  }

  // schedule scalars in registers; the per-step tables are read straight from the kernel arguments
  const int sch_first = sch.first, sch_nmb = sch.n_mb, sch_bs = sch.batch_size, n_steps = sch.n_steps;
  const long long sch_total = sch.total;
  // row-sharded data parallelism: this rank's rows of a (global) minibatch are [row_lo, row_lim(r))
  const int row_lo = SHARD ? sh.rank * sh.rows_per_rank : 0;
  auto row_lim = [&](const MbRows& r) { return SHARD ? min(r.batch, row_lo + sh.rows_per_rank) : r.batch; };
  // (epoch, minibatch) -> rows; the gradient workgroups carry the pair from step to step instead of dividing every time
  auto rows_at = [=](int e, int mb) {
    const long long start = (long long)mb * sch_bs;
    const long long left = sch_total - start;
    MbRows r{obs, actions, old_logp, adv, ret, perm + (long long)e * sch_total + start,
             (int)(left < sch_bs ? left : sch_bs), T, n_envs};
    return r;
  };
  auto rows_of = [=](int s) {
    const int gs = sch_first + s;
    const int e = gs / sch_nmb, mb = gs - e * sch_nmb;
    const long long start = (long long)mb * sch_bs;
    const long long left = sch_total - start;
    MbRows r{obs, actions, old_logp, adv, ret, perm + (long long)e * sch_total + start,
             (int)(left < sch_bs ? left : sch_bs), T, n_envs};
    return r;
  };

  // Loss statistics of step q: sum of the per-block partials in block order, written by one wave.
  auto write_loss_stats = [&](int q, float total_norm, float coef) {
    if (tid < 64 && stats) {
      int wz;   // opaque zero: lane index and row addresses of this rarely taken branch are formed here, not kept alive
      asm volatile("s_mov_b32 %0, 0" : "=s"(wz));
      const int ln = (tid + wz) & 63;
      const float* sb = w.statpart + (q % UPD_SD) * nblk * 8;
      float st = 0.f;
      if (ln < 5) {
        int b = 0;
        for (; b + 8 <= nblk; b += 8) {
          float t[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) t[u] = sb[(b + u) * 8 + ln];
#pragma unroll
          for (int u = 0; u < 8; ++u) st += t[u];
        }
        for (; b < nblk; ++b) st += sb[b * 8 + ln];
        st *= 1.f / (float)rows_of(q).batch;
      }
      const float st0 = __shfl(st, 0, 64), st1 = __shfl(st, 1, 64), st2 = __shfl(st, 2, 64);
      float* so = stats + (long long)(sch_first + q) * 8;
      if (ln < 5) so[ln] = st;
      if (ln == 5) so[5] = st0 + ent_coef * st2 + vf_coef * st1;
      if (ln == 6) so[6] = total_norm;
      if (ln == 7) so[7] = coef;
    }
  };

  if (vb == nblk) {  // ---------------- statistics block
    // It also writes the loss statistics of finished steps (all but the last one of the launch):
    // step q's partials and its (norm, coef) pair are published by barrier q+1's release fences.
    int q = 0;
    auto drain = [&](int upto /* exclusive */) {
      if constexpr (SHARD || !LOCAL) return;   // (gradient workgroup 0 writes them itself: it holds the sums)
      for (; q < upto; ++q)
        write_loss_stats(q, *reinterpret_cast<const volatile float*>(w.normcoef + (q % UPD_SD) * 2),
                         *reinterpret_cast<const volatile float*>(w.normcoef + (q % UPD_SD) * 2 + 1));
    };
    long long sb_acc[3] = {0, 0, 0}, sb_prev = tstamp ? wall_clock64() : 0;
#define SB_TS(k) do { if (tstamp && tid == 0) { const long long tn = wall_clock64(); sb_acc[k] += tn - sb_prev; sb_prev = tn; } } while (0)
    for (int s = 0; s < n_steps; ++s) {
      if (s >= UPD_RING) {  // the slot is free once every gradient block has arrived at barrier s - RING
        if (tid < 64) {
          const bool ok = spin_arrivals(arrivals, (unsigned)(s - UPD_RING + 1), 1, err);   // (workgroup 0's steps)
          if (tid == 0) {
            s_ok = ok;
            __threadfence();
          }
        }
        __syncthreads();
        if (!s_ok) return;
        SB_TS(0);
        drain(s - UPD_RING);  // barriers 0 .. s-RING complete => steps 0 .. s-RING-1 fully published
        SB_TS(1);
      }
      const MbRows r = rows_of(s);
      float* slot = w.ring + (s % UPD_RING) * UPD_RS;
      if (n_slices == 0) {
        prepare_stats_t<512>(d, r.obs, r.adv, r.idx, r.batch, T, n_envs, update_norm, nm, nv, ncount, slot + 2 * MAXD,
                             lds, sch_total < (1ll << 31) ? reinterpret_cast<int*>(lds + PREP_LDS_FLOATS) : nullptr);
      } else {
        // large minibatch: the slicer blocks left per-slice moments; merge them in slice order (Chan),
        // then the same running update / advantage statistics as the one-block form
        const int ns = (r.batch + UPD_SLICE - 1) / UPD_SLICE;
        if (tid < 64) {   // lane j polls slicer j (all of them at once: one round trip, not one per slice)
          unsigned it = 0;
          bool ok = true;
          for (;;) {
            const bool here = tid >= ns || __hip_atomic_load(sliced + min(tid, ns - 1), __ATOMIC_RELAXED,
                                                              __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)(s + 1);
            if (__all(here)) break;
            __builtin_amdgcn_s_sleep(2);
            if (++it > (1u << 24) || ((it & 255u) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
              ok = false;
              break;
            }
          }
          if (tid == 0) {
            if (!ok) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_ok = ok;
            __threadfence();
          }
        }
        __syncthreads();
        if (!s_ok) return;
        const float* pr = w.pring + (long long)(s % UPD_RING) * UPD_SLICES_MAX * UPD_PRS;
        const bool norm_on = d.has_norm && update_norm;
        if (tid <= D) {
          const int c = tid;  // columns 0..D-1: features; D: advantages
          float n_acc = 0.f, m_acc = 0.f, M2 = 0.f;
          for (int j0 = 0; j0 < ns; j0 += 8) {   // eight slices' moments requested together, merged in slice order
            float nb[8], mb[8], qb[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              const float* pj = pr + min(j0 + u, ns - 1) * UPD_PRS;
              nb[u] = pj[2 * MAXD + 2];
              mb[u] = c < D ? pj[c] : pj[2 * MAXD + 0];
              qb[u] = c < D ? pj[MAXD + c] : pj[2 * MAXD + 1];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 8; ++u)
              if (j0 + u < ns) {
                const float tot = n_acc + nb[u], dlt = mb[u] - m_acc;
                M2 = M2 + qb[u] + dlt * dlt * n_acc * nb[u] / tot;
                m_acc = m_acc + dlt * nb[u] / tot;
                n_acc = tot;
              }
          }
          if (c == D) {
            slot[2 * MAXD + 0] = m_acc;
            slot[2 * MAXD + 1] = r.batch > 1 ? sqrtf(M2 / (float)(r.batch - 1)) : 0.f;
          } else if (norm_on) {
            const int cnt = *ncount;
            const float bmean = m_acc, bvar = M2 / (float)r.batch;
            const float fcount = (float)cnt, fn = (float)r.batch, tot = (float)((long long)cnt + r.batch);
            const float delta = bmean - nm[c];
            nm[c] = nm[c] + delta * fn / tot;
            float rvv = nv[c] * fcount;
            rvv = rvv + bvar * fn;
            rvv = rvv + delta * delta * fcount * fn / tot;
            nv[c] = rvv / tot;
          }
        }
        __syncthreads();
        if (tid == 0 && norm_on) *ncount = rn_count_add(*ncount, r.batch);
      }
      __syncthreads();
      if (d.has_norm && tid < D) {
        // mean and 1/sqrt(var + eps): the square root and the division are done ONCE per column here (this block runs
        // ahead of the chain) instead of once per element in every gradient block (8 IEEE sqrt + 8 IEEE divisions per
        // lane on the chain's critical path); the product is within 1 ulp of the quotient `networks.py:91` forms
        slot[tid] = nm[tid];
        slot[MAXD + tid] = 1.f / sqrtf(nv[tid] + d.norm_eps);
      }
      __syncthreads();
      if (tid == 0) {
        __threadfence();
        __hip_atomic_store(published, (unsigned)(s + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      SB_TS(2);
    }
    if (tstamp && tid == 0) { tstamp[12] += sb_acc[0]; tstamp[13] += sb_acc[1]; tstamp[14] += sb_acc[2]; }
#undef SB_TS
    if (tid < 64) {
      const bool ok = spin_arrivals(arrivals, (unsigned)n_steps, 1, err);
      if (tid == 0) {
        s_ok = ok;
        __threadfence();
      }
    }
    __syncthreads();
    if (!s_ok) return;
    drain(n_steps - 1);  // the last step's statistics are written by gradient block 0 itself
    return;
  }
  if (vb > nblk) {  // ---------------- statistics slicers (large minibatches only)
    const int j = vb - nblk - 1;
    if (j >= n_slices) return;
    for (int s = 0; s < n_steps; ++s) {
      const MbRows r = rows_of(s);
      const int r0 = j * UPD_SLICE;
      const int nr = min(UPD_SLICE, r.batch - r0);
      if (s >= UPD_RING) {  // the partial slot is free once the merger has published step s - RING
        if (tid == 0) s_ok = spin_until(published, (unsigned)(s - UPD_RING + 1), err);
        __syncthreads();
        if (!s_ok) return;
      }
      if (nr > 0) {
        float* pj = w.pring + ((long long)(s % UPD_RING) * UPD_SLICES_MAX + j) * UPD_PRS;
        prepare_stats_t<512>(d, r.obs, r.adv, r.idx + r0, nr, T, n_envs, update_norm, nullptr, nullptr, nullptr, nullptr,
                             lds, reinterpret_cast<int*>(lds + PREP_LDS_FLOATS), pj);
      }
      __syncthreads();
      if (tid == 0) {
        __threadfence();
        __hip_atomic_store(sliced + j, (unsigned)(s + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    return;
  }

  // ---------------- gradient blocks
  constexpr int PROWS = SMALL ? 16 : ROWS;   // rows a minibatch of this workgroup can have
  float* sP = lds + L::total;
  float* sPt = sP + w.P4;                     // transposed towers, padded rows (UPD_TSTR; see there)
  float* stg = sPt + upd_pt4(w.P4);           // UpdStage: the NEXT minibatch's rows of this block
  unsigned short* dstT = reinterpret_cast<unsigned short*>(stg + UpdStage::total(d.discrete ? 1 : d.act_dim));  // [P4] index of parameter i in the transposed copy
  // (the 64 floats at lds + L::scratch are the block reductions' scratch)
  const int lane = tid & 63;
  float rm[NPT], rv[NPT];
#pragma unroll
  for (int k = 0; k < NPT; ++k) {
    const int i = tid + k * 512;
    rm[k] = rv[k] = 0.f;
    if (i < o.total) {
      const float pi_ = P[i];
      sP[i] = pi_;
      rm[k] = m[i];
      rv[k] = v[i];
      int dst = 0xffff;   // (not a tower matrix: no transposed copy inside this kernel)
      auto tr = [&](int base, int rows, int cols) {
        if (i >= base && i < base + rows * cols) {
          const int rr = (i - base) / cols, cc = (i - base) % cols;
          dst = upd_tbase(base) + cc * UPD_TSTR + rr;
        }
      };
      tr(o.pW1, H, D); tr(o.pW2, H, H); tr(o.vW1, H, D); tr(o.vW2, H, H);
      dstT[i] = (unsigned short)dst;
      if (dst != 0xffff) sPt[dst] = pi_;   // (= Pt's value of that element: the global copy is rewritten when the launch ends)
    }
  }
  // Row prefetch: the gathers of step s+1 (two dependent global loads per element) are issued
  // before the grid barrier of step s and land in registers behind the barrier wait and the
  // update; they are parked in the LDS staging area just before step s+1 starts.
  // (i) one step ahead of (ii): wave 7 resolves permutation entry -> rollout-tile row offset for the
  // block's 64 rows of minibatch s (a dependent global load plus a division) and leaves them in LDS;
  // any later block barrier publishes them. (ii) every thread then issues its row gathers at once.
  auto prefetch_resolve = [&](const MbRows& r) {
    if (tid >= 448) {
      int tz;   // opaque zero: the reciprocal of T behind `f / T` is re-derived per step instead of being spilled
      asm volatile("s_mov_b32 %0, 0" : "=s"(tz));
      const unsigned Tq = (unsigned)(T + tz);
      const int i0 = row_lo + vb * ROWS;
      int src = 0;
      if (i0 + lane < row_lim(r)) {
        const long long flat = r.idx[i0 + lane];
        if (sch_total < (1ll << 31)) {  // 32-bit divide (the 64-bit one is a long software sequence)
          const unsigned f = (unsigned)flat, env = f / Tq, t = f - env * Tq;
          src = (int)(t * (unsigned)n_envs + env);
        } else {
          src = (int)rollout_offset(flat, T, n_envs);
        }
      }
      reinterpret_cast<int*>(stg)[UpdStage::nxt + lane] = src;
    }
  };
  // (ii) global -> LDS directly (no staging registers to keep alive across the update): each wave
  // instruction drops its 64 dwords at a wave-uniform LDS base + 4*lane, which is exactly the
  // linear staging order e = it*512 + wave*64 + lane. Addresses are clamped to valid rows/columns;
  // slots outside the observation width or the batch are masked when the stage is read.
  typedef __attribute__((address_space(3))) void* lds_void_p;
  typedef __attribute__((address_space(1))) const void* glb_void_p;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // Every row offset is read from LDS BEFORE the first LDS-direct load is issued: the compiler orders an LDS read
  // that follows a `global_load_lds` behind `s_waitcnt vmcnt(0)` (the DMA might alias it), which made the nine
  // gathers of a wave nine serial round trips to memory (3.6 us per step on the barrier path). `zero` is an opaque
  // 0 refreshed every step, so the element -> (row, column) arithmetic is redone here (a dozen VALU operations)
  // instead of being hoisted out of the step loop into 27 spilled registers.
  // Round 5: the observation rows travel as 16-BYTE pieces (`global_load_lds` width 16: LDS destination = wave-uniform base +
  // 16 x lane). Slot e = row * Q + piece, Q = ceil(D / 4): the rows land 4 Q floats apart in slot order -- exactly the order
  // `chain_stage_rows` walks them (one ds_read_b128 per slot) --, ONE pass of five waves at D = 17 instead of three passes of
  // all eight, a third of the address arithmetic. A row's last piece runs up to 12 bytes into the next row of the rollout
  // tile (masked when the slot is read; the tile's last time slice T is never a minibatch row, so the tensor is not left).
  // The per-row scalars are loaded by wave 7 (the x pieces occupy the low waves).
  auto prefetch_issue = [&](const MbRows& r, int zero) {
    const int* nxt = reinterpret_cast<const int*>(stg) + UpdStage::nxt;
    constexpr int SCW = 7;   // wave that loads the per-row scalars
    int src0 = 0;
    if (wave == SCW) src0 = nxt[lane];
    const int Q = (D + 3) >> 2;
    constexpr int NQT = (PROWS * (MAXD / 4) + 511) / 512;
    int qsrc[NQT], qoff[NQT];
    const unsigned rcpQ = 0xffffffffu / (unsigned)Q + 1u;
#pragma unroll
    for (int it = 0; it < NQT; ++it) {
      const int e0 = it * 512 + wave * 64 + zero;  // wave-uniform
      qsrc[it] = qoff[it] = 0;
      if (e0 < PROWS * Q) {
        const int e = min(e0 + lane, PROWS * Q - 1);
        const int rr = Q == 1 ? e : (int)__umulhi((unsigned)e, rcpQ);
        qoff[it] = 4 * (e - rr * Q);
        qsrc[it] = nxt[rr];
      }
    }
    const int aw_ = d.discrete ? 1 : d.act_dim;
    constexpr int NAT = (PROWS * MAXA + 511) / 512;
    int asrc[NAT], acol[NAT];
    const unsigned rcpA = 0xffffffffu / (unsigned)aw_ + 1u;   // (aw_ = 1: 0, not used)
#pragma unroll
    for (int it = 0; it < NAT; ++it) {
      const int e0 = it * 512 + wave * 64 + zero;  // wave-uniform
      acol[it] = asrc[it] = 0;
      if (e0 < PROWS * aw_) {
        const int e = min(e0 + lane, PROWS * aw_ - 1);
        const int rr = aw_ == 1 ? e : (int)__umulhi((unsigned)e, rcpA);
        acol[it] = e - rr * aw_;
        asrc[it] = nxt[rr];
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int it = 0; it < NQT; ++it) {
      const int e0 = it * 512 + wave * 64;  // wave-uniform
      if (e0 < PROWS * Q)
        __builtin_amdgcn_global_load_lds((glb_void_p)(r.obs + (long long)qsrc[it] * D + qoff[it]),
                                         (lds_void_p)(stg + UpdStage::x + 4 * e0), 16, 0, 0);
    }
#pragma unroll
    for (int it = 0; it < NAT; ++it) {
      const int e0 = it * 512 + wave * 64;  // wave-uniform
      if (e0 < PROWS * aw_)
        __builtin_amdgcn_global_load_lds((glb_void_p)(r.actions + (long long)asrc[it] * aw_ + acol[it]),
                                         (lds_void_p)(stg + UpdStage::act + e0), 4, 0, 0);
    }
    if (wave == SCW) {
      __builtin_amdgcn_global_load_lds((glb_void_p)(r.old_logp + src0), (lds_void_p)(stg + UpdStage::oldlp), 4, 0, 0);
      __builtin_amdgcn_global_load_lds((glb_void_p)(r.adv + src0), (lds_void_p)(stg + UpdStage::adv), 4, 0, 0);
      __builtin_amdgcn_global_load_lds((glb_void_p)(r.ret + src0), (lds_void_p)(stg + UpdStage::ret), 4, 0, 0);
    }
  };
  auto prefetch_park = [&]() {  // row offsets of the staged minibatch (its action loads use them)
    if (wave == 0) stg[UpdStage::src + lane] = stg[UpdStage::nxt + lane];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the LDS-direct loads have landed
  };
  for (int e = tid; e < (SMALL ? L::total : MAXD * L::RS); e += 512) lds[L::x + e] = 0.f;   // (the chain only rewrites the columns
                                                                                            // it uses; SMALL: rows 16.. of EVERY tile)
  int cur_e = sch_first / sch_nmb, cur_mb = sch_first - cur_e * sch_nmb;   // (epoch, minibatch) of step s
  if (n_steps > 0) {
    prefetch_resolve(rows_at(cur_e, cur_mb));
    __syncthreads();
    prefetch_issue(rows_at(cur_e, cur_mb), 0);
    prefetch_park();
  }
  __syncthreads();

  // The statistics-ring slot of step s+1 is normally published long before step s ends: it is then
  // copied into LDS behind barrier s (whose acquire fence covers it) and step s+1 starts without a
  // wait, a fence or a dependent global load. `have_ring`: the LDS copy holds this step's slot.
  bool have_ring = false;
  // one gradient workgroup: its gradient never leaves the CU (LDS image inside the consumed part of the staging area)
  // (LOCAL is chosen by the host: nblk == 1 and the parameter vector fits the area; a kernel of its own, because both
  // chain forms in one kernel cost 14 spilled registers in the 9-parameters-per-thread instantiation)
  constexpr bool local = LOCAL;
  if (tstamp && tid == 0) tprev = wall_clock64();
  for (int s = 0; s < n_steps; ++s) {
    const MbRows r = rows_at(cur_e, cur_mb);
    const bool wrap = cur_mb + 1 == sch_nmb;
    const int nxt_e = wrap ? cur_e + 1 : cur_e, nxt_mb = wrap ? 0 : cur_mb + 1;   // step s + 1
    cur_e = nxt_e;
    cur_mb = nxt_mb;
    const float* slot = w.ring + (s % UPD_RING) * UPD_RS;
    float adv_mean, adv_std;
    if (!have_ring) {
      if (tid == 0) {
        s_ok = spin_until(published, (unsigned)(s + 1), err);
        __threadfence();
      }
      __syncthreads();
      if (!s_ok) return;
      // (rare: the statistics block had not published this slot when the previous barrier was passed) copy it into
      // the LDS slot too, so that the chain below ALWAYS reads its statistics through LDS addresses -- a pointer that
      // may be global or LDS compiles to flat loads, 16 of them per lane in the staging phase
      {
        int sz;   // opaque zero: this rarely taken copy must not leave a hoisted (and spilled) address behind
        asm volatile("s_mov_b32 %0, 0" : "=s"(sz));
        if (tid < 2 * MAXD + 8) stg[UpdStage::ring + tid] = *reinterpret_cast<const volatile float*>(slot + tid + sz);
      }
      __syncthreads();
      if constexpr (!LOCAL) {   // (several workgroups stage ahead, see below: here only step 0 or late statistics)
        int zz;
        asm volatile("s_mov_b32 %0, 0" : "=s"(zz));
        const int wv_ = __builtin_amdgcn_readfirstlane((tid + zz) >> 6);
        chain_stage_rows<false>(d, stg + UpdStage::ring, stg + UpdStage::ring + MAXD, row_lo + vb * ROWS, row_lim(r), lds, stg,
                                wv_ >> 2, wv_ & 3, lane);
      }
    }
    slot = stg + UpdStage::ring;
    // (several workgroups: how far the statistics workgroup is, read HERE -- before the statistics slot of step s + 1 is
    //  requested below, so a slot it calls published was published when its load was issued)
    // (one gradient workgroup: read here too. It used to be read behind the minibatch by thread 0, followed by a full fence
    //  and a barrier of its own: a round trip through the fabric plus the wait for the wave's row gathers, ~0.8 us per step
    //  with every wave waiting. The value of the step's start says "published" almost always -- the statistics workgroup
    //  runs a ring ahead --, and the acquire half of the fence is all the slot's loads need.)
    const int pub_early = (int)__hip_atomic_load(published, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    adv_mean = slot[2 * MAXD];
    adv_std = slot[2 * MAXD + 1];
    UPD_TS(0);
    // Adam's scalars of this step: requested now (a global load), consumed after the grid barrier
    const float step_size = w.tab[s], bc2_sqrt = w.tab[UPD_MAX_STEPS + s];
    if (s + 1 < n_steps) prefetch_resolve(rows_at(nxt_e, nxt_mb));  // published by the block barriers inside the minibatch
    const int P8 = w.P4 + 8;
    const unsigned lseq = lseq_base + (unsigned)s + 1u;
    // row-sharded exchange: the receive areas are double-buffered by the parity of the GLOBAL sequence number (the exchange
    // context's counter, which only grows), not of this launch's step index: a launch with an odd number of steps would
    // otherwise end and the next one begin on the same buffer, and a fast rank could overwrite a record a slow peer has not
    // read yet (the ranks are only ordered by the words themselves)
    const int xpar = SHARD ? (int)((sh.seq_base + (unsigned)s) & 1u) : 0;
    unsigned long long* slabs_s = w.slabs64 + (long long)(s & 1) * nblk * P8;
    float* stat_base = w.statpart + (s % UPD_SD) * nblk * 8;
    int oz;
    asm volatile("s_mov_b32 %0, 0" : "=s"(oz));
    mfma32_minibatch_chain<KS1, LOCAL, SMALL, !LOCAL>(d, slot, slot + MAXD, adv_mean, adv_std, r, row_lo + vb * ROWS, row_lim(r),
                                       normalize_adv, clip, ent_coef, vf_coef,
                                       reinterpret_cast<float*>(slabs_s + (long long)vb * P8), stat_base + vb * 8, lds, sP,
                                       stg, oz, tstamp ? tstamp + 16 : nullptr, lseq, ll_xcd);
    // (the minibatch ends with a block barrier: the LDS tiles are free; several workgroups: the slab words are on their
    //  way, nobody waits for them here)
    UPD_TS(1);
    UPD_TS(7);
    UPD_TS(8);
    float g[NPT];
    unsigned long long tail_w = 0ull;                                  // (several workgroups: the sum vector's tail, see hop 2)
    const bool want_tail = !LOCAL && vb == 0 && tid < 64 && stats;
    if (local) {
      // the gradient image out of the staging area before the next minibatch's rows are loaded over it
      int gz;
      asm volatile("s_mov_b32 %0, 0" : "=s"(gz));
#pragma unroll
      for (int k = 0; k < NPT; ++k) g[k] = stg[UpdStage::x + gz + min(tid + k * 512, o.total - 1)];
      if constexpr (SHARD) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the loss-statistic partials go into the record
      if (tid == 0) {
        s_ok = 1;
        s_pub = pub_early;
        // acquire, by hand: the LDS store above has waited for the counter's load; the CU's L1 is invalidated before any
        // wave requests the slot below. (The fence builtin also waits for this wave's write-through stores of the
        // minibatch -- a trip to memory and back -- with every other wave at the barrier.)
        asm volatile("buffer_inv sc1" ::: "memory");
      }
      __syncthreads();
    }
    if (s + 1 < n_steps) {
      int pz;
      asm volatile("s_mov_b32 %0, 0" : "=s"(pz));
      prefetch_issue(rows_at(nxt_e, nxt_mb), pz);
    }
    if constexpr (!LOCAL) {
      if (s + 1 < n_steps && wave < 3) {
        if (pub_early >= s + 2) {
          const float* nslot = w.ring + ((s + 1) % UPD_RING) * UPD_RS;
          __builtin_amdgcn_global_load_lds((glb_void_p)(nslot + wave * MAXD + lane),
                                           (lds_void_p)(stg + UpdStage::ring + wave * MAXD), 4, 0, 0);
        }
        if (lane == 0) s_pubw[wave] = pub_early;
      }
    }
    UPD_TS(9);
    if constexpr (LOCAL) {
      UPD_TS(10);
      UPD_TS(11);
    } else {
      // ---- Several gradient workgroups: the exchange of the step, without a grid barrier. Every workgroup's slab is on
      // its way as (value, sequence) words; workgroup b owns SLICE b of the parameter range:
      //   hop 1  it polls slice b of ALL slabs (nblk x SL words over its lanes, each lane spinning on its own words), sums
      //          them in slab order through an LDS scratch, and publishes the slice of the sum vector (again as words);
      //   hop 2  every workgroup polls the whole sum vector (its parameters-per-thread words per lane).
      // Two one-way trips through memory; nobody waits for a store acknowledgement, a release fence or an arrival count
      // (before: write-through stores acknowledged, fence, arrive, spin, acquire, then every workgroup read all nblk slabs --
      // 224 KB through one CU's L1 per step at 16 slabs, two levels beyond 32). Slabs and sum vector are double-buffered by
      // step parity: a word of step s + 2 can only be written after every workgroup has published (hence finished reading
      // for) step s + 1. Row-sharded data parallelism adds one hop between the two: the slice of this RANK's sum goes to every
      // rank's receive area, the slice's owner sums the ranks' words in rank order and publishes the global slice.
      using u64 = unsigned long long;
      const int SL = (P8 + nblk - 1) / nblk;   // slice length
      const unsigned rcpSL = 0xffffffffu / (unsigned)SL + 1u;   // = ceil(2^32 / SL), 32-bit divide
      float* red = lds + L::a1;                // scratch [nblk][SL] (the activation tiles are free until the next minibatch)
      auto valid_el = [&](int gi) { return gi < o.total || (gi >= w.P4 && gi < w.P4 + 5); };   // written elements only
      bool fail = false;
      const long long t0 = wall_clock64();
      const long long local_timeout = 200000000ll;   // 2 s of the 100 MHz clock
      u64* sums_s = w.sums64 + (long long)(s & 1) * P8;
      // one slice sum on its way: to the sum vector, or (row-sharded) into every rank's receive area
      auto emit = [&](int gi, float part) {
        if constexpr (SHARD) {
          const unsigned xseq = sh.seq_base + (unsigned)s + 1u;
          for (int rr = 0; rr < sh.world; ++rr)
            ll_store_system(sh.peer_recv[rr] + (long long)(xpar * sh.world + (sh.loopback ? rr : sh.rank)) * P8 + gi, part, xseq);
        } else if (ll_xcd) {
          ll_store_xcd(sums_s + gi, part, lseq);
        } else {
          ll_store_agent(sums_s + gi, part, lseq);
        }
      };
      {
      {   // hop 1: this workgroup's slice of every slab. (Tried: sixteen slabs with a 16-lane row per element and the
          // row sums on the DPP path -- no LDS scratch, no barrier --: the loads then touch sixteen cache lines per wave
          // instruction instead of four: 18.6 against 17.1 us per step.)
        u64 t[NPT];
        int gi_[NPT];
        unsigned need = 0u;
        int ez;
        asm volatile("s_mov_b32 %0, 0" : "=s"(ez));
#pragma unroll
        for (int k = 0; k < NPT; ++k) {
          const int f = tid + ez + k * 512;
          const int j = (int)__umulhi((unsigned)f, rcpSL), e = f - j * SL;
          const int gi = vb * SL + e;
          gi_[k] = min(j, nblk - 1) * P8 + min(gi, P8 - 1);
          if (f < nblk * SL && valid_el(gi)) need |= 1u << k;
        }
        unsigned it = 0;
        for (;; ) {
#pragma unroll
          for (int k = 0; k < NPT; ++k)   // (unconditional loads at clamped addresses, all in flight together)
            t[k] = __hip_atomic_load(slabs_s + gi_[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __builtin_amdgcn_sched_barrier(0);
          bool ok = true;
#pragma unroll
          for (int k = 0; k < NPT; ++k) ok = ok && (!((need >> k) & 1u) || (unsigned)(t[k] >> 32) == lseq);
          if (ok) break;
          __builtin_amdgcn_s_sleep(1);
          if ((++it & 255u) == 0 &&
              (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) || wall_clock64() - t0 > local_timeout)) {
            fail = true;
            break;
          }
        }
        if (tstamp && tid == 0) tstamp[40] += it;   // (measurement build: hop 1's unsuccessful polls of thread 0)
#pragma unroll
        for (int k = 0; k < NPT; ++k) {
          const int f = tid + k * 512;
          if (f < nblk * SL) red[f] = ((need >> k) & 1u) ? __uint_as_float((unsigned)t[k]) : 0.f;
        }
      }
      UPD_TS(10);
      // (thread 0 used to read `published` into s_pub here: a round trip through the fabric ahead of this barrier in every
      //  workgroup, for a value that is overwritten below before anybody reads it)
      if (fail) s_fail = 1;
      __syncthreads();
      if (s_fail) {
        if (tid == 0) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
      }
      for (int e = tid; e < SL; e += 512) {   // slice sums in slab order (reads eight at a time), then on their way
        const int gi = vb * SL + e;
        float part = 0.f;
        int j = 0;
        for (; j + 8 <= nblk; j += 8) {
          float tt[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) tt[u] = red[(j + u) * SL + e];
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int u = 0; u < 8; ++u) part += tt[u];
        }
        for (; j < nblk; ++j) part += red[j * SL + e];
        if (valid_el(gi)) emit(gi, part);
      }
      }
      if constexpr (SHARD) {   // the ranks' words of this slice, summed in rank order; the global slice published locally
        const unsigned xseq = sh.seq_base + (unsigned)s + 1u;
        const u64* rbase = sh.recv + (long long)(xpar * sh.world) * P8;
        for (int e = tid; e < SL; e += 512) {
          const int gi = vb * SL + e;
          if (!valid_el(gi)) continue;
          u64 t[SHARD_WORLD_MAX];
          for (unsigned it = 0;;) {
#pragma unroll
            for (int rr = 0; rr < SHARD_WORLD_MAX; ++rr)
              t[rr] = __hip_atomic_load(rbase + (long long)min(rr, sh.world - 1) * P8 + gi, __ATOMIC_RELAXED,
                                        __HIP_MEMORY_SCOPE_SYSTEM);
            __builtin_amdgcn_sched_barrier(0);
            bool ok = true;
#pragma unroll
            for (int rr = 0; rr < SHARD_WORLD_MAX; ++rr) ok = ok && (rr >= sh.world || (unsigned)(t[rr] >> 32) == xseq);
            if (ok) break;
            __builtin_amdgcn_s_sleep(1);
            if ((++it & 255u) == 0 && (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ||
                                       wall_clock64() - t0 > sh.timeout_ticks)) {
              fail = true;
              break;
            }
          }
          if (fail) {   // a peer's record never came: NO word leaves under this step's number (a sibling workgroup would take it
                        // for the global slice and step its parameters before the error word is seen)
            __hip_atomic_store(err, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
          }
          float part = 0.f;
#pragma unroll
          for (int rr = 0; rr < SHARD_WORLD_MAX; ++rr)
            if (rr < sh.world) part += __uint_as_float((unsigned)t[rr]);
          if (ll_xcd) ll_store_xcd(sums_s + gi, part, lseq);
          else ll_store_agent(sums_s + gi, part, lseq);
        }
      }
      UPD_TS(11);
      // ---- while the sum vector is on its way: the NEXT minibatch's rows are normalised into the x tile -- 1.3 us of
      // the next chain that depends on the data only, not on the parameters Adam is about to write. Its rows (prefetched
      // behind this minibatch's chain) and its statistics slot (requested with them by waves 0-2, each if ITS early look at
      // the statistics workgroup's progress said published) HAVE landed: loads return in order and hop 1's polls, issued
      // after them, have returned in every wave ahead of hop 1's barrier -- no wait here (a `vmcnt(0)` would also wait for
      // the acknowledgement of the slice words just sent).
      const bool stage_ahead = (s + 1 < n_steps) && min(min(s_pubw[0], s_pubw[1]), s_pubw[2]) >= s + 2;   // workgroup-uniform
      if (stage_ahead) {
        const MbRows rn = rows_at(nxt_e, nxt_mb);
        int zz;
        asm volatile("s_mov_b32 %0, 0" : "=s"(zz));
        const int wv_ = __builtin_amdgcn_readfirstlane((tid + zz) >> 6);
        chain_stage_rows<false>(d, stg + UpdStage::ring, stg + UpdStage::ring + MAXD, row_lo + vb * ROWS, row_lim(rn), lds, stg,
                                wv_ >> 2, wv_ & 3, lane);
      }
      if (tid == 0) s_pub = stage_ahead ? s + 2 : 0;   // (what `have_ring` is formed from below)
      {   // hop 2: the whole sum vector. Wave 0 of workgroup 0 also takes the vector's TAIL (the loss-statistic sums it
          // writes out after the norm) in the same trip: polled behind the norm, it was one more round trip through the
          // fabric per step in the one workgroup every other workgroup's hop 1 then waits for.
        u64 t[NPT];
        int ez;
        asm volatile("s_mov_b32 %0, 0" : "=s"(ez));
        unsigned it = 0;
        for (; !fail;) {
#pragma unroll
          for (int k = 0; k < NPT; ++k)
            t[k] = __hip_atomic_load(sums_s + min(tid + ez + k * 512, o.total - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          tail_w = __hip_atomic_load(sums_s + w.P4 + ez + min(lane, 4), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __builtin_amdgcn_sched_barrier(0);
          bool ok = !want_tail || (unsigned)(tail_w >> 32) == lseq;
#pragma unroll
          for (int k = 0; k < NPT; ++k) ok = ok && (unsigned)(t[k] >> 32) == lseq;
          if (ok) break;
          __builtin_amdgcn_s_sleep(1);
          if ((++it & 255u) == 0 &&
              (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ||
               wall_clock64() - t0 > (SHARD ? sh.timeout_ticks : local_timeout))) {
            fail = true;
            break;
          }
        }
        if (tstamp && tid == 0) tstamp[41] += it;   // (hop 2's unsuccessful polls)
#pragma unroll
        for (int k = 0; k < NPT; ++k) g[k] = __uint_as_float((unsigned)t[k]);
      }
      if (fail) s_fail = 1;   // (no barrier of its own: s_fail and s_pub are read behind the norm's block barrier below)
      // (the statistics workgroup's ring follows workgroup 0's progress: every workgroup's minibatch s is behind it)
      if (vb == 0 && tid == 0) __hip_atomic_fetch_add(arrivals, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if constexpr (LOCAL) have_ring = (s + 1 < n_steps) && (s_pub >= s + 2);
    if (LOCAL && have_ring && wave < 3) {
      // the next step's statistics slot: global -> LDS directly (waves 0 / 1: mean / 1 / std columns, wave 2: the
      // advantage statistics -- 64 lanes wide, the staging slot has room), landed by prefetch_park's wait. Held in
      // registers across the update instead, the first value was spilled right here behind a `vmcnt` wait.
      const float* nslot = w.ring + ((s + 1) % UPD_RING) * UPD_RS;
      __builtin_amdgcn_global_load_lds((glb_void_p)(nslot + wave * MAXD + lane),
                                       (lds_void_p)(stg + UpdStage::ring + wave * MAXD), 4, 0, 0);
    }
    UPD_TS(2);

    // global norm, clip, Adam -- identical in every workgroup (g: the summed gradient; one gradient workgroup: its own)
    float sq = 0.f;
    if constexpr (SHARD && LOCAL) {
      // ---- exchange (one gradient workgroup per rank): g = this rank's partial gradient (identical in all of its workgroups) -> every rank's sum.
      // Every value travels as ONE 8-byte word (float bits, sequence number of the step): a naturally aligned 8-byte store
      // arrives whole, so the receiver needs no separate flag, the sender no acknowledgement wait, fence or flag store --
      // one one-way trip per step (the LL protocol of RCCL, which relies on the same 8-byte atomicity over xGMI).
      const int W = sh.world, J = sh.pieces;
      const unsigned seq = sh.seq_base + (unsigned)s + 1u;
      const int REC = w.P4 + 8;
      const int kp = (NPT + J - 1) / J;              // per-thread parameter slots per piece
      auto pack = [&](float v) { return ((unsigned long long)seq << 32) | (unsigned long long)__float_as_uint(v); };
      // (1) send: pair (peer p, piece j) = index p * J + j, dealt round-robin over this rank's workgroups
      for (int qi = vb; qi < W * J; qi += nblk) {    // (workgroup-uniform)
        const int p = qi / J, j = qi - p * J;
        const int src = sh.loopback ? p : sh.rank;
        unsigned long long* dst = sh.peer_recv[p] + (long long)(xpar * W + src) * REC;
#pragma unroll
        for (int k = 0; k < NPT; ++k) {
          const int i = tid + k * 512;
          if (k / kp == j && i < o.total) __hip_atomic_store(dst + i, pack(g[k]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        if (j == 0 && tid < 64) {
          // tail: this rank's loss-statistic sums (write-through partials of its workgroups). Lane (b8, k) = (lane >> 3,
          // lane & 7) loads the partials of workgroups b8, b8 + 8, ... all at once (one L2 round trip per eight workgroups, not
          // one per workgroup), then the eight lane groups fold in a fixed order.
          float st = 0.f;
          for (int b0 = 0; b0 < nblk; b0 += 8) {
            const int b = b0 + (lane >> 3);
            const float v = __hip_atomic_load(stat_base + min(b, nblk - 1) * 8 + (lane & 7), __ATOMIC_RELAXED,
                                              __HIP_MEMORY_SCOPE_AGENT);
            st += (b < nblk && (lane & 7) < 5) ? v : 0.f;
          }
          st += __shfl_down(st, 8, 64);
          st += __shfl_down(st, 16, 64);
          st += __shfl_down(st, 32, 64);
          if (lane < 8) __hip_atomic_store(dst + w.P4 + lane, pack(st), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
      }
      // (2) receive: the records in rank order, SHARD_RB at a time (SHARD_RB x NPT 8-byte loads in flight, all of them
      // past the caches); a thread spins on its own elements until every one carries this step's sequence number. A peer can
      // be one step ahead -- it then writes the OTHER parity's records -- never two (it needs this rank's record of the
      // step between).
      const unsigned long long* rbase = sh.recv + (long long)(xpar * W) * REC;
#pragma unroll
      for (int k = 0; k < NPT; ++k) g[k] = 0.f;
      bool fail = false;
      const long long t0 = wall_clock64();
      for (int r0 = 0; r0 < W && !fail; r0 += SHARD_RB) {
        unsigned long long t[SHARD_RB][NPT];
        int ez;   // opaque zero: element offsets are re-formed per batch instead of living across the step
        asm volatile("s_mov_b32 %0, 0" : "=s"(ez));
        for (unsigned it = 0;;) {
#pragma unroll
          for (int u = 0; u < SHARD_RB; ++u)
#pragma unroll
            for (int k = 0; k < NPT; ++k)   // (unconditional loads at clamped record indices: see "a load behind a branch")
              t[u][k] = __hip_atomic_load(rbase + (unsigned)min(r0 + u, W - 1) * (unsigned)REC +
                                              (unsigned)min(tid + ez + k * 512, o.total - 1),
                                          __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          __builtin_amdgcn_sched_barrier(0);
          bool ok = true;
#pragma unroll
          for (int u = 0; u < SHARD_RB; ++u)
#pragma unroll
            for (int k = 0; k < NPT; ++k) ok = ok && (unsigned)(t[u][k] >> 32) == seq;
          if (ok) break;
          __builtin_amdgcn_s_sleep(1);
          if ((++it & 255u) == 0 &&
              (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) || wall_clock64() - t0 > sh.timeout_ticks)) {
            fail = true;
            break;
          }
        }
#pragma unroll
        for (int u = 0; u < SHARD_RB; ++u)
          if (r0 + u < W) {   // (workgroup-uniform)
#pragma unroll
            for (int k = 0; k < NPT; ++k) g[k] += __uint_as_float((unsigned)t[u][k]);
          }
      }
      if (fail) {
        __hip_atomic_store(err, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_ok = 0;
      }
      __syncthreads();   // (s_ok was set to 1 by the grid wait above -- or, one workgroup, is set here)
      if (!s_ok) return;
    }
#pragma unroll
    for (int k = 0; k < NPT; ++k) {
      if (tid + k * 512 >= o.total) g[k] = 0.f;
      sq += g[k] * g[k];
    }
    UPD_TS(4);
    int rz;   // opaque zero: the reduction scratch address is re-formed here instead of living in a (spilled) register
    asm volatile("s_mov_b32 %0, 0" : "=s"(rz));
    const float total_sq = block_sum512_dpp(sq, lds + L::scratch + rz, s & 1);
    if constexpr (!LOCAL) {   // (the block sum's barrier published hop 2's verdict and thread 0's s_pub)
      if (s_fail) {
        if (tid == 0) __hip_atomic_store(err, SHARD ? 2u : 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
      }
      have_ring = (s + 1 < n_steps) && (s_pub >= s + 2);
    }
    UPD_TS(5);
    const float total_norm = sqrtf(total_sq);
    const float coef = fminf(max_norm / (total_norm + 1e-6f), 1.0f);  // torch clip_grad_norm_
    if constexpr (!LOCAL) {
      if (vb == 0 && tid < 64 && stats) {   // the loss-statistic sums are the sum vector's tail: same rows as write_loss_stats
        const float st = __uint_as_float((unsigned)tail_w) * (1.f / (float)r.batch);   // (arrived with hop 2)
        const float st0 = __shfl(st, 0, 64), st1 = __shfl(st, 1, 64), st2 = __shfl(st, 2, 64);
        float* so = stats + (long long)(sch_first + s) * 8;
        if (lane < 5) so[lane] = st;
        if (lane == 5) so[5] = st0 + ent_coef * st2 + vf_coef * st1;
        if (lane == 6) so[6] = total_norm;
        if (lane == 7) so[7] = coef;
      }
    } else if constexpr (SHARD) {
      if (vb == 0 && tid < 64 && stats) {   // every rank's sums are in the step's records: same rows as write_loss_stats
        const unsigned long long* rbase = sh.recv + (long long)(xpar * sh.world) * (w.P4 + 8) + w.P4;
        const unsigned seq = sh.seq_base + (unsigned)s + 1u;
        // lane (rank, k) = (lane >> 3, lane & 7) takes that rank's sum k: ONE round trip for all ranks (the tails travel
        // with piece 0 of each record; the wait is bounded like every other), then the ranks fold in rank order
        const int rk = min(lane >> 3, sh.world - 1);
        unsigned long long tv;
        const long long t0 = wall_clock64();
        bool late = false;
        do {
          tv = __hip_atomic_load(rbase + rk * (w.P4 + 8) + (lane & 7), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          late = !__all((unsigned)(tv >> 32) == seq);
        } while (late && wall_clock64() - t0 < sh.timeout_ticks);
        if (late && lane == 0)   // (stale tail words would be logged as this step's statistics: the host raises instead)
          __hip_atomic_store(err, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        float st = (lane >> 3) < sh.world ? __uint_as_float((unsigned)tv) : 0.f;
        {
          float acc = __shfl(st, lane & 7, 64);
          for (int r_ = 1; r_ < sh.world; ++r_) acc += __shfl(st, r_ * 8 + (lane & 7), 64);
          st = acc * (1.f / (float)r.batch);
        }
        const float st0 = __shfl(st, 0, 64), st1 = __shfl(st, 1, 64), st2 = __shfl(st, 2, 64);
        float* so = stats + (long long)(sch_first + s) * 8;
        if (lane < 5) so[lane] = st;
        if (lane == 5) so[5] = st0 + ent_coef * st2 + vf_coef * st1;
        if (lane == 6) so[6] = total_norm;
        if (lane == 7) so[7] = coef;
      }
    } else if (vb == 0) {
      if (s + 1 < n_steps) {  // the statistics block writes this step's loss statistics later
        if (tid == 0) {   // (write-through: the local form's arrive has no release fence ahead of it)
          slab_store(w.normcoef + (s % UPD_SD) * 2, total_norm);
          slab_store(w.normcoef + (s % UPD_SD) * 2 + 1, coef);
        }
      } else {
        write_loss_stats(s, total_norm, coef);
      }
    }
    {
      // torch.optim.Adam's step on the thread's NPT parameters. All LDS reads (parameter, index into the transposed
      // copy) first, then the arithmetic, then all LDS writes: written element by element, every iteration's reads had to
      // wait for the previous iteration's stores (the compiler cannot tell sPt[dstT[i]] from sP[i']) -- eight serial
      // LDS round trips + eight undivided IEEE division / square-root chains, 1.6 us per step.
      float pv[NPT];
      int dt[NPT];
#pragma unroll
      for (int k = 0; k < NPT; ++k) {
        const int ic = min(tid + k * 512, o.total - 1);
        pv[k] = sP[ic];
        dt[k] = dstT[ic];
      }
      __builtin_amdgcn_sched_barrier(0);
      // Round 5: torch's `sqrt(v) / sqrt(bc2) + eps` and `m / denom` on the hardware's v_sqrt_f32 / v_rcp_f32 (1 ulp each) with
      // one Newton step on the reciprocal, the division by sqrt(bc2) as a multiplication by its reciprocal (formed once per
      // step): ~9 VALU operations per parameter instead of ~35 (two IEEE division expansions and a square-root expansion --
      // ~400 instructions per wave and step in EVERY workgroup). The step differs from torch's by <= ~2 ulp of lr * m / denom
      // (~1e-10 absolute at lr = 3e-4); the full-size parity tests re-measured: `profiles/r05_ppo_ab.md`.
      const float inv_bc2 = 1.f / bc2_sqrt;
#pragma unroll
      for (int k = 0; k < NPT; ++k) {
        const float gi = g[k] * coef;
        float mi = rm[k];
        mi = mi + (gi - mi) * (1.f - beta1);
        const float vi = rv[k] * beta2 + (1.f - beta2) * gi * gi;
        const float denom = __builtin_amdgcn_sqrtf(vi) * inv_bc2 + eps;
        float rd = __builtin_amdgcn_rcpf(denom);
        rd = __builtin_fmaf(rd, __builtin_fmaf(-denom, rd, 1.f), rd);
        pv[k] = pv[k] - step_size * (mi * rd);
        rm[k] = mi;     // (threads past the parameter count carry zeros: g = 0 keeps m = v = 0; nothing of theirs is stored)
        rv[k] = vi;
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int k = 0; k < NPT; ++k) {
        const int i = tid + k * 512;
        if (i < o.total) {
          sP[i] = pv[k];
          if (dt[k] != 0xffff) sPt[dt[k]] = pv[k];
        }
      }
    }
    UPD_TS(6);
    if (s + 1 < n_steps && (LOCAL || !have_ring)) prefetch_park();   // (its vmcnt(0) also covers the statistics slot's LDS-direct loads;
                                                                     //  several workgroups: staged ahead, nothing left to land)
    else if (local) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (local && tid == 0)   // only the statistics workgroup listens: this step's loss partials and norm / clip pair were
                             // stored write-through and every wave has drained its stores ahead of the barrier above
      __hip_atomic_fetch_add(arrivals + 16 * (vb & (UPD_ARR - 1)), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    UPD_TS(3);
  }
#undef UPD_TS
  if (tstamp && vb == 0 && tid == 0)
    for (int k = 0; k < 12; ++k) tstamp[k] += tacc[k];
  if (vb == 0 && tid == 0)
    __hip_atomic_store(w.ctrl + 9, lseq_base + (unsigned)n_steps, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (vb == 0) {
#pragma unroll
    for (int k = 0; k < NPT; ++k) {
      const int i = tid + k * 512;
      if (i < o.total) {
        const float pi_ = sP[i];
        P[i] = pi_;
        int dst = i;   // element i's place in the global transposed copy (`transpose_params_kernel`'s layout)
        auto tr = [&](int base, int rows, int cols) {
          if (i >= base && i < base + rows * cols) {
            const int rr = (i - base) / cols, cc = (i - base) % cols;
            dst = base + cc * rows + rr;
          }
        };
        tr(o.pW1, H, D); tr(o.pW2, H, H); tr(o.vW1, H, D); tr(o.vW2, H, H);
        Pt[dst] = pi_;
        m[i] = rm[k];
        v[i] = rv[k];
      }
    }
  }
}

inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

template <int H>
size_t lds_bytes() { return Lds<H>::total * sizeof(float); }

template <typename K>
int set_lds(K kern, size_t bytes) {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)bytes);
  return e == hipSuccess ? IA_OK : (int)e;
}

bool g_epoch_split = false;  // tuning/debug: two launches per minibatch for 64-wide towers too
bool g_epoch_whole = false;  // tuning/debug: the one-launch epoch with whole row-block workgroups (8 waves, both towers)
bool g_epoch_barriers = false;   // tuning/debug: the one-tower epoch kernel with grid barriers instead of the word exchange
bool g_epoch_chain2 = true;      // the word-exchange epoch kernel with round 5's transposed register-resident chain (observation
                                 // widths <= 32); false: round 4's gradient body everywhere
bool g_epoch_rows32 = true;      // round 6: that kernel on 32-row blocks where they fit (false: 64-row blocks, eight waves, everywhere)
bool g_epoch_quarters = true;    // round 6: ... with eight waves per 32-row block (two row groups x four feature quarters)
long long* g_epoch_dbg = nullptr;  // measurement: phase ticks of workgroup 0 of ppo_epoch_persistent_kernel
}  // namespace

extern "C" {

int64_t ia_policy_param_count(const ia_policy_desc* d) {
  if (!pol_ok(d)) return IA_ERR_ARG;
  return pol_offsets(d->obs_dim, d->act_dim, d->hidden, d->discrete).total;
}

int ia_policy_transpose(const ia_policy_desc* d, const float* params, float* params_t, void* stream) {
  if (!pol_ok(d)) return IA_ERR_ARG;
  hipLaunchKernelGGL(transpose_params_kernel, dim3(8), dim3(256), 0, (hipStream_t)stream, *d, params, params_t);
  IA_CHECK_LAUNCH();
  return IA_OK;
}

int ia_policy_act(const ia_policy_desc* d, const float* params, const float* params_t, const float* norm_mean,
                  const float* norm_var, const float* obs, int n, const float* noise, const float* low,
                  const float* high, float* actions, float* clipped, float* values, float* logp, void* stream) {
  if (!pol_ok(d) || n <= 0) return IA_ERR_ARG;
  int rc;
  if (d->hidden == 32) {
    static bool attr = false;
    const size_t bytes = ALds<32>::total * sizeof(float);
    if (!attr) { if ((rc = set_lds(policy_act_mfma_kernel<32, false>, bytes))) return rc; attr = true; }
    hipLaunchKernelGGL((policy_act_mfma_kernel<32, false>), dim3(cdiv(n, ROWS)), dim3(512), bytes, (hipStream_t)stream, *d,
                       params, params_t, norm_mean, norm_var, obs, n, noise, low, high, actions, clipped, values, logp);
  } else if (d->hidden == 64) {
    static bool attr = false;
    const size_t bytes = ALds<64>::total * sizeof(float);
    if (!attr) { if ((rc = set_lds(policy_act_mfma_kernel<64, false>, bytes))) return rc; attr = true; }
    hipLaunchKernelGGL((policy_act_mfma_kernel<64, false>), dim3(cdiv(n, ROWS)), dim3(512), bytes, (hipStream_t)stream, *d,
                       params, params_t, norm_mean, norm_var, obs, n, noise, low, high, actions, clipped, values, logp);
  } else {
    return IA_ERR_UNSUPPORTED;
  }
  IA_CHECK_LAUNCH();
  return IA_OK;
}

// The rollout's act steps as ONE resident launch driven through flags in pinned host memory (policy_rollout_mailbox_kernel).
// Strides are in floats between consecutive steps (0: the same tile every step). IA_ERR_UNSUPPORTED: shapes the
// matrix-core act kernel does not cover -- the caller then launches `ia_policy_act` per step.
int ia_policy_rollout_mailbox(const ia_policy_desc* d, const float* params, const float* params_t, const float* norm_mean,
                              const float* norm_var, int n, const float* low, const float* high, const float* obs,
                              int64_t s_obs, const float* noise, int64_t s_noise, float* actions, int64_t s_act,
                              float* clipped, int64_t s_clip, float* values, int64_t s_val, float* logp, int64_t s_lp,
                              float* last_val, int T, const int32_t* ready, int32_t* done, double timeout_s,
                              void* stream) {
  if (!pol_ok(d) || n <= 0 || T <= 0 || !ready || !done || !obs || !actions || !clipped || !values || !logp)
    return IA_ERR_ARG;
  if (d->hidden != 32 && d->hidden != 64) return IA_ERR_UNSUPPORTED;
  {
    // every workgroup stays resident for the whole rollout and the host waits for ALL of them at every step: more
    // workgroups than the device can hold at once would leave the surplus waiting for the residents forever
    static int dev_cus = 0;
    if (dev_cus == 0) {
      int dev = 0;
      if (hipGetDevice(&dev) != hipSuccess ||
          hipDeviceGetAttribute(&dev_cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
        return IA_ERR_ARG;
    }
    if (cdiv(n, ROWS) > dev_cus) return IA_ERR_UNSUPPORTED;   // (one workgroup per CU is always possible)
  }
  ActMailbox mb{obs, s_obs, noise, s_noise, actions, s_act, clipped, s_clip, values, s_val, logp, s_lp, last_val, T,
                reinterpret_cast<const int*>(ready), reinterpret_cast<int*>(done), (long long)(timeout_s * 1e8)};
  int rc;
  if (d->hidden == 32) {
    static bool attr = false;
    const size_t bytes = ALds<32>::total * sizeof(float);
    if (!attr) { if ((rc = set_lds(policy_rollout_mailbox_kernel<32>, bytes))) return rc; attr = true; }
    hipLaunchKernelGGL((policy_rollout_mailbox_kernel<32>), dim3(cdiv(n, ROWS)), dim3(512), bytes, (hipStream_t)stream,
                       *d, params, params_t, norm_mean, norm_var, n, low, high, mb);
  } else {
    static bool attr = false;
    const size_t bytes = ALds<64>::total * sizeof(float);
    if (!attr) { if ((rc = set_lds(policy_rollout_mailbox_kernel<64>, bytes))) return rc; attr = true; }
    hipLaunchKernelGGL((policy_rollout_mailbox_kernel<64>), dim3(cdiv(n, ROWS)), dim3(512), bytes, (hipStream_t)stream,
                       *d, params, params_t, norm_mean, norm_var, n, low, high, mb);
  }
  IA_CHECK_LAUNCH();
  return IA_OK;
}

int ia_policy_logits_mailbox(const ia_policy_desc* d, const float* params, const float* params_t, const float* norm_mean,
                             const float* norm_var, int n, const float* obs, int64_t s_obs, float* logits, float* values,
                             int64_t s_val, int T, const int32_t* ready, int32_t* done, double timeout_s, void* stream) {
  if (!pol_ok(d) || n <= 0 || T <= 0 || !ready || !done || !obs || !logits) return IA_ERR_ARG;
  {
    static int dev_cus = 0;
    if (dev_cus == 0) {
      int dev = 0;
      if (hipGetDevice(&dev) != hipSuccess ||
          hipDeviceGetAttribute(&dev_cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
        return IA_ERR_ARG;
    }
    if (cdiv(n, ROWS) > dev_cus) return IA_ERR_UNSUPPORTED;   // (every workgroup must be resident: see the act mailbox)
  }
  LogitsMailbox mb{obs, s_obs, logits, values, s_val, T, reinterpret_cast<const int*>(ready),
                   reinterpret_cast<int*>(done), (long long)(timeout_s * 1e8)};
  int rc;
  {   // the MFMA body (both towers of 64 rows per 512-thread workgroup)
    if (d->hidden == 32) {
      const size_t bytes = ALds<32>::total * sizeof(float);
      if ((rc = set_lds(policy_logits_mailbox_mfma_kernel<32>, bytes))) return rc;
      hipLaunchKernelGGL(policy_logits_mailbox_mfma_kernel<32>, dim3(cdiv(n, ROWS)), dim3(512), bytes, (hipStream_t)stream, *d,
                         params, params_t, norm_mean, norm_var, n, mb);
    } else {
      const size_t bytes = ALds<64>::total * sizeof(float);
      if ((rc = set_lds(policy_logits_mailbox_mfma_kernel<64>, bytes))) return rc;
      hipLaunchKernelGGL(policy_logits_mailbox_mfma_kernel<64>, dim3(cdiv(n, ROWS)), dim3(512), bytes, (hipStream_t)stream, *d,
                         params, params_t, norm_mean, norm_var, n, mb);
    }
    IA_CHECK_LAUNCH();
    return IA_OK;
  }
  return IA_ERR_UNSUPPORTED;
}

int ia_policy_evaluate(const ia_policy_desc* d, const float* params, const float* params_t, const float* norm_mean,
                       const float* norm_var, const float* obs, const float* actions, int n, float* logp,
                       float* values, float* entropy, void* stream) {
  if (!pol_ok(d) || n <= 0) return IA_ERR_ARG;
  int rc;
  if (d->hidden == 32 && (actions != nullptr || logp == nullptr)) {
    static bool attr = false;   // the MFMA layer chain of the rollout step, given actions instead of sampling
    const size_t bytes = ALds<32>::total * sizeof(float);
    if (!attr) { if ((rc = set_lds(policy_act_mfma_kernel<32, true>, bytes))) return rc; attr = true; }
    hipLaunchKernelGGL((policy_act_mfma_kernel<32, true>), dim3(cdiv(n, ROWS)), dim3(512), bytes, (hipStream_t)stream, *d,
                       params, params_t, norm_mean, norm_var, obs, n, actions, (const float*)nullptr,
                       (const float*)nullptr, (float*)nullptr, entropy, values, logp);
  } else if (d->hidden == 64 && (actions != nullptr || logp == nullptr)) {
    static bool attr = false;
    const size_t bytes = ALds<64>::total * sizeof(float);
    if (!attr) { if ((rc = set_lds(policy_act_mfma_kernel<64, true>, bytes))) return rc; attr = true; }
    hipLaunchKernelGGL((policy_act_mfma_kernel<64, true>), dim3(cdiv(n, ROWS)), dim3(512), bytes, (hipStream_t)stream, *d,
                       params, params_t, norm_mean, norm_var, obs, n, actions, (const float*)nullptr,
                       (const float*)nullptr, (float*)nullptr, entropy, values, logp);
  } else if (d->hidden == 32) {
    if ((rc = set_lds(policy_eval_kernel<32>, lds_bytes<32>()))) return rc;
    hipLaunchKernelGGL(policy_eval_kernel<32>, dim3(cdiv(n, ROWS)), dim3(ROWS), lds_bytes<32>(),
                       (hipStream_t)stream, *d, params, params_t, norm_mean, norm_var, obs, actions, n, logp, values,
                       entropy);
  } else {
    if ((rc = set_lds(policy_eval_kernel<64>, lds_bytes<64>()))) return rc;
    hipLaunchKernelGGL(policy_eval_kernel<64>, dim3(cdiv(n, ROWS)), dim3(ROWS), lds_bytes<64>(),
                       (hipStream_t)stream, *d, params, params_t, norm_mean, norm_var, obs, actions, n, logp, values,
                       entropy);
  }
  IA_CHECK_LAUNCH();
  return IA_OK;
}

int ia_policy_logits(const ia_policy_desc* d, const float* params, const float* params_t, const float* norm_mean,
                     const float* norm_var, const float* obs, int n, float* logits, float* values, void* stream) {
  if (!pol_ok(d) || n <= 0 || !logits) return IA_ERR_ARG;
  int rc;
  {   // the MFMA body: the same arithmetic as the mailbox kernel's steps
    if (d->hidden == 32) {
      const size_t bytes = ALds<32>::total * sizeof(float);
      if ((rc = set_lds(policy_logits_mfma_kernel<32>, bytes))) return rc;
      hipLaunchKernelGGL(policy_logits_mfma_kernel<32>, dim3(cdiv(n, ROWS)), dim3(512), bytes, (hipStream_t)stream, *d, params,
                         params_t, norm_mean, norm_var, obs, n, logits, values);
    } else {
      const size_t bytes = ALds<64>::total * sizeof(float);
      if ((rc = set_lds(policy_logits_mfma_kernel<64>, bytes))) return rc;
      hipLaunchKernelGGL(policy_logits_mfma_kernel<64>, dim3(cdiv(n, ROWS)), dim3(512), bytes, (hipStream_t)stream, *d, params,
                         params_t, norm_mean, norm_var, obs, n, logits, values);
    }
    IA_CHECK_LAUNCH();
    return IA_OK;
  }
  return IA_ERR_UNSUPPORTED;
}

int ia_gae(const float* rewards, const float* values, const float* episode_starts, const float* last_values,
           const float* last_dones, int T, int n, float gamma, float gae_lambda, float* advantages,
           float* returns, void* stream) {
  if (T <= 0 || n <= 0) return IA_ERR_ARG;
  // gamma*lambda is formed in double on the host side of the reference, then rounded to f32
  const float gl = (float)((double)gamma * (double)gae_lambda);
  hipLaunchKernelGGL(gae_kernel, dim3(cdiv(n, 64)), dim3(64), 0, (hipStream_t)stream, rewards, values,
                     episode_starts, last_values, last_dones, T, n, gamma, gl, advantages, returns);
  IA_CHECK_LAUNCH();
  return IA_OK;
}

int ia_timeout_bootstrap(float* rewards, const float* terminal_values, const uint8_t* truncated, float gamma,
                         int64_t n, void* stream) {
  if (n <= 0) return IA_ERR_ARG;
  hipLaunchKernelGGL(bootstrap_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, rewards,
                     terminal_values, truncated, gamma, (long long)n);
  IA_CHECK_LAUNCH();
  return IA_OK;
}

// (everything but the word areas of ppo_epoch_ll_kernel, which follow at the next 8-byte boundary)
static int64_t ppo_ws_plain_floats(const ia_policy_desc* d, int batch, int64_t gather_rows) {
  const int nblk = cdiv(batch, ROWS);
  const int P = pol_offsets(d->obs_dim, d->act_dim, d->hidden, d->discrete).total;
  const int aw = d->discrete ? 1 : d->act_dim;
  const int64_t n_mb = (gather_rows + batch - 1) / batch;
  const int64_t n = 8 + (int64_t)nblk * 8 + (int64_t)nblk * P + P + gather_rows * (d->obs_dim + aw + 3) +
                    n_mb * (EPS_PART + EPS_SEQ);   // + ia_ppo_epoch's per-minibatch statistics (raw moments, published form)
  return (n + 1) & ~(int64_t)1;
}

int64_t ia_ppo_ws_floats(const ia_policy_desc* d, int batch, int64_t gather_rows) {
  if (!pol_ok(d) || batch <= 0 || gather_rows < batch) return IA_ERR_ARG;
  const int P = pol_offsets(d->obs_dim, d->act_dim, d->hidden, d->discrete).total;
  // (64-wide towers: the word exchange, laid out for 32-row blocks where they apply -- twice the slabs)
  const int r64 = epoch_ll_row_blocks(cdiv(batch, ROWS), P), r32 = epoch_ll_row_blocks(cdiv(batch, 32), P);
  const int64_t ll = d->hidden == 64 ? 2 * epoch_ll_words(r32 <= 32 ? r32 : r64, P) : 0;
  return ppo_ws_plain_floats(d, batch, gather_rows) + ll;
}

}  // extern "C" (helpers below are C++)

namespace {

long long* g_tstamp = nullptr;  // debug: device buffer of >= 16 clocks (ia_ppo_debug_timing)

struct Gathered {  // contiguous (permuted-order) copies of the minibatch rows, inside ws
  float *obs, *act, *logp, *adv, *ret;
};
inline Gathered gathered_region(const ia_policy_desc* d, float* ws, int max_batch, long long rows) {
  const int P = pol_offsets(d->obs_dim, d->act_dim, d->hidden, d->discrete).total;
  const int nblk = cdiv(max_batch, ROWS);
  const int aw = d->discrete ? 1 : d->act_dim;
  Gathered g;
  g.obs = ws + 8 + (long long)nblk * 8 + (long long)nblk * P + P;
  g.act = g.obs + rows * d->obs_dim;
  g.logp = g.act + rows * aw;
  g.adv = g.logp + rows;
  g.ret = g.adv + rows;
  return g;
}

struct PpoArgs {
  const ia_policy_desc* d;
  float *params, *params_t, *norm_mean, *norm_var;
  int32_t* norm_count;
  int update_norm;
  const float *obs, *actions, *old_logp, *advantages, *returns;
  int T, n_envs, normalize_adv;
  float clip_range, ent_coef, vf_coef, max_grad_norm;
  float *exp_avg, *exp_avg_sq;
  float beta1, beta2, adam_eps;
  float* ws;
  hipStream_t st;
  const float* advstat = nullptr;   // {advantage mean, std} of this minibatch (null: ws[0..1], left by the previous launch)
};

int launch_gather(const PpoArgs& a, const int64_t* perm, long long rows, const Gathered& g,
                  unsigned long long* zero_words = nullptr, long long n_zero = 0, unsigned* zero_ctl = nullptr) {
  const int aw = a.d->discrete ? 1 : a.d->act_dim;
  const long long elems = rows * (a.d->obs_dim + aw + 3);
  hipLaunchKernelGGL(ppo_epoch_gather_kernel, dim3(cdiv(elems, 256)), dim3(256), 0, a.st, a.obs, a.actions, a.old_logp,
                     a.advantages, a.returns, perm, rows, a.T, a.n_envs, a.d->obs_dim, aw, g.obs, g.act, g.logp, g.adv,
                     g.ret, zero_words, n_zero, zero_ctl);
  IA_CHECK_LAUNCH();
  return IA_OK;
}

// view of `a` whose row 0 is row `start` of the gathered arrays (kernels then run with idx == null)
PpoArgs at_rows(const PpoArgs& a, const Gathered& g, long long start) {
  const int aw = a.d->discrete ? 1 : a.d->act_dim;
  PpoArgs v = a;
  v.obs = g.obs + start * a.d->obs_dim;
  v.actions = g.act + start * aw;
  v.old_logp = g.logp + start;
  v.advantages = g.adv + start;
  v.returns = g.ret + start;
  return v;
}

int launch_prepare(const PpoArgs& a, const int64_t* idx, int batch) {
  static bool attr = false;
  const size_t bytes = PREP_LDS_FLOATS * sizeof(float);
  if (!attr) { int rc = set_lds(ppo_prepare_kernel, bytes); if (rc) return rc; attr = true; }
  const PolOff o = pol_offsets(a.d->obs_dim, a.d->act_dim, a.d->hidden, a.d->discrete);
  const PpoWs w = ppo_ws(a.ws, cdiv(batch, ROWS), o.total);
  hipLaunchKernelGGL(ppo_prepare_kernel, dim3(1), dim3(PREP_THREADS), bytes, a.st, *a.d, a.obs, a.advantages, idx,
                     batch, a.T, a.n_envs, a.update_norm, a.norm_mean, a.norm_var, a.norm_count, w.advstat);
  IA_CHECK_LAUNCH();
  return IA_OK;
}


template <int H>
int launch_grad(const PpoArgs& a, const int64_t* idx, int batch) {
  const int nblk = cdiv(batch, ROWS);
  // matrix-core gradient kernel: H = 32 with the parameters copied into LDS, H = 64 reading its fragments from L2
  const int P4 = (pol_offsets(a.d->obs_dim, a.d->act_dim, H, a.d->discrete).total + 3) & ~3;
  const size_t mbytes = (GLds<H>::total + (H == 32 ? 2 * P4 : 0)) * sizeof(float);
  static bool attr2 = false;
  if (!attr2) { int rc = set_lds(ppo_grad_mfma_kernel<H>, 160 * 1024); if (rc) return rc; attr2 = true; }
  hipLaunchKernelGGL(ppo_grad_mfma_kernel<H>, dim3(nblk), dim3(512), mbytes, a.st, *a.d, a.params, a.params_t,
                     a.norm_mean, a.norm_var, a.obs, a.actions, a.old_logp, a.advantages, a.returns, idx, batch, a.T,
                     a.n_envs, a.normalize_adv, a.clip_range, a.ent_coef, a.vf_coef, a.ws, nblk, g_tstamp, a.advstat);
  IA_CHECK_LAUNCH();
  return IA_OK;
}

// grad + apply for one minibatch whose statistics are already in ws; the apply kernel also
// prepares minibatch `next_idx` (next_batch == 0: nothing follows). NOTE: both minibatches share
// `ws`, whose slab region is sized by the LARGER batch; advstat sits at ws[0..7] for any size.
int launch_minibatch_next(const PpoArgs& a, int batch, float step_size, float bc2_sqrt, float* stats,
                          const PpoArgs& nxt, int next_batch);

// reduce + clip + Adam (+ the next minibatch's statistics: in an extra workgroup when `stats_block`, else behind Adam)
int launch_apply_split(const PpoArgs& a, int batch, float step_size, float bc2_sqrt, float* stats, const float* nobs,
                       const float* nadv, const int64_t* next_idx, int next_batch, int stats_block) {
  static bool attr = false;
  const size_t bytes = PREP_LDS_FLOATS * sizeof(float);
  if (!attr) { int rc = set_lds(ppo_apply_split_kernel, bytes); if (rc) return rc; attr = true; }
  const int nblk = cdiv(batch, ROWS);
  const int total = pol_offsets(a.d->obs_dim, a.d->act_dim, a.d->hidden, a.d->discrete).total;
  const int G = std::max(1, std::min(std::min(nblk, 16), cdiv(total, PREP_THREADS)));
  hipLaunchKernelGGL(ppo_apply_split_kernel, dim3(G + (stats_block ? 1 : 0)), dim3(PREP_THREADS), bytes, a.st, *a.d,
                     a.params, a.params_t, a.exp_avg, a.exp_avg_sq, a.ws, nblk, batch, a.max_grad_norm, a.ent_coef,
                     a.vf_coef, a.beta1, a.beta2, a.adam_eps, step_size, bc2_sqrt, stats, nobs, nadv, next_idx, next_batch,
                     a.T, a.n_envs, a.update_norm, a.norm_mean, a.norm_var, a.norm_count, G, stats_block);
  IA_CHECK_LAUNCH();
  return IA_OK;
}

int launch_minibatch(const PpoArgs& a, const int64_t* idx, int batch, float step_size, float bc2_sqrt, float* stats,
                     const int64_t* next_idx, int next_batch) {
  int rc = a.d->hidden == 32 ? launch_grad<32>(a, idx, batch) : launch_grad<64>(a, idx, batch);
  if (rc) return rc;
  return launch_apply_split(a, batch, step_size, bc2_sqrt, stats, a.obs, a.advantages, next_idx, next_batch, 0);
}

// contiguous-rows form: `a` = this minibatch's rows, `nxt` = the next minibatch's rows (for its statistics)
int launch_minibatch_next(const PpoArgs& a, int batch, float step_size, float bc2_sqrt, float* stats,
                          const PpoArgs& nxt, int next_batch) {
  int rc = a.d->hidden == 32 ? launch_grad<32>(a, nullptr, batch) : launch_grad<64>(a, nullptr, batch);
  if (rc) return rc;
  return launch_apply_split(a, batch, step_size, bc2_sqrt, stats, nxt.obs, nxt.advantages, nullptr, next_batch,
                            next_batch > 0 ? 1 : 0);
}

int launch_apply_phase(const PpoArgs& a, int batch, float step_size, float bc2_sqrt, float* stats, int phases) {
  static bool attr = false;
  const size_t bytes = PREP_LDS_FLOATS * sizeof(float);
  if (!attr) { int rc = set_lds(ppo_apply_kernel, bytes); if (rc) return rc; attr = true; }
  hipLaunchKernelGGL(ppo_apply_kernel, dim3(1), dim3(PREP_THREADS), bytes, a.st, *a.d, a.params, a.params_t, a.exp_avg,
                     a.exp_avg_sq, a.ws, cdiv(batch, ROWS), batch, a.max_grad_norm, a.ent_coef, a.vf_coef, a.beta1,
                     a.beta2, a.adam_eps, step_size, bc2_sqrt, stats, a.obs, a.advantages, nullptr, 0, a.T, a.n_envs, 0,
                     a.norm_mean, a.norm_var, a.norm_count, phases);
  IA_CHECK_LAUNCH();
  return IA_OK;
}

}  // namespace

extern "C" {

int ia_ppo_minibatch(const ia_policy_desc* d, float* params, float* params_t, float* norm_mean, float* norm_var,
                     int32_t* norm_count, int update_norm, const float* obs, const float* actions,
                     const float* old_logp, const float* advantages, const float* returns, const int64_t* idx,
                     int batch, int T, int n_envs, int normalize_adv, float clip_range, float ent_coef,
                     float vf_coef, float max_grad_norm, float* exp_avg, float* exp_avg_sq, float beta1,
                     float beta2, float adam_eps, float step_size, float bc2_sqrt, float* ws, float* stats,
                     void* stream) {
  if (!pol_ok(d) || batch <= 0) return IA_ERR_ARG;
  PpoArgs a{d, params, params_t, norm_mean, norm_var, norm_count, update_norm, obs, actions, old_logp, advantages,
            returns, T, n_envs, normalize_adv, clip_range, ent_coef, vf_coef, max_grad_norm, exp_avg, exp_avg_sq,
            beta1, beta2, adam_eps, ws, (hipStream_t)stream};
  const Gathered g = gathered_region(d, ws, batch, batch);
  int rc = launch_gather(a, idx, batch, g);
  if (rc) return rc;
  const PpoArgs v = at_rows(a, g, 0);
  rc = launch_prepare(v, nullptr, batch);
  if (rc) return rc;
  return launch_minibatch(v, nullptr, batch, step_size, bc2_sqrt, stats, nullptr, 0);
}

// Data-parallel split of a minibatch step (one rank per GPU): `_grad` = statistics + forward/backward
// + fixed-order slab reduction into the flat gradient at ia_ppo_grad_ptr(); the caller all-reduces
// that buffer over RCCL; `_apply` = clip_grad_norm_ + Adam on the reduced gradient.
int ia_ppo_minibatch_grad(const ia_policy_desc* d, float* params, float* params_t, float* norm_mean, float* norm_var,
                          int32_t* norm_count, int update_norm, const float* obs, const float* actions,
                          const float* old_logp, const float* advantages, const float* returns, const int64_t* idx,
                          int batch, int T, int n_envs, int normalize_adv, float clip_range, float ent_coef,
                          float vf_coef, float* ws, void* stream) {
  if (!pol_ok(d) || batch <= 0) return IA_ERR_ARG;
  PpoArgs a{d, params, params_t, norm_mean, norm_var, norm_count, update_norm, obs, actions, old_logp, advantages,
            returns, T, n_envs, normalize_adv, clip_range, ent_coef, vf_coef, 0.f, nullptr, nullptr, 0.f, 0.f, 0.f, ws,
            (hipStream_t)stream};
  const Gathered g = gathered_region(d, ws, batch, batch);
  int rc = launch_gather(a, idx, batch, g);
  if (rc) return rc;
  const PpoArgs v = at_rows(a, g, 0);
  rc = launch_prepare(v, nullptr, batch);
  if (rc) return rc;
  rc = d->hidden == 32 ? launch_grad<32>(v, nullptr, batch) : launch_grad<64>(v, nullptr, batch);
  if (rc) return rc;
  return launch_apply_phase(v, batch, 0.f, 1.f, nullptr, 1);
}

int ia_ppo_debug_timing(void* device_buffer_16xi64) {
  g_tstamp = (long long*)device_buffer_16xi64;
  return IA_OK;
}

// Tuning / measurement: 1 = ia_ppo_epoch launches the gradient and apply kernels per minibatch even where the
// one-launch-per-epoch kernel applies (64-wide towers).
int ia_ppo_epoch_split(int on) {
  g_epoch_split = on == 1;
  g_epoch_whole = on == 2;
  g_epoch_barriers = on == 3;
  g_epoch_rows32 = on != 6;   // 6: 64-row blocks on eight waves also where the 32-row blocks apply
  g_epoch_quarters = on != 7;            // 7: 32-row blocks on four waves (feature halves) instead of eight (quarters)
  g_epoch_chain2 = on != 4;   // 4: round 4's gradient body in the word-exchange kernel also where round 5's chain applies
  return IA_OK;
}

int ia_ppo_epoch_debug_timing(void* device_buffer_64xi64) {
  g_epoch_dbg = (long long*)device_buffer_64xi64;
  return IA_OK;
}

int64_t ia_ppo_grad_offset(const ia_policy_desc* d, int batch) {
  if (!pol_ok(d) || batch <= 0) return IA_ERR_ARG;
  const int nblk = cdiv(batch, ROWS);
  const int P = pol_offsets(d->obs_dim, d->act_dim, d->hidden, d->discrete).total;
  return 8 + (int64_t)nblk * 8 + (int64_t)nblk * P;  // float offset of the reduced gradient inside ws
}

int ia_ppo_minibatch_apply(const ia_policy_desc* d, float* params, float* params_t, int batch, float ent_coef,
                           float vf_coef, float max_grad_norm, float* exp_avg, float* exp_avg_sq, float beta1,
                           float beta2, float adam_eps, float step_size, float bc2_sqrt, float* ws, float* stats,
                           void* stream) {
  if (!pol_ok(d) || batch <= 0) return IA_ERR_ARG;
  PpoArgs a{d, params, params_t, nullptr, nullptr, nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, 0,
            0.f, ent_coef, vf_coef, max_grad_norm, exp_avg, exp_avg_sq, beta1, beta2, adam_eps, ws,
            (hipStream_t)stream};
  return launch_apply_phase(a, batch, step_size, bc2_sqrt, stats, 2);
}

// One full PPO epoch (SB3 PPO.train inner loop over RolloutBuffer.get): `perm` is the host-drawn
// np.random.permutation(T*n_envs) already resident on the device; minibatches are consecutive
// slices of it (the last one may be short). Adam's bias corrections are formed in double per step.
// Launches: 1 prepare + 2 per minibatch.
// `n_epochs` consecutive epochs as ONE sequence of minibatches (`perm` = the epochs' permutations back to back): valid when
// the rollout divides into whole minibatches (no minibatch then straddles two epochs and every minibatch is the one the
// per-epoch call makes); gather, statistics and the epoch kernels run over n_epochs x T x n_envs rows.
static int ppo_epochs_impl(const ia_policy_desc* d, float* params, float* params_t, float* norm_mean, float* norm_var,
                 int32_t* norm_count, int update_norm, const float* obs, const float* actions, const float* old_logp,
                 const float* advantages, const float* returns, const int64_t* perm, int n_epochs, int T, int n_envs,
                 int batch_size, int normalize_adv, float clip_range, float ent_coef, float vf_coef,
                 float max_grad_norm, float* exp_avg, float* exp_avg_sq, double lr, double beta1, double beta2,
                 float adam_eps, int64_t adam_steps_done, float* ws, float* stats, void* stream) {
  if (!pol_ok(d) || batch_size <= 0 || n_epochs < 1) return IA_ERR_ARG;
  if (n_epochs > 1 && ((long long)T * n_envs) % batch_size != 0) return IA_ERR_UNSUPPORTED;
  PpoArgs a{d, params, params_t, norm_mean, norm_var, norm_count, update_norm, obs, actions, old_logp, advantages,
            returns, T, n_envs, normalize_adv, clip_range, ent_coef, vf_coef, max_grad_norm, exp_avg, exp_avg_sq,
            (float)beta1, (float)beta2, adam_eps, ws, (hipStream_t)stream};
  const long long total = (long long)T * n_envs * n_epochs;
  int64_t step = adam_steps_done;
  int mb = 0;
  auto size_at = [&](long long start) { return (int)((total - start) < batch_size ? (total - start) : batch_size); };
  const Gathered g = gathered_region(d, ws, size_at(0), total);
  // which epoch kernel (64-wide towers; see below): the word-exchange form has its areas and control words cleared by
  // the gather launch
  static int dev_cus = 0;
  if (dev_cus == 0) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&dev_cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
      return IA_ERR_ARG;
  }
  constexpr size_t EPOCH_SPLIT_LDS = 160 * 1024 - 1024;   // dynamic LDS the one-tower kernel may ask for (it has static LDS too)
  constexpr size_t EPOCH_LL_LDS = 160 * 1024 - 2048;
  const bool one_launch = d->hidden == 64 && !g_epoch_split && g_tstamp == nullptr;
  const int nrb = cdiv(size_at(0), ROWS);
  const PolOff po = pol_offsets(d->obs_dim, d->act_dim, d->hidden, d->discrete);
  const int P = po.total;
  // one tower per workgroup (two workgroups of four waves per row block, the tower's parameters resident in LDS) when
  // both fit: 2 nrb workgroups co-resident, chunks of <= 4 x 256 parameters, the larger tower's images beside its tiles
  const int tlen = 64 * d->obs_dim + 64 + 64 * 64 + 64, lenB = std::max(po.cW - po.aW, P - po.cW) + 3;
  const size_t sbytes = (size_t)(((GLds<64, 1>::total + 3) & ~3) + 2 * tlen + 4 * d->obs_dim + 128 + lenB + MAXA) * sizeof(float);
  const bool towers_fit = one_launch && !g_epoch_whole && 2 * nrb <= dev_cus && 2 * nrb <= 64;
  // the word-exchange form of the one-tower kernel (no grid barriers); its word areas sit behind everything else in ws.
  // Small minibatches (down to ONE row block: SB3's default batch_size = 64): the launch has as many workgroups as it takes
  // for chunks of <= 1 024 parameters -- those beyond the row blocks run no gradient phase, they sum, clip and step their
  // chunk (`epoch_ll_row_blocks`)
  const int nrb_l = epoch_ll_row_blocks(nrb, P);
  const bool llx = towers_fit && !g_epoch_barriers && sbytes + 16 <= EPOCH_LL_LDS && nrb_l <= 32 && 2 * nrb_l <= dev_cus;
  const bool split = towers_fit && sbytes <= EPOCH_SPLIT_LDS && (llx || cdiv(P, 2 * nrb) <= 4 * 256);
  // round 6: the transposed-chain kernel on 32-ROW blocks (twice the workgroups, half the chain and half the weight-gradient
  // tiles per compute unit) where its 2 x ceil(b / 32) workgroups and slabs fit the exchange's limits (minibatches <= 1 024 rows)
  const int nrb32_l = epoch_ll_row_blocks(cdiv(size_at(0), 32), P);
  const bool rows32 = llx && g_epoch_chain2 && g_epoch_rows32 && d->obs_dim <= 32 && nrb32_l <= 32 && 2 * nrb32_l <= dev_cus &&
                      2 * nrb32_l <= 64;
  const int nrb_x = rows32 ? nrb32_l : nrb_l;   // row blocks the word areas are laid out for
  unsigned long long* ll_base = llx ? reinterpret_cast<unsigned long long*>(ws + ppo_ws_plain_floats(d, size_at(0), total)) : nullptr;
  int rc = launch_gather(a, perm, total, g, ll_base, llx ? epoch_ll_words(nrb_x, P) : 0,
                         llx ? reinterpret_cast<unsigned*>(ws) + 4 : nullptr);   // (words 4, 5, 6: counter, error word, ticket)
  if (rc) return rc;
  // statistics of every minibatch of the epoch, one launch ahead of the chain (ppo_epoch_stats_kernel)
  const int aw = d->discrete ? 1 : d->act_dim;
  const int n_mb = (int)((total + batch_size - 1) / batch_size);
  float* seq = g.ret + total;                       // [n_mb][EPS_SEQ]
  float* part = seq + (long long)n_mb * EPS_SEQ;    // [n_mb][EPS_PART]
  (void)aw;
  {
    static bool attr = false;
    const size_t bytes = PREP_LDS_FLOATS * sizeof(float);
    if (!attr) { rc = set_lds(ppo_epoch_stats_kernel, bytes); if (rc) return rc; attr = true; }
    if (!llx && hipMemsetAsync(ws + 6, 0, sizeof(unsigned), a.st) != hipSuccess) return IA_ERR_ARG;   // its ticket (ws is not pre-zeroed)
    hipLaunchKernelGGL(ppo_epoch_stats_kernel, dim3(n_mb), dim3(PREP_THREADS), bytes, a.st, *d, g.obs, g.adv, total,
                       batch_size, update_norm, norm_mean, norm_var, norm_count, seq, part,
                       reinterpret_cast<unsigned*>(ws) + 6);
    IA_CHECK_LAUNCH();
  }
  const bool snap = d->has_norm && update_norm;
  if (one_launch) {
    // one launch per (<= 64 minibatches of the) epoch when every gradient workgroup can be resident at once
    const int nwg = split ? 2 * nrb : nrb;
    if (llx || (nwg <= dev_cus && nwg <= 64 && cdiv(P, nwg) <= 4 * (split ? 256 : 512))) {
      static bool attr = false, attr_s = false;
      const size_t mbytes = split ? sbytes : GLds<64>::total * sizeof(float);
      const int aw_ = d->discrete ? 1 : d->act_dim;
      const size_t l2bytes = (size_t)(d->obs_dim <= 16 ? T64Geo<1>(d->obs_dim, aw_).total : T64Geo<2>(d->obs_dim, aw_).total) *
                                 sizeof(float) + 16;
      if (llx && g_epoch_chain2 && d->obs_dim <= 32 && l2bytes <= EPOCH_LL_LDS) {
        // round 5's gradient phase (transposed register-resident chain, images with ds_read_b128 fragments)
        auto k1h = ppo_epoch_ll2_kernel<1, 8, 64>;   // 64-row blocks: eight waves (feature halves), two per SIMD
        auto k2h = ppo_epoch_ll2_kernel<2, 8, 64>;
        auto k1r = ppo_epoch_ll2_kernel<1, 4, 32>;   // round 6: 32-row blocks, feature halves
        auto k2r = ppo_epoch_ll2_kernel<2, 4, 32>;
        auto k1q = ppo_epoch_ll2_kernel<1, 8, 32>;   // round 6: 32-row blocks, feature quarters (two waves per SIMD)
        auto k2q = ppo_epoch_ll2_kernel<2, 8, 32>;
        const int nw = rows32 ? (g_epoch_quarters ? 8 : 4) : 8;
        const int ki = (d->obs_dim <= 16 ? 0 : 1) + (rows32 ? (nw == 8 ? 4 : 2) : 0);
        decltype(k1h) kerns[6] = {k1h, k2h, k1r, k2r, k1q, k2q};
        auto kern = kerns[ki];
        static bool attr_2[6] = {false, false, false, false, false, false};
        if (!attr_2[ki]) { rc = set_lds(kern, EPOCH_LL_LDS); if (rc) return rc; attr_2[ki] = true; }
        EpochLl el{};
        el.base = ll_base;   // (cleared, like the error word, by the gather launch above)
        for (int first = 0; first < n_mb; first += EpochSteps::MAX) {
          EpochSteps es{};
          es.first = first;
          es.n = std::min(EpochSteps::MAX, n_mb - first);
          el.seq0 = (unsigned)(adam_steps_done + first + 1);
          for (int k = 0; k < es.n; ++k) {
            ++step;
            es.step_size[k] = (float)(lr / (1.0 - pow(beta1, (double)step)));
            es.bc2_sqrt[k] = (float)sqrt(1.0 - pow(beta2, (double)step));
          }
          hipLaunchKernelGGL(kern, dim3(2 * nrb_x), dim3(64 * nw), l2bytes, a.st, *d, params, params_t, exp_avg, exp_avg_sq, norm_mean,
                             norm_var, g.obs, g.act, g.logp, g.adv, g.ret, total, batch_size, T, n_envs, normalize_adv,
                             clip_range, ent_coef, vf_coef, max_grad_norm, (float)beta1, (float)beta2, adam_eps, ws, el, seq,
                             snap ? 1 : 0, stats, es, g_epoch_dbg);
          IA_CHECK_LAUNCH();
        }
        return IA_OK;
      }
      if (llx) {
        const int nll = tlen <= 20 * 256 ? 20 : (tlen <= 24 * 256 ? 24 : 33);
        static bool attr_l[3] = {false, false, false};
        auto k20 = ppo_epoch_ll_kernel<64, 20>;
        auto k24 = ppo_epoch_ll_kernel<64, 24>;
        auto k33 = ppo_epoch_ll_kernel<64, 33>;
        auto kern = nll == 20 ? k20 : (nll == 24 ? k24 : k33);
        const int ai = nll == 20 ? 0 : (nll == 24 ? 1 : 2);
        if (!attr_l[ai]) { rc = set_lds(kern, EPOCH_LL_LDS); if (rc) return rc; attr_l[ai] = true; }
        EpochLl el{};
        el.base = ll_base;   // (cleared, like the error word, by the gather launch above)
        for (int first = 0; first < n_mb; first += EpochSteps::MAX) {
          EpochSteps es{};
          es.first = first;
          es.n = std::min(EpochSteps::MAX, n_mb - first);
          el.seq0 = (unsigned)(adam_steps_done + first + 1);
          for (int k = 0; k < es.n; ++k) {
            ++step;
            es.step_size[k] = (float)(lr / (1.0 - pow(beta1, (double)step)));
            es.bc2_sqrt[k] = (float)sqrt(1.0 - pow(beta2, (double)step));
          }
          hipLaunchKernelGGL(kern, dim3(2 * nrb_l), dim3(256), sbytes + 16, a.st, *d, params, params_t, exp_avg, exp_avg_sq, norm_mean,
                             norm_var, g.obs, g.act, g.logp, g.adv, g.ret, total, batch_size, T, n_envs, normalize_adv,
                             clip_range, ent_coef, vf_coef, max_grad_norm, (float)beta1, (float)beta2, adam_eps, ws, el, seq,
                             snap ? 1 : 0, stats, es, g_epoch_dbg);
          IA_CHECK_LAUNCH();
        }
        return IA_OK;
      }
      if (!split && !attr) { rc = set_lds(ppo_epoch_persistent_kernel<64>, mbytes); if (rc) return rc; attr = true; }
      if (split && !attr_s) {
        rc = set_lds(ppo_epoch_persistent_kernel<64, true>, EPOCH_SPLIT_LDS);
        if (rc) return rc;
        attr_s = true;
      }
      if (hipMemsetAsync(ws + 4, 0, 2 * sizeof(unsigned), a.st) != hipSuccess) return IA_ERR_ARG;   // barrier counter, error word
      for (int first = 0; first < n_mb; first += EpochSteps::MAX) {
        EpochSteps es{};
        es.first = first;
        es.n = std::min(EpochSteps::MAX, n_mb - first);
        for (int k = 0; k < es.n; ++k) {
          ++step;
          es.step_size[k] = (float)(lr / (1.0 - pow(beta1, (double)step)));
          es.bc2_sqrt[k] = (float)sqrt(1.0 - pow(beta2, (double)step));
        }
        if (first > 0 && hipMemsetAsync(ws + 4, 0, sizeof(unsigned), a.st) != hipSuccess) return IA_ERR_ARG;
        if (split)
          hipLaunchKernelGGL((ppo_epoch_persistent_kernel<64, true>), dim3(nwg), dim3(256), mbytes, a.st, *d, params,
                             params_t, exp_avg, exp_avg_sq, norm_mean, norm_var, g.obs, g.act, g.logp, g.adv, g.ret, total,
                             batch_size, T, n_envs, normalize_adv, clip_range, ent_coef, vf_coef, max_grad_norm,
                             (float)beta1, (float)beta2, adam_eps, ws, seq, snap ? 1 : 0, stats, es, g_epoch_dbg);
        else
          hipLaunchKernelGGL(ppo_epoch_persistent_kernel<64>, dim3(nwg), dim3(512), mbytes, a.st, *d, params, params_t,
                             exp_avg, exp_avg_sq, norm_mean, norm_var, g.obs, g.act, g.logp, g.adv, g.ret, total, batch_size,
                             T, n_envs, normalize_adv, clip_range, ent_coef, vf_coef, max_grad_norm, (float)beta1,
                             (float)beta2, adam_eps, ws, seq, snap ? 1 : 0, stats, es, g_epoch_dbg);
        IA_CHECK_LAUNCH();
      }
      return IA_OK;
    }
  }
  for (long long start = 0; start < total; start += batch_size, ++mb) {
    const int b = size_at(start);
    ++step;
    const double bc1 = 1.0 - pow(beta1, (double)step);
    const double bc2 = 1.0 - pow(beta2, (double)step);
    PpoArgs v = at_rows(a, g, start);
    float* sq = seq + (long long)mb * EPS_SEQ;
    v.advstat = sq;
    if (snap) { v.norm_mean = sq + 8; v.norm_var = sq + 8 + MAXD; }
    rc = d->hidden == 32 ? launch_grad<32>(v, nullptr, b) : launch_grad<64>(v, nullptr, b);
    if (rc) return rc;
    rc = launch_apply_split(v, b, (float)(lr / bc1), (float)sqrt(bc2), stats ? stats + mb * 8 : nullptr, nullptr, nullptr,
                            nullptr, 0, 0);
    if (rc) return rc;
  }
  return IA_OK;
}


int ia_ppo_epoch(const ia_policy_desc* d, float* params, float* params_t, float* norm_mean, float* norm_var,
                 int32_t* norm_count, int update_norm, const float* obs, const float* actions, const float* old_logp,
                 const float* advantages, const float* returns, const int64_t* perm, int T, int n_envs,
                 int batch_size, int normalize_adv, float clip_range, float ent_coef, float vf_coef,
                 float max_grad_norm, float* exp_avg, float* exp_avg_sq, double lr, double beta1, double beta2,
                 float adam_eps, int64_t adam_steps_done, float* ws, float* stats, void* stream) {
  return ppo_epochs_impl(d, params, params_t, norm_mean, norm_var, norm_count, update_norm, obs, actions, old_logp,
                         advantages, returns, perm, 1, T, n_envs, batch_size, normalize_adv, clip_range, ent_coef, vf_coef,
                         max_grad_norm, exp_avg, exp_avg_sq, lr, beta1, beta2, adam_eps, adam_steps_done, ws, stats, stream);
}

int ia_ppo_epochs(const ia_policy_desc* d, float* params, float* params_t, float* norm_mean, float* norm_var,
                  int32_t* norm_count, int update_norm, const float* obs, const float* actions, const float* old_logp,
                  const float* advantages, const float* returns, const int64_t* perms, int n_epochs, int T, int n_envs,
                  int batch_size, int normalize_adv, float clip_range, float ent_coef, float vf_coef,
                  float max_grad_norm, float* exp_avg, float* exp_avg_sq, double lr, double beta1, double beta2,
                  float adam_eps, int64_t adam_steps_done, float* ws, float* stats, void* stream) {
  return ppo_epochs_impl(d, params, params_t, norm_mean, norm_var, norm_count, update_norm, obs, actions, old_logp,
                         advantages, returns, perms, n_epochs, T, n_envs, batch_size, normalize_adv, clip_range, ent_coef,
                         vf_coef, max_grad_norm, exp_avg, exp_avg_sq, lr, beta1, beta2, adam_eps, adam_steps_done, ws, stats,
                         stream);
}

// LDS of a gradient block: minibatch tiles, both parameter copies, the staged next minibatch, the transpose map
inline size_t upd_grad_lds_bytes(int P4, int aw) {
  return 16 /* base rounded up to 16 bytes */ + (CLds::total + (size_t)P4 + upd_pt4(P4) + UpdStage::total(aw)) * sizeof(float) +
         P4 * sizeof(unsigned short);
}

// Workspace of ia_ppo_update in floats; 0 when the persistent kernel does not cover the shape
// (the caller then runs ia_ppo_epoch per epoch).
static int64_t upd_ws_floats(const ia_policy_desc* d, int batch_size, int world) {
  if (!pol_ok(d) || batch_size <= 0 || world < 1 || world > SHARD_WORLD_MAX) return IA_ERR_ARG;
  const int P = pol_offsets(d->obs_dim, d->act_dim, d->hidden, d->discrete).total;
  const int nblk = cdiv(batch_size, ROWS);
  const int P4 = (P + 3) & ~3;
  if (d->hidden != 32 || P > UPD_NPT_WIDE * 512 || upd_grad_lds_bytes(P4, d->discrete ? 1 : d->act_dim) > 160 * 1024 || nblk > UPD_NBLK_MAX ||
      cdiv((long long)batch_size * world, UPD_SLICE) > UPD_SLICES_MAX)
    return 0;
  // (every gradient workgroup polls nblk x ceil((P4 + 8) / nblk) slab words with its parameters-per-thread x 512 lanes)
  if (nblk > 1 && P4 + 8 + nblk - 1 > UPD_NPT_WIDE * 512) return 0;
  return UPD_CTRL + 2 * UPD_MAX_STEPS + UPD_RING * UPD_RS + UPD_SD * 2 + UPD_SD * (int64_t)nblk * 8 +
         4 * (int64_t)nblk * (P4 + 8) + 4 * (int64_t)(P4 + 8) + (int64_t)UPD_RING * UPD_SLICES_MAX * UPD_PRS;
}
int64_t ia_ppo_update_ws_floats(const ia_policy_desc* d, int batch_size) { return upd_ws_floats(d, batch_size, 1); }
// Row-sharded data-parallel update (ia_ppo_update_sharded): workspace for `rows_per_rank` rows of each global minibatch of
// world x rows_per_rank rows (0: shape not covered), and the size of a rank's receive area.
int64_t ia_ppo_update_sharded_ws_floats(const ia_policy_desc* d, int rows_per_rank, int world) {
  return upd_ws_floats(d, rows_per_rank, world);
}
int64_t ia_ppo_shard_recv_bytes(const ia_policy_desc* d, int world) {
  if (!pol_ok(d) || world < 1 || world > SHARD_WORLD_MAX) return IA_ERR_ARG;
  const int P = pol_offsets(d->obs_dim, d->act_dim, d->hidden, d->discrete).total;
  return 2 * (int64_t)world * (((P + 3) & ~3) + 8) * (int64_t)sizeof(unsigned long long);
}

bool g_upd_xcd_pack = false;   // opt-in (ia_ppo_update_xcd_pack): several gradient workgroups that fit one XCD packed there
constexpr int HOST_TAB_SLOTS = 8;
struct HostTab { float* buf = nullptr; hipEvent_t done; };
HostTab g_host_tab[HOST_TAB_SLOTS];
int g_host_tab_next = 0;
int ia_ppo_update_xcd_pack(int on) { g_upd_xcd_pack = on != 0; return IA_OK; }
int g_upd_assume_cus = 0;
int ia_ppo_update_assume_cus(int n) { g_upd_assume_cus = n; return IA_OK; }

// A whole PPO.train: n_epochs passes over consecutive minibatches of perm[e][T*n_envs] (SB3
// RolloutBuffer.get order), in ONE persistent launch per <= UPD_MAX_STEPS optimiser steps.
// stats: [n_epochs * n_minibatches][8] or NULL. ws must be zero-initialised once by the caller;
// word 8 of it is a sticky error flag (non-zero: a grid wait timed out, results invalid).
// `shard` (ia_ppo_update_sharded): `batch_size` is the rank's rows per minibatch, `n_envs` the global tile's width; the
// schedule (minibatch size world x batch_size, statistics slices) is the global one, the gradient workgroups are the rank's.
static int ppo_update_launch(const ia_policy_desc* d, float* params, float* params_t, float* norm_mean, float* norm_var,
                  int32_t* norm_count, int update_norm, const float* obs, const float* actions, const float* old_logp,
                  const float* advantages, const float* returns, const int64_t* perm, int n_epochs, int T, int n_envs,
                  int batch_size, int normalize_adv, float clip_range, float ent_coef, float vf_coef,
                  float max_grad_norm, float* exp_avg, float* exp_avg_sq, double lr, double beta1, double beta2,
                  float adam_eps, int64_t adam_steps_done, float* ws, float* stats, void* stream, const ShardArgs* shard) {
  const int world = shard ? shard->world : 1;
  if (!pol_ok(d) || batch_size <= 0 || n_epochs <= 0 || upd_ws_floats(d, batch_size, world) <= 0) return IA_ERR_ARG;
  const long long total = (long long)T * n_envs;
  const int batch_global = batch_size * world;
  const int n_mb = cdiv(total, batch_global);
  const int nblk = cdiv(batch_size, ROWS);
  const int P = pol_offsets(d->obs_dim, d->act_dim, 32, d->discrete).total;
  const int P4 = (P + 3) & ~3;
  const size_t grad_bytes = upd_grad_lds_bytes(P4, d->discrete ? 1 : d->act_dim);
  const size_t prep_bytes = (PREP_LDS_FLOATS + (size_t)cdiv(batch_global, ROWS) * ROWS) * sizeof(float);  // + row offsets
  const size_t bytes = grad_bytes > prep_bytes ? grad_bytes : prep_bytes;
  const bool wide = P > UPD_NPT * 512 || (nblk > 1 && P4 + 8 + nblk - 1 > UPD_NPT * 512);
  const bool timing = g_tstamp != nullptr;
  // instantiations: {8, 9} parameters per thread x {production, phase clocks} x first-layer fragments for <= 32 / <= 64
  // observation columns (the narrow form keeps 16 registers less across the chain)
  const bool ks16 = d->obs_dim > 32;
  // one gradient workgroup whose parameter vector fits the consumed part of the staging area: the gradient stays in LDS
  const bool local = nblk == 1 && P4 <= UpdStage::nxt;
  using KernelT = decltype(&ppo_update_persistent_kernel<UPD_NPT, false, 8, false>);
  static const KernelT kernels[32] = {
      ppo_update_persistent_kernel<UPD_NPT, false, 8, false>,      ppo_update_persistent_kernel<UPD_NPT, false, 16, false>,
      ppo_update_persistent_kernel<UPD_NPT, true, 8, false>,       ppo_update_persistent_kernel<UPD_NPT, true, 16, false>,
      ppo_update_persistent_kernel<UPD_NPT_WIDE, false, 8, false>, ppo_update_persistent_kernel<UPD_NPT_WIDE, false, 16, false>,
      ppo_update_persistent_kernel<UPD_NPT_WIDE, true, 8, false>,  ppo_update_persistent_kernel<UPD_NPT_WIDE, true, 16, false>,
      ppo_update_persistent_kernel<UPD_NPT, false, 8, true>,       ppo_update_persistent_kernel<UPD_NPT, false, 16, true>,
      ppo_update_persistent_kernel<UPD_NPT, true, 8, true>,        ppo_update_persistent_kernel<UPD_NPT, true, 16, true>,
      ppo_update_persistent_kernel<UPD_NPT_WIDE, false, 8, true>,  ppo_update_persistent_kernel<UPD_NPT_WIDE, false, 16, true>,
      ppo_update_persistent_kernel<UPD_NPT_WIDE, true, 8, true>,   ppo_update_persistent_kernel<UPD_NPT_WIDE, true, 16, true>,
      // row-sharded data parallelism (no phase-clock build): [16 + local * 4 + wide * 2 + ks16]
      ppo_update_persistent_kernel<UPD_NPT, false, 8, false, true>,      ppo_update_persistent_kernel<UPD_NPT, false, 16, false, true>,
      ppo_update_persistent_kernel<UPD_NPT_WIDE, false, 8, false, true>, ppo_update_persistent_kernel<UPD_NPT_WIDE, false, 16, false, true>,
      ppo_update_persistent_kernel<UPD_NPT, false, 8, true, true>,       ppo_update_persistent_kernel<UPD_NPT, false, 16, true, true>,
      ppo_update_persistent_kernel<UPD_NPT_WIDE, false, 8, true, true>,  ppo_update_persistent_kernel<UPD_NPT_WIDE, false, 16, true, true>,
      // one gradient workgroup, minibatches of <= 16 rows (the tuned AIRL file): six of the eight waves skip the layer chain,
      // each tower's one wave has a SIMD to itself: [24 + shard * 4 + wide * 2 + ks16]
      ppo_update_persistent_kernel<UPD_NPT, false, 8, true, false, true>,       ppo_update_persistent_kernel<UPD_NPT, false, 16, true, false, true>,
      ppo_update_persistent_kernel<UPD_NPT_WIDE, false, 8, true, false, true>,  ppo_update_persistent_kernel<UPD_NPT_WIDE, false, 16, true, false, true>,
      ppo_update_persistent_kernel<UPD_NPT, false, 8, true, true, true>,        ppo_update_persistent_kernel<UPD_NPT, false, 16, true, true, true>,
      ppo_update_persistent_kernel<UPD_NPT_WIDE, false, 8, true, true, true>,   ppo_update_persistent_kernel<UPD_NPT_WIDE, false, 16, true, true, true>};
  const bool small = local && batch_size <= 16 && !timing;
  const int vi_k = small ? 24 + (shard ? 4 : 0) + wide * 2 + ks16
                         : (shard ? 16 + local * 4 + wide * 2 + ks16 : local * 8 + wide * 4 + timing * 2 + ks16);
  const KernelT kernel = kernels[vi_k];
  static size_t attr_bytes[32] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  if (bytes > attr_bytes[vi_k]) {
    const int rc = set_lds(kernel, bytes);
    if (rc) return rc;
    attr_bytes[vi_k] = bytes;
  }
  hipStream_t st = (hipStream_t)stream;
  const int steps_total = n_epochs * n_mb;
  int64_t step = adam_steps_done;
  for (int first = 0; first < steps_total; first += UPD_MAX_STEPS) {
    UpdSched sch;
    sch.n_steps = steps_total - first < UPD_MAX_STEPS ? steps_total - first : UPD_MAX_STEPS;
    sch.first = first; sch.n_mb = n_mb; sch.batch_size = batch_global; sch.total = total;
    // Adam's per-step scalars, formed in double on the host exactly as torch.optim.Adam does, staged
    // through a small ring of pinned buffers (a slot is reused only after its copy has completed).
    HostTab& ht = g_host_tab[g_host_tab_next];
    g_host_tab_next = (g_host_tab_next + 1) % HOST_TAB_SLOTS;
    if (ht.buf == nullptr) {
      hipError_t e0 = hipHostMalloc(reinterpret_cast<void**>(&ht.buf), 2 * UPD_MAX_STEPS * sizeof(float), 0);
      if (e0 != hipSuccess) return (int)e0;
      e0 = hipEventCreateWithFlags(&ht.done, hipEventDisableTiming);
      if (e0 != hipSuccess) return (int)e0;
    } else {
      hipError_t e0 = hipEventSynchronize(ht.done);
      if (e0 != hipSuccess) return (int)e0;
    }
    for (int s = 0; s < sch.n_steps; ++s) {
      ++step;
      const double bc1 = 1.0 - pow(beta1, (double)step);
      const double bc2 = 1.0 - pow(beta2, (double)step);
      ht.buf[s] = (float)(lr / bc1);
      ht.buf[UPD_MAX_STEPS + s] = (float)sqrt(bc2);
    }
    const UpdWs uw = upd_ws(ws, nblk, P);
    hipError_t ec = hipMemcpyAsync(uw.tab, ht.buf, 2 * UPD_MAX_STEPS * sizeof(float), hipMemcpyHostToDevice, st);
    if (ec != hipSuccess) return (int)ec;
    ec = hipEventRecord(ht.done, st);
    if (ec != hipSuccess) return (int)ec;
    hipError_t e = hipMemsetAsync(ws, 0, 8 * sizeof(unsigned), st);  // published / second-level arrivals (error word stays)
    if (e != hipSuccess) return (int)e;
    e = hipMemsetAsync(ws + 16, 0, (UPD_CTRL - 16) * sizeof(unsigned), st);  // per-slicer progress, arrival counters
    if (e != hipSuccess) return (int)e;
    // packing onto one XCD only works while all workgroups fit its 32 CUs (each takes a whole CU's LDS)
    // minibatches beyond UPD_SLICE rows: their statistics are cut into UPD_SLICE-row slices, one extra block
    // each, merged in order by the statistics block. (One block needs 22 us for the gathered moments of a
    // 1024-row minibatch -- as long as a whole gradient step, so the chain kept waiting 1-2 us per step for
    // it; two slices + merge take ~14 us and the ring runs ahead again.)
    const int n_slices = (batch_global > UPD_SLICE && total < (1ll << 31)) ? cdiv(batch_global, UPD_SLICE) : 0;
    const bool pack = g_upd_xcd_pack && nblk > 1 && nblk + 1 + n_slices <= 32;
    const int grid = (nblk + 1 + n_slices) * (pack ? 8 : 1);
    {
      // The grid barriers inside need every workgroup resident at once. A plain launch performs no such check
      // (and a cooperative launch costs +15-19 us per launch, MI355X_MICROARCH.md "coop-launch"), so the same
      // test is made here: workgroups per CU by the occupancy query (LDS-bound: one) times the CU count.
      // Not enough room -> IA_ERR_UNSUPPORTED, the caller runs ia_ppo_epoch (two launches per minibatch).
      static int dev_cus = 0, per_cu[32] = {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1,
                                            -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1};
      static size_t per_cu_bytes[32] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
      const int vi = vi_k;
      if (dev_cus == 0) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&dev_cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
          return IA_ERR_ARG;
      }
      const int cu_count = g_upd_assume_cus > 0 ? g_upd_assume_cus : dev_cus;
      if (per_cu[vi] < 0 || per_cu_bytes[vi] != bytes) {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(kernel), 512, bytes) !=
            hipSuccess)
          return IA_ERR_ARG;
        per_cu[vi] = nb;
        per_cu_bytes[vi] = bytes;
      }
      if ((long long)per_cu[vi] * cu_count < (pack ? grid / 8 : grid)) return IA_ERR_UNSUPPORTED;
    }
    ShardArgs sa;
    if (shard) {
      sa = *shard;
      sa.seq_base = shard->seq_base + (unsigned)first;
    } else {
      memset(&sa, 0, sizeof(sa));
    }
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(512), bytes, st, *d, params, params_t, exp_avg,
                       exp_avg_sq, norm_mean, norm_var, norm_count, update_norm, obs, actions, old_logp, advantages,
                       returns, perm, T, n_envs, normalize_adv, clip_range, ent_coef, vf_coef, max_grad_norm,
                       (float)beta1, (float)beta2, adam_eps, ws, nblk, n_slices, stats, sch, pack ? 1 : 0, g_tstamp, sa);
    IA_CHECK_LAUNCH();
  }
  return IA_OK;
}

int ia_ppo_update(const ia_policy_desc* d, float* params, float* params_t, float* norm_mean, float* norm_var,
                  int32_t* norm_count, int update_norm, const float* obs, const float* actions, const float* old_logp,
                  const float* advantages, const float* returns, const int64_t* perm, int n_epochs, int T, int n_envs,
                  int batch_size, int normalize_adv, float clip_range, float ent_coef, float vf_coef,
                  float max_grad_norm, float* exp_avg, float* exp_avg_sq, double lr, double beta1, double beta2,
                  float adam_eps, int64_t adam_steps_done, float* ws, float* stats, void* stream) {
  return ppo_update_launch(d, params, params_t, norm_mean, norm_var, norm_count, update_norm, obs, actions, old_logp,
                           advantages, returns, perm, n_epochs, T, n_envs, batch_size, normalize_adv, clip_range, ent_coef,
                           vf_coef, max_grad_norm, exp_avg, exp_avg_sq, lr, beta1, beta2, adam_eps, adam_steps_done, ws,
                           stats, stream, nullptr);
}

// The same [SB3 PPO.train] with each GLOBAL minibatch's rows sharded over `world` ranks (one process per GPU): `obs` ...
// `returns` are the all-gathered rollout tile [T, n_envs] (n_envs = world x the rank's environments), `perm` the
// permutations every rank shares, `rows_per_rank` the rank's rows of each minibatch of world x rows_per_rank rows.
// `recv`: this rank's receive area (ia_ppo_shard_recv_bytes), zeroed once at allocation and never again; `peer_recv[r]`:
// rank r's area as mapped into this process (own included). `seq_base`: optimiser steps exchanged through these areas so
// far (the same on every rank; it must only ever grow). Every rank must launch the same sequence of updates. `loopback` (tools: the cost of a world-W step on one
// process): every peer area is this process's own and the rank writes its record once per source rank.
int ia_ppo_update_sharded(const ia_policy_desc* d, float* params, float* params_t, float* norm_mean, float* norm_var,
                          int32_t* norm_count, int update_norm, const float* obs, const float* actions,
                          const float* old_logp, const float* advantages, const float* returns, const int64_t* perm,
                          int n_epochs, int T, int n_envs, int rows_per_rank, int normalize_adv, float clip_range,
                          float ent_coef, float vf_coef, float max_grad_norm, float* exp_avg, float* exp_avg_sq, double lr,
                          double beta1, double beta2, float adam_eps, int64_t adam_steps_done, float* ws, float* stats,
                          int world, int rank, uint32_t seq_base, void* recv, void* const* peer_recv, int loopback,
                          double timeout_s, void* stream) {
  if (world < 1 || world > SHARD_WORLD_MAX || rank < 0 || rank >= world || !recv || !peer_recv) return IA_ERR_ARG;
  ShardArgs sa;
  sa.world = world; sa.rank = rank; sa.rows_per_rank = rows_per_rank; sa.loopback = loopback != 0;
  const int nblk = cdiv(rows_per_rank, ROWS);
  int J = nblk / world;
  sa.pieces = J < 1 ? 1 : (J > SHARD_PIECES_MAX ? SHARD_PIECES_MAX : J);
  sa.seq_base = seq_base;
  sa.timeout_ticks = (long long)(timeout_s * 1e8);
  sa.recv = static_cast<unsigned long long*>(recv);
  for (int r = 0; r < SHARD_WORLD_MAX; ++r) {
    sa.peer_recv[r] = r < world ? static_cast<unsigned long long*>(peer_recv[r]) : nullptr;
    if (r < world && !sa.peer_recv[r]) return IA_ERR_ARG;
  }
  return ppo_update_launch(d, params, params_t, norm_mean, norm_var, norm_count, update_norm, obs, actions, old_logp,
                           advantages, returns, perm, n_epochs, T, n_envs, rows_per_rank, normalize_adv, clip_range,
                           ent_coef, vf_coef, max_grad_norm, exp_avg, exp_avg_sq, lr, beta1, beta2, adam_eps,
                           adam_steps_done, ws, stats, stream, &sa);
}

}  // extern "C"
