// Actor-critic heads and the PPO loss for policy towers of ARBITRARY shape (any SB3 `net_arch`).
// The fused kernels of policy.hip cover the reference configs' two equal tanh towers [H, H], H in {32, 64}; every
// other `net_arch` (deeper / unequal / pi != vf towers) runs its towers on the generic fp32-MFMA stacks of mlp.hip
// (`ia_mlp_forward / _backward`) and takes from here what sits between them:
//   * the DiagGaussian head of a rollout step (sample with host noise, clip to the Box, log-prob)   ia_gauss_act
//   * log-prob / entropy of given actions (evaluate_actions, AIRL's log pi)                          ia_gauss_eval
//   * the minibatch advantage moments (mean, unbiased std; one block, fixed order)                  ia_adv_moments
//   * [SB3 PPO.train] loss of one minibatch: clipped surrogate + value MSE + entropy bonus -> the gradient w.r.t.
//     the head outputs (mean or logits), the values and log_std, plus the logged statistics         ia_ppo_head_loss
// All kernels are row-parallel streams over [rows, act_dim] tiles (HBM-bound, a few MB at most); reductions run
// in a fixed order (block partials, then one block), so results are run-to-run identical.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "common.h"

namespace {

constexpr int PG_THREADS = 256;
constexpr int PG_MAX_ACT = 64;          // width of the per-block partial rows below
constexpr int PG_STAT = 5;              // pg_loss, value_loss, entropy_loss, approx_kl, clip_fraction
constexpr float HALF_LOG_2PI = 0.91893853320467274178f;

inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// [torch Normal.log_prob] summed over the action dimension: -(a-mu)^2 / (2 var) - log_std - log(sqrt(2 pi))
__device__ __forceinline__ float gauss_logp_term(float a, float mu, float log_std) {
  const float sd = expf(log_std);
  const float d = a - mu;
  return -(d * d) / (2.0f * (sd * sd)) - log_std - HALF_LOG_2PI;
}

__global__ __launch_bounds__(PG_THREADS) void gauss_act_kernel(const float* __restrict__ mean, const float* __restrict__ log_std,
                                                               const float* __restrict__ noise, const float* __restrict__ low,
                                                               const float* __restrict__ high, int n, int A,
                                                               float* __restrict__ actions, float* __restrict__ clipped,
                                                               float* __restrict__ logp) {
  const int r = blockIdx.x * PG_THREADS + threadIdx.x;
  if (r >= n) return;
  float lp = 0.0f;
  for (int a = 0; a < A; ++a) {
    const float mu = mean[(size_t)r * A + a], ls = log_std[a];
    const float act = mu + expf(ls) * noise[(size_t)r * A + a];     // [SB3 DiagGaussianDistribution.sample] = rsample
    actions[(size_t)r * A + a] = act;
    if (clipped) clipped[(size_t)r * A + a] = fminf(fmaxf(act, low[a]), high[a]);
    lp += gauss_logp_term(act, mu, ls);
  }
  if (logp) logp[r] = lp;
}

__global__ __launch_bounds__(PG_THREADS) void gauss_eval_kernel(const float* __restrict__ mean, const float* __restrict__ log_std,
                                                                const float* __restrict__ actions, int n, int A,
                                                                float* __restrict__ logp, float* __restrict__ entropy) {
  const int r = blockIdx.x * PG_THREADS + threadIdx.x;
  if (r >= n) return;
  float lp = 0.0f, ent = 0.0f;
  for (int a = 0; a < A; ++a) {
    const float ls = log_std[a];
    lp += gauss_logp_term(actions[(size_t)r * A + a], mean[(size_t)r * A + a], ls);
    ent += 0.5f + HALF_LOG_2PI + ls;
  }
  if (logp) logp[r] = lp;
  if (entropy) entropy[r] = ent;
}

// block-wide sum in a fixed order: lanes -> wave (xor butterfly), waves -> thread 0 (ascending)
__device__ __forceinline__ float block_sum(float v, float* sm) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = 0.0f;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += sm[w];
  return t;
}

// out[0] = mean(x), out[1] = unbiased std(x)   ([SB3 ppo.py] `advantages.mean()`, `advantages.std()`)
__global__ __launch_bounds__(1024) void adv_moments_kernel(const float* __restrict__ x, int n, float* __restrict__ out) {
  __shared__ float sm[16];
  float s = 0.0f;
  for (int i = threadIdx.x; i < n; i += 1024) s += x[i];
  const float mean = block_sum(s, sm) / (float)n;
  float q = 0.0f;
  for (int i = threadIdx.x; i < n; i += 1024) {
    const float d = x[i] - mean;
    q += d * d;
  }
  const float ss = block_sum(q, sm);
  if (threadIdx.x == 0) {
    out[0] = mean;
    out[1] = sqrtf(ss / (float)(n - 1));
  }
}

// [torch clip_grad_norm_] on one flat gradient: total = ||g||_2, g *= min(1, max_norm / (total + 1e-6)); one block
__global__ __launch_bounds__(1024) void clip_grad_norm_kernel(float* __restrict__ g, long long n, float max_norm,
                                                              float* __restrict__ norm_out) {
  __shared__ float sm[16];
  float q = 0.0f;
  for (long long i = threadIdx.x; i < n; i += 1024) q += g[i] * g[i];
  const float total = sqrtf(block_sum(q, sm));
  const float coef = fminf(max_norm / (total + 1e-6f), 1.0f);
  for (long long i = threadIdx.x; i < n; i += 1024) g[i] *= coef;
  if (threadIdx.x == 0 && norm_out) norm_out[0] = total;
}

// The same for long gradients (NatureCNN: 1.7 M entries took the one block 700 us, 11 of the image-GAIL round's 93 ms of GPU
// time): two launches over a `ws` of CLIP_BLOCKS floats. (1) block b leaves the sum of squares of its chunk in ws[b];
// (2) EVERY block folds the partials in the same order (identical total and coefficient everywhere) and scales its chunk.
constexpr int CLIP_BLOCKS = 256;
__global__ __launch_bounds__(1024) void clip_sumsq_kernel(const float* __restrict__ g, long long n, long long chunk,
                                                          float* __restrict__ ws) {
  __shared__ float sm[16];
  const long long lo = (long long)blockIdx.x * chunk, hi = lo + chunk < n ? lo + chunk : n;
  float q[4] = {0.f, 0.f, 0.f, 0.f};
  long long i = lo + threadIdx.x;
  for (; i + 3 * 1024 < hi; i += 4 * 1024) {
    float v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = g[i + u * 1024];
#pragma unroll
    for (int u = 0; u < 4; ++u) q[u] += v[u] * v[u];
  }
  for (int u = 0; i < hi; i += 1024, ++u) q[u] += g[i] * g[i];
  const float s = block_sum((q[0] + q[1]) + (q[2] + q[3]), sm);
  if (threadIdx.x == 0) ws[blockIdx.x] = s;
}
__global__ __launch_bounds__(1024) void clip_scale_kernel(float* __restrict__ g, long long n, long long chunk, int nb,
                                                          float max_norm, const float* __restrict__ ws,
                                                          float* __restrict__ norm_out) {
  __shared__ float sm[16];
  const float part = (int)threadIdx.x < nb ? ws[threadIdx.x] : 0.f;
  const float total = sqrtf(block_sum(part, sm));
  const float coef = fminf(max_norm / (total + 1e-6f), 1.0f);
  const long long lo = (long long)blockIdx.x * chunk, hi = lo + chunk < n ? lo + chunk : n;
  if (coef < 1.0f)
    for (long long i = lo + threadIdx.x; i < hi; i += 1024) g[i] *= coef;
  if (blockIdx.x == 0 && threadIdx.x == 0 && norm_out) norm_out[0] = total;
}

struct HeadLoss {
  const float* out;       // [B, A] Gaussian means or Categorical logits
  const float* log_std;   // [A] (Box heads)
  const float* values;    // [B]
  const float* actions;   // [B, A] (Box) or [B] fp32 indices (Discrete)
  const float* old_logp;  // [B]
  const float* adv;       // [B]
  const float* ret;       // [B]
  const float* adv_ms;    // [2] minibatch (mean, std) or null = advantages used as they are
  int B, A, discrete;
  float clip, ent_coef, vf_coef;
  float* d_out;           // [B, A]
  float* d_values;        // [B]
  float* part;            // [blocks][PG_MAX_ACT + 8]: per-block sums of dlog_std rows and the statistics
};

// One thread per minibatch row; formulas are those of policy.hip's fused step (ppo_grad_kernel), which follow
// [SB3 ppo.py:train]: ratio = exp(logp - old), surrogate min(adv*ratio, adv*clamp(ratio)), torch's tie rule for the
// gradient of `min` (0.5 / 0.5), value loss mse(returns, values), entropy loss -mean(entropy).
__global__ __launch_bounds__(PG_THREADS) void ppo_head_loss_kernel(HeadLoss h) {
  __shared__ float sm[PG_THREADS / 64];
  const int r = blockIdx.x * PG_THREADS + threadIdx.x;
  const bool live = r < h.B;
  const int A = h.A;
  const float invB = 1.0f / (float)h.B;
  float st[PG_STAT] = {0.f, 0.f, 0.f, 0.f, 0.f};
  float dlogp = 0.0f;
  float lse = 0.0f, ent = 0.0f;
  if (live) {
    float advn = h.adv[r];
    if (h.adv_ms) advn = (advn - h.adv_ms[0]) / (h.adv_ms[1] + 1e-8f);
    float logp = 0.0f;
    if (h.discrete) {
      const float* l = h.out + (size_t)r * A;
      float mx = l[0];
      for (int a = 1; a < A; ++a) mx = fmaxf(mx, l[a]);
      float se = 0.0f;
      for (int a = 0; a < A; ++a) se += expf(l[a] - mx);
      lse = mx + logf(se);
      for (int a = 0; a < A; ++a) {
        const float lpa = l[a] - lse;
        ent -= expf(lpa) * lpa;
      }
      logp = l[(int)h.actions[r]] - lse;
    } else {
      for (int a = 0; a < A; ++a) {
        const float ls = h.log_std[a];
        logp += gauss_logp_term(h.actions[(size_t)r * A + a], h.out[(size_t)r * A + a], ls);
        ent += 0.5f + HALF_LOG_2PI + ls;
      }
    }
    const float log_ratio = logp - h.old_logp[r];
    const float ratio = expf(log_ratio);
    const float lo = 1.0f - h.clip, hi = 1.0f + h.clip;
    const float pl1 = advn * ratio, pl2 = advn * fminf(fmaxf(ratio, lo), hi);
    const float g1 = pl1 < pl2 ? 1.0f : (pl1 == pl2 ? 0.5f : 0.0f);
    const float g2 = pl2 < pl1 ? 1.0f : (pl1 == pl2 ? 0.5f : 0.0f);
    const float inrange = (ratio >= lo && ratio <= hi) ? 1.0f : 0.0f;
    dlogp = -invB * advn * (g1 + g2 * inrange) * ratio;
    const float v = h.values[r], dv = v - h.ret[r];
    h.d_values[r] = h.vf_coef * 2.0f * dv * invB;
    st[0] = -fminf(pl1, pl2);
    st[1] = dv * dv;
    st[2] = -ent;
    st[3] = (ratio - 1.0f) - log_ratio;
    st[4] = fabsf(ratio - 1.0f) > h.clip ? 1.0f : 0.0f;
  }
  float* part = h.part + (size_t)blockIdx.x * (PG_MAX_ACT + 8);
  if (h.discrete) {
    if (live) {
      const float* l = h.out + (size_t)r * A;
      const int act = (int)h.actions[r];
      for (int a = 0; a < A; ++a) {
        const float lpa = l[a] - lse, p = expf(lpa);
        // d(-H)/dl_a = p_a (log p_a + H)
        h.d_out[(size_t)r * A + a] = dlogp * ((a == act ? 1.0f : 0.0f) - p) + h.ent_coef * invB * p * (lpa + ent);
      }
    }
  } else {
    for (int a = 0; a < A; ++a) {     // uniform trip count: every thread takes part in the block sums
      float dls = 0.0f;
      if (live) {
        const float ls = h.log_std[a], sd = expf(ls), var = sd * sd;
        const float d = h.actions[(size_t)r * A + a] - h.out[(size_t)r * A + a];
        h.d_out[(size_t)r * A + a] = dlogp * d / var;
        dls = dlogp * (d * d / var - 1.0f) - h.ent_coef * invB;
      }
      const float t = block_sum(dls, sm);
      if (threadIdx.x == 0) part[a] = t;
    }
  }
#pragma unroll
  for (int k = 0; k < PG_STAT; ++k) {
    const float t = block_sum(st[k], sm);
    if (threadIdx.x == 0) part[PG_MAX_ACT + k] = t;
  }
}

// sums the block partials in ascending block order: dlog_std[A] (+=0: written) and stats[8]
__global__ __launch_bounds__(128) void ppo_head_finish_kernel(const float* __restrict__ part, int nblocks, int A, int discrete,
                                                              int B, float ent_coef, float vf_coef,
                                                              float* __restrict__ dlog_std, float* __restrict__ stats) {
  const int c = threadIdx.x;
  __shared__ float s[8];
  if (c < PG_MAX_ACT + PG_STAT) {
    const bool is_stat = c >= PG_MAX_ACT;
    if (is_stat || (!discrete && c < A)) {
      float t = 0.0f;
      for (int b = 0; b < nblocks; ++b) t += part[(size_t)b * (PG_MAX_ACT + 8) + c];
      if (is_stat) s[c - PG_MAX_ACT] = t / (float)B;
      else if (dlog_std) dlog_std[c] = t;
    }
  }
  __syncthreads();
  if (c == 0 && stats) {
    for (int k = 0; k < PG_STAT; ++k) stats[k] = s[k];
    stats[5] = s[0] + ent_coef * s[2] + vf_coef * s[1];
    stats[6] = 0.0f;
    stats[7] = 0.0f;
  }
}

}  // namespace

extern "C" {

int ia_gauss_act(const float* mean, const float* log_std, const float* noise, const float* low, const float* high, int n,
                 int A, float* actions, float* clipped, float* logp, void* stream) {
  if (!mean || !log_std || !noise || !actions || n <= 0 || A <= 0 || (clipped && (!low || !high))) return IA_ERR_ARG;
  hipLaunchKernelGGL(gauss_act_kernel, dim3(cdiv(n, PG_THREADS)), dim3(PG_THREADS), 0, (hipStream_t)stream, mean, log_std,
                     noise, low, high, n, A, actions, clipped, logp);
  IA_CHECK_LAUNCH();
  return IA_OK;
}

int ia_gauss_eval(const float* mean, const float* log_std, const float* actions, int n, int A, float* logp, float* entropy,
                  void* stream) {
  if (!mean || !log_std || !actions || n <= 0 || A <= 0) return IA_ERR_ARG;
  hipLaunchKernelGGL(gauss_eval_kernel, dim3(cdiv(n, PG_THREADS)), dim3(PG_THREADS), 0, (hipStream_t)stream, mean, log_std,
                     actions, n, A, logp, entropy);
  IA_CHECK_LAUNCH();
  return IA_OK;
}

int ia_adv_moments(const float* x, int n, float* out2, void* stream) {
  if (!x || !out2 || n < 2) return IA_ERR_ARG;
  hipLaunchKernelGGL(adv_moments_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, x, n, out2);
  IA_CHECK_LAUNCH();
  return IA_OK;
}

long long ia_clip_grad_norm_ws_floats(void) { return CLIP_BLOCKS; }

int ia_clip_grad_norm(float* grad, long long n, float max_norm, float* norm_out, float* ws, void* stream) {
  if (!grad || n <= 0) return IA_ERR_ARG;
  if (!ws || n < 65536) {   // short gradients (or no workspace): one block
    hipLaunchKernelGGL(clip_grad_norm_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, grad, n, max_norm, norm_out);
    IA_CHECK_LAUNCH();
    return IA_OK;
  }
  long long chunk = (n + CLIP_BLOCKS - 1) / CLIP_BLOCKS;
  chunk = (chunk + 4095) / 4096 * 4096;   // whole 4 x 1024-element trips
  const int nb = (int)((n + chunk - 1) / chunk);
  hipLaunchKernelGGL(clip_sumsq_kernel, dim3(nb), dim3(1024), 0, (hipStream_t)stream, grad, n, chunk, ws);
  IA_CHECK_LAUNCH();
  hipLaunchKernelGGL(clip_scale_kernel, dim3(nb), dim3(1024), 0, (hipStream_t)stream, grad, n, chunk, nb, max_norm, ws, norm_out);
  IA_CHECK_LAUNCH();
  return IA_OK;
}

long long ia_ppo_head_loss_ws_floats(int B) { return B <= 0 ? 0 : (long long)cdiv(B, PG_THREADS) * (PG_MAX_ACT + 8); }

int ia_ppo_head_loss(int discrete, const float* out, const float* log_std, const float* values, const float* actions,
                     const float* old_logp, const float* adv, const float* ret, const float* adv_ms, int B, int A,
                     float clip_range, float ent_coef, float vf_coef, float* d_out, float* d_values, float* dlog_std,
                     float* ws, float* stats, void* stream) {
  if (!out || !values || !actions || !old_logp || !adv || !ret || !d_out || !d_values || !ws || B <= 0 || A <= 0)
    return IA_ERR_ARG;
  if (!discrete && (!log_std || !dlog_std)) return IA_ERR_ARG;
  if (A > PG_MAX_ACT) return IA_ERR_UNSUPPORTED;
  HeadLoss h{out, log_std, values, actions, old_logp, adv, ret, adv_ms, B, A, discrete, clip_range, ent_coef, vf_coef,
             d_out, d_values, ws};
  const int nb = cdiv(B, PG_THREADS);
  hipLaunchKernelGGL(ppo_head_loss_kernel, dim3(nb), dim3(PG_THREADS), 0, (hipStream_t)stream, h);
  IA_CHECK_LAUNCH();
  hipLaunchKernelGGL(ppo_head_finish_kernel, dim3(1), dim3(128), 0, (hipStream_t)stream, ws, nb, A, discrete, B, ent_coef,
                     vf_coef, dlog_std, stats);
  IA_CHECK_LAUNCH();
  return IA_OK;
}

}  // extern "C"
