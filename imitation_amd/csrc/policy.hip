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
  for (int j = 0; j < H; ++j) a1row[j] = tanhf(acc[j]);
#pragma unroll
  for (int j = 0; j < H; ++j) acc[j] = b2[j];
  for (int k = 0; k < H; ++k) {
    const float ak = a1row[k];
#pragma unroll
    for (int j = 0; j < H; ++j) acc[j] = fmaf(W2t[k * H + j], ak, acc[j]);
  }
#pragma unroll
  for (int j = 0; j < H; ++j) {
    a2[j] = tanhf(acc[j]);
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
__global__ __launch_bounds__(ROWS) void policy_act_kernel(ia_policy_desc d, const float* __restrict__ P,
                                                          const float* __restrict__ Pt, const float* __restrict__ nm,
                                                          const float* __restrict__ nv, const float* __restrict__ obs,
                                                          int n, const float* __restrict__ noise,
                                                          const float* __restrict__ low, const float* __restrict__ high,
                                                          float* __restrict__ actions, float* __restrict__ clipped,
                                                          float* __restrict__ values, float* __restrict__ logp) {
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
  tower_forward<H>(Pt + o.pW1, P + o.pb1, Pt + o.pW2, P + o.pb2, D, xrow, a1row, nullptr, a2);
  head_forward<H>(P + o.aW, P + o.ab, A, a2, outrow);
  tower_forward<H>(Pt + o.vW1, P + o.vb1, Pt + o.vW2, P + o.vb2, D, xrow, a1row, nullptr, a2);
  float v = P[o.cb];
#pragma unroll
  for (int k = 0; k < H; ++k) v = fmaf(P[o.cW + k], a2[k], v);
  if (!valid) return;
  values[row] = v;
  if (!d.discrete) {
    float lp = 0.f;
    for (int a = 0; a < A; ++a) {
      const float ls = P[o.log_std + a];
      const float mu = outrow[a];
      // Normal.rsample: loc + eps * scale  (two roundings, as on the host)
      const float act = __fadd_rn(mu, __fmul_rn(noise[(long long)row * A + a], expf(ls)));
      actions[(long long)row * A + a] = act;
      clipped[(long long)row * A + a] = fminf(fmaxf(act, low[a]), high[a]);
      lp += gauss_logp_term(act, mu, ls);
    }
    logp[row] = lp;
  } else {
    float mx = outrow[0];
    for (int a = 1; a < A; ++a) mx = fmaxf(mx, outrow[a]);
    float se = 0.f;
    for (int a = 0; a < A; ++a) se += expf(outrow[a] - mx);
    const float lse = mx + logf(se);
    const float u = noise[row];
    float c = 0.f;
    int pick = A - 1;
    for (int a = 0; a < A; ++a) {
      c += expf(outrow[a] - lse);
      if (u < c) { pick = a; break; }
    }
    actions[row] = (float)pick;
    clipped[row] = (float)pick;
    logp[row] = outrow[pick] - lse;
  }
}

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
  // order NumPy evaluates `r + gamma*nv*nnt - v` and `delta + gamma*lambda*nnt*last`.
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  float last = 0.f;
  for (int t = T - 1; t >= 0; --t) {
    float nnt, nv;
    if (t == T - 1) {
      nnt = __fsub_rn(1.0f, last_dones[e]);
      nv = last_values[e];
    } else {
      nnt = __fsub_rn(1.0f, starts[(long long)(t + 1) * n + e]);
      nv = values[(long long)(t + 1) * n + e];
    }
    const float v = values[(long long)t * n + e];
    const float delta =
        __fsub_rn(__fadd_rn(rewards[(long long)t * n + e], __fmul_rn(__fmul_rn(gamma, nv), nnt)), v);
    last = __fadd_rn(delta, __fmul_rn(__fmul_rn(gl, nnt), last));
    adv[(long long)t * n + e] = last;
    ret[(long long)t * n + e] = __fadd_rn(last, v);
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

__global__ __launch_bounds__(256) void ppo_prepare_kernel(ia_policy_desc d, const float* __restrict__ obs,
                                                          const float* __restrict__ adv,
                                                          const int64_t* __restrict__ idx, int batch, int T,
                                                          int n_envs, int update_norm, float* __restrict__ nm,
                                                          float* __restrict__ nv, int32_t* __restrict__ ncount,
                                                          float* __restrict__ advstat) {
  __shared__ float red[256];
  __shared__ float bc;
  const int tid = threadIdx.x;
  // minibatch advantage mean / unbiased std (two pass) -- PPO.train: (A-mean)/(std+1e-8)
  float s = 0.f;
  for (int i = tid; i < batch; i += 256) s += adv[rollout_offset(idx[i], T, n_envs)];
  red[tid] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
  if (tid == 0) bc = red[0] / (float)batch;
  __syncthreads();
  const float mean = bc;
  float q = 0.f;
  for (int i = tid; i < batch; i += 256) {
    const float dl = adv[rollout_offset(idx[i], T, n_envs)] - mean;
    q += dl * dl;
  }
  __syncthreads();
  red[tid] = q;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
  if (tid == 0) {
    advstat[0] = mean;
    advstat[1] = batch > 1 ? sqrtf(red[0] / (float)(batch - 1)) : 0.f;
  }
  if (!(d.has_norm && update_norm)) return;
  // RunningNorm.update_stats on the minibatch observations (train mode, util/networks.py:111-134)
  const int D = d.obs_dim;
  const int cl = tid & 63, rl = tid >> 6;  // 4 row lanes x 64 columns
  __shared__ float r2[4][65];
  float cs = 0.f;
  if (cl < D)
    for (int i = rl; i < batch; i += 4) cs += obs[rollout_offset(idx[i], T, n_envs) * D + cl];
  r2[rl][cl] = cs;
  __syncthreads();
  float bmean = 0.f;
  if (cl < D) bmean = (r2[0][cl] + r2[1][cl] + r2[2][cl] + r2[3][cl]) / (float)batch;
  __syncthreads();
  float cq = 0.f;
  if (cl < D)
    for (int i = rl; i < batch; i += 4) {
      const float dl = obs[rollout_offset(idx[i], T, n_envs) * D + cl] - bmean;
      cq += dl * dl;
    }
  r2[rl][cl] = cq;
  __syncthreads();
  const int cnt = *ncount;
  if (rl == 0 && cl < D) {
    const float bvar = (r2[0][cl] + r2[1][cl] + r2[2][cl] + r2[3][cl]) / (float)batch;
    const float fcount = (float)cnt, fn = (float)batch, tot = (float)(cnt + batch);
    const float delta = bmean - nm[cl];
    nm[cl] = nm[cl] + delta * fn / tot;
    float rv = nv[cl] * fcount;
    rv = rv + bvar * fn;
    rv = rv + delta * delta * fcount * fn / tot;
    nv[cl] = rv / tot;
  }
  __syncthreads();
  if (tid == 0) *ncount = cnt + batch;
}

// G[J][K] (+= over the wave's 64 rows) = sum_r U[r][j] * V[r][k], U/V are LDS tiles with odd
// strides; written (not accumulated) to `dst` with leading dimension ldk. J,K multiples of 32
// are handled tile by tile; `jmax`/`kmax` clip the store.
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

template <int H>
__device__ void tower_backward_and_grads(const float* __restrict__ W2, const float* __restrict__ W1, int D,
                                         float (&dz2)[H] /* in: grad wrt a2 pre-tanh' ; used as scratch */,
                                         const float (&a2)[H], float* lds, int tid, float* __restrict__ slab,
                                         int oW1, int ob1, int oW2, int ob2) {
  using L = Lds<H>;
  float* a1row = lds + L::a1 + tid * L::HS;
  float* dzrow = lds + L::dz + tid * L::HS;
  // dz2 = da2 * (1 - a2^2)
#pragma unroll
  for (int j = 0; j < H; ++j) {
    dz2[j] = dz2[j] * (1.f - a2[j] * a2[j]);
    dzrow[j] = dz2[j];
  }
  __syncthreads();
  // dW2[j][k] = sum_r dz2[r][j] a1[r][k] ; db2[j] = sum_r dz2[r][j]
  for (int j0 = 0; j0 < H; j0 += 32)
    for (int k0 = 0; k0 < H; k0 += 32)
      mfma_outer_store(lds + L::dz, L::HS, lds + L::a1, L::HS, j0, k0, H, H, slab + oW2, H, tid);
  if (tid < H) {
    float s = 0.f;
    for (int r = 0; r < ROWS; ++r) s += lds[L::dz + r * L::HS + tid];
    slab[ob2 + tid] = s;
  }
  // da1[k] = sum_j W2[j][k] dz2[j]  (rows of W2 are contiguous: scalar loads)
  float da1[H];
#pragma unroll
  for (int k = 0; k < H; ++k) da1[k] = 0.f;
  for (int j = 0; j < H; ++j) {
    const float g = dzrow[j];
#pragma unroll
    for (int k = 0; k < H; ++k) da1[k] = fmaf(W2[j * H + k], g, da1[k]);
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < H; ++k) {
    const float a = a1row[k];
    dzrow[k] = da1[k] * (1.f - a * a);  // dz1
  }
  __syncthreads();
  // dW1[j][k] = sum_r dz1[r][j] x[r][k]
  for (int j0 = 0; j0 < H; j0 += 32)
    for (int k0 = 0; k0 < D; k0 += 32)
      mfma_outer_store(lds + L::dz, L::HS, lds + L::x, L::XS, j0, k0, H, D, slab + oW1, D, tid);
  if (tid < H) {
    float s = 0.f;
    for (int r = 0; r < ROWS; ++r) s += lds[L::dz + r * L::HS + tid];
    slab[ob1 + tid] = s;
  }
  __syncthreads();
  (void)W1;
}

template <int H>
__global__ __launch_bounds__(ROWS) void ppo_grad_kernel(ia_policy_desc d, const float* __restrict__ P,
                                                        const float* __restrict__ Pt, const float* __restrict__ nm,
                                                        const float* __restrict__ nv, const float* __restrict__ obs,
                                                        const float* __restrict__ actions,
                                                        const float* __restrict__ old_logp,
                                                        const float* __restrict__ adv, const float* __restrict__ ret,
                                                        const int64_t* __restrict__ idx, int batch, int T, int n_envs,
                                                        int normalize_adv, float clip, float ent_coef, float vf_coef,
                                                        float* __restrict__ ws, int nblk) {
  using L = Lds<H>;
  extern __shared__ float lds[];
  const int tid = threadIdx.x, i = blockIdx.x * ROWS + tid;
  const bool valid = i < batch;
  const int D = d.obs_dim, A = d.act_dim;
  const PolOff o = pol_offsets(D, A, H, d.discrete);
  const PpoWs w = ppo_ws(ws, nblk, o.total);
  float* slab = w.slabs + (long long)blockIdx.x * o.total;
  const long long src = valid ? rollout_offset(idx[i], T, n_envs) : 0;
  const int aw = d.discrete ? 1 : A;  // stored action width
  float* xrow = lds + L::x + tid * L::XS;
  float* a1row = lds + L::a1 + tid * L::HS;
  float* a2row = lds + L::a2 + tid * L::HS;
  float* outrow = lds + L::out + tid * L::AS;
  float* doutrow = lds + L::dout + tid * L::AS;
  // zero the padded feature columns the MFMA tiles may touch beyond D
  for (int k = D; k < L::XS; ++k) xrow[k] = 0.f;
  for (int a = 0; a < L::AS; ++a) doutrow[a] = 0.f;
  load_features(d, obs + src * D, nm, nv, valid, xrow);
  const float invB = 1.f / (float)batch;

  // ---------------- policy tower ----------------
  float a2[H];
  tower_forward<H>(Pt + o.pW1, P + o.pb1, Pt + o.pW2, P + o.pb2, D, xrow, a1row, a2row, a2);
  head_forward<H>(P + o.aW, P + o.ab, A, a2, outrow);
  float logp = 0.f, entropy = 0.f;
  float lse = 0.f;
  int act_i = 0;
  if (!d.discrete) {
    for (int a = 0; a < A; ++a) {
      const float ls = P[o.log_std + a];
      logp += gauss_logp_term(actions[src * aw + a], outrow[a], ls);
      entropy += 0.5f + LOG_SQRT_2PI + logf(expf(ls));
    }
  } else {
    float mx = outrow[0];
    for (int a = 1; a < A; ++a) mx = fmaxf(mx, outrow[a]);
    float se = 0.f;
    for (int a = 0; a < A; ++a) se += expf(outrow[a] - mx);
    lse = mx + logf(se);
    act_i = (int)actions[src];
    logp = outrow[act_i] - lse;
    for (int a = 0; a < A; ++a) {
      const float l = outrow[a] - lse;
      entropy -= expf(l) * l;
    }
  }
  float advn = adv[src];
  if (normalize_adv && batch > 1) advn = (advn - w.advstat[0]) / (w.advstat[1] + 1e-8f);
  const float log_ratio = logp - old_logp[src];
  const float ratio = expf(log_ratio);
  const float lo = 1.f - clip, hi = 1.f + clip;
  const float pl1 = advn * ratio;
  const float pl2 = advn * fminf(fmaxf(ratio, lo), hi);
  // d(-mean(min(pl1,pl2)))/d ratio with torch's tie rule (equal -> half to each branch)
  const float g1 = pl1 < pl2 ? 1.f : (pl1 == pl2 ? 0.5f : 0.f);
  const float g2 = pl2 < pl1 ? 1.f : (pl1 == pl2 ? 0.5f : 0.f);
  const float inrange = (ratio >= lo && ratio <= hi) ? 1.f : 0.f;
  float dlogp = valid ? -invB * advn * (g1 + g2 * inrange) * ratio : 0.f;

  // d loss / d head outputs
  float* auxrow = lds + L::aux + tid * L::AS;
  if (!d.discrete) {
    for (int a = 0; a < A; ++a) {
      const float ls = P[o.log_std + a];
      const float sd = expf(ls), var = sd * sd;
      const float diff = actions[src * aw + a] - outrow[a];
      doutrow[a] = dlogp * diff / var;
      // d logp/d log_std = diff^2/var - 1 ; entropy_loss = -mean(entropy) -> -ent_coef/B per row
      auxrow[a] = valid ? dlogp * (diff * diff / var - 1.f) - ent_coef * invB : 0.f;
    }
  } else {
    for (int a = 0; a < A; ++a) {
      const float l = outrow[a] - lse, p = expf(l);
      const float dH = -p * (l + entropy);  // d entropy / d logit_a
      float g = dlogp * ((a == act_i ? 1.f : 0.f) - p);
      g += valid ? -ent_coef * invB * dH : 0.f;
      doutrow[a] = g;
    }
  }
  __syncthreads();
  // head grads: dWa[a][k] = sum_r dout[r][a] a2[r][k]; dba[a] = sum_r dout[r][a]
  for (int k0 = 0; k0 < H; k0 += 32)
    mfma_outer_store(lds + L::dout, L::AS, lds + L::a2, L::HS, 0, k0, A, H, slab + o.aW, H, tid);
  if (tid < A) {
    float s = 0.f;
    for (int r = 0; r < ROWS; ++r) s += lds[L::dout + r * L::AS + tid];
    slab[o.ab + tid] = s;
  }
  if (!d.discrete) {
    // log_std gradient: column sums of the per-row terms staged in the `aux` tile
    if (tid < A) {
      float s = 0.f;
      for (int r = 0; r < ROWS; ++r) s += lds[L::aux + r * L::AS + tid];
      slab[o.log_std + tid] = s;
    }
  }
  // da2[k] = sum_a Wa[a][k] dout[a]
  float da2[H];
#pragma unroll
  for (int k = 0; k < H; ++k) da2[k] = 0.f;
  for (int a = 0; a < A; ++a) {
    const float g = doutrow[a];
#pragma unroll
    for (int k = 0; k < H; ++k) da2[k] = fmaf(P[o.aW + a * H + k], g, da2[k]);
  }
  __syncthreads();
  tower_backward_and_grads<H>(P + o.pW2, P + o.pW1, D, da2, a2, lds, tid, slab, o.pW1, o.pb1, o.pW2, o.pb2);

  // ---------------- value tower ----------------
  tower_forward<H>(Pt + o.vW1, P + o.vb1, Pt + o.vW2, P + o.vb2, D, xrow, a1row, a2row, a2);
  float v = P[o.cb];
#pragma unroll
  for (int k = 0; k < H; ++k) v = fmaf(P[o.cW + k], a2[k], v);
  const float verr = ret[src] - v;
  const float dv = valid ? vf_coef * 2.f * (v - ret[src]) * invB : 0.f;  // F.mse_loss(returns, values)
  for (int a = 0; a < L::AS; ++a) doutrow[a] = 0.f;
  doutrow[0] = dv;
  __syncthreads();
  for (int k0 = 0; k0 < H; k0 += 32)
    mfma_outer_store(lds + L::dout, L::AS, lds + L::a2, L::HS, 0, k0, 1, H, slab + o.cW, H, tid);
  if (tid == 0) {
    float s = 0.f;
    for (int r = 0; r < ROWS; ++r) s += lds[L::dout + r * L::AS];
    slab[o.cb] = s;
  }
#pragma unroll
  for (int k = 0; k < H; ++k) da2[k] = P[o.cW + k] * dv;
  __syncthreads();
  tower_backward_and_grads<H>(P + o.vW2, P + o.vW1, D, da2, a2, lds, tid, slab, o.vW1, o.vb1, o.vW2, o.vb2);

  // ---------------- loss statistics (SB3 logger values) ----------------
  float st[6];
  st[0] = valid ? -fminf(pl1, pl2) : 0.f;                                   // policy_gradient_loss
  st[1] = valid ? verr * verr : 0.f;                                         // value_loss
  st[2] = valid ? -entropy : 0.f;                                            // entropy_loss
  st[3] = valid ? (expf(log_ratio) - 1.f) - log_ratio : 0.f;                 // approx_kl
  st[4] = valid ? (fabsf(ratio - 1.f) > clip ? 1.f : 0.f) : 0.f;             // clip_fraction
  st[5] = 0.f;
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    float x = st[k];
    for (int s = 32; s > 0; s >>= 1) x += __shfl_down(x, s, 64);
    if (tid == 0) w.statpart[blockIdx.x * 8 + k] = x;
  }
}

__global__ __launch_bounds__(1024) void ppo_apply_kernel(ia_policy_desc d, float* __restrict__ P,
                                                         float* __restrict__ Pt, float* __restrict__ m,
                                                         float* __restrict__ v, float* __restrict__ ws, int nblk,
                                                         int batch, float max_norm, float ent_coef, float vf_coef,
                                                         float beta1, float beta2, float eps, float step_size,
                                                         float bc2_sqrt, float* __restrict__ stats) {
  __shared__ float red[16];
  __shared__ float coef;
  const int H = d.hidden, D = d.obs_dim;
  const PolOff o = pol_offsets(D, d.act_dim, H, d.discrete);
  const PpoWs w = ppo_ws(ws, nblk, o.total);
  const int tid = threadIdx.x;
  float sq = 0.f;
  for (int i = tid; i < o.total; i += blockDim.x) {
    float g = 0.f;
    for (int b = 0; b < nblk; ++b) g += w.slabs[(long long)b * o.total + i];
    w.grad[i] = g;
    sq += g * g;
  }
  for (int s = 32; s > 0; s >>= 1) sq += __shfl_down(sq, s, 64);
  if ((tid & 63) == 0) red[tid >> 6] = sq;
  __syncthreads();
  if (tid == 0) {
    float t = 0.f;
    for (int k = 0; k < (int)(blockDim.x >> 6); ++k) t += red[k];
    const float total_norm = sqrtf(t);
    // torch.nn.utils.clip_grad_norm_: coef = max_norm/(norm+1e-6), clamped to 1
    coef = fminf(max_norm / (total_norm + 1e-6f), 1.0f);
    if (stats) {
      float st[6] = {0, 0, 0, 0, 0, 0};
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
  for (int i = tid; i < o.total; i += blockDim.x) {
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
    if ((rc = set_lds(policy_act_kernel<32>, lds_bytes<32>()))) return rc;
    hipLaunchKernelGGL(policy_act_kernel<32>, dim3(cdiv(n, ROWS)), dim3(ROWS), lds_bytes<32>(), (hipStream_t)stream,
                       *d, params, params_t, norm_mean, norm_var, obs, n, noise, low, high, actions, clipped, values,
                       logp);
  } else {
    if ((rc = set_lds(policy_act_kernel<64>, lds_bytes<64>()))) return rc;
    hipLaunchKernelGGL(policy_act_kernel<64>, dim3(cdiv(n, ROWS)), dim3(ROWS), lds_bytes<64>(), (hipStream_t)stream,
                       *d, params, params_t, norm_mean, norm_var, obs, n, noise, low, high, actions, clipped, values,
                       logp);
  }
  IA_CHECK_LAUNCH();
  return IA_OK;
}

int ia_policy_evaluate(const ia_policy_desc* d, const float* params, const float* params_t, const float* norm_mean,
                       const float* norm_var, const float* obs, const float* actions, int n, float* logp,
                       float* values, float* entropy, void* stream) {
  if (!pol_ok(d) || n <= 0) return IA_ERR_ARG;
  int rc;
  if (d->hidden == 32) {
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

int64_t ia_ppo_ws_floats(const ia_policy_desc* d, int batch) {
  if (!pol_ok(d) || batch <= 0) return IA_ERR_ARG;
  const int nblk = cdiv(batch, ROWS);
  const int P = pol_offsets(d->obs_dim, d->act_dim, d->hidden, d->discrete).total;
  return 8 + (int64_t)nblk * 8 + (int64_t)nblk * P + P;
}

int ia_ppo_minibatch(const ia_policy_desc* d, float* params, float* params_t, float* norm_mean, float* norm_var,
                     int32_t* norm_count, int update_norm, const float* obs, const float* actions,
                     const float* old_logp, const float* advantages, const float* returns, const int64_t* idx,
                     int batch, int T, int n_envs, int normalize_adv, float clip_range, float ent_coef,
                     float vf_coef, float max_grad_norm, float* exp_avg, float* exp_avg_sq, float beta1,
                     float beta2, float adam_eps, float step_size, float bc2_sqrt, float* ws, float* stats,
                     void* stream) {
  if (!pol_ok(d) || batch <= 0) return IA_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int nblk = cdiv(batch, ROWS);
  const PolOff o = pol_offsets(d->obs_dim, d->act_dim, d->hidden, d->discrete);
  const PpoWs w = ppo_ws(ws, nblk, o.total);
  hipLaunchKernelGGL(ppo_prepare_kernel, dim3(1), dim3(256), 0, st, *d, obs, advantages, idx, batch, T, n_envs,
                     update_norm, norm_mean, norm_var, norm_count, w.advstat);
  IA_CHECK_LAUNCH();
  int rc;
  if (d->hidden == 32) {
    if ((rc = set_lds(ppo_grad_kernel<32>, lds_bytes<32>()))) return rc;
    hipLaunchKernelGGL(ppo_grad_kernel<32>, dim3(nblk), dim3(ROWS), lds_bytes<32>(), st, *d, params, params_t,
                       norm_mean, norm_var, obs, actions, old_logp, advantages, returns, idx, batch, T, n_envs,
                       normalize_adv, clip_range, ent_coef, vf_coef, ws, nblk);
  } else {
    if ((rc = set_lds(ppo_grad_kernel<64>, lds_bytes<64>()))) return rc;
    hipLaunchKernelGGL(ppo_grad_kernel<64>, dim3(nblk), dim3(ROWS), lds_bytes<64>(), st, *d, params, params_t,
                       norm_mean, norm_var, obs, actions, old_logp, advantages, returns, idx, batch, T, n_envs,
                       normalize_adv, clip_range, ent_coef, vf_coef, ws, nblk);
  }
  IA_CHECK_LAUNCH();
  hipLaunchKernelGGL(ppo_apply_kernel, dim3(1), dim3(1024), 0, st, *d, params, params_t, exp_avg, exp_avg_sq, ws,
                     nblk, batch, max_grad_norm, ent_coef, vf_coef, beta1, beta2, adam_eps, step_size, bc2_sqrt,
                     stats);
  IA_CHECK_LAUNCH();
  return IA_OK;
}

// One full PPO epoch (SB3 PPO.train inner loop over RolloutBuffer.get): `perm` is the host-drawn
// np.random.permutation(T*n_envs) already resident on the device; minibatches are consecutive
// slices of it (the last one may be short). Adam's bias corrections are formed in double per step.
int ia_ppo_epoch(const ia_policy_desc* d, float* params, float* params_t, float* norm_mean, float* norm_var,
                 int32_t* norm_count, int update_norm, const float* obs, const float* actions, const float* old_logp,
                 const float* advantages, const float* returns, const int64_t* perm, int T, int n_envs,
                 int batch_size, int normalize_adv, float clip_range, float ent_coef, float vf_coef,
                 float max_grad_norm, float* exp_avg, float* exp_avg_sq, double lr, double beta1, double beta2,
                 float adam_eps, int64_t adam_steps_done, float* ws, float* stats, void* stream) {
  if (!pol_ok(d) || batch_size <= 0) return IA_ERR_ARG;
  const long long total = (long long)T * n_envs;
  int64_t step = adam_steps_done;
  int mb = 0;
  for (long long start = 0; start < total; start += batch_size, ++mb) {
    const int b = (int)((total - start) < batch_size ? (total - start) : batch_size);
    ++step;
    const double bc1 = 1.0 - pow(beta1, (double)step);
    const double bc2 = 1.0 - pow(beta2, (double)step);
    int rc = ia_ppo_minibatch(d, params, params_t, norm_mean, norm_var, norm_count, update_norm, obs, actions,
                              old_logp, advantages, returns, perm + start, b, T, n_envs, normalize_adv, clip_range,
                              ent_coef, vf_coef, max_grad_norm, exp_avg, exp_avg_sq, (float)beta1, (float)beta2,
                              adam_eps, (float)(lr / bc1), (float)sqrt(bc2), ws, stats ? stats + mb * 8 : nullptr,
                              stream);
    if (rc) return rc;
  }
  return IA_OK;
}

}  // extern "C"
