// One AIRL discriminator update for the scripts' default shaped reward net (adversarial/airl.py:99-132,
// rewards/reward_nets.py:674-809: BasicShapedRewardNet = reward MLP D_b -> 32 -> 1 on [s | a | s' | done] plus potential
// MLP D_p -> 32 -> 32 -> 1 evaluated on s' and on s, ReLU) without the ~40 launches of the generic stack path
// (three dense stacks forward and backward through the GEMM family, row-dot heads, logit / BCE / routing kernels --
// each 2-15 us for nets this small: 380 us per update, profiles/r02_airl_kernel_stats.md).
//
//   airl_rows_kernel: ONE pass over the 2 x minibatch rows on the matrix cores. A wave owns 32 rows and computes every
//     layer TRANSPOSED (H^T = W . X^T, v_mfma_f32_32x32x2_f32): the accumulator of one layer -- lane = row, registers =
//     16 of the 32 features -- is, element by element, the B operand of the next layer's MFMAs (the contraction index
//     is simply visited in the accumulator's feature order, with the weight fragments pre-permuted to match in LDS),
//     so activations and deltas never leave the registers between layers:
//     normalise the three inputs (statistics as given: the caller has applied the train-mode updates, potential:
//     after the next-state batch for h(s'), after the state batch for h(s), util/networks.py:79-91),
//     forward the three stacks, logits = g + gamma (1 - done) h(s') - h(s) - log pi  (reward_nets.py:727-733, airl.py:118),
//     BCE-with-logits + its statistics (common.py:27-92, 360-368; same sums as bce_kernel), d logits routed to the three
//     outputs, deltas back through the ReLU layers (W2^T . delta^T is one more MFMA chain).
//     It writes what the weight gradients contract over the rows: normalised inputs, first hidden activations of the
//     potential and the hidden-layer deltas; the 1-wide output layers' gradients are reduced inside the workgroup
//     (transpose-and-halve butterflies: 16 shuffles for 16 sums) and stored as this workgroup's slab of the split-K
//     partial buffer.
//   The three hidden-layer weight gradients are then the existing split-K TN GEMMs (one slab per 128 rows) and
//   ia_reduce_partials_adam finishes: 1 + 3 + 1 launches behind the batch assembly and the statistics updates.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/imitation_hip.h"
#include "common.h"
#include "rn_common.h"

namespace {

constexpr int AH = 32;              // hidden width of all three stacks
constexpr int A_WAVES = 4;
constexpr int A_THREADS = 64 * A_WAVES;
constexpr int A_ROWS = 32 * A_WAVES;    // rows per workgroup = rows per split-K slab of the weight-gradient GEMMs
constexpr int A_CH = 8;                 // input chunks of 8 columns a stack's first layer may have
constexpr int A_D_MAX = 8 * A_CH;

struct AirlArgs {
  const float *Xb, *Sn, *Sc;        // [R, ldb] base inputs, [R, ldp] next / current observations (raw)
  int ldb, Db, ldp, Dp;
  const float *dones, *logp;        // [R]
  const float *bmean, *bvar;        // base input norm (null: none)
  const float *pmeanA, *pvarA;      // potential norm after the next-state update (null: none)
  const float *pmeanB, *pvarB;      // ... after the current-state update
  float beps, peps;
  const float *Pb, *Pp;             // flat parameters: base [W1 32xDb | b1 | wout 32 | bout], potential [W1 32xDp | b1 | W2 | b2 | wout | bout]
  float gamma, scale;
  int R, n_expert;
  float *Ab; int ldab;              // [R, ldab] normalised base inputs
  float *Db1;                       // [R, 32]
  float *Ap; int ldap;              // [2R, ldap] normalised potential inputs: rows r (next) and R + r (current)
  float *H1, *Dp1, *Dp2;            // [2R, 32] each
  float *part; long long part_stride;   // split-K partial buffer [slabs][part_stride]; this kernel fills the output layers
  int off_b_wout, off_b_bout, off_p_wout, off_p_bout;   // their offsets inside a slab
  float *logits, *stats, *bce_part;
  unsigned *ticket;
};

__device__ __forceinline__ f32x16 mfma(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// Accumulator element j of a lane in half h (= lane >> 5) is feature 8 (j / 4) + 4 h + j % 4 of row lane & 31: four
// runs of four consecutive features, so per-feature vectors and row-major stores go by float4.
__device__ __forceinline__ f32x16 per_feature(const float* __restrict__ v, int half) {
  f32x16 r;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x4 t = *reinterpret_cast<const f32x4*>(v + 8 * q + 4 * half);
#pragma unroll
    for (int u = 0; u < 4; ++u) r[4 * q + u] = t[u];
  }
  return r;
}

__device__ __forceinline__ void store_features(float* __restrict__ row32, const f32x16& v, int half) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    f32x4 t;
#pragma unroll
    for (int u = 0; u < 4; ++u) t[u] = v[4 * q + u];
    *reinterpret_cast<f32x4*>(row32 + 8 * q + 4 * half) = t;
  }
}

// Sums of 16 per-lane values over the 32 lanes of a half: every step trades half of the values with the partner
// lane and adds, so 8 + 4 + 2 + 1 + 1 shuffles do the work of 16 butterflies. Lane l ends up with the sum of element
// 8 b4 + 4 b3 + 2 b2 + b1 (bits of l); the pair (l, l ^ 1) holds the same one. Fixed order: deterministic.
__device__ __forceinline__ float half_sums16(const f32x16& p, int lane) {
  float a8[8], a4[4], a2[2];
  bool b = (lane & 16) != 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) a8[i] = (b ? p[i + 8] : p[i]) + __shfl_xor(b ? p[i] : p[i + 8], 16, 64);
  b = (lane & 8) != 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) a4[i] = (b ? a8[i + 4] : a8[i]) + __shfl_xor(b ? a8[i] : a8[i + 4], 8, 64);
  b = (lane & 4) != 0;
#pragma unroll
  for (int i = 0; i < 2; ++i) a2[i] = (b ? a4[i + 2] : a4[i]) + __shfl_xor(b ? a4[i] : a4[i + 2], 4, 64);
  b = (lane & 2) != 0;
  float a1 = (b ? a2[1] : a2[0]) + __shfl_xor(b ? a2[0] : a2[1], 2, 64);
  a1 += __shfl_xor(a1, 1, 64);
  return a1;
}

// Three independent vectors at once: each level's shuffles of all three are issued together, so the 16-step chain of
// ~150-clock cross-lane operations is walked once, not three times.
__device__ __forceinline__ void half_sums16x3(const f32x16& p0, const f32x16& p1, const f32x16& p2, int lane, float& r0,
                                              float& r1, float& r2) {
  const f32x16* p[3] = {&p0, &p1, &p2};
  float a8[3][8], a4[3][4], a2[3][2], a1[3];
  bool b = (lane & 16) != 0;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int v = 0; v < 3; ++v) a8[v][i] = (b ? (*p[v])[i + 8] : (*p[v])[i]) + __shfl_xor(b ? (*p[v])[i] : (*p[v])[i + 8], 16, 64);
  b = (lane & 8) != 0;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int v = 0; v < 3; ++v) a4[v][i] = (b ? a8[v][i + 4] : a8[v][i]) + __shfl_xor(b ? a8[v][i] : a8[v][i + 4], 8, 64);
  b = (lane & 4) != 0;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int v = 0; v < 3; ++v) a2[v][i] = (b ? a4[v][i + 2] : a4[v][i]) + __shfl_xor(b ? a4[v][i] : a4[v][i + 2], 4, 64);
  b = (lane & 2) != 0;
#pragma unroll
  for (int v = 0; v < 3; ++v) a1[v] = (b ? a2[v][1] : a2[v][0]) + __shfl_xor(b ? a2[v][0] : a2[v][1], 2, 64);
#pragma unroll
  for (int v = 0; v < 3; ++v) a1[v] += __shfl_xor(a1[v], 1, 64);
  r0 = a1[0];
  r1 = a1[1];
  r2 = a1[2];
}

// All chunks of one input row, issued up front (the kernel's only HBM-latency exposure): chunk q of a lane in half h is
// columns 8 q + 4 h .. + 3.
__device__ __forceinline__ void load_row(const float* __restrict__ xrow, int ld, int chunks, int half, f32x4 (&raw)[A_CH]) {
  // branch-free: behind a (wave-uniform) `if (q < chunks)` every load was followed by its own s_waitcnt at the end of the
  // block -- 24 serial memory round trips, 5.4 us, before the kernel had done anything. Unused chunks re-read the row's
  // last 16 bytes and are masked.
  f32x4 v[A_CH];
#pragma unroll
  for (int q = 0; q < A_CH; ++q) v[q] = *reinterpret_cast<const f32x4*>(xrow + min(8 * q + 4 * half, ld - 4));
#pragma unroll
  for (int q = 0; q < A_CH; ++q) {
    const bool on = q < chunks && 8 * q + 4 * half < ld;
#pragma unroll
    for (int u = 0; u < 4; ++u) raw[q][u] = on ? v[q][u] : 0.f;
  }
}

// First layer of a stack for the wave's 32 rows: normalise, store the normalised rows, relu(b1 + W1 xn) transposed.
__device__ __forceinline__ f32x16 first_layer(const f32x4 (&raw)[A_CH], int chunks, const float* __restrict__ mean,
                                              const float* __restrict__ istd, const f32x4* __restrict__ wf,
                                              const float* __restrict__ b1, float* __restrict__ xn_out, int ldo,
                                              bool live, int half) {
  f32x16 acc = per_feature(b1, half);
#pragma unroll
  for (int q = 0; q < A_CH; ++q) {
    if (q < chunks) {
      const int k0 = 8 * q + 4 * half;
      const f32x4 m = *reinterpret_cast<const f32x4*>(mean + k0);
      const f32x4 is = *reinterpret_cast<const f32x4*>(istd + k0);
      const f32x4 w = wf[q * 64];
      const f32x4 xn = (raw[q] - m) * is;          // (columns past the input width: istd = 0)
      if (live && k0 < ldo) *reinterpret_cast<f32x4*>(xn_out + k0) = xn;
#pragma unroll
      for (int u = 0; u < 4; ++u) acc = mfma(w[u], xn[u], acc);
    }
  }
#pragma unroll
  for (int j = 0; j < 16; ++j) acc[j] = fmaxf(acc[j], 0.f);
  return acc;
}

// acc0 + W . v for a 32 x 32 layer whose A fragments `wf` follow the accumulator's feature order
__device__ __forceinline__ f32x16 chain32(f32x16 acc, const f32x16& v, const f32x4* __restrict__ wf) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x4 w = wf[q * 64];
#pragma unroll
    for (int u = 0; u < 4; ++u) acc = mfma(w[u], v[4 * q + u], acc);
  }
  return acc;
}

__device__ __forceinline__ float dot_features(const f32x16& h, const f32x16& w) {
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 16; ++j) s += h[j] * w[j];
  return s + __shfl_xor(s, 32, 64);
}

long long* g_airl_tstamp = nullptr;   // debug: >= 16 shader clocks of workgroup 0's first wave (ia_airl_debug_timing)
#define AIRL_TS(slot) do { if (ts && bid == 0 && threadIdx.x == 0) ts[slot] = clock64(); } while (0)

struct AirlLds {
  f32x4 bW1f[A_CH * 64], pW1f[A_CH * 64], W2f[4 * 64], W2tf[4 * 64];
  float vec[5 * AH + 4];            // b_b1, b_wout, p_b1, p_b2, p_wout, then bout_b, bout_p
  float bmean[A_D_MAX], bistd[A_D_MAX], pmA[A_D_MAX], piA[A_D_MAX], pmB[A_D_MAX], piB[A_D_MAX];
  float red[A_WAVES][72];
  int is_last;
};

// (`bid` of `nblocks`: this workgroup's place among the launch's workgroups of THIS body -- the update's rows and the
//  gradient penalty's rows can share one launch, `airl_rows_gp_kernel`)
__device__ __forceinline__ void airl_rows_body(const AirlArgs& a, long long* __restrict__ ts, AirlLds& S, const int bid,
                                               const unsigned nblocks) {
  AIRL_TS(0);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5;
  const int Cb = (a.Db + 7) >> 3, Cp = (a.Dp + 7) >> 3;
  const int r = bid * A_ROWS + wave * 32 + (lane & 31);
  const bool live = r < a.R;
  const int rr = live ? r : a.R - 1;             // (rows past the end recompute the last one; nothing of theirs is stored or summed)
  f32x4 rawb[A_CH], rawn[A_CH], rawc[A_CH];
  load_row(a.Xb + (long long)rr * a.ldb, a.ldb, Cb, half, rawb);
  load_row(a.Sn + (long long)rr * a.ldp, a.ldp, Cp, half, rawn);
  load_row(a.Sc + (long long)rr * a.ldp, a.ldp, Cp, half, rawc);
  const float done = a.dones[rr], lp = a.logp[rr];
  AIRL_TS(1);

  const float* Pb = a.Pb;
  const float* Pp = a.Pp;
  const int ob_b1 = AH * a.Db, ob_wout = ob_b1 + AH, ob_bout = ob_wout + AH;
  const int op_b1 = AH * a.Dp, op_W2 = op_b1 + AH, op_b2 = op_W2 + AH * AH, op_wout = op_b2 + AH, op_bout = op_wout + AH;
  // weight fragments: entry (q, lane) = the four A values A[m = lane & 31][k = 8 q + 4 (lane >> 5) + u]. Staged
  // branch-free -- every global load of a thread (clamped addresses, masked afterwards) is issued before its first LDS
  // store: as loops with the loads inside, the staging was 14 serial memory round trips (3.3 us).
  {
    constexpr int NE = A_CH * 64 / A_THREADS;       // first-layer entries per thread and stack
    static_assert(A_CH * 64 % A_THREADS == 0 && 4 * 64 == A_THREADS, "staging assumes 256 threads");
    f32x4 wb[NE], wp[NE], w2, w2t;
#pragma unroll
    for (int it = 0; it < NE; ++it) {
      const int e = tid + it * A_THREADS;
      const int m = e & 31, k0 = 8 * (e >> 6) + 4 * ((e >> 5) & 1);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        wb[it][u] = Pb[m * a.Db + min(k0 + u, a.Db - 1)];
        wp[it][u] = Pp[m * a.Dp + min(k0 + u, a.Dp - 1)];
      }
    }
    {
      const int m = tid & 31, k0 = 8 * (tid >> 6) + 4 * ((tid >> 5) & 1);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        w2[u] = Pp[op_W2 + m * AH + k0 + u];          // forward:  A[m = out][k = in]
        w2t[u] = Pp[op_W2 + (k0 + u) * AH + m];       // backward: A[m = in][k = out]
      }
    }
    const int vt = min(tid, AH - 1), kt = min(tid, A_D_MAX - 1);
    const float v0 = Pb[ob_b1 + vt], v1 = Pb[ob_wout + vt], v2 = Pp[op_b1 + vt], v3 = Pp[op_b2 + vt], v4 = Pp[op_wout + vt];
    const float vb0 = Pb[ob_bout], vb1 = Pp[op_bout];
    const int kb = min(kt, a.Db - 1), kp = min(kt, a.Dp - 1);
    const bool hb_ = a.bmean != nullptr, hp_ = a.pmeanA != nullptr;
    const float sbm = hb_ ? a.bmean[kb] : 0.f, sbv = hb_ ? a.bvar[kb] : 1.f;
    const float sam = hp_ ? a.pmeanA[kp] : 0.f, sav = hp_ ? a.pvarA[kp] : 1.f;
    const float sqm = hp_ ? a.pmeanB[kp] : 0.f, sqv = hp_ ? a.pvarB[kp] : 1.f;
#pragma unroll
    for (int it = 0; it < NE; ++it) {
      const int e = tid + it * A_THREADS;
      const int k0 = 8 * (e >> 6) + 4 * ((e >> 5) & 1);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        wb[it][u] = k0 + u < a.Db ? wb[it][u] : 0.f;
        wp[it][u] = k0 + u < a.Dp ? wp[it][u] : 0.f;
      }
      S.bW1f[e] = wb[it];
      S.pW1f[e] = wp[it];
    }
    S.W2f[tid] = w2;
    S.W2tf[tid] = w2t;
    if (tid < AH) {
      S.vec[tid] = v0;
      S.vec[AH + tid] = v1;
      S.vec[2 * AH + tid] = v2;
      S.vec[3 * AH + tid] = v3;
      S.vec[4 * AH + tid] = v4;
    }
    if (tid == 0) {
      S.vec[5 * AH] = vb0;
      S.vec[5 * AH + 1] = vb1;
    }
    if (tid < A_D_MAX) {
      const int k = tid;
      const bool inb = k < a.Db, inp = k < a.Dp;
      S.bmean[k] = inb ? sbm : 0.f;
      S.bistd[k] = inb ? (hb_ ? 1.f / sqrtf(sbv + a.beps) : 1.f) : 0.f;
      S.pmA[k] = inp ? sam : 0.f;
      S.piA[k] = inp ? (hp_ ? 1.f / sqrtf(sav + a.peps) : 1.f) : 0.f;
      S.pmB[k] = inp ? sqm : 0.f;
      S.piB[k] = inp ? (hp_ ? 1.f / sqrtf(sqv + a.peps) : 1.f) : 0.f;
    }
  }
  __syncthreads();
  AIRL_TS(2);

  const float* b_b1 = S.vec;
  const float* p_b1 = S.vec + 2 * AH;
  const f32x16 b_wout = per_feature(S.vec + AH, half), p_wout = per_feature(S.vec + 4 * AH, half);
  const f32x16 p_b2 = per_feature(S.vec + 3 * AH, half);
  const f32x4* bW1f = S.bW1f + lane;
  const f32x4* pW1f = S.pW1f + lane;
  const f32x4* W2f = S.W2f + lane;
  const f32x4* W2tf = S.W2tf + lane;

  // forward (reference order: base, potential(next_state), potential(state))
  const f32x16 hb = first_layer(rawb, Cb, S.bmean, S.bistd, bW1f, b_b1, a.Ab + (long long)rr * a.ldab, a.ldab, live, half);
  const float g = S.vec[5 * AH] + dot_features(hb, b_wout);
  const f32x16 h1n = first_layer(rawn, Cp, S.pmA, S.piA, pW1f, p_b1, a.Ap + (long long)rr * a.ldap, a.ldap, live, half);
  f32x16 h2n = chain32(p_b2, h1n, W2f);
  const f32x16 h1c = first_layer(rawc, Cp, S.pmB, S.piB, pW1f, p_b1, a.Ap + (long long)(a.R + rr) * a.ldap, a.ldap, live, half);
  f32x16 h2c = chain32(p_b2, h1c, W2f);
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    h2n[j] = fmaxf(h2n[j], 0.f);
    h2c[j] = fmaxf(h2c[j], 0.f);
  }
  const float h_next = S.vec[5 * AH + 1] + dot_features(h2n, p_wout);
  const float h_cur = S.vec[5 * AH + 1] + dot_features(h2c, p_wout);
  AIRL_TS(3);

  // logits (reward_nets.py:727-733 order, then airl.py:118) and BCE (bce_kernel's expressions); both halves hold them
  float f = g + a.gamma * ((1.f - done) * h_next);
  f = f - h_cur;
  const float x = f - lp;
  const float y = rr < a.n_expert ? 1.f : 0.f;
  const float lse = log1pf(expf(-fabsf(x)));
  const float p = 1.f / (1.f + expf(-x));
  const float dlog = live ? (p - y) * (a.scale / (float)a.R) : 0.f;
  const float dg = dlog, dhn = a.gamma * (1.f - done) * dlog, dhc = -dlog;
  f32x16 sc;                                     // scalars to sum over the rows: elements 0..7
#pragma unroll
  for (int j = 0; j < 16; ++j) sc[j] = 0.f;
  if (live) {
    if (half == 0) a.logits[r] = x;
    const bool is_gen_pred = x < 0.f, is_gen_true = y == 0.f, ok = is_gen_pred == is_gen_true;
    sc[0] = (1.f - y) * x - (fminf(x, 0.f) - lse);
    sc[1] = ok ? 1.f : 0.f;
    sc[2] = (ok && !is_gen_true) ? 1.f : 0.f;
    sc[3] = (ok && is_gen_true) ? 1.f : 0.f;
    sc[4] = is_gen_pred ? 1.f : 0.f;
    sc[5] = (1.f - p) * x - (fminf(x, 0.f) - lse);
    sc[6] = dg;                                  // d bout of the base stack
    sc[7] = dhn + dhc;                           // d bout of the potential
  }

  // backward: deltas of the hidden layers for the TN GEMMs, output-layer gradients reduced here
  f32x16 gb, gp, t;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    gb[j] = dg * hb[j];
    gp[j] = dhn * h2n[j] + dhc * h2c[j];
    t[j] = hb[j] > 0.f ? dg * b_wout[j] : 0.f;
  }
  if (live) store_features(a.Db1 + (long long)rr * AH, t, half);
  auto pot_back = [&](const f32x16& h1, const f32x16& h2, float dh, long long row) {
    f32x16 d2, z;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      d2[j] = h2[j] > 0.f ? dh * p_wout[j] : 0.f;
      z[j] = 0.f;
    }
    f32x16 d1 = chain32(z, d2, W2tf);
#pragma unroll
    for (int j = 0; j < 16; ++j) d1[j] = h1[j] > 0.f ? d1[j] : 0.f;
    if (live) {
      store_features(a.Dp2 + row * AH, d2, half);
      store_features(a.Dp1 + row * AH, d1, half);
      store_features(a.H1 + row * AH, h1, half);
    }
  };
  AIRL_TS(4);
  pot_back(h1n, h2n, dhn, (long long)rr);
  pot_back(h1c, h2c, dhc, (long long)(a.R + rr));
  AIRL_TS(5);
  float s_b, s_p, s_s;
  half_sums16x3(gb, gp, sc, lane, s_b, s_p, s_s);
  {
    const int j = ((lane >> 4) & 1) * 8 + ((lane >> 3) & 1) * 4 + ((lane >> 2) & 1) * 2 + ((lane >> 1) & 1);
    const int feat = 8 * (j >> 2) + 4 * half + (j & 3);
    if ((lane & 1) == 0) {
      S.red[wave][feat] = s_b;
      S.red[wave][AH + feat] = s_p;
      if (half == 0 && j < 8) S.red[wave][2 * AH + j] = s_s;
    }
  }
  __syncthreads();
  if (tid < 72) {
    float tsum = S.red[0][tid];
#pragma unroll
    for (int w = 1; w < A_WAVES; ++w) tsum += S.red[w][tid];
    float* slab = a.part + (long long)bid * a.part_stride;
    if (tid < AH) slab[a.off_b_wout + tid] = tsum;
    else if (tid < 2 * AH) slab[a.off_p_wout + tid - AH] = tsum;
    else if (tid < 2 * AH + 6)   // (written THROUGH to memory: the hand-off below then needs no L2 write-back)
      __hip_atomic_store(a.bce_part + bid * 8 + tid - 2 * AH, tsum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    else if (tid == 2 * AH + 6) slab[a.off_b_bout] = tsum;
    else slab[a.off_p_bout] = tsum;
  }
  // statistics: the workgroup that draws the last ticket folds the partials. Only the six partial sums cross
  // workgroups inside this launch (everything else is read by the NEXT launches), so instead of an agent-scope release
  // fence -- which writes back the ~160 KB of GEMM operands this workgroup has just stored: 5 us -- the partials go
  // out as system-scope stores, are waited for (vmcnt), and are read back with system-scope loads.
  AIRL_TS(6);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  AIRL_TS(7);
  if (tid == 0) {
    const unsigned tk = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    AIRL_TS(8);
    S.is_last = (tk == nblocks - 1);
  }
  __syncthreads();
  if (!S.is_last) return;
  {
    // fold: thread (k, j) sums workgroups j, j + 32, ... of statistic k with all its loads in flight, then lane j = 0
    // adds the 32 partial sums in order -- fixed order, ~2 memory round trips instead of one per workgroup
    const int k = tid >> 5, j = tid & 31;
    float t = 0.f;
    if (k < 6)
      for (unsigned b = j; b < nblocks; b += 32)
        t += __hip_atomic_load(a.bce_part + b * 8 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    float* fold = &S.red[0][0];          // (the block sums above were consumed before the last barrier: reuse 8 x 32 floats)
    if (k < 8) fold[k * 32 + j] = t;
    __syncthreads();
    if (tid < 6) {
      float tsum = 0.f;
      for (int q = 0; q < 32; ++q) tsum += fold[tid * 32 + q];
      if (tid == 0) tsum = tsum / (float)a.R * a.scale;
      a.stats[tid] = tsum;
    }
  }
  if (tid == 6) a.stats[6] = (float)a.n_expert;
  if (tid == 7) a.stats[7] = (float)(a.R - a.n_expert);
  if (tid == 0) __hip_atomic_store(a.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(A_THREADS) void airl_rows_kernel(AirlArgs a, long long* __restrict__ ts) {
  __shared__ __attribute__((aligned(16))) AirlLds S;
  airl_rows_body(a, ts, S, blockIdx.x, gridDim.x);
}


// ---- batch assembly + input statistics in one launch ---------------------------------------------------------------
// One pass of the assembly: `ncols` consecutive columns of a destination matrix, all of one kind, taken from column
// `off` on of a per-row source array (expert table for the slab's first rows, generator table for the rest).
enum { AP_F32 = 0, AP_ONEHOT = 1, AP_DONE = 2, AP_INDEX = 3 };
struct AirlPass {
  const void *p0, *p1;   // the two tables' arrays: float [n, stride] (AP_F32), int64 [n] (AP_ONEHOT / AP_INDEX), uint8 [n] (AP_DONE)
  int kind, stride, off, ncols;
  float* X; int ldx, xcol;    // destination, its row stride, first destination column
  float* ws; int wsD;         // slab moments [slabs][2][wsD] of the destination matrix (null: not wanted)
  long long x_stride, ws_stride;   // several batches in one launch (blockIdx.z = batch): element offsets per batch
};
constexpr int AP_MAX = 12;

struct AirlPrep {
  const int64_t *idx0, *idx1;   // rows of the two tables (null: the first n)
  long long idx_stride;         // ... of batch blockIdx.z: idx + z * idx_stride
  int n0, R;
  int n_pass;
  AirlPass pass[AP_MAX];
};

__global__ __launch_bounds__(256) void airl_prepare_kernel(AirlPrep a) {
  // Workgroup (slab, pass): 256 rows of one pass. Element (row, c) is gathered once, stored, and feeds the slab moments
  // with rn_partial_kernel's row partition and summation order (mlp.hip:104: 8 row lanes x 32 rows each, lanes folded
  // in order), so the statistics equal those of a stand-alone update of the stored matrix bit for bit.
  __shared__ float red[8][33];
  constexpr int RPT = RN_ROWS_PER_BLOCK / 8;
  const int r0 = blockIdx.x * RN_ROWS_PER_BLOCK;
  const int rows = min(RN_ROWS_PER_BLOCK, a.R - r0);
  const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
  // table row and source of the thread's rows (rows past the slab's end repeat its last one; never stored or summed)
  int src[RPT];                 // (tables have < 2^31 rows)
  unsigned wmask = 0;           // bit k: the thread's k-th row comes from the generator source
  static_assert(RPT <= 32, "one bit per row");
#pragma unroll
  for (int k = 0; k < RPT; ++k) {
    const int i = r0 + min(rl + 8 * k, rows - 1);
    const int w = i >= a.n0 ? 1 : 0, j = i - (w ? a.n0 : 0);
    wmask |= (unsigned)w << k;
    const int64_t* ix = w ? a.idx1 : a.idx0;
    src[k] = ix ? (int)ix[(long long)blockIdx.z * a.idx_stride + j] : j;
  }
  {
    AirlPass P = a.pass[blockIdx.y];
    P.X += (long long)blockIdx.z * P.x_stride;
    if (P.ws != nullptr) P.ws += (long long)blockIdx.z * P.ws_stride;
    for (int c0 = 0; c0 < P.ncols; c0 += 32) {
      const int c = c0 + cl, cc = min(c, P.ncols - 1);
      const bool col_on = c < P.ncols;
      float v[RPT];
      if (P.kind == AP_F32) {
        const float *q0 = (const float*)P.p0 + cc, *q1 = (const float*)P.p1 + cc;
#pragma unroll
        for (int k = 0; k < RPT; ++k) v[k] = (((wmask >> k) & 1u) ? q1 : q0)[(long long)src[k] * P.stride];
      } else if (P.kind == AP_DONE) {
        const uint8_t *q0 = (const uint8_t*)P.p0, *q1 = (const uint8_t*)P.p1;
#pragma unroll
        for (int k = 0; k < RPT; ++k) v[k] = (((wmask >> k) & 1u) ? q1 : q0)[src[k]] ? 1.f : 0.f;
      } else {
        const int64_t *q0 = (const int64_t*)P.p0, *q1 = (const int64_t*)P.p1;
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
          const int64_t t = (((wmask >> k) & 1u) ? q1 : q0)[src[k]];
          v[k] = P.kind == AP_ONEHOT ? (t == cc ? 1.f : 0.f) : (float)t;
        }
      }
      float* xp = P.X + (long long)r0 * P.ldx + P.xcol + c;
#pragma unroll
      for (int k = 0; k < RPT; ++k)
        if (col_on && rl + 8 * k < rows) xp[(long long)(rl + 8 * k) * P.ldx] = v[k];
      if (P.ws == nullptr) continue;
      float sum = 0.f;
#pragma unroll
      for (int k = 0; k < RPT; ++k) sum += (col_on && rl + 8 * k < rows) ? v[k] : 0.f;
      red[rl][cl] = sum;
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
        q += (col_on && rl + 8 * k < rows) ? dlt * dlt : 0.f;
      }
      red[rl][cl] = q;
      __syncthreads();
      if (rl == 0 && col_on) {
        float t = 0.f;
        for (int k = 0; k < 8; ++k) t += red[k][cl];
        P.ws[((long long)blockIdx.x * 2 + 0) * P.wsD + P.xcol + c] = mean;
        P.ws[((long long)blockIdx.x * 2 + 1) * P.wsD + P.xcol + c] = t;
      }
      __syncthreads();
    }
  }
}

// The train-mode statistics updates of one forward (util/networks.py:111-134; reference order reward_nets.py:708-710):
// base input norm with the assembled batch; potential input norm with the next-state batch -- the statistics h(s') is
// normalised with, copied to `snapA` -- and then with the state batch. One wave per column; the workgroup that draws
// the last ticket bumps the sample counts once every column has read them.
__global__ __launch_bounds__(64) void airl_stats_merge_kernel(const float* __restrict__ ws_b, const float* __restrict__ ws_n,
                                                              const float* __restrict__ ws_c, int groups, long long gstride,
                                                              int R, int Db, int Dp,
                                                              float* __restrict__ bmean, float* __restrict__ bvar,
                                                              int32_t* __restrict__ bcount, float* __restrict__ pmean,
                                                              float* __restrict__ pvar, int32_t* __restrict__ pcount,
                                                              float* __restrict__ snapA, unsigned* __restrict__ ticket) {
  // `groups` data-parallel ranks contribute R rows each (slab moments `gstride` floats apart): every rank merges all of
  // them in rank order, i.e. makes the update a single process would make on the concatenated batch
  const int lane = threadIdx.x;
  const int bpg = (R + RN_ROWS_PER_BLOCK - 1) / RN_ROWS_PER_BLOCK;
  const int nblocks = groups * bpg;
  const int Rt = groups * R;
  const int nbase = ws_b ? Db : 0;
  if ((int)blockIdx.x < nbase) {
    const int c = blockIdx.x, cnt = *bcount;
    float bm, bq;
    rn_wave_batch_moments(ws_b, nblocks, bpg, R, Db, c, lane, bm, bq, gstride);
    if (lane == 0) {
      float mc = bmean[c], vc = bvar[c];
      rn_absorb(mc, vc, cnt, Rt, bm, bq / (float)Rt);
      bmean[c] = mc;
      bvar[c] = vc;
    }
  } else {
    const int c = blockIdx.x - nbase, cnt = *pcount;
    float bm, bq, cm, cq;
    rn_wave_batch_moments(ws_n, nblocks, bpg, R, Dp, c, lane, bm, bq, gstride);
    rn_wave_batch_moments(ws_c, nblocks, bpg, R, Dp, c, lane, cm, cq, gstride);
    if (lane == 0) {
      float mc = pmean[c], vc = pvar[c];
      rn_absorb(mc, vc, cnt, Rt, bm, bq / (float)Rt);
      snapA[c] = mc;
      snapA[Dp + c] = vc;
      rn_absorb(mc, vc, rn_count_add(cnt, Rt), Rt, cm, cq / (float)Rt);
      pmean[c] = mc;
      pvar[c] = vc;
    }
  }
  if (lane == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned tk = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tk == gridDim.x - 1) {
      if (ws_b) *bcount = rn_count_add(*bcount, Rt);
      if (ws_n) *pcount = rn_count_add(rn_count_add(*pcount, Rt), Rt);
      __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}


// ---- gradient penalty of the shaped reward (opt-in extension, imitation_amd/grad_penalty.py) on the same chains -------------
// f(s, a, s') = g([s | a | s' | d]) + gamma (1 - d) h(s') - h(s) at x_hat = e x_expert + (1 - e) x_generator; penalty
// coef * mean_i (|grad f|_2 - target)^2 over every input block. The stacks are ReLU, so with the masks of the forward
// at x_hat fixed the input gradient is a product of the weight matrices, and so is the penalty's parameter gradient:
//   forward (masks)            h = relu(.)                                   -- the forward chains of airl_rows_kernel
//   input gradient, dOut = 1   u_L = w_out * m_L, u_l = m_l * (W_{l+1}^T u_{l+1}), gn = W_1^T u_1   (transposed chains)
//   rows                       T = signed combination of the three gn / sigma, n = |T|, Q = coef/B 2 (n - t)/n T,
//                              Cn = Q mapped back into each stack's normalised input (cross-stack columns via LDS tiles)
//   second pass                dW_1 += u_1^T Cn, dV_1 = m_1 * (W_1 Cn), dW_2 += u_2^T dV_1, dV_2 = m_2 * (W_2 dV_1),
//                              dw_out += column sums of dV_L; biases get nothing (the input gradient does not depend on them)
// The row-contracting products (dW_1, dW_2) are split-K TN GEMMs on what this kernel writes, as in the update itself.
constexpr int GP_TS = 68;                    // row stride of the per-wave exchange tiles (floats)
struct AirlGpArgs {
  const float *Xb, *Sn, *Sc; int ldb, Db, ldp, Dp;   // assembled [expert | generator] batches, 2B rows
  const float *dones, *e;                            // [2B], [B]
  const float *bmean, *bvar, *pmean, *pvar; float beps, peps;   // frozen statistics (null: none)
  const float *Pb, *Pp;
  int od, ad, use_state, use_action, use_next, use_done;
  float gamma, coef, target; int B;
  float *U1b, *Cb; int ldcb;          // [B, 32], [B, ldcb]
  float *U1p, *Cp; int ldcp;          // [2B, 32], [2B, ldcp]: rows r (next) and B + r (current)
  float *U2p, *V1p;                   // [2B, 32] each
  float *part; long long part_stride; int off_b_wout, off_p_wout;
  float *pen_part, *pen_out; unsigned* ticket;
};

struct GpLds {   // (carved from dynamic LDS)
  f32x4 bW1f[A_CH * 64], pW1f[A_CH * 64], W2f[4 * 64], W2tf[4 * 64];
  f32x4 bW1Tf[2 * 4 * 64], pW1Tf[2 * 4 * 64];     // A fragments of W1^T: [column tile][q][lane]
  float vec[5 * AH + 4];
  float bmean[A_D_MAX], bistd[A_D_MAX], pmean[A_D_MAX], pistd[A_D_MAX];
  float red[A_WAVES][72];
  float tiles[A_WAVES][3][32 * GP_TS];            // per wave: G_b, G_n, G_c (input gradients w.r.t. the raw inputs)
  int is_last;
};

// interpolated, normalised chunks of one row pair (r, r + B)
__device__ __forceinline__ void gp_mix(const f32x4 (&x0)[A_CH], const f32x4 (&x1)[A_CH], float w, int chunks,
                                       const float* __restrict__ mean, const float* __restrict__ istd, int half,
                                       f32x4 (&xn)[A_CH]) {
#pragma unroll
  for (int q = 0; q < A_CH; ++q) {
    xn[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (q < chunks) {
      const int k0 = 8 * q + 4 * half;
      const f32x4 m = *reinterpret_cast<const f32x4*>(mean + k0);
      const f32x4 is = *reinterpret_cast<const f32x4*>(istd + k0);
      xn[q] = ((w * x0[q] + (1.f - w) * x1[q]) - m) * is;
    }
  }
}

// relu(b1 + W1 xn) of a stack's first layer (transposed chain; `xn` in chunk layout)
__device__ __forceinline__ f32x16 gp_first(const f32x4 (&xn)[A_CH], int chunks, const f32x4* __restrict__ wf,
                                           const float* __restrict__ b1, int half, bool relu) {
  f32x16 acc = per_feature(b1, half);
#pragma unroll
  for (int q = 0; q < A_CH; ++q)
    if (q < chunks) {
      const f32x4 w = wf[q * 64];
#pragma unroll
      for (int u = 0; u < 4; ++u) acc = mfma(w[u], xn[q][u], acc);
    }
  if (relu) {
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = fmaxf(acc[j], 0.f);
  }
  return acc;
}

// gn = W1^T u for the column tiles of a stack (lane = row, registers = 16 of the tile's 32 columns)
__device__ __forceinline__ void gp_input_grad(const f32x16& u, const f32x4* __restrict__ wtf, int tiles, f32x16 (&gn)[2]) {
#pragma unroll
  for (int t = 0; t < 2; ++t) {
#pragma unroll
    for (int j = 0; j < 16; ++j) gn[t][j] = 0.f;
    if (t < tiles) gn[t] = chain32(gn[t], u, wtf + t * 4 * 64);
  }
}

__device__ __forceinline__ void airl_gp_rows_body(const AirlGpArgs& a, GpLds& S, const int bid, const unsigned nblocks) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, lrow = lane & 31;
  const int Cb = (a.Db + 7) >> 3, Cp = (a.Dp + 7) >> 3;
  const int Tb = (a.Db + 31) >> 5, Tp = (a.Dp + 31) >> 5;
  const int r = bid * A_ROWS + wave * 32 + lrow;
  const bool live = r < a.B;
  const int rr = live ? r : a.B - 1;
  f32x4 b0[A_CH], b1[A_CH], n0[A_CH], n1[A_CH], c0[A_CH], c1[A_CH];
  load_row(a.Xb + (long long)rr * a.ldb, a.ldb, Cb, half, b0);
  load_row(a.Xb + (long long)(rr + a.B) * a.ldb, a.ldb, Cb, half, b1);
  load_row(a.Sn + (long long)rr * a.ldp, a.ldp, Cp, half, n0);
  load_row(a.Sn + (long long)(rr + a.B) * a.ldp, a.ldp, Cp, half, n1);
  load_row(a.Sc + (long long)rr * a.ldp, a.ldp, Cp, half, c0);
  load_row(a.Sc + (long long)(rr + a.B) * a.ldp, a.ldp, Cp, half, c1);
  const float w = a.e[rr];
  const float dhat = w * a.dones[rr] + (1.f - w) * a.dones[rr + a.B];

  const float* Pb = a.Pb;
  const float* Pp = a.Pp;
  const int ob_b1 = AH * a.Db, ob_wout = ob_b1 + AH;
  const int op_b1 = AH * a.Dp, op_W2 = op_b1 + AH, op_b2 = op_W2 + AH * AH, op_wout = op_b2 + AH;
  {   // fragments, staged branch-free: all of a thread's global loads (clamped, masked afterwards) before its LDS stores
    constexpr int NE = A_CH * 64 / A_THREADS;
    f32x4 wb[NE], wp[NE], tb[NE], tp[NE], w2, w2t;
#pragma unroll
    for (int it = 0; it < NE; ++it) {
      const int e = tid + it * A_THREADS;
      const int m = e & 31, k0 = 8 * (e >> 6) + 4 * ((e >> 5) & 1);
      // W1^T entries: A[m = column 32 t + (lane & 31)][k = feature 8 q + 4 half + u]
      const int t = e >> 8, f0 = 8 * ((e >> 6) & 3) + 4 * ((e >> 5) & 1), col = 32 * t + m;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        wb[it][u] = Pb[m * a.Db + min(k0 + u, a.Db - 1)];
        wp[it][u] = Pp[m * a.Dp + min(k0 + u, a.Dp - 1)];
        tb[it][u] = Pb[(f0 + u) * a.Db + min(col, a.Db - 1)];
        tp[it][u] = Pp[(f0 + u) * a.Dp + min(col, a.Dp - 1)];
      }
    }
    {
      const int m = tid & 31, k0 = 8 * (tid >> 6) + 4 * ((tid >> 5) & 1);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        w2[u] = Pp[op_W2 + m * AH + k0 + u];
        w2t[u] = Pp[op_W2 + (k0 + u) * AH + m];
      }
    }
#pragma unroll
    for (int it = 0; it < NE; ++it) {
      const int e = tid + it * A_THREADS;
      const int k0 = 8 * (e >> 6) + 4 * ((e >> 5) & 1), col = 32 * (e >> 8) + (e & 31);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        wb[it][u] = k0 + u < a.Db ? wb[it][u] : 0.f;
        wp[it][u] = k0 + u < a.Dp ? wp[it][u] : 0.f;
        tb[it][u] = col < a.Db ? tb[it][u] : 0.f;
        tp[it][u] = col < a.Dp ? tp[it][u] : 0.f;
      }
      S.bW1f[e] = wb[it];
      S.pW1f[e] = wp[it];
      S.bW1Tf[e] = tb[it];
      S.pW1Tf[e] = tp[it];
    }
    S.W2f[tid] = w2;
    S.W2tf[tid] = w2t;
  }
  if (tid < AH) {
    S.vec[tid] = Pb[ob_b1 + tid];
    S.vec[AH + tid] = Pb[ob_wout + tid];
    S.vec[2 * AH + tid] = Pp[op_b1 + tid];
    S.vec[3 * AH + tid] = Pp[op_b2 + tid];
    S.vec[4 * AH + tid] = Pp[op_wout + tid];
  }
  if (tid < A_D_MAX) {
    const int k = tid;
    const bool inb = k < a.Db, inp = k < a.Dp;
    S.bmean[k] = (inb && a.bmean) ? a.bmean[k] : 0.f;
    S.bistd[k] = inb ? (a.bmean ? 1.f / sqrtf(a.bvar[k] + a.beps) : 1.f) : 0.f;
    S.pmean[k] = (inp && a.pmean) ? a.pmean[k] : 0.f;
    S.pistd[k] = inp ? (a.pmean ? 1.f / sqrtf(a.pvar[k] + a.peps) : 1.f) : 0.f;
  }
  __syncthreads();
  const f32x16 b_wout = per_feature(S.vec + AH, half), p_wout = per_feature(S.vec + 4 * AH, half);
  const f32x16 p_b2 = per_feature(S.vec + 3 * AH, half);
  const f32x4* bW1f = S.bW1f + lane;
  const f32x4* pW1f = S.pW1f + lane;
  const f32x4* W2f = S.W2f + lane;
  const f32x4* W2tf = S.W2tf + lane;

  // ---- forward at the interpolates (masks) and the input gradients with dOut = 1
  f32x4 xb[A_CH], xn[A_CH], xc[A_CH];
  gp_mix(b0, b1, w, Cb, S.bmean, S.bistd, half, xb);
  gp_mix(n0, n1, w, Cp, S.pmean, S.pistd, half, xn);
  gp_mix(c0, c1, w, Cp, S.pmean, S.pistd, half, xc);
  const f32x16 hb = gp_first(xb, Cb, bW1f, S.vec, half, true);
  const f32x16 h1n = gp_first(xn, Cp, pW1f, S.vec + 2 * AH, half, true);
  const f32x16 h1c = gp_first(xc, Cp, pW1f, S.vec + 2 * AH, half, true);
  const f32x16 h2n = chain32(p_b2, h1n, W2f), h2c = chain32(p_b2, h1c, W2f);   // (pre-activations: only the sign is used)
  f32x16 u1b, u2n, u2c, z;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    u1b[j] = hb[j] > 0.f ? b_wout[j] : 0.f;
    u2n[j] = h2n[j] > 0.f ? p_wout[j] : 0.f;
    u2c[j] = h2c[j] > 0.f ? p_wout[j] : 0.f;
    z[j] = 0.f;
  }
  f32x16 u1n = chain32(z, u2n, W2tf), u1c = chain32(z, u2c, W2tf);
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    u1n[j] = h1n[j] > 0.f ? u1n[j] : 0.f;
    u1c[j] = h1c[j] > 0.f ? u1c[j] : 0.f;
  }
  f32x16 gb[2], gnx[2], gc[2];
  gp_input_grad(u1b, S.bW1Tf + lane, Tb, gb);
  gp_input_grad(u1n, S.pW1Tf + lane, Tp, gnx);
  gp_input_grad(u1c, S.pW1Tf + lane, Tp, gc);
  // G = gn / sigma (gradient w.r.t. the raw input); register (t, j) of a lane is column 32 t + 8 (j / 4) + 4 half + j % 4
  auto colof = [&](int t, int j) { return 32 * t + 8 * (j >> 2) + 4 * half + (j & 3); };
  float* Gb = S.tiles[wave][0] + lrow * GP_TS;
  float* Gn = S.tiles[wave][1] + lrow * GP_TS;
  float* Gc = S.tiles[wave][2] + lrow * GP_TS;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c0_ = 32 * t + 8 * q + 4 * half;
      f32x4 vb, vn, vc;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        gb[t][4 * q + u] *= S.bistd[c0_ + u];
        gnx[t][4 * q + u] *= S.pistd[c0_ + u];
        gc[t][4 * q + u] *= S.pistd[c0_ + u];
        vb[u] = gb[t][4 * q + u];
        vn[u] = gnx[t][4 * q + u];
        vc[u] = gc[t][4 * q + u];
      }
      *reinterpret_cast<f32x4*>(Gb + c0_) = vb;
      *reinterpret_cast<f32x4*>(Gn + c0_) = vn;
      *reinterpret_cast<f32x4*>(Gc + c0_) = vc;
    }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();

  // ---- rows: T, |T|, coefficients. s' and s blocks are summed in the potential's column layout (own registers of
  // gnx / gc + the base's entry from the tile), the action / done blocks in the base's.
  const int od = a.od, ad = a.ad;
  const int o_a = a.use_state ? od : 0, o_n = o_a + (a.use_action ? ad : 0), o_d = o_n + (a.use_next ? od : 0);
  const float cdisc = a.gamma * (1.f - dhat);
  f32x16 Ts[2], Tn[2];
  float sq = 0.f;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int col = colof(t, j);
      const bool on = col < od;
      const float ts = on ? (a.use_state ? Gb[min(col, A_D_MAX - 1)] : 0.f) - gc[t][j] : 0.f;
      const float tn = on ? (a.use_next ? Gb[min(o_n + col, A_D_MAX - 1)] : 0.f) + cdisc * gnx[t][j] : 0.f;
      Ts[t][j] = ts;
      Tn[t][j] = tn;
      sq += ts * ts + tn * tn;
      const bool in_a = a.use_action && col >= o_a && col < o_a + ad;
      const bool in_d = a.use_done && col == o_d;
      const float tb = (in_a || in_d) ? gb[t][j] : 0.f;
      sq += tb * tb;
    }
  sq += __shfl_xor(sq, 32, 64);
  const float nrm = sqrtf(sq);
  const float kq = (live && nrm > 0.f) ? a.coef / (float)a.B * 2.f * (nrm - a.target) / nrm : 0.f;
  const float pen_row = (live && half == 0) ? (nrm - a.target) * (nrm - a.target) : 0.f;
  // Cn in each stack's chunk layout (= the B operand of the second pass)
  f32x16 Cnb[2], Cnn[2], Cnc[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int col = colof(t, j);
      Cnn[t][j] = cdisc * kq * Tn[t][j] * S.pistd[col];
      Cnc[t][j] = -kq * Ts[t][j] * S.pistd[col];
      float tb;
      if (a.use_state && col < od) tb = gb[t][j] - Gc[col];
      else if (a.use_next && col >= o_n && col < o_n + od) tb = gb[t][j] + cdisc * Gn[min(max(col - o_n, 0), A_D_MAX - 1)];
      else tb = gb[t][j];
      Cnb[t][j] = col < a.Db ? kq * tb * S.bistd[col] : 0.f;
    }

  // ---- second pass
  auto store_chunks = [&](float* dst, int ld, const f32x16 (&v)[2]) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int k0 = 32 * t + 8 * q + 4 * half;
        if (k0 < ld) {
          f32x4 o;
#pragma unroll
          for (int u = 0; u < 4; ++u) o[u] = v[t][4 * q + u];
          *reinterpret_cast<f32x4*>(dst + k0) = o;
        }
      }
  };
  auto times_W1 = [&](const f32x16 (&cn)[2], const f32x4* wf, int chunks) {   // W1 . Cn  (32 features per row)
    f32x16 acc = z;
#pragma unroll
    for (int q = 0; q < A_CH; ++q)
      if (q < chunks) {
        const f32x4 wv = wf[q * 64];
#pragma unroll
        for (int u = 0; u < 4; ++u) acc = mfma(wv[u], cn[q >> 2][4 * (q & 3) + u], acc);
      }
    return acc;
  };
  f32x16 dV1b = times_W1(Cnb, bW1f, Cb);
  f32x16 dV1n = times_W1(Cnn, pW1f, Cp), dV1c = times_W1(Cnc, pW1f, Cp);
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    dV1b[j] = hb[j] > 0.f ? dV1b[j] : 0.f;
    dV1n[j] = h1n[j] > 0.f ? dV1n[j] : 0.f;
    dV1c[j] = h1c[j] > 0.f ? dV1c[j] : 0.f;
  }
  f32x16 dV2n = chain32(z, dV1n, W2f), dV2c = chain32(z, dV1c, W2f);
  f32x16 gwp, sc;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    gwp[j] = (h2n[j] > 0.f ? dV2n[j] : 0.f) + (h2c[j] > 0.f ? dV2c[j] : 0.f);
    sc[j] = 0.f;
  }
  sc[0] = pen_row;
  if (live) {
    store_features(a.U1b + (long long)rr * AH, u1b, half);
    store_chunks(a.Cb + (long long)rr * a.ldcb, a.ldcb, Cnb);
    store_features(a.U1p + (long long)rr * AH, u1n, half);
    store_features(a.U1p + (long long)(a.B + rr) * AH, u1c, half);
    store_chunks(a.Cp + (long long)rr * a.ldcp, a.ldcp, Cnn);
    store_chunks(a.Cp + (long long)(a.B + rr) * a.ldcp, a.ldcp, Cnc);
    store_features(a.U2p + (long long)rr * AH, u2n, half);
    store_features(a.U2p + (long long)(a.B + rr) * AH, u2c, half);
    store_features(a.V1p + (long long)rr * AH, dV1n, half);
    store_features(a.V1p + (long long)(a.B + rr) * AH, dV1c, half);
  }
  // (rows past the end: kq = 0, so every Cn and dV of theirs is zero)
  float s_b, s_p, s_s;
  half_sums16x3(dV1b, gwp, sc, lane, s_b, s_p, s_s);
  {
    const int j = ((lane >> 4) & 1) * 8 + ((lane >> 3) & 1) * 4 + ((lane >> 2) & 1) * 2 + ((lane >> 1) & 1);
    const int feat = 8 * (j >> 2) + 4 * half + (j & 3);
    if ((lane & 1) == 0) {
      S.red[wave][feat] = s_b;
      S.red[wave][AH + feat] = s_p;
      if (half == 0 && j == 0) S.red[wave][2 * AH] = s_s;
    }
  }
  __syncthreads();
  if (tid < 2 * AH + 1) {
    float tsum = S.red[0][tid];
#pragma unroll
    for (int wv = 1; wv < A_WAVES; ++wv) tsum += S.red[wv][tid];
    float* slab = a.part + (long long)bid * a.part_stride;
    if (tid < AH) slab[a.off_b_wout + tid] = tsum;
    else if (tid < 2 * AH) slab[a.off_p_wout + tid - AH] = tsum;
    else __hip_atomic_store(a.pen_part + bid, tsum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  // (hand-off as in airl_rows_kernel: the one value that crosses workgroups goes out write-through, no L2 write-back)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    const unsigned tk = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tk == nblocks - 1) {
      float tsum = 0.f;
      for (unsigned b = 0; b < nblocks; ++b)
        tsum += __hip_atomic_load(a.pen_part + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      a.pen_out[0] = tsum / (float)a.B;
      __hip_atomic_store(a.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

__global__ __launch_bounds__(A_THREADS) void airl_gp_rows_kernel(AirlGpArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char gp_raw[];
  airl_gp_rows_body(a, *reinterpret_cast<GpLds*>(gp_raw), blockIdx.x, gridDim.x);
}

// The update's rows AND its gradient penalty's rows in ONE launch: the penalty's pass reads the assembled batches, the frozen
// statistics and the parameters -- nothing the update's row pass writes -- so the two passes (64 workgroups each at
// 8 192 + 8 192 rows: a quarter of the chip, 21 and 42 us) run side by side instead of one behind the other. The penalty's
// workgroups, the longer ones, take the first block ids.
__global__ __launch_bounds__(A_THREADS) void airl_rows_gp_kernel(AirlArgs a, AirlGpArgs g, int nb_gp, long long* __restrict__ ts) {
  extern __shared__ __attribute__((aligned(16))) unsigned char both_raw[];
  if ((int)blockIdx.x < nb_gp) airl_gp_rows_body(g, *reinterpret_cast<GpLds*>(both_raw), blockIdx.x, nb_gp);
  else airl_rows_body(a, ts, *reinterpret_cast<AirlLds*>(both_raw), blockIdx.x - nb_gp, gridDim.x - nb_gp);
}


// ---- GAIL with the reference's DEFAULT discriminator: BasicRewardNet D -> 32 -> 32 -> 1 (ReLU) --------------------------------
// rewards/reward_nets.py:394-397,430-457 (hid_sizes = (32, 32): what every tuned GAIL config of the reference trains) on
// the same register-resident transposed chains as the potential stack above. One launch does the whole minibatch of
// adversarial/common.py:352-373 short of the optimiser step: normalise (statistics as given), forward, BCE-with-logits +
// the statistics of compute_train_stats, deltas -- and the WEIGHT GRADIENTS too: the workgroup's 128 rows of
// (x_n, h1, delta1, delta2) go through LDS tiles, then wave 0 contracts delta2^T . h1 (dW2), waves 1 / 2 delta1^T . x_n
// (dW1, 32 input columns each) over the 128 rows on the matrix cores and wave 3 takes the bias / output-layer sums. Each
// workgroup leaves one slab of the flat gradient layout; ia_reduce_partials(_adam) finishes. Two launches per update.
struct Disc32Args {
  const float* X; int ldx, D, R, n_expert;
  const float *mean, *var; float eps;     // null: no input normalisation
  const float* P;                         // W1[32 x D] b1[32] W2[32 x 32] b2[32] W3[32] b3
  float scale;
  float* part; long long pstride;         // [workgroups][pstride] gradient slabs, torch parameter order
  float *logits, *dlogits, *stats, *bce_part;
  unsigned* ticket;
  float* pred_out; int out_act;           // prediction (ia_disc32_predict): out_act(logit) of every row is all that is wanted
};

constexpr int D32_TX = 65, D32_TH = 33;   // LDS row strides (odd: conflict-free column walks)
struct Disc32Lds {
  f32x4 W1f[A_CH * 64], W2f[4 * 64], W2tf[4 * 64];
  float vec[3 * AH + 4];                  // b1, b2, w3, then b3
  float mean[A_D_MAX], istd[A_D_MAX];
  float red[A_WAVES][40];                 // per wave: 32 dW3 partial sums + 8 scalars
  float Tx[A_ROWS * D32_TX];              // normalised inputs of the workgroup's rows
  float Th1[A_ROWS * D32_TH], Td1[A_ROWS * D32_TH], Td2[A_ROWS * D32_TH];
  int is_last;
};

__device__ __forceinline__ void store_features_lds(float* __restrict__ row, const f32x16& v, int half) {
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int u = 0; u < 4; ++u) row[8 * q + 4 * half + u] = v[4 * q + u];
}

// C[m][n] = sum over the workgroup's 128 rows of TA[row][m] * TB[row][n]: one wave, 64 k-steps of v_mfma_f32_32x32x2_f32
// (A[m][k] from lane (m, k & 1), B[k][n] from lane (n, k & 1)); the 16 operand pairs of a quarter are requested together.
__device__ __forceinline__ f32x16 rows_outer(const float* __restrict__ TA, int lda, const float* __restrict__ TB, int ldb,
                                             int lane) {
  f32x16 acc;
#pragma unroll
  for (int j = 0; j < 16; ++j) acc[j] = 0.f;
  const int c = lane & 31, h = lane >> 5;
#pragma unroll
  for (int s0 = 0; s0 < A_ROWS / 2; s0 += 16) {
    float av[16], bv[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int row = 2 * (s0 + s) + h;
      av[s] = TA[row * lda + c];
      bv[s] = TB[row * ldb + c];
    }
#pragma unroll
    for (int s = 0; s < 16; ++s) acc = mfma(av[s], bv[s], acc);
  }
  return acc;
}

__global__ __launch_bounds__(A_THREADS) void disc32_rows_kernel(Disc32Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char d32_smem[];
  Disc32Lds& S = *reinterpret_cast<Disc32Lds*>(d32_smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5;
  const int D = a.D, C = (D + 7) >> 3;
  const int rl = wave * 32 + (lane & 31);           // row inside the workgroup
  const int r = blockIdx.x * A_ROWS + rl;
  const bool live = r < a.R;
  const int rr = live ? r : a.R - 1;                // (rows past the end recompute the last one with a zero delta)
  f32x4 raw[A_CH];
  load_row(a.X + (long long)rr * a.ldx, a.ldx, C, half, raw);

  const float* P = a.P;
  const int o_b1 = AH * D, o_W2 = o_b1 + AH, o_b2 = o_W2 + AH * AH, o_w3 = o_b2 + AH, o_b3 = o_w3 + AH;
  {
    // weight fragments (entry (q, lane) = A[m = lane & 31][k = 8 q + 4 (lane >> 5) + u]), vectors and statistics: every
    // global load of a thread is issued before its first LDS store (clamped addresses, masked afterwards)
    constexpr int NE = A_CH * 64 / A_THREADS;
    f32x4 w1[NE], w2, w2t;
#pragma unroll
    for (int it = 0; it < NE; ++it) {
      const int e = tid + it * A_THREADS;
      const int m = e & 31, k0 = 8 * (e >> 6) + 4 * ((e >> 5) & 1);
#pragma unroll
      for (int u = 0; u < 4; ++u) w1[it][u] = P[m * D + min(k0 + u, D - 1)];
    }
    {
      const int m = tid & 31, k0 = 8 * (tid >> 6) + 4 * ((tid >> 5) & 1);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        w2[u] = P[o_W2 + m * AH + k0 + u];            // forward:  A[m = out][k = in]
        w2t[u] = P[o_W2 + (k0 + u) * AH + m];         // backward: A[m = in][k = out]
      }
    }
    const int vt = min(tid, AH - 1), kt = min(tid, D - 1);
    const float v0 = P[o_b1 + vt], v1 = P[o_b2 + vt], v2 = P[o_w3 + vt], vb = P[o_b3];
    const bool hn = a.mean != nullptr;
    const float sm = hn ? a.mean[kt] : 0.f, sv = hn ? a.var[kt] : 1.f;
#pragma unroll
    for (int it = 0; it < NE; ++it) {
      const int e = tid + it * A_THREADS;
      const int k0 = 8 * (e >> 6) + 4 * ((e >> 5) & 1);
#pragma unroll
      for (int u = 0; u < 4; ++u) w1[it][u] = k0 + u < D ? w1[it][u] : 0.f;
      S.W1f[e] = w1[it];
    }
    S.W2f[tid] = w2;
    S.W2tf[tid] = w2t;
    if (tid < AH) {
      S.vec[tid] = v0;
      S.vec[AH + tid] = v1;
      S.vec[2 * AH + tid] = v2;
    }
    if (tid == 0) S.vec[3 * AH] = vb;
    if (tid < A_D_MAX) {
      const bool in = tid < D;
      S.mean[tid] = in ? sm : 0.f;
      S.istd[tid] = in ? (hn ? 1.f / sqrtf(sv + a.eps) : 1.f) : 0.f;   // (x - mean) / sqrt(var + eps), networks.py:91
    }
  }
  __syncthreads();

  const f32x16 w3 = per_feature(S.vec + 2 * AH, half), b2 = per_feature(S.vec + AH, half);
  const f32x4* W1f = S.W1f + lane;
  const f32x4* W2f = S.W2f + lane;
  const f32x4* W2tf = S.W2tf + lane;

  // forward: h1 = relu(b1 + W1 xn) (xn into the LDS tile on the way), h2 = relu(b2 + W2 h1), logit = b3 + w3 . h2
  f32x16 h1 = per_feature(S.vec, half);
  {
    float* xrow = S.Tx + rl * D32_TX;
#pragma unroll
    for (int q = 0; q < A_CH; ++q) {
      if (q < C) {
        const int k0 = 8 * q + 4 * half;
        const f32x4 m = *reinterpret_cast<const f32x4*>(S.mean + k0);
        const f32x4 is = *reinterpret_cast<const f32x4*>(S.istd + k0);
        const f32x4 w = W1f[q * 64];
        const f32x4 xn = (raw[q] - m) * is;          // (columns past the input width: istd = 0)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          xrow[k0 + u] = xn[u];
          h1 = mfma(w[u], xn[u], h1);
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 16; ++j) h1[j] = fmaxf(h1[j], 0.f);
  f32x16 h2 = chain32(b2, h1, W2f);
#pragma unroll
  for (int j = 0; j < 16; ++j) h2[j] = fmaxf(h2[j], 0.f);
  const float x = S.vec[3 * AH] + dot_features(h2, w3);
  if (a.pred_out != nullptr) {   // (kernel argument: uniform) `RewardNet.predict_th` of the rows, rewards/reward_nets.py:176-204
    if (live && half == 0) a.pred_out[r] = ia_apply_act(x, a.out_act);
    return;
  }

  // BCE-with-logits, its gradient and the statistics (adversarial/common.py:27-92, 360-368; bce_kernel's expressions)
  const float y = rr < a.n_expert ? 1.f : 0.f;
  const float lse = log1pf(expf(-fabsf(x)));
  const float p = 1.f / (1.f + expf(-x));
  const float dlog = live ? (p - y) * (a.scale / (float)a.R) : 0.f;
  f32x16 sc;
#pragma unroll
  for (int j = 0; j < 16; ++j) sc[j] = 0.f;
  if (live) {
    if (half == 0) {
      a.logits[r] = x;
      if (a.dlogits) a.dlogits[r] = dlog;
    }
    const bool is_gen_pred = x < 0.f, is_gen_true = y == 0.f, ok = is_gen_pred == is_gen_true;
    sc[0] = (1.f - y) * x - (fminf(x, 0.f) - lse);
    sc[1] = ok ? 1.f : 0.f;
    sc[2] = (ok && !is_gen_true) ? 1.f : 0.f;
    sc[3] = (ok && is_gen_true) ? 1.f : 0.f;
    sc[4] = is_gen_pred ? 1.f : 0.f;
    sc[5] = (1.f - p) * x - (fminf(x, 0.f) - lse);
    sc[6] = dlog;                                  // d b3
  }

  // backward: delta2 = relu'(h2) dlog w3, delta1 = relu'(h1) (W2^T delta2); output-layer gradient rows dlog * h2
  f32x16 d2, g3, z;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    g3[j] = dlog * h2[j];
    d2[j] = h2[j] > 0.f ? dlog * w3[j] : 0.f;
    z[j] = 0.f;
  }
  f32x16 d1 = chain32(z, d2, W2tf);
#pragma unroll
  for (int j = 0; j < 16; ++j) d1[j] = h1[j] > 0.f ? d1[j] : 0.f;
  store_features_lds(S.Th1 + rl * D32_TH, h1, half);
  store_features_lds(S.Td1 + rl * D32_TH, d1, half);
  store_features_lds(S.Td2 + rl * D32_TH, d2, half);
  {
    const float s_g = half_sums16(g3, lane), s_s = half_sums16(sc, lane);
    const int j = ((lane >> 4) & 1) * 8 + ((lane >> 3) & 1) * 4 + ((lane >> 2) & 1) * 2 + ((lane >> 1) & 1);
    const int feat = 8 * (j >> 2) + 4 * half + (j & 3);
    if ((lane & 1) == 0) {
      S.red[wave][feat] = s_g;
      if (half == 0 && j < 8) S.red[wave][AH + j] = s_s;
    }
  }
  __syncthreads();

  // weight gradients of the workgroup's 128 rows, one role per wave; slab = [W1 | b1 | W2 | b2 | W3 | b3]
  float* slab = a.part + (long long)blockIdx.x * a.pstride;
  if (wave == 0) {
    const f32x16 g = rows_outer(S.Td2, D32_TH, S.Th1, D32_TH, lane);
#pragma unroll
    for (int j = 0; j < 16; ++j) slab[o_W2 + (8 * (j >> 2) + 4 * half + (j & 3)) * AH + (lane & 31)] = g[j];
  } else if (wave == 1 || (wave == 2 && D > 32)) {
    const int c0 = (wave - 1) * 32, c = c0 + (lane & 31);
    const f32x16 g = rows_outer(S.Td1, D32_TH, S.Tx + c0, D32_TX, lane);
    if (c < D) {
#pragma unroll
      for (int j = 0; j < 16; ++j) slab[(8 * (j >> 2) + 4 * half + (j & 3)) * D + c] = g[j];
    }
  } else if (wave == 3) {
    // bias gradients = column sums of the delta tiles (lanes 0..31: b2 from delta2, lanes 32..63: b1 from delta1)
    const float* T = (half ? S.Td1 : S.Td2) + (lane & 31);
    float s = 0.f;
#pragma unroll 16
    for (int row = 0; row < A_ROWS; ++row) s += T[row * D32_TH];
    slab[(half ? o_b1 : o_b2) + (lane & 31)] = s;
    if (lane < 40) {
      float t = S.red[0][lane];
#pragma unroll
      for (int w = 1; w < A_WAVES; ++w) t += S.red[w][lane];
      if (lane < AH) slab[o_w3 + lane] = t;
      else if (lane < AH + 6)   // (written THROUGH to memory: the hand-off below then needs no L2 write-back)
        __hip_atomic_store(a.bce_part + blockIdx.x * 8 + lane - AH, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      else if (lane == AH + 6) slab[o_b3] = t;
    }
  }
  // statistics: the workgroup that draws the last ticket folds the partials (the hand-off of airl_rows_kernel)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    const unsigned tk = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    S.is_last = (tk == gridDim.x - 1);
  }
  __syncthreads();
  if (!S.is_last) return;
  {
    const int k = tid >> 5, j = tid & 31;
    float t = 0.f;
    if (k < 6)
      for (unsigned b = j; b < gridDim.x; b += 32)
        t += __hip_atomic_load(a.bce_part + b * 8 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    float* fold = S.Tx;                    // (the tiles were consumed before the last barriers)
    if (k < 8) fold[k * 32 + j] = t;
    __syncthreads();
    if (tid < 6) {
      float tsum = 0.f;
      for (int q = 0; q < 32; ++q) tsum += fold[tid * 32 + q];
      if (tid == 0) tsum = tsum / (float)a.R * a.scale;
      a.stats[tid] = tsum;
    }
  }
  if (tid == 6) a.stats[6] = (float)a.n_expert;
  if (tid == 7) a.stats[7] = (float)(a.R - a.n_expert);
  if (tid == 0) __hip_atomic_store(a.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}


// ---- host side: argument blocks of the two row passes and of their split-K weight-gradient products --------------------------
struct AirlStepPlan { AirlArgs a; IaGemm ws3[3]; int nblk, nb; long long ptot; };
int airl_plan_step(const float* Xb, int ldb, int Db, const float* Sn, const float* Sc, int ldp, int Dp, const float* dones,
                   const float* logp, const float* bmean, const float* bvar, float beps, const float* pmeanA,
                   const float* pvarA, const float* pmeanB, const float* pvarB, float peps, const float* params_base,
                   const float* params_pot, float gamma, float scale, int R, int n_expert, float* Ab, int ldab, float* Db1,
                   float* Ap, int ldap, float* H1, float* Dp1, float* Dp2, float* partials, float* logits, float* stats,
                   float* bce_part, unsigned* ticket, AirlStepPlan& pl) {
  if (!Xb || !Sn || !Sc || !dones || !logp || !params_base || !params_pot || !Ab || !Db1 || !Ap || !H1 || !Dp1 || !Dp2 ||
      !partials || !logits || !stats || !bce_part || !ticket || R <= 0 || n_expert < 0 || n_expert > R)
    return IA_ERR_ARG;
  if (!ia_airl_fused_ok(Db, Dp, AH, AH, AH) || ldb % 4 || ldp % 4 || ldab % 4 || ldap % 4 || ldb < Db || ldp < Dp ||
      ldab < ((Db + 3) & ~3) || ldap < ((Dp + 3) & ~3))
    return IA_ERR_UNSUPPORTED;
  const int nb = AH * Db + AH + AH + 1, np = AH * Dp + AH + AH * AH + AH + AH + 1;
  const long long ptot = (long long)nb + np;
  const int nblk = ia_airl_fused_slabs(R);
  AirlArgs a{};
  a.Xb = Xb; a.Sn = Sn; a.Sc = Sc; a.ldb = ldb; a.Db = Db; a.ldp = ldp; a.Dp = Dp; a.dones = dones; a.logp = logp;
  a.bmean = bmean; a.bvar = bvar; a.pmeanA = pmeanA; a.pvarA = pvarA; a.pmeanB = pmeanB; a.pvarB = pvarB;
  a.beps = beps; a.peps = peps; a.Pb = params_base; a.Pp = params_pot; a.gamma = gamma; a.scale = scale; a.R = R;
  a.n_expert = n_expert; a.Ab = Ab; a.ldab = ldab; a.Db1 = Db1; a.Ap = Ap; a.ldap = ldap; a.H1 = H1; a.Dp1 = Dp1; a.Dp2 = Dp2;
  a.part = partials; a.part_stride = ptot;
  a.off_b_wout = AH * Db + AH; a.off_b_bout = a.off_b_wout + AH;
  a.off_p_wout = nb + AH * Dp + AH + AH * AH + AH; a.off_p_bout = a.off_p_wout + AH;
  a.logits = logits; a.stats = stats; a.bce_part = bce_part; a.ticket = ticket;
  // hidden-layer weight gradients: dW = delta^T . input over the rows, one split-K slab per 128 (256) rows; bias
  // gradients = column sums of delta. Slab s of `partials` is what workgroup s of the rows kernel wrote into.
  auto wgrad = [&](const float* delta, const float* in, int ldin, int N, int K, long long w_off, long long b_off) {
    IaGemm w{};
    w.A = delta; w.lda = AH; w.B = in; w.ldb = ldin; w.M = AH; w.N = N; w.K = K;
    w.C = partials + w_off; w.ldc = N; w.splits = nblk; w.k_per_split = ((K + nblk - 1) / nblk + 31) / 32 * 32;
    w.c_split_stride = ptot; w.dbias = partials + b_off; w.dbias_split_stride = ptot;
    return w;
  };
  pl.a = a; pl.nblk = nblk; pl.nb = nb; pl.ptot = ptot;
  pl.ws3[0] = wgrad(Db1, Ab, ldab, Db, R, 0, (long long)AH * Db);
  pl.ws3[1] = wgrad(Dp1, Ap, ldap, Dp, 2 * R, nb, (long long)nb + AH * Dp);
  pl.ws3[2] = wgrad(Dp2, H1, AH, AH, 2 * R, (long long)nb + AH * Dp + AH, (long long)nb + AH * Dp + AH + AH * AH);
  return IA_OK;
}

struct AirlGpPlan { AirlGpArgs a; IaGemm gs3[3]; int nblk; long long ptot; };
int airl_plan_gp(const float* Xb, int ldb, int Db, const float* Sn, const float* Sc, int ldp, int Dp, const float* dones,
                 const float* e, const float* bmean, const float* bvar, float beps, const float* pmean, const float* pvar,
                 float peps, const float* params_base, const float* params_pot, int obs_dim, int act_dim, int use_state,
                 int use_action, int use_next_state, int use_done, float gamma, float coef, float target, int B, float* U1b,
                 float* Cb, float* U1p, float* Cp, float* U2p, float* V1p, float* partials, float* pen_part, float* pen_out,
                 unsigned* ticket, AirlGpPlan& pl) {
  if (!Xb || !Sn || !Sc || !dones || !e || !params_base || !params_pot || !U1b || !Cb || !U1p || !Cp || !U2p || !V1p ||
      !partials || !pen_part || !pen_out || !ticket || B <= 0)
    return IA_ERR_ARG;
  const int Dchk = (use_state ? obs_dim : 0) + (use_action ? act_dim : 0) + (use_next_state ? obs_dim : 0) + (use_done ? 1 : 0);
  if (!ia_airl_fused_ok(Db, Dp, AH, AH, AH) || Dchk != Db || Dp != obs_dim || ldb % 4 || ldp % 4 || ldb < Db || ldp < Dp)
    return IA_ERR_UNSUPPORTED;
  const int nb = AH * Db + AH + AH + 1, np = AH * Dp + AH + AH * AH + AH + AH + 1;
  const long long ptot = (long long)nb + np;
  const int nblk = ia_airl_fused_slabs(B);
  AirlGpArgs a{};
  a.Xb = Xb; a.Sn = Sn; a.Sc = Sc; a.ldb = ldb; a.Db = Db; a.ldp = ldp; a.Dp = Dp; a.dones = dones; a.e = e;
  a.bmean = bmean; a.bvar = bvar; a.pmean = pmean; a.pvar = pvar; a.beps = beps; a.peps = peps; a.Pb = params_base;
  a.Pp = params_pot; a.od = obs_dim; a.ad = act_dim; a.use_state = use_state; a.use_action = use_action;
  a.use_next = use_next_state; a.use_done = use_done; a.gamma = gamma; a.coef = coef; a.target = target; a.B = B;
  a.U1b = U1b; a.Cb = Cb; a.ldcb = ldb; a.U1p = U1p; a.Cp = Cp; a.ldcp = ldp; a.U2p = U2p; a.V1p = V1p;
  a.part = partials; a.part_stride = ptot; a.off_b_wout = AH * Db + AH; a.off_p_wout = nb + AH * Dp + AH + AH * AH + AH;
  a.pen_part = pen_part; a.pen_out = pen_out; a.ticket = ticket;
  auto wgrad = [&](const float* u, const float* in, int ldin, int N, int K, long long w_off) {
    IaGemm g{};
    g.A = u; g.lda = AH; g.B = in; g.ldb = ldin; g.M = AH; g.N = N; g.K = K;
    g.C = partials + w_off; g.ldc = N; g.splits = nblk; g.k_per_split = ((K + nblk - 1) / nblk + 31) / 32 * 32;
    g.c_split_stride = ptot;
    return g;
  };
  pl.a = a; pl.nblk = nblk; pl.ptot = ptot;
  pl.gs3[0] = wgrad(U1b, Cb, ldb, Db, B, 0);
  pl.gs3[1] = wgrad(U1p, Cp, ldp, Dp, 2 * B, nb);
  pl.gs3[2] = wgrad(U2p, V1p, AH, AH, 2 * B, (long long)nb + AH * Dp + AH);
  return IA_OK;
}

int airl_gp_lds_attr() {   // the penalty's row pass (alone, or sharing a launch with the update's) needs > 64 KB of LDS
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(airl_gp_rows_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)sizeof(GpLds)) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(airl_rows_gp_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)(sizeof(GpLds) > sizeof(AirlLds) ? sizeof(GpLds) : sizeof(AirlLds))) != hipSuccess)
      return IA_ERR_ARG;
    attr = true;
  }
  return IA_OK;
}
}  // namespace

extern "C" {

/* Debug / measurement: shader-clock stamps of the row kernel's phases (workgroup 0) into `buf` (>= 16 int64; null: off). */
int ia_airl_debug_timing(long long* buf) { g_airl_tstamp = buf; return IA_OK; }

/* Geometry the fused AIRL update covers: reward MLP Db -> 32 -> 1, potential MLP Dp -> 32 -> 32 -> 1 (ReLU). */
int ia_airl_fused_ok(int Db, int Dp, int hb, int hp1, int hp2) {
  return (hb == AH && hp1 == AH && hp2 == AH && Db >= 1 && Db <= A_D_MAX && Dp >= 1 && Dp <= A_D_MAX) ? 1 : 0;
}

/* Slabs of the split-K partial buffer (= workgroups of the rows kernel) an update of R rows uses. */
int ia_airl_fused_slabs(int R) { return R > 0 ? (R + A_ROWS - 1) / A_ROWS : 0; }

/* One discriminator update, device half: rows kernel + the three hidden-layer weight-gradient GEMMs; with `adam`
 * given (parameters of the two stacks contiguous, base first), also the split-K reduction fused with the Adam step
 * (ia_reduce_partials_adam) -- otherwise the caller
 * reduces `partials` ([ia_airl_fused_slabs(R)][n_params], base stack's parameters first) -- e.g. ia_reduce_partials_adam.
 * Inputs: assembled raw batches Xb[R, ldb] / Sn, Sc[R, ldp] ([expert | generator] rows), dones, log pi; statistics
 * as the train-mode updates left them (nullable: no input norm); flat parameters of the two stacks. Workspaces:
 * Ab[R, ldab], Db1[R, 32], Ap[2R, ldap], H1 / Dp1 / Dp2 [2R, 32], bce_part [slabs * 8], ticket (zeroed once). */
int ia_airl_step_shaped(const float* Xb, int ldb, int Db, const float* Sn, const float* Sc, int ldp, int Dp,
                        const float* dones, const float* logp, const float* bmean, const float* bvar, float beps,
                        const float* pmeanA, const float* pvarA, const float* pmeanB, const float* pvarB, float peps,
                        const float* params_base, const float* params_pot, float gamma, float scale, int R, int n_expert,
                        float* Ab, int ldab, float* Db1, float* Ap, int ldap, float* H1, float* Dp1, float* Dp2,
                        float* partials, float* logits, float* stats, float* bce_part, unsigned* ticket,
                        const ia_adam_args* adam, void* stream) {
  AirlStepPlan pl;
  int rc = airl_plan_step(Xb, ldb, Db, Sn, Sc, ldp, Dp, dones, logp, bmean, bvar, beps, pmeanA, pvarA, pmeanB, pvarB, peps,
                          params_base, params_pot, gamma, scale, R, n_expert, Ab, ldab, Db1, Ap, ldap, H1, Dp1, Dp2, partials,
                          logits, stats, bce_part, ticket, pl);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(airl_rows_kernel, dim3(pl.nblk), dim3(A_THREADS), 0, st, pl.a, g_airl_tstamp);
  IA_CHECK_LAUNCH();
  // (one grouped launch: 3 x nblk workgroups together instead of three latency-bound launches of nblk each)
  rc = ia_launch_gemm_group_tn(pl.ws3, 3, st);
  if (rc || !adam) return rc;
  if (!adam->grads || !adam->exp_avg || !adam->exp_avg_sq || params_pot != params_base + pl.nb) return IA_ERR_ARG;
  return ia_reduce_partials_adam(partials, pl.nblk, pl.ptot, 1.0f, adam->grads, const_cast<float*>(params_base), adam->exp_avg,
                                 adam->exp_avg_sq, adam->beta1, adam->beta2, adam->eps, adam->weight_decay,
                                 adam->step_size, adam->bc2_sqrt, stream);
}

/* Batch assembly of one update in ONE launch (adversarial/common.py:564-603, rewards/reward_nets.py:441-457): rows
 * idx0 (n0 of them, expert) then idx1 (R - n0, generator) of the two transition tables -> Xb[R, ldb] ([state | action,
 * one-hot when act_i64 | next state | done] as flagged), Sn / Sc[R, ldp] (next observations, observations) and
 * dones[R]; with ws_* given, also the RunningNorm slab moments of each matrix ([ceil(R/256)][2][D], the layout and
 * arithmetic of ia_running_norm_partial); with pol_obs / pol_act given, the unpadded observation and action rows the
 * generator policy's log pi(a|s) reads. Padding columns of the outputs are not written. */
int ia_airl_prepare(const float* obs0, const float* act0_f32, const int64_t* act0_i64, const float* next0,
                    const uint8_t* done0, const int64_t* idx0, int n0, const float* obs1, const float* act1_f32,
                    const int64_t* act1_i64, const float* next1, const uint8_t* done1, const int64_t* idx1, int n1,
                    int obs_dim, int act_dim, int use_state, int use_action, int use_next_state, int use_done, float* Xb,
                    int ldb, float* Sn, float* Sc, int ldp, float* dones, float* ws_b, float* ws_n, float* ws_c,
                    float* pol_obs, float* pol_act, void* stream) {
  const int Db = (use_state ? obs_dim : 0) + (use_action ? act_dim : 0) + (use_next_state ? obs_dim : 0) + (use_done ? 1 : 0);
  if (!obs0 || !next0 || !done0 || !obs1 || !next1 || !done1 || n0 < 0 || n1 < 0 || n0 + n1 <= 0 || obs_dim <= 0 ||
      act_dim <= 0 || Db <= 0 || !Xb || !Sn || !Sc || !dones || ldb < Db || ldp < obs_dim)
    return IA_ERR_ARG;
  if ((use_action || pol_act) && ((!act0_f32 && !act0_i64) || (!act1_f32 && !act1_i64))) return IA_ERR_ARG;
  if ((act0_i64 != nullptr) != (act1_i64 != nullptr)) return IA_ERR_ARG;
  AirlPrep a{};
  a.idx0 = idx0; a.idx1 = idx1; a.n0 = n0; a.R = n0 + n1;
  const bool disc = act0_i64 != nullptr;
  const int od = obs_dim, ad = act_dim;
  int n = 0, col = 0;
  auto add = [&](const void* p0, const void* p1, int kind, int stride, int ncols, float* X, int ldx, int xcol, float* ws,
                 int wsD) { a.pass[n++] = AirlPass{p0, p1, kind, stride, 0, ncols, X, ldx, xcol, ws, wsD, 0, 0}; };
  if (use_state) { add(obs0, obs1, AP_F32, od, od, Xb, ldb, col, ws_b, Db); col += od; }
  if (use_action) {
    if (disc) add(act0_i64, act1_i64, AP_ONEHOT, 1, ad, Xb, ldb, col, ws_b, Db);
    else add(act0_f32, act1_f32, AP_F32, ad, ad, Xb, ldb, col, ws_b, Db);
    col += ad;
  }
  if (use_next_state) { add(next0, next1, AP_F32, od, od, Xb, ldb, col, ws_b, Db); col += od; }
  if (use_done) add(done0, done1, AP_DONE, 1, 1, Xb, ldb, col, ws_b, Db);
  add(next0, next1, AP_F32, od, od, Sn, ldp, 0, ws_n, od);
  add(obs0, obs1, AP_F32, od, od, Sc, ldp, 0, ws_c, od);
  add(done0, done1, AP_DONE, 1, 1, dones, 1, 0, nullptr, 0);
  if (pol_obs) add(obs0, obs1, AP_F32, od, od, pol_obs, od, 0, nullptr, 0);
  if (pol_act) {
    if (disc) add(act0_i64, act1_i64, AP_INDEX, 1, 1, pol_act, 1, 0, nullptr, 0);
    else add(act0_f32, act1_f32, AP_F32, ad, ad, pol_act, ad, 0, nullptr, 0);
  }
  a.n_pass = n;
  hipLaunchKernelGGL(airl_prepare_kernel, dim3((a.R + RN_ROWS_PER_BLOCK - 1) / RN_ROWS_PER_BLOCK, n), dim3(256), 0,
                     (hipStream_t)stream, a);
  IA_CHECK_LAUNCH();
  return IA_OK;
}

/* The train-mode RunningNorm updates of one shaped-net forward from ia_airl_prepare's slab moments, one launch: base
 * input norm (ws_b null: skipped); potential input norm (ws_n / ws_c null: skipped) with the next-state batch, its
 * (mean, var) copied to snapA[2][Dp], then with the state batch. `groups` > 1: the moments of that many data-parallel
 * ranks (R rows each, `group_stride` floats apart, 0 = back to back), merged in rank order. `ticket`: one zeroed word
 * (left zeroed). */
int ia_airl_stats_merge(const float* ws_b, const float* ws_n, const float* ws_c, int groups, int64_t group_stride, int R,
                        int Db, int Dp, float* bmean, float* bvar, int32_t* bcount, float* pmean, float* pvar,
                        int32_t* pcount, float* snapA, unsigned* ticket, void* stream) {
  if (R <= 0 || groups <= 0 || group_stride < 0 || !ticket || (!ws_b && !ws_n) || (ws_b && (!bmean || !bvar || !bcount || Db <= 0)) ||
      ((ws_n != nullptr) != (ws_c != nullptr)) || (ws_n && (!pmean || !pvar || !pcount || !snapA || Dp <= 0)))
    return IA_ERR_ARG;
  const int blocks = (ws_b ? Db : 0) + (ws_n ? Dp : 0);
  hipLaunchKernelGGL(airl_stats_merge_kernel, dim3(blocks), dim3(64), 0, (hipStream_t)stream, ws_b, ws_n, ws_c, groups,
                     (long long)group_stride, R, Db, Dp, bmean, bvar, bcount, pmean, pvar, pcount, snapA, ticket);
  IA_CHECK_LAUNCH();
  return IA_OK;
}

/* Gradient penalty of AIRL's shaped reward (OPT-IN extension, no reference counterpart; imitation_amd/grad_penalty.py:
 * coef * mean_i (|grad_(s,a,s',d) f|_2 - target)^2 at the interpolates e x_expert + (1-e) x_generator of the assembled
 * batches, statistics frozen) for the geometry of ia_airl_fused_ok, its parameter gradient ADDED to `grads`
 * ([base | potential] flat layout): one MFMA row kernel, three split-K weight-gradient GEMMs, one accumulate.
 * Workspaces: U1b[B,32], Cb[B,ldb], U1p / U2p / V1p[2B,32], Cp[2B,ldp], partials [ia_airl_fused_slabs(B)][n_params]
 * ZEROED once by the caller (bias entries are never written), pen_part[slabs], ticket (zeroed). pen_out[0] = the mean of
 * (|grad f| - target)^2. */
int ia_airl_gp_shaped(const float* Xb, int ldb, int Db, const float* Sn, const float* Sc, int ldp, int Dp,
                      const float* dones, const float* e, const float* bmean, const float* bvar, float beps,
                      const float* pmean, const float* pvar, float peps, const float* params_base,
                      const float* params_pot, int obs_dim, int act_dim, int use_state, int use_action,
                      int use_next_state, int use_done, float gamma, float coef, float target, int B, float* U1b,
                      float* Cb, float* U1p, float* Cp, float* U2p, float* V1p, float* partials, float* pen_part,
                      float* pen_out, unsigned* ticket, float* grads, void* stream) {
  if (!grads) return IA_ERR_ARG;
  AirlGpPlan pl;
  int rc = airl_plan_gp(Xb, ldb, Db, Sn, Sc, ldp, Dp, dones, e, bmean, bvar, beps, pmean, pvar, peps, params_base, params_pot,
                        obs_dim, act_dim, use_state, use_action, use_next_state, use_done, gamma, coef, target, B, U1b, Cb, U1p,
                        Cp, U2p, V1p, partials, pen_part, pen_out, ticket, pl);
  if (rc) return rc;
  if ((rc = airl_gp_lds_attr())) return rc;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(airl_gp_rows_kernel, dim3(pl.nblk), dim3(A_THREADS), sizeof(GpLds), st, pl.a);
  IA_CHECK_LAUNCH();
  rc = ia_launch_gemm_group_tn(pl.gs3, 3, st);
  if (rc) return rc;
  return ia_reduce_partials(partials, pl.nblk, pl.ptot, 1.0f, 1, grads, stream);
}

// A round's AIRL updates in ONE host call (`for _ in range(n_disc_updates_per_round): train_disc()`,
// adversarial/common.py:454-458): update k = the four entries an update is made of, with a[k]'s arguments, in order. What
// this removes is the per-update Python of the binding (four calls and their argument marshalling: ~150 us per update,
// which made the AIRL round host-bound).
int ia_airl_round(const ia_airl_update_args* a, int n, void* stream) {
  if (!a || n <= 0) return IA_ERR_ARG;
  for (int k = 0; k < n; ++k) {
    const ia_airl_update_args& u = a[k];
    const int R = u.n0 + u.n1;
    int rc = ia_airl_prepare(u.obs0, u.act0_f32, u.act0_i64, u.next0, u.done0, u.idx0, u.n0, u.obs1, u.act1_f32, u.act1_i64,
                             u.next1, u.done1, u.idx1, u.n1, u.obs_dim, u.act_dim, u.use_state, u.use_action, u.use_next_state,
                             u.use_done, u.Xb, u.ldb, u.Sn, u.Sc, u.ldp, u.dones, u.ws_b, u.ws_n, u.ws_c, u.pol_obs, u.pol_act,
                             stream);
    if (rc) return rc;
    if (u.ws_b || u.ws_n || u.ws_c) {
      rc = ia_airl_stats_merge(u.ws_b, u.ws_n, u.ws_c, 1, 0, R, u.Db, u.Dp, u.bmean, u.bvar, u.bcount, u.pmean, u.pvar,
                               u.pcount, u.snapA, u.merge_ticket, stream);
      if (rc) return rc;
    }
    rc = ia_policy_evaluate(u.pol, u.pol_params, u.pol_params_t, u.pol_norm_mean, u.pol_norm_var, u.pol_obs, u.pol_act, R,
                            u.logp, nullptr, nullptr, stream);
    if (rc) return rc;
    if (u.gp_e == nullptr) {
      rc = ia_airl_step_shaped(u.Xb, u.ldb, u.Db, u.Sn, u.Sc, u.ldp, u.Dp, u.dones, u.logp, u.f_bmean, u.f_bvar, u.beps, u.pmeanA,
                               u.pvarA, u.pmeanB, u.pvarB, u.peps, u.params_base, u.params_pot, u.gamma, u.scale, R, u.n_expert,
                               u.Ab, u.ldab, u.Db1, u.Ap, u.ldap, u.H1, u.Dp1, u.Dp2, u.partials, u.logits, u.stats, u.bce_part,
                               u.ticket, &u.adam, stream);
      if (rc) return rc;
      continue;
    }
    // With the gradient penalty (round 6): the update's row pass and the penalty's in ONE launch (independent: the penalty
    // reads the assembled batches, the statistics as the merge left them and the parameters), their six split-K
    // weight-gradient products in ONE grouped launch, and both slab sets reduced inside the optimiser step's launch --
    // three launches where there were seven (rows, products, reduce | penalty rows, products, reduce + add | Adam), the
    // same kernels' bodies on the same operands and the same order of every sum: bit-identical (tools/ppo_bits.py).
    if (u.n0 != u.n1) return IA_ERR_ARG;
    AirlStepPlan sp;
    AirlGpPlan gp;
    rc = airl_plan_step(u.Xb, u.ldb, u.Db, u.Sn, u.Sc, u.ldp, u.Dp, u.dones, u.logp, u.f_bmean, u.f_bvar, u.beps, u.pmeanA, u.pvarA,
                        u.pmeanB, u.pvarB, u.peps, u.params_base, u.params_pot, u.gamma, u.scale, R, u.n_expert, u.Ab, u.ldab,
                        u.Db1, u.Ap, u.ldap, u.H1, u.Dp1, u.Dp2, u.partials, u.logits, u.stats, u.bce_part, u.ticket, sp);
    if (rc) return rc;
    rc = airl_plan_gp(u.Xb, u.ldb, u.Db, u.Sn, u.Sc, u.ldp, u.Dp, u.dones, u.gp_e, u.f_bmean, u.f_bvar, u.beps, u.pmeanB, u.pvarB,
                      u.peps, u.params_base, u.params_pot, u.obs_dim, u.act_dim, u.use_state, u.use_action, u.use_next_state,
                      u.use_done, u.gamma, u.gp_coef, u.gp_target, u.n0, u.U1b, u.Cb, u.U1p, u.Cp, u.U2p, u.V1p, u.gp_partials,
                      u.pen_part, u.pen_out, u.gp_ticket, gp);
    if (rc) return rc;
    if (!u.adam.grads || !u.adam.exp_avg || !u.adam.exp_avg_sq || u.params_pot != u.params_base + sp.nb ||
        sp.nblk != u.n_slabs || sp.ptot != u.n_params)
      return IA_ERR_ARG;
    if ((rc = airl_gp_lds_attr())) return rc;
    hipStream_t st = (hipStream_t)stream;
    constexpr size_t lds_both = sizeof(GpLds) > sizeof(AirlLds) ? sizeof(GpLds) : sizeof(AirlLds);
    hipLaunchKernelGGL(airl_rows_gp_kernel, dim3(gp.nblk + sp.nblk), dim3(A_THREADS), lds_both, st, sp.a, gp.a, gp.nblk,
                       g_airl_tstamp);
    IA_CHECK_LAUNCH();
    const IaGemm six[6] = {gp.gs3[0], gp.gs3[1], gp.gs3[2], sp.ws3[0], sp.ws3[1], sp.ws3[2]};
    if ((rc = ia_launch_gemm_group_tn(six, 6, st))) return rc;
    rc = ia_reduce2_partials_adam(u.partials, sp.nblk, u.gp_partials, gp.nblk, sp.ptot, 1.0f, u.adam.grads,
                                  const_cast<float*>(u.params_base), u.adam.exp_avg, u.adam.exp_avg_sq, u.adam.beta1,
                                  u.adam.beta2, u.adam.eps, u.adam.weight_decay, u.adam.step_size, u.adam.bc2_sqrt, st);
    if (rc) return rc;
  }
  return IA_OK;
}

}  // extern "C"

// ---- the 32-wide BasicRewardNet behind the fused entry points of disc_fused.hip (ia_disc_fused_ws_floats,
//      ia_disc_assemble_round, ia_disc_step_basic -> ia_disc_step_fused, ia_disc_fused_adam dispatch here by shape) ----------
bool ia_disc32_shape_ok(const ia_mlp_desc* d, int ldx) {
  if (!d || d->n_layers != 3 || d->hidden_act != IA_ACT_RELU) return false;
  const int D = d->dims[0];
  return d->dims[1] == AH && d->dims[2] == AH && d->dims[3] == 1 && D >= 1 && D <= A_D_MAX && ldx >= D && ldx % 4 == 0 &&
         ldx <= A_D_MAX;
}

// Prediction on the row kernel (the reference's default 32 x 32 stack, up to 64 inputs): one launch, see ia_disc_fused_predict.
int ia_disc32_predict(const ia_mlp_desc* d, const float* params, const float* X, int ldx, int R, const float* mean,
                      const float* var, float eps, int out_act, float* out, hipStream_t stream) {
  if (!ia_disc32_shape_ok(d, ldx) || !params || !X || !out || R <= 0) return IA_ERR_ARG;
  Disc32Args k{};
  k.X = X; k.ldx = ldx; k.D = d->dims[0]; k.R = R; k.n_expert = 0;
  k.mean = mean; k.var = var; k.eps = eps;
  k.P = params; k.scale = 0.f;
  k.pred_out = out; k.out_act = out_act;
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(disc32_rows_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)sizeof(Disc32Lds)) != hipSuccess)
      return IA_ERR_ARG;
    attr = true;
  }
  hipLaunchKernelGGL(disc32_rows_kernel, dim3((R + A_ROWS - 1) / A_ROWS), dim3(A_THREADS), sizeof(Disc32Lds), stream, k);
  IA_CHECK_LAUNCH();
  return IA_OK;
}

namespace {
struct D32Ws { float* part; float* bce_part; unsigned* ticket; long long total; int nblk; long long P; };
inline D32Ws d32_layout(const ia_mlp_desc* d, int R, float* base) {
  D32Ws w;
  w.nblk = (R + A_ROWS - 1) / A_ROWS;
  w.P = (long long)AH * d->dims[0] + AH + AH * AH + AH + AH + 1;
  long long o = 0;
  w.part = base + o; o += (long long)w.nblk * w.P;
  w.bce_part = base + o; o += (long long)w.nblk * 8;
  w.ticket = reinterpret_cast<unsigned*>(base + o); o += 4;
  w.total = o;
  return w;
}
}  // namespace

int64_t ia_disc32_ws_floats(const ia_mlp_desc* d, int R) { return R > 0 ? d32_layout(d, R, nullptr).total : 0; }

// Batch assembly of `n_updates` updates in one launch (update k: index rows idx + k * idx_stride -> X + k * x_stride, slab
// moments -> rn_ws + k * rn_stride): adversarial/common.py:564-603 + rewards/reward_nets.py:441-457 through the pass kernel.
int ia_disc32_assemble(const ia_disc_step_args* a, int n_updates, int64_t idx_stride, int64_t x_stride, int64_t rn_stride,
                       float* rn_ws, hipStream_t stream) {
  const int od = a->obs_dim, ad = a->act_dim, D = a->desc->dims[0];
  const int Dchk = (a->use_state ? od : 0) + (a->use_action ? ad : 0) + (a->use_next_state ? od : 0) + (a->use_done ? 1 : 0);
  if (Dchk != D || a->n0 < 0 || a->n1 < 0 || a->n0 + a->n1 <= 0 || !a->X || n_updates <= 0) return IA_ERR_ARG;
  AirlPrep p{};
  p.idx0 = a->idx0; p.idx1 = a->idx1; p.idx_stride = idx_stride; p.n0 = a->n0; p.R = a->n0 + a->n1;
  // an empty side never selects its table: alias the other one so that no null pointer is formed into an address
  const bool e0 = a->n0 == 0, e1 = a->n1 == 0;
  const float *obs0 = e0 ? a->obs1 : a->obs0, *obs1 = e1 ? a->obs0 : a->obs1;
  const float *nx0 = e0 ? a->next1 : a->next0, *nx1 = e1 ? a->next0 : a->next1;
  const uint8_t *dn0 = e0 ? a->done1 : a->done0, *dn1 = e1 ? a->done0 : a->done1;
  const float *af0 = e0 ? a->act1_f32 : a->act0_f32, *af1 = e1 ? a->act0_f32 : a->act1_f32;
  const int64_t *ai0 = e0 ? a->act1_i64 : a->act0_i64, *ai1 = e1 ? a->act0_i64 : a->act1_i64;
  const bool disc = ai0 != nullptr;
  int n = 0, col = 0;
  auto add = [&](const void* p0, const void* p1, int kind, int stride, int ncols) {
    p.pass[n++] = AirlPass{p0, p1, kind, stride, 0, ncols, a->X, a->ldx, col, rn_ws, D, x_stride, rn_stride};
    col += ncols;
  };
  if (a->use_state) add(obs0, obs1, AP_F32, od, od);
  if (a->use_action) {
    if (disc) add(ai0, ai1, AP_ONEHOT, 1, ad);
    else add(af0, af1, AP_F32, ad, ad);
  }
  if (a->use_next_state) add(nx0, nx1, AP_F32, od, od);
  if (a->use_done) add(dn0, dn1, AP_DONE, 1, 1);
  p.n_pass = n;
  hipLaunchKernelGGL(airl_prepare_kernel, dim3((p.R + RN_ROWS_PER_BLOCK - 1) / RN_ROWS_PER_BLOCK, n, n_updates), dim3(256),
                     0, stream, p);
  IA_CHECK_LAUNCH();
  return IA_OK;
}

// Slab moments of the OBSERVATION columns of a round's batches in ONE launch: what the policy's train-mode feature
// RunningNorm absorbs when `train_disc` evaluates log pi(a|s) on a batch (common.py:606-615) depends on the
// sampled rows only, so the moments of all n_updates batches are taken ahead of the updates (ia_running_norm_merge_seq
// then applies them in order and keeps the per-update snapshots). Batch k = rows idx0 + k*idx_stride (n0, first table) then
// idx1 + k*idx_stride (n1, second table); X [n_updates][n0+n1][ldx] is scratch for the gathered rows; rn_ws + k*rn_stride
// receives batch k's moments in ia_running_norm_partial's layout and arithmetic. (Before: two gathers and one moment
// launch per update.)
extern "C" int ia_obs_moments_round(const float* obs0, const int64_t* idx0, int n0, const float* obs1, const int64_t* idx1,
                                    int n1, int obs_dim, int n_updates, int64_t idx_stride, float* X, int ldx, float* rn_ws,
                                    int64_t rn_stride, void* stream) {
  if (!obs0 || !obs1 || !idx0 || !idx1 || !X || !rn_ws || obs_dim < 1 || ldx < obs_dim) return IA_ERR_ARG;
  ia_mlp_desc d{};
  d.n_layers = 1;
  d.dims[0] = obs_dim;
  ia_disc_step_args a{};
  a.desc = &d;
  a.obs0 = obs0; a.idx0 = idx0; a.n0 = n0;
  a.obs1 = obs1; a.idx1 = idx1; a.n1 = n1;
  a.obs_dim = obs_dim; a.act_dim = 0; a.use_state = 1;
  a.X = X; a.ldx = ldx;
  return ia_disc32_assemble(&a, n_updates, idx_stride, (int64_t)(n0 + n1) * ldx, rn_stride, rn_ws, (hipStream_t)stream);
}

// One minibatch of train_disc for the 32-wide stack: the contract of ia_disc_step_basic (assemble unless pre_assembled,
// train-mode statistics, forward + BCE + backward with the weight gradients in ONE launch, slab reduction (+ Adam)).
int ia_disc32_step(const ia_disc_step_args* a, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  const ia_mlp_desc* d = a->desc;
  const int R = a->n0 + a->n1, D = d->dims[0];
  if (!ia_disc32_shape_ok(d, a->ldx) || !a->fused_ws || !a->X || !a->logits || !a->stats || !a->grads) return IA_ERR_ARG;
  const D32Ws w = d32_layout(d, R, a->fused_ws);
  int rc;
  if (!a->pre_assembled) {
    const bool upd = a->norm_mean != nullptr && a->update_norm;
    if (upd && !a->rn_ws) return IA_ERR_ARG;
    if ((rc = ia_disc32_assemble(a, 1, 0, 0, 0, upd ? a->rn_ws : nullptr, stream))) return rc;
    if (upd) {
      if ((rc = ia_running_norm_merge(a->rn_ws, 1, R, D, D, a->norm_mean, a->norm_var, a->norm_count, stream_))) return rc;
      if (a->pnorm_mean && a->pnorm_dim > 0 && a->pnorm_dim <= D &&
          (rc = ia_running_norm_merge(a->rn_ws, 1, R, a->pnorm_dim, D, a->pnorm_mean, a->pnorm_var, a->pnorm_count, stream_)))
        return rc;
    }
  }
  Disc32Args k{};
  k.X = a->X; k.ldx = a->ldx; k.D = D; k.R = R; k.n_expert = a->n_expert;
  k.mean = a->norm_mean; k.var = a->norm_var; k.eps = a->norm_eps;
  k.P = a->params; k.scale = a->loss_scale;
  k.part = w.part; k.pstride = w.P;
  k.logits = a->logits; k.dlogits = a->dlogits; k.stats = a->stats; k.bce_part = w.bce_part; k.ticket = w.ticket;
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(disc32_rows_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)sizeof(Disc32Lds)) != hipSuccess)
      return IA_ERR_ARG;
    attr = true;
  }
  hipLaunchKernelGGL(disc32_rows_kernel, dim3(w.nblk), dim3(A_THREADS), sizeof(Disc32Lds), stream, k);
  IA_CHECK_LAUNCH();
  if (a->adam && !a->accumulate)
    return ia_reduce_partials_adam(w.part, w.nblk, w.P, 1.0f, a->grads, a->params, a->exp_avg, a->exp_avg_sq, a->beta1,
                                   a->beta2, a->adam_eps, a->weight_decay, a->step_size, a->bc2_sqrt, stream_);
  if ((rc = ia_reduce_partials(w.part, w.nblk, w.P, 1.0f, a->accumulate, a->grads, stream_))) return rc;
  if (a->adam)
    return ia_adam_step(a->params, a->grads, a->exp_avg, a->exp_avg_sq, w.P, a->beta1, a->beta2, a->adam_eps,
                        a->weight_decay, a->step_size, a->bc2_sqrt, stream_);
  return IA_OK;
}
