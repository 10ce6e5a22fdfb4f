// Fused discriminator update for BasicRewardNet-shaped stacks  D -> H -> H -> 1  (ReLU, H in {128, 256},
// D <= 24 in every kernel below; D <= 63 in rows of up to 64 floats through disc_fb_kernel<H, 64, 64>):
// adversarial/common.py:352-373 (one minibatch of train_disc) in FIVE launches instead of 16,
// with the hidden activations of a 64-row tile chained through LDS instead of round-tripping HBM.
//
//   K1 assemble : gather + concat (+ one-hot) of [expert | generator] rows -> X, RunningNorm slab moments
//                 from the LDS copy of the slab, last-block Chan merge (util/networks.py:111-134); spare
//                 blocks transpose W2 -> W2T ([in][out]) so that both big layers stream contiguous K chunks
//   K2 forward  : per 64-row tile: normalise x on load, layer 1 (K = D) on MFMA -> h1 in LDS (+ HBM, for
//                 the weight gradients), layer 2 (K = H) with A = the LDS tile and B = W2T streamed through a
//                 3-stage LDS ring, logit layer + BCE-with-logits + its gradient + the 8 statistics partials
//                 + dW3/db3 partials + dh2 = dlogit * w3 * relu'(h2), all in the epilogue (h2 never leaves
//                 the registers)
//   K3 backward : per 64-row tile: dh1 = (dh2 . W2) * relu'(h1) with A = dh2 chunks and B = W2 chunks
//                 through LDS rings, then dW1/db1 partials = dh1^T . xn from the LDS tile (dh1 never
//                 reaches HBM)
//   K4 wgrad 2  : dW2/db2 = split-K TN GEMM over (dh2, h1)                      (gemm.hip)
//   K5 reduce   : fixed-order slab reduction of the three partial sets, gradient (accumulation), Adam,
//                 and the statistics row
// fp32 throughout (v_mfma_f32_32x32x2_f32: exact fp32 FMA chains). Deterministic: every reduction has a
// fixed order.
#include "common.h"
#include "rn_common.h"
#include "disc_reduce.h"
#include "../../include/imitation_hip.h"

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));

// Tile kernels (K2, K3) are templated on BM = rows per workgroup: 4 waves per 32 rows, each wave 32 rows x H/4
// columns. BM = 64 (512 threads, one workgroup per CU) is the default. BM = 32 (256 threads, <= 75 KB LDS) puts two
// workgroups on a CU -- measured no faster at R = 16 384 (116 vs 107 us per update): all 512 workgroups start
// together and run in lockstep, so their prologues / epilogues coincide instead of hiding behind each other's
// MFMA loop, and every workgroup streams all of W2 through LDS (twice the ring traffic per CU).
constexpr int FB_K = 16;     // K chunk of the streamed operand
constexpr int NS = 2;        // ring stages
constexpr int XP = 25;       // padded row length of the x tile / W1 in LDS (D <= 24), odd -> conflict-free
constexpr int XP3 = 33;      // x tile as a 32-wide B operand (K3)
constexpr int A_LD = FB_K + 1;
// Input widths: DW = 24 (D <= 24: K = 24 in layer 1, [xn | 1] as ONE 32-wide operand of the first-layer gradient) and
// DW = 64 (D <= 63, X rows up to 64 floats: layer 1 in 16-column steps over the row length, [xn | 1] as two 32-wide
// blocks). Row lengths of the W1 image / of the x tile in LDS, both odd:
constexpr int xp_of(int DW) { return DW == 24 ? XP : 65; }
constexpr int xp3_of(int DW) { return DW == 24 ? XP3 : 65; }

__device__ __forceinline__ int rowoff(int r) { return (r & 3) + 8 * (r >> 2); }
// phase clock of (block 0, thread 0) into dbg[slot] when measurement is switched on (scalar branch otherwise)
#define FUSED_STAMP(a, slot)                                                                     \
  do {                                                                                           \
    if ((a).dbg != nullptr && blockIdx.x == 0 && threadIdx.x == 0) (a).dbg[slot] = __builtin_readcyclecounter(); \
  } while (0)

struct FusedArgs {
  const float* X; int ldx; int R; int D;
  const float* mean; const float* var; float eps;     // mean == nullptr: no input normalisation
  const float* params;                                 // W1[H,D] b1[H] W2[H,H] b2[H] W3[H] b3[1]
  const float* W2T;                                    // [in][out]
  const float* W1P;                                    // [H][XP]: W1 rows zero-padded to the LDS row length
  float* h1; float* dh2;                               // [R,H] each
  unsigned long long* h1mask;                          // [tiles][8 waves][TN*16]: ballot(h1 > 0) per accumulator register
  float* logits; float* dlogits; int n_expert; float loss_scale;
  float* part;                                         // [tiles][8] statistics partials
  float* dump;                                         // [1024]: where the slice stores of rows past R go (disc_fb_kernel)
  float* P1; float* P3;                                // per-tile partial slabs: [tiles][H*D+H], [tiles][2H+1] = [dW3 | db3 | db2]
  long long* dbg;                                      // measurement only (ia_disc_fused_debug_timing): phase clocks of block 0
  // gradient-penalty passes (MODE 1 / 2 of the tile kernels; R = interpolated rows, X = the [2R, ldx] assembled batch)
  const float* gp_e;                                   // [R] weights: x_hat_i = e_i X[i] + (1 - e_i) X[R + i]
  unsigned long long* h2mask;                          // ballot(h2 > 0), the layout of h1mask
  float* gp_C;                                         // [R][ldx] row coefficients d(coef / R sum pen) / d xn
  float* gp_pen;                                       // [tiles] partial sums of (|grad_x D| - target)^2
  float gp_coef, gp_target;
  int out_act;                                         // disc_fwd_kernel MODE 3 (prediction): activation of the output
};

// x tile of rows [row0, row0+64): raw X normalised on the fly -> xs[row][ld], columns >= D and rows >= R zero.
// (X - mean) / sqrt(var + eps): util/networks.py:91 as ia_running_norm_apply computes it.
// HAT: the row is the interpolate e X[i] + (1 - e) X[R + i] (ia_gp_interpolate's expression), then normalised.
// RAW: rows of `X` as they are (the penalty's row coefficients).
template <int BM, bool HAT = false, bool RAW = false, bool WIDE = false>
__device__ __forceinline__ void load_x_tile(const FusedArgs& a, const float* __restrict__ X, int row0,
                                            float* __restrict__ xs, int ld, int tid0) {
  const int quads = a.ldx >> 2;
  auto piece = [&](int tid) {
    const int row = tid / quads, q = tid - row * quads;
    const int gi = row0 + row;
    const long long ro = (long long)min(gi, a.R - 1) * a.ldx + 4 * q;
    f4 v = *reinterpret_cast<const f4*>(X + ro);
    if constexpr (HAT) {
      const f4 g = *reinterpret_cast<const f4*>(X + (long long)a.R * a.ldx + ro);
      const float w = a.gp_e[min(gi, a.R - 1)];
      v.x = w * v.x + (1.f - w) * g.x; v.y = w * v.y + (1.f - w) * g.y;
      v.z = w * v.z + (1.f - w) * g.z; v.w = w * v.w + (1.f - w) * g.w;
    }
    const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = 4 * q + j;
      float x = vv[j];
      if (!RAW && a.mean != nullptr) {
        const int cc = min(c, a.D - 1);
        x = (x - a.mean[cc]) / sqrtf(a.var[cc] + a.eps);
      }
      xs[row * ld + c] = (gi < a.R && c < a.D) ? x : 0.f;
    }
  };
  // rows of up to 24 floats: at most one 16-byte piece per thread of the BM * 8; WIDE (up to 64 floats): up to two
  if constexpr (WIDE) {
    for (int e = tid0; e < BM * quads; e += BM * 8) piece(e);
  } else {
    if (tid0 < BM * quads) piece(tid0);
  }
}

// ------------------------------------------------------------------------------------------- K2
// Cross-lane sums by halving butterflies: N values per lane, summed across a group of lanes; every step
// trades half of the values with the partner lane, so N values cost N - 1 + log2(lanes / N) exchanges instead
// of N * log2(lanes). Fixed order -> deterministic.
// reduce16_in_half: 16 values across the 32 lanes of a wave half; lane `l` ends with the total of value
// index ((l>>4)&1)*8 + ((l>>3)&1)*4 + ((l>>2)&1)*2 + ((l>>1)&1) (lanes l and l^1 hold the same one).
template <int N, int O>
__device__ __forceinline__ void halve_step(float* __restrict__ v, int lane) {
  const bool up = (lane & O) != 0;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const float send = up ? v[i] : v[i + N];
    const float keep = up ? v[i + N] : v[i];
    v[i] = keep + __shfl_xor(send, O, 64);
  }
}
__device__ __forceinline__ float reduce16_in_half(float (&v)[16], int lane) {
  halve_step<8, 16>(v, lane);
  halve_step<4, 8>(v, lane);
  halve_step<2, 4>(v, lane);
  halve_step<1, 2>(v, lane);
  return v[0] + __shfl_xor(v[0], 1, 64);
}
// reduce8_in_wave: 8 values across the 64 lanes of a wave; lane 8k ends with the total of value index k.
__device__ __forceinline__ float reduce8_in_wave(float (&v)[8], int lane) {
  halve_step<4, 32>(v, lane);
  halve_step<2, 16>(v, lane);
  halve_step<1, 8>(v, lane);
  float t = v[0];
  t += __shfl_xor(t, 4, 64);
  t += __shfl_xor(t, 2, 64);
  t += __shfl_xor(t, 1, 64);
  return t;
}

// One K chunk (FB_K = 16 deep) of a wave's 32 x (TN*32) tile: all fragments of the chunk are requested, then the
// MFMAs issue (with two workgroups per CU the other workgroup's waves fill this wave's LDS round trip).
// a_at(ks) = the lane's A fragment of k-step ks; Bs = chunk base + lane column.
template <int H, int TN, class AAt>
__device__ __forceinline__ void chunk_mma(f32x16 (&acc)[TN], const AAt& a_at, const float* __restrict__ Bs, int lh) {
  constexpr int KS = FB_K / 2;
  float af[KS], bf[KS][TN];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    af[ks] = a_at(ks);
#pragma unroll
    for (int t = 0; t < TN; ++t) bf[ks][t] = Bs[(2 * ks + lh) * H + t * 32];
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
#pragma unroll
    for (int t = 0; t < TN; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[ks], bf[ks][t], acc[t], 0, 0, 0);
  __builtin_amdgcn_sched_barrier(0);
}

// MODE 0: the BCE update's forward (above). MODE 1: first pass of the gradient penalty -- the same two layers at the
// interpolated rows, ballots of both ReLU masks, u2 = relu'(h2) * w3 (= dD/dz2) into the dh2 buffer; no loss.
// MODE 2: second pass -- x := the row coefficients C, v1 = relu'(h1) * (C . W1^T) (into the h1 buffer: the operand of
// dW2 += u2^T v1), t = v1 . W2^T, dW3 partial = column sums of relu'(h2) * t (biases get no gradient: with the masks
// fixed the input gradient does not depend on them).
// The same chunk in two halves, for loops that request chunk c+1's fragments BEFORE the MFMAs of chunk c issue (register
// double buffering: the LDS round trip of a chunk hides behind the previous chunk's 1 024 matrix-pipe cycles instead of
// relying on the SIMD's other wave being out of phase -- the per-chunk barrier puts the two back in phase every time).
template <int H, int TN, class AAt>
__device__ __forceinline__ void chunk_frags(float (&af)[FB_K / 2], float (&bf)[FB_K / 2][TN], const AAt& a_at,
                                            const float* __restrict__ Bs, int lh) {
#pragma unroll
  for (int ks = 0; ks < FB_K / 2; ++ks) {
    af[ks] = a_at(ks);
#pragma unroll
    for (int t = 0; t < TN; ++t) bf[ks][t] = Bs[(2 * ks + lh) * H + t * 32];
  }
  __builtin_amdgcn_sched_barrier(0);
}
template <int TN>
__device__ __forceinline__ void chunk_mfmas(f32x16 (&acc)[TN], const float (&af)[FB_K / 2], const float (&bf)[FB_K / 2][TN]) {
#pragma unroll
  for (int ks = 0; ks < FB_K / 2; ++ks)
#pragma unroll
    for (int t = 0; t < TN; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[ks], bf[ks][t], acc[t], 0, 0, 0);
  __builtin_amdgcn_sched_barrier(0);
}
// MFMAs of the chunk in flight with the NEXT chunk's fragment reads dealt between them: one MFMA (64 pipe cycles), then
// two LDS reads, ... -- a wave can have 15 LDS operations outstanding, so 24 reads issued in one burst stall the issue of
// the first MFMA until the first dozen return.
template <int H, int TN, class AAt>
__device__ __forceinline__ void chunk_mfmas_and_next_frags(f32x16 (&acc)[TN], const float (&af)[FB_K / 2],
                                                           const float (&bf)[FB_K / 2][TN], float (&afn)[FB_K / 2],
                                                           float (&bfn)[FB_K / 2][TN], const AAt& a_nx,
                                                           const float* __restrict__ Bs, int lh) {
#pragma unroll
  for (int ks = 0; ks < FB_K / 2; ++ks) {
    afn[ks] = a_nx(ks);
#pragma unroll
    for (int t = 0; t < TN; ++t) bfn[ks][t] = Bs[(2 * ks + lh) * H + t * 32];
  }
#pragma unroll
  for (int ks = 0; ks < FB_K / 2; ++ks)
#pragma unroll
    for (int t = 0; t < TN; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[ks], bf[ks][t], acc[t], 0, 0, 0);
}
// Issue order of one loop iteration of disc_fb_kernel (everything between two barriers is one scheduling region): the
// chunk's MFMAs lead -- their operands have been in registers since the previous iteration -- and the iteration's other
// work is dealt between them: the ring write of chunk g+2 (NWR 16-byte LDS writes), the request of chunk g+3 (NWR
// global loads), the LDS reads (next chunk's fragments + the 16-byte slice of the tile that trickles out to HBM), the
// slice's store.
template <int TN, int NWR>
__device__ __forceinline__ void schedule_iteration() {
  __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
  __builtin_amdgcn_sched_group_barrier(0x200, NWR, 0);
  __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
  __builtin_amdgcn_sched_group_barrier(0x020, NWR, 0);
#pragma unroll
  for (int i = 2; i < (FB_K / 2) * TN; ++i) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // one MFMA
    __builtin_amdgcn_sched_group_barrier(0x100, TN == 2 ? 2 : 3, 0);   // LDS reads
  }
  __builtin_amdgcn_sched_group_barrier(0x040, 1, 0);
  __builtin_amdgcn_sched_barrier(0);
}

template <int H, int BM, int MODE = 0>
__global__ __launch_bounds__(BM * 8) void disc_fwd_kernel(FusedArgs a) {
  constexpr int NT = BM * 8, NW = NT / 64;
  constexpr int TN = H / 128;          // 32-column MFMA tiles per wave
  constexpr int WC = TN * 32;          // columns per wave
  constexpr int LDH = H + 1;
  constexpr int NCH = H / FB_K;
  constexpr int BST = FB_K * H;        // floats per B stage
  constexpr int BV = BST / 4 / NT;     // float4 per thread per B chunk
  static_assert(BST % (4 * NT) == 0, "B chunk must divide among the threads");
  static_assert(H * XP <= NS * BST, "the W1 image borrows the ring during layer 1");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* h1s = smem;                   // [BM][LDH]   h1 tile, later the dh2 tile
  float* bs = h1s + BM * LDH;          // NS x [FB_K][H] ring; holds the W1 image [H][XP] during layer 1
  float* red = bs + NS * BST;          // [4][BM] logit partials per column group
  float* dls = red + 4 * BM;           // [BM] dlogit of the tile's rows
  float* w3red = dls + BM;             // [2][BM/32][H] dW3 / db2 partials per row group
  float* xs = w3red + 2 * (BM / 32) * H;   // [BM][XP]
  float* w1s = bs;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wm = wave >> 2, wn = wave & 3;
  const int row0 = blockIdx.x * BM;
  const int D = a.D;
  const float* W1 = a.params;
  const float* b1 = W1 + (long long)H * D;
  const float* b2 = b1 + H + (long long)H * H;
  const float* w3 = b2 + H;
  const float* b3 = w3 + H;

  // ---- prologue: every first-use operand is requested up front (coalesced 16-byte copies)
  FUSED_STAMP(a, 0);
  f4 rb[BV];
  auto bload = [&](int c) {
#pragma unroll
    for (int i = 0; i < BV; ++i)
      rb[i] = *reinterpret_cast<const f4*>(a.W2T + (long long)c * BST + (long long)(tid + i * NT) * 4);
  };
  auto bstore = [&](int c) {
    float* S = bs + (c % NS) * BST;
#pragma unroll
    for (int i = 0; i < BV; ++i) *reinterpret_cast<f4*>(S + (tid + i * NT) * 4) = rb[i];
  };
  constexpr int W1Q = H * XP / 4;                      // float4 of the W1 image
  constexpr int W1V = (W1Q + NT - 1) / NT;
  static_assert((H * XP) % 4 == 0, "W1 image is copied in 16-byte pieces");
  f4 w1v[W1V];
#pragma unroll
  for (int i = 0; i < W1V; ++i) w1v[i] = reinterpret_cast<const f4*>(a.W1P)[min(tid + i * NT, W1Q - 1)];
  float b1v[TN], b2v[TN], w3v[TN];
#pragma unroll
  for (int t = 0; t < TN; ++t) {
    const int col = wn * WC + t * 32 + li;
    b1v[t] = MODE == 2 ? 0.f : b1[col]; b2v[t] = MODE == 2 ? 0.f : b2[col]; w3v[t] = w3[col];
  }
  const float b3v = b3[0];
  unsigned int hm_lo = 0u, hm_hi = 0u, h2_lo = 0u, h2_hi = 0u;   // MODE 2: the first pass's ballots (word (t, r) in lane t*16+r)
  if constexpr (MODE == 2) {
    const long long mo = ((long long)blockIdx.x * NW + wave) * (TN * 16) + min(lane, TN * 16 - 1);
    const unsigned long long w1 = a.h1mask[mo], w2 = a.h2mask[mo];
    hm_lo = (unsigned int)w1; hm_hi = (unsigned int)(w1 >> 32);
    h2_lo = (unsigned int)w2; h2_hi = (unsigned int)(w2 >> 32);
  }
  if constexpr (MODE == 0 || MODE == 3) load_x_tile<BM>(a, a.X, row0, xs, XP, tid);
  else if constexpr (MODE == 1) load_x_tile<BM, true>(a, a.X, row0, xs, XP, tid);
  else load_x_tile<BM, false, true>(a, a.gp_C, row0, xs, XP, tid);
  for (int e = tid; e < BM * (24 - a.ldx); e += NT) {  // columns [ldx, 24) of the K = 24 operand
    const int w = 24 - a.ldx;
    const int row = e / w;
    xs[row * XP + a.ldx + e - row * w] = 0.f;
  }
  bload(0);                                            // stays in registers until layer 1 is done with the ring
#pragma unroll
  for (int i = 0; i < W1V; ++i)
    if (tid + i * NT < W1Q) reinterpret_cast<f4*>(w1s)[tid + i * NT] = w1v[i];
  __syncthreads();
  FUSED_STAMP(a, 1);

  // ---- layer 1: h1 = relu(xn . W1^T + b1), K = 24
  f32x16 acc[TN];
#pragma unroll
  for (int t = 0; t < TN; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  {
    float af[12], bf[12][TN];
#pragma unroll
    for (int ks = 0; ks < 12; ++ks) {
      af[ks] = xs[(wm * 32 + li) * XP + 2 * ks + lh];
#pragma unroll
      for (int t = 0; t < TN; ++t) bf[ks][t] = w1s[(wn * WC + t * 32 + li) * XP + 2 * ks + lh];
    }
#pragma unroll
    for (int ks = 0; ks < 12; ++ks)
#pragma unroll
      for (int t = 0; t < TN; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[ks], bf[ks][t], acc[t], 0, 0, 0);
  }
  FUSED_STAMP(a, 2);
  unsigned long long mword = 0ull;
#pragma unroll
  for (int t = 0; t < TN; ++t) {
    const int col = wn * WC + t * 32 + li;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = wm * 32 + 4 * lh + rowoff(r);
      if constexpr (MODE == 2) {
        const unsigned int wlo = __builtin_amdgcn_readlane(hm_lo, t * 16 + r), whi = __builtin_amdgcn_readlane(hm_hi, t * 16 + r);
        const bool on = (((lh ? whi : wlo) >> li) & 1u) != 0u;
        h1s[row * LDH + col] = on ? acc[t][r] : 0.f;
      } else {
        const float v = fmaxf(acc[t][r] + b1v[t], 0.f);
        h1s[row * LDH + col] = v;
        // relu'(h1) for the backward tile kernel: one 64-bit ballot per accumulator register instead of a
        // re-read of the tile; lane (t*16 + r) keeps word (t, r) -> ONE coalesced store per wave below
        if constexpr (MODE != 3) {
          const unsigned long long m = __ballot(v > 0.f);
          if (lane == t * 16 + r) mword = m;
        }
      }
      acc[t][r] = 0.f;
    }
  }
  if (MODE != 2 && MODE != 3 && lane < TN * 16) a.h1mask[((long long)blockIdx.x * NW + wave) * (TN * 16) + lane] = mword;
  __syncthreads();   // every wave is done with the W1 image: the ring is the ring from here on
  bstore(0);
  if (NCH > 1) bload(1);
  __syncthreads();
  FUSED_STAMP(a, 3);

  // ---- layer 2: h2 = relu(h1 . W2^T + b2): A = the LDS tile, B = W2T chunks through the ring. Iteration c: the
  // registers (chunk c+1) go to stage (c+1)%2 -- last read in iteration c-1, before barrier c-1 --, chunk c+2 is
  // requested, chunk c is multiplied out of stage c%2 (complete since barrier c-1), barrier c.
  {
    const float* Ar = h1s + (wm * 32 + li) * LDH + lh;
    const int boff = wn * WC + li;
    // the tile's h1 goes out to HBM (for the weight gradients) in slices, one per chunk, read back from LDS
    // row-major: coalesced stores trickling beside the MFMAs instead of one burst of all workgroups. 8 bytes per
    // thread and chunk, UNCONDITIONALLY (rows past R land in a dump slot): behind a branch the compiler cannot count
    // the store in vmcnt and makes the next ring write wait for the store's acknowledgement (see disc_fb_kernel)
    static_assert(BM * H / NCH / 2 == NT, "one 8-byte piece per thread and chunk");
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      if (c + 1 < NCH) bstore(c + 1);
      if (c + 2 < NCH) bload(c + 2);
      const int e2 = c * NT + tid;
      const int hrow = e2 / (H / 2), hcol = (e2 % (H / 2)) * 2;
      float2 hv;
      if constexpr (MODE != 1 && MODE != 3) { hv.x = h1s[hrow * LDH + hcol]; hv.y = h1s[hrow * LDH + hcol + 1]; }
      auto a_at = [&](int ks) { return Ar[c * FB_K + 2 * ks]; };
      chunk_mma<H, TN>(acc, a_at, bs + (c % NS) * BST + boff, lh);
      if constexpr (MODE != 1 && MODE != 3) {
        float* dst = row0 + hrow < a.R ? a.h1 + (long long)(row0 + hrow) * H + hcol : a.dump + 2 * tid;
        *reinterpret_cast<float2*>(dst) = hv;
      }
      __syncthreads();
    }
  }
  FUSED_STAMP(a, 4);

  if constexpr (MODE == 1) {
    // ---- first penalty pass: u2 = relu'(h2) * w3 (rows past R: 0) -> the dh2 tile; ballots of relu'(h2)
    unsigned long long m2w = 0ull;
#pragma unroll
    for (int t = 0; t < TN; ++t) {
      const int col = wn * WC + t * 32 + li;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * 32 + 4 * lh + rowoff(r);
        const bool on = fmaxf(acc[t][r] + b2v[t], 0.f) > 0.f;
        h1s[row * LDH + col] = (on && row0 + row < a.R) ? w3v[t] : 0.f;
        const unsigned long long m = __ballot(on);
        if (lane == t * 16 + r) m2w = m;
      }
    }
    if (lane < TN * 16) a.h2mask[((long long)blockIdx.x * NW + wave) * (TN * 16) + lane] = m2w;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < BM * H / 4 / NT; ++i) {
      const int e4 = tid + i * NT;
      const int row = e4 / (H / 4), c4 = (e4 % (H / 4)) * 4;
      const float* hp = h1s + row * LDH + c4;
      f4 v;
      v.x = hp[0]; v.y = hp[1]; v.z = hp[2]; v.w = hp[3];
      if (row0 + row < a.R) *reinterpret_cast<f4*>(a.dh2 + (long long)(row0 + row) * H + c4) = v;
    }
    return;
  }
  if constexpr (MODE == 2) {
    // ---- second penalty pass: dW3 partial = column sums of relu'(h2) * t over the tile's rows; db3 = 0
#pragma unroll
    for (int t = 0; t < TN; ++t) {
      const int col = wn * WC + t * 32 + li;
      float s = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const unsigned int wlo = __builtin_amdgcn_readlane(h2_lo, t * 16 + r), whi = __builtin_amdgcn_readlane(h2_hi, t * 16 + r);
        const bool on = (((lh ? whi : wlo) >> li) & 1u) != 0u;
        s += on ? acc[t][r] : 0.f;
      }
      s += __shfl_xor(s, 32, 64);
      if (lh == 0) w3red[wm * H + col] = s;
    }
    __syncthreads();
    if (tid < H) {
      float s = w3red[tid];
#pragma unroll
      for (int g = 1; g < BM / 32; ++g) s += w3red[g * H + tid];
      a.P3[(long long)blockIdx.x * (2 * H + 1) + tid] = s;
    }
    if (tid == 0) a.P3[(long long)blockIdx.x * (2 * H + 1) + H] = 0.f;
    return;
  }
  // ---- epilogue: logit, BCE, dlogit, statistics, dW3/db3 partials, dh2
  float h2[TN][16];
  float p[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < TN; ++t) {
      h2[t][r] = fmaxf(acc[t][r] + b2v[t], 0.f);
      s += h2[t][r] * w3v[t];
    }
    p[r] = s;
  }
  {
    const float tot = reduce16_in_half(p, lane);
    const int r = ((lane >> 4) & 1) * 8 + ((lane >> 3) & 1) * 4 + ((lane >> 2) & 1) * 2 + ((lane >> 1) & 1);
    if ((lane & 1) == 0) red[wn * BM + wm * 32 + 4 * lh + rowoff(r)] = tot;
  }
  __syncthreads();
  FUSED_STAMP(a, 5);
  if constexpr (MODE == 3) {
    // ---- prediction (`RewardNet.predict_th` of the rows, `rewards/reward_nets.py:176-204`): the output, through its
    //      activation (GAIL's generator reward: softplus, `gail.py:75-83`), is all that leaves the tile
    if (wave == 0 && lane < BM && row0 + lane < a.R) {
      const float x = ((red[lane] + red[BM + lane]) + red[2 * BM + lane]) + red[3 * BM + lane] + b3v;
      a.logits[row0 + lane] = ia_apply_act(x, a.out_act);
    }
    return;
  }
  if (wave == 0) {  // one row per lane (lanes >= BM idle but take part in the reduction)
    const int row = min(lane, BM - 1);
    const int gi = row0 + row;
    const bool valid = lane < BM && gi < a.R;
    const float x = ((red[row] + red[BM + row]) + red[2 * BM + row]) + red[3 * BM + row] + b3v;
    // adversarial/common.py:360-368 + 27-92, the arithmetic of bce_kernel (mlp.hip)
    const float y = gi < a.n_expert ? 1.f : 0.f;
    const float lse = log1pf(expf(-fabsf(x)));
    const float pr = 1.f / (1.f + expf(-x));
    const float inv = a.loss_scale / (float)a.R;
    const float dl = valid ? (pr - y) * inv : 0.f;
    if (lane < BM) dls[lane] = dl;
    if (valid) {
      a.logits[gi] = x;
      if (a.dlogits) a.dlogits[gi] = dl;
    }
    const bool is_gen_pred = x < 0.f, is_gen_true = y == 0.f;
    const bool ok = is_gen_pred == is_gen_true;
    float vals[8];
    vals[0] = valid ? (1.f - y) * x - (fminf(x, 0.f) - lse) : 0.f;
    vals[1] = (valid && ok) ? 1.f : 0.f;
    vals[2] = (valid && ok && !is_gen_true) ? 1.f : 0.f;
    vals[3] = (valid && ok && is_gen_true) ? 1.f : 0.f;
    vals[4] = (valid && is_gen_pred) ? 1.f : 0.f;
    vals[5] = valid ? (1.f - pr) * x - (fminf(x, 0.f) - lse) : 0.f;
    vals[6] = dl;  // db3 partial
    vals[7] = 0.f;
    const float tot = reduce8_in_wave(vals, lane);
    const int k = lane >> 3;
    if ((lane & 7) == 0) {
      if (k < 6) a.part[(long long)blockIdx.x * 8 + k] = tot;
      else if (k == 6) a.P3[(long long)blockIdx.x * (2 * H + 1) + H] = tot;
    }
  }
  __syncthreads();
  FUSED_STAMP(a, 6);
  float dlr[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) dlr[r] = dls[wm * 32 + 4 * lh + rowoff(r)];
#pragma unroll
  for (int t = 0; t < TN; ++t) {
    const int col = wn * WC + t * 32 + li;
    float s = 0.f, sb = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = wm * 32 + 4 * lh + rowoff(r);
      s += dlr[r] * h2[t][r];
      const float dv = h2[t][r] > 0.f ? dlr[r] * w3v[t] : 0.f;
      h1s[row * LDH + col] = dv;   // dh2 (the h1 tile is dead by now)
      sb += dv;                    // db2 = column sums of dh2 (the split-K product no longer carries them)
    }
    s += __shfl_xor(s, 32, 64);
    sb += __shfl_xor(sb, 32, 64);
    if (lh == 0) { w3red[wm * H + col] = s; w3red[(BM / 32 + wm) * H + col] = sb; }
  }
  __syncthreads();
  if (tid < H) {
    float s = w3red[tid], sb = w3red[(BM / 32) * H + tid];
#pragma unroll
    for (int g = 1; g < BM / 32; ++g) { s += w3red[g * H + tid]; sb += w3red[(BM / 32 + g) * H + tid]; }
    a.P3[(long long)blockIdx.x * (2 * H + 1) + tid] = s;
    a.P3[(long long)blockIdx.x * (2 * H + 1) + H + 1 + tid] = sb;
  }
  // dh2 tile -> HBM row-major in 16-byte stores (a dword-per-lane epilogue is store-issue bound)
#pragma unroll
  for (int i = 0; i < BM * H / 4 / NT; ++i) {
    const int e4 = tid + i * NT;
    const int row = e4 / (H / 4), c4 = (e4 % (H / 4)) * 4;
    const float* hp = h1s + row * LDH + c4;
    f4 v;
    v.x = hp[0]; v.y = hp[1]; v.z = hp[2]; v.w = hp[3];
    if (row0 + row < a.R) *reinterpret_cast<f4*>(a.dh2 + (long long)(row0 + row) * H + c4) = v;
  }
  FUSED_STAMP(a, 7);
}

// ------------------------------------------------------------------------------------------- K3
// MODE 1 (gradient penalty, between the two passes; BM = 32): dh2 := u2, so the chain leaves u1 = relu'(h1) * (u2 . W2)
// = dD/dz1 in the LDS tile; then gn = u1 . W1 (the input gradient w.r.t. the normalised row: K = H split over the four
// waves, partial tiles summed in fixed order), the row coefficients of gp_row_coeffs_kernel (mlp.hip) -- g = gn / sigma,
// n = |g|, pen = (n - target)^2, C = coef / R * 2 (n - target) / n * g / sigma -- into the x tile (and HBM, for the
// second pass), and the same tile product as the BCE pass: dW1 partial = u1^T . C (no bias column).
template <int H, int BM, int MODE = 0>
__global__ __launch_bounds__(BM * 8) void disc_bwd_kernel(FusedArgs a) {
  constexpr int NT = BM * 8, NW = NT / 64;
  constexpr int TN = H / 128;
  constexpr int WC = TN * 32;
  constexpr int LDH = H + 1;
  constexpr int NCH = H / FB_K;
  constexpr int BST = FB_K * H;
  constexpr int BV = BST / 4 / NT;
  constexpr int AST = BM * A_LD;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* xs = smem;                    // [BM][XP3]: xn | 1 | 0 ... (the ones column makes db1 a column of dW1)
  float* d1s = xs + BM * XP3;          // [BM][LDH]  dh1 tile
  float* bs = d1s + BM * LDH;          // NS x [FB_K][H]  W2 chunks (k = out unit, n = in unit); later the P1 slab image
  float* as = bs + NS * BST;           // NS x [BM][A_LD]  dh2 chunks

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wm = wave >> 2, wn = wave & 3;
  const int row0 = blockIdx.x * BM;
  const int D = a.D;
  const float* W2 = a.params + (long long)H * D + H;

  FUSED_STAMP(a, 8);
  f4 rb[BV], ra;
  const int arow = tid >> 2, aq = tid & 3;
  const bool a_thread = tid < BM * 4;
  const float* abase = a.dh2 + (long long)min(row0 + arow, a.R - 1) * H + aq * 4;
  const bool a_valid = a_thread && (row0 + arow < a.R);
  auto gload = [&](int c) {
#pragma unroll
    for (int i = 0; i < BV; ++i)
      rb[i] = *reinterpret_cast<const f4*>(W2 + (long long)c * BST + (long long)(tid + i * NT) * 4);
    if (a_thread) ra = *reinterpret_cast<const f4*>(abase + c * FB_K);
  };
  auto lstore = [&](int c) {
    float* S = bs + (c % NS) * BST;
#pragma unroll
    for (int i = 0; i < BV; ++i) *reinterpret_cast<f4*>(S + (tid + i * NT) * 4) = rb[i];
    if (a_thread) {
      float* d = as + (c % NS) * AST + arow * A_LD + aq * 4;
      d[0] = a_valid ? ra.x : 0.f; d[1] = a_valid ? ra.y : 0.f; d[2] = a_valid ? ra.z : 0.f; d[3] = a_valid ? ra.w : 0.f;
    }
  };
  gload(0);
  // relu'(h1) of the tile: the forward kernel's ballots, same (wave, tile, register) decomposition
  // (lane l < TN*16 holds word l: one coalesced load; word (t, r) is broadcast with readlane where it is used)
  const unsigned long long hmw = a.h1mask[((long long)blockIdx.x * NW + wave) * (TN * 16) + min(lane, TN * 16 - 1)];
  const unsigned int hm_lo = (unsigned int)hmw, hm_hi = (unsigned int)(hmw >> 32);
  if constexpr (MODE == 0) {
    load_x_tile<BM>(a, a.X, row0, xs, XP3, tid);
    for (int e = tid; e < BM * (XP3 - 1 - a.ldx); e += NT) {  // columns [ldx, 32) of the B operand
      const int w = XP3 - 1 - a.ldx;
      const int row = e / w, c = a.ldx + e - row * w;
      xs[row * XP3 + c] = 0.f;
    }
  }
  lstore(0);
  if (NCH > 1) gload(1);
  __syncthreads();
  if (MODE == 0 && tid < BM) xs[tid * XP3 + D] = 1.f;  // after the tile writes above (column D < 32 is a zero column there)
  FUSED_STAMP(a, 9);

  f32x16 acc[TN];
#pragma unroll
  for (int t = 0; t < TN; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  {
    const float* Ab = as + (wm * 32 + li) * A_LD + lh;
    const int boff = wn * WC + li;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      if (c + 1 < NCH) lstore(c + 1);
      if (c + 2 < NCH) gload(c + 2);
      auto a_at = [&](int ks) { return Ab[(c % NS) * AST + 2 * ks]; };
      chunk_mma<H, TN>(acc, a_at, bs + (c % NS) * BST + boff, lh);
      __syncthreads();
    }
  }
  FUSED_STAMP(a, 10);
#pragma unroll
  for (int t = 0; t < TN; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const unsigned int wlo = __builtin_amdgcn_readlane(hm_lo, t * 16 + r), whi = __builtin_amdgcn_readlane(hm_hi, t * 16 + r);
      const bool on = (((lh ? whi : wlo) >> li) & 1u) != 0u;
      d1s[(wm * 32 + 4 * lh + rowoff(r)) * LDH + wn * WC + t * 32 + li] = on ? acc[t][r] : 0.f;
    }
  __syncthreads();
  FUSED_STAMP(a, 11);
  if constexpr (MODE == 1) {
    static_assert(MODE == 0 || BM == 32, "the penalty pass runs 32-row tiles (four waves split K)");
    // gn partials: wave w multiplies u1[:, 64w' ...] by W1 rows of its K quarter; B fragment = W1[k][c] from the
    // padded image (columns >= D are zero there up to XP; lanes past it read column 0 and are masked)
    float* gpart = bs;                                   // [NW][32][33] (the rings are free by now)
    static_assert(NW * 32 * 33 <= NS * BST + NS * AST, "gn partial tiles fit the two rings");
    {
      constexpr int KQ = H / NW;                         // k range of this wave
      f32x16 accg;
#pragma unroll
      for (int r = 0; r < 16; ++r) accg[r] = 0.f;
      const bool cok = li < D;
#pragma unroll 8
      for (int s2 = 0; s2 < KQ / 2; ++s2) {
        const int k = wave * KQ + 2 * s2 + lh;
        const float af = d1s[li * LDH + k];
        const float w1 = a.W1P[k * XP + min(li, XP - 1)];
        accg = __builtin_amdgcn_mfma_f32_32x32x2f32(af, cok ? w1 : 0.f, accg, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) gpart[(wave * 32 + 4 * lh + rowoff(r)) * 33 + li] = accg[r];
    }
    __syncthreads();
    float pen_w = 0.f;
    {
      const float inv = (a.mean != nullptr && li < D) ? 1.f / sqrtf(a.var[min(li, D - 1)] + a.eps) : 1.f;
#pragma unroll
      for (int it = 0; it < 32 / (2 * NW); ++it) {
        const int row = wave * (32 / NW) + 2 * it + lh;  // two rows per wave and iteration: lane half = row, li = column
        const float gn = ((gpart[row * 33 + li] + gpart[(32 + row) * 33 + li]) + gpart[(64 + row) * 33 + li]) +
                         gpart[(96 + row) * 33 + li];
        const float g = li < D ? gn * inv : 0.f;
        float sq = g * g;
        sq += __shfl_xor(sq, 16, 64); sq += __shfl_xor(sq, 8, 64); sq += __shfl_xor(sq, 4, 64);
        sq += __shfl_xor(sq, 2, 64); sq += __shfl_xor(sq, 1, 64);
        const float n = sqrtf(sq);
        const bool valid = row0 + row < a.R;
        const float kk = (valid && n > 0.f) ? a.gp_coef / (float)a.R * 2.f * (n - a.gp_target) / n : 0.f;
        const float cv = li < D ? kk * gn * inv * inv : 0.f;
        xs[row * XP3 + li] = cv;
        if (valid && li < a.ldx) a.gp_C[(long long)(row0 + row) * a.ldx + li] = cv;
        const float pr = valid ? (n - a.gp_target) * (n - a.gp_target) : 0.f;
        pen_w += __shfl(pr, 0, 64) + __shfl(pr, 32, 64);
      }
    }
    if (lane == 0) xs[wave * XP3 + 32] = pen_w;          // (column 32 of the x tile is padding)
    __syncthreads();
    if (tid == 0) a.gp_pen[blockIdx.x] = ((xs[32] + xs[XP3 + 32]) + xs[2 * XP3 + 32]) + xs[3 * XP3 + 32];
  }

  // ---- [dW1 | db1] partial [H, D + 1] = dh1^T . [xn | 1] over the tile's BM rows: one 32-unit M tile per wave
  //      and pass; the slab image [W1 grad [H][D] | b1 grad [H]] is assembled in LDS (the ring is free by now) ...
  const long long n1 = (long long)H * D + H;
  float* P1 = a.P1 + (long long)blockIdx.x * n1;
  for (int mt = wave; mt < H / 32; mt += NW) {
    f32x16 acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc1[r] = 0.f;
    float af[BM / 2], bf[BM / 2];
#pragma unroll
    for (int s = 0; s < BM / 2; ++s) {
      const int k = 2 * s + lh;
      af[s] = d1s[k * LDH + mt * 32 + li];
      bf[s] = xs[k * XP3 + li];
    }
#pragma unroll
    for (int s = 0; s < BM / 2; ++s) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s], bf[s], acc1, 0, 0, 0);
    if (li <= D) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = mt * 32 + 4 * lh + rowoff(r);
        bs[li < D ? i * D + li : H * D + i] = acc1[r];
      }
    }
  }
  __syncthreads();
  // ... and leaves in 16-byte coalesced stores (a dword-per-lane store of 23-float rows is store-issue bound)
  for (int e = tid; e < (int)(n1 / 4); e += NT)
    reinterpret_cast<f4*>(P1)[e] = reinterpret_cast<const f4*>(bs)[e];
  FUSED_STAMP(a, 12);
}

// ------------------------------------------------------------------------------------------- K2 + K3 in one launch
// The BCE update's forward AND backward of a tile in one workgroup (round 3; the two-launch form above stays for the
// gradient penalty's passes and as `ia_disc_fused_split_tiles(1)`): after the forward epilogue the dh2 tile sits in LDS
// where h1 was, so the input-gradient chain takes its A operand from there -- no dh2 re-read from HBM, no second
// x / mask load, no second dispatch; dh2 trickles out to HBM beside the chain's MFMAs exactly as h1 does beside layer 2
// (the weight-gradient GEMM needs both), W2's first chunks are requested while the BCE epilogue runs, and relu'(h1)
// stays in the registers that took the ballots. Same arithmetic, same summation order as K2 + K3: bit-identical
// outputs (tests/test_disc_fused_gpu.py compares the two forms).
// LDS of disc_fb_kernel in floats: h1 tile | region R (ring, then the scratch rows, then padding) | x tile. Region R
// holds the W1 image during layer 1 and the first-layer slab image at the end, whichever of the three is larger.
template <int H, int BM, int DW>
constexpr int fb_region_floats() {
  constexpr int ring = 3 * FB_K * H + 4 * BM + BM + 2 * (BM / 32) * H;
  constexpr int img = H * xp_of(DW) > H * (DW + 1) ? H * xp_of(DW) : H * (DW + 1);
  return ring > img ? ring : img;
}
template <int H, int BM, int DW>
constexpr int fb_lds_floats() { return BM * (H + 1) + fb_region_floats<H, BM, DW>() + BM * xp3_of(DW); }

template <int H, int BM, int DW>
__device__ __forceinline__ void disc_fb_body(const FusedArgs& a, const int bid) {
  constexpr int NT = BM * 8, NW = NT / 64;
  constexpr int XPW = xp_of(DW), XP3W = xp3_of(DW);   // row lengths of the W1 image / the x tile
  constexpr bool WIDE = DW != 24;
  constexpr int TN = H / 128;
  constexpr int WC = TN * 32;
  constexpr int LDH = H + 1;
  constexpr int NCH = H / FB_K;
  constexpr int BST = FB_K * H;
  constexpr int BV = BST / 4 / NT;
  static_assert(BST % (4 * NT) == 0, "B chunk must divide among the threads");
  constexpr int NR = 3;                // ring stages: chunk g sits in stage g % 3 (g < NCH: W2T chunk g, then W2 chunk g - NCH)
  static_assert(WIDE || H * XP <= NR * BST, "the W1 image borrows the ring during layer 1");
  static_assert(WIDE || H * 24 + H <= NR * BST, "the first-layer slab image borrows the ring at the end");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* h1s = smem;                   // [BM][LDH]   h1 tile -> dh2 tile -> dh1 tile
  float* bs = h1s + BM * LDH;          // NR x [FB_K][H] ring (W2T chunks, then W2 chunks); W1 image during layer 1;
                                       // the P1 slab image at the end (WIDE: both run on over the scratch rows below,
                                       // which are idle at those times, and some padding)
  float* red = bs + NR * BST;          // [4][BM] logit partials per column group
  float* dls = red + 4 * BM;           // [BM] dlogit of the tile's rows
  float* w3red = dls + BM;             // [2][BM/32][H] dW3 / db2 partials per row group
  float* xs = bs + fb_region_floats<H, BM, DW>();   // [BM][XP3W]: xn | 0 ... (layer 1's A operand; later [xn | 1 | 0 ...] as dW1's B)
  float* w1s = bs;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wm = wave >> 2, wn = wave & 3;
  const int row0 = bid * BM;
  const int D = a.D;
  const float* W1 = a.params;
  const float* b1 = W1 + (long long)H * D;
  const float* W2 = b1 + H;
  const float* b2 = W2 + (long long)H * H;
  const float* w3 = b2 + H;
  const float* b3 = w3 + H;

  // ---- prologue: every first-use operand is requested up front (coalesced 16-byte copies)
  FUSED_STAMP(a, 0);
  // chunk stream g = 0 .. 2 NCH - 1 through registers (requested three iterations ahead of its MFMAs) and the ring
  f4 rb[BV], rb1[BV];
  auto bload = [&](f4 (&r)[BV], int g) {
    const float* src = g < NCH ? a.W2T + (long long)g * BST : W2 + (long long)(g - NCH) * BST;
#pragma unroll
    for (int i = 0; i < BV; ++i) r[i] = *reinterpret_cast<const f4*>(src + (long long)(tid + i * NT) * 4);
  };
  auto bstore = [&](const f4 (&r)[BV], int g) {
    float* S = bs + (g % NR) * BST;
#pragma unroll
    for (int i = 0; i < BV; ++i) *reinterpret_cast<f4*>(S + (tid + i * NT) * 4) = r[i];
  };
  static_assert(H * XPW % 4 == 0, "the W1 image is copied in 16-byte pieces");
  constexpr int W1Q = H * XPW / 4;                     // float4 of the W1 image
  constexpr int W1V = (W1Q + NT - 1) / NT;
  f4 w1v[W1V];
#pragma unroll
  for (int i = 0; i < W1V; ++i) w1v[i] = reinterpret_cast<const f4*>(a.W1P)[min(tid + i * NT, W1Q - 1)];
  float b1v[TN], b2v[TN], w3v[TN];
#pragma unroll
  for (int t = 0; t < TN; ++t) {
    const int col = wn * WC + t * 32 + li;
    b1v[t] = b1[col]; b2v[t] = b2[col]; w3v[t] = w3[col];
  }
  const float b3v = b3[0];
  load_x_tile<BM, false, false, WIDE>(a, a.X, row0, xs, XP3W, tid);
  for (int e = tid; e < BM * (XP3W - 1 - a.ldx); e += NT) {  // columns [ldx, 32) (WIDE: [ldx, 64))
    const int w = XP3W - 1 - a.ldx;
    const int row = e / w, c = a.ldx + e - row * w;
    xs[row * XP3W + c] = 0.f;
  }
  bload(rb, 0);                                        // stay in registers until layer 1 is done with the ring
  bload(rb1, 1);
#pragma unroll
  for (int i = 0; i < W1V; ++i)
    if (tid + i * NT < W1Q) reinterpret_cast<f4*>(w1s)[tid + i * NT] = w1v[i];
  __syncthreads();
  FUSED_STAMP(a, 1);

  // ---- layer 1: h1 = relu(xn . W1^T + b1), K = 24 (WIDE: 16 columns at a time over the row length; the columns from
  //      D on are zero in both operands)
  f32x16 acc[TN];
#pragma unroll
  for (int t = 0; t < TN; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  if constexpr (!WIDE) {
    float af[12], bf[12][TN];
#pragma unroll
    for (int ks = 0; ks < 12; ++ks) {
      af[ks] = xs[(wm * 32 + li) * XP3 + 2 * ks + lh];
#pragma unroll
      for (int t = 0; t < TN; ++t) bf[ks][t] = w1s[(wn * WC + t * 32 + li) * XP + 2 * ks + lh];
    }
#pragma unroll
    for (int ks = 0; ks < 12; ++ks)
#pragma unroll
      for (int t = 0; t < TN; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[ks], bf[ks][t], acc[t], 0, 0, 0);
  } else {
    const int nk16 = (a.ldx + 15) >> 4;
    const float* xr = xs + (wm * 32 + li) * XP3W + lh;
    const float* wr = w1s + (wn * WC + li) * XPW + lh;
    float af[2][8], bf[2][8][TN];
    auto frags = [&](float (&fa_)[8], float (&fb_)[8][TN], int kc) {
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        fa_[ks] = xr[kc * 16 + 2 * ks];
#pragma unroll
        for (int t = 0; t < TN; ++t) fb_[ks][t] = wr[t * 32 * XPW + kc * 16 + 2 * ks];
      }
    };
    frags(af[0], bf[0], 0);
    // at most four steps (uniform guards); the next step's fragment reads are issued ahead of this step's MFMAs
#pragma unroll
    for (int kc = 0; kc < 4; ++kc) {
      if (kc < nk16) {
        if (kc + 1 < 4) frags(af[(kc + 1) & 1], bf[(kc + 1) & 1], min(kc + 1, nk16 - 1));
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
#pragma unroll
          for (int t = 0; t < TN; ++t)
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kc & 1][ks], bf[kc & 1][ks][t], acc[t], 0, 0, 0);
      }
    }
  }
  FUSED_STAMP(a, 2);
  unsigned long long mword = 0ull;                     // relu'(h1): word (t, r) in lane t*16 + r, kept for the backward
#pragma unroll
  for (int t = 0; t < TN; ++t) {
    const int col = wn * WC + t * 32 + li;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = wm * 32 + 4 * lh + rowoff(r);
      const float v = fmaxf(acc[t][r] + b1v[t], 0.f);
      h1s[row * LDH + col] = v;
      const unsigned long long m = __ballot(v > 0.f);
      if (lane == t * 16 + r) mword = m;
      acc[t][r] = 0.f;
    }
  }
  __syncthreads();   // every wave is done with the W1 image: the ring is the ring from here on
  bstore(rb, 0);
  bstore(rb1, 1);
  bload(rb, 2);
  __syncthreads();
  FUSED_STAMP(a, 3);

  // ---- layer 2 (see disc_fwd_kernel); its last iteration requests the backward chain's first W2 chunk
  // the tile's h1 (layer 2) / dh2 (backward chain) leaves for HBM in slices, one per chunk: EVERY thread moves 8 bytes
  // (read from LDS at the top of the iteration, stored behind the MFMAs) -- with 16-byte pieces only half of the waves
  // had the LDS round trip + store at the end of their iteration, and the other half waited for them at the barrier
  static_assert(BM * H / NCH / 2 == NT, "one 8-byte piece per thread and chunk");
  const float* Ar = h1s + (wm * 32 + li) * LDH + lh;
  const int boff = wn * WC + li;
  float fa[2][FB_K / 2], fb[2][FB_K / 2][TN];          // fragments of the chunk in flight / the next one
  {
    auto a_at = [&](int ks) { return Ar[2 * ks]; };
    chunk_frags<H, TN>(fa[0], fb[0], a_at, bs + boff, lh);
  }
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    bstore(rb, c + 2);                                 // (c + 2 >= NCH: the backward chain's first W2 chunks)
    bload(rb, c + 3);
    const int e2 = c * NT + tid;
    const int hrow = e2 / (H / 2), hcol = (e2 % (H / 2)) * 2;
    const float hv0 = h1s[hrow * LDH + hcol], hv1 = h1s[hrow * LDH + hcol + 1];
    if (c + 1 < NCH) {
      auto a_nx = [&](int ks) { return Ar[(c + 1) * FB_K + 2 * ks]; };
      chunk_mfmas_and_next_frags<H, TN>(acc, fa[c & 1], fb[c & 1], fa[(c + 1) & 1], fb[(c + 1) & 1], a_nx,
                                        bs + ((c + 1) % NR) * BST + boff, lh);
    } else {
      chunk_mfmas<TN>(acc, fa[c & 1], fb[c & 1]);
    }
    {
      // UNCONDITIONAL store (rows past R land in a dump slot): behind a branch the compiler cannot count the store in
      // vmcnt and makes the next iteration's ring write wait for every outstanding operation -- i.e. for this store's
      // acknowledgement, ~400 cycles per chunk
      float2 hv; hv.x = hv0; hv.y = hv1;
      float* dst = row0 + hrow < a.R ? a.h1 + (long long)(row0 + hrow) * H + hcol : a.dump + 2 * tid;
      *reinterpret_cast<float2*>(dst) = hv;
    }
    schedule_iteration<TN, BV>();
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
  }
  FUSED_STAMP(a, 4);

  // ---- epilogue: logit, BCE, dlogit, statistics, dW3/db3 partials, dh2 (into the h1 tile)
  float h2[TN][16];
  float p[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < TN; ++t) {
      h2[t][r] = fmaxf(acc[t][r] + b2v[t], 0.f);
      s += h2[t][r] * w3v[t];
    }
    p[r] = s;
  }
  {
    const float tot = reduce16_in_half(p, lane);
    const int r = ((lane >> 4) & 1) * 8 + ((lane >> 3) & 1) * 4 + ((lane >> 2) & 1) * 2 + ((lane >> 1) & 1);
    if ((lane & 1) == 0) red[wn * BM + wm * 32 + 4 * lh + rowoff(r)] = tot;
  }
  __syncthreads();
  FUSED_STAMP(a, 5);
  if (wave == 0) {  // one row per lane (lanes >= BM idle but take part in the reduction)
    const int row = min(lane, BM - 1);
    const int gi = row0 + row;
    const bool valid = lane < BM && gi < a.R;
    const float x = ((red[row] + red[BM + row]) + red[2 * BM + row]) + red[3 * BM + row] + b3v;
    // adversarial/common.py:360-368 + 27-92, the arithmetic of bce_kernel (mlp.hip)
    const float y = gi < a.n_expert ? 1.f : 0.f;
    const float lse = log1pf(expf(-fabsf(x)));
    const float pr = 1.f / (1.f + expf(-x));
    const float inv = a.loss_scale / (float)a.R;
    const float dl = valid ? (pr - y) * inv : 0.f;
    if (lane < BM) dls[lane] = dl;
    if (valid) {
      a.logits[gi] = x;
      if (a.dlogits) a.dlogits[gi] = dl;
    }
    const bool is_gen_pred = x < 0.f, is_gen_true = y == 0.f;
    const bool ok = is_gen_pred == is_gen_true;
    float vals[8];
    vals[0] = valid ? (1.f - y) * x - (fminf(x, 0.f) - lse) : 0.f;
    vals[1] = (valid && ok) ? 1.f : 0.f;
    vals[2] = (valid && ok && !is_gen_true) ? 1.f : 0.f;
    vals[3] = (valid && ok && is_gen_true) ? 1.f : 0.f;
    vals[4] = (valid && is_gen_pred) ? 1.f : 0.f;
    vals[5] = valid ? (1.f - pr) * x - (fminf(x, 0.f) - lse) : 0.f;
    vals[6] = dl;  // db3 partial
    vals[7] = 0.f;
    const float tot = reduce8_in_wave(vals, lane);
    const int k = lane >> 3;
    if ((lane & 7) == 0) {
      if (k < 6) a.part[(long long)bid * 8 + k] = tot;
      else if (k == 6) a.P3[(long long)bid * (2 * H + 1) + H] = tot;
    }
  }
  __syncthreads();
  FUSED_STAMP(a, 6);
  {
    float dlr[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) dlr[r] = dls[wm * 32 + 4 * lh + rowoff(r)];
#pragma unroll
    for (int t = 0; t < TN; ++t) {
      const int col = wn * WC + t * 32 + li;
      float s = 0.f, sb = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * 32 + 4 * lh + rowoff(r);
        s += dlr[r] * h2[t][r];
        const float dv = h2[t][r] > 0.f ? dlr[r] * w3v[t] : 0.f;
        h1s[row * LDH + col] = dv;   // dh2 (the h1 tile is dead by now)
        sb += dv;                    // db2 = column sums of dh2
        acc[t][r] = 0.f;
      }
      s += __shfl_xor(s, 32, 64);
      sb += __shfl_xor(sb, 32, 64);
      if (lh == 0) { w3red[wm * H + col] = s; w3red[(BM / 32 + wm) * H + col] = sb; }
    }
  }
  __syncthreads();
  if (tid < H) {
    float s = w3red[tid], sb = w3red[(BM / 32) * H + tid];
#pragma unroll
    for (int g = 1; g < BM / 32; ++g) { s += w3red[g * H + tid]; sb += w3red[(BM / 32 + g) * H + tid]; }
    a.P3[(long long)bid * (2 * H + 1) + tid] = s;
    a.P3[(long long)bid * (2 * H + 1) + H + 1 + tid] = sb;
  }
  FUSED_STAMP(a, 7);

  // ---- backward chain: dh1 = (dh2 . W2) * relu'(h1), A = the dh2 tile in LDS, B = W2 chunks through the ring; the
  //      tile's dh2 goes out to HBM in slices beside the MFMAs
  {
    auto a_at = [&](int ks) { return Ar[2 * ks]; };
    chunk_frags<H, TN>(fa[0], fb[0], a_at, bs + (NCH % NR) * BST + boff, lh);
  }
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    if (c + 2 < NCH) bstore(rb, NCH + c + 2);
    if (c + 3 < NCH) bload(rb, NCH + c + 3);
    const int e2 = c * NT + tid;
    const int hrow = e2 / (H / 2), hcol = (e2 % (H / 2)) * 2;
    const float hv0 = h1s[hrow * LDH + hcol], hv1 = h1s[hrow * LDH + hcol + 1];
    if (c + 1 < NCH) {
      auto a_nx = [&](int ks) { return Ar[(c + 1) * FB_K + 2 * ks]; };
      chunk_mfmas_and_next_frags<H, TN>(acc, fa[c & 1], fb[c & 1], fa[(c + 1) & 1], fb[(c + 1) & 1], a_nx,
                                        bs + ((NCH + c + 1) % NR) * BST + boff, lh);
    } else {
      chunk_mfmas<TN>(acc, fa[c & 1], fb[c & 1]);
    }
    {
      float2 hv; hv.x = hv0; hv.y = hv1;
      float* dst = row0 + hrow < a.R ? a.dh2 + (long long)(row0 + hrow) * H + hcol : a.dump + 2 * tid;
      *reinterpret_cast<float2*>(dst) = hv;
    }
    schedule_iteration<TN, BV>();
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
  }
  FUSED_STAMP(a, 10);
  {
    const unsigned int hm_lo = (unsigned int)mword, hm_hi = (unsigned int)(mword >> 32);
#pragma unroll
    for (int t = 0; t < TN; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const unsigned int wlo = __builtin_amdgcn_readlane(hm_lo, t * 16 + r), whi = __builtin_amdgcn_readlane(hm_hi, t * 16 + r);
        const bool on = (((lh ? whi : wlo) >> li) & 1u) != 0u;
        h1s[(wm * 32 + 4 * lh + rowoff(r)) * LDH + wn * WC + t * 32 + li] = on ? acc[t][r] : 0.f;
      }
  }
  if (tid < BM) xs[tid * XP3W + D] = 1.f;  // the ones column makes db1 a column of dW1 (column D was a zero column so far)
  __syncthreads();
  FUSED_STAMP(a, 11);

  // ---- [dW1 | db1] partial [H, D + 1] = dh1^T . [xn | 1] over the tile's BM rows (see disc_bwd_kernel); WIDE: the
  //      D + 1 <= 64 columns as two 32-wide blocks
  const long long n1 = (long long)H * D + H;
  float* P1 = a.P1 + (long long)bid * n1;
  for (int mt = wave; mt < H / 32; mt += NW) {
    float af[BM / 2];
#pragma unroll
    for (int s = 0; s < BM / 2; ++s) af[s] = h1s[(2 * s + lh) * LDH + mt * 32 + li];
#pragma unroll
    for (int nb = 0; nb < (WIDE ? 2 : 1); ++nb) {
      f32x16 acc1;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc1[r] = 0.f;
      float bf[BM / 2];
#pragma unroll
      for (int s = 0; s < BM / 2; ++s) bf[s] = xs[(2 * s + lh) * XP3W + nb * 32 + li];
#pragma unroll
      for (int s = 0; s < BM / 2; ++s) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s], bf[s], acc1, 0, 0, 0);
      const int col = nb * 32 + li;
      if (col <= D) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int i = mt * 32 + 4 * lh + rowoff(r);
          bs[col < D ? i * D + col : H * D + i] = acc1[r];
        }
      }
    }
  }
  __syncthreads();
  for (int e = tid; e < (int)(n1 / 4); e += NT)
    reinterpret_cast<f4*>(P1)[e] = reinterpret_cast<const f4*>(bs)[e];
  FUSED_STAMP(a, 12);
}

template <int H, int BM, int DW = 24>
__global__ __launch_bounds__(BM * 8) void disc_fb_kernel(FusedArgs a) { disc_fb_body<H, BM, DW>(a, blockIdx.x); }

// ------------------------------------------------------------------------------------------- gradient penalty, one launch
// The three penalty passes of a 32-row tile (disc_fwd_kernel<H,32,1>, disc_bwd_kernel<H,32,1>, disc_fwd_kernel<H,32,2>)
// in ONE workgroup: what they hand from pass to pass -- relu'(h1), relu'(h2), u2, u1, the row coefficients C, v1 -- is
// per tile, so the masks stay in the registers that took the ballots, C never leaves LDS, and the three chunk loops
// (h1 W2^T, u2 W2, v1 W2^T: 3 NCH chunks of W2T | W2 | W2T) run as one stream through the 3-stage ring with the
// register-double-buffered fragments of disc_fb_kernel. The W1 image stays resident (layer 1, the input gradient u1 W1
// and C W1^T all read it). Only the second pass's GEMM operands u2 and v1 go to HBM, in slices beside the MFMAs of the
// loop that reads them. Same arithmetic, same order as the three launches: bit-identical (`ia_disc_fused_split_tiles(1)`
// keeps them; tests/test_disc_fused_gpu.py compares).
// DW = 64 (rows of up to 64 floats, D <= 63 -- Ant-width GAIL nets, `use_next_state` / `use_done`): layer 1 over K = 64,
// the input gradient and the row coefficients as two 32-column blocks (the four waves: K halves x column blocks, two
// columns per lane in the norm), the first-layer slab straight to HBM (its image would not fit beside the 65-float W1
// rows), and the input gradient's partial tiles in the ring slot the second chunk loop has just finished with.
// G = 2 (H = 256, DW = 24: config P's penalty): TWO 32-row tiles per workgroup -- 512 threads, two waves per SIMD -- that
// share the chunk stream's ring and the W1 image and have their own tile, x tile and partial sums; per tile the same
// instructions in the same order as G = 1 (bit-identical). The input-gradient partial tiles of ONE group at a time sit in the
// ring stage the second chunk loop has just left (rows 32 floats apart to fit its 4 096), the first-layer slab goes straight
// to HBM: 151 KB of LDS. With one 256-thread workgroup per CU (138 KB) the kernel ran at one wave per SIMD, 58 TFLOP/s
// against the BCE tile kernel's 80.
// NWC = 8 (same shape): ONE tile per workgroup, its columns split over EIGHT waves (32 columns each instead of 64) -- 512
// threads, two waves per SIMD, and still one workgroup per tile: B = 8 192 interpolates fill all 256 CUs, where G = 2 leaves
// half of them idle (measured: 80 us per launch against 60-66 for G = 1). The input gradient's four K quarters and the row
// norms stay with waves 0-3 (the same partial tiles, the same sums); every other phase divides by columns or by slab tiles.
template <int H, int DW, int G, int NWC>
__device__ __forceinline__ void disc_gp_body(const FusedArgs& a, const int bid) {
  constexpr int BM = 32, NW = 4, NTG = 64 * NWC, NT = NTG * G;   // NW: waves that share the input-gradient phase; NTG: threads of a row group
  static_assert(G == 1 || (G == 2 && DW == 24 && H == 256 && NWC == 4), "two row groups: the 256-wide narrow-row shape only");
  static_assert(NWC == 4 || (NWC == 8 && G == 1 && H == 256), "eight column groups: the 256-wide shapes only");
  constexpr bool WIDE = DW == 64;
  constexpr int XP = xp_of(DW), XP3 = xp3_of(DW);   // (shadow the narrow row lengths of the file scope)
  constexpr int KS1 = WIDE ? 32 : 12;               // layer 1: k steps of 2
  constexpr int TN = H / (32 * NWC);                  // 32-column MFMA tiles per wave
  constexpr int WC = TN * 32;
  constexpr int LDH = H + 1;
  constexpr int NCH = H / FB_K;
  constexpr int BST = FB_K * H;
  constexpr int BV = BST / 4 / NT;
  constexpr int NR = 3;
  constexpr bool SCR_IN_RING = (WIDE && BST >= 2 * 32 * 64) || G == 2;
  constexpr int GS = G == 2 ? 32 : 33;                // row length of an input-gradient partial tile (narrow rows)
  // gn partial tiles, then (narrow) the first-layer slab image; WIDE: 2 K halves x [32][64] partial tiles only
  constexpr int SCR = SCR_IN_RING ? 0 : (WIDE ? 2 * 32 * 64 : ((H * 25 > NW * 32 * 33) ? H * 25 : NW * 32 * 33));
  static_assert(!SCR_IN_RING || WIDE || NW * 32 * GS <= BST, "a group's partial tiles must fit one ring stage");
  static_assert(BST % (4 * NT) == 0, "B chunk must divide among the threads");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* h1s = smem;                   // G x [BM][LDH]   h1 -> u2 -> u1 -> v1 (this wave's group: below)
  float* bs = h1s + G * BM * LDH;      // NR x [FB_K][H] ring
  float* w1s = bs + NR * BST;          // [H][XP] W1 image, resident
  float* scr = SCR_IN_RING ? bs + ((2 * NCH - 1) % NR) * BST : w1s + H * XP;   // [SCR] (WIDE, H = 256: the slot of the
                                                                               //  second loop's last chunk)
  float* w3red = w1s + H * XP + SCR;   // G x [H]
  float* xs = w3red + G * H;           // G x [BM][XP3]: x_hat (normalised), later the row coefficients C

  const int tid = threadIdx.x, lane = tid & 63;
  const int wg = G == 1 ? 0 : (tid >> 8);        // row group of this wave
  const int wave = (tid >> 6) % NWC;             // wave inside its group
  const int tg = tid & (NTG - 1);                // thread inside its group
  const int li = lane & 31, lh = lane >> 5;
  const int wn = wave;
  const int tile = bid * G + wg;          // the 32-row tile of this group
  const bool tile_ok = G == 1 || tile * BM < a.R;   // (an odd tile count leaves the last workgroup's second group idle)
  const int row0 = tile * BM;
  h1s += wg * BM * LDH;
  w3red += wg * H;
  xs += wg * BM * XP3;
  const int D = a.D;
  const float* W1 = a.params;
  const float* b1 = W1 + (long long)H * D;
  const float* W2 = b1 + H;
  const float* b2 = W2 + (long long)H * H;
  const float* w3 = b2 + H;

  f4 rb[BV], rb1[BV];
  auto bload = [&](f4 (&r)[BV], int g) {
    const float* src = g < NCH ? a.W2T + (long long)g * BST
                               : (g < 2 * NCH ? W2 + (long long)(g - NCH) * BST : a.W2T + (long long)(g - 2 * NCH) * BST);
#pragma unroll
    for (int i = 0; i < BV; ++i) r[i] = *reinterpret_cast<const f4*>(src + (long long)(tid + i * NT) * 4);
  };
  auto bstore = [&](const f4 (&r)[BV], int g) {
    float* S = bs + (g % NR) * BST;
#pragma unroll
    for (int i = 0; i < BV; ++i) *reinterpret_cast<f4*>(S + (tid + i * NT) * 4) = r[i];
  };
  constexpr int W1Q = H * XP / 4;
  constexpr int W1V = (W1Q + NT - 1) / NT;
  f4 w1v[W1V];
#pragma unroll
  for (int i = 0; i < W1V; ++i) w1v[i] = reinterpret_cast<const f4*>(a.W1P)[min(tid + i * NT, W1Q - 1)];
  float b1v[TN], b2v[TN], w3v[TN];
#pragma unroll
  for (int t = 0; t < TN; ++t) {
    const int col = wn * WC + t * 32 + li;
    b1v[t] = b1[col]; b2v[t] = b2[col]; w3v[t] = w3[col];
  }
  load_x_tile<BM, true, false, WIDE>(a, a.X, row0, xs, XP3, tg);
  for (int e = tg; e < BM * (XP3 - 1 - a.ldx); e += NTG) {  // columns [ldx, 32) (WIDE: [ldx, 64))
    const int w = XP3 - 1 - a.ldx;
    const int row = e / w, c = a.ldx + e - row * w;
    xs[row * XP3 + c] = 0.f;
  }
  bload(rb, 0);
  bload(rb1, 1);
#pragma unroll
  for (int i = 0; i < W1V; ++i)
    if (tid + i * NT < W1Q) reinterpret_cast<f4*>(w1s)[tid + i * NT] = w1v[i];
  bstore(rb, 0);
  bstore(rb1, 1);
  bload(rb, 2);
  __syncthreads();

  f32x16 acc[TN];
  // layer-1 shaped product of the x tile with W1 (K = 24 / 64): at x_hat (first pass) and at C (second pass)
  auto layer1 = [&]() {
#pragma unroll
    for (int t = 0; t < TN; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float af[KS1], bf[KS1][TN];
#pragma unroll
    for (int ks = 0; ks < KS1; ++ks) {
      af[ks] = xs[li * XP3 + 2 * ks + lh];
#pragma unroll
      for (int t = 0; t < TN; ++t) bf[ks][t] = w1s[(wn * WC + t * 32 + li) * XP + 2 * ks + lh];
    }
#pragma unroll
    for (int ks = 0; ks < KS1; ++ks)
#pragma unroll
      for (int t = 0; t < TN; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[ks], bf[ks][t], acc[t], 0, 0, 0);
  };
  auto bit_of = [&](unsigned int lo, unsigned int hi, int idx) {
    const unsigned int wlo = __builtin_amdgcn_readlane(lo, idx), whi = __builtin_amdgcn_readlane(hi, idx);
    return (((lh ? whi : wlo) >> li) & 1u) != 0u;
  };

  // ---- h1 = relu(x_hat W1^T + b1); relu'(h1) ballots (word (t, r) in lane t*16 + r)
  layer1();
  unsigned long long mword1 = 0ull;
#pragma unroll
  for (int t = 0; t < TN; ++t) {
    const int col = wn * WC + t * 32 + li;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = 4 * lh + rowoff(r);
      const float v = fmaxf(acc[t][r] + b1v[t], 0.f);
      h1s[row * LDH + col] = v;
      const unsigned long long m = __ballot(v > 0.f);
      if (lane == t * 16 + r) mword1 = m;
      acc[t][r] = 0.f;
    }
  }
  const unsigned int m1_lo = (unsigned int)mword1, m1_hi = (unsigned int)(mword1 >> 32);
  __syncthreads();

  // ---- one chunk loop of the stream: acc += tile(h1s) . chunk(g0 + c); the tile's rows go to `out` in slices (or not)
  constexpr int PW = BM * H / NCH / NTG;            // floats of the tile a thread sends to HBM per chunk (2; 8 waves: 1)
  static_assert(PW * NTG * NCH == BM * H && (PW == 1 || PW == 2), "the tile leaves in NCH slices of PW floats per thread");
  const float* Ar = h1s + li * LDH + lh;
  const int boff = wn * WC + li;
  float fa[2][FB_K / 2], fb[2][FB_K / 2][TN];
  auto chunk_loop = [&](const int g0, float* __restrict__ out) {
    {
      auto a_at = [&](int ks) { return Ar[2 * ks]; };
      chunk_frags<H, TN>(fa[0], fb[0], a_at, bs + (g0 % NR) * BST + boff, lh);
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int g = g0 + c;
      if (g + 2 < 3 * NCH) bstore(rb, g + 2);
      if (g + 3 < 3 * NCH) bload(rb, g + 3);
      const int e2 = c * NTG + tg;
      const int hrow = e2 / (H / PW), hcol = (e2 % (H / PW)) * PW;
      float hv0 = 0.f, hv1 = 0.f;
      if (out != nullptr) { hv0 = h1s[hrow * LDH + hcol]; if constexpr (PW == 2) hv1 = h1s[hrow * LDH + hcol + 1]; }
      if (c + 1 < NCH) {
        auto a_nx = [&](int ks) { return Ar[(c + 1) * FB_K + 2 * ks]; };
        chunk_mfmas_and_next_frags<H, TN>(acc, fa[c & 1], fb[c & 1], fa[(c + 1) & 1], fb[(c + 1) & 1], a_nx,
                                          bs + ((g + 1) % NR) * BST + boff, lh);
      } else {
        chunk_mfmas<TN>(acc, fa[c & 1], fb[c & 1]);
      }
      if (out != nullptr) {   // (wave-uniform; the store itself is unconditional: rows past R land in the dump slot)
        float* dst = row0 + hrow < a.R ? out + (long long)(row0 + hrow) * H + hcol : a.dump + PW * tid;
        if constexpr (PW == 2) { float2 hv; hv.x = hv0; hv.y = hv1; *reinterpret_cast<float2*>(dst) = hv; }
        else *dst = hv0;
      }
      __builtin_amdgcn_sched_barrier(0);
      __syncthreads();
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // ---- z2 = h1 W2^T + b2: relu'(h2) ballots, u2 = relu'(h2) w3 (rows past R: 0) into the tile
  chunk_loop(0, nullptr);
  unsigned long long mword2 = 0ull;
#pragma unroll
  for (int t = 0; t < TN; ++t) {
    const int col = wn * WC + t * 32 + li;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = 4 * lh + rowoff(r);
      const bool on = fmaxf(acc[t][r] + b2v[t], 0.f) > 0.f;
      h1s[row * LDH + col] = (on && row0 + row < a.R) ? w3v[t] : 0.f;
      const unsigned long long m = __ballot(on);
      if (lane == t * 16 + r) mword2 = m;
      acc[t][r] = 0.f;
    }
  }
  const unsigned int m2_lo = (unsigned int)mword2, m2_hi = (unsigned int)(mword2 >> 32);
  __syncthreads();

  // ---- u1 = relu'(h1) (u2 W2); u2 leaves for HBM beside the MFMAs
  chunk_loop(NCH, a.dh2);
#pragma unroll
  for (int t = 0; t < TN; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const bool on = bit_of(m1_lo, m1_hi, t * 16 + r);
      h1s[(4 * lh + rowoff(r)) * LDH + wn * WC + t * 32 + li] = on ? acc[t][r] : 0.f;
    }
  __syncthreads();

  // ---- gn = u1 W1 (K = H split over the four waves), row coefficients -> xs := C, penalty partial (disc_bwd_kernel MODE 1)
  if constexpr (WIDE) {
   if (NWC == NW || wave < NW) {   // (eight column waves: the two K halves x two column blocks stay with waves 0-3)
    constexpr int KH = H / 2;
    const int kh = wave >> 1, col = (wave & 1) * 32 + li;
    f32x16 accg;
#pragma unroll
    for (int r = 0; r < 16; ++r) accg[r] = 0.f;
    const bool cok = col < D;
#pragma unroll 8
    for (int s2 = 0; s2 < KH / 2; ++s2) {
      const int k = kh * KH + 2 * s2 + lh;
      const float af = h1s[li * LDH + k];
      const float w1 = w1s[k * XP + col];
      accg = __builtin_amdgcn_mfma_f32_32x32x2f32(af, cok ? w1 : 0.f, accg, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) scr[(kh * 32 + 4 * lh + rowoff(r)) * 64 + col] = accg[r];
   }
  }
  // narrow rows: the four K quarters of a group's input gradient as partial tiles in `scr` (rows GS floats apart) ...
  auto gn_partials = [&]() {
    constexpr int KQ = H / NW;
    f32x16 accg;
#pragma unroll
    for (int r = 0; r < 16; ++r) accg[r] = 0.f;
    const bool cok = li < D;
#pragma unroll 8
    for (int s2 = 0; s2 < KQ / 2; ++s2) {
      const int k = wave * KQ + 2 * s2 + lh;
      const float af = h1s[li * LDH + k];
      const float w1 = w1s[k * XP + min(li, XP - 1)];
      accg = __builtin_amdgcn_mfma_f32_32x32x2f32(af, cok ? w1 : 0.f, accg, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) scr[(wave * 32 + 4 * lh + rowoff(r)) * GS + li] = accg[r];
  };
  if constexpr (!WIDE && G == 1) { if (NWC == NW || wave < NW) gn_partials(); }
  if constexpr (G == 1) __syncthreads();
  float pen_w = 0.f;
  if constexpr (WIDE) {
   if (NWC == NW || wave < NW) {
    const int c1 = 32 + li;
    const float inv0 = (a.mean != nullptr && li < D) ? 1.f / sqrtf(a.var[min(li, D - 1)] + a.eps) : 1.f;
    const float inv1 = (a.mean != nullptr && c1 < D) ? 1.f / sqrtf(a.var[min(c1, D - 1)] + a.eps) : 1.f;
#pragma unroll
    for (int it = 0; it < 32 / (2 * NW); ++it) {
      const int row = wave * (32 / NW) + 2 * it + lh;
      const float gn0 = scr[row * 64 + li] + scr[(32 + row) * 64 + li];
      const float gn1 = scr[row * 64 + c1] + scr[(32 + row) * 64 + c1];
      const float g0 = li < D ? gn0 * inv0 : 0.f, g1 = c1 < D ? gn1 * inv1 : 0.f;
      float sq = g0 * g0 + g1 * g1;
      sq += __shfl_xor(sq, 16, 64); sq += __shfl_xor(sq, 8, 64); sq += __shfl_xor(sq, 4, 64);
      sq += __shfl_xor(sq, 2, 64); sq += __shfl_xor(sq, 1, 64);
      const float n = sqrtf(sq);
      const bool valid = row0 + row < a.R;
      const float kk = (valid && n > 0.f) ? a.gp_coef / (float)a.R * 2.f * (n - a.gp_target) / n : 0.f;
      xs[row * XP3 + li] = li < D ? kk * gn0 * inv0 * inv0 : 0.f;
      xs[row * XP3 + c1] = c1 < D ? kk * gn1 * inv1 * inv1 : 0.f;
      const float pr = valid ? (n - a.gp_target) * (n - a.gp_target) : 0.f;
      pen_w += __shfl(pr, 0, 64) + __shfl(pr, 32, 64);
    }
   }
  }
  // ... and the rows' norms, coefficients C (into the x tile) and penalty terms from the partial tiles' fixed-order sums
  auto gn_rows = [&]() {
    float pw = 0.f;
    const float inv = (a.mean != nullptr && li < D) ? 1.f / sqrtf(a.var[min(li, D - 1)] + a.eps) : 1.f;
#pragma unroll
    for (int it = 0; it < 32 / (2 * NW); ++it) {
      const int row = wave * (32 / NW) + 2 * it + lh;
      const float gn = ((scr[row * GS + li] + scr[(32 + row) * GS + li]) + scr[(64 + row) * GS + li]) +
                       scr[(96 + row) * GS + li];
      const float g = li < D ? gn * inv : 0.f;
      float sq = g * g;
      sq += __shfl_xor(sq, 16, 64); sq += __shfl_xor(sq, 8, 64); sq += __shfl_xor(sq, 4, 64);
      sq += __shfl_xor(sq, 2, 64); sq += __shfl_xor(sq, 1, 64);
      const float n = sqrtf(sq);
      const bool valid = row0 + row < a.R;
      const float kk = (valid && n > 0.f) ? a.gp_coef / (float)a.R * 2.f * (n - a.gp_target) / n : 0.f;
      xs[row * XP3 + li] = li < D ? kk * gn * inv * inv : 0.f;
      const float pr = valid ? (n - a.gp_target) * (n - a.gp_target) : 0.f;
      pw += __shfl(pr, 0, 64) + __shfl(pr, 32, 64);
    }
    return pw;
  };
  if constexpr (!WIDE && G == 1) { if (NWC == NW || wave < NW) pen_w = gn_rows(); }
  if constexpr (G > 1) {
    // one group at a time through the one free ring stage (every barrier is the whole workgroup's)
#pragma unroll
    for (int gsel = 0; gsel < G; ++gsel) {
      if (wg == gsel) gn_partials();
      __syncthreads();
      if (wg == gsel) pen_w = gn_rows();
      __syncthreads();
    }
  }
  constexpr int PC = XP3 - 1;   // the spare last column of the tile
  if (lane == 0 && (NWC == NW || wave < NW)) xs[wave * XP3 + PC] = pen_w;
  __syncthreads();
  if (tg == 0 && tile_ok) a.gp_pen[tile] = ((xs[PC] + xs[XP3 + PC]) + xs[2 * XP3 + PC]) + xs[3 * XP3 + PC];

  // ---- first-layer slab [dW1 | 0] = u1^T . C (no bias column), image in `scr`
  {
    const long long n1 = (long long)H * D + H;
    float* P1 = a.P1 + (long long)tile * n1;
    __syncthreads();   // (the gn partial tiles in `scr` have been read)
    for (int mt = wave; mt < H / 32; mt += NWC) {
      float af[BM / 2];
#pragma unroll
      for (int s = 0; s < BM / 2; ++s) af[s] = h1s[(2 * s + lh) * LDH + mt * 32 + li];
#pragma unroll
      for (int nb = 0; nb < (WIDE ? 2 : 1); ++nb) {
        f32x16 acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc1[r] = 0.f;
        float bf[BM / 2];
#pragma unroll
        for (int s = 0; s < BM / 2; ++s) bf[s] = xs[(2 * s + lh) * XP3 + nb * 32 + li];
#pragma unroll
        for (int s = 0; s < BM / 2; ++s) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s], bf[s], acc1, 0, 0, 0);
        const int col = nb * 32 + li;
        if (col <= D && tile_ok) {   // (column D: the bias column of the slab, 0 here -- column D of C is 0)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int i = mt * 32 + 4 * lh + rowoff(r);
            ((WIDE || G > 1) ? P1 : scr)[col < D ? i * D + col : H * D + i] = acc1[r];
          }
        }
      }
    }
    if constexpr (!WIDE && G == 1) {
      __syncthreads();
      for (int e = tid; e < (int)(n1 / 4); e += NT)
        reinterpret_cast<f4*>(P1)[e] = reinterpret_cast<const f4*>(scr)[e];
    } else {
      __syncthreads();   // (the slab went straight to HBM: every wave is past its reads of u1 before v1 overwrites the tile)
    }
  }

  // ---- v1 = relu'(h1) (C W1^T) into the tile (every wave is past its reads of u1: the barrier above)
  layer1();
#pragma unroll
  for (int t = 0; t < TN; ++t) {
    const int col = wn * WC + t * 32 + li;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const bool on = bit_of(m1_lo, m1_hi, t * 16 + r);
      h1s[(4 * lh + rowoff(r)) * LDH + col] = on ? acc[t][r] : 0.f;
      acc[t][r] = 0.f;
    }
  }
  __syncthreads();

  // ---- t = v1 W2^T; v1 leaves for HBM beside the MFMAs; last-layer slab = column sums of relu'(h2) t
  chunk_loop(2 * NCH, a.h1);
#pragma unroll
  for (int t = 0; t < TN; ++t) {
    const int col = wn * WC + t * 32 + li;
    float sum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) sum += bit_of(m2_lo, m2_hi, t * 16 + r) ? acc[t][r] : 0.f;
    sum += __shfl_xor(sum, 32, 64);
    if (lh == 0) w3red[col] = sum;
  }
  __syncthreads();
  if (tg < H && tile_ok) a.P3[(long long)tile * (2 * H + 1) + tg] = w3red[tg];
  if (tg == 0 && tile_ok) a.P3[(long long)tile * (2 * H + 1) + H] = 0.f;
}

template <int H, int DW = 24, int G = 1, int NWC = 4>
__global__ __launch_bounds__(64 * NWC * G) void disc_gp_kernel(FusedArgs a) { disc_gp_body<H, DW, G, NWC>(a, blockIdx.x); }

// The update's tile pass AND the penalty's in ONE launch (256-wide stack, rows of up to 24 or up to 64 floats, 64-row update tiles,
// the penalty's eight-column-wave form: both bodies are 512-thread workgroups of one tile). The penalty's pass reads the batch, the
// statistics and the parameters -- nothing the update's pass writes. Each pass is about ONE wave of 256 workgroups on a chip
// where the PPO kernel holds 19 compute units: as launches of their own each pays for a second, nearly empty wave (alone
// 51 / 48 us, beside the PPO kernel 65 / 61); in one launch the stragglers of one fill the other's gaps. The penalty's
// tiles take the first block ids. Same bodies, same operands: bit-identical per tile.
template <int H, int DW>
__global__ __launch_bounds__(512) void disc_fb_gp_kernel(FusedArgs fa, FusedArgs ga, int n_gp) {
  if ((int)blockIdx.x < n_gp) disc_gp_body<H, DW, 1, 8>(ga, blockIdx.x);
  else disc_fb_body<H, 64, DW>(fa, blockIdx.x - n_gp);
}

// ------------------------------------------------------------------------------------------- K1
struct AssembleArgs {
  const float* obs[2]; const float* act_f32[2]; const int64_t* act_i64[2]; const float* next[2];
  const uint8_t* done[2]; const int64_t* idx[2]; int n[2];
  int obs_dim, act_dim, use_state, use_action, use_next, use_done;
  float* X; int ldx; int R; int D;
  int update_norm;                       // slab moments + merge
  float* rn_ws;                          // [slabs][2][D]
  float* mean; float* var; int32_t* count;
  float* pmean; float* pvar; int32_t* pcount; int pdim;   // second norm over the first pdim columns
  unsigned int* ticket;
  const float* W2; float* W2T; int H;    // transposer blocks (blockIdx >= slabs)
  const float* W1; float* W1P; int xp;   // last block: W1 [H][D] -> [H][xp], zero padded
  // a whole round in one launch: blockIdx.y = update k; its index rows / X / slab moments sit k strides further
  long long idx_stride, x_stride, rn_stride;
};

constexpr int AS_NT = 1024;

__global__ __launch_bounds__(AS_NT) void disc_assemble_kernel(AssembleArgs a) {
  __shared__ __attribute__((aligned(16))) float xt[RN_ROWS_PER_BLOCK * 24];  // the slab, X's own layout (ldx <= 24)
  __shared__ float red[8][33];
  __shared__ int s_last;
  const int tid = threadIdx.x;
  const int slabs = (a.R + RN_ROWS_PER_BLOCK - 1) / RN_ROWS_PER_BLOCK;
  if (blockIdx.y > 0) {
    const long long k = blockIdx.y;
    if (a.idx[0]) a.idx[0] += k * a.idx_stride;
    if (a.idx[1]) a.idx[1] += k * a.idx_stride;
    a.X += k * a.x_stride;
    a.rn_ws += k * a.rn_stride;
  }
  if (a.W1P != nullptr && blockIdx.x == gridDim.x - 1) {
    for (int e = tid; e < a.H * a.xp; e += AS_NT) {
      const int n = e / a.xp, k = e - n * a.xp;
      a.W1P[e] = k < a.D ? a.W1[n * a.D + k] : 0.f;
    }
    return;
  }
  if ((int)blockIdx.x >= slabs) {
    // ---- W2 [H][H] -> W2T, one 64x64 tile per block through LDS (stride 65)
    float* tt = xt;  // 64*65 = 4160 floats <= 6144
    const int tb = blockIdx.x - slabs, tpr = a.H / 64;
    const int r0 = (tb / tpr) * 64, c0 = (tb % tpr) * 64;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = tid + i * AS_NT, r = e >> 6, c = e & 63;
      tt[r * 65 + c] = a.W2[(long long)(r0 + r) * a.H + c0 + c];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = tid + i * AS_NT, r = e >> 6, c = e & 63;
      a.W2T[(long long)(c0 + r) * a.H + r0 + c] = tt[c * 65 + r];
    }
    return;
  }
  const int r0 = blockIdx.x * RN_ROWS_PER_BLOCK;
  const int rows = min(RN_ROWS_PER_BLOCK, a.R - r0);
  const int ldx = a.ldx, D = a.D;
  {
    // four threads per row; thread q takes columns q, q+4, ... (adversarial/common.py:592-603 +
    // rewards/reward_nets.py:441-457, the element rule of gather_concat_kernel). Every load is
    // unconditional at a clamped address so that the batch is in flight at once.
    const int row = tid >> 2, q = tid & 3;
    const int gi = min(r0 + row, a.R - 1);
    const int s = gi >= a.n[0] ? 1 : 0;
    const int li = gi - (s ? a.n[0] : 0);
    const long long src = a.idx[s] ? a.idx[s][li] : (long long)li;
    const long long av = a.act_i64[s] ? a.act_i64[s][src] : 0;
    const float dn = (a.use_done && a.done[s][src]) ? 1.f : 0.f;
    constexpr int NC = 6;  // columns per thread (ldx <= 24)
    float v[NC];
    int kind[NC], oo[NC];
#pragma unroll
    for (int j = 0; j < NC; ++j) {
      const int c = min(q + 4 * j, ldx - 1);
      int o = c, k = 0;  // 0 zero, 1 f32 load, 2 one-hot, 3 done
      const float* p = a.obs[s];
      if (a.use_state) {
        if (o < a.obs_dim) { k = 1; p = a.obs[s] + src * a.obs_dim + o; }
        o -= a.obs_dim;
      }
      if (k == 0 && a.use_action) {
        if (o >= 0 && o < a.act_dim) {
          if (a.act_i64[s]) { k = 2; }
          else { k = 1; p = a.act_f32[s] + src * a.act_dim + o; }
        }
        if (k == 0) o -= a.act_dim;
      }
      if (k == 0 && a.use_next) {
        if (o >= 0 && o < a.obs_dim) { k = 1; p = a.next[s] + src * a.obs_dim + o; }
        if (k == 0) o -= a.obs_dim;
      }
      if (k == 0 && a.use_done && o == 0) k = 3;
      kind[j] = k; oo[j] = o;
      v[j] = *p;
    }
#pragma unroll
    for (int j = 0; j < NC; ++j) {
      const int c = q + 4 * j;
      if (c < ldx) {
        float x = 0.f;
        if (kind[j] == 1) x = v[j];
        else if (kind[j] == 2) x = (av == oo[j]) ? 1.f : 0.f;
        else if (kind[j] == 3) x = dn;
        xt[row * ldx + c] = (r0 + row < a.R) ? x : 0.f;
      }
    }
  }
  __syncthreads();
  // the slab in LDS has X's own layout: coalesced 16-byte copy out
  for (int e = tid; e < rows * ldx / 4; e += AS_NT)
    reinterpret_cast<f4*>(a.X + (long long)r0 * ldx)[e] = reinterpret_cast<const f4*>(xt)[e];
  if (!a.update_norm) return;

  // ---- slab moments: the arithmetic (and summation order) of rn_partial_kernel, reading the LDS copy
  {
    const bool act = tid < 256;
    const int cl = tid & 31, rl = (tid >> 5) & 7;
    constexpr int RPT = RN_ROWS_PER_BLOCK / 8;
    for (int c0 = 0; c0 < D; c0 += 32) {
      const int c = c0 + cl;
      float v[RPT];
      float mean = 0.f;
      if (act) {
#pragma unroll
        for (int k = 0; k < RPT; ++k) v[k] = xt[min(rl + 8 * k, rows - 1) * ldx + min(c, D - 1)];
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < RPT; ++k) s += (c < D && rl + 8 * k < rows) ? v[k] : 0.f;
        red[rl][cl] = s;
      }
      __syncthreads();
      if (act && rl == 0) {
        float t = 0.f;
        for (int k = 0; k < 8; ++k) t += red[k][cl];
        red[0][cl] = t / (float)rows;
      }
      __syncthreads();
      if (act) mean = red[0][cl];
      __syncthreads();
      if (act) {
        float q = 0.f;
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
          const float dlt = v[k] - mean;
          q += (c < D && rl + 8 * k < rows) ? dlt * dlt : 0.f;
        }
        red[rl][cl] = q;
      }
      __syncthreads();
      if (act && rl == 0 && c < D) {
        float t = 0.f;
        for (int k = 0; k < 8; ++k) t += red[k][cl];
        a.rn_ws[((long long)blockIdx.x * 2 + 0) * D + c] = mean;
        a.rn_ws[((long long)blockIdx.x * 2 + 1) * D + c] = t;  // M2 of the slab
      }
      __syncthreads();
    }
  }
  if (a.mean == nullptr) return;
  // ---- last slab block merges (util/networks.py:111-134). Hand-off per the gfx950 rules: plain stores ->
  // __syncthreads -> one-lane agent-scope release (+ explicit vmcnt(0)) -> relaxed ticket; last block:
  // one-lane agent-scope acquire -> __syncthreads -> plain loads.
  if (tid == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned int t = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = (t == (unsigned int)slabs - 1u);
    if (s_last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
  if (!s_last) return;
  const int cnt = *a.count;
  const int pcnt = a.pcount ? *a.pcount : 0;
  __syncthreads();
  const int lane = tid & 63, wave = tid >> 6;
  const int pdim = a.pmean ? a.pdim : 0;
  for (int job = wave; job < D + pdim; job += AS_NT / 64) {
    const bool isp = job >= D;
    const int c = isp ? job - D : job;
    float b_mean, b_M2;
    rn_wave_batch_moments(a.rn_ws, slabs, slabs, a.R, D, c, lane, b_mean, b_M2);
    if (lane == 0) {
      float* mp = isp ? a.pmean : a.mean;
      float* vp = isp ? a.pvar : a.var;
      float mc = mp[c], vc = vp[c];
      rn_absorb(mc, vc, isp ? pcnt : cnt, a.R, b_mean, b_M2 / (float)a.R);
      mp[c] = mc;
      vp[c] = vc;
    }
  }
  if (tid == 0) {
    *a.count = rn_count_add(cnt, a.R);
    if (pdim > 0) *a.pcount = rn_count_add(pcnt, a.R);
    __hip_atomic_store(a.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// ------------------------------------------------------------------------------------------- K5
// (disc_reduce.h: ReduceArgs, disc_reduce_block)
__global__ __launch_bounds__(256) void disc_reduce_kernel(ReduceArgs a) { disc_reduce_block(a, blockIdx.x, gridDim.x); }

// ------------------------------------------------------------------------------------------- K5b
// Data-parallel form of the tail of K5: the slab reduction has left this rank's gradient in `grads` (K5 with adam = 0),
// ONE all-reduce (sum) over the ranks has run on it, and this launch finishes the update: grads *= gscale (1 / world: the
// rank mean, adversarial/common.py:352-373 on the concatenated batch), torch.optim.Adam's step, and the refresh of the
// W2T / padded-W1 images the tile kernels of the NEXT pre-assembled update read.
struct AdamRefreshArgs {
  long long n; float* grads; float gscale;
  float* p; float* m; float* v; float beta1, beta2, eps, wd, step_size, bc2_sqrt;
  float* W2T; float* W1P; int H; int D; int xp;
};

__global__ __launch_bounds__(256) void disc_adam_refresh_kernel(AdamRefreshArgs a) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= a.n) return;
  float grad = a.grads[i] * a.gscale;
  a.grads[i] = grad;
  // torch/optim/adam.py _single_tensor_adam: lerp, mul+addcmul, sqrt/bc2_sqrt + eps, addcdiv (K5's expressions)
  const float pi = a.p[i];
  if (a.wd != 0.f) grad = grad + a.wd * pi;
  float mi = a.m[i];
  mi = mi + (grad - mi) * (1.f - a.beta1);
  const float vi = a.v[i] * a.beta2 + (1.f - a.beta2) * grad * grad;
  const float denom = sqrtf(vi) / a.bc2_sqrt + a.eps;
  const float pn = pi - a.step_size * (mi / denom);
  a.p[i] = pn;
  a.m[i] = mi;
  a.v[i] = vi;
  const long long nW1 = (long long)a.H * a.D, n1 = nW1 + a.H;
  if (i < nW1) {
    const int n = (int)(i / a.D), k = (int)(i - (long long)n * a.D);
    a.W1P[n * a.xp + k] = pn;
  } else if (i >= n1 && i < n1 + (long long)a.H * a.H) {
    const int j = (int)(i - n1), r = j / a.H, c = j - r * a.H;
    a.W2T[(long long)c * a.H + r] = pn;
  }
}

inline int cdivi(long long a, long long b) { return (int)((a + b - 1) / b); }

// The input width class of a stack the tile kernels cover: 24 (D <= 24, every kernel of this file incl. the penalty's),
// 64 (D <= 63 in rows of up to 64 floats -- use_next_state / use_done nets, Ant-sized GAIL inputs: the one-launch tile
// pass `disc_fb_kernel<H, 64, 64>`, assembled by the pass kernel of airl_fused.hip), 0: not covered.
inline int fused_dw(const ia_mlp_desc* d, int ldx) {
  if (!d || d->n_layers != 3 || d->hidden_act != IA_ACT_RELU) return 0;
  const int D = d->dims[0], H = d->dims[1];
  if (d->dims[2] != H || d->dims[3] != 1) return 0;
  if (H != 128 && H != 256) return 0;
  if (D < 1 || ldx < D || ldx % 4 != 0) return 0;
  if (D <= 24 && ldx <= 24) return 24;
  return (D <= 63 && ldx <= 64) ? 64 : 0;
}
inline bool fused_shape_ok(const ia_mlp_desc* d, int ldx) { return fused_dw(d, ldx) != 0; }

struct FusedWs { float* P1; float* P3; float* part; float* W2T; float* W1P; unsigned long long* h1mask; unsigned int* ticket; float* dump; long long total; };

inline FusedWs fused_ws_layout(const ia_mlp_desc* d, int R, float* base) {
  const long long D = d->dims[0], H = d->dims[1];
  const int xp = D <= 24 ? XP : xp_of(64);
  const long long tiles = cdivi(R, 32);   // sized for 32-row tiles (64-row tiles use half of it)
  FusedWs w;
  long long o = 0;
  w.P1 = base + o; o += tiles * (H * D + H);
  w.P3 = base + o; o += tiles * (2 * H + 1);
  w.part = base + o; o += tiles * 8;
  o = (o + 3) / 4 * 4;                     // 16-byte aligned W2T rows
  w.W2T = base + o; o += H * H;
  w.W1P = base + o; o += H * xp; o = (o + 3) / 4 * 4;
  // (64-row tiles write 8 waves' words per tile: 2 * ceil(R / 64) half-tiles, one more than ceil(R / 32) when that is odd)
  w.h1mask = reinterpret_cast<unsigned long long*>(base + o); o += 2 * (long long)cdivi(R, 64) * 4 * (H / 128) * 16 * 2;
  w.ticket = reinterpret_cast<unsigned int*>(base + o); o += 4;
  w.dump = base + o; o += 1024;
  w.total = o;
  return w;
}

// Workspace of the fused gradient penalty over B interpolated rows (32-row tiles): the second pass's operands
// v1 / u2 [B, H], the row coefficients C [B, ldx], both passes' mask ballots, per-tile partial slabs of the first and
// last layer, penalty partial sums, split-K slabs of dW2 (the bias entries of those slabs are never written: the
// caller zeroes the workspace once and they stay zero).
struct GpWs {
  float* v1; float* u2; float* C; unsigned long long* m1; unsigned long long* m2; float* P1; float* P3; float* pen;
  float* partials; int splits; long long total;
};

inline GpWs gp_ws_layout(const ia_mlp_desc* d, int B, int ldx, float* base) {
  const long long D = d->dims[0], H = d->dims[1];
  const long long tiles = cdivi(B, 32), tot = H * D + H + H * H + H + H + 1;
  GpWs w;
  long long o = 0;
  w.v1 = base + o; o += (long long)B * H;
  w.u2 = base + o; o += (long long)B * H;
  w.C = base + o; o += (long long)B * ldx;
  o = (o + 3) / 4 * 4;
  w.m1 = reinterpret_cast<unsigned long long*>(base + o); o += tiles * 4 * (H / 128) * 16 * 2;
  w.m2 = reinterpret_cast<unsigned long long*>(base + o); o += tiles * 4 * (H / 128) * 16 * 2;
  w.P1 = base + o; o += tiles * (H * D + H);
  w.P3 = base + o; o += tiles * (2 * H + 1);
  w.pen = base + o; o += tiles;
  o = (o + 3) / 4 * 4;
  w.splits = B >= 8192 ? 32 : (B >= 256 ? B / 256 : 1);
  w.partials = base + o; o += w.splits * tot;
  w.total = o;
  return w;
}

int g_fused_bm = 64;   // rows per tile workgroup (tuning: ia_disc_fused_tile_rows)
bool g_fused_split = false;   // forward and backward tile passes as two launches (ia_disc_fused_split_tiles)
bool g_side_reduce = true;    // the product-independent part of the closing reduction inside the split-K product's launch
int g_gp_groups = 8;          // form of the one-launch penalty pass at H = 256 (ia_disc_fused_gp_groups): 8 = one tile, eight column
                              // waves; 2 = two tiles per workgroup; 1 = one tile, four waves
template <int H>
int launch_gp_tiles_wide(const FusedArgs& ga, int B, hipStream_t stream) {
  constexpr int BM = 32, XPW = xp_of(64), XP3W = xp3_of(64);
  constexpr int SCR = FB_K * H >= 2 * 32 * 64 ? 0 : 2 * 32 * 64;
  constexpr size_t smem_g = sizeof(float) * (BM * (H + 1) + 3 * FB_K * H + H * XPW + SCR + H + BM * XP3W);
  static_assert(smem_g <= 160 * 1024, "the wide penalty tile must fit one CU's LDS");
  static bool attr_g = false;
  if (!attr_g) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(disc_gp_kernel<H, 64>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_g);
    if (e != hipSuccess) return (int)e;
    attr_g = true;
  }
  if constexpr (H == 256) {
    if (g_gp_groups == 8) {   // columns over eight waves (two waves per SIMD), as the narrow-row pass
      static bool attr_g8 = false;
      if (!attr_g8) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(disc_gp_kernel<H, 64, 1, 8>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_g);
        if (e != hipSuccess) return (int)e;
        attr_g8 = true;
      }
      hipLaunchKernelGGL((disc_gp_kernel<H, 64, 1, 8>), dim3(cdivi(B, BM)), dim3(512), smem_g, stream, ga);
      IA_CHECK_LAUNCH();
      return IA_OK;
    }
  }
  hipLaunchKernelGGL((disc_gp_kernel<H, 64>), dim3(cdivi(B, BM)), dim3(256), smem_g, stream, ga);
  IA_CHECK_LAUNCH();
  return IA_OK;
}

template <int H>
int launch_gp_tiles(const FusedArgs& ga, int B, hipStream_t stream) {
  constexpr int BM = 32;
  constexpr size_t smem_f = sizeof(float) * (BM * (H + 1) + NS * FB_K * H + 4 * BM + BM + 2 * (BM / 32) * H + BM * XP);
  constexpr size_t smem_b = sizeof(float) * (BM * XP3 + BM * (H + 1) + NS * FB_K * H + NS * BM * A_LD);
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(disc_fwd_kernel<H, BM, 1>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_f);
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(disc_fwd_kernel<H, BM, 2>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_f);
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(disc_bwd_kernel<H, BM, 1>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_b);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  const int tiles = cdivi(B, BM);
  if constexpr (H == 256) {
    if (!g_fused_split && g_gp_groups == 8) {   // one tile per 512-thread workgroup, columns over eight waves (disc_gp_kernel)
      constexpr int SCR8 = (H * 25 > 4 * 32 * 33) ? H * 25 : 4 * 32 * 33;
      constexpr size_t smem_g8 = sizeof(float) * (BM * (H + 1) + 3 * FB_K * H + H * XP + SCR8 + H + BM * XP3);
      static bool attr_g8 = false;
      if (!attr_g8) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(disc_gp_kernel<H, 24, 1, 8>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_g8);
        if (e != hipSuccess) return (int)e;
        attr_g8 = true;
      }
      hipLaunchKernelGGL((disc_gp_kernel<H, 24, 1, 8>), dim3(tiles), dim3(512), smem_g8, stream, ga);
      IA_CHECK_LAUNCH();
      return IA_OK;
    }
    if (!g_fused_split && g_gp_groups == 2) {   // two 32-row tiles per 512-thread workgroup: two waves per SIMD (disc_gp_kernel)
      constexpr size_t smem_g2 = sizeof(float) * (2 * BM * (H + 1) + 3 * FB_K * H + H * XP + 2 * H + 2 * BM * XP3);
      static_assert(smem_g2 <= 160 * 1024, "the two-group penalty workgroup must fit one CU's LDS");
      static bool attr_g2 = false;
      if (!attr_g2) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(disc_gp_kernel<H, 24, 2>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_g2);
        if (e != hipSuccess) return (int)e;
        attr_g2 = true;
      }
      hipLaunchKernelGGL((disc_gp_kernel<H, 24, 2>), dim3(cdivi(tiles, 2)), dim3(512), smem_g2, stream, ga);
      IA_CHECK_LAUNCH();
      return IA_OK;
    }
  }
  if (!g_fused_split) {
    constexpr int SCR = (H * 25 > 4 * 32 * 33) ? H * 25 : 4 * 32 * 33;
    constexpr size_t smem_g = sizeof(float) * (BM * (H + 1) + 3 * FB_K * H + H * XP + SCR + H + BM * XP3);
    static bool attr_g = false;
    if (!attr_g) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(disc_gp_kernel<H>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_g);
      if (e != hipSuccess) return (int)e;
      attr_g = true;
    }
    hipLaunchKernelGGL((disc_gp_kernel<H>), dim3(tiles), dim3(256), smem_g, stream, ga);
    IA_CHECK_LAUNCH();
    return IA_OK;
  }
  hipLaunchKernelGGL((disc_fwd_kernel<H, BM, 1>), dim3(tiles), dim3(BM * 8), smem_f, stream, ga);
  IA_CHECK_LAUNCH();
  hipLaunchKernelGGL((disc_bwd_kernel<H, BM, 1>), dim3(tiles), dim3(BM * 8), smem_b, stream, ga);
  IA_CHECK_LAUNCH();
  hipLaunchKernelGGL((disc_fwd_kernel<H, BM, 2>), dim3(tiles), dim3(BM * 8), smem_f, stream, ga);
  IA_CHECK_LAUNCH();
  return IA_OK;
}


template <int H, int BM>
int launch_fused_tiles(const FusedArgs& fa, int R, hipStream_t stream) {
  constexpr size_t smem_f = sizeof(float) * (BM * (H + 1) + NS * FB_K * H + 4 * BM + BM + 2 * (BM / 32) * H + BM * XP);
  constexpr size_t smem_b = sizeof(float) * (BM * XP3 + BM * (H + 1) + NS * FB_K * H + NS * BM * A_LD);
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(disc_fwd_kernel<H, BM>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_f);
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(disc_bwd_kernel<H, BM>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_b);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  constexpr size_t smem_fb = sizeof(float) * (BM * (H + 1) + 3 * FB_K * H + 4 * BM + BM + 2 * (BM / 32) * H + BM * XP3);
  static bool attr_fb = false;
  if (!attr_fb) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(disc_fb_kernel<H, BM>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_fb);
    if (e != hipSuccess) return (int)e;
    attr_fb = true;
  }
  static_assert(smem_fb == sizeof(float) * fb_lds_floats<H, BM, 24>(), "disc_fb_kernel's LDS map");
  const int tiles = cdivi(R, BM);
  if (!g_fused_split) {
    hipLaunchKernelGGL((disc_fb_kernel<H, BM>), dim3(tiles), dim3(BM * 8), smem_fb, stream, fa);
    IA_CHECK_LAUNCH();
    return IA_OK;
  }
  hipLaunchKernelGGL((disc_fwd_kernel<H, BM>), dim3(tiles), dim3(BM * 8), smem_f, stream, fa);
  IA_CHECK_LAUNCH();
  hipLaunchKernelGGL((disc_bwd_kernel<H, BM>), dim3(tiles), dim3(BM * 8), smem_b, stream, fa);
  IA_CHECK_LAUNCH();
  return IA_OK;
}

// the update's tile pass and the penalty's in one launch (`disc_fb_gp_kernel`): 256-wide stack, narrow rows, 64-row update
// tiles, the penalty's eight-column-wave form. `false`: this configuration is not it (the caller launches them apart).
bool g_fb_gp_merged = true;   // (ia_disc_fused_gp_merged: same-box A/Bs)
inline bool fb_gp_merged_ok(int H, bool wide, int bm) {
  return g_fb_gp_merged && H == 256 && bm == 64 && !g_fused_split && g_gp_groups == 8;
}
template <int DW>
int launch_fb_gp_merged(const FusedArgs& fa, const FusedArgs& ga, int R, int B, hipStream_t stream) {
  constexpr int H = 256, BMG = 32;
  constexpr bool WIDE = DW == 64;
  constexpr int XPW = xp_of(DW), XP3W = xp3_of(DW);
  constexpr int SCR8 = WIDE ? (FB_K * H >= 2 * 32 * 64 ? 0 : 2 * 32 * 64) : ((H * 25 > 4 * 32 * 33) ? H * 25 : 4 * 32 * 33);
  constexpr size_t smem_g8 = sizeof(float) * (BMG * (H + 1) + 3 * FB_K * H + H * XPW + SCR8 + H + BMG * XP3W);
  constexpr size_t smem_fb = sizeof(float) * fb_lds_floats<H, 64, DW>();
  constexpr size_t smem = smem_g8 > smem_fb ? smem_g8 : smem_fb;
  static_assert(smem <= 160 * 1024, "one workgroup per CU");
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(disc_fb_gp_kernel<H, DW>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return (int)e;
    attr = true;
  }
  const int n_gp = cdivi(B, BMG), n_fb = cdivi(R, 64);
  hipLaunchKernelGGL((disc_fb_gp_kernel<H, DW>), dim3(n_gp + n_fb), dim3(512), smem, stream, fa, ga, n_gp);
  IA_CHECK_LAUNCH();
  return IA_OK;
}

// rows of 25 .. 64 floats: the one-launch tile pass only (64-row tiles)
template <int H>
int launch_fused_tiles_wide(const FusedArgs& fa, int R, hipStream_t stream) {
  constexpr size_t smem = sizeof(float) * fb_lds_floats<H, 64, 64>();
  static_assert(smem <= 160 * 1024, "one workgroup per CU");
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(disc_fb_kernel<H, 64, 64>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return (int)e;
    attr = true;
  }
  hipLaunchKernelGGL((disc_fb_kernel<H, 64, 64>), dim3(cdivi(R, 64)), dim3(512), smem, stream, fa);
  IA_CHECK_LAUNCH();
  return IA_OK;
}

}  // namespace

namespace { long long* g_fused_dbg = nullptr; }
// the reference's default 32 x 32 stack takes the register-resident row kernel of airl_fused.hip behind the same entries
bool ia_disc32_shape_ok(const ia_mlp_desc* d, int ldx);
int64_t ia_disc32_ws_floats(const ia_mlp_desc* d, int R);
int ia_disc32_assemble(const ia_disc_step_args* a, int n_updates, int64_t idx_stride, int64_t x_stride, int64_t rn_stride,
                       float* rn_ws, hipStream_t stream);
int ia_disc32_step(const ia_disc_step_args* a, void* stream);
int ia_disc32_predict(const ia_mlp_desc* d, const float* params, const float* X, int ldx, int R, const float* mean,
                      const float* var, float eps, int out_act, float* out, hipStream_t stream);
extern "C" int ia_disc_fused_tile_rows(int rows) { g_fused_bm = rows == 32 ? 32 : 64; return IA_OK; }
extern "C" int ia_disc_fused_split_tiles(int on) { g_fused_split = on != 0; return IA_OK; }
extern "C" int ia_disc_fused_side_reduce(int on) { g_side_reduce = on != 0; return IA_OK; }
extern "C" int ia_disc_fused_gp_groups(int groups) {
  g_gp_groups = (groups == 1 || groups == 2) ? groups : 8;
  g_fb_gp_merged = groups != 80;   // 80: eight column waves, but the penalty's pass as a launch of its own (A/Bs, tests)
  return IA_OK;
}
extern "C" int ia_disc_fused_debug_timing(void* device_buffer_16xi64) {
  g_fused_dbg = static_cast<long long*>(device_buffer_16xi64);
  return IA_OK;
}

extern "C" int64_t ia_disc_fused_gp_ws_floats(const ia_mlp_desc* d, int B, int ldx) {
  if (B <= 0 || fused_dw(d, ldx) == 0) return 0;   // (the penalty's tile kernel: rows of up to 24 / up to 64 floats)
  return gp_ws_layout(d, B, ldx, nullptr).total;
}

extern "C" int64_t ia_disc_fused_ws_floats(const ia_mlp_desc* d, int R, int ldx) {
  if (R > 0 && ia_disc32_shape_ok(d, ldx)) return ia_disc32_ws_floats(d, R);
  if (R <= 0 || !fused_shape_ok(d, ldx)) return 0;
  return fused_ws_layout(d, R, nullptr).total;
}

namespace {
void fill_assemble_sources(AssembleArgs& as, const ia_disc_step_args* a) {
  as.obs[0] = a->obs0; as.act_f32[0] = a->act0_f32; as.act_i64[0] = a->act0_i64; as.next[0] = a->next0;
  as.done[0] = a->done0; as.idx[0] = a->idx0; as.n[0] = a->n0;
  as.obs[1] = a->obs1; as.act_f32[1] = a->act1_f32; as.act_i64[1] = a->act1_i64; as.next[1] = a->next1;
  as.done[1] = a->done1; as.idx[1] = a->idx1; as.n[1] = a->n1;
  if (a->n0 == 0) { as.obs[0] = a->obs1; as.next[0] = a->next1; as.done[0] = a->done1; }
  if (a->n1 == 0) { as.obs[1] = a->obs0; as.next[1] = a->next0; as.done[1] = a->done0; }
  as.obs_dim = a->obs_dim; as.act_dim = a->act_dim; as.use_state = a->use_state; as.use_action = a->use_action;
  as.use_next = a->use_next_state; as.use_done = a->use_done;
}
}  // namespace

// A whole round's batch assembly in ONE launch: update k (blockIdx.y) gathers rows idx0 + k*idx_stride /
// idx1 + k*idx_stride of the two tables into X + k*x_stride and leaves its RunningNorm slab moments at
// rn_ws + k*rn_stride (no merge: ia_running_norm_merge_seq applies them in order and keeps per-update snapshots).
extern "C" int ia_disc_assemble_round(const ia_disc_step_args* a, int n_updates, int64_t idx_stride, int64_t x_stride,
                                      int64_t rn_stride, void* stream) {
  if (a && n_updates > 0 && (ia_disc32_shape_ok(a->desc, a->ldx) || fused_dw(a->desc, a->ldx) == 64))
    return ia_disc32_assemble(a, n_updates, idx_stride, x_stride, rn_stride, a->rn_ws, (hipStream_t)stream);
  if (!a || n_updates <= 0 || !fused_shape_ok(a->desc, a->ldx) || a->n0 + a->n1 <= 0) return IA_ERR_ARG;
  AssembleArgs as{};
  fill_assemble_sources(as, a);
  const int R = a->n0 + a->n1;
  as.X = a->X; as.ldx = a->ldx; as.R = R; as.D = a->desc->dims[0];
  as.update_norm = a->rn_ws != nullptr;
  as.rn_ws = a->rn_ws; as.mean = nullptr;
  as.idx_stride = idx_stride; as.x_stride = x_stride; as.rn_stride = rn_stride;
  hipLaunchKernelGGL(disc_assemble_kernel, dim3(cdivi(R, RN_ROWS_PER_BLOCK), n_updates), dim3(AS_NT), 0,
                     (hipStream_t)stream, as);
  IA_CHECK_LAUNCH();
  return IA_OK;
}

// W2T and the padded W1 image of the CURRENT parameters into the fused workspace (once before a batch of
// pre-assembled updates; the updates' own Adam steps keep them current).
extern "C" int ia_disc_fused_prepare(const ia_mlp_desc* d, const float* params, int R, int ldx, float* fused_ws,
                                     void* stream) {
  if (fused_ws && R > 0 && ia_disc32_shape_ok(d, ldx)) return IA_OK;   // (its row kernel reads the parameters themselves)
  if (!fused_shape_ok(d, ldx) || !fused_ws || R <= 0) return IA_ERR_ARG;
  const int D = d->dims[0], H = d->dims[1];
  const FusedWs w = fused_ws_layout(d, R, fused_ws);
  AssembleArgs as{};
  as.R = 0; as.D = D; as.ldx = ldx; as.H = H;
  as.W2 = params + (long long)H * D + H; as.W2T = w.W2T;
  as.W1 = params; as.W1P = w.W1P; as.xp = xp_of(fused_dw(d, ldx));
  hipLaunchKernelGGL(disc_assemble_kernel, dim3((H / 64) * (H / 64) + 1), dim3(AS_NT), 0, (hipStream_t)stream, as);
  IA_CHECK_LAUNCH();
  return IA_OK;
}

// PREDICTION on the tile kernel: out[r] = act(MLP(normalise(X[r]))) for R assembled rows -- `RewardNet.predict_th` of a
// whole rollout tile (`rewards/reward_nets.py:176-204`; the relabelling behind the rollout's last step,
// `rewards/reward_wrapper.py:110-115`) as TWO launches (weight images of the current parameters, then the forward pass of
// 64-row tiles with the hidden activations chained through LDS: disc_fwd_kernel<H, 64, 3>) instead of the five of
// ia_running_norm_apply + ia_mlp_forward, and without the [R, H] activations going out to HBM twice. D <= 24 stacks
// D -> H -> H -> 1 (ReLU, H = 128 / 256), and the reference's default D <= 64 -> 32 -> 32 -> 1 stack on the register-resident row
// kernel of airl_fused.hip (ONE launch); 0 floats of workspace = shape not covered (use ia_mlp_forward).
extern "C" int64_t ia_disc_fused_predict_ws_floats(const ia_mlp_desc* d, int ldx) {
  if (ia_disc32_shape_ok(d, ldx)) return 4;   // (the 32 x 32 row kernel needs no workspace: a token size)
  if (fused_dw(d, ldx) != 24) return 0;
  return fused_ws_layout(d, 64, nullptr).total;
}
namespace {
template <int H>
int launch_predict_tiles(const FusedArgs& fa, int R, hipStream_t stream) {
  constexpr int BM = 64;
  constexpr size_t smem_f = sizeof(float) * (BM * (H + 1) + NS * FB_K * H + 4 * BM + BM + 2 * (BM / 32) * H + BM * XP);
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(disc_fwd_kernel<H, BM, 3>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_f);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  hipLaunchKernelGGL((disc_fwd_kernel<H, BM, 3>), dim3(cdivi(R, BM)), dim3(BM * 8), smem_f, stream, fa);
  IA_CHECK_LAUNCH();
  return IA_OK;
}
}  // namespace
extern "C" int ia_disc_fused_predict(const ia_mlp_desc* d, const float* params, const float* X, int ldx, int R,
                                     const float* norm_mean, const float* norm_var, float norm_eps, int out_act,
                                     float* predict_ws, float* out, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!d || !params || !X || !predict_ws || !out || R <= 0) return IA_ERR_ARG;
  if (ia_disc32_shape_ok(d, ldx))   // the reference's default 32 x 32 stack (<= 64 inputs): the register-resident row kernel
    return ia_disc32_predict(d, params, X, ldx, R, norm_mean, norm_var, norm_eps, out_act, out, stream);
  if (fused_dw(d, ldx) != 24) return IA_ERR_UNSUPPORTED;
  const int D = d->dims[0], H = d->dims[1];
  const FusedWs w = fused_ws_layout(d, 64, predict_ws);
  AssembleArgs as{};
  as.R = 0; as.D = D; as.ldx = ldx; as.H = H;
  as.W2 = params + (long long)H * D + H; as.W2T = w.W2T;
  as.W1 = params; as.W1P = w.W1P; as.xp = XP;
  hipLaunchKernelGGL(disc_assemble_kernel, dim3((H / 64) * (H / 64) + 1), dim3(AS_NT), 0, stream, as);
  IA_CHECK_LAUNCH();
  FusedArgs fa{};
  fa.X = X; fa.ldx = ldx; fa.R = R; fa.D = D;
  fa.mean = norm_mean; fa.var = norm_var; fa.eps = norm_eps;
  fa.params = params; fa.W2T = w.W2T; fa.W1P = w.W1P;
  fa.logits = out; fa.dump = w.dump; fa.out_act = out_act;
  return H == 256 ? launch_predict_tiles<256>(fa, R, stream) : launch_predict_tiles<128>(fa, R, stream);
}

// Data-parallel tail of a fused update (see disc_adam_refresh_kernel): adam->grads holds the all-reduced (summed)
// gradient; scaled by grad_scale in place, Adam step on `params`, W2T / W1 images of fused_ws refreshed.
extern "C" int ia_disc_fused_adam(const ia_mlp_desc* d, float* params, float grad_scale, int R, int ldx, float* fused_ws,
                                  const ia_adam_args* adam, void* stream) {
  const bool narrow = ia_disc32_shape_ok(d, ldx);   // no weight images to refresh: H = D = 0 below
  if ((!narrow && !fused_shape_ok(d, ldx)) || !fused_ws || !params || !adam || !adam->grads || !adam->exp_avg ||
      !adam->exp_avg_sq || R <= 0)
    return IA_ERR_ARG;
  const int D = d->dims[0], H = d->dims[1];
  FusedWs w{};
  if (!narrow) w = fused_ws_layout(d, R, fused_ws);
  AdamRefreshArgs ra{};
  ra.n = (long long)H * D + H + (long long)H * H + H + H + 1;
  ra.grads = adam->grads; ra.gscale = grad_scale;
  ra.p = params; ra.m = adam->exp_avg; ra.v = adam->exp_avg_sq;
  ra.beta1 = adam->beta1; ra.beta2 = adam->beta2; ra.eps = adam->eps; ra.wd = adam->weight_decay;
  ra.step_size = adam->step_size; ra.bc2_sqrt = adam->bc2_sqrt;
  ra.W2T = w.W2T; ra.W1P = w.W1P; ra.H = narrow ? 0 : H; ra.D = narrow ? 1 : D;
  ra.xp = narrow ? XP : xp_of(fused_dw(d, ldx));
  hipLaunchKernelGGL(disc_adam_refresh_kernel, dim3(cdivi(ra.n, 256)), dim3(256), 0, (hipStream_t)stream, ra);
  IA_CHECK_LAUNCH();
  return IA_OK;
}

// The fused form of ia_disc_step_basic (mlp.hip dispatches here when a->fused_ws is set and the shape
// qualifies). Same contract, same outputs (logits, dlogits, stats, rn_ws slab moments, grads / Adam).
int ia_disc_step_fused(const ia_disc_step_args* a, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  const ia_mlp_desc* d = a->desc;
  const int R = a->n0 + a->n1, D = d->dims[0], H = d->dims[1];
  if (a->fused_ws && ia_disc32_shape_ok(d, a->ldx)) return a->gp_e ? IA_ERR_UNSUPPORTED : ia_disc32_step(a, stream_);
  const int dw = fused_dw(d, a->ldx);
  if (dw == 0 || !a->fused_ws) return IA_ERR_UNSUPPORTED;
  const bool wide = dw != 24;
  const FusedWs w = fused_ws_layout(d, R, a->fused_ws);
  const int bm = (g_fused_bm == 64 || wide) ? 64 : 32;
  const int tiles = cdivi(R, bm), slabs = cdivi(R, RN_ROWS_PER_BLOCK);
  const long long nW1 = (long long)H * D, n1 = nW1 + H, n2 = (long long)H * H + H, n3 = H + 1, tot = n1 + n2 + n3;

  AssembleArgs as{};
  fill_assemble_sources(as, a);
  as.X = a->X; as.ldx = a->ldx; as.R = R; as.D = D;
  as.update_norm = (a->norm_mean != nullptr && a->update_norm) ? 1 : 0;
  as.rn_ws = a->rn_ws; as.mean = a->norm_mean; as.var = a->norm_var; as.count = a->norm_count;
  const bool use_p = as.update_norm && a->pnorm_mean && a->pnorm_dim > 0 && a->pnorm_dim <= D;
  as.pmean = use_p ? a->pnorm_mean : nullptr; as.pvar = a->pnorm_var; as.pcount = use_p ? a->pnorm_count : nullptr;
  as.pdim = use_p ? a->pnorm_dim : 0;
  as.ticket = w.ticket;
  as.W2 = a->params + n1; as.W2T = w.W2T; as.H = H;
  as.W1 = a->params; as.W1P = w.W1P; as.xp = xp_of(dw);
  int rc;
  if (!a->pre_assembled && !wide) {   // (pre-assembled: X / slab moments come from ia_disc_assemble_round, the statistics to
                             //  normalise with from the caller, W2T / W1P from ia_disc_fused_prepare + the Adam steps)
    hipLaunchKernelGGL(disc_assemble_kernel, dim3(slabs + (H / 64) * (H / 64) + 1), dim3(AS_NT), 0, stream, as);
    IA_CHECK_LAUNCH();
  } else if (!a->pre_assembled) {
    // wide rows: the pass kernel of airl_fused.hip assembles (any row length) and leaves the slab moments, the merges are
    // their own launches (ia_disc32_step's sequence), and the weight images come from the transposer blocks
    if (as.update_norm && !a->rn_ws) return IA_ERR_ARG;
    if ((rc = ia_disc32_assemble(a, 1, 0, 0, 0, as.update_norm ? a->rn_ws : nullptr, stream))) return rc;
    if (as.update_norm) {
      if ((rc = ia_running_norm_merge(a->rn_ws, 1, R, D, D, a->norm_mean, a->norm_var, a->norm_count, stream_))) return rc;
      if (use_p && (rc = ia_running_norm_merge(a->rn_ws, 1, R, a->pnorm_dim, D, a->pnorm_mean, a->pnorm_var,
                                               a->pnorm_count, stream_)))
        return rc;
    }
    if ((rc = ia_disc_fused_prepare(d, a->params, R, a->ldx, a->fused_ws, stream_))) return rc;
  }

  FusedArgs fa{};
  fa.X = a->X; fa.ldx = a->ldx; fa.R = R; fa.D = D;
  fa.mean = a->norm_mean; fa.var = a->norm_var; fa.eps = a->norm_eps;
  fa.params = a->params; fa.W2T = w.W2T; fa.W1P = w.W1P;
  fa.h1 = a->hidden; fa.dh2 = a->hidden + (long long)R * H;
  fa.h1mask = w.h1mask;
  fa.logits = a->logits; fa.dlogits = a->dlogits; fa.n_expert = a->n_expert; fa.loss_scale = a->loss_scale;
  fa.part = w.part; fa.P1 = w.P1; fa.P3 = w.P3; fa.dump = w.dump;
  fa.dbg = g_fused_dbg;
  // (with the penalty, where both passes are one-tile 512-thread workgroups: ONE launch for the two of them, below)
  const bool merged = a->gp_e != nullptr && fb_gp_merged_ok(H, wide, bm);
  if (merged) rc = IA_OK;
  else if (wide) rc = H == 256 ? launch_fused_tiles_wide<256>(fa, R, stream) : launch_fused_tiles_wide<128>(fa, R, stream);
  else if (bm == 64) rc = H == 256 ? launch_fused_tiles<256, 64>(fa, R, stream) : launch_fused_tiles<128, 64>(fa, R, stream);
  else rc = H == 256 ? launch_fused_tiles<256, 32>(fa, R, stream) : launch_fused_tiles<128, 32>(fa, R, stream);
  if (rc) return rc;

  // dW2 [H,H] = dh2^T . h1 (+ db2 = column sums of dh2), split-K slabs inside `partials` ([splits][tot])
  // (Tried and removed: a dedicated kernel whose waves take their MFMA fragments straight from global memory --
  //  no LDS, no barriers, 16 splits. 54 us against 26.6 us: a dword-per-lane global load is ~50 cycles of the
  //  vector-memory pipe per instruction, and fragments need one per operand and k-step; the LDS-staged GEMM
  //  moves the same data in 16-byte loads.)
  const int splits = a->splits;
  IaGemm g_main{};
  {
    const int kps = (((R + splits - 1) / splits) + 31) / 32 * 32;
    IaGemm g{};
    g.A = fa.dh2; g.lda = H;
    g.B = fa.h1; g.ldb = H;
    g.M = H; g.N = H; g.K = R;
    g.C = a->partials + n1; g.ldc = H;
    g.splits = splits; g.k_per_split = kps; g.c_split_stride = tot;
    g.dbias = nullptr; g.dbias_split_stride = tot;   // (db2: column sums of dh2 in the tile pass -- 3.5 us less here)
    g_main = g;
  }
  const bool gp = a->gp_e != nullptr;
  // What of the closing reduction does not depend on a split-K product rides in the LAST product's launch (the penalty's
  // tile passes read the parameters: not in the first one then).
  const bool side = g_side_reduce && H * (D + 1) % 64 == 0;
  if ((gp || !side) && !merged) {
    if ((rc = ia_launch_gemm(IA_GEMM_TN, g_main, stream))) return rc;
  }

  // opt-in gradient penalty (grad_penalty.py's definition) on the interpolates of the batch's expert / generator rows:
  // three tile launches + the split-K product u2^T v1; its slabs join the reduction below
  GpWs gw{};
  int gtiles = 0;
  IaGemm g_pen{};
  if (gp) {
    const int B = a->n0;
    if (!a->gp_ws || !a->gp_out || a->n1 != B || a->n_expert != B) return IA_ERR_ARG;
    gw = gp_ws_layout(d, B, a->ldx, a->gp_ws);
    gtiles = cdivi(B, 32);
    FusedArgs ga = fa;
    ga.R = B; ga.h1 = gw.v1; ga.dh2 = gw.u2; ga.h1mask = gw.m1; ga.h2mask = gw.m2; ga.P1 = gw.P1; ga.P3 = gw.P3;
    ga.gp_e = a->gp_e; ga.gp_C = gw.C; ga.gp_pen = gw.pen; ga.gp_coef = a->gp_coef; ga.gp_target = a->gp_target;
    ga.dbg = nullptr;
    if (merged) {   // both tile passes, then the update's split-K product (which the first branch above left for here)
      if ((rc = wide ? launch_fb_gp_merged<64>(fa, ga, R, B, stream) : launch_fb_gp_merged<24>(fa, ga, R, B, stream))) return rc;
      rc = ia_launch_gemm(IA_GEMM_TN, g_main, stream);
    } else if (wide) {
      rc = H == 256 ? launch_gp_tiles_wide<256>(ga, B, stream) : launch_gp_tiles_wide<128>(ga, B, stream);
    } else {
      rc = H == 256 ? launch_gp_tiles<256>(ga, B, stream) : launch_gp_tiles<128>(ga, B, stream);
    }
    if (rc) return rc;
    const int kps = (((B + gw.splits - 1) / gw.splits) + 31) / 32 * 32;
    IaGemm g{};
    g.A = gw.u2; g.lda = H;
    g.B = gw.v1; g.ldb = H;
    g.M = H; g.N = H; g.K = B;
    g.C = gw.partials + n1; g.ldc = H;
    g.splits = gw.splits; g.k_per_split = kps; g.c_split_stride = tot;
    g.dbias = nullptr; g.dbias_split_stride = tot;
    g_pen = g;
    if (!side && (rc = ia_launch_gemm(IA_GEMM_TN, g, stream))) return rc;
  }

  ReduceArgs ra{};
  const long long p3s = 2 * (long long)H + 1;   // tile slab of the last layer: [dW3 | db3 | db2]
  if (gp) {   // (no bias gradients from the penalty: segment 2 has no second set)
    ra.src2[0] = gw.P1; ra.stride2[0] = n1; ra.cnt2[0] = gtiles;
    ra.src2[1] = gw.partials + n1; ra.stride2[1] = tot; ra.cnt2[1] = gw.splits;
    ra.src2[3] = gw.P3; ra.stride2[3] = p3s; ra.cnt2[3] = gtiles;
    ra.pen = gw.pen; ra.pen_tiles = gtiles; ra.gp_rows = a->n0; ra.gp_out = a->gp_out;
  }
  ra.src[0] = w.P1; ra.stride[0] = n1; ra.cnt[0] = tiles; ra.seg_end[0] = n1;
  ra.src[1] = a->partials + n1; ra.stride[1] = tot; ra.cnt[1] = splits; ra.seg_end[1] = n1 + (long long)H * H;
  ra.src[2] = w.P3 + H + 1; ra.stride[2] = p3s; ra.cnt[2] = tiles; ra.seg_end[2] = n1 + n2;
  ra.src[3] = w.P3; ra.stride[3] = p3s; ra.cnt[3] = tiles; ra.seg_end[3] = tot;
  ra.n = tot; ra.accumulate = a->accumulate; ra.grads = a->grads;
  ra.adam = (a->adam && !a->accumulate) ? 1 : 0;
  ra.p = a->params; ra.m = a->exp_avg; ra.v = a->exp_avg_sq;
  ra.beta1 = a->beta1; ra.beta2 = a->beta2; ra.eps = a->adam_eps; ra.wd = a->weight_decay;
  ra.step_size = a->step_size; ra.bc2_sqrt = a->bc2_sqrt;
  ra.part = w.part; ra.tiles = tiles; ra.R = R; ra.n_expert = a->n_expert; ra.loss_scale = a->loss_scale;
  ra.stats = a->stats;
  ra.W2T = w.W2T; ra.W1P = w.W1P; ra.H = H; ra.D = D; ra.xp = xp_of(dw);
  if (side) {
    // first / last layer, b2, the statistics row: beside the product; dW2's split slabs: behind it
    ra.r_begin[0] = 0; ra.r_end[0] = n1; ra.nb0 = (int)(n1 / 64);
    ra.r_begin[1] = n1 + (long long)H * H; ra.r_end[1] = tot; ra.do_stats = 1;
    if ((rc = ia_launch_gemm_tn_side(gp ? g_pen : g_main, ra, stream))) return rc;
    ra.r_begin[0] = n1; ra.r_end[0] = n1 + (long long)H * H; ra.nb0 = H * H / 64;
    ra.r_begin[1] = 0; ra.r_end[1] = 0; ra.do_stats = 0;
  } else {
    ra.r_begin[0] = 0; ra.r_end[0] = tot; ra.nb0 = cdivi(tot, 64);
    ra.r_begin[1] = 0; ra.r_end[1] = 0; ra.do_stats = 1;
  }
  hipLaunchKernelGGL(disc_reduce_kernel, dim3(disc_reduce_blocks(ra)), dim3(256), 0, stream, ra);
  IA_CHECK_LAUNCH();
  if (a->adam && a->accumulate)
    return ia_adam_step(a->params, a->grads, a->exp_avg, a->exp_avg_sq, tot, a->beta1, a->beta2, a->adam_eps,
                        a->weight_decay, a->step_size, a->bc2_sqrt, stream_);
  return IA_OK;
}
