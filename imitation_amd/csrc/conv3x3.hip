// The reward CNN's 3 x 3 "same" convolutions at FULL image resolution (`rewards/reward_nets.py:460-610` -> `util/networks.py:286-357`:
// Conv2d(3, stride 1, padding "same") - ReLU per hidden channel count, 32 channels by default, AdaptiveAvgPool2d(1)): kernels
// written for THAT geometry instead of the general im2col GEMMs of gemm.hip / conv.hip, which the image-GAIL round ran at 0.19 of
// the fp32-MFMA peak (`profiles/r05_image_gail.md`).
//
//   conv3x3_c32_wgrad_kernel   dW[co][ky][kx][ci] = sum over (b, oy, ox) of dz[b, oy, ox, co] * x[b, oy+ky-1, ox+kx-1, ci], Cin = Cout = 32.
//       A 7.2 M-row x 32 x 288 TN product at 1 024 frames of 84 x 84. The split-K GEMM cut it into 1 024 row slabs x 5 tiles of
//       64 x 64 and re-read the im2col view of x nine times over (3.6 ms, 36 TFLOP/s). Here a workgroup owns whole IMAGES: it walks the
//       output rows of an image with the three input rows an output row needs resident in LDS (every input row is loaded once and
//       used by three output rows), the WHOLE 32 x 288 gradient in the accumulators of its waves (wave (ky, half): taps (ky, 0..2),
//       three 32 x 32 tiles, over one half of a row's positions; the halves are added in a fixed order when the workgroup is
//       through with its images), positions along the MFMA's K index: A[m = co][k = position] and B[k = position][n = ci] are both
//       conflict-free ds_read_b32 of channel-last rows. HBM traffic: x and dz once each.
//   avgpool_relu_backward_kernel   dz[b, p, c] = y[b, p, c] > 0 ? dout[b, c] / HW : 0 -- the backward of "ReLU then global average
//       pool" in one pass (read y, write dz) instead of the pool's broadcast (write) + relu_backward (read, read, write).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int C3_C = 32;          // channels in and out
constexpr int C3_WMAX = 128;      // widest image row the LDS rows are sized for
constexpr int C3_THREADS = 384;   // six waves: (kernel row ky = wave % 3) x (half of the positions of a row = wave / 3)

// LDS: a ring of FOUR input rows (three in use + the one being loaded), each [W + 2 pixels][32] with a zero pixel at either
// end (the "same" padding in x), and TWO dz rows [W (+1 when odd)][32].
struct C3Geo {
  int W, H, Wp;            // Wp = W rounded up to even: positions go through the MFMA in pairs
  int xrow, zrow;          // floats per LDS row
  __host__ __device__ C3Geo(int H_, int W_) : W(W_), H(H_), Wp((W_ + 1) & ~1), xrow((((W_ + 1) & ~1) + 2) * C3_C), zrow(((W_ + 1) & ~1) * C3_C) {}
  __host__ __device__ int total() const { return 4 * xrow + 2 * zrow; }
};

__global__ __launch_bounds__(C3_THREADS) void conv3x3_c32_wgrad_kernel(const float* __restrict__ dz, const float* __restrict__ x,
                                                                       int B, int H, int W, float* __restrict__ part,
                                                                       float* __restrict__ dbp) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const C3Geo g(H, W);
  float* xr = lds;                    // [4][xrow]
  float* zr = lds + 4 * g.xrow;       // [2][zrow]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ky = wv % 3, half = wv / 3;
  const int m = lane & 31, kk = lane >> 5;
  const int row_f4 = W * C3_C / 4;    // float4 pieces of one image row
  constexpr int NLD = (C3_WMAX * C3_C / 4 + C3_THREADS - 1) / C3_THREADS;   // pieces per thread and row

  f32x16 acc[3];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[t][j] = 0.f;
  float bsum = 0.f;

  // zero pixels of every x row (never overwritten) and the odd-width tail of the dz rows
  for (int e = tid; e < 4 * 2 * C3_C; e += C3_THREADS) {
    const int r = e / (2 * C3_C), q = e - r * 2 * C3_C;
    xr[r * g.xrow + (q < C3_C ? q : (g.Wp + 1) * C3_C + (q - C3_C))] = 0.f;
  }
  if (g.Wp != W) {
    for (int e = tid; e < 4 * C3_C; e += C3_THREADS) xr[(e / C3_C) * g.xrow + (W + 1) * C3_C + (e % C3_C)] = 0.f;
    for (int e = tid; e < 2 * C3_C; e += C3_THREADS) zr[(e / C3_C) * g.zrow + W * C3_C + (e % C3_C)] = 0.f;
  }

  auto load_row = [&](const float* __restrict__ src, f32x4 (&v)[NLD]) {   // one image row, global -> registers (all in flight)
    const f32x4* s4 = reinterpret_cast<const f32x4*>(src);
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int e = tid + i * C3_THREADS;
      v[i] = s4[min(e, row_f4 - 1)];
    }
  };
  auto store_row = [&](float* __restrict__ dst, const f32x4 (&v)[NLD]) {
    f32x4* d4 = reinterpret_cast<f32x4*>(dst);
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int e = tid + i * C3_THREADS;
      if (e < row_f4) d4[e] = v[i];
    }
  };
  auto zero_row = [&](float* __restrict__ dst) {
    f32x4* d4 = reinterpret_cast<f32x4*>(dst);
    for (int e = tid; e < row_f4; e += C3_THREADS) d4[e] = f32x4{0.f, 0.f, 0.f, 0.f};
  };
  const int steps = g.Wp >> 1;                                   // position pairs of a row
  const int s_lo = half ? (steps + 1) >> 1 : 0, s_hi = half ? steps : (steps + 1) >> 1;

  for (int b = blockIdx.x; b < B; b += gridDim.x) {
    const float* xb = x + (long long)b * H * W * C3_C;
    const float* zb = dz + (long long)b * H * W * C3_C;
    // Ring slot of input row r: (r + 1) & 3 (row -1 -- zeros -- in slot 0). Output row oy reads rows oy - 1 .. oy + 1 =
    // slots oy & 3 .. (oy + 2) & 3; row oy + 2 goes to the FREE slot (oy + 3) & 3 and dz row oy + 1 to the other dz slot while
    // row oy is being contracted (both were last read for output row oy - 1): one barrier per output row.
    __syncthreads();   // (the previous image's last reads)
    zero_row(xr + 0 * g.xrow + C3_C);
    {
      f32x4 v0[NLD], v1[NLD], vz[NLD];
      load_row(xb, v0);
      if (H > 1) load_row(xb + (long long)W * C3_C, v1);
      load_row(zb, vz);
      store_row(xr + 1 * g.xrow + C3_C, v0);
      if (H > 1) store_row(xr + 2 * g.xrow + C3_C, v1);
      else zero_row(xr + 2 * g.xrow + C3_C);
      store_row(zr, vz);
    }
    __syncthreads();
    for (int oy = 0; oy < H; ++oy) {
      f32x4 vx[NLD], vz[NLD];
      const bool have_x = oy + 2 < H, have_z = oy + 1 < H;
      if (have_x) load_row(xb + (long long)(oy + 2) * W * C3_C, vx);
      if (have_z) load_row(zb + (long long)(oy + 1) * W * C3_C, vz);
      const float* xrow = xr + ((oy + ky) & 3) * g.xrow;   // this wave's input row oy + ky - 1
      const float* zrow = zr + (oy & 1) * g.zrow;
      // positions in pairs along K: A[m = co][k] = dz[ox0 + k][co], B[k][n = ci] = x[ox0 + k + kx - 1][ci] (LDS pixel ox0 + k + kx)
      const float* ap = zrow + kk * C3_C + m;
      const float* bp = xrow + kk * C3_C + m;
      for (int s = s_lo; s < s_hi; ++s) {
        const float a = ap[s * 2 * C3_C];
        const float b0 = bp[s * 2 * C3_C], b1 = bp[s * 2 * C3_C + C3_C], b2 = bp[s * 2 * C3_C + 2 * C3_C];
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b2, acc[2], 0, 0, 0);
        if (ky == 1) bsum += a;   // (wave-uniform) bias gradient: sum of dz over the positions
      }
      if (have_x) store_row(xr + ((oy + 3) & 3) * g.xrow + C3_C, vx);
      else zero_row(xr + ((oy + 3) & 3) * g.xrow + C3_C);
      if (have_z) store_row(zr + ((oy + 1) & 1) * g.zrow, vz);
      __syncthreads();
    }
  }
  // the two halves of the positions: the upper half's accumulators through LDS (the rows are free), added by the lower half's
  // wave of the same ky -- a fixed order
  float* ex = lds;   // [3 ky][3 kx][16][64]
  if (half == 1) {
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
      for (int j = 0; j < 16; ++j) ex[((ky * 3 + kx) * 16 + j) * 64 + lane] = acc[kx][j];
    if (ky == 1) ex[9 * 16 * 64 + lane] = bsum;
  }
  __syncthreads();
  if (half == 1) return;
#pragma unroll
  for (int kx = 0; kx < 3; ++kx)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[kx][j] += ex[((ky * 3 + kx) * 16 + j) * 64 + lane];
  if (ky == 1) bsum += ex[9 * 16 * 64 + lane];
  // slab of this workgroup: part[wg][co][(ky * 3 + kx) * 32 + ci]; accumulator register j of lane (n = lane & 31, half = lane >> 5)
  // is row m = 8 (j / 4) + 4 half + j % 4, column n
  float* slab = part + (long long)blockIdx.x * C3_C * 9 * C3_C;
#pragma unroll
  for (int kx = 0; kx < 3; ++kx)
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int co = 8 * (j >> 2) + 4 * kk + (j & 3);
      slab[co * 9 * C3_C + (ky * 3 + kx) * C3_C + m] = acc[kx][j];
    }
  if (ky == 1) {
    bsum += __shfl_xor(bsum, 32, 64);
    if (lane < 32) dbp[(long long)blockIdx.x * C3_C + lane] = bsum;
  }
}

// ---- conv3x3_c32_conv_kernel: y = act(bias + W * x) [* (mask > 0)] for 32 -> 32 channels, the forward of the reward CNN's second
// convolution AND -- with the flipped / transposed weights Wd[ci][ky][kx][co] = W[co][2 - ky][2 - kx][ci] -- its input gradient (then
// `mask` = the layer below's ReLU output: its backward in the epilogue). The implicit GEMM of gemm.hip ran these 133 GFLOP calls at
// 66-71 TFLOP/s (128 x 32 tiles, K = 288 in nine 32-deep chunks, the operand gathered chunk by chunk).
// A workgroup takes a BAND of 8 output rows of one image: 8 x W positions (672 = 21 tiles of 32 at W = 84), their 10 input rows in LDS
// with a zero pixel at either end and a pixel stride of 33 floats (B[k = (tap, ci)][n = position]: lanes walk positions -> 33 n + ci:
// conflict-free), the weights as [k][co] (A[m = co][k]: lanes walk co: conflict-free); per tile 144 MFMAs 32x32x2 whose A / B
// fragments are one ds_read_b32 each; bias, activation and mask on the accumulators, 16-byte stores.
constexpr int CV_BAND = 8, CV_PS = 33;
constexpr int CV_THREADS = 512;

__global__ __launch_bounds__(CV_THREADS) void conv3x3_c32_conv_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                      const float* __restrict__ bias, const float* __restrict__ mask,
                                                                      int H, int W, int bands, int relu, float* __restrict__ y) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int px = W + 2;
  float* ws = lds;                           // [288][32]: ws[k * 32 + co]
  float* xs = lds + 9 * C3_C * C3_C;         // [(BAND + 2) rows][px][33]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 31, kk = lane >> 5;
  const int b = blockIdx.x / bands, band = blockIdx.x - b * bands;
  const int oy0 = band * CV_BAND, rows = min(CV_BAND, H - oy0);
  // weights: `w` arrives as [k = (tap, ci)][co] (the caller transposes the 36 KB once per call): 16-byte copies, all of a thread's
  // loads in flight before its stores (element by element the prologue -- one workgroup per CU, nothing to hide it behind -- was a
  // third of the kernel: 32 serial round trips to L2 per thread and 64-way conflicted LDS scatters)
  {
    constexpr int NWQ = (9 * C3_C * C3_C / 4 + CV_THREADS - 1) / CV_THREADS;   // 5
    f32x4 wv[NWQ];
#pragma unroll
    for (int i = 0; i < NWQ; ++i) wv[i] = reinterpret_cast<const f32x4*>(w)[min(tid + i * CV_THREADS, 9 * C3_C * C3_C / 4 - 1)];
#pragma unroll
    for (int i = 0; i < NWQ; ++i)
      if (tid + i * CV_THREADS < 9 * C3_C * C3_C / 4) reinterpret_cast<f32x4*>(ws)[tid + i * CV_THREADS] = wv[i];
  }
  // input rows oy0 - 1 .. oy0 + rows (zeros outside the image), pixels -1 .. W, 16-byte loads (seven in flight per thread and
  // pass), pixel stride 33 in LDS
  const float* xb = x + (long long)b * H * W * C3_C;
  const int n4 = (rows + 2) * px * (C3_C / 4);
  constexpr int NB = 7;
  for (int e0 = 0; e0 < n4; e0 += NB * CV_THREADS) {
    f32x4 v[NB];
    int dst[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int e = e0 + tid + i * CV_THREADS;
      const int q = e & 7, pp = e >> 3;
      const int r = pp / px, p = pp - r * px;
      const int iy = oy0 - 1 + r, ix = p - 1;
      const bool in = e < n4 && iy >= 0 && iy < H && ix >= 0 && ix < W;
      dst[i] = e < n4 ? (r * px + p) * CV_PS + 4 * q : -1;
      v[i] = *reinterpret_cast<const f32x4*>(xb + ((long long)min(max(iy, 0), H - 1) * W + min(max(ix, 0), W - 1)) * C3_C + 4 * q);
      if (!in) v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int i = 0; i < NB; ++i)
      if (dst[i] >= 0) {
        float* d = xs + dst[i];
        d[0] = v[i][0]; d[1] = v[i][1]; d[2] = v[i][2]; d[3] = v[i][3];
      }
  }
  __syncthreads();
  float bv[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) bv[j] = bias ? bias[8 * (j >> 2) + 4 * kk + (j & 3)] : 0.f;
  const int npos = rows * W;
  const long long out0 = ((long long)b * H + oy0) * W;
  for (int t0 = wave * 32; t0 < npos; t0 += (CV_THREADS / 64) * 32) {
    const int p = min(t0 + n, npos - 1);
    const int r = p / W, ox = p - r * W;
    const float* bbase = xs + (r * px + ox) * CV_PS + kk;   // element (tap, ci = 2 j' + kk)
    const float* abase = ws + kk * C3_C + n;
    f32x16 acc;   // (the bias joins in the epilogue, as in the GEMM this replaces: the same sums in the same order, bit for bit --
                  //  pre-activations within an ulp of zero keep their sign, and with it the ReLU mask of the backward pass)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = 0.f;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const float* bt = bbase + ((tap / 3) * px + (tap % 3)) * CV_PS;
      const float* at = abase + tap * C3_C * C3_C;
#pragma unroll
      for (int c2 = 0; c2 < C3_C / 2; ++c2)   // k-step: channels 2 c2 + kk of this tap
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(at[2 * c2 * C3_C], bt[2 * c2], acc, 0, 0, 0);
    }
    if (t0 + n < npos) {
      const long long o = (out0 + p) * C3_C + 4 * kk;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 v;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float pre = acc[4 * q + u] + bv[4 * q + u];
          v[u] = relu ? fmaxf(pre, 0.f) : pre;
        }
        if (mask != nullptr) {
          const f32x4 mq = *reinterpret_cast<const f32x4*>(mask + o + 8 * q);
#pragma unroll
          for (int u = 0; u < 4; ++u) v[u] = mq[u] > 0.f ? v[u] : 0.f;
        }
        *reinterpret_cast<f32x4*>(y + o + 8 * q) = v;
      }
    }
  }
}

// ---- the reward CNN's FIRST convolution: 4 input channels (the frame stack), 32 output channels, 3 x 3 "same" -----------------------
// K = 36 is no 32-deep chunk of a kernel row, so the general path materialised the im2col matrix (1 024 x 7 056 x 36 floats = 1 GB
// written, read by the forward GEMM and read again by the weight gradient: 2.8 ms of an update's 11). Both products straight from
// the 4-channel rows instead:
//   conv3x3_c4_fwd_kernel    y = relu(b + W x): a workgroup takes a band of output rows of one image, its input rows (+ halo, zero
//       pixels at the borders) in LDS; per 32 positions 18 MFMAs 32x32x2 with A[m = co][k = (tap, ci)] = the weights (registers) and
//       B[k][n = position] gathered from the LDS rows; bias and ReLU on the accumulators, 16-byte stores. Bound by the 925 MB of y.
//   conv3x3_c4_wgrad_kernel  dW[co][(tap, ci)] and db over the positions of whole images per workgroup: A[m = co][k = position] = dz
//       (LDS row), B[k = position][n = (tap, ci)] from the LDS rows of x, two 32-wide column tiles (36 columns). Reads dz once.
constexpr int C4_CI = 4, C4_K = 36;
constexpr int C4_BAND = 12;        // output rows per forward workgroup
constexpr int C4_THREADS = 256;

__global__ __launch_bounds__(C4_THREADS) void conv3x3_c4_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                    const float* __restrict__ bias, int H, int W, int bands,
                                                                    int relu, float* __restrict__ y) {
  extern __shared__ __attribute__((aligned(16))) float lds[];   // [(BAND + 2) rows][(W + 2) pixels][4]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 31, kk = lane >> 5;
  const int b = blockIdx.x / bands, band = blockIdx.x - b * bands;
  const int oy0 = band * C4_BAND, rows = min(C4_BAND, H - oy0);
  const int rs = (W + 2) * C4_CI;   // floats per LDS row
  // weights as the A operand: A[m = co][k] of step j is W[co][2 j + k] (torch layout [co][ky][kx][ci] = [co][36])
  float wa[C4_K / 2];
#pragma unroll
  for (int j = 0; j < C4_K / 2; ++j) wa[j] = w[n * C4_K + 2 * j + kk];
  // B operand: element e = 2 j + k = (tap, ci) of position p sits at LDS offset p_base + (ky * rs + kx * 4 + ci)
  int boff[C4_K / 2];
#pragma unroll
  for (int j = 0; j < C4_K / 2; ++j) {
    const int e = 2 * j + kk, tap = e >> 2, ci = e & 3;
    boff[j] = (tap / 3) * rs + (tap % 3) * C4_CI + ci;
  }
  float bv[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) bv[j] = bias[8 * (j >> 2) + 4 * kk + (j & 3)];
  // input rows oy0 - 1 .. oy0 + rows (zero outside the image), pixels -1 .. W
  const float* xb = x + (long long)b * H * W * C4_CI;
  const int px = W + 2;
  for (int e = tid; e < (rows + 2) * px; e += C4_THREADS) {
    const int r = e / px, p = e - r * px;
    const int iy = oy0 - 1 + r, ix = p - 1;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = *reinterpret_cast<const f32x4*>(xb + ((long long)iy * W + ix) * C4_CI);
    *reinterpret_cast<f32x4*>(lds + (r * px + p) * C4_CI) = v;
  }
  __syncthreads();
  const int npos = rows * W;
  float* yb = y + ((long long)b * H + oy0) * W * C3_C;
  for (int t0 = wave * 32; t0 < npos; t0 += 4 * 32) {
    const int p = min(t0 + n, npos - 1);
    const int r = p / W, ox = p - r * W;
    const float* base = lds + (r * px + ox) * C4_CI;
    float bq[C4_K / 2];
#pragma unroll
    for (int j = 0; j < C4_K / 2; ++j) bq[j] = base[boff[j]];
    f32x16 acc;   // (bias in the epilogue: the sums of the GEMM this replaces, bit for bit)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = 0.f;
#pragma unroll
    for (int j = 0; j < C4_K / 2; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[j], bq[j], acc, 0, 0, 0);
    if (t0 + n < npos) {
      float* o = yb + (long long)p * C3_C + 4 * kk;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 v;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float pre = acc[4 * q + u] + bv[4 * q + u];
          v[u] = relu ? fmaxf(pre, 0.f) : pre;
        }
        *reinterpret_cast<f32x4*>(o + 8 * q) = v;
      }
    }
  }
}

constexpr int C4W_THREADS = 256;   // four waves: quarters of a row's position pairs; each wave both column tiles
__global__ __launch_bounds__(C4W_THREADS) void conv3x3_c4_wgrad_kernel(const float* __restrict__ dz, const float* __restrict__ x,
                                                                       int B, int H, int W, float* __restrict__ part,
                                                                       float* __restrict__ dbp) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  // LDS: ring of four x rows [(Wp + 2) pixels][4] (zero pixels at either end) + two dz rows [Wp][32]
  const int Wp = (W + 1) & ~1;
  const int xrow = (Wp + 2) * C4_CI, zrow = Wp * C3_C;
  float* xr = lds;
  float* zr = lds + 4 * xrow;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 31, kk = lane >> 5;
  // B[k = position][n = e]: e = 32 t + m (t = 0, 1; e < 36) -> (ky, kx, ci); the wave's rows differ by ky PER LANE here
  int boff[2];
  bool bon[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int e = 32 * t + m;
    bon[t] = e < C4_K;
    const int ec = bon[t] ? e : 0, tap = ec >> 2, ci = ec & 3;
    boff[t] = (tap / 3) * 4 * xrow /* ring slot stride, resolved per output row below */ + (tap % 3) * C4_CI + ci;
  }
  f32x16 acc[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[t][j] = 0.f;
  float bsum = 0.f;
  for (int e = tid; e < 4 * xrow; e += C4W_THREADS) xr[e] = 0.f;
  for (int e = tid; e < 2 * zrow; e += C4W_THREADS) zr[e] = 0.f;
  const int zf4 = W * C3_C / 4, xf4 = W;   // float4 pieces of a dz row / an x row (one pixel each)
  constexpr int NZ = (C3_WMAX * C3_C / 4 + C4W_THREADS - 1) / C4W_THREADS;
  const int steps = Wp >> 1;
  const int s_lo = (steps * wv) >> 2, s_hi = (steps * (wv + 1)) >> 2;
  for (int b = blockIdx.x; b < B; b += gridDim.x) {
    const float* xb = x + (long long)b * H * W * C4_CI;
    const float* zb = dz + (long long)b * H * W * C3_C;
    __syncthreads();
    // ring slot of input row r: (r + 1) & 3; row -1 = zeros
    for (int e = tid; e < W; e += C4W_THREADS) *reinterpret_cast<f32x4*>(xr + 0 * xrow + (e + 1) * C4_CI) = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int e = tid; e < 2 * xf4; e += C4W_THREADS) {
      const int r = e / xf4, p = e - r * xf4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (r < H) v = *reinterpret_cast<const f32x4*>(xb + ((long long)r * W + p) * C4_CI);
      *reinterpret_cast<f32x4*>(xr + (r + 1) * xrow + (p + 1) * C4_CI) = v;
    }
    for (int e = tid; e < zf4; e += C4W_THREADS) reinterpret_cast<f32x4*>(zr)[e] = reinterpret_cast<const f32x4*>(zb)[e];
    __syncthreads();
    for (int oy = 0; oy < H; ++oy) {
      f32x4 vz[NZ], vx = {0.f, 0.f, 0.f, 0.f};
      const bool have_x = oy + 2 < H, have_z = oy + 1 < H;
      if (have_z) {
        const f32x4* s4 = reinterpret_cast<const f32x4*>(zb + (long long)(oy + 1) * W * C3_C);
#pragma unroll
        for (int i = 0; i < NZ; ++i) vz[i] = s4[min(tid + i * C4W_THREADS, zf4 - 1)];
      }
      if (have_x && tid < xf4) vx = *reinterpret_cast<const f32x4*>(xb + ((long long)(oy + 2) * W + tid) * C4_CI);
      // x row oy + ky - 1 is in slot (oy + ky) & 3: the lane's tap decides
      const float* zrow_p = zr + (oy & 1) * zrow + kk * C3_C + m;
      const float* bp[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int ky = boff[t] / (4 * xrow), rest = boff[t] - ky * 4 * xrow;
        bp[t] = xr + ((oy + ky) & 3) * xrow + rest + kk * C4_CI;
      }
      for (int s = s_lo; s < s_hi; ++s) {
        const float a = zrow_p[s * 2 * C3_C];
        const float b0 = bp[0][s * 2 * C4_CI];
        const float b1 = bon[1] ? bp[1][s * 2 * C4_CI] : 0.f;
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc[1], 0, 0, 0);
        bsum += a;
      }
      if (tid < xf4) *reinterpret_cast<f32x4*>(xr + ((oy + 3) & 3) * xrow + (tid + 1) * C4_CI) = vx;   // (zeros past the image)
      if (have_z) {
        f32x4* d4 = reinterpret_cast<f32x4*>(zr + ((oy + 1) & 1) * zrow);
#pragma unroll
        for (int i = 0; i < NZ; ++i)
          if (tid + i * C4W_THREADS < zf4) d4[tid + i * C4W_THREADS] = vz[i];
      }
      __syncthreads();
    }
  }
  // the four waves' partial sums through LDS, added in wave order by wave 0
  __syncthreads();
  float* ex = lds;   // [3 waves][2][16][64] + [3][64]
  if (wv > 0) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int j = 0; j < 16; ++j) ex[(((wv - 1) * 2 + t) * 16 + j) * 64 + lane] = acc[t][j];
    ex[3 * 2 * 16 * 64 + (wv - 1) * 64 + lane] = bsum;
  }
  __syncthreads();
  if (wv > 0) return;
  for (int w2 = 0; w2 < 3; ++w2) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[t][j] += ex[((w2 * 2 + t) * 16 + j) * 64 + lane];
    bsum += ex[3 * 2 * 16 * 64 + w2 * 64 + lane];
  }
  float* slab = part + (long long)blockIdx.x * C3_C * C4_K;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int e = 32 * t + m;
    if (e < C4_K)
#pragma unroll
      for (int j = 0; j < 16; ++j) slab[(8 * (j >> 2) + 4 * kk + (j & 3)) * C4_K + e] = acc[t][j];
  }
  bsum += __shfl_xor(bsum, 32, 64);
  if (lane < 32) dbp[(long long)blockIdx.x * C3_C + lane] = bsum;
}

__global__ __launch_bounds__(256) void avgpool_relu_backward_kernel(const float* __restrict__ dout, const float* __restrict__ y,
                                                                    int HW, int C, float* __restrict__ dz, long long total4) {
  const int Q = C >> 2;
  const float inv = (float)HW;
  const f32x4* y4 = reinterpret_cast<const f32x4*>(y);
  f32x4* z4 = reinterpret_cast<f32x4*>(dz);
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total4; e += (long long)gridDim.x * blockDim.x) {
    const int q = (int)(e % Q);
    const long long b = e / ((long long)HW * Q);
    const f32x4 v = y4[e];
    const f32x4 gq = *reinterpret_cast<const f32x4*>(dout + b * C + 4 * q);
    f32x4 o;
#pragma unroll
    for (int u = 0; u < 4; ++u) o[u] = v[u] > 0.f ? gq[u] / inv : 0.f;   // (the pool's own quotient, then the mask)
    z4[e] = o;
  }
}

}  // namespace

extern "C" {

/* Weight and bias gradient of a 3 x 3, stride-1, "same"-padded convolution with 32 input and 32 output channels on channel-last
 * tensors (dz[B, H, W, 32], x[B, H, W, 32]) as `ia_conv3x3_c32_wgrad_slabs(B)` slabs: part[slabs][32][3][3][32] (torch's
 * [Cout, KH, KW, Cin] weight layout per slab), dbp[slabs][32]; reduce with ia_reduce_partials. IA_ERR_UNSUPPORTED (-2) for
 * W > 128. */
int ia_conv3x3_c32_wgrad_slabs(int B) { return B < 1 ? 0 : (B < 1024 ? B : 1024); }

int ia_conv3x3_c32_wgrad(const float* dz, const float* x, int B, int H, int W, float* part, float* dbp, void* stream) {
  if (!dz || !x || !part || !dbp || B <= 0 || H <= 0 || W <= 0) return IA_ERR_ARG;
  if (W > C3_WMAX) return IA_ERR_UNSUPPORTED;
  const C3Geo g(H, W);
  constexpr int EX_FLOATS = 9 * 16 * 64 + 64;   // the position halves' exchange at the end (narrow images: more than the rows)
  const size_t bytes = (size_t)(g.total() > EX_FLOATS ? g.total() : EX_FLOATS) * sizeof(float);
  static size_t attr_bytes = 0;
  if (bytes > attr_bytes) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_c32_wgrad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)bytes) != hipSuccess)
      return IA_ERR_ARG;
    attr_bytes = bytes;
  }
  hipLaunchKernelGGL(conv3x3_c32_wgrad_kernel, dim3(ia_conv3x3_c32_wgrad_slabs(B)), dim3(C3_THREADS), bytes, (hipStream_t)stream,
                     dz, x, B, H, W, part, dbp);
  IA_CHECK_LAUNCH();
  return IA_OK;
}

/* y[B, H, W, 32] = act(bias + conv3x3(x[B, H, W, 32], w)) on channel-last tensors, stride 1, "same" padding. `wt`: the weights
 * TRANSPOSED, [ky][kx][ci][co] (= torch's [co][ky][kx][ci] with co moved last); `bias` nullable; `relu` != 0: ReLU; `mask` (nullable,
 * laid out like y): outputs zeroed where mask <= 0. With wt[ky][kx][co][ci] = W[co][2 - ky][2 - kx][ci] on dz this is the convolution's
 * input gradient (mask: the ReLU output below). -2 when a band of rows does not fit 160 KB of LDS (W > 94). */
int ia_conv3x3_c32_conv(const float* x, const float* w, const float* bias, const float* mask, int B, int H, int W, int relu, float* y,
                        void* stream) {
  if (!x || !w || !y || B <= 0 || H <= 0 || W <= 0) return IA_ERR_ARG;
  const size_t bytes = ((size_t)9 * C3_C * C3_C + (size_t)(CV_BAND + 2) * (W + 2) * CV_PS) * sizeof(float);
  if (bytes > 160 * 1024) return IA_ERR_UNSUPPORTED;
  static size_t attr_bytes = 0;
  if (bytes > attr_bytes) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_c32_conv_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)bytes) != hipSuccess)
      return IA_ERR_ARG;
    attr_bytes = bytes;
  }
  const int bands = (H + CV_BAND - 1) / CV_BAND;
  hipLaunchKernelGGL(conv3x3_c32_conv_kernel, dim3((unsigned)((long long)B * bands)), dim3(CV_THREADS), bytes, (hipStream_t)stream, x, w,
                     bias, mask, H, W, bands, relu, y);
  IA_CHECK_LAUNCH();
  return IA_OK;
}

/* The same for the reward CNN's FIRST convolution (4 input channels: the frame stack): forward y[B, H, W, 32] = act(b + W x) of
 * x[B, H, W, 4] with w[32][3][3][4] (`relu` != 0: ReLU), and its weight / bias gradient as ia_conv3x3_c32_wgrad_slabs(B) slabs
 * part[slabs][32][36], dbp[slabs][32] -- no [B H W, 36] column matrix. -2 for W > 128. */
int ia_conv3x3_c4_forward(const float* x, const float* w, const float* bias, int B, int H, int W, int relu, float* y, void* stream) {
  if (!x || !w || !bias || !y || B <= 0 || H <= 0 || W <= 0) return IA_ERR_ARG;
  if (W > C3_WMAX) return IA_ERR_UNSUPPORTED;
  const int bands = (H + C4_BAND - 1) / C4_BAND;
  const size_t bytes = (size_t)(C4_BAND + 2) * (W + 2) * C4_CI * sizeof(float);
  hipLaunchKernelGGL(conv3x3_c4_fwd_kernel, dim3((unsigned)((long long)B * bands)), dim3(C4_THREADS), bytes, (hipStream_t)stream, x, w,
                     bias, H, W, bands, relu, y);
  IA_CHECK_LAUNCH();
  return IA_OK;
}

int ia_conv3x3_c4_wgrad(const float* dz, const float* x, int B, int H, int W, float* part, float* dbp, void* stream) {
  if (!dz || !x || !part || !dbp || B <= 0 || H <= 0 || W <= 0) return IA_ERR_ARG;
  if (W > C3_WMAX) return IA_ERR_UNSUPPORTED;
  const int Wp = (W + 1) & ~1;
  const int rows_f = 4 * (Wp + 2) * C4_CI + 2 * Wp * C3_C, ex_f = 3 * 2 * 16 * 64 + 3 * 64;
  const size_t bytes = (size_t)(rows_f > ex_f ? rows_f : ex_f) * sizeof(float);
  hipLaunchKernelGGL(conv3x3_c4_wgrad_kernel, dim3(ia_conv3x3_c32_wgrad_slabs(B)), dim3(C4W_THREADS), bytes, (hipStream_t)stream, dz,
                     x, B, H, W, part, dbp);
  IA_CHECK_LAUNCH();
  return IA_OK;
}

/* Backward of "ReLU, then AdaptiveAvgPool2d(1)" on channel-last y[B, HW, C] (C % 4 == 0): dz = y > 0 ? dout[b, c] / HW : 0. */
int ia_avgpool_relu_backward(const float* dout, const float* y, int B, int HW, int C, float* dz, void* stream) {
  if (!dout || !y || !dz || B <= 0 || HW <= 0 || C <= 0) return IA_ERR_ARG;
  if (C % 4) return IA_ERR_UNSUPPORTED;
  const long long total4 = (long long)B * HW * (C / 4);
  const long long blocks = (total4 + 255) / 256;
  hipLaunchKernelGGL(avgpool_relu_backward_kernel, dim3((unsigned)(blocks < 65536 * 4 ? blocks : 65536 * 4)), dim3(256), 0,
                     (hipStream_t)stream, dout, y, HW, C, dz, total4);
  IA_CHECK_LAUNCH();
  return IA_OK;
}

}  // extern "C"
