// First layer of SB3's NatureCNN (Conv2d(4, 32, 8, stride 4) on uint8 [4, H, W] frame stacks, `x / 255` folded in) as an
// IMPLICIT GEMM: the explicit path (conv.hip) writes a [B*OH*OW, 256] fp32 column buffer -- 1.68 GB at batch 4096 of
// 84x84 frames -- and streams it twice more (forward GEMM, weight-gradient GEMM); here the A operand of
// v_mfma_f32_32x32x2_f32 is formed straight from the frames: a workgroup keeps whole images in LDS as BYTES (28 KB per
// 84x84x4 image), a lane reads the dword that holds four consecutive kernel columns of its output pixel and converts
// bytes to fp32 in registers (v_cvt_f32_ubyteN). HBM traffic per batch: 115 MB of frames + the outputs, instead of
// 3 x 1.68 GB.
//   forward : out[b, oh, ow, co] = relu(bias[co] + sum_{c,i,j} x[b, c, 4 oh + i, 4 ow + j] / 255 * W[co, c, i, j])
//             M-tile = 32 output pixels, N = 32 output channels, K = 256; all 128 weight fragments of a lane stay in
//             registers for the whole (persistent) workgroup; one wave = one 32x32 tile at a time
//   wgrad   : dW[co, k] = sum_m dout[m, co] * col[m, k], reduction over the pixels of the workgroup's images with the
//             accumulators in registers (wave w owns input channel w: two 32-wide k tiles), per-workgroup partials
//             written once and summed in workgroup order by ia_reduce_partials (deterministic); db = column sums.
// The first layer needs no input gradient.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "common.h"

namespace {

constexpr int C1 = 4, KW1 = 8, ST1 = 4, CO1 = 32, KTOT1 = C1 * KW1 * KW1, KSTEPS1 = KTOT1 / 2;
constexpr int FWD_NIMG = 2;   // images per workgroup iteration: 26 tiles over 4 waves (7, 7, 6, 6) instead of 13 (4, 3, 3, 3)

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// wfrag[kk][lane] = W[co = lane & 31][k = 2 kk + (lane >> 5)]: the B operand of k-step kk as one coalesced load
__global__ __launch_bounds__(256) void conv1_wfrag_kernel(const float* __restrict__ W, float* __restrict__ wfrag) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= KSTEPS1 * 64) return;
  const int kk = e >> 6, l = e & 63;
  wfrag[e] = W[(l & 31) * KTOT1 + 2 * kk + (l >> 5)];
}

__global__ __launch_bounds__(256, 2) void conv1_fwd_kernel(const uint8_t* __restrict__ x, const float* __restrict__ wfrag,
                                                           const float* __restrict__ bias, float scale, int H, int W, int OH,
                                                           int OW, int B, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, li = lane & 31;
  const int HW = H * W, img_bytes = C1 * HW, npix = OH * OW, ntile = (npix + 31) >> 5;
  float bw[KSTEPS1];
#pragma unroll
  for (int kk = 0; kk < KSTEPS1; ++kk) bw[kk] = wfrag[kk * 64 + lane];
  const float bj = bias[li];
  const unsigned sh = 8u * (unsigned)h;
  for (int b0 = blockIdx.x * FWD_NIMG; b0 < B; b0 += gridDim.x * FWD_NIMG) {
    const int nimg = min(FWD_NIMG, B - b0);
    __syncthreads();   // the previous iteration's tiles have been read
    {
      const uint4* src = reinterpret_cast<const uint4*>(x + (size_t)b0 * img_bytes);
      uint4* dst = reinterpret_cast<uint4*>(smem);
      const int n16 = nimg * img_bytes / 16;
      for (int i = tid; i < n16; i += 256) dst[i] = src[i];
    }
    __syncthreads();
    // two tiles per wave at a time: their MFMA chains are independent, so neither waits for the other's accumulator,
    // and the 128 weight fragments are used twice per read-out; a last odd tile runs alone
    const int nt = nimg * ntile;
    for (int u = wave; u < nt; u += 8) {
      const bool two = u + 4 < nt;
      const int u1 = two ? u + 4 : u;
      const int im0 = u / ntile, mt0 = u - im0 * ntile, im1 = u1 / ntile, mt1 = u1 - im1 * ntile;
      const int m0 = min(mt0 * 32 + li, npix - 1), m1 = min(mt1 * 32 + li, npix - 1);
      const int oh0 = m0 / OW, ow0 = m0 - oh0 * OW, oh1 = m1 / OW, ow1 = m1 - oh1 * OW;
      const uint8_t* base0 = smem + im0 * img_bytes + (oh0 * ST1) * W + ow0 * ST1;
      const uint8_t* base1 = smem + im1 * img_bytes + (oh1 * ST1) * W + ow1 * ST1;
      f32x16 acc0, acc1;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;
      if (two) {
#pragma unroll
        for (int c = 0; c < C1; ++c) {
#pragma unroll
          for (int ki = 0; ki < KW1; ++ki) {
            const uint32_t* p = reinterpret_cast<const uint32_t*>(base0 + c * HW + ki * W);
            const uint32_t* q = reinterpret_cast<const uint32_t*>(base1 + c * HW + ki * W);
            const uint32_t d0 = p[0] >> sh, d1 = p[1] >> sh, e0 = q[0] >> sh, e1 = q[1] >> sh;
            const int kk0 = c * 32 + ki * 4;
            acc0 = mfma32((float)(d0 & 0xffu) * scale, bw[kk0 + 0], acc0);
            acc1 = mfma32((float)(e0 & 0xffu) * scale, bw[kk0 + 0], acc1);
            acc0 = mfma32((float)((d0 >> 16) & 0xffu) * scale, bw[kk0 + 1], acc0);
            acc1 = mfma32((float)((e0 >> 16) & 0xffu) * scale, bw[kk0 + 1], acc1);
            acc0 = mfma32((float)(d1 & 0xffu) * scale, bw[kk0 + 2], acc0);
            acc1 = mfma32((float)(e1 & 0xffu) * scale, bw[kk0 + 2], acc1);
            acc0 = mfma32((float)((d1 >> 16) & 0xffu) * scale, bw[kk0 + 3], acc0);
            acc1 = mfma32((float)((e1 >> 16) & 0xffu) * scale, bw[kk0 + 3], acc1);
          }
        }
      } else {
#pragma unroll
        for (int c = 0; c < C1; ++c) {
#pragma unroll
          for (int ki = 0; ki < KW1; ++ki) {
            const uint32_t* p = reinterpret_cast<const uint32_t*>(base0 + c * HW + ki * W);
            const uint32_t d0 = p[0] >> sh, d1 = p[1] >> sh;   // lane half h takes bytes h and h + 2 of each dword
            const int kk0 = c * 32 + ki * 4;
            acc0 = mfma32((float)(d0 & 0xffu) * scale, bw[kk0 + 0], acc0);
            acc0 = mfma32((float)((d0 >> 16) & 0xffu) * scale, bw[kk0 + 1], acc0);
            acc0 = mfma32((float)(d1 & 0xffu) * scale, bw[kk0 + 2], acc0);
            acc0 = mfma32((float)((d1 >> 16) & 0xffu) * scale, bw[kk0 + 3], acc0);
          }
        }
      }
      float* orow0 = out + ((size_t)(b0 + im0) * npix) * CO1 + li;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = mt0 * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (row < npix) orow0[(size_t)row * CO1] = fmaxf(acc0[r] + bj, 0.f);
      }
      if (two) {
        float* orow1 = out + ((size_t)(b0 + im1) * npix) * CO1 + li;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = mt1 * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          if (row < npix) orow1[(size_t)row * CO1] = fmaxf(acc1[r] + bj, 0.f);
        }
      }
    }
  }
}

__global__ __launch_bounds__(256, 2) void conv1_wgrad_kernel(const uint8_t* __restrict__ x, const float* __restrict__ dout,
                                                             float scale, int H, int W, int OH, int OW, int B,
                                                             float* __restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, li = lane & 31;
  const int HW = H * W, img_bytes = C1 * HW, npix = OH * OW;
  float* dl = reinterpret_cast<float*>(smem + ((img_bytes + 15) & ~15));   // [npix][32]
  const int koff = wave * HW + (li >> 3) * W + (li & 7);   // k = wave * 64 + li (+32): c = wave, ki = li >> 3 (+4), kj = li & 7
  f32x16 acc0, acc1;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;
  float bsum = 0.f;
  for (int b = blockIdx.x; b < B; b += gridDim.x) {
    __syncthreads();
    {
      const uint4* src = reinterpret_cast<const uint4*>(x + (size_t)b * img_bytes);
      uint4* dst = reinterpret_cast<uint4*>(smem);
      for (int i = tid; i < img_bytes / 16; i += 256) dst[i] = src[i];
      const uint4* dsrc = reinterpret_cast<const uint4*>(dout + (size_t)b * npix * CO1);
      uint4* ddst = reinterpret_cast<uint4*>(dl);
      for (int i = tid; i < npix * CO1 / 4; i += 256) ddst[i] = dsrc[i];
    }
    __syncthreads();
    // pixel m = oh * OW + 2 j + h of step (oh, j): OW is even, so a step never straddles two output rows and the
    // frame / dout addresses advance by constants (the inner loop unrolls; its reads run ahead of the MFMAs)
    const int half = OW >> 1;
    for (int oh = 0; oh < OH; ++oh) {
      const uint8_t* ap = smem + koff + oh * ST1 * W + h * ST1;
      const float* bp = dl + (oh * OW + h) * CO1 + li;
      constexpr int U = 5;   // steps whose operands are requested together (84 x 84 frames: 10 steps per row)
      int j = 0;
      for (; j + U <= half; j += U) {
        uint8_t x0[U], x1[U];
        float bv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          x0[u] = ap[(j + u) * 2 * ST1];
          x1[u] = ap[(j + u) * 2 * ST1 + 4 * W];
          bv[u] = bp[(j + u) * 2 * CO1];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          acc0 = mfma32((float)x0[u] * scale, bv[u], acc0);
          acc1 = mfma32((float)x1[u] * scale, bv[u], acc1);
          bsum += bv[u];
        }
      }
      for (; j < half; ++j) {
        const float a0 = (float)ap[j * 2 * ST1] * scale;
        const float a1 = (float)ap[j * 2 * ST1 + 4 * W] * scale;
        const float b1 = bp[j * 2 * CO1];
        acc0 = mfma32(a0, b1, acc0);
        acc1 = mfma32(a1, b1, acc1);
        bsum += b1;
      }
    }
  }
  float* pb = part + (size_t)blockIdx.x * (CO1 * KTOT1 + CO1);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int i = (r & 3) + 8 * (r >> 2) + 4 * h;   // k within the tile; column = lane & 31 = co
    pb[li * KTOT1 + wave * 64 + i] = acc0[r];
    pb[li * KTOT1 + wave * 64 + 32 + i] = acc1[r];
  }
  if (wave == 0) {
    bsum += __shfl_xor(bsum, 32, 64);
    if (h == 0) pb[CO1 * KTOT1 + li] = bsum;
  }
}

// out[i] (+)= sum_b part[b][i]: 64 outputs per block, four waves take a quarter of the workgroup partials each
// (ascending), the quarters are added in order -- fixed summation order
__global__ __launch_bounds__(256) void conv1_wgrad_reduce_kernel(const float* __restrict__ part, int nblk, int accumulate,
                                                                 float* __restrict__ dW, float* __restrict__ db) {
  constexpr int PER = CO1 * KTOT1 + CO1;
  __shared__ float q[4][64];
  const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int e = min((int)blockIdx.x * 64 + lane, PER - 1);
  const int per_g = (nblk + 3) >> 2, lo = g * per_g, hi = min(nblk, lo + per_g);
  float s = 0.f;
  int b = lo;
  for (; b + 8 <= hi; b += 8) {
    float t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) t[u] = part[(size_t)(b + u) * PER + e];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += t[u];
  }
  for (; b < hi; ++b) s += part[(size_t)b * PER + e];
  q[g][lane] = s;
  __syncthreads();
  if (g == 0 && (int)blockIdx.x * 64 + lane < PER) {
    const float tot = ((q[0][lane] + q[1][lane]) + q[2][lane]) + q[3][lane];
    float* dst = e < CO1 * KTOT1 ? dW + e : db + (e - CO1 * KTOT1);
    *dst = accumulate ? *dst + tot : tot;
  }
}

inline bool conv1_shape_ok(int C, int H, int W, int KH, int KW, int S, int Cout) {
  if (C != C1 || KH != KW1 || KW != KW1 || S != ST1 || Cout != CO1) return false;
  if (H < KW1 || W < KW1 || W % 4 != 0 || (C * H * W) % 16 != 0) return false;
  const int OH = (H - KW1) / ST1 + 1, OW = (W - KW1) / ST1 + 1;
  const size_t fwd = (size_t)FWD_NIMG * C * H * W;
  const size_t wg = (((size_t)C * H * W + 15) & ~(size_t)15) + (size_t)((OH * OW + 1) & ~1) * CO1 * sizeof(float);
  return OW >= 2 && OW % 2 == 0 && fwd <= 64 * 1024 && wg <= 80 * 1024;   // two workgroups per CU keep their images in LDS
}

inline int conv1_grid(int B, int per_block) {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
      cus = 256;
  }
  const int want = (B + per_block - 1) / per_block;
  return want < 2 * cus ? want : 2 * cus;
}

}  // namespace

extern "C" {

int ia_conv1_u8_implicit_ok(int C, int H, int W, int KH, int KW, int S, int Cout) {
  return conv1_shape_ok(C, H, W, KH, KW, S, Cout) ? 1 : 0;
}

// ws: 128 * 64 floats (the weight fragments in lane order)
int ia_conv1_u8_forward(const uint8_t* x, int B, int H, int W, const float* weight, const float* bias, float scale,
                        float* ws, float* out, void* stream) {
  if (!x || !weight || !bias || !ws || !out || B <= 0 || !conv1_shape_ok(C1, H, W, KW1, KW1, ST1, CO1)) return IA_ERR_ARG;
  if ((reinterpret_cast<uintptr_t>(x) & 15) != 0) return IA_ERR_ARG;
  const int OH = (H - KW1) / ST1 + 1, OW = (W - KW1) / ST1 + 1;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(conv1_wfrag_kernel, dim3(KSTEPS1 * 64 / 256), dim3(256), 0, st, weight, ws);
  IA_CHECK_LAUNCH();
  const size_t bytes = (size_t)FWD_NIMG * C1 * H * W;
  static size_t attr = 0;
  if (bytes > attr) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv1_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)bytes) != hipSuccess)
      return IA_ERR_ARG;
    attr = bytes;
  }
  hipLaunchKernelGGL(conv1_fwd_kernel, dim3(conv1_grid(B, FWD_NIMG)), dim3(256), bytes, st, x, ws, bias, scale, H, W, OH, OW,
                     B, out);
  IA_CHECK_LAUNCH();
  return IA_OK;
}

long long ia_conv1_u8_wgrad_ws_floats(int B) {
  return B <= 0 ? 0 : (long long)conv1_grid(B, 1) * (CO1 * KTOT1 + CO1);
}

// dout: [B * OH * OW, 32] (gradient w.r.t. the pre-activation, i.e. already masked by the ReLU); dW [32, 256], db [32]
int ia_conv1_u8_wgrad(const uint8_t* x, int B, int H, int W, const float* dout, float scale, float* ws, int accumulate,
                      float* dW, float* db, void* stream) {
  if (!x || !dout || !ws || !dW || !db || B <= 0 || !conv1_shape_ok(C1, H, W, KW1, KW1, ST1, CO1)) return IA_ERR_ARG;
  if ((reinterpret_cast<uintptr_t>(x) & 15) != 0 || (reinterpret_cast<uintptr_t>(dout) & 15) != 0) return IA_ERR_ARG;
  const int OH = (H - KW1) / ST1 + 1, OW = (W - KW1) / ST1 + 1, npix = OH * OW;
  const size_t bytes = (((size_t)C1 * H * W + 15) & ~(size_t)15) + (size_t)npix * CO1 * sizeof(float);
  static size_t attr = 0;
  if (bytes > attr) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv1_wgrad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)bytes) != hipSuccess)
      return IA_ERR_ARG;
    attr = bytes;
  }
  hipStream_t st = (hipStream_t)stream;
  const int grid = conv1_grid(B, 1);
  hipLaunchKernelGGL(conv1_wgrad_kernel, dim3(grid), dim3(256), bytes, st, x, dout, scale, H, W, OH, OW, B, ws);
  IA_CHECK_LAUNCH();
  hipLaunchKernelGGL(conv1_wgrad_reduce_kernel, dim3((CO1 * KTOT1 + CO1 + 63) / 64), dim3(256), 0, st, ws, grid, accumulate,
                     dW, db);
  IA_CHECK_LAUNCH();
  return IA_OK;
}

}  // extern "C"
