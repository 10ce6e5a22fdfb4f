// Image path of the BC step (BASELINE config 4: NatureCNN on [C,H,W] uint8 frames):
// convolutions as im2col + the fp32 MFMA GEMMs of gemm.hip (activations NHWC, weights in torch's
// [Cout, Cin*KH*KW] layout = the GEMM's B operand as stored), their input gradient as a GEMM + a gather-
// form col2im (each input element sums the windows that cover it: no atomics, fixed order), and the
// Categorical head (log-prob of the expert action, entropy, gradient of the BC loss w.r.t. the logits).
// First version: explicit column buffers in HBM (288 GB: batch 4096 x 84x84x4 needs 1.7 GB for the first
// layer) -- every kernel here is a streaming gather bound by HBM; the flops are in the GEMMs.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "common.h"

namespace {

inline int cdiv_ll(long long a, long long b) { return (int)((a + b - 1) / b); }

// Both im2col kernels: a block walks IM2COL_ROWS consecutive rows m = (b, oh, ow); the row -> input base
// address arithmetic is wave-uniform (scalar unit), every thread keeps the input offsets of its own
// columns k = tid + 256 t in registers (c, i, j resolved once), so an element costs one add, one load and
// one coalesced store (the first version spent its time in per-element 64-bit divisions: 1.1 TB/s).
constexpr int IM2COL_ROWS = 16;
constexpr int IM2COL_KPT = 4;   // columns per thread: K <= 1024

// col[m][k = (c, i, j)] = x[b, c, oh*S+i, ow*S+j] * scale     (x: uint8, channel-first)
__global__ __launch_bounds__(256) void im2col_u8_nchw_kernel(const uint8_t* __restrict__ x, int C, int H, int W, int KH,
                                                             int KW, int S, int OH, int OW, float scale,
                                                             float* __restrict__ col, long long M) {
  const int K = C * KH * KW;
  int koff[IM2COL_KPT];
#pragma unroll
  for (int t = 0; t < IM2COL_KPT; ++t) {
    const int k = min((int)threadIdx.x + 256 * t, K - 1);
    const int c = k / (KH * KW), r = k - c * (KH * KW);
    const int i = r / KW, j = r - i * KW;
    koff[t] = (c * H + i) * W + j;
  }
  const long long m0 = (long long)blockIdx.x * IM2COL_ROWS;
  for (int rr = 0; rr < IM2COL_ROWS; ++rr) {
    const long long m = m0 + rr;
    if (m >= M) return;
    const int p = (int)(m % (OH * OW));
    const long long b = m / (OH * OW);
    const int oh = p / OW, ow = p - oh * OW;
    const uint8_t* src = x + (b * C * H + oh * S) * W + ow * S;
    float* dst = col + m * K;
#pragma unroll
    for (int t = 0; t < IM2COL_KPT; ++t) {
      const int k = threadIdx.x + 256 * t;
      if (k < K) dst[k] = (float)src[koff[t]] * scale;
    }
  }
}

// KW == 8 with 4-byte aligned windows (NatureCNN's first layer: 8x8, stride 4, W % 4 == 0): a thread owns one
// (c, i) pair of a row = 8 consecutive bytes in, 8 consecutive floats out (two dword loads, two 16-byte
// stores); 256 threads cover 256 / (C*KH) rows at a time.
__global__ __launch_bounds__(256) void im2col_u8_nchw_kw8_kernel(const uint8_t* __restrict__ x, int C, int H, int W,
                                                                 int KH, int S, int OH, int OW, float scale,
                                                                 float* __restrict__ col, long long M, int rows_per_pass) {
  const int CI = C * KH;                       // (c, i) pairs per row; K = CI * 8
  const int sub = threadIdx.x / CI, ci = threadIdx.x - sub * CI;
  if (sub >= rows_per_pass) return;
  const int c = ci / KH, i = ci - c * KH;
  const int koff = (c * H + i) * W;
  typedef float f4v __attribute__((ext_vector_type(4)));
  const long long m0 = (long long)blockIdx.x * IM2COL_ROWS * rows_per_pass;
  for (int rr = 0; rr < IM2COL_ROWS; ++rr) {
    const long long m = m0 + (long long)rr * rows_per_pass + sub;
    if (m >= M) return;
    const int p = (int)(m % (OH * OW));
    const long long b = m / (OH * OW);
    const int oh = p / OW, ow = p - oh * OW;
    const uint32_t* src = reinterpret_cast<const uint32_t*>(x + (b * C * H + oh * S) * W + ow * S + koff);
    const uint32_t lo = src[0], hi = src[1];
    f4v a = {(float)(lo & 255u) * scale, (float)((lo >> 8) & 255u) * scale, (float)((lo >> 16) & 255u) * scale,
             (float)(lo >> 24) * scale};
    f4v bq = {(float)(hi & 255u) * scale, (float)((hi >> 8) & 255u) * scale, (float)((hi >> 16) & 255u) * scale,
              (float)(hi >> 24) * scale};
    f4v* dst = reinterpret_cast<f4v*>(col + m * (CI * 8) + ci * 8);
    dst[0] = a;
    dst[1] = bq;
  }
}

// col[m][k = (i, j, c)] = x[b, oh*S+i, ow*S+j, c]              (x: fp32, channel-last)
// Column order (i, j, c) -- the weights of these layers are kept as [Cout, KH, KW, Cin] on the device -- so a
// row is KH runs of KW*C contiguous input floats: reads and writes are both coalesced.
__global__ __launch_bounds__(256) void im2col_f32_nhwc_kernel(const float* __restrict__ x, int C, int H, int W, int KH,
                                                              int KW, int S, int OH, int OW, float* __restrict__ col,
                                                              long long M) {
  const int K = C * KH * KW;
  int koff[IM2COL_KPT];
#pragma unroll
  for (int t = 0; t < IM2COL_KPT; ++t) {
    const int k = min((int)threadIdx.x + 256 * t, K - 1);
    const int i = k / (KW * C), r = k - i * (KW * C);
    koff[t] = i * W * C + r;
  }
  const long long m0 = (long long)blockIdx.x * IM2COL_ROWS;
  for (int rr = 0; rr < IM2COL_ROWS; ++rr) {
    const long long m = m0 + rr;
    if (m >= M) return;
    const int p = (int)(m % (OH * OW));
    const long long b = m / (OH * OW);
    const int oh = p / OW, ow = p - oh * OW;
    const float* src = x + ((b * H + oh * S) * W + ow * S) * C;
    float* dst = col + m * K;
#pragma unroll
    for (int t = 0; t < IM2COL_KPT; ++t) {
      const int k = threadIdx.x + 256 * t;
      if (k < K) dst[k] = src[koff[t]];
    }
  }
}

// dx[b, h, w, c] = sum over the windows (oh, ow, i, j) with oh*S+i == h, ow*S+j == w of dcol[(b,oh,ow)][(i,j,c)],
// in (i, j) order; multiplied by [mask[b,h,w,c] > 0] when a mask (the ReLU output of that layer) is given.
// (Consecutive threads = consecutive c: every read of a window is a contiguous run of C floats.)
__global__ void col2im_nhwc_kernel(const float* __restrict__ dcol, int C, int H, int W, int KH, int KW, int S, int OH,
                                   int OW, const float* __restrict__ mask, float* __restrict__ dx, long long total) {
  const int K = C * KH * KW;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(e % C);
    const long long t = e / C;
    const int w = (int)(t % W);
    const long long t2 = t / W;
    const int h = (int)(t2 % H);
    const long long b = t2 / H;
    // windows covering (h, w): oh in [ceil((h-KH+1)/S), floor(h/S)] n [0, OH), likewise ow -- at most
    // ceil(KH/S) x ceil(KW/S) terms, visited in decreasing (oh, ow) = increasing (i, j) order
    float s = 0.f;
    const int oh_hi = min(h / S, OH - 1), oh_lo = max(0, (h - KH + S) / S);
    const int ow_hi = min(w / S, OW - 1), ow_lo = max(0, (w - KW + S) / S);
    for (int oh = oh_hi; oh >= oh_lo; --oh) {
      const int i = h - oh * S;
      for (int ow = ow_hi; ow >= ow_lo; --ow) {
        const int j = w - ow * S;
        s += dcol[((b * OH + oh) * OW + ow) * (long long)K + (i * KW + j) * C + c];
      }
    }
    if (mask != nullptr && !(mask[e] > 0.f)) s = 0.f;
    dx[e] = s;
  }
}

// The same for C % 4 == 0 and fewer than 2^31 channel quads: a thread owns FOUR consecutive channels of one input pixel
// (16-byte loads and stores, a quarter of the memory instructions) and decodes its pixel with 32-bit arithmetic (the
// element-per-thread form above spends four 64-bit divisions per element).
__global__ __launch_bounds__(256) void col2im_nhwc_v4_kernel(const float* __restrict__ dcol, int C, int H, int W, int KH,
                                                             int KW, int S, int OH, int OW, const float* __restrict__ mask,
                                                             float* __restrict__ dx, int total4) {
  const int K = C * KH * KW, q = C >> 2;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total4; e += gridDim.x * blockDim.x) {
    const int t = e / q, cq = e - t * q;
    const int t2 = t / W, w = t - t2 * W;
    const int b = t2 / H, h = t2 - b * H;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    const int oh_hi = min(h / S, OH - 1), oh_lo = max(0, (h - KH + S) / S);
    const int ow_hi = min(w / S, OW - 1), ow_lo = max(0, (w - KW + S) / S);
    for (int oh = oh_hi; oh >= oh_lo; --oh) {
      const int i = h - oh * S;
      for (int ow = ow_hi; ow >= ow_lo; --ow) {
        const int j = w - ow * S;
        s += *reinterpret_cast<const f32x4*>(dcol + ((long long)(b * OH + oh) * OW + ow) * K + (i * KW + j) * C + cq * 4);
      }
    }
    if (mask != nullptr) {
      const f32x4 m = *reinterpret_cast<const f32x4*>(mask + (long long)e * 4);
      s.x = m.x > 0.f ? s.x : 0.f;
      s.y = m.y > 0.f ? s.y : 0.f;
      s.z = m.z > 0.f ? s.z : 0.f;
      s.w = m.w > 0.f ? s.w : 0.f;
    }
    *reinterpret_cast<f32x4*>(dx + (long long)e * 4) = s;
  }
}

// One row per lane. logits [B][ldl]; act = expert action index (fp32). log-softmax z, p = exp(z):
//   logp = z[act], H = -sum_k p_k z_k  (torch Categorical.log_prob / .entropy),
//   dlogits_k = c_lp * ([k == act] - p_k) + c_ent * (-p_k (z_k + H))     (= d(c_lp*logp + c_ent*H)/dlogit_k)
__global__ void categorical_loss_kernel(const float* __restrict__ logits, int ldl, const float* __restrict__ act, int B,
                                        int A, float c_lp, float c_ent, float* __restrict__ logp,
                                        float* __restrict__ entropy, float* __restrict__ dlogits) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= B) return;
  const float* row = logits + (long long)r * ldl;
  float mx = row[0];
  for (int k = 1; k < A; ++k) mx = fmaxf(mx, row[k]);
  float se = 0.f;
  for (int k = 0; k < A; ++k) se += expf(row[k] - mx);
  const float lse = mx + logf(se);
  const int a = (int)act[r];
  float H = 0.f;
  for (int k = 0; k < A; ++k) {
    const float z = row[k] - lse;
    H -= expf(z) * z;
  }
  logp[r] = row[a] - lse;
  entropy[r] = H;
  if (dlogits != nullptr) {
    float* drow = dlogits + (long long)r * ldl;
    for (int k = 0; k < A; ++k) {
      const float z = row[k] - lse, p = expf(z);
      drow[k] = c_lp * ((k == a ? 1.f : 0.f) - p) + c_ent * (-p * (z + H));
    }
  }
}

// ---- padded convolutions of the reward CNN (util/networks.py:286-357 `build_cnn`: Conv2d(k, stride, padding) -
// ReLU ..., AdaptiveAvgPool2d(1), Linear). Same im2col / GEMM / col2im scheme with a zero border of P pixels
// resolved in the index arithmetic (no padded copy of the activations).

// col[m = (b, oh, ow)][k = (i, j, c)] = x[b, oh*S + i - P, ow*S + j - P, c] (0 outside the image); x channel-last
__global__ void im2col_f32_nhwc_pad_kernel(const float* __restrict__ x, int C, int H, int W, int KH, int KW, int S,
                                           int P, int OH, int OW, float* __restrict__ col, long long total) {
  const int K = C * KH * KW;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int k = (int)(e % K);
    const long long m = e / K;
    const int c = k % C, ij = k / C;
    const int i = ij / KW, j = ij - i * KW;
    const int p = (int)(m % ((long long)OH * OW));
    const long long b = m / ((long long)OH * OW);
    const int oh = p / OW, ow = p - oh * OW;
    const int h = oh * S + i - P, w = ow * S + j - P;
    col[e] = (h >= 0 && h < H && w >= 0 && w < W) ? x[((b * H + h) * W + w) * C + c] : 0.f;
  }
}

// dx[b, h, w, c] = sum over the windows (oh, ow, i, j) with oh*S + i - P == h, ow*S + j - P == w, in (i, j) order
__global__ void col2im_nhwc_pad_kernel(const float* __restrict__ dcol, int C, int H, int W, int KH, int KW, int S,
                                       int P, int OH, int OW, float* __restrict__ dx, long long total) {
  const int K = C * KH * KW;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(e % C);
    const long long t = e / C;
    const int w = (int)(t % W);
    const long long t2 = t / W;
    const int h = (int)(t2 % H);
    const long long b = t2 / H;
    const int hp = h + P, wp = w + P;   // position in the (virtually) padded image
    float s = 0.f;
    const int oh_hi = min(hp / S, OH - 1), oh_lo = max(0, (hp - KH + S) / S);
    const int ow_hi = min(wp / S, OW - 1), ow_lo = max(0, (wp - KW + S) / S);
    for (int oh = oh_hi; oh >= oh_lo; --oh) {
      const int i = hp - oh * S;
      for (int ow = ow_hi; ow >= ow_lo; --ow) {
        const int j = wp - ow * S;
        s += dcol[((b * OH + oh) * OW + ow) * (long long)K + (i * KW + j) * C + c];
      }
    }
    dx[e] = s;
  }
}

__global__ void relu_backward_kernel(const float* __restrict__ dy, const float* __restrict__ y, long long n,
                                     float* __restrict__ out) {
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x)
    out[e] = y[e] > 0.f ? dy[e] : 0.f;
}

// out[b, c] = mean over the HW positions of y[b, :, c] (AdaptiveAvgPool2d(1) on channel-last activations): one
// block per image. Channel QUADS across the threads (consecutive threads read consecutive 16-byte pieces of a position's
// channel row: a wave instruction covers 1 KB), the positions dealt over G = 256 / (C / 4) thread groups; every thread
// sums its positions in order in four interleaved chains (loads of four positions in flight), the groups are folded in
// group order through LDS -- a fixed summation order, whatever the grid. (The first form -- one thread per channel walking
// all HW positions with a stride of C floats, 32 of 256 threads busy at C = 32 -- took 2.7 ms per 1 024 x 84 x 84 x 32
// call: 16 of the image-GAIL round's 93 ms of GPU time, `profiles/r05_image_gail.md`.)
__global__ __launch_bounds__(256) void avgpool_nhwc_kernel(const float* __restrict__ y, int HW, int C,
                                                           float* __restrict__ out) {
  __shared__ float4 red[256];
  const long long b = blockIdx.x;
  const int tid = threadIdx.x;
  const int Q = C >> 2;                      // channel quads
  if ((C & 3) == 0 && Q <= 256) {
    const int G = 256 / Q;                   // position groups
    const int q = tid % Q, g = tid / Q;
    float4 acc[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    if (g < G) {
      const float4* src = reinterpret_cast<const float4*>(y + b * HW * C) + q;
      int p = g;
      for (; p + 3 * G < HW; p += 4 * G) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = src[(long long)(p + u * G) * Q];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          acc[u].x += v[u].x; acc[u].y += v[u].y; acc[u].z += v[u].z; acc[u].w += v[u].w;
        }
      }
      for (int u = 0; p < HW; p += G, ++u) {
        const float4 v = src[(long long)p * Q];
        acc[u].x += v.x; acc[u].y += v.y; acc[u].z += v.z; acc[u].w += v.w;
      }
    }
    float4 s;
    s.x = (acc[0].x + acc[1].x) + (acc[2].x + acc[3].x);
    s.y = (acc[0].y + acc[1].y) + (acc[2].y + acc[3].y);
    s.z = (acc[0].z + acc[1].z) + (acc[2].z + acc[3].z);
    s.w = (acc[0].w + acc[1].w) + (acc[2].w + acc[3].w);
    red[tid] = s;
    __syncthreads();
    if (tid < Q) {
      float4 t = red[tid];
      for (int g2 = 1; g2 < G; ++g2) {
        const float4 v = red[g2 * Q + tid];
        t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
      }
      const float inv = (float)HW;
      float* o = out + b * C + 4 * tid;
      o[0] = t.x / inv; o[1] = t.y / inv; o[2] = t.z / inv; o[3] = t.w / inv;
    }
    return;
  }
  for (int c = tid; c < C; c += 256) {       // other channel counts: a thread per channel, positions in order
    const float* src = y + b * HW * C + c;
    float s = 0.f;
    for (int p = 0; p < HW; ++p) s += src[(long long)p * C];
    out[b * C + c] = s / (float)HW;
  }
}

__global__ void avgpool_nhwc_backward_kernel(const float* __restrict__ dout, int HW, int C, float* __restrict__ dy,
                                             long long total) {
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(e % C);
    const long long b = e / ((long long)HW * C);
    dy[e] = dout[b * C + c] / (float)HW;
  }
}

inline dim3 stream_grid(long long total) {
  const long long blocks = (total + 255) / 256;
  return dim3((unsigned)(blocks < 65536 * 4 ? (blocks > 0 ? blocks : 1) : 65536 * 4));
}

}  // namespace

extern "C" {

int ia_im2col_u8_nchw(const uint8_t* x, int B, int C, int H, int W, int KH, int KW, int S, float scale, float* col,
                      void* stream) {
  if (!x || !col || B <= 0 || C <= 0 || KH <= 0 || KW <= 0 || S <= 0 || H < KH || W < KW) return IA_ERR_ARG;
  const int OH = (H - KH) / S + 1, OW = (W - KW) / S + 1;
  if (C * KH * KW > 256 * IM2COL_KPT) return IA_ERR_ARG;
  const long long M = (long long)B * OH * OW;
  if (KW == 8 && S % 4 == 0 && W % 4 == 0 && C * KH <= 256 && (reinterpret_cast<uintptr_t>(x) & 3) == 0) {
    const int rpp = 256 / (C * KH);
    const long long per_block = (long long)IM2COL_ROWS * rpp;
    hipLaunchKernelGGL(im2col_u8_nchw_kw8_kernel, dim3((unsigned)((M + per_block - 1) / per_block)), dim3(256), 0,
                       (hipStream_t)stream, x, C, H, W, KH, S, OH, OW, scale, col, M, rpp);
    IA_CHECK_LAUNCH();
    return IA_OK;
  }
  hipLaunchKernelGGL(im2col_u8_nchw_kernel, dim3((unsigned)((M + IM2COL_ROWS - 1) / IM2COL_ROWS)), dim3(256), 0,
                     (hipStream_t)stream, x, C, H, W, KH, KW, S, OH, OW, scale, col, M);
  IA_CHECK_LAUNCH();
  return IA_OK;
}

int ia_im2col_f32_nhwc(const float* x, int B, int H, int W, int C, int KH, int KW, int S, float* col, void* stream) {
  if (!x || !col || B <= 0 || C <= 0 || KH <= 0 || KW <= 0 || S <= 0 || H < KH || W < KW) return IA_ERR_ARG;
  const int OH = (H - KH) / S + 1, OW = (W - KW) / S + 1;
  if (C * KH * KW > 256 * IM2COL_KPT) return IA_ERR_ARG;
  const long long M = (long long)B * OH * OW;
  hipLaunchKernelGGL(im2col_f32_nhwc_kernel, dim3((unsigned)((M + IM2COL_ROWS - 1) / IM2COL_ROWS)), dim3(256), 0,
                     (hipStream_t)stream, x, C, H, W, KH, KW, S, OH, OW, col, M);
  IA_CHECK_LAUNCH();
  return IA_OK;
}

int ia_col2im_nhwc(const float* dcol, int B, int H, int W, int C, int KH, int KW, int S, const float* relu_mask,
                   float* dx, void* stream) {
  if (!dcol || !dx || B <= 0 || C <= 0 || KH <= 0 || KW <= 0 || S <= 0 || H < KH || W < KW) return IA_ERR_ARG;
  const int OH = (H - KH) / S + 1, OW = (W - KW) / S + 1;
  const long long total = (long long)B * H * W * C;
  const bool aligned = ((reinterpret_cast<uintptr_t>(dcol) | reinterpret_cast<uintptr_t>(dx) |
                         reinterpret_cast<uintptr_t>(relu_mask)) & 15) == 0;
  if (C % 4 == 0 && total / 4 < (1ll << 31) && aligned) {
    hipLaunchKernelGGL(col2im_nhwc_v4_kernel, stream_grid(total / 4), dim3(256), 0, (hipStream_t)stream, dcol, C, H, W, KH,
                       KW, S, OH, OW, relu_mask, dx, (int)(total / 4));
    IA_CHECK_LAUNCH();
    return IA_OK;
  }
  hipLaunchKernelGGL(col2im_nhwc_kernel, stream_grid(total), dim3(256), 0, (hipStream_t)stream, dcol, C, H, W, KH, KW,
                     S, OH, OW, relu_mask, dx, total);
  IA_CHECK_LAUNCH();
  return IA_OK;
}

int ia_im2col_f32_nhwc_pad(const float* x, int B, int H, int W, int C, int KH, int KW, int S, int P, float* col,
                           void* stream) {
  if (!x || !col || B <= 0 || C <= 0 || KH <= 0 || KW <= 0 || S <= 0 || P < 0 || H + 2 * P < KH || W + 2 * P < KW)
    return IA_ERR_ARG;
  const int OH = (H + 2 * P - KH) / S + 1, OW = (W + 2 * P - KW) / S + 1;
  const long long total = (long long)B * OH * OW * C * KH * KW;
  hipLaunchKernelGGL(im2col_f32_nhwc_pad_kernel, stream_grid(total), dim3(256), 0, (hipStream_t)stream, x, C, H, W, KH,
                     KW, S, P, OH, OW, col, total);
  IA_CHECK_LAUNCH();
  return IA_OK;
}

int ia_col2im_nhwc_pad(const float* dcol, int B, int H, int W, int C, int KH, int KW, int S, int P, float* dx,
                       void* stream) {
  if (!dcol || !dx || B <= 0 || C <= 0 || KH <= 0 || KW <= 0 || S <= 0 || P < 0 || H + 2 * P < KH || W + 2 * P < KW)
    return IA_ERR_ARG;
  const int OH = (H + 2 * P - KH) / S + 1, OW = (W + 2 * P - KW) / S + 1;
  const long long total = (long long)B * H * W * C;
  hipLaunchKernelGGL(col2im_nhwc_pad_kernel, stream_grid(total), dim3(256), 0, (hipStream_t)stream, dcol, C, H, W, KH,
                     KW, S, P, OH, OW, dx, total);
  IA_CHECK_LAUNCH();
  return IA_OK;
}

int ia_relu_backward(const float* dy, const float* y, int64_t n, float* out, void* stream) {
  if (!dy || !y || !out || n <= 0) return IA_ERR_ARG;
  hipLaunchKernelGGL(relu_backward_kernel, stream_grid(n), dim3(256), 0, (hipStream_t)stream, dy, y, (long long)n, out);
  IA_CHECK_LAUNCH();
  return IA_OK;
}

int ia_avgpool_nhwc(const float* y, int B, int HW, int C, float* out, void* stream) {
  if (!y || !out || B <= 0 || HW <= 0 || C <= 0) return IA_ERR_ARG;
  hipLaunchKernelGGL(avgpool_nhwc_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, y, HW, C, out);
  IA_CHECK_LAUNCH();
  return IA_OK;
}

int ia_avgpool_nhwc_backward(const float* dout, int B, int HW, int C, float* dy, void* stream) {
  if (!dout || !dy || B <= 0 || HW <= 0 || C <= 0) return IA_ERR_ARG;
  const long long total = (long long)B * HW * C;
  hipLaunchKernelGGL(avgpool_nhwc_backward_kernel, stream_grid(total), dim3(256), 0, (hipStream_t)stream, dout, HW, C,
                     dy, total);
  IA_CHECK_LAUNCH();
  return IA_OK;
}

int ia_categorical_loss(const float* logits, int ldl, const float* actions, int B, int A, float logp_coef,
                        float ent_coef, float* logp, float* entropy, float* dlogits, void* stream) {
  if (!logits || !actions || !logp || !entropy || B <= 0 || A <= 0 || ldl < A) return IA_ERR_ARG;
  hipLaunchKernelGGL(categorical_loss_kernel, dim3(cdiv_ll(B, 128)), dim3(128), 0, (hipStream_t)stream, logits, ldl,
                     actions, B, A, logp_coef, ent_coef, logp, entropy, dlogits);
  IA_CHECK_LAUNCH();
  return IA_OK;
}

}  // extern "C"
