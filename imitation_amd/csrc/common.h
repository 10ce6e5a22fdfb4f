// Shared declarations for the gfx950 (MI355X / CDNA4) kernels of the GAIL/AIRL round.
// Wave = 64 lanes; fp32 MFMA = v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain, 157 TF peak).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define IA_OK 0
#define IA_ERR_ARG (-1)
#define IA_ERR_UNSUPPORTED (-2)

#define IA_ACT_NONE 0
#define IA_ACT_RELU 1
#define IA_ACT_TANH 2
#define IA_ACT_SOFTPLUS 3  // max(x,0)+log1p(exp(-|x|)) == -logsigmoid(-x)  (gail.py:75-83)

#define IA_CHECK_LAUNCH()                         \
  do {                                            \
    hipError_t e__ = hipGetLastError();           \
    if (e__ != hipSuccess) return (int)e__;       \
  } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- GEMM (gemm.hip) -------------------------------------------------------------------
enum { IA_GEMM_NT = 0, IA_GEMM_NN = 1, IA_GEMM_TN = 2 };

// Implicit im2col view of a channel-last activation tensor x[B, H, W, C] (no column buffer): row m = (b, oh, ow),
// column k = (i, j, c) -> x[b, oh*S + i, ow*S + j, c]. A kernel row (fixed i) is one contiguous run of `seg` = KW*C
// floats; divisions go through precomputed 64-bit reciprocals (`m*`), exact for every 32-bit index.
struct IaIm {
  int on;                       // 0: the operand is a plain matrix
  int OW, OHW, W, C, S, HWC;    // output width, output pixels per image, input width / channels / stride, H*W*C
  int seg, rstride;             // KW*C, W*C
  unsigned long long mOW, mOHW, mseg;   // floor((2^64 - 1) / d) + 1: __umul64hi(n, m) == n / d for every 32-bit n
  int pad, H;                   // zero padding on every side (taps outside the image read as 0), input height
  unsigned long long mC;
  // optional scatter of the OUTPUT rows (NT): row m = (b, y', x') of a cm_OW-wide grid with cm_OHW cells per image
  // goes to row b*cm_HW + (y'*cm_S + cm_py)*cm_W + x'*cm_S + cm_px of C (transposed convolution by sub-pixel classes)
  // cm_on == 2: all cm_S^2 classes in ONE GEMM -- output column = class * cm_C + channel, class = py * cm_S + px
  int cm_on, cm_OW, cm_OHW, cm_S, cm_py, cm_px, cm_W, cm_HW, cm_C;
  unsigned long long cm_mOW, cm_mOHW, cm_mC;
};

struct IaGemm {
  IaIm im;              // NT: A is the view; TN: B is (the k-contiguous [rows, K] operand of the convolution)
  const float* A;
  const float* B;
  float* C;
  int M, N, K;          // C is [M,N]; K is the reduction length
  int lda, ldb, ldc;
  const float* bias;    // NT: [N] added before the activation (may be null)
  int act;              // NT: epilogue activation; NN: derivative kind applied with P
  const float* P;       // NN: post-activation values of the layer whose input-grad this is
  int ldp;
  int nt_split;         // NT only: the K splits below apply, slab s of C receives split s's products WITHOUT bias / activation
  int splits;           // TN: number of K splits (grid.z)
  int k_per_split;      // TN: rows of K handled by one split (multiple of 32)
  long long c_split_stride;   // TN: elements between consecutive split slabs of C
  float* dbias;               // TN: per-split column sums of A (bias gradient), may be null
  long long dbias_split_stride;  // TN: elements between split slabs of dbias
};

int ia_launch_gemm(int mode, const IaGemm& g, hipStream_t stream);
// n <= 6 independent split-K TN GEMMs whose outputs have <= 32 rows, in one launch
int ia_launch_gemm_group_tn(const IaGemm* gs, int n, hipStream_t stream);
// mlp.hip: grads = scale * sum(slabs of `partials`) + sum(slabs of `partials2`), then torch's Adam step -- one launch
int ia_reduce2_partials_adam(const float* partials, int splits, const float* partials2, int splits2, int64_t n, float scale,
                             float* grads, float* params, float* exp_avg, float* exp_avg_sq, float beta1, float beta2,
                             float eps, float weight_decay, float step_size, float bc2_sqrt, hipStream_t stream);

__device__ __forceinline__ float ia_softplus(float x) {
  return fmaxf(x, 0.0f) + log1pf(expf(-fabsf(x)));
}
__device__ __forceinline__ float ia_apply_act(float x, int act) {
  switch (act) {
    case IA_ACT_RELU: return fmaxf(x, 0.0f);
    case IA_ACT_TANH: return tanhf(x);
    case IA_ACT_SOFTPLUS: return ia_softplus(x);
    default: return x;
  }
}
// derivative expressed through the POST-activation value p
__device__ __forceinline__ float ia_act_grad_from_post(float p, int act) {
  switch (act) {
    case IA_ACT_RELU: return p > 0.0f ? 1.0f : 0.0f;
    case IA_ACT_TANH: return 1.0f - p * p;
    default: return 1.0f;
  }
}
