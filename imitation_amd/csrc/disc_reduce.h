// The slab reduction + Adam step that ends a fused discriminator update (disc_fused.hip K5), as a device function shared
// with gemm.hip: the part that does not depend on the split-K product dW2 rides in that product's launch.
#pragma once
#include "common.h"

struct ReduceArgs {
  const float* src[4]; long long stride[4]; int cnt[4]; long long seg_end[4];   // [W1 b1] | W2 | b2 | [W3 b3]
  long long n; int accumulate; float* grads;
  int adam; float* p; float* m; float* v; float beta1, beta2, eps, wd, step_size, bc2_sqrt;
  const float* part; int tiles; int R; int n_expert; float loss_scale; float* stats;
  float* W2T; float* W1P; int H; int D; int xp;   // images of W2 / W1 ([H][xp]) the tile kernels read: refreshed with the Adam step
  // gradient penalty: a second slab set per segment, summed behind the first (cnt2 = 0: none), and the penalty's mean
  const float* src2[4]; long long stride2[4]; int cnt2[4];
  const float* pen; int pen_tiles; int gp_rows; float* gp_out;
  // which part of the update THIS launch covers: up to two element ranges (multiples of 64 apart from the last end), the
  // first over `nb0` blocks; `do_stats`: one more block folds the statistics partials (and the penalty's mean)
  long long r_begin[2], r_end[2]; int nb0; int do_stats;
};

// 64 parameters per block; wave q folds quarter q of the element's slabs in slab order, the four quarter
// sums are combined in fixed order -> deterministic. The last block (`do_stats`) folds the statistics partials.
// A device function: the launch that covers what does NOT depend on the split-K product (first / last layer, b2) runs as
// the leading workgroups of that product's own launch (gemm.hip, ia_launch_gemm_tn_side) -- memory-bound work beside a
// matrix-pipe-bound kernel; `disc_reduce_kernel` behind the product then only folds dW2's split slabs.
__device__ __forceinline__ void disc_reduce_block(const ReduceArgs& a, const int vb, const int nvb) {
  __shared__ float red[4][64];
  const int tid = threadIdx.x, lane = tid & 63, q = tid >> 6;
  if (a.do_stats && vb == nvb - 1) {
    float vals[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int t = tid; t < a.tiles; t += 256) {
#pragma unroll
      for (int k = 0; k < 6; ++k) vals[k] += a.part[(long long)t * 8 + k];
    }
    __shared__ float sred[6][4];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      float v = vals[k];
      for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
      if (lane == 0) sred[k][q] = v;
    }
    __syncthreads();
    if (tid < 6) {
      float t = ((sred[tid][0] + sred[tid][1]) + sred[tid][2]) + sred[tid][3];
      if (tid == 0) t = t / (float)a.R * a.loss_scale;
      a.stats[tid] = t;
    }
    if (tid == 6) a.stats[6] = (float)a.n_expert;
    if (tid == 7) a.stats[7] = (float)(a.R - a.n_expert);
    if (a.gp_out != nullptr) {   // mean_i (|grad_x D(x_hat_i)| - target)^2 from the tiles' partial sums, fixed order
      float v = 0.f;
      for (int t = tid; t < a.pen_tiles; t += 256) v += a.pen[t];
      for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
      __syncthreads();
      if (lane == 0) sred[0][q] = v;
      __syncthreads();
      if (tid == 0) a.gp_out[0] = (((sred[0][0] + sred[0][1]) + sred[0][2]) + sred[0][3]) / (float)a.gp_rows;
    }
    return;
  }
  const int rg = vb < a.nb0 ? 0 : 1;
  const long long i = a.r_begin[rg] + (long long)(vb - (rg ? a.nb0 : 0)) * 64 + lane;
  const long long r_end = a.r_end[rg];
  const long long ic = i < r_end ? i : r_end - 1;
  const int seg = ic < a.seg_end[0] ? 0 : (ic < a.seg_end[1] ? 1 : (ic < a.seg_end[2] ? 2 : 3));
  const long long base = seg == 0 ? 0 : a.seg_end[seg - 1];
  const float* src = a.src[seg] + (ic - base);
  const long long st = a.stride[seg];
  const int cnt = a.cnt[seg];
  const int lo = (int)((long long)q * cnt / 4), hi = (int)((long long)(q + 1) * cnt / 4);
  float s = 0.f;
  int k = lo;
  for (; k + 8 <= hi; k += 8) {
    float t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) t[u] = src[(long long)(k + u) * st];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += t[u];
  }
  for (; k < hi; ++k) s += src[(long long)k * st];
  if (a.cnt2[seg] > 0) {
    const float* src2 = a.src2[seg] + (ic - base);
    const long long st2 = a.stride2[seg];
    const int cnt2 = a.cnt2[seg];
    const int lo2 = (int)((long long)q * cnt2 / 4), hi2 = (int)((long long)(q + 1) * cnt2 / 4);
    int k2 = lo2;
    for (; k2 + 8 <= hi2; k2 += 8) {
      float t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = src2[(long long)(k2 + u) * st2];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += t[u];
    }
    for (; k2 < hi2; ++k2) s += src2[(long long)k2 * st2];
  }
  red[q][lane] = s;
  __syncthreads();
  if (q != 0 || i >= r_end) return;
  float grad = ((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane];
  if (a.accumulate) grad = a.grads[i] + grad;
  a.grads[i] = grad;
  if (!a.adam) return;
  // torch/optim/adam.py _single_tensor_adam: lerp, mul+addcmul, sqrt/bc2_sqrt + eps, addcdiv
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
  // the next update may skip the assemble launch (pre-assembled rounds): W2T / the padded W1 image follow here
  const long long nW1 = (long long)a.H * a.D, n1 = nW1 + a.H;
  if (i < nW1) {
    const int n = (int)(i / a.D), k = (int)(i - (long long)n * a.D);
    a.W1P[n * a.xp + k] = pn;
  } else if (i >= n1 && i < n1 + (long long)a.H * a.H) {
    const int j = (int)(i - n1), r = j / a.H, c = j - r * a.H;
    a.W2T[(long long)c * a.H + r] = pn;
  }
}


// blocks a launch of `disc_reduce_block` needs for the ranges in `a`
static inline int disc_reduce_blocks(const ReduceArgs& a) {
  const int nb1 = a.r_end[1] > a.r_begin[1] ? (int)((a.r_end[1] - a.r_begin[1] + 63) / 64) : 0;
  return a.nb0 + nb1 + (a.do_stats ? 1 : 0);
}

// gemm.hip: split-K TN product (64 x 64 tiles) whose first `disc_reduce_blocks(side)` workgroups run `side`
int ia_launch_gemm_tn_side(const IaGemm& g, const ReduceArgs& side, hipStream_t stream);
