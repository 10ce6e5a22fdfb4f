// RunningNorm device helpers shared by mlp.hip (stand-alone kernels) and disc_fused.hip (the fused
// discriminator update). util/networks.py:111-134 (Chan et al. merge, the reference's operation order).
#pragma once
#include "common.h"

constexpr int RN_ROWS_PER_BLOCK = 256;

// Chan combination of two (count, mean, M2) moment triples.
__device__ __forceinline__ void chan_combine(float& n, float& m, float& M2, float nb, float mb, float qb) {
  if (nb == 0.f) return;
  const float tot = n + nb;
  const float dlt = mb - m;
  M2 = M2 + qb + dlt * dlt * n * nb / tot;
  m = m + dlt * nb / tot;
  n = tot;
}

// One WAVE merges the slab moments of column `c` (`nblocks` slabs of RN_ROWS_PER_BLOCK rows in groups
// of `bpg`, each group covering `rpg` rows) and returns (batch mean, batch M2) in every lane: lane l
// folds slabs l, l+64, ... sequentially, then a fixed butterfly (xor 32,16,...,1) combines the 64
// partial triples -> deterministic.
// `gstride` (floats; 0 = groups back to back): distance between the moment blocks of consecutive groups, for groups
// that arrive inside a wider all-gathered record.
__device__ __forceinline__ void rn_wave_batch_moments(const float* __restrict__ ws, int nblocks, int bpg, int rpg,
                                                      int ws_ld, int c, int lane, float& b_mean, float& b_M2,
                                                      long long gstride = 0) {
  float n_acc = 0.f, m_acc = 0.f, M2 = 0.f;
  const long long gs = gstride > 0 ? gstride : (long long)bpg * 2 * ws_ld;
  for (int b = lane; b < nblocks; b += 64) {
    const int g = b / bpg, bb = b - g * bpg;
    const float nb = (float)min(RN_ROWS_PER_BLOCK, rpg - bb * RN_ROWS_PER_BLOCK);
    const float* w = ws + g * gs + (long long)bb * 2 * ws_ld;
    chan_combine(n_acc, m_acc, M2, nb, w[c], w[ws_ld + c]);
  }
  for (int o = 32; o > 0; o >>= 1) {
    const float nb = __shfl_xor(n_acc, o, 64), mb = __shfl_xor(m_acc, o, 64), qb = __shfl_xor(M2, o, 64);
    // both partners must compute the identical combination: order the pair by lane id
    if ((lane & o) == 0) chan_combine(n_acc, m_acc, M2, nb, mb, qb);
    else { float n2 = nb, m2 = mb, q2 = qb; chan_combine(n2, m2, q2, n_acc, m_acc, M2); n_acc = n2; m_acc = m2; M2 = q2; }
  }
  b_mean = __shfl(m_acc, 0, 64);
  b_M2 = __shfl(M2, 0, 64);
}

// util/networks.py:123-134, same operation order: running (mean, var) with `cnt` samples absorb a batch of
// R samples with moments (b_mean, b_var).
__device__ __forceinline__ void rn_absorb(float& mean, float& var, int cnt, int R, float b_mean, float b_var) {
  const float fcount = (float)cnt, fn = (float)R;
  const float tot = (float)((long long)cnt + R);  // == (float)(cnt + R) below 2^31; no wrap above
  const float delta = b_mean - mean;
  mean = mean + delta * fn / tot;
  float rv = var * fcount;
  rv = rv + b_var * fn;
  rv = rv + delta * delta * fcount * fn / tot;
  var = rv / tot;
}

// The sample count is kept int32 like the reference's buffer (util/networks.py:72) but SATURATES at
// INT32_MAX instead of wrapping: this framework reaches 2^31 samples within a minute of config-P
// training, the reference never does (its arithmetic below that point is unchanged).
__device__ __forceinline__ int rn_count_add(int cnt, long long add) {
  const long long t = (long long)cnt + add;
  return t > 2147483647LL ? 2147483647 : (int)t;
}
