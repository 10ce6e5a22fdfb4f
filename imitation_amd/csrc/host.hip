// Host-side helpers of the C ABI (no device code): index draws that must reproduce NumPy's
// legacy global generator bit for bit, runnable off the Python thread (ctypes drops the GIL).
#include <chrono>
#include <cstdint>
#include <thread>
#include <vector>
#include <immintrin.h>

#include "common.h"

namespace {

constexpr int MT_N = 624, MT_M = 397;

inline void mt_refill(uint32_t* key) {
  constexpr uint32_t A = 0x9908b0dfu, UP = 0x80000000u, LO = 0x7fffffffu;
  int i = 0;
  for (; i < MT_N - MT_M; ++i) {
    const uint32_t y = (key[i] & UP) | (key[i + 1] & LO);
    key[i] = key[i + MT_M] ^ (y >> 1) ^ (-(int32_t)(y & 1) & A);
  }
  for (; i < MT_N - 1; ++i) {
    const uint32_t y = (key[i] & UP) | (key[i + 1] & LO);
    key[i] = key[i + (MT_M - MT_N)] ^ (y >> 1) ^ (-(int32_t)(y & 1) & A);
  }
  const uint32_t y = (key[MT_N - 1] & UP) | (key[0] & LO);
  key[MT_N - 1] = key[MT_M - 1] ^ (y >> 1) ^ (-(int32_t)(y & 1) & A);
}

inline uint32_t mt_next32(uint32_t* key, int& pos) {
  if (pos == MT_N) {
    mt_refill(key);
    pos = 0;
  }
  uint32_t y = key[pos++];
  y ^= y >> 11;
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= y >> 18;
  return y;
}

// NumPy legacy `random_interval`: masked rejection on 32-bit draws (max < 2^32 here).
inline uint64_t legacy_interval(uint32_t* key, int& pos, uint64_t max) {
  if (max == 0) return 0;
  uint64_t mask = max;
  mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16; mask |= mask >> 32;
  uint64_t v;
  while ((v = (mt_next32(key, pos) & mask)) > max) {}
  return v;
}

// MT19937 `init_by_array` (Matsumoto & Nishimura's published seeding routine; what
// `np.random.RandomState(seed_array)` runs on the uint32 words of the array). Leaves pos = 624.
inline void mt_init_by_array(const uint32_t* init, int len, uint32_t* mt) {
  mt[0] = 19650218u;
  for (int i = 1; i < MT_N; ++i) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
  int i = 1, j = 0;
  for (int k = MT_N > len ? MT_N : len; k; --k) {
    mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1664525u)) + init[j] + (uint32_t)j;
    if (++i >= MT_N) { mt[0] = mt[MT_N - 1]; i = 1; }
    if (++j >= len) j = 0;
  }
  for (int k = MT_N - 1; k; --k) {
    mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1566083941u)) - (uint32_t)i;
    if (++i >= MT_N) { mt[0] = mt[MT_N - 1]; i = 1; }
  }
  mt[0] = 0x80000000u;
}

inline void legacy_permutation(uint32_t* key, int& p, int64_t n, int64_t* x) {
  for (int64_t i = 0; i < n; ++i) x[i] = i;
  for (int64_t i = n - 1; i >= 1; --i) {
    const int64_t j = (int64_t)legacy_interval(key, p, (uint64_t)i);
    const int64_t t = x[i];
    x[i] = x[j];
    x[j] = t;
  }
}

}  // namespace

// `count` independent permutations, out[c] = np.random.RandomState(seeds[c, :seed_len]).permutation(n),
// each drawn by its own host thread (the data-parallel PPO update: every rank derives the same
// per-epoch minibatch order from a shared seed while its rollout runs).
extern "C" int ia_host_mt19937_seeded_permutations(const uint32_t* seeds, int seed_len, int64_t n, int count,
                                                   int64_t* out) {
  if (seeds == nullptr || out == nullptr || seed_len <= 0 || n < 0 || count < 0 || n > 0xffffffffLL) return IA_ERR_ARG;
  auto one = [=](int c) {
    uint32_t key[MT_N];
    mt_init_by_array(seeds + (int64_t)c * seed_len, seed_len, key);
    int pos = MT_N;
    legacy_permutation(key, pos, n, out + (int64_t)c * n);
  };
  std::vector<std::thread> workers;
  workers.reserve(count > 1 ? count - 1 : 0);
  for (int c = 1; c < count; ++c) workers.emplace_back(one, c);
  if (count > 0) one(0);
  for (auto& t : workers) t.join();
  return IA_OK;
}

// `count` consecutive `np.random.permutation(n)` draws of the legacy MT19937 RandomState whose
// state is (key[624], *pos): out[c*n .. c*n+n) = arange(n) shuffled by the legacy Fisher-Yates
// loop (`for i in n-1..1: j = random_interval(i); swap(x[i], x[j])`). key / pos are advanced in
// place exactly as NumPy would have ([SB3 RolloutBuffer.get] draws one per PPO epoch).
extern "C" int ia_host_mt19937_permutations(uint32_t* key, int* pos, int64_t n, int count, int64_t* out) {
  if (key == nullptr || pos == nullptr || out == nullptr || n < 0 || count < 0 || *pos < 0 || *pos > MT_N ||
      n > 0xffffffffLL)
    return IA_ERR_ARG;
  int p = *pos;
  for (int c = 0; c < count; ++c) legacy_permutation(key, p, n, out + (int64_t)c * n);
  *pos = p;
  return IA_OK;
}

// Host half of the rollout mailbox (policy_rollout_mailbox_kernel): spin until every one of the `n` flags the device
// writes into pinned host memory has reached `target`. 0 = reached, 1 = timed out, -1 = a flag went negative (the
// kernel gave up: aborted or its own time-out). Called through ctypes, i.e. without the GIL.
extern "C" int ia_host_wait_i32(const volatile int32_t* flags, int n, int target, double timeout_s) {
  if (!flags || n <= 0) return IA_ERR_ARG;
  const auto t0 = std::chrono::steady_clock::now();
  unsigned spins = 0;
  for (;;) {
    bool ok = true;
    for (int i = 0; i < n; ++i) {
      const int32_t v = flags[i];
      if (v < 0) return -1;
      if (v < target) { ok = false; break; }
    }
    if (ok) return 0;
    _mm_pause();
    if ((++spins & 4095u) == 0 &&
        std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s)
      return 1;
  }
}
