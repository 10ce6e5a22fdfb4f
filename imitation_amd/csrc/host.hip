// Host-side helpers of the C ABI (no device code): index draws that must reproduce NumPy's
// legacy global generator bit for bit, runnable off the Python thread (ctypes drops the GIL).
#include <chrono>
#include <cstring>
#include <cstdint>
#include <thread>
#include <vector>
#include <immintrin.h>

#include "common.h"
#include "../../include/imitation_hip.h"

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

// The same, followed -- on a COPY of the generator -- by `rows` x `row_len` draws of `np.random.randint(high, size=row_len)`
// (legacy `RandomState.randint`, int64: `_rand_int64` -> `random_bounded_uint64_fill` with use_masked = 1, i.e. for
// high - 1 < 2^32 - 1 one masked rejection loop over 32-bit draws per element; high == 1: zeros, no draw). key / pos are
// left BEHIND THE PERMUTATIONS (what `PPO.train` installs as the global state when it adopts them); key_post / pos_post
// receive the state behind the randint rows (what the discriminator round installs when it adopts those: the replay
// ring's sixteen `sample_indices` calls of a round, `algorithms/adversarial/common.py:557-575` through
// `data/buffer.py:366-377`). One call, no GIL in between: the helper thread never competes with the rollout loop.
extern "C" int ia_host_mt19937_permutations_then_randint(uint32_t* key, int* pos, int64_t n, int count, int64_t* out,
                                                         int64_t high, int64_t rows, int64_t row_len, int64_t* out_rows,
                                                         uint32_t* key_post, int* pos_post) {
  const int rc = ia_host_mt19937_permutations(key, pos, n, count, out);
  if (rc != IA_OK) return rc;
  if (rows <= 0) return IA_OK;
  if (out_rows == nullptr || key_post == nullptr || pos_post == nullptr || high < 1 || high > 0x100000000LL || row_len < 0)
    return IA_ERR_ARG;
  for (int i = 0; i < MT_N; ++i) key_post[i] = key[i];
  int p = *pos;
  const uint64_t rng = (uint64_t)(high - 1);
  for (int64_t i = 0; i < rows * row_len; ++i)
    out_rows[i] = rng == 0xffffffffULL ? (int64_t)mt_next32(key_post, p) : (int64_t)legacy_interval(key_post, p, rng);
  *pos_post = p;
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

// ---------------------------------------------------------------------------------------------------------------------
// Peer-mapped device memory for the row-sharded data-parallel PPO update (ia_ppo_update_sharded): each rank (one process
// per GPU) owns one block -- receive area + flags + handshake words -- that every other rank writes into directly from
// inside its persistent kernel (xGMI between GPUs; the same protocol between two processes on one GPU, which is how it is
// tested on one-GPU boxes). The block is fine-grained (coherent at system scope, never cached in a remote L2) when the
// runtime grants it, plain device memory otherwise; it is shared through hipIpc handles the ranks exchange over
// torch.distributed. These are the ONLY entries of the library that allocate / map / free device memory or synchronise.
extern "C" int ia_peer_alloc(size_t bytes, void** out, int* fine_grained) {
  if (!out || bytes == 0) return IA_ERR_ARG;
  void* p = nullptr;
  int fine = 1;
  hipError_t e = hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained);
  if (e != hipSuccess || p == nullptr) {
    (void)hipGetLastError();
    fine = 0;
    e = hipMalloc(&p, bytes);
    if (e != hipSuccess) return (int)e;
  }
  e = hipMemset(p, 0, bytes);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e != hipSuccess) {
    (void)hipFree(p);
    return (int)e;
  }
  *out = p;
  if (fine_grained) *fine_grained = fine;
  return IA_OK;
}
extern "C" int ia_peer_free(void* p) { return p ? (int)hipFree(p) : IA_OK; }
extern "C" int ia_peer_ipc_export(void* p, unsigned char* handle64) {
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t is 64 bytes");
  if (!p || !handle64) return IA_ERR_ARG;
  hipIpcMemHandle_t h;
  const hipError_t e = hipIpcGetMemHandle(&h, p);
  if (e != hipSuccess) return (int)e;
  memcpy(handle64, &h, 64);
  return IA_OK;
}
extern "C" int ia_peer_ipc_open(const unsigned char* handle64, void** out) {
  if (!handle64 || !out) return IA_ERR_ARG;
  hipIpcMemHandle_t h;
  memcpy(&h, handle64, 64);
  void* p = nullptr;
  const hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
  if (e != hipSuccess) return (int)e;
  *out = p;
  return IA_OK;
}
extern "C" int ia_peer_ipc_close(void* p) { return p ? (int)hipIpcCloseMemHandle(p) : IA_OK; }

namespace {
struct HsPeers { uint32_t* p[8]; };
// One wave: write `token` into word [rank] of every rank's handshake area, then wait until all `world` words of the own
// area carry it. result: 1 = every peer's write arrived (mapping, peer writes and system-scope polling all work), 0 = not.
__global__ void peer_handshake_kernel(int world, int rank, uint32_t token, uint32_t* own, HsPeers peers,
                                      long long timeout_ticks, int* result) {
  const int lane = threadIdx.x;
  if (lane == 0)
    for (int r = 0; r < world; ++r) __hip_atomic_store(peers.p[r] + rank, token, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  const long long t0 = wall_clock64();
  int ok = 0;
  for (;;) {
    const uint32_t v = __hip_atomic_load(own + (lane < world ? lane : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (__all(lane >= world || v == token)) { ok = 1; break; }
    __builtin_amdgcn_s_sleep(8);
    if (wall_clock64() - t0 > timeout_ticks) break;
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
  if (lane == 0) __hip_atomic_store(result, ok, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
}  // namespace

// Enqueues the handshake on `stream`; `result` is one int the device can write (device or pinned host memory).
extern "C" int ia_peer_handshake(int world, int rank, uint32_t token, uint32_t* own_words, uint32_t* const* peer_words,
                                 double timeout_s, int* result, void* stream) {
  if (world < 1 || world > 8 || rank < 0 || rank >= world || !own_words || !peer_words || !result || token == 0)
    return IA_ERR_ARG;
  HsPeers pp;
  for (int r = 0; r < 8; ++r) {
    pp.p[r] = r < world ? peer_words[r] : nullptr;
    if (r < world && !pp.p[r]) return IA_ERR_ARG;
  }
  hipLaunchKernelGGL(peer_handshake_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, world, rank, token, own_words, pp,
                     (long long)(timeout_s * 1e8), result);
  IA_CHECK_LAUNCH();
  return IA_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// The tail of a rollout in ONE host call: relabelling of the whole [T, n] tile by the discriminator-based reward
// (`rewards/reward_wrapper.py:110-115` per step == once on the tile, since nothing on this path changes between steps:
// assembly `ia_gather_concat`, prediction `ia_disc_fused_predict`), the rewards' copy to the pinned host tile (episode-return
// bookkeeping, `rewards/reward_wrapper.py:117-133`), and [SB3 RolloutBuffer.compute_returns_and_advantage] (`ia_gae`) -- the
// launches of those four calls in their order, nothing else: the Python between them (argument marshalling of ~60 scalars,
// table views, context managers: ~75 us) sat on the critical path between the last environment step and the PPO launch.
extern "C" int ia_rollout_tail(const ia_rollout_tail_args* a, void* stream) {
  if (!a || !a->desc || !a->X || !a->rewards || a->T <= 0 || a->n <= 0) return IA_ERR_ARG;
  const int rows = a->T * a->n;
  int rc = ia_gather_concat(a->obs, a->act_f32, a->act_i64, a->next_obs, a->dones, nullptr, rows, a->obs_dim, a->act_dim,
                            a->use_state, a->use_action, a->use_next_state, a->use_done, a->X, a->ldx, 0, stream);
  if (rc) return rc;
  rc = ia_disc_fused_predict(a->desc, a->params, a->X, a->ldx, rows, a->norm_mean, a->norm_var, a->norm_eps, a->out_act,
                             a->predict_ws, a->rewards, stream);
  if (rc) return rc;
  if (a->rewards_host) {
    hipError_t e = hipMemcpyAsync(a->rewards_host, a->rewards, sizeof(float) * (size_t)rows, hipMemcpyDeviceToHost,
                                  (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
  }
  return ia_gae(a->rewards, a->values, a->episode_starts, a->last_values, a->last_dones, a->T, a->n, a->gamma,
                a->gae_lambda, a->advantages, a->returns, stream);
}
