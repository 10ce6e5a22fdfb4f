// Host-only helper of the synthetic benchmark environment (imitation_amd/vec_env.py): fills the NEXT
// step's standard-normal draws of a numpy.random.Generator on a helper thread, through NumPy's own
// `random_standard_normal_fill` (linked from numpy/random/lib/libnpyrandom.a, the routine
// `Generator.standard_normal` calls), so stream and values are exactly those of drawing inside the
// step. Not part of the libimitation_hip C ABI: no device work, nothing of the reference's path.
// One worker (thread) per environment, created / destroyed with it.
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <mutex>
#include <thread>

#include <numpy/random/bitgen.h>

extern "C" void random_standard_normal_fill(bitgen_t*, intptr_t, double*);

namespace {

struct Worker {
  std::atomic<int> state{0};  // 0 idle, 1 job posted, 2 job done, 3 exit requested
  bitgen_t* bg = nullptr;
  int64_t n_a = 0, n_b = 0;
  double *a = nullptr, *b = nullptr;
  std::mutex m;
  std::condition_variable cv;
  std::thread th;

  void serve() {
    for (;;) {
      // jobs arrive every ~150 us while a rollout runs: spin about that long, then sleep until the next post
      int spins = 0, s;
      while ((s = state.load(std::memory_order_acquire)) != 1 && s != 3) {
        if (++spins < 20000) {
          __builtin_ia32_pause();
        } else {
          std::unique_lock<std::mutex> lk(m);
          cv.wait(lk, [&] { const int v = state.load(std::memory_order_acquire); return v == 1 || v == 3; });
        }
      }
      if (s == 3) return;
      random_standard_normal_fill(bg, (intptr_t)n_a, a);               // process noise first ...
      if (n_b > 0) random_standard_normal_fill(bg, (intptr_t)n_b, b);  // ... then the reset observations
      state.store(2, std::memory_order_release);
    }
  }

  void signal(int v) {
    {
      std::lock_guard<std::mutex> lk(m);
      state.store(v, std::memory_order_release);
    }
    cv.notify_one();
  }
};

}  // namespace

extern "C" void* ia_env_noise_create(void) {
  Worker* w = new Worker;
  w->th = std::thread([w] { w->serve(); });
  return w;
}

// Post one job: a[0..n_a) then b[0..n_b) from generator `bitgen` (address of its bitgen_t). One job at a
// time; the caller must not touch the generator or the buffers until ia_env_noise_wait() has returned.
extern "C" int ia_env_noise_post(void* handle, void* bitgen, int64_t n_a, double* a, int64_t n_b, double* b) {
  Worker* w = static_cast<Worker*>(handle);
  if (w == nullptr || bitgen == nullptr || n_a < 0 || n_b < 0 || (n_a > 0 && a == nullptr) ||
      (n_b > 0 && b == nullptr))
    return 1;
  if (w->state.load(std::memory_order_acquire) != 0) return 2;  // previous job not collected
  w->bg = static_cast<bitgen_t*>(bitgen);
  w->n_a = n_a; w->a = a; w->n_b = n_b; w->b = b;
  w->signal(1);
  return 0;
}

// Blocks (spinning: the fill is normally finished already) until the posted job is done.
extern "C" int ia_env_noise_wait(void* handle) {
  Worker* w = static_cast<Worker*>(handle);
  if (w == nullptr || w->state.load(std::memory_order_acquire) == 0) return 1;  // nothing posted
  while (w->state.load(std::memory_order_acquire) != 2) __builtin_ia32_pause();
  w->state.store(0, std::memory_order_release);
  return 0;
}

// Finishes a running job (its buffers must still be alive), stops the thread, frees the worker.
extern "C" void ia_env_noise_destroy(void* handle) {
  Worker* w = static_cast<Worker*>(handle);
  if (w == nullptr) return;
  if (w->state.load(std::memory_order_acquire) != 0) ia_env_noise_wait(handle);
  w->signal(3);
  w->th.join();
  delete w;
}
