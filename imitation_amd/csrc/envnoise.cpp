// Host-only helper of the synthetic benchmark environment (imitation_amd/vec_env.py): fills the NEXT
// step's standard-normal draws of a numpy.random.Generator on a helper thread, through NumPy's own
// `random_standard_normal_fill` (linked from numpy/random/lib/libnpyrandom.a, the routine
// `Generator.standard_normal` calls), so stream and values are exactly those of drawing inside the
// step. Not part of the libimitation_hip C ABI: no device work, nothing of the reference's path.
// One worker (thread) per environment, created / destroyed with it.
#include <pthread.h>
#include <sched.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <fstream>
#include <mutex>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include <numpy/random/bitgen.h>

extern "C" void random_standard_normal_fill(bitgen_t*, intptr_t, double*);

namespace {

struct Worker {
  std::atomic<int> state{0};  // 0 idle, 1 job posted, 2 job done, 3 exit requested
  bitgen_t* bg = nullptr;
  // a job = n_steps consecutive env steps: step j draws a[j*stride .. +n_a) and then b[j*stride .. +n_b[j])
  int n_steps = 0;
  int64_t n_a = 0, stride = 0;
  double a_scale = 1.0, b_scale = 1.0;  // the draws are stored already multiplied by these (one IEEE multiply each)
  const int64_t* n_b = nullptr;
  double *a = nullptr, *b = nullptr;
  std::atomic<int> steps_done{0};  // of the current job
  std::mutex m;
  std::condition_variable cv;
  std::thread th;
  int64_t fill_ns = 0, fills = 0;  // diagnostics (ia_env_noise_stats)
  int cpu = -1;
  int main_cpu = -1;  // CPU of the posting thread the helper was last placed next to

  void serve() {
    for (;;) {
      // Jobs arrive every ~150 us while a rollout runs and not at all during the ~4 ms generator update
      // between rollouts. Stay hot across both: a sleeping thread costs a futex wake plus a C-state exit
      // (50-100 us) PER STEP, which makes steps longer, which makes the thread fall asleep again -- spin
      // for 20 ms (by the clock, `pause` lengths differ between CPUs), then sleep until the next post.
      int s;
      unsigned spins = 0;
      auto t0 = std::chrono::steady_clock::now();
      bool hot = true;
      while ((s = state.load(std::memory_order_acquire)) != 1 && s != 3) {
        if (hot) {
          __builtin_ia32_pause();
          if ((++spins & 1023u) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(20)) hot = false;
        } else {
          std::unique_lock<std::mutex> lk(m);
          cv.wait(lk, [&] { const int v = state.load(std::memory_order_acquire); return v == 1 || v == 3; });
        }
      }
      if (s == 3) return;
      const auto f0 = std::chrono::steady_clock::now();
      for (int j = 0; j < n_steps; ++j) {
        double* aj = a + j * stride;
        double* bj = b + j * stride;
        random_standard_normal_fill(bg, (intptr_t)n_a, aj);                      // process noise first ...
        if (n_b[j] > 0) random_standard_normal_fill(bg, (intptr_t)n_b[j], bj);  // ... then the reset observations
        for (int64_t i = 0; i < n_a; ++i) aj[i] = a_scale * aj[i];
        for (int64_t i = 0; i < n_b[j]; ++i) bj[i] = b_scale * bj[i];
        steps_done.store(j + 1, std::memory_order_release);
      }
      fill_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - f0).count();
      ++fills;
      cpu = sched_getcpu();
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

// "80-87,208-215" -> cpu ids
std::vector<int> parse_cpu_list(const std::string& path) {
  std::vector<int> out;
  std::ifstream f(path);
  std::string text;
  if (!f || !std::getline(f, text)) return out;
  std::stringstream ss(text);
  std::string part;
  while (std::getline(ss, part, ',')) {
    int lo = 0, hi = 0;
    if (std::sscanf(part.c_str(), "%d-%d", &lo, &hi) == 2) { for (int c = lo; c <= hi; ++c) out.push_back(c); }
    else if (std::sscanf(part.c_str(), "%d", &lo) == 1) out.push_back(lo);
  }
  return out;
}

// Keep the helper on a core that shares the L3 slice with the thread that posts the jobs (and is not its
// SMT sibling): the generator state and the 139 KB draw buffer bounce between the two threads every step,
// and from another socket the same fill takes 2x as long. Re-evaluated whenever the poster has moved.
void place_near(Worker* w, int cpu) {
  w->main_cpu = cpu;
  const std::string base = "/sys/devices/system/cpu/cpu" + std::to_string(cpu);
  const std::vector<int> l3 = parse_cpu_list(base + "/cache/index3/shared_cpu_list");
  const std::vector<int> sib = parse_cpu_list(base + "/topology/thread_siblings_list");
  cpu_set_t allowed;
  CPU_ZERO(&allowed);
  if (sched_getaffinity(0, sizeof(allowed), &allowed) != 0) return;
  const size_t n = l3.size();
  size_t at = 0;
  for (size_t i = 0; i < n; ++i) if (l3[i] == cpu) at = i;
  for (size_t k = 1; k < n; ++k) {
    const int c = l3[(at + k) % n];
    bool is_sib = false;
    for (int x : sib) is_sib = is_sib || x == c;
    if (is_sib || !CPU_ISSET(c, &allowed)) continue;
    cpu_set_t one;
    CPU_ZERO(&one);
    CPU_SET(c, &one);
    if (pthread_setaffinity_np(w->th.native_handle(), sizeof(one), &one) == 0) return;
  }
}

}  // namespace

extern "C" void* ia_env_noise_create(void) {
  Worker* w = new Worker;
  w->th = std::thread([w] { w->serve(); });
  return w;
}

// Post one job of n_steps env steps: step j fills a[j*stride .. +n_a) and then b[j*stride .. +n_b[j]) from
// generator `bitgen` (address of its bitgen_t), in that order, and stores them multiplied by a_scale / b_scale. One job at a time; the caller must not touch
// the generator, the buffers or n_b until ia_env_noise_finish() has returned.
extern "C" int ia_env_noise_post(void* handle, void* bitgen, int n_steps, int64_t n_a, int64_t stride, double* a,
                                 const int64_t* n_b, double* b, double a_scale, double b_scale) {
  Worker* w = static_cast<Worker*>(handle);
  if (w == nullptr || bitgen == nullptr || n_steps <= 0 || n_a < 0 || stride < n_a || a == nullptr ||
      n_b == nullptr || b == nullptr)
    return 1;
  if (w->state.load(std::memory_order_acquire) != 0) return 2;  // previous job not collected
  const int cpu = sched_getcpu();
  if (cpu >= 0 && cpu != w->main_cpu) place_near(w, cpu);
  w->bg = static_cast<bitgen_t*>(bitgen);
  w->n_steps = n_steps; w->n_a = n_a; w->stride = stride; w->a = a; w->n_b = n_b; w->b = b;
  w->a_scale = a_scale; w->b_scale = b_scale;
  w->steps_done.store(0, std::memory_order_release);
  w->signal(1);
  return 0;
}

// Blocks (spinning: the fill is normally finished already) until step j of the posted job has been drawn.
extern "C" int ia_env_noise_wait_step(void* handle, int j) {
  Worker* w = static_cast<Worker*>(handle);
  if (w == nullptr || w->state.load(std::memory_order_acquire) == 0 || j < 0 || j >= w->n_steps) return 1;
  while (w->steps_done.load(std::memory_order_acquire) <= j) __builtin_ia32_pause();
  return 0;
}

// Blocks until the whole posted job is done and marks the worker idle again.
extern "C" int ia_env_noise_finish(void* handle) {
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
  if (w->state.load(std::memory_order_acquire) != 0) ia_env_noise_finish(handle);
  w->signal(3);
  w->th.join();
  delete w;
}

// Diagnostics: out[0] = total fill time (ns), out[1] = fills, out[2] = CPU the helper last ran on,
// out[3] = CPU of the calling thread.
extern "C" void ia_env_noise_stats(void* handle, int64_t* out) {
  Worker* w = static_cast<Worker*>(handle);
  out[0] = w->fill_ns; out[1] = w->fills; out[2] = w->cpu; out[3] = sched_getcpu();
}
