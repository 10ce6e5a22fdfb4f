"""Per-step phase timing of the persistent PPO update on an all-gathered (data-parallel sized) tile:
world copies of the config-P rollout through a stand-in DataParallel. Usage: python tools/ppo_update_timing_dp.py [world]"""
import os
import sys
import time

import torch as th

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from imitation_amd import _lib as L  # noqa: E402


from tools.dp_overhead import NullDP  # noqa: E402  (stand-in DataParallel: identity all-reduce, repeating all-gather)


world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
th.set_num_threads(1)
cfg = dict(bench.CFG_P)
tr = bench.build_trainer(bench.hip_namespace(), cfg, "cuda", seed=0, dp=NullDP(world))
per_round = cfg["n_envs"] * cfg["n_steps"]
tr.train(2 * per_round)
th.cuda.synchronize()
algo = tr.gen_algo
algo.defer_train_stats = True
buf = th.zeros(32, dtype=th.int64, device="cuda")
L.load().ia_ppo_debug_timing(buf.data_ptr())
names = ("wait statistics", "minibatch fwd/bwd", "grid barrier", "park+sync", "slab reduce (both levels)", "norm",
         "stats+Adam", "release fence", "atomic add", "prefetch issue", "spin", "acquire fence")
for rep in range(2):
    algo._dpg["perms"].start(algo._dpg["perm_np"])
    buf.zero_()
    th.cuda.synchronize()
    t = time.perf_counter()
    algo.train()
    h = time.perf_counter() - t
    th.cuda.synchronize()
    d = time.perf_counter() - t
    steps = algo.n_epochs * algo._n_mb
    ticks = buf.cpu().numpy()[:12]
    print(f"world {world}: train() host {1e3 * h:.2f} ms, device idle after {1e3 * d:.2f} ms, {steps} steps; per step: " +
          ", ".join(f"{n} {t_ / 100.0 / steps:.2f}" for n, t_ in zip(names, ticks)))
L.load().ia_ppo_debug_timing(None)
# production instantiation (no phase clocks): HIP events around the launch
algo.update_events = (th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True))
ms = []
for rep in range(6):
    algo._dpg["perms"].start(algo._dpg["perm_np"])
    algo.train()
    th.cuda.synchronize()
    ms.append(algo.update_events[0].elapsed_time(algo.update_events[1]))
steps = algo.n_epochs * algo._n_mb
print(f"world {world}: production ia_ppo_update on the gathered tile: median {sorted(ms)[len(ms) // 2]:.3f} ms = "
      f"{1e3 * sorted(ms)[len(ms) // 2] / steps:.2f} us/step")
