"""Production-build time of the persistent PPO update (ia_ppo_update) at config P: HIP events around the launch,
median over rounds. Usage: python tools/ppo_step_us.py [xcd_pack 0|1] [rounds] [bench variant]"""
import os
import sys

import numpy as np
import torch as th

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from imitation_amd import _lib as L  # noqa: E402

pack = int(sys.argv[1]) if len(sys.argv) > 1 else 0
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 12
th.set_num_threads(1)
if len(sys.argv) > 3:
    tr, per_round = bench.build_variant(sys.argv[3])
else:
    cfg = dict(bench.CFG_P)
    tr = bench.build_trainer(bench.hip_namespace(), cfg, "cuda")
    per_round = cfg["n_envs"] * cfg["n_steps"]
L.load().ia_ppo_update_xcd_pack(pack)
tr.train(3 * per_round)
th.cuda.synchronize()
algo = tr.gen_algo
algo.update_events = (th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True))
ms = []
for _ in range(rounds):
    tr.train(per_round)
    th.cuda.synchronize()
    ms.append(algo.update_events[0].elapsed_time(algo.update_events[1]))
steps = algo.n_epochs * algo._n_mb
print((sys.argv[3] + " " if len(sys.argv) > 3 else "") + f"xcd_pack={pack}: ia_ppo_update median {np.median(ms):.3f} ms (min {min(ms):.3f}) for {steps} steps = "
      f"{1e3 * np.median(ms) / steps:.2f} us/step")
