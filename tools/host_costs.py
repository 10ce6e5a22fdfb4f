"""Host-side cost of the pieces of one round's enqueue work at config P (no device sync inside the
timed calls unless noted). Usage: python tools/host_costs.py"""
import os
import sys
import time

import numpy as np
import torch as th

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

th.set_num_threads(1)
t = time.perf_counter()
for _ in range(50):
    np.random.permutation(16384)
print(f"np.random.permutation(16384): {(time.perf_counter() - t) / 50 * 1e6:.1f} us")
cfg = dict(bench.CFG_P)
tr = bench.build_trainer(bench.hip_namespace(), cfg, "cuda")
per_round = cfg["n_envs"] * cfg["n_steps"]
tr.train(3 * per_round)
th.cuda.synchronize()
algo = tr.gen_algo
algo.defer_train_stats = True
for name, fn in (("PPO.train enqueue (10 epochs)", algo.train),):
    for rep in range(3):
        th.cuda.synchronize()
        t = time.perf_counter()
        fn()
        h = time.perf_counter() - t
        th.cuda.synchronize()
        d = time.perf_counter() - t
        print(f"{name}: host {1e3 * h:.2f} ms, until device idle {1e3 * d:.2f} ms")
from imitation_amd import networks  # noqa: E402
for rep in range(3):
    th.cuda.synchronize()
    t = time.perf_counter()
    tr._overlap_k = 0
    pend = tr._disc_round()
    h = time.perf_counter() - t
    th.cuda.synchronize()
    d = time.perf_counter() - t
    tr._finish_disc_round(pend)
    print(f"disc round enqueue (16 updates): host {1e3 * h:.2f} ms, until device idle {1e3 * d:.2f} ms")
