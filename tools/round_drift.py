"""Per-round wall time over a long run at config P (windows of 20 rounds): does the round time drift?"""
import os
import sys
import time

import numpy as np
import torch as th

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

th.set_num_threads(1)
cfg = dict(bench.CFG_P)
tr = bench.build_trainer(bench.hip_namespace(), cfg, "cuda")
per = cfg["n_envs"] * cfg["n_steps"]
tr.train(3 * per)
th.cuda.synchronize()
n_win, win = int(sys.argv[1]) if len(sys.argv) > 1 else 10, 20
if len(sys.argv) > 2 and sys.argv[2] == "nogc":
    import gc
    gc.collect()
    gc.freeze()
    gc.disable()
for w in range(n_win):
    t0 = time.perf_counter()
    tr.train(win * per)
    th.cuda.synchronize()
    print(f"rounds {w * win:4d}-{(w + 1) * win - 1:4d}: {1e3 * (time.perf_counter() - t0) / win:.3f} ms/round")
