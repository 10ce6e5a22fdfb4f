"""Cost of the data-parallel code path WITHOUT the collectives themselves (one GPU): a stand-in
DataParallel of world 2 whose all-reduce is the identity and whose all-gather repeats the local
buffer. What remains is the per-step kernel and host work the synchronous schedule adds over the
single-GPU schedule (per-minibatch launches, no stream overlap). Usage: python tools/dp_overhead.py"""
import os
import sys
import time

import torch as th

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


class NullDP:
    rank = 0

    def __init__(self, world):
        self.world = world

    def allreduce_mean_(self, flat):
        return flat

    def broadcast_(self, tensors, src=0):
        pass

    def all_gather_flat(self, local):
        return th.cat([local] * self.world)

    def shared_seed(self):
        return 1234


th.set_num_threads(1)
cfg = dict(bench.CFG_P)
per_round = cfg["n_envs"] * cfg["n_steps"]
only = sys.argv[1] if len(sys.argv) > 1 else None   # substring of the case name, e.g. "dp 8"
for name, dp, glob in (("single", None, True), ("dp 2, per-minibatch all-reduce path", NullDP(2), False),
                       ("dp 2, global-minibatch update", NullDP(2), True),
                       ("dp 8, global-minibatch update", NullDP(8), True)):
    if only is not None and only not in name:
        continue
    tr = bench.build_trainer(bench.hip_namespace(), cfg, "cuda", seed=0, dp=dp)
    tr.gen_algo.dp_global_minibatch = glob
    tr.train(3 * per_round)
    th.cuda.synchronize()
    t = time.perf_counter()
    tr.train(10 * per_round)
    th.cuda.synchronize()
    dt = (time.perf_counter() - t) / 10
    print(f"{name:38s} {1e3 * dt:7.2f} ms/round  {per_round / dt / 1e3:8.1f} k env-steps/s per GPU")
