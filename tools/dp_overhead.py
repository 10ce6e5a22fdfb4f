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
    world, rank = 2, 0

    def allreduce_mean_(self, flat):
        return flat

    def broadcast_(self, tensors, src=0):
        pass

    def all_gather_flat(self, local):
        return th.cat([local, local])


th.set_num_threads(1)
cfg = dict(bench.CFG_P)
per_round = cfg["n_envs"] * cfg["n_steps"]
for name, dp in (("single", None), ("dp-path, null collectives", NullDP())):
    tr = bench.build_trainer(bench.hip_namespace(), cfg, "cuda", seed=0, dp=dp)
    tr.train(3 * per_round)
    th.cuda.synchronize()
    t = time.perf_counter()
    tr.train(10 * per_round)
    th.cuda.synchronize()
    dt = (time.perf_counter() - t) / 10
    print(f"{name:28s} {1e3 * dt:7.2f} ms/round  {per_round / dt / 1e3:8.1f} k env-steps/s per GPU")
