"""Where the host waits inside a pipelined round's `drain` (previous round's statistics): times `_TrainRecord.read_back`,
`finalize_train` and `_finish_disc_round` per round. Usage: python tools/drain_probe.py <variant | P> [rounds]"""
import os
import sys
import time

import numpy as np
import torch as th

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from imitation_amd import ppo as P  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "P"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 20
th.set_num_threads(1)
if name == "P":
    cfg = dict(bench.CFG_P)
    tr, per = bench.build_trainer(bench.hip_namespace(), cfg, "cuda"), cfg["n_envs"] * cfg["n_steps"]
else:
    tr, per = bench.build_variant(name)
tr.train(5 * per)
th.cuda.synchronize()
acc = {}


def timed(obj, attr, key):
    orig = getattr(obj, attr)

    def f(*a, **k):
        t = time.perf_counter()
        r = orig(*a, **k)
        acc.setdefault(key, []).append(1e3 * (time.perf_counter() - t))
        return r

    setattr(obj, attr, f)


orig_rb = P._TrainRecord.read_back


def rb(self, stream):
    t = time.perf_counter()
    q = self.ready.query()
    orig_rb(self, stream)
    acc.setdefault("read_back (ready event already complete: %s)" % q, []).append(1e3 * (time.perf_counter() - t))


P._TrainRecord.read_back = rb
timed(tr.gen_algo, "finalize_train", "finalize_train")
timed(tr, "_finish_disc_round", "_finish_disc_round")
timed(tr.gen_algo, "collect_rollouts", "collect_rollouts")
t = time.perf_counter()
tr.train(rounds * per)
th.cuda.synchronize()
print(f"{name}: {1e3 * (time.perf_counter() - t) / rounds:.3f} ms/round")
for k, v in acc.items():
    print(f"  {k:60s} n={len(v):3d} median {np.median(v):.3f} ms  max {max(v):.3f}")
