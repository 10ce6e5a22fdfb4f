"""Per-phase wall-clock breakdown of the HIP GAIL round at config P (host time with device syncs
at phase boundaries) + cProfile of the host code. Usage: python tools/phase_profile.py [rounds]"""
import cProfile
import os
import pstats
import sys
import time

import torch as th

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
cfg = dict(bench.CFG_P)
tr = bench.build_trainer(bench.hip_namespace(), cfg, "cuda")
per_round = cfg["n_envs"] * cfg["n_steps"]
tr.train(2 * per_round)
th.cuda.synchronize()

algo = tr.gen_algo
acc = {}


def timed(name, fn):
    def wrapper(*a, **k):
        th.cuda.synchronize()
        t = time.perf_counter()
        r = fn(*a, **k)
        th.cuda.synchronize()
        acc[name] = acc.get(name, 0.0) + time.perf_counter() - t
        return r
    return wrapper


algo.collect_rollouts = timed("collect_rollouts", algo.collect_rollouts)
algo.train = timed("ppo_train", algo.train)
tr.train_disc = timed("train_disc", tr.train_disc)
tr.venv_buffering.pop_transitions_and_lens = timed("pop_transitions", tr.venv_buffering.pop_transitions_and_lens)
tr._gen_replay_buffer.store = timed("replay_store", tr._gen_replay_buffer.store)
t0 = time.perf_counter()
tr.train(rounds * per_round)
th.cuda.synchronize()
total = time.perf_counter() - t0
print(f"total {1e3 * total / rounds:.2f} ms/round")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print(f"  {k:20s} {1e3 * v / rounds:8.2f} ms/round")
pr = cProfile.Profile()
pr.enable()
tr.train(2 * per_round)
th.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
