"""Serial per-phase wall-clock costs of one bench.py variant's round (each phase followed by a device sync), next to
the overlapped round. Usage: python tools/variant_phases.py [variant] [rounds]"""
import os
import sys
import time

import torch as th

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from imitation_amd import networks  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "3_airl_ant_1024x16"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
th.set_num_threads(1)
tr, per_round = bench.build_variant(name)
tr.train(3 * per_round)
th.cuda.synchronize()
algo = tr.gen_algo
acc = {}


def timed(key, fn, *a, **k):
    th.cuda.synchronize()
    t = time.perf_counter()
    r = fn(*a, **k)
    host = time.perf_counter() - t
    th.cuda.synchronize()
    acc[key] = acc.get(key, 0.0) + time.perf_counter() - t
    acc[key + " (host)"] = acc.get(key + " (host)", 0.0) + host
    return r


cb = algo._init_callback(tr.gen_callback)
for _ in range(rounds):
    timed("collect_rollouts", algo.collect_rollouts, algo.env, cb, algo.rollout_buffer, algo.n_steps)
    timed("ppo_train", algo.train)
    gs, lens = timed("pop_transitions", tr.venv_buffering.pop_transitions_and_lens)
    timed("replay_store", tr._gen_replay_buffer.store, gs)
    for _ in range(tr.n_disc_updates_per_round):
        with networks.training(tr.reward_train):
            timed("train_disc", tr.train_disc)
print(name, "serial phase costs, ms/round (host = time until the call returned):")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print(f"  {k:28s} {1e3 * v / rounds:8.3f}")
t0 = time.perf_counter()
tr.train(rounds * per_round)
th.cuda.synchronize()
print(f"overlapped round: {1e3 * (time.perf_counter() - t0) / rounds:.2f} ms/round")
