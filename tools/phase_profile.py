"""Per-phase wall-clock breakdown of the HIP GAIL round at config P. Phases are run back-to-back
WITHOUT overlap and with a device sync after each, so the numbers are the serial costs; the real
round overlaps the PPO update with the discriminator updates (GAIL).
Usage: python tools/phase_profile.py [rounds]"""
import os
import sys
import time

import torch as th

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from imitation_amd import networks  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
th.set_num_threads(1)
cfg = dict(bench.CFG_P)
tr = bench.build_trainer(bench.hip_namespace(), cfg, "cuda")
per_round = cfg["n_envs"] * cfg["n_steps"]
tr.train(2 * per_round)
th.cuda.synchronize()
algo = tr.gen_algo
acc = {}


def timed(name, fn, *a, **k):
    th.cuda.synchronize()
    t = time.perf_counter()
    r = fn(*a, **k)
    th.cuda.synchronize()
    acc[name] = acc.get(name, 0.0) + time.perf_counter() - t
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
print("serial phase costs, ms/round:")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print(f"  {k:20s} {1e3 * v / rounds:8.2f}")
t0 = time.perf_counter()
tr.train(rounds * per_round)
th.cuda.synchronize()
print(f"overlapped round: {1e3 * (time.perf_counter() - t0) / rounds:.2f} ms/round")
