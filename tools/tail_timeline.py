"""Host timeline of a round's TAIL at config P: from the return of the rollout's last environment step to the return of the
`ia_ppo_update` call (median over the rounds, us). Usage: python tools/tail_timeline.py [rounds]"""
import os
import statistics
import sys
import time

import torch as th

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from imitation_amd import _lib as L, ppo, reward_nets  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
th.set_num_threads(1)
cfg = dict(bench.CFG_P)
tr = bench.build_trainer(bench.hip_namespace(), cfg, "cuda")
per = cfg["n_envs"] * cfg["n_steps"]
tr.train(4 * per)
th.cuda.synchronize()
marks, cur = [], {}
tick = time.perf_counter


def wrap(obj, name, label, before=False):
    f = getattr(obj, name)

    def g(*a, **k):
        if before:
            cur[label + " entered"] = tick()
        r = f(*a, **k)
        cur[label + " returned"] = tick()
        return r
    setattr(obj, name, g)


orig_step = ppo.step_arrays


def step_arrays(base):
    r = orig_step(base)
    cur["last env step returned"] = tick()
    return r


ppo.step_arrays = step_arrays
wrap(ppo.RolloutBuffer, "upload_host_tiles", "upload_host_tiles")
wrap(type(tr.reward_train), "predict_processed_rollout", "relabel (predict_processed_rollout)", before=True)
wrap(ppo.PPO, "collect_rollouts", "collect_rollouts")
wrap(ppo.PPO, "train", "PPO.train", before=True)
lib = L.load()
f_upd = lib.ia_ppo_update


def upd(*a):
    cur["ia_ppo_update entered"] = tick()
    r = f_upd(*a)
    cur["ia_ppo_update returned"] = tick()
    marks.append(dict(cur))
    return r


lib.ia_ppo_update = upd
tr.train(rounds * per)
th.cuda.synchronize()
keys = ["upload_host_tiles returned", "relabel (predict_processed_rollout) entered", "relabel (predict_processed_rollout) returned",
        "collect_rollouts returned", "PPO.train entered", "ia_ppo_update entered", "ia_ppo_update returned"]
print(f"host, us after the last environment step returned (median of {len(marks)} rounds):")
for k in keys:
    if not all(k in m for m in marks):   # (the one-call tail has no separate relabelling call)
        continue
    print(f"  {k:52s} {1e6 * statistics.median(m[k] - m['last env step returned'] for m in marks):8.1f}")
