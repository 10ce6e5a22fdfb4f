"""Host-side sections of the rollout loop at config P (seconds accumulated over the 16 steps of each
rollout inside pipelined training). Usage: python tools/rollout_sections.py [rounds]"""
import os
import sys

import torch as th

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 10
th.set_num_threads(1)
cfg = dict(bench.CFG_P)
tr = bench.build_trainer(bench.hip_namespace(), cfg, "cuda")
per_round = cfg["n_envs"] * cfg["n_steps"]
tr.train(3 * per_round)
th.cuda.synchronize()
tr.gen_algo.rollout_profile = {}
tr.train(rounds * per_round)
th.cuda.synchronize()
prof = tr.gen_algo.rollout_profile
tot = sum(prof.values())
print(f"per rollout (16 steps), average over {rounds} rounds:")
for k, v in sorted(prof.items(), key=lambda kv: -kv[1]):
    print(f"  {k:28s} {1e3 * v / rounds:7.3f} ms   {1e6 * v / rounds / cfg['n_steps']:7.1f} us/step")
print(f"  {'sum':28s} {1e3 * tot / rounds:7.3f} ms")

# device time between the last env step and the start of the PPO update (copies, relabel, GAE, launch)
algo = tr.gen_algo
tails = []
for _ in range(5):
    algo.update_events = (th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True))
    tr.train(per_round)
    th.cuda.synchronize()
    tails.append((algo.tail_event.elapsed_time(algo.update_events[0]), algo.update_events[0].elapsed_time(algo.update_events[1])))
algo.update_events = None
print("  last env step -> PPO update starts (ms):", ", ".join(f"{a:.3f}" for a, _ in tails),
      "| PPO update (ms):", ", ".join(f"{b:.3f}" for _, b in tails))
env = tr.venv  # helper-thread diagnostics of the synthetic env (if the prefetch helper is active)
while not hasattr(env, "_worker") and hasattr(env, "venv"):
    env = env.venv
if getattr(env, "_worker", 0):
    import ctypes as C
    out = (C.c_int64 * 4)()
    env._helper.ia_env_noise_stats.argtypes = [C.c_void_p, C.c_void_p]
    env._helper.ia_env_noise_stats(env._worker, out)
    print(f"  env noise helper: {out[0] / max(out[1], 1) / 1e3:.1f} us per fill over {out[1]} fills, helper on cpu {out[2]}, "
          f"main thread on cpu {out[3]}")
