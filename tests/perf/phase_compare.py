"""Per-phase wall time of one GAIL round at config P, same phase boundaries on every side:
  collect rollouts (policy steps, env steps, reward relabelling, GAE) | pop + flatten + replay store |
  16 x train_disc (expert batch, generator batch, forward/backward/Adam, statistics) | PPO.train | log dump
for the HIP trainer (phases run back to back with a device sync after each, i.e. WITHOUT the overlap
of the real schedule), the CPU oracle (1 torch thread and all cores) and -- where /root/reference
exists (not on the GPU box) -- the reference's own modules under the import shim.
Usage: python tests/perf/phase_compare.py hip|oracle|reference [torch threads]"""
import os
import sys
import time

import numpy as np
import torch as th

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from tests import harness  # noqa: E402

impl = sys.argv[1] if len(sys.argv) > 1 else "hip"
threads = int(sys.argv[2]) if len(sys.argv) > 2 else (1 if impl == "hip" else th.get_num_threads())
th.set_num_threads(threads)
cfg = dict(bench.CFG_P)
ns = harness.namespace(impl)
ns.configure_logger = (lambda d, f=ns.configure_logger: f(d))
device = "cuda" if impl == "hip" else "cpu"
if impl == "reference":  # the reference's Transitions need an infos column
    tr = bench.build_trainer(ns, cfg, device)
else:
    tr = bench.build_trainer(bench.hip_namespace() if impl == "hip" else bench.oracle_namespace(), cfg, device)
per_round = cfg["n_envs"] * cfg["n_steps"]
acc = {}
sync = th.cuda.synchronize if impl == "hip" else (lambda: None)
depth = [0]


def wrap(obj, attr, name):
    if not hasattr(obj, attr):
        return False
    orig = getattr(obj, attr)

    def f(*a, **k):
        outer = depth[0] == 0
        depth[0] += 1
        if outer:
            sync()
            t = time.perf_counter()
        try:
            return orig(*a, **k)
        finally:
            depth[0] -= 1
            if outer:
                sync()
                acc[name] = acc.get(name, 0.0) + time.perf_counter() - t

    setattr(obj, attr, f)
    return True


algo = tr.gen_algo
if impl == "hip":  # strictly sequential schedule, one phase at a time
    tr.pipeline_rounds = False
    tr._overlap_beside_ppo = False
rounds_warm, rounds = (2, 5) if impl == "hip" else (0, 1)
if rounds_warm:
    tr.train(rounds_warm * per_round)
wrap(algo, "collect_rollouts", "collect rollouts (policy + env steps, relabel, GAE)")
wrap(algo, "train", "PPO.train (10 epochs x 16 minibatches)")
buf = tr.venv_buffering
for a in ("pop_transitions_and_lens", "pop_order_and_lens", "pop_trajectories", "pop_transitions"):
    wrap(buf, a, "pop + flatten trajectories")
rb = tr._gen_replay_buffer
for a in ("store", "store_from_rollout"):
    wrap(rb, a, "replay store")
if not wrap(tr, "_disc_round", "16 x train_disc (batches, fwd/bwd/Adam, stats)"):
    wrap(tr, "train_disc", "16 x train_disc (batches, fwd/bwd/Adam, stats)")
wrap(tr, "_finish_disc_round", "16 x train_disc (batches, fwd/bwd/Adam, stats)")
wrap(tr.logger, "dump", "logger.dump")
sync()
t0 = time.perf_counter()
tr.train(rounds * per_round)
sync()
total = (time.perf_counter() - t0) / rounds
label = {"hip": "HIP trainer, phases serialised", "oracle": "CPU oracle", "reference": "reference modules (import shim)"}[impl]
print(f"## {label}; torch threads = {threads}, os.cpu_count() = {os.cpu_count()}; {rounds} round(s)")
print(f"| phase | ms per round | share |\n|---|---|---|")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print(f"| {k} | {1e3 * v / rounds:.2f} | {100 * v / rounds / total:.1f} % |")
rest = total - sum(acc.values()) / rounds
print(f"| other (loop, callbacks, wrapper bookkeeping outside the phases) | {1e3 * rest:.2f} | {100 * rest / total:.1f} % |")
print(f"| **whole round** | **{1e3 * total:.2f}** | {per_round / total:,.0f} env-steps/s |")
