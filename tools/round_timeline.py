"""Host and device timeline of overlapped GAIL rounds at config P: when the host finishes each
enqueue step, and when each stream finishes its work (HIP events), relative to the round start.
Usage: python tools/round_timeline.py [rounds] [world] [variant]   (variant: a bench.py VARIANTS name instead of P)"""
import os
import sys
import time

import torch as th

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
th.set_num_threads(1)
cfg = dict(bench.CFG_P)
world = int(sys.argv[2]) if len(sys.argv) > 2 else 1   # > 1: data-parallel code path with stand-in collectives
dp = None
if world > 1:
    class NullDP:  # all-reduce = identity, all-gather = the local buffer repeated (see tools/dp_overhead.py)
        rank = 0

        def __init__(self, w):
            self.world = w

        def allreduce_mean_(self, flat):
            return flat

        def broadcast_(self, tensors, src=0):
            pass

        def all_gather_flat(self, local):
            return th.cat([local] * self.world)

        def shared_seed(self):
            return 1234

    dp = NullDP(world)
if len(sys.argv) > 3:
    tr, per_round = bench.build_variant(sys.argv[3])
else:
    tr = bench.build_trainer(bench.hip_namespace(), cfg, "cuda", seed=0, dp=dp)
    per_round = cfg["n_envs"] * cfg["n_steps"]
tr.train(3 * per_round)
th.cuda.synchronize()
algo = tr.gen_algo
marks = []
t0 = [0.0]
ev0 = [None]
dev_marks = []


def host_mark(name):
    marks.append((name, 1e3 * (time.perf_counter() - t0[0])))


def dev_mark(name, stream=None):
    e = th.cuda.Event(enable_timing=True)
    e.record(stream if stream is not None else th.cuda.current_stream())
    dev_marks.append((name, e))


def wrap(obj, attr, name, dev=False, stream_of=None):
    orig = getattr(obj, attr)

    def f(*a, **k):
        r = orig(*a, **k)
        host_mark(name)
        if dev:
            dev_mark(name)
        return r

    setattr(obj, attr, f)


wrap(algo, "collect_rollouts", "collect_rollouts returned", dev=True)
wrap(algo, "train", "ppo train enqueued", dev=True)
wrap(tr, "_disc_round", "disc round enqueued", dev=True)
wrap(tr.venv_buffering, "pop_transitions_and_lens", "  pop_transitions returned")
wrap(tr._gen_replay_buffer, "store", "  replay store returned")
wrap(tr, "_replay_policy_norm_updates", "norm replay enqueued", dev=True)
_orig_fin = tr._finish_disc_round


def _fin(p):
    host_mark("  drain starts (disc r-1 done event waited)")
    r = _orig_fin(p)
    host_mark("disc stats logged")
    return r


tr._finish_disc_round = _fin
wrap(algo, "finalize_train", "ppo stats logged")
# multi-round call: rounds overlap (the discriminator updates of round r run behind the environment
# stepping of round r+1), so print absolute host / device times of everything in one train() call
marks.clear()
dev_marks.clear()
th.cuda.synchronize()
t0[0] = time.perf_counter()
s = th.cuda.Event(enable_timing=True)
s.record()
tr.train(rounds * per_round)
host_mark("train() returned")
th.cuda.synchronize()
host_mark("device idle")
print(f"{rounds} rounds in one train() call: {marks[-1][1] / rounds:.2f} ms/round")
print("host timeline (ms):")
for n, t in marks:
    print(f"  {t:8.2f}  {n}")
print("device timeline (ms; completion of the work enqueued up to that point on that stream):")
for n, e in dev_marks:
    print(f"  {s.elapsed_time(e):8.2f}  {n}")
