"""Where the persistent PPO update (ia_ppo_update) spends its time at config P (or a bench variant): block 0 accumulates
100 MHz ticks per phase {wait for statistics, minibatch fwd/bwd, grid barrier, reduce+clip+Adam}.
Usage: python tools/ppo_update_timing.py [xcd_pack 0|1] [bench variant, e.g. T_gail_half_cheetah_tuned_verbatim]"""
import os
import sys
import time

import torch as th

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from imitation_amd import _lib as L  # noqa: E402

pack = int(sys.argv[1]) if len(sys.argv) > 1 else 1
th.set_num_threads(1)
if len(sys.argv) > 2:
    tr, per_round = bench.build_variant(sys.argv[2])
else:
    cfg = dict(bench.CFG_P)
    tr = bench.build_trainer(bench.hip_namespace(), cfg, "cuda")
    per_round = cfg["n_envs"] * cfg["n_steps"]
tr.train(2 * per_round)
th.cuda.synchronize()
algo = tr.gen_algo
algo.defer_train_stats = True
L.load().ia_ppo_update_xcd_pack(pack)
buf = th.zeros(64, dtype=th.int64, device="cuda")
L.load().ia_ppo_debug_timing(buf.data_ptr())
for rep in range(3):
    buf.zero_()
    th.cuda.synchronize()
    t = time.perf_counter()
    algo.train()
    th.cuda.synchronize()
    d = time.perf_counter() - t
    steps = algo.n_epochs * algo._n_mb
    ticks = buf.cpu().numpy()[:12]
    names = ("wait statistics", "minibatch fwd/bwd", "hop 2: poll the sum vector (+ barrier)", "park+sync", "(to the norm)", "norm (block sum)", "stats+Adam", "-", "-", "prefetch issue", "hop 1: poll the slice of all slabs", "slice sums + publish (+ barrier)")
    sb = buf.cpu().numpy()[12:15]
    print("statistics block per step: " + ", ".join(f"{n} {t_ / 100.0 / steps:.2f} us" for n, t_ in
                                                      zip(("wait for a free ring slot", "loss statistics of finished steps",
                                                           "minibatch statistics + publish"), sb)))
    place = buf.cpu().numpy()[42:50]
    print("packed launch: (XCC id, words stored at workgroup scope) of gradient workgroups 0-7:",
          [(int(v) >> 1, bool(int(v) & 1)) for v in place])
    polls = buf.cpu().numpy()[40:42]
    print(f"unsuccessful polls per step (thread 0 of workgroup 0): hop 1 {polls[0] / steps:.2f}, hop 2 {polls[1] / steps:.2f}")
    print(f"xcd_pack={pack}: train() {1e3 * d:.2f} ms for {steps} steps; per step: " +
          ", ".join(f"{n} {t_ / 100.0 / steps:.2f} us" for n, t_ in zip(names, ticks)))
t = buf.cpu().numpy()[16:32]
import numpy as np  # noqa: E402
d = np.diff(np.concatenate([t[:7], t[8:9]]))   # (slot 7 is not stamped: the gradient tiles are one phase)
names = ["0 stage/fragments", "1 layer1", "2 layer2", "3 heads", "4 loss", "5 head grads, dz2, dz1 + barrier",
         "6 gradient tiles (dW*, db*, statistics) + barrier"]
print("shader clocks per minibatch phase (last step, block 0):"); print("  tile phase: head tiles / column sums", t[7] - t[6], "clk, dW2 / dW1 tiles + bias sums + barrier", t[8] - t[7], "clk")
for n_, v in zip(names, d):
    print(f"  {n_:18s} {v:8d} clk  ~{v / 2.4e3:6.2f} us @2.4GHz")
print(f"  loss detail (clk): fragment requests + head outputs read {t[12] - t[4]}, log-prob / entropy over the actions "
      f"{t[13] - t[12]}, ratio / surrogate / d log-prob {t[14] - t[13]}, head gradients + statistics written {t[15] - t[14]}, "
      f"to the end of the phase {t[5] - t[15]}")
print(f"  phase 0 detail (clk): loss scalars / action loads issued {t[9] - t[0]}, rows normalised into the x tile "
      f"{t[10] - t[9]}, weight fragments LDS->VGPR + Gaussian constants {t[11] - t[10]}, block barrier {t[1] - t[11]}")
L.load().ia_ppo_debug_timing(None)
