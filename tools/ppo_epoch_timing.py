"""Phase ticks of the one-launch-per-epoch PPO kernel (64-wide towers): workgroup 0 accumulates 100 MHz ticks in
{gradient, barrier, slab sum, barrier, norm + Adam, barrier}. Usage: python tools/ppo_epoch_timing.py [bench variant]"""
import os
import sys

import torch as th

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from imitation_amd import _lib as L  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "P_mlp64_1024x16"
th.set_num_threads(1)
tr, per = bench.build_variant(name)
tr.train(3 * per)
th.cuda.synchronize()
buf = th.zeros(8, dtype=th.int64, device="cuda")
L.load().ia_ppo_epoch_debug_timing(buf.data_ptr())
rounds = 4
tr.train(rounds * per)
th.cuda.synchronize()
L.load().ia_ppo_epoch_debug_timing(None)
algo = tr.gen_algo
steps = rounds * algo.n_epochs * algo._n_mb
t = buf.cpu().numpy()[:6] / 100.0 / steps
names = ("gradient", "barrier", "slab sum + partial norm", "barrier", "norm + Adam + statistics", "barrier")
print(f"{name}: per optimiser step (workgroup 0): " + ", ".join(f"{n} {v:.2f} us" for n, v in zip(names, t)) +
      f"; sum {t.sum():.2f} us")
