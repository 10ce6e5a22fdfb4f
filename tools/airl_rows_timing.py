"""Shader-clock phases of `airl_rows_kernel` (workgroup 0, first lane) inside AIRL Ant-shaped training rounds.
Usage: python tools/airl_rows_timing.py"""
import os
import sys

import numpy as np
import torch as th

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from imitation_amd import _lib as L  # noqa: E402

th.set_num_threads(1)
tr, per = bench.build_variant("3_airl_ant_1024x16_mb1024")
tr.train(3 * per)
buf = th.zeros(16, dtype=th.int64, device="cuda")
L.load().ia_airl_debug_timing(buf.data_ptr())
tr.train(2 * per)
th.cuda.synchronize()
L.load().ia_airl_debug_timing(None)
t = buf.cpu().numpy()
names = ["row loads issued", "weights staged in LDS (+ barrier)", "three forwards + outputs", "logits / BCE / base deltas stored",
         "potential deltas (2 chains) + stores", "output-layer sums (shuffles) + slab", "stores acknowledged + barrier",
         "ticket (write-through hand-off)"]
for n, d in zip(names, np.diff(t[:9])):
    print(f"  {n:44s} {d:8d} clk  ~{d / 2.4e3:6.2f} us @2.4GHz")
print("  total", t[8] - t[0], "clk")
