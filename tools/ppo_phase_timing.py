"""Shader-clock timing of the ppo_grad kernel phases (block 0) at config P.
Usage: python tools/ppo_phase_timing.py [variant]   (e.g. P_mlp64_1024x16: the per-minibatch kernels of H = 64)"""
import os, sys
import numpy as np, torch as th
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from imitation_amd import _lib as L
cfg = dict(bench.CFG_P)
th.set_num_threads(1)
tr = bench.build_variant(sys.argv[1])[0] if len(sys.argv) > 1 else bench.build_trainer(bench.hip_namespace(), cfg, "cuda")
tr.train(2 * 16384)
buf = th.zeros(16, dtype=th.int64, device="cuda")
L.load().ia_ppo_debug_timing(buf.data_ptr())
tr.train(16384)
th.cuda.synchronize()
t = buf.cpu().numpy()
L.load().ia_ppo_debug_timing(None)
d = np.diff(t[:9])
names = ["0 gather", "1 layer1", "2 layer2", "3 heads", "4 loss", "5 head grads/dz2", "6 dW2/da1", "7 dW1"]
print("shader clocks per phase (last minibatch, block 0):")
for n, v in zip(names, d):
    print(f"  {n:18s} {v:8d} clk  ~{v / 2.1e3:6.2f} us @2.1GHz")
print("  total", t[8] - t[0], "clk")
print("  phase0 detail: start->obs-loads-issued", t[9]-t[0], " ->param copy issued", t[10]-t[9], " ->LDS staged+sync", t[11]-t[10], " ->fragments built", t[1]-t[11])
