"""BC supervised step with the NatureCNN policy at BASELINE config 4's shape (84x84x4 uint8 frames, batch 4096,
Discrete(6)): samples/s of `imitation_amd.bc.BC` on the GPU next to the CPU oracle (torch CPU, same policy and
loss) on a bounded sample. Usage: python tests/perf/bc_bench.py [batch] [steps]"""
import os
import sys
import tempfile
import time

import numpy as np
import torch as th

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import imitation_amd as p  # noqa: E402
from imitation_amd import spaces  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
shape, A = (4, 84, 84), 6
osp, asp = spaces.Box(0, 255, shape, np.uint8), spaces.Discrete(A)
rng = np.random.default_rng(0)
n = 2 * B
obs = rng.integers(0, 256, (n, *shape), dtype=np.uint8)
acts = rng.integers(0, A, n).astype(np.int64)
demos = p.Transitions(obs=obs, acts=acts, next_obs=obs, dones=np.zeros(n, bool))

th.manual_seed(0)
pol = p.cnn_policy.ActorCriticCnnPolicy(osp, asp, lambda _: 1.0)
tr = p.bc.BC(observation_space=osp, action_space=asp, rng=rng, policy=pol, demonstrations=demos, batch_size=B,
             device="cuda", custom_logger=p.configure_logger(tempfile.mkdtemp(), []))
tr.train(n_batches=2, log_interval=10 ** 9)   # warm-up (buffers, first launches)
th.cuda.synchronize()
t0 = time.perf_counter()
tr.train(n_batches=steps, log_interval=10 ** 9)
th.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
g = pol.geom
fwd = sum(2.0 * B * oh * ow * (cin * k * k) * cout for cin, _, _, cout, k, _, oh, ow in g) + 2.0 * B * pol.n_flatten * 512 \
    + 2.0 * B * 512 * (A + 1)
dgrad = sum(2.0 * B * oh * ow * (cin * k * k) * cout for cin, _, _, cout, k, _, oh, ow in g[1:]) + 2.0 * B * pol.n_flatten * 512 \
    + 2.0 * B * 512 * A
flops = 2 * fwd + dgrad   # forward + weight gradients (same contraction sizes) + input gradients
print(f"HIP  BC step, NatureCNN {shape}, batch {B}: {1e3 * dt:.2f} ms/step = {B / dt / 1e3:.1f} k samples/s, "
      f"{flops / dt / 1e12:.1f} TFLOP/s of GEMM work ({flops / 1e9:.0f} GFLOP per step; fp32 MFMA peak 157.3)")

from oracle import imitation_restated as o  # noqa: E402  (CPU baseline leg: the checker timed as the baseline)
from oracle import sb3_restated as sb  # noqa: E402

for threads in (8, 32):
    th.set_num_threads(threads)
    th.manual_seed(0)
    Bc = min(B, 512)
    demos_c = o.Transitions(obs=obs[:2 * Bc], acts=acts[:2 * Bc], next_obs=obs[:2 * Bc], dones=np.zeros(2 * Bc, bool))
    tc = o.BC(observation_space=osp, action_space=asp, rng=rng, policy=sb.ActorCriticCnnPolicy(osp, asp, lambda _: 1.0),
              demonstrations=demos_c, batch_size=Bc, custom_logger=o.configure_logger(tempfile.mkdtemp(), []))
    tc.train(n_batches=1, log_interval=10 ** 9)
    t0 = time.perf_counter()
    tc.train(n_batches=2, log_interval=10 ** 9)
    dc = (time.perf_counter() - t0) / 2
    print(f"CPU oracle ({threads} torch threads), batch {Bc}: {1e3 * dc:.0f} ms/step = {Bc / dc / 1e3:.2f} k samples/s "
          f"-> GPU / CPU = {(B / dt) / (Bc / dc):.0f}x")
