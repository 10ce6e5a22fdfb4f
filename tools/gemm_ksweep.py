"""K-sweep of the fp32 MFMA GEMM at the discriminator's M, N: the slope is the steady-state
K-loop rate, the intercept the fixed (launch + prologue + epilogue) cost per launch.
Usage: GEMM_CFGS=1,0 python tools/gemm_ksweep.py [iters]"""
import os
import sys

import torch as th

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imitation_amd import _lib as L  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
R, N = 16384, 256
lib = L.load()
for cfg in [int(c) for c in os.environ.get("GEMM_CFGS", "-1").split(",")]:
    lib.ia_gemm_set_config(cfg)
    for mode in (0, 1):
        pts = []
        for K in (64, 128, 256, 512, 1024, 2048):
            A = th.randn(R, K, device="cuda")
            B = th.randn(N, K, device="cuda") if mode == 0 else th.randn(K, N, device="cuda")
            Cc = th.empty(R, N, device="cuda")
            bias = th.randn(N, device="cuda")

            def run():
                L.call("ia_gemm_f32", mode, L.ptr(A), K, L.ptr(B), B.shape[1], L.ptr(Cc), N, R, N, K,
                       L.ptr(bias) if mode == 0 else None, 1, None, N, 1, None, L.stream())
            for _ in range(5):
                run()
            th.cuda.synchronize()
            e0, e1 = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                run()
            e1.record()
            th.cuda.synchronize()
            us = 1e3 * e0.elapsed_time(e1) / iters
            pts.append((K, us))
        (k0, t0), (k1, t1) = pts[2], pts[-1]
        slope = (t1 - t0) / (k1 - k0) * 256
        print(f"cfg {cfg:2d} mode {mode}: " + "  ".join(f"K={k}:{t:7.2f}us" for k, t in pts) +
              f"   slope {slope:6.2f} us/256k ({2.0 * R * N * 256 / slope / 1e6:5.1f} TF steady)  intercept {t0 - slope:6.2f} us")
