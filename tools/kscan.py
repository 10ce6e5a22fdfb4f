import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch as th
from imitation_amd import _lib as L
L.load()
dev="cuda"
for cfg in (1, 0):
    L.load().ia_gemm_set_config(cfg)
    for R in (4096, 16384):
        for K in (32, 64, 256, 1024):
            A, B = th.randn(R, K, device=dev), th.randn(256, K, device=dev)
            Cc = th.empty(R, 256, device=dev); bias = th.randn(256, device=dev)
            run = lambda: L.call("ia_gemm_f32", 0, L.ptr(A), K, L.ptr(B), K, L.ptr(Cc), 256, R, 256, K, L.ptr(bias), 1, None, 0, 1, None, L.stream())
            for _ in range(5): run()
            th.cuda.synchronize()
            e0, e1 = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(100): run()
            e1.record(); th.cuda.synchronize()
            us = 10 * e0.elapsed_time(e1)
            print(f"cfg {cfg} R={R:6d} K={K:5d}: {us:8.2f} us  {2.0*R*256*K/us/1e6:7.2f} TF")
