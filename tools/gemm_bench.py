"""Microbenchmark of the fp32 MFMA GEMM (C-ABI `ia_gemm_f32`) on the discriminator shapes.
Back-to-back launches timed with torch (HIP) events on the launch stream.
Usage: python tools/gemm_bench.py [iters]"""
import os
import sys

import torch as th

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imitation_amd import _lib as L  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
dev = "cuda"
R = 16384
shapes = [  # (mode, M, N, K, splits, label)
    (0, R, 256, 24, 1, "fwd  L1  [R,24]x[256,24]^T"),
    (0, R, 256, 256, 1, "fwd  L2  [R,256]x[256,256]^T"),
    (0, R, 1, 256, 1, "fwd  L3  [R,256]x[1,256]^T"),
    (1, R, 256, 1, 1, "dgrad L3 [R,1]x[1,256]"),
    (1, R, 256, 256, 1, "dgrad L2 [R,256]x[256,256]"),
    (2, 256, 256, R, 64, "wgrad L2 [R,256]^T x [R,256] split 64"),
    (2, 256, 256, R, 32, "wgrad L2 split 32"),
    (2, 256, 24, R, 64, "wgrad L1 [R,256]^T x [R,24]"),
    (2, 1, 256, R, 64, "wgrad L3 [R,1]^T x [R,256]"),
    (0, 4096, 256, 256, 1, "fwd  L2 R=4096"),
    (0, 65536, 256, 256, 1, "fwd  L2 R=65536"),
]
L.load()
cfgs = [int(c) for c in os.environ.get("GEMM_CFGS", "-1").split(",")]
shapes = [(m, M, N, K, sp, f"cfg{c:2d} " + lab, c) for (m, M, N, K, sp, lab) in shapes for c in cfgs]
for mode, M, N, K, splits, label, cfg in shapes:
    L.load().ia_gemm_set_config(cfg)
    if mode == 0:
        A, B = th.randn(M, K, device=dev), th.randn(N, K, device=dev)
    elif mode == 1:
        A, B = th.randn(M, K, device=dev), th.randn(K, N, device=dev)
    else:
        A, B = th.randn(K, M, device=dev), th.randn(K, N, device=dev)
    Cc = th.empty(splits, M, N, device=dev)
    bias = th.randn(N, device=dev)
    P = th.rand(M, N, device=dev)
    db = th.empty(splits, M, device=dev)
    def run():
        L.call("ia_gemm_f32", mode, L.ptr(A), A.shape[1], L.ptr(B), B.shape[1], L.ptr(Cc), N, M, N, K,
               L.ptr(bias) if mode == 0 else None, 1, L.ptr(P) if mode == 1 else None, N, splits,
               L.ptr(db) if (mode == 2 and not os.environ.get("NO_DBIAS")) else None, L.stream())
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
    fl = 2.0 * M * N * K
    print(f"{label:42s} {us:8.2f} us  {fl / us / 1e6:8.2f} TFLOP/s  ({100 * fl / us / 1e6 / 157.3:5.1f}% of fp32 MFMA peak)")
