"""Workload for the PMC passes: the three 256x256 discriminator GEMMs at R=16384 plus calibration
kernels with known traffic (fp32 copy of 64 MiB: reads 64 MiB, writes 64 MiB)."""
import os, sys
import torch as th
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imitation_amd import _lib as L
L.load()
dev = "cuda"
R = 16384
src = th.randn(16 * 1024 * 1024, device=dev)   # 64 MiB
dst = th.empty_like(src)
for _ in range(3):
    dst.copy_(src)                              # calibration: __amd_rocclr_copyBuffer or elementwise copy
A = th.randn(R, 256, device=dev); W = th.randn(256, 256, device=dev); Cc = th.empty(R, 256, device=dev)
bias = th.randn(256, device=dev); P = th.rand(R, 256, device=dev)
parts = th.empty(64, 256, 256, device=dev); db = th.empty(64, 256, device=dev)
for _ in range(5):
    L.call("ia_gemm_f32", 0, L.ptr(A), 256, L.ptr(W), 256, L.ptr(Cc), 256, R, 256, 256, L.ptr(bias), 1, None, 0, 1, None, L.stream())
    L.call("ia_gemm_f32", 1, L.ptr(A), 256, L.ptr(W), 256, L.ptr(Cc), 256, R, 256, 256, None, 1, L.ptr(P), 256, 1, None, L.stream())
    L.call("ia_gemm_f32", 2, L.ptr(A), 256, L.ptr(P), 256, L.ptr(parts), 256, 256, 256, R, None, 0, None, 0, 64, L.ptr(db), L.stream())
th.cuda.synchronize()
