"""Gradient penalty on the discriminator -- an OPT-IN extension (default off).

BASELINE.json's north star and its config 3 name a "grad-penalty fused with BCE"; the reference has none
(no gradient-penalty code anywhere under `/root/reference/src`), so there is no reference behaviour to
match and nothing changes unless `AdversarialTrainer(..., disc_grad_penalty_coef=c)` is given with `c > 0`.

Definition (the WGAN-GP form, Gulrajani et al. 2017, applied to the discriminator logit `D`):

    x_hat_i = e_i * x_expert_i + (1 - e_i) * x_gen_i,   e_i ~ U(0, 1) from torch's global CPU generator
    penalty  = coef * mean_i ( || grad_x D(x_hat_i) ||_2 - target )^2

for the rows `x = [state | action | ...]` a `BasicRewardNet` concatenates (`rewards/reward_nets.py:441-457`), with the
input `RunningNorm` (if any) applied with FROZEN statistics. For a ReLU stack `D` is piecewise linear in `x`, so its
input gradient is `W1^T diag(m1) W2^T diag(m2) ... w_L` with the ReLU masks `m_l` locally constant: the penalty's
parameter gradient needs no double-backward graph, only products of the same matrices --

    forward at x_hat           -> hidden activations (masks)                  ia_mlp_forward
    dD/dx (dOut = 1)           -> u_l = dD/dz_l per hidden layer, gn = dD/dxn ia_mlp_backward
    rows: n = |gn / sigma|, pen, Cn = d(coef/B sum pen)/d gn                   ia_gp_row_coeffs
    for l = 1..L:  dW_l += u_l^T . dV_{l-1}            (split-K TN GEMM)       ia_gemm_f32
                   dV_l  = m_l * (dV_{l-1} . W_l^T)     (NT GEMM + mask)        ia_gemm_f32, ia_relu_backward
    (dV_0 = Cn; biases get no gradient: with the masks fixed the input gradient does not depend on them)

-- every contraction on the fp32 MFMA GEMMs of libimitation_hip.so. Checked against torch's double-backward
(`create_graph=True`) in float64 (`tests/test_grad_penalty_gpu.py`).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence, Tuple

import torch as th

from imitation_amd import _lib as L


def _round_up(v: int, m: int) -> int:
    return (v + m - 1) // m * m


def penalty_and_param_grad(flat: th.Tensor, dims: Sequence[int], act: int, X: th.Tensor, ldx: int, B: int,
                           e: th.Tensor, mean: Optional[th.Tensor], var: Optional[th.Tensor], eps: float, coef: float,
                           target: float = 1.0) -> Tuple[th.Tensor, th.Tensor]:
    """`X[2B, ldx]` = [expert rows | generator rows] (un-normalised concatenated inputs), `e[B]` the interpolation
    weights on the device. Returns `(mean_i (|grad D| - target)^2  [0-dim device tensor], gflat)` with
    `gflat = d(coef * that mean) / d flat` in the flat parameter layout (bias entries zero)."""
    if act != L.ACT_RELU:
        raise NotImplementedError("the gradient penalty is implemented for ReLU stacks (piecewise-linear discriminators)")
    dims = [int(d) for d in dims]
    if dims[-1] != 1:
        raise NotImplementedError("the gradient penalty needs a scalar discriminator output")
    dev, D, nl = flat.device, dims[0], len(dims) - 1
    ld = _round_up(D, 4)
    desc = L.mlp_desc(dims, act)
    s = L.stream()
    Xn = th.empty(B, ld, device=dev)
    L.call("ia_gp_interpolate", L.ptr(X), ldx, B, D, L.ptr(e), L.ptr(mean), L.ptr(var), float(eps), L.ptr(Xn), ld, s)
    hid_w = sum(dims[1:-1])
    hidden, dhidden = th.empty(max(1, B * hid_w), device=dev), th.empty(max(1, B * hid_w), device=dev)
    out, ones, gn = th.empty(B, 1, device=dev), th.ones(B, 1, device=dev), th.empty(B, ld, device=dev)
    L.call("ia_mlp_forward", C.byref(desc), L.ptr(flat), L.ptr(Xn), ld, B, L.ptr(hidden), L.ptr(out), L.ACT_NONE, s)
    splits = max(1, min(64, B // 256))
    partials = th.empty(splits, flat.numel(), device=dev)
    L.call("ia_mlp_backward", C.byref(desc), L.ptr(flat), L.ptr(Xn), ld, B, L.ptr(hidden), L.ptr(ones), L.ptr(dhidden),
           L.ptr(partials), splits, L.ptr(gn), s)
    Cn, pen = th.empty(B, ld, device=dev), th.empty(B, device=dev)
    L.call("ia_gp_row_coeffs", L.ptr(gn), ld, B, D, L.ptr(var), float(eps), float(coef), float(target), L.ptr(Cn),
           L.ptr(pen), s)

    gflat = th.zeros_like(flat)
    w_off, h_off, o, ho = [], [], 0, 0
    for l in range(nl):
        w_off.append(o)
        o += dims[l] * dims[l + 1] + dims[l + 1]
        if l < nl - 1:
            h_off.append(ho)
            ho += B * dims[l + 1]
    dV, ldv = Cn, ld          # dV_{l-1}: [B, dims[l-1]] with row stride ldv
    for l in range(nl):       # 0-based Linear l: dims[l] -> dims[l+1]
        n_in, n_out = dims[l], dims[l + 1]
        u = ones if l == nl - 1 else dhidden[h_off[l]:h_off[l] + B * n_out]
        wp = th.empty(splits, n_out, n_in, device=dev)
        # dW_l [n_out, n_in] = u^T . dV   (split-K over the B rows, slabs reduced in fixed order)
        L.call("ia_gemm_f32", L.GEMM_TN, L.ptr(u), n_out, L.ptr(dV), ldv, L.ptr(wp), n_in, n_out, n_in, B, None, 0, None,
               0, splits, None, s)
        L.call("ia_reduce_partials", L.ptr(wp), splits, n_out * n_in, 1.0, 0,
               L.ptr(gflat[w_off[l]:w_off[l] + n_out * n_in]), s)
        if l < nl - 1:
            # dU_l [B, n_out] = dV . W_l^T ; dV_l = relu'(h_l) * dU_l
            dU = th.empty(B, n_out, device=dev)
            W = flat[w_off[l]:w_off[l] + n_out * n_in]
            L.call("ia_gemm_f32", L.GEMM_NT, L.ptr(dV), ldv, L.ptr(W), n_in, L.ptr(dU), n_out, B, n_out, n_in, None, 0,
                   None, 0, 1, None, s)
            h = hidden[h_off[l]:h_off[l] + B * n_out]
            dVn = th.empty(B, n_out, device=dev)
            L.call("ia_relu_backward", L.ptr(dU), L.ptr(h), B * n_out, L.ptr(dVn), s)
            dV, ldv = dVn, n_out
    return pen.mean(), gflat
