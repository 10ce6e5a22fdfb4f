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

AIRL (BASELINE config 3: "BasicShapedRewardNet + grad-penalty"): the penalised function is the shaped reward

    f(s, a, s') = g([s | a | s' | d]) + gamma (1 - d) h(s') - h(s)            rewards/reward_nets.py:727-733

at the interpolated transition (d = the interpolated done flag, held constant; the `- log pi(a|s)` of AIRL's logit is an
input of the discriminator and carries no parameters), with the norm taken over the gradient w.r.t. EVERY input block.
That gradient is a signed combination of the three stacks' input gradients (`ia_gp_shaped_coeffs`), each stack still
piecewise linear: one first pass per stack evaluation, the row coefficients, one second pass per stack evaluation
(`shaped_penalty_and_param_grad`). For the geometry of the fused AIRL update (widths 32, inputs <= 64 columns) the
trainer uses `ShapedRewardNet.fused_grad_penalty` instead: the same arithmetic in one MFMA row kernel
(`ia_airl_gp_shaped`, csrc/airl_fused.hip), checked against the same float64 graphs.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence, Tuple

import torch as th

from imitation_amd import _lib as L


def _round_up(v: int, m: int) -> int:
    return (v + m - 1) // m * m


def _layer_offsets(dims, B):
    w_off, h_off, o, ho = [], [], 0, 0
    nl = len(dims) - 1
    for l in range(nl):
        w_off.append(o)
        o += dims[l] * dims[l + 1] + dims[l + 1]
        if l < nl - 1:
            h_off.append(ho)
            ho += B * dims[l + 1]
    return w_off, h_off


def _first_pass(flat: th.Tensor, dims, desc, Xn: th.Tensor, ld: int, B: int):
    """Forward at the (normalised) interpolates and dD/dx with dOut = 1: hidden activations (the ReLU masks),
    `u_l = dD/dz_l` per hidden layer, `gn = dD/dXn [B, ld]`."""
    dev, s = flat.device, L.stream()
    hid_w = sum(dims[1:-1])
    hidden, dhidden = th.empty(max(1, B * hid_w), device=dev), th.empty(max(1, B * hid_w), device=dev)
    out, ones, gn = th.empty(B, 1, device=dev), th.ones(B, 1, device=dev), th.empty(B, ld, device=dev)
    L.call("ia_mlp_forward", C.byref(desc), L.ptr(flat), L.ptr(Xn), ld, B, L.ptr(hidden), L.ptr(out), L.ACT_NONE, s)
    splits = max(1, min(64, B // 256))
    partials = th.empty(splits, flat.numel(), device=dev)
    L.call("ia_mlp_backward", C.byref(desc), L.ptr(flat), L.ptr(Xn), ld, B, L.ptr(hidden), L.ptr(ones), L.ptr(dhidden),
           L.ptr(partials), splits, L.ptr(gn), s)
    return hidden, dhidden, ones, gn


def _second_pass(flat: th.Tensor, dims, hidden, dhidden, ones, Cn: th.Tensor, ld: int, B: int, gflat: th.Tensor,
                 accumulate: bool) -> None:
    """Parameter gradient of `sum_rows Cn . gn` with the masks fixed: for l = 1..L `dW_l (+)= u_l^T dV_{l-1}`,
    `dV_l = m_l * (dV_{l-1} W_l^T)`, `dV_0 = Cn`; written (or added) into `gflat`'s weight entries."""
    dev, s, nl = flat.device, L.stream(), len(dims) - 1
    splits = max(1, min(64, B // 256))
    w_off, h_off = _layer_offsets(dims, B)
    dV, ldv = Cn, ld          # dV_{l-1}: [B, dims[l-1]] with row stride ldv
    for l in range(nl):       # 0-based Linear l: dims[l] -> dims[l+1]
        n_in, n_out = dims[l], dims[l + 1]
        u = ones if l == nl - 1 else dhidden[h_off[l]:h_off[l] + B * n_out]
        wp = th.empty(splits, n_out, n_in, device=dev)
        # dW_l [n_out, n_in] = u^T . dV   (split-K over the B rows, slabs reduced in fixed order)
        L.call("ia_gemm_f32", L.GEMM_TN, L.ptr(u), n_out, L.ptr(dV), ldv, L.ptr(wp), n_in, n_out, n_in, B, None, 0, None,
               0, splits, None, s)
        L.call("ia_reduce_partials", L.ptr(wp), splits, n_out * n_in, 1.0, int(accumulate),
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


def _check(act: int, dims) -> None:
    if act != L.ACT_RELU:
        raise NotImplementedError("the gradient penalty is implemented for ReLU stacks (piecewise-linear discriminators)")
    if dims[-1] != 1:
        raise NotImplementedError("the gradient penalty needs a scalar discriminator output")


def penalty_and_param_grad(flat: th.Tensor, dims: Sequence[int], act: int, X: th.Tensor, ldx: int, B: int,
                           e: th.Tensor, mean: Optional[th.Tensor], var: Optional[th.Tensor], eps: float, coef: float,
                           target: float = 1.0) -> Tuple[th.Tensor, th.Tensor]:
    """`X[2B, ldx]` = [expert rows | generator rows] (un-normalised concatenated inputs), `e[B]` the interpolation
    weights on the device. Returns `(mean_i (|grad D| - target)^2  [0-dim device tensor], gflat)` with
    `gflat = d(coef * that mean) / d flat` in the flat parameter layout (bias entries zero)."""
    dims = [int(d) for d in dims]
    _check(act, dims)
    dev, D = flat.device, dims[0]
    ld = _round_up(D, 4)
    desc = L.mlp_desc(dims, act)
    s = L.stream()
    Xn = th.empty(B, ld, device=dev)
    L.call("ia_gp_interpolate", L.ptr(X), ldx, B, D, L.ptr(e), L.ptr(mean), L.ptr(var), float(eps), L.ptr(Xn), ld, s)
    hidden, dhidden, ones, gn = _first_pass(flat, dims, desc, Xn, ld, B)
    Cn, pen = th.empty(B, ld, device=dev), th.empty(B, device=dev)
    L.call("ia_gp_row_coeffs", L.ptr(gn), ld, B, D, L.ptr(var), float(eps), float(coef), float(target), L.ptr(Cn),
           L.ptr(pen), s)
    gflat = th.zeros_like(flat)
    _second_pass(flat, dims, hidden, dhidden, ones, Cn, ld, B, gflat, False)
    return pen.mean(), gflat


def shaped_penalty_and_param_grad(base_flat: th.Tensor, base_dims: Sequence[int], pot_flat: th.Tensor,
                                  pot_dims: Sequence[int], act: int, Xb: th.Tensor, ldb: int, Sn: th.Tensor,
                                  Sc: th.Tensor, ldp: int, dones: th.Tensor, B: int, e: th.Tensor, obs_dim: int,
                                  act_dim: int, flags, base_norm, pot_norm, gamma: float, coef: float,
                                  target: float = 1.0) -> Tuple[th.Tensor, th.Tensor, th.Tensor]:
    """The penalty for AIRL's shaped reward net (`rewards/reward_nets.py:674-809`): the penalised function is the
    shaped reward `f(s, a, s') = g([s | a | s' | d]) + gamma (1 - d) h(s') - h(s)` at the interpolated transition (`d` =
    the interpolated done flag, a constant; AIRL's `- log pi` is an input of the logit, not part of the net), its
    gradient taken w.r.t. every input block. `Xb[2B, ldb]`, `Sn / Sc[2B, ldp]`, `dones[2B]` = the assembled
    [expert | generator] batches; `base_norm / pot_norm` = `(mean, var, eps)` (frozen) or None; `flags` =
    `(use_state, use_action, use_next_state, use_done)` of the base net. Returns `(mean penalty, d/d base params,
    d/d potential params)`; the three stacks' second passes start from `ia_gp_shaped_coeffs`."""
    base_dims, pot_dims = [int(d) for d in base_dims], [int(d) for d in pot_dims]
    _check(act, base_dims)
    _check(act, pot_dims)
    dev, s = base_flat.device, L.stream()
    Db, Dp = base_dims[0], pot_dims[0]
    lb, lp = _round_up(Db, 4), _round_up(Dp, 4)
    bm, bv, be = base_norm if base_norm is not None else (None, None, 0.0)
    pm, pv, pe = pot_norm if pot_norm is not None else (None, None, 0.0)
    Xn_b, Xn_n, Xn_c = th.empty(B, lb, device=dev), th.empty(B, lp, device=dev), th.empty(B, lp, device=dev)
    dhat = th.empty(B, 4, device=dev)
    L.call("ia_gp_interpolate", L.ptr(Xb), ldb, B, Db, L.ptr(e), L.ptr(bm), L.ptr(bv), float(be), L.ptr(Xn_b), lb, s)
    L.call("ia_gp_interpolate", L.ptr(Sn), ldp, B, Dp, L.ptr(e), L.ptr(pm), L.ptr(pv), float(pe), L.ptr(Xn_n), lp, s)
    L.call("ia_gp_interpolate", L.ptr(Sc), ldp, B, Dp, L.ptr(e), L.ptr(pm), L.ptr(pv), float(pe), L.ptr(Xn_c), lp, s)
    L.call("ia_gp_interpolate", L.ptr(dones), 1, B, 1, L.ptr(e), None, None, 0.0, L.ptr(dhat), 4, s)
    dh = dhat[:, 0].contiguous()
    bdesc, pdesc = L.mlp_desc(base_dims, act), L.mlp_desc(pot_dims, act)
    fb = _first_pass(base_flat, base_dims, bdesc, Xn_b, lb, B)
    fn = _first_pass(pot_flat, pot_dims, pdesc, Xn_n, lp, B)
    fc = _first_pass(pot_flat, pot_dims, pdesc, Xn_c, lp, B)
    Cb, Cnx, Cc = th.empty(B, lb, device=dev), th.empty(B, lp, device=dev), th.empty(B, lp, device=dev)
    pen = th.empty(B, device=dev)
    L.call("ia_gp_shaped_coeffs", L.ptr(fb[3]), lb, L.ptr(fn[3]), L.ptr(fc[3]), lp, L.ptr(dh), B, obs_dim, act_dim,
           *[int(f) for f in flags], L.ptr(bv), float(be), L.ptr(pv), float(pe), float(gamma), float(coef), float(target),
           L.ptr(Cb), L.ptr(Cnx), L.ptr(Cc), L.ptr(pen), s)
    gb, gp = th.zeros_like(base_flat), th.zeros_like(pot_flat)
    _second_pass(base_flat, base_dims, fb[0], fb[1], fb[2], Cb, lb, B, gb, False)
    _second_pass(pot_flat, pot_dims, fn[0], fn[1], fn[2], Cnx, lp, B, gp, False)
    _second_pass(pot_flat, pot_dims, fc[0], fc[1], fc[2], Cc, lp, B, gp, True)
    return pen.mean(), gb, gp
