"""PyTorch-ROCm custom ops over the C ABI (`include/imitation_hip.h`): the operator boundary the
reference's reward-net plugin API sits on (`rewards/reward_nets.py:16-50`: a `RewardNet` is an
`nn.Module` whose `forward` carries an autograd graph, trained by `loss.backward()` at
`algorithms/adversarial/common.py:353-369`).

Two layers:

* `torch.library` ops `imitation_amd::*` -- thin, stateless wrappers of the HIP kernels (device tensors in,
  device tensors out, launched on the current HIP stream; CUDA/ROCm device only: there is no CPU
  implementation, calling them with CPU tensors raises);
* `torch.autograd.Function`s on top (`mlp`, `running_norm_apply`, `bce_expert_first`) so that an
  `nn.Module` built from them (`imitation_amd.modules`) trains through `loss.backward()` with every
  forward / backward contraction running in libimitation_hip.so.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence, Tuple

import torch as th
from torch import Tensor

from imitation_amd import _lib as L

ACT_NONE, ACT_RELU, ACT_TANH = L.ACT_NONE, L.ACT_RELU, L.ACT_TANH


def _dev(t: Tensor, what: str) -> Tensor:
    if not t.is_cuda:
        raise RuntimeError(f"imitation_amd ops compute on MI355X only (no CPU fallback): `{what}` is on {t.device}")
    return t.contiguous().float() if (t.dtype != th.float32 or not t.is_contiguous()) else t


def _desc(dims: Sequence[int], act: int) -> L.MlpDesc:
    return L.mlp_desc([int(d) for d in dims], int(act))


def _splits(R: int) -> int:
    return max(1, min(64, R // 256))


# ------------------------------------------------------------------------------------ torch.library ops

@th.library.custom_op("imitation_amd::mlp_forward", mutates_args=(), device_types="cuda")
def mlp_forward(x: Tensor, flat: Tensor, dims: List[int], act: int, out_act: int) -> Tuple[Tensor, Tensor]:
    """`build_mlp` stack (`util/networks.py:204-283`, no input norm) on rows `x[R, dims[0]]` with the flat
    parameter vector `flat` (torch `parameters()` order W0, b0, W1, b1, ...). Returns `(out[R, dims[-1]],
    hidden)` where `hidden` holds the post-activation hidden layers the backward needs."""
    x, flat = _dev(x, "x"), _dev(flat, "flat")
    R = x.shape[0]
    d = _desc(dims, act)
    hidden = th.empty(max(1, R * sum(dims[1:-1])), device=x.device)
    out = th.empty(R, dims[-1], device=x.device)
    L.call("ia_mlp_forward", C.byref(d), L.ptr(flat), L.ptr(x), x.shape[1], R, L.ptr(hidden), L.ptr(out), out_act,
           L.stream())
    return out, hidden


@mlp_forward.register_fake
def _(x, flat, dims, act, out_act):
    R = x.shape[0]
    return x.new_empty(R, dims[-1]), x.new_empty(max(1, R * sum(dims[1:-1])))


@th.library.custom_op("imitation_amd::mlp_backward", mutates_args=(), device_types="cuda")
def mlp_backward(x: Tensor, flat: Tensor, hidden: Tensor, dout: Tensor, dims: List[int], act: int) -> Tuple[Tensor, Tensor]:
    """Autograd of `mlp_forward` for `dout[R, dims[-1]]` (gradient at the pre-activation output): returns
    `(dflat, dx)` -- the flat parameter gradient (split-K slabs reduced in fixed order: deterministic) and
    the input gradient `[R, dims[0]]`."""
    x, flat, hidden, dout = _dev(x, "x"), _dev(flat, "flat"), _dev(hidden, "hidden"), _dev(dout, "dout")
    R = x.shape[0]
    d = _desc(dims, act)
    splits = _splits(R)
    n = flat.numel()
    partials = th.empty(splits, n, device=x.device)
    dhidden = th.empty_like(hidden)
    dx = th.empty_like(x)
    L.call("ia_mlp_backward", C.byref(d), L.ptr(flat), L.ptr(x), x.shape[1], R, L.ptr(hidden), L.ptr(dout),
           L.ptr(dhidden), L.ptr(partials), splits, L.ptr(dx), L.stream())
    dflat = th.empty(n, device=x.device)
    L.call("ia_reduce_partials", L.ptr(partials), splits, n, 1.0, 0, L.ptr(dflat), L.stream())
    return dflat, dx


@mlp_backward.register_fake
def _(x, flat, hidden, dout, dims, act):
    return th.empty_like(flat), th.empty_like(x)


@th.library.custom_op("imitation_amd::running_norm_update", mutates_args=("mean", "var", "count"), device_types="cuda")
def running_norm_update(x: Tensor, mean: Tensor, var: Tensor, count: Tensor) -> None:
    """`RunningNorm.update_stats` (`util/networks.py:111-134`, Chan merge; `count` int32) with batch `x[R, F]`."""
    x = _dev(x, "x")
    R, F = x.shape
    ws = th.empty(int(L.load().ia_running_norm_ws_floats(R, F)), device=x.device)
    L.call("ia_running_norm_update", L.ptr(x), F, R, F, L.ptr(mean), L.ptr(var), L.ptr(count), L.ptr(ws), L.stream())


@th.library.custom_op("imitation_amd::running_norm_apply", mutates_args=(), device_types="cuda")
def running_norm_apply(x: Tensor, mean: Tensor, var: Tensor, eps: float) -> Tensor:
    """`(x - mean) / sqrt(var + eps)` (`util/networks.py:91`) on `x[R, F]`."""
    x = _dev(x, "x")
    R, F = x.shape
    y = th.empty_like(x)
    L.call("ia_running_norm_apply", L.ptr(x), F, R, F, L.ptr(_dev(mean, "mean")), L.ptr(_dev(var, "var")), float(eps),
           L.ptr(y), F, L.stream())
    return y


@running_norm_apply.register_fake
def _(x, mean, var, eps):
    return th.empty_like(x)


@th.library.custom_op("imitation_amd::bce_expert_first", mutates_args=(), device_types="cuda")
def bce_expert_first_raw(logits: Tensor, n_expert: int, scale: float) -> Tuple[Tensor, Tensor]:
    """`adversarial/common.py:360-368` + `27-92`: BCE-with-logits over `logits[R]` whose first `n_expert` rows are
    labelled 1 (expert) and the rest 0, mean loss scaled by `scale`. Returns `(stats[8], dlogits[R])`:
    stats = {loss, n_correct, n_correct_expert, n_correct_gen, n_pred_gen, entropy_sum, n_expert, n_gen},
    dlogits = d loss / d logits."""
    logits = _dev(logits, "logits").reshape(-1)
    R = logits.numel()
    ws = th.zeros(int(L.load().ia_bce_ws_floats(R)), device=logits.device)
    stats = th.empty(8, device=logits.device)
    dlogits = th.empty(R, device=logits.device)
    L.call("ia_bce_logits", L.ptr(logits), R, int(n_expert), float(scale), L.ptr(dlogits), L.ptr(stats), L.ptr(ws),
           L.stream())
    return stats, dlogits


@bce_expert_first_raw.register_fake
def _(logits, n_expert, scale):
    return logits.new_empty(8), logits.new_empty(logits.numel())


@th.library.custom_op("imitation_amd::gather_rows", mutates_args=(), device_types="cuda")
def gather_rows(src: Tensor, idx: Tensor) -> Tensor:
    """`src[idx]` for a row-major fp32 table `src[N, W]` and int64 `idx[n]` (batch assembly, `common.py:564-603`)."""
    src = _dev(src, "src")
    flat = src.reshape(src.shape[0], -1)
    out = th.empty(idx.numel(), flat.shape[1], device=src.device)
    L.call("ia_gather_rows", L.ptr(flat), L.ptr(idx.contiguous()), idx.numel(), flat.shape[1], L.ptr(out), L.stream())
    return out.reshape(idx.numel(), *src.shape[1:])


@gather_rows.register_fake
def _(src, idx):
    return src.new_empty(idx.numel(), *src.shape[1:])


@th.library.custom_op("imitation_amd::adam_step", mutates_args=("p", "m", "v"), device_types="cuda")
def adam_step(p: Tensor, g: Tensor, m: Tensor, v: Tensor, beta1: float, beta2: float, eps: float, weight_decay: float,
              step_size: float, bc2_sqrt: float) -> None:
    """`torch.optim.Adam` single-tensor step on contiguous fp32 buffers (`step_size = lr / (1 - b1^t)`,
    `bc2_sqrt = sqrt(1 - b2^t)` formed by the caller in double, as torch does)."""
    L.call("ia_adam_step", L.ptr(p), L.ptr(_dev(g, "g")), L.ptr(m), L.ptr(v), p.numel(), beta1, beta2, eps, weight_decay,
           step_size, bc2_sqrt, L.stream())


def conv_is_implicit(cin: int, kw: int, x_numel: int) -> bool:
    """Shapes whose convolution GEMMs read the activations through the implicit im2col view (`ia_gemm_f32_im2col_pad`):
    channel quads, a 32-deep K chunk inside one kernel row, 32-bit element offsets."""
    return cin % 4 == 0 and (kw * cin) % 32 == 0 and x_numel < 2 ** 31


def conv_is_direct_c4(cin: int, cout: int, kh: int, kw: int, stride: int, pad: int, w_in: int) -> bool:
    """The reward CNN's first convolution (`csrc/conv3x3.hip`): 3 x 3 "same" from the 4-channel frame stack to 32 channels --
    forward and weight gradient straight from the 4-channel rows, no column matrix."""
    return cin == 4 and cout == 32 and kh == kw == 3 and stride == 1 and pad == 1 and w_in <= 128


def conv_is_direct_c32(cin: int, cout: int, kh: int, kw: int, stride: int, pad: int, w_in: int) -> bool:
    """The reward CNN's second convolution (`csrc/conv3x3.hip`, `conv3x3_c32_conv_kernel`): 3 x 3 "same", 32 -> 32 channels, image
    rows that fit the kernel's LDS band (10 rows of W + 2 pixels at 33 floats + the weights in 160 KB)."""
    return cin == 32 and cout == 32 and kh == kw == 3 and stride == 1 and pad == 1 and (9 * 1024 + 10 * (w_in + 2) * 33) * 4 <= 160 * 1024


@th.library.custom_op("imitation_amd::conv2d_nhwc_forward", mutates_args=(), device_types="cuda")
def conv2d_nhwc_forward(x: Tensor, w: Tensor, b: Tensor, stride: int, pad: int, relu: bool) -> Tuple[Tensor, Tensor]:
    """Convolution of channel-last activations `x[B, H, W, Cin]` with `w[Cout, KH, KW, Cin]`, bias `b[Cout]`, zero
    padding `pad`, optional fused ReLU: im2col (border resolved in the index arithmetic) + the fp32 MFMA GEMM.
    Returns `(y[B, OH, OW, Cout], col)`; `col` is kept for the weight gradient."""
    x, w, b = _dev(x, "x"), _dev(w, "w"), _dev(b, "b")
    B, H, W, Cin = x.shape
    Cout, KH, KW, _ = w.shape
    OH, OW = (H + 2 * pad - KH) // stride + 1, (W + 2 * pad - KW) // stride + 1
    K, M = KH * KW * Cin, B * OH * OW
    y = th.empty(B, OH, OW, Cout, device=x.device)
    if conv_is_direct_c4(Cin, Cout, KH, KW, stride, pad, W):
        L.call("ia_conv3x3_c4_forward", L.ptr(x), L.ptr(w), L.ptr(b), B, H, W, int(relu), L.ptr(y), L.stream())
        return y, th.empty(0, K, device=x.device)
    if conv_is_direct_c32(Cin, Cout, KH, KW, stride, pad, W):
        wt = w.permute(1, 2, 3, 0).contiguous()   # [ky][kx][ci][co]: the kernel's LDS image, 36 KB
        L.call("ia_conv3x3_c32_conv", L.ptr(x), L.ptr(wt), L.ptr(b), None, B, H, W, int(relu), L.ptr(y), L.stream())
        return y, th.empty(0, K, device=x.device)
    if conv_is_implicit(Cin, KW, B * H * W * Cin):
        # the GEMM reads its operand through the padded im2col view of `x`: no column buffer (`col` comes back empty;
        # the backward op takes `x` instead)
        L.call("ia_gemm_f32_im2col_pad", L.GEMM_NT, L.ptr(x), K, L.ptr(w), K, L.ptr(y), Cout, M, Cout, K, L.ptr(b),
               ACT_RELU if relu else ACT_NONE, 1, None, H, W, Cin, KH, KW, stride, pad, None, None, L.stream())
        return y, th.empty(0, K, device=x.device)
    col = th.empty(M, K, device=x.device)
    L.call("ia_im2col_f32_nhwc_pad", L.ptr(x), B, H, W, Cin, KH, KW, stride, pad, L.ptr(col), L.stream())
    L.call("ia_gemm_f32", L.GEMM_NT, L.ptr(col), K, L.ptr(w), K, L.ptr(y), Cout, M, Cout, K, L.ptr(b),
           ACT_RELU if relu else ACT_NONE, None, 0, 1, None, L.stream())
    return y, col


@conv2d_nhwc_forward.register_fake
def _(x, w, b, stride, pad, relu):
    B, H, W, Cin = x.shape
    Cout, KH, KW, _ = w.shape
    OH, OW = (H + 2 * pad - KH) // stride + 1, (W + 2 * pad - KW) // stride + 1
    rows = 0 if (conv_is_implicit(Cin, KW, B * H * W * Cin) or conv_is_direct_c4(Cin, Cout, KH, KW, stride, pad, W)
                 or conv_is_direct_c32(Cin, Cout, KH, KW, stride, pad, W)) else B * OH * OW
    return x.new_empty(B, OH, OW, Cout), x.new_empty(rows, KH * KW * Cin)


@th.library.custom_op("imitation_amd::conv2d_nhwc_backward", mutates_args=(), device_types="cuda")
def conv2d_nhwc_backward(dy: Tensor, y: Tensor, col: Tensor, x: Tensor, w: Tensor, in_h: int, in_w: int, stride: int,
                         pad: int, relu: bool, need_dx: bool, mask_dx: bool = False) -> Tuple[Tensor, Tensor, Tensor]:
    """Backward of `conv2d_nhwc_forward`: `(dx[B, H, W, Cin] (zeros when not needed), dw, db)`. ReLU backward from
    the saved output, weight gradient = split-K TN GEMM on the kept columns -- or, when the forward ran without a
    column buffer (`col` empty), on the implicit view of the input `x` -- slabs reduced in fixed order; input gradient =
    NN GEMM + gather-form col2im. `mask_dx`: the input `x` is itself a ReLU output whose backward the caller wants applied to
    `dx` here (zero where x <= 0: in the input-gradient GEMM's epilogue, instead of a pass of its own over the tensor)."""
    dy, y, col, x, w = _dev(dy, "dy"), _dev(y, "y"), _dev(col, "col"), _dev(x, "x"), _dev(w, "w")
    B, OH, OW, Cout = y.shape
    _, KH, KW, Cin = w.shape
    K, M = KH * KW * Cin, B * OH * OW
    dz = dy
    if relu:
        dz = th.empty_like(dy)
        L.call("ia_relu_backward", L.ptr(dy), L.ptr(y), dy.numel(), L.ptr(dz), L.stream())
    if (col.shape[0] == 0 and KH == KW == 3 and stride == 1 and pad == 1 and Cin == 32 and Cout == 32 and in_w <= 128
            and (OH, OW) == (in_h, in_w)):
        # the reward CNN's own geometry: whole images per workgroup, the 32 x 288 gradient in its accumulators, x and dz read
        # once (`csrc/conv3x3.hip`; the split-K GEMM below ran this 7.2 M-row product at 36 TFLOP/s)
        splits = int(L.load().ia_conv3x3_c32_wgrad_slabs(B))
        part = th.empty(splits, Cout, K, device=dy.device)
        dbp = th.empty(splits, Cout, device=dy.device)
        L.call("ia_conv3x3_c32_wgrad", L.ptr(dz), L.ptr(x), B, in_h, in_w, L.ptr(part), L.ptr(dbp), L.stream())
        return _conv_backward_tail(dz, w, part, dbp, splits, B, in_h, in_w, OH, OW, Cin, Cout, KH, KW, K, M, stride, pad, need_dx,
                                   x if mask_dx else None)
    if col.shape[0] == 0 and conv_is_direct_c4(Cin, Cout, KH, KW, stride, pad, in_w):
        splits = int(L.load().ia_conv3x3_c32_wgrad_slabs(B))
        part = th.empty(splits, Cout, K, device=dy.device)
        dbp = th.empty(splits, Cout, device=dy.device)
        L.call("ia_conv3x3_c4_wgrad", L.ptr(dz), L.ptr(x), B, in_h, in_w, L.ptr(part), L.ptr(dbp), L.stream())
        return _conv_backward_tail(dz, w, part, dbp, splits, B, in_h, in_w, OH, OW, Cin, Cout, KH, KW, K, M, stride, pad, need_dx,
                                   x if mask_dx else None)
    # split-K over the rows: 64 splits left the weight gradient of a 1 024-frame 84 x 84 batch (7.2 M rows against a 32 x 288
    # output: 5 tiles) on 320 workgroups of 113 k rows each -- 7.7 ms per call, 17 TFLOP/s (`profiles/r05_image_gail.md`)
    splits = int(min(1024, max(1, M // 2048)))
    part = th.empty(splits, Cout, K, device=dy.device)
    dbp = th.empty(splits, Cout, device=dy.device)
    if col.shape[0] == 0:
        L.call("ia_gemm_f32_im2col_pad", L.GEMM_TN, L.ptr(dz), Cout, L.ptr(x), K, L.ptr(part), K, Cout, K, M, None, 0,
               splits, L.ptr(dbp), in_h, in_w, Cin, KH, KW, stride, pad, None, None, L.stream())
    else:
        L.call("ia_gemm_f32", L.GEMM_TN, L.ptr(dz), Cout, L.ptr(col), K, L.ptr(part), K, Cout, K, M, None, 0, None, 0,
               splits, L.ptr(dbp), L.stream())
    return _conv_backward_tail(dz, w, part, dbp, splits, B, in_h, in_w, OH, OW, Cin, Cout, KH, KW, K, M, stride, pad, need_dx,
                                   x if mask_dx else None)


def _conv_backward_tail(dz, w, part, dbp, splits, B, in_h, in_w, OH, OW, Cin, Cout, KH, KW, K, M, stride, pad, need_dx, mask=None):
    """Slab reduction of the weight / bias gradient (fixed order) and the input gradient of `conv2d_nhwc_backward`
    (`mask`: zero it where that tensor -- the convolution's input, a ReLU output -- is <= 0)."""
    dy = dz
    dw, db = th.empty(Cout, KH, KW, Cin, device=dy.device), th.empty(Cout, device=dy.device)
    L.call("ia_reduce_partials", L.ptr(part), splits, Cout * K, 1.0, 0, L.ptr(dw), L.stream())
    L.call("ia_reduce_partials", L.ptr(dbp), splits, Cout, 1.0, 0, L.ptr(db), L.stream())
    dx = th.zeros(B, in_h, in_w, Cin, device=dy.device) if not need_dx else th.empty(B, in_h, in_w, Cin, device=dy.device)
    if need_dx and (OH, OW) == (in_h, in_w) and conv_is_direct_c32(Cout, Cin, KH, KW, stride, pad, in_w):
        # the same geometry read backwards: the convolution of `dz` with the flipped / transposed kernel, the layer below's ReLU
        # mask in the epilogue (`conv3x3_c32_conv_kernel`)
        wd = w.flip(1, 2).permute(1, 2, 0, 3).contiguous()   # [ky][kx][co][ci] of the flipped kernel: contraction index first
        L.call("ia_conv3x3_c32_conv", L.ptr(dz), L.ptr(wd), None, L.ptr(mask), B, in_h, in_w, 0, L.ptr(dx), L.stream())
    elif need_dx and stride == 1 and KH - 1 - pad >= 0 and KH == KW and conv_is_implicit(Cout, KW, dz.numel()):
        # d loss / d input of a stride-1 convolution = the padded (KH - 1 - pad) stride-1 convolution of `dz` with the
        # flipped / transposed kernel Wd[c, i, j, co] = W[co, KH-1-i, KW-1-j, c]: again an implicit GEMM over the im2col VIEW
        # of dz -- no [M, KH*KW*Cin] column-gradient buffer (8.3 GB per call at 1 024 frames of 84 x 84 x 32) and no col2im
        # pass over it (`ia_gemm_f32_im2col_pad`, the form `cnn_policy._dgrad_implicit` uses for NatureCNN)
        wd = w.flip(1, 2).permute(3, 1, 2, 0).contiguous()
        Kd = KH * KW * Cout
        L.call("ia_gemm_f32_im2col_pad", L.GEMM_NT, L.ptr(dz), Kd, L.ptr(wd), Kd, L.ptr(dx), Cin, B * in_h * in_w, Cin, Kd,
               None, ACT_NONE, 1, None, OH, OW, Cout, KH, KW, 1, KH - 1 - pad, None, L.ptr(mask), L.stream())
    elif need_dx:
        dcol = th.empty(M, K, device=dy.device)
        L.call("ia_gemm_f32", L.GEMM_NN, L.ptr(dz), Cout, L.ptr(w), K, L.ptr(dcol), K, M, K, Cout, None, 0, None, 0, 1,
               None, L.stream())
        L.call("ia_col2im_nhwc_pad", L.ptr(dcol), B, in_h, in_w, Cin, KH, KW, stride, pad, L.ptr(dx), L.stream())
        if mask is not None:
            L.call("ia_relu_backward", L.ptr(dx), L.ptr(mask), dx.numel(), L.ptr(dx), L.stream())
    return dx, dw, db


@conv2d_nhwc_backward.register_fake
def _(dy, y, col, x, w, in_h, in_w, stride, pad, relu, need_dx, mask_dx=False):
    B = y.shape[0]
    return y.new_empty(B, in_h, in_w, w.shape[3]), th.empty_like(w), y.new_empty(w.shape[0])


@th.library.custom_op("imitation_amd::avgpool_nhwc", mutates_args=(), device_types="cuda")
def avgpool_nhwc(y: Tensor) -> Tensor:
    """`nn.AdaptiveAvgPool2d(1)` + flatten on channel-last `y[B, H, W, C]` -> `[B, C]`."""
    y = _dev(y, "y")
    B, H, W, Cc = y.shape
    out = th.empty(B, Cc, device=y.device)
    L.call("ia_avgpool_nhwc", L.ptr(y), B, H * W, Cc, L.ptr(out), L.stream())
    return out


@avgpool_nhwc.register_fake
def _(y):
    return y.new_empty(y.shape[0], y.shape[3])


@th.library.custom_op("imitation_amd::avgpool_nhwc_backward", mutates_args=(), device_types="cuda")
def avgpool_nhwc_backward(dout: Tensor, h: int, w: int) -> Tensor:
    dout = _dev(dout, "dout")
    B, Cc = dout.shape
    dy = th.empty(B, h, w, Cc, device=dout.device)
    L.call("ia_avgpool_nhwc_backward", L.ptr(dout), B, h * w, Cc, L.ptr(dy), L.stream())
    return dy


@avgpool_nhwc_backward.register_fake
def _(dout, h, w):
    return dout.new_empty(dout.shape[0], h, w, dout.shape[1])


# ------------------------------------------------------------------------------------ autograd functions

class _Conv(th.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, stride, pad, relu, x_is_relu, dy_is_masked):
        y, col = th.ops.imitation_amd.conv2d_nhwc_forward(x, w, b, stride, pad, relu)
        ctx.save_for_backward(y, col, x, w)
        ctx.cfg = (x.shape[1], x.shape[2], stride, pad, relu and not dy_is_masked, x_is_relu)
        return y

    @staticmethod
    def backward(ctx, dy):
        y, col, x, w = ctx.saved_tensors
        in_h, in_w, stride, pad, relu, x_is_relu = ctx.cfg
        dx, dw, db = th.ops.imitation_amd.conv2d_nhwc_backward(dy.contiguous(), y, col, x, w, in_h, in_w, stride, pad,
                                                               relu, bool(ctx.needs_input_grad[0]), x_is_relu)
        return (dx if ctx.needs_input_grad[0] else None), dw, db, None, None, None, None, None


def conv2d_nhwc(x: Tensor, w: Tensor, b: Tensor, stride: int = 1, pad: int = 0, relu: bool = False, x_is_relu: bool = False,
                dy_is_masked: bool = False) -> Tensor:
    """Differentiable channel-last convolution (+ fused ReLU): `x[B, H, W, Cin]`, `w[Cout, KH, KW, Cin]`.
    A chain of ReLU convolutions can hand the ReLU backward of layer i to layer i + 1's input-gradient GEMM (its epilogue
    zeroes dx where its input -- layer i's ReLU output -- is <= 0): layer i + 1 is built with `x_is_relu=True`, layer i with
    `dy_is_masked=True` (the gradient it receives already carries its own mask: no `ia_relu_backward` pass). Same values."""
    return _Conv.apply(_dev(x, "x"), w, b, int(stride), int(pad), bool(relu), bool(x_is_relu), bool(dy_is_masked))


class _AvgPool(th.autograd.Function):
    @staticmethod
    def forward(ctx, y):
        ctx.hw = (y.shape[1], y.shape[2])
        return th.ops.imitation_amd.avgpool_nhwc(y)

    @staticmethod
    def backward(ctx, dout):
        return th.ops.imitation_amd.avgpool_nhwc_backward(dout.contiguous(), *ctx.hw)


def avgpool_nhwc_fn(y: Tensor) -> Tensor:
    return _AvgPool.apply(_dev(y, "y"))


@th.library.custom_op("imitation_amd::avgpool_relu_backward", mutates_args=(), device_types="cuda")
def avgpool_relu_backward(dout: Tensor, y: Tensor) -> Tensor:
    """Backward of "ReLU, then `AdaptiveAvgPool2d(1)`" through the saved post-activation `y[B, H, W, C]`: the pool's
    `dout / (H W)` under the ReLU's mask in ONE pass (read y, write dz) -- apart, the broadcast writes a full-resolution
    tensor that `ia_relu_backward` reads back together with `y` (2.8 GB moved per 1 024 x 84 x 84 x 32 call instead of 1.85)."""
    dout, y = _dev(dout, "dout"), _dev(y, "y")
    B, H, W, Cc = y.shape
    dz = th.empty_like(y)
    L.call("ia_avgpool_relu_backward", L.ptr(dout), L.ptr(y), B, H * W, Cc, L.ptr(dz), L.stream())
    return dz


@avgpool_relu_backward.register_fake
def _(dout, y):
    return th.empty_like(y)


class _ConvReluPool(th.autograd.Function):
    """`Conv2d - ReLU - AdaptiveAvgPool2d(1) - Flatten` (the tail of `build_cnn`, `util/networks.py:340-349`) as one node: the same
    two forward launches as `conv2d_nhwc(relu=True)` + `avgpool_nhwc_fn`, a backward that forms the convolution's
    pre-activation gradient in one pass."""

    @staticmethod
    def forward(ctx, x, w, b, stride, pad, x_is_relu):
        y, col = th.ops.imitation_amd.conv2d_nhwc_forward(x, w, b, stride, pad, True)
        ctx.save_for_backward(y, col, x, w)
        ctx.cfg = (x.shape[1], x.shape[2], stride, pad, x_is_relu)
        return th.ops.imitation_amd.avgpool_nhwc(y)

    @staticmethod
    def backward(ctx, dout):
        y, col, x, w = ctx.saved_tensors
        in_h, in_w, stride, pad, x_is_relu = ctx.cfg
        dz = th.ops.imitation_amd.avgpool_relu_backward(dout.contiguous(), y)
        dx, dw, db = th.ops.imitation_amd.conv2d_nhwc_backward(dz, y, col, x, w, in_h, in_w, stride, pad, False,
                                                               bool(ctx.needs_input_grad[0]), x_is_relu)
        return (dx if ctx.needs_input_grad[0] else None), dw, db, None, None, None


def conv2d_relu_avgpool_nhwc(x: Tensor, w: Tensor, b: Tensor, stride: int = 1, pad: int = 0, x_is_relu: bool = False) -> Tensor:
    """`avgpool_nhwc_fn(conv2d_nhwc(x, w, b, stride, pad, relu=True))` with the fused backward (C % 4 == 0); `x_is_relu` as
    in `conv2d_nhwc`."""
    if w.shape[0] % 4:
        return avgpool_nhwc_fn(conv2d_nhwc(x, w, b, stride, pad, relu=True, x_is_relu=x_is_relu))
    return _ConvReluPool.apply(_dev(x, "x"), w, b, int(stride), int(pad), bool(x_is_relu))


class _Mlp(th.autograd.Function):
    @staticmethod
    def forward(ctx, x, flat, dims, act):
        out, hidden = th.ops.imitation_amd.mlp_forward(x, flat, list(dims), act, ACT_NONE)
        ctx.save_for_backward(x, flat, hidden)
        ctx.dims, ctx.act = list(dims), act
        return out

    @staticmethod
    def backward(ctx, dout):
        x, flat, hidden = ctx.saved_tensors
        dflat, dx = th.ops.imitation_amd.mlp_backward(x, flat, hidden, dout.contiguous(), ctx.dims, ctx.act)
        return (dx if ctx.needs_input_grad[0] else None), (dflat if ctx.needs_input_grad[1] else None), None, None


def mlp(x: Tensor, flat: Tensor, dims: Sequence[int], act: int = ACT_RELU) -> Tensor:
    """Differentiable dense stack: `out[R, dims[-1]]`; gradients flow to `x` and to `flat`."""
    return _Mlp.apply(_dev(x, "x"), _dev(flat, "flat"), tuple(int(d) for d in dims), int(act))


class _NormApply(th.autograd.Function):
    @staticmethod
    def forward(ctx, x, mean, var, eps):
        ctx.save_for_backward(var)
        ctx.eps = eps
        return th.ops.imitation_amd.running_norm_apply(x, mean, var, eps)

    @staticmethod
    def backward(ctx, dy):
        (var,) = ctx.saved_tensors
        # d/dx (x - mean) / sqrt(var + eps) with the statistics held constant (they are buffers, `networks.py:72-77`)
        return th.ops.imitation_amd.running_norm_apply(dy.contiguous(), th.zeros_like(var), var, ctx.eps), None, None, None


def running_norm_apply_fn(x: Tensor, mean: Tensor, var: Tensor, eps: float) -> Tensor:
    return _NormApply.apply(_dev(x, "x"), mean, var, float(eps))


class _Bce(th.autograd.Function):
    @staticmethod
    def forward(ctx, logits, n_expert, scale):
        stats, dlogits = th.ops.imitation_amd.bce_expert_first(logits, n_expert, scale)
        ctx.save_for_backward(dlogits)
        ctx.shape = logits.shape
        ctx.mark_non_differentiable(stats)
        return stats[0].clone(), stats

    @staticmethod
    def backward(ctx, dloss, _dstats):
        (dlogits,) = ctx.saved_tensors
        return (dlogits * dloss).reshape(ctx.shape), None, None


def bce_expert_first(logits: Tensor, n_expert: int, scale: float = 1.0) -> Tuple[Tensor, Tensor]:
    """Differentiable `binary_cross_entropy_with_logits(logits, [1]*n_expert + [0]*(R - n_expert)) * scale`
    plus the statistics row of `compute_train_stats` (not differentiable). Returns `(loss, stats[8])`."""
    return _Bce.apply(_dev(logits, "logits"), int(n_expert), float(scale))


# ------------------------------------------------------------------------------------ optimiser

class HipAdam(th.optim.Optimizer):
    """`torch.optim.Adam` (no amsgrad) over ordinary `nn.Parameter`s, each step one `imitation_amd::adam_step`
    launch per parameter tensor. What `AdversarialTrainer` builds for an `nn.Module` reward net when
    `disc_opt_cls` is `torch.optim.Adam` (`adversarial/common.py:218-221`)."""

    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0,
                 amsgrad: bool = False):
        if amsgrad:
            raise NotImplementedError("amsgrad is not implemented on the HIP path")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))

    @th.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = th.zeros_like(p, memory_format=th.contiguous_format)
                    st["exp_avg_sq"] = th.zeros_like(p, memory_format=th.contiguous_format)
                st["step"] += 1
                t = st["step"]
                th.ops.imitation_amd.adam_step(p.data, p.grad.contiguous(), st["exp_avg"], st["exp_avg_sq"], b1, b2,
                                               group["eps"], group["weight_decay"], group["lr"] / (1.0 - b1 ** t),
                                               (1.0 - b2 ** t) ** 0.5)
        return loss
