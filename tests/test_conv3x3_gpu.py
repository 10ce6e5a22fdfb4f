"""The reward CNN's dedicated kernels (`csrc/conv3x3.hip`): weight / bias gradient of the 3 x 3 "same" convolution with 32 -> 32
channels against float64 autograd of `torch.nn.functional.conv2d` on the CPU (`util/networks.py:286-357` builds the layer
out of `nn.Conv2d`), and the fused "ReLU then global average pool" backward against its two separate passes, bit for bit."""
import numpy as np
import pytest
import torch as th

from imitation_amd import _lib as L
from imitation_amd import ops

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,H,W", [(1, 1, 2), (3, 5, 7), (2, 9, 16), (5, 12, 31), (2, 84, 84), (1030, 4, 6)])
def test_conv3x3_c32_weight_gradient_matches_float64_autograd(B, H, W):
    g = th.Generator().manual_seed(B * 1000 + H * 10 + W)
    x = th.randn(B, H, W, 32, generator=g)
    dz = th.randn(B, H, W, 32, generator=g)
    w64 = th.zeros(32, 32, 3, 3, dtype=th.float64, requires_grad=True)
    b64 = th.zeros(32, dtype=th.float64, requires_grad=True)
    y = th.nn.functional.conv2d(x.double().permute(0, 3, 1, 2), w64, b64, padding=1)
    (y * dz.double().permute(0, 3, 1, 2)).sum().backward()
    want_w = w64.grad.permute(0, 2, 3, 1).contiguous()   # [Cout, KH, KW, Cin]: the layout of the slabs
    want_b = b64.grad
    xd, dzd = x.cuda(), dz.cuda()
    slabs = int(L.load().ia_conv3x3_c32_wgrad_slabs(B))
    assert slabs == min(B, 1024)
    part = th.full((slabs, 32, 288), float("nan"), device="cuda")
    dbp = th.full((slabs, 32), float("nan"), device="cuda")
    L.call("ia_conv3x3_c32_wgrad", L.ptr(dzd), L.ptr(xd), B, H, W, L.ptr(part), L.ptr(dbp), L.stream())
    got_w = part.double().sum(0).reshape(32, 3, 3, 32).cpu()
    got_b = dbp.double().sum(0).cpu()
    scale = float(want_w.abs().max()) + 1e-12
    assert float((got_w - want_w).abs().max()) <= 2e-5 * scale * max(1.0, np.sqrt(B * H * W / 64.0)), (B, H, W)
    assert float((got_b - want_b).abs().max()) <= 2e-5 * (float(want_b.abs().max()) + 1e-12) * max(1.0, np.sqrt(B * H * W / 64.0))


def test_conv_backward_op_takes_the_dedicated_kernel_and_matches_the_general_path():
    """`conv2d_nhwc_backward` at the reward CNN's geometry (dedicated kernel) against the same op at a geometry one channel
    group off (general split-K GEMM) is not comparable -- so: against float64 autograd, through the public op."""
    g = th.Generator().manual_seed(5)
    B, H, W = 4, 10, 12
    x = th.randn(B, H, W, 32, generator=g).cuda()
    w = (0.1 * th.randn(32, 3, 3, 32, generator=g)).cuda()
    b = (0.1 * th.randn(32, generator=g)).cuda()
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True)
    out = ops.conv2d_relu_avgpool_nhwc(xr, wr, br, 1, 1)
    coef = th.randn(B, 32, generator=g).cuda()
    (out * coef).sum().backward()
    x64 = x.double().cpu().permute(0, 3, 1, 2).requires_grad_(True)
    w64 = w.double().cpu().permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    b64 = b.double().cpu().requires_grad_(True)
    y64 = th.relu(th.nn.functional.conv2d(x64, w64, b64, padding=1)).mean(dim=(2, 3))
    (y64 * coef.double().cpu()).sum().backward()
    assert th.allclose(out.double().cpu(), y64.detach(), rtol=1e-4, atol=1e-5)
    assert th.allclose(wr.grad.double().cpu(), w64.grad.permute(0, 2, 3, 1), rtol=2e-4, atol=2e-6)
    assert th.allclose(br.grad.double().cpu(), b64.grad, rtol=2e-4, atol=2e-6)
    assert th.allclose(xr.grad.double().cpu(), x64.grad.permute(0, 2, 3, 1), rtol=2e-4, atol=2e-6)


@pytest.mark.parametrize("B,HW,C", [(3, 35, 32), (2, 7056, 32), (5, 9, 8)])
def test_avgpool_relu_backward_equals_its_two_passes_bit_for_bit(B, HW, C):
    g = th.Generator().manual_seed(HW)
    y = th.randn(B, HW, C, generator=g).cuda()
    dout = th.randn(B, C, generator=g).cuda()
    dy = th.empty_like(y)
    L.call("ia_avgpool_nhwc_backward", L.ptr(dout), B, HW, C, L.ptr(dy), L.stream())
    want = th.empty_like(y)
    L.call("ia_relu_backward", L.ptr(dy), L.ptr(y), y.numel(), L.ptr(want), L.stream())
    got = th.full_like(y, float("nan"))
    L.call("ia_avgpool_relu_backward", L.ptr(dout), L.ptr(y), B, HW, C, L.ptr(got), L.stream())
    assert th.equal(got, want)


@pytest.mark.parametrize("B,H,W", [(1, 1, 1), (3, 5, 7), (2, 13, 16), (4, 25, 31), (2, 84, 84), (1030, 3, 4)])
def test_conv3x3_c4_forward_and_gradient_match_float64_autograd(B, H, W):
    """The reward CNN's first convolution straight from the 4-channel rows (no column matrix), through the public op."""
    g = th.Generator().manual_seed(7 * B + H + W)
    x = th.randn(B, H, W, 4, generator=g).cuda()
    w = (0.3 * th.randn(32, 3, 3, 4, generator=g)).cuda()
    b = (0.3 * th.randn(32, generator=g)).cuda()
    wr, br = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    y = ops.conv2d_nhwc(x, wr, br, 1, 1, relu=True)
    coef = th.randn(B, H, W, 32, generator=g).cuda()
    (y * coef).sum().backward()
    w64 = w.double().cpu().permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    b64 = b.double().cpu().requires_grad_(True)
    y64 = th.relu(th.nn.functional.conv2d(x.double().cpu().permute(0, 3, 1, 2), w64, b64, padding=1))
    (y64 * coef.double().cpu().permute(0, 3, 1, 2)).sum().backward()
    assert th.allclose(y.double().cpu(), y64.detach().permute(0, 2, 3, 1), rtol=1e-5, atol=1e-5)
    tol = 2e-5 * max(1.0, np.sqrt(B * H * W / 64.0))
    assert float((wr.grad.double().cpu() - w64.grad.permute(0, 2, 3, 1)).abs().max()) <= tol * (float(w64.grad.abs().max()) + 1e-12)
    assert float((br.grad.double().cpu() - b64.grad).abs().max()) <= tol * (float(b64.grad.abs().max()) + 1e-12)


def test_rollout_tiles_carry_image_frames_as_uint8_and_fill_the_fp32_tiles():
    """`RolloutBuffer(obs_u8=True)`: the pinned tiles the env loop writes and their twins in the transfer block are uint8, the
    fp32 device tiles every consumer reads are filled behind the one copy -- the same values as the fp32 transport, exactly."""
    from imitation_amd.ppo import RolloutBuffer
    T, n, D = 3, 5, 4 * 6 * 6
    rng = np.random.default_rng(0)
    frames = rng.integers(0, 256, (T + 1, n, D), dtype=np.uint8)
    nxt = rng.integers(0, 256, (T, n, D), dtype=np.uint8)
    outs = []
    for u8 in (True, False):
        rb = RolloutBuffer(T, n, D, 1, th.device("cuda"), obs_u8=u8)
        assert rb.h_obs.dtype == (th.uint8 if u8 else th.float32) and rb.obs.dtype == th.float32
        assert rb.obs.shape == (T + 1, n, D) and rb.next_fixed.shape == (T, n, D)
        rb.h_obs.numpy()[...] = frames
        rb.h_next.numpy()[...] = nxt
        rb.h_dones.numpy()[...] = 1
        rb.upload_host_tiles()
        th.cuda.synchronize()
        outs.append((rb.obs.clone(), rb.next_fixed.clone(), rb.dones.clone()))
    for a, b in zip(*outs):
        assert th.equal(a, b)
    assert th.equal(outs[0][0].cpu(), th.as_tensor(frames).float())


@pytest.mark.parametrize("B,H,W", [(1, 1, 1), (2, 5, 7), (3, 9, 16), (2, 17, 31), (2, 84, 84), (5, 8, 90)])
def test_conv3x3_c32_forward_and_input_gradient_match_float64_autograd(B, H, W):
    """The 32 -> 32 layer's forward and input gradient on `conv3x3_c32_conv_kernel` (bands of 8 output rows, partial last band,
    odd widths), through a two-layer chain of the public op so that the ReLU mask in the input gradient's epilogue is used."""
    g = th.Generator().manual_seed(11 * B + H + W)
    x = th.randn(B, H, W, 32, generator=g).cuda()
    w1 = (0.1 * th.randn(32, 3, 3, 32, generator=g)).cuda()
    w2 = (0.1 * th.randn(32, 3, 3, 32, generator=g)).cuda()
    b1, b2 = (0.1 * th.randn(32, generator=g)).cuda(), (0.1 * th.randn(32, generator=g)).cuda()
    xr = x.clone().requires_grad_(True)
    w1r, w2r = w1.clone().requires_grad_(True), w2.clone().requires_grad_(True)
    h = ops.conv2d_nhwc(xr, w1r, b1, 1, 1, relu=True, dy_is_masked=True)
    y = ops.conv2d_nhwc(h, w2r, b2, 1, 1, relu=True, x_is_relu=True)
    coef = th.randn(B, H, W, 32, generator=g).cuda()
    (y * coef).sum().backward()
    x64 = x.double().cpu().permute(0, 3, 1, 2).requires_grad_(True)
    w164 = w1.double().cpu().permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    w264 = w2.double().cpu().permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    h64 = th.relu(th.nn.functional.conv2d(x64, w164, b1.double().cpu(), padding=1))
    y64 = th.relu(th.nn.functional.conv2d(h64, w264, b2.double().cpu(), padding=1))
    (y64 * coef.double().cpu().permute(0, 3, 1, 2)).sum().backward()
    assert th.allclose(y.double().cpu(), y64.detach().permute(0, 2, 3, 1), rtol=2e-5, atol=2e-5)
    tol = 3e-5 * max(1.0, np.sqrt(B * H * W / 64.0))
    for got, want in ((xr.grad, x64.grad), (w1r.grad, w164.grad), (w2r.grad, w264.grad)):
        want = want.permute(0, 2, 3, 1)
        assert float((got.double().cpu() - want).abs().max()) <= tol * (float(want.abs().max()) + 1e-12)
