"""Opt-in discriminator gradient penalty (`imitation_amd/grad_penalty.py`), `-m gpu`. The reference has no gradient
penalty, so the oracle here is torch's double-backward (`create_graph=True`) in float64 on the same
definition; with the coefficient at its default 0 nothing changes (every other parity test runs that way)."""
import numpy as np
import pytest
import torch as th

from tests import harness

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not th.cuda.is_available():
        pytest.skip("no GPU")


@pytest.mark.parametrize("dims,B,norm", [((23, 64, 32, 1), 200, True), ((23, 256, 256, 1), 1024, True),
                                         ((6, 32, 1), 77, False)])
def test_penalty_and_parameter_gradient_match_torch_double_backward(dims, B, norm):
    from imitation_amd import _lib as L, grad_penalty

    g = th.Generator().manual_seed(0)
    D = dims[0]
    n = sum(i * j + j for i, j in zip(dims[:-1], dims[1:]))
    flat = th.randn(n, generator=g) * 0.3
    X = th.randn(2 * B, D, generator=g) * 1.5 + 0.2
    e = th.rand(B, generator=g)
    mean, var = (th.randn(D, generator=g) * 0.1, th.rand(D, generator=g) + 0.5) if norm else (None, None)
    coef, target, eps = 7.0, 1.0, 1e-5
    # ---- torch float64 reference with a double-backward graph
    fr = flat.double().requires_grad_(True)
    xh = (e.double()[:, None] * X[:B].double() + (1 - e.double()[:, None]) * X[B:].double()).requires_grad_(True)
    h = (xh - mean.double()) / th.sqrt(var.double() + eps) if norm else xh
    o = 0
    for li, (i, j) in enumerate(zip(dims[:-1], dims[1:])):
        W = fr[o:o + i * j].view(j, i); o += i * j
        b = fr[o:o + j]; o += j
        h = h @ W.T + b
        if li < len(dims) - 2:
            h = th.relu(h)
    (gx,) = th.autograd.grad(h.sum(), xh, create_graph=True)
    pen_rows = (gx.norm(dim=1) - target) ** 2
    (coef * pen_rows.mean()).backward()
    # ---- HIP
    ld = (D + 3) // 4 * 4
    Xd = th.zeros(2 * B, ld, device=DEV)
    Xd[:, :D] = X.to(DEV)
    pen, gflat = grad_penalty.penalty_and_param_grad(flat.to(DEV), dims, L.ACT_RELU, Xd, ld, B, e.to(DEV),
                                                     None if mean is None else mean.to(DEV),
                                                     None if var is None else var.to(DEV), eps, coef, target)
    th.testing.assert_close(pen.cpu().double(), pen_rows.mean().detach(), rtol=2e-5, atol=1e-6)
    scale = float(fr.grad.abs().max())
    th.testing.assert_close(gflat.cpu().double(), fr.grad, rtol=5e-4, atol=2e-5 * scale)


def test_trainer_with_gradient_penalty_state_holder_and_module_paths_agree(tmp_path):
    """GAIL with `disc_grad_penalty_coef > 0`: the fused state-holder path and the `nn.Module` path draw the same
    interpolation weights from torch's generator and must end with the same parameters; the penalty term moves
    the parameters (vs coefficient 0) and everything stays finite."""
    import imitation_amd as p

    cfg = dict(harness.CASES["gail_box"])
    outs = {}
    for name, module_net, coef in (("holder", False, 5.0), ("module", True, 5.0), ("off", False, 0.0)):
        tr, _ = harness.build_trainer("hip", cfg, str(tmp_path / name), device="cuda", module_net=module_net)
        tr.disc_grad_penalty_coef = coef
        tr.train_gen()
        stats = [tr.train_disc() for _ in range(5)]
        outs[name] = ({k: v.detach().cpu().numpy().copy() for k, v in tr._reward_net.state_dict().items()}, stats,
                      None if tr.last_grad_penalty is None else float(tr.last_grad_penalty))
    a, b, off = outs["holder"], outs["module"], outs["off"]
    assert off[2] is None and a[2] is not None and np.isfinite(a[2]) and a[2] >= 0
    for k in a[0]:
        assert np.isfinite(a[0][k]).all(), k
        if k.endswith("count"):
            assert np.array_equal(a[0][k], b[0][k]), k
        else:
            np.testing.assert_allclose(b[0][k], a[0][k], rtol=2e-4, atol=5e-5, err_msg=k)
    assert any(not np.allclose(a[0][k], off[0][k]) for k in a[0] if "weight" in k), "the penalty changed nothing"
    np.testing.assert_allclose(a[2], b[2], rtol=1e-3)


def test_gradient_penalty_on_the_fused_shape_and_unsupported_nets(tmp_path):
    import imitation_amd as p

    cfg = dict(harness.CASES["gail_fused"])
    tr, _ = harness.build_trainer("hip", cfg, str(tmp_path / "f"), device="cuda")
    tr.disc_grad_penalty_coef = 1.0
    tr.train(2 * cfg["n_envs"] * cfg["n_steps"])        # pipelined rounds fall back to per-update assembly
    assert np.isfinite(float(tr.last_grad_penalty))
    assert all(bool(th.isfinite(v.float()).all()) for v in tr._reward_net.state_dict().values())
    at, _ = harness.build_trainer("hip", harness.CASES["airl_box"], str(tmp_path / "a"), device="cuda")
    at.disc_grad_penalty_coef = 1.0
    at.train_gen()
    with pytest.raises(NotImplementedError, match="BasicRewardNet"):
        at.train_disc()
