"""The PyTorch-ROCm custom-op boundary (`imitation_amd.ops`, `imitation_amd.modules`), `-m gpu`:
the `imitation_amd::*` torch.library ops and their autograd functions against torch's own autograd
(float64) on the same inputs, and `nn.Module` reward nets trained by `loss.backward()` through them
against (a) the fused state-holder path and (b) the reference's golden runs.

Tolerances: op-level `rtol 2e-5, atol 2e-5 * sqrt(K)` (fp32 MFMA FMA chains vs float64); end-to-end
the adversarial tests' `rtol 2e-4 / atol 5e-5`."""
import math
import os

import numpy as np
import pytest
import torch as th

from tests import harness

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
DEV = "cuda"


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not th.cuda.is_available():
        pytest.skip("no GPU")


def test_ops_are_registered_with_torch_library():
    from imitation_amd import ops  # noqa: F401

    for name in ("mlp_forward", "mlp_backward", "running_norm_update", "running_norm_apply", "bce_expert_first",
                 "gather_rows", "adam_step"):
        assert hasattr(th.ops.imitation_amd, name), name
    with pytest.raises((NotImplementedError, RuntimeError)):   # no CPU kernels: fails loudly
        th.ops.imitation_amd.running_norm_apply(th.zeros(4, 3), th.zeros(3), th.ones(3), 1e-5)


@pytest.mark.parametrize("dims,act,R", [((23, 256, 256, 1), "relu", 4096), ((6, 32, 32, 1), "tanh", 333),
                                        ((11, 64, 3), "relu", 100)])
def test_mlp_autograd_matches_torch(dims, act, R):
    from imitation_amd import ops

    g = th.Generator().manual_seed(0)
    n = sum(i * j + j for i, j in zip(dims[:-1], dims[1:]))
    flat = (th.randn(n, generator=g) * 0.2)
    x = th.randn(R, dims[0], generator=g)
    w_out = th.randn(R, dims[-1], generator=g)
    # torch float64 reference
    xr, fr = x.double().requires_grad_(True), flat.double().requires_grad_(True)
    h, o = xr, 0
    fn = th.relu if act == "relu" else th.tanh
    for li, (i, j) in enumerate(zip(dims[:-1], dims[1:])):
        W = fr[o:o + i * j].view(j, i); o += i * j
        b = fr[o:o + j]; o += j
        h = h @ W.T + b
        if li < len(dims) - 2:
            h = fn(h)
    (h * w_out.double()).sum().backward()
    # HIP
    xd, fd = x.to(DEV).requires_grad_(True), flat.to(DEV).requires_grad_(True)
    out = ops.mlp(xd, fd, dims, ops.ACT_RELU if act == "relu" else ops.ACT_TANH)
    (out * w_out.to(DEV)).sum().backward()
    K = max(dims)
    tol = dict(rtol=2e-5, atol=2e-5 * math.sqrt(K))
    th.testing.assert_close(out.detach().cpu().double(), h.detach(), rtol=2e-5, atol=2e-5 * math.sqrt(K))
    th.testing.assert_close(xd.grad.cpu().double(), xr.grad, **tol)
    th.testing.assert_close(fd.grad.cpu().double(), fr.grad, rtol=3e-5, atol=3e-5 * math.sqrt(R))


def test_running_norm_module_and_bce_match_torch():
    from imitation_amd import modules, ops

    g = th.Generator().manual_seed(1)
    rn = modules.RunningNorm(7).to(DEV)
    mean, M2, cnt = th.zeros(7, dtype=th.float64), th.zeros(7, dtype=th.float64), 0
    allx = []
    for k in range(3):
        x = th.randn(50 + 13 * k, 7, generator=g) * (1 + k) + k
        allx.append(x)
        xd = x.to(DEV).requires_grad_(True)
        y = rn(xd)                                           # train mode: update, then normalise
        cat = th.cat(allx).double()
        mu, var = cat.mean(0), cat.var(0, unbiased=False)
        th.testing.assert_close(rn.running_mean.cpu().double(), mu, rtol=1e-5, atol=1e-5)
        th.testing.assert_close(rn.running_var.cpu().double(), var, rtol=1e-5, atol=1e-5)
        assert int(rn.count) == len(cat)
        th.testing.assert_close(y.detach().cpu().double(), (x.double() - mu) / th.sqrt(var + 1e-5), rtol=1e-4, atol=1e-5)
        y.sum().backward()                                   # statistics are constants for autograd
        th.testing.assert_close(xd.grad.cpu().double(), (1 / th.sqrt(var + 1e-5)).expand(len(x), 7), rtol=1e-5, atol=1e-6)
    rn.eval()
    before = rn.running_mean.clone()
    rn(th.randn(5, 7, device=DEV))
    assert th.equal(before, rn.running_mean)
    # BCE with the [expert | generator] label layout
    logits = th.randn(301, generator=g) * 3
    ne = 120
    ld = logits.to(DEV).requires_grad_(True)
    loss, stats = ops.bce_expert_first(ld, ne, 0.5)
    (loss * 2.0).backward()
    lr = logits.double().requires_grad_(True)
    y = th.cat([th.ones(ne), th.zeros(301 - ne)]).double()
    ref = th.nn.functional.binary_cross_entropy_with_logits(lr, y) * 0.5
    (ref * 2.0).backward()
    th.testing.assert_close(loss.cpu().double(), ref.detach(), rtol=1e-5, atol=1e-6)
    th.testing.assert_close(ld.grad.cpu().double(), lr.grad, rtol=1e-5, atol=1e-8)
    s = stats.cpu().numpy()
    pred_gen = (logits < 0).numpy()
    assert s[4] == pred_gen.sum() and s[6] == ne and s[7] == 301 - ne
    assert s[1] == (pred_gen == (y.numpy() == 0)).sum()


def test_hip_adam_optimizer_matches_torch_adam():
    from imitation_amd import ops

    th.manual_seed(0)
    ps = [th.randn(17, 5), th.randn(5)]
    a = [p.clone().to(DEV).requires_grad_(True) for p in ps]
    b = [p.clone().double().requires_grad_(True) for p in ps]
    oa, ob = ops.HipAdam(a, lr=3e-3, weight_decay=1e-2), th.optim.Adam(b, lr=3e-3, weight_decay=1e-2)
    for k in range(5):
        for opt, params in ((oa, a), (ob, b)):
            opt.zero_grad()
            loss = sum(((p * (k + 1)).sin() ** 2).sum() for p in params)
            loss.backward()
            opt.step()
    for x, y in zip(a, b):
        th.testing.assert_close(x.detach().cpu().double(), y.detach(), rtol=1e-5, atol=1e-6)


def test_module_reward_net_train_disc_matches_fused_path(tmp_path):
    """One `train_gen` + several `train_disc` through `modules.BasicRewardNet` (loss.backward() on the custom ops,
    `ops.HipAdam`) against the fused `ia_disc_step_basic` path on the same seeds: identical batches (host RNG
    order), statistics / parameters / norm buffers within the end-to-end tolerance; state dicts interchange."""
    cfg = dict(harness.CASES["gail_box"], demo_minibatch=None)
    outs = {}
    for module_net in (False, True):
        tr, _ = harness.build_trainer("hip", cfg, str(tmp_path / f"m{module_net}"), device="cuda", module_net=module_net)
        assert tr._module_net == module_net
        tr.train_gen()
        stats = [tr.train_disc() for _ in range(4)]
        sd = {k: v.detach().cpu().numpy().copy() for k, v in tr._reward_net.state_dict().items()}
        outs[module_net] = (stats, sd, tr)
    (sa, da, ta), (sb, db, tb) = outs[False], outs[True]
    assert set(da) == set(db)
    for k in da:
        if k.endswith("count"):
            assert np.array_equal(da[k], db[k]), k
        else:
            np.testing.assert_allclose(db[k], da[k], rtol=2e-4, atol=5e-5, err_msg=k)
    for x, y in zip(sa, sb):
        for k in x:
            np.testing.assert_allclose(y[k], x[k], rtol=2e-4, atol=5e-5, err_msg=k)
    # interchange: the module's state dict loads into the state-holder net and vice versa
    ta._reward_net.load_state_dict(tb._reward_net.state_dict())
    tb._reward_net.load_state_dict({k: th.as_tensor(v) for k, v in da.items()})
    # the public RewardFn surface agrees
    rng = np.random.default_rng(3)
    s = rng.standard_normal((16, cfg["obs_dim"])).astype(np.float32)
    a = rng.uniform(-1, 1, (16, cfg["act_dim"])).astype(np.float32)
    tb._reward_net.load_state_dict(ta._reward_net.state_dict())
    np.testing.assert_allclose(tb.reward_train.predict_processed(s, a, s, np.zeros(16, bool)),
                               ta.reward_train.predict_processed(s, a, s, np.zeros(16, bool)), rtol=2e-4, atol=5e-5)


@pytest.mark.parametrize("case", ["gail_box", "gail_discrete", "airl_box", "airl_ema"])
def test_module_reward_net_trainer_matches_reference_golden(case, tmp_path):
    """Full GAIL / AIRL runs with `nn.Module` reward nets (autograd through the HIP ops, gradient accumulation
    over minibatches by repeated `backward()`, per-step `predict_processed` relabelling) against the
    reference's own golden runs -- the same comparison the fused path passes."""
    cfg = harness.CASES[case]
    gold = dict(np.load(os.path.join(GOLDEN, f"{case}.npz")))
    got = harness.run_case("hip", case, str(tmp_path), device="cuda", module_net=True)
    assert set(got) == set(gold), set(got) ^ set(gold)
    for key in gold:
        x, y = np.asarray(got[key]), np.asarray(gold[key])
        assert x.shape == y.shape, (key, x.shape, y.shape)
        if key in harness.EXACT_KEYS or y.dtype.kind in "biu":
            assert np.array_equal(x, y), key
        else:
            np.testing.assert_allclose(x.astype(np.float64), y.astype(np.float64), rtol=2e-4, atol=5e-5,
                                       equal_nan=True, err_msg=key)


def test_user_defined_reward_net_plugs_in(tmp_path):
    """A reward net the framework has never seen -- a user's `modules.RewardNet` subclass mixing an HIP-op MLP
    with plain torch-on-ROCm ops -- trains through `GAIL.train()`: the plugin contract of
    `rewards/reward_nets.py:16-50`."""
    import imitation_amd as p
    from imitation_amd import modules

    class TwoHeadNet(modules.RewardNet):
        def __init__(self, osp, asp):
            super().__init__(osp, asp)
            d = int(np.prod(osp.shape)) + int(np.prod(asp.shape))
            self.trunk = modules.Mlp(d, (32,), out_size=8)
            self.head = th.nn.Linear(8, 1)   # ordinary torch layer: rocBLAS-free tiny matvec via autograd

        def forward(self, state, action, next_state, done):
            z = th.tanh(self.trunk(th.cat([state.flatten(1), action.flatten(1)], 1)))
            return self.head(z).squeeze(-1) - 0.1 * done

    cfg = harness.CASES["gail_box"]
    th.manual_seed(0)
    np.random.seed(0)
    from imitation_amd.vec_env import SyntheticVecEnv
    venv = SyntheticVecEnv(num_envs=8, obs_dim=cfg["obs_dim"], act_dim=cfg["act_dim"], horizon=10, seed=0)
    algo = p.PPO(p.FeedForward32Policy, venv, n_steps=16, batch_size=32, n_epochs=2, seed=0, device="cuda")
    net = TwoHeadNet(venv.observation_space, venv.action_space)
    tr = p.GAIL(demonstrations=p.Transitions(**harness.make_demo_arrays(cfg)), demo_batch_size=64, venv=venv,
                gen_algo=algo, reward_net=net, n_disc_updates_per_round=3, disc_opt_cls=th.optim.SGD,
                disc_opt_kwargs=dict(lr=0.05), custom_logger=p.configure_logger(str(tmp_path), []))
    before = {k: v.detach().clone() for k, v in net.state_dict().items()}
    tr.train(3 * 8 * 16)
    after = net.state_dict()
    assert all(th.isfinite(v).all() for v in after.values())
    assert any(not th.equal(before[k], after[k]) for k in before), "the optimiser never moved the plugged-in net"
    assert tr._disc_step == 9
    losses = [tr.train_disc()["disc_loss"] for _ in range(30)]
    assert np.mean(losses[-5:]) < np.mean(losses[:5])


def test_cnn_reward_net_matches_reference_golden_and_torch_autograd():
    """`modules.CnnRewardNet` (padded 3x3 convolutions as im2col + MFMA GEMM, global average pool, linear head,
    one-hot action / done selection) against (a) the reference's own `CnnRewardNet` outputs for the same weights
    (`tests/golden/cnn_reward_net.npz`, made by the reference under the shim) and (b) torch autograd (float64
    `nn.Conv2d` stack) for every parameter gradient and the input gradient."""
    from imitation_amd import modules, ops, spaces

    g = np.load(os.path.join(GOLDEN, "cnn_reward_net.npz"))
    osp, asp = spaces.Box(0, 255, (12, 10, 3), np.uint8), spaces.Discrete(5)
    net = modules.CnnRewardNet(osp, asp, use_next_state=True, use_done=True).to(DEV)
    net.load_state_dict({k[3:]: th.as_tensor(g[k]) for k in g.files if k.startswith("sd/")})
    rews = net.predict_processed(g["obs"], g["acts"], g["next_obs"], g["dones"])
    np.testing.assert_allclose(rews, g["rews"], rtol=2e-5, atol=2e-6)

    # (b) gradients of a generic stack: 5 -> 16 -> 8 channels, 4 outputs, 9 x 7 images, batch 6
    th.manual_seed(0)
    cnn = modules.Cnn(5, (16, 8), out_size=4).to(DEV)
    x = th.randn(6, 5, 9, 7)
    wsum = th.randn(6, 4)
    ref = th.nn.Sequential(th.nn.Conv2d(5, 16, 3, padding="same"), th.nn.ReLU(), th.nn.Conv2d(16, 8, 3, padding="same"),
                           th.nn.ReLU(), th.nn.AdaptiveAvgPool2d(1), th.nn.Flatten(), th.nn.Linear(8, 4)).double()
    sd = cnn.state_dict()
    with th.no_grad():
        for src, dst in (("conv0", ref[0]), ("conv1", ref[2]), ("dense_final", ref[6])):
            dst.weight.copy_(sd[f"{src}.weight"].double().cpu())
            dst.bias.copy_(sd[f"{src}.bias"].double().cpu())
    xr = x.double().requires_grad_(True)
    (ref(xr) * wsum.double()).sum().backward()
    xd = x.to(DEV).requires_grad_(True)
    out = cnn(xd)
    (out * wsum.to(DEV)).sum().backward()
    th.testing.assert_close(out.detach().cpu().double(), ref(x.double()).detach(), rtol=2e-5, atol=2e-5)
    th.testing.assert_close(xd.grad.cpu().double(), xr.grad, rtol=5e-5, atol=2e-5)
    for src, dst in (("conv0", ref[0]), ("conv1", ref[2]), ("dense_final", ref[6])):
        th.testing.assert_close(getattr(cnn, src).weight.grad.cpu().double(), dst.weight.grad, rtol=5e-5, atol=5e-5)
        th.testing.assert_close(getattr(cnn, src).bias.grad.cpu().double(), dst.bias.grad, rtol=5e-5, atol=5e-5)


def test_gail_discriminator_on_image_observations_trains(tmp_path):
    """`train_disc` of a GAIL trainer whose reward net is the image `CnnRewardNet` (explicit expert / generator
    image batches, uint8 frames): loss.backward() through the convolution ops, discriminator loss decreases
    (`tests/algorithms/test_adversarial.py:256-282` of the reference, on images)."""
    import imitation_amd as p
    from imitation_amd import modules, spaces
    from imitation_amd.vec_env import SyntheticVecEnv

    th.manual_seed(0)
    np.random.seed(0)
    osp, asp = spaces.Box(0, 255, (10, 10, 2), np.uint8), spaces.Discrete(3)

    class ImageEnv(SyntheticVecEnv):   # only the spaces matter here: train_disc gets explicit samples
        pass

    venv = SyntheticVecEnv(num_envs=4, obs_dim=8, act_dim=3, horizon=5, seed=0, n_discrete=3)
    venv.observation_space = osp
    rng = np.random.default_rng(0)

    def batch(bright):
        obs = rng.integers(0, 128, (32, 10, 10, 2)).astype(np.uint8) + (100 if bright else 0)
        return dict(obs=obs, acts=rng.integers(0, 3, 32), next_obs=obs.copy(), dones=np.zeros(32, bool))

    algo = p.PPO(p.FeedForward32Policy, SyntheticVecEnv(num_envs=4, obs_dim=8, act_dim=3, horizon=5, seed=0, n_discrete=3),
                 n_steps=4, batch_size=8, seed=0, device="cuda")
    net = modules.CnnRewardNet(osp, asp, hid_channels=(8, 8))
    demos = p.Transitions(obs=np.zeros((64, 8), np.float32), acts=np.zeros(64, np.int64),
                          next_obs=np.zeros((64, 8), np.float32), dones=np.zeros(64, bool))
    tr = p.GAIL(demonstrations=demos, demo_batch_size=32, venv=algo.get_env(), gen_algo=algo, reward_net=net,
                disc_opt_kwargs=dict(lr=1e-2), custom_logger=p.configure_logger(str(tmp_path), []))
    tr.venv = venv   # spaces of the image task for batch assembly (one-hot width, observation shape)
    losses = [tr.train_disc(expert_samples=batch(True), gen_samples=batch(False))["disc_loss"] for _ in range(60)]
    assert np.isfinite(losses).all() and np.mean(losses[-5:]) < 0.8 * np.mean(losses[:5]), losses


def test_mlp_module_with_dropout_runs_layer_by_layer():
    """`build_mlp(dropout_prob > 0)` (`util/networks.py:210,270-271`): eval mode equals the dropout-free stack,
    train mode drops units (different outputs for the same input) and trains through `backward()`."""
    from imitation_amd import modules

    th.manual_seed(0)
    net = modules.Mlp(7, (32, 32), out_size=1, dropout_prob=0.5, squeeze_output=True).to(DEV)
    ref = modules.Mlp(7, (32, 32), out_size=1, squeeze_output=True).to(DEV)
    ref.load_state_dict(net.state_dict())
    x = th.randn(64, 7, device=DEV)
    net.eval()
    th.testing.assert_close(net(x), ref(x), rtol=1e-5, atol=1e-6)
    net.train()
    a, b = net(x), net(x)
    assert not th.allclose(a, b)
    a.sum().backward()
    assert all(p_.grad is not None and th.isfinite(p_.grad).all() for p_ in net.parameters())


@pytest.mark.parametrize("B,H,W,Cin,Cout,k,s,p,relu", [
    (96, 84, 84, 32, 32, 3, 1, 1, True),     # 677 376 output rows: the view's row decode must be exact far past 2^16 rows
    (5, 20, 20, 32, 64, 4, 2, 0, True),      # NatureCNN layer 2 geometry
    (3, 9, 11, 64, 64, 3, 1, 1, False),      # non-square, padded
    (4, 12, 12, 4, 16, 3, 1, 1, True),       # Cin = 4: (KW * Cin) % 32 != 0 -> explicit column buffer path
])
def test_conv2d_nhwc_op_matches_torch(B, H, W, Cin, Cout, k, s, p, relu):
    """`ops.conv2d_nhwc` (implicit im2col view when Cin % 4 == 0 and KW*Cin % 32 == 0, explicit column buffer otherwise)
    against `torch.nn.functional.conv2d` autograd: output, input gradient, weight and bias gradients."""
    from imitation_amd import ops
    g = th.Generator().manual_seed(B + Cin)
    x = th.randn(B, H, W, Cin, generator=g).cuda().requires_grad_()
    w = (th.randn(Cout, k, k, Cin, generator=g) / np.sqrt(k * k * Cin)).cuda().requires_grad_()
    b = (0.1 * th.randn(Cout, generator=g)).cuda().requires_grad_()
    assert ops.conv_is_implicit(Cin, k, x.numel()) == (Cin % 4 == 0 and (k * Cin) % 32 == 0)
    y = ops.conv2d_nhwc(x, w, b, stride=s, pad=p, relu=relu)
    up = th.randn(y.shape, generator=g).cuda()
    (y * up).sum().backward()
    xr, wr, br = (t.detach().clone().requires_grad_() for t in (x, w, b))
    yr = th.nn.functional.conv2d(xr.permute(0, 3, 1, 2), wr.permute(0, 3, 1, 2), br, stride=s, padding=p)
    yr = (th.relu(yr) if relu else yr).permute(0, 2, 3, 1)
    (yr * up).sum().backward()
    for name, a_, r_ in (("y", y, yr), ("dx", x.grad, xr.grad), ("dw", w.grad, wr.grad), ("db", b.grad, br.grad)):
        scale = float(r_.abs().max()) + 1e-12
        assert float((a_ - r_).abs().max()) <= 5e-5 * scale, (name, float((a_ - r_).abs().max()), scale)


def test_dropout_net_and_tensorboard_flag_train_through_gail(tmp_path):
    """What used to raise now runs: a `BasicRewardNet(dropout_prob > 0)` built through the state-holder constructor is the
    `nn.Module` net and trains through `GAIL.train` (module path: `loss.backward()` on the HIP ops); `init_tensorboard=True`
    (`common.py:223-227`) warns and trains on."""
    import imitation_amd as p
    from imitation_amd import modules
    from imitation_amd.vec_env import SyntheticVecEnv

    cfg = harness.CASES["gail_box"]
    th.manual_seed(0)
    np.random.seed(0)
    venv = SyntheticVecEnv(num_envs=cfg["n_envs"], obs_dim=17, act_dim=6, horizon=cfg["horizon"], seed=0)
    algo = p.PPO(p.FeedForward32Policy, venv, n_steps=cfg["n_steps"], batch_size=cfg["ppo_batch"], n_epochs=2, seed=0,
                 device="cuda")
    net = p.BasicRewardNet(venv.observation_space, venv.action_space, hid_sizes=(32, 32), dropout_prob=0.2,
                           normalize_input_layer=p.RunningNorm)
    assert isinstance(net, modules.BasicRewardNet)
    demos = p.Transitions(**harness.make_demo_arrays(cfg))
    with pytest.warns(RuntimeWarning, match="init_tensorboard"):
        tr = p.GAIL(demonstrations=demos, demo_batch_size=64, venv=venv, gen_algo=algo, reward_net=net,
                    n_disc_updates_per_round=2, custom_logger=p.configure_logger(str(tmp_path), []), init_tensorboard=True)
    before = {k: v.detach().clone() for k, v in net.state_dict().items()}
    tr.train(2 * cfg["n_envs"] * cfg["n_steps"])
    th.cuda.synchronize()
    after = net.state_dict()
    assert any(not th.equal(before[k], after[k]) for k in before if k.endswith("weight"))
    assert all(bool(th.isfinite(v.float()).all()) for v in after.values())
