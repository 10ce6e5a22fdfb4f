"""General-tower actor-critic policies (`imitation_amd/general_policy.py`, `csrc/ppo_general.hip`): the head kernels
against torch (float64 autograd of [SB3 ppo.py] PPO.train's loss), and policy-level agreement of a general-tower
policy with the fused kernels on a shape both cover. The end-to-end evidence is the reference's golden runs
`gail_towers`, `gail_discrete_towers`, `airl_towers` (tests/test_adversarial_gpu.py)."""
import ctypes as C

import numpy as np
import pytest
import torch as th
from torch import nn

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _need_gpu():
    if not th.cuda.is_available():
        pytest.skip("no GPU")


def _torch_loss(discrete, out, log_std, values, actions, old, adv, ret, normalize, clip, ent_coef, vf_coef):
    """[SB3 ppo.py] PPO.train loss of one minibatch in float64 with autograd."""
    if normalize:
        adv = (adv - adv.mean()) / (adv.std() + 1e-8)
    if discrete:
        dist = th.distributions.Categorical(logits=out)
        logp, ent = dist.log_prob(actions.long()), dist.entropy()
    else:
        dist = th.distributions.Normal(out, th.ones_like(out) * log_std.exp())
        logp, ent = dist.log_prob(actions).sum(1), dist.entropy().sum(1)
    ratio = th.exp(logp - old)
    pl1, pl2 = adv * ratio, adv * th.clamp(ratio, 1 - clip, 1 + clip)
    pg = -th.min(pl1, pl2).mean()
    vl = th.nn.functional.mse_loss(ret, values)
    el = -ent.mean()
    loss = pg + ent_coef * el + vf_coef * vl
    with th.no_grad():
        lr = logp - old
        kl = ((th.exp(lr) - 1) - lr).mean()
        cf = ((ratio - 1).abs() > clip).double().mean()
    return loss, th.stack([pg, vl, el, kl, cf, loss]).detach()


@pytest.mark.parametrize("discrete,B,A,normalize", [(False, 700, 6, True), (False, 33, 1, False), (True, 515, 5, True),
                                                    (True, 7, 2, True)])
def test_head_loss_matches_torch_autograd(discrete, B, A, normalize):
    from imitation_amd import _lib as L
    g = th.Generator().manual_seed(B + A)
    out = th.randn(B, A, generator=g, dtype=th.float64)
    log_std = th.randn(A, generator=g, dtype=th.float64) * 0.3
    values, ret = th.randn(B, generator=g, dtype=th.float64), th.randn(B, generator=g, dtype=th.float64)
    adv = th.randn(B, generator=g, dtype=th.float64) * 2 + 0.3
    if discrete:
        actions = th.randint(0, A, (B,), generator=g).double()
        old = th.distributions.Categorical(logits=out).log_prob(actions.long()) + 0.3 * th.randn(B, generator=g, dtype=th.float64)
    else:
        actions = out + th.randn(B, A, generator=g, dtype=th.float64)
        old = (th.distributions.Normal(out, log_std.exp().expand_as(out)).log_prob(actions).sum(1)
               + 0.3 * th.randn(B, generator=g, dtype=th.float64))
    clip, ent_coef, vf_coef = 0.2, 0.03, 0.5
    o, ls, v = out.clone().requires_grad_(), log_std.clone().requires_grad_(), values.clone().requires_grad_()
    loss, st_ref = _torch_loss(discrete, o, ls, v, actions, old, adv, ret, normalize, clip, ent_coef, vf_coef)
    loss.backward()

    dev = "cuda"
    f = lambda t: t.float().to(dev).contiguous()  # noqa: E731
    d_out, d_val, dls = th.empty(B, A, device=dev), th.empty(B, device=dev), th.zeros(A, device=dev)
    ms, stats = th.empty(2, device=dev), th.empty(8, device=dev)
    ws = th.empty(int(L.load().ia_ppo_head_loss_ws_floats(B)), device=dev)
    adv_d = f(adv)
    if normalize:
        L.call("ia_adv_moments", L.ptr(adv_d), B, L.ptr(ms), L.stream())
        np.testing.assert_allclose(ms.cpu().numpy(), [adv.mean().item(), adv.std().item()], rtol=1e-5)
    out_d, ls_d, val_d, act_d, old_d, ret_d = f(out), f(log_std), f(values), f(actions), f(old), f(ret)  # (kept alive)
    L.call("ia_ppo_head_loss", int(discrete), L.ptr(out_d), L.ptr(ls_d), L.ptr(val_d), L.ptr(act_d),
           L.ptr(old_d), L.ptr(adv_d), L.ptr(ret_d), L.ptr(ms) if normalize else None, B, A, clip, ent_coef, vf_coef,
           L.ptr(d_out), L.ptr(d_val), None if discrete else L.ptr(dls), L.ptr(ws), L.ptr(stats), L.stream())
    tol = dict(rtol=2e-4, atol=2e-6)
    np.testing.assert_allclose(d_out.cpu().double().numpy(), o.grad.numpy(), **tol)
    np.testing.assert_allclose(d_val.cpu().double().numpy(), v.grad.numpy(), **tol)
    if not discrete:
        np.testing.assert_allclose(dls.cpu().double().numpy(), ls.grad.numpy(), rtol=5e-4, atol=1e-5)
    np.testing.assert_allclose(stats[:6].cpu().double().numpy(), st_ref.numpy(), rtol=2e-4, atol=1e-5)


def test_gauss_act_and_clip_grad_norm():
    from imitation_amd import _lib as L
    n, A = 300, 4
    g = th.Generator().manual_seed(1)
    mean, noise = th.randn(n, A, generator=g), th.randn(n, A, generator=g)
    log_std = th.tensor([0.1, -0.4, 0.0, 0.7])
    low, high = -th.ones(A) * 0.8, th.ones(A) * 1.1
    d = lambda t: t.cuda().contiguous()  # noqa: E731
    acts, clip, lp = th.empty(n, A, device="cuda"), th.empty(n, A, device="cuda"), th.empty(n, device="cuda")
    mean_d, ls_d, noise_d, low_d, high_d = d(mean), d(log_std), d(noise), d(low), d(high)
    L.call("ia_gauss_act", L.ptr(mean_d), L.ptr(ls_d), L.ptr(noise_d), L.ptr(low_d), L.ptr(high_d), n, A,
           L.ptr(acts), L.ptr(clip), L.ptr(lp), L.stream())
    ref = mean + log_std.exp() * noise
    dist = th.distributions.Normal(mean, log_std.exp().expand_as(mean))
    np.testing.assert_allclose(acts.cpu().numpy(), ref.numpy(), rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(clip.cpu().numpy(), th.max(th.min(ref, high), low).numpy(), rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(lp.cpu().numpy(), dist.log_prob(ref).sum(1).numpy(), rtol=1e-5, atol=1e-5)
    ent, lp2 = th.empty(n, device="cuda"), th.empty(n, device="cuda")
    L.call("ia_gauss_eval", L.ptr(mean_d), L.ptr(ls_d), L.ptr(acts), n, A, L.ptr(lp2), L.ptr(ent), L.stream())
    np.testing.assert_allclose(lp2.cpu().numpy(), lp.cpu().numpy(), rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(ent.cpu().numpy(), dist.entropy().sum(1).numpy(), rtol=1e-6)

    cws = th.empty(int(L.load().ia_clip_grad_norm_ws_floats()), device="cuda")
    for numel, ws in ((5000, None), (5000, cws), (1_700_003, cws), (65_536, cws)):   # one block / the grid form (NatureCNN length)
        for scale in (0.01 / (numel / 5000) ** 0.5, 10.0):   # below / above the clipping threshold
            gr = th.randn(numel, generator=g) * scale
            gd, norm = d(gr), th.empty(1, device="cuda")
            L.call("ia_clip_grad_norm", L.ptr(gd), gd.numel(), 0.5, L.ptr(norm), None if ws is None else L.ptr(ws), L.stream())
            p = nn.Parameter(th.zeros(numel))
            p.grad = gr.clone()
            total = th.nn.utils.clip_grad_norm_([p], 0.5)
            if numel <= 5000:
                np.testing.assert_allclose(norm.item(), total.item(), rtol=1e-5)
                np.testing.assert_allclose(gd.cpu().numpy(), p.grad.numpy(), rtol=1e-5, atol=1e-9)
            # long vectors: torch-CPU's own float32 accumulation is 2e-5 off at 1.7 M entries -- the float64 norm decides
            total64 = float(gr.double().norm())
            np.testing.assert_allclose(norm.item(), total64, rtol=2e-6)
            want = gr.double() * min(1.0, 0.5 / (total64 + 1e-6))
            np.testing.assert_allclose(gd.cpu().numpy(), want.float().numpy(), rtol=2e-6, atol=1e-9)


@pytest.mark.parametrize("discrete", [False, True])
def test_general_towers_agree_with_fused_kernels_on_a_shared_shape(discrete):
    """[32, 32] tanh towers run on the fused kernels; forced through `general_policy.adopt` the SAME parameters must
    give the same actions / values / log-probs and the same PPO update (both follow [SB3 PPO.train])."""
    import imitation_amd as ia
    from imitation_amd import general_policy, spaces
    from imitation_amd.vec_env import SyntheticVecEnv

    def make(general):
        th.manual_seed(3)
        np.random.seed(3)
        venv = SyntheticVecEnv(num_envs=8, obs_dim=7, act_dim=3, horizon=9, seed=0, n_discrete=3 if discrete else None)
        if general:
            orig = general_policy.fused_arch
            general_policy.fused_arch = lambda *a: False
        try:
            algo = ia.PPO(ia.FeedForward32Policy, venv, n_steps=16, batch_size=32, n_epochs=2, ent_coef=0.02, seed=0,
                          policy_kwargs=dict(features_extractor_class=ia.NormalizeFeaturesExtractor,
                                             features_extractor_kwargs=dict(normalize_class=ia.RunningNorm)),
                          device="cuda")
        finally:
            if general:
                general_policy.fused_arch = orig
        return algo

    a, b = make(False), make(True)
    assert a.policy.fused and not b.policy.fused
    assert th.equal(a.policy._flat, b.policy._flat)
    assert list(a.policy.state_dict()) == list(b.policy.state_dict())
    for algo in (a, b):   # same global torch / NumPy streams (action noise, minibatch permutations) for both runs
        th.manual_seed(11)
        np.random.seed(11)
        algo.learn(16 * 8 * 3)
    for k, v in a.policy.state_dict().items():
        np.testing.assert_allclose(b.policy.state_dict()[k].cpu().numpy(), v.cpu().numpy(), rtol=2e-4, atol=2e-5, err_msg=k)
    obs = np.random.default_rng(0).standard_normal((50, 7)).astype(np.float32)
    acts = a.policy.predict(obs, deterministic=True)[0]
    np.testing.assert_allclose(b.policy.predict(obs, deterministic=True)[0], acts, rtol=1e-3, atol=1e-4)
    va, lpa, ea = a.policy.evaluate_actions(obs, acts)
    vb, lpb, eb = b.policy.evaluate_actions(obs, acts)
    for x, y in ((va, vb), (lpa, lpb), (ea, eb)):
        np.testing.assert_allclose(y.cpu().numpy(), x.cpu().numpy(), rtol=1e-3, atol=1e-4)


@pytest.mark.parametrize("discrete", [False, True])
def test_general_towers_update_replayed_as_a_graph_equals_the_launch_sequence(discrete, monkeypatch):
    """From the third update on `GeneralTowers.ppo_update` replays one captured hipGraph (Adam's step-dependent scalars
    from a device-side count) instead of ~35 launches per minibatch: same parameters, optimiser state, step count and
    logged statistics as the launch-by-launch run (device `pow` vs host `pow` may differ in the last bit of the step size)."""
    import imitation_amd as ia
    from imitation_amd import general_policy
    from imitation_amd.vec_env import SyntheticVecEnv

    outs = {}
    for graphs in (True, False):
        monkeypatch.setattr(general_policy, "GRAPH_UPDATES", graphs)
        th.manual_seed(5)
        np.random.seed(5)
        venv = SyntheticVecEnv(num_envs=8, obs_dim=7, act_dim=3, horizon=9, seed=0, n_discrete=3 if discrete else None)
        algo = ia.PPO(ia.ActorCriticPolicy, venv, n_steps=16, batch_size=48, n_epochs=2, ent_coef=0.02, seed=0,
                      policy_kwargs=dict(net_arch=dict(pi=[24, 16], vf=[40]),
                                         features_extractor_class=ia.NormalizeFeaturesExtractor,
                                         features_extractor_kwargs=dict(normalize_class=ia.RunningNorm)), device="cuda")
        assert not algo.policy.fused
        algo.learn(16 * 8 * 6)       # six updates: eager, capture + replay, four replays
        pol = algo.policy
        used = [v for v in pol.__dict__.get("_update_graphs", {}).values() if v != "warm"]
        assert (len(used) == 1) == graphs
        outs[graphs] = ({k: v.detach().cpu().numpy().copy() for k, v in pol.state_dict().items()},
                        pol.optimizer.exp_avg.cpu().numpy().copy(), pol.optimizer.step_count,
                        {k: v for k, v in algo.logger.name_to_value.items() if k.startswith("train/")})
    a, b = outs[True], outs[False]
    assert a[2] == b[2] == 6 * 2 * 3
    for k in a[0]:
        np.testing.assert_allclose(a[0][k], b[0][k], rtol=2e-5, atol=2e-6, err_msg=k)
    np.testing.assert_allclose(a[1], b[1], rtol=2e-5, atol=1e-7)
    assert set(a[3]) == set(b[3])
    for k in a[3]:
        np.testing.assert_allclose(a[3][k], b[3][k], rtol=1e-4, atol=1e-6, err_msg=k)


def test_general_towers_state_dict_checkpoint_and_bc(tmp_path):
    import imitation_amd as ia
    from imitation_amd import bc, spaces
    osp = spaces.Box(-np.ones(5, dtype=np.float32), np.ones(5, dtype=np.float32))
    asp = spaces.Box(-np.ones(2, dtype=np.float32), np.ones(2, dtype=np.float32))
    th.manual_seed(0)
    p = ia.ActorCriticPolicy(osp, asp, lambda _: 1e-3, net_arch=dict(pi=[16, 8, 8], vf=[12]), activation_fn=nn.ReLU).to("cuda")
    assert list(p.state_dict())[:3] == ["log_std", "mlp_extractor.policy_net.0.weight", "mlp_extractor.policy_net.0.bias"]
    assert "mlp_extractor.policy_net.4.weight" in p.state_dict() and p.state_dict()["value_net.weight"].shape == (1, 12)
    q = ia.ActorCriticPolicy(osp, asp, lambda _: 1e-3, net_arch=dict(pi=[16, 8, 8], vf=[12]), activation_fn=nn.ReLU).to("cuda")
    q.load_state_dict(p.state_dict())
    obs = np.random.default_rng(1).standard_normal((9, 5)).astype(np.float32)
    np.testing.assert_array_equal(q.predict(obs, deterministic=True)[0], p.predict(obs, deterministic=True)[0])
    # BC accepts general-tower policies (explicit forward / head-gradient / backward launches; goldens in test_bc.py)
    trainer = bc.BC(observation_space=osp, action_space=asp, rng=np.random.default_rng(0), policy=p)
    assert trainer._explicit and not trainer._fused


@pytest.mark.parametrize("discrete", [True, False])
def test_cnn_policy_ppo_matches_sb3_restated(discrete):
    """PPO with the NatureCNN actor-critic policy on uint8 frames (`PPO("CnnPolicy", ...)`): rollouts, GAE and the
    minibatch updates against the SB3 restatement (oracle, CPU) on the same seeds -- Categorical and DiagGaussian
    heads. Same outlier rule as the image golden: a few conv weights may take a different early Adam step."""
    import imitation_amd as ia
    from imitation_amd.vec_env import SyntheticImageVecEnv
    from oracle import sb3_restated as sb

    def run(ppo_cls, policy_cls, device):
        th.manual_seed(5)
        np.random.seed(5)
        venv = SyntheticImageVecEnv(num_envs=4, shape=(4, 36, 36), act_dim=2, horizon=6, seed=1,
                                    n_discrete=3 if discrete else None)
        algo = ppo_cls(policy_cls, venv, n_steps=8, batch_size=16, n_epochs=2, ent_coef=0.01, learning_rate=1e-4, seed=0,
                       device=device)
        algo.learn(4 * 8 * 2)
        rb = algo.rollout_buffer
        out = {f"policy/{k}": v.detach().cpu().numpy() for k, v in algo.policy.state_dict().items()}
        for k in ("values", "log_probs", "advantages", "returns", "actions"):
            v = getattr(rb, k)
            out[f"rollout/{k}"] = (v.detach().cpu().numpy() if isinstance(v, th.Tensor) else np.asarray(v)).reshape(-1)
        return out

    ref = run(sb.PPO, sb.ActorCriticCnnPolicy, "cpu")
    got = run(ia.PPO, "CnnPolicy", "cuda")
    assert set(ref) == set(got)
    for k in ref:
        x, y = got[k].astype(np.float64), ref[k].astype(np.float64)
        err = np.abs(x - y)
        bad = err > 5e-5 + 2e-4 * np.abs(y)
        assert bad.mean() <= 1e-3 and (not bad.any() or err[bad].max() <= 4 * 1e-4), (k, bad.mean(), err.max())
