"""The fused discriminator updates (`csrc/disc_fused.hip`: D -> H -> H -> 1 with H in {128, 256}, D <= 63; `csrc/airl_fused.hip`
`disc32_rows_kernel`: the reference's default 32 x 32 stack with up to 64 inputs) through `BasicRewardNet.disc_step_c`
against a float64 torch autograd restatement of one `train_disc` minibatch (`adversarial/common.py:352-373`,
`rewards/reward_nets.py:441-457`, `util/networks.py:79-91,111-134`): logits, the statistics row, the flat gradient,
the RunningNorm state, Adam's step. Tolerances: forward 1e-5 relative, gradients 2e-5 * sqrt(rows) absolute headroom
for fp32 summation order; counts exact."""
import numpy as np
import pytest
import torch as th

import imitation_amd as p
from imitation_amd import _lib as L
from imitation_amd import networks, reward_nets, spaces
from imitation_amd.networks import HipAdam, TransitionTable

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not th.cuda.is_available():
        pytest.skip("no GPU")
    L.load()


def _tables(n_tab, od, ad, discrete, seed):
    g = np.random.default_rng(seed)
    obs = g.standard_normal((n_tab, od)).astype(np.float32) * 1.7 + 0.3
    nxt = g.standard_normal((n_tab, od)).astype(np.float32)
    acts = g.integers(0, ad, n_tab).astype(np.int64) if discrete else g.uniform(-1, 1, (n_tab, ad)).astype(np.float32)
    dones = (g.random(n_tab) < 0.3).astype(np.uint8)
    t = TransitionTable(th.as_tensor(obs).to(DEV), th.as_tensor(acts).to(DEV), th.as_tensor(nxt).to(DEV),
                        th.as_tensor(dones).to(DEV), discrete)
    return t, dict(obs=obs, acts=acts, next_obs=nxt, dones=dones)


def _concat(host, idx, flags, ad, discrete):
    cols = []
    if flags[0]:
        cols.append(host["obs"][idx])
    if flags[1]:
        a = host["acts"][idx]
        cols.append(np.eye(ad, dtype=np.float32)[a] if discrete else a)
    if flags[2]:
        cols.append(host["next_obs"][idx])
    if flags[3]:
        cols.append(host["dones"][idx].astype(np.float32)[:, None])
    return np.concatenate(cols, axis=1)


CASES = [
    # (obs_dim, act_dim, discrete, flags (state, action, next_state, done), hid, n_expert, n_gen, norm)
    (17, 6, False, (True, True, False, False), (32, 32), 64, 64, True),        # the reference default at HalfCheetah width
    (17, 6, False, (True, True, True, True), (32, 32), 200, 200, True),        # use_next_state + use_done: 41 inputs
    (27, 8, False, (True, True, False, False), (32, 32), 300, 211, True),      # Ant width (35), ragged: 511 rows
    (4, 2, True, (True, True, False, False), (32, 32), 33, 31, True),          # CartPole: one-hot actions, 6 inputs
    (29, 6, False, (True, True, True, False), (32, 32), 128, 128, False),      # 64 inputs (the maximum), no input norm
    (17, 6, False, (True, True, False, False), (128, 128), 100, 92, True),
    (17, 6, False, (True, True, False, False), (256, 256), 256, 256, True),
    # rows of 25 .. 64 floats through the wide one-launch tile pass (`disc_fb_kernel<H, 64, 64>`)
    (27, 8, False, (True, True, False, False), (256, 256), 300, 211, True),    # Ant GAIL: 35 inputs, ragged 511 rows
    (17, 6, False, (True, True, True, True), (256, 256), 1000, 1000, True),    # use_next_state + use_done: 41 inputs
    (28, 6, False, (True, True, True, True), (128, 128), 100, 93, True),       # 63 inputs (the maximum)
    (11, 4, True, (True, True, True, False), (256, 256), 64, 64, False),       # one-hot actions, 26 inputs, no input norm
]


@pytest.mark.parametrize("od,ad,discrete,flags,hid,n0,n1,norm", CASES)
def test_fused_disc_step_matches_float64_autograd(od, ad, discrete, flags, hid, n0, n1, norm):
    th.manual_seed(3)
    osp = spaces.Box(-np.inf, np.inf, (od,), np.float32)
    asp = spaces.Discrete(ad) if discrete else spaces.Box(-1, 1, (ad,), np.float32)
    kw = dict(normalize_input_layer=p.RunningNorm) if norm else {}
    net = reward_nets.BasicRewardNet(osp, asp, use_state=flags[0], use_action=flags[1], use_next_state=flags[2],
                                     use_done=flags[3], hid_sizes=hid, **kw).to(DEV)
    mlp = net.mlp
    R = n0 + n1
    assert net.fused_ws_of(R) is not None, "this shape must take a fused path"
    e_tab, e_host = _tables(700, od, ad, discrete, 1)
    g_tab, g_host = _tables(500, od, ad, discrete, 2)
    rng = np.random.default_rng(5)
    e_idx, g_idx = rng.integers(0, 700, n0), rng.integers(0, 500, n1)
    if norm:   # non-trivial running statistics before the update (an earlier batch of 77 rows)
        warm = th.as_tensor(rng.standard_normal((77, mlp.dims[0])).astype(np.float32) * 0.5 + 0.2).to(DEV)
        mlp.norm.update_stats(warm)
    params0 = mlp.flat.detach().cpu().double().clone()
    st0 = None if not norm else (mlp.norm.running_mean.cpu().double().clone(), mlp.norm.running_var.cpu().double().clone(),
                                 int(mlp.norm.count))
    opt = HipAdam(mlp.flat, mlp.grad, lr=1e-3)
    stats = th.zeros(8, device=DEV)
    bce_ws = th.zeros(int(L.load().ia_bce_ws_floats(R)), device=DEV)
    src = [(e_tab, th.as_tensor(e_idx).to(DEV), n0), (g_tab, th.as_tensor(g_idx).to(DEV), n1)]
    with networks.training(net):
        ws = net.disc_step_c(src, n0, 0.5, stats, bce_ws, accumulate=False, adam=opt)
    th.cuda.synchronize()

    # ---- float64 restatement
    X = th.as_tensor(np.concatenate([_concat(e_host, e_idx, flags, ad, discrete),
                                     _concat(g_host, g_idx, flags, ad, discrete)])).double()
    if norm:
        mean, var, cnt = st0
        bm, bv = X.mean(0), X.var(0, unbiased=False)
        tot = cnt + R
        delta = bm - mean
        new_mean = mean + delta * R / tot
        new_var = (var * cnt + bv * R + delta ** 2 * cnt * R / tot) / tot
        assert int(mlp.norm.count) == tot
        th.testing.assert_close(mlp.norm.running_mean.cpu().double(), new_mean, rtol=1e-5, atol=1e-6)
        th.testing.assert_close(mlp.norm.running_var.cpu().double(), new_var, rtol=1e-5, atol=1e-6)
        Xn = (X - new_mean) / th.sqrt(new_var + 1e-5)
    else:
        Xn = X
    D, H1, H2 = mlp.dims[0], hid[0], hid[1]
    P = params0.clone().requires_grad_(True)
    o = 0
    W1 = P[o:o + H1 * D].view(H1, D); o += H1 * D
    b1 = P[o:o + H1]; o += H1
    W2 = P[o:o + H2 * H1].view(H2, H1); o += H2 * H1
    b2 = P[o:o + H2]; o += H2
    W3 = P[o:o + H2].view(1, H2); o += H2
    b3 = P[o:o + 1]
    logits = (th.relu(th.relu(Xn @ W1.T + b1) @ W2.T + b2) @ W3.T + b3).reshape(-1)
    y = th.cat([th.ones(n0), th.zeros(n1)]).double()
    loss = th.nn.functional.binary_cross_entropy_with_logits(logits, y) * 0.5
    loss.backward()
    got_logits = ws["out"].reshape(-1).cpu().double()
    th.testing.assert_close(got_logits, logits.detach(), rtol=1e-5, atol=2e-5)
    gtol = 2e-5 * max(1.0, float(P.grad.abs().max()))
    th.testing.assert_close(mlp.grad.cpu().double(), P.grad, rtol=2e-4, atol=gtol)
    s = stats.cpu().double().numpy()
    assert abs(s[0] - float(loss)) < 1e-5 * max(1.0, abs(float(loss)))
    pred_gen, true_gen = logits.detach() < 0, y == 0
    ok = pred_gen == true_gen
    assert s[1] == float(ok.sum()) and s[2] == float((ok & ~true_gen).sum()) and s[3] == float((ok & true_gen).sum())
    assert s[4] == float(pred_gen.sum()) and s[6] == n0 and s[7] == n1
    pr = th.sigmoid(logits.detach())
    ent = th.nn.functional.binary_cross_entropy_with_logits(logits.detach(), pr, reduction="sum")
    assert abs(s[5] - float(ent)) < 1e-4 * max(1.0, abs(float(ent)))
    # Adam's first step: p - lr * g / (|g| + eps) (bias-corrected moments of a single gradient)
    g = P.grad
    want = params0 - 1e-3 * g / (g.abs() + 1e-8)
    moved = g.abs() > 1e-7   # (elements with a vanishing gradient take a step that depends on its last bits)
    th.testing.assert_close(mlp.flat.cpu().double()[moved], want[moved], rtol=0, atol=2e-5)


@pytest.mark.parametrize("hid,od", [((32, 32), 17), ((128, 128), 17), ((256, 256), 27)])
def test_round_assembly_equals_per_update_assembly(hid, od):
    """`assemble_round` + pre-assembled updates (what a pipelined round runs) == one-call updates in sequence: same
    parameters, statistics rows and RunningNorm state, bit for bit."""
    ad, n_upd, mb = (8 if od == 27 else 6), 3, 96
    osp, asp = spaces.Box(-np.inf, np.inf, (od,), np.float32), spaces.Box(-1, 1, (ad,), np.float32)
    e_tab, _ = _tables(400, od, ad, False, 1)
    g_tab, _ = _tables(300, od, ad, False, 2)
    rng = np.random.default_rng(9)
    idx_all = th.as_tensor(np.stack([np.stack([rng.integers(0, 400, mb), rng.integers(0, 300, mb)])
                                     for _ in range(n_upd)])).to(DEV).contiguous()
    outs = []
    for pre in (False, True):
        th.manual_seed(11)
        net = reward_nets.BasicRewardNet(osp, asp, hid_sizes=hid, normalize_input_layer=p.RunningNorm).to(DEV)
        opt = HipAdam(net.mlp.flat, net.mlp.grad, lr=1e-3)
        stats = th.zeros(n_upd, 8, device=DEV)
        bce_ws = th.zeros(int(L.load().ia_bce_ws_floats(2 * mb)), device=DEV)
        with networks.training(net):
            rw = net.assemble_round(e_tab, g_tab, idx_all, n_upd, mb) if pre else None
            assert (rw is not None) == pre
            for k in range(n_upd):
                src = [(e_tab, idx_all[k, 0], mb), (g_tab, idx_all[k, 1], mb)]
                net.disc_step_c(src, mb, 1.0, stats[k], bce_ws, accumulate=False, adam=opt,
                                pre=(rw, k) if pre else None)
        th.cuda.synchronize()
        outs.append((net.mlp.flat.cpu().clone(), stats.cpu().clone(), net.mlp.norm.running_mean.cpu().clone(),
                     net.mlp.norm.running_var.cpu().clone(), int(net.mlp.norm.count)))
    for a, b in zip(*outs):
        assert (a == b) if isinstance(a, int) else th.equal(a, b)
    assert outs[0][4] == n_upd * 2 * mb


def test_gradient_accumulation_through_the_narrow_fused_step():
    """Two half minibatches accumulated (`demo_minibatch_size < demo_batch_size`, `common.py:352-373`) == the gradient of
    the whole batch under frozen statistics (eval-mode norm), within summation order."""
    od, ad, mb = 11, 3, 80
    osp, asp = spaces.Box(-np.inf, np.inf, (od,), np.float32), spaces.Box(-1, 1, (ad,), np.float32)
    e_tab, _ = _tables(400, od, ad, False, 1)
    g_tab, _ = _tables(300, od, ad, False, 2)
    rng = np.random.default_rng(4)
    ei, gi = th.as_tensor(rng.integers(0, 400, 2 * mb)).to(DEV), th.as_tensor(rng.integers(0, 300, 2 * mb)).to(DEV)
    th.manual_seed(2)
    net = reward_nets.BasicRewardNet(osp, asp, hid_sizes=(32, 32)).to(DEV)
    stats = th.zeros(8, device=DEV)
    bce_ws = th.zeros(int(L.load().ia_bce_ws_floats(4 * mb)), device=DEV)
    with networks.training(net):
        net.disc_step_c([(e_tab, ei, 2 * mb), (g_tab, gi, 2 * mb)], 2 * mb, 1.0, stats, bce_ws, accumulate=False)
        whole = net.mlp.grad.cpu().clone()
        for h in range(2):
            sl = slice(h * mb, (h + 1) * mb)
            net.disc_step_c([(e_tab, ei[sl].contiguous(), mb), (g_tab, gi[sl].contiguous(), mb)], mb, 0.5, stats, bce_ws,
                            accumulate=h > 0)
    th.cuda.synchronize()
    th.testing.assert_close(net.mlp.grad.cpu(), whole, rtol=2e-4, atol=2e-6)


@pytest.mark.parametrize("hid,B,norm,od", [((256, 256), 256, True, 17), ((128, 128), 200, True, 17),
                                           ((256, 256), 1000, False, 11), ((256, 256), 4096, True, 17),
                                           # rows of up to 64 floats (`disc_gp_kernel<H, 64>`): D = 35 (Ant-width), D = 63
                                           ((256, 256), 1000, True, 29), ((256, 256), 500, True, 57),
                                           ((128, 128), 200, False, 29), ((256, 256), 4096, True, 29),
                                           ((128, 128), 333, True, 57)])
def test_fused_gradient_penalty_matches_float64_double_backward(hid, B, norm, od):
    """The opt-in gradient penalty inside the 128 / 256-wide fused update (`disc_fwd_kernel` / `disc_bwd_kernel`
    MODE 1 / 2): gradient of [BCE + coef * mean (|grad_x D(x_hat)| - target)^2] against a float64 double-backward graph,
    the penalty's mean, and -- same call with the coefficient's passes switched off -- that the penalty is what
    differs."""
    th.manual_seed(11)
    ad = 6
    osp = spaces.Box(-np.inf, np.inf, (od,), np.float32)
    asp = spaces.Box(-1, 1, (ad,), np.float32)
    kw = dict(normalize_input_layer=p.RunningNorm) if norm else {}
    net = reward_nets.BasicRewardNet(osp, asp, hid_sizes=hid, **kw).to(DEV)
    mlp = net.mlp
    with th.no_grad():   # small default-initialised weights leave the input gradient far below the target: scale up
        mlp.flat.mul_(2.5)
    R = 2 * B
    assert net.fused_gp_ws(B) is not None
    e_tab, e_host = _tables(5000, od, ad, False, 1)
    g_tab, g_host = _tables(5000, od, ad, False, 2)
    rng = np.random.default_rng(5)
    e_idx, g_idx = rng.integers(0, 5000, B), rng.integers(0, 5000, B)
    if norm:
        warm = th.as_tensor(rng.standard_normal((77, mlp.dims[0])).astype(np.float32) * 0.5 + 0.2).to(DEV)
        mlp.norm.update_stats(warm)
    params0 = mlp.flat.detach().cpu().double().clone()
    stats = th.zeros(8, device=DEV)
    bce_ws = th.zeros(int(L.load().ia_bce_ws_floats(R)), device=DEV)
    src = [(e_tab, th.as_tensor(e_idx).to(DEV), B), (g_tab, th.as_tensor(g_idx).to(DEV), B)]
    e = th.rand(B)
    coef, target = 3.0, 1.0
    with networks.training(net):
        ws = net.disc_step_c(src, B, 1.0, stats, bce_ws, accumulate=False, adam=None, gp=(e.to(DEV), coef, target))
    th.cuda.synchronize()
    got = mlp.grad.detach().cpu().double().clone()
    pen_got = float(ws["gp_out"][0])
    mean, var = (mlp.norm.running_mean.cpu().double(), mlp.norm.running_var.cpu().double()) if norm else (None, None)

    flags = (True, True, False, False)
    X = th.as_tensor(np.concatenate([_concat(e_host, e_idx, flags, ad, False),
                                     _concat(g_host, g_idx, flags, ad, False)])).double()
    D, H = mlp.dims[0], hid[0]
    P = params0.clone().requires_grad_(True)
    o = 0
    W1 = P[o:o + H * D].view(H, D); o += H * D
    b1 = P[o:o + H]; o += H
    W2 = P[o:o + H * H].view(H, H); o += H * H
    b2 = P[o:o + H]; o += H
    W3 = P[o:o + H].view(1, H); o += H
    b3 = P[o:o + 1]

    def f(x):
        xn = (x - mean) / th.sqrt(var + 1e-5) if norm else x
        return (th.relu(th.relu(xn @ W1.T + b1) @ W2.T + b2) @ W3.T + b3).reshape(-1)

    y = th.cat([th.ones(B), th.zeros(B)]).double()
    bce = th.nn.functional.binary_cross_entropy_with_logits(f(X), y)
    xh = (e.double()[:, None] * X[:B] + (1 - e.double()[:, None]) * X[B:]).requires_grad_(True)
    (gx,) = th.autograd.grad(f(xh).sum(), xh, create_graph=True)
    pen_rows = (gx.norm(dim=1) - target) ** 2
    (g_bce,) = th.autograd.grad(bce, P, retain_graph=True)
    (g_pen,) = th.autograd.grad(coef * pen_rows.mean(), P)
    assert float(pen_rows.mean()) > 1e-3, "degenerate case: no penalty"
    np.testing.assert_allclose(pen_got, float(pen_rows.mean()), rtol=5e-5)
    scale = float(g_pen.abs().max())
    # the penalty's share of the gradient, then the whole
    th.testing.assert_close(got - g_bce, g_pen, rtol=2e-3, atol=3e-5 * scale)
    th.testing.assert_close(got, g_bce + g_pen, rtol=5e-4, atol=2e-5 * max(scale, float(g_bce.abs().max())))
    assert float((g_pen[o - H - H * H - H:o - H - H].abs()).max()) > 0     # W2 entries carry a penalty gradient
    # bias entries get no penalty gradient (masks fixed): the float64 graph agrees
    assert float(g_pen[H * D:H * D + H].abs().max()) == 0.0


@pytest.mark.parametrize("hid,n0,n1,rows", [((256, 256), 1000, 1000, 64), ((128, 128), 300, 211, 64), ((256, 256), 200, 200, 32)])
def test_one_launch_tile_pass_is_bit_identical_to_forward_plus_backward_launches(hid, n0, n1, rows):
    """`disc_fb_kernel` (forward + backward of a tile in one workgroup) against `disc_fwd_kernel` + `disc_bwd_kernel`
    (`ia_disc_fused_split_tiles(1)`): same arithmetic in the same order -> logits, statistics, gradient, parameters after
    Adam and the saved activations bit for bit."""
    od, ad = 17, 6
    osp = spaces.Box(-np.inf, np.inf, (od,), np.float32)
    asp = spaces.Box(-1, 1, (ad,), np.float32)
    e_tab, _ = _tables(3000, od, ad, False, 1)
    g_tab, _ = _tables(3000, od, ad, False, 2)
    rng = np.random.default_rng(9)
    e_idx, g_idx = th.as_tensor(rng.integers(0, 3000, n0)).to(DEV), th.as_tensor(rng.integers(0, 3000, n1)).to(DEV)
    R = n0 + n1
    outs = []
    lib = L.load()
    try:
        lib.ia_disc_fused_tile_rows(rows)
        for split in (1, 0):
            lib.ia_disc_fused_split_tiles(split)
            lib.ia_disc_fused_side_reduce(1 - split)   # (the closing reduction: one launch / split around the product)
            th.manual_seed(3)
            net = reward_nets.BasicRewardNet(osp, asp, hid_sizes=hid, normalize_input_layer=p.RunningNorm).to(DEV)
            mlp = net.mlp
            opt = HipAdam(mlp.flat, mlp.grad, lr=1e-3)
            stats = th.zeros(8, device=DEV)
            bce_ws = th.zeros(int(lib.ia_bce_ws_floats(R)), device=DEV)
            rows_out = []
            for _ in range(2):
                with networks.training(net):
                    ws = net.disc_step_c([(e_tab, e_idx, n0), (g_tab, g_idx, n1)], n0, 1.0, stats, bce_ws,
                                         accumulate=False, adam=opt)
                th.cuda.synchronize()
                rows_out.append((ws["out"].clone(), stats.clone(), mlp.grad.clone(), mlp.flat.clone(),
                                 ws["hidden"][:2 * R * hid[0]].clone()))
            outs.append(rows_out)
    finally:
        lib.ia_disc_fused_split_tiles(0)
        lib.ia_disc_fused_side_reduce(1)
        lib.ia_disc_fused_tile_rows(64)
    names = ("logits", "statistics", "gradient", "parameters", "saved activations")
    for k, (a, b) in enumerate(zip(*outs)):
        for name, x, y in zip(names, a, b):
            if not th.equal(x, y):
                ii = th.nonzero((x != y).reshape(-1)).reshape(-1)
                raise AssertionError(f"update {k}: {name} differ at {ii.numel()} of {x.numel()} places, first {ii[:6].tolist()}, "
                                     f"values {x.reshape(-1)[ii[:3]].tolist()} vs {y.reshape(-1)[ii[:3]].tolist()}")


@pytest.mark.parametrize("hid,B,norm,form", [((256, 256), 1000, True, 8), ((128, 128), 200, True, 8), ((256, 256), 4096, False, 8),
                                             ((256, 256), 1100, True, 8), ((256, 256), 1100, True, 80), ((256, 256), 1000, True, 1),
                                             ((256, 256), 1100, True, 2),   # 35 tiles: the last two-tile workgroup is half idle
                                             ((256, 256), 4096, False, 2)])
def test_one_launch_penalty_pass_is_bit_identical_to_its_three_launches(hid, B, norm, form):
    """`disc_gp_kernel` (the penalty's three tile passes in one workgroup: masks in registers, C in LDS) against
    `disc_fwd_kernel<.,32,1>` + `disc_bwd_kernel<.,32,1>` + `disc_fwd_kernel<.,32,2>` (`ia_disc_fused_split_tiles(1)`):
    gradient (BCE + penalty), the penalty's mean and the second pass's GEMM operands bit for bit. `form`: the 256-wide pass as
    one tile per workgroup with eight column waves (8, default -- since round 6 in ONE launch with the update's own tile pass,
    `disc_fb_gp_kernel`; 80: the same form as a launch of its own), two tiles per workgroup (2), or four waves (1):
    `ia_disc_fused_gp_groups`."""
    od, ad = 17, 6
    osp = spaces.Box(-np.inf, np.inf, (od,), np.float32)
    asp = spaces.Box(-1, 1, (ad,), np.float32)
    kw = dict(normalize_input_layer=p.RunningNorm) if norm else {}
    e_tab, _ = _tables(6000, od, ad, False, 1)
    g_tab, _ = _tables(6000, od, ad, False, 2)
    rng = np.random.default_rng(9)
    e_idx, g_idx = th.as_tensor(rng.integers(0, 6000, B)).to(DEV), th.as_tensor(rng.integers(0, 6000, B)).to(DEV)
    e = th.rand(B, generator=th.Generator().manual_seed(4)).to(DEV)
    R = 2 * B
    outs = []
    lib = L.load()
    try:
        lib.ia_disc_fused_gp_groups(form)
        for split in (1, 0):
            lib.ia_disc_fused_split_tiles(split)
            lib.ia_disc_fused_side_reduce(1 - split)   # (the closing reduction: one launch / split around the product)
            th.manual_seed(3)
            net = reward_nets.BasicRewardNet(osp, asp, hid_sizes=hid, **kw).to(DEV)
            with th.no_grad():
                net.mlp.flat.mul_(2.5)
            stats = th.zeros(8, device=DEV)
            bce_ws = th.zeros(int(lib.ia_bce_ws_floats(R)), device=DEV)
            with networks.training(net):
                ws = net.disc_step_c([(e_tab, e_idx, B), (g_tab, g_idx, B)], B, 1.0, stats, bce_ws, accumulate=False,
                                     adam=None, gp=(e, 3.0, 1.0))
            th.cuda.synchronize()
            n_act = 2 * B * hid[0]      # v1 | u2 at the head of the penalty workspace
            outs.append((net.mlp.grad.clone(), ws["gp_out"].clone(), ws["gp_ws"][:n_act].clone(), stats.clone()))
    finally:
        lib.ia_disc_fused_split_tiles(0)
        lib.ia_disc_fused_side_reduce(1)
        lib.ia_disc_fused_gp_groups(8)
    for name, x, y in zip(("gradient", "penalty", "v1 | u2", "statistics"), *outs):
        if not th.equal(x, y):
            ii = th.nonzero((x != y).reshape(-1)).reshape(-1)
            raise AssertionError(f"{name} differ at {ii.numel()} of {x.numel()} places, first {ii[:6].tolist()}, values "
                                 f"{x.reshape(-1)[ii[:3]].tolist()} vs {y.reshape(-1)[ii[:3]].tolist()}")
    assert float(outs[0][1]) > 1e-3


@pytest.mark.parametrize("od,ad,B", [(29, 6, 1000), (27, 8, 333)])
def test_wide_row_penalty_pass_forms_are_bit_identical(od, ad, B):
    """`disc_gp_kernel<256, 64>` (rows of up to 64 floats: Ant-width nets) with its columns over eight waves (default, two
    waves per SIMD) against the four-wave form (`ia_disc_fused_gp_groups(1)`): gradient, penalty mean, the second pass's GEMM
    operands and the statistics bit for bit (the float64 double-backward tests of tests/test_grad_penalty_gpu.py cover the
    values themselves)."""
    osp = spaces.Box(-np.inf, np.inf, (od,), np.float32)
    asp = spaces.Box(-1, 1, (ad,), np.float32)
    e_tab, _ = _tables(6000, od, ad, False, 1)
    g_tab, _ = _tables(6000, od, ad, False, 2)
    rng = np.random.default_rng(9)
    e_idx, g_idx = th.as_tensor(rng.integers(0, 6000, B)).to(DEV), th.as_tensor(rng.integers(0, 6000, B)).to(DEV)
    e = th.rand(B, generator=th.Generator().manual_seed(4)).to(DEV)
    R = 2 * B
    outs = []
    lib = L.load()
    try:
        for form in (8, 80, 1):
            lib.ia_disc_fused_gp_groups(form)
            th.manual_seed(3)
            net = reward_nets.BasicRewardNet(osp, asp, hid_sizes=(256, 256), normalize_input_layer=p.RunningNorm).to(DEV)
            with th.no_grad():
                net.mlp.flat.mul_(2.5)
            assert net.mlp.ldx > 24 and net.fused_gp_ws(B) is not None   # the wide-row fused penalty
            stats = th.zeros(8, device=DEV)
            bce_ws = th.zeros(int(lib.ia_bce_ws_floats(R)), device=DEV)
            with networks.training(net):
                ws = net.disc_step_c([(e_tab, e_idx, B), (g_tab, g_idx, B)], B, 1.0, stats, bce_ws, accumulate=False,
                                     adam=None, gp=(e, 3.0, 1.0))
            th.cuda.synchronize()
            n_act = 2 * B * 256
            outs.append((net.mlp.grad.clone(), ws["gp_out"].clone(), ws["gp_ws"][:n_act].clone(), stats.clone()))
    finally:
        lib.ia_disc_fused_gp_groups(8)
    for other in outs[1:]:   # (8: in one launch with the update's tile pass; 80: the same form apart; 1: four waves)
        for name, x, y in zip(("gradient", "penalty", "v1 | u2", "statistics"), outs[0], other):
            assert th.equal(x, y), name
    assert float(outs[0][1]) > 1e-3


@pytest.mark.parametrize("hid,od,ad,discrete,R,norm,softplus", [
    ((256, 256), 17, 6, False, 16384, True, True),     # config P's relabelling tile
    ((256, 256), 17, 6, False, 1000, False, False),    # ragged last tile, no input norm, raw logits
    ((128, 128), 4, 2, True, 333, True, True),         # CartPole width (one-hot actions: 6 inputs)
    ((256, 256), 11, 3, False, 64, True, False),
    ((128, 128), 11, 3, False, 1, True, True),         # a single transition (`predict_th` of one row)
    ((256, 256), 17, 6, False, 65, False, True),       # one row into the second tile
    ((32, 32), 17, 6, False, 16384, True, True),       # the reference's default discriminator: the row kernel's prediction mode
    ((32, 32), 29, 6, False, 500, True, False),        # ... at 35 inputs (rows of 36 floats), ragged last workgroup
    ((32, 32), 4, 2, True, 33, False, True)])
def test_fused_prediction_matches_the_layer_by_layer_forward(hid, od, ad, discrete, R, norm, softplus):
    """`ia_disc_fused_predict` (`DenseStack.forward_rows(keep_hidden=False)`: the relabelling of a rollout tile,
    `rewards/reward_wrapper.py:110-115` / `rewards/reward_nets.py:176-204`) against `ia_running_norm_apply` + `ia_mlp_forward`
    on the same assembled rows and against a float64 restatement; GAIL's softplus (`gail.py:75-83`) in the epilogue."""
    osp = spaces.Box(-np.inf, np.inf, (od,), np.float32)
    asp = spaces.Discrete(ad) if discrete else spaces.Box(-1, 1, (ad,), np.float32)
    kw = dict(normalize_input_layer=p.RunningNorm) if norm else {}
    th.manual_seed(5)
    net = reward_nets.BasicRewardNet(osp, asp, hid_sizes=hid, **kw).to(DEV)
    tab, host = _tables(R, od, ad, discrete, 3)
    if norm:   # statistics of some other batch (train-mode pass), then predictions in eval mode
        other, _ = _tables(500, od, ad, discrete, 4)
        with networks.training(net):
            net._forward_table([(other, None, 500)], "warm")
    mlp = net.mlp
    act = L.ACT_SOFTPLUS if softplus else L.ACT_NONE
    outs = {}
    for fused in (True, False):
        mlp.FUSED_PREDICT = fused
        try:
            with networks.evaluating(net):
                outs[fused] = net._forward_table([(tab, None, R)], "pred", act).clone()
        finally:
            mlp.FUSED_PREDICT = True
    th.cuda.synchronize()
    assert mlp._predict_ws() is not None, "the shape must be covered by the tile kernel"
    a, b = outs[True].double().cpu().numpy(), outs[False].double().cpu().numpy()
    # float64 restatement
    X = _concat(host, np.arange(R), net.flags, ad, discrete).astype(np.float64)
    if norm:
        nrm = mlp.norm
        X = (X - nrm.running_mean.double().cpu().numpy()) / np.sqrt(nrm.running_var.double().cpu().numpy() + nrm.eps)
    flat = mlp.flat.double().cpu().numpy()
    D, H = mlp.dims[0], hid[0]
    o = 0
    W1 = flat[o:o + H * D].reshape(H, D); o += H * D
    b1 = flat[o:o + H]; o += H
    W2 = flat[o:o + H * H].reshape(H, H); o += H * H
    b2 = flat[o:o + H]; o += H
    w3 = flat[o:o + H]; o += H
    b3 = flat[o]
    h = np.maximum(X @ W1.T + b1, 0)
    h = np.maximum(h @ W2.T + b2, 0)
    ref = h @ w3 + b3
    if softplus:
        ref = np.maximum(ref, 0) + np.log1p(np.exp(-np.abs(ref)))
    scale = max(1.0, float(np.abs(ref).max()))
    assert np.abs(a - ref).max() < 2e-5 * scale, np.abs(a - ref).max()
    assert np.abs(b - ref).max() < 2e-5 * scale
    # the two device paths against each other (measured: <= 7.5e-8 absolute on these cases)
    assert np.abs(a - b).max() < 2e-6 * scale, np.abs(a - b).max()
    print(f"fused prediction vs layer-by-layer: max |diff| {np.abs(a - b).max():.3g} (bit-identical: {np.array_equal(a, b)})")
