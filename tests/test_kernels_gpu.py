"""Kernel-level parity (`-m gpu`): every C-ABI entry point of libimitation_hip.so against the
oracle's torch-CPU arithmetic on the same seeded inputs. Tolerances are stated per test;
integer / index work is compared bit-exactly."""
import ctypes as C
import math

import numpy as np
import pytest
import torch as th
from torch.nn import functional as F

from imitation_amd import _lib as L

pytestmark = pytest.mark.gpu

DEV = "cuda"


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not th.cuda.is_available():
        pytest.skip("no GPU")
    L.load()


def dev(x, dtype=th.float32):
    return th.as_tensor(np.ascontiguousarray(x)).to(DEV, dtype).contiguous()


_KEEP = []


@pytest.fixture(autouse=True)
def _release_temporaries():
    yield
    if th.cuda.is_available():
        th.cuda.synchronize()
    _KEEP.clear()


def dptr(x, dtype=th.float32):
    """Uploads `x` and returns its raw device pointer; the tensor stays alive until the test ends
    (raw pointers carry no ownership, and the caching allocator would otherwise recycle the block)."""
    t = dev(x, dtype)
    _KEEP.append(t)
    return L.ptr(t)


def rnd(*shape, seed=0, scale=1.0):
    g = th.Generator().manual_seed(seed)
    return th.randn(*shape, generator=g) * scale


def gemm(mode, A, B, M, N, K, bias=None, act=0, P=None, splits=1, want_db=False, lda=None, ldb=None):
    ldc = N
    Cs = th.full((splits if mode == 2 else 1, M, ldc), float("nan"), device=DEV)
    db = th.zeros(splits, M, device=DEV) if want_db else None
    L.call("ia_gemm_f32", mode, L.ptr(A), lda or A.shape[-1], L.ptr(B), ldb or B.shape[-1], L.ptr(Cs), ldc, M, N, K,
           L.ptr(bias), act, L.ptr(P), N if P is not None else 0, splits, L.ptr(db), L.stream())
    th.cuda.synchronize()
    return Cs, db


@pytest.mark.parametrize("M,N,K", [(128, 128, 32), (300, 256, 23), (1000, 32, 256), (64, 1, 256),
                                   (16384, 256, 256), (17, 5, 3), (257, 65, 33)])
def test_gemm_nt(M, N, K):
    A, B, b = rnd(M, K, seed=1), rnd(N, K, seed=2), rnd(N, seed=3)
    for act, f in [(0, lambda x: x), (1, th.relu), (2, th.tanh), (3, F.softplus)]:
        got, _ = gemm(0, dev(A), dev(B), M, N, K, bias=dev(b), act=act)
        ref = f(A.double() @ B.double().T + b.double()).float()
        th.testing.assert_close(got[0].cpu(), ref, rtol=2e-5, atol=2e-5 * math.sqrt(K))


@pytest.mark.parametrize("M,N,K,splits", [(64, 512, 3136, 12), (256, 512, 3136, 8), (5, 512, 3136, 12), (70, 36, 100, 3),
                                          (33, 64, 64, 1), (64, 128, 96, 7)])
def test_gemm_nt_split_along_k(M, N, K, splits):
    """`ia_gemm_f32_nt_splitk` (the NatureCNN `linear` layer at rollout / minibatch sizes): C = act(A . B^T + bias) with the
    product in `splits` K slabs (empty trailing slabs included), bias and activation behind their ordered sum; float64 reference."""
    A, B, b = rnd(M, K, seed=1), rnd(N, K, seed=2), rnd(N, seed=3)
    dA, dB, db = dev(A), dev(B), dev(b)   # (kept alive across the launches)
    for act, f in [(0, lambda x: x), (1, th.relu)]:
        parts = th.full((splits, M, N), float("nan"), device=DEV)
        out = th.full((M, N), float("nan"), device=DEV)
        L.call("ia_gemm_f32_nt_splitk", L.ptr(dA), K, L.ptr(dB), K, L.ptr(parts), L.ptr(out), N, M, N, K, L.ptr(db), act, splits,
               L.stream())
        th.cuda.synchronize()
        ref = f(A.double() @ B.double().T + b.double()).float()
        th.testing.assert_close(out.cpu(), ref, rtol=2e-5, atol=2e-5 * math.sqrt(K))
        got2 = th.empty_like(out)   # deterministic
        L.call("ia_gemm_f32_nt_splitk", L.ptr(dA), K, L.ptr(dB), K, L.ptr(parts), L.ptr(got2), N, M, N, K, L.ptr(db), act, splits,
               L.stream())
        th.cuda.synchronize()
        assert th.equal(out, got2)


@pytest.mark.parametrize("M,N,K", [(128, 128, 32), (1000, 23, 256), (16384, 256, 256), (70, 33, 1), (513, 64, 32)])
def test_gemm_nn_with_activation_grad(M, N, K):
    A, B = rnd(M, K, seed=1), rnd(K, N, seed=2)
    P = th.tanh(rnd(M, N, seed=3))
    got, _ = gemm(1, dev(A), dev(B), M, N, K, act=2, P=dev(P))
    ref = ((A.double() @ B.double()) * (1 - P.double() ** 2)).float()
    th.testing.assert_close(got[0].cpu(), ref, rtol=2e-5, atol=2e-5 * math.sqrt(K))
    Pr = th.relu(rnd(M, N, seed=4))
    got, _ = gemm(1, dev(A), dev(B), M, N, K, act=1, P=dev(Pr))
    ref = ((A.double() @ B.double()) * (Pr > 0)).float()
    th.testing.assert_close(got[0].cpu(), ref, rtol=2e-5, atol=2e-5 * math.sqrt(K))


@pytest.mark.parametrize("M,N,K,splits", [(256, 256, 16384, 64), (256, 23, 1000, 4), (1, 256, 777, 3),
                                          (32, 17, 64, 1), (64, 64, 100, 7)])
def test_gemm_tn_splitk_and_bias_sums(M, N, K, splits):
    A, B = rnd(K, M, seed=1), rnd(K, N, seed=2)  # A is [K,M], B is [K,N]
    got, db = gemm(2, dev(A), dev(B), M, N, K, splits=splits, want_db=True)
    ref = (A.double().T @ B.double()).float()
    th.testing.assert_close(got.sum(0).cpu(), ref, rtol=3e-5, atol=3e-5 * math.sqrt(K))
    th.testing.assert_close(db.sum(0).cpu(), A.double().sum(0).float(), rtol=3e-5, atol=3e-5 * math.sqrt(K))


def _torch_mlp(dims, act):
    layers = []
    for i in range(len(dims) - 1):
        layers.append(th.nn.Linear(dims[i], dims[i + 1]))
        if i < len(dims) - 2:
            layers.append(act())
    return th.nn.Sequential(*layers)


@pytest.mark.parametrize("dims,act,R", [((23, 256, 256, 1), "relu", 16384), ((24, 32, 32, 1), "relu", 1000),
                                        ((11, 32, 1), "relu", 130), ((6, 64, 32, 1), "tanh", 77), ((5, 1), "relu", 64)])
def test_mlp_forward_backward(dims, act, R):
    th.manual_seed(0)
    net = _torch_mlp(dims, th.nn.ReLU if act == "relu" else th.nn.Tanh)
    flat = th.cat([p.detach().reshape(-1) for p in net.parameters()])
    X = rnd(R, dims[0], seed=5)
    dOut = rnd(R, dims[-1], seed=6) / R
    out_ref = net(X)
    out_ref.backward(dOut)
    gref = th.cat([p.grad.reshape(-1) for p in net.parameters()])

    d = L.mlp_desc(dims, L.ACT_RELU if act == "relu" else L.ACT_TANH)
    lib = L.load()
    assert lib.ia_mlp_param_count(C.byref(d)) == flat.numel()
    hid = int(lib.ia_mlp_hidden_floats_per_row(C.byref(d)))
    params, Xd = dev(flat), dev(X)
    hidden = th.empty(max(1, R * hid), device=DEV)
    dhidden = th.empty_like(hidden)
    out = th.empty(R, dims[-1], device=DEV)
    L.call("ia_mlp_forward", C.byref(d), L.ptr(params), L.ptr(Xd), dims[0], R, L.ptr(hidden), L.ptr(out), 0,
           L.stream())
    th.testing.assert_close(out.cpu(), out_ref.detach(), rtol=1e-4, atol=1e-5)
    splits = max(1, min(64, R // 256))
    partials = th.full((splits, flat.numel()), float("nan"), device=DEV)
    dX = th.empty(R, dims[0], device=DEV)
    L.call("ia_mlp_backward", C.byref(d), L.ptr(params), L.ptr(Xd), dims[0], R, L.ptr(hidden), dptr(dOut),
           L.ptr(dhidden), L.ptr(partials), splits, L.ptr(dX), L.stream())
    grads = th.zeros(flat.numel(), device=DEV)
    L.call("ia_reduce_partials", L.ptr(partials), splits, flat.numel(), 1.0, 0, L.ptr(grads), L.stream())
    th.cuda.synchronize()
    scale = gref.abs().max().item()
    th.testing.assert_close(grads.cpu(), gref, rtol=2e-4, atol=2e-5 * max(scale, 1e-3))
    # accumulate=1 with scale 0.5 adds half again
    L.call("ia_reduce_partials", L.ptr(partials), splits, flat.numel(), 0.5, 1, L.ptr(grads), L.stream())
    th.testing.assert_close(grads.cpu(), 1.5 * gref, rtol=2e-4, atol=3e-5 * max(scale, 1e-3))
    # softplus output activation == -logsigmoid(-x)  (gail.py:75-83)
    L.call("ia_mlp_forward", C.byref(d), L.ptr(params), L.ptr(Xd), dims[0], R, L.ptr(hidden), L.ptr(out), 3,
           L.stream())
    th.testing.assert_close(out.cpu(), -F.logsigmoid(-out_ref.detach()), rtol=1e-4, atol=1e-5)


def test_adam_matches_torch():
    th.manual_seed(1)
    p0 = th.randn(5000)
    for wd in (0.0, 0.01):
        p = th.nn.Parameter(p0.clone())
        opt = th.optim.Adam([p], lr=1e-3, eps=1e-5, weight_decay=wd)
        pd, m, v = dev(p0), th.zeros(5000, device=DEV), th.zeros(5000, device=DEV)
        for step in range(1, 6):
            g = rnd(5000, seed=step)
            p.grad = g.clone()
            opt.step()
            bc1, bc2 = 1 - 0.9 ** step, 1 - 0.999 ** step
            L.call("ia_adam_step", L.ptr(pd), dptr(g), L.ptr(m), L.ptr(v), 5000, 0.9, 0.999, 1e-5, wd,
                   1e-3 / bc1, math.sqrt(bc2), L.stream())
        th.testing.assert_close(pd.cpu(), p.detach(), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("R,D", [(16384, 23), (100, 17), (1, 4), (513, 64), (2048, 1)])
def test_running_norm_update_and_apply(R, D):
    from oracle.imitation_restated import RunningNorm
    rn = RunningNorm(D)
    mean, var = th.zeros(D, device=DEV), th.ones(D, device=DEV)
    count = th.zeros((), dtype=th.int32, device=DEV)
    ws = th.empty(int(L.load().ia_running_norm_ws_floats(R, D)), device=DEV)
    for it in range(3):
        X = rnd(R, D, seed=it) * (1 + it) + it
        rn.train()
        ref = rn(X)
        Xd = dev(X)
        L.call("ia_running_norm_update", L.ptr(Xd), D, R, D, L.ptr(mean), L.ptr(var), L.ptr(count), L.ptr(ws),
               L.stream())
        ld = D + 3
        Y = th.full((R, ld), float("nan"), device=DEV)
        L.call("ia_running_norm_apply", L.ptr(Xd), D, R, D, L.ptr(mean), L.ptr(var), 1e-5, L.ptr(Y), ld, L.stream())
        assert int(count.item()) == int(rn.count)  # int32 count: exact
        th.testing.assert_close(mean.cpu(), rn.running_mean, rtol=1e-5, atol=1e-6)
        th.testing.assert_close(var.cpu(), rn.running_var, rtol=1e-4, atol=1e-6)
        th.testing.assert_close(Y[:, :D].cpu(), ref, rtol=1e-4, atol=1e-5)
        assert th.all(Y[:, D:] == 0)


def test_gather_concat_bit_exact():
    rng = np.random.default_rng(0)
    N, Do, Da = 500, 17, 6
    obs, nxt = rng.standard_normal((N, Do)).astype(np.float32), rng.standard_normal((N, Do)).astype(np.float32)
    act = rng.uniform(-1, 1, (N, Da)).astype(np.float32)
    acti = rng.integers(0, Da, N)
    dones = rng.random(N) < 0.3
    idx = rng.integers(0, N, 200)
    for use in [(1, 1, 0, 0), (1, 1, 1, 1), (0, 1, 1, 0), (1, 0, 0, 1)]:
        for discrete in (False, True):
            parts = []
            if use[0]:
                parts.append(obs[idx])
            if use[1]:
                parts.append(np.eye(Da, dtype=np.float32)[acti[idx]] if discrete else act[idx])
            if use[2]:
                parts.append(nxt[idx])
            if use[3]:
                parts.append(dones[idx].astype(np.float32)[:, None])
            ref = np.concatenate(parts, 1)
            ld = (ref.shape[1] + 3) // 4 * 4
            X = th.full((300, ld), float("nan"), device=DEV)
            # keep every device buffer alive across the call (raw pointers carry no ownership)
            d_obs, d_act, d_acti, d_nxt = dev(obs), dev(act), dev(acti, th.int64), dev(nxt)
            d_done, d_idx = dev(dones.astype(np.uint8), th.uint8), dev(idx, th.int64)
            L.call("ia_gather_concat", L.ptr(d_obs), None if discrete else L.ptr(d_act),
                   L.ptr(d_acti) if discrete else None, L.ptr(d_nxt), L.ptr(d_done), L.ptr(d_idx), 200, Do, Da, *use,
                   L.ptr(X), ld, 100, L.stream())
            got = X.cpu().numpy()
            assert np.array_equal(got[100:300, :ref.shape[1]], ref)
            assert np.all(got[100:300, ref.shape[1]:] == 0)
            assert np.all(np.isnan(got[:100]))


@pytest.mark.parametrize("n0,n1,discrete,use", [(200, 200, False, (1, 1, 0, 0)), (300, 213, True, (1, 1, 1, 1)),
                                                (64, 64, False, (1, 1, 1, 0))])
def test_airl_prepare_and_stats_merge_bit_exact(n0, n1, discrete, use):
    """`ia_airl_prepare` (assembly of the base / next-state / state batches + their RunningNorm slab moments in one
    launch) and `ia_airl_stats_merge` (the three train-mode updates in one launch) against the per-matrix kernels they
    replace: every output bit for bit."""
    rng = np.random.default_rng(1)
    N, Do, Da = 700, 11, 5
    R = n0 + n1
    mk = lambda: dict(obs=dev(rng.standard_normal((N, Do)).astype(np.float32)),
                      nxt=dev(rng.standard_normal((N, Do)).astype(np.float32) * 2 + 1),
                      act=dev(rng.uniform(-1, 1, (N, Da)).astype(np.float32)),
                      acti=dev(rng.integers(0, Da, N), th.int64), done=dev((rng.random(N) < 0.3).astype(np.uint8), th.uint8))
    t0, t1 = mk(), mk()
    i0, i1 = dev(rng.integers(0, N, n0), th.int64), None          # generator source: the first n1 rows (null index)
    Db = use[0] * Do + use[1] * Da + use[2] * Do + use[3]
    ldb, ldp = (Db + 3) // 4 * 4, (Do + 3) // 4 * 4
    ref = dict(Xb=th.zeros(R, ldb, device=DEV), Sn=th.zeros(R, ldp, device=DEV), Sc=th.zeros(R, ldp, device=DEV),
               d4=th.zeros(R, 4, device=DEV))
    for t, idx, n, row in ((t0, i0, n0, 0), (t1, i1, n1, n0)):
        af, ai = (None, L.ptr(t["acti"])) if discrete else (L.ptr(t["act"]), None)
        for obs, flags, X, ld in ((t["obs"], use, ref["Xb"], ldb), (t["nxt"], (1, 0, 0, 0), ref["Sn"], ldp),
                                  (t["obs"], (1, 0, 0, 0), ref["Sc"], ldp), (t["obs"], (0, 0, 0, 1), ref["d4"], 4)):
            L.call("ia_gather_concat", L.ptr(obs), af, ai, L.ptr(t["nxt"]), L.ptr(t["done"]), L.ptr(idx), n, Do, Da,
                   *flags, L.ptr(X), ld, row, L.stream())
    nrn = -(-R // 256)
    got = dict(Xb=th.zeros(R, ldb, device=DEV), Sn=th.zeros(R, ldp, device=DEV), Sc=th.zeros(R, ldp, device=DEV),
               dones=th.empty(R, device=DEV))
    wsg = {k: th.full((nrn * 2 * D,), float("nan"), device=DEV) for k, D in (("b", Db), ("n", Do), ("c", Do))}
    acts = lambda t: (None, L.ptr(t["acti"])) if discrete else (L.ptr(t["act"]), None)
    pol_obs = th.full((R, Do), float("nan"), device=DEV)
    pol_act = th.full((R, 1 if discrete else Da), float("nan"), device=DEV)
    L.call("ia_airl_prepare", L.ptr(t0["obs"]), *acts(t0), L.ptr(t0["nxt"]), L.ptr(t0["done"]), L.ptr(i0), n0,
           L.ptr(t1["obs"]), *acts(t1), L.ptr(t1["nxt"]), L.ptr(t1["done"]), L.ptr(i1), n1, Do, Da, *use,
           L.ptr(got["Xb"]), ldb, L.ptr(got["Sn"]), L.ptr(got["Sc"]), ldp, L.ptr(got["dones"]), L.ptr(wsg["b"]),
           L.ptr(wsg["n"]), L.ptr(wsg["c"]), L.ptr(pol_obs), L.ptr(pol_act), L.stream())
    for k in ("Xb", "Sn", "Sc"):
        assert th.equal(got[k], ref[k]), k
    assert th.equal(got["dones"], ref["d4"][:, 0])
    assert th.equal(pol_obs, ref["Sc"][:, :Do])
    rows = lambda t, idx, n: (t if idx is None else t[idx])[:n]
    want_act = th.cat([rows(t0["acti" if discrete else "act"], i0, n0), rows(t1["acti" if discrete else "act"], i1, n1)])
    assert th.equal(pol_act, want_act.float().reshape(R, -1))
    # statistics: three stand-alone updates (base; potential with the next-state batch, then with the state batch)
    st = lambda D: [th.linspace(-1, 1, D, device=DEV), th.linspace(0.5, 2, D, device=DEV),
                    th.full((), 1000, dtype=th.int32, device=DEV)]
    rb, rp, gb, gp = st(Db), st(Do), st(Db), st(Do)
    for X, ld, D, (m, v, c) in ((ref["Xb"], ldb, Db, rb), (ref["Sn"], ldp, Do, rp)):
        ws = th.empty(int(L.load().ia_running_norm_ws_floats(R, D)), device=DEV)
        L.call("ia_running_norm_update", L.ptr(X), ld, R, D, L.ptr(m), L.ptr(v), L.ptr(c), L.ptr(ws), L.stream())
    snap_ref = th.stack([rp[0].clone(), rp[1].clone()])
    ws = th.empty(int(L.load().ia_running_norm_ws_floats(R, Do)), device=DEV)
    L.call("ia_running_norm_update", L.ptr(ref["Sc"]), ldp, R, Do, L.ptr(rp[0]), L.ptr(rp[1]), L.ptr(rp[2]), L.ptr(ws),
           L.stream())
    snap, ticket = th.empty(2, Do, device=DEV), th.zeros(1, dtype=th.int32, device=DEV)
    L.call("ia_airl_stats_merge", L.ptr(wsg["b"]), L.ptr(wsg["n"]), L.ptr(wsg["c"]), 1, 0, R, Db, Do, L.ptr(gb[0]), L.ptr(gb[1]),
           L.ptr(gb[2]), L.ptr(gp[0]), L.ptr(gp[1]), L.ptr(gp[2]), L.ptr(snap), L.ptr(ticket), L.stream())
    for a, b in zip(gb + gp + [snap], rb + rp + [snap_ref]):
        assert th.equal(a, b)
    assert int(ticket.item()) == 0 and int(gp[2].item()) == 1000 + 2 * R


@pytest.mark.parametrize("R,ne", [(16384, 8192), (128, 64), (7, 0), (10, 10)])
def test_bce_logits_and_stats(R, ne):
    from oracle.imitation_restated import compute_train_stats
    logits = rnd(R, seed=3, scale=2.0).requires_grad_(True)
    labels = th.cat([th.ones(ne), th.zeros(R - ne)])
    scale = 0.25
    loss = F.binary_cross_entropy_with_logits(logits, labels) * scale
    loss.backward()
    ref = compute_train_stats(logits.detach(), labels.long(), loss.detach())
    dl, st = th.empty(R, device=DEV), th.empty(8, device=DEV)
    ws = th.zeros(int(L.load().ia_bce_ws_floats(R)), device=DEV)
    for _ in range(2):  # twice: the ticket word must be back at zero after a call
        L.call("ia_bce_logits", dptr(logits.detach()), R, ne, scale, L.ptr(dl), L.ptr(st), L.ptr(ws), L.stream())
    s = st.cpu().numpy()
    th.testing.assert_close(dl.cpu(), logits.grad, rtol=1e-5, atol=1e-9)
    assert abs(s[0] - ref["disc_loss"]) <= 1e-5 * max(1, abs(ref["disc_loss"]))
    assert s[1] / R == pytest.approx(ref["disc_acc"], abs=1e-7)          # integer counts: exact
    if ne > 0:
        assert s[2] / ne == pytest.approx(ref["disc_acc_expert"], abs=1e-7)
    assert s[3] / max(1, R - ne) == pytest.approx(ref["disc_acc_gen"], abs=1e-7)
    assert (R - s[4]) / R == pytest.approx(ref["disc_proportion_expert_pred"], abs=1e-7)
    assert s[5] / R == pytest.approx(ref["disc_entropy"], rel=1e-4)
    assert (s[6], s[7]) == (ne, R - ne)


def test_airl_logits_and_grad_routing():
    R = 1000
    g, hc, hn, lp = (rnd(R, seed=i).requires_grad_(i < 3) for i in range(4))
    done = (rnd(R, seed=9) > 0.5)
    gamma = 0.97
    f = g + gamma * ((1 - done.float()) * hn) - hc
    logits = f - lp
    dl = rnd(R, seed=11)
    logits.backward(dl)
    out = th.empty(R, device=DEV)
    dd = dev(done.float())  # the AIRL kernels take 0/1 fp32 dones (as gathered by ia_gather_concat)
    L.call("ia_airl_logits", dptr(g.detach()), dptr(hc.detach()), dptr(hn.detach()), L.ptr(dd),
           dptr(lp), gamma, R, L.ptr(out), L.stream())
    th.testing.assert_close(out.cpu(), logits.detach(), rtol=1e-6, atol=1e-6)
    dg, dhc, dhn = (th.empty(R, device=DEV) for _ in range(3))
    L.call("ia_airl_route_grad", dptr(dl), L.ptr(dd), gamma, R, L.ptr(dg), L.ptr(dhc), L.ptr(dhn), L.stream())
    th.testing.assert_close(dg.cpu(), g.grad)
    th.testing.assert_close(dhc.cpu(), hc.grad)
    th.testing.assert_close(dhn.cpu(), hn.grad, rtol=1e-6, atol=1e-7)


# ------------------------------------------------------------------ generator-side kernels


def _oracle_policy(D, A, H, discrete, norm, seed=0):
    from imitation_amd import spaces
    from oracle import imitation_restated as o
    from oracle import sb3_restated as sb
    th.manual_seed(seed)
    os_ = spaces.Box(-np.inf, np.inf, (D,), np.float32)
    as_ = spaces.Discrete(A) if discrete else spaces.Box(-1, 1, (A,), np.float32)
    kw = dict(features_extractor_class=o.NormalizeFeaturesExtractor) if norm else {}
    pol = sb.ActorCriticPolicy(os_, as_, lambda _: 3e-4, net_arch=[H, H], **kw)
    with th.no_grad():  # ortho init leaves the action head tiny; make the test less degenerate
        for p in pol.parameters():
            p.add_(0.1 * th.randn_like(p))
    if norm:
        pol.features_extractor.normalize.running_mean.copy_(th.randn(D) * 0.3)
        pol.features_extractor.normalize.running_var.copy_(th.rand(D) + 0.5)
        pol.features_extractor.normalize.count.fill_(1000)
    return pol


class DevPolicy:
    """Device mirror of an oracle policy (test helper; the product wrapper lives in imitation_amd)."""

    def __init__(self, pol, D, A, H, discrete, norm):
        self.d = L.PolicyDesc(D, A, H, int(discrete), int(norm), 1e-5)
        flat = th.cat([p.detach().reshape(-1) for p in pol.parameters()])
        assert L.load().ia_policy_param_count(C.byref(self.d)) == flat.numel()
        self.P = dev(flat)
        self.Pt = th.empty_like(self.P)
        L.call("ia_policy_transpose", C.byref(self.d), L.ptr(self.P), L.ptr(self.Pt), L.stream())
        if norm:
            rn = pol.features_extractor.normalize
            self.nm, self.nv = dev(rn.running_mean), dev(rn.running_var)
            self.nc = rn.count.clone().to(DEV, th.int32)
        else:
            self.nm = self.nv = self.nc = None
        self.m, self.v = th.zeros_like(self.P), th.zeros_like(self.P)


@pytest.mark.parametrize("D,A,H,discrete,norm,n", [(17, 6, 32, False, True, 1024), (4, 2, 64, True, False, 100),
                                                   (11, 3, 64, False, False, 65), (27, 8, 32, False, True, 7),
                                                   (64, 16, 32, True, True, 130)])
def test_policy_act_and_evaluate(D, A, H, discrete, norm, n):
    pol = _oracle_policy(D, A, H, discrete, norm)
    pol.set_training_mode(False)
    dp = DevPolicy(pol, D, A, H, discrete, norm)
    obs = rnd(n, D, seed=4)
    aw = 1 if discrete else A
    noise = th.rand(n, generator=th.Generator().manual_seed(5)) if discrete else rnd(n, A, seed=5)
    low, high = dev(-th.ones(A)), dev(th.ones(A))
    acts, clip = th.empty(n, aw, device=DEV), th.empty(n, aw, device=DEV)
    vals, logp = th.empty(n, device=DEV), th.empty(n, device=DEV)
    L.call("ia_policy_act", C.byref(dp.d), L.ptr(dp.P), L.ptr(dp.Pt), L.ptr(dp.nm), L.ptr(dp.nv), dptr(obs), n,
           dptr(noise), L.ptr(low), L.ptr(high), L.ptr(acts), L.ptr(clip), L.ptr(vals), L.ptr(logp), L.stream())
    with th.no_grad():
        feats = pol.extract_features(obs, pol.features_extractor)
        lat_pi, lat_vf = pol.mlp_extractor(feats)
        v_ref = pol.value_net(lat_vf).flatten()
        head = pol.action_net(lat_pi)
    th.testing.assert_close(vals.cpu(), v_ref, rtol=1e-4, atol=1e-5)
    a_cpu = acts.cpu()
    if not discrete:
        a_ref = head + noise * pol.log_std.exp()
        th.testing.assert_close(a_cpu, a_ref.detach(), rtol=1e-4, atol=1e-5)
        assert th.equal(clip.cpu(), a_cpu.clamp(-1, 1))  # clipping itself is exact
        act_for_eval = a_cpu
    else:
        probs = th.softmax(head, 1)
        cdf = probs.cumsum(1)
        picked = a_cpu.flatten().long()
        # inverse-CDF sampling: u must fall in the picked bin (boundaries within fp32 slack)
        lo = th.where(picked > 0, cdf.gather(1, (picked - 1).clamp(min=0)[:, None]).flatten(), th.zeros(n))
        hi = cdf.gather(1, picked[:, None]).flatten()
        assert th.all(noise >= lo - 1e-5) and th.all((noise <= hi + 1e-5) | (picked == A - 1))
        act_for_eval = picked
    with th.no_grad():
        v2, lp_ref, ent_ref = pol.evaluate_actions(obs, act_for_eval)
    th.testing.assert_close(logp.cpu(), lp_ref, rtol=1e-4, atol=2e-5)
    lp2, v_out, ent = th.empty(n, device=DEV), th.empty(n, device=DEV), th.empty(n, device=DEV)
    L.call("ia_policy_evaluate", C.byref(dp.d), L.ptr(dp.P), L.ptr(dp.Pt), L.ptr(dp.nm), L.ptr(dp.nv),
           dptr(obs), dptr(act_for_eval.float().reshape(n, aw)), n, L.ptr(lp2), L.ptr(v_out), L.ptr(ent),
           L.stream())
    th.testing.assert_close(lp2.cpu(), lp_ref, rtol=1e-4, atol=2e-5)
    th.testing.assert_close(v_out.cpu(), v2.flatten(), rtol=1e-4, atol=1e-5)
    th.testing.assert_close(ent.cpu(), ent_ref, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("T,n", [(16, 1024), (1000, 64), (1, 3), (5, 1)])
def test_gae_bit_exact(T, n):
    from imitation_amd import spaces
    from oracle import sb3_restated as sb
    rng = np.random.default_rng(T * 1000 + n)
    buf = sb.RolloutBuffer(T, spaces.Box(-1, 1, (2,)), spaces.Box(-1, 1, (1,)), gamma=0.99, gae_lambda=0.95, n_envs=n)
    buf.rewards[:] = rng.standard_normal((T, n))
    buf.values[:] = rng.standard_normal((T, n))
    buf.episode_starts[:] = rng.random((T, n)) < 0.1
    last_v = th.as_tensor(rng.standard_normal(n).astype(np.float32))
    dones = rng.random(n) < 0.2
    buf.compute_returns_and_advantage(last_v, dones)
    adv, ret = th.empty(T, n, device=DEV), th.empty(T, n, device=DEV)
    L.call("ia_gae", dptr(buf.rewards), dptr(buf.values), dptr(buf.episode_starts),
           dptr(last_v), dptr(dones.astype(np.float32)), T, n, 0.99, 0.95, L.ptr(adv), L.ptr(ret),
           L.stream())
    # same fp32 operations in the same order, no FMA contraction: bit-exact
    assert np.array_equal(adv.cpu().numpy(), buf.advantages)
    assert np.array_equal(ret.cpu().numpy(), buf.returns)


def test_timeout_bootstrap():
    r, v = rnd(1000, seed=1), rnd(1000, seed=2)
    tr = (rnd(1000, seed=3) > 0.8)
    rd = dev(r)
    L.call("ia_timeout_bootstrap", L.ptr(rd), dptr(v), dptr(tr.numpy().astype(np.uint8), th.uint8), 0.99,
           1000, L.stream())
    ref = r.clone()
    ref[tr] += np.float32(0.99) * v[tr]
    assert th.equal(rd.cpu(), ref)


@pytest.mark.parametrize("D,A,H,discrete,norm,T,n,bs", [(17, 6, 32, False, True, 16, 64, 256),
                                                        (4, 2, 64, True, False, 8, 16, 50),
                                                        (11, 3, 32, False, False, 4, 32, 128),
                                                        (5, 4, 64, False, True, 3, 7, 21),
                                                        (4, 3, 32, True, True, 8, 16, 40),
                                                        (64, 16, 32, False, True, 4, 32, 64),
                                                        (33, 1, 32, False, False, 2, 70, 140),
                                                        # Discrete head with 13 actions: every lane of a row's quad
                                                        # owns several of them in the loss phase
                                                        (20, 13, 32, True, True, 4, 32, 96),
                                                        # Ant-shaped (BASELINE config 3): 4209 parameters, the
                                                        # 9-parameters-per-thread build of the persistent kernel
                                                        (27, 8, 32, False, True, 8, 64, 256),
                                                        # large (data-parallel-sized) minibatches: 32 / 128
                                                        # gradient blocks, two-level slab reduction
                                                        (17, 6, 32, False, True, 16, 256, 2048),
                                                        (17, 6, 32, False, True, 8, 2048, 8192),
                                                        # SB3's default MlpPolicy (64 x 64) at config-P minibatch size and
                                                        # with a short last minibatch: the one-launch-per-epoch kernel
                                                        # (minibatch steps as phases between grid barriers)
                                                        (17, 6, 64, False, True, 16, 256, 1024),
                                                        (4, 2, 64, True, True, 9, 100, 384),
                                                        # ... and beyond 1 024 rows (64-row blocks on eight waves; up to 1 024:
                                                        # 32-row blocks), second observation-width class
                                                        (27, 8, 64, False, True, 8, 512, 2048),
                                                        # ... down to ONE row block per minibatch (SB3's default batch_size
                                                        # = 64: two workgroups, each owning half the parameters -- the
                                                        # 20-per-thread chunk form) and two row blocks
                                                        (4, 2, 64, True, True, 16, 8, 64),
                                                        (17, 6, 64, False, True, 8, 32, 128),
                                                        # minibatches of <= 16 rows (the reference's tuned AIRL file): the
                                                        # one-workgroup kernel whose waves of rows 16.. idle; Ant width,
                                                        # and a Discrete head with a short last minibatch
                                                        (27, 8, 32, False, True, 4, 12, 16),
                                                        (4, 3, 32, True, True, 4, 10, 16)])
@pytest.mark.parametrize("path", ["epoch", "epochs_one_call", "epoch_whole", "epoch_barriers", "epoch_ll", "update",
                                  "update_spread", "update_shard1"])
def test_ppo_epochs_match_oracle(D, A, H, discrete, norm, T, n, bs, path):
    """Two PPO epochs on a synthetic rollout: parameters, Adam state, RunningNorm state and the
    logged loss statistics against SB3-restated `PPO.train` on the same permutations -- through
    one `ia_ppo_epoch` call per epoch, and through the single persistent `ia_ppo_update` launch
    (working blocks packed on one XCD, or spread over all of them). `update_shard1`: the row-sharded data-parallel form
    (`ia_ppo_update_sharded`) with a world of one -- the whole exchange machinery (8-byte value / sequence words through
    the peer block, loss statistics through the record tails, two launches on a growing sequence base) on one process;
    two ranks run in tests/test_distributed.py."""
    if path in ("epoch_whole", "epoch_barriers", "epoch_ll") and H != 64:
        pytest.skip("forms of the 64-wide one-launch epoch only: whole row-block workgroups / grid barriers / round 4's "
                    "gradient body in the word-exchange kernel (default there: `ppo_epoch_ll2_kernel`, path `epoch`)")
    if path == "epochs_one_call" and (T * n) % bs != 0:
        pytest.skip("`ia_ppo_epochs` runs the epochs as one sequence of minibatches: whole minibatches per epoch only")
    if path not in ("epoch", "epochs_one_call", "epoch_whole", "epoch_barriers", "epoch_ll") and H != 32:
        pytest.skip("the persistent update covers hidden = 32")
    from imitation_amd import spaces
    from oracle import imitation_restated as o
    from oracle import sb3_restated as sb
    pol_ref = _oracle_policy(D, A, H, discrete, norm, seed=3)
    dp = DevPolicy(pol_ref, D, A, H, discrete, norm)
    rng = np.random.default_rng(0)
    aw = 1 if discrete else A
    os_ = spaces.Box(-np.inf, np.inf, (D,), np.float32)
    as_ = spaces.Discrete(A) if discrete else spaces.Box(-1, 1, (A,), np.float32)
    algo = sb.PPO(sb.ActorCriticPolicy, None, n_steps=T, batch_size=bs, n_epochs=2, ent_coef=0.05, _init_setup_model=False)
    algo.observation_space, algo.action_space, algo.n_envs = os_, as_, n
    algo.policy = pol_ref
    algo.lr_schedule = sb.constant_fn(3e-4)
    algo.clip_range = sb.constant_fn(0.2)
    algo._logger = sb.Logger(None, [])
    buf = sb.RolloutBuffer(T, os_, as_, gamma=0.99, gae_lambda=0.95, n_envs=n)
    buf.observations[:] = rng.standard_normal((T, n, D))
    buf.actions[:] = rng.integers(0, A, (T, n, 1)) if discrete else rng.standard_normal((T, n, A))
    buf.values[:] = rng.standard_normal((T, n))
    buf.advantages[:] = rng.standard_normal((T, n))
    buf.returns[:] = buf.advantages + buf.values
    with th.no_grad():
        pol_ref.set_training_mode(False)
        acts_t = th.as_tensor(buf.actions.reshape(T * n, aw))
        _, lp, _ = pol_ref.evaluate_actions(th.as_tensor(buf.observations.reshape(T * n, D)),
                                            acts_t.long().flatten() if discrete else acts_t)
    buf.log_probs[:] = (lp.numpy() + 0.1 * rng.standard_normal(T * n)).reshape(T, n)
    buf.full = True
    algo.rollout_buffer = buf
    # (one time slice more than T behind the observations, as the rollout tile has: `ia_ppo_update*` read rows in 16-byte pieces,
    #  up to 12 bytes past the last row -- include/imitation_hip.h; an exact-size tensor at the end of a mapped region faulted)
    d_obs = dev(np.concatenate([buf.observations, np.zeros_like(buf.observations[:1])]))[:T]
    d_act = dev(buf.actions)
    d_lp, d_adv, d_ret = dev(buf.log_probs), dev(buf.advantages), dev(buf.returns)

    np.random.seed(123)
    perms = [np.random.permutation(T * n) for _ in range(2)]
    np.random.seed(123)
    algo.train()  # oracle: 2 epochs

    ws = th.empty(int(L.load().ia_ppo_ws_floats(C.byref(dp.d), bs, (2 if path == "epochs_one_call" else 1) * T * n)),
                  device=DEV)
    n_mb = -(-T * n // bs)
    stats = th.zeros(2, n_mb, 8, device=DEV)
    if path == "epochs_one_call":   # both epochs as one sequence of minibatches (`ia_ppo_epochs`)
        d_perm = th.as_tensor(np.stack(perms)).to(DEV)
        L.call("ia_ppo_epochs", C.byref(dp.d), L.ptr(dp.P), L.ptr(dp.Pt), L.ptr(dp.nm), L.ptr(dp.nv), L.ptr(dp.nc),
               int(norm), L.ptr(d_obs), L.ptr(d_act), L.ptr(d_lp), L.ptr(d_adv), L.ptr(d_ret), L.ptr(d_perm), 2, T, n, bs,
               1, 0.2, 0.05, 0.5, 0.5, L.ptr(dp.m), L.ptr(dp.v), 3e-4, 0.9, 0.999, 1e-5, 0, L.ptr(ws), L.ptr(stats),
               L.stream())
        th.cuda.synchronize()
    elif path in ("epoch", "epoch_whole", "epoch_barriers", "epoch_ll"):
        L.load().ia_ppo_epoch_split({"epoch": 0, "epoch_whole": 2, "epoch_barriers": 3, "epoch_ll": 4}[path])
        try:
            for e in range(2):
                L.call("ia_ppo_epoch", C.byref(dp.d), L.ptr(dp.P), L.ptr(dp.Pt), L.ptr(dp.nm), L.ptr(dp.nv), L.ptr(dp.nc),
                       int(norm), L.ptr(d_obs), L.ptr(d_act), L.ptr(d_lp), L.ptr(d_adv), L.ptr(d_ret),
                       dptr(perms[e], th.int64), T, n, bs, 1, 0.2, 0.05, 0.5, 0.5, L.ptr(dp.m), L.ptr(dp.v), 3e-4, 0.9,
                       0.999, 1e-5, e * n_mb, L.ptr(ws), L.ptr(stats[e]), L.stream())
            th.cuda.synchronize()
        finally:
            L.load().ia_ppo_epoch_split(0)
    elif path == "update_shard1":
        from imitation_amd.distributed import PeerExchange
        nws = int(L.load().ia_ppo_update_sharded_ws_floats(C.byref(dp.d), bs, 1))
        if nws == 0:
            pytest.skip("shape not covered by the persistent kernel (parameter copies do not fit LDS)")
        uws = th.zeros(nws, device=DEV)
        d_perm = th.as_tensor(np.stack(perms)).to(DEV)
        ex = PeerExchange.loopback(1, dp.d)
        assert ex.ok
        try:
            for e in range(2):   # one launch per epoch: the sequence base carries over
                L.call("ia_ppo_update_sharded", C.byref(dp.d), L.ptr(dp.P), L.ptr(dp.Pt), L.ptr(dp.nm), L.ptr(dp.nv),
                       L.ptr(dp.nc), int(norm), L.ptr(d_obs), L.ptr(d_act), L.ptr(d_lp), L.ptr(d_adv), L.ptr(d_ret),
                       L.ptr(d_perm[e:e + 1]), 1, T, n, bs, 1, 0.2, 0.05, 0.5, 0.5, L.ptr(dp.m), L.ptr(dp.v), 3e-4, 0.9, 0.999,
                       1e-5, e * n_mb, L.ptr(uws), L.ptr(stats[e]), 1, 0, ex.take_steps(n_mb), ex.recv, ex.peer_recv, 1, 5.0,
                       L.stream())
            th.cuda.synchronize()
            assert int(uws[8:9].view(th.int32).item()) == 0, "a wait timed out"
        finally:
            th.cuda.synchronize()
            ex.close()
    else:
        nws = int(L.load().ia_ppo_update_ws_floats(C.byref(dp.d), bs))
        if nws == 0:
            pytest.skip("shape not covered by the persistent kernel (parameter copies do not fit LDS)")
        uws = th.zeros(nws, device=DEV)
        d_perm = th.as_tensor(np.stack(perms)).to(DEV)
        L.load().ia_ppo_update_xcd_pack(1 if path == "update" else 0)
        try:
            L.call("ia_ppo_update", C.byref(dp.d), L.ptr(dp.P), L.ptr(dp.Pt), L.ptr(dp.nm), L.ptr(dp.nv), L.ptr(dp.nc),
                   int(norm), L.ptr(d_obs), L.ptr(d_act), L.ptr(d_lp), L.ptr(d_adv), L.ptr(d_ret), L.ptr(d_perm), 2, T,
                   n, bs, 1, 0.2, 0.05, 0.5, 0.5, L.ptr(dp.m), L.ptr(dp.v), 3e-4, 0.9, 0.999, 1e-5, 0, L.ptr(uws),
                   L.ptr(stats), L.stream())
            th.cuda.synchronize()
        finally:
            L.load().ia_ppo_update_xcd_pack(0)
        assert int(uws[8:9].view(th.int32).item()) == 0, "grid wait timed out"
    th.cuda.synchronize()
    flat_ref = th.cat([p.detach().reshape(-1) for p in pol_ref.parameters()])
    k = 2 * n_mb  # optimiser steps taken; tolerance scaled as the reference's own test does
    th.testing.assert_close(dp.P.cpu(), flat_ref, rtol=(1 + k) * 1e-5, atol=(1 + k) * 2e-6)
    if norm:
        rn = pol_ref.features_extractor.normalize
        assert int(dp.nc.item()) == int(rn.count)
        th.testing.assert_close(dp.nm.cpu(), rn.running_mean, rtol=1e-4, atol=1e-5)
        th.testing.assert_close(dp.nv.cpu(), rn.running_var, rtol=1e-4, atol=1e-5)
    lg = algo.logger.name_to_value
    st = stats.cpu().numpy().reshape(-1, 8)
    assert st[:, 0].mean() == pytest.approx(lg["train/policy_gradient_loss"], rel=2e-3, abs=2e-5)
    assert st[:, 1].mean() == pytest.approx(lg["train/value_loss"], rel=2e-3)
    assert st[:, 2].mean() == pytest.approx(lg["train/entropy_loss"], rel=2e-3)
    assert st[n_mb:, 3].mean() == pytest.approx(lg["train/approx_kl"], rel=5e-3, abs=1e-6)
    assert st[:, 4].mean() == pytest.approx(lg["train/clip_fraction"], abs=2.0 / (T * n))
    assert st[-1, 5] == pytest.approx(lg["train/loss"], rel=2e-3, abs=2e-5)
    # the transposed copies must track the parameters
    Pt2 = th.empty_like(dp.P)
    L.call("ia_policy_transpose", C.byref(dp.d), L.ptr(dp.P), L.ptr(Pt2), L.stream())
    assert th.equal(Pt2, dp.Pt)


@pytest.mark.parametrize("D,A,discrete,T,n,bs", [(17, 6, False, 16, 256, 1024), (4, 2, True, 9, 100, 384),
                                                  (27, 8, False, 8, 512, 2048), (40, 3, False, 10, 100, 448)])
def test_word_exchange_epoch_is_bit_identical_to_the_barrier_form(D, A, discrete, T, n, bs):
    """64-wide towers: `ppo_epoch_ll_kernel` (`ia_ppo_epoch_split(4)`: slabs, partial sums of squares and new parameters handed
    over as 8-byte value / sequence words, no grid barrier) against `ppo_epoch_persistent_kernel<64, true>` (`ia_ppo_epoch_split(3)`):
    the same sums in the same order -> parameters, transposed copy, Adam moments and logged statistics bit for bit over
    three epochs (the sequence numbers continue from call to call; observation widths of all three poll-size classes)."""
    pol_ref = _oracle_policy(D, A, 64, discrete, True, seed=5)
    rng = np.random.default_rng(1)
    aw = 1 if discrete else A
    obs = rng.standard_normal((T, n, D)).astype(np.float32)
    acts = (rng.integers(0, A, (T, n, 1)) if discrete else rng.standard_normal((T, n, A))).astype(np.float32)
    lp, adv = rng.standard_normal((T, n)).astype(np.float32) * 0.1 - 1.0, rng.standard_normal((T, n)).astype(np.float32)
    ret = rng.standard_normal((T, n)).astype(np.float32)
    d_obs, d_act, d_lp, d_adv, d_ret = dev(obs), dev(acts), dev(lp), dev(adv), dev(ret)
    n_mb = -(-T * n // bs)
    perms = [rng.permutation(T * n) for _ in range(3)]
    outs = []
    try:
        for mode in (3, 4, 0, 0):
            dp = DevPolicy(pol_ref, D, A, 64, discrete, True)
            ws = th.empty(int(L.load().ia_ppo_ws_floats(C.byref(dp.d), bs, T * n)), device=DEV)
            ws.uniform_(-1e30, 1e30)   # (the workspace arrives uninitialised)
            stats = th.zeros(3, n_mb, 8, device=DEV)
            L.load().ia_ppo_epoch_split(mode)
            for e in range(3):
                L.call("ia_ppo_epoch", C.byref(dp.d), L.ptr(dp.P), L.ptr(dp.Pt), L.ptr(dp.nm), L.ptr(dp.nv), L.ptr(dp.nc),
                       1, L.ptr(d_obs), L.ptr(d_act), L.ptr(d_lp), L.ptr(d_adv), L.ptr(d_ret),
                       dptr(perms[e], th.int64), T, n, bs, 1, 0.2, 0.05, 0.5, 0.5, L.ptr(dp.m), L.ptr(dp.v), 3e-4, 0.9,
                       0.999, 1e-5, e * n_mb, L.ptr(ws), L.ptr(stats[e]), L.stream())
            th.cuda.synchronize()
            assert int(ws[5:6].view(th.int32).item()) == 0, "a wait timed out"
            outs.append((dp.P.clone(), dp.Pt.clone(), dp.m.clone(), dp.v.clone(), stats.clone(), dp.nm.clone(), dp.nv.clone()))
    finally:
        L.load().ia_ppo_epoch_split(0)
    names = ("parameters", "transposed copy", "exp_avg", "exp_avg_sq", "statistics", "norm mean", "norm var")
    for name, x, y in zip(names, outs[0], outs[1]):
        assert th.equal(x, y), f"{name}: {int((x != y).sum())} of {x.numel()} differ, max {float((x - y).abs().max()):.3e}"
    assert float(outs[0][0].abs().sum()) > 0 and bool(th.isfinite(outs[1][0]).all())
    # the default kernel (tower-resident parameters, transposed chain: other summation orders; observation widths up to 32,
    # beyond them the word-exchange kernel again) agrees within the tolerance of the oracle comparison
    k = 3 * n_mb
    for name, x, y in zip(names, outs[0], outs[2]):
        th.testing.assert_close(x, y, rtol=(1 + k) * 1e-5, atol=(1 + k) * 2e-6, msg=lambda m, name=name: f"{name}: {m}")
    assert bool(th.isfinite(outs[2][0]).all())
    # ... and is deterministic: the same launch sequence again from the same state, bit for bit
    assert len(outs) == 4, "the default form runs twice"
    for name, x, y in zip(names, outs[2], outs[3]):
        assert th.equal(x, y), f"default form, run to run, {name}: {int((x != y).sum())} of {x.numel()} differ"


@pytest.mark.parametrize("D,A,discrete,T,n,bs", [(17, 6, False, 16, 256, 1024), (4, 2, True, 9, 100, 384),
                                                  (27, 8, False, 8, 512, 2048), (11, 3, False, 8, 32, 64),
                                                  (16, 16, False, 5, 77, 200), (32, 5, True, 6, 64, 128)])
def test_epoch_chain_forms_agree(D, A, discrete, T, n, bs):
    """64-wide towers, `ppo_epoch_ll2_kernel`'s forms: 64-row blocks on eight waves (`ia_ppo_epoch_split(6)`: every wave half of
    a layer's output tiles, two waves per SIMD; the form of 1 025 - 2 048-row minibatches), 32-row blocks on four waves (7:
    feature halves) and on eight (0, the default up to 1 024 rows: feature QUARTERS). Every tile is accumulated by the same MFMAs
    in the same order in 7 and 0, so with a clip threshold nothing reaches (the sum of squares is folded by twice as many waves:
    the norm may differ in its last bit) parameters, transposed copy, Adam moments and the loss statistics are bit-identical over
    three epochs, and agree to the last few bits with the default threshold. The 64-row form (the rows' contraction in one slab
    per 64 rows instead of two) agrees with them within the tolerance of the oracle comparison; the default is deterministic.
    (Round 5's four-wave 64-row form, to which 6 was bit-identical in the same sense, was retired: `profiles/r06_mlp64.md`.)"""
    pol_ref = _oracle_policy(D, A, 64, discrete, True, seed=5)
    rng = np.random.default_rng(1)
    obs = rng.standard_normal((T, n, D)).astype(np.float32)
    acts = (rng.integers(0, A, (T, n, 1)) if discrete else rng.standard_normal((T, n, A))).astype(np.float32)
    lp, adv = rng.standard_normal((T, n)).astype(np.float32) * 0.1 - 1.0, rng.standard_normal((T, n)).astype(np.float32)
    ret = rng.standard_normal((T, n)).astype(np.float32)
    d_obs, d_act, d_lp, d_adv, d_ret = dev(obs), dev(acts), dev(lp), dev(adv), dev(ret)
    n_mb = -(-T * n // bs)
    perms = [rng.permutation(T * n) for _ in range(3)]
    names = ("parameters", "transposed copy", "exp_avg", "exp_avg_sq", "norm mean", "norm var")
    k = 3 * n_mb
    for max_norm in (1e9, 0.5):
        outs = []
        try:
            for mode in (6, 0, 0, 7):
                dp = DevPolicy(pol_ref, D, A, 64, discrete, True)
                ws = th.empty(int(L.load().ia_ppo_ws_floats(C.byref(dp.d), bs, T * n)), device=DEV)
                ws.uniform_(-1e30, 1e30)   # (the workspace arrives uninitialised)
                stats = th.zeros(3, n_mb, 8, device=DEV)
                L.load().ia_ppo_epoch_split(mode)
                for e in range(3):
                    L.call("ia_ppo_epoch", C.byref(dp.d), L.ptr(dp.P), L.ptr(dp.Pt), L.ptr(dp.nm), L.ptr(dp.nv), L.ptr(dp.nc),
                           1, L.ptr(d_obs), L.ptr(d_act), L.ptr(d_lp), L.ptr(d_adv), L.ptr(d_ret),
                           dptr(perms[e], th.int64), T, n, bs, 1, 0.2, 0.05, 0.5, max_norm, L.ptr(dp.m), L.ptr(dp.v), 3e-4, 0.9,
                           0.999, 1e-5, e * n_mb, L.ptr(ws), L.ptr(stats[e]), L.stream())
                th.cuda.synchronize()
                assert int(ws[5:6].view(th.int32).item()) == 0, "a wait timed out"
                outs.append(((dp.P.clone(), dp.Pt.clone(), dp.m.clone(), dp.v.clone(), dp.nm.clone(), dp.nv.clone()), stats.clone()))
        finally:
            L.load().ia_ppo_epoch_split(0)
        (xs, st_x), (zs, st_z), (zs2, st_z2), (hs, st_h) = outs
        assert float(xs[0].abs().sum()) > 0 and bool(th.isfinite(xs[0]).all()) and bool(th.isfinite(zs[0]).all())
        for name, x, z in zip(names, xs, zs):   # 64-row blocks against the default: other slabs, same gradient within rounding
            th.testing.assert_close(x, z, rtol=(1 + k) * 1e-5, atol=(1 + k) * 2e-6, msg=lambda m, name=name: f"64-row form, {name}: {m}")
        th.testing.assert_close(st_x, st_z, rtol=1e-4, atol=1e-5)
        for name, z, z2 in zip(names, zs, zs2):
            assert th.equal(z, z2), f"default form, run to run, {name}"
        assert th.equal(st_z, st_z2)
        if max_norm > 1e8:   # quarters against halves on the same 32-row blocks (64-row blocks when they do not apply: 6 == 6)
            for name, z, hh in zip(names, zs, hs):
                assert th.equal(z, hh), f"quarters / halves, {name}: {int((z != hh).sum())} of {z.numel()} differ"
            assert th.equal(st_z[..., :6], st_h[..., :6])
        else:
            for name, z, hh in zip(names, zs, hs):
                th.testing.assert_close(z, hh, rtol=(1 + k) * 1e-6, atol=(1 + k) * 2e-7, msg=lambda m, name=name: f"quarters / halves, {name}: {m}")


@pytest.mark.parametrize("shape,B,A", [((4, 36, 36), 8, 6), ((3, 44, 52), 5, 4), ((1, 36, 40), 33, 18),
                                       ((4, 84, 84), 5, 6), ((4, 44, 60), 3, 4)])
def test_cnn_policy_forward_and_gradient_match_torch(shape, B, A):
    """NatureCNN actor-critic policy on the HIP path (im2col + MFMA GEMMs, col2im, Categorical head) against
    the SB3-restated torch policy with the same weights: values / log-probs / entropies, and the gradient of
    `c_lp * sum(logp) + c_ent * sum(entropy)` w.r.t. every parameter against torch autograd."""
    from imitation_amd import spaces
    from imitation_amd.cnn_policy import ActorCriticCnnPolicy
    from oracle import sb3_restated as sb

    osp, asp = spaces.Box(0, 255, shape, np.uint8), spaces.Discrete(A)
    th.manual_seed(5)
    ref = sb.ActorCriticCnnPolicy(osp, asp, lambda _: 1.0)
    pol = ActorCriticCnnPolicy(osp, asp, lambda _: 1.0).to(DEV)
    pol.load_state_dict(ref.state_dict())
    for k, v in ref.state_dict().items():                      # layout round trip (linear.0 columns are permuted)
        assert th.equal(pol.state_dict()[k].cpu(), v), k
    rng = np.random.default_rng(1)
    obs = rng.integers(0, 256, (B, *shape), dtype=np.uint8)
    acts = rng.integers(0, A, B)
    c_lp, c_ent = -0.7 / B, -0.01 / B
    vals, logp, ent = pol.evaluate_actions(obs, acts, logp_coef=c_lp, ent_coef=c_ent, want_grad=True)
    grad = th.zeros_like(pol._flat)
    pol.backward(B, grad)
    rv, rl, re = ref.evaluate_actions(th.as_tensor(obs), th.as_tensor(acts))
    (c_lp * rl.sum() + c_ent * re.sum()).backward()
    th.testing.assert_close(logp.cpu(), rl.detach(), rtol=2e-5, atol=2e-5)
    th.testing.assert_close(ent.cpu(), re.detach(), rtol=2e-5, atol=2e-5)
    th.testing.assert_close(vals.cpu(), rv.detach(), rtol=2e-5, atol=2e-5)
    named = dict(ref.named_parameters())
    for i, name in enumerate(pol._names):
        if name == "value_net":
            continue   # no gradient reaches the value head from this loss (torch leaves .grad None)
        o, n, ob, nb = pol._offsets[i]
        gw = pol._to_torch_layout(i, grad[o:o + n]).reshape(pol._shapes[i]).cpu()
        rw, rb = named[f"{name}.weight"].grad, named[f"{name}.bias"].grad
        scale = float(rw.abs().max()) + 1e-12
        assert float((gw - rw).abs().max()) <= 2e-5 * scale + 1e-8, (name, float((gw - rw).abs().max()), scale)
        assert float((grad[ob:ob + nb].cpu() - rb).abs().max()) <= 2e-5 * (float(rb.abs().max()) + 1e-12) + 1e-8, name


@pytest.mark.parametrize("shape,B", [((4, 84, 84), 7), ((4, 36, 36), 19), ((4, 44, 60), 4)])
def test_implicit_first_layer_equals_the_column_buffer_path(shape, B):
    """`csrc/conv1_implicit.hip` (A operand formed from the uint8 frames in LDS) and `ia_gemm_f32_im2col` (layers 2 / 3:
    the GEMM reads its operand through the im2col view of the activations) -- no column buffers -- against the
    explicit im2col + GEMM path of the same policy: activations of the first layer and the gradient of every parameter
    (the two paths sum the same products in different orders: tolerance 2e-5 of the array's scale)."""
    from imitation_amd import spaces
    from imitation_amd.cnn_policy import ActorCriticCnnPolicy

    osp, asp = spaces.Box(0, 255, shape, np.uint8), spaces.Discrete(5)
    th.manual_seed(2)
    pol = ActorCriticCnnPolicy(osp, asp, lambda _: 1.0).to(DEV)
    assert pol.implicit_conv1, "this shape is covered by the implicit kernels"
    rng = np.random.default_rng(3)
    obs = rng.integers(0, 256, (B, *shape), dtype=np.uint8)
    acts = rng.integers(0, 5, B)
    assert pol.implicit_convs, "NatureCNN's layers 2 and 3 are covered by the implicit-im2col GEMMs"
    res = {}
    for mode in ("implicit", "implicit+dgrad", "explicit"):   # (+dgrad: input gradients as implicit transposed convolutions)
        pol.implicit_conv1 = pol.implicit_convs = mode != "explicit"
        pol.implicit_dgrad = mode == "implicit+dgrad"
        pol._bufs = {}
        vals, logp, _ = pol.evaluate_actions(obs, acts, logp_coef=-0.5 / B, ent_coef=-0.02 / B, want_grad=True)
        grad = th.zeros_like(pol._flat)
        pol.backward(B, grad)
        assert ("dcol1" in pol._bufs[B]) == (mode != "implicit+dgrad")
        res[mode] = (pol._bufs[B]["act0"].clone(), vals.clone(), logp.clone(), grad)
    for mode in ("implicit", "implicit+dgrad"):
        for x, y in zip(res[mode], res["explicit"]):
            scale = float(y.abs().max()) + 1e-12
            assert float((x - y).abs().max()) <= 2e-5 * scale + 1e-8, (mode, float((x - y).abs().max()), scale)
