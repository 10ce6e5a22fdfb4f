"""Data-parallel path: world_size-2 `gloo` tests. The CPU test checks the collective logic on host
tensors; the GPU test runs two ranks of the full HIP GAIL trainer on ONE MI355X (gloo staging the
buckets through the host) and checks the replicas stay bit-identical and the moment merge matches a
single-process update on the concatenated batch."""
import os
import socket
import tempfile

import numpy as np
import pytest
import torch as th
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _init(rank, world, port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)


def _cpu_worker(rank, world, port, out_dir):
    _init(rank, world, port)
    from imitation_amd.distributed import DataParallel, merge_moments_reference
    dp = DataParallel()
    g = th.Generator().manual_seed(rank)
    flat = th.randn(1000, generator=g)
    local = flat.clone()
    dp.allreduce_mean_(flat)
    everyone = [th.randn(1000, generator=th.Generator().manual_seed(r)) for r in range(world)]
    assert th.allclose(flat, sum(everyone) / world, atol=1e-6)
    gathered = dp.all_gather_flat(local)
    assert th.equal(gathered, th.cat(everyone))
    t = th.full((5,), float(rank))
    dp.broadcast_([t])
    assert th.all(t == 0)
    # moment merge == moments of the concatenated batch
    x = th.randn(300, 7, generator=th.Generator().manual_seed(50 + rank)) * (1 + rank) + rank
    means = dp.all_gather_flat(x.mean(0)).reshape(world, 7)
    m2s = dp.all_gather_flat(((x - x.mean(0)) ** 2).sum(0)).reshape(world, 7)
    mean, var = merge_moments_reference(means, m2s, th.full((world,), 300))
    allx = th.cat([th.randn(300, 7, generator=th.Generator().manual_seed(50 + r)) * (1 + r) + r for r in range(world)])
    assert th.allclose(mean, allx.mean(0), atol=1e-5) and th.allclose(var, allx.var(0, unbiased=False), atol=1e-4)
    open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
    dist.destroy_process_group()


def test_data_parallel_collectives_gloo_cpu(tmp_path):
    port = _free_port()
    mp.spawn(_cpu_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert os.path.exists(tmp_path / "ok0") and os.path.exists(tmp_path / "ok1")


def _gpu_worker(rank, world, port, out_dir):
    _init(rank, world, port)
    th.cuda.set_device(0)
    th.set_num_threads(1)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import imitation_amd as p
    from imitation_amd.distributed import DataParallel
    from tests import harness
    cfg = dict(harness.CASES["gail_box"], rounds=2)
    from imitation_amd.vec_env import SyntheticVecEnv

    def run(batch_moments: bool, global_mb: bool = False, pipeline: bool = True, airl: bool = False,
            general: bool = False):
        th.manual_seed(100 + rank)      # different initial weights per rank: the broadcast must fix that
        np.random.seed(100 + rank)
        venv = SyntheticVecEnv(num_envs=cfg["n_envs"], obs_dim=17, act_dim=6, horizon=cfg["horizon"], seed=rank)
        pk = dict(features_extractor_class=p.NormalizeFeaturesExtractor,
                  features_extractor_kwargs=dict(normalize_class=p.RunningNorm))
        if general:   # towers outside the fused kernels' shapes: the general minibatch loop, one gradient all-reduce per step
            pk = dict(pk, net_arch=dict(pi=[24], vf=[16, 16]))
        algo = p.PPO(p.ActorCriticPolicy if general else p.FeedForward32Policy, venv, n_steps=cfg["n_steps"],
                     batch_size=cfg["ppo_batch"], n_epochs=2, ent_coef=0.1, policy_kwargs=pk, device="cuda")
        algo.dp_batch_moments = batch_moments
        algo.dp_global_minibatch = global_mb
        algo.dp_update_form = "sharded"   # (the default, "auto", times both forms first: its own test below)
        if airl:
            net = p.BasicShapedRewardNet(venv.observation_space, venv.action_space, reward_hid_sizes=(32,),
                                         potential_hid_sizes=(32, 32), use_next_state=True,
                                         normalize_input_layer=p.RunningNorm)
        else:
            net = p.BasicRewardNet(venv.observation_space, venv.action_space, hid_sizes=(32, 32),
                                   normalize_input_layer=p.RunningNorm)
        demos = p.Transitions(**harness.make_demo_arrays(cfg, seed=1 + rank))
        tr = (p.AIRL if airl else p.GAIL)(
            demonstrations=demos, demo_batch_size=64, venv=venv, gen_algo=algo, reward_net=net,
            n_disc_updates_per_round=2, custom_logger=p.configure_logger(tempfile.mkdtemp(), []),
            data_parallel=DataParallel())
        tr.pipeline_rounds = pipeline
        tr.train((cfg["rounds"] + 1) * cfg["n_envs"] * cfg["n_steps"] if global_mb else
                 cfg["rounds"] * cfg["n_envs"] * cfg["n_steps"])
        th.cuda.synchronize()
        sd = {f"disc/{k}": v.cpu() for k, v in tr._reward_net.state_dict().items()}
        sd.update({f"pol/{k}": v.cpu() for k, v in algo.policy.state_dict().items()})
        if global_mb:  # the gathered tile of the last update and this rank's own shard of it
            g, rb = algo._dpg, algo.rollout_buffer
            # rows sharded over the ranks, records exchanged inside the kernels through hipIpc-mapped memory (two
            # processes on one GPU): every update of this run took that path
            assert g["shard"] is not None and g["shard"]["ex"].ok and not g["shard"]["ex"].loop
            assert algo.dp_sharded_updates == cfg["rounds"] + 1
            sd["tile/obs_global"], sd["tile/adv_global"] = g["obs"].cpu(), g["adv"].cpu()
            sd["tile/obs_local"], sd["tile/adv_local"] = rb.obs[: rb.buffer_size].cpu(), rb.adv.cpu()
            sd["tile/perm"] = g["perm_dev"].cpu()
        return sd

    th.save(run(True), os.path.join(out_dir, f"state{rank}.pt"))
    th.save(run(False), os.path.join(out_dir, f"state{rank}_per_minibatch.pt"))
    th.save(run(True, global_mb=True), os.path.join(out_dir, f"state{rank}_global.pt"))
    th.save(run(True, global_mb=True, pipeline=False), os.path.join(out_dir, f"state{rank}_global_seq.pt"))
    th.save(run(True, global_mb=True, airl=True), os.path.join(out_dir, f"state{rank}_airl.pt"))
    th.save(run(True, global_mb=True, pipeline=False, airl=True), os.path.join(out_dir, f"state{rank}_airl_seq.pt"))
    th.save(run(True, general=True), os.path.join(out_dir, f"state{rank}_general.pt"))
    dist.destroy_process_group()


@pytest.mark.gpu
def test_two_ranks_one_gpu_replicas_identical(tmp_path):
    if not th.cuda.is_available():
        pytest.skip("no GPU")
    port = _free_port()
    mp.spawn(_gpu_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a, b = th.load(tmp_path / "state0.pt"), th.load(tmp_path / "state1.pt")
    assert set(a) == set(b)
    for k in a:
        assert th.equal(a[k], b[k]), k  # bit-identical replicas (same reduced grads, same merged moments)
    # one all-gather of all minibatches' feature-norm moments per PPO update == one per minibatch
    c = th.load(tmp_path / "state0_per_minibatch.pt")
    for k in a:
        assert th.equal(a[k], c[k]), k
    # global-minibatch update (no per-step collective): replicas identical, every rank holds the same
    # gathered tile [T, world*n] = [rank 0 envs | rank 1 envs], and the same permutations
    g0, g1 = th.load(tmp_path / "state0_global.pt"), th.load(tmp_path / "state1_global.pt")
    for k in g0:
        if not k.startswith("tile/") or k.endswith("_global") or k == "tile/perm":
            assert th.equal(g0[k], g1[k]), k
    n = g0["tile/obs_local"].shape[1]
    assert th.equal(g0["tile/obs_global"][:, :n], g0["tile/obs_local"]) and th.equal(g0["tile/obs_global"][:, n:],
                                                                                      g1["tile/obs_local"])
    assert th.equal(g0["tile/adv_global"][:, n:], g1["tile/adv_local"])
    perm = g0["tile/perm"]
    assert perm.shape[1] == 2 * g0["tile/adv_local"].numel() and th.equal(perm.sort(dim=1).values[0],
                                                                           th.arange(perm.shape[1]))
    assert not th.equal(g0["pol/action_net.weight"], a["pol/action_net.weight"])   # it did train differently
    # rounds pipelined across the rollout (discriminator collectives behind the env stepping) == sequential
    q0 = th.load(tmp_path / "state0_global_seq.pt")
    for k in g0:
        assert th.equal(g0[k], q0[k]), k
    # AIRL under data parallelism: per-update feature statistics from the (all-gathered) merge snapshots;
    # replicas identical, pipelined == sequential
    r0, r1, rs = (th.load(tmp_path / f) for f in ("state0_airl.pt", "state1_airl.pt", "state0_airl_seq.pt"))
    for k in r0:
        if not k.startswith("tile/") or k.endswith("_global") or k == "tile/perm":
            assert th.equal(r0[k], r1[k]), k
        assert th.equal(r0[k], rs[k]), k
    # general-tower policies (any `net_arch`) under data parallelism: gradient all-reduce per optimiser step, feature
    # statistics merged across ranks -> identical replicas, and the policy did train
    t0, t1 = th.load(tmp_path / "state0_general.pt"), th.load(tmp_path / "state1_general.pt")
    assert "pol/mlp_extractor.value_net.2.weight" in t0
    for k in t0:
        assert th.equal(t0[k], t1[k]), k
    assert all(bool(th.isfinite(v.float()).all()) for v in t0.values())
    # every rank contributed: disc input norm saw world * (2 rounds * 2 updates * 128 rows)
    assert int(a["disc/mlp.normalize_input.count"]) == 2 * 2 * 2 * 128
    assert int(g0["disc/mlp.normalize_input.count"]) == 2 * 3 * 2 * 128


# ---------------------------------------------------------------------------------------------------
# What single-process run does a world-W run equal?
#   PPO   : n_envs' = W * n_envs (the env batches side by side, rank order), batch_size' = W * batch_size,
#           the data-parallel run's own permutation stream -- [SB3 PPO.train] on that rollout;
#   disc  : demo_batch_size' = W * demo_batch_size with expert / generator batches = the ranks' batches
#           concatenated -- `train_disc` on that batch (mean-reduced BCE: the rank-mean of per-rank mean
#           gradients IS the gradient of the global mean; moments merged over all rows).

# PPO geometries of the equivalence test: the harness case itself (one 64-row gradient workgroup per rank: the kernel form
# whose gradient stays in LDS) and a wide one -- 64 envs x 16 steps per rank, PPO minibatch 512 per rank: eight gradient
# workgroups per rank, global minibatches of 1 024 rows whose statistics are taken in 512-row slices, records travelling
# in four pieces per peer
# ... and config P's own per-rank geometry (bench.py: 1 024 envs x 16 steps, PPO minibatch 1 024 = sixteen gradient
# workgroups + three per rank, global minibatches of world x 1 024 rows)
_EQUIV_GEOM = {"one_workgroup": {}, "eight_workgroups": dict(n_envs=64, ppo_batch=512),
               "config_p": dict(n_envs=1024, ppo_batch=1024)}
# (world, geometry, data-parallel form of the PPO update): world 4 is the first time more than two writers meet in the
# exchange's parity slots and the rank-order slice sums (4 x 11 resident workgroups sharded, 4 x 35 replicated: co-resident
# on one GPU)
_EQUIV_RUNS = [(2, "one_workgroup", "sharded"), (2, "eight_workgroups", "sharded"), (2, "config_p", "sharded"),
               (2, "config_p", "replicated"), (4, "eight_workgroups", "sharded"), (4, "eight_workgroups", "replicated"),
               (4, "one_workgroup", "sharded"),
               # the last rank reports that it cannot map peer memory (test-only `IA_PEER_FAIL_RANK`): all four ranks must
               # land on the replicated update and still equal the single process
               (4, "one_workgroup", "handshake_fails")]


def _equiv_worker(rank, world, port, out_dir, geom="one_workgroup", form="sharded"):
    _init(rank, world, port)
    th.cuda.set_device(0)
    th.set_num_threads(1)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import imitation_amd as p
    from imitation_amd.distributed import DataParallel
    from imitation_amd.vec_env import SyntheticVecEnv
    from tests import harness
    cfg = dict(harness.CASES["gail_box"], **_EQUIV_GEOM[geom])
    th.manual_seed(100 + rank)
    np.random.seed(100 + rank)
    venv = SyntheticVecEnv(num_envs=cfg["n_envs"], obs_dim=17, act_dim=6, horizon=cfg["horizon"], seed=rank)
    pk = dict(features_extractor_class=p.NormalizeFeaturesExtractor,
              features_extractor_kwargs=dict(normalize_class=p.RunningNorm))
    algo = p.PPO(p.FeedForward32Policy, venv, n_steps=cfg["n_steps"], batch_size=cfg["ppo_batch"], n_epochs=2,
                 ent_coef=0.1, policy_kwargs=pk, device="cuda")
    algo.dp_global_minibatch = True
    algo.dp_update_form = "sharded" if form == "handshake_fails" else form
    if form == "handshake_fails":
        os.environ["IA_PEER_FAIL_RANK"] = str(world - 1)
    net = p.BasicRewardNet(venv.observation_space, venv.action_space, hid_sizes=(32, 32),
                           normalize_input_layer=p.RunningNorm)
    demos = p.Transitions(**harness.make_demo_arrays(cfg, seed=1 + rank))
    tr = p.GAIL(demonstrations=demos, demo_batch_size=64, venv=venv, gen_algo=algo, reward_net=net,
                n_disc_updates_per_round=2, custom_logger=p.configure_logger(tempfile.mkdtemp(), []),
                data_parallel=DataParallel())
    tr.pipeline_rounds = False
    out = {}
    # ---- discriminator: three updates on explicit per-rank batches
    cpu = lambda sd: {k: v.detach().cpu().clone() for k, v in sd.items()}
    out["disc_pre"] = cpu(tr._reward_net.state_dict())
    rng = np.random.default_rng(200 + rank)
    out["batches"] = []
    for _ in range(3):
        mk = lambda: dict(obs=rng.standard_normal((64, 17)).astype(np.float32),
                          acts=rng.uniform(-1, 1, (64, 6)).astype(np.float32),
                          next_obs=rng.standard_normal((64, 17)).astype(np.float32), dones=rng.random(64) < 0.1)
        e, g = mk(), mk()
        out["batches"].append((e, g))
        tr.train_disc(expert_samples=e, gen_samples=g)
    out["disc_post"] = cpu(tr._reward_net.state_dict())
    out["pol_norm_post_disc"] = cpu(algo.policy.features_extractor.normalize.state_dict())
    # ---- generator: one rollout + the data-parallel PPO update on the gathered tile
    orig_train = algo.train

    def hooked(*a, **k):
        th.cuda.synchronize()
        out["pol_pre"] = cpu(algo.policy.state_dict())
        return orig_train(*a, **k)

    algo.train = hooked
    tr.train_gen()
    th.cuda.synchronize()
    out["pol_post"] = cpu(algo.policy.state_dict())
    g = algo._dpg
    if form == "sharded":
        # the update above sharded each global minibatch's rows over the ranks (records exchanged inside the kernels)
        assert g["shard"] is not None and algo.dp_sharded_updates == 1 and not algo.dp_handshake_failed
    else:
        assert g["shard"] is None and getattr(algo, "dp_sharded_updates", 0) == 0   # whole global minibatch on every rank
        assert algo.dp_handshake_failed == (form == "handshake_fails") and g["forms"] == ["replicated"]
    out["tile"] = {k: g[k].cpu().clone() for k in ("obs", "acts", "logp", "adv", "ret")}
    out["perm"] = g["perm_dev"].cpu().clone()
    out["stats"] = algo._stats_dev.cpu().clone() if algo._records is None else algo._records[(algo._rec_i - 1) % 2].stats.cpu().clone()
    out["hyper"] = dict(T=cfg["n_steps"], n=cfg["n_envs"], bs=cfg["ppo_batch"], n_epochs=2, ent_coef=0.1)
    th.save(out, os.path.join(out_dir, f"equiv{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("world,geom,form", _EQUIV_RUNS, ids=[f"w{w}-{g}-{f}" for w, g, f in _EQUIV_RUNS])
def test_world_n_equals_single_process_on_the_concatenated_batch(tmp_path, world, geom, form):
    """`world` ranks sharing the box's one GPU (gloo collectives, hipIpc-mapped exchange areas) == ONE process on the
    ranks' batches side by side: the PPO update against [SB3 PPO.train] restated on the concatenated rollout, the
    discriminator against a single-process `train_disc` on the concatenated batches. Unmeasured on multi-GPU hardware."""
    if not th.cuda.is_available():
        pytest.skip("no GPU")
    port = _free_port()
    W = world
    mp.spawn(_equiv_worker, args=(W, port, str(tmp_path), geom, form), nprocs=W, join=True)
    rs = [th.load(tmp_path / f"equiv{r}.pt", weights_only=False) for r in range(W)]
    r0 = rs[0]
    for r1 in rs[1:]:
        for k in r0["disc_post"]:
            assert th.equal(r0["disc_post"][k], r1["disc_post"][k]), k
        for k in r0["pol_post"]:
            assert th.equal(r0["pol_post"][k], r1["pol_post"][k]), k

    # ---- (1) PPO: world W == [SB3 PPO.train] restated, single process, on the side-by-side env batch with
    #          batch_size W x (per-rank batch) and the data-parallel run's permutations
    from imitation_amd import spaces
    from oracle import imitation_restated as o
    from oracle import sb3_restated as sb
    h = r0["hyper"]
    T, n2, D, A = h["T"], W * h["n"], 17, 6
    os_, as_ = spaces.Box(-np.inf, np.inf, (D,), np.float32), spaces.Box(-1, 1, (A,), np.float32)
    pol = sb.ActorCriticPolicy(os_, as_, lambda _: 3e-4, net_arch=[32, 32], features_extractor_class=o.NormalizeFeaturesExtractor)
    pol.load_state_dict(r0["pol_pre"])
    algo = sb.PPO(sb.ActorCriticPolicy, None, n_steps=T, batch_size=W * h["bs"], n_epochs=h["n_epochs"],
                  ent_coef=h["ent_coef"], _init_setup_model=False)
    algo.observation_space, algo.action_space, algo.n_envs = os_, as_, n2
    algo.policy = pol
    algo.lr_schedule, algo.clip_range = sb.constant_fn(3e-4), sb.constant_fn(0.2)
    algo._logger = sb.Logger(None, [])
    buf = sb.RolloutBuffer(T, os_, as_, gamma=0.99, gae_lambda=0.95, n_envs=n2)
    t = r0["tile"]
    buf.observations[:] = t["obs"].numpy()
    buf.actions[:] = t["acts"].numpy()
    buf.log_probs[:] = t["logp"].numpy()
    buf.advantages[:] = t["adv"].numpy()
    buf.returns[:] = t["ret"].numpy()
    buf.values[:] = buf.returns - buf.advantages
    buf.full = True
    algo.rollout_buffer = buf
    perms = [p_.numpy() for p_ in r0["perm"]]
    real = np.random.permutation
    it = iter(perms)
    np.random.permutation = lambda n_: next(it)        # the shared-seed stream of the data-parallel run
    try:
        algo.train()
    finally:
        np.random.permutation = real
    k = h["n_epochs"] * (T * n2 // (W * h["bs"]))
    # the logged loss statistics travel in the records' tails: all ranks hold the same rows, equal to the oracle's means
    for r1 in rs[1:]:
        assert th.equal(r0["stats"], r1["stats"])
    st, lg = r0["stats"].numpy().reshape(-1, 8), algo.logger.name_to_value
    assert st[:, 0].mean() == pytest.approx(lg["train/policy_gradient_loss"], rel=2e-3, abs=2e-5)
    assert st[:, 1].mean() == pytest.approx(lg["train/value_loss"], rel=2e-3)
    assert st[:, 2].mean() == pytest.approx(lg["train/entropy_loss"], rel=2e-3)
    assert st[-1, 5] == pytest.approx(lg["train/loss"], rel=2e-3, abs=2e-5)
    for name, ref in pol.state_dict().items():
        got = r0["pol_post"][name]
        if name.endswith("count"):
            assert int(got) == int(ref), name
        else:
            th.testing.assert_close(got.float(), ref.float(), rtol=(1 + k) * 1e-5, atol=(1 + k) * 2e-6, msg=name)

    # ---- (2) discriminator: world W == one process with demo_batch_size W x 64 on the concatenated batches
    import imitation_amd as p
    from imitation_amd.vec_env import SyntheticVecEnv
    from tests import harness
    cfg = harness.CASES["gail_box"]
    th.manual_seed(100)
    np.random.seed(100)   # rank 0's construction seeds: its initial weights are what the broadcast spread
    venv = SyntheticVecEnv(num_envs=cfg["n_envs"], obs_dim=17, act_dim=6, horizon=cfg["horizon"], seed=0)
    pk = dict(features_extractor_class=p.NormalizeFeaturesExtractor,
              features_extractor_kwargs=dict(normalize_class=p.RunningNorm))
    a1 = p.PPO(p.FeedForward32Policy, venv, n_steps=cfg["n_steps"], batch_size=cfg["ppo_batch"], n_epochs=2,
               ent_coef=0.1, policy_kwargs=pk, device="cuda")
    net = p.BasicRewardNet(venv.observation_space, venv.action_space, hid_sizes=(32, 32),
                           normalize_input_layer=p.RunningNorm)
    tr = p.GAIL(demonstrations=p.Transitions(**harness.make_demo_arrays(cfg, seed=1)), demo_batch_size=W * 64, venv=venv,
                gen_algo=a1, reward_net=net, n_disc_updates_per_round=2,
                custom_logger=p.configure_logger(str(tmp_path / "single"), []))
    for kk, v in tr._reward_net.state_dict().items():
        assert th.equal(v.cpu(), r0["disc_pre"][kk]), kk          # same starting point
    cat = lambda parts: {kk: np.concatenate([a[kk] for a in parts]) for kk in parts[0]}
    for per_rank in zip(*[r["batches"] for r in rs]):
        tr.train_disc(expert_samples=cat([e for e, _ in per_rank]), gen_samples=cat([g_ for _, g_ in per_rank]))
    for kk, v in tr._reward_net.state_dict().items():
        if kk.endswith("count"):
            assert int(v) == int(r0["disc_post"][kk]), kk
        else:
            th.testing.assert_close(r0["disc_post"][kk], v.cpu(), rtol=2e-4, atol=5e-5, msg=kk)
    rn = a1.policy.features_extractor.normalize.state_dict()
    for kk, v in rn.items():   # the policy-feature-norm side effect of the updates (App. C.2) merges over all rows too
        if kk.endswith("count"):
            assert int(v) == int(r0["pol_norm_post_disc"][kk]), kk
        else:
            th.testing.assert_close(r0["pol_norm_post_disc"][kk], v.cpu(), rtol=2e-4, atol=5e-5, msg=kk)


def _rccl_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    th.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=th.device("cuda", 0))
    from imitation_amd.distributed import DataParallel
    dp = DataParallel()
    assert not dp._stage, "nccl (= RCCL) works on device buffers in place"
    dp._min_world = 1                     # issue the collectives although one rank needs none
    flat = th.arange(1000, dtype=th.float32, device="cuda")
    ref = flat.clone()
    dp.allreduce_mean_(flat)
    assert th.equal(flat, ref)
    g = dp.all_gather_flat(ref)
    assert g.is_cuda and th.equal(g, ref)
    t = th.full((7,), 3.0, device="cuda")
    dp.broadcast_([t])
    assert th.all(t == 3.0)
    s = dp.shared_seed()
    assert 0 <= s < 2 ** 31
    th.cuda.synchronize()
    open(os.path.join(out_dir, "rccl_ok"), "w").write("ok")
    dist.destroy_process_group()


@pytest.mark.gpu
def test_rccl_branches_execute_on_one_gpu(tmp_path):
    """Backend `nccl` (= RCCL on ROCm) with a world of one on the box's single GPU: every `DataParallel` method
    goes through its device-buffer branch (all_reduce, all_gather_into_tensor, broadcast, the shared seed on a
    device tensor) -- the code the 8-GPU run executes, minus the second peer."""
    if not th.cuda.is_available():
        pytest.skip("no GPU")
    port = _free_port()
    mp.spawn(_rccl_worker, args=(1, port, str(tmp_path)), nprocs=1, join=True)
    assert os.path.exists(tmp_path / "rccl_ok")


# ---------------------------------------------------------------------------------------------------
# Round 3: the FUSED discriminator updates under data parallelism. GAIL's four-launch update
# (`disc_fused.hip`, 256-wide) with the round's slab moments all-gathered once, slab reduce | all-reduce | Adam + weight
# images; AIRL's fused update (`airl_fused.hip`) with the three moment sets all-gathered per update.

def _fused_worker(rank, world, port, out_dir):
    _init(rank, world, port)
    th.cuda.set_device(0)
    th.set_num_threads(1)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import imitation_amd as p
    from imitation_amd import reward_nets
    from imitation_amd.distributed import DataParallel
    from imitation_amd.vec_env import SyntheticVecEnv
    from tests import harness
    cfg = harness.CASES["gail_box"]
    calls = {"gail": 0, "airl": 0}
    orig_adam, orig_finish = reward_nets.BasicRewardNet.fused_adam_step, reward_nets.ShapedRewardNet.fused_finish

    def counting_adam(self, *a, **k):
        calls["gail"] += 1
        return orig_adam(self, *a, **k)

    def counting_finish(self, *a, **k):
        calls["airl"] += 1
        return orig_finish(self, *a, **k)

    reward_nets.BasicRewardNet.fused_adam_step = counting_adam
    reward_nets.ShapedRewardNet.fused_finish = counting_finish

    def run(airl: bool, pipeline: bool, hid=(256, 256), normalize_output: bool = False, module: bool = False,
            gp: float = 0.0, wide: bool = False):
        th.manual_seed(100 + rank)
        np.random.seed(100 + rank)
        venv = SyntheticVecEnv(num_envs=cfg["n_envs"], obs_dim=17, act_dim=6, horizon=cfg["horizon"], seed=rank)
        pk = dict(features_extractor_class=p.NormalizeFeaturesExtractor,
                  features_extractor_kwargs=dict(normalize_class=p.RunningNorm))
        algo = p.PPO(p.FeedForward32Policy, venv, n_steps=cfg["n_steps"], batch_size=cfg["ppo_batch"], n_epochs=2,
                     ent_coef=0.1, policy_kwargs=pk, device="cuda")
        if airl:
            net = p.BasicShapedRewardNet(venv.observation_space, venv.action_space, reward_hid_sizes=(32,),
                                         potential_hid_sizes=(32, 32), use_next_state=True,
                                         normalize_input_layer=p.RunningNorm)
        elif module:   # autograd-capable nn.Module net on the custom ops (the operator boundary), general kernels
            net = p.modules.BasicRewardNet(venv.observation_space, venv.action_space, hid_sizes=(32, 32),
                                           normalize_input_layer=p.modules.RunningNorm)
        else:
            net = p.BasicRewardNet(venv.observation_space, venv.action_space, hid_sizes=hid,
                                   normalize_input_layer=p.RunningNorm,
                                   **(dict(use_next_state=True, use_done=True) if wide else {}))   # wide: 41 inputs
        if normalize_output:   # the scripts' default wrapper: output statistics over the env batches of ALL ranks
            net = p.NormalizedRewardNet(net, p.RunningNorm)
        demos = p.Transitions(**harness.make_demo_arrays(cfg, seed=1 + rank))
        tr = (p.AIRL if airl else p.GAIL)(
            demonstrations=demos, demo_batch_size=128, venv=venv, gen_algo=algo, reward_net=net,
            n_disc_updates_per_round=3, custom_logger=p.configure_logger(tempfile.mkdtemp(), []),
            data_parallel=DataParallel())
        tr.pipeline_rounds = pipeline
        tr.disc_grad_penalty_coef = gp   # (opt-in; > 0: the penalty's tile passes inside the fused 256-wide update)
        tr.train(3 * cfg["n_envs"] * cfg["n_steps"])
        th.cuda.synchronize()
        if gp > 0:
            assert np.isfinite(float(tr.last_grad_penalty)) and float(tr.last_grad_penalty) >= 0
        sd = {f"disc/{k}": v.cpu() for k, v in tr._reward_net.state_dict().items()}
        sd.update({f"pol/{k}": v.cpu() for k, v in algo.policy.state_dict().items()})
        return sd

    for name, kw in (("gail", dict(airl=False, pipeline=True)), ("gail_seq", dict(airl=False, pipeline=False)),
                     ("gail128", dict(airl=False, pipeline=True, hid=(128, 128))),
                     ("gail_gp", dict(airl=False, pipeline=True, gp=2.0)),
                     ("gail_wide", dict(airl=False, pipeline=True, wide=True)),
                     ("gail32", dict(airl=False, pipeline=True, hid=(32, 32))),
                     ("airl", dict(airl=True, pipeline=True)), ("airl_seq", dict(airl=True, pipeline=False)),
                     ("airl_norm", dict(airl=True, pipeline=True, normalize_output=True)),
                     ("module", dict(airl=False, pipeline=True, module=True))):
        before = dict(calls)
        sd = run(**kw)
        sd["_fused_calls"] = th.tensor([calls["gail"] - before["gail"], calls["airl"] - before["airl"]])
        th.save(sd, os.path.join(out_dir, f"fused{rank}_{name}.pt"))
    dist.destroy_process_group()


@pytest.mark.gpu
def test_fused_updates_under_data_parallelism_replicas_identical(tmp_path):
    """Two ranks on one GPU (gloo): the fused GAIL update (H = 256 and 128) and the fused AIRL update keep their
    kernels under data parallelism, replicas stay bit-identical, pipelined == sequential schedule."""
    if not th.cuda.is_available():
        pytest.skip("no GPU")
    port = _free_port()
    mp.spawn(_fused_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    ld = lambda r, n: th.load(tmp_path / f"fused{r}_{n}.pt")
    for name in ("gail", "gail_seq", "gail128", "gail_gp", "gail_wide", "gail32", "airl", "airl_seq", "airl_norm", "module"):
        a, b = ld(0, name), ld(1, name)
        calls = a.pop("_fused_calls"); b.pop("_fused_calls")
        # 3 rounds x 3 updates, every one through the fused kernels (the nn.Module net: through the custom ops)
        assert int(calls[1 if name.startswith("airl") else 0]) == (0 if name == "module" else 9), (name, calls)
        for k in a:
            assert th.equal(a[k], b[k]), (name, k)
        assert all(bool(th.isfinite(v.float()).all()) for v in a.values())
    # the penalty changed the parameters (same seeds otherwise) and went through the fused update + fused Adam launch
    assert any(not th.equal(ld(0, "gail")[k], ld(0, "gail_gp")[k]) for k in ld(0, "gail") if k.startswith("disc/mlp.dense"))
    for name in ("gail", "airl"):
        a, s = ld(0, name), ld(0, name + "_seq")
        for k in a:
            if k != "_fused_calls":
                assert th.equal(a[k], s[k]), (name, k)
    # every rank contributed to the input statistics: world x (3 rounds x 3 updates x 256 rows)
    assert int(ld(0, "gail")["disc/mlp.normalize_input.count"]) == 2 * 3 * 3 * 256
    assert int(ld(0, "module")["disc/mlp.normalize_input.count"]) == 2 * 3 * 3 * 256
    # ... and to the output statistics of NormalizedRewardNet: world x (3 rollouts x 16 steps x 8 envs)
    assert int(ld(0, "airl_norm")["disc/normalize_output_layer.count"]) == 2 * 3 * 16 * 8


# geometry of the fused-update equivalence test: the harness case (8 envs, 128-row expert batches) and config P's per-rank
# geometry (bench.py: 1 024 envs x 16 steps in the ring, 8 192-row expert batches = 16 384-row updates per rank)
_FUSED_GEOM = {"gail_box": dict(demo_batch=128), "config_p": dict(n_envs=1024, demo_batch=8192, n_demo=20000)}


def _fused_equiv_worker(rank, world, port, out_dir, geom="gail_box"):
    _init(rank, world, port)
    th.cuda.set_device(0)
    th.set_num_threads(1)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import imitation_amd as p
    from imitation_amd.distributed import DataParallel
    from imitation_amd.vec_env import SyntheticVecEnv
    from tests import harness
    cfg = dict(harness.CASES["gail_box"], **_FUSED_GEOM[geom])
    th.manual_seed(100 + rank)
    np.random.seed(100 + rank)
    venv = SyntheticVecEnv(num_envs=cfg["n_envs"], obs_dim=17, act_dim=6, horizon=cfg["horizon"], seed=rank)
    pk = dict(features_extractor_class=p.NormalizeFeaturesExtractor,
              features_extractor_kwargs=dict(normalize_class=p.RunningNorm))
    algo = p.PPO(p.FeedForward32Policy, venv, n_steps=cfg["n_steps"], batch_size=cfg["ppo_batch"], n_epochs=2,
                 ent_coef=0.1, policy_kwargs=pk, device="cuda")
    net = p.BasicRewardNet(venv.observation_space, venv.action_space, hid_sizes=(256, 256),
                           normalize_input_layer=p.RunningNorm)
    demos = harness.make_demo_arrays(cfg, seed=1 + rank)
    tr = p.GAIL(demonstrations=p.Transitions(**demos), demo_batch_size=cfg["demo_batch"], venv=venv, gen_algo=algo,
                reward_net=net, n_disc_updates_per_round=3, custom_logger=p.configure_logger(tempfile.mkdtemp(), []),
                data_parallel=DataParallel())
    tr.pipeline_rounds = False
    tr.train_gen()                       # one rollout (+ the data-parallel PPO update): fills the replay ring
    th.cuda.synchronize()
    cpu = lambda sd: {k: v.detach().cpu().clone() for k, v in sd.items()}
    out = dict(disc_pre=cpu(tr._reward_net.state_dict()),
               pol_norm_pre=cpu(algo.policy.features_extractor.normalize.state_dict()),
               demos=demos, ring={k: v.copy() for k, v in tr._gen_replay_buffer._arrays.items()},
               ring_n=tr._gen_replay_buffer.size(), idx=[])
    e_next, g_sample = tr._expert_stream.next_indices, tr._gen_replay_buffer.sample_indices

    def rec_e():
        i = e_next()
        out["idx"].append(("e", np.array(i)))
        return i

    def rec_g(n):
        i = g_sample(n)
        out["idx"].append(("g", np.array(i)))
        return i

    tr._expert_stream.next_indices, tr._gen_replay_buffer.sample_indices = rec_e, rec_g
    algo.policy.set_training_mode(True)   # as PPO.train leaves it
    tr._overlap_k = 0
    pending = tr._disc_round(prepass=True)      # the round-level assembly + three fused updates
    tr._replay_policy_norm_updates()
    tr._finish_disc_round(pending)
    th.cuda.synchronize()
    out["disc_post"] = cpu(tr._reward_net.state_dict())
    out["pol_norm_post"] = cpu(algo.policy.features_extractor.normalize.state_dict())
    th.save(out, os.path.join(out_dir, f"fequiv{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("geom", list(_FUSED_GEOM))
def test_world_two_fused_updates_equal_single_process_on_the_concatenated_batch(tmp_path, geom):
    """The fused path (H = 256) under data parallelism == ONE process running the same fused round on the union of
    the ranks' batches (expert / replay tables and index batches concatenated in rank order)."""
    if not th.cuda.is_available():
        pytest.skip("no GPU")
    port = _free_port()
    mp.spawn(_fused_equiv_worker, args=(2, port, str(tmp_path), geom), nprocs=2, join=True)
    r0 = th.load(tmp_path / "fequiv0.pt", weights_only=False)
    r1 = th.load(tmp_path / "fequiv1.pt", weights_only=False)
    for k in r0["disc_post"]:
        assert th.equal(r0["disc_post"][k], r1["disc_post"][k]), k
        assert th.equal(r0["disc_pre"][k], r1["disc_pre"][k]), k
    assert not th.equal(r0["disc_post"]["mlp.dense0.weight"], r0["disc_pre"]["mlp.dense0.weight"])

    import imitation_amd as p
    from imitation_amd.vec_env import SyntheticVecEnv
    from tests import harness
    cfg = dict(harness.CASES["gail_box"], **_FUSED_GEOM[geom])
    th.manual_seed(100)
    np.random.seed(100)
    venv = SyntheticVecEnv(num_envs=cfg["n_envs"], obs_dim=17, act_dim=6, horizon=cfg["horizon"], seed=0)
    pk = dict(features_extractor_class=p.NormalizeFeaturesExtractor,
              features_extractor_kwargs=dict(normalize_class=p.RunningNorm))
    algo = p.PPO(p.FeedForward32Policy, venv, n_steps=cfg["n_steps"], batch_size=cfg["ppo_batch"], n_epochs=2,
                 ent_coef=0.1, policy_kwargs=pk, device="cuda")
    net = p.BasicRewardNet(venv.observation_space, venv.action_space, hid_sizes=(256, 256),
                           normalize_input_layer=p.RunningNorm)
    cat = lambda a, b: {k: np.concatenate([a[k], b[k]]) for k in a}
    n_demo, cap = len(r0["demos"]["obs"]), r0["ring_n"]
    assert cap == r1["ring_n"] == len(r0["ring"]["obs"])
    tr = p.GAIL(demonstrations=p.Transitions(**cat(r0["demos"], r1["demos"])), demo_batch_size=2 * cfg["demo_batch"], venv=venv,
                gen_algo=algo, reward_net=net, n_disc_updates_per_round=3, gen_replay_buffer_capacity=2 * cap,
                custom_logger=p.configure_logger(str(tmp_path / "single"), []))
    tr._reward_net.load_state_dict(r0["disc_pre"])
    algo.policy.features_extractor.normalize.load_state_dict(r0["pol_norm_pre"])
    tr._gen_replay_buffer.store(p.Transitions(**cat(r0["ring"], r1["ring"])))
    stream = []
    for (k0, i0), (k1, i1) in zip(r0["idx"], r1["idx"]):
        assert k0 == k1
        stream.append((k0, np.concatenate([i0, i1 + (n_demo if k0 == "e" else cap)])))
    it = iter(stream)

    def nxt(kind):
        k, i = next(it)
        assert k == kind
        return i

    tr._expert_stream.next_indices = lambda: nxt("e")
    tr._gen_replay_buffer.sample_indices = lambda n: nxt("g")
    algo.policy.set_training_mode(True)
    tr._overlap_k = 0
    pending = tr._disc_round(prepass=True)
    assert pending is not None
    tr._replay_policy_norm_updates()
    tr._finish_disc_round(pending)
    th.cuda.synchronize()
    for kk, v in tr._reward_net.state_dict().items():
        if kk.endswith("count"):
            assert int(v) == int(r0["disc_post"][kk]) == 3 * 2 * 2 * cfg["demo_batch"], kk
        else:
            th.testing.assert_close(r0["disc_post"][kk], v.cpu(), rtol=2e-4, atol=5e-5, msg=kk)
    for kk, v in algo.policy.features_extractor.normalize.state_dict().items():
        if kk.endswith("count"):
            assert int(v) == int(r0["pol_norm_post"][kk]), kk
        else:
            th.testing.assert_close(r0["pol_norm_post"][kk], v.cpu(), rtol=2e-4, atol=5e-5, msg=kk)


@pytest.mark.gpu
def test_bench_two_rank_launch_line_on_one_gpu(tmp_path):
    """The driver's N > 1 command (`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`) with two
    ranks sharing the box's one GPU (`IA_BENCH_SHARE_GPU=1`: collectives through gloo): config P, data-parallel, fused
    discriminator updates; one JSON line from rank 0."""
    if not th.cuda.is_available():
        pytest.skip("no GPU")
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, IA_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3",
           "--warmup", "2", "--prof-rounds", "1"]
    res = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-4000:]
    line = [l for l in res.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["value"] > 0
    assert out["config"]["parallelism"] == "dp2"
    assert out["details"]["roofline_disc_update"]["path"].startswith("fused")   # not the general 16-launch update
    assert out["summary"]["disc_update"]["path"] == "fused" and out["tail_summary"] == out["summary"]
    # the data-parallel form of the PPO update was a MEASURED choice: both forms ran during the warm-up rounds (rows sharded
    # over the ranks with the in-kernel record exchange through hipIpc memory; the whole global minibatch on every rank)
    ch = out["config"]["ppo_update_choice"]
    assert ch is not None and ch["chosen"] in ("sharded", "replicated") and ch["sharded_ms"] > 0 and ch["replicated_ms"] > 0
    assert out["config"]["ppo_update"].startswith("row-sharded" if ch["chosen"] == "sharded" else "replicated")
    assert all(not isinstance(v, (dict, list)) for v in out["roofline"].values())   # flat: scalars survive any parser


def _bench_line(tmp_path, cmd, extra_env=None):
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, IA_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0", **(extra_env or {}))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    res = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-4000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines          # ONE JSON line, from rank 0
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_bare_command_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2 ...` with no launcher environment (the shape of the driver's N = 1 command with another N):
    the script re-runs itself under `torch.distributed.run`, one rank per GPU (here: two ranks on the one GPU), and rank 0
    prints the one JSON line."""
    if not th.cuda.is_available():
        pytest.skip("no GPU")
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = _bench_line(tmp_path, [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup",
                                 "5", "--prof-rounds", "1"])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["value"] > 0 and out["config"]["parallelism"] == "dp2"
    ch = out["config"]["ppo_update_choice"]   # five warm-up rounds: the four timed trials and the verdict
    assert ch is not None and ch["chosen"] in ("sharded", "replicated")
    by_rank = out["config"]["ppo_update_by_rank"]
    assert [b["rank"] for b in by_rank] == [0, 1] and {b["form"] for b in by_rank} == {ch["chosen"]}
    assert not any(b["handshake_failed"] for b in by_rank)


@pytest.mark.gpu
def test_bench_failed_peer_handshake_lands_every_rank_on_the_replicated_update(tmp_path):
    """One rank cannot export / map peer memory (test-only switch `IA_PEER_FAIL_RANK`): the ranks' common verdict sends
    ALL of them to the replicated update -- no rank runs the sharded kernel against peers that are not there -- and the
    line says so."""
    if not th.cuda.is_available():
        pytest.skip("no GPU")
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = _bench_line(tmp_path, [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup",
                                 "2", "--prof-rounds", "1"], extra_env={"IA_PEER_FAIL_RANK": "1"})
    assert out["n_gpus"] == 2 and out["value"] > 0
    by_rank = out["config"]["ppo_update_by_rank"]
    assert len(by_rank) == 2
    for b in by_rank:
        assert b["form"] == "replicated" and b["handshake_failed"] and b["sharded_updates"] == 0, b
    assert out["config"]["ppo_update"].startswith("replicated") and "handshake failed" in out["config"]["ppo_update"]
    assert out["config"]["ppo_update_choice"] is None
