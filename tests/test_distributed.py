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

    def run(batch_moments: bool, global_mb: bool = False, pipeline: bool = True, airl: bool = False):
        th.manual_seed(100 + rank)      # different initial weights per rank: the broadcast must fix that
        np.random.seed(100 + rank)
        venv = SyntheticVecEnv(num_envs=cfg["n_envs"], obs_dim=17, act_dim=6, horizon=cfg["horizon"], seed=rank)
        pk = dict(features_extractor_class=p.NormalizeFeaturesExtractor,
                  features_extractor_kwargs=dict(normalize_class=p.RunningNorm))
        algo = p.PPO(p.FeedForward32Policy, venv, n_steps=cfg["n_steps"], batch_size=cfg["ppo_batch"], n_epochs=2,
                     ent_coef=0.1, policy_kwargs=pk, device="cuda")
        algo.dp_batch_moments = batch_moments
        algo.dp_global_minibatch = global_mb
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
    # every rank contributed: disc input norm saw world * (2 rounds * 2 updates * 128 rows)
    assert int(a["disc/mlp.normalize_input.count"]) == 2 * 2 * 2 * 128
    assert int(g0["disc/mlp.normalize_input.count"]) == 2 * 3 * 2 * 128
