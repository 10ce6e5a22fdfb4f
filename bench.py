#!/usr/bin/env python
"""Headline benchmark: env-steps/sec of full GAIL rounds (generator rollout + PPO update +
discriminator updates) on synthetic HalfCheetah-shaped data, BASELINE.json config[1].

    python bench.py --gpus N --steps K --warmup W

A "step" is ONE adversarial round (`AdversarialTrainer.train` body, reference
`algorithms/adversarial/common.py:453-461`) at config P of SURVEY 8d: n_envs=1024 per GPU,
n_steps=16 (16 384 transitions/round/GPU), obs 17 / act 6, disc BasicRewardNet 256x256 +
RunningNorm, demo_batch_size 8192 (16 384-row disc updates), 16 disc updates/round, PPO
FeedForward32Policy + NormalizeFeaturesExtractor, minibatch 1024, 10 epochs. Host env workers
(NumPy) stay on the CPU. For N>1 it runs one rank per GPU (env-batch sharded: 1024 envs per
rank, weak scaling) with RCCL gradient / moment all-reduces every optimiser step.

Prints ONE JSON line (rank 0). Extra objects: `roofline` for the kernel with the largest share of GPU
time (the persistent PPO update: a latency chain, so its figure of merit is us_per_step), with the whole
discriminator update (`disc_update`, the throughput kernel family SURVEY 8d's roofline target is about)
and the GEMM family (`gemm`) as siblings, all measured with HIP events on the launch streams right
after the timed region; `variants` = SURVEY 8d's other configurations through the same trainer;
`cpu_baseline` = the oracle (CPU restatement of the reference round, torch CPU ops) timed on this box's
host cores for a few rounds of the same workload.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch as th

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CFG_P = dict(n_envs=1024, n_steps=16, obs_dim=17, act_dim=6, horizon=1000, disc_hid=(256, 256),
             demo_batch=8192, n_disc=16, ppo_batch=1024, n_epochs=10, ent_coef=0.1, lr=3e-4,
             n_demo_traj=64)
PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
DISC_FLOP_PER_UPDATE = 16384 * 418304  # 6.85 GFLOP per 16 384-row update


def make_demos(cfg, seed=1):
    """64 x 1000-step synthetic demonstrations from a fixed random tanh-linear expert."""
    from imitation_amd.vec_env import SyntheticVecEnv
    env = SyntheticVecEnv(num_envs=cfg["n_demo_traj"], obs_dim=cfg["obs_dim"], act_dim=cfg["act_dim"],
                          horizon=cfg["horizon"], seed=seed)
    rng = np.random.default_rng(seed)
    We = rng.standard_normal((cfg["obs_dim"], cfg["act_dim"])).astype(np.float32)
    obs = env.reset()
    O, A, N, D = [], [], [], []
    for _ in range(cfg["horizon"]):
        act = np.tanh(obs @ We).astype(np.float32)
        env.step_async(act)
        nobs, _, dones, nxt, _ = env.step_wait_arrays()
        O.append(obs); A.append(act); N.append(nxt); D.append(dones)
        obs = nobs
    # trajectory-major order like flatten_trajectories
    st = lambda xs: np.stack(xs).swapaxes(0, 1).reshape(-1, *xs[0].shape[1:])
    return dict(obs=st(O), acts=st(A), next_obs=st(N), dones=st(D))


def build_trainer(ns, cfg, device, seed=0, dp=None):
    from imitation_amd.vec_env import SyntheticVecEnv
    th.manual_seed(seed)
    np.random.seed(seed)
    venv = SyntheticVecEnv(num_envs=cfg["n_envs"], obs_dim=cfg["obs_dim"], act_dim=cfg["act_dim"],
                           horizon=cfg["horizon"], seed=seed)
    pk = dict(features_extractor_class=ns.NormalizeFeaturesExtractor,
              features_extractor_kwargs=dict(normalize_class=ns.RunningNorm))
    algo = ns.PPO(ns.FeedForward32Policy, venv, n_steps=cfg["n_steps"], batch_size=cfg["ppo_batch"],
                  n_epochs=cfg["n_epochs"], ent_coef=cfg["ent_coef"], learning_rate=cfg["lr"], seed=seed,
                  policy_kwargs=pk, device=device)
    if dp is not None:
        # the benchmark MEASURES which data-parallel form of the PPO update is faster on this node during its warm-up
        # rounds (the library's default is the row-sharded form, deterministic; see `PPO.dp_update_form`)
        algo.dp_update_form = os.environ.get("IA_DP_UPDATE_FORM", "auto") if algo.dp_row_sharded else "replicated"
    net = ns.BasicRewardNet(venv.observation_space, venv.action_space, hid_sizes=cfg["disc_hid"],
                            normalize_input_layer=ns.RunningNorm)
    demos = ns.Transitions(**make_demos(cfg))
    kw = {} if dp is None else dict(data_parallel=dp)
    return ns.GAIL(demonstrations=demos, demo_batch_size=cfg["demo_batch"], venv=venv, gen_algo=algo,
                   reward_net=net, n_disc_updates_per_round=cfg["n_disc"],
                   gen_replay_buffer_capacity=cfg["n_envs"] * cfg["n_steps"],
                   custom_logger=ns.configure_logger(tempfile.mkdtemp(prefix="bench-")), **kw)


def hip_namespace():
    import types
    import imitation_amd as p
    return types.SimpleNamespace(GAIL=p.GAIL, AIRL=p.AIRL, PPO=p.PPO, FeedForward32Policy=p.FeedForward32Policy,
                                 ActorCriticPolicy=p.ActorCriticPolicy,
                                 NormalizeFeaturesExtractor=p.NormalizeFeaturesExtractor, RunningNorm=p.RunningNorm,
                                 BasicRewardNet=p.BasicRewardNet, BasicShapedRewardNet=p.BasicShapedRewardNet,
                                 NormalizedRewardNet=p.NormalizedRewardNet, Transitions=lambda **kw: p.Transitions(**kw),
                                 configure_logger=lambda d: p.configure_logger(d, []), device="cuda",
                                 CnnPolicy="CnnPolicy", CnnRewardNet=p.modules.CnnRewardNet)


def oracle_namespace():
    import types
    from oracle import imitation_restated as o
    from oracle import sb3_restated as sb
    return types.SimpleNamespace(GAIL=o.GAIL, AIRL=o.AIRL, PPO=sb.PPO, FeedForward32Policy=o.FeedForward32Policy,
                                 ActorCriticPolicy=sb.ActorCriticPolicy,
                                 NormalizeFeaturesExtractor=o.NormalizeFeaturesExtractor, RunningNorm=o.RunningNorm,
                                 BasicRewardNet=o.BasicRewardNet, BasicShapedRewardNet=o.BasicShapedRewardNet,
                                 NormalizedRewardNet=o.NormalizedRewardNet, Transitions=lambda **kw: o.Transitions(**kw),
                                 configure_logger=lambda d: o.configure_logger(d, []), device="cpu",
                                 CnnPolicy=sb.ActorCriticCnnPolicy, CnnRewardNet=o.CnnRewardNet)


def cpu_baseline(cfg, host_threads):
    """Oracle (kind="port") on the host cores: full rounds of the same workload, timed after
    construction (the first rollout pays one env reset). The oracle's torch-CPU ops scale badly past a few
    threads at these sizes (profiles/r01_phase_compare.md: 1 thread 5.9 k, 8 threads 7.7 k, 32 threads
    7.5 k, 128 threads 1.8 k env-steps/s on the GPU box), so the reported value is the 8-thread run
    -- the fastest setting found -- over 4 rounds; the reference's own test default, 1 thread
    (`tests/conftest.py:37-38`), is timed for one round beside it."""
    per_round = cfg["n_envs"] * cfg["n_steps"]
    res = {}
    for threads, rounds in ((min(8, host_threads), 4), (1, 1)):
        th.set_num_threads(threads)
        tr = build_trainer(oracle_namespace(), cfg, "cpu")
        t0 = time.perf_counter()
        tr.train(rounds * per_round)
        res[threads] = (rounds, time.perf_counter() - t0)
    th.set_num_threads(host_threads)
    best = min(8, host_threads)
    rounds, dt = res[best]
    one = res.get(1, res[best])
    return {"value": rounds * per_round / dt, "unit": "env-steps/s", "cores": best, "kind": "port",
            "note": "oracle/ restatement, bit-identical to the reference's own modules run under oracle/ref_shim.py (CPU "
                    "suite); /root/reference does not exist on the GPU box, so the verbatim modules were timed once, in "
                    "the build container: profiles/r01_phase_compare.md (the oracle is not faster than they are)",
            "sample": f"{rounds} rounds = {rounds * per_round} env-steps incl. {rounds * cfg['n_disc']} disc updates + "
                      f"{rounds * 160} PPO minibatch steps, {dt:.1f} s, torch threads={best} (fastest setting), "
                      f"os.cpu_count()={os.cpu_count()}",
            "one_thread": {"value": one[0] * per_round / one[1], "unit": "env-steps/s",
                           "sample": f"{one[0]} round, {one[1]:.1f} s, torch threads=1"}}


def _pmc_traffic():
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        return json.load(open(pmc)) if os.path.exists(pmc) else {}
    except Exception:
        return {}


def disc_update_timing(trainer, cfg):
    """One round's worth of discriminator updates (n_disc x `ia_disc_step_basic`) ALONE on the stream, bracketed
    by HIP events on that stream: the whole-update number SURVEY 8d's roofline target is about
    (6.85 GFLOP per 16 384-row update against the fp32-MFMA peak; 3.6 MB algorithmic bytes against HBM)."""
    from imitation_amd import networks
    n = trainer.n_disc_updates_per_round
    th.cuda.synchronize()
    best = None
    for _ in range(3):
        # all host index draws (and their uploads) first, as the pipelined trainer does, so that the timed region
        # holds nothing but the updates' own launches
        trainer._use_ring, trainer._overlap_k = True, 0
        try:
            drawn = [trainer._batch_sources(None, None) for _ in range(n)]
            th.cuda.synchronize()
            e0, e1 = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
            e0.record()
            t_host = time.perf_counter()
            round_ws = trainer._assemble_round(drawn)   # one launch for all n batches + the ordered norm merges
            fused = round_ws is not None
            for k in range(n):
                with networks.training(trainer.reward_train):
                    trainer._disc_update(None, None, trainer._stats_ring[k], drawn=drawn[k], quirk_done=True,
                                         pre=None if round_ws is None else (round_ws, k))
            e1.record()
            host_us = 1e6 * (time.perf_counter() - t_host) / n
        finally:
            trainer._use_ring = False
        th.cuda.synchronize()
        us = 1e3 * e0.elapsed_time(e1) / n
        if best is None or us < best:
            best, best_host = us, host_us
    R = 2 * cfg["demo_batch"]
    D, (H1, H2) = cfg["obs_dim"] + cfg["act_dim"], cfg["disc_hid"]
    flop = R * (2.0 * (D * H1 + H1 * H2 + H2) * 2 + 2.0 * (H1 * H2 + H2))   # fwd + wgrad + dgrad
    n_par = D * H1 + H1 + H1 * H2 + H2 + H2 + 1
    alg_bytes = R * (D * 4 + 4) + 7 * 4 * n_par                            # rows in, logit out, parameter/Adam traffic
    tf = flop / (best * 1e-6) / 1e12
    return {"kernel": "discriminator update (ia_disc_step_basic: assemble+moments+merge | tile pass: forward+BCE+head "
                      "gradient+dgrad+first-layer wgrad in one workgroup | split-K wgrad | slab reduce+Adam+statistics)",
            "bound": "mfma", "us": best, "host_enqueue_us": best_host,
            "path": "fused (round assembly + 3 launches per update)" if fused else "general (16 launches per update)",
            "launches_per_update": 3 if fused else 16, "launches_per_round_shared": 4 if fused else 0, "rows": R, "flop": flop,
            "achieved": tf, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": tf / PEAK_F32_MFMA_TFLOPS,
            "algorithmic_bytes": alg_bytes, "achieved_hbm_gbs": alg_bytes / (best * 1e-6) / 1e9,
            "frac_hbm": alg_bytes / (best * 1e-6) / 8e12,
            "traffic": _pmc_traffic().get("disc_update_fused"),
            "note": "compute-bound (arithmetic intensity ~1900 flop/B fused): the fp32-MFMA fraction binds, the HBM "
                    "fraction is reported for completeness; measured alone, best of 3 rounds of n_disc updates"}


def gemm_roofline(trainer, cfg, rounds):
    """Runs `rounds` more rounds with every GEMM launch and the persistent PPO launch bracketed by HIP events on
    their streams. The top-level record names the kernel with the LARGEST share of GPU time (the persistent PPO
    update: a latency chain); the discriminator update as a whole and the GEMM family are siblings."""
    from imitation_amd import _lib as L
    lib = L.load()
    lib.ia_prof_enable(1)
    algo = trainer.gen_algo
    ppo_ms, disc_in_rounds = [], []
    for _ in range(rounds):  # the persistent PPO update launch bracketed by events on its stream
        algo.update_events = (th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True))
        trainer.train(cfg["n_envs"] * cfg["n_steps"])
        th.cuda.synchronize()
        if getattr(algo, "_upd_ws", None) is not None or getattr(algo, "_dpg", None):
            ppo_ms.append(algo.update_events[0].elapsed_time(algo.update_events[1]))
        t = getattr(trainer, "_disc_timing", None)   # (start, end, ran behind the PPO update?) of the round's updates, in situ
        if t is not None and t[0] is not None:
            disc_in_rounds.append((1e3 * t[0].elapsed_time(t[1]) / trainer.n_disc_updates_per_round, bool(t[2])))
    algo.update_events = None
    ms, fl = (C.c_double * 12)(), (C.c_double * 12)()
    cnt = (C.c_longlong * 12)()
    lib.ia_prof_collect(ms, fl, cnt)
    lib.ia_prof_enable(0)
    names = {0: "NT(fwd)", 1: "NN(dgrad)", 2: "TN(wgrad)"}
    tiles = {0: "128x128", 1: "64x64", 2: "128x32", 3: "32x128"}
    per = []
    for k in range(12):
        if cnt[k]:
            per.append(dict(kernel=f"ia_gemm_kernel {names[k // 4]} tile {tiles[k % 4]}", launches=int(cnt[k]),
                            avg_us=1e3 * ms[k] / cnt[k], tflops=fl[k] / (ms[k] * 1e-3) / 1e12, total_ms=ms[k]))
    per.sort(key=lambda d: -d["total_ms"])
    gemm = None
    if per:
        top = per[0]
        all_ms, all_fl = sum(p["total_ms"] for p in per), sum(p["tflops"] * p["total_ms"] for p in per)
        gemm = {"bound": "mfma", "kernel": top["kernel"], "achieved": top["tflops"], "peak": PEAK_F32_MFMA_TFLOPS,
                "unit": "TFLOP/s", "frac": top["tflops"] / PEAK_F32_MFMA_TFLOPS, "avg_launch_us": top["avg_us"],
                "launches": top["launches"], "traffic": _pmc_traffic().get(top["kernel"]),
                "all_gemm_tflops": all_fl / all_ms if all_ms else None, "kernels": per[:6],
                "note": "measured inside training rounds, i.e. BESIDE the persistent PPO kernel and the act kernels"}
    disc = disc_update_timing(trainer, cfg)
    if disc_in_rounds:
        # the same updates INSIDE training rounds: beside the persistent PPO kernel and the act kernels (or behind the PPO
        # update, beside the next rollout's act kernels), events on the discriminator stream around the round's updates
        us = sorted(u for u, _ in disc_in_rounds)[len(disc_in_rounds) // 2]
        disc["us_in_rounds"] = us
        disc["in_rounds_schedule"] = "behind the PPO update" if disc_in_rounds[-1][1] else "beside the PPO update"
        disc["achieved_in_rounds"] = disc["flop"] / (us * 1e-6) / 1e12
        disc["frac_in_rounds"] = disc["achieved_in_rounds"] / PEAK_F32_MFMA_TFLOPS
    ppo = None
    if ppo_ms:
        # forward + backward of both 32x32 towers ~ 3 x 2 x (weights touched) flops per row and step
        D, A, H = cfg["obs_dim"], cfg["act_dim"], 32
        per_row = 6.0 * ((D * H + H * H + H * A) + (D * H + H * H + H))
        steps = algo.n_epochs * algo._n_mb
        world = getattr(getattr(algo, "dp", None), "world", 1)   # the data-parallel update runs on the gathered tile
        rows = world * min(algo.batch_size, cfg["n_envs"] * cfg["n_steps"])
        fl_ppo = per_row * rows * steps
        avg = sum(ppo_ms) / len(ppo_ms)
        disc_ms_round = disc.get("us_in_rounds", disc["us"]) * 1e-3 * trainer.n_disc_updates_per_round
        ppo = {"kernel": "ppo_update_persistent_kernel (whole PPO.train, one launch)", "bound": "latency",
               "avg_launch_us": 1e3 * avg, "optimizer_steps_per_launch": steps, "us_per_step": 1e3 * avg / steps,
               "achieved": fl_ppo / (avg * 1e-3) / 1e12, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
               "frac": fl_ppo / (avg * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS,
               "traffic": _pmc_traffic().get("ppo_update_persistent_kernel"),
               # of the two kernel families that own the round's GPU time (in-situ discriminator time); the share of ALL
               # GPU time, act / relabel / GAE / copies included, is in the rocprofv3 summary under profiles/
               "share_of_ppo_plus_disc_gpu_time": avg / (avg + disc_ms_round),
               "algorithmic_bytes_per_launch": steps * rows * (D + A + 4) * 4.0,
               **_ppo_floor(D, A, rows, 1e3 * avg / steps),
               "note": "largest kernel by GPU time: a chain of dependent 1024-row optimiser steps on nblk+3 workgroups "
                       "(the reference's minibatch semantics), bound by per-step latency (grid barrier, slab reduce, "
                       "Adam), not by MFMA or HBM throughput: read us_per_step, not frac. The throughput kernels are "
                       "under `disc_update` (whole discriminator update) and `gemm`"}
    top = dict(ppo) if ppo is not None else dict(disc)
    top["disc_update"] = disc
    top["gemm"] = gemm
    top["ppo_update"] = ppo
    return top


def _ppo_floor(D, A, rows, us_per_step):
    """The computed floor of one optimiser step (`tools/ppo_step_floor.py`: longest dependent path priced with the guide's
    issue / MFMA / LDS / fabric constants, nothing measured) beside the measured step, so the latency kernel's number has a
    denominator: `us_per_step_over_floor` is the figure to read, not `frac`."""
    try:
        from tools.ppo_step_floor import model
        m = model(D, A, rows)
        return {"floor_us": m["floor_us"], "floor_chain_us": m["chain_us"], "us_per_step_over_floor": us_per_step / m["floor_us"]}
    except Exception as e:   # the figure is a derived annotation: never take the line down
        return {"floor_us": None, "floor_error": f"{type(e).__name__}: {e}"}


# ---- the other BASELINE.json / SURVEY 8d configurations, driver-timed under `variants` -----------------
# Every entry is the exact configuration it runs (echoed into the JSON line under `config`), n_envs = 1024 per GPU like
# the headline. "verbatim" = the reference's tuned hyper-parameter files, value by value.
def _v(**kw):
    return kw


_PPO_P = dict(batch_size=1024, n_epochs=10, ent_coef=0.1, learning_rate=3e-4)   # config P (scripts/config/train_adversarial.py:105-130)
VARIANTS = {
    # SURVEY 8d variant H: rollouts of a whole horizon (1 024 000 transitions per round), config P's real update counts
    "H_horizon_1024x1000": _v(algo="gail", n_envs=1024, n_steps=1000, obs=17, act=6, ppo=_PPO_P, demo_batch=8192, n_disc=16,
                              capacity=16384, net=dict(hid_sizes=(256, 256)), rounds=2, warm=1),
    # scripts/config/tuned_hps/gail_seals_half_cheetah_best_hp_eval.json:2-44 verbatim (rl.batch_size 4096 -> n_steps 4)
    "T_gail_half_cheetah_tuned_verbatim": _v(
        algo="gail", n_envs=1024, n_steps=4, obs=17, act=6,
        ppo=dict(batch_size=64, clip_range=0.1, ent_coef=3.992371122209408e-6, gae_lambda=0.95, gamma=0.95,
                 learning_rate=0.00026250519057717037, max_grad_norm=0.8, n_epochs=5, vf_coef=0.11483689492120866),
        demo_batch=8192, n_disc=8, capacity=512, net=dict(), normalize_output=True, rounds=30, warm=5),
    # scripts/config/tuned_hps/airl_seals_ant_best_hp_eval.json:2-44 verbatim (rl.batch_size 8192 -> n_steps 8; PPO
    # minibatch 16: 5 120 optimiser steps per round)
    "3_airl_ant_tuned_verbatim": _v(
        algo="airl", n_envs=1024, n_steps=8, obs=27, act=8,
        ppo=dict(batch_size=16, clip_range=0.3, ent_coef=3.27750078482474e-6, gae_lambda=0.8, gamma=0.995,
                 learning_rate=3.249429831179079e-5, max_grad_norm=0.9, n_epochs=10, vf_coef=0.4351450387648799),
        demo_batch=8192, n_disc=16, capacity=8192, net=dict(), normalize_output=True, rounds=3, warm=1),
    # BASELINE config 3's shape with config P's generator schedule (PPO minibatch 1024): the fused AIRL update
    "3_airl_ant_1024x16_mb1024": _v(algo="airl", n_envs=1024, n_steps=16, obs=27, act=8, ppo=dict(_PPO_P, ent_coef=0.01),
                                    demo_batch=8192, n_disc=16, capacity=16384, net=dict(), normalize_output=True,
                                    rounds=30, warm=5),
    # ... "+ grad-penalty" as BASELINE words config 3 (opt-in extension, no reference counterpart: coefficient 10)
    "3_airl_ant_1024x16_mb1024_gp10": _v(algo="airl", n_envs=1024, n_steps=16, obs=27, act=8, ppo=dict(_PPO_P, ent_coef=0.01),
                                         demo_batch=8192, n_disc=16, capacity=16384, net=dict(), normalize_output=True,
                                         grad_penalty=10.0, rounds=30, warm=5),
    # config P with the opt-in gradient penalty on the 256 x 256 discriminator
    "P_gp10": _v(algo="gail", n_envs=1024, n_steps=16, obs=17, act=6, ppo=_PPO_P, demo_batch=8192, n_disc=16, capacity=16384,
                 net=dict(hid_sizes=(256, 256)), grad_penalty=10.0, rounds=30, warm=5),
    # config P with the reference's DEFAULT discriminator (BasicRewardNet 32 x 32)
    "P_disc32": _v(algo="gail", n_envs=1024, n_steps=16, obs=17, act=6, ppo=_PPO_P, demo_batch=8192, n_disc=16, capacity=16384,
                   net=dict(), rounds=30, warm=5),
    # GAIL at Ant width (35 inputs) with use_next_state + use_done on top (63 inputs), default 32 x 32 net
    "P_ant_gail_d35": _v(algo="gail", n_envs=1024, n_steps=16, obs=27, act=8, ppo=_PPO_P, demo_batch=8192, n_disc=16,
                         capacity=16384, net=dict(), rounds=30, warm=5),
    # ... the 256 x 256 discriminator at Ant width (rows of 36 floats: the wide tile kernels), without and with the
    # opt-in gradient penalty (`disc_gp_kernel<256, 64>` inside the update)
    "P_ant_gail_d35_h256": _v(algo="gail", n_envs=1024, n_steps=16, obs=27, act=8, ppo=_PPO_P, demo_batch=8192, n_disc=16,
                              capacity=16384, net=dict(hid_sizes=(256, 256)), rounds=30, warm=5),
    "P_ant_gail_d35_gp10": _v(algo="gail", n_envs=1024, n_steps=16, obs=27, act=8, ppo=_PPO_P, demo_batch=8192, n_disc=16,
                              capacity=16384, net=dict(hid_sizes=(256, 256)), grad_penalty=10.0, rounds=30, warm=5),
    "P_ant_gail_d63_next_done": _v(algo="gail", n_envs=1024, n_steps=16, obs=27, act=8, ppo=_PPO_P, demo_batch=8192, n_disc=16,
                                   capacity=16384, net=dict(use_next_state=True, use_done=True), rounds=30, warm=5),
    # BASELINE config 1 in its canonical library form (docs/algorithms/gail.rst:36-91): 8 envs, SB3 MlpPolicy 64 x 64,
    # Discrete head on the reference's sampling stream
    "1_cartpole_8x256_mlp64": _v(algo="gail", n_envs=8, n_steps=256, obs=4, act=2, discrete=True, policy="mlp64",
                                 ppo=dict(batch_size=64, n_epochs=5, ent_coef=0.0, learning_rate=4e-4, gamma=0.95),
                                 demo_batch=1024, n_disc=4, capacity=2048, net=dict(), rounds=4, warm=3),
    # the same environment stepped through a generic gym-style VecEnv (dict infos, terminal_observation, Monitor's
    # `episode` entries): the per-env Python branch of the wrappers (rewards/reward_wrapper.py:98-109)
    "1_cartpole_8x256_mlp64_generic_vecenv": _v(algo="gail", n_envs=8, n_steps=256, obs=4, act=2, discrete=True,
                                                policy="mlp64", generic_vecenv=True,
                                                ppo=dict(batch_size=64, n_epochs=5, ent_coef=0.0, learning_rate=4e-4, gamma=0.95),
                                                demo_batch=1024, n_disc=4, capacity=2048, net=dict(), rounds=4, warm=3),
    # config P stepped through a generic SB3-protocol VecEnv at WIDTH (what a 1 024-env DummyVecEnv / SubprocVecEnv of
    # Monitor-wrapped environments hands the trainer, `util/util.py:158-166`): one info dict per env and step,
    # `terminal_observation` / `TimeLimit.truncated` / `episode` on the (staggered) episode ends -- the wrappers' per-env
    # branch (`rewards/reward_wrapper.py:92-133`, `data/wrappers.py:69-91`) instead of `ArrayVecEnv.step_wait_arrays`
    # ... and the SAME environment (staggered episode ends, variable-horizon flag) through the array protocol: the pair
    # isolates what the dict protocol costs -- the env's own 1 024 dicts per step and the trainer's per-env branch
    "P_stagger_arrays_1024": _v(algo="gail", n_envs=1024, n_steps=16, obs=17, act=6, ppo=_PPO_P, demo_batch=8192, n_disc=16,
                                capacity=16384, net=dict(hid_sizes=(256, 256)), stagger=True, rounds=30, warm=5),
    "P_generic_vecenv_1024": _v(algo="gail", n_envs=1024, n_steps=16, obs=17, act=6, ppo=_PPO_P, demo_batch=8192, n_disc=16,
                                capacity=16384, net=dict(hid_sizes=(256, 256)), generic_vecenv=True, stagger=True,
                                rounds=30, warm=5),
    # config P with SB3's default `MlpPolicy` (64 x 64 tanh towers) as the generator instead of FeedForward32Policy
    "P_mlp64_1024x16": _v(algo="gail", n_envs=1024, n_steps=16, obs=17, act=6, policy="mlp64", ppo=dict(_PPO_P, ent_coef=0.01),
                          demo_batch=8192, n_disc=16, capacity=16384, net=dict(hid_sizes=(256, 256)), rounds=30, warm=5),
    # policy towers outside the fused kernels' shapes (any SB3 `net_arch`): the general minibatch loop
    "towers_1024x16_pi128x64_vf256": _v(algo="gail", n_envs=1024, n_steps=16, obs=17, act=6, policy="mlp64",
                                        net_arch=dict(pi=[128, 64], vf=[256]),
                                        ppo=dict(batch_size=2048, n_epochs=4, ent_coef=0.01), demo_batch=8192, n_disc=4,
                                        capacity=16384, net=dict(hid_sizes=(256, 256)), rounds=30, warm=5),
    # GAIL on uint8 image observations: CnnPolicy generator + CnnRewardNet discriminator (run_image_variant)
    "image_gail_64x16_cnn": None,
    # BASELINE config 5: BC supervised step, NatureCNN policy on 84 x 84 x 4 uint8 frames, batch 4096 (run_bc_variant)
    "5_bc_cnn_4096": None,
}


def build_image_variant(p=None):
    """GAIL on image observations end to end: uint8 [4, 84, 84] frames, `PPO("CnnPolicy")` generator
    (NatureCNN, Categorical head), `CnnRewardNet` discriminator (HIP: `modules.CnnRewardNet`, trained through the operator
    boundary; `p`: namespace of another implementation, e.g. the oracle for the CPU figure)."""
    from imitation_amd.vec_env import SyntheticImageVecEnv
    p = p or hip_namespace()
    n_envs, n_steps, shape, n_act = 64, 16, (4, 84, 84), 6
    th.manual_seed(0)
    np.random.seed(0)
    venv = SyntheticImageVecEnv(num_envs=n_envs, shape=shape, act_dim=3, horizon=500, seed=0, n_discrete=n_act)
    algo = p.PPO(p.CnnPolicy, venv, n_steps=n_steps, batch_size=256, n_epochs=4, ent_coef=0.01, learning_rate=1e-4, seed=0,
                 device=p.device)
    net = p.CnnRewardNet(venv.observation_space, venv.action_space, hwc_format=False)
    rng = np.random.default_rng(1)
    n = 2048
    frames = rng.integers(0, 256, (n + 1, *shape)).astype(np.uint8)
    demos = p.Transitions(obs=frames[:-1], acts=rng.integers(0, n_act, n).astype(np.int64), next_obs=frames[1:],
                          dones=np.zeros(n, bool))
    tr = p.GAIL(demonstrations=demos, demo_batch_size=512, venv=venv, gen_algo=algo, reward_net=net,
                n_disc_updates_per_round=2, custom_logger=p.configure_logger(tempfile.mkdtemp(prefix="bench-var-")))
    return tr, n_envs * n_steps


def run_image_variant(rounds=10, warm=3):   # (45-50 ms rounds on the host's clock: three-round samples spread by 8 %)
    tr, per = build_image_variant()
    tr.train(warm * per)
    th.cuda.synchronize()
    t0 = time.perf_counter()
    tr.train(rounds * per)
    th.cuda.synchronize()
    dt = time.perf_counter() - t0
    finite = all(bool(th.isfinite(v.float()).all()) for v in tr.gen_algo.policy.state_dict().values())
    return {"env_steps_per_s": rounds * per / dt, "ms_per_round": 1e3 * dt / rounds, "rounds": rounds,
            "env_steps_per_round": per, "finite": finite,
            "config": "GAIL, 64 envs x 16 steps of uint8 4x84x84 frames, CnnPolicy (NatureCNN) + CnnRewardNet, PPO "
                      "minibatch 256 x 4 epochs, demo batch 512 x 2 updates"}


def run_bc_variant(batch=4096, steps=10):
    """BASELINE config 5: `bc.BC` supervised steps (`algorithms/bc.py:94-156,464-510`) with the NatureCNN policy on
    synthetic uint8 4 x 84 x 84 frames, Discrete(6), batch 4096: samples/s and the GEMM work rate of a step."""
    import imitation_amd as p
    from imitation_amd import spaces
    shape, A = (4, 84, 84), 6
    osp, asp = spaces.Box(0, 255, shape, np.uint8), spaces.Discrete(A)
    rng = np.random.default_rng(0)
    n = 2 * batch
    obs = rng.integers(0, 256, (n, *shape), dtype=np.uint8)
    acts = rng.integers(0, A, n).astype(np.int64)
    demos = p.Transitions(obs=obs, acts=acts, next_obs=obs, dones=np.zeros(n, bool))
    th.manual_seed(0)
    pol = p.cnn_policy.ActorCriticCnnPolicy(osp, asp, lambda _: 1.0)
    tr = p.bc.BC(observation_space=osp, action_space=asp, rng=rng, policy=pol, demonstrations=demos, batch_size=batch,
                 device="cuda", custom_logger=p.configure_logger(tempfile.mkdtemp(prefix="bench-bc-"), []))
    tr.train(n_batches=2, log_interval=10 ** 9)
    th.cuda.synchronize()
    dt = None
    for _ in range(2):   # best of two timed passes (one pass in twenty ran at half speed on a busy host: 115 MB of frames
        t0 = time.perf_counter()   # per batch are gathered on the host)
        tr.train(n_batches=steps, log_interval=10 ** 9)
        th.cuda.synchronize()
        d = (time.perf_counter() - t0) / steps
        dt = d if dt is None else min(dt, d)
    g = pol.geom
    fwd = sum(2.0 * batch * oh * ow * (cin * k * k) * cout for cin, _, _, cout, k, _, oh, ow in g) \
        + 2.0 * batch * pol.n_flatten * 512 + 2.0 * batch * 512 * (A + 1)
    dgrad = sum(2.0 * batch * oh * ow * (cin * k * k) * cout for cin, _, _, cout, k, _, oh, ow in g[1:]) \
        + 2.0 * batch * pol.n_flatten * 512 + 2.0 * batch * 512 * A
    flops = 2 * fwd + dgrad
    finite = all(bool(th.isfinite(v.float()).all()) for v in pol.state_dict().values())
    return {"samples_per_s": batch / dt, "ms_per_step": 1e3 * dt, "steps": steps, "batch": batch, "finite": finite,
            "gemm_tflops": flops / dt / 1e12, "frac_of_fp32_mfma_peak": flops / dt / 1e12 / PEAK_F32_MFMA_TFLOPS,
            "config": "BC, NatureCNN ActorCriticCnnPolicy, uint8 4x84x84 frames, Discrete(6), batch 4096, Adam"}


def build_variant(name, p=None):
    """The trainer of one non-image variant, untrained (`p`: namespace of an implementation, default the HIP product)."""
    from imitation_amd.vec_env import SyntheticVecEnv
    if name == "image_gail_64x16_cnn":
        return build_image_variant(p)
    p = p or hip_namespace()
    v = VARIANTS[name]
    n_envs, n_steps, od, ad = v["n_envs"], v["n_steps"], v["obs"], v["act"]
    discrete = v.get("discrete", False)
    th.manual_seed(0)
    np.random.seed(0)
    venv = SyntheticVecEnv(num_envs=n_envs, obs_dim=od, act_dim=ad, horizon=1000 if n_envs > 8 else 500, seed=0,
                           n_discrete=ad if discrete else None, stagger=bool(v.get("stagger")))
    if v.get("generic_vecenv"):
        from imitation_amd.vec_env import GymStyleVecEnv
        venv = GymStyleVecEnv(venv)
    pk = dict(features_extractor_class=p.NormalizeFeaturesExtractor,
              features_extractor_kwargs=dict(normalize_class=p.RunningNorm))
    mlp64 = v.get("policy") == "mlp64"
    policy = p.ActorCriticPolicy if mlp64 else p.FeedForward32Policy   # SB3 MlpPolicy default = 64 x 64, no feature norm
    if v.get("net_arch"):
        pk = dict(pk, net_arch=v["net_arch"])
    elif mlp64:
        pk = {}
    algo = p.PPO(policy, venv, n_steps=n_steps, seed=0, policy_kwargs=pk, device=p.device, **v["ppo"])
    if v["algo"] == "gail":
        net = p.BasicRewardNet(venv.observation_space, venv.action_space, normalize_input_layer=p.RunningNorm, **v["net"])
        cls = p.GAIL
    else:
        net = p.BasicShapedRewardNet(venv.observation_space, venv.action_space, normalize_input_layer=p.RunningNorm, **v["net"])
        cls = p.AIRL
    if v.get("normalize_output"):
        net = p.NormalizedRewardNet(net, p.RunningNorm)
    rng = np.random.default_rng(1)
    n = max(4 * v["demo_batch"], 20000)
    obs = rng.standard_normal((n, od)).astype(np.float32)
    acts = rng.integers(0, ad, n).astype(np.int64) if discrete else rng.uniform(-1, 1, (n, ad)).astype(np.float32)
    demos = p.Transitions(obs=obs, acts=acts, next_obs=(0.9 * obs).astype(np.float32), dones=np.zeros(n, bool))
    extra = dict(disc_grad_penalty_coef=v["grad_penalty"]) if v.get("grad_penalty") else {}   # (opt-in extension, HIP only)
    tr = cls(demonstrations=demos, demo_batch_size=v["demo_batch"], venv=venv, gen_algo=algo, reward_net=net,
             n_disc_updates_per_round=v["n_disc"], gen_replay_buffer_capacity=v["capacity"],
             custom_logger=p.configure_logger(tempfile.mkdtemp(prefix="bench-var-")),
             allow_variable_horizon=bool(v.get("stagger")), **extra)   # (staggered first episodes are shorter)
    return tr, n_envs * n_steps


# variants that get the CPU oracle timed beside them (one round each, after one warm-up round: the reference's shipped
# configurations, for which the headline's cpu_baseline says nothing)
CPU_VARIANTS = ("T_gail_half_cheetah_tuned_verbatim", "3_airl_ant_tuned_verbatim", "1_cartpole_8x256_mlp64",
                "P_generic_vecenv_1024", "image_gail_64x16_cnn")


def variant_cpu_baseline(name, host_threads):
    th.set_num_threads(min(8, host_threads))
    try:
        tr, per = build_variant(name, oracle_namespace())
        warm = 0 if name == "image_gail_64x16_cnn" else 1   # (a CPU round of 1 024 frames through two NatureCNNs: tens of seconds)
        if warm:
            tr.train(per)
        t0 = time.perf_counter()
        tr.train(per)
        dt = time.perf_counter() - t0
    finally:
        th.set_num_threads(1)
    return {"value": per / dt, "unit": "env-steps/s", "cores": min(8, host_threads), "kind": "port",
            "sample": f"1 round = {per} env-steps after {warm} warm-up round(s), {dt:.1f} s, torch threads={min(8, host_threads)}",
            "note": "oracle/ restatement (bit-identical to the reference's modules under oracle/ref_shim.py, which cannot "
                    "travel to the GPU box); the reference's own modules were timed once, in the build container: "
                    "profiles/r01_phase_compare.md"}


def run_variant(name, rounds=None, warm=None):
    if name == "image_gail_64x16_cnn":
        return run_image_variant(rounds or 10, warm or 3)
    if name == "5_bc_cnn_4096":
        return run_bc_variant()
    v = VARIANTS[name]
    rounds, warm = rounds or v["rounds"], warm or v["warm"]
    tr, per = build_variant(name)
    tr.train(warm * per)
    th.cuda.synchronize()
    t0 = time.perf_counter()
    tr.train(rounds * per)
    th.cuda.synchronize()
    dt = time.perf_counter() - t0
    finite = all(bool(th.isfinite(v_.float()).all()) for v_ in tr.gen_algo.policy.state_dict().values())
    cfg = {k: (list(x) if isinstance(x, tuple) else x) for k, x in v.items() if k not in ("rounds", "warm")}
    cfg["net"] = {k: (list(x) if isinstance(x, tuple) else x) for k, x in v["net"].items()}
    steps = tr.gen_algo.n_epochs * tr.gen_algo._n_mb
    out = {"env_steps_per_s": rounds * per / dt, "ms_per_round": 1e3 * dt / rounds, "rounds": rounds,
           "env_steps_per_round": per, "ppo_optimizer_steps_per_round": steps, "finite": finite, "config": cfg}
    out.update(_variant_disc_record(tr, v))
    return out


def _variant_disc_record(tr, v):
    """The LAST timed round's discriminator updates by the events on their stream (`AdversarialTrainer._disc_timing`): us per
    update inside the round and, for the two-hidden-layer stacks, the fp32-MFMA fraction of the update's flops -- BCE update
    2 R (3 (D H1 + H1 H2 + H2) - D H1) over R = 2 x demo_batch rows (forward, weight gradient, input gradient but the first
    layer's) plus, with the opt-in penalty, 2 B (4 D H1 + 4 H1 H2 + H2) over B = demo_batch interpolates (forward, input
    gradient, the second pass through both layers and the two weight-gradient products); AIRL: the base stack on R rows and
    the potential stack on 2 R (next state, state), the penalty likewise on B and 2 B."""
    t = getattr(tr, "_disc_timing", None)
    if t is None or t[0] is None:
        return {}
    try:
        th.cuda.synchronize()
        us = 1e3 * t[0].elapsed_time(t[1]) / tr.n_disc_updates_per_round
    except Exception:
        return {}
    rec = {"disc_update_us_in_rounds": us, "disc_updates_schedule": "behind the PPO update" if t[2] else "beside the PPO update"}
    hid = tuple(v["net"].get("hid_sizes", (32, 32)))
    if len(hid) != 2:
        return rec
    H1, H2 = hid
    B, gp = v["demo_batch"], v.get("grad_penalty", 0.0) > 0.0
    R = 2 * B
    upd = lambda rows, D: 2.0 * rows * (3 * (D * H1 + H1 * H2 + H2) - D * H1)
    pen = lambda rows, D: 2.0 * rows * (4 * D * H1 + 4 * H1 * H2 + H2)
    if v["algo"] == "gail":
        D = v["obs"] + v["act"] + (v["obs"] if v["net"].get("use_next_state") else 0) + (1 if v["net"].get("use_done") else 0)
        flop = upd(R, D) + (pen(B, D) if gp else 0.0)
    else:   # AIRL's shaped net (defaults: base on [s | a | s'] ... as flagged; potential 32 x 32 on s', s)
        Db = v["obs"] + v["act"]
        flop = upd(R, Db) + upd(2 * R, v["obs"]) + ((pen(B, Db) + pen(2 * B, v["obs"])) if gp else 0.0)
    rec["disc_update_flop"] = flop
    rec["disc_update_frac_in_rounds"] = flop / (us * 1e-6) / 1e12 / PEAK_F32_MFMA_TFLOPS
    return rec


def _dp_form_text(algo):
    """Which data-parallel form of the persistent PPO update the run used, and on what grounds (`PPO.dp_update_form`)."""
    names = {"sharded": "row-sharded, in-kernel exchange", "replicated": "replicated on the gathered tile"}
    ch = getattr(algo, "dp_choice", None)
    if ch is not None:
        return (f"{names[ch['chosen']]} -- measured on this node during the warm-up rounds: row-sharded "
                f"{ch['sharded_ms']:.3f} ms, replicated {ch['replicated_ms']:.3f} ms per update (slowest rank)")
    g = algo._dpg if isinstance(getattr(algo, "_dpg", None), dict) else None
    if g is None:
        return "per-minibatch gradient all-reduce"
    if algo.dp_update_form in g["forms"] or len(g["forms"]) == 1:
        form = algo.dp_update_form if algo.dp_update_form in g["forms"] else g["forms"][0]
        why = "asked for" if algo.dp_update_form == form else "the only form available"
        if getattr(algo, "dp_handshake_failed", False):
            why = "the peer-memory handshake failed on at least one rank: every rank fell back to this form"
        return f"{names[form]} ({why})"
    return "row-sharded / replicated alternating (the timed trials were not over)"


def _self_launch(n_gpus: int) -> int:
    """`python bench.py --gpus N` without a launcher's environment: re-runs this file under
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1` with the same arguments
    (the driver's own N > 1 command shape) and returns its exit status. Rank 0 prints the one JSON line."""
    import socket
    import subprocess
    if os.environ.get("IA_BENCH_SHARE_GPU") != "1" and th.cuda.device_count() < n_gpus:
        raise SystemExit(f"bench.py: --gpus {n_gpus} but this node shows {th.cuda.device_count()} GPU(s)")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n_gpus) // n_gpus)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)   # ~1 s of timed rounds at config P
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--prof-rounds", type=int, default=2)
    ap.add_argument("--no-variants", action="store_true")
    ap.add_argument("--long-rounds", type=int, default=200)   # total rounds of the longer sample behind the timed region
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # bare `python bench.py --gpus N`: become the launcher -- one rank per GPU under torch.distributed.run on this
        # node, the ranks' stdout (rank 0's ONE JSON line) and exit status passed through
        sys.exit(_self_launch(args.gpus))
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                         f"(`python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus}`), "
                         "or run `python bench.py --gpus N` bare and it launches the ranks itself")
    # Test hook for boxes with fewer GPUs than ranks (not a benchmark mode): IA_BENCH_SHARE_GPU=1 puts
    # every rank on cuda:0 and moves the collectives through gloo (RCCL refuses two ranks on one GPU).
    share = os.environ.get("IA_BENCH_SHARE_GPU") == "1"
    th.cuda.set_device(0 if share else local_rank)
    dp = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=th.device("cuda", local_rank))
        from imitation_amd.distributed import DataParallel
        dp = DataParallel()

    cfg = dict(CFG_P)
    per_round = cfg["n_envs"] * cfg["n_steps"]
    # The HIP path's host side is sequential index bookkeeping; torch's intra-op thread pool only
    # hurts it (torch.randperm(64000) -- the expert permutation, same values for any thread count
    # -- takes 8 ms with 128 threads vs 0.7 ms with 1). The CPU baseline below gets all cores back.
    host_threads = th.get_num_threads()
    th.set_num_threads(1)
    trainer = build_trainer(hip_namespace(), cfg, "cuda", seed=rank, dp=dp)
    if args.warmup > 0:
        trainer.train(args.warmup * per_round)

    def barrier():
        th.cuda.synchronize()
        if world > 1:
            dist.barrier()
            th.cuda.synchronize()

    def over_ranks(x):   # the slowest rank's time
        if world > 1:
            tmax = th.tensor([x], device="cuda", dtype=th.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            x = float(tmax.item())
        return x

    barrier()
    trainer.round_wall_stamps = stamps = []   # (a host time stamp per round: the spread inside the timed call)
    t0 = time.perf_counter()
    trainer.train(args.steps * per_round)
    barrier()
    dt = over_ranks(time.perf_counter() - t0)
    trainer.round_wall_stamps = None
    # spread of the timed rounds (rank 0's host stamps; the first interval starts at t0) ...
    gaps = np.diff(np.array([t0] + stamps)) * 1e3
    spread = None
    if len(gaps) >= 2:
        spread = {"ms_per_step_median": float(np.median(gaps)), "ms_per_step_p10": float(np.percentile(gaps, 10)),
                  "ms_per_step_p90": float(np.percentile(gaps, 90)), "ms_per_step_min": float(gaps.min()),
                  "ms_per_step_max": float(gaps.max()), "rounds": int(len(gaps)),
                  "note": "host time stamps at the end of each round's iteration inside the ONE timed train() call; "
                          "under the pipelined schedule a stamp leads the device by up to one round's updates, so single "
                          "gaps are indicative and their sum is what `ms_per_step` measures"}
    # ... and a longer sample right behind the timed region, so that K x 3.8 ms is never the only number: the same call
    # over `--long-rounds` rounds (default 200 minus K), timed the same way; `value_long` covers both regions together
    long_rounds = max(0, args.long_rounds - args.steps)
    value_long = None
    if long_rounds > 0:
        barrier()
        t1 = time.perf_counter()
        trainer.train(long_rounds * per_round)
        barrier()
        dt_long = over_ranks(time.perf_counter() - t1)
        value_long = {"value": world * (args.steps + long_rounds) * per_round / (dt + dt_long), "unit": "env-steps/s",
                      "rounds": args.steps + long_rounds, "ms_per_step": 1e3 * (dt + dt_long) / (args.steps + long_rounds),
                      "value_extra_rounds_only": world * long_rounds * per_round / dt_long}

    # the profiled rounds are training rounds too: under data parallelism EVERY rank must take part in
    # their collectives (only rank 0's measurement is reported)
    roof = gemm_roofline(trainer, cfg, args.prof_rounds) if args.prof_rounds > 0 else None
    if rank != 0:
        roof = None
    variants = None
    if rank == 0 and world == 1 and not args.no_variants:
        # SURVEY 8d's other configurations through the same trainer (a few rounds each, after the timed region)
        variants = {}
        for name in VARIANTS:
            try:
                variants[name] = run_variant(name)
            except Exception as e:  # a variant must never take the headline down with it
                variants[name] = {"error": f"{type(e).__name__}: {e}"}
    base = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        base = cpu_baseline(cfg, host_threads)
        if variants:
            for name in CPU_VARIANTS:   # the oracle beside the reference's shipped configurations (one round each)
                if name in variants and "error" not in variants[name]:
                    try:
                        cb = variant_cpu_baseline(name, host_threads)
                        variants[name]["cpu_baseline"] = cb
                        variants[name]["speedup_vs_cpu_baseline"] = variants[name]["env_steps_per_s"] / cb["value"]
                    except Exception as e:
                        variants[name]["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"}
    by_rank = None
    if world > 1:
        # what every rank actually ran (the form is the ranks' COMMON verdict by construction; the line shows it per rank)
        g = trainer.gen_algo._dpg if isinstance(getattr(trainer.gen_algo, "_dpg", None), dict) else None
        ch = getattr(trainer.gen_algo, "dp_choice", None)
        mine = {"rank": rank,
                "form": (ch["chosen"] if ch else (trainer.gen_algo.dp_update_form
                                                  if g and trainer.gen_algo.dp_update_form in g["forms"]
                                                  else (g["forms"][0] if g else "per-minibatch all-reduce"))),
                "handshake_failed": bool(getattr(trainer.gen_algo, "dp_handshake_failed", False)),
                "sharded_updates": int(getattr(trainer.gen_algo, "dp_sharded_updates", 0))}
        by_rank = [None] * world
        dist.all_gather_object(by_rank, mine)
        dist.barrier()

    if rank == 0:
        value = world * args.steps * per_round / dt
        disc = (roof or {}).get("disc_update") or {}
        ppo = (roof or {}).get("ppo_update") or {}
        gemm = (roof or {}).get("gemm")
        r3 = lambda x: None if x is None else float(f"{x:.4g}")
        # compact record of everything a reader needs first (the driver keeps scalar fields and the END of the line):
        summary = {
            "headline_env_steps_per_s": r3(value), "ms_per_round": r3(1e3 * dt / args.steps), "n_gpus": world,
            "ms_per_round_p10_median_p90": ([r3(spread["ms_per_step_p10"]), r3(spread["ms_per_step_median"]),
                                             r3(spread["ms_per_step_p90"])] if spread else None),
            "env_steps_per_s_over_200_rounds": r3(value_long["value"]) if value_long else None,
            "ppo": {"us_per_step": r3(ppo.get("us_per_step")), "floor_us": r3(ppo.get("floor_us")),
                    "us_per_step_over_floor": r3(ppo.get("us_per_step_over_floor")), "launch_us": r3(ppo.get("avg_launch_us")),
                    "steps_per_launch": ppo.get("optimizer_steps_per_launch"), "frac_mfma": r3(ppo.get("frac")),
                    "traffic": ppo.get("traffic"), "algorithmic_bytes": ppo.get("algorithmic_bytes_per_launch")},
            "disc_update": {"us": r3(disc.get("us")), "us_in_rounds": r3(disc.get("us_in_rounds")),
                            "frac": r3(disc.get("frac")), "frac_in_rounds": r3(disc.get("frac_in_rounds")),
                            "traffic": disc.get("traffic"), "algorithmic_bytes": disc.get("algorithmic_bytes"),
                            "path": (disc.get("path") or "").split(" ")[0]},
            "cpu_env_steps_per_s": r3(base["value"]) if base else None,
            "speedup_vs_cpu": r3(value / base["value"]) if base else None,
            "variants_env_steps_per_s": ({k: r3(v.get("env_steps_per_s", v.get("samples_per_s"))) for k, v in variants.items()}
                                         if variants else None),
            "variants_disc_update_us_frac_in_rounds": ({k: [r3(v.get("disc_update_us_in_rounds")), r3(v.get("disc_update_frac_in_rounds"))]
                                                        for k, v in variants.items() if v.get("disc_update_us_in_rounds")}
                                                       if variants else None),
            "variants_cpu_env_steps_per_s": ({k: r3(v["cpu_baseline"].get("value")) for k, v in variants.items()
                                              if isinstance(v.get("cpu_baseline"), dict)} if variants else None),
        }
        # `roofline`: the kernel with the largest share of GPU time, FLAT (scalars only), with the throughput family's
        # figures beside it under disc_* names; the long-form records follow under `details`
        flat = None
        if roof:
            flat = {k: v for k, v in roof.items() if not isinstance(v, (dict, list)) and k != "note"}
            flat.update({"disc_update_us": disc.get("us"), "disc_update_us_in_rounds": disc.get("us_in_rounds"),
                         "disc_update_frac": disc.get("frac"), "disc_update_frac_in_rounds": disc.get("frac_in_rounds"),
                         "disc_update_traffic": disc.get("traffic"), "disc_update_bound": disc.get("bound"),
                         "note": "dominant kernel by GPU time is a latency chain (read us_per_step); the MFMA-bound "
                                 "family is disc_update_*" +
                                 (f" ({disc['flop'] / 1e9:.3g} GFLOP per update vs {disc.get('peak', 157.3)} TFLOP/s)"
                                  if disc.get("flop") else "")})
        out = {
            "metric": "env-steps/sec (gen+disc round) GAIL HalfCheetah n_envs=1024",
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            **({k: v for k, v in spread.items() if k.startswith("ms_per_step_")} if spread else {}),
            "value_200": (value_long["value"] if value_long and value_long["rounds"] == 200 else None),
            "spread": spread, "value_long": value_long,
            "summary": summary,
            "config": {"workload": "GAIL round, config P (BASELINE.json configs[1]): seals/HalfCheetah-shaped "
                                   "obs17/act6 synthetic VecEnv, n_envs=1024/GPU x n_steps=16, disc BasicRewardNet "
                                   "256x256+RunningNorm, demo_batch 8192, 16 disc updates/round, PPO 32x32 "
                                   "minibatch 1024 x 10 epochs", "env_steps_per_round_per_gpu": per_round,
                       "parallelism": f"dp{world}" if world > 1 else "single",
                       "ppo_update": _dp_form_text(trainer.gen_algo) if world > 1 else "single GPU",
                       "ppo_update_choice": getattr(trainer.gen_algo, "dp_choice", None) if world > 1 else None,
                       "ppo_update_by_rank": by_rank},
            "roofline": flat, "cpu_baseline": base,
            "speedup_vs_cpu_baseline": (value / base["value"]) if base else None,
            "details": {"roofline_disc_update": disc or None, "roofline_gemm": gemm, "roofline_ppo_update": ppo or None},
            "variants": variants,
            # the same compact record once more at the END of the line (a reader that keeps only the tail still gets it)
            "tail_summary": summary,
        }
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
