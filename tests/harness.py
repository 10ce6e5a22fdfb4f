"""Shared test harness: runs one seeded GAIL/AIRL case through an implementation
(`reference` under the oracle shim, `oracle`, or the HIP product `hip`) and returns a flat
dict of arrays that the parity tests compare.

The three implementations expose the same constructor surface (that is the drop-in claim),
so a case is a config dict + a namespace of classes.
"""
from __future__ import annotations

import types as pytypes
from typing import Any, Dict

import numpy as np
import torch as th

from imitation_amd.vec_env import SyntheticVecEnv

CASES: Dict[str, Dict[str, Any]] = {
    # GAIL, Box actions, grad accumulation (minibatch 32 of 64), ring truncation + wrap
    # (capacity 96 < 128 transitions/round), episodes ending mid-rollout (horizon 10, n_steps 16).
    "gail_box": dict(algo="gail", n_envs=8, horizon=10, obs_dim=17, act_dim=6, n_discrete=None,
                     n_steps=16, ppo_batch=32, n_epochs=2, ent_coef=0.1, disc_hid=(32, 32),
                     demo_batch=64, demo_minibatch=32, n_disc=2, capacity=96, n_demo=500, rounds=3,
                     norm_policy=True, norm_disc=True, obs_dtype="float32"),
    # float64 observations (HalfCheetah-like dtype), wider disc, no input norm on the policy.
    "gail_f64": dict(algo="gail", n_envs=4, horizon=7, obs_dim=5, act_dim=3, n_discrete=None,
                     n_steps=8, ppo_batch=16, n_epochs=3, ent_coef=0.0, disc_hid=(64, 32),
                     demo_batch=16, demo_minibatch=None, n_disc=3, capacity=None, n_demo=100, rounds=2,
                     norm_policy=False, norm_disc=True, obs_dtype="float64"),
    # Discrete actions (CartPole-shaped): one-hot action preprocessing, Categorical policy.
    "gail_discrete": dict(algo="gail", n_envs=8, horizon=6, obs_dim=4, act_dim=2, n_discrete=2,
                          n_steps=8, ppo_batch=16, n_epochs=2, ent_coef=0.01, disc_hid=(32, 32),
                          demo_batch=32, demo_minibatch=None, n_disc=2, capacity=None, n_demo=200, rounds=2,
                          norm_policy=True, norm_disc=True, obs_dtype="float32"),
    # BASELINE config 1 in its canonical library form (docs/algorithms/gail.rst:36-91), scaled down: 8 envs, Discrete
    # actions, SB3 `MlpPolicy` (two 64-wide tanh towers, no feature norm), PPO minibatch 64-style (a quarter of
    # the rollout) x 5 epochs, gamma 0.95, lr 4e-4, ent_coef 0, replay capacity smaller than a round.
    "gail_cartpole": dict(algo="gail", n_envs=8, horizon=9, obs_dim=4, act_dim=2, n_discrete=2,
                          n_steps=16, ppo_batch=32, n_epochs=5, ent_coef=0.0, disc_hid=(32, 32),
                          demo_batch=64, demo_minibatch=None, n_disc=4, capacity=64, n_demo=300, rounds=3,
                          norm_policy=False, norm_disc=True, obs_dtype="float32", policy="mlp64",
                          ppo_kwargs=dict(gamma=0.95, learning_rate=4e-4)),
    # the fused five-launch discriminator update (D -> H -> H -> 1, H = 128) and, in pipelined rounds, the
    # one-launch round assembly + pre-assembled four-launch updates
    "gail_fused": dict(algo="gail", n_envs=8, horizon=10, obs_dim=17, act_dim=6, n_discrete=None,
                       n_steps=16, ppo_batch=32, n_epochs=2, ent_coef=0.1, disc_hid=(128, 128),
                       demo_batch=192, demo_minibatch=None, n_disc=3, capacity=None, n_demo=500, rounds=3,
                       norm_policy=True, norm_disc=True, obs_dtype="float32"),
    # SURVEY 8d variant H ("horizon" rollouts: n_steps far beyond the episode length -- long GAE scans, many episodes
    # completing inside one rollout, the replay ring holding a whole long rollout) at reduced width.
    "gail_horizon": dict(algo="gail", n_envs=4, horizon=25, obs_dim=17, act_dim=6, n_discrete=None,
                         n_steps=200, ppo_batch=100, n_epochs=2, ent_coef=0.1, disc_hid=(32, 32),
                         demo_batch=128, demo_minibatch=None, n_disc=2, capacity=None, n_demo=600, rounds=2,
                         norm_policy=True, norm_disc=True, obs_dtype="float32"),
    # SURVEY 8d variant T (tuned_hps/gail_seals_half_cheetah_best_hp_eval.json shape): rounds of n_steps = 4, PPO
    # minibatch 1/8 of the rollout x 5 epochs, gamma 0.95, clip 0.1, replay capacity SMALLER than a round
    # (truncation on every store) and demo batches larger than the ring (heavy sampling with replacement).
    "gail_tuned": dict(algo="gail", n_envs=16, horizon=6, obs_dim=17, act_dim=6, n_discrete=None,
                       n_steps=4, ppo_batch=8, n_epochs=5, ent_coef=0.0, disc_hid=(32, 32),
                       demo_batch=128, demo_minibatch=None, n_disc=4, capacity=32, n_demo=500, rounds=4,
                       norm_policy=True, norm_disc=True, obs_dtype="float32",
                       ppo_kwargs=dict(gamma=0.95, clip_range=0.1, gae_lambda=0.9)),
    # Policy towers outside the reference configs' [H, H] shapes (any SB3 `net_arch`): unequal pi / vf
    # towers of different depth behind the feature RunningNorm, Box actions.
    "gail_towers": dict(algo="gail", n_envs=8, horizon=10, obs_dim=17, act_dim=6, n_discrete=None,
                        n_steps=16, ppo_batch=32, n_epochs=3, ent_coef=0.05, disc_hid=(32, 32),
                        demo_batch=64, demo_minibatch=None, n_disc=2, capacity=None, n_demo=400, rounds=3,
                        norm_policy=True, norm_disc=True, obs_dtype="float32", policy="mlp64",
                        policy_kwargs=dict(net_arch=dict(pi=[48, 24], vf=[40, 40, 16]))),
    # Discrete head on a one-layer pi tower and NO vf tower (value_net reads the features), ReLU; remainder minibatch
    # (rollout of 40 rows in minibatches of 16).
    "gail_discrete_towers": dict(algo="gail", n_envs=5, horizon=6, obs_dim=4, act_dim=3, n_discrete=3,
                                 n_steps=8, ppo_batch=16, n_epochs=2, ent_coef=0.01, disc_hid=(32, 32),
                                 demo_batch=20, demo_minibatch=None, n_disc=2, capacity=None, n_demo=200, rounds=2,
                                 norm_policy=False, norm_disc=True, obs_dtype="float32", policy="mlp64",
                                 policy_kwargs=dict(net_arch=dict(pi=[24], vf=[]), activation_fn="relu")),
    # AIRL (log pi of the generator inside the discriminator logit) on three-layer ReLU towers.
    "airl_towers": dict(algo="airl", n_envs=8, horizon=10, obs_dim=11, act_dim=3, n_discrete=None,
                        n_steps=16, ppo_batch=64, n_epochs=2, ent_coef=0.0, disc_hid=(32,),
                        demo_batch=64, demo_minibatch=None, n_disc=2, capacity=None, n_demo=300, rounds=2,
                        norm_policy=True, norm_disc=True, obs_dtype="float32", normalize_output=True, policy="mlp64",
                        policy_kwargs=dict(net_arch=[64, 48, 32], activation_fn="relu")),
    # GAIL on IMAGE observations: uint8 [C, H, W] frames, SB3 `CnnPolicy` (NatureCNN) generator with a
    # Categorical head, the reference's `CnnRewardNet` discriminator (state + action, channel-first frames).
    "gail_image": dict(algo="gail", image=(4, 36, 36), n_envs=4, horizon=7, obs_dim=None, act_dim=3, n_discrete=4,
                       n_steps=8, ppo_batch=16, n_epochs=2, ent_coef=0.01, disc_hid=None,
                       demo_batch=16, demo_minibatch=None, n_disc=2, capacity=None, n_demo=64, rounds=2,
                       norm_policy=False, norm_disc=False, obs_dtype="uint8", ppo_kwargs=dict(learning_rate=1e-4)),
    # AIRL on image observations: log pi(a|s) of the NatureCNN policy inside the discriminator logit
    "airl_image": dict(algo="airl", image=(4, 36, 36), n_envs=4, horizon=7, obs_dim=None, act_dim=3, n_discrete=3,
                       n_steps=8, ppo_batch=16, n_epochs=2, ent_coef=0.01, disc_hid=None,
                       demo_batch=16, demo_minibatch=8, n_disc=2, capacity=None, n_demo=64, rounds=2,
                       norm_policy=False, norm_disc=False, obs_dtype="uint8", ppo_kwargs=dict(learning_rate=1e-4)),
    # tuned_hps/gail_seals_half_cheetah_best_hp_eval.json:2-44 VERBATIM at reduced width (16 envs instead of the scripts'
    # env count; every ratio kept: rl.batch_size / n_envs = 4 steps, replay capacity = 1/8 of a round, demo batches far
    # larger than the ring, 8 updates per round): PPO clip 0.1, ent_coef 3.99e-6, gae_lambda 0.95, gamma 0.95, lr 2.625e-4,
    # max_grad_norm 0.8, 5 epochs, vf_coef 0.1148; BasicRewardNet at its DEFAULT 32 x 32 with input RunningNorm, wrapped in
    # NormalizedRewardNet (`reward.normalize_output_layer`: GAIL's processed reward bypasses it, gail.py:65-73).
    "gail_tuned_hps": dict(algo="gail", n_envs=16, horizon=6, obs_dim=17, act_dim=6, n_discrete=None,
                           n_steps=4, ppo_batch=8, n_epochs=5, ent_coef=3.992371122209408e-6, disc_hid=(32, 32),
                           demo_batch=128, demo_minibatch=None, n_disc=8, capacity=8, n_demo=500, rounds=4,
                           norm_policy=True, norm_disc=True, obs_dtype="float32", normalize_output=True,
                           ppo_kwargs=dict(clip_range=0.1, gae_lambda=0.95, gamma=0.95,
                                           learning_rate=0.00026250519057717037, max_grad_norm=0.8,
                                           vf_coef=0.11483689492120866)),
    # tuned_hps/airl_seals_ant_best_hp_eval.json:2-44 VERBATIM at reduced width (16 envs x 8 steps per round, Ant-shaped
    # obs 27 / act 8): PPO minibatch 16 (as written), clip 0.3, ent_coef 3.28e-6, gae_lambda 0.8, gamma 0.995, lr 3.25e-5,
    # max_grad_norm 0.9, 10 epochs, vf_coef 0.435; BasicShapedRewardNet defaults (reward 32, potential 32 x 32, no next
    # state in the reward input) + input RunningNorm + NormalizedRewardNet; capacity = one round, 16 updates per round.
    "airl_tuned_hps": dict(algo="airl", n_envs=16, horizon=12, obs_dim=27, act_dim=8, n_discrete=None,
                           n_steps=8, ppo_batch=16, n_epochs=10, ent_coef=3.27750078482474e-6, disc_hid=(32,),
                           demo_batch=64, demo_minibatch=None, n_disc=16, capacity=128, n_demo=400, rounds=2,
                           norm_policy=True, norm_disc=True, obs_dtype="float32", normalize_output=True,
                           use_next_state=False,
                           ppo_kwargs=dict(clip_range=0.3, gae_lambda=0.8, gamma=0.995,
                                           learning_rate=3.249429831179079e-5, max_grad_norm=0.9,
                                           vf_coef=0.4351450387648799)),
    # EMANorm (`util/networks.py:137-201`) wherever the reference takes a normalisation layer for the reward net: input
    # norms of both stacks and the NormalizedRewardNet output layer (AIRL: the processed reward IS normalised).
    "airl_ema": dict(algo="airl", n_envs=8, horizon=10, obs_dim=11, act_dim=3, n_discrete=None,
                     n_steps=16, ppo_batch=32, n_epochs=2, ent_coef=0.0, disc_hid=(32,),
                     demo_batch=64, demo_minibatch=None, n_disc=3, capacity=None, n_demo=300, rounds=3,
                     norm_policy=True, norm_disc=True, obs_dtype="float32", normalize_output=True, disc_norm="ema"),
    # BASELINE config 1's plumbing ("CPU SubprocVecEnv"): the environment speaks ONLY the gym / SB3 VecEnv protocol --
    # per-env info dicts with `terminal_observation`, `TimeLimit.truncated` and Monitor's `episode` entries
    # (`vec_env.GymStyleVecEnv`) -- so the wrappers take their generic per-env branch (`rewards/reward_wrapper.py:98-109`,
    # `data/wrappers.py:69-91`); CartPole-shaped, Discrete actions, SB3 MlpPolicy 64 x 64, variable-length bookkeeping.
    "gail_generic_vecenv": dict(algo="gail", n_envs=8, horizon=9, obs_dim=4, act_dim=2, n_discrete=2,
                                n_steps=16, ppo_batch=32, n_epochs=3, ent_coef=0.0, disc_hid=(32, 32),
                                demo_batch=64, demo_minibatch=None, n_disc=3, capacity=96, n_demo=300, rounds=3,
                                norm_policy=False, norm_disc=True, obs_dtype="float32", policy="mlp64",
                                generic_vecenv=True, reward_scale=1.0, ppo_kwargs=dict(gamma=0.95, learning_rate=4e-4)),
    # BasicRewardNet(use_next_state=True, use_done=True) (`rewards/reward_nets.py:441-457`): 17 + 6 + 17 + 1 = 41 inputs
    # through the default 32 x 32 stack; episodes end inside the rollout so the done column is not constant.
    "gail_next_done": dict(algo="gail", n_envs=8, horizon=5, obs_dim=17, act_dim=6, n_discrete=None,
                           n_steps=16, ppo_batch=32, n_epochs=2, ent_coef=0.01, disc_hid=(32, 32),
                           demo_batch=96, demo_minibatch=None, n_disc=3, capacity=None, n_demo=400, rounds=3,
                           norm_policy=True, norm_disc=True, obs_dtype="float32",
                           disc_kwargs=dict(use_next_state=True, use_done=True)),
    # 18 + 6 + 18 = 42 inputs (rows of 44 floats) through the wide fused update (128-wide stack, rows of 25 .. 64 floats:
    # `disc_fb_kernel<H, 64, 64>`), pipelined rounds with the one-launch round assembly. (Without the done column: with it
    # this configuration amplifies a 1e-6 parameter perturbation of the REFERENCE ITSELF to 2e-3 within three rounds --
    # ReLU kinks; the column's assembly pass is covered by `gail_next_done` and by tests/test_disc_fused_gpu.py.)
    "gail_fused_wide": dict(algo="gail", n_envs=8, horizon=5, obs_dim=18, act_dim=6, n_discrete=None,
                            n_steps=16, ppo_batch=32, n_epochs=2, ent_coef=0.01, disc_hid=(128, 128),
                            demo_batch=192, demo_minibatch=None, n_disc=3, capacity=None, n_demo=500, rounds=3,
                            norm_policy=True, norm_disc=True, obs_dtype="float32",
                            disc_kwargs=dict(use_next_state=True)),
    # AIRL, shaped reward net, NormalizedRewardNet output norm (script default), use_next_state.
    "airl_box": dict(algo="airl", n_envs=8, horizon=10, obs_dim=11, act_dim=3, n_discrete=None,
                     n_steps=16, ppo_batch=32, n_epochs=2, ent_coef=0.0, disc_hid=(32,),
                     demo_batch=64, demo_minibatch=None, n_disc=2, capacity=None, n_demo=300, rounds=2,
                     norm_policy=True, norm_disc=True, obs_dtype="float32", normalize_output=True),
}


def make_demo_arrays(cfg, seed: int = 1) -> Dict[str, np.ndarray]:
    rng = np.random.default_rng(seed)
    n, od, ad = cfg["n_demo"], cfg["obs_dim"], cfg["act_dim"]
    dt = np.dtype(cfg["obs_dtype"])
    if cfg.get("image"):
        obs = rng.integers(0, 256, (n + 1, *cfg["image"])).astype(np.uint8)
        acts = rng.integers(0, cfg["n_discrete"], n).astype(np.int64)
        dones = np.zeros(n, dtype=bool)
        dones[cfg["horizon"] - 1::cfg["horizon"]] = True
        return dict(obs=obs[:-1], acts=acts, next_obs=obs[1:], dones=dones)
    obs = rng.standard_normal((n, od)).astype(dt)
    if cfg["n_discrete"] is None:
        acts = rng.uniform(-1, 1, (n, ad)).astype(np.float32)
    else:
        acts = rng.integers(0, cfg["n_discrete"], n).astype(np.int64)
    nxt = (0.9 * obs + 0.1 * rng.standard_normal((n, od))).astype(dt)
    dones = np.zeros(n, dtype=bool)
    dones[cfg["horizon"] - 1::cfg["horizon"]] = True
    return dict(obs=obs, acts=acts, next_obs=nxt, dones=dones)


def namespace(impl: str) -> pytypes.SimpleNamespace:
    """Classes of one implementation under common names."""
    if impl == "reference":
        from oracle import ref_shim

        ref_shim.install()
        from imitation.algorithms.adversarial.airl import AIRL
        from imitation.algorithms.adversarial.gail import GAIL
        from imitation.data import types as rtypes
        from imitation.policies.base import FeedForward32Policy, NormalizeFeaturesExtractor
        from imitation.rewards import reward_nets as rn
        from imitation.util import logger as rlog
        from imitation.util.networks import EMANorm, RunningNorm
        from oracle import sb3_restated as sb

        def transitions(**kw):
            n = len(kw["obs"])
            return rtypes.Transitions(infos=np.array([{}] * n), **kw)

        return pytypes.SimpleNamespace(
            GAIL=GAIL, AIRL=AIRL, PPO=sb.PPO, FeedForward32Policy=FeedForward32Policy,
            ActorCriticPolicy=sb.ActorCriticPolicy, ActorCriticCnnPolicy=sb.ActorCriticCnnPolicy,
            CnnRewardNet=rn.CnnRewardNet,
            NormalizeFeaturesExtractor=NormalizeFeaturesExtractor, RunningNorm=RunningNorm, EMANorm=EMANorm,
            BasicRewardNet=rn.BasicRewardNet, BasicShapedRewardNet=rn.BasicShapedRewardNet,
            NormalizedRewardNet=rn.NormalizedRewardNet, Transitions=transitions,
            configure_logger=lambda d: rlog.configure(d, []))
    if impl == "oracle":
        from oracle import imitation_restated as o
        from oracle import sb3_restated as sb

        return pytypes.SimpleNamespace(
            GAIL=o.GAIL, AIRL=o.AIRL, PPO=sb.PPO, FeedForward32Policy=o.FeedForward32Policy,
            ActorCriticPolicy=sb.ActorCriticPolicy, ActorCriticCnnPolicy=sb.ActorCriticCnnPolicy,
            CnnRewardNet=o.CnnRewardNet,
            NormalizeFeaturesExtractor=o.NormalizeFeaturesExtractor, RunningNorm=o.RunningNorm, EMANorm=o.EMANorm,
            BasicRewardNet=o.BasicRewardNet, BasicShapedRewardNet=o.BasicShapedRewardNet,
            NormalizedRewardNet=o.NormalizedRewardNet, Transitions=lambda **kw: o.Transitions(**kw),
            configure_logger=lambda d: o.configure_logger(d, []))
    if impl == "hip":
        import imitation_amd as p

        return pytypes.SimpleNamespace(
            GAIL=p.GAIL, AIRL=p.AIRL, PPO=p.PPO, FeedForward32Policy=p.FeedForward32Policy,
            ActorCriticPolicy=p.ActorCriticPolicy, ActorCriticCnnPolicy=p.cnn_policy.ActorCriticCnnPolicy,
            CnnRewardNet=p.modules.CnnRewardNet,
            NormalizeFeaturesExtractor=p.NormalizeFeaturesExtractor, RunningNorm=p.RunningNorm, EMANorm=p.EMANorm,
            BasicRewardNet=p.BasicRewardNet, BasicShapedRewardNet=p.BasicShapedRewardNet,
            NormalizedRewardNet=p.NormalizedRewardNet, Transitions=lambda **kw: p.Transitions(**kw),
            configure_logger=lambda d: p.configure_logger(d, []))
    raise ValueError(impl)


def build_trainer(impl: str, cfg, log_dir: str, device: str = "cpu", module_net: bool = False):
    ns = namespace(impl)
    if module_net:  # product only: the autograd-capable nn.Module reward nets on the HIP custom ops
        from imitation_amd import modules as m

        ns.BasicRewardNet, ns.BasicShapedRewardNet, ns.NormalizedRewardNet = (m.BasicRewardNet, m.BasicShapedRewardNet,
                                                                             m.NormalizedRewardNet)
        disc_norm = m.EMANorm if cfg.get("disc_norm") == "ema" else m.RunningNorm
    else:
        disc_norm = ns.EMANorm if cfg.get("disc_norm") == "ema" else ns.RunningNorm
    th.manual_seed(0)
    np.random.seed(0)
    if cfg.get("image"):
        from imitation_amd.vec_env import SyntheticImageVecEnv
        venv = SyntheticImageVecEnv(num_envs=cfg["n_envs"], shape=cfg["image"], act_dim=cfg["act_dim"],
                                    horizon=cfg["horizon"], seed=0, n_discrete=cfg["n_discrete"])
    else:
        venv = SyntheticVecEnv(num_envs=cfg["n_envs"], obs_dim=cfg["obs_dim"], act_dim=cfg["act_dim"],
                               horizon=cfg["horizon"], seed=0, obs_dtype=np.dtype(cfg["obs_dtype"]),
                               n_discrete=cfg["n_discrete"], reward_scale=cfg.get("reward_scale", 0.0))
        if cfg.get("generic_vecenv"):
            from imitation_amd.vec_env import GymStyleVecEnv
            venv = GymStyleVecEnv(venv)
    pk = {}
    if cfg["norm_policy"]:
        pk = dict(features_extractor_class=ns.NormalizeFeaturesExtractor,
                  features_extractor_kwargs=dict(normalize_class=ns.RunningNorm))
    policy_cls = ns.ActorCriticPolicy if cfg.get("policy") == "mlp64" else ns.FeedForward32Policy  # SB3 MlpPolicy
    extra = dict(cfg.get("policy_kwargs", {}))
    if "activation_fn" in extra:
        extra["activation_fn"] = {"relu": th.nn.ReLU, "tanh": th.nn.Tanh}[extra["activation_fn"]]
    pk = dict(pk, **extra)
    if cfg.get("image"):
        policy_cls = ns.ActorCriticCnnPolicy
    algo = ns.PPO(policy_cls, venv, n_steps=cfg["n_steps"], batch_size=cfg["ppo_batch"],
                  n_epochs=cfg["n_epochs"], ent_coef=cfg["ent_coef"], seed=0, policy_kwargs=pk, device=device,
                  **cfg.get("ppo_kwargs", {}))
    kw = dict(normalize_input_layer=disc_norm) if cfg["norm_disc"] else {}
    if cfg.get("image"):
        net = ns.CnnRewardNet(venv.observation_space, venv.action_space, hwc_format=False)
        cls = ns.GAIL if cfg["algo"] == "gail" else ns.AIRL
    elif cfg["algo"] == "gail":
        net = ns.BasicRewardNet(venv.observation_space, venv.action_space, hid_sizes=cfg["disc_hid"], **kw,
                                **cfg.get("disc_kwargs", {}))
        if cfg.get("normalize_output"):
            net = ns.NormalizedRewardNet(net, disc_norm)
        cls = ns.GAIL
    else:
        net = ns.BasicShapedRewardNet(venv.observation_space, venv.action_space,
                                      reward_hid_sizes=cfg["disc_hid"], potential_hid_sizes=(32, 32),
                                      use_next_state=cfg.get("use_next_state", True), **kw)
        if cfg.get("normalize_output"):
            net = ns.NormalizedRewardNet(net, disc_norm)
        cls = ns.AIRL
    demos = ns.Transitions(**make_demo_arrays(cfg))
    trainer = cls(demonstrations=demos, demo_batch_size=cfg["demo_batch"], venv=venv, gen_algo=algo,
                  reward_net=net, demo_minibatch_size=cfg["demo_minibatch"],
                  n_disc_updates_per_round=cfg["n_disc"], gen_replay_buffer_capacity=cfg["capacity"],
                  custom_logger=ns.configure_logger(log_dir), allow_variable_horizon=False,
                  **({"gen_train_timesteps": cfg["gen_train_timesteps"]} if cfg.get("gen_train_timesteps") else {}))
    return trainer, venv


def _np(x):
    if isinstance(x, th.Tensor):
        return x.detach().cpu().numpy()
    return np.asarray(x)


def snapshot(trainer) -> Dict[str, np.ndarray]:
    """State that must agree across implementations after a run."""
    out: Dict[str, np.ndarray] = {}
    for k, v in trainer._reward_net.state_dict().items():
        out[f"disc/{k}"] = _np(v)
    for k, v in trainer.gen_algo.policy.state_dict().items():
        out[f"policy/{k}"] = _np(v)
    buf = trainer._gen_replay_buffer
    arrays = buf._buffer._arrays if hasattr(buf, "_buffer") else buf._arrays
    for k in ("obs", "acts", "next_obs", "dones"):
        out[f"replay/{k}"] = _np(arrays[k])
    inner = buf._buffer if hasattr(buf, "_buffer") else buf
    out["replay/_idx"] = np.asarray(inner._idx)
    out["replay/_n_data"] = np.asarray(inner._n_data)
    rb = trainer.gen_algo.rollout_buffer
    for k in ("rewards", "values", "log_probs", "advantages", "returns", "actions", "observations"):
        out[f"rollout/{k}"] = _np(getattr(rb, k)).reshape(-1)
    out["counters"] = np.asarray([trainer._global_step, trainer._disc_step, trainer.gen_algo.num_timesteps])
    eps = list(getattr(trainer.gen_algo, "ep_info_buffer", None) or [])
    if eps:   # gym-style envs only: Monitor's episode entries as SB3 collects them ([SB3 collect_rollouts] App. A.4)
        out["ep_info"] = np.asarray([[e["r"], e["l"]] for e in eps], dtype=np.float64)
        infos = arrays["infos"] if "infos" in arrays else getattr(inner, "_infos", None)
        out["replay/infos_episode_lens"] = np.asarray([i["episode"]["l"] if (i and "episode" in i) else 0
                                                       for i in infos], dtype=np.int64)
        out["replay/infos_has_terminal_obs"] = np.asarray([bool(i) and "terminal_observation" in i for i in infos])
    return out


def run_case(impl: str, name: str, log_dir: str, device: str = "cpu", sync_disc: bool = False,
             pipeline: bool = True, discrete_sampling: str = None, module_net: bool = False) -> Dict[str, np.ndarray]:
    cfg = CASES[name]
    trainer, venv = build_trainer(impl, cfg, log_dir, device, module_net=module_net)
    if discrete_sampling is not None:  # product only: "inverse_cdf" = the in-kernel fast sampler
        trainer.gen_algo.policy.discrete_sampling = discrete_sampling
    if hasattr(trainer, "pipeline_rounds"):
        trainer.pipeline_rounds = pipeline
    stats = []
    if hasattr(trainer, "_log_disc_stats") and not sync_disc:
        # product: record where the statistics are logged, so `train()` keeps its own schedule
        # (all updates of a round enqueued before any statistics are read back)
        orig_log = trainer._log_disc_stats

        def recording_log(*a, **kw):
            s = orig_log(*a, **kw)
            stats.append([float(s[k]) for k in sorted(s)])
            return s

        trainer._log_disc_stats = recording_log
    else:
        orig = trainer.train_disc

        def recording_train_disc(**kw):
            s = orig(**kw)
            stats.append([float(s[k]) for k in sorted(s)])
            return s

        trainer.train_disc = recording_train_disc  # a replaced train_disc is called once per update
    trainer.train(cfg["rounds"] * cfg["n_envs"] * cfg["n_steps"])
    out = snapshot(trainer)
    out["disc_stats"] = np.asarray(stats, dtype=np.float64)
    # one more reward query on fixed inputs through the public RewardFn surface
    rng = np.random.default_rng(7)
    n = 32
    if cfg.get("image"):
        draw = lambda: rng.integers(0, 256, (n, *cfg["image"])).astype(np.uint8)  # noqa: E731
    else:
        draw = lambda: rng.standard_normal((n, cfg["obs_dim"])).astype(np.dtype(cfg["obs_dtype"]))  # noqa: E731
    s = draw()
    if cfg["n_discrete"] is None:
        a = rng.uniform(-1, 1, (n, cfg["act_dim"])).astype(np.float32)
    else:
        a = rng.integers(0, cfg["n_discrete"], n)
    ns_ = draw()
    d = rng.random(n) < 0.2
    out["reward_train_predict"] = _np(trainer.reward_train.predict(s, a, ns_, d))
    out["reward_test_predict"] = _np(trainer.reward_test.predict(s, a, ns_, d))
    return out


# Keys whose values are integer/boolean bookkeeping and must match bit-exactly.
EXACT_KEYS = ("replay/dones", "replay/_idx", "replay/_n_data", "counters")


# --------------------------------------------------------------------------------------------
# data/rollout.py cases: trajectory collection through each implementation.

ROLLOUT_CASES: Dict[str, Dict[str, Any]] = {
    # Plain callable policy: exercises episode bookkeeping, retirement of envs and the final
    # shuffle only (bit-exact across implementations).
    "rollout_callable": dict(kind="callable", n_envs=6, obs_dim=5, act_dim=3, horizon=7, n_discrete=None,
                             min_timesteps=60, min_episodes=11),
    # Freshly initialised FeedForward32Policy with RunningNorm features, stochastic actions.
    "rollout_policy": dict(kind="policy", n_envs=8, obs_dim=9, act_dim=4, horizon=9, n_discrete=None,
                           min_timesteps=100, min_episodes=None, deterministic=False),
    # Discrete actions, deterministic (mode) prediction.
    "rollout_discrete_det": dict(kind="policy", n_envs=5, obs_dim=4, act_dim=2, horizon=6, n_discrete=3,
                                 min_timesteps=None, min_episodes=12, deterministic=True),
    # Discrete actions SAMPLED ([SB3 CategoricalDistribution.sample] = torch.multinomial on the global generator).
    "rollout_discrete_sto": dict(kind="policy", n_envs=6, obs_dim=5, act_dim=2, horizon=7, n_discrete=4,
                                 min_timesteps=90, min_episodes=None, deterministic=False),
}


def rollout_namespace(impl: str):
    if impl == "reference":
        from oracle import ref_shim

        ref_shim.install()
        from imitation.data import rollout as r

        return pytypes.SimpleNamespace(generate_trajectories=r.generate_trajectories,
                                       make_sample_until=r.make_sample_until, rollout_stats=r.rollout_stats,
                                       discounted_sum=r.discounted_sum)
    if impl == "oracle":
        from oracle import imitation_restated as o

        return pytypes.SimpleNamespace(generate_trajectories=o.generate_trajectories,
                                       make_sample_until=o.make_sample_until, rollout_stats=o.rollout_stats,
                                       discounted_sum=o.discounted_sum)
    from imitation_amd import rollout as r

    return pytypes.SimpleNamespace(generate_trajectories=r.generate_trajectories,
                                   make_sample_until=r.make_sample_until, rollout_stats=r.rollout_stats,
                                   discounted_sum=r.discounted_sum)


def run_rollout_case(impl: str, name: str, device: str = "cpu") -> Dict[str, np.ndarray]:
    cfg = ROLLOUT_CASES[name]
    ns, rns = namespace(impl), rollout_namespace(impl)
    th.manual_seed(0)
    np.random.seed(0)
    venv = SyntheticVecEnv(num_envs=cfg["n_envs"], obs_dim=cfg["obs_dim"], act_dim=cfg["act_dim"],
                           horizon=cfg["horizon"], seed=3, n_discrete=cfg["n_discrete"], stagger=True,
                           reward_scale=1.0)
    if cfg["kind"] == "callable":
        W = np.random.default_rng(11).standard_normal((cfg["obs_dim"], cfg["act_dim"]))
        policy = lambda obs, state, starts: (np.tanh(4.0 * obs @ W).astype(np.float32), None)  # noqa: E731
        kw = {}
    else:
        pk = dict(features_extractor_class=ns.NormalizeFeaturesExtractor,
                  features_extractor_kwargs=dict(normalize_class=ns.RunningNorm))
        algo = ns.PPO(ns.FeedForward32Policy, venv, n_steps=8, batch_size=8, seed=0, policy_kwargs=pk, device=device)
        # give the feature normaliser non-trivial statistics (one train-mode pass), then roll out in eval mode
        warm = np.random.default_rng(12).standard_normal((64, cfg["obs_dim"])).astype(np.float32) * 0.3 + 0.1
        algo.policy.set_training_mode(True)
        algo.policy.predict_values(th.as_tensor(warm, device=device) if impl != "hip" else warm)
        policy = algo
        kw = dict(deterministic_policy=cfg["deterministic"])
    until = rns.make_sample_until(min_timesteps=cfg["min_timesteps"], min_episodes=cfg["min_episodes"])
    trajs = rns.generate_trajectories(policy, venv, until, rng=np.random.default_rng(5), **kw)
    out = dict(lens=np.asarray([len(t.acts) for t in trajs]),
               obs=np.concatenate([np.asarray(t.obs) for t in trajs]),
               acts=np.concatenate([np.asarray(t.acts) for t in trajs]),
               rews=np.concatenate([np.asarray(t.rews) for t in trajs]),
               terminal=np.asarray([t.terminal for t in trajs]))
    stats = rns.rollout_stats(trajs)
    out["stats"] = np.asarray([stats[k] for k in sorted(stats)], dtype=np.float64)
    out["disc_sum"] = np.asarray([rns.discounted_sum(np.asarray(t.rews, dtype=np.float64), 0.9) for t in trajs])
    return out


# ----------------------------------------------------------------------------------------------
# Behavioural cloning (MLP slice): seeded cases shared by the golden generator
# (reference `algorithms/bc.py` under the shim), the oracle pin and the HIP parity test.
BC_CASES: Dict[str, Dict[str, Any]] = {
    # default policy (FeedForward32Policy, Flatten features), Box actions, two epochs + drop_last remainder
    "bc_box": dict(obs_dim=5, act_dim=3, n_discrete=None, n_demo=200, batch_size=32, ent_weight=1e-3,
                   train=dict(n_epochs=2), norm_policy=False, log_interval=3),
    # Discrete actions, n_batches that stops inside the second epoch, larger entropy weight
    "bc_discrete": dict(obs_dim=4, act_dim=2, n_discrete=2, n_demo=150, batch_size=16, ent_weight=1e-2,
                        train=dict(n_batches=13), norm_policy=False, log_interval=4),
    # train-mode NormalizeFeaturesExtractor(RunningNorm): `evaluate_actions` updates the statistics every batch
    "bc_norm": dict(obs_dim=7, act_dim=2, n_discrete=None, n_demo=128, batch_size=64, ent_weight=1e-3,
                    train=dict(n_epochs=3), norm_policy=True, log_interval=1),
    # gradient accumulation (minibatch 8 of 24; batches straddle epoch ends; incomplete last batch) + L2 term
    # image observations (uint8 [C, H, W]) through the NatureCNN actor-critic policy (BASELINE config 4 shape,
    # scaled down: 4 x 36 x 36 is the smallest image the 8/4 - 4/2 - 3/1 convolution stack accepts), Discrete(6)
    "bc_cnn": dict(image=(4, 36, 36), obs_dim=None, act_dim=6, n_discrete=6, n_demo=96, batch_size=32,
                   ent_weight=1e-3, train=dict(n_epochs=2), norm_policy=False, log_interval=2),
    # MLP policies outside the fused kernels' shapes (any SB3 `net_arch`): unequal ReLU towers behind the feature
    # RunningNorm, Box actions, gradient accumulation + L2; and a Discrete head on a single tanh layer
    "bc_towers": dict(obs_dim=7, act_dim=3, n_discrete=None, n_demo=120, batch_size=24, minibatch_size=12,
                      ent_weight=1e-3, l2_weight=1e-2, train=dict(n_epochs=2), norm_policy=True, log_interval=2,
                      policy_kwargs=dict(net_arch=dict(pi=[48, 24], vf=[16]), activation_fn="relu")),
    "bc_towers_discrete": dict(obs_dim=5, act_dim=3, n_discrete=3, n_demo=90, batch_size=16, ent_weight=1e-2,
                               train=dict(n_epochs=2), norm_policy=False, log_interval=2,
                               policy_kwargs=dict(net_arch=[40])),
    # DiagGaussian head on the NatureCNN policy
    "bc_cnn_box": dict(image=(4, 36, 36), obs_dim=None, act_dim=2, n_discrete=None, n_demo=64, batch_size=32,
                       ent_weight=1e-3, train=dict(n_epochs=2), norm_policy=False, log_interval=1),
    "bc_accum_l2": dict(obs_dim=6, act_dim=2, n_discrete=None, n_demo=100, batch_size=24, minibatch_size=8,
                        ent_weight=1e-3, l2_weight=1e-2, train=dict(n_epochs=2), norm_policy=False, log_interval=2),
}


def bc_namespace(impl: str) -> pytypes.SimpleNamespace:
    if impl == "reference":
        ns = namespace("reference")
        from imitation.algorithms import bc as rbc
        from oracle import sb3_restated as sb

        ns.BC, ns.ActorCriticCnnPolicy = rbc.BC, sb.ActorCriticCnnPolicy
        return ns
    if impl == "oracle":
        ns = namespace("oracle")
        from oracle import imitation_restated as o
        from oracle import sb3_restated as sb

        ns.BC, ns.ActorCriticCnnPolicy = o.BC, sb.ActorCriticCnnPolicy
        return ns
    ns = namespace("hip")
    import imitation_amd as p

    ns.BC = p.bc.BC
    ns.ActorCriticCnnPolicy = getattr(getattr(p, "cnn_policy", None), "ActorCriticCnnPolicy", None)
    return ns


def run_bc_case(impl: str, name: str, log_dir: str, device: str = "cpu") -> Dict[str, np.ndarray]:
    """Trains BC on seeded synthetic demonstrations; returns the policy state and every logged row."""
    from imitation_amd import spaces

    cfg = BC_CASES[name]
    ns = bc_namespace(impl)
    th.manual_seed(0)
    np.random.seed(0)
    rng = np.random.default_rng(3)
    n, od, ad = cfg["n_demo"], cfg["obs_dim"], cfg["act_dim"]
    policy = None
    if cfg.get("image"):
        obs = rng.integers(0, 256, (n, *cfg["image"]), dtype=np.uint8)
        if cfg["n_discrete"] is None:
            acts = np.tanh(obs.reshape(n, -1)[:, :ad] / 128.0 - 1.0 + 0.1 * rng.standard_normal((n, ad))).astype(np.float32)
            act_space = spaces.Box(-1.0, 1.0, (ad,), np.float32)
        else:
            acts = (obs.reshape(n, -1)[:, :7].sum(axis=1) % cfg["n_discrete"]).astype(np.int64)
            act_space = spaces.Discrete(cfg["n_discrete"])
        obs_space = spaces.Box(0, 255, cfg["image"], np.uint8)
        policy = ns.ActorCriticCnnPolicy(observation_space=obs_space, action_space=act_space,
                                         lr_schedule=lambda _: 1.0)
    else:
        obs = rng.standard_normal((n, od)).astype(np.float32)
        if cfg["n_discrete"] is None:
            acts = np.tanh(obs[:, :ad] + 0.1 * rng.standard_normal((n, ad))).astype(np.float32)
            act_space = spaces.Box(-1.0, 1.0, (ad,), np.float32)
        else:
            acts = (obs[:, 0] > 0).astype(np.int64)
            act_space = spaces.Discrete(cfg["n_discrete"])
        obs_space = spaces.Box(-np.inf, np.inf, (od,), np.float32)
    demos = ns.Transitions(obs=obs, acts=acts, next_obs=obs.copy(), dones=np.zeros(n, dtype=bool))
    extra = dict(cfg.get("policy_kwargs", {}))
    if "activation_fn" in extra:
        extra["activation_fn"] = {"relu": th.nn.ReLU, "tanh": th.nn.Tanh}[extra["activation_fn"]]
    if cfg["norm_policy"] or extra:
        nk = (dict(features_extractor_class=ns.NormalizeFeaturesExtractor,
                   features_extractor_kwargs=dict(normalize_class=ns.RunningNorm)) if cfg["norm_policy"] else {})
        pcls = ns.ActorCriticPolicy if extra else ns.FeedForward32Policy
        policy = pcls(observation_space=obs_space, action_space=act_space, lr_schedule=lambda _: 1.0, **nk, **extra)
    logger = ns.configure_logger(log_dir)
    rows = []
    orig_dump = logger.dump

    def dump(step=0):
        kv = dict(logger.name_to_value) if hasattr(logger, "name_to_value") else {}
        if not kv and hasattr(logger, "default_logger"):
            kv = dict(logger.default_logger.name_to_value)
        rows.append([float(kv[k]) for k in sorted(kv) if k.startswith("bc/") or k == "batch_size"])
        return orig_dump(step)

    logger.dump = dump
    kw = dict(device=device) if impl == "hip" else {}
    trainer = ns.BC(observation_space=obs_space, action_space=act_space, rng=np.random.default_rng(0), policy=policy,
                    demonstrations=demos, batch_size=cfg["batch_size"], ent_weight=cfg["ent_weight"],
                    minibatch_size=cfg.get("minibatch_size"), l2_weight=cfg.get("l2_weight", 0.0),
                    custom_logger=logger, **kw)
    trainer.train(log_interval=cfg["log_interval"], progress_bar=False, **cfg["train"])
    out = {f"policy/{k}": _np(v) for k, v in trainer.policy.state_dict().items()
           if not (cfg.get("image") and k.startswith(("pi_features_extractor.", "vf_features_extractor.")))}  # aliases
    out["log_rows"] = np.asarray(rows, dtype=np.float64)
    out["torch_rng_after"] = th.get_rng_state().numpy().copy()   # the loader consumed the global generator identically
    return out
