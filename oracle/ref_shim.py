"""ORACLE tooling (local container only): run the reference's OWN code under a shim.

`/root/reference` is pure Python but imports `gymnasium` / `stable_baselines3` /
`torch.utils.tensorboard`, none of which are installed. `install()`
registers `oracle.sb3_restated` + `imitation_amd.spaces` under those module names so that
`imitation.algorithms.adversarial.{common,gail,airl}`, `imitation.rewards.*`,
`imitation.data.*`, `imitation.util.*`, `imitation.policies.base` import UNMODIFIED from
`/root/reference/src` ("Tier R"). Used only by `tests/golden/make_golden.py` (to produce
committed fixtures) and by local-only tests that are skipped when /root/reference is
absent (the GPU box never has it).
"""
from __future__ import annotations

import os
import sys
import types

REFERENCE_SRC = "/root/reference/src"


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_SRC, "imitation"))


def _mod(name: str, **attrs) -> types.ModuleType:
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install() -> None:
    """Idempotently install the shim modules and put the reference on `sys.path`."""
    if "stable_baselines3" in sys.modules and getattr(sys.modules["stable_baselines3"], "_IS_ORACLE_SHIM", False):
        return
    from imitation_amd import spaces as sp
    from imitation_amd import vec_env as ve
    from oracle import sb3_restated as sb

    class _Env:  # gymnasium.Env / Wrapper placeholders (only subclassed, never stepped here)
        pass

    class _Wrapper(_Env):
        def __init__(self, env):
            self.env = env

    class _DictSpace(sp.Space):
        pass

    g_spaces = _mod("gymnasium.spaces", Space=sp.Space, Box=sp.Box, Discrete=sp.Discrete, Dict=_DictSpace)
    g_spaces.utils = _mod("gymnasium.spaces.utils", flatdim=sp.flatdim)
    _mod("gymnasium", Space=sp.Space, Env=_Env, Wrapper=_Wrapper, spaces=g_spaces, make=None)

    class _DummyVecEnv(ve.VecEnv):
        pass

    class _SubprocVecEnv(ve.VecEnv):
        pass

    class _Monitor(_Wrapper):
        pass

    common = _mod("stable_baselines3.common")
    common.vec_env = _mod("stable_baselines3.common.vec_env", VecEnv=ve.VecEnv, VecEnvWrapper=ve.VecEnvWrapper,
                          DummyVecEnv=_DummyVecEnv, SubprocVecEnv=_SubprocVecEnv)
    common.base_class = _mod("stable_baselines3.common.base_class", BaseAlgorithm=sb.BaseAlgorithm)
    common.on_policy_algorithm = _mod("stable_baselines3.common.on_policy_algorithm",
                                      OnPolicyAlgorithm=sb.OnPolicyAlgorithm)
    common.policies = _mod("stable_baselines3.common.policies", BasePolicy=sb.BasePolicy,
                           ActorCriticPolicy=sb.ActorCriticPolicy, ActorCriticCnnPolicy=sb.ActorCriticCnnPolicy)
    common.distributions = _mod("stable_baselines3.common.distributions",
                                DiagGaussianDistribution=sb.DiagGaussianDistribution,
                                SquashedDiagGaussianDistribution=sb.SquashedDiagGaussianDistribution,
                                CategoricalDistribution=sb.CategoricalDistribution)
    common.preprocessing = _mod("stable_baselines3.common.preprocessing", preprocess_obs=sb.preprocess_obs,
                                get_flattened_obs_dim=sb.get_flattened_obs_dim, is_image_space=sb.is_image_space,
                                get_action_dim=sb.get_action_dim)
    common.logger = _mod("stable_baselines3.common.logger", Logger=sb.Logger, KVWriter=sb.KVWriter,
                         HumanOutputFormat=sb.HumanOutputFormat, make_output_format=sb.make_output_format)
    common.callbacks = _mod("stable_baselines3.common.callbacks", BaseCallback=sb.BaseCallback,
                            CallbackList=sb.CallbackList)
    common.monitor = _mod("stable_baselines3.common.monitor", Monitor=_Monitor)
    common.utils = _mod("stable_baselines3.common.utils", check_for_correct_spaces=sb.check_for_correct_spaces,
                        set_random_seed=sb.set_random_seed, obs_as_tensor=sb.obs_as_tensor,
                        get_device=sb.get_device)
    common.torch_layers = _mod("stable_baselines3.common.torch_layers", FlattenExtractor=sb.FlattenExtractor,
                               BaseFeaturesExtractor=sb.BaseFeaturesExtractor, MlpExtractor=sb.MlpExtractor,
                               CombinedExtractor=sb.CombinedExtractor, NatureCNN=sb.NatureCNN)
    sac = _mod("stable_baselines3.sac")
    sac.policies = _mod("stable_baselines3.sac.policies", SACPolicy=sb.SACPolicy)
    ppo = _mod("stable_baselines3.ppo", PPO=sb.PPO)
    _mod("stable_baselines3", common=common, sac=sac, ppo=ppo, PPO=sb.PPO, _IS_ORACLE_SHIM=True)

    try:
        import torch.utils.tensorboard  # noqa: F401
    except Exception:  # tensorboard is not installed; `common.py:9` imports it at module top
        import torch.utils

        class _SummaryWriter:
            def __init__(self, *a, **k):
                pass

            def add_histogram(self, *a, **k):
                pass

        torch.utils.tensorboard = _mod("torch.utils.tensorboard", SummaryWriter=_SummaryWriter)

    try:
        import jsonpickle  # noqa: F401
    except Exception:
        # `data/huggingface_utils.py:5` imports jsonpickle (absent here) to encode per-step info dicts.
        # For dicts of JSON types its text equals json's, which is all the serialisation tests write.
        import json

        _mod("jsonpickle", encode=lambda o, **k: json.dumps(o), decode=lambda s, **k: json.loads(s))

    if REFERENCE_SRC not in sys.path:
        sys.path.insert(0, REFERENCE_SRC)
