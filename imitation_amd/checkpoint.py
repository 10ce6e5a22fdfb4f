"""Checkpoints of an adversarial trainer.

Two things, kept apart on purpose:

1. `save(trainer, save_path)` -- the **reference's artefact layout**
   (`scripts/train_adversarial.py:25-35`, `policies/serialize.py:182-201`):

       save_path/reward_train.pt      save_path/reward_test.pt      save_path/gen_policy/model.zip

   `model.zip` carries SB3's member names: `policy.pth` (state dict under SB3's parameter names, so
   `sb3_policy.load_state_dict(torch.load(...))` works and vice versa), `policy.optimizer.pth`
   (torch-Adam state dict: one `{step, exp_avg, exp_avg_sq}` entry per parameter tensor in
   `parameters()` order, the fixture's `param_groups` fields), `pytorch_variables.pth`,
   `_stable_baselines3_version`, `system_info.txt` and a JSON `data` with the hyper-parameters under
   SB3's field names. SB3 stores classes in `data` as cloudpickles of its own classes; those cannot be
   produced here, so classes are recorded by dotted name and `load_policy_zip` only needs the tensors.
   The `.pt` files hold `{"class": name, "state_dict": ...}` with the reference's state-dict keys (the
   reference pickles the module object itself, which would need its classes to unpickle).

2. `save_checkpoint / load_checkpoint` -- **everything needed to resume bit-exactly**: both
   networks, both Adam states, RunningNorm buffers, replay ring, expert-stream position, counters,
   wrapper bookkeeping, the torch and NumPy global generators, and the env's own state when it offers
   `get_state()/set_state()`. (`tests/test_checkpoint_gpu.py`: train, save, continue == restore into
   a differently initialised trainer, continue.)
"""
from __future__ import annotations

import io
import json
import os
import platform
import zipfile
from typing import Any, Dict

import numpy as np
import torch as th

from imitation_amd import reward_nets
from imitation_amd.ppo import PPO

SB3_FORMAT_VERSION = "2.2.1"   # the release line the reference pins (`setup.py:206`)


def _cpu(sd: Dict[str, th.Tensor]) -> Dict[str, th.Tensor]:
    return {k: v.detach().cpu().clone() for k, v in sd.items()}


# ------------------------------------------------------------------ reference artefact layout


def _torch_adam_state(policy) -> Dict[str, Any]:
    """The flat Adam buffers sliced per parameter tensor, as `torch.optim.Adam.state_dict()`."""
    opt = policy.optimizer
    g = opt.param_groups[0]
    state, o = {}, 0
    names = [n for n, _ in policy.named_parameters()]
    for i, (_, p) in enumerate(policy.named_parameters()):
        n = p.numel()
        state[i] = {"step": th.tensor(float(opt.step_count)),
                    "exp_avg": opt.exp_avg[o:o + n].view(p.shape).cpu().clone(),
                    "exp_avg_sq": opt.exp_avg_sq[o:o + n].view(p.shape).cpu().clone()}
        o += n
    group = {"lr": g["lr"], "betas": tuple(g["betas"]), "eps": g["eps"], "weight_decay": g.get("weight_decay", 0),
             "amsgrad": False, "maximize": False, "foreach": None, "capturable": False, "differentiable": False,
             "fused": None, "params": list(range(len(names)))}
    return {"state": state if opt.step_count else {}, "param_groups": [group]}


def _load_torch_adam_state(policy, sd: Dict[str, Any]) -> None:
    opt = policy.optimizer
    st = sd.get("state", {})
    o, step = 0, 0
    for i, (_, p) in enumerate(policy.named_parameters()):
        n = p.numel()
        if i in st:
            opt.exp_avg[o:o + n].copy_(th.as_tensor(st[i]["exp_avg"]).reshape(-1))
            opt.exp_avg_sq[o:o + n].copy_(th.as_tensor(st[i]["exp_avg_sq"]).reshape(-1))
            step = int(float(st[i]["step"]))
        o += n
    opt.step_count = step
    for k in ("lr", "betas", "eps"):
        if sd.get("param_groups"):
            opt.param_groups[0][k] = sd["param_groups"][0][k]


def _save_tensor_file(z: zipfile.ZipFile, name: str, obj) -> None:
    buf = io.BytesIO()
    th.save(obj, buf)
    z.writestr(name, buf.getvalue())


def save_policy_zip(path, algo: PPO) -> None:
    """`model.save(path)` of SB3 [base_class.BaseAlgorithm.save -> save_util.save_to_zip_file]."""
    pol = algo.policy
    data = {
        "policy_class": {":type:": "class", ":name:": f"{type(pol).__module__}.{type(pol).__qualname__}"},
        "verbose": algo.verbose, "policy_kwargs": {k: str(v) for k, v in (algo.policy_kwargs or {}).items()},
        "num_timesteps": algo.num_timesteps, "_total_timesteps": getattr(algo, "_total_timesteps", 0),
        "_num_timesteps_at_start": getattr(algo, "_num_timesteps_at_start", 0), "seed": algo.seed,
        "learning_rate": algo.learning_rate if isinstance(algo.learning_rate, (int, float)) else str(algo.learning_rate),
        "_current_progress_remaining": algo._current_progress_remaining, "_n_updates": algo._n_updates,
        "observation_space": repr(algo.observation_space), "action_space": repr(algo.action_space),
        "n_envs": algo.n_envs, "n_steps": algo.n_steps, "gamma": algo.gamma, "gae_lambda": algo.gae_lambda,
        "ent_coef": algo.ent_coef, "vf_coef": algo.vf_coef, "max_grad_norm": algo.max_grad_norm,
        "batch_size": algo.batch_size, "n_epochs": algo.n_epochs, "normalize_advantage": algo.normalize_advantage,
        "target_kl": None, "clip_range_vf": None, "use_sde": False,
    }
    os.makedirs(os.path.dirname(os.path.abspath(str(path))), exist_ok=True)
    with zipfile.ZipFile(str(path), "w") as z:
        z.writestr("data", json.dumps(data, indent=1, default=str))
        _save_tensor_file(z, "pytorch_variables.pth", {})
        _save_tensor_file(z, "policy.pth", _cpu(pol.state_dict()))
        _save_tensor_file(z, "policy.optimizer.pth", _torch_adam_state(pol))
        z.writestr("_stable_baselines3_version", SB3_FORMAT_VERSION)
        z.writestr("system_info.txt", f"- OS: {platform.platform()}\n- Python: {platform.python_version()}\n"
                                      f"- PyTorch: {th.__version__}\n- Numpy: {np.__version__}\n"
                                      "- Writer: imitation_amd (SB3 zip member layout)\n")


def load_policy_zip(path, algo_or_policy, load_optimizer: bool = True) -> Dict[str, Any]:
    """Reads `policy.pth` (+ `policy.optimizer.pth`) of an SB3-layout zip -- one written here or by SB3
    itself (e.g. the reference's expert fixtures) -- into a policy of matching architecture.
    Returns the parsed `data` member."""
    pol = getattr(algo_or_policy, "policy", algo_or_policy)
    with zipfile.ZipFile(str(path)) as z:
        sd = th.load(io.BytesIO(z.read("policy.pth")), map_location="cpu", weights_only=False)
        pol.load_state_dict(sd)
        if load_optimizer and "policy.optimizer.pth" in z.namelist():
            _load_torch_adam_state(pol, th.load(io.BytesIO(z.read("policy.optimizer.pth")), map_location="cpu",
                                                weights_only=False))
        data = json.loads(z.read("data"))
    if isinstance(algo_or_policy, PPO):
        algo_or_policy.num_timesteps = int(data.get("num_timesteps", algo_or_policy.num_timesteps))
        algo_or_policy._n_updates = int(data.get("_n_updates", algo_or_policy._n_updates))
    return data


def save_reward_net(path, net) -> None:
    if isinstance(net, th.nn.Module):
        # an `nn.Module` reward net (imitation_amd.modules) is written the way the reference writes its own:
        # `th.save(module, path)` (`scripts/train_adversarial.py:25-35`), readable by `th.load` + `.predict*`
        th.save(net, str(path))
        return
    chain, n = [], net
    while n is not None:
        chain.append(type(n).__name__)
        n = getattr(n, "base", None)
    th.save({"class": chain[0], "wrapper_chain": chain, "state_dict": _cpu(net.state_dict())}, str(path))


def load_reward_net(path, net) -> None:
    blob = th.load(str(path), map_location="cpu", weights_only=False)
    if isinstance(blob, th.nn.Module):
        blob = blob.state_dict()
    net.load_state_dict(blob["state_dict"] if isinstance(blob, dict) and "state_dict" in blob else blob)


def save(trainer, save_path) -> None:
    """`scripts/train_adversarial.py:25-35`: discriminator and generator artefacts."""
    save_path = str(save_path)
    os.makedirs(save_path, exist_ok=True)
    save_reward_net(os.path.join(save_path, "reward_train.pt"), trainer.reward_train)
    save_reward_net(os.path.join(save_path, "reward_test.pt"), trainer.reward_test)
    save_policy_zip(os.path.join(save_path, "gen_policy", "model.zip"), trainer.gen_algo)


# ------------------------------------------------------------------------ resumable checkpoint


def _env_state(venv):
    base = venv
    while hasattr(base, "venv"):
        base = base.venv
    return base.get_state() if hasattr(base, "get_state") else None


def _set_env_state(venv, state) -> None:
    base = venv
    while hasattr(base, "venv"):
        base = base.venv
    if state is not None:
        base.set_state(state)


def _cpu_nested(x):
    if isinstance(x, th.Tensor):
        return x.detach().cpu()
    if isinstance(x, dict):
        return {k: _cpu_nested(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return type(x)(_cpu_nested(v) for v in x)
    return x


def save_checkpoint(trainer, path) -> None:
    """Complete training state at a round boundary (call between `train()` calls)."""
    algo = trainer.gen_algo
    if trainer.venv_buffering.n_transitions:
        raise RuntimeError("checkpoint only at a round boundary (the buffering wrapper still holds transitions)")
    if getattr(trainer, "_pre_expert_rows", None) or getattr(trainer, "_gp_round_pre", None) is not None:
        # (draws taken ahead for a round that did not run to its end -- `AdversarialTrainer._round_predraw`: the generator
        #  is already behind them, a resumed run would draw them a second time)
        raise RuntimeError("checkpoint only at a round boundary (a round's index rows are drawn but not consumed)")
    th.cuda.synchronize()
    net, opt, ring = trainer._reward_net, trainer._disc_opt, trainer._gen_replay_buffer
    es = getattr(trainer, "_expert_stream", None)
    blob = {
        "format": 1,
        "reward_net": _cpu(net.state_dict()),
        # the fused flat-buffer Adam keeps (step, exp_avg, exp_avg_sq); any torch optimiser (a `disc_opt_cls` other than
        # Adam, or the optimiser of an `nn.Module` reward net) goes through its own state_dict
        "disc_opt": ({"step": opt.step_count, "exp_avg": opt.exp_avg.cpu(), "exp_avg_sq": opt.exp_avg_sq.cpu()}
                     if hasattr(opt, "step_count") else {"torch_state_dict": _cpu_nested(opt.state_dict())}),
        "policy": _cpu(algo.policy.state_dict()),
        "policy_opt": {"step": algo.policy.optimizer.step_count, "exp_avg": algo.policy.optimizer.exp_avg.cpu(),
                       "exp_avg_sq": algo.policy.optimizer.exp_avg_sq.cpu()},
        "ppo": {"num_timesteps": algo.num_timesteps, "_n_updates": algo._n_updates,
                "_current_progress_remaining": algo._current_progress_remaining,
                "_last_obs": None if algo._last_obs is None else np.array(algo._last_obs),
                "_last_episode_starts": None if algo._last_episode_starts is None else np.array(algo._last_episode_starts),
                "_total_timesteps": getattr(algo, "_total_timesteps", 0),
                "_num_timesteps_at_start": getattr(algo, "_num_timesteps_at_start", 0),
                "ep_info_buffer": list(getattr(algo, "ep_info_buffer", []) or [])},
        "ring": {"obs": ring._obs.cpu(), "acts": ring._acts.cpu(), "next": ring._next.cpu(), "dones": ring._dones.cpu(),
                 "_idx": ring._idx, "_n_data": ring._n_data,
                 # host ring of info dicts at the same positions (None: no stored transition carried any)
                 "infos": None if ring._infos is None else list(ring._infos)},
        "expert_stream": None if es is None else {"perm": None if es._perm is None else es._perm.copy(), "pos": es._pos},
        "counters": {"_global_step": trainer._global_step, "_disc_step": trainer._disc_step},
        "buffering": {"_last_obs": np.array(trainer.venv_buffering._last_obs),
                      "_timesteps": trainer.venv_buffering._timesteps.copy(),
                      "_ep_lens": list(trainer.venv_buffering._ep_lens),
                      "_init_reset": trainer.venv_buffering._init_reset},
        "reward_wrapper": None if not hasattr(trainer.venv_wrapped, "episode_rewards") else {
            "episode_rewards": list(trainer.venv_wrapped.episode_rewards),
            "_cumulative_rew": trainer.venv_wrapped._cumulative_rew.copy(),
            "_old_obs": np.array(trainer.venv_wrapped._old_obs)},
        "rng": {"torch": th.get_rng_state(), "numpy": np.random.get_state()},
        "env": _env_state(trainer.venv),
    }
    os.makedirs(os.path.dirname(os.path.abspath(str(path))), exist_ok=True)
    th.save(blob, str(path))


def load_checkpoint(trainer, path) -> None:
    """Restores a `save_checkpoint` file into a trainer built with the same configuration."""
    blob = th.load(str(path), map_location="cpu", weights_only=False)
    assert blob.get("format") == 1, "unknown checkpoint format"
    algo = trainer.gen_algo
    dev = trainer._device
    trainer._reward_net.load_state_dict(blob["reward_net"])
    o = trainer._disc_opt
    if "torch_state_dict" in blob["disc_opt"]:
        o.load_state_dict(blob["disc_opt"]["torch_state_dict"])  # torch moves the state to the parameters' device
    else:
        o.step_count = int(blob["disc_opt"]["step"])
        o.exp_avg.copy_(blob["disc_opt"]["exp_avg"].to(dev))
        o.exp_avg_sq.copy_(blob["disc_opt"]["exp_avg_sq"].to(dev))
    algo.policy.load_state_dict(blob["policy"])
    po = algo.policy.optimizer
    po.step_count = int(blob["policy_opt"]["step"])
    po.exp_avg.copy_(blob["policy_opt"]["exp_avg"].to(dev))
    po.exp_avg_sq.copy_(blob["policy_opt"]["exp_avg_sq"].to(dev))
    p = blob["ppo"]
    algo.num_timesteps, algo._n_updates = p["num_timesteps"], p["_n_updates"]
    algo._current_progress_remaining = p["_current_progress_remaining"]
    algo._last_obs, algo._last_episode_starts = p["_last_obs"], p["_last_episode_starts"]
    algo._total_timesteps, algo._num_timesteps_at_start = p["_total_timesteps"], p["_num_timesteps_at_start"]
    if getattr(algo, "ep_info_buffer", None) is not None:
        algo.ep_info_buffer.clear()
        algo.ep_info_buffer.extend(p["ep_info_buffer"])
    ring, r = trainer._gen_replay_buffer, blob["ring"]
    ring._obs.copy_(r["obs"].to(dev)); ring._acts.copy_(r["acts"].to(dev))
    ring._next.copy_(r["next"].to(dev)); ring._dones.copy_(r["dones"].to(dev))
    ring._idx, ring._n_data = int(r["_idx"]), int(r["_n_data"])
    infos = r.get("infos")
    ring._infos = None if infos is None else np.array(list(infos) + [{}] * (ring.capacity - len(infos)), dtype=object)
    es = getattr(trainer, "_expert_stream", None)
    if es is not None and blob["expert_stream"] is not None:
        es._perm, es._pos = blob["expert_stream"]["perm"], int(blob["expert_stream"]["pos"])
    trainer._global_step, trainer._disc_step = blob["counters"]["_global_step"], blob["counters"]["_disc_step"]
    b = blob["buffering"]
    bw = trainer.venv_buffering
    bw._last_obs, bw._timesteps, bw._ep_lens = b["_last_obs"], b["_timesteps"], list(b["_ep_lens"])
    bw._init_reset, bw._steps, bw.n_transitions = b["_init_reset"], [], 0
    w = blob["reward_wrapper"]
    if w is not None:
        rw = trainer.venv_wrapped
        rw.episode_rewards.clear(); rw.episode_rewards.extend(w["episode_rewards"])
        rw._cumulative_rew, rw._old_obs = w["_cumulative_rew"], w["_old_obs"]
    _set_env_state(trainer.venv, blob["env"])
    th.set_rng_state(blob["rng"]["torch"])
    np.random.set_state(blob["rng"]["numpy"])
    th.cuda.synchronize()
