"""Trajectory collection with the device policy (`data/rollout.py`, SURVEY 8f "next" row 1).

Same surface as the reference -- `make_min_episodes`, `make_min_timesteps`, `make_sample_until`,
`policy_to_callable`, `generate_trajectories`, `rollout_stats`, `flatten_trajectories_with_rew`,
`generate_transitions`, `rollout`, `discounted_sum` -- with two differences in mechanism:

* actions of a `PPO` / `ActorCriticPolicy` come from the HIP policy kernel through `.predict()`
  (eval mode: RunningNorm statistics are read, not updated; host noise draw in SB3's order);
* steps are logged as whole `[n_envs, ...]` arrays in a growing time-major buffer and each
  trajectory is cut out as one slice when its episode ends, instead of one dict per env per step
  in a `TrajectoryAccumulator` (`data/rollout.py:57-190`).

Emission order is the reference's: within a step, finished envs in ascending index; the list is
shuffled with `rng.shuffle` at the end (`data/rollout.py:471-476`).
"""
from __future__ import annotations

import dataclasses
import logging
from typing import Any, Callable, Dict, List, Mapping, Optional, Sequence, Union

import numpy as np

from imitation_amd import data_types as dt
from imitation_amd.policies import ActorCriticPolicy
from imitation_amd.ppo import OnPolicyAlgorithm
from imitation_amd.vec_env import VecEnv

GenTrajTerminationFn = Callable[[Sequence[dt.TrajectoryWithRew]], bool]


def make_min_episodes(n: int) -> GenTrajTerminationFn:
    """`data/rollout.py:193-204`."""
    assert n >= 1
    return lambda trajectories: len(trajectories) >= n


def make_min_timesteps(n: int) -> GenTrajTerminationFn:
    """`data/rollout.py:207-223`."""
    assert n >= 1

    def f(trajectories):
        return sum(len(t.obs) - 1 for t in trajectories) >= n

    return f


def make_sample_until(min_timesteps: Optional[int] = None, min_episodes: Optional[int] = None) -> GenTrajTerminationFn:
    """`data/rollout.py:226-272`: all given conditions must hold."""
    if min_timesteps is None and min_episodes is None:
        raise ValueError("At least one of min_timesteps and min_episodes needs to be non-None")
    conditions = []
    if min_timesteps is not None:
        if min_timesteps <= 0:
            raise ValueError(f"min_timesteps={min_timesteps} if provided must be positive")
        conditions.append(make_min_timesteps(min_timesteps))
    if min_episodes is not None:
        if min_episodes <= 0:
            raise ValueError(f"min_episodes={min_episodes} if provided must be positive")
        conditions.append(make_min_episodes(min_episodes))

    def sample_until(trajs):
        return all(cond(trajs) for cond in conditions)

    return sample_until


def policy_to_callable(policy, venv: VecEnv, deterministic_policy: bool = False):
    """`data/rollout.py:288-379`: None -> `action_space.sample()` per env; algorithm / policy ->
    `.predict()` (the HIP kernel); any other callable is used as is."""
    if policy is None:
        def get_actions(observations, states, episode_starts):
            acts = [venv.action_space.sample() for _ in range(len(observations))]
            return np.stack(acts, axis=0), None
    elif isinstance(policy, (OnPolicyAlgorithm, ActorCriticPolicy)):
        def get_actions(observations, states, episode_starts):
            return policy.predict(observations, state=states, episode_start=episode_starts,
                                  deterministic=deterministic_policy)
    elif callable(policy):
        if deterministic_policy:
            raise ValueError("Cannot set deterministic_policy=True when policy is a callable, "
                             "since deterministic_policy argument is ignored.")
        get_actions = policy
    else:
        raise TypeError("Policy must be None, a stable-baselines policy or algorithm, "
                        f"or a Callable, got {type(policy)} instead")
    if isinstance(policy, OnPolicyAlgorithm):  # [SB3 check_for_correct_spaces]
        if venv.observation_space != policy.observation_space:
            raise ValueError(f"Observation spaces do not match: {venv.observation_space} != {policy.observation_space}")
        if venv.action_space != policy.action_space:
            raise ValueError(f"Action spaces do not match: {venv.action_space} != {policy.action_space}")
    return get_actions


class _StepLog:
    """Time-major log of vectorised steps: row t holds, per env, the action taken at t, the
    observation it produced (terminal observation if the episode ended there) and the reward."""

    def __init__(self, n_envs: int):
        self.n = n_envs
        self.t = 0
        self._cap = 0
        self.acts = self.nxt = self.rews = None
        self.infos: List[Optional[list]] = []

    def _grow(self, acts, nxt, rews):
        cap = max(64, 2 * self._cap)

        def alloc(old, like):
            new = np.empty((cap,) + like.shape, dtype=like.dtype)
            if old is not None:
                new[:self.t] = old[:self.t]
            return new

        self.acts, self.nxt, self.rews = alloc(self.acts, acts), alloc(self.nxt, nxt), alloc(self.rews, rews)
        self._cap = cap

    def append(self, acts, nxt, rews, infos) -> None:
        if self.t == self._cap:
            self._grow(np.asarray(acts), np.asarray(nxt), np.asarray(rews))
        self.acts[self.t], self.nxt[self.t], self.rews[self.t] = acts, nxt, rews
        self.infos.append(infos)
        self.t += 1

    def cut(self, first_obs, s: int, e: int, env: int, terminal: bool) -> dt.TrajectoryWithRew:
        """Steps s..e (inclusive) of `env` as one trajectory starting from `first_obs`."""
        obs = np.concatenate([first_obs[None], self.nxt[s:e + 1, env]])
        # per trajectory: Monitor-style wrappers report content ('episode', 'rollout') on the done step only, so a
        # step's list may be missing (nothing beyond what the trajectory already encodes) -- those steps get {}
        infos = None
        if any(self.infos[t] is not None for t in range(s, e + 1)):
            infos = np.array([self.infos[t][env] if self.infos[t] is not None else {} for t in range(s, e + 1)])
        return dt.TrajectoryWithRew(obs=obs, acts=self.acts[s:e + 1, env].copy(),
                                    rews=self.rews[s:e + 1, env].astype(np.float64 if not np.issubdtype(
                                        self.rews.dtype, np.floating) else self.rews.dtype), infos=infos,
                                    terminal=terminal)


def generate_trajectories(policy, venv: VecEnv, sample_until: GenTrajTerminationFn, rng: np.random.Generator, *,
                          deterministic_policy: bool = False) -> Sequence[dt.TrajectoryWithRew]:
    """`data/rollout.py:382-506`. Every env keeps stepping until `sample_until` holds and its own
    current episode has ended (the reference's guard against a bias towards short episodes);
    episodes of envs that were retired are dropped."""
    get_actions = policy_to_callable(policy, venv, deterministic_policy)
    trajectories: List[dt.TrajectoryWithRew] = []
    obs = venv.reset()
    assert isinstance(obs, np.ndarray), "Dict / tuple observations are not supported on this path."
    n = venv.num_envs
    log = _StepLog(n)
    first_obs = np.array(obs, copy=True)          # first observation of each env's running episode
    start = np.zeros(n, dtype=np.int64)           # log row where it started
    active = np.ones(n, dtype=bool)
    state = None
    dones = np.zeros(n, dtype=bool)
    while np.any(active):
        acts, state = get_actions(obs, state, dones)
        obs, rews, dones, infos = venv.step(acts)
        assert isinstance(obs, np.ndarray), "Dict / tuple observations are not supported on this path."
        dones = np.asarray(dones, dtype=bool) & active
        nxt = obs
        if dones.any():
            nxt = np.array(obs, copy=True)
            for i in np.flatnonzero(dones):       # `data/rollout.py:171-177`: real terminal observation
                nxt[i] = infos[i]["terminal_observation"]
        log.append(acts, nxt, rews, infos if _has_content(infos) else None)
        t = log.t - 1
        for i in np.flatnonzero(dones):
            trajectories.append(log.cut(first_obs[i], int(start[i]), t, int(i), terminal=True))
            first_obs[i] = obs[i]
            start[i] = t + 1
        if sample_until(trajectories):
            active &= ~dones
    rng.shuffle(trajectories)
    obs_shape, act_shape = venv.observation_space.shape, venv.action_space.shape
    for traj in trajectories:
        k = len(traj.acts)
        assert traj.obs.shape == (k + 1,) + obs_shape, f"expected shape {(k + 1,) + obs_shape}, got {traj.obs.shape}"
        assert traj.acts.shape == (k,) + act_shape, f"expected shape {(k,) + act_shape}, got {traj.acts.shape}"
        assert traj.rews.shape == (k,), f"expected shape {(k,)}, got {traj.rews.shape}"
    return trajectories


def _has_content(infos) -> bool:
    """Array envs report only terminal_observation / TimeLimit.truncated, which the trajectory
    already encodes; keep `infos` only when an env reports something else (e.g. Monitor data)."""
    for info in infos:
        for k in info:
            if k not in ("terminal_observation", "TimeLimit.truncated"):
                return True
    return False


def rollout_stats(trajectories: Sequence[dt.TrajectoryWithRew]) -> Mapping[str, float]:
    """`data/rollout.py:509-560`: `n_traj`, `{monitor_,}return_*` and `len_*` min/mean/std/max."""
    assert len(trajectories) > 0
    out: Dict[str, float] = {"n_traj": len(trajectories)}
    desc = {"return": np.asarray([sum(t.rews) for t in trajectories]),
            "len": np.asarray([len(t.rews) for t in trajectories])}
    monitor = []
    for t in trajectories:
        if t.infos is not None:
            r = t.infos[-1].get("episode", {}).get("r")
            if r is not None:
                monitor.append(r)
    if monitor:
        desc["monitor_return"] = np.asarray(monitor)
        out["monitor_return_len"] = len(monitor)
    for name, vals in desc.items():
        for stat in ("min", "mean", "std", "max"):
            out[f"{name}_{stat}"] = getattr(np, stat)(vals).item()
    return out


def flatten_trajectories_with_rew(trajectories: Sequence[dt.TrajectoryWithRew]) -> dt.TransitionsWithRew:
    """`data/rollout.py:613-621`."""
    return dt.flatten_trajectories(trajectories)


flatten_trajectories = flatten_trajectories_with_rew


def generate_transitions(policy, venv: VecEnv, n_timesteps: int, rng: np.random.Generator, *, truncate: bool = True,
                         **kwargs: Any) -> dt.TransitionsWithRew:
    """`data/rollout.py:624-665`."""
    trajs = generate_trajectories(policy, venv, sample_until=make_min_timesteps(n_timesteps), rng=rng, **kwargs)
    trans = flatten_trajectories_with_rew(trajs)
    if truncate and n_timesteps is not None:
        fields = {f.name: getattr(trans, f.name) for f in dataclasses.fields(trans)}
        trans = dt.TransitionsWithRew(**{k: (v[:n_timesteps] if v is not None else None) for k, v in fields.items()})
    return trans


def unwrap_traj(traj: dt.TrajectoryWithRew) -> dt.TrajectoryWithRew:
    """`data/rollout.py:30-54`: original obs / rews recorded by a `RolloutInfoWrapper`."""
    ep_info = traj.infos[-1]["rollout"]
    res = dataclasses.replace(traj, obs=ep_info["obs"], rews=ep_info["rews"])
    assert len(res.obs) == len(res.acts) + 1 and len(res.rews) == len(res.acts)
    return res


def rollout(policy, venv: VecEnv, sample_until: GenTrajTerminationFn, rng: np.random.Generator, *,
            unwrap: bool = True, exclude_infos: bool = True, verbose: bool = True,
            **kwargs: Any) -> Sequence[dt.TrajectoryWithRew]:
    """`data/rollout.py:668-725`. Trajectories without infos (array envs) have nothing to unwrap."""
    trajs = generate_trajectories(policy, venv, sample_until, rng=rng, **kwargs)
    if unwrap:
        trajs = [unwrap_traj(t) if t.infos is not None else t for t in trajs]
    if exclude_infos:
        trajs = [dataclasses.replace(t, infos=None) for t in trajs]
    if verbose:
        logging.info(f"Rollout stats: {rollout_stats(trajs)}")
    return trajs


def discounted_sum(arr: np.ndarray, gamma: float) -> Union[np.ndarray, float]:
    """`data/rollout.py:728-756`: sum_t gamma^t arr[t] over the FIRST axis (Horner form)."""
    assert arr.ndim in (1, 2)
    if gamma == 1.0:
        return arr.sum(axis=0)
    return np.polynomial.polynomial.polyval(gamma, arr)
