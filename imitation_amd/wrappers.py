"""VecEnv wrappers of the adversarial trainer with array (not per-env Python) bookkeeping.

`BufferingWrapper` (`data/wrappers.py:13-169`) and `RewardVecEnvWrapper`
(`rewards/reward_wrapper.py:40-133`) keep their reference surface -- `reset`, `step_async`,
`step_wait`, `pop_trajectories`, `pop_transitions`, `n_transitions`, `episode_rewards`,
`make_log_callback` -- but record whole `[n_envs, ...]` step arrays and derive trajectory
order with `data_types.segment_order`, instead of appending one dict per env per step
(the reference's dominant CPU cost at n_envs=1024, SURVEY 8a "where the time goes").

The fused GPU collector (`ppo.PPO.collect_rollouts`) steps the innermost env itself and feeds
these wrappers through `record_step` / `record_rewards`, which is state-equivalent to calling
`step_wait` on them once per env step.
"""
from __future__ import annotations

import collections
from typing import Any, Callable, Deque, Dict, List, Optional, Sequence, Tuple

import numpy as np

from imitation_amd import data_types as dt
from imitation_amd.vec_env import ArrayVecEnv, VecEnv, VecEnvWrapper


def step_arrays(venv: VecEnv):
    """`(obs, rews, dones, next_obs_fixed, truncated, infos_or_None)` of a finished step."""
    if isinstance(venv, ArrayVecEnv):
        obs, rews, dones, nxt, trunc = venv.step_wait_arrays()
        return obs, rews, dones, nxt, trunc, None
    obs, rews, dones, infos = venv.step_wait()
    dones = np.asarray(dones, dtype=bool)
    nxt = np.array(obs, copy=True)
    trunc = np.zeros(len(dones), dtype=bool)
    ended = np.flatnonzero(dones)    # `reward_wrapper.py:98-109`, SB3 collect_rollouts timeout rule: ended envs only
    if len(ended):
        nxt[ended] = np.stack([infos[i]["terminal_observation"] for i in ended]).reshape(len(ended), *nxt.shape[1:])
        trunc[ended] = [bool(infos[i].get("TimeLimit.truncated", False)) for i in ended]
    return obs, rews, dones, nxt, trunc, infos


def infos_from_arrays(dones, nxt, trunc) -> List[Dict[str, Any]]:
    infos: List[Dict[str, Any]] = [{} for _ in range(len(dones))]
    for i in np.flatnonzero(dones):
        infos[i]["terminal_observation"] = nxt[i].copy()
        infos[i]["TimeLimit.truncated"] = bool(trunc[i])
    return infos


class BufferingWrapper(VecEnvWrapper):
    """Saves transitions of the underlying VecEnv; `pop_*` return them in the reference's order."""

    def __init__(self, venv: VecEnv, error_on_premature_reset: bool = True):
        super().__init__(venv)
        self.error_on_premature_reset = error_on_premature_reset
        self._init_reset = False
        self._saved_acts = None
        self._last_obs: Optional[np.ndarray] = None
        self._timesteps: Optional[np.ndarray] = None
        self.n_transitions: Optional[int] = None
        self._steps: List[Tuple[np.ndarray, ...]] = []   # (obs_before, acts, next_fixed, rews, dones)
        # per step: the env's info dicts when they carry more than what the arrays already encode
        # (terminal_observation / TimeLimit.truncated), else None -- array envs never pay for dict lists
        self._infos: List[Optional[list]] = []
        self.last_infos: Optional[np.ndarray] = None     # infos of the last pop, in its emission order (or None)
        self._ep_lens: List[int] = []

    def reset(self, **kwargs):
        if self._init_reset and self.error_on_premature_reset and self.n_transitions > 0:
            raise RuntimeError("BufferingWrapper reset() before samples were accessed")
        self._init_reset = True
        self.n_transitions = 0
        obs = self.venv.reset(**kwargs)
        self._last_obs = obs
        self._steps, self._infos = [], []
        self._timesteps = np.zeros((len(obs),), dtype=int)
        return obs

    def step_async(self, actions):
        assert self._init_reset
        assert self._saved_acts is None
        self.venv.step_async(actions)
        self._saved_acts = actions

    def step_wait(self):
        assert self._init_reset
        assert self._saved_acts is not None
        acts, self._saved_acts = self._saved_acts, None
        obs, rews, dones, nxt, trunc, infos = step_arrays(self.venv)
        self.record_step(acts, obs, nxt, rews, dones, infos)
        if infos is None:
            infos = infos_from_arrays(dones, nxt, trunc)
        return obs, rews, dones, infos

    def record_step(self, acts, new_obs, next_fixed, rews, dones, infos=None) -> None:
        """State update of one `step_wait` given the step's arrays (`wrappers.py:69-91`)."""
        dones = np.asarray(dones, dtype=bool)
        self._steps.append((self._last_obs, np.array(acts, copy=True), next_fixed, np.asarray(rews), dones))
        # the env's own dicts, verbatim, whenever it produced any (`data/wrappers.py:69-91` keeps every step's info);
        # array envs (infos None) never pay for dict lists
        # (a shallow copy of the LIST: an env may refill the same list object every step)
        self._infos.append(list(infos) if infos is not None else None)
        self._last_obs = new_obs
        self.n_transitions += self.num_envs
        self._timesteps += 1
        if dones.any():
            self._ep_lens += list(self._timesteps[dones])
            self._timesteps[dones] = 0

    def _stacked(self):
        obs, acts, nxt, rews, dones = (np.stack(x) for x in zip(*self._steps))
        return obs, acts, nxt, rews, dones

    def _infos_in_order(self, order: np.ndarray, n: int) -> Optional[np.ndarray]:
        """`infos` of the recorded steps in emission order (time-major offsets `t*n + env`): the env's own dicts
        (`data/wrappers.py:69-91` keeps every step's dict verbatim); None for array envs, which produce none."""
        if not any(i is not None for i in self._infos):
            return None
        # one object row per step (`np.fromiter`: no shape discovery inside the dicts) and ONE fancy index for the whole
        # round, instead of a Python loop over its 16 384 entries
        tile = np.empty((len(self._infos), n), dtype=object)
        for t, row in enumerate(self._infos):
            # (steps recorded without dicts between steps with them: an empty dict each, as before)
            it = row if row is not None else ({} for _ in range(n))
            try:
                tile[t] = np.fromiter(it, dtype=object, count=n)
            except (TypeError, ValueError):   # NumPy < 1.23: no object dtype in `fromiter`
                for j, d in enumerate(row if row is not None else ({} for _ in range(n))):
                    tile[t, j] = d
        return tile.reshape(-1)[order]

    def pop_transitions_and_lens(self) -> Tuple[Optional[dt.TransitionsWithRew], List[int]]:
        """Fast path of `pop_trajectories` + `flatten_trajectories_with_rew`
        (`adversarial/common.py:422-424`): the same rows in the same order, built with one gather."""
        if self.n_transitions == 0:
            return None, []
        obs, acts, nxt, rews, dones = self._stacked()
        T, n = dones.shape
        order, _, _ = dt.segment_order(dones)
        flat = lambda a: a.reshape(T * n, *a.shape[2:])[order]
        self.last_infos = self._infos_in_order(order, n)
        trans = dt.TransitionsWithRew(obs=flat(obs), acts=flat(acts), next_obs=flat(nxt), dones=flat(dones),
                                      infos=self.last_infos,
                                      rews=flat(rews).astype(np.float64, copy=False)
                                      if not np.issubdtype(rews.dtype, np.floating) else flat(rews))
        lens, self._ep_lens = self._ep_lens, []
        self._steps, self._infos = [], []
        self.n_transitions = 0
        return trans, lens

    def pop_order_and_lens(self):
        """Device-friendly pop: `(order, ep_lens, T)` where `order` lists the time-major offsets
        `t*n_envs + env` of the recorded steps in the reference's emission order; the caller gathers
        the rows itself (the GPU collector already holds them in HBM). Clears the buffer."""
        if self.n_transitions == 0:
            return None, [], 0
        dones = np.stack([st[4] for st in self._steps])
        order, _, _ = dt.segment_order(dones)
        self.last_infos = self._infos_in_order(order, dones.shape[1])
        lens, self._ep_lens = self._ep_lens, []
        self._steps, self._infos = [], []
        self.n_transitions = 0
        return order, lens, dones.shape[0]

    def pop_transitions(self) -> dt.TransitionsWithRew:
        if self.n_transitions == 0:
            raise RuntimeError("Called pop_transitions on an empty BufferingWrapper")
        return self.pop_transitions_and_lens()[0]

    def pop_trajectories(self) -> Tuple[Sequence[dt.TrajectoryWithRew], Sequence[int]]:
        """`wrappers.py:132-148`: completed trajectories in completion order, then the
        in-progress fragments by env index (materialised only when somebody asks for them)."""
        if self.n_transitions == 0:
            return [], []
        obs, acts, nxt, rews, dones = self._stacked()
        T, n = dones.shape
        trajs: List[dt.TrajectoryWithRew] = []
        t_idx, e_idx = np.nonzero(dones)
        last = np.full(n, -1)
        have = any(i is not None for i in self._infos)

        def infos_of(s: int, t: int, e: int):   # steps s..t (inclusive) of env e
            if not have:
                return None
            return np.array([self._infos[k][e] if self._infos[k] is not None else {} for k in range(s, t + 1)],
                            dtype=object)

        for t, e in zip(t_idx, e_idx):
            s = last[e] + 1
            trajs.append(dt.TrajectoryWithRew(obs=np.concatenate([obs[s:t + 1, e], nxt[t:t + 1, e]]),
                                              acts=acts[s:t + 1, e], rews=rews[s:t + 1, e], infos=infos_of(s, t, e),
                                              terminal=True))
            last[e] = t
        for e in range(n):
            s = last[e] + 1
            if s <= T - 1:
                trajs.append(dt.TrajectoryWithRew(obs=np.concatenate([obs[s:T, e], nxt[T - 1:T, e]]),
                                                  acts=acts[s:T, e], rews=rews[s:T, e], infos=infos_of(s, T - 1, e),
                                                  terminal=False))
        lens, self._ep_lens = self._ep_lens, []
        self._steps, self._infos = [], []
        self.n_transitions = 0
        return trajs, lens

    pop_finished_trajectories = pop_trajectories


class WrappedRewardCallback:
    """`rewards/reward_wrapper.py:15-37`: logs the mean wrapped episode reward at rollout start."""

    def __init__(self, episode_rewards: Deque[float]):
        self.episode_rewards = episode_rewards
        self.model = None

    def init_callback(self, model) -> None:
        self.model = model

    def on_training_start(self, locals_, globals_) -> None:
        pass

    def on_rollout_start(self) -> None:
        if len(self.episode_rewards) == 0:
            return
        mean = sum(self.episode_rewards) / len(self.episode_rewards)
        self.model.logger.record("rollout/ep_rew_wrapped_mean", mean)

    def on_step(self) -> bool:
        return True

    def on_rollout_end(self) -> None:
        pass

    def on_training_end(self) -> None:
        pass

    def update_locals(self, locals_) -> None:
        pass


class RewardVecEnvWrapper(VecEnvWrapper):
    """Replaces the env reward by `reward_fn(old_obs, act, next_obs_fixed, dones)` every step."""

    def __init__(self, venv: VecEnv, reward_fn: Callable, ep_history: int = 100):
        assert not isinstance(venv, RewardVecEnvWrapper)
        super().__init__(venv)
        self.episode_rewards: Deque = collections.deque(maxlen=ep_history)
        self._cumulative_rew = np.zeros((venv.num_envs,))
        self.reward_fn = reward_fn
        self._old_obs = None
        self._actions = None
        self.reset()

    def make_log_callback(self) -> WrappedRewardCallback:
        return WrappedRewardCallback(self.episode_rewards)

    @property
    def envs(self):
        return self.venv.envs

    def reset(self):
        self._old_obs = self.venv.reset()
        return self._old_obs

    def step_async(self, actions):
        self._actions = actions
        return self.venv.step_async(actions)

    def step_wait(self):
        obs, old_rews, dones, infos = self.venv.step_wait()
        dones = np.asarray(dones, dtype=bool)
        fixed = np.array(obs, copy=True)
        for i in np.flatnonzero(dones):
            fixed[i] = infos[i]["terminal_observation"]
        rews = self.reward_fn(self._old_obs, self._actions, fixed, np.array(dones))
        assert len(rews) == len(obs), "must return one rew for each env"
        self.record_rewards(rews[None], dones[None], obs)
        for info, r in zip(infos, old_rews):
            info["original_env_rew"] = r
        return obs, rews, dones, infos

    def record_rewards(self, rews: np.ndarray, dones: np.ndarray, last_obs) -> None:
        """Episode-return bookkeeping of `step_wait` (`reward_wrapper.py:117-130`) for `[T, n]`
        arrays of wrapped rewards and dones."""
        for t in range(rews.shape[0]):
            self._cumulative_rew += rews[t]
            d = dones[t]
            if d.any():
                self.episode_rewards.extend(self._cumulative_rew[d].tolist())
                self._cumulative_rew[d] = 0
        self._old_obs = last_obs
