"""Host-side vectorised environment protocol + synthetic environments.

Environment workers stay on host CPU (BASELINE.json north_star). This module provides
the SB3 `VecEnv` surface the reference's trainer relies on:
`num_envs, observation_space, action_space, reset(), step_async(), step_wait(), step()`
with auto-reset semantics -- on `dones[i]` the returned `obs[i]` is the post-reset
observation, `infos[i]["terminal_observation"]` holds the true last observation and
`infos[i]["TimeLimit.truncated"]` flags time-limit endings (relied on by
`rewards/reward_wrapper.py:100-104`, `data/rollout.py:161-167`).

In addition to the dict-of-infos protocol, environments may implement the *array* fast
path `step_wait_arrays()` which returns `(obs, rews, dones, next_obs_fixed, truncated)`
without building `num_envs` Python dicts per step; the GPU rollout collector uses it
when present and falls back to parsing `infos` otherwise.
"""
from __future__ import annotations

import abc
import os
import weakref
from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np

from imitation_amd import spaces


class VecEnv(abc.ABC):
    """Abstract asynchronous vectorised environment (SB3 `VecEnv` surface)."""

    def __init__(self, num_envs: int, observation_space: spaces.Space, action_space: spaces.Space):
        self.num_envs = int(num_envs)
        self.observation_space = observation_space
        self.action_space = action_space

    @abc.abstractmethod
    def reset(self) -> np.ndarray:
        ...

    @abc.abstractmethod
    def step_async(self, actions: np.ndarray) -> None:
        ...

    @abc.abstractmethod
    def step_wait(self) -> Tuple[np.ndarray, np.ndarray, np.ndarray, List[Dict[str, Any]]]:
        ...

    def step(self, actions: np.ndarray):
        self.step_async(actions)
        return self.step_wait()

    def close(self) -> None:
        pass

    def seed(self, seed: Optional[int] = None) -> Sequence[Optional[int]]:
        return [seed] * self.num_envs

    @property
    def unwrapped(self) -> "VecEnv":
        return self.venv.unwrapped if isinstance(self, VecEnvWrapper) else self

    # SB3 API used by `BaseAlgorithm.set_env` checks; kept for duck-typing.
    def env_is_wrapped(self, wrapper_class, indices=None):
        return [False] * self.num_envs


class VecEnvWrapper(VecEnv):
    """Base wrapper: forwards everything to `venv` unless overridden."""

    def __init__(self, venv: VecEnv, observation_space=None, action_space=None):
        self.venv = venv
        super().__init__(
            venv.num_envs,
            observation_space or venv.observation_space,
            action_space or venv.action_space,
        )

    def step_async(self, actions):
        self.venv.step_async(actions)

    def reset(self):
        return self.venv.reset()

    def step_wait(self):
        return self.venv.step_wait()

    def close(self):
        return self.venv.close()

    def seed(self, seed=None):
        return self.venv.seed(seed)

    def __getattr__(self, name):
        if name.startswith("_") or name == "venv":
            raise AttributeError(name)
        return getattr(self.venv, name)


_ENV_NOISE_LIB: List[Any] = []


def _env_noise_lib():
    """ctypes handle of libimitation_envnoise.so (built next to libimitation_hip.so), or None."""
    if not _ENV_NOISE_LIB:
        import ctypes as C

        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libimitation_envnoise.so")
        lib = None
        if os.path.exists(path):
            lib = C.CDLL(path)
            lib.ia_env_noise_create.argtypes, lib.ia_env_noise_create.restype = [], C.c_void_p
            lib.ia_env_noise_post.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_void_p,
                                              C.c_void_p, C.c_void_p, C.c_double, C.c_double]
            lib.ia_env_noise_post.restype = C.c_int
            lib.ia_env_noise_wait_step.argtypes, lib.ia_env_noise_wait_step.restype = [C.c_void_p, C.c_int], C.c_int
            lib.ia_env_noise_finish.argtypes, lib.ia_env_noise_finish.restype = [C.c_void_p], C.c_int
            lib.ia_env_noise_destroy.argtypes, lib.ia_env_noise_destroy.restype = [C.c_void_p], None
        _ENV_NOISE_LIB.append(lib)
    return _ENV_NOISE_LIB[0]


def _destroy_worker(lib, worker, *keep_alive) -> None:
    lib.ia_env_noise_destroy(worker)


class ArrayVecEnv(VecEnv):
    """VecEnv whose native step produces arrays; dict infos are derived from them."""

    @abc.abstractmethod
    def step_wait_arrays(self) -> Tuple[np.ndarray, np.ndarray, np.ndarray, np.ndarray, np.ndarray]:
        """Returns `(obs, rews, dones, next_obs_fixed, truncated)`.

        `obs` is post-auto-reset; `next_obs_fixed[i]` is the true successor observation
        (== terminal observation where `dones[i]`, else `obs[i]`); `truncated[i]` is True
        iff the episode ended by time limit."""

    def step_wait(self):
        obs, rews, dones, nxt, trunc = self.step_wait_arrays()
        infos: List[Dict[str, Any]] = [{} for _ in range(self.num_envs)]
        for i in np.flatnonzero(dones):
            infos[i]["terminal_observation"] = nxt[i].copy()
            infos[i]["TimeLimit.truncated"] = bool(trunc[i])
        return obs, rews, dones, infos


class GymStyleVecEnv(VecEnv):
    """A plain SB3-protocol VecEnv around an array environment: ONLY `reset / step_async / step_wait`, the latter with
    one info DICT per env -- `terminal_observation` and `TimeLimit.truncated` on episode ends (App. A.9) plus the
    `episode = {"r", "l", "t"}` entry [SB3 Monitor] adds there (`util/util.py:150`, read by `data/rollout.py:536-547`
    and SB3's `ep_info_buffer`). This is what a `DummyVecEnv` / `SubprocVecEnv` of Monitor-wrapped gym environments
    hands the trainer; it is deliberately NOT an `ArrayVecEnv`, so the wrappers and the rollout collector take their
    generic per-env branch (`wrappers.step_arrays`, `rewards/reward_wrapper.py:98-109`)."""

    def __init__(self, env: "ArrayVecEnv"):
        super().__init__(env.num_envs, env.observation_space, env.action_space)
        self._env = env
        self._ret = np.zeros(env.num_envs, dtype=np.float64)
        self._len = np.zeros(env.num_envs, dtype=np.int64)
        self._t = 0

    def reset(self):
        self._ret[:] = 0.0
        self._len[:] = 0
        return self._env.reset()

    def seed(self, seed=None):
        return self._env.seed(seed)

    def set_lookahead(self, steps: int) -> None:
        # (the synthetic env's own draw-ahead horizon: forwarded so that BOTH protocols step the same environment at the
        #  same cost -- what differs between them is then the dict protocol alone)
        if hasattr(self._env, "set_lookahead"):
            self._env.set_lookahead(steps)

    def step_async(self, actions):
        self._env.step_async(actions)

    def step_wait(self):
        obs, rews, dones, nxt, trunc = self._env.step_wait_arrays()
        self._ret += rews
        self._len += 1
        self._t += 1
        infos: List[Dict[str, Any]] = [{} for _ in range(self.num_envs)]
        for i in np.flatnonzero(dones):
            infos[i]["terminal_observation"] = nxt[i].copy()
            infos[i]["TimeLimit.truncated"] = bool(trunc[i])
            infos[i]["episode"] = {"r": float(self._ret[i]), "l": int(self._len[i]), "t": float(self._t)}
            self._ret[i], self._len[i] = 0.0, 0
        return obs, rews, dones, infos

    def close(self):
        self._env.close()


class SyntheticVecEnv(ArrayVecEnv):
    """HalfCheetah-shaped synthetic control environment.

    Dynamics `o' = 0.9*o + 0.1*tanh(W a) + 0.05*xi`, `W in R^{obs x act}`, `xi ~ N(0,1)`,
    env reward `-reward_scale*|a|^2` (0 by default), fixed horizon with `TimeLimit.truncated=True`.
    Discrete action spaces are embedded through a fixed table of `n` action vectors.
    All randomness comes from a private `np.random.Generator` (never the global streams
    the trainer's index draws use).
    """

    def __init__(
        self,
        num_envs: int = 1024,
        obs_dim: int = 17,
        act_dim: int = 6,
        horizon: int = 1000,
        seed: int = 0,
        obs_dtype=np.float32,
        n_discrete: Optional[int] = None,
        stagger: bool = False,
        reward_scale: float = 0.0,
        prefetch_noise: bool = True,
    ):
        obs_space = spaces.Box(-np.inf, np.inf, (obs_dim,), obs_dtype)
        if n_discrete is None:
            act_space: spaces.Space = spaces.Box(-1.0, 1.0, (act_dim,), np.float32)
        else:
            act_space = spaces.Discrete(n_discrete)
        super().__init__(num_envs, obs_space, act_space)
        self.obs_dim, self.act_dim, self.horizon = obs_dim, act_dim, int(horizon)
        self._seed0 = seed
        self._rng = np.random.default_rng(seed)
        wrng = np.random.default_rng(10_000 + seed)
        self._W = wrng.standard_normal((act_dim, obs_dim)).astype(np.float64) / np.sqrt(act_dim)
        self._table = (
            wrng.uniform(-1, 1, (n_discrete, act_dim)) if n_discrete is not None else None
        )
        self._obs = np.zeros((num_envs, obs_dim), dtype=np.float64)
        self._t = np.zeros(num_envs, dtype=np.int64)
        self._stagger = stagger
        self._reward_scale = float(reward_scale)
        self._actions: Optional[np.ndarray] = None
        # The generator draws of the NEXT steps (process noise, then the fresh observations of the
        # environments whose episode ends in that step -- known in advance, the horizon is fixed) do not
        # depend on the actions: a helper thread of libimitation_envnoise.so fills them through NumPy's
        # own `random_standard_normal_fill`, `lookahead` steps per job, while the learner is busy (the
        # policy step of a transition, or -- with `lookahead` = rollout length, set by `PPO` -- the whole
        # generator update between two rollouts), as subprocess env workers would overlap with the
        # learner. Same stream, same order, same values as drawing inside the step (the helper is
        # optional: without the library the step draws inline).
        self._helper = _env_noise_lib() if (prefetch_noise and os.environ.get("IA_ENV_PREFETCH", "1") != "0") else None
        self.lookahead = 1   # env steps per helper job (buffers grow on demand)
        self._xi = self._fresh_buf = self._n_fresh = None
        self._pending = None  # [generator state before the job's draws, n_done per step, steps consumed]
        self._worker = self._bitgen_addr = 0
        if self._helper is not None:
            self._bitgen_addr = int(self._rng.bit_generator.ctypes.bit_generator.value)
            self._worker = self._helper.ia_env_noise_create()
            # the worker may still be drawing from this generator into the job buffers when the env goes
            # away: the finaliser keeps both alive until the thread has been joined
            self._keep = []
            weakref.finalize(self, _destroy_worker, self._helper, self._worker, self._rng, self._keep)

    def _fresh(self, n: int) -> np.ndarray:
        return 0.1 * self._rng.standard_normal((n, self.obs_dim))

    def _plan_next_draws(self) -> None:
        """Posts the draws of the next `lookahead` steps (only when no job is outstanding)."""
        if self._helper is None or self._pending is not None:
            return
        K, n, D = max(1, int(self.lookahead)), self.num_envs, self.obs_dim
        if self._xi is None or self._xi.shape[0] != K:
            self._xi, self._fresh_buf = np.empty((K, n, D)), np.empty((K, n, D))
            self._n_fresh = np.zeros(K, dtype=np.int64)
            self._keep[:] = [self._xi, self._fresh_buf, self._n_fresh]
        t = self._t.copy()
        n_done = []
        for j in range(K):  # episode ends of the next K steps (fixed horizon: independent of the actions)
            t += 1
            ends = t >= self.horizon
            n_done.append(int(np.count_nonzero(ends)))
            t[ends] = 0
        self._n_fresh[:] = np.asarray(n_done) * D
        state0 = self._rng.bit_generator.state
        rc = self._helper.ia_env_noise_post(self._worker, self._bitgen_addr, K, n * D, n * D, self._xi.ctypes.data,
                                            self._n_fresh.ctypes.data, self._fresh_buf.ctypes.data, 0.05, 0.1)
        assert rc == 0, rc
        self._pending = [state0, n_done, 0]

    def set_lookahead(self, steps: int) -> None:
        """Draw-ahead horizon = the learner's rollout length: the job that is posted at the end of a
        rollout's last step then covers exactly the next rollout and is filled during the generator
        update in between. Re-plans from the current position."""
        if int(steps) != self.lookahead:
            self._cancel_planned_draws()
            self.lookahead = int(steps)
            self._plan_next_draws()

    def _take_planned_draws(self):
        """-> (0.05 * xi, 0.1 * fresh-draws or None, n_done) of the next step of the outstanding job (the helper
        stores the draws already scaled: the same single IEEE multiply NumPy would do)."""
        _, n_done, j = self._pending
        rc = self._helper.ia_env_noise_wait_step(self._worker, j)
        assert rc == 0, rc
        out = (self._xi[j], (self._fresh_buf[j, :n_done[j]] if n_done[j] else None), n_done[j])
        self._pending[2] = j + 1
        if j + 1 == len(n_done):   # job used up (the views above stay valid until the next post)
            rc = self._helper.ia_env_noise_finish(self._worker)
            assert rc == 0, rc
            self._pending = None
        return out

    def _state_at_cursor(self):
        """Generator state as if exactly the consumed steps of the outstanding job had been drawn."""
        state0, n_done, used = self._pending
        g = np.random.Generator(type(self._rng.bit_generator)())
        g.bit_generator.state = state0
        for j in range(used):
            g.standard_normal((self.num_envs, self.obs_dim))
            if n_done[j]:
                g.standard_normal((n_done[j], self.obs_dim))
        return g.bit_generator.state

    def _cancel_planned_draws(self) -> None:
        """Drops the draws that were made ahead: the generator is back where inline draws would have left it."""
        if self._pending is not None:
            cursor = self._state_at_cursor()
            rc = self._helper.ia_env_noise_finish(self._worker)
            assert rc == 0, rc
            self._rng.bit_generator.state = cursor
            self._pending = None

    def get_state(self):
        """Everything `step` depends on (used by `checkpoint.save_checkpoint`)."""
        rng = self._state_at_cursor() if self._pending is not None else self._rng.bit_generator.state
        return {"rng": rng, "obs": self._obs.copy(), "t": self._t.copy()}

    def set_state(self, state) -> None:
        self._cancel_planned_draws()
        self._rng.bit_generator.state = state["rng"]
        self._obs, self._t = state["obs"].copy(), state["t"].copy()
        self._actions = None

    def reset(self) -> np.ndarray:
        self._cancel_planned_draws()
        self._obs = self._fresh(self.num_envs)
        self._t[:] = 0
        if self._stagger:
            # Desynchronise episode ends across envs (keeps a fixed horizon per episode
            # only when stagger=False; used by tests that want dones in every step).
            self._t[:] = self._rng.integers(0, self.horizon, self.num_envs)
        self._plan_next_draws()
        return self._obs.astype(self.observation_space.dtype)

    def step_async(self, actions: np.ndarray) -> None:
        self._actions = np.asarray(actions)

    def step_wait_arrays(self):
        a = self._actions
        assert a is not None, "step_async must be called first"
        self._actions = None
        if self._table is not None:
            a = self._table[np.asarray(a).reshape(-1).astype(np.int64)]
        a = a.reshape(self.num_envs, self.act_dim).astype(np.float64)
        nxt = 0.9 * self._obs + 0.1 * np.tanh(a @ self._W)
        fresh = None
        if self._pending is not None:
            sxi, fresh, planned = self._take_planned_draws()
        else:
            sxi, planned = 0.05 * self._rng.standard_normal(nxt.shape), -1
        nxt += sxi
        self._t += 1
        dones = self._t >= self.horizon
        dt = self.observation_space.dtype
        next_fixed = nxt.astype(dt)
        n_done = int(dones.sum())
        if n_done:
            assert planned in (-1, n_done)
            nxt[dones] = fresh if fresh is not None else self._fresh(n_done)
            self._t[dones] = 0
        self._obs = nxt
        obs = nxt.astype(dt)
        self._plan_next_draws()
        if self._reward_scale:
            rews = (-self._reward_scale * (a * a).sum(axis=1)).astype(np.float32)
        else:
            rews = np.zeros(self.num_envs, dtype=np.float32)
        return obs, rews, dones, next_fixed, dones.copy()


class SyntheticImageVecEnv(ArrayVecEnv):
    """Image-observation synthetic environment: uint8 `[C, H, W]` frames (the space SB3's `CnnPolicy` / the
    reference's `CnnRewardNet` take) rendered from a low-dimensional latent with `SyntheticVecEnv`'s dynamics,
    `z' = 0.9 z + 0.1 tanh(W a) + 0.05 xi`, `frame = uint8(clip(127.5 + 60 R z, 0, 255))`; fixed horizon with
    `TimeLimit.truncated`. Discrete (table of action vectors) or Box actions."""

    def __init__(self, num_envs: int = 8, shape=(4, 36, 36), act_dim: int = 3, horizon: int = 20, seed: int = 0,
                 n_discrete: Optional[int] = None, latent_dim: int = 6):
        obs_space = spaces.Box(0, 255, tuple(shape), np.uint8)
        act_space: spaces.Space = (spaces.Box(-1.0, 1.0, (act_dim,), np.float32) if n_discrete is None
                                   else spaces.Discrete(n_discrete))
        super().__init__(num_envs, obs_space, act_space)
        self.act_dim, self.horizon, self.latent_dim = act_dim, int(horizon), latent_dim
        self._rng = np.random.default_rng(seed)
        wrng = np.random.default_rng(20_000 + seed)
        self._W = wrng.standard_normal((act_dim, latent_dim)) / np.sqrt(act_dim)
        self._R = wrng.standard_normal((latent_dim, int(np.prod(shape)))) / np.sqrt(latent_dim)
        self._R32 = self._R.astype(np.float32)
        self._table = wrng.uniform(-1, 1, (n_discrete, act_dim)) if n_discrete is not None else None
        self._z = np.zeros((num_envs, latent_dim))
        self._t = np.zeros(num_envs, dtype=np.int64)
        self._actions: Optional[np.ndarray] = None

    def render_frames(self, z: np.ndarray) -> np.ndarray:
        # (a clipped linear map, in float32: rendering 64 frames of 4 x 84 x 84 must not dominate a round's host time)
        # einsum, not BLAS: a fixed summation order (the frames must be the same on every host) and no thread pool
        f = np.einsum("nl,lk->nk", z.astype(np.float32), self._R32)
        f *= np.float32(60.0)
        f += np.float32(127.5)
        np.clip(f, 0.0, 255.0, out=f)
        return f.astype(np.uint8).reshape(len(z), *self.observation_space.shape)

    def reset(self) -> np.ndarray:
        self._z = self._rng.standard_normal((self.num_envs, self.latent_dim))
        self._t[:] = 0
        return self.render_frames(self._z)

    def step_async(self, actions: np.ndarray) -> None:
        self._actions = np.asarray(actions)

    def step_wait_arrays(self):
        a = self._actions
        assert a is not None, "step_async must be called first"
        self._actions = None
        if self._table is not None:
            a = self._table[np.asarray(a).reshape(-1).astype(np.int64)]
        a = a.reshape(self.num_envs, self.act_dim).astype(np.float64)
        z = 0.9 * self._z + 0.1 * np.tanh(a @ self._W) + 0.05 * self._rng.standard_normal(self._z.shape)
        self._t += 1
        dones = self._t >= self.horizon
        next_fixed = self.render_frames(z)
        n_done = int(dones.sum())
        if n_done:
            z[dones] = self._rng.standard_normal((n_done, self.latent_dim))
            self._t[dones] = 0
        self._z = z
        obs = next_fixed.copy()
        if n_done:
            obs[dones] = self.render_frames(z[dones])
        rews = (-0.1 * (a * a).sum(axis=1)).astype(np.float32)   # (the adversarial trainers replace it)
        return obs, rews, dones, next_fixed, dones.copy()


class CountingVecEnv(ArrayVecEnv):
    """Deterministic bookkeeping env in the spirit of the reference's `_CountingEnv`
    (`tests/data/test_wrappers.py:14-74`): `obs = t`, `rew = 10 t`, per-env episode
    lengths; terminal (not truncated) endings."""

    def __init__(self, episode_lengths: Sequence[int], obs_dim: int = 1, act_dim: int = 1):
        n = len(episode_lengths)
        super().__init__(
            n,
            spaces.Box(-np.inf, np.inf, (obs_dim,), np.float32),
            spaces.Box(-np.inf, np.inf, (act_dim,), np.float32),
        )
        self._lens = np.asarray(episode_lengths, dtype=np.int64)
        self._t = np.zeros(n, dtype=np.int64)
        self._obs_dim = obs_dim
        self._actions = None

    def _mk(self, t):
        return np.repeat(t.astype(np.float32)[:, None], self._obs_dim, axis=1)

    def reset(self):
        self._t[:] = 0
        return self._mk(self._t)

    def step_async(self, actions):
        self._actions = actions

    def step_wait_arrays(self):
        self._actions = None
        self._t += 1
        rews = (10.0 * self._t).astype(np.float32)
        dones = self._t >= self._lens
        next_fixed = self._mk(self._t)
        self._t[dones] = 0
        obs = self._mk(self._t)
        return obs, rews, dones, next_fixed, np.zeros_like(dones)
