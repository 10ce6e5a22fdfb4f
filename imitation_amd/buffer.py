"""Generator replay ring resident in HBM (`data/buffer.py:30-416`).

FIFO ring over `obs, acts, next_obs, dones` in HBM plus `infos` objects in a host-side ring at the same
positions (`rews` are dropped, `buffer.py:316-329,409-412`); the host ring exists only once some stored
transition carried a non-empty info dict (array envs never allocate it: `sample` then returns `{}`s). All index arithmetic stays on the host and follows
the reference exactly -- `store` splits at the wrap point (`:184-192`), `_idx=(idx+n)%cap`,
`_n_data=min(n_data+n,cap)` (`:208-214`), over-capacity stores keep only the LAST `capacity`
rows (`:174-178`), `sample` draws `np.random.randint(size, size=n)` from the GLOBAL NumPy
stream (`:231`) -- so on identical seeds the sampled rows are bit-identical. Observations are
stored as fp32 (the value `.float()` produces at `rewards/reward_nets.py:90-110`).
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch as th

from imitation_amd import data_types as dt
from imitation_amd import spaces
from imitation_amd.networks import TransitionTable


class ReplayBuffer:
    def __init__(self, capacity: int, venv=None, *, obs_shape=None, act_shape=None, obs_dtype=None, act_dtype=None,
                 device="cuda"):
        if venv is not None:
            if obs_shape is not None or act_shape is not None or obs_dtype is not None or act_dtype is not None:
                raise ValueError("Cannot specify both shapes/dtypes and also an environment.")
            obs_shape, obs_dtype = tuple(venv.observation_space.shape), venv.observation_space.dtype
            act_shape, act_dtype = tuple(venv.action_space.shape), venv.action_space.dtype
            self.discrete = isinstance(venv.action_space, spaces.Discrete)
        else:
            if any(x is None for x in (obs_shape, act_shape, obs_dtype, act_dtype)):
                raise ValueError("Shape or dtype missing and no environment specified.")
            self.discrete = len(tuple(act_shape)) == 0 and np.issubdtype(np.dtype(act_dtype), np.integer)
        self.capacity = int(capacity)
        self.obs_shape, self.act_shape = tuple(obs_shape), tuple(act_shape)
        self.obs_dtype, self.act_dtype = np.dtype(obs_dtype), np.dtype(act_dtype)
        self.device = th.device(device)
        od, ad = int(np.prod(self.obs_shape)), int(np.prod(self.act_shape))
        self._obs = th.zeros(self.capacity, od, device=self.device)
        self._next = th.zeros(self.capacity, od, device=self.device)
        self._acts = (th.zeros(self.capacity, dtype=th.int64, device=self.device) if self.discrete
                      else th.zeros(self.capacity, ad, device=self.device))
        self._dones = th.zeros(self.capacity, dtype=th.uint8, device=self.device)
        self._infos: Optional[np.ndarray] = None   # host ring of info dicts, allocated on first use
        self._n_data = 0
        self._idx = 0
        self.table = TransitionTable(self._obs, self._acts, self._next, self._dones, self.discrete)

    @classmethod
    def from_data(cls, transitions, capacity: Optional[int] = None, truncate_ok: bool = False, device="cuda"):
        cap = len(transitions.obs) if capacity is None else capacity
        inst = cls(cap, obs_shape=transitions.obs.shape[1:], act_shape=transitions.acts.shape[1:],
                   obs_dtype=transitions.obs.dtype, act_dtype=transitions.acts.dtype, device=device)
        inst.store(transitions, truncate_ok=truncate_ok)
        return inst

    def size(self) -> int:
        assert 0 <= self._n_data <= self.capacity
        return self._n_data

    def _put(self, lo: int, obs, acts, nxt, dones) -> None:
        hi = lo + len(obs)
        dev = self.device
        self._obs[lo:hi].copy_(th.from_numpy(np.ascontiguousarray(obs, dtype=np.float32)).reshape(len(obs), -1))
        self._next[lo:hi].copy_(th.from_numpy(np.ascontiguousarray(nxt, dtype=np.float32)).reshape(len(obs), -1))
        if self.discrete:
            self._acts[lo:hi].copy_(th.from_numpy(np.ascontiguousarray(acts, dtype=np.int64)).reshape(-1))
        else:
            self._acts[lo:hi].copy_(th.from_numpy(np.ascontiguousarray(acts, dtype=np.float32)).reshape(len(obs), -1))
        self._dones[lo:hi].copy_(th.from_numpy(np.ascontiguousarray(dones, dtype=np.uint8)))

    def _put_infos(self, lo: int, infos, m: int) -> None:
        """Host ring of `infos` (`buffer.py:316-329` stores them like any other key)."""
        if infos is None and self._infos is None:
            return
        if self._infos is None:
            self._infos = np.array([{} for _ in range(self.capacity)], dtype=object)
        if infos is None:
            self._infos[lo:lo + m] = [{} for _ in range(m)]
        elif isinstance(infos, np.ndarray) and infos.dtype == object:
            self._infos[lo:lo + m] = infos          # (object rows copy as references: no Python loop)
        else:
            self._infos[lo:lo + m] = list(infos)

    @staticmethod
    def _meaningful(infos) -> Optional[np.ndarray]:
        if infos is None:
            return None
        if isinstance(infos, np.ndarray) and infos.dtype == object:
            return infos if any(infos.tolist()) else None   # (a dict is true iff it is non-empty)
        return np.asarray(infos, dtype=object) if any(len(i) for i in infos) else None

    def store(self, transitions, truncate_ok: bool = True) -> None:
        obs, acts = np.asarray(transitions.obs), np.asarray(transitions.acts)
        nxt, dones = np.asarray(transitions.next_obs), np.asarray(transitions.dones)
        infos = self._meaningful(getattr(transitions, "infos", None))
        n = len(obs)
        if n == 0:
            raise ValueError("Trying to store empty data.")
        if obs.shape[1:] != self.obs_shape or acts.shape[1:] != self.act_shape:
            raise ValueError("Wrong data shape")
        if n > self.capacity:
            if not truncate_ok:
                raise ValueError("Not enough capacity to store data.")
            obs, acts, nxt, dones = (a[-self.capacity:] for a in (obs, acts, nxt, dones))
            infos = None if infos is None else infos[-self.capacity:]
            n = self.capacity
        if self._idx + n > self.capacity:
            rem = self.capacity - self._idx
            self._put(self._idx, obs[:rem], acts[:rem], nxt[:rem], dones[:rem])
            self._put(0, obs[rem:], acts[rem:], nxt[rem:], dones[rem:])
            self._put_infos(self._idx, None if infos is None else infos[:rem], rem)
            self._put_infos(0, None if infos is None else infos[rem:], n - rem)
        else:
            self._put(self._idx, obs, acts, nxt, dones)
            self._put_infos(self._idx, infos, n)
        self._idx = (self._idx + n) % self.capacity
        self._n_data = min(self._n_data + n, self.capacity)

    def store_from_rollout(self, rb, order: np.ndarray, truncate_ok: bool = True, infos=None) -> None:
        """`store` for transitions that already live in HBM (the PPO rollout tile `rb`): the rows
        `order` (time-major offsets, reference emission order) are gathered device-to-device into
        the ring; same index arithmetic as `store`."""
        from imitation_amd import _lib as L
        n = len(order)
        if n == 0:
            raise ValueError("Trying to store empty data.")
        if n > self.capacity:
            if not truncate_ok:
                raise ValueError("Not enough capacity to store data.")
            order = order[-self.capacity:]
            infos = None if infos is None else infos[-self.capacity:]
            n = self.capacity
        infos = self._meaningful(infos)
        order_dev = th.from_numpy(np.ascontiguousarray(order)).to(self.device)
        T, ne = rb.buffer_size, rb.n_envs
        od = self._obs.shape[1]
        src_obs = rb.obs.reshape((T + 1) * ne, od)
        src_next = rb.next_fixed.reshape(T * ne, od)
        src_act = rb.clipped.reshape(T * ne, -1)
        src_done = rb.dones.reshape(T * ne)

        def put(lo: int, sel: th.Tensor) -> None:
            m = sel.numel()
            L.call("ia_gather_rows", L.ptr(src_obs), L.ptr(sel), m, od, L.ptr(self._obs[lo:lo + m]), L.stream())
            L.call("ia_gather_rows", L.ptr(src_next), L.ptr(sel), m, od, L.ptr(self._next[lo:lo + m]), L.stream())
            if self.discrete:
                self._acts[lo:lo + m] = src_act[sel, 0].long()
            else:
                L.call("ia_gather_rows", L.ptr(src_act), L.ptr(sel), m, src_act.shape[1],
                       L.ptr(self._acts[lo:lo + m]), L.stream())
            self._dones[lo:lo + m] = src_done[sel]

        if self._idx + n > self.capacity:
            rem = self.capacity - self._idx
            put(self._idx, order_dev[:rem].contiguous())
            put(0, order_dev[rem:].contiguous())
            self._put_infos(self._idx, None if infos is None else infos[:rem], rem)
            self._put_infos(0, None if infos is None else infos[rem:], n - rem)
        else:
            put(self._idx, order_dev)
            self._put_infos(self._idx, infos, n)
        self._idx = (self._idx + n) % self.capacity
        self._n_data = min(self._n_data + n, self.capacity)

    def sample_indices(self, n_samples: int) -> np.ndarray:
        if self.size() == 0:
            raise ValueError("Buffer is empty")
        return np.random.randint(self.size(), size=n_samples)

    def sample(self, n_samples: int) -> dt.Transitions:
        """Host-facing sample (API parity); the trainer's fused path uses `sample_indices`."""
        ind = self.sample_indices(n_samples)
        arr = self._arrays
        return dt.Transitions(obs=arr["obs"][ind], acts=arr["acts"][ind], next_obs=arr["next_obs"][ind],
                              dones=arr["dones"][ind], infos=None if self._infos is None else self._infos[ind])

    @property
    def _arrays(self) -> Dict[str, np.ndarray]:
        """Host copy of the ring in the reference's dtypes/shapes (tests, checkpoints)."""
        c = self.capacity
        acts = self._acts.cpu().numpy()
        return {
            "obs": self._obs.cpu().numpy().reshape(c, *self.obs_shape).astype(self.obs_dtype),
            "acts": acts.astype(self.act_dtype).reshape(c, *self.act_shape),
            "next_obs": self._next.cpu().numpy().reshape(c, *self.obs_shape).astype(self.obs_dtype),
            "dones": self._dones.cpu().numpy().astype(bool),
        }
