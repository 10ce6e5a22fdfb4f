"""Host-side data containers and index bookkeeping of the GAIL/AIRL round.

Everything here is integer / layout work that stays on the host so that, on identical
seeds, batch composition is bit-identical to the reference:

* `Transitions` / `TrajectoryWithRew` -- reference `data/types.py:335-638` (validation rules).
* `flatten_trajectories` -- `data/rollout.py:563-621`.
* `ExpertIndexStream` -- the index sequence (and torch global-RNG consumption) of
  `algorithms/base.py:226-288` `make_data_loader(..., shuffle=True, drop_last=True)` wrapped in
  `util/util.py:215-241` `endless_iter`; the expert rows themselves live in HBM and are
  gathered there, so only 8-byte indices cross PCIe.
* `segment_order` -- the row order in which `BufferingWrapper.pop_trajectories` +
  `flatten_trajectories_with_rew` (`data/wrappers.py:93-148`, `data/rollout.py:613-621`)
  emit a rollout's transitions (completed episodes in completion order, then in-progress
  fragments by env index).
"""
from __future__ import annotations

import dataclasses
from typing import Iterable, Iterator, List, Optional, Sequence, Tuple

import numpy as np
import torch as th


@dataclasses.dataclass(frozen=True)
class Transitions:
    """A batch of obs-act-obs-done transitions (`data/types.py:580-621`)."""

    obs: np.ndarray
    acts: np.ndarray
    next_obs: np.ndarray
    dones: np.ndarray
    infos: Optional[np.ndarray] = None

    def __post_init__(self):
        n = len(self.obs)
        if len(self.acts) != n:
            raise ValueError(f"obs and acts must have same number of timesteps: {n} != {len(self.acts)}")
        if self.obs.shape != self.next_obs.shape:
            raise ValueError(f"obs and next_obs must have same shape: {self.obs.shape} != {self.next_obs.shape}")
        if self.obs.dtype != self.next_obs.dtype:
            raise ValueError(f"obs and next_obs must have the same dtype: {self.obs.dtype} != {self.next_obs.dtype}")
        if self.dones.shape != (n,):
            raise ValueError(f"dones must be 1D array, one entry for each timestep: {self.dones.shape} != ({n},)")
        if self.dones.dtype != bool:
            raise ValueError(f"dones must be boolean, not {self.dones.dtype}")
        if self.infos is None:
            object.__setattr__(self, "infos", np.array([{}] * n))
        elif len(self.infos) != n:
            raise ValueError(f"obs and infos must have same number of timesteps: {n} != {len(self.infos)}")

    def __len__(self) -> int:
        return len(self.obs)

    def __getitem__(self, key):
        d = {f.name: getattr(self, f.name)[key] for f in dataclasses.fields(self)}
        return dataclasses.replace(self, **d) if isinstance(key, slice) else d


@dataclasses.dataclass(frozen=True)
class TransitionsWithRew(Transitions):
    rews: Optional[np.ndarray] = None

    def __post_init__(self):
        super().__post_init__()
        if self.rews is None or self.rews.shape != (len(self.obs),):
            raise ValueError("rewards must be 1D array, one entry for each timestep")
        if not np.issubdtype(self.rews.dtype, np.floating):
            raise ValueError(f"rewards dtype {self.rews.dtype} not a float")


@dataclasses.dataclass(frozen=True)
class TrajectoryWithRew:
    """`data/types.py:335-440`: `obs` has one more row than `acts`/`rews`."""

    obs: np.ndarray
    acts: np.ndarray
    rews: np.ndarray
    infos: Optional[np.ndarray]
    terminal: bool

    def __post_init__(self):
        if len(self.obs) != len(self.acts) + 1:
            raise ValueError(f"expected one more observations than actions: {len(self.obs)} != {len(self.acts)} + 1")
        if len(self.acts) == 0:
            raise ValueError("Degenerate trajectory: must have at least one action.")
        if self.rews.shape != (len(self.acts),):
            raise ValueError("rewards must be 1D array, one entry for each action")

    def __len__(self) -> int:
        return len(self.acts)


def flatten_trajectories(trajectories: Sequence[TrajectoryWithRew]) -> TransitionsWithRew:
    """`data/rollout.py:563-621`: `dones` is True only on the last step of terminal trajectories."""
    obs, nxt, acts, dones, infos, rews = [], [], [], [], [], []
    for tr in trajectories:
        acts.append(tr.acts)
        obs.append(tr.obs[:-1])
        nxt.append(tr.obs[1:])
        d = np.zeros(len(tr), dtype=bool)
        d[-1] = tr.terminal
        dones.append(d)
        infos.append(tr.infos if tr.infos is not None else np.array([{}] * len(tr)))
        rews.append(tr.rews)
    return TransitionsWithRew(obs=np.concatenate(obs), acts=np.concatenate(acts), next_obs=np.concatenate(nxt),
                              dones=np.concatenate(dones), infos=np.concatenate(infos), rews=np.concatenate(rews))


def trajectories_from_legacy_npz(path: str) -> List[TrajectoryWithRew]:
    """Decodes the reference's legacy `.npz` rollout layout (`data/serialize.py:50-67`)."""
    d = np.load(path, allow_pickle=True)
    n_traj = len(d["indices"]) + 1
    acts = np.split(d["acts"], d["indices"])
    rews = np.split(d["rews"], d["indices"])
    obs = np.split(d["obs"], d["indices"] + np.arange(1, n_traj))
    return [TrajectoryWithRew(obs=o, acts=a, rews=r, infos=None, terminal=bool(t))
            for o, a, r, t in zip(obs, acts, rews, d["terminal"])]


def as_transitions(demonstrations) -> Transitions:
    """Accepts what `make_data_loader` accepts for the device-resident path: `Transitions` or a
    sequence of trajectories (`algorithms/base.py:254-263`). Duck-types foreign containers
    (e.g. the reference's own dataclasses) by attribute."""
    if isinstance(demonstrations, Transitions):
        return demonstrations
    if all(hasattr(demonstrations, k) for k in ("obs", "acts", "next_obs", "dones")):
        return Transitions(obs=np.asarray(demonstrations.obs), acts=np.asarray(demonstrations.acts),
                           next_obs=np.asarray(demonstrations.next_obs), dones=np.asarray(demonstrations.dones))
    if isinstance(demonstrations, Sequence) and len(demonstrations) and hasattr(demonstrations[0], "terminal"):
        trajs = [t if isinstance(t, TrajectoryWithRew) else TrajectoryWithRew(
            obs=np.asarray(t.obs), acts=np.asarray(t.acts),
            rews=np.asarray(getattr(t, "rews", np.zeros(len(t.acts), np.float32))), infos=None,
            terminal=bool(t.terminal)) for t in demonstrations]
        tr = flatten_trajectories(trajs)
        return Transitions(obs=tr.obs, acts=tr.acts, next_obs=tr.next_obs, dones=tr.dones)
    raise TypeError(f"`demonstrations` unexpected type {type(demonstrations)}")


class ExpertIndexStream:
    """Index view of the reference's endless shuffled expert `DataLoader`.

    Consumes torch's GLOBAL CPU generator exactly like
    `endless_iter(DataLoader(transitions, batch_size, shuffle=True, drop_last=True))` does
    (torch 2.x single-process loader):
      * every `iter(loader)` draws one int64 (`_base_seed`);
      * the first `next()` on it makes `RandomSampler` draw one int64 seed for a private
        generator, from which `randperm(N)` is taken;
      * `endless_iter` creates one throw-away iterator for its `iter(x) == x` guard
        (`util/util.py:236`), then one more whose first batch is fetched and discarded
        (`util/util.py:240,353-355`), then a fresh iterator per epoch.
    `tests/test_host_logic.py` pins this against a real `torch.utils.data.DataLoader`.
    """

    def __init__(self, n_samples: int, batch_size: int):
        if batch_size <= 0:
            raise ValueError(f"batch_size={batch_size} must be positive.")
        if n_samples < batch_size:
            raise ValueError(f"Number of transitions in `demonstrations` {n_samples} "
                             f"is smaller than batch size {batch_size}.")
        self.n, self.batch_size = int(n_samples), int(batch_size)
        self.batches_per_epoch = self.n // self.batch_size
        self._draw_int64()          # guard iterator: base seed only
        self._draw_int64()          # first-element iterator: base seed ...
        self._permutation()         # ... + sampler seed + randperm (batch discarded)
        self._perm: Optional[np.ndarray] = None
        self._pos = 0

    @staticmethod
    def _draw_int64() -> int:
        return int(th.empty((), dtype=th.int64).random_().item())

    def _permutation(self) -> np.ndarray:
        seed = self._draw_int64()
        g = th.Generator()
        g.manual_seed(seed)
        return th.randperm(self.n, generator=g).numpy()

    def next_indices(self) -> np.ndarray:
        """Row indices (int64, length `batch_size`) of the next expert batch."""
        if self._perm is None or self._pos >= self.batches_per_epoch:
            self._draw_int64()                 # iter(loader): base seed
            self._perm = self._permutation()   # first next(): sampler seed + randperm
            self._pos = 0
        b = self._perm[self._pos * self.batch_size:(self._pos + 1) * self.batch_size]
        self._pos += 1
        return b


_NO_DONE_ORDER: dict = {}


def segment_order(dones: np.ndarray) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """Row order of a rollout's transitions after `pop_trajectories` + flatten.

    Args:
        dones: bool `[T, n_envs]`, `dones[t, e]` = env `e` finished an episode at step `t`.
    Returns:
        `(order, seg_end_step, seg_env)`: `order` are time-major offsets `t*n_envs+e` (int64,
        length `T*n_envs`) -- completed episodes first, sorted by (completion step, env), each
        contributing its steps in time order; then the in-progress fragments by env index
        (`data/wrappers.py:93-111,146-148`). The other two arrays describe the completed
        segments in emission order (used for episode-length bookkeeping).
    """
    dones = np.asarray(dones, dtype=bool)
    T, n = dones.shape
    if not dones.any():
        # no episode ended inside the rollout (15 rounds of 16 at the reference's 1 000-step horizons): only in-progress
        # fragments, by env index, each in time order -- the same array every time, kept per shape
        order = _NO_DONE_ORDER.get((T, n))
        if order is None:
            if len(_NO_DONE_ORDER) > 8:
                _NO_DONE_ORDER.clear()
            order = (np.arange(T, dtype=np.int64)[None, :] * n + np.arange(n, dtype=np.int64)[:, None]).reshape(-1)
            _NO_DONE_ORDER[(T, n)] = order
        empty = np.zeros(0, dtype=np.int64)
        return order.copy(), empty, empty   # (callers own their array)
    t_idx, e_idx = np.nonzero(dones)               # sorted by t then e: exactly the completion order
    # start step of each completed segment = previous done step of the same env + 1
    prev_done = np.full((T, n), -1, dtype=np.int64)
    step_grid = np.where(dones, np.arange(T, dtype=np.int64)[:, None], -1)
    running = np.maximum.accumulate(step_grid, axis=0)      # last done step at or before t
    prev_done[1:] = running[:-1]
    seg_start = prev_done[t_idx, e_idx] + 1
    seg_len = t_idx - seg_start + 1
    last_done = running[-1]                                  # per env, -1 if none
    part_env = np.nonzero(last_done < T - 1)[0]
    part_start = last_done[part_env] + 1
    part_len = T - part_start
    starts = np.concatenate([seg_start, part_start])
    lens = np.concatenate([seg_len, part_len])
    envs = np.concatenate([e_idx, part_env])
    total = int(lens.sum())
    assert total == T * n
    seg_id = np.repeat(np.arange(len(lens)), lens)
    first = np.cumsum(lens) - lens
    within = np.arange(total, dtype=np.int64) - first[seg_id]
    order = (starts[seg_id] + within) * n + envs[seg_id]
    return order.astype(np.int64), t_idx.astype(np.int64), e_idx.astype(np.int64)
