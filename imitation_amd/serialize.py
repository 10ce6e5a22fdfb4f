"""Trajectory on-disk formats of the reference (`data/serialize.py`, `data/huggingface_utils.py`;
SURVEY 8f next row 2): what `demonstrations=` of a trainer is loaded from.

* HuggingFace `datasets` directory written by `serialize.save`: one row per trajectory with the
  columns `obs [T+1, ...]`, `acts [T, ...]`, `infos [T]` (one encoded string per step),
  `terminal`, and `rews [T]` for trajectories with rewards (`huggingface_utils.py:88-133`);
* the legacy `.npz` layout (`serialize.py:50-67`): concatenated `obs / acts / infos / rews`,
  split points `indices`, per-trajectory `terminal`;
* the legacy pickle of a trajectory sequence.

The reference encodes each step's info dict with `jsonpickle`. That package is not part of this
image; for info dicts made of JSON types `jsonpickle.encode` and `json.dumps` produce the same text,
so those are written / read with `json`, and anything else is refused loudly rather than written
in a dialect the reference could not read back. `rollout.rollout(exclude_infos=True)` (the default)
yields exactly the `{}` case.

`load_transitions` goes straight to the flattened device-table input of the trainers.
"""
from __future__ import annotations

import json
import os
import warnings
from typing import Any, List, Mapping, Optional, Sequence

import numpy as np

from imitation_amd import data_types as dt


def _encode_info(info: Mapping[str, Any]) -> str:
    try:
        return json.dumps(dict(info), separators=(", ", ": "))
    except TypeError as e:
        raise TypeError("only info dicts made of JSON types can be written without `jsonpickle` "
                        f"(got {info!r}); drop them with rollout.rollout(exclude_infos=True)") from e


def _decode_info(text: str) -> Any:
    out = json.loads(text)
    if isinstance(out, dict) and any(k.startswith("py/") for k in out):
        raise TypeError("this info entry was written by `jsonpickle` with Python-object tags; it cannot be "
                        "decoded without that package")
    return out


class _LazyInfos(Sequence):
    """`huggingface_utils.py:51-85`: infos are decoded only when somebody looks at them."""

    def __init__(self, encoded: Sequence[str]):
        self._encoded = encoded
        self._cache: dict = {}

    def __len__(self) -> int:
        return len(self._encoded)

    def __getitem__(self, idx):
        if isinstance(idx, slice):
            return [self[i] for i in range(*idx.indices(len(self)))]
        if idx < 0:
            idx += len(self)
        if idx not in self._cache:
            self._cache[idx] = _decode_info(self._encoded[idx])
        return self._cache[idx]


def trajectories_to_dict(trajectories: Sequence[dt.TrajectoryWithRew]) -> dict:
    """`huggingface_utils.py:88-133`."""
    has_rew = [getattr(t, "rews", None) is not None for t in trajectories]
    if any(has_rew) and not all(has_rew):
        raise ValueError("Some trajectories have rewards but not all")
    out = dict(
        obs=[np.asarray(t.obs) for t in trajectories],
        acts=[np.asarray(t.acts) for t in trajectories],
        infos=[[_encode_info(i) for i in (t.infos if t.infos is not None else [{}] * len(t.acts))]
               for t in trajectories],
        terminal=[bool(t.terminal) for t in trajectories],
    )
    if all(has_rew) and len(trajectories):
        out["rews"] = [np.asarray(t.rews) for t in trajectories]
    return out


def save(path, trajectories: Sequence[dt.TrajectoryWithRew]) -> None:
    """`serialize.py:15-24`: HuggingFace `datasets` directory."""
    import datasets

    datasets.Dataset.from_dict(trajectories_to_dict(trajectories)).save_to_disk(str(path))


class TrajectoryDatasetSequence(Sequence):
    """`huggingface_utils.py:11-48`: a `datasets.Dataset` presented as a sequence of trajectories."""

    def __init__(self, dataset):
        # Same decode as the reference: rows come back as Python lists and go through `np.asarray`, so
        # float columns are float64 and integer columns int64 whatever width was stored.
        self._dataset = dataset.with_transform(
            lambda batch: {k: (np.asarray(v) if k != "infos" else v) for k, v in batch.items()})
        self._has_rew = "rews" in dataset.features

    def __len__(self) -> int:
        return len(self._dataset)

    def __getitem__(self, idx):
        if isinstance(idx, slice):
            return [self[i] for i in range(*idx.indices(len(self)))]
        row = self._dataset[int(idx)]
        acts = np.asarray(row["acts"])
        rews = np.asarray(row["rews"]) if self._has_rew else np.zeros(len(acts), dtype=np.float32)
        return dt.TrajectoryWithRew(obs=np.asarray(row["obs"]), acts=acts, rews=rews,
                                    infos=_LazyInfos(list(row["infos"])), terminal=bool(row["terminal"]))


def _from_legacy_mapping(data: Mapping[str, np.ndarray]) -> List[dt.TrajectoryWithRew]:
    """`serialize.py:50-67`. (`indices` holds the split points between trajectories.)"""
    n = len(data["indices"]) + 1
    acts = np.split(data["acts"], data["indices"])
    obs = np.split(data["obs"], data["indices"] + np.arange(n - 1) + 1)
    infos = np.split(data["infos"], data["indices"]) if "infos" in data else [None] * n
    rews = np.split(data["rews"], data["indices"]) if "rews" in data else [np.zeros(len(a), np.float32) for a in acts]
    return [dt.TrajectoryWithRew(obs=o, acts=a, rews=r, infos=i, terminal=bool(t))
            for o, a, r, i, t in zip(obs, acts, rews, infos, data["terminal"])]


def load(path) -> Sequence[dt.TrajectoryWithRew]:
    """`serialize.py:27-73`: directory -> HF dataset; file -> legacy `.npz` or pickle."""
    path = str(path)
    if os.path.isdir(path):
        import datasets

        ds = datasets.load_from_disk(path)
        if not isinstance(ds, datasets.Dataset):
            raise ValueError(f"Expected to load a `datasets.Dataset` but got {type(ds)}")
        return TrajectoryDatasetSequence(ds)
    data = np.load(path, allow_pickle=True)
    if isinstance(data, Mapping):
        warnings.warn("Loading old npz version of Trajectories", DeprecationWarning)
        return _from_legacy_mapping(data)
    if isinstance(data, Sequence):  # pickle of the reference's own Trajectory objects: duck-typed
        warnings.warn("Loading old pickle version of Trajectories", DeprecationWarning)
        return [dt.TrajectoryWithRew(obs=np.asarray(t.obs), acts=np.asarray(t.acts),
                                     rews=np.asarray(getattr(t, "rews", np.zeros(len(t.acts), np.float32))),
                                     infos=getattr(t, "infos", None), terminal=bool(t.terminal)) for t in data]
    raise ValueError("Expected either an .npz file or a pickled sequence of trajectories; "
                     f"got a pickled object of type {type(data).__name__}")


def load_with_rewards(path) -> Sequence[dt.TrajectoryWithRew]:
    """`serialize.py:76-88`."""
    return load(path)


def load_transitions(path, n_max: Optional[int] = None) -> dt.TransitionsWithRew:
    """Flattened `(obs, acts, next_obs, dones, rews)` of a saved demo set -- what the trainers upload
    into the device-resident expert table (`algorithms/base.py:254-263` accepts the same)."""
    trajs = load(path)
    trajs = [trajs[i] for i in range(len(trajs))]
    flat = dt.flatten_trajectories([dt.TrajectoryWithRew(obs=t.obs, acts=t.acts, rews=t.rews, infos=None,
                                                         terminal=t.terminal) for t in trajs])
    if n_max is not None:
        import dataclasses

        flat = dt.TransitionsWithRew(**{f.name: (getattr(flat, f.name)[:n_max] if getattr(flat, f.name) is not None
                                                  else None) for f in dataclasses.fields(flat)})
    return flat
