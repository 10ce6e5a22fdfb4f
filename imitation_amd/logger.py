"""Metric logging with the reference's key scheme (`util/logger.py:71-342`).

Inside `accumulate_means(name)` every `record(k, v)` goes to a per-name sub-logger under
`raw/<name>/k` and is mean-accumulated on the root logger as `mean/<name>/k`; outside it
goes straight to the root. Writers: "stdout", "log", "csv", "json". The base `Logger`
duck-types the SB3 `Logger` surface the trainer and PPO use (`record`, `record_mean`,
`dump`, `name_to_value`).
"""
from __future__ import annotations

import collections
import contextlib
import csv
import datetime
import json
import os
import sys
import tempfile
from typing import Any, Dict, List, Optional, Sequence


class KVWriter:
    def write(self, key_values: Dict[str, Any], key_excluded: Dict[str, Any], step: int = 0) -> None:
        raise NotImplementedError

    def close(self) -> None:
        pass


def _excluded(key_excluded, key, fmt) -> bool:
    ex = key_excluded.get(key)
    return ex is not None and fmt in ex


class HumanWriter(KVWriter):
    def __init__(self, target, name: str = "stdout", max_length: int = 50):
        self.name, self.max_length = name, max_length
        self.own = isinstance(target, str)
        self.file = open(target, "w") if self.own else target

    def write(self, key_values, key_excluded, step=0):
        rows = []
        for k in sorted(key_values):
            if _excluded(key_excluded, k, self.name):
                continue
            v = key_values[k]
            rows.append((k[: self.max_length], f"{v:.4g}" if isinstance(v, float) else str(v)))
        if not rows:
            return
        w0 = max(len(r[0]) for r in rows)
        w1 = max(len(r[1]) for r in rows)
        bar = "-" * (w0 + w1 + 7)
        self.file.write("\n".join([bar] + [f"| {a:<{w0}} | {b:<{w1}} |" for a, b in rows] + [bar]) + "\n")
        self.file.flush()

    def close(self):
        if self.own:
            self.file.close()


class CsvWriter(KVWriter):
    def __init__(self, path: str):
        self.path = path
        self.keys: List[str] = []
        self.rows: List[Dict[str, Any]] = []

    def write(self, key_values, key_excluded, step=0):
        row = {k: v for k, v in key_values.items() if not _excluded(key_excluded, k, "csv")}
        new = [k for k in row if k not in self.keys]
        self.rows.append(row)
        if new:  # header grew: rewrite the file (rare: first few dumps only)
            self.keys.extend(new)
            with open(self.path, "w", newline="") as f:
                w = csv.DictWriter(f, fieldnames=self.keys)
                w.writeheader()
                w.writerows(self.rows)
        else:
            with open(self.path, "a", newline="") as f:
                csv.DictWriter(f, fieldnames=self.keys).writerow(row)


class JsonWriter(KVWriter):
    def __init__(self, path: str):
        self.file = open(path, "w")

    def write(self, key_values, key_excluded, step=0):
        self.file.write(json.dumps({k: (v.item() if hasattr(v, "item") else v) for k, v in key_values.items()}) + "\n")
        self.file.flush()

    def close(self):
        self.file.close()


def make_output_format(fmt: str, log_dir: str, log_suffix: str = "") -> KVWriter:
    os.makedirs(log_dir, exist_ok=True)
    if fmt == "stdout":
        return HumanWriter(sys.stdout, "stdout")
    if fmt == "log":
        return HumanWriter(os.path.join(log_dir, f"log{log_suffix}.txt"), "log")
    if fmt == "csv":
        return CsvWriter(os.path.join(log_dir, f"progress{log_suffix}.csv"))
    if fmt == "json":
        return JsonWriter(os.path.join(log_dir, f"progress{log_suffix}.json"))
    raise ValueError(f"Unknown format specified: {fmt}")


class Logger:
    def __init__(self, folder: Optional[str], output_formats: Sequence[KVWriter]):
        self.name_to_value: Dict[str, Any] = collections.defaultdict(float)
        self.name_to_count: Dict[str, int] = collections.defaultdict(int)
        self.name_to_excluded: Dict[str, Any] = {}
        self.dir = folder
        self.output_formats = list(output_formats)

    def record(self, key: str, value: Any, exclude=None) -> None:
        self.name_to_value[key] = value
        self.name_to_excluded[key] = exclude

    def record_mean(self, key: str, value, exclude=None) -> None:
        if value is None:
            return
        n = self.name_to_count[key]
        self.name_to_value[key] = self.name_to_value[key] * n / (n + 1) + value / (n + 1)
        self.name_to_count[key] = n + 1
        self.name_to_excluded[key] = exclude

    def dump(self, step: int = 0) -> None:
        for w in self.output_formats:
            w.write(self.name_to_value, self.name_to_excluded, step)
        self.name_to_value.clear()
        self.name_to_count.clear()
        self.name_to_excluded.clear()

    def get_dir(self) -> Optional[str]:
        return self.dir

    def log(self, *args, **kwargs) -> None:
        pass

    def close(self) -> None:
        for w in self.output_formats:
            w.close()


class HierarchicalLogger(Logger):
    def __init__(self, default_logger: Logger, format_strs: Sequence[str] = ("stdout", "log", "csv")):
        self.default_logger = default_logger
        self.current_logger: Optional[Logger] = None
        self._cached: Dict[str, Logger] = {}
        self._accumulate_prefixes: List[str] = []
        self._key_prefixes: List[str] = []
        self._name: Optional[str] = None
        self.format_strs = list(format_strs)
        super().__init__(default_logger.dir, [])

    # the maps visible from outside are those of whichever logger is active (`logger.py:155-158`)
    @property
    def name_to_value(self):
        return self._logger.name_to_value

    @name_to_value.setter
    def name_to_value(self, v):
        pass

    @property
    def name_to_count(self):
        return self._logger.name_to_count

    @name_to_count.setter
    def name_to_count(self, v):
        pass

    @property
    def name_to_excluded(self):
        return self._logger.name_to_excluded

    @name_to_excluded.setter
    def name_to_excluded(self, v):
        pass

    @contextlib.contextmanager
    def add_accumulate_prefix(self, prefix: str):
        if self.current_logger is not None:
            raise RuntimeError("Cannot add prefix when accumulate_means context is already active.")
        self._accumulate_prefixes.append(prefix)
        try:
            yield
        finally:
            self._accumulate_prefixes.pop()

    @contextlib.contextmanager
    def add_key_prefix(self, prefix: str):
        if self.current_logger is None:
            raise RuntimeError("Cannot add key prefix when accumulate_means context is not active.")
        self._key_prefixes.append(prefix)
        try:
            yield
        finally:
            self._key_prefixes.pop()

    @contextlib.contextmanager
    def accumulate_means(self, name: str):
        if self.current_logger is not None:
            raise RuntimeError("Nested `accumulate_means` context")
        subdir = os.path.join(*self._accumulate_prefixes, name)
        if subdir not in self._cached:
            folder = os.path.join(self.default_logger.dir, "raw", subdir)
            os.makedirs(folder, exist_ok=True)
            self._cached[subdir] = Logger(folder, [make_output_format(f, folder) for f in self.format_strs])
        self.current_logger, self._name = self._cached[subdir], name
        try:
            yield
        finally:
            self.current_logger, self._name = None, None

    # ---- deferred rounds (imitation_amd extension) --------------------------------------------------
    # A trainer that lets round r's device work finish in the background while round r+1 already
    # records its first values parks round r's still-undumped root entries in a stash and later
    # replays the rest of round r's logging into it -- from INSIDE round r+1's own context -- so that
    # every file receives exactly the rows, in exactly the order, of the strictly sequential schedule.
    def detach_pending(self):
        """Moves the root logger's undumped entries out (the root starts empty again)."""
        d = self.default_logger
        stash = (d.name_to_value, d.name_to_count, d.name_to_excluded)
        d.name_to_value = collections.defaultdict(float)
        d.name_to_count = collections.defaultdict(int)
        d.name_to_excluded = {}
        return stash

    @contextlib.contextmanager
    def replaying(self, stash):
        """Temporarily leaves any active `accumulate_means` context and swaps the stashed root entries
        in; on exit the interrupted context and the current entries are back."""
        d = self.default_logger
        saved_ctx = (self.current_logger, self._name, self._key_prefixes, self._accumulate_prefixes)
        saved_maps = (d.name_to_value, d.name_to_count, d.name_to_excluded)
        self.current_logger, self._name, self._key_prefixes, self._accumulate_prefixes = None, None, [], []
        d.name_to_value, d.name_to_count, d.name_to_excluded = stash
        try:
            yield
        finally:
            d.name_to_value, d.name_to_count, d.name_to_excluded = saved_maps
            self.current_logger, self._name, self._key_prefixes, self._accumulate_prefixes = saved_ctx

    def record(self, key, val, exclude=None):
        if self.current_logger is None:
            self.default_logger.record(key, val, exclude)
            return
        mid = [*self._accumulate_prefixes, self._name, *self._key_prefixes, key]
        self.current_logger.record("/".join(["raw", *mid]), val, exclude)
        self.default_logger.record_mean("/".join(["mean", *mid]), val, exclude)

    @property
    def _logger(self) -> Logger:
        return self.current_logger if self.current_logger is not None else self.default_logger

    def dump(self, step=0):
        self._logger.dump(step)

    def get_dir(self):
        return self._logger.get_dir()

    def record_mean(self, key, val, exclude=None):
        self.default_logger.record_mean(key, val, exclude)

    def close(self):
        self.default_logger.close()
        for lg in self._cached.values():
            lg.close()


def configure(folder: Optional[str] = None, format_strs: Optional[Sequence[str]] = None) -> HierarchicalLogger:
    """`util/logger.py:387-417`."""
    if folder is None:
        stamp = datetime.datetime.now().strftime("imitation-%Y-%m-%d-%H-%M-%S-%f")
        folder = os.path.join(tempfile.gettempdir(), stamp)
    folder = str(folder)
    os.makedirs(folder, exist_ok=True)
    if format_strs is None:
        format_strs = ["stdout", "log", "csv"]
    root = Logger(folder, [make_output_format(f, folder) for f in format_strs])
    return HierarchicalLogger(root, [f for f in format_strs if f != "wandb"])
