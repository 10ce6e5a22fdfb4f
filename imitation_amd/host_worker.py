"""Persistent host helper threads for GIL-free work that runs beside the Python main loop
(NumPy generator fills, the C permutation helpers): no thread creation per use."""
from __future__ import annotations

import queue
import threading
from typing import Callable, Dict


class HostWorker:
    """One daemon thread serving jobs in order. `submit(fn)` returns an event set when `fn()` returned."""

    _named: Dict[str, "HostWorker"] = {}

    def __init__(self, name: str):
        self._queue: "queue.SimpleQueue" = queue.SimpleQueue()
        threading.Thread(target=self._serve, daemon=True, name=f"imitation_amd-{name}").start()

    @classmethod
    def named(cls, name: str) -> "HostWorker":
        """Process-wide worker `name` (created on first use)."""
        if name not in cls._named:
            cls._named[name] = HostWorker(name)
        return cls._named[name]

    def _serve(self) -> None:
        while True:
            job, done = self._queue.get()
            try:
                job()
            finally:
                done.set()

    def submit(self, fn: Callable[[], None]) -> threading.Event:
        done = threading.Event()
        self._queue.put((fn, done))
        return done
