"""PPO generator behind the SB3 `BaseAlgorithm` / `OnPolicyAlgorithm` surface the adversarial
trainer touches (`adversarial/common.py:243-251,414-419`; SURVEY 8b "Generator protocol",
App. A.3-A.7), running on MI355X:

* `collect_rollouts`: per env step ONE fused policy kernel on the `[n_envs, obs]` observation
  tensor (`ia_policy_act`), a 24 KB D2H of the clipped actions, the host VecEnv step, and an
  async H2D of the next observations. Reward relabelling by the discriminator, the time-limit
  value bootstrap, the final value estimate and GAE run once per rollout on the whole
  `[T, n_envs]` tile (identical arithmetic: nothing they depend on changes inside a rollout).
* `train`: host-drawn `np.random.permutation` per epoch (same global-RNG call sequence as SB3's
  `RolloutBuffer.get`), all epochs enqueued back-to-back through `ia_ppo_epoch` with no host
  synchronisation; statistics come back in one D2H at the end.
"""
from __future__ import annotations

import collections
import ctypes as C
import random
import os
import sys
import threading
import time
import warnings
from typing import Any, Dict, Optional

import numpy as np
import torch as th

from imitation_amd import _lib as L
from imitation_amd.host_worker import HostWorker
from imitation_amd import logger as imit_logger
from imitation_amd import policies as pol_mod
from imitation_amd import spaces
from imitation_amd.networks import TransitionTable, require_device
from imitation_amd.wrappers import BufferingWrapper, RewardVecEnvWrapper, step_arrays


def set_random_seed(seed: int) -> None:
    """[SB3 utils.set_random_seed]."""
    random.seed(seed)
    np.random.seed(seed)
    th.manual_seed(seed)


def _schedule(v):
    return v if callable(v) else (lambda _progress, _v=float(v): _v)


def _has_user_step_hook(callback) -> bool:
    """True when user code runs inside the rollout's step loop through `callback.on_step` (anything but the built-in
    no-op, the reward-logging callback of `RewardVecEnvWrapper`, or lists of those)."""
    from imitation_amd.wrappers import WrappedRewardCallback
    if isinstance(callback, _CallbackList):
        return any(_has_user_step_hook(c) for c in callback.cbs)
    return not (type(callback) is _NullCallback or isinstance(callback, WrappedRewardCallback))


class _NullCallback:
    def init_callback(self, model): pass
    def on_training_start(self, l, g): pass
    def on_rollout_start(self): pass
    def on_step(self): return True
    def on_rollout_end(self): pass
    def on_training_end(self): pass
    def update_locals(self, l): pass


class _CallbackList(_NullCallback):
    def __init__(self, cbs):
        self.cbs = list(cbs)

    def init_callback(self, model):
        for c in self.cbs: c.init_callback(model)

    def on_training_start(self, l, g):
        for c in self.cbs: c.on_training_start(l, g)

    def on_rollout_start(self):
        for c in self.cbs: c.on_rollout_start()

    def on_step(self):
        ok = True
        for c in self.cbs: ok = c.on_step() and ok
        return ok

    def on_rollout_end(self):
        for c in self.cbs: c.on_rollout_end()

    def on_training_end(self):
        for c in self.cbs: c.on_training_end()


class _PermutationPredraw:
    """Draws the `n_epochs` minibatch permutations of the NEXT PPO update while the rollout is
    still stepping the environments, on a private copy of the state of NumPy's global generator
    (`ia_host_mt19937_permutations`: the legacy MT19937 Fisher-Yates draw, bit for bit, in C --
    ctypes releases the GIL, so the helper thread does not slow the rollout down).

    SB3 draws them from the global generator at the start of `PPO.train` (`RolloutBuffer.get`), i.e.
    right after the rollout. The copy is only adopted if, at that point, the global generator is
    still in the state it was copied from (nobody -- e.g. an environment -- drew from it during the
    rollout); the global state then jumps to the copy's post-draw state, which is exactly what
    drawing in place would have produced. Otherwise the speculation is dropped and the permutations
    are drawn in place. Either way values and generator state equal the reference's."""

    def __init__(self, n_epochs: int, size: int):
        self.n_epochs, self.size = n_epochs, size
        self._thread: Optional[threading.Event] = None
        self._state0 = None
        self._key: Optional[np.ndarray] = None
        self._pos = C.c_int(0)
        self._rc = 0
        self._out: Optional[np.ndarray] = None
        self._rr = self._rr_armed = None
        self._bg = None

    # The state of NumPy's global generator is read, compared and moved DIRECTLY in the bit generator's `mt19937_state`
    # (`uint32 key[624]; int pos;` at `_bit_generator.ctypes.state_address`): `np.random.get_state()` / `set_state()` cost
    # ~20-45 us each (tuple + array copies, validation) and sat between the last environment step and the PPO launch
    # (`finish`), ahead of the discriminator round's enqueue (`take_randint`) and at the top of the rollout (`start`).
    # That layout is NumPy-private, so it is PROVEN once per process on a private `np.random.MT19937` instance (`_raw_ok`:
    # the bytes at the address equal the public `.state` before and after draws, and a state written through the address
    # reads back through `.state`); when the proof fails -- another NumPy lays the struct out differently -- the same
    # protocol runs on the public `get_state` / `set_state` instead (slower, same draws, same post-states:
    # `tests/test_host_logic.py::test_permutation_predraw_falls_back_to_public_state_api`).
    _STATE_BYTES = 624 * 4 + 4
    _KEY_OFFSET = 0          # of `key[624]` from `state_address` (tests move it to trip the proof)
    _raw_proof: Optional[bool] = None

    @classmethod
    def _raw_ok(cls) -> bool:
        if cls._raw_proof is None:
            ok = False
            try:
                bg = np.random.MT19937(20260930)
                addr = int(bg.ctypes.state_address) + cls._KEY_OFFSET

                def same():
                    st = bg.state["state"]
                    raw = C.string_at(addr, cls._STATE_BYTES)
                    return (raw[:624 * 4] == np.ascontiguousarray(st["key"], dtype=np.uint32).tobytes()
                            and int.from_bytes(raw[624 * 4:], sys.byteorder, signed=True) == int(st["pos"]))

                ok = same()
                bg.random_raw(7)          # pos moves inside the block
                ok = ok and same()
                bg.random_raw(700)        # the block is regenerated
                ok = ok and same()
                other = np.random.MT19937(7).state["state"]
                key = np.ascontiguousarray(other["key"], dtype=np.uint32)
                C.memmove(addr, key.ctypes.data, 624 * 4)
                C.c_int.from_address(addr + 624 * 4).value = 321
                st = bg.state["state"]
                ok = ok and np.array_equal(st["key"], key) and int(st["pos"]) == 321
                want = np.random.MT19937(7)
                want.state = dict(want.state, state=dict(key=key, pos=321))
                ok = ok and bool(np.array_equal(bg.random_raw(5), want.random_raw(5)))
            except Exception:   # (anything missing or different: the public API)
                ok = False
            cls._raw_proof = ok
        return cls._raw_proof

    @classmethod
    def _global_mt(cls):
        """(bit generator, address of its mt19937_state -- or None: go through `get_state` / `set_state`) of NumPy's global
        `RandomState`, or None when it is not MT19937 (no speculation: every draw happens in place)."""
        try:
            bg = getattr(np.random.mtrand._rand, "_bit_generator", None)
        except AttributeError:
            bg = None
        if bg is None:
            try:
                if np.random.get_state()[0] != "MT19937":
                    return None
            except Exception:
                return None
            return None, None
        if type(bg).__name__ != "MT19937":
            return None
        if cls._raw_ok():
            try:
                return bg, int(bg.ctypes.state_address) + cls._KEY_OFFSET
            except AttributeError:
                pass
        return bg, None

    def _raw_state(self, addr) -> bytes:
        """`key[624]` + `pos` of the global generator as bytes (address None: through the public API)."""
        if addr is None:
            st = np.random.get_state()
            return (np.ascontiguousarray(st[1], dtype=np.uint32).tobytes()
                    + int(st[2]).to_bytes(4, sys.byteorder, signed=True))
        return C.string_at(addr, self._STATE_BYTES)

    @staticmethod
    def _move_global(addr, key: np.ndarray, pos: int, bg=None) -> None:
        if addr is None:   # public API: the Gaussian cache of the legacy state stays as it is (no draw here touches it)
            st = np.random.get_state()
            np.random.set_state((st[0], np.array(key, dtype=np.uint32), int(pos)) + tuple(st[3:]))
            return
        lock = getattr(bg, "lock", None)   # (the bit generator's own lock: another thread's draw never sees half a state)
        if lock is not None:
            with lock:
                C.memmove(addr, key.ctypes.data, 624 * 4)
                C.c_int.from_address(addr + 624 * 4).value = int(pos)
        else:
            C.memmove(addr, key.ctypes.data, 624 * 4)
            C.c_int.from_address(addr + 624 * 4).value = int(pos)

    def start(self, out: np.ndarray, randint_spec=None) -> None:
        """`randint_spec = (high, rows, row_len)` (optional): behind the permutations the SAME C call draws, on a copy of the
        generator, `rows` x `np.random.randint(high, size=row_len)` -- the replay ring's index rows of the round's
        discriminator updates (`take_randint`)."""
        assert out.dtype == np.int64 and out.flags.c_contiguous and out.shape == (self.n_epochs, self.size)
        self._rr = self._rr_armed = None   # (rows armed by an earlier round belong to an earlier generator state)
        mt = self._global_mt()
        if mt is None:
            self._thread = None
            return
        self._bg, addr = mt
        raw = self._raw_state(addr)
        self._state0 = raw
        self._key = np.frombuffer(raw, dtype=np.uint32, count=624).copy()
        self._pos = C.c_int(int.from_bytes(raw[624 * 4:], sys.byteorder, signed=True))
        self._out = out
        lib = L.load()
        if randint_spec is not None:
            high, rows, row_len = (int(x) for x in randint_spec)
            rr = dict(spec=(high, rows, row_len), rows=np.empty((rows, row_len), dtype=np.int64),
                      key=np.empty(624, dtype=np.uint32), pos=C.c_int(0))
            self._rr = rr

            def work():
                self._rc = lib.ia_host_mt19937_permutations_then_randint(
                    self._key.ctypes.data, C.byref(self._pos), self.size, self.n_epochs, out.ctypes.data, high, rows, row_len,
                    rr["rows"].ctypes.data, rr["key"].ctypes.data, C.byref(rr["pos"]))
        else:
            def work():
                self._rc = lib.ia_host_mt19937_permutations(self._key.ctypes.data, C.byref(self._pos), self.size,
                                                            self.n_epochs, out.ctypes.data)

        self._thread = HostWorker.named("permutations").submit(work)

    def ready(self) -> bool:
        """The helper has filled `out` (whether the draw is ADOPTED is decided by `finish`)."""
        t = self._thread
        return t is not None and t.is_set() and self._rc == 0

    def finish(self, out: np.ndarray) -> bool:
        """True if `out` now holds the permutations and the global generator has advanced past them."""
        t, self._thread = self._thread, None
        self._rr_armed = None
        if t is None:
            return False
        t.wait()
        mt = self._global_mt()
        if (self._rc != 0 or out is not self._out or mt is None or mt[0] is not self._bg
                or self._raw_state(mt[1]) != self._state0):
            return False
        self._move_global(mt[1], self._key, self._pos.value, mt[0])
        self._rr_armed = self._rr   # (valid while the global generator stays where this call has just put it)
        return True

    def take_randint(self, high: int, rows: int, row_len: int):
        """The `rows` x `np.random.randint(high, size=row_len)` rows drawn behind the adopted permutations -- handed over
        (and the global generator moved behind them) only if the request is the one they were drawn for and the global
        generator is still where `finish` left it; None otherwise (the caller draws in place)."""
        rr, self._rr_armed = getattr(self, "_rr_armed", None), None
        if rr is None or rr["spec"] != (int(high), int(rows), int(row_len)):
            return None
        mt = self._global_mt()
        if mt is None or mt[0] is not self._bg:
            return None
        raw = self._raw_state(mt[1])
        if (int.from_bytes(raw[624 * 4:], sys.byteorder, signed=True) != int(self._pos.value)
                or raw[:624 * 4] != self._key.tobytes()):
            return None
        self._move_global(mt[1], rr["key"], rr["pos"].value, mt[0])
        return rr["rows"]


class _SharedPermutations:
    """Minibatch permutations of the data-parallel PPO update over the all-gathered rollout: every rank
    must use the SAME permutations, so they come from generators seeded identically on all ranks
    (`[shared seed, update index, epoch]`), one per epoch, drawn concurrently by host threads inside
    `ia_host_mt19937_seeded_permutations` (no GIL) while the rollout runs."""

    def __init__(self, seed: int, n_epochs: int, size: int):
        self.seed, self.n_epochs, self.size = int(seed), n_epochs, size
        self.update = 0
        self._threads = []

    def start(self, out: np.ndarray) -> None:
        assert out.dtype == np.int64 and out.flags.c_contiguous and out.shape == (self.n_epochs, self.size)
        lib = L.load()
        seeds = np.array([[self.seed, self.update, e] for e in range(self.n_epochs)], dtype=np.uint32)

        def work():  # one ctypes call (no GIL); the library fans out one host thread per epoch
            self._rc = lib.ia_host_mt19937_seeded_permutations(seeds.ctypes.data, 3, self.size, self.n_epochs,
                                                                 out.ctypes.data)

        self._rc = 0
        self._threads = [HostWorker.named("permutations").submit(work)]
        self.update += 1

    def finish(self) -> None:
        for t in self._threads:
            t.wait()
        self._threads = []
        L.check(self._rc, "ia_host_mt19937_seeded_permutations")


class _TrainRecord:
    """Everything `finalize_train` reads about ONE `PPO.train`, kept apart from the live buffers so it can
    be read back after the next rollout -- or even the next PPO update -- has been enqueued: loss
    statistics (written by the kernels directly), value / return tiles, log_std, the error word.
    Read-back goes through a side stream, so it never queues behind a running PPO update."""

    def __init__(self, stats_like: th.Tensor, val_like: th.Tensor, log_std: Optional[th.Tensor]):
        dev = stats_like.device
        self.stats = th.zeros_like(stats_like)
        self.val, self.ret = th.empty_like(val_like), th.empty_like(val_like)
        self.log_std = None if log_std is None else th.empty_like(log_std)
        self.err = th.zeros(1, dtype=th.int32, device=dev)
        pin = lambda t: th.empty(t.shape, dtype=t.dtype).pin_memory()
        self.h_stats, self.h_val, self.h_ret, self.h_err = pin(self.stats), pin(self.val), pin(self.ret), pin(self.err)
        self.h_log_std = None if log_std is None else pin(self.log_std)
        self.ready = th.cuda.Event()
        self.clip_range = 0.0
        self.n_updates = 0

    def read_back(self, stream: th.cuda.Stream) -> None:
        with th.cuda.stream(stream):
            stream.wait_event(self.ready)
            self.h_stats.copy_(self.stats, non_blocking=True)
            self.h_val.copy_(self.val, non_blocking=True)
            self.h_ret.copy_(self.ret, non_blocking=True)
            self.h_err.copy_(self.err, non_blocking=True)
            if self.log_std is not None:
                self.h_log_std.copy_(self.log_std, non_blocking=True)
        stream.synchronize()


class RolloutBuffer:
    """Device-resident `[T, n_envs, ...]` rollout tile (time-major, fp32) + the pinned host
    staging the env loop writes into. Properties named like SB3's `RolloutBuffer` fields return
    host copies in SB3's layout (post-`get()`: env-major flattened; `rewards`: `[T, n]`)."""

    def __init__(self, T: int, n: int, obs_dim: int, act_width: int, device, obs_u8: bool = False):
        """`obs_u8` (image observations, uint8 frames): the pinned tiles the env loop writes (`h_obs`, `h_next`) and their twins
        in the transfer block are uint8 -- the frames as the environment hands them over --; the fp32 device tiles every consumer
        reads (`obs`, `next_fixed`: relabelling, replay rows, PPO's row gathers) are filled from them by two conversion kernels
        behind the one copy. At 64 envs x 16 steps of 4 x 84 x 84 frames: 60 MB over PCIe instead of 240 MB, and no uint8 ->
        fp32 pass over 3.6 M elements per env step on the host (`profiles/r06_image_gail.md`)."""
        self.buffer_size, self.n_envs, self.obs_dim, self.act_width = T, n, obs_dim, act_width
        self.obs_u8 = bool(obs_u8)
        f = lambda *s: th.zeros(*s, device=device)
        # Everything the env loop produces on the host goes to the device in ONE copy after the last step:
        # the host-written tiles are views into one pinned block, their device twins views into one device
        # block with the same layout (seven separate copies cost ~10 us of copy-engine latency each, on the
        # critical path between the last env step and the PPO update).
        frame_dt = th.uint8 if self.obs_u8 else th.float32
        # (the two frame tiles first, the small tiles contiguous behind them: `upload_host_tiles(from_step=)` takes the frames'
        #  tails and the small tiles as three copies)
        spec = [("_obs_x" if self.obs_u8 else "obs", (T + 1, n, obs_dim), frame_dt),
                ("_next_x" if self.obs_u8 else "next_fixed", (T, n, obs_dim), frame_dt), ("clipped", (T, n, act_width), th.float32),
                ("starts", (T, n), th.float32), ("last_done", (n,), th.float32), ("dones", (T, n), th.uint8),
                ("trunc", (T, n), th.uint8)]
        nbytes = sum(-(-int(np.prod(shape)) * th.empty(0, dtype=dt).element_size() // 256) * 256 for _, shape, dt in spec)
        self._dev_block = th.zeros(nbytes, dtype=th.uint8, device=device)
        self._host_block = th.zeros(nbytes, dtype=th.uint8).pin_memory()
        off = 0
        host_names = dict(obs="h_obs", clipped="h_clip", next_fixed="h_next", starts="h_starts",
                          last_done="h_last_done", dones="h_dones", trunc="h_trunc", _obs_x="h_obs", _next_x="h_next")
        for name, shape, dt in spec:
            size = int(np.prod(shape)) * th.empty(0, dtype=dt).element_size()
            setattr(self, name, self._dev_block[off:off + size].view(dt).view(*shape))
            setattr(self, host_names[name], self._host_block[off:off + size].view(dt).view(*shape))
            if name == "clipped":
                self._small_off = off   # (first byte of the small tiles)
            off += -(-size // 256) * 256
        if self.obs_u8:   # (one slice more than T: `ia_ppo_update*`'s 16-byte row pieces, include/imitation_hip.h)
            self.obs, self.next_fixed = f(T + 1, n, obs_dim), f(T, n, obs_dim)
        self.acts = f(T, n, act_width)
        self.rew, self.val, self.logp, self.adv, self.ret = (f(T, n) for _ in range(5))
        self.term_val, self.last_val = f(T, n), f(n)
        self.noise = f(n, max(act_width, 1))
        pin = lambda *s, dtype=th.float32: th.zeros(*s, dtype=dtype).pin_memory()
        self.h_rew = pin(T, n)
        self.h_noise = pin(n, max(act_width, 1))
        self.h_noise_tile = None   # [T, n, act_width] pinned: a whole rollout's policy noise in one draw (PPO.predraw_noise)
        self.full = False

    def ensure_host_sampling_tiles(self, n_actions: int) -> None:
        """Pinned tiles of the host-sampled Discrete rollout step (`ActorCriticPolicy.make_multinomial_step`)."""
        if getattr(self, "h_logits", None) is None or self.h_logits.shape[1] != n_actions:
            self.h_logits = th.zeros(self.n_envs, n_actions).pin_memory()
            self.h_logp = th.zeros(self.buffer_size, self.n_envs).pin_memory()

    def upload_host_tiles(self, from_step: int = 0) -> None:
        """One H2D copy of everything the env loop wrote on the host (current stream); uint8 frames -> the fp32 tiles.
        `from_step` > 0: the frame rows of steps < from_step are on the device already (`upload_host_steps`): the frames'
        tails and the small tiles, three copies."""
        ox, nx = (self._obs_x, self._next_x) if self.obs_u8 else (self.obs, self.next_fixed)
        if from_step <= 0:
            self._dev_block.copy_(self._host_block, non_blocking=True)
        else:
            ox[from_step:].copy_(self.h_obs[from_step:], non_blocking=True)
            nx[from_step:].copy_(self.h_next[from_step:], non_blocking=True)
            self._dev_block[self._small_off:].copy_(self._host_block[self._small_off:], non_blocking=True)
        if self.obs_u8:
            self.obs[from_step:].copy_(self._obs_x[from_step:])
            self.next_fixed[from_step:].copy_(self._next_x[from_step:])

    def upload_host_steps(self, t1: int) -> None:
        """Steps [0, t1) of what a reward net reads (observations, next observations, clipped actions, dones) ahead of the whole
        upload -- the rows are final once their steps are done (current stream)."""
        ox, nx = (self._obs_x, self._next_x) if self.obs_u8 else (self.obs, self.next_fixed)
        ox[:t1].copy_(self.h_obs[:t1], non_blocking=True)
        nx[:t1].copy_(self.h_next[:t1], non_blocking=True)
        self.clipped[:t1].copy_(self.h_clip[:t1], non_blocking=True)
        self.dones[:t1].copy_(self.h_dones[:t1], non_blocking=True)
        if self.obs_u8:
            self.obs[:t1].copy_(self._obs_x[:t1])
            self.next_fixed[:t1].copy_(self._next_x[:t1])

    def reset(self) -> None:
        self.full = False

    def _env_major(self, t: th.Tensor) -> np.ndarray:
        a = t.detach().cpu().numpy()
        return a.swapaxes(0, 1).reshape(self.buffer_size * self.n_envs, *a.shape[2:])

    observations = property(lambda s: s._env_major(s.obs[: s.buffer_size]))
    actions = property(lambda s: s._env_major(s.acts))
    values = property(lambda s: s._env_major(s.val))
    log_probs = property(lambda s: s._env_major(s.logp))
    advantages = property(lambda s: s._env_major(s.adv))
    returns = property(lambda s: s._env_major(s.ret))
    rewards = property(lambda s: s.rew.detach().cpu().numpy())
    episode_starts = property(lambda s: s.starts.detach().cpu().numpy())


class OnPolicyAlgorithm:
    """Marker base so `isinstance(gen_algo, OnPolicyAlgorithm)` (`common.py:250`) keeps its meaning."""


class PPO(OnPolicyAlgorithm):
    def __init__(self, policy, env, learning_rate=3e-4, n_steps: int = 2048, batch_size: int = 64, n_epochs: int = 10,
                 gamma: float = 0.99, gae_lambda: float = 0.95, clip_range=0.2, clip_range_vf=None,
                 normalize_advantage: bool = True, ent_coef: float = 0.0, vf_coef: float = 0.5,
                 max_grad_norm: float = 0.5, use_sde: bool = False, sde_sample_freq: int = -1,
                 target_kl: Optional[float] = None, stats_window_size: int = 100, tensorboard_log=None,
                 policy_kwargs: Optional[Dict[str, Any]] = None, verbose: int = 0, seed: Optional[int] = None,
                 device="auto", _init_setup_model: bool = True):
        if use_sde or clip_range_vf is not None or target_kl is not None:
            raise NotImplementedError("gSDE / value clipping / target_kl are off in every reference config")
        if isinstance(policy, str):
            from imitation_amd.cnn_policy import ActorCriticCnnPolicy
            policy = {"MlpPolicy": pol_mod.ActorCriticPolicy, "CnnPolicy": ActorCriticCnnPolicy}[policy]
        self.policy_class = policy
        self.policy_kwargs = dict(policy_kwargs or {})
        self.device = th.device("cuda" if device == "auto" else device)
        self.learning_rate, self.n_steps, self.batch_size, self.n_epochs = learning_rate, n_steps, batch_size, n_epochs
        self.gamma, self.gae_lambda, self.clip_range = gamma, gae_lambda, clip_range
        self.normalize_advantage, self.ent_coef, self.vf_coef = normalize_advantage, ent_coef, vf_coef
        self.max_grad_norm, self.seed, self.verbose = max_grad_norm, seed, verbose
        self.num_timesteps = 0
        self._total_timesteps = 0
        self._num_timesteps_at_start = 0
        self._n_updates = 0
        self._current_progress_remaining = 1.0
        self._last_obs = None
        self._last_episode_starts = None
        self._stats_window_size = stats_window_size
        self.ep_info_buffer = None
        self._logger: Optional[imit_logger.Logger] = None
        self._custom_logger = False
        self.start_time = 0
        self.env = env
        self.policy: Optional[pol_mod.ActorCriticPolicy] = None
        self.rollout_buffer: Optional[RolloutBuffer] = None
        if env is not None:
            self.observation_space, self.action_space, self.n_envs = env.observation_space, env.action_space, env.num_envs
        if normalize_advantage:
            assert batch_size > 1, "`batch_size` must be greater than 1 (advantage normalisation)"
        if _init_setup_model:
            self._setup_model()

    # ---- SB3 BaseAlgorithm surface ----------------------------------------------------------
    def _setup_model(self) -> None:
        self.lr_schedule = _schedule(self.learning_rate)
        self.clip_range = _schedule(self.clip_range)
        if self.seed is not None:
            set_random_seed(self.seed)
            self.action_space.seed(self.seed)
            if self.env is not None:
                self.env.seed(self.seed)
        self.policy = self.policy_class(self.observation_space, self.action_space, self.lr_schedule,
                                        **self.policy_kwargs).to(self.device)
        self._alloc()

    def _alloc(self) -> None:
        if self.device.type != "cuda":
            return
        p = self.policy
        osp_ = self.observation_space   # image observations (uint8 frames, an image policy): uint8 transport tiles
        frames_u8 = (getattr(p, "takes_uint8_frames", False) and getattr(osp_, "dtype", None) == np.uint8)
        self.rollout_buffer = RolloutBuffer(self.n_steps, self.n_envs, p.obs_dim, 1 if p.discrete else p.act_dim,
                                            self.device, obs_u8=frames_u8)
        total = self.n_steps * self.n_envs
        self._n_mb = -(-total // self.batch_size)
        # policies outside the fused kernels' shapes (`general_policy.GeneralTowers`) run their own minibatch loop
        # (whole minibatches per epoch: room for every epoch's gathered rows, `ia_ppo_epochs` runs them as one sequence)
        # (... up to 4 M gathered rows: beyond that the per-epoch calls keep the workspace at one epoch's size)
        self._ppo_ws_epochs = (self.n_epochs if (p.fused and total % self.batch_size == 0
                                                 and self.n_epochs * total <= (1 << 22)) else 1)
        self.epochs_one_call = True   # (tuning / tests: False = one `ia_ppo_epoch` call per epoch)
        self._ppo_ws = th.empty(int(L.load().ia_ppo_ws_floats(C.byref(p.desc), min(self.batch_size, total),
                                                              self._ppo_ws_epochs * total)),
                                device=self.device) if p.fused else None
        self._perm_host = th.zeros(self.n_epochs, total, dtype=th.int64).pin_memory()
        self._perm_dev = th.zeros(self.n_epochs, total, dtype=th.int64, device=self.device)
        self._perm_np = self._perm_host.numpy()
        self._perm_uploaded = None     # event of an upload made beside the rollout (`_upload_permutations_early`)
        self._predraw = _PermutationPredraw(self.n_epochs, total)
        self.update_events = None  # optional (start, end) torch events around the persistent update launch
        self._stats_dev = th.zeros(self.n_epochs, self._n_mb, 8, device=self.device)
        n_upd = int(L.load().ia_ppo_update_ws_floats(C.byref(p.desc), min(self.batch_size, total))) if p.fused else 0
        # persistent whole-update kernel (hidden = 32); None -> one ia_ppo_epoch call per epoch
        self._upd_ws = th.zeros(n_upd, device=self.device) if n_upd > 0 else None
        self.dp = None  # set by the trainer for data-parallel runs (imitation_amd.distributed.DataParallel)
        # When True, `train()` only ENQUEUES the update and returns; the caller overlaps other GPU
        # work with it and later calls `finalize_train()` (one D2H of the statistics + logging).
        self.defer_train_stats = False
        self._pending_train = None
        self._h_upd_err, self._upd_err_pending = None, False
        # called (if set) after the last env step of a rollout and before the reward relabelling -- the
        # first point of a rollout that depends on the discriminator (see AdversarialTrainer.train)
        self.before_relabel = None
        # `enqueue_first`: inside `learn`, enqueue the PPO update BEFORE the iteration's host-side logging
        # (then call `after_enqueue`, then log) so that nothing on the host delays the launch; the records
        # are made in the same order as always. Set by the pipelined adversarial trainer.
        self.enqueue_first = False
        self.after_enqueue = None
        # ... and, ahead of both (right behind the update's launch): `after_train_enqueued(last_iteration)` -- the pipelined
        # adversarial trainer enqueues the round's discriminator updates there, before any host-side wait of this iteration
        self.after_train_enqueued = None
        # hook of the adversarial trainer (`adversarial/common.py: _replay_rows_spec`): (high, rows, row_len) of the
        # `np.random.randint` rows its discriminator round will draw behind this update's permutations (argument: rows of
        # the rollout) -- the permutation helper's C call draws them too, on a copy of the generator
        self.randint_spec_for_round = None
        # hook of the pipelined adversarial trainer (`adversarial/common.py: _round_predraw`), called right behind the
        # rollout's noise draw -- where the host waits for the previous update anyway: what the coming discriminator round
        # will take from torch's global CPU generator (expert index rows, interpolation weights) is drawn THERE, in the order
        # the round itself draws it, when the whole rollout's noise has been taken in one draw (nothing else reads that
        # generator until the round: the condition `predraw_noise` states). Argument: whether this is the last rollout of
        # the running `learn()` -- only then is the round's first draw the next thing the sequential schedule takes
        self.after_noise_predraw = None
        # relabelling + reward copy + GAE behind a rollout's last step as one host call where the reward net allows it
        # (`_rollout_tail_args`; False: the general path, call by call -- tests compare)
        self.rollout_tail_one_call = True
        # `nn.Module` reward nets relabelled in bulk behind the rollout: the rows of the rollout's first three quarters are
        # uploaded and relabelled WHILE the host steps the environments through the last quarter (rows are independent; the
        # reward net is final once the previous round's updates are through -- `before_relabel` -- and those run behind the
        # first part of the rollout). False: everything behind the last step (tests compare)
        self.relabel_early = True
        self._tail_args = None
        self._post_enqueue_work = []
        self._act_stream = None
        self.rollout_post_ahead = True   # (tuning / A-B: False posts a mailbox step only at the top of its own iteration)
        self.predraw_noise = os.environ.get("IA_PREDRAW_NOISE", "1") != "0"   # see `collect_rollouts`
        # the rollout's act steps as one resident launch driven through flags in pinned host memory (`collect_rollouts`)
        self.rollout_mailbox = os.environ.get("IA_ROLLOUT_MAILBOX", "1") != "0"
        self.rollout_mailbox_timeout_s = 120.0   # how long the resident kernel waits for ONE environment step before it
                                                 # leaves (the rollout then continues with per-step launches)
        # ... and how long when user code runs inside the step loop (a `callback` with its own `on_step`, an arbitrary
        # host `reward_fn`): a DEVICE-WIDE wait made there (torch.cuda.synchronize(), empty_cache(), the caching
        # allocator's out-of-memory retry) cannot return before the resident kernel has left, i.e. stalls for this long
        # (INTEGRATION.md "The rollout mailbox and device-wide waits"). After ANY device-side time-out the mailbox stays
        # off for the rest of training.
        self.rollout_mailbox_hooked_timeout_s = 5.0
        self.rollout_window_ms = None
        self.rollout_profile = None
        self._dp_obs = th.zeros(min(self.batch_size, total), p.obs_dim, device=self.device)
        self._dp_ws_pre = None
        self._dpg = None   # lazily built state of the global-minibatch data-parallel update
        self._records = None
        self._rec_i = 0
        self._fin_stream = None
        self.dp_batch_moments = True  # False: exchange the feature-norm moments once per minibatch (tests)
        self.dp_global_minibatch = True  # False: per-minibatch gradient all-reduce (`_train_data_parallel`)
        # Data-parallel update on the global minibatch, ROWS SHARDED over the ranks (default): each rank's persistent kernel
        # takes its `batch_size` rows of every world x batch_size-row minibatch and the ranks exchange one 14 KB record per
        # optimiser step inside the kernels through peer-mapped memory (`distributed.PeerExchange`, `ia_ppo_update_sharded`).
        # False -- or a failed peer handshake -- runs the whole global minibatch redundantly on every rank instead.
        self.dp_row_sharded = os.environ.get("IA_DP_ROW_SHARDED", "1") != "0"
        # Which data-parallel form of the persistent update runs: "sharded" (default: each rank its rows of the global
        # minibatch, one record per optimiser step exchanged inside the kernels; a failed peer handshake -> "replicated" on
        # every rank, `dp_handshake_failed`), "replicated" (every rank the whole global minibatch, no per-step exchange) or
        # "auto" (opt-in; `bench.py` asks for it): both are TIMED on the node during the first updates of a run --
        # alternating, the first launch of each form discarded -- and the form whose slowest rank is faster is kept on all
        # ranks (one all-gather of two numbers); `dp_choice` records both timings and the verdict. The two forms sum the
        # gradient slabs in different orders (same values, last bits differ), so a choice that depends on measured time
        # is NOT the default: a seeded run on a given node always takes the same form. `IA_DP_ROW_SHARDED=0` still means
        # "replicated".
        self.dp_update_form = os.environ.get("IA_DP_UPDATE_FORM", "sharded") if self.dp_row_sharded else "replicated"
        self.dp_choice = None
        self.dp_handshake_failed = False
        self.dp_exchange_timeout_s = 30.0   # a rank waits this long for a peer's record of ONE optimiser step

    @property
    def logger(self):
        return self._logger

    def set_logger(self, logger) -> None:
        self._logger = logger
        self._custom_logger = True

    def get_env(self):
        return self.env

    def set_env(self, env, force_reset: bool = True) -> None:
        if env.num_envs != self.n_envs:
            raise ValueError("The number of environments to be set is different from the number of environments in "
                             f"the model: ({env.num_envs} != {self.n_envs})")
        if env.observation_space != self.observation_space:
            raise ValueError(f"Observation spaces do not match: {self.observation_space} != {env.observation_space}")
        if env.action_space != self.action_space:
            raise ValueError(f"Action spaces do not match: {self.action_space} != {env.action_space}")
        if force_reset:
            self._last_obs = None
        self.env = env

    def predict(self, observation, state=None, episode_start=None, deterministic: bool = False):
        return self.policy.predict(observation, state, episode_start, deterministic)

    def save(self, path) -> None:
        """[SB3 BaseAlgorithm.save]: a zip with SB3's member layout (`checkpoint.save_policy_zip`)."""
        from imitation_amd import checkpoint

        path = str(path)
        checkpoint.save_policy_zip(path if path.endswith(".zip") else path + ".zip", self)

    def load_parameters(self, path, load_optimizer: bool = True):
        """Policy (+ Adam state) from an SB3-layout zip into this model; returns its `data` dict."""
        from imitation_amd import checkpoint

        return checkpoint.load_policy_zip(path, self, load_optimizer=load_optimizer)

    # ---- learn loop (App. A.3) ---------------------------------------------------------------
    def _init_callback(self, callback):
        if callback is None:
            callback = _NullCallback()
        elif isinstance(callback, (list, tuple)):
            callback = _CallbackList(callback)
        callback.init_callback(self)
        return callback

    def _setup_learn(self, total_timesteps: int, callback, reset_num_timesteps: bool):
        self.start_time = time.time_ns()
        if self.ep_info_buffer is None or reset_num_timesteps:
            self.ep_info_buffer = collections.deque(maxlen=self._stats_window_size)
        if reset_num_timesteps:
            self.num_timesteps = 0
        else:
            total_timesteps += self.num_timesteps
        self._total_timesteps = total_timesteps
        self._num_timesteps_at_start = self.num_timesteps
        if reset_num_timesteps or self._last_obs is None:
            self._last_obs = self.env.reset()
            self._last_episode_starts = np.ones((self.env.num_envs,), dtype=bool)
        if not self._custom_logger and self._logger is None:
            self._logger = imit_logger.Logger(None, [])
        return total_timesteps, self._init_callback(callback)

    def learn(self, total_timesteps: int, callback=None, log_interval: int = 1, tb_log_name: str = "PPO",
              reset_num_timesteps: bool = True, progress_bar: bool = False):
        require_device(self.device)
        iteration = 0
        total_timesteps, callback = self._setup_learn(total_timesteps, callback, reset_num_timesteps)
        callback.on_training_start(locals(), globals())
        while self.num_timesteps < total_timesteps:
            if not self.collect_rollouts(self.env, callback, self.rollout_buffer, self.n_steps):
                break
            iteration += 1
            self._current_progress_remaining = 1.0 - float(self.num_timesteps) / float(total_timesteps)
            lr_later = None
            if self.enqueue_first:
                lr_later = self.train(record_lr=False)
                if self.after_train_enqueued is not None:
                    self.after_train_enqueued(self.num_timesteps >= total_timesteps)
                for work in self._post_enqueue_work:
                    work()
                self._post_enqueue_work = []
                if self.after_enqueue is not None:
                    self.after_enqueue()
            if log_interval is not None and iteration % log_interval == 0:
                elapsed = max((time.time_ns() - self.start_time) / 1e9, sys.float_info.epsilon)
                fps = int((self.num_timesteps - self._num_timesteps_at_start) / elapsed)
                self.logger.record("time/iterations", iteration, exclude="tensorboard")
                if len(self.ep_info_buffer) > 0 and len(self.ep_info_buffer[0]) > 0:
                    self.logger.record("rollout/ep_rew_mean", float(np.mean([e["r"] for e in self.ep_info_buffer])))
                    self.logger.record("rollout/ep_len_mean", float(np.mean([e["l"] for e in self.ep_info_buffer])))
                self.logger.record("time/fps", fps)
                self.logger.record("time/time_elapsed", int(elapsed), exclude="tensorboard")
                self.logger.record("time/total_timesteps", self.num_timesteps, exclude="tensorboard")
                self.logger.dump(step=self.num_timesteps)
            if lr_later is None:
                self.train()
            else:  # what `train()` records before it enqueues anything
                self.logger.record("train/learning_rate", lr_later)
        callback.on_training_end()
        return self

    # ---- rollout collection (App. A.4) --------------------------------------------------------
    @staticmethod
    def _unwrap(env):
        """(reward_wrapper | None, buffering | None, innermost env to step)."""
        rw = env if isinstance(env, RewardVecEnvWrapper) else None
        inner = env.venv if rw is not None else env
        bw = inner if isinstance(inner, BufferingWrapper) else None
        base = inner.venv if bw is not None else inner
        return rw, bw, base

    def collect_rollouts(self, env, callback, rb: RolloutBuffer, n_rollout_steps: int) -> bool:
        assert self._last_obs is not None, "No previous observation was provided"
        from imitation_amd.reward_nets import RewardNet  # local import: avoid a cycle
        pol = self.policy
        pol.set_training_mode(False)
        rb.reset()
        callback.on_rollout_start()
        rw, bw, base = self._unwrap(env)
        fused_net = None
        if rw is not None:
            owner = getattr(rw.reward_fn, "__self__", None)
            if isinstance(owner, RewardNet) and getattr(rw.reward_fn, "__name__", "") == "predict_processed":
                fused_net = owner
        # an `nn.Module` reward net (operator boundary) whose prediction does not depend on the call sequence (no
        # NormalizedRewardNet statistics update per call, no user override of predict_processed / predict / predict_th
        # anywhere in the wrapper chain): relabel the whole tile behind the last step, in chunks, instead of one host
        # round trip per environment step; anything else is called once per step like `RewardVecEnvWrapper` does
        module_net = None
        if rw is not None and fused_net is None:
            from imitation_amd import modules as _modules
            if (isinstance(owner, _modules.RewardNet) and getattr(rw.reward_fn, "__name__", "") == "predict_processed"
                    and _modules.bulk_relabel_ok(owner)):
                module_net = owner
        T, n = rb.buffer_size, rb.n_envs
        assert n_rollout_steps == T
        if hasattr(base, "set_lookahead"):  # host env that can draw its noise one rollout ahead (SyntheticVecEnv)
            base.set_lookahead(T)
        if self._dp_global():
            self._dpg["perms"].start(self._dpg["perm_np"])   # shared across ranks; consumed by the next train()
        else:
            self._perm_uploaded = None
            spec = self.randint_spec_for_round(T * n) if self.randint_spec_for_round is not None else None
            self._predraw.start(self._perm_np, randint_spec=spec)  # consumed by the `train()` that follows
        stream = th.cuda.current_stream()
        rb.h_obs[0].copy_(th.as_tensor(np.asarray(self._last_obs)).reshape(n, -1))
        starts = np.asarray(self._last_episode_starts, dtype=bool)
        h_rew_np, h_dones_np, h_trunc_np = rb.h_rew.numpy(), rb.h_dones.numpy(), rb.h_trunc.numpy()
        h_next_np, h_obs_np, h_starts_np = rb.h_next.numpy(), rb.h_obs.numpy(), rb.h_starts.numpy()
        per_step_rews = []
        prof = self.rollout_profile  # optional dict of per-section host seconds (tools/rollout_sections.py)
        tick = time.perf_counter
        # The per-step exchange with the host env workers goes through pinned (device-mapped) host
        # memory directly: the act kernel reads this step's observations and noise from it and writes
        # the clipped actions into it, so a step is ONE launch + ONE wait -- no copy-engine hops in the
        # latency chain. The device copies of the observation / clipped-action tiles (needed from the
        # reward relabelling on) are filled by two bulk copies after the last step.
        # The act kernels run on their own HIGH-PRIORITY stream: while the host steps the environments the
        # device works through the previous round's discriminator updates (pipelined rounds), whose GEMMs
        # fill every CU; with equal priority each 15 us act kernel queued behind them for ~60 us.
        if self._act_stream is None:
            self._act_stream = L.side_stream(self.device, "act", priority=-1)
        act_stream = self._act_stream
        act_stream.wait_stream(stream)     # parameters / statistics written by the previous update
        host_sampling = pol.samples_on_host  # Discrete head on the reference's torch.multinomial stream
        predrawn = False
        mailbox = None
        hooked = _has_user_step_hook(callback) or (rw is not None and fused_net is None and module_net is None)
        mb_timeout = min(self.rollout_mailbox_timeout_s, self.rollout_mailbox_hooked_timeout_s) if hooked \
            else self.rollout_mailbox_timeout_s
        with th.cuda.stream(act_stream):
            if host_sampling:
                rb.ensure_host_sampling_tiles(pol.act_dim)
                act_step = pol.make_multinomial_step(rb.h_obs, rb.h_logits, rb.h_clip, rb.val, rb.h_logp)
                if self.rollout_mailbox and hasattr(pol, "make_multinomial_mailbox"):
                    mailbox = pol.make_multinomial_mailbox(rb.h_obs, rb.h_logits, rb.h_clip, rb.val, rb.h_logp, T,
                                                           timeout_s=mb_timeout)
            else:
                # The rollout's T Gaussian noise tiles in ONE draw at its start, when that is the same stream:
                # `normal_()` on a contiguous [T, n, A] tensor consumes torch's generator exactly as T draws of
                # [n, A] do if n * A is a multiple of 16 (its vectorised path fills whole 16-element blocks;
                # `tests/test_host_logic.py`) and nothing else reads that generator between the steps (no env of
                # the path does; set `predraw_noise = False` for in-process envs that use torch's global RNG).
                # The draw lands where the host waits for the previous PPO update anyway; 15 x 13 us per round
                # leave the step loop.
                width = rb.h_noise.shape[1]
                noise_tile = rb.h_noise
                if self.predraw_noise and not pol.discrete and (n * width) % 16 == 0:
                    if rb.h_noise_tile is None:
                        rb.h_noise_tile = th.zeros(T, n, width).pin_memory()
                    noise_tile = rb.h_noise_tile
                    pol.draw_noise_into(noise_tile)
                    if self.after_noise_predraw is not None and not hooked:   # (no user code inside the step loop)
                        # is this the LAST rollout of the running `learn()`? (a round of k > 1 rollouts draws the noise of
                        # rollouts 2..k from the same generator before the discriminator round draws anything)
                        self.after_noise_predraw(self.num_timesteps + n * T >= self._total_timesteps)
                predrawn = noise_tile.dim() == 3
                act_step = pol.make_act_step(rb.h_obs, noise_tile, rb.acts, rb.h_clip, rb.val, rb.logp)  # eval mode: no norm update
                # ONE resident launch for the rollout's T act steps, driven through flags in pinned host memory: a step
                # then costs the host a flag write and a poll instead of a launch and a stream synchronisation
                # (`ActorCriticPolicy.make_rollout_mailbox`; None: shapes its kernel does not cover)
                if self.rollout_mailbox and hasattr(pol, "make_rollout_mailbox"):
                    mailbox = pol.make_rollout_mailbox(rb.h_obs, noise_tile, rb.acts, rb.h_clip, rb.val, rb.logp, T,
                                                       timeout_s=mb_timeout, last_val=rb.last_val)
        h_clip_np = rb.h_clip.numpy()
        try:
            return self._rollout_steps(env, callback, rb, T, n, pol, rw, bw, base, fused_net, module_net, act_step, mailbox,
                                       act_stream, host_sampling, predrawn, starts, per_step_rews, prof, tick, h_clip_np,
                                       h_rew_np, h_dones_np, h_trunc_np, h_next_np, h_obs_np, h_starts_np, stream)
        finally:
            if mailbox is not None:
                mailbox[2]()

    def _rollout_tail_args(self, fused_net, rb, T: int, n: int):
        """The argument block of `ia_rollout_tail` for this rollout buffer and reward net (built once, re-used while nothing
        it points at has moved), or None when the reward is not one fused-shape stack (`RewardNet.rollout_tail_plan`)."""
        plan = fused_net.rollout_tail_plan()
        if plan is None:
            return None
        basic, out_act = plan
        mlp, nrm = basic.mlp, basic.mlp.norm
        ws = mlp.workspace(T * n, "rollout")
        pws = mlp._predict_ws()
        key = (id(basic), out_act, T, n, basic.flags, ws["X"].data_ptr(), pws.data_ptr(), mlp.flat.data_ptr(),
               None if nrm is None else (nrm.running_mean.data_ptr(), nrm.running_var.data_ptr(), float(nrm.eps)),
               rb.obs.data_ptr(), rb.clipped.data_ptr(), rb.next_fixed.data_ptr(), rb.dones.data_ptr(), rb.rew.data_ptr(),
               rb.h_rew.data_ptr(), rb.val.data_ptr(), rb.starts.data_ptr(), rb.last_val.data_ptr(), rb.last_done.data_ptr(),
               rb.adv.data_ptr(), rb.ret.data_ptr(), float(self.gamma), float(self.gae_lambda))
        if self._tail_args is None or self._tail_args[0] != key:
            a = L.RolloutTailArgs()
            a.obs, a.act_f32, a.act_i64 = rb.obs.data_ptr(), rb.clipped.data_ptr(), None
            a.next_obs, a.dones = rb.next_fixed.data_ptr(), rb.dones.data_ptr()
            a.obs_dim, a.act_dim = basic.obs_dim, basic.act_dim
            a.use_state, a.use_action, a.use_next_state, a.use_done = (int(f) for f in basic.flags)
            a.X, a.ldx, a.desc = ws["X"].data_ptr(), mlp.ldx, C.pointer(mlp.desc)
            a.params = mlp.flat.data_ptr()
            a.norm_mean = None if nrm is None else nrm.running_mean.data_ptr()
            a.norm_var = None if nrm is None else nrm.running_var.data_ptr()
            a.norm_eps, a.out_act = (0.0 if nrm is None else float(nrm.eps)), int(out_act)
            a.predict_ws, a.rewards, a.rewards_host = pws.data_ptr(), rb.rew.data_ptr(), rb.h_rew.data_ptr()
            a.values, a.episode_starts = rb.val.data_ptr(), rb.starts.data_ptr()
            a.last_values, a.last_dones = rb.last_val.data_ptr(), rb.last_done.data_ptr()
            a.T, a.n, a.gamma, a.gae_lambda = T, n, float(self.gamma), float(self.gae_lambda)
            a.advantages, a.returns = rb.adv.data_ptr(), rb.ret.data_ptr()
            self._tail_args = (key, a, mlp.desc)   # (the descriptor object stays alive with its pointer)
        return self._tail_args[1]

    def _upload_permutations_early(self, stream) -> None:
        """The pre-drawn permutations of the update that follows this rollout go to the device as soon as the helper
        thread has them (a side stream, beside the environment stepping). `train()` adopts the upload if it adopts the
        draw (`_PermutationPredraw.finish`), else it draws and uploads in place as before."""
        if self._perm_uploaded is not None or self._dp_global() or not self._predraw.ready():
            return
        up = L.side_stream(self.device, "upload")
        up.wait_stream(stream)              # (the previous update, the last reader of the device copy, has been enqueued there)
        with th.cuda.stream(up):
            self._perm_dev.copy_(self._perm_host, non_blocking=True)
            ev = th.cuda.Event()
            ev.record()
        self._perm_uploaded = ev

    def _relabel_module_rows(self, rb, module_net, pol, row_lo: int, row_hi: int, T: int, n: int) -> None:
        """`rb.rew` rows [row_lo, row_hi) (time-major) <- `module_net.predict_th` of the rollout tile's rows (current stream)."""
        osp = self.observation_space
        S, NS = rb.obs[:T].reshape(T * n, *osp.shape), rb.next_fixed.reshape(T * n, *osp.shape)
        A_ = rb.clipped.reshape(T * n).long() if pol.discrete else rb.clipped.reshape((T * n, *self.action_space.shape))
        D_, out = rb.dones.reshape(T * n), rb.rew.view(-1)
        # (chunks of 1 024 rows: rows are independent, the chunk size only bounds the transient activations -- 1.9 GB for the
        #  default CnnRewardNet on 84 x 84 frames -- and a 256-row chunk left the convolutions' launches four times as many)
        for lo in range(row_lo, row_hi, 1024):
            hi = min(row_hi, lo + 1024)
            out[lo:hi] = module_net.predict_th(S[lo:hi], A_[lo:hi], NS[lo:hi], D_[lo:hi])

    def _rollout_steps(self, env, callback, rb, T, n, pol, rw, bw, base, fused_net, module_net, act_step, mailbox,
                       act_stream, host_sampling, predrawn, starts, per_step_rews, prof, tick, h_clip_np, h_rew_np,
                       h_dones_np, h_trunc_np, h_next_np, h_obs_np, h_starts_np, stream) -> bool:
        """The step loop and the tail of `collect_rollouts` (its own function so that the rollout mailbox is closed on
        every way out)."""
        # Mailbox rollouts post step t + 1 AS SOON AS its observations are in their pinned row -- ahead of the rest of step
        # t's bookkeeping (next-observation / done / reward rows, callbacks, the buffering wrapper), which then runs while
        # the device computes the actions: ~10 us per env step off the host's critical path. (Needs the step's noise
        # in place already: pre-drawn tiles or host-side sampling.)
        post_ahead = self.rollout_post_ahead and (host_sampling or predrawn)
        posted = -1   # the last step posted to the mailbox
        early_at = (3 * T) // 4 if (module_net is not None and self.relabel_early and T >= 8) else 0
        early_T = 0   # steps whose rows are relabelled already
        for t in range(T):
            t0 = tick() if prof is not None else 0.0
            if not host_sampling and not predrawn:
                pol.draw_noise_into(rb.h_noise)
            t1 = tick() if prof is not None else 0.0
            if mailbox is not None:
                if posted < t:
                    mailbox[0](t)
                    posted = t
            else:
                act_step(t)
            t2 = tick() if prof is not None else 0.0
            if mailbox is not None and not mailbox[1](t):
                # the resident kernel gave up waiting (an env step longer than its time-out): per-step launches from here
                # on, this step included (the workgroups that did run it wrote the same values) -- and for the rest of
                # training: whatever was slow (or made a device-wide wait that could only return once the kernel had
                # left) will be there again next rollout
                mailbox = None
                if self.rollout_mailbox:
                    self.rollout_mailbox = False
                    warnings.warn("rollout mailbox: the resident act kernel timed out waiting for an environment step; "
                                  "continuing with per-step launches for the rest of training "
                                  "(PPO.rollout_mailbox = True re-enables it)", RuntimeWarning)
                act_stream.synchronize()
                with th.cuda.stream(act_stream):
                    act_step(t)
            if mailbox is None and not host_sampling:   # (the host-sampling step has waited for its own launch already)
                act_stream.synchronize()   # (so everything the act kernels wrote is complete before `stream` reads it)
            if t == 0:
                t_first_step = tick()      # the previous update has finished: the device is free from here on
                self._check_update_error()
            t3 = tick() if prof is not None else 0.0
            acts_np = h_clip_np[t]
            acts_np = acts_np.reshape(n).astype(np.int64) if pol.discrete else acts_np.reshape(
                (n, *self.action_space.shape)).copy()
            old_obs = self._last_obs
            base.step_async(acts_np)
            new_obs, env_rews, dones, nxt, trunc, infos = step_arrays(base)
            h_obs_np[t + 1] = new_obs.reshape(n, -1)
            if mailbox is not None and post_ahead and t + 1 < T:
                mailbox[0](t + 1)
                posted = t + 1
            if prof is not None:
                t4 = tick()
                for k, v in (("noise draw", t1 - t0), ("act launch", t2 - t1),
                             ("wait for the device" if t else "wait for the device, step 0 (previous update)", t3 - t2), ("env step", t4 - t3)):
                    prof[k] = prof.get(k, 0.0) + v
                prof["_t_book"] = t4
            self.num_timesteps += n
            if not callback.on_step():
                return False
            if infos is not None:   # [SB3 _update_info_buffer]; only non-empty dicts can carry an `episode` entry
                for info in filter(None, infos):
                    if info.get("episode") is not None:
                        self.ep_info_buffer.extend([info["episode"]])
            if bw is not None:
                bw.record_step(acts_np, new_obs, nxt, env_rews, dones, infos)
            h_next_np[t] = nxt.reshape(n, -1)
            h_dones_np[t], h_trunc_np[t], h_starts_np[t] = dones, trunc, starts
            if rw is not None and fused_net is None and module_net is None:  # arbitrary host reward function: per-step call
                r = rw.reward_fn(old_obs, acts_np, nxt, np.array(dones))
                per_step_rews.append(np.asarray(r, dtype=np.float32))
                h_rew_np[t] = per_step_rews[-1]
            else:
                h_rew_np[t] = env_rews
            self._last_obs, starts = new_obs, np.asarray(dones, dtype=bool)
            if t >= 1 and self._perm_uploaded is None:
                self._upload_permutations_early(stream)
            if t + 1 == early_at:
                # (the copies first: a copy enqueued behind the stream's wait for the updates would hold back every later copy)
                rb.upload_host_steps(early_at)
                if self.before_relabel is not None:
                    self.before_relabel()
                self._relabel_module_rows(rb, module_net, pol, 0, early_at * n, T, n)
                early_T = early_at
            if prof is not None:
                prof["bookkeeping"] = prof.get("bookkeeping", 0.0) + tick() - prof.pop("_t_book")
        self._last_episode_starts = starts
        if self.update_events is not None:  # tools: device time from the last env step to the start of the update
            self.tail_event = th.cuda.Event(enable_timing=True)
            self.tail_event.record()
        last_val_done = False
        if mailbox is not None and not host_sampling:
            mailbox[0](T)          # the observation behind the last step is in its pinned row: the resident kernel evaluates
            last_val_done = True   # V of it (the GAE bootstrap) while the host goes on, and leaves
            stream.wait_stream(act_stream)
        rb.h_last_done.copy_(th.as_tensor(starts.astype(np.float32)))
        rb.upload_host_tiles(from_step=early_T)
        if host_sampling:  # actions (= the clipped tile for Discrete heads) and log-probs were produced on the host
            rb.acts.copy_(rb.clipped)
            rb.logp.copy_(rb.h_logp, non_blocking=True)
        # host time the device had to itself for other streams' work during this rollout (see
        # `AdversarialTrainer._train_pipelined`: where the discriminator updates are scheduled)
        self.rollout_window_ms = 1e3 * (tick() - t_first_step)
        if self.before_relabel is not None and early_T == 0:   # (else: the stream has waited for the updates already)
            self.before_relabel()
        tail = None
        if (fused_net is not None and rw is not None and self.enqueue_first and self.rollout_tail_one_call and last_val_done
                and not pol.discrete and not h_trunc_np.any()):
            tail = self._rollout_tail_args(fused_net, rb, T, n)
        if tail is not None:
            # relabelling of the tile, the rewards' copy to the pinned host tile and GAE as ONE host call (`ia_rollout_tail`:
            # the launches of the general path below, in its order; ~75 us of Python between the last environment step and
            # the PPO launch otherwise)
            L.call("ia_rollout_tail", C.byref(tail), L.stream())
            copied = th.cuda.Event()
            copied.record()
            last_obs, dones_host = self._last_obs, h_dones_np.astype(bool)

            def bookkeeping():
                copied.synchronize()
                rw.record_rewards(rb.h_rew.numpy().copy(), dones_host, last_obs)

            self._post_enqueue_work.append(bookkeeping)
            rb.full = True
            self.rollout_done_event = th.cuda.Event()
            self.rollout_done_event.record()
            callback.on_rollout_end()
            return True
        if fused_net is not None:  # discriminator reward relabelling on the whole [T, n] tile
            acts_tbl = rb.clipped.reshape(T * n).long() if pol.discrete else rb.clipped.reshape(T * n, -1)
            table = TransitionTable(rb.obs[:T].reshape(T * n, -1), acts_tbl, rb.next_fixed.reshape(T * n, -1),
                                    rb.dones.reshape(T * n), pol.discrete)
            rb.rew.copy_(fused_net.predict_processed_rollout(table, T, n).reshape(T, n))
        elif module_net is not None:
            self._relabel_module_rows(rb, module_net, pol, early_T * n, T * n, T, n)
        else:
            rb.rew.copy_(rb.h_rew, non_blocking=True)
        if rw is not None:
            if (fused_net is not None or module_net is not None) and self.enqueue_first:
                # episode-return bookkeeping needs the relabelled rewards on the host but nothing on the
                # device needs it: copy now, consume after the PPO update has been enqueued
                rb.h_rew.copy_(rb.rew, non_blocking=True)
                copied = th.cuda.Event()
                copied.record()
                last_obs, dones_host = self._last_obs, h_dones_np.astype(bool)

                def bookkeeping():
                    copied.synchronize()
                    rw.record_rewards(rb.h_rew.numpy().copy(), dones_host, last_obs)

                self._post_enqueue_work.append(bookkeeping)
            else:
                wrapped = (rb.rew.cpu().numpy() if (fused_net is not None or module_net is not None)
                           else np.stack(per_step_rews))
                rw.record_rewards(wrapped, h_dones_np.astype(bool), self._last_obs)
        if h_trunc_np.any():  # rewards[i] += gamma * V(terminal_obs_i) for time-limit endings
            pol.values_rows(rb.next_fixed.reshape(T * n, -1), rb.term_val.reshape(T * n))
            L.call("ia_timeout_bootstrap", L.ptr(rb.rew), L.ptr(rb.term_val), L.ptr(rb.trunc), float(self.gamma),
                   T * n, L.stream())
        if not last_val_done:
            pol.values_rows(rb.obs[T], rb.last_val)
        L.call("ia_gae", L.ptr(rb.rew), L.ptr(rb.val), L.ptr(rb.starts), L.ptr(rb.last_val), L.ptr(rb.last_done), T,
               n, float(self.gamma), float(self.gae_lambda), L.ptr(rb.adv), L.ptr(rb.ret), L.stream())
        rb.full = True
        # lets another stream consume the finished rollout tile without waiting for the PPO update
        self.rollout_done_event = th.cuda.Event()
        self.rollout_done_event.record()
        callback.on_rollout_end()
        return True

    def _dp_global(self) -> bool:
        """Data-parallel PPO on the all-gathered rollout (see `_train_dp_global`); builds its state on
        first use. False: single process, or a shape the persistent kernel does not cover (then the
        per-minibatch all-reduce path `_train_data_parallel` runs)."""
        dp = self.dp
        if dp is None or dp.world <= 1 or not self.dp_global_minibatch or not self.policy.fused:
            return False
        if self._dpg is None:
            pol, rb = self.policy, self.rollout_buffer
            T, n, W = rb.buffer_size, rb.n_envs, dp.world
            aw = 1 if pol.discrete else pol.act_dim
            bg = min(W * self.batch_size, W * T * n)
            rows = bg // W
            # the row-sharded form is asked for FIRST: it is the one that still fits when the global minibatch has more row
            # blocks than one GPU's persistent kernel takes; the replicated form only where its shape is supported
            n_ws_s = int(L.load().ia_ppo_update_sharded_ws_floats(C.byref(pol.desc), rows, W)) if rows * W == bg else 0
            n_ws = int(L.load().ia_ppo_update_ws_floats(C.byref(pol.desc), bg))
            dev, cols = self.device, pol.obs_dim + aw + 3
            shard = None
            if self.dp_update_form != "replicated" and n_ws_s > 0:
                from imitation_amd.distributed import PeerExchange
                make = getattr(dp, "make_peer_exchange", None)   # (stand-in DataParallel objects of the tools: loopback)
                ex = make(pol.desc) if make is not None else PeerExchange(dp, pol.desc)
                if ex.ok:   # (the ranks' COMMON verdict: mapping + peer writes + system-scope polling work everywhere)
                    shard = dict(ex=ex, rows=rows, ws=th.zeros(n_ws_s, device=dev))
                else:
                    ex.close()
                    self.dp_handshake_failed = True   # (the ranks' common verdict: the same on every rank)
                    warnings.warn("data-parallel PPO update: the peer-memory handshake failed; every rank runs the "
                                  "whole global minibatch instead of its row shard", RuntimeWarning)
            if shard is None and n_ws <= 0:
                self._dpg = False
            else:
                perm_host = th.zeros(self.n_epochs, W * T * n, dtype=th.int64).pin_memory()
                forms = (["sharded"] if shard is not None else []) + (["replicated"] if n_ws > 0 else [])
                self._dpg = dict(
                    shard=shard, forms=forms, n_ws=n_ws, ws=None,   # (the replicated form's workspace: on first use)
                    auto=dict(k=0, events=[], ms={"sharded": [], "replicated": []}, chosen=None),
                    W=W, aw=aw, cols=cols, batch=bg,
                    send=th.empty(T, n, cols, device=dev),
                    # (one slice more than the T the kernels index: `ia_ppo_update*` fetch observation rows as 16-byte
                    #  pieces, the last piece of a row runs up to 12 bytes past it -- include/imitation_hip.h)
                    obs=th.empty(T + 1, W * n, pol.obs_dim, device=dev)[:T], acts=th.empty(T, W * n, aw, device=dev),
                    logp=th.empty(T, W * n, device=dev), adv=th.empty(T, W * n, device=dev),
                    ret=th.empty(T, W * n, device=dev), perm_host=perm_host, perm_np=perm_host.numpy(),
                    perm_dev=th.zeros(self.n_epochs, W * T * n, dtype=th.int64, device=dev),
                    perms=_SharedPermutations(dp.shared_seed(), self.n_epochs, W * T * n))
        return bool(self._dpg)

    def _train_dp_global(self, g, lr: float, clip_range: float, stats_dev: th.Tensor) -> None:
        """Data-parallel PPO update with NO per-step collective: ONE all-gather of the round's rollout
        shards (obs, actions, log-probs, advantages, returns; 1.8 MB per rank at config P), then every
        rank runs the identical persistent update on the global tile `[T, world*n_envs]` -- minibatches
        of world x batch_size rows, permutations shared by construction, deterministic kernels, hence
        bit-identical replicas. The PPO update is a latency-bound chain of dependent optimiser steps
        whose per-step time barely depends on the minibatch size (more 64-row workgroups, a two-level
        slab reduction), so replicating it costs far less than 160 latency-bound all-reduces per round
        would; the throughput-bound parts (rollout, discriminator) stay sharded."""
        pol, rb, dp = self.policy, self.rollout_buffer, self.dp
        T, n, W, D, aw = rb.buffer_size, rb.n_envs, g["W"], pol.obs_dim, g["aw"]
        send = g["send"]
        send[..., :D] = rb.obs[:T]
        send[..., D:D + aw] = rb.acts
        send[..., D + aw], send[..., D + aw + 1], send[..., D + aw + 2] = rb.logp, rb.adv, rb.ret
        allr = dp.all_gather_flat(send.reshape(-1)).view(W, T, n, g["cols"]).permute(1, 0, 2, 3).reshape(T, W * n, -1)
        g["obs"].copy_(allr[..., :D])
        g["acts"].copy_(allr[..., D:D + aw])
        g["logp"].copy_(allr[..., D + aw]); g["adv"].copy_(allr[..., D + aw + 1]); g["ret"].copy_(allr[..., D + aw + 2])
        g["perms"].finish()
        g["perm_dev"].copy_(g["perm_host"], non_blocking=True)
        rn = pol.features_extractor.normalize
        og = pol.optimizer.param_groups[0]
        if self.update_events is not None:
            self.update_events[0].record()
        form, timed = self._dp_pick_form(g)
        ev = (th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)) if timed else None
        if ev is not None:
            ev[0].record()
        sh = g["shard"] if form == "sharded" else None
        if sh is None and g["ws"] is None:
            g["ws"] = th.zeros(g["n_ws"], device=self.device)
        if sh is not None:
            # rows sharded over the ranks, one record per optimiser step exchanged inside the kernels
            ex = sh["ex"]
            steps = self.n_epochs * (-(-(W * T * n) // g["batch"]))
            L.call("ia_ppo_update_sharded", C.byref(pol.desc), L.ptr(pol._flat), L.ptr(pol._flat_t),
                   L.ptr(rn.running_mean) if rn else None, L.ptr(rn.running_var) if rn else None,
                   L.ptr(rn.count) if rn else None, int(rn is not None), L.ptr(g["obs"]), L.ptr(g["acts"]),
                   L.ptr(g["logp"]), L.ptr(g["adv"]), L.ptr(g["ret"]), L.ptr(g["perm_dev"]), self.n_epochs, T, W * n,
                   sh["rows"], int(self.normalize_advantage), float(clip_range), float(self.ent_coef),
                   float(self.vf_coef), float(self.max_grad_norm), L.ptr(pol.optimizer.exp_avg),
                   L.ptr(pol.optimizer.exp_avg_sq), float(lr), float(og["betas"][0]), float(og["betas"][1]),
                   float(og["eps"]), pol.optimizer.step_count, L.ptr(sh["ws"]), L.ptr(stats_dev), ex.world, ex.rank,
                   ex.take_steps(steps), ex.recv, ex.peer_recv, int(ex.loop),
                   float(self.dp_exchange_timeout_s), L.stream())
            self.dp_sharded_updates = getattr(self, "dp_sharded_updates", 0) + 1
        else:
            L.call("ia_ppo_update", C.byref(pol.desc), L.ptr(pol._flat), L.ptr(pol._flat_t),
                   L.ptr(rn.running_mean) if rn else None, L.ptr(rn.running_var) if rn else None,
                   L.ptr(rn.count) if rn else None, int(rn is not None), L.ptr(g["obs"]), L.ptr(g["acts"]), L.ptr(g["logp"]),
                   L.ptr(g["adv"]), L.ptr(g["ret"]), L.ptr(g["perm_dev"]), self.n_epochs, T, W * n, g["batch"],
                   int(self.normalize_advantage), float(clip_range), float(self.ent_coef), float(self.vf_coef),
                   float(self.max_grad_norm), L.ptr(pol.optimizer.exp_avg), L.ptr(pol.optimizer.exp_avg_sq),
                   float(lr), float(og["betas"][0]), float(og["betas"][1]), float(og["eps"]), pol.optimizer.step_count,
                   L.ptr(g["ws"]), L.ptr(stats_dev), L.stream())
        if ev is not None:
            ev[1].record()
            g["auto"]["events"].append((form, ev))
        if self.update_events is not None:
            self.update_events[1].record()
        pol.optimizer.step_count += self.n_epochs * self._n_mb

    _DP_AUTO_TRIALS = 4   # timed updates of the "auto" choice: sharded / replicated alternating, the first of each discarded
                          # (the driver's five warm-up rounds hold the trials and the verdict)

    def _dp_pick_form(self, g):
        """(form of this update, whether it is timed). One form available, or one asked for: that one. "auto": the first
        `_DP_AUTO_TRIALS` updates alternate between the forms under HIP events; then every rank takes the minimum of its
        kept timings per form, the ranks' MAXIMA (one all-gather of two numbers: a step is as slow as the slowest rank)
        decide for all of them alike, and `dp_choice` keeps the record."""
        forms, auto = g["forms"], g["auto"]
        if len(forms) == 1 or self.dp_update_form in forms:
            return (self.dp_update_form if self.dp_update_form in forms else forms[0]), False
        if auto["chosen"] is not None:
            return auto["chosen"], False
        k = auto["k"]
        auto["k"] = k + 1
        if k < self._DP_AUTO_TRIALS:
            return forms[k % 2], True
        for form, (e0, e1) in auto["events"]:
            e1.synchronize()
            auto["ms"][form].append(e0.elapsed_time(e1))
        mine = [min(auto["ms"][f][1:]) for f in ("sharded", "replicated")]
        allr = self.dp.all_gather_flat(th.tensor(mine, dtype=th.float32, device=self.device)).view(-1, 2).cpu()
        worst = allr.max(dim=0).values.tolist()
        auto["chosen"] = "sharded" if worst[0] <= worst[1] else "replicated"
        auto["events"] = []
        self.dp_choice = dict(sharded_ms=worst[0], replicated_ms=worst[1], chosen=auto["chosen"],
                              trials={f: [round(x, 4) for x in v] for f, v in auto["ms"].items()})
        return auto["chosen"], False

    def close(self) -> None:
        """Releases what the data-parallel update holds outside torch's allocator: the peer-mapped exchange block and the
        hipIpc mappings of the other ranks' blocks (`distributed.PeerExchange`)."""
        g = self._dpg if isinstance(self._dpg, dict) else None
        if g is not None and g.get("shard") is not None:
            g["shard"]["ex"].close()
            g["shard"] = None
            g["forms"] = [f for f in g["forms"] if f != "sharded"]
            if not g["forms"]:
                self._dpg = None   # nothing left to run: the next update builds the state (and its exchange) again

    def _train_data_parallel(self, perm: np.ndarray, lr: float, clip_range: float) -> None:
        """Minibatch loop with one RCCL all-reduce of the flat policy gradient per optimiser step
        (and a moment all-gather for the feature RunningNorm); replicas stay identical."""
        pol, rb, dp = self.policy, self.rollout_buffer, self.dp
        T, n = rb.buffer_size, rb.n_envs
        total = T * n
        rn = pol.features_extractor.normalize
        g = pol.optimizer.param_groups[0]
        b1, b2 = g["betas"]
        P = pol._flat.numel()
        offs_host = ((perm % T) * n + perm // T).astype(np.int64)  # time-major row of each permuted index
        offs = th.from_numpy(offs_host).to(self.device)
        obs_rows = rb.obs.reshape((T + 1) * n, -1)
        # The feature-RunningNorm update of a minibatch depends on the data only, so the slab moments of
        # ALL minibatches of this update are formed first and exchanged in ONE all-gather; the per-step
        # loop then merges them locally (same partial / merge kernels and order as the per-minibatch
        # exchange, hence the same statistics) and keeps a single collective per optimiser step.
        pre = None
        if rn is not None and total % self.batch_size == 0 and self.dp_batch_moments:
            bsz, steps = self.batch_size, self.n_epochs * self._n_mb
            need = int(L.load().ia_running_norm_ws_floats(bsz, pol.obs_dim))
            if self._dp_ws_pre is None or self._dp_ws_pre.numel() != steps * need:
                self._dp_ws_pre = th.empty(steps, need, device=self.device)
            for e in range(self.n_epochs):
                for mb, start in enumerate(range(0, total, bsz)):
                    L.call("ia_gather_rows", L.ptr(obs_rows), L.ptr(offs[e, start:start + bsz]), bsz, pol.obs_dim,
                           L.ptr(self._dp_obs), L.stream())
                    L.call("ia_running_norm_partial", L.ptr(self._dp_obs), pol.obs_dim, bsz, pol.obs_dim,
                           L.ptr(self._dp_ws_pre[e * self._n_mb + mb]), L.stream())
            gathered = dp.all_gather_flat(self._dp_ws_pre.reshape(-1))
            pre = gathered.view(dp.world, steps, need).permute(1, 0, 2).contiguous()   # [step][rank][ws]
        for e in range(self.n_epochs):
            for mb, start in enumerate(range(0, total, self.batch_size)):
                b = min(self.batch_size, total - start)
                idx = self._perm_dev[e, start:start + b]
                if pre is not None:
                    L.call("ia_running_norm_merge", L.ptr(pre[e * self._n_mb + mb]), dp.world, b, pol.obs_dim,
                           pol.obs_dim, L.ptr(rn.running_mean), L.ptr(rn.running_var), L.ptr(rn.count), L.stream())
                elif rn is not None:
                    L.call("ia_gather_rows", L.ptr(obs_rows), L.ptr(offs[e, start:start + b]), b, pol.obs_dim,
                           L.ptr(self._dp_obs), L.stream())
                    rn.update_stats(self._dp_obs, ldx=pol.obs_dim, rows=b)
                L.call("ia_ppo_minibatch_grad", C.byref(pol.desc), L.ptr(pol._flat), L.ptr(pol._flat_t),
                       L.ptr(rn.running_mean) if rn else None, L.ptr(rn.running_var) if rn else None,
                       L.ptr(rn.count) if rn else None, 0, L.ptr(rb.obs), L.ptr(rb.acts), L.ptr(rb.logp),
                       L.ptr(rb.adv), L.ptr(rb.ret), L.ptr(idx), b, T, n, int(self.normalize_advantage),
                       float(clip_range), float(self.ent_coef), float(self.vf_coef), L.ptr(self._ppo_ws), L.stream())
                off = int(L.load().ia_ppo_grad_offset(C.byref(pol.desc), b))
                dp.allreduce_mean_(self._ppo_ws[off:off + P])
                pol.optimizer.step_count += 1
                bc1 = 1.0 - b1 ** pol.optimizer.step_count
                bc2 = 1.0 - b2 ** pol.optimizer.step_count
                L.call("ia_ppo_minibatch_apply", C.byref(pol.desc), L.ptr(pol._flat), L.ptr(pol._flat_t), b,
                       float(self.ent_coef), float(self.vf_coef), float(self.max_grad_norm),
                       L.ptr(pol.optimizer.exp_avg), L.ptr(pol.optimizer.exp_avg_sq), float(b1), float(b2),
                       float(g["eps"]), float(lr / bc1), float(bc2 ** 0.5), L.ptr(self._ppo_ws),
                       L.ptr(self._stats_dev[e, mb]), L.stream())

    # ---- PPO update (App. A.7) ----------------------------------------------------------------
    def train(self, record_lr: bool = True):
        pol, rb = self.policy, self.rollout_buffer
        if self._pending_train is not None:
            # a deferred record nobody has collected yet (`learn` spanning several iterations inside one
            # `train_gen`): log it now, in order, instead of overwriting it (the reference logs every iteration)
            self.finalize_train()
        pol.set_training_mode(True)
        lr = self.lr_schedule(self._current_progress_remaining)
        if record_lr:
            self.logger.record("train/learning_rate", lr)
        pol.optimizer.param_groups[0]["lr"] = lr
        clip_range = self.clip_range(self._current_progress_remaining)
        T, n = rb.buffer_size, rb.n_envs
        dpg = self._dpg if self._dp_global() else None
        perm = self._perm_np
        if dpg is None:
            early, self._perm_uploaded = self._perm_uploaded, None
            if not self._predraw.finish(perm):
                if early is not None:
                    # the speculative upload may still be reading the pinned buffer (and writing the device copy) this
                    # branch is about to overwrite: let it finish first
                    early.synchronize()
                early = None
                for e in range(self.n_epochs):  # one global-NumPy draw per epoch, as RolloutBuffer.get does
                    perm[e] = np.random.permutation(T * n)
            if early is not None:   # uploaded beside the rollout (`_upload_permutations_early`): 1.3 MB of PCIe time that
                th.cuda.current_stream().wait_event(early)   # would sit between the last env step and the update
            else:
                self._perm_dev.copy_(self._perm_host, non_blocking=True)
        rn = pol.features_extractor.normalize
        g = pol.optimizer.param_groups[0]
        rec = None
        if self.defer_train_stats:
            if self._records is None:
                self._records = [_TrainRecord(self._stats_dev, rb.val, pol.log_std) for _ in range(2)]
                self._fin_stream = L.side_stream(self.device, "readback")
            rec = self._records[self._rec_i % 2]
            self._rec_i += 1
        stats_dev = rec.stats if rec is not None else self._stats_dev
        if not pol.fused:
            # general towers: [SB3 PPO.train]'s minibatch loop on the generic stacks (one gradient all-reduce per
            # optimiser step when data-parallel; the feature RunningNorm exchanges its moments itself)
            sd = self._stats_dev if (rec is not None and self.dp is not None and self.dp.world > 1) else stats_dev
            pol.ppo_update(rb, self._perm_dev, self.n_epochs, self.batch_size, self.normalize_advantage, clip_range,
                           self.ent_coef, self.vf_coef, self.max_grad_norm, sd, dp=self.dp)
        elif dpg is not None:
            self._train_dp_global(dpg, lr, clip_range, stats_dev)
        elif self.dp is not None and self.dp.world > 1:
            self._train_data_parallel(perm, lr, clip_range)
        single = pol.fused and not (self.dp is not None and self.dp.world > 1)
        if single and self._upd_ws is not None:
            if self.update_events is not None:  # measurement hook (bench.py): events on the launch stream
                self.update_events[0].record()
            rc = L.load().ia_ppo_update(
                C.byref(pol.desc), L.ptr(pol._flat), L.ptr(pol._flat_t),
                L.ptr(rn.running_mean) if rn else None, L.ptr(rn.running_var) if rn else None,
                L.ptr(rn.count) if rn else None, int(rn is not None), L.ptr(rb.obs), L.ptr(rb.acts), L.ptr(rb.logp),
                L.ptr(rb.adv), L.ptr(rb.ret), L.ptr(self._perm_dev), self.n_epochs, T, n, min(self.batch_size, T * n),
                int(self.normalize_advantage), float(clip_range), float(self.ent_coef), float(self.vf_coef),
                float(self.max_grad_norm), L.ptr(pol.optimizer.exp_avg), L.ptr(pol.optimizer.exp_avg_sq),
                float(lr), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]), pol.optimizer.step_count,
                L.ptr(self._upd_ws), L.ptr(stats_dev), L.stream())
            if rc == L.ERR_UNSUPPORTED:
                # not every workgroup of the persistent kernel would be resident at once on this device (its grid
                # barriers need that): nothing was launched; this and all later updates take the per-epoch kernels
                self._upd_ws = None
            else:
                L.check(rc, "ia_ppo_update")
                if self.update_events is not None:
                    self.update_events[1].record()
                pol.optimizer.step_count += self.n_epochs * self._n_mb
                # sticky error word of the launch -> pinned host word, checked before the next rollout acts on the
                # updated parameters (`_check_update_error`)
                if self._h_upd_err is None:
                    self._h_upd_err = th.zeros(1, dtype=th.int32).pin_memory()
                self._h_upd_err.copy_(self._upd_ws[8:9].view(th.int32), non_blocking=True)
                self._upd_err_pending = True
        per_epoch = single and self._upd_ws is None
        if per_epoch and self.n_epochs > 1 and self._ppo_ws_epochs == self.n_epochs and self.epochs_one_call:
            # every epoch of the update as one sequence of minibatches (the rollout divides into whole minibatches): gather
            # and statistics of all epochs in one launch each, the epoch kernels back to back
            rc = L.load().ia_ppo_epochs(
                C.byref(pol.desc), L.ptr(pol._flat), L.ptr(pol._flat_t),
                L.ptr(rn.running_mean) if rn else None, L.ptr(rn.running_var) if rn else None,
                L.ptr(rn.count) if rn else None, int(rn is not None), L.ptr(rb.obs), L.ptr(rb.acts), L.ptr(rb.logp),
                L.ptr(rb.adv), L.ptr(rb.ret), L.ptr(self._perm_dev), self.n_epochs, T, n, self.batch_size,
                int(self.normalize_advantage), float(clip_range), float(self.ent_coef), float(self.vf_coef),
                float(self.max_grad_norm), L.ptr(pol.optimizer.exp_avg), L.ptr(pol.optimizer.exp_avg_sq),
                float(lr), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]), pol.optimizer.step_count,
                L.ptr(self._ppo_ws), L.ptr(stats_dev), L.stream())
            if rc != L.ERR_UNSUPPORTED:
                L.check(rc, "ia_ppo_epochs")
                pol.optimizer.step_count += self.n_epochs * self._n_mb
                per_epoch = False
        for e in range(self.n_epochs if per_epoch else 0):
            L.call("ia_ppo_epoch", C.byref(pol.desc), L.ptr(pol._flat), L.ptr(pol._flat_t),
                   L.ptr(rn.running_mean) if rn else None, L.ptr(rn.running_var) if rn else None,
                   L.ptr(rn.count) if rn else None, int(rn is not None), L.ptr(rb.obs), L.ptr(rb.acts), L.ptr(rb.logp),
                   L.ptr(rb.adv), L.ptr(rb.ret), L.ptr(self._perm_dev[e]), T, n, self.batch_size,
                   int(self.normalize_advantage), float(clip_range), float(self.ent_coef), float(self.vf_coef),
                   float(self.max_grad_norm), L.ptr(pol.optimizer.exp_avg), L.ptr(pol.optimizer.exp_avg_sq),
                   float(lr), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]), pol.optimizer.step_count,
                   L.ptr(self._ppo_ws), L.ptr(stats_dev[e]), L.stream())
            pol.optimizer.step_count += self._n_mb
        self._n_updates += self.n_epochs
        if rec is not None:
            # (behind the update's launch, not ahead of it: the update neither writes these tiles nor do they change before
            #  the next rollout's act kernels, which are ordered behind this stream)
            rec.val.copy_(rb.val)   # the tile is reused by the next rollout before the statistics are read
            rec.ret.copy_(rb.ret)
            if self.dp is not None and self.dp.world > 1 and dpg is None:   # that path wrote the shared statistics tile
                rec.stats.copy_(self._stats_dev)
            if rec.log_std is not None:
                rec.log_std.copy_(pol.log_std)
            err_ws = self._upd_ws
            if dpg is not None:   # the data-parallel launch has its own workspace (word 8: 1 = a grid wait, 2 = a peer's
                err_ws = dpg["shard"]["ws"] if dpg["shard"] is not None else dpg["ws"]   # record did not arrive in time)
            if err_ws is not None:
                rec.err.copy_(err_ws[8:9].view(th.int32))
            rec.ready.record()
            rec.clip_range, rec.n_updates = clip_range, self._n_updates
            self._pending_train = rec
        else:
            self._pending_train = clip_range
            self.finalize_train()
        return lr

    def _check_update_error(self) -> None:
        """After a synchronisation that covers the last `ia_ppo_update`: raise (and clear the sticky device word)
        if a grid-wide wait inside it timed out -- BEFORE a rollout acts on parameters it left half-updated."""
        if not self._upd_err_pending:
            return
        self._upd_err_pending = False
        if int(self._h_upd_err[0]) != 0:
            self._upd_ws[8:9].zero_()
            raise RuntimeError("ia_ppo_update: a grid-wide wait timed out inside the persistent PPO kernel (its "
                               "workgroups were not co-resident); the parameters of this update are invalid")

    def finalize_train(self) -> None:
        """Statistics read-back + logging of the last `train()` (SB3 PPO.train's logger block)."""
        if self._pending_train is None:
            return
        pending, self._pending_train = self._pending_train, None
        pol, rb = self.policy, self.rollout_buffer
        if isinstance(pending, _TrainRecord):
            pending.read_back(self._fin_stream)
            clip_range, n_updates = pending.clip_range, pending.n_updates
            st, err = pending.h_stats.numpy(), int(pending.h_err.item())
            vals, rets = pending.h_val.numpy().reshape(-1), pending.h_ret.numpy().reshape(-1)
            std = None if pending.h_log_std is None else float(th.exp(pending.h_log_std).mean().item())
        else:
            clip_range, n_updates = pending, self._n_updates
            st = self._stats_dev.cpu().numpy()  # one synchronisation per train()
            err_ws = self._upd_ws
            if self._dpg:
                err_ws = self._dpg["shard"]["ws"] if self._dpg["shard"] is not None else self._dpg["ws"]
            err = 0 if err_ws is None else int(err_ws[8:9].view(th.int32).item())
            vals, rets = rb.val.cpu().numpy().reshape(-1), rb.ret.cpu().numpy().reshape(-1)
            std = None if pol.discrete else float(th.exp(pol.log_std.cpu()).mean().item())  # host exp, as on every schedule
        if err != 0:
            raise RuntimeError("ia_ppo_update: " + ("a peer rank's gradient record did not arrive in time"
                                                    if err == 2 else "a grid-wide wait timed out")
                               + " inside the persistent PPO kernel; the parameters of this update are invalid")
        var_y = np.var(rets)
        ev = np.nan if var_y == 0 else 1 - np.var(rets - vals) / var_y
        self.logger.record("train/entropy_loss", float(st[..., 2].mean()))
        self.logger.record("train/policy_gradient_loss", float(st[..., 0].mean()))
        self.logger.record("train/value_loss", float(st[..., 1].mean()))
        self.logger.record("train/approx_kl", float(st[-1, :, 3].mean()))
        self.logger.record("train/clip_fraction", float(st[..., 4].mean()))
        self.logger.record("train/loss", float(st[-1, -1, 5]))
        self.logger.record("train/explained_variance", float(ev))
        if std is not None:
            self.logger.record("train/std", std)
        self.logger.record("train/n_updates", n_updates, exclude="tensorboard")
        self.logger.record("train/clip_range", clip_range)
