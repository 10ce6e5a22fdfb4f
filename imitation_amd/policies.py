"""Actor-critic policy behind the SB3 `ActorCriticPolicy` surface the trainer touches
(`adversarial/common.py:262-266,490-496`; `policies/base.py:92-149`), evaluated by the fused
policy kernels of libimitation_hip.so.

Fused architecture = what the reference's GAIL/AIRL configs use: Flatten or
`NormalizeFeaturesExtractor(RunningNorm)` features, two separate tanh towers of equal width
H in {32, 64} (`FeedForward32Policy` = [32, 32]; SB3 `MlpPolicy` default = [64, 64]),
DiagGaussian (Box) or Categorical (Discrete) head. Any other `net_arch` (deeper, unequal, `dict(pi=..., vf=...)`,
ReLU towers) keeps this class's surface but executes on the generic MLP stacks: `general_policy.GeneralTowers`. Parameters are ONE flat fp32 buffer in torch
`parameters()` order (log_std, pi tower, vf tower, action_net, value_net) + a transposed
shadow copy the kernels read their weight rows from.
"""
from __future__ import annotations

import ctypes as C
import functools
from typing import Dict, Iterator, List, Optional, Tuple

import numpy as np
import torch as th
from torch import nn

from imitation_amd import _lib as L
from imitation_amd import spaces
from imitation_amd.networks import HipAdam, RunningNorm, require_device


class FlattenExtractor:
    """[SB3 torch_layers.FlattenExtractor]."""

    def __init__(self, observation_space):
        self.features_dim = spaces.flatdim(observation_space)
        self.normalize: Optional[RunningNorm] = None


class NormalizeFeaturesExtractor(FlattenExtractor):
    """`policies/base.py:123-149`: flatten then `normalize_class(features_dim)`."""

    def __init__(self, observation_space, normalize_class=RunningNorm):
        super().__init__(observation_space)
        if normalize_class is not RunningNorm:
            raise NotImplementedError("feature normalisation of the HIP policies: imitation_amd.RunningNorm (the policy "
                                      "kernels merge Chan statistics inside the PPO update; EMANorm is implemented for "
                                      "reward nets only)")
        self.normalize = normalize_class(self.features_dim)


def categorical_sample(logits: th.Tensor):
    """`d = torch.distributions.Categorical(logits=logits); a = d.sample(); return a, d.log_prob(a)` op for op -- the
    normalisation `logits - logsumexp`, `softmax` of that, ONE `torch.multinomial(probs, 1, True)` on the global
    generator, a gather of the normalised logits -- without the distribution object (argument validation, lazy
    properties, broadcasting helpers: 45 us of a 125 us rollout step at 8 environments). Same values, same draws
    (`tests/test_host_logic.py::test_categorical_sample_equals_torch_distributions`)."""
    lg = logits - logits.logsumexp(dim=-1, keepdim=True)
    probs = th.nn.functional.softmax(lg, dim=-1)
    a = th.multinomial(probs.reshape(-1, probs.shape[-1]), 1, True).T.reshape(probs.shape[:-1])
    return a, lg.gather(-1, a.unsqueeze(-1)).squeeze(-1)


_RACE_OK = None   # does `_multinomial_one` reproduce `torch.multinomial(p, 1, True)` on this torch build? (checked on first use)


def _multinomial_one(probs: th.Tensor, generator=None) -> np.ndarray:
    """`torch.multinomial(probs, 1, True)[:, 0]` of a 2-D float32 tile as torch itself forms it (ATen `multinomial`, one sample
    per row: `q = empty_like(p).exponential_(1)`, `argmax(p / q)` -- the exponential race): the ONE generator call is the same
    call on the same tensor, the division and the argmax are NumPy's (IEEE division; first maximum). torch's own entry spends
    most of its ~10 us per 8 x 2 tile on argument checks (`max`, `min`, `sum`, `any` reductions of the probabilities) that a
    softmax's output does not need."""
    q = th.empty_like(probs).exponential_(1.0, generator=generator)
    return np.argmax(probs.numpy() / q.numpy(), axis=-1)


def _race_reproduces_multinomial() -> bool:
    """The guard of `_multinomial_one`: on private generators, the same actions AND the same generator state as
    `torch.multinomial` over several shapes and draws -- an ATen that samples differently fails it and the rollout step keeps
    calling `torch.multinomial`."""
    global _RACE_OK
    if _RACE_OK is None:
        ok = True
        gp = th.Generator().manual_seed(11)
        for shape in ((8, 2), (64, 6), (1024, 6), (5, 13), (1, 3)):
            g1, g2 = th.Generator().manual_seed(7), th.Generator().manual_seed(7)
            for _ in range(12):
                p = th.softmax(th.randn(*shape, generator=gp) * 3.0, -1)
                ref = th.multinomial(p, 1, True, generator=g1)[:, 0].numpy()
                ok = ok and np.array_equal(ref, _multinomial_one(p, g2)) and th.equal(g1.get_state(), g2.get_state())
        _RACE_OK = bool(ok)
    return _RACE_OK


def categorical_sample_into(logits: th.Tensor, logp_row: np.ndarray, act_row: np.ndarray, rows: np.ndarray) -> None:
    """`categorical_sample` of a host `[n, A]` logits tile with the results written straight into NumPy rows (the pinned rollout
    tiles' views): `logp_row[n]` float32, `act_row[n]` float32 (the action indices; exact), `rows = np.arange(n)`. The three
    torch calls that fix the values and the draw are the same; the gather and the two row copies -- seven small torch calls of
    3-5 us each per environment step at 8 environments -- are NumPy indexing."""
    lg = logits - logits.logsumexp(dim=-1, keepdim=True)
    probs = th.nn.functional.softmax(lg, dim=-1)
    if probs.dtype == th.float32 and probs.dim() == 2 and _race_reproduces_multinomial():
        a = _multinomial_one(probs)
    else:
        a = th.multinomial(probs, 1, True).numpy()[:, 0]
    logp_row[:] = lg.numpy()[rows, a]
    act_row[:] = a


class ActorCriticPolicy:
    fused = True   # False once `general_policy.adopt` re-classed the instance (towers outside the fused kernels)

    def __init__(self, observation_space, action_space, lr_schedule, net_arch=None, activation_fn=nn.Tanh,
                 ortho_init: bool = True, use_sde: bool = False, log_std_init: float = 0.0,
                 squash_output: bool = False, features_extractor_class=FlattenExtractor,
                 features_extractor_kwargs=None, share_features_extractor: bool = True, normalize_images: bool = True,
                 optimizer_class=th.optim.Adam, optimizer_kwargs=None):
        if use_sde or squash_output or not share_features_extractor:
            raise NotImplementedError("gSDE / squashing / separate extractors are outside the reference's PPO path")
        if optimizer_class is not th.optim.Adam:
            raise NotImplementedError("the PPO step implements Adam (SB3 default)")
        if net_arch is None:
            net_arch = dict(pi=[64, 64], vf=[64, 64])
        if isinstance(net_arch, dict):   # [SB3 MlpExtractor]: a missing key = no hidden layer in that tower
            pi, vf = list(net_arch.get("pi", [])), list(net_arch.get("vf", []))
        else:
            pi, vf = list(net_arch), list(net_arch)
        from imitation_amd import general_policy
        fused = general_policy.fused_arch(pi, vf, activation_fn)
        self.observation_space, self.action_space = observation_space, action_space
        self.net_arch, self.hidden = net_arch, (int(pi[0]) if fused else None)
        self.discrete = isinstance(action_space, spaces.Discrete)
        self.obs_dim = spaces.flatdim(observation_space)
        self.act_dim = action_space.n if self.discrete else int(np.prod(action_space.shape))
        self.features_extractor = features_extractor_class(observation_space, **(features_extractor_kwargs or {}))
        self.pi_features_extractor = self.vf_features_extractor = self.features_extractor
        self.optimizer_kwargs = dict(optimizer_kwargs or {})
        self.optimizer_kwargs.setdefault("eps", 1e-5)  # SB3 ActorCriticPolicy default for Adam
        self.training = True
        self._squash_output = False
        # Discrete heads: "multinomial" = [SB3 CategoricalDistribution.sample] itself on the host -- the head's
        # logits come back from the device and `torch.distributions.Categorical(logits).sample()` consumes
        # torch's global CPU generator exactly as the reference does (identical seeds => identical actions);
        # "inverse_cdf" = one host U(0,1) per row, sampled inside the rollout kernel (same distribution,
        # different stream, one launch + no logits round trip per step).
        self.discrete_sampling = "multinomial"
        self._lr0 = float(lr_schedule(1))
        self._low = self._high = None
        self.optimizer: Optional[HipAdam] = None
        if not fused:
            # any other `net_arch` / ReLU towers: same parameters and API, executed on the generic MLP stacks
            general_policy.adopt(self, pi, vf, activation_fn, ortho_init, log_std_init)
            return

        # Host construction in SB3's order so the torch global RNG is consumed identically:
        # pi tower, vf tower, action_net, log_std, value_net; then orthogonal re-initialisation.
        H, D = self.hidden, self.obs_dim
        pi_net = nn.Sequential(nn.Linear(D, H), nn.Tanh(), nn.Linear(H, H), nn.Tanh())
        vf_net = nn.Sequential(nn.Linear(D, H), nn.Tanh(), nn.Linear(H, H), nn.Tanh())
        action_net = nn.Linear(H, self.act_dim)
        log_std = None if self.discrete else th.ones(self.act_dim) * log_std_init
        value_net = nn.Linear(H, 1)
        if ortho_init:
            def init(m, gain):
                if isinstance(m, nn.Linear):
                    nn.init.orthogonal_(m.weight, gain=gain)
                    m.bias.data.fill_(0.0)
            for mod, gain in ((pi_net, np.sqrt(2)), (vf_net, np.sqrt(2)), (action_net, 0.01), (value_net, 1)):
                mod.apply(functools.partial(init, gain=gain))
        parts: List[th.Tensor] = [] if self.discrete else [log_std]
        for lin in (pi_net[0], pi_net[2], vf_net[0], vf_net[2], action_net, value_net):
            parts += [lin.weight.detach().reshape(-1), lin.bias.detach().reshape(-1)]
        self._flat = th.cat(parts).contiguous()
        self._flat_t: Optional[th.Tensor] = None
        self.desc = L.PolicyDesc(self.obs_dim, self.act_dim, H, int(self.discrete),
                                 int(self.features_extractor.normalize is not None),
                                 self.features_extractor.normalize.eps if self.features_extractor.normalize else 1e-5)

    # ---- layout -------------------------------------------------------------------------
    def _layout(self) -> List[Tuple[str, Tuple[int, ...]]]:
        H, D, A = self.hidden, self.obs_dim, self.act_dim
        lay = [] if self.discrete else [("log_std", (A,))]
        for tower in ("policy_net", "value_net"):
            lay += [(f"mlp_extractor.{tower}.0.weight", (H, D)), (f"mlp_extractor.{tower}.0.bias", (H,)),
                    (f"mlp_extractor.{tower}.2.weight", (H, H)), (f"mlp_extractor.{tower}.2.bias", (H,))]
        lay += [("action_net.weight", (A, H)), ("action_net.bias", (A,)), ("value_net.weight", (1, H)),
                ("value_net.bias", (1,))]
        return lay

    def named_parameters(self) -> Iterator[Tuple[str, th.Tensor]]:
        o = 0
        for name, shape in self._layout():
            n = int(np.prod(shape))
            yield name, self._flat[o:o + n].view(shape)
            o += n

    def parameters(self) -> Iterator[th.Tensor]:
        for _, p in self.named_parameters():
            yield p

    @property
    def log_std(self) -> Optional[th.Tensor]:
        return None if self.discrete else self._flat[: self.act_dim]

    def state_dict(self) -> Dict[str, th.Tensor]:
        sd: Dict[str, th.Tensor] = {}
        named = dict(self.named_parameters())
        if not self.discrete:
            sd["log_std"] = named.pop("log_std")
        rn = self.features_extractor.normalize
        if rn is not None:  # SB3 registers the shared extractor under three names
            for ext in ("features_extractor", "pi_features_extractor", "vf_features_extractor"):
                sd.update(rn.state_dict(f"{ext}.normalize."))
        sd.update(named)
        return sd

    def load_state_dict(self, sd) -> None:
        for k, v in self.named_parameters():
            v.copy_(th.as_tensor(sd[k]))
        rn = self.features_extractor.normalize
        if rn is not None:
            rn.load_state_dict(sd, "features_extractor.normalize.")
        self._sync_transposed()

    # ---- module-like plumbing --------------------------------------------------------------
    @property
    def device(self) -> th.device:
        return self._flat.device

    @property
    def squash_output(self) -> bool:
        return self._squash_output

    def to(self, device):
        device = th.device(device)
        self._flat = self._flat.to(device).contiguous()
        if self.features_extractor.normalize is not None:
            self.features_extractor.normalize.to(device)
        if device.type == "cuda":
            self._flat_t = th.empty_like(self._flat)
            self._sync_transposed()
            self.optimizer = HipAdam(self._flat, th.zeros_like(self._flat), lr=self._lr0, **self.optimizer_kwargs)
            if not self.discrete:
                self._low = th.as_tensor(self.action_space.low.reshape(-1), dtype=th.float32, device=device)
                self._high = th.as_tensor(self.action_space.high.reshape(-1), dtype=th.float32, device=device)
            else:
                self._low = self._high = th.zeros(self.act_dim, device=device)
        return self

    def _sync_transposed(self) -> None:
        if self._flat_t is not None:
            L.call("ia_policy_transpose", C.byref(self.desc), L.ptr(self._flat), L.ptr(self._flat_t), L.stream())

    def set_training_mode(self, mode: bool) -> None:
        self.train(mode)

    def train(self, mode: bool = True):
        self.training = mode
        if self.features_extractor.normalize is not None:
            self.features_extractor.normalize.train(mode)
        return self

    def eval(self):
        return self.train(False)

    def _norm_ptrs(self):
        rn = self.features_extractor.normalize
        return (None, None) if rn is None else (L.ptr(rn.running_mean), L.ptr(rn.running_var))

    def _obs_dev(self, obs) -> th.Tensor:
        """Observation batch -> contiguous fp32 `[n, obs_dim]` on device ([SB3 preprocess_obs] Box: .float())."""
        t = obs if isinstance(obs, th.Tensor) else th.as_tensor(np.ascontiguousarray(obs))
        return t.to(self.device, th.float32).reshape(t.shape[0], -1).contiguous()

    def _maybe_update_norm(self, obs_dev: th.Tensor) -> None:
        rn = self.features_extractor.normalize
        if rn is not None and self.training:  # train-mode forward updates before normalising (networks.py:81-87)
            rn.update_stats(obs_dev)

    # ---- SB3 API -----------------------------------------------------------------------------
    def sample_noise(self, n: int) -> th.Tensor:
        """Host draw from torch's global generator with the shape SB3's sampling uses
        (`Normal.rsample` -> standard normal `[n, act_dim]`). Discrete heads use one U(0,1) per row for
        inverse-CDF sampling (same distribution as `Categorical.sample`, different stream)."""
        if self.discrete:
            return th.rand(n)
        return th.distributions.utils._standard_normal((n, self.act_dim), dtype=th.float32, device="cpu")

    def draw_noise_into(self, out: th.Tensor) -> None:
        """`sample_noise` straight into a (pinned) contiguous host buffer: the same generator
        consumption as `torch.empty(shape).normal_()` / `.uniform_()` on a fresh tensor, minus the
        allocation and the copy."""
        if self.discrete:
            out.uniform_()
        else:
            out.normal_()

    def make_act_step(self, obs_tile: th.Tensor, noise_dev: th.Tensor, acts: th.Tensor, clipped: th.Tensor,
                      val: th.Tensor, logp: th.Tensor):
        """Rollout-step launcher over time-major tiles `[T(+1), n, ...]` with every pointer resolved once:
        `step(t)` is one ctypes call (eval mode: no statistics update). Used by `PPO.collect_rollouts`."""
        assert not (self.training and self.features_extractor.normalize is not None), \
            "the rollout step runs in eval mode (a train-mode forward would update the feature statistics)"
        lib, desc = L.load(), C.byref(self.desc)
        fn = lib.ia_policy_act
        n = obs_tile.shape[1]
        nm, nv = self._norm_ptrs()
        P, Pt = L.ptr(self._flat), L.ptr(self._flat_t)
        low, high = L.ptr(self._low), L.ptr(self._high)
        # `noise_dev`: one [n, A] tile refilled before every step, or [T, n, A] with the whole rollout's draws
        b_noise, s_noise = noise_dev.data_ptr(), (noise_dev.stride(0) * 4 if noise_dev.dim() == 3 else 0)
        b_obs, s_obs = obs_tile.data_ptr(), obs_tile.stride(0) * 4
        b_act, s_act = acts.data_ptr(), acts.stride(0) * 4
        b_clip, s_clip = clipped.data_ptr(), clipped.stride(0) * 4
        b_val, s_val = val.data_ptr(), val.stride(0) * 4
        b_lp, s_lp = logp.data_ptr(), logp.stride(0) * 4
        stream = L.stream()

        def step(t: int) -> None:
            rc = fn(desc, P, Pt, nm, nv, b_obs + t * s_obs, n, b_noise + t * s_noise, low, high, b_act + t * s_act,
                    b_clip + t * s_clip,
                    b_val + t * s_val, b_lp + t * s_lp, stream)
            if rc != 0:
                L.check(rc, "ia_policy_act")

        return step

    def make_rollout_mailbox(self, obs_tile: th.Tensor, noise_dev: th.Tensor, acts: th.Tensor, clipped: th.Tensor,
                             val: th.Tensor, logp: th.Tensor, T: int, timeout_s: float = 120.0,
                             last_val: Optional[th.Tensor] = None):
        """The rollout's act steps as ONE resident launch (`ia_policy_rollout_mailbox`) on the current stream: returns
        `(post, wait, close)` -- `post(t)` tells the device that step t's observations (and noise) are in their pinned
        tiles, `wait(t)` returns True once every workgroup has acknowledged step t (its clipped actions are in host memory;
        spins in C without the GIL) and False when the device has left the rollout (it waits `timeout_s` for a step, e.g.
        an environment that took minutes: the caller then runs this and the remaining steps as per-step launches, which
        rewrite the same values), `close()` aborts a kernel that has steps left (error paths) -- or None when the shape is
        not covered (the caller launches `make_act_step`'s kernel per step). Same tiles, same values as `make_act_step`.
        `last_val`: the kernel also evaluates V(`obs_tile[T]`) into it when the host posts step T (`post(T)`, no wait
        needed: the stream the kernel runs on is what later launches are ordered behind)."""
        assert not (self.training and self.features_extractor.normalize is not None), \
            "the rollout step runs in eval mode (a train-mode forward would update the feature statistics)"
        lib = L.load()
        n = obs_tile.shape[1]
        nblk = (n + 63) // 64
        ready, done = self._mailbox_flags_for(nblk)
        nm, nv = self._norm_ptrs()
        stride = lambda t: t.stride(0)
        rc = lib.ia_policy_rollout_mailbox(
            C.byref(self.desc), L.ptr(self._flat), L.ptr(self._flat_t), nm, nv, n, L.ptr(self._low), L.ptr(self._high),
            obs_tile.data_ptr(), stride(obs_tile), noise_dev.data_ptr(), stride(noise_dev) if noise_dev.dim() == 3 else 0,
            acts.data_ptr(), stride(acts), clipped.data_ptr(), stride(clipped), val.data_ptr(), stride(val),
            logp.data_ptr(), stride(logp), L.ptr(last_val), T, ready.data_ptr(), done.data_ptr(), float(timeout_s),
            L.stream())
        if rc == L.ERR_UNSUPPORTED:
            return None
        L.check(rc, "ia_policy_rollout_mailbox")
        ready_np = ready.numpy()
        done_ptr = done.data_ptr()
        wait_fn = lib.ia_host_wait_i32
        state = {"posted": 0, "gone": False}
        total = T + 1 if last_val is not None else T    # steps the kernel stays for
        stream_obj = th.cuda.current_stream()

        def post(t: int) -> None:
            ready_np[0] = t + 1
            state["posted"] = t + 1

        def wait(t: int) -> bool:
            rc_ = wait_fn(done_ptr, nblk, t + 1, float(timeout_s) + 30.0)
            if rc_ == 0:
                return True
            # rc -1: the device left on its own time-out. rc 1: the host's own (longer) time-out -- the device clock only
            # starts once the kernel runs, so a kernel still queued behind other work ends up here: abort it and take the
            # same per-step fallback (the caller synchronises the stream, where the kernel -- started or not -- sees -1)
            ready_np[0] = -1
            state["gone"] = True        # (nothing left to abort)
            return False

        def close() -> None:
            if not state["gone"] and state["posted"] < total:
                ready_np[0] = -1   # the kernel's workgroups leave at their next poll
                stream_obj.synchronize()   # ... and HAVE left before anybody can hand these flags to a new launch

        return post, wait, close

    @property
    def samples_on_host(self) -> bool:
        """Discrete head sampled by torch.multinomial on the host (the reference's RNG stream)."""
        return self.discrete and self.discrete_sampling == "multinomial"

    def make_multinomial_step(self, obs_tile: th.Tensor, h_logits: th.Tensor, h_clip: th.Tensor, val: th.Tensor,
                              h_logp: th.Tensor):
        """Rollout-step launcher for Discrete heads on the reference's sampling stream: `step(t)` = one launch
        (logits of the step's observations straight into pinned host memory + values), one wait, then
        [SB3 CategoricalDistribution] `sample()` / `log_prob()` on the host exactly as the reference composes
        them (`torch.distributions.Categorical(logits=...)`: `torch.multinomial` on the global generator).
        Actions land in `h_clip[t]` (fp32 indices), log-probs in `h_logp[t]` (both pinned host tiles)."""
        assert self.discrete and not (self.training and self.features_extractor.normalize is not None)
        lib, desc = L.load(), C.byref(self.desc)
        fn = lib.ia_policy_logits
        n = obs_tile.shape[1]
        nm, nv = self._norm_ptrs()
        P, Pt = L.ptr(self._flat), L.ptr(self._flat_t)
        b_obs, s_obs = obs_tile.data_ptr(), obs_tile.stride(0) * 4
        b_val, s_val = val.data_ptr(), val.stride(0) * 4
        lg = h_logits.data_ptr()
        stream_obj = th.cuda.current_stream()
        stream = stream_obj.cuda_stream

        h_logp_np, h_clip_np, rows_np = h_logp.numpy(), h_clip.numpy().reshape(h_clip.shape[0], n), np.arange(n)

        def step(t: int) -> None:
            rc = fn(desc, P, Pt, nm, nv, b_obs + t * s_obs, n, lg, b_val + t * s_val, stream)
            if rc != 0:
                L.check(rc, "ia_policy_logits")
            stream_obj.synchronize()
            categorical_sample_into(h_logits, h_logp_np[t], h_clip_np[t], rows_np)

        return step

    def _mailbox_flags_for(self, nblk: int):
        """A (ready, done) pair in pinned host memory, zeroed. Pairs ROTATE over three slots: a kernel that was told to
        abort (`ready = -1`) keeps seeing that value in ITS pair however soon the next rollout starts (`close()` also waits
        for it to leave; the rotation covers a `close()` that raised)."""
        ring = getattr(self, "_mailbox_ring", None)
        if ring is None or ring[0][1].numel() != nblk:
            ring = self._mailbox_ring = [(th.zeros(1, dtype=th.int32).pin_memory(), th.zeros(nblk, dtype=th.int32).pin_memory())
                                         for _ in range(3)]
            self._mailbox_next = 0
        box = ring[self._mailbox_next]
        self._mailbox_next = (self._mailbox_next + 1) % len(ring)
        box[0].zero_()
        box[1].zero_()
        return box

    def make_multinomial_mailbox(self, obs_tile: th.Tensor, h_logits: th.Tensor, h_clip: th.Tensor, val: th.Tensor,
                                 h_logp: th.Tensor, T: int, timeout_s: float = 120.0):
        """`make_multinomial_step` with the T launches folded into ONE resident kernel (`ia_policy_logits_mailbox`):
        `(post, wait, close)` as `make_rollout_mailbox`; `wait(t)` returns True after the step's logits have arrived in
        the pinned tile AND the host has sampled from them exactly as `make_multinomial_step` does (actions in
        `h_clip[t]`, log-probs in `h_logp[t]`). None: not covered."""
        assert self.discrete and not (self.training and self.features_extractor.normalize is not None)
        lib = L.load()
        n = obs_tile.shape[1]
        nblk = (n + 63) // 64
        ready, done = self._mailbox_flags_for(nblk)
        nm, nv = self._norm_ptrs()
        rc = lib.ia_policy_logits_mailbox(C.byref(self.desc), L.ptr(self._flat), L.ptr(self._flat_t), nm, nv, n,
                                          obs_tile.data_ptr(), obs_tile.stride(0), h_logits.data_ptr(), val.data_ptr(),
                                          val.stride(0), T, ready.data_ptr(), done.data_ptr(), float(timeout_s), L.stream())
        if rc == L.ERR_UNSUPPORTED:
            return None
        L.check(rc, "ia_policy_logits_mailbox")
        ready_np, done_ptr, wait_fn = ready.numpy(), done.data_ptr(), lib.ia_host_wait_i32
        state = {"acked": 0}
        stream_obj = th.cuda.current_stream()
        h_logp_np, h_clip_np, rows_np = h_logp.numpy(), h_clip.numpy().reshape(h_clip.shape[0], n), np.arange(n)

        def post(t: int) -> None:
            ready_np[0] = t + 1

        def wait(t: int) -> bool:
            rc_ = wait_fn(done_ptr, nblk, t + 1, float(timeout_s) + 30.0)
            if rc_ != 0:    # -1: the device's time-out; 1: the host's (kernel never started) -- same fallback, see above
                ready_np[0] = -1
                state["acked"] = T
                return False
            state["acked"] = t + 1
            categorical_sample_into(h_logits, h_logp_np[t], h_clip_np[t], rows_np)
            return True

        def close() -> None:
            if state["acked"] < T:
                ready_np[0] = -1
                stream_obj.synchronize()

        return post, wait, close

    def act(self, obs_dev: th.Tensor, noise_dev: th.Tensor, actions: th.Tensor, clipped: th.Tensor,
            values: th.Tensor, logp: th.Tensor) -> None:
        """Device-to-device rollout step (no allocation): fills actions/clipped/values/logp."""
        self._maybe_update_norm(obs_dev)
        nm, nv = self._norm_ptrs()
        L.call("ia_policy_act", C.byref(self.desc), L.ptr(self._flat), L.ptr(self._flat_t), nm, nv, L.ptr(obs_dev),
               obs_dev.shape[0], L.ptr(noise_dev), L.ptr(self._low), L.ptr(self._high), L.ptr(actions),
               L.ptr(clipped), L.ptr(values), L.ptr(logp), L.stream())

    def forward(self, obs, deterministic: bool = False):
        """[SB3 ActorCriticPolicy.forward] -> (actions, values, log_prob) device tensors."""
        require_device(self.device)
        o = self._obs_dev(obs)
        n = o.shape[0]
        if self.samples_on_host and not deterministic:
            self._maybe_update_norm(o)
            nm, nv = self._norm_ptrs()
            logits, vals = th.empty(n, self.act_dim, device=self.device), th.empty(n, device=self.device)
            L.call("ia_policy_logits", C.byref(self.desc), L.ptr(self._flat), L.ptr(self._flat_t), nm, nv, L.ptr(o), n,
                   L.ptr(logits), L.ptr(vals), L.stream())
            dist = th.distributions.Categorical(logits=logits.cpu())
            a = dist.sample()
            return (a.to(self.device).reshape((n, *self.action_space.shape)), vals.reshape(n, 1),
                    dist.log_prob(a).to(self.device))
        if deterministic:  # mode: zero Gaussian noise / negative uniform = argmax (no RNG draw, as in SB3)
            noise = th.full((n,), -1.0) if self.discrete else th.zeros(n, self.act_dim)
        else:
            noise = self.sample_noise(n)
        aw = 1 if self.discrete else self.act_dim
        acts, clip = th.empty(n, aw, device=self.device), th.empty(n, aw, device=self.device)
        vals, logp = th.empty(n, device=self.device), th.empty(n, device=self.device)
        self.act(o, noise.to(self.device), acts, clip, vals, logp)
        if self.discrete:
            acts = acts.reshape(n).long()
        return acts.reshape((n, *self.action_space.shape)), vals.reshape(n, 1), logp

    __call__ = forward

    def evaluate_actions(self, obs, actions):
        """[SB3 evaluate_actions] without autograd: (values [n,1], log_prob [n], entropy [n])."""
        require_device(self.device)
        o = self._obs_dev(obs)
        n = o.shape[0]
        a = actions if isinstance(actions, th.Tensor) else th.as_tensor(np.ascontiguousarray(actions))
        a = a.to(self.device, th.float32).reshape(n, -1).contiguous()
        self._maybe_update_norm(o)
        nm, nv = self._norm_ptrs()
        logp, vals, ent = (th.empty(n, device=self.device) for _ in range(3))
        L.call("ia_policy_evaluate", C.byref(self.desc), L.ptr(self._flat), L.ptr(self._flat_t), nm, nv, L.ptr(o),
               L.ptr(a), n, L.ptr(logp), L.ptr(vals), L.ptr(ent), L.stream())
        return vals.reshape(n, 1), logp, ent

    def log_prob_rows(self, obs_dev: th.Tensor, acts_dev: th.Tensor, out: th.Tensor,
                      norm_snapshot: Optional[th.Tensor] = None) -> None:
        """log pi(a|s) for device rows (train-mode norm update included), no allocation.
        `norm_snapshot` `[2, obs_dim]` (mean, var): normalise with these statistics and leave the live ones
        alone -- the train-mode update this call would have made was already applied elsewhere
        (`ia_running_norm_merge_seq` snapshots, pipelined AIRL rounds)."""
        if norm_snapshot is None:
            self._maybe_update_norm(obs_dev)
            nm, nv = self._norm_ptrs()
        else:
            assert norm_snapshot.is_contiguous() and norm_snapshot.shape == (2, self.obs_dim)
            nm, nv = norm_snapshot.data_ptr(), norm_snapshot.data_ptr() + 4 * self.obs_dim
        L.call("ia_policy_evaluate", C.byref(self.desc), L.ptr(self._flat), L.ptr(self._flat_t), nm, nv,
               L.ptr(obs_dev), L.ptr(acts_dev), obs_dev.shape[0], L.ptr(out), None, None, L.stream())

    def predict_values(self, obs) -> th.Tensor:
        require_device(self.device)
        o = self._obs_dev(obs)
        self._maybe_update_norm(o)
        vals = th.empty(o.shape[0], device=self.device)
        self.values_rows(o, vals)
        return vals.reshape(-1, 1)

    def values_rows(self, obs_dev: th.Tensor, out: th.Tensor) -> None:
        nm, nv = self._norm_ptrs()
        L.call("ia_policy_evaluate", C.byref(self.desc), L.ptr(self._flat), L.ptr(self._flat_t), nm, nv,
               L.ptr(obs_dev), None, obs_dev.shape[0], None, L.ptr(out), None, L.stream())

    def predict(self, observation, state=None, episode_start=None, deterministic: bool = False):
        """[SB3 BasePolicy.predict]: numpy in, clipped numpy actions out."""
        self.set_training_mode(False)
        obs = np.asarray(observation)
        vectorized = obs.shape != tuple(self.observation_space.shape)
        obs = obs.reshape((-1, *self.observation_space.shape))
        acts, _, _ = self.forward(obs, deterministic=deterministic)
        acts = acts.cpu().numpy()
        if not self.discrete:
            acts = np.clip(acts, self.action_space.low, self.action_space.high)
        return (acts if vectorized else acts[0]), state


class FeedForward32Policy(ActorCriticPolicy):
    """`policies/base.py:92-104`."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs, net_arch=[32, 32])
