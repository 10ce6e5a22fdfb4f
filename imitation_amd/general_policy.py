"""Actor-critic policies whose `net_arch` the fused policy kernels do not cover.

[SB3 ActorCriticPolicy] accepts any `net_arch` -- a list (both towers alike) or `dict(pi=[...], vf=[...])` of any depth
and widths -- and `nn.Tanh` or `nn.ReLU` towers. The reference's GAIL/AIRL configs use `[32, 32]` / `[64, 64]` tanh
towers, which `policies.ActorCriticPolicy` runs on the fused kernels of `csrc/policy.hip`; every other shape lands
here. Same flat parameter buffer in torch `parameters()` order, same construction order (=> same initial weights on
the same torch seed), same `state_dict` keys; what differs is the execution:

    features (RunningNorm)      ia_running_norm_update / _apply
    pi tower + action_net       ia_mlp_forward / ia_mlp_backward       (fp32 MFMA GEMMs, one stack)
    vf tower + value_net        ia_mlp_forward / ia_mlp_backward       (one stack)
    DiagGaussian / Categorical  ia_gauss_act / ia_gauss_eval / ia_categorical_loss   (csrc/ppo_general.hip)
    PPO minibatch               ia_gather_rows, ia_adv_moments, ia_ppo_head_loss, ia_clip_grad_norm, ia_adam_step

i.e. [SB3 PPO.train]'s loop body as ~25 launches per minibatch instead of one persistent launch per update -- the
generality path, not the benchmark path. `ActorCriticPolicy.__init__` switches an instance to this class when it
sees such a `net_arch` (`adopt`), so user code keeps constructing `ActorCriticPolicy` / `PPO("MlpPolicy", ...)`.
"""
from __future__ import annotations

import ctypes as C
import functools
from typing import Dict, Iterator, List, Optional, Tuple

import numpy as np
import torch as th
from torch import nn

from imitation_amd import _lib as L
from imitation_amd.networks import HipAdam, require_device


def fused_arch(pi, vf, activation_fn) -> bool:
    """Two equal tanh towers [H, H], H in (32, 64): the shapes of csrc/policy.hip."""
    return (activation_fn is nn.Tanh and len(pi) == len(vf) == 2 and pi[0] == pi[1] == vf[0] == vf[1]
            and pi[0] in (32, 64))


def head_grad_of_coefficients(discrete: bool, out: th.Tensor, log_std_ptr, values: th.Tensor, acts: th.Tensor,
                              logp: th.Tensor, n: int, A: int, logp_coef: float, ent_coef: float, ws) -> None:
    """d/d(head outputs, log_std) of `logp_coef * sum(log_prob) + ent_coef * sum(entropy)` (the BC loss terms) through
    `ia_ppo_head_loss`: with old log-probs = current ones the ratio is 1 and the clipped surrogate's gradient is
    `-adv / n` per row, so `adv = -logp_coef * n`, PPO's `ent_coef = -ent_coef * n`, `vf_coef = 0`, no clipping, no
    advantage normalisation give exactly that gradient for Categorical and DiagGaussian heads alike. Results land in
    `ws["d_out"]` (`[n, A]`) and, for Box heads, `ws["dls"]` (`[A]`)."""
    dev = out.device
    for k, shape in (("d_out", (n, A)), ("d_val", (n, 1)), ("adv", (n,)), ("ret", (n,)), ("dls", (max(A, 1),))):
        if k not in ws or ws[k].shape != shape:
            ws[k] = th.empty(*shape, device=dev)
    if "loss_ws" not in ws:
        ws["loss_ws"] = th.empty(int(L.load().ia_ppo_head_loss_ws_floats(n)), device=dev)
    ws["adv"].fill_(-float(logp_coef) * n)
    ws["ret"].zero_()   # (enters only the value-loss terms, which carry coefficient 0)
    L.call("ia_ppo_head_loss", int(discrete), L.ptr(out), log_std_ptr, L.ptr(values), L.ptr(acts), L.ptr(logp), L.ptr(ws["adv"]),
           L.ptr(ws["ret"]), None, n, A, 1.0e30, -float(ent_coef) * n, 0.0, L.ptr(ws["d_out"]), L.ptr(ws["d_val"]),
           None if discrete else L.ptr(ws["dls"]), L.ptr(ws["loss_ws"]), None, L.stream())


def adopt(policy, pi, vf, activation_fn, ortho_init: bool, log_std_init: float) -> None:
    """Re-class `policy` (an `ActorCriticPolicy` whose common attributes are set) to the general-tower execution,
    keeping the user's class in the MRO, and build its parameters."""
    cls = type(policy)
    if not issubclass(cls, GeneralTowers):
        policy.__class__ = type(cls.__name__, (GeneralTowers, cls), {"__module__": cls.__module__})
    policy._build_general(pi, vf, activation_fn, ortho_init, log_std_init)


# tuning / tests: False keeps the PPO update of the generic stacks as individual launches (no hipGraph replay)
GRAPH_UPDATES = True


class GeneralTowers:
    fused = False

    # ---- construction ----------------------------------------------------------------------------------------
    def _build_general(self, pi, vf, activation_fn, ortho_init: bool, log_std_init: float) -> None:
        if activation_fn not in (nn.Tanh, nn.ReLU):
            raise NotImplementedError("policy towers on the HIP path are nn.Tanh or nn.ReLU")
        pi, vf = [int(h) for h in pi], [int(h) for h in vf]
        if len(pi) > L.IA_MAX_LAYERS - 1 or len(vf) > L.IA_MAX_LAYERS - 1:
            raise NotImplementedError(f"at most {L.IA_MAX_LAYERS - 1} hidden layers per tower")
        D, A = self.obs_dim, self.act_dim
        self.pi_arch, self.vf_arch = pi, vf
        self.hidden, self.desc = None, None
        self._hidden_act = L.ACT_TANH if activation_fn is nn.Tanh else L.ACT_RELU

        # Host construction in SB3's order (pi tower, vf tower, action_net, log_std, value_net), then the
        # orthogonal re-initialisation: the torch global generator is consumed exactly as SB3 consumes it.
        def tower(sizes):
            mods, last = [], D
            for h in sizes:
                mods += [nn.Linear(last, h), activation_fn()]
                last = h
            return nn.Sequential(*mods), last

        pi_net, lat_pi = tower(pi)
        vf_net, lat_vf = tower(vf)
        action_net = nn.Linear(lat_pi, A)
        log_std = None if self.discrete else th.ones(A) * log_std_init
        value_net = nn.Linear(lat_vf, 1)
        if ortho_init:
            def init(m, gain):
                if isinstance(m, nn.Linear):
                    nn.init.orthogonal_(m.weight, gain=gain)
                    m.bias.data.fill_(0.0)
            for mod, gain in ((pi_net, np.sqrt(2)), (vf_net, np.sqrt(2)), (action_net, 0.01), (value_net, 1)):
                mod.apply(functools.partial(init, gain=gain))
        parts: List[th.Tensor] = [] if self.discrete else [log_std]
        lins = [m for m in pi_net if isinstance(m, nn.Linear)] + [m for m in vf_net if isinstance(m, nn.Linear)]
        for lin in lins + [action_net, value_net]:
            parts += [lin.weight.detach().reshape(-1), lin.bias.detach().reshape(-1)]
        self._flat = th.cat(parts).contiguous()
        self._flat_t = None

        def count(sizes):
            n, last = 0, D
            for h in sizes:
                n += last * h + h
                last = h
            return n

        n_ls = 0 if self.discrete else A
        self._n_pi, self._n_vf = count(pi), count(vf)
        self._n_an, self._n_vn = lat_pi * A + A, lat_vf + 1
        self._o_pi = n_ls
        self._o_vf = self._o_pi + self._n_pi
        self._o_an = self._o_vf + self._n_vf
        self._o_vn = self._o_an + self._n_an
        assert self._o_vn + self._n_vn == self._flat.numel()
        self._desc_pi = L.mlp_desc([D] + pi + [A], self._hidden_act)
        self._desc_vf = L.mlp_desc([D] + vf + [1], self._hidden_act)
        self._hid_pi, self._hid_vf = sum(pi), sum(vf)
        self._pi_stack: Optional[th.Tensor] = None   # [pi tower | action_net] contiguous: the stack ia_mlp_* reads
        self._vf_stack: Optional[th.Tensor] = None   # [vf tower | value_net]
        self._ws: Dict[Tuple[str, int], Dict[str, th.Tensor]] = {}

    def _layout(self) -> List[Tuple[str, Tuple[int, ...]]]:
        D, A = self.obs_dim, self.act_dim
        lay: List[Tuple[str, Tuple[int, ...]]] = [] if self.discrete else [("log_std", (A,))]
        for tower, sizes in (("policy_net", self.pi_arch), ("value_net", self.vf_arch)):
            last = D
            for i, h in enumerate(sizes):   # Linear modules sit at the even positions of the nn.Sequential
                lay += [(f"mlp_extractor.{tower}.{2 * i}.weight", (h, last)), (f"mlp_extractor.{tower}.{2 * i}.bias", (h,))]
                last = h
        lat_pi = self.pi_arch[-1] if self.pi_arch else D
        lat_vf = self.vf_arch[-1] if self.vf_arch else D
        lay += [("action_net.weight", (A, lat_pi)), ("action_net.bias", (A,)), ("value_net.weight", (1, lat_vf)),
                ("value_net.bias", (1,))]
        return lay

    def to(self, device):
        device = th.device(device)
        self._flat = self._flat.to(device).contiguous()
        if self.features_extractor.normalize is not None:
            self.features_extractor.normalize.to(device)
        if device.type == "cuda":
            self._pi_stack = th.empty(self._n_pi + self._n_an, device=device)
            self._vf_stack = th.empty(self._n_vf + self._n_vn, device=device)
            self._g_pi, self._g_vf = th.empty_like(self._pi_stack), th.empty_like(self._vf_stack)
            self._sync_transposed()
            self.optimizer = HipAdam(self._flat, th.zeros_like(self._flat), lr=self._lr0, **self.optimizer_kwargs)
            if not self.discrete:
                self._low = th.as_tensor(self.action_space.low.reshape(-1), dtype=th.float32, device=device)
                self._high = th.as_tensor(self.action_space.high.reshape(-1), dtype=th.float32, device=device)
            else:
                self._low = self._high = th.zeros(self.act_dim, device=device)
            self._ws = {}
            self.__dict__.pop("_update_graphs", None)   # captured updates hold the old buffers' addresses
        return self

    def _sync_transposed(self) -> None:
        """Refresh the two contiguous stacks from the flat buffer (after a load, a broadcast or an optimiser step)."""
        if self._pi_stack is None:
            return
        f = self._flat
        if f.is_cuda:   # the four pieces in one launch (four device-to-device copies per optimiser step before)
            key = (f.data_ptr(), self._pi_stack.data_ptr(), self._vf_stack.data_ptr())
            if getattr(self, "_sync_args", (None,))[0] != key:
                offs = (self._o_pi, self._o_an, self._o_vf, self._o_vn)
                dsts = (self._pi_stack.data_ptr(), self._pi_stack.data_ptr() + 4 * self._n_pi,
                        self._vf_stack.data_ptr(), self._vf_stack.data_ptr() + 4 * self._n_vf)
                self._sync_args = (key, (C.c_void_p * 4)(*(f.data_ptr() + 4 * o for o in offs)), (C.c_void_p * 4)(*dsts),
                                   (C.c_int64 * 4)(self._n_pi, self._n_an, self._n_vf, self._n_vn))
            _, src, dst, ns = self._sync_args
            L.call("ia_copy_pieces", 4, src, dst, ns, L.stream())
            return
        self._pi_stack[: self._n_pi].copy_(f[self._o_pi:self._o_pi + self._n_pi])
        self._pi_stack[self._n_pi:].copy_(f[self._o_an:self._o_an + self._n_an])
        self._vf_stack[: self._n_vf].copy_(f[self._o_vf:self._o_vf + self._n_vf])
        self._vf_stack[self._n_vf:].copy_(f[self._o_vn:self._o_vn + self._n_vn])

    # ---- forward pieces --------------------------------------------------------------------------------------
    def _buffers(self, tag: str, n: int) -> Dict[str, th.Tensor]:
        ws = self._ws.get((tag, n))
        if ws is None:
            dev, D, A = self.device, self.obs_dim, self.act_dim
            aw = 1 if self.discrete else A
            ws = dict(xn=th.empty(n, D, device=dev), out=th.empty(n, A, device=dev), val=th.empty(n, 1, device=dev),
                      hid_pi=th.empty(max(1, n * self._hid_pi), device=dev),
                      hid_vf=th.empty(max(1, n * self._hid_vf), device=dev))
            if tag == "step":
                ws.update(obs=th.empty(n, D, device=dev), noise=th.empty(n, A, device=dev),
                          act=th.empty(n, aw, device=dev), clip=th.empty(n, aw, device=dev))
            if tag == "train":
                splits = max(1, min(32, n // 256))
                ws.update(obs=th.empty(n, D, device=dev), act=th.empty(n, aw, device=dev),
                          old=th.empty(n, device=dev), adv=th.empty(n, device=dev), ret=th.empty(n, device=dev),
                          d_out=th.empty(n, A, device=dev), d_val=th.empty(n, 1, device=dev),
                          dhid=th.empty(max(1, n * max(self._hid_pi, self._hid_vf)), device=dev),
                          part=th.empty(splits, max(self._pi_stack.numel(), self._vf_stack.numel()), device=dev),
                          ms=th.empty(2, device=dev), splits=splits,
                          loss_ws=th.empty(int(L.load().ia_ppo_head_loss_ws_floats(n)), device=dev))
            self._ws[(tag, n)] = ws
        return ws

    def _run(self, o: th.Tensor, n: int, ws, pi: bool = True, vf: bool = True, norm=None) -> None:
        """Normalised features, then the requested stacks: `ws["out"]` (means / logits), `ws["val"]`."""
        D, s = self.obs_dim, L.stream()
        rn = self.features_extractor.normalize
        x = o
        if rn is not None:
            mean, var = (L.ptr(rn.running_mean), L.ptr(rn.running_var)) if norm is None else norm
            L.call("ia_running_norm_apply", L.ptr(o), D, n, D, mean, var, float(rn.eps), L.ptr(ws["xn"]), D, s)
            x = ws["xn"]
        ws["x"] = x
        if pi:
            L.call("ia_mlp_forward", C.byref(self._desc_pi), L.ptr(self._pi_stack), L.ptr(x), D, n, L.ptr(ws["hid_pi"]),
                   L.ptr(ws["out"]), L.ACT_NONE, s)
        if vf:
            L.call("ia_mlp_forward", C.byref(self._desc_vf), L.ptr(self._vf_stack), L.ptr(x), D, n, L.ptr(ws["hid_vf"]),
                   L.ptr(ws["val"]), L.ACT_NONE, s)

    def _log_std_ptr(self):
        return None if self.discrete else L.ptr(self._flat)   # log_std = the first act_dim entries

    @property
    def samples_on_host(self) -> bool:
        """Discrete heads always sample with torch.multinomial on the host (the reference's stream)."""
        return self.discrete

    # ---- rollout steps ---------------------------------------------------------------------------------------
    def make_rollout_mailbox(self, *args, **kwargs):
        """(the resident rollout kernel is the fused towers' act kernel: general towers launch per step)"""
        return None

    def make_multinomial_mailbox(self, *args, **kwargs):
        return None

    def make_act_step(self, obs_tile: th.Tensor, noise_host: th.Tensor, acts: th.Tensor, clipped: th.Tensor,
                      val: th.Tensor, logp: th.Tensor):
        """Rollout-step launcher (Box heads): stage this step's observations and noise from the pinned host tiles,
        both stacks, the Gaussian head; actions / values / log-probs into the device tiles, clipped actions into
        the pinned host tile the env workers read. Eval mode (no statistics update)."""
        assert not self.discrete, "Discrete general-tower policies sample on the host (make_multinomial_step)"
        assert not (self.training and self.features_extractor.normalize is not None)
        n, A = obs_tile.shape[1], self.act_dim
        ws = self._buffers("step", n)
        stream_obj = th.cuda.current_stream()   # the stream this launcher was made on (PPO: its act stream)

        def step(t: int) -> None:
            with th.cuda.stream(stream_obj):
                ws["obs"].copy_(obs_tile[t], non_blocking=True)
                ws["noise"].copy_((noise_host[t] if noise_host.dim() == 3 else noise_host).reshape(n, A), non_blocking=True)
                self._run(ws["obs"], n, ws)
                L.call("ia_gauss_act", L.ptr(ws["out"]), self._log_std_ptr(), L.ptr(ws["noise"]), L.ptr(self._low),
                       L.ptr(self._high), n, A, L.ptr(acts[t]), L.ptr(ws["clip"]), L.ptr(logp[t]), L.stream())
                val[t].copy_(ws["val"].reshape(n))
                clipped[t].copy_(ws["clip"], non_blocking=True)

        return step

    def make_multinomial_step(self, obs_tile: th.Tensor, h_logits: th.Tensor, h_clip: th.Tensor, val: th.Tensor,
                              h_logp: th.Tensor):
        """Discrete heads on the reference's sampling stream (see `ActorCriticPolicy.make_multinomial_step`)."""
        assert self.discrete and not (self.training and self.features_extractor.normalize is not None)
        n = obs_tile.shape[1]
        ws = self._buffers("step", n)
        stream_obj = th.cuda.current_stream()
        h_logp_np, h_clip_np, rows_np = h_logp.numpy(), h_clip.numpy().reshape(h_clip.shape[0], n), np.arange(n)

        def step(t: int) -> None:
            with th.cuda.stream(stream_obj):
                ws["obs"].copy_(obs_tile[t], non_blocking=True)
                self._run(ws["obs"], n, ws)
                h_logits.copy_(ws["out"], non_blocking=True)
                val[t].copy_(ws["val"].reshape(n))
            stream_obj.synchronize()
            from imitation_amd.policies import categorical_sample_into   # (local: policies imports this module)
            categorical_sample_into(h_logits, h_logp_np[t], h_clip_np[t], rows_np)

        return step

    def act(self, obs_dev: th.Tensor, noise_dev: th.Tensor, actions: th.Tensor, clipped: th.Tensor,
            values: th.Tensor, logp: th.Tensor) -> None:
        assert not self.discrete
        n = obs_dev.shape[0]
        ws = self._buffers("eval", n)
        self._maybe_update_norm(obs_dev)
        self._run(obs_dev, n, ws)
        L.call("ia_gauss_act", L.ptr(ws["out"]), self._log_std_ptr(), L.ptr(noise_dev), L.ptr(self._low), L.ptr(self._high),
               n, self.act_dim, L.ptr(actions), L.ptr(clipped), L.ptr(logp), L.stream())
        values.reshape(n).copy_(ws["val"].reshape(n))

    def forward(self, obs, deterministic: bool = False):
        """[SB3 ActorCriticPolicy.forward] -> (actions, values, log_prob) device tensors."""
        require_device(self.device)
        o = self._obs_dev(obs)
        n = o.shape[0]
        ws = self._buffers("eval", n)
        if self.discrete:
            self._maybe_update_norm(o)
            self._run(o, n, ws)
            if deterministic:   # [SB3 CategoricalDistribution.mode]: argmax of the probabilities
                a_dev = th.argmax(ws["out"], dim=1)
                lp, a_f = th.empty(n, device=self.device), a_dev.float()
                self._head_eval(ws, a_f, n, lp, None)
            else:
                dist = th.distributions.Categorical(logits=ws["out"].cpu())
                a = dist.sample()
                a_dev, lp = a.to(self.device), dist.log_prob(a).to(self.device)
            return a_dev.reshape((n, *self.action_space.shape)), ws["val"].clone(), lp
        noise = th.zeros(n, self.act_dim) if deterministic else self.sample_noise(n)
        acts, clip = th.empty(n, self.act_dim, device=self.device), th.empty(n, self.act_dim, device=self.device)
        vals, logp = th.empty(n, device=self.device), th.empty(n, device=self.device)
        self.act(o, noise.to(self.device), acts, clip, vals, logp)
        return acts.reshape((n, *self.action_space.shape)), vals.reshape(n, 1), logp

    __call__ = forward

    # ---- evaluation ------------------------------------------------------------------------------------------
    def _head_eval(self, ws, a: th.Tensor, n: int, logp, entropy) -> None:
        if self.discrete:
            ent = entropy if entropy is not None else th.empty(n, device=self.device)
            L.call("ia_categorical_loss", L.ptr(ws["out"]), self.act_dim, L.ptr(a), n, self.act_dim, 0.0, 0.0, L.ptr(logp),
                   L.ptr(ent), None, L.stream())
        else:
            L.call("ia_gauss_eval", L.ptr(ws["out"]), self._log_std_ptr(), L.ptr(a), n, self.act_dim, L.ptr(logp),
                   L.ptr(entropy), L.stream())

    def evaluate_actions(self, obs, actions, logp_coef: float = 0.0, ent_coef: float = 0.0, want_grad: bool = False):
        """[SB3 evaluate_actions] without autograd: (values [n,1], log_prob [n], entropy [n]). With `want_grad` (the BC
        step, `bc.py:138-156`), the gradient of `logp_coef * sum(log_prob) + ent_coef * sum(entropy)` w.r.t. the head
        outputs and log_std is left for `backward()`."""
        require_device(self.device)
        o = self._obs_dev(obs)
        n = o.shape[0]
        a = actions if isinstance(actions, th.Tensor) else th.as_tensor(np.ascontiguousarray(actions))
        a = a.to(self.device, th.float32).reshape(n, -1).contiguous()
        self._maybe_update_norm(o)
        ws = self._buffers("train" if want_grad else "eval", n)
        self._run(o, n, ws)
        logp, ent = th.empty(n, device=self.device), th.empty(n, device=self.device)
        self._head_eval(ws, a, n, logp, ent)
        if want_grad:
            head_grad_of_coefficients(self.discrete, ws["out"], self._log_std_ptr(), ws["val"], a, logp, n, self.act_dim,
                                      logp_coef, ent_coef, ws)
        return ws["val"].clone(), logp, ent

    def backward(self, B: int, grad: th.Tensor) -> None:
        """Adds to `grad` (flat) the parameter gradient of the loss whose head gradient the last
        `evaluate_actions(..., want_grad=True)` on `B` rows left behind (no value term: the vf stack gets none)."""
        ws = self._buffers("train", B)
        s, sp, P = L.stream(), ws["splits"], self._pi_stack.numel()
        part = ws["part"].reshape(-1)[: sp * P]
        L.call("ia_mlp_backward", C.byref(self._desc_pi), L.ptr(self._pi_stack), L.ptr(ws["x"]), self.obs_dim, B,
               L.ptr(ws["hid_pi"]), L.ptr(ws["d_out"]), L.ptr(ws["dhid"]), L.ptr(part), sp, None, s)
        L.call("ia_reduce_partials", L.ptr(part), sp, P, 1.0, 0, L.ptr(self._g_pi), s)
        for (o0, n0), lo in (((self._o_pi, self._n_pi), 0), ((self._o_an, self._n_an), self._n_pi)):
            L.call("ia_reduce_partials", L.ptr(self._g_pi[lo:lo + n0]), 1, n0, 1.0, 1, L.ptr(grad[o0:o0 + n0]), s)
        if not self.discrete:
            L.call("ia_reduce_partials", L.ptr(ws["dls"]), 1, self.act_dim, 1.0, 1, L.ptr(grad), s)

    def log_prob_rows(self, obs_dev: th.Tensor, acts_dev: th.Tensor, out: th.Tensor,
                      norm_snapshot: Optional[th.Tensor] = None) -> None:
        n = obs_dev.shape[0]
        norm = None
        if norm_snapshot is None:
            self._maybe_update_norm(obs_dev)
        else:
            assert norm_snapshot.is_contiguous() and norm_snapshot.shape == (2, self.obs_dim)
            norm = (norm_snapshot.data_ptr(), norm_snapshot.data_ptr() + 4 * self.obs_dim)
        ws = self._buffers("eval", n)
        self._run(obs_dev, n, ws, vf=False, norm=norm)
        a = acts_dev if acts_dev.dtype == th.float32 else acts_dev.float()
        self._head_eval(ws, a.reshape(n, -1).contiguous(), n, out, None)

    def values_rows(self, obs_dev: th.Tensor, out: th.Tensor) -> None:
        n = obs_dev.shape[0]
        ws = self._buffers("eval", n)
        self._run(obs_dev, n, ws, pi=False)
        out.reshape(n).copy_(ws["val"].reshape(n))

    # ---- [SB3 PPO.train] ---------------------------------------------------------------------------------------
    def ppo_update(self, rb, perm_dev: th.Tensor, n_epochs: int, batch_size: int, normalize_advantage: bool,
                   clip_range: float, ent_coef: float, vf_coef: float, max_grad_norm: float, stats: th.Tensor,
                   dp=None) -> None:
        """`_ppo_update_launches` (the ~35 launches per minibatch of the generic stacks are host-bound: ~250 us per
        step of launch overhead against ~100 us of kernels), replayed as ONE hipGraph from the third update with the
        same configuration on: the first runs eagerly (it also performs every kernel's one-time attribute set-up),
        the second is captured -- Adam then takes its step-dependent scalars from a device-side step count
        (`HipAdam.begin_device_steps`) -- and every later one is a single graph launch. Every buffer the sequence
        touches is persistent (rollout tile, permutation, workspaces, parameters); the statistics land in a buffer
        owned by the graph and are copied out. Not used under data parallelism (collectives between the launches)."""
        opt = self.optimizer
        n_steps = n_epochs * -(-(rb.buffer_size * rb.n_envs) // batch_size)
        g0 = opt.param_groups[0] if isinstance(opt, HipAdam) else None
        if not (GRAPH_UPDATES and g0 is not None and (dp is None or dp.world == 1) and self._flat.is_cuda):
            return self._ppo_update_launches(rb, perm_dev, n_epochs, batch_size, normalize_advantage, clip_range,
                                             ent_coef, vf_coef, max_grad_norm, stats, dp)
        # every buffer the captured launches read or write is part of the key (a rebuilt rollout tile, a second `.to()`
        # or a reloaded optimiser must not replay into freed memory); the learning rate is NOT: it reaches the captured
        # Adam launches through device memory (`HipAdam.sync_device_step`), so a schedule keeps replaying
        key = (rb.obs.data_ptr(), rb.acts.data_ptr(), rb.logp.data_ptr(), rb.adv.data_ptr(), rb.ret.data_ptr(),
               rb.buffer_size, rb.n_envs, perm_dev.data_ptr(), self._flat.data_ptr(), self._pi_stack.data_ptr(),
               self._vf_stack.data_ptr(), opt.exp_avg.data_ptr(), opt.exp_avg_sq.data_ptr(), opt.grad.data_ptr(), n_epochs,
               batch_size, bool(normalize_advantage), float(clip_range), float(ent_coef), float(vf_coef),
               float(max_grad_norm), tuple(stats.shape), bool(self.training), tuple(g0["betas"]), float(g0["eps"]),
               float(g0["weight_decay"]), th.cuda.current_device())
        graphs = self.__dict__.setdefault("_update_graphs", {})
        entry = graphs.get(key)
        if entry is None:
            if len(graphs) >= 4:     # (schedules that change every round would capture forever)
                graphs.clear()
            graphs[key] = "warm"
            return self._ppo_update_launches(rb, perm_dev, n_epochs, batch_size, normalize_advantage, clip_range,
                                             ent_coef, vf_coef, max_grad_norm, stats, dp)
        if entry == "warm":
            buf = th.zeros_like(stats)
            graph = th.cuda.CUDAGraph()
            opt.begin_device_steps()
            try:
                th.cuda.synchronize()
                with th.cuda.graph(graph, capture_error_mode="thread_local"):
                    self._ppo_update_launches(rb, perm_dev, n_epochs, batch_size, normalize_advantage, clip_range,
                                              ent_coef, vf_coef, max_grad_norm, buf, None)
            finally:
                opt.end_device_steps(0)
            entry = graphs[key] = (graph, buf)
        graph, buf = entry
        opt.sync_device_step()
        graph.replay()
        opt.step_count += n_steps
        stats.copy_(buf)

    def _ppo_update_launches(self, rb, perm_dev: th.Tensor, n_epochs: int, batch_size: int, normalize_advantage: bool,
                             clip_range: float, ent_coef: float, vf_coef: float, max_grad_norm: float, stats: th.Tensor,
                             dp=None) -> None:
        """All epochs x minibatches of one PPO update on the rollout tile `rb` (time-major `[T, n_envs, ...]`);
        `perm_dev[e]` = the epoch's `np.random.permutation(T * n_envs)` over SB3's env-major flattening
        ([SB3 RolloutBuffer.swap_and_flatten]: index i = env * T + t). `stats[e, mb, :8]` receives the minibatch
        statistics. Train mode: every minibatch forward first updates the feature statistics."""
        T, n = rb.buffer_size, rb.n_envs
        total, D, A = T * n, self.obs_dim, self.act_dim
        aw = 1 if self.discrete else A
        s = L.stream()
        offs = (perm_dev % T) * n + perm_dev // T          # time-major row of every permuted index
        obs_rows, act_rows = rb.obs.reshape((T + 1) * n, D), rb.acts.reshape(total, aw)
        vecs = [(rb.logp.reshape(total, 1), "old"), (rb.adv.reshape(total, 1), "adv"), (rb.ret.reshape(total, 1), "ret")]
        rn = self.features_extractor.normalize
        opt = self.optimizer
        grad = opt.grad
        if getattr(self, "_clip_ws", None) is None:   # partials of the grid-wide gradient norm (long gradients)
            self._clip_ws = th.empty(int(L.load().ia_clip_grad_norm_ws_floats()), device=self.device)
        clip_ws = self._clip_ws
        for e in range(n_epochs):
            for mb, start in enumerate(range(0, total, batch_size)):
                b = min(batch_size, total - start)
                ws = self._buffers("train", b)
                idx = offs[e, start:start + b]
                L.call("ia_gather_rows", L.ptr(obs_rows), L.ptr(idx), b, D, L.ptr(ws["obs"]), s)
                L.call("ia_gather_rows", L.ptr(act_rows), L.ptr(idx), b, aw, L.ptr(ws["act"]), s)
                for src, key in vecs:
                    L.call("ia_gather_rows", L.ptr(src), L.ptr(idx), b, 1, L.ptr(ws[key]), s)
                if rn is not None and self.training:
                    rn.update_stats(ws["obs"])
                self._run(ws["obs"], b, ws)
                ms = None
                if normalize_advantage and b > 1:
                    L.call("ia_adv_moments", L.ptr(ws["adv"]), b, L.ptr(ws["ms"]), s)
                    ms = L.ptr(ws["ms"])
                L.call("ia_ppo_head_loss", int(self.discrete), L.ptr(ws["out"]), self._log_std_ptr(), L.ptr(ws["val"]),
                       L.ptr(ws["act"]), L.ptr(ws["old"]), L.ptr(ws["adv"]), L.ptr(ws["ret"]), ms, b, A,
                       float(clip_range), float(ent_coef), float(vf_coef), L.ptr(ws["d_out"]), L.ptr(ws["d_val"]),
                       None if self.discrete else L.ptr(grad), L.ptr(ws["loss_ws"]), L.ptr(stats[e, mb]), s)
                sp = ws["splits"]
                for desc, stack, hid, dout, g, pieces in (
                        (self._desc_pi, self._pi_stack, ws["hid_pi"], ws["d_out"], self._g_pi,
                         ((self._o_pi, self._n_pi), (self._o_an, self._n_an))),
                        (self._desc_vf, self._vf_stack, ws["hid_vf"], ws["d_val"], self._g_vf,
                         ((self._o_vf, self._n_vf), (self._o_vn, self._n_vn)))):
                    P = stack.numel()
                    part = ws["part"].reshape(-1)[: sp * P]
                    L.call("ia_mlp_backward", C.byref(desc), L.ptr(stack), L.ptr(ws["x"]), D, b, L.ptr(hid), L.ptr(dout),
                           L.ptr(ws["dhid"]), L.ptr(part), sp, None, s)
                    # the tower's slabs reduced straight into the two pieces of the flat gradient (its layers | its head): the
                    # same sums as one reduction of the whole stack followed by two copies
                    (o0, n0), (o1, n1) = pieces
                    if n0:   # (a tower without hidden layers has no first piece)
                        L.call("ia_reduce_partials_strided", L.ptr(part), sp, n0, P, 1.0, 0, L.ptr(grad[o0:o0 + n0]), s)
                    L.call("ia_reduce_partials_strided", L.ptr(part[n0:]), sp, n1, P, 1.0, 0, L.ptr(grad[o1:o1 + n1]), s)
                if dp is not None and dp.world > 1:
                    dp.allreduce_mean_(grad)
                L.call("ia_clip_grad_norm", L.ptr(grad), grad.numel(), float(max_grad_norm), None, L.ptr(clip_ws), s)
                opt.step()
                self._sync_transposed()

    def named_parameters(self) -> Iterator[Tuple[str, th.Tensor]]:
        o = 0
        for name, shape in self._layout():
            k = int(np.prod(shape))
            yield name, self._flat[o:o + k].view(shape)
            o += k
