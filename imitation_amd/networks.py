"""Device-resident building blocks of the reward / discriminator networks.

Host-side mirror of `util/networks.py` (`RunningNorm` `:98-134`, `build_mlp` `:204-283`,
`training`/`evaluating` `:12-34`) whose arithmetic runs in libimitation_hip.so. Objects are
light state holders: parameters live in one flat fp32 HBM buffer per network (torch
`parameters()` order) so that one Adam launch / one all-reduce bucket covers a whole net.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import os
from typing import Dict, Iterator, List, Optional, Sequence, Tuple

import numpy as np
import torch as th
from torch import nn

from imitation_amd import _lib as L


def require_device(device: th.device) -> None:
    if device.type != "cuda":
        raise RuntimeError(
            "imitation_amd computes on MI355X only (no CPU fallback): move the network / algorithm to "
            f"'cuda' first (got device {device}).")


@contextlib.contextmanager
def training_mode(m, mode: bool):
    old = m.training
    m.train(mode)
    try:
        yield m
    finally:
        m.train(old)


def training(m):
    return training_mode(m, True)


def evaluating(m):
    return training_mode(m, False)


class RunningNorm:
    """`util/networks.py:47-134`: running mean / variance (Chan merge), int32 count.

    In train mode `normalize` first updates the statistics with the batch, then normalises it
    with the updated statistics (`networks.py:79-91`)."""

    def __init__(self, num_features: int, eps: float = 1e-5):
        self.num_features, self.eps = int(num_features), float(eps)
        self.running_mean = th.zeros(num_features)
        self.running_var = th.ones(num_features)
        self.count = th.zeros((), dtype=th.int32)
        self.training = True
        self._ws: Optional[th.Tensor] = None
        self.dp = None  # imitation_amd.distributed.DataParallel: merge moments across ranks

    is_chan = True   # Chan-merged statistics: what the fused updates' in-kernel merges implement (EMANorm: False)

    def __getstate__(self):
        state = dict(self.__dict__)   # (never pickle the process-group handle, see modules.RunningNorm.__getstate__)
        state["dp"] = None
        return state

    # -- nn.Module-like plumbing
    def train(self, mode: bool = True):
        self.training = mode
        return self

    def eval(self):
        return self.train(False)

    def to(self, device):
        device = th.device(device)
        self.running_mean = self.running_mean.to(device)
        self.running_var = self.running_var.to(device)
        self.count = self.count.to(device)
        self._ws = None
        return self

    @property
    def device(self) -> th.device:
        return self.running_mean.device

    def state_dict(self, prefix: str = "") -> Dict[str, th.Tensor]:
        return {prefix + "running_mean": self.running_mean, prefix + "running_var": self.running_var,
                prefix + "count": self.count}

    def load_state_dict(self, sd, prefix: str = "") -> None:
        self.running_mean.copy_(th.as_tensor(sd[prefix + "running_mean"]))
        self.running_var.copy_(th.as_tensor(sd[prefix + "running_var"]))
        self.count.copy_(th.as_tensor(sd[prefix + "count"]).to(th.int32))

    def reset_running_stats(self) -> None:
        self.running_mean.zero_()
        self.running_var.fill_(1)
        self.count.zero_()

    # -- compute
    def update_stats(self, x: th.Tensor, ldx: Optional[int] = None, rows: Optional[int] = None) -> None:
        """`RunningNorm.update_stats` on a device batch `x[rows, ldx]` (first `num_features` columns)."""
        require_device(self.device)
        if x.dim() == 1:
            x = x.reshape(-1, 1)
        R = rows if rows is not None else x.shape[0]
        ld = ldx if ldx is not None else x.shape[1]
        need = int(L.load().ia_running_norm_ws_floats(R, self.num_features))
        if self._ws is None or self._ws.numel() != need:
            self._ws = th.empty(need, device=self.device)
        if self.dp is not None and self.dp.world > 1:
            # every rank contributes R rows: all-gather the slab moments, identical merge everywhere
            L.call("ia_running_norm_partial", L.ptr(x), ld, R, self.num_features, L.ptr(self._ws), L.stream())
            ws_all = self.dp.all_gather_flat(self._ws)
            L.call("ia_running_norm_merge", L.ptr(ws_all), self.dp.world, R, self.num_features, self.num_features,
                   L.ptr(self.running_mean), L.ptr(self.running_var), L.ptr(self.count), L.stream())
            return
        L.call("ia_running_norm_update", L.ptr(x), ld, R, self.num_features, L.ptr(self.running_mean),
               L.ptr(self.running_var), L.ptr(self.count), L.ptr(self._ws), L.stream())

    def apply(self, x: th.Tensor, out: th.Tensor, ldx: int, ldy: int, rows: int) -> None:
        L.call("ia_running_norm_apply", L.ptr(x), ldx, rows, self.num_features, L.ptr(self.running_mean),
               L.ptr(self.running_var), self.eps, L.ptr(out), ldy, L.stream())

    def __call__(self, x: th.Tensor) -> th.Tensor:
        """Module-style forward on a device tensor `[B, F]` (or `[B]` when F == 1)."""
        require_device(self.device)
        flat = x.reshape(x.shape[0], -1).contiguous().float()
        if self.training:
            self.update_stats(flat)
        out = th.empty_like(flat)
        self.apply(flat, out, flat.shape[1], flat.shape[1], flat.shape[0])
        return out.reshape(x.shape)

    forward = __call__


class EMANorm(RunningNorm):
    """`util/networks.py:137-201`: exponentially weighted statistics (extra buffers `inv_learning_rate`,
    `num_batches`). Same surface as `RunningNorm`; the fused one-call updates (which merge Chan statistics inside
    their kernels) step aside for it and the stack-by-stack path runs (`RunningNorm.is_chan`)."""

    is_chan = False

    def __init__(self, num_features: int, decay: float = 0.99, eps: float = 1e-5):
        super().__init__(num_features, eps=eps)
        if not 0 < decay < 1:
            raise ValueError("decay must be between 0 and 1")
        self.decay = float(decay)
        self.inv_learning_rate = th.zeros(())
        self.num_batches = th.zeros((), dtype=th.int32)

    def to(self, device):
        super().to(device)
        self.inv_learning_rate = self.inv_learning_rate.to(self.device)
        self.num_batches = self.num_batches.to(self.device)
        return self

    def state_dict(self, prefix: str = "") -> Dict[str, th.Tensor]:
        sd = super().state_dict(prefix)
        sd.update({prefix + "inv_learning_rate": self.inv_learning_rate, prefix + "num_batches": self.num_batches})
        return sd

    def load_state_dict(self, sd, prefix: str = "") -> None:
        super().load_state_dict(sd, prefix)
        self.inv_learning_rate.copy_(th.as_tensor(sd[prefix + "inv_learning_rate"]))
        self.num_batches.copy_(th.as_tensor(sd[prefix + "num_batches"]).to(th.int32))

    def reset_running_stats(self) -> None:
        super().reset_running_stats()
        self.inv_learning_rate.zero_()
        self.num_batches.zero_()

    def update_stats(self, x: th.Tensor, ldx: Optional[int] = None, rows: Optional[int] = None) -> None:
        require_device(self.device)
        if x.dim() == 1:
            x = x.reshape(-1, 1)
        R = rows if rows is not None else x.shape[0]
        ld = ldx if ldx is not None else x.shape[1]
        need = int(L.load().ia_running_norm_ws_floats(R, self.num_features))
        if self._ws is None or self._ws.numel() != need:
            self._ws = th.empty(need, device=self.device)
        L.call("ia_running_norm_partial", L.ptr(x), ld, R, self.num_features, L.ptr(self._ws), L.stream())
        ws, groups = self._ws, 1
        if self.dp is not None and self.dp.world > 1:
            ws, groups = self.dp.all_gather_flat(self._ws), self.dp.world
        L.call("ia_ema_norm_merge", L.ptr(ws), groups, R, self.num_features, self.num_features, L.ptr(self.running_mean),
               L.ptr(self.running_var), L.ptr(self.count), L.ptr(self.inv_learning_rate), L.ptr(self.num_batches),
               self.decay, L.stream())


_ACT_CODES = {nn.ReLU: L.ACT_RELU, nn.Tanh: L.ACT_TANH, None: L.ACT_NONE}


def _round_up(v: int, m: int) -> int:
    return (v + m - 1) // m * m


class DenseStack:
    """`build_mlp` (`util/networks.py:204-283`) on device: optional input `RunningNorm`, hidden
    `Linear`+activation layers, final `Linear`. Initial weights are drawn by constructing
    `torch.nn.Linear` layers on the host in the reference's order, so the torch global RNG is
    consumed identically and seeds carry over."""

    def __init__(self, in_size: int, hid_sizes: Sequence[int], out_size: int = 1, activation=nn.ReLU,
                 dropout_prob: float = 0.0, squeeze_output: bool = False, flatten_input: bool = False,
                 normalize_input_layer=None, name: Optional[str] = None):
        if dropout_prob > 0.0:
            raise NotImplementedError("dropout is off in every reference GAIL/AIRL config; not built for HIP")
        if activation not in _ACT_CODES:
            raise NotImplementedError(f"activation {activation} not supported on the HIP path (ReLU/Tanh)")
        if squeeze_output and out_size != 1:
            raise ValueError("squeeze_output is only applicable when out_size=1")
        self.dims = [int(in_size), *[int(h) for h in hid_sizes], int(out_size)]
        if len(self.dims) - 1 > L.IA_MAX_LAYERS:
            raise NotImplementedError(f"at most {L.IA_MAX_LAYERS} Linear layers")
        self.squeeze_output = squeeze_output
        self.prefix = "" if name is None else f"{name}_"
        self.norm: Optional[RunningNorm] = None
        if normalize_input_layer is not None:
            if not (isinstance(normalize_input_layer, type) and issubclass(normalize_input_layer, RunningNorm)):
                raise NotImplementedError("input normalisation: imitation_amd.RunningNorm or imitation_amd.EMANorm")
            self.norm = normalize_input_layer(in_size)
        self.desc = L.mlp_desc(self.dims, _ACT_CODES[activation])
        layers = [nn.Linear(self.dims[i], self.dims[i + 1]) for i in range(len(self.dims) - 1)]
        self.n_params = sum(l.weight.numel() + l.bias.numel() for l in layers)
        self._init_flat = th.cat([t.detach().reshape(-1) for l in layers for t in (l.weight, l.bias)])
        self.flat: Optional[th.Tensor] = None        # view into the owner's flat parameter buffer
        self.grad: Optional[th.Tensor] = None        # view into the owner's flat gradient buffer
        self.ldx = _round_up(self.dims[0], 4)         # 16-byte aligned rows for the float4 tile loads
        self.hidden_per_row = sum(self.dims[1:-1])
        self.training = True
        self._ws: Dict[Tuple[int, str], Dict[str, th.Tensor]] = {}

    # -- parameter plumbing
    def bind(self, flat: th.Tensor, grad: Optional[th.Tensor]) -> None:
        self.flat, self.grad = flat, grad
        self._ws = {}

    def layer_names(self) -> List[str]:
        n = len(self.dims) - 1
        return [f"{self.prefix}dense{i}" for i in range(n - 1)] + [f"{self.prefix}dense_final"]

    def named_parameters(self, prefix: str = "") -> Iterator[Tuple[str, th.Tensor]]:
        o = 0
        for name, (i, j) in zip(self.layer_names(), zip(self.dims[:-1], self.dims[1:])):
            yield f"{prefix}{name}.weight", self.flat[o:o + i * j].view(j, i)
            o += i * j
            yield f"{prefix}{name}.bias", self.flat[o:o + j]
            o += j

    def named_grads(self, prefix: str = "") -> Iterator[Tuple[str, th.Tensor]]:
        o = 0
        for name, (i, j) in zip(self.layer_names(), zip(self.dims[:-1], self.dims[1:])):
            yield f"{prefix}{name}.weight", self.grad[o:o + i * j].view(j, i)
            o += i * j
            yield f"{prefix}{name}.bias", self.grad[o:o + j]
            o += j

    def state_dict(self, prefix: str = "") -> Dict[str, th.Tensor]:
        sd: Dict[str, th.Tensor] = {}
        if self.norm is not None:
            sd.update(self.norm.state_dict(f"{prefix}{self.prefix}normalize_input."))
        sd.update(dict(self.named_parameters(prefix)))
        return sd

    def load_state_dict(self, sd, prefix: str = "") -> None:
        if self.norm is not None:
            self.norm.load_state_dict(sd, f"{prefix}{self.prefix}normalize_input.")
        for k, v in self.named_parameters(prefix):
            v.copy_(th.as_tensor(sd[k]))

    def train(self, mode: bool = True):
        self.training = mode
        if self.norm is not None:
            self.norm.train(mode)
        return self

    def to(self, device):
        if self.norm is not None:
            self.norm.to(device)
        self._ws = {}
        return self

    # -- compute
    def workspace(self, R: int, tag: str) -> Dict[str, th.Tensor]:
        key = (R, tag)
        ws = self._ws.get(key)
        if ws is None:
            dev = self.flat.device
            ws = {
                "X": th.zeros(R, self.ldx, device=dev),
                "Xn": th.zeros(R, self.ldx, device=dev) if self.norm is not None else None,
                "hidden": th.empty(max(1, R * self.hidden_per_row), device=dev),
                "out": th.empty(R, self.dims[-1], device=dev),
            }
            self._ws[key] = ws
        return ws

    def train_workspace(self, R: int, tag: str) -> Dict[str, th.Tensor]:
        ws = self.workspace(R, tag)
        if "dhidden" not in ws:
            dev = self.flat.device
            ws["dhidden"] = th.empty(max(1, R * self.hidden_per_row), device=dev)
            ws["splits"] = max(1, min(64, R // 256))
            ws["partials"] = th.empty(ws["splits"], self.n_params, device=dev)
        return ws

    # prediction of whole tiles (the reward relabelling behind a rollout) on the fused tile kernel where its shape is covered
    # (`ia_disc_fused_predict`: D <= 24 -> H -> H -> 1, ReLU, H = 128 / 256); False: always the layer-by-layer launches
    FUSED_PREDICT = os.environ.get("IA_FUSED_PREDICT", "1") != "0"

    def _predict_ws(self) -> Optional[th.Tensor]:
        ws = getattr(self, "_pred_ws", False)
        if ws is False:
            n = 0
            if len(self.dims) == 4 and self.dims[-1] == 1 and (self.norm is None or self.norm.is_chan):
                n = int(L.load().ia_disc_fused_predict_ws_floats(C.byref(self.desc), self.ldx))
            ws = self._pred_ws = th.zeros(n, device=self.flat.device) if n > 0 else None
        return ws

    def forward_rows(self, ws: Dict[str, th.Tensor], R: int, out_act: int = L.ACT_NONE, keep_hidden: bool = True) -> th.Tensor:
        """Runs the stack on `ws["X"][:R]` (already assembled). Returns `ws["out"]` `[R, out]`. `keep_hidden=False`: the
        caller will not back-propagate through this pass (predictions): covered shapes take the two-launch tile kernel,
        which keeps the hidden activations in LDS."""
        if not keep_hidden and self.FUSED_PREDICT and not (self.norm is not None and self.training):
            pws = self._predict_ws()
            if pws is not None:
                nrm = self.norm
                L.call("ia_disc_fused_predict", C.byref(self.desc), L.ptr(self.flat), L.ptr(ws["X"]), self.ldx, R,
                       L.ptr(nrm.running_mean) if nrm is not None else None, L.ptr(nrm.running_var) if nrm is not None else None,
                       float(nrm.eps) if nrm is not None else 0.0, out_act, L.ptr(pws), L.ptr(ws["out"]), L.stream())
                return ws["out"]
        x = ws["X"]
        if self.norm is not None:
            if self.training:
                self.norm.update_stats(x, ldx=self.ldx, rows=R)
            self.norm.apply(x, ws["Xn"], self.ldx, self.ldx, R)
            x = ws["Xn"]
        ws["_in"] = x
        L.call("ia_mlp_forward", C.byref(self.desc), L.ptr(self.flat), L.ptr(x), self.ldx, R, L.ptr(ws["hidden"]),
               L.ptr(ws["out"]), out_act, L.stream())
        return ws["out"]

    def backward_rows(self, ws: Dict[str, th.Tensor], R: int, d_out: th.Tensor, accumulate: bool,
                      scale: float = 1.0, adam: Optional["HipAdam"] = None) -> None:
        """Back-propagates `d_out[R, out]` through the activations saved by `forward_rows` and
        (accumulates) the parameter gradient into `self.grad`. With `adam` (only when this stack IS
        the whole optimised buffer and nothing is accumulated) the split-K reduction and the Adam step
        run as one launch."""
        L.call("ia_mlp_backward", C.byref(self.desc), L.ptr(self.flat), L.ptr(ws["_in"]), self.ldx, R,
               L.ptr(ws["hidden"]), L.ptr(d_out), L.ptr(ws["dhidden"]), L.ptr(ws["partials"]), ws["splits"], None,
               L.stream())
        if adam is not None:
            assert not accumulate and adam.flat.data_ptr() == self.flat.data_ptr() and adam.flat.numel() == self.n_params
            adam.fused_reduce_step(ws["partials"], ws["splits"], scale)
            return
        L.call("ia_reduce_partials", L.ptr(ws["partials"]), ws["splits"], self.n_params, scale, int(accumulate),
               L.ptr(self.grad), L.stream())


class TransitionTable:
    """A set of row-aligned device arrays describing transitions (the expert demonstrations,
    the generator replay ring, or a rollout): `obs/next_obs` fp32 `[N, obs_dim]`, `acts` fp32
    `[N, act_dim]` (Box) or int64 `[N]` (Discrete), `dones` uint8 `[N]`."""

    def __init__(self, obs: th.Tensor, acts: th.Tensor, next_obs: th.Tensor, dones: th.Tensor, discrete: bool):
        self.obs, self.acts, self.next_obs, self.dones, self.discrete = obs, acts, next_obs, dones, discrete

    def __len__(self) -> int:
        return self.obs.shape[0]


def gather_concat(table: TransitionTable, idx: Optional[th.Tensor], n: int, obs_dim: int, act_dim: int,
                  flags: Tuple[bool, bool, bool, bool], X: th.Tensor, ldx: int, row0: int,
                  state_from_next: bool = False) -> None:
    """`adversarial/common.py:592-603` + `rewards/reward_nets.py:441-457`: X[row0:row0+n] =
    [state | action (one-hot if discrete) | next_state | done] of rows `idx` of `table`."""
    obs = table.next_obs if state_from_next else table.obs
    L.call("ia_gather_concat", L.ptr(obs), None if table.discrete else L.ptr(table.acts),
           L.ptr(table.acts) if table.discrete else None, L.ptr(table.next_obs), L.ptr(table.dones), L.ptr(idx), n,
           obs_dim, act_dim, int(flags[0]), int(flags[1]), int(flags[2]), int(flags[3]), L.ptr(X), ldx, row0,
           L.stream())


class HipAdam:
    """`torch.optim.Adam` (single param group) over one flat device buffer, stepping with the
    fused HIP kernel. Mirrors the constructor kwargs and `state_dict` fields that matter."""

    def __init__(self, flat: th.Tensor, grad: th.Tensor, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 0.0, amsgrad: bool = False, **unused):
        if amsgrad:
            raise NotImplementedError("amsgrad is not implemented on the HIP path")
        self.flat, self.grad = flat, grad
        self.param_groups = [dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay)]
        self.exp_avg = th.zeros_like(flat)
        self.exp_avg_sq = th.zeros_like(flat)
        self.step_count = 0

    def zero_grad(self, set_to_none: bool = False) -> None:
        self.grad.zero_()

    # ---- device-resident step count: for update loops that are captured once and replayed as a hipGraph ----------
    def begin_device_steps(self) -> None:
        """From here on `step()` takes its bias-correction scalars from a device-side step count (two launches per
        step, no host-dependent kernel arguments), so a captured sequence of steps can be replayed; the host count is
        advanced by the caller (`end_device_steps`)."""
        if getattr(self, "_dev_step", None) is None:
            self._dev_step = th.zeros(1, dtype=th.int64, device=self.flat.device)
            self._dev_scal = th.zeros(2, device=self.flat.device)
            self._dev_lr = th.zeros(1, dtype=th.float64, device=self.flat.device)
        self._dev_lr.fill_(float(self.param_groups[0]["lr"]))
        self._dev_mode = True

    def end_device_steps(self, steps_run: int = 0) -> None:
        self._dev_mode = False
        self.step_count += int(steps_run)

    def sync_device_step(self) -> None:
        """Sets the device-side count and learning rate to the host's (before a replay)."""
        self._dev_step.fill_(self.step_count)
        self._dev_lr.fill_(float(self.param_groups[0]["lr"]))

    def step(self) -> None:
        g = self.param_groups[0]
        if getattr(self, "_dev_mode", False):
            b1, b2 = g["betas"]
            s = L.stream()
            L.call("ia_adam_step_scalars", L.ptr(self._dev_step), float(g["lr"]), L.ptr(self._dev_lr), float(b1),
                   float(b2), L.ptr(self._dev_scal), s)
            L.call("ia_adam_step_dev", L.ptr(self.flat), L.ptr(self.grad), L.ptr(self.exp_avg), L.ptr(self.exp_avg_sq),
                   self.flat.numel(), b1, b2, g["eps"], g["weight_decay"], L.ptr(self._dev_scal), s)
            return
        self.step_count += 1
        b1, b2 = g["betas"]
        bc1 = 1.0 - b1 ** self.step_count
        bc2 = 1.0 - b2 ** self.step_count
        L.call("ia_adam_step", L.ptr(self.flat), L.ptr(self.grad), L.ptr(self.exp_avg), L.ptr(self.exp_avg_sq),
               self.flat.numel(), b1, b2, g["eps"], g["weight_decay"], g["lr"] / bc1, bc2 ** 0.5, L.stream())

    def fused_reduce_step(self, partials: th.Tensor, splits: int, scale: float) -> None:
        """grad = scale * sum_s partials[s]; then the Adam step -- one launch."""
        g = self.param_groups[0]
        self.step_count += 1
        b1, b2 = g["betas"]
        bc1 = 1.0 - b1 ** self.step_count
        bc2 = 1.0 - b2 ** self.step_count
        L.call("ia_reduce_partials_adam", L.ptr(partials), splits, self.flat.numel(), scale, L.ptr(self.grad),
               L.ptr(self.flat), L.ptr(self.exp_avg), L.ptr(self.exp_avg_sq), b1, b2, g["eps"], g["weight_decay"],
               g["lr"] / bc1, bc2 ** 0.5, L.stream())

    def next_step_args(self):
        """Counts one optimiser step and returns its `ia_adam_args` (by reference) for an entry point that finishes
        with the reduction + Adam launch itself (`ia_airl_step_shaped`)."""
        g = self.param_groups[0]
        self.step_count += 1
        b1, b2 = g["betas"]
        a = getattr(self, "_c_args", None)
        if a is None:
            a = self._c_args = L.AdamArgs()
        a.grads, a.exp_avg, a.exp_avg_sq = self.grad.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr()
        a.beta1, a.beta2, a.eps, a.weight_decay = b1, b2, g["eps"], g["weight_decay"]
        a.step_size = g["lr"] / (1.0 - b1 ** self.step_count)
        a.bc2_sqrt = (1.0 - b2 ** self.step_count) ** 0.5
        return C.byref(a)

    def state_dict(self):
        return {"state": {0: {"step": self.step_count, "exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq}},
                "param_groups": [dict(g, params=[0]) for g in self.param_groups]}

    def load_state_dict(self, sd) -> None:
        st = sd["state"][0]
        self.step_count = int(st["step"])
        self.exp_avg.copy_(st["exp_avg"])
        self.exp_avg_sq.copy_(st["exp_avg_sq"])
        for g, s in zip(self.param_groups, sd["param_groups"]):
            g.update({k: v for k, v in s.items() if k != "params"})
