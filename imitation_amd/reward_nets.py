"""Reward / discriminator networks behind the reference's `RewardNet` plugin API
(`rewards/reward_nets.py`), computing on MI355X through libimitation_hip.so.

Same constructors, same `forward / preprocess / predict_th / predict / predict_processed`
surface, same `state_dict` keys (so checkpoints interchange with the reference), but the
objects are plain state holders (not autograd modules): the discriminator update is driven
by `AdversarialTrainer.train_disc` through `disc_forward` / `disc_backward`, which run the
fused HIP forward+backward and accumulate into one flat gradient buffer.
"""
from __future__ import annotations

import abc
from typing import Callable, Dict, Iterator, List, Optional, Sequence, Tuple, Type

import numpy as np
import torch as th
from torch import nn

from imitation_amd import _lib as L
from imitation_amd import spaces
from imitation_amd.networks import (DenseStack, RunningNorm, TransitionTable, evaluating, gather_concat,
                                    require_device)


# tuning / tests: False forces the general (unfused) discriminator update even where the fused one applies
FUSED_DISC_STEP = True
FUSED_AIRL_STEP = True    # shaped reward nets of the default geometry take `fused_prepare` / `fused_finish` (tests flip it)


class ParamStore:
    """All dense stacks of one reward-net tree share a flat parameter / gradient buffer."""

    def __init__(self):
        self.stacks: List[DenseStack] = []
        self.flat: Optional[th.Tensor] = None
        self.grad: Optional[th.Tensor] = None

    def add(self, stack: DenseStack) -> None:
        self.stacks.append(stack)
        self.materialize(self.flat.device if self.flat is not None else th.device("cpu"))

    def materialize(self, device) -> None:
        device = th.device(device)
        parts = []
        for s in self.stacks:
            parts.append(s.flat.detach().to(device) if s.flat is not None else s._init_flat.to(device))
        self.flat = th.cat(parts).contiguous()
        self.grad = th.zeros_like(self.flat)
        o = 0
        for s in self.stacks:
            s.to(device)
            s.bind(self.flat[o:o + s.n_params], self.grad[o:o + s.n_params])
            o += s.n_params


def preprocess_space(x: th.Tensor, space, normalize_images: bool = True) -> th.Tensor:
    """[SB3 preprocess_obs] for the spaces on the path."""
    if isinstance(space, spaces.Box):
        x = x.float()
        if normalize_images and space.dtype == np.uint8 and len(space.shape) == 3:
            x = x / 255.0
        return x
    if isinstance(space, spaces.Discrete):
        return th.nn.functional.one_hot(x.long(), num_classes=space.n).float()
    raise NotImplementedError(f"Preprocessing not implemented for {space}")


class RewardNet(abc.ABC):
    """`rewards/reward_nets.py:16-224`."""

    def __init__(self, observation_space, action_space, normalize_images: bool = True):
        self.observation_space, self.action_space = observation_space, action_space
        self.normalize_images = normalize_images
        self.training = True
        self._store: Optional[ParamStore] = None

    # ---- module-like plumbing -----------------------------------------------------------
    def _children(self) -> List["RewardNet"]:
        return []

    def _named_stacks(self) -> List[Tuple[str, DenseStack]]:
        return []

    def _named_norms(self) -> List[Tuple[str, RunningNorm]]:
        return []

    def train(self, mode: bool = True):
        self.training = mode
        for _, s in self._named_stacks():
            s.train(mode)
        for _, n in self._named_norms():
            n.train(mode)
        for c in self._children():
            c.train(mode)
        return self

    def eval(self):
        return self.train(False)

    def to(self, device):
        if self._store is not None:
            self._store.materialize(device)
        for _, n in self._named_norms():
            n.to(device)
        for c in self._children():
            for _, n in c._named_norms():
                n.to(device)
        return self

    @property
    def device(self) -> th.device:
        if self._store is not None and self._store.flat is not None:
            return self._store.flat.device
        return th.device("cpu")

    @property
    def dtype(self) -> th.dtype:
        return th.float32

    def named_parameters(self) -> Iterator[Tuple[str, th.Tensor]]:
        for prefix, s in self._named_stacks():
            yield from s.named_parameters(prefix)

    def parameters(self) -> Iterator[th.Tensor]:
        for _, p in self.named_parameters():
            yield p

    def state_dict(self) -> Dict[str, th.Tensor]:
        sd: Dict[str, th.Tensor] = {}
        for prefix, s in self._named_stacks():
            sd.update(s.state_dict(prefix))
        for prefix, n in self._named_norms():
            sd.update(n.state_dict(prefix))
        return sd

    def load_state_dict(self, sd) -> None:
        for prefix, s in self._named_stacks():
            s.load_state_dict(sd, prefix)
        for prefix, n in self._named_norms():
            n.load_state_dict(sd, prefix)

    # ---- reference API ------------------------------------------------------------------
    def preprocess(self, state, action, next_state, done):
        """`reward_nets.py:52-118`: to device, float / one-hot, done -> float32."""
        dev = self.device
        to = lambda a: th.as_tensor(np.ascontiguousarray(a) if isinstance(a, np.ndarray) else a).to(dev)
        s = preprocess_space(to(state), self.observation_space, self.normalize_images)
        a = preprocess_space(to(action), self.action_space, self.normalize_images)
        ns = preprocess_space(to(next_state), self.observation_space, self.normalize_images)
        d = to(done).to(th.float32)
        assert s.shape == ns.shape and len(a) == len(s)
        return s, a, ns, d

    def _table_from_tensors(self, state, action, next_state, done) -> TransitionTable:
        n = state.shape[0]
        return TransitionTable(state.reshape(n, -1).float().contiguous(), action.reshape(n, -1).float().contiguous(),
                               next_state.reshape(n, -1).float().contiguous(),
                               (done.reshape(n) != 0).to(th.uint8).contiguous(), discrete=False)

    def forward(self, state: th.Tensor, action: th.Tensor, next_state: th.Tensor, done: th.Tensor) -> th.Tensor:
        """Rewards for preprocessed device tensors (`reward_nets.py:42-50`); no autograd graph."""
        require_device(self.device)
        table = self._table_from_tensors(state, action, next_state, done)
        return self._forward_table([(table, None, len(table))], "fwd").clone()

    __call__ = forward

    @abc.abstractmethod
    def _forward_table(self, sources, tag: str, out_act: int = L.ACT_NONE) -> th.Tensor:
        """Rewards `[n]` for rows assembled from `sources = [(table, idx_or_None, n), ...]`."""

    def predict_th(self, state, action, next_state, done) -> th.Tensor:
        with evaluating(self):
            rew = self.forward(*self.preprocess(state, action, next_state, done))
        assert rew.shape == np.shape(state)[:1]
        return rew

    def predict(self, state, action, next_state, done) -> np.ndarray:
        return self.predict_th(state, action, next_state, done).detach().cpu().numpy().flatten()

    def predict_processed(self, state, action, next_state, done, **kwargs) -> np.ndarray:
        del kwargs
        return self.predict(state, action, next_state, done)

    # ---- fused paths used by the trainer / rollout collector ----------------------------
    def predict_processed_rollout(self, table: TransitionTable, T: int, n: int) -> th.Tensor:
        """`predict_processed` for a whole rollout `[T*n]` rows (time-major), result on device.
        Equivalent to calling `predict_processed` once per env step (`reward_wrapper.py:110-115`)
        because nothing on this path changes between steps except output-norm statistics, which
        `NormalizedRewardNet` replays sequentially."""
        with evaluating(self):
            return self._forward_table([(table, None, T * n)], "rollout")

    def forward_plan(self):
        """`(BasicRewardNet, output activation)` when `_forward_table` of this net is, row by row, ONE stack the fused
        prediction kernels cover followed by that activation; None otherwise."""
        return None

    def rollout_tail_plan(self):
        """`forward_plan()` when `predict_processed_rollout` of this net IS its `_forward_table` (the default; a wrapper that
        post-processes the tile -- `NormalizedRewardNet` as the outermost net -- is not): the rollout collector then relabels,
        copies the rewards out and runs GAE in one host call (`ia_rollout_tail`, `PPO._rollout_tail_args`); None: the
        general path."""
        if type(self).predict_processed_rollout is not RewardNet.predict_processed_rollout:
            return None
        return self.forward_plan()

    def disc_forward(self, sources, mb_rows: int, logp: Optional[th.Tensor]) -> th.Tensor:
        raise NotImplementedError(f"{type(self).__name__} cannot be trained as a discriminator on the HIP path")

    def disc_backward(self, d_logits: th.Tensor, accumulate: bool, adam=None) -> bool:
        """Returns True when the optimiser step was fused into the backward (see `DenseStack`)."""
        raise NotImplementedError


def _dims_of(space) -> int:
    return spaces.flatdim(space)


def _module_twin(cls_name: str, args, kwargs):
    """`dropout_prob > 0` (`util/networks.py:270-271`; off in every shipped GAIL / AIRL configuration): the fused
    state-holder stacks have no dropout, the `nn.Module` reward nets of `imitation_amd.modules` do (layer by layer on the
    HIP ops, masks from the device generator) -- so the constructor hands back the module net of the same name and
    arguments, which `AdversarialTrainer` trains through `loss.backward()`. Normalisation layer classes are mapped to
    their module counterparts."""
    from imitation_amd import modules, networks
    kw = dict(kwargs)
    nl = kw.get("normalize_input_layer")
    if nl is networks.EMANorm:
        kw["normalize_input_layer"] = modules.EMANorm
    elif nl is networks.RunningNorm:
        kw["normalize_input_layer"] = modules.RunningNorm
    return getattr(modules, cls_name)(*args, **kw)


class BasicRewardNet(RewardNet):
    """`rewards/reward_nets.py:383-457`: MLP over the concatenation of the enabled inputs."""

    def __new__(cls, *args, **kwargs):
        if cls is BasicRewardNet and float(kwargs.get("dropout_prob", 0.0) or 0.0) > 0.0:
            return _module_twin("BasicRewardNet", args, kwargs)   # (not an instance of cls: __init__ is not run on it)
        return super().__new__(cls)

    def __init__(self, observation_space, action_space, use_state: bool = True, use_action: bool = True,
                 use_next_state: bool = False, use_done: bool = False, **kwargs):
        super().__init__(observation_space, action_space)
        self.use_state, self.use_action = use_state, use_action
        self.use_next_state, self.use_done = use_next_state, use_done
        self.obs_dim, self.act_dim = _dims_of(observation_space), _dims_of(action_space)
        size = (self.obs_dim if use_state else 0) + (self.act_dim if use_action else 0) + \
               (self.obs_dim if use_next_state else 0) + (1 if use_done else 0)
        full = {"hid_sizes": (32, 32), **kwargs, "in_size": size, "out_size": 1, "squeeze_output": True}
        self.mlp = DenseStack(**full)
        self._store = ParamStore()
        self._store.add(self.mlp)

    @property
    def flags(self):
        return (self.use_state, self.use_action, self.use_next_state, self.use_done)

    def _named_stacks(self):
        return [("mlp.", self.mlp)]

    def _assemble(self, sources, ws) -> int:
        row = 0
        for table, idx, n in sources:
            # preprocessed tensors arrive one-hot already (discrete=False); raw tables may be int64
            gather_concat(table, idx, n, self.obs_dim, self.act_dim, self.flags, ws["X"], self.mlp.ldx, row)
            row += n
        return row

    def _forward_table(self, sources, tag, out_act=L.ACT_NONE):
        R = sum(n for _, _, n in sources)
        ws = self.mlp.workspace(R, tag)
        self._assemble(sources, ws)
        return self.mlp.forward_rows(ws, R, out_act, keep_hidden=False).reshape(R)

    def forward_plan(self):
        if type(self) is not BasicRewardNet or not self.mlp.FUSED_PREDICT or self.mlp._predict_ws() is None:
            return None   # (a subclass may have its own forward; shapes outside the prediction kernels)
        return self, L.ACT_NONE

    def disc_forward(self, sources, mb_rows, logp):
        R = sum(n for _, _, n in sources)
        ws = self.mlp.train_workspace(R, "disc")
        self._assemble(sources, ws)
        self._disc_ws, self._disc_R = ws, R
        return self.mlp.forward_rows(ws, R).reshape(R)

    def disc_backward(self, d_logits, accumulate, adam=None):
        fuse = adam is not None and not accumulate and self._store.flat.numel() == self.mlp.n_params
        self.mlp.backward_rows(self._disc_ws, self._disc_R, d_logits, accumulate, adam=adam if fuse else None)
        return fuse

    def assemble_round(self, e_tab: TransitionTable, g_tab: TransitionTable, idx_all: th.Tensor, n_updates: int,
                       mb: int, dp=None) -> Optional[Dict[str, th.Tensor]]:
        """Batch assembly of a whole round's discriminator updates in ONE launch (fused shapes only; returns None
        otherwise): update k gathers expert rows `idx_all[k, 0]` and generator rows `idx_all[k, 1]` into its own
        X and leaves its RunningNorm slab moments; one `ia_running_norm_merge_seq` launch then applies the n
        updates of the input norm in order and keeps the statistics AFTER each one -- what update k's own forward
        normalises with (`util/networks.py:79-91`). The updates themselves (`disc_step_c(..., pre=(ws, k))`) are
        four launches each. `dp` (`distributed.DataParallel`, world > 1): every rank assembles its own batches; the
        slab moments of the whole round are exchanged in ONE all-gather and every rank applies the same merges over
        world x 2*mb rows per update -- the statistics of a single process on the concatenated batches."""
        import ctypes as C
        mlp, R = self.mlp, 2 * mb
        if FUSED_DISC_STEP is False or int(L.load().ia_disc_fused_ws_floats(C.byref(mlp.desc), R, mlp.ldx)) <= 0:
            return None
        if mlp.norm is not None and not mlp.norm.is_chan:   # EMANorm: the fused kernels merge Chan statistics
            return None
        if idx_all.shape[1:] != (2, mb) or not idx_all.is_contiguous():
            return None
        dev, D = mlp.flat.device, mlp.dims[0]
        key = ("round", n_updates, R)
        rw = mlp._ws.get(key)
        if rw is None:
            need = int(L.load().ia_running_norm_ws_floats(R, D))
            rw = {"X_all": th.zeros(n_updates, R, mlp.ldx, device=dev), "rn_all": th.empty(n_updates, need, device=dev),
                  "snap": th.empty(n_updates, 2, D, device=dev), "need": need}
            mlp._ws[key] = rw
        ws = self._step_workspace(R)
        a = ws["args"]
        for k, t in enumerate((e_tab, g_tab)):
            setattr(a, f"obs{k}", L.ptr(t.obs))
            setattr(a, f"act{k}_f32", None if t.discrete else L.ptr(t.acts))
            setattr(a, f"act{k}_i64", L.ptr(t.acts) if t.discrete else None)
            setattr(a, f"next{k}", L.ptr(t.next_obs))
            setattr(a, f"done{k}", L.ptr(t.dones))
            setattr(a, f"idx{k}", idx_all[0, k].data_ptr())
            setattr(a, f"n{k}", mb)
        nrm = mlp.norm
        update = nrm is not None and mlp.training
        a.X, a.rn_ws = L.ptr(rw["X_all"]), (L.ptr(rw["rn_all"]) if update else None)
        L.call("ia_disc_assemble_round", C.byref(a), n_updates, 2 * mb, R * mlp.ldx, rw["need"], L.stream())
        a.X, a.rn_ws = L.ptr(ws["X"]), L.ptr(ws["rn_ws"])
        rw["has_moments"] = update
        rw["rn_seq"], rw["rn_stride"], rw["rn_groups"] = rw["rn_all"], rw["need"], 1
        if update:
            if dp is not None and dp.world > 1:   # [rank][update][moments] -> [update][rank][moments]
                allm = dp.all_gather_flat(rw["rn_all"].reshape(-1))
                rw["rn_seq"] = allm.view(dp.world, n_updates, rw["need"]).permute(1, 0, 2).contiguous()
                rw["rn_stride"], rw["rn_groups"] = dp.world * rw["need"], dp.world
            L.call("ia_running_norm_merge_seq", L.ptr(rw["rn_seq"]), n_updates, rw["rn_stride"], rw["rn_groups"], R, D, D,
                   L.ptr(nrm.running_mean), L.ptr(nrm.running_var), L.ptr(nrm.count), L.ptr(rw["snap"]), L.stream())
        L.call("ia_disc_fused_prepare", C.byref(mlp.desc), L.ptr(mlp.flat), R, mlp.ldx, L.ptr(ws["fused_ws"]), L.stream())
        return rw

    def _step_workspace(self, R: int) -> Dict[str, th.Tensor]:
        import ctypes as C
        mlp = self.mlp
        ws = mlp.train_workspace(R, "disc")
        if "dlogits" not in ws:
            dev = mlp.flat.device
            ws["dlogits"] = th.empty(R, device=dev)
            ws["rn_ws"] = th.empty(max(1, int(L.load().ia_running_norm_ws_floats(R, mlp.dims[0]))), device=dev)
            a = L.DiscStepArgs()
            a.desc = C.pointer(mlp.desc)
            a.obs_dim, a.act_dim = self.obs_dim, self.act_dim
            a.use_state, a.use_action = int(self.use_state), int(self.use_action)
            a.use_next_state, a.use_done = int(self.use_next_state), int(self.use_done)
            a.X, a.Xn, a.ldx = L.ptr(ws["X"]), L.ptr(ws["Xn"]), mlp.ldx
            a.hidden, a.dhidden = L.ptr(ws["hidden"]), L.ptr(ws["dhidden"])
            a.logits, a.dlogits, a.partials = L.ptr(ws["out"]), L.ptr(ws["dlogits"]), L.ptr(ws["partials"])
            a.splits, a.rn_ws = ws["splits"], L.ptr(ws["rn_ws"])
            # D -> H -> H -> 1 stacks: workspace of the five-launch fused update (0 floats: general path)
            nf = 0 if FUSED_DISC_STEP is False else int(L.load().ia_disc_fused_ws_floats(C.byref(mlp.desc), R, mlp.ldx))
            if mlp.norm is not None and not mlp.norm.is_chan:
                nf = 0
            ws["fused_ws"] = th.zeros(nf, device=dev) if nf > 0 else None
            a.fused_ws = L.ptr(ws["fused_ws"])
            if nf > 0:  # 32 K-splits of the second layer's weight gradient: half the partial-slab traffic of 64
                a.splits = min(ws["splits"], 32)
            ws["args"] = a
        return ws

    def fused_gp_ws(self, mb: int) -> Optional[th.Tensor]:
        """Workspace of the gradient penalty fused into the 128 / 256-wide tile update over `mb` interpolated rows
        (None: the shape takes `grad_penalty.penalty_and_param_grad`, stack by stack)."""
        import ctypes as C
        mlp = self.mlp
        ws = self._step_workspace(2 * mb)
        if "gp_ws" not in ws:
            nf = 0
            if ws["fused_ws"] is not None:
                nf = int(L.load().ia_disc_fused_gp_ws_floats(C.byref(mlp.desc), mb, mlp.ldx))
            ws["gp_ws"] = th.zeros(nf, device=mlp.flat.device) if nf > 0 else None
            ws["gp_out"] = th.zeros(1, device=mlp.flat.device)
        return ws["gp_ws"]

    def disc_step_c(self, sources, n_expert: int, loss_scale: float, stats: th.Tensor, bce_ws: th.Tensor,
                    accumulate: bool, adam=None, pnorm: Optional[RunningNorm] = None, pnorm_dim: int = 0, pre=None,
                    gp=None):
        """One discriminator minibatch through the single C entry `ia_disc_step_basic` (assemble ->
        norm -> forward -> BCE -> backward -> reduce [-> Adam]): ONE host call instead of ~25.
        Returns the workspace dict (logits in ws["out"], slab moments in ws["rn_ws"]).
        `gp = (e [n_expert] device weights, coef, target)`: the opt-in gradient penalty inside the same update
        (`fused_gp_ws(n_expert)` must not be None); its mean lands in `ws["gp_out"]`."""
        import ctypes as C
        ws = self._step_workspace(sources[0][2] + sources[1][2])
        a = ws["args"]
        self._fill_step_args(a, ws, sources, n_expert, loss_scale, stats, bce_ws, accumulate, adam, pnorm, pnorm_dim, pre, gp)
        L.call("ia_disc_step_basic", C.byref(a), L.stream())
        return self._step_used(ws, pre)

    def disc_round_c(self, n: int, mb: int, loss_scale: float, stats_rows: th.Tensor, bce_ws: th.Tensor, adam, rw,
                     gp=None):
        """The n pre-assembled updates of one round (`rw` = `assemble_round`'s workspace) through ONE C call
        (`ia_disc_round_basic`): the same launches in the same order as n `disc_step_c(..., pre=(rw, k))` calls with
        the optimiser step fused -- the per-update Python (argument structs, ~110 us each, more than the update's
        kernels take to enqueue) is paid once per round. `gp = ([n] list of device weight vectors, coef, target)`."""
        import ctypes as C
        ws = self._step_workspace(2 * mb)
        arr = ws.get("round_args")
        if arr is None or len(arr) < n:
            arr = ws["round_args"] = (type(ws["args"]) * n)()
            ws["round_args_key"] = None
        base = ws["args"]
        src = [(None, None, mb), (None, None, mb)]
        # Everything but the Adam scalars of the step and the weight vectors' addresses is the same round after round (the
        # round workspace, the statistics rows and the optimiser state live in fixed buffers): the n argument blocks are
        # filled once and re-used while nothing they were built from has moved -- 16 x ~20 us of attribute stores per round
        # that sat between the PPO launch and the round's first kernel (`tools/host_profile.py P_gp10`)
        mlp, nrm, g = self.mlp, self.mlp.norm, adam.param_groups[0]
        gws = None if gp is None else self.fused_gp_ws(mb)
        key = (n, mb, float(loss_scale), stats_rows.data_ptr(), stats_rows.stride(0), bce_ws.data_ptr(),
               rw["X_all"].data_ptr(), rw["X_all"].stride(0), rw["has_moments"], rw["rn_all"].data_ptr() if rw["has_moments"] else 0,
               rw["snap"].data_ptr() if rw["has_moments"] else 0, mlp.flat.data_ptr(), mlp.grad.data_ptr(),
               mlp.training, self.use_state, None if nrm is None else (nrm.running_mean.data_ptr(), nrm.eps),
               id(adam), adam.exp_avg.data_ptr(), adam.exp_avg_sq.data_ptr(), g["lr"], tuple(g["betas"]), g["eps"],
               g["weight_decay"], None if gp is None else (float(gp[1]), float(gp[2]), -1 if gws is None else gws.data_ptr(),
                                                           ws["gp_out"].data_ptr()))
        if ws.get("round_args_key") == key and (gp is None or gws is not None):
            b1, b2 = g["betas"]
            for k in range(n):
                a = arr[k]
                adam.step_count += 1
                a.step_size = g["lr"] / (1.0 - b1 ** adam.step_count)
                a.bc2_sqrt = (1.0 - b2 ** adam.step_count) ** 0.5
                if gp is not None:
                    e = gp[0][k]
                    if e.numel() != mb:
                        raise RuntimeError("the fused gradient penalty needs one weight per expert / generator row pair")
                    a.gp_e = L.ptr(e)
        else:
            for k in range(n):
                a = arr[k]
                C.memmove(C.byref(a), C.byref(base), C.sizeof(base))   # (the fields no update changes: desc, work areas, ...)
                self._fill_step_args(a, ws, src, mb, loss_scale, stats_rows[k], bce_ws, False, adam, None, 0, (rw, k),
                                     None if gp is None else (gp[0][k], gp[1], gp[2]))
            ws["round_args_key"] = key
        L.call("ia_disc_round_basic", arr, n, L.stream())
        return self._step_used(ws, (rw, n - 1))

    def _fill_step_args(self, a, ws, sources, n_expert, loss_scale, stats, bce_ws, accumulate, adam, pnorm, pnorm_dim,
                        pre, gp) -> None:
        (t0, i0, n0), (t1, i1, n1) = sources
        mlp = self.mlp
        a.params, a.grads = L.ptr(mlp.flat), L.ptr(mlp.grad)
        nrm = mlp.norm
        a.norm_mean = L.ptr(nrm.running_mean) if nrm is not None else None
        a.norm_var = L.ptr(nrm.running_var) if nrm is not None else None
        a.norm_count = L.ptr(nrm.count) if nrm is not None else None
        a.norm_eps = nrm.eps if nrm is not None else 0.0
        a.update_norm = int(nrm is not None and mlp.training)
        a.pre_assembled = 0
        a.X, a.rn_ws = L.ptr(ws["X"]), L.ptr(ws["rn_ws"])
        a.n0, a.n1 = n0, n1
        if pre is not None:   # (round workspace of `assemble_round`, update number): rows and statistics are ready
            rw, k = pre
            a.pre_assembled, a.update_norm = 1, 0
            a.X = rw["X_all"][k].data_ptr()
            if rw["has_moments"]:
                a.rn_ws = rw["rn_all"][k].data_ptr()
                a.norm_mean, a.norm_var = rw["snap"][k, 0].data_ptr(), rw["snap"][k, 1].data_ptr()
        for k, (t, i, n) in enumerate(((t0, i0, n0), (t1, i1, n1)) if pre is None else ()):  # (pre-assembled: unused)
            setattr(a, f"obs{k}", L.ptr(t.obs))
            setattr(a, f"act{k}_f32", None if t.discrete else L.ptr(t.acts))
            setattr(a, f"act{k}_i64", L.ptr(t.acts) if t.discrete else None)
            setattr(a, f"next{k}", L.ptr(t.next_obs))
            setattr(a, f"done{k}", L.ptr(t.dones))
            setattr(a, f"idx{k}", L.ptr(i))
            setattr(a, f"n{k}", n)
        a.n_expert, a.loss_scale = n_expert, loss_scale
        a.bce_ws, a.stats = L.ptr(bce_ws), L.ptr(stats)
        a.accumulate = int(accumulate)
        a.adam = 0
        if adam is not None:
            g = adam.param_groups[0]
            adam.step_count += 1
            b1, b2 = g["betas"]
            a.adam = 1
            a.exp_avg, a.exp_avg_sq = L.ptr(adam.exp_avg), L.ptr(adam.exp_avg_sq)
            a.beta1, a.beta2, a.adam_eps, a.weight_decay = b1, b2, g["eps"], g["weight_decay"]
            a.step_size = g["lr"] / (1.0 - b1 ** adam.step_count)
            a.bc2_sqrt = (1.0 - b2 ** adam.step_count) ** 0.5
        use_p = pnorm is not None and a.update_norm and self.use_state and 0 < pnorm_dim <= mlp.dims[0]
        a.pnorm_mean = L.ptr(pnorm.running_mean) if use_p else None
        a.pnorm_var = L.ptr(pnorm.running_var) if use_p else None
        a.pnorm_count = L.ptr(pnorm.count) if use_p else None
        a.pnorm_dim = pnorm_dim if use_p else 0
        a.gp_e = None
        if gp is not None:
            e, coef, target = gp
            gws = self.fused_gp_ws(n_expert)
            if gws is None or n0 != n1 or n0 != n_expert or e.numel() != n_expert:
                raise RuntimeError("the fused gradient penalty needs equal expert / generator halves on the 128 / 256-wide "
                                   "fused update")
            a.gp_e, a.gp_coef, a.gp_target = L.ptr(e), float(coef), float(target)
            a.gp_ws, a.gp_out = L.ptr(gws), L.ptr(ws["gp_out"])

    def _step_used(self, ws, pre):
        # what this step read / normalised with (the opt-in gradient penalty evaluates the net at the same point)
        nrm = self.mlp.norm
        ws["X_used"] = ws["X"] if pre is None else pre[0]["X_all"][pre[1]]
        if nrm is None:
            ws["norm_used"] = None
        elif pre is not None and pre[0]["has_moments"]:
            ws["norm_used"] = (pre[0]["snap"][pre[1], 0], pre[0]["snap"][pre[1], 1], nrm.eps)
        else:
            ws["norm_used"] = (nrm.running_mean, nrm.running_var, nrm.eps)
        return ws


    def fused_ws_of(self, R: int) -> Optional[th.Tensor]:
        """The fused-update workspace of R-row updates (None: the general path runs this shape)."""
        return self._step_workspace(R)["fused_ws"]

    def fused_adam_step(self, adam, R: int, grad_scale: float) -> None:
        """Data-parallel tail of a fused update: the flat gradient (already summed over the ranks) scaled by
        `grad_scale`, Adam's step, and the refresh of the weight images the tile kernels read (`ia_disc_fused_adam`)."""
        import ctypes as C
        mlp = self.mlp
        ws = self._step_workspace(R)
        if adam.flat.data_ptr() != mlp.flat.data_ptr() or adam.flat.numel() != mlp.n_params:
            raise RuntimeError("the fused update needs the optimiser over the net's flat parameter buffer")
        L.call("ia_disc_fused_adam", C.byref(mlp.desc), L.ptr(mlp.flat), float(grad_scale), R, mlp.ldx,
               L.ptr(ws["fused_ws"]), adam.next_step_args(), L.stream())


class RewardNetWrapper(RewardNet):
    """`rewards/reward_nets.py:227-272`."""

    def __init__(self, base: RewardNet):
        super().__init__(base.observation_space, base.action_space, base.normalize_images)
        self._base = base
        self._store = base._store

    @property
    def base(self) -> RewardNet:
        return self._base

    def _children(self):
        return [self._base]

    def _named_stacks(self):
        return [(f"_base.{p}", s) for p, s in self._base._named_stacks()]

    def _named_norms(self):
        return [(f"_base.{p}", n) for p, n in self._base._named_norms()]

    def preprocess(self, state, action, next_state, done):
        return self.base.preprocess(state, action, next_state, done)


class ForwardWrapper(RewardNetWrapper):
    """`rewards/reward_nets.py:275-300`."""

    def __init__(self, base: RewardNet):
        super().__init__(base)
        if isinstance(base, PredictProcessedWrapper):
            raise ValueError("ForwardWrapper cannot be applied on top of PredictProcessedWrapper!")


class PredictProcessedWrapper(RewardNetWrapper):
    """`rewards/reward_nets.py:303-353`: forward / predict / predict_th pass through."""

    def _forward_table(self, sources, tag, out_act=L.ACT_NONE):
        return self.base._forward_table(sources, tag, out_act)

    def forward_plan(self):
        # (`forward` passes through: `rewards/reward_nets.py:303-353` -- also for NormalizedRewardNet, which normalises in
        #  `predict_processed` only; a subclass may have changed `_forward_table`)
        if type(self)._forward_table is not PredictProcessedWrapper._forward_table:
            return None
        return self.base.forward_plan()

    def predict(self, state, action, next_state, done):
        return self.base.predict(state, action, next_state, done)

    def predict_th(self, state, action, next_state, done):
        return self.base.predict_th(state, action, next_state, done)

    def disc_forward(self, sources, mb_rows, logp):
        return self.base.disc_forward(sources, mb_rows, logp)

    def disc_backward(self, d_logits, accumulate, adam=None):
        return self.base.disc_backward(d_logits, accumulate, adam)


class BasicPotentialMLP:
    """`rewards/reward_nets.py:812-839`."""

    def __init__(self, observation_space, hid_sizes: Sequence[int], **kwargs):
        self.obs_dim = _dims_of(observation_space)
        self._potential_net = DenseStack(in_size=self.obs_dim, hid_sizes=hid_sizes, squeeze_output=True,
                                         flatten_input=True, **kwargs)


class ShapedRewardNet(ForwardWrapper):
    """`rewards/reward_nets.py:674-736`: f = base(s,a,s',d) + gamma*(1-d)*h(s') - h(s)."""

    def __init__(self, base: RewardNet, potential, discount_factor: float):
        super().__init__(base)
        if not isinstance(base, BasicRewardNet) or not isinstance(potential, BasicPotentialMLP):
            raise NotImplementedError("HIP ShapedRewardNet is built for BasicRewardNet + BasicPotentialMLP")
        self.potential = potential
        self.discount_factor = float(discount_factor)
        self._store.add(potential._potential_net)
        self._aux: Dict[int, Dict[str, th.Tensor]] = {}

    def _named_stacks(self):
        return super()._named_stacks() + [("potential._potential_net.", self.potential._potential_net)]

    def to(self, device):
        self._aux = {}
        return super().to(device)

    def _aux_ws(self, R: int) -> Dict[str, th.Tensor]:
        ws = self._aux.get(R)
        if ws is None:
            dev = self.device
            ws = {k: th.empty(R, device=dev) for k in ("dones", "logits", "dg", "dh_cur", "dh_next")}
            ws["dones4"] = th.empty(R, 4, device=dev)
            self._aux[R] = ws
        return ws

    def _shaped(self, sources, tag: str, train_ws: bool, logp: Optional[th.Tensor]) -> th.Tensor:
        base, pot = self._base, self.potential._potential_net
        R = sum(n for _, _, n in sources)
        mk = (lambda s, t: s.train_workspace(R, t)) if train_ws else (lambda s, t: s.workspace(R, t))
        wg, wn, wc = mk(base.mlp, tag), mk(pot, tag + "_next"), mk(pot, tag + "_cur")
        aux = self._aux_ws(R)
        row = 0
        for table, idx, n in sources:
            gather_concat(table, idx, n, base.obs_dim, base.act_dim, base.flags, wg["X"], base.mlp.ldx, row)
            gather_concat(table, idx, n, base.obs_dim, base.act_dim, (True, False, False, False), wn["X"], pot.ldx,
                          row, state_from_next=True)
            gather_concat(table, idx, n, base.obs_dim, base.act_dim, (True, False, False, False), wc["X"], pot.ldx, row)
            gather_concat(table, idx, n, base.obs_dim, base.act_dim, (False, False, False, True), aux["dones4"], 4, row)
            row += n
        # reference order (reward_nets.py:708-710): base, potential(next_state), potential(state)
        g = base.mlp.forward_rows(wg, R)
        h_next = pot.forward_rows(wn, R)
        h_cur = pot.forward_rows(wc, R)
        aux["dones"].copy_(aux["dones4"][:, 0])
        L.call("ia_airl_logits", L.ptr(g), L.ptr(h_cur), L.ptr(h_next), L.ptr(aux["dones"]), L.ptr(logp),
               self.discount_factor, R, L.ptr(aux["logits"]), L.stream())
        self._last = (wg, wn, wc, aux, R)
        return aux["logits"]

    def _forward_table(self, sources, tag, out_act=L.ACT_NONE):
        out = self._shaped(sources, tag, False, None)
        if out_act == L.ACT_SOFTPLUS:
            out = -th.nn.functional.logsigmoid(-out)
        return out

    def disc_forward(self, sources, mb_rows, logp):
        if logp is None:
            raise TypeError("Non-None `log_policy_act_prob` is required for this method.")
        return self._shaped(sources, "disc", True, logp)

    # ---- fused update (csrc/airl_fused.hip): the scripts' default geometry in 5 launches behind the batch assembly
    def fused_step_ok(self) -> bool:
        base, pot = self._base.mlp, self.potential._potential_net
        if len(base.dims) != 3 or len(pot.dims) != 4 or base.dims[-1] != 1 or pot.dims[-1] != 1:
            return False
        if base.desc.hidden_act != L.ACT_RELU or pot.desc.hidden_act != L.ACT_RELU or not FUSED_AIRL_STEP:
            return False
        if any(n is not None and not n.is_chan for n in (base.norm, pot.norm)):   # EMANorm: stack-by-stack path
            return False
        return bool(L.load().ia_airl_fused_ok(base.dims[0], pot.dims[0], base.dims[1], pot.dims[1], pot.dims[2]))

    def fused_prepare(self, sources, pol_obs: Optional[th.Tensor] = None, pol_act: Optional[th.Tensor] = None,
                      dp=None, _args_only: bool = False) -> None:
        """First half of one whole `train_disc` minibatch (= batch) of `common.py:317-374` for this net: the batch
        assembly (`ia_airl_prepare`: base inputs, next-state and state batches, dones -- and, when asked, the rows the
        generator policy's log pi(a|s) reads) and the train-mode input statistics (`ia_airl_stats_merge`; potential:
        next-state batch, then state batch, `reward_nets.py:708-710` order). Two launches. `dp` (world > 1): the three
        moment sets of all ranks are exchanged in one all-gather in between and merged in rank order -- the statistics
        update of one process on the concatenated batch."""
        base, pot = self._base, self.potential._potential_net
        bm = base.mlp
        (t0, i0, n0), (t1, i1, n1) = sources
        R = n0 + n1
        dev = self.device
        ws = self._aux.get(("fused", R))
        if ws is None:
            nblk = int(L.load().ia_airl_fused_slabs(R))
            P = bm.n_params + pot.n_params
            nrn = -(-R // 256)
            ws = dict(Xb=th.zeros(R, bm.ldx, device=dev), Sn=th.zeros(R, pot.ldx, device=dev),
                      Sc=th.zeros(R, pot.ldx, device=dev), dones=th.empty(R, device=dev),
                      Ab=th.empty(R, bm.ldx, device=dev), Db1=th.empty(R, 32, device=dev),
                      Ap=th.empty(2 * R, pot.ldx, device=dev), H1=th.empty(2 * R, 32, device=dev),
                      Dp1=th.empty(2 * R, 32, device=dev), Dp2=th.empty(2 * R, 32, device=dev),
                      part=th.empty(nblk, P, device=dev), logits=th.empty(R, device=dev),
                      bce_part=th.zeros(nblk * 8, device=dev), ticket=th.zeros(2, dtype=th.int32, device=dev),
                      snapA=th.empty(2, pot.dims[0], device=dev), nblk=nblk,
                      ws_all=th.empty(nrn * 2 * (bm.dims[0] + 2 * pot.dims[0]), device=dev))
            nb_, np_ = nrn * 2 * bm.dims[0], nrn * 2 * pot.dims[0]   # one record [base | next | current]: one all-gather
            ws["ws_b"], ws["ws_n"], ws["ws_c"] = (ws["ws_all"][:nb_], ws["ws_all"][nb_:nb_ + np_],
                                                  ws["ws_all"][nb_ + np_:])
            ws["flags"] = [int(f) for f in base.flags]
            ws["out"] = (ws["Xb"].data_ptr(), bm.ldx, ws["Sn"].data_ptr(), ws["Sc"].data_ptr(), pot.ldx,
                         ws["dones"].data_ptr())
            self._aux[("fused", R)] = ws
        bn, pn = bm.norm, pot.norm
        upd_b = bn is not None and bm.training
        upd_p = pn is not None and pot.training
        pb = ws["ws_b"].data_ptr() if upd_b else None
        pnx, pc = (ws["ws_n"].data_ptr(), ws["ws_c"].data_ptr()) if upd_p else (None, None)
        if _args_only:   # (`airl_round_c`: the workspace and what this call would have left in `_fused`, no launch)
            self._fused = (ws, R, n0, upd_p)
            return
        acts = lambda t: (None, t.acts.data_ptr()) if t.discrete else (t.acts.data_ptr(), None)
        st = L.stream()
        L.call("ia_airl_prepare", t0.obs.data_ptr(), *acts(t0), t0.next_obs.data_ptr(), t0.dones.data_ptr(), L.ptr(i0), n0,
               t1.obs.data_ptr(), *acts(t1), t1.next_obs.data_ptr(), t1.dones.data_ptr(), L.ptr(i1), n1, base.obs_dim,
               base.act_dim, *ws["flags"], *ws["out"], pb, pnx, pc, L.ptr(pol_obs), L.ptr(pol_act), st)
        if upd_b or upd_p:   # h(s') is normalised with the statistics after ITS update (snapA), h(s) after the second one
            groups, gstride = 1, 0
            if dp is not None and dp.world > 1:
                rec = ws["ws_all"].numel()
                allm = dp.all_gather_flat(ws["ws_all"])
                groups, gstride = dp.world, rec
                nb_, np_ = ws["ws_b"].numel(), ws["ws_n"].numel()
                base_p = allm.data_ptr()
                pb = base_p if upd_b else None
                pnx, pc = (base_p + 4 * nb_, base_p + 4 * (nb_ + np_)) if upd_p else (None, None)
                ws["_gathered"] = allm   # keeps the buffer alive until the merge has been enqueued (and beyond: reused)
            L.call("ia_airl_stats_merge", pb, pnx, pc, groups, gstride, R, bm.dims[0], pot.dims[0],
                   *((bn.running_mean.data_ptr(), bn.running_var.data_ptr(), bn.count.data_ptr()) if upd_b else (None,) * 3),
                   *((pn.running_mean.data_ptr(), pn.running_var.data_ptr(), pn.count.data_ptr()) if upd_p else (None,) * 3),
                   ws["snapA"].data_ptr(), ws["ticket"].data_ptr() + 4, st)
        self._fused = (ws, R, n0, upd_p)

    def airl_round_c(self, drawn, mb: int, pol, pol_obs: th.Tensor, pol_act: th.Tensor, logp: th.Tensor, snaps,
                     scale: float, stats_rows: th.Tensor, adam, gp=None) -> th.Tensor:
        """The n updates of one round (`drawn[k]` = ((expert table, index rows), (generator table, index rows)), `mb` rows
        each) through ONE C call (`ia_airl_round`): per update the calls `fused_prepare` + `policy.log_prob_rows` +
        `fused_finish` make, in their order, with the optimiser step fused. `snaps` `[n, 2, obs_dim]` (or None: the
        policy's live statistics, which no update changes): the statistics update k's log pi(a|s) is normalised with.
        `gp = ([n] device weight vectors, coef, target)`: the opt-in gradient penalty of every update (`fused_grad_penalty`
        between the slab reduction and the optimiser step; its mean of the last update is in `self._fused[0]["gp"]["pen"]`).
        Returns the logits of the last update."""
        import ctypes as C
        base, pot = self._base, self.potential._potential_net
        bm = base.mlp
        (t0, i0), (t1, i1) = drawn[0]
        self.fused_prepare([(t0, i0, mb), (t1, i1, mb)], pol_obs, pol_act, _args_only=True)   # (workspace of this batch size)
        ws, R, n0, upd_p = self._fused
        bn, pn = bm.norm, pot.norm
        upd_b = bn is not None and bm.training
        if self._store.grad.numel() != bm.n_params + pot.n_params or self._store.flat.data_ptr() != bm.flat.data_ptr():
            raise RuntimeError("the fused AIRL update needs a parameter store of exactly the reward and potential stacks")
        if adam.flat.data_ptr() != bm.flat.data_ptr() or adam.flat.numel() != bm.n_params + pot.n_params:
            raise RuntimeError("the fused AIRL update needs the optimiser over the net's flat parameter buffer")
        t = L.AirlUpdateArgs()   # what no update of the round changes
        for k, tb in enumerate((t0, t1)):
            setattr(t, f"obs{k}", tb.obs.data_ptr())
            setattr(t, f"act{k}_f32", None if tb.discrete else tb.acts.data_ptr())
            setattr(t, f"act{k}_i64", tb.acts.data_ptr() if tb.discrete else None)
            setattr(t, f"next{k}", tb.next_obs.data_ptr())
            setattr(t, f"done{k}", tb.dones.data_ptr())
            setattr(t, f"n{k}", mb)
        t.obs_dim, t.act_dim = base.obs_dim, base.act_dim
        t.use_state, t.use_action, t.use_next_state, t.use_done = ws["flags"]
        t.Xb, t.ldb, t.Sn, t.Sc, t.ldp, t.dones = ws["out"]
        t.ws_b = ws["ws_b"].data_ptr() if upd_b else None
        t.ws_n, t.ws_c = (ws["ws_n"].data_ptr(), ws["ws_c"].data_ptr()) if upd_p else (None, None)
        t.pol_obs, t.pol_act = L.ptr(pol_obs), L.ptr(pol_act)
        t.Db, t.Dp = bm.dims[0], pot.dims[0]
        if upd_b:
            t.bmean, t.bvar, t.bcount = bn.running_mean.data_ptr(), bn.running_var.data_ptr(), bn.count.data_ptr()
        if upd_p:
            t.pmean, t.pvar, t.pcount = pn.running_mean.data_ptr(), pn.running_var.data_ptr(), pn.count.data_ptr()
        t.snapA, t.merge_ticket = ws["snapA"].data_ptr(), ws["ticket"].data_ptr() + 4
        t.pol = C.pointer(pol.desc)
        t.pol_params, t.pol_params_t, t.logp = L.ptr(pol._flat), L.ptr(pol._flat_t), L.ptr(logp)
        live_nm, live_nv = pol._norm_ptrs()
        if bn is not None:
            t.f_bmean, t.f_bvar, t.beps = bn.running_mean.data_ptr(), bn.running_var.data_ptr(), float(bn.eps)
        if pn is not None:
            t.pmeanB, t.pvarB, t.peps = pn.running_mean.data_ptr(), pn.running_var.data_ptr(), float(pn.eps)
            t.pmeanA, t.pvarA = ((ws["snapA"].data_ptr(), ws["snapA"].data_ptr() + 4 * pot.dims[0]) if upd_p
                                 else (t.pmeanB, t.pvarB))
        t.params_base, t.params_pot = bm.flat.data_ptr(), pot.flat.data_ptr()
        t.gamma, t.scale, t.n_expert = self.discount_factor, float(scale), mb
        t.Ab, t.ldab, t.Db1, t.Ap, t.ldap = ws["Ab"].data_ptr(), bm.ldx, ws["Db1"].data_ptr(), ws["Ap"].data_ptr(), pot.ldx
        t.H1, t.Dp1, t.Dp2 = ws["H1"].data_ptr(), ws["Dp1"].data_ptr(), ws["Dp2"].data_ptr()
        t.partials, t.logits = ws["part"].data_ptr(), ws["logits"].data_ptr()
        t.bce_part, t.ticket = ws["bce_part"].data_ptr(), ws["ticket"].data_ptr()
        g = adam.param_groups[0]
        b1, b2 = g["betas"]
        t.adam.grads, t.adam.exp_avg, t.adam.exp_avg_sq = adam.grad.data_ptr(), adam.exp_avg.data_ptr(), adam.exp_avg_sq.data_ptr()
        t.adam.beta1, t.adam.beta2, t.adam.eps, t.adam.weight_decay = b1, b2, g["eps"], g["weight_decay"]
        if gp is not None:
            gws = self._gp_workspace(ws, mb)
            t.gp_coef, t.gp_target, t.n_slabs, t.n_params = float(gp[1]), float(gp[2]), ws["nblk"], bm.n_params + pot.n_params
            for fld, key in (("U1b", "U1b"), ("Cb", "Cb"), ("U1p", "U1p"), ("Cp", "Cp"), ("U2p", "U2p"), ("V1p", "V1p"),
                             ("gp_partials", "part"), ("pen_part", "pen_part"), ("pen_out", "pen"), ("gp_ticket", "ticket")):
                setattr(t, fld, gws[key].data_ptr())
        n = len(drawn)
        arr = ws.get("round_args")
        if arr is None or len(arr) < n:
            arr = ws["round_args"] = (L.AirlUpdateArgs * n)()
        so = pol.obs_dim * 4
        stats_base, stats_stride = stats_rows.data_ptr(), stats_rows.stride(0) * 4
        snap_base = None if snaps is None else snaps.data_ptr()
        for k in range(n):
            a = arr[k]
            C.memmove(C.byref(a), C.byref(t), C.sizeof(t))
            (_, e_idx), (_, g_idx) = drawn[k]
            a.idx0, a.idx1 = L.ptr(e_idx), L.ptr(g_idx)
            a.stats = stats_base + k * stats_stride
            if snap_base is None:
                a.pol_norm_mean, a.pol_norm_var = live_nm, live_nv
            else:
                a.pol_norm_mean, a.pol_norm_var = snap_base + k * 2 * so, snap_base + k * 2 * so + so
            adam.step_count += 1
            a.adam.step_size = g["lr"] / (1.0 - b1 ** adam.step_count)
            a.adam.bc2_sqrt = (1.0 - b2 ** adam.step_count) ** 0.5
            if gp is not None:
                a.gp_e = L.ptr(gp[0][k])
        L.call("ia_airl_round", arr, n, L.stream())
        return ws["logits"]

    def _gp_workspace(self, ws, B: int):
        """Work areas of `ia_airl_gp_shaped` for B interpolated rows (kept with the fused workspace of 2 B rows)."""
        gws = ws.get("gp")
        if gws is None:
            bm, pot = self._base.mlp, self.potential._potential_net
            dev = self.device
            nblk = int(L.load().ia_airl_fused_slabs(B))
            gws = ws["gp"] = dict(U1b=th.empty(B, 32, device=dev), Cb=th.empty(B, bm.ldx, device=dev),
                                  U1p=th.empty(2 * B, 32, device=dev), Cp=th.empty(2 * B, pot.ldx, device=dev),
                                  U2p=th.empty(2 * B, 32, device=dev), V1p=th.empty(2 * B, 32, device=dev),
                                  part=th.zeros(nblk, bm.n_params + pot.n_params, device=dev),
                                  pen_part=th.empty(nblk, device=dev), pen=th.empty(1, device=dev),
                                  ticket=th.zeros(1, dtype=th.int32, device=dev))
        return gws

    def fused_finish(self, logp: th.Tensor, scale: float, stats_dev: th.Tensor, adam) -> th.Tensor:
        """Second half: `ia_airl_step_shaped` on the prepared batch (forward, logits, BCE + statistics, deltas,
        weight-gradient GEMMs, split-K reduction fused with Adam: five launches). Returns the logits `[R]`.
        `adam=None`: the reduced gradient is left in the store's flat gradient buffer instead."""
        base, pot = self._base.mlp, self.potential._potential_net
        ws, R, n_expert, upd_p = self._fused
        bn, pn = base.norm, pot.norm
        A = B = (None, None)
        if pn is not None:
            B = (pn.running_mean.data_ptr(), pn.running_var.data_ptr())
            A = (ws["snapA"].data_ptr(), ws["snapA"].data_ptr() + 4 * pot.dims[0]) if upd_p else B
        # the slabs are [base | potential] wide: the store must hold exactly these two stacks (also when the caller
        # reduces into the store's gradient itself, `adam=None`)
        if self._store.grad.numel() != base.n_params + pot.n_params or self._store.flat.data_ptr() != base.flat.data_ptr():
            raise RuntimeError("the fused AIRL update needs a parameter store of exactly the reward and potential stacks")
        if adam is not None and (adam.flat.data_ptr() != base.flat.data_ptr()
                                 or adam.flat.numel() != base.n_params + pot.n_params):
            raise RuntimeError("the fused AIRL update needs the optimiser over the net's flat parameter buffer")
        L.call("ia_airl_step_shaped", ws["out"][0], base.ldx, base.dims[0], ws["out"][2], ws["out"][3], pot.ldx,
               pot.dims[0], ws["out"][5], L.ptr(logp),
               *((bn.running_mean.data_ptr(), bn.running_var.data_ptr(), float(bn.eps)) if bn is not None else (None, None, 0.0)),
               *A, *B, float(pn.eps) if pn is not None else 0.0,
               base.flat.data_ptr(), pot.flat.data_ptr(), self.discount_factor, float(scale), R, n_expert,
               ws["Ab"].data_ptr(), base.ldx, ws["Db1"].data_ptr(), ws["Ap"].data_ptr(), pot.ldx, ws["H1"].data_ptr(),
               ws["Dp1"].data_ptr(), ws["Dp2"].data_ptr(), ws["part"].data_ptr(), ws["logits"].data_ptr(),
               L.ptr(stats_dev), ws["bce_part"].data_ptr(), ws["ticket"].data_ptr(),
               adam.next_step_args() if adam is not None else None, L.stream())
        if adam is None:   # the caller adds to the gradient (gradient penalty) and steps the optimiser itself
            g = self._store.grad
            L.call("ia_reduce_partials", ws["part"].data_ptr(), ws["nblk"], g.numel(), 1.0, 0, g.data_ptr(), L.stream())
        return ws["logits"]

    def fused_grad_penalty(self, e: th.Tensor, coef: float, target: float) -> th.Tensor:
        """Opt-in gradient penalty (`grad_penalty.py`) of the shaped reward on the batch `fused_prepare` assembled last,
        input statistics as they stand (frozen): `ia_airl_gp_shaped` ADDS the parameter gradient to the store's flat
        gradient buffer. Returns the 1-element device tensor mean (|grad f| - target)^2."""
        base, pot = self._base, self.potential._potential_net
        bm = base.mlp
        ws, R, n0, _ = self._fused
        B = R // 2
        if n0 != B or R != 2 * B:
            raise ValueError("the gradient penalty interpolates expert and generator rows pairwise: equal halves needed")
        gws = self._gp_workspace(ws, B)
        bn, pn = bm.norm, pot.norm
        stat = lambda n: (n.running_mean.data_ptr(), n.running_var.data_ptr(), float(n.eps)) if n is not None else (None, None, 0.0)
        L.call("ia_airl_gp_shaped", ws["out"][0], bm.ldx, bm.dims[0], ws["out"][2], ws["out"][3], pot.ldx, pot.dims[0],
               ws["out"][5], L.ptr(e), *stat(bn), *stat(pn), bm.flat.data_ptr(), pot.flat.data_ptr(), base.obs_dim,
               base.act_dim, *ws["flags"], self.discount_factor, float(coef), float(target), B,
               *[gws[k].data_ptr() for k in ("U1b", "Cb", "U1p", "Cp", "U2p", "V1p", "part", "pen_part", "pen", "ticket")],
               self._store.grad.data_ptr(), L.stream())
        return gws["pen"]

    def fused_batches(self):
        """(Xb, ldb, Sn, Sc, ldp, dones) of the batch `fused_prepare` assembled last."""
        ws = self._fused[0]
        return ws["Xb"], self._base.mlp.ldx, ws["Sn"], ws["Sc"], self.potential._potential_net.ldx, ws["dones"]

    def disc_step_fused(self, sources, logp: th.Tensor, scale: float, stats_dev: th.Tensor, adam) -> th.Tensor:
        self.fused_prepare(sources)
        return self.fused_finish(logp, scale, stats_dev, adam)

    def disc_backward(self, d_logits, accumulate, adam=None):
        wg, wn, wc, aux, R = self._last
        L.call("ia_airl_route_grad", L.ptr(d_logits), L.ptr(aux["dones"]), self.discount_factor, R, L.ptr(aux["dg"]),
               L.ptr(aux["dh_cur"]), L.ptr(aux["dh_next"]), L.stream())
        pot = self.potential._potential_net
        self._base.mlp.backward_rows(wg, R, aux["dg"], accumulate)
        pot.backward_rows(wn, R, aux["dh_next"], accumulate)
        pot.backward_rows(wc, R, aux["dh_cur"], True)
        return False


class BasicShapedRewardNet(ShapedRewardNet):
    """`rewards/reward_nets.py:739-809`."""

    def __init__(self, observation_space, action_space, *, reward_hid_sizes: Sequence[int] = (32,),
                 potential_hid_sizes: Sequence[int] = (32, 32), use_state: bool = True, use_action: bool = True,
                 use_next_state: bool = False, use_done: bool = False, discount_factor: float = 0.99, **kwargs):
        base = BasicRewardNet(observation_space, action_space, use_state=use_state, use_action=use_action,
                              use_next_state=use_next_state, use_done=use_done, hid_sizes=reward_hid_sizes, **kwargs)
        pot = BasicPotentialMLP(observation_space, hid_sizes=potential_hid_sizes, **kwargs)
        super().__init__(base, pot, discount_factor=discount_factor)


class NormalizedRewardNet(PredictProcessedWrapper):
    """`rewards/reward_nets.py:613-671`: normalises `predict_processed` with a running norm."""

    def __init__(self, base: RewardNet, normalize_output_layer: Type):
        super().__init__(base)
        if not (isinstance(normalize_output_layer, type) and issubclass(normalize_output_layer, RunningNorm)):
            raise NotImplementedError("output normalisation: imitation_amd.RunningNorm or imitation_amd.EMANorm")
        self.normalize_output_layer = normalize_output_layer(1)

    def _named_norms(self):
        return super()._named_norms() + [("normalize_output_layer.", self.normalize_output_layer)]

    def to(self, device):
        self.normalize_output_layer.to(device)
        return super().to(device)

    def predict_processed(self, state, action, next_state, done, update_stats: bool = True, **kwargs) -> np.ndarray:
        with evaluating(self):
            raw = th.as_tensor(self.base.predict_processed(state, action, next_state, done, **kwargs),
                               device=self.device)
            rew = self.normalize_output_layer(raw.reshape(-1, 1)).reshape(-1).cpu().numpy().flatten()
        if update_stats:
            self.normalize_output_layer.update_stats(raw.reshape(-1, 1).contiguous())
        assert rew.shape == np.shape(state)[:1]
        return rew

    def predict_processed_rollout(self, table, T, n, update_stats: bool = True):
        raw = self.base.predict_processed_rollout(table, T, n).contiguous()
        out = th.empty_like(raw)
        nl = self.normalize_output_layer
        if not nl.is_chan:   # EMANorm: step by step (normalise with the statistics so far, then update), like the wrapper
            r2, o2 = raw.view(T, n, 1), out.view(T, n, 1)
            for t in range(T):
                nl.apply(r2[t], o2[t], 1, 1, n)
                if update_stats:
                    nl.update_stats(r2[t])
            return out
        if nl.dp is not None and nl.dp.world > 1:
            # data parallelism: this rank relabelled its own env batch; per step the statistics absorb the batch of ALL
            # ranks (one all-gather of the [T, 2] per-step moments), so every rank keeps the statistics of one process on
            # the env batches side by side
            mom = th.empty(T, 2, device=raw.device)
            L.call("ia_reward_step_moments", L.ptr(raw), T, n, L.ptr(mom), L.stream())
            allm = nl.dp.all_gather_flat(mom.reshape(-1))
            L.call("ia_reward_norm_sequential_groups", L.ptr(raw), T, n, nl.eps, int(update_stats), L.ptr(allm),
                   nl.dp.world, L.ptr(nl.running_mean), L.ptr(nl.running_var), L.ptr(nl.count), L.ptr(out), L.stream())
            return out
        L.call("ia_reward_norm_sequential", L.ptr(raw), T, n, nl.eps, int(update_stats), L.ptr(nl.running_mean),
               L.ptr(nl.running_var), L.ptr(nl.count), L.ptr(out), L.stream())
        return out
