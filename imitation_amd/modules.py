"""`nn.Module` reward networks on the HIP custom ops (`imitation_amd.ops`): the reference's reward-net
PLUGIN contract (`rewards/reward_nets.py:16-224`) -- `forward(state, action, next_state, done)` carries an
autograd graph, `AdversarialTrainer.train_disc` trains it with `loss.backward()`
(`algorithms/adversarial/common.py:353-372`) -- with the forward / backward arithmetic in libimitation_hip.so.

`BasicRewardNet` here has the reference's module tree and `state_dict` keys (`mlp.normalize_input.*`,
`mlp.dense0.*`, ..., `mlp.dense_final.*`), so checkpoints interchange with the reference and with the
fused state-holder nets of `imitation_amd.reward_nets` (same keys). User-defined reward nets subclass
`RewardNet` below and compose `Mlp` / `RunningNorm` / any differentiable torch-on-ROCm op in `forward`.

The state-holder nets (`imitation_amd.reward_nets`) remain the fast path: one fused C call per update.
"""
from __future__ import annotations

import abc
import collections
from typing import Iterable, Optional, Sequence, Tuple, Type

import numpy as np
import torch as th
from torch import nn

from imitation_amd import ops, spaces
from imitation_amd.reward_nets import preprocess_space


class RunningNorm(nn.Module):
    """`util/networks.py:47-134`: buffers `running_mean`, `running_var`, int32 `count`; a train-mode forward
    first merges the batch into the statistics (Chan), then normalises with them."""

    def __init__(self, num_features: int, eps: float = 1e-5):
        super().__init__()
        self.eps = eps
        self.register_buffer("running_mean", th.zeros(num_features))
        self.register_buffer("running_var", th.ones(num_features))
        self.register_buffer("count", th.zeros((), dtype=th.int32))
        self.dp = None   # imitation_amd.distributed.DataParallel (set by the trainer): merge the moments of all ranks

    def __getstate__(self):
        # `th.save(net)` (save_reward_net) must not pickle the process-group handle: a net reloaded in a single process
        # would otherwise call collectives without a group; the trainer sets `dp` again when it adopts a net
        state = dict(self.__dict__)
        state["dp"] = None
        return state

    def reset_running_stats(self) -> None:
        self.running_mean.zero_()
        self.running_var.fill_(1)
        self.count.zero_()

    def update_stats(self, batch: th.Tensor) -> None:
        x = batch.detach().reshape(batch.shape[0], -1).float().contiguous()
        if self.dp is not None and self.dp.world > 1:
            # every rank contributes its rows: slab moments all-gathered, the same Chan merge everywhere -- the update
            # of one process on the concatenated batch, as `networks.RunningNorm` does it
            from imitation_amd import _lib as L
            R, F = x.shape
            ws = th.empty(int(L.load().ia_running_norm_ws_floats(R, F)), device=x.device)
            L.call("ia_running_norm_partial", L.ptr(x), F, R, F, L.ptr(ws), L.stream())
            ws_all = self.dp.all_gather_flat(ws)
            L.call("ia_running_norm_merge", L.ptr(ws_all), self.dp.world, R, F, F, L.ptr(self.running_mean),
                   L.ptr(self.running_var), L.ptr(self.count), L.stream())
            return
        th.ops.imitation_amd.running_norm_update(x, self.running_mean, self.running_var, self.count)

    def forward(self, x: th.Tensor) -> th.Tensor:
        flat = x.reshape(x.shape[0], -1)
        if self.training:
            with th.no_grad():
                self.update_stats(flat)
        return ops.running_norm_apply_fn(flat, self.running_mean, self.running_var, self.eps).reshape(x.shape)


class EMANorm(RunningNorm):
    """`util/networks.py:137-201`: exponentially weighted statistics; buffers `inv_learning_rate`, `num_batches`."""

    def __init__(self, num_features: int, decay: float = 0.99, eps: float = 1e-5):
        super().__init__(num_features, eps=eps)
        if not 0 < decay < 1:
            raise ValueError("decay must be between 0 and 1")
        self.decay = float(decay)
        self.register_buffer("inv_learning_rate", th.zeros(()))
        self.register_buffer("num_batches", th.zeros((), dtype=th.int32))

    def reset_running_stats(self) -> None:
        super().reset_running_stats()
        self.inv_learning_rate.zero_()
        self.num_batches.zero_()

    def update_stats(self, batch: th.Tensor) -> None:
        from imitation_amd import _lib as L
        x = batch.detach().reshape(batch.shape[0], -1).float().contiguous()
        R, F = x.shape
        ws = th.empty(int(L.load().ia_running_norm_ws_floats(R, F)), device=x.device)
        L.call("ia_running_norm_partial", L.ptr(x), F, R, F, L.ptr(ws), L.stream())
        groups = 1
        if self.dp is not None and self.dp.world > 1:
            ws, groups = self.dp.all_gather_flat(ws), self.dp.world
        L.call("ia_ema_norm_merge", L.ptr(ws), groups, R, F, F, L.ptr(self.running_mean), L.ptr(self.running_var),
               L.ptr(self.count), L.ptr(self.inv_learning_rate), L.ptr(self.num_batches), self.decay, L.stream())


_ACTS = {nn.ReLU: ops.ACT_RELU, nn.Tanh: ops.ACT_TANH}


class Mlp(nn.Module):
    """`build_mlp` (`util/networks.py:204-283`): children named as the reference names them (`normalize_input`,
    `dense{i}`, `dense_final`), ordinary `nn.Linear` parameters (so any torch optimiser and the reference's
    `state_dict` keys apply); `forward` packs them into the flat vector the `imitation_amd::mlp_*` ops take and
    autograd routes the flat gradient back to the individual parameters."""

    def __init__(self, in_size: int, hid_sizes: Iterable[int], out_size: int = 1, name: Optional[str] = None,
                 activation: Type[nn.Module] = nn.ReLU, dropout_prob: float = 0.0, squeeze_output: bool = False,
                 flatten_input: bool = False, normalize_input_layer: Optional[Type[nn.Module]] = None):
        super().__init__()
        if activation not in _ACTS:
            raise NotImplementedError(f"activation {activation} not supported on the HIP path (ReLU/Tanh)")
        if squeeze_output and out_size != 1:
            raise ValueError("squeeze_output is only applicable when out_size=1")
        prefix = "" if name is None else f"{name}_"
        # dropout (`util/networks.py:270-271`, off in every reference GAIL/AIRL config): the stack then runs layer
        # by layer -- each Linear on the HIP op, activation + `nn.Dropout` in between (the mask comes from the
        # DEVICE generator, so dropout runs are not seed-comparable with the reference's CPU masks)
        self.dropout_prob = float(dropout_prob)
        self._activation = activation() if dropout_prob > 0.0 else None
        self._dropout = nn.Dropout(dropout_prob) if dropout_prob > 0.0 else None
        self.dims = [int(in_size), *[int(h) for h in hid_sizes], int(out_size)]
        self.act = _ACTS[activation]
        self.squeeze_output, self.flatten_input = squeeze_output, flatten_input
        self._norm_name = None
        if normalize_input_layer:
            try:
                layer = normalize_input_layer(in_size)
            except TypeError as exc:
                raise ValueError(f"normalize_input_layer={normalize_input_layer} is not a valid normalization layer "
                                 "type accepting only one argument (in_size).") from exc
            self._norm_name = f"{prefix}normalize_input"
            self.add_module(self._norm_name, layer)
        self._dense = []
        for i in range(len(self.dims) - 2):
            self._dense.append(f"{prefix}dense{i}")
            self.add_module(self._dense[-1], nn.Linear(self.dims[i], self.dims[i + 1]))
        self._dense.append(f"{prefix}dense_final")
        self.add_module(self._dense[-1], nn.Linear(self.dims[-2], self.dims[-1]))

    def flat_parameters(self) -> th.Tensor:
        parts = []
        for n in self._dense:
            lin = getattr(self, n)
            parts += [lin.weight.reshape(-1), lin.bias.reshape(-1)]
        return th.cat(parts)

    def add_flat_grad(self, gflat: th.Tensor) -> None:
        """Adds a gradient given in the flat layout of `flat_parameters()` to the individual `.grad`s."""
        o = 0
        for n in self._dense:
            lin = getattr(self, n)
            for p_ in (lin.weight, lin.bias):
                piece = gflat[o:o + p_.numel()].view_as(p_)
                p_.grad = piece.clone() if p_.grad is None else p_.grad.add_(piece)
                o += p_.numel()

    def forward(self, x: th.Tensor) -> th.Tensor:
        if self.flatten_input:
            x = x.reshape(x.shape[0], -1)
        if self._norm_name is not None:
            x = getattr(self, self._norm_name)(x)
        if self._dropout is None:
            out = ops.mlp(x, self.flat_parameters(), self.dims, self.act)
        else:
            out = x
            for k, n in enumerate(self._dense):
                lin = getattr(self, n)
                out = ops.mlp(out, th.cat([lin.weight.reshape(-1), lin.bias.reshape(-1)]),
                              [lin.in_features, lin.out_features], ops.ACT_NONE)
                if k < len(self._dense) - 1:
                    out = self._dropout(self._activation(out))
        return out.squeeze(-1) if self.squeeze_output else out


class RewardNet(nn.Module, abc.ABC):
    """`rewards/reward_nets.py:16-224`, verbatim contract: `forward` on preprocessed device tensors (with
    grad), `preprocess` (numpy -> device tensors, one-hot / float), `predict_th`, `predict`,
    `predict_processed`, `device`, `dtype`."""

    def __init__(self, observation_space, action_space, normalize_images: bool = True):
        super().__init__()
        self.observation_space, self.action_space = observation_space, action_space
        self.normalize_images = normalize_images

    @abc.abstractmethod
    def forward(self, state: th.Tensor, action: th.Tensor, next_state: th.Tensor, done: th.Tensor) -> th.Tensor:
        """Rewards `[B]` for preprocessed tensors."""

    def preprocess(self, state, action, next_state, done) -> Tuple[th.Tensor, th.Tensor, th.Tensor, th.Tensor]:
        dev = self.device
        to = lambda a: th.as_tensor(np.ascontiguousarray(a) if isinstance(a, np.ndarray) else a).to(dev)
        s = preprocess_space(to(state), self.observation_space, self.normalize_images)
        a = preprocess_space(to(action), self.action_space, self.normalize_images)
        ns = preprocess_space(to(next_state), self.observation_space, self.normalize_images)
        d = to(done).to(th.float32)
        assert s.shape == ns.shape and len(a) == len(s)
        return s, a, ns, d

    def predict_th(self, state, action, next_state, done) -> th.Tensor:
        was = self.training
        self.eval()
        try:
            with th.no_grad():
                rew = self(*self.preprocess(state, action, next_state, done))
        finally:
            self.train(was)
        assert rew.shape == np.shape(state)[:1]
        return rew

    def predict(self, state, action, next_state, done) -> np.ndarray:
        return self.predict_th(state, action, next_state, done).detach().cpu().numpy().flatten()

    def predict_processed(self, state, action, next_state, done, **kwargs) -> np.ndarray:
        del kwargs
        return self.predict(state, action, next_state, done)

    @property
    def device(self) -> th.device:
        try:
            return next(self.parameters()).device
        except StopIteration:
            return th.device("cpu")

    @property
    def dtype(self) -> th.dtype:
        try:
            return next(self.parameters()).dtype
        except StopIteration:
            return th.get_default_dtype()


class BasicRewardNet(RewardNet):
    """`rewards/reward_nets.py:383-457`: MLP over the concatenation of the enabled, flattened inputs."""

    def __init__(self, observation_space, action_space, use_state: bool = True, use_action: bool = True,
                 use_next_state: bool = False, use_done: bool = False, **kwargs):
        super().__init__(observation_space, action_space)
        size = 0
        self.use_state, self.use_action = use_state, use_action
        self.use_next_state, self.use_done = use_next_state, use_done
        if use_state:
            size += spaces.flatdim(observation_space)
        if use_action:
            size += spaces.flatdim(action_space)
        if use_next_state:
            size += spaces.flatdim(observation_space)
        if use_done:
            size += 1
        full = {"hid_sizes": (32, 32), **kwargs, "in_size": size, "out_size": 1, "squeeze_output": True}
        self.mlp = Mlp(**full)

    def concat_inputs(self, state, action, next_state, done) -> th.Tensor:
        n = state.shape[0]
        picked = ((self.use_state, state), (self.use_action, action), (self.use_next_state, next_state),
                  (self.use_done, done))
        return th.cat([t.reshape(n, -1).float() for on, t in picked if on], dim=1)

    def forward(self, state, action, next_state, done):
        n = state.shape[0]
        row = self.concat_inputs(state, action, next_state, done)
        rew = self.mlp(row)
        assert rew.shape == (n,)
        return rew


class Cnn(nn.Module):
    """`build_cnn` (`util/networks.py:286-357`): `Conv2d(kernel_size, stride, padding) - activation` per hidden
    channel count, `AdaptiveAvgPool2d(1)`, `Flatten`, `Linear(out_size)`. Children carry the reference's names
    (`conv{i}`, `dense_final`) and torch's parameter layouts ([Cout, Cin, KH, KW]); `forward` takes channel-FIRST
    input like the reference's, runs channel-last internally (im2col + MFMA GEMM per layer) and permutes the
    weights on the fly."""

    def __init__(self, in_channels: int, hid_channels: Iterable[int], out_size: int = 1, name: Optional[str] = None,
                 activation: Type[nn.Module] = nn.ReLU, kernel_size: int = 3, stride: int = 1, padding="same",
                 dropout_prob: float = 0.0, squeeze_output: bool = False):
        super().__init__()
        if dropout_prob > 0.0:
            raise NotImplementedError("dropout is not built for HIP")
        if activation is not nn.ReLU:
            raise NotImplementedError("the HIP convolution fuses ReLU (the reference's default)")
        if squeeze_output and out_size != 1:
            raise ValueError("squeeze_output is only applicable when out_size=1")
        if padding == "same":
            if stride != 1 or kernel_size % 2 == 0:
                raise ValueError("padding='same' needs stride 1 and an odd kernel (as torch.nn.Conv2d does)")
            pad = kernel_size // 2
        else:
            pad = int(padding)
        prefix = "" if name is None else f"{name}_"
        self.stride, self.pad, self.squeeze_output = int(stride), pad, squeeze_output
        self._convs = []
        prev = int(in_channels)
        for i, ch in enumerate(hid_channels):
            self._convs.append(f"{prefix}conv{i}")
            self.add_module(self._convs[-1], nn.Conv2d(prev, int(ch), kernel_size, stride=stride, padding=padding))
            prev = int(ch)
        self._final = f"{prefix}dense_final"
        self.add_module(self._final, nn.Linear(prev, int(out_size)))
        self.dims_final = [prev, int(out_size)]

    def forward(self, x: th.Tensor) -> th.Tensor:
        h = x.permute(0, 2, 3, 1).contiguous()                  # [B, C, H, W] -> channel-last
        # every layer's ReLU backward rides in the NEXT layer's input-gradient epilogue (`ops.conv2d_nhwc`)
        for i, n in enumerate(self._convs[:-1]):
            conv = getattr(self, n)
            h = ops.conv2d_nhwc(h, conv.weight.permute(0, 2, 3, 1).contiguous(), conv.bias, self.stride, self.pad, relu=True,
                                x_is_relu=i > 0, dy_is_masked=True)
        if self._convs:   # the last convolution, its ReLU and the pool as one autograd node (fused backward)
            conv = getattr(self, self._convs[-1])
            pooled = ops.conv2d_relu_avgpool_nhwc(h, conv.weight.permute(0, 2, 3, 1).contiguous(), conv.bias, self.stride,
                                                  self.pad, x_is_relu=len(self._convs) > 1)
        else:
            pooled = ops.avgpool_nhwc_fn(h)
        fin = getattr(self, self._final)
        out = ops.mlp(pooled, th.cat([fin.weight.reshape(-1), fin.bias.reshape(-1)]), self.dims_final, ops.ACT_NONE)
        return out.squeeze(-1) if self.squeeze_output else out


def build_cnn(*args, **kwargs) -> Cnn:
    """`util/networks.py:286-357` signature."""
    return Cnn(*args, **kwargs)


def _is_image_space(space) -> bool:
    return (isinstance(space, spaces.Box) and len(space.shape) == 3 and space.dtype == np.uint8
            and bool(np.all(space.low == 0) and np.all(space.high == 255)))


class CnnRewardNet(RewardNet):
    """`rewards/reward_nets.py:460-610`: a CNN over the (channel-concatenated) image state / next state whose output
    has one entry per discrete action (two per action when `use_done`); the reward is the entry the one-hot action
    (and the done flag) selects."""

    def __init__(self, observation_space, action_space, use_state: bool = True, use_action: bool = True,
                 use_next_state: bool = False, use_done: bool = False, hwc_format: bool = True, **kwargs):
        super().__init__(observation_space, action_space)
        self.use_state, self.use_action = use_state, use_action
        self.use_next_state, self.use_done, self.hwc_format = use_next_state, use_done, hwc_format
        if not (use_state or use_next_state):
            raise ValueError("CnnRewardNet must take current or next state as input.")
        if not _is_image_space(observation_space):
            raise ValueError("CnnRewardNet requires observations to be images.")
        if use_action and not isinstance(action_space, spaces.Discrete):
            raise ValueError("CnnRewardNet can only use Discrete action spaces.")
        ch = observation_space.shape[-1] if hwc_format else observation_space.shape[0]
        n_in = ch * (int(use_state) + int(use_next_state))
        n_out = (int(action_space.n) if use_action else 1) * (2 if use_done else 1)
        full = {"hid_channels": (32, 32), **kwargs, "in_channels": n_in, "out_size": n_out, "squeeze_output": n_out == 1}
        self.cnn = Cnn(**full)

    def forward(self, state, action, next_state, done):
        images = [t for on, t in ((self.use_state, state), (self.use_next_state, next_state)) if on]
        if self.hwc_format:   # [B, H, W, C] frames: channel-first for the CNN's input contract
            images = [t.permute(0, 3, 1, 2) for t in images]
        out = self.cnn(th.cat(images, dim=1))
        if not self.use_action and not self.use_done:
            return out
        n = state.shape[0]
        if self.use_action:
            sel = action.reshape(n, -1).float()
            if self.use_done:   # first half of the outputs: done = False, second half: done = True
                d = done.reshape(n, 1).float()
                sel = th.cat([sel * (1 - d), sel * d], dim=1)
        else:
            sel = nn.functional.one_hot(done.long(), num_classes=2).float()
        return (out * sel).sum(dim=1)


class BasicPotentialMLP(nn.Module):
    """`rewards/reward_nets.py:812-839`."""

    def __init__(self, observation_space, hid_sizes: Sequence[int], **kwargs):
        super().__init__()
        self._potential_net = Mlp(in_size=spaces.flatdim(observation_space), hid_sizes=hid_sizes, squeeze_output=True,
                                  flatten_input=True, **kwargs)

    def forward(self, state: th.Tensor) -> th.Tensor:
        return self._potential_net(state)


class ShapedRewardNet(RewardNet):
    """`rewards/reward_nets.py:674-736`: `base(s, a, s', d) + gamma * (1 - d) * potential(s') - potential(s)`,
    composed by autograd from the two sub-modules."""

    def __init__(self, base: RewardNet, potential: nn.Module, discount_factor: float):
        super().__init__(base.observation_space, base.action_space, base.normalize_images)
        self._base = base
        self.potential = potential
        self.discount_factor = discount_factor

    @property
    def base(self) -> RewardNet:
        return self._base

    def forward(self, state, action, next_state, done):
        g = self._base(state, action, next_state, done)
        h_next, h_cur = self.potential(next_state).reshape(-1), self.potential(state).reshape(-1)
        alive = 1 - done.float()                     # no bootstrapping through a true episode end
        shaped = g + self.discount_factor * (alive * h_next)   # same association as the reference (`:727-733`)
        shaped = shaped - h_cur
        assert shaped.shape == state.shape[:1]
        return shaped


class BasicShapedRewardNet(ShapedRewardNet):
    """`rewards/reward_nets.py:739-809`."""

    def __init__(self, observation_space, action_space, *, reward_hid_sizes: Sequence[int] = (32,),
                 potential_hid_sizes: Sequence[int] = (32, 32), use_state: bool = True, use_action: bool = True,
                 use_next_state: bool = False, use_done: bool = False, discount_factor: float = 0.99, **kwargs):
        base = BasicRewardNet(observation_space, action_space, use_state=use_state, use_action=use_action,
                              use_next_state=use_next_state, use_done=use_done, hid_sizes=reward_hid_sizes, **kwargs)
        pot = BasicPotentialMLP(observation_space, hid_sizes=potential_hid_sizes, **kwargs)
        super().__init__(base, pot, discount_factor=discount_factor)


class RewardNetWrapper(RewardNet):
    """`rewards/reward_nets.py:227-272`."""

    def __init__(self, base: RewardNet):
        super().__init__(base.observation_space, base.action_space, base.normalize_images)
        self._base = base

    @property
    def base(self) -> RewardNet:
        return self._base

    @property
    def device(self) -> th.device:
        return self.base.device

    @property
    def dtype(self) -> th.dtype:
        return self.base.dtype

    def preprocess(self, state, action, next_state, done):
        return self.base.preprocess(state, action, next_state, done)


class PredictProcessedWrapper(RewardNetWrapper):
    """`rewards/reward_nets.py:303-353`: `forward` / `predict` / `predict_th` pass through to the base."""

    def forward(self, state, action, next_state, done):
        return self.base.forward(state, action, next_state, done)

    def predict(self, state, action, next_state, done):
        return self.base.predict(state, action, next_state, done)

    def predict_th(self, state, action, next_state, done):
        return self.base.predict_th(state, action, next_state, done)


class NormalizedRewardNet(PredictProcessedWrapper):
    """`rewards/reward_nets.py:613-671`: `predict_processed` normalised by a running norm of the raw rewards
    (statistics of the calls so far; updated AFTER normalising when `update_stats`)."""

    def __init__(self, base: RewardNet, normalize_output_layer: Type[nn.Module]):
        super().__init__(base=base)
        self.normalize_output_layer = normalize_output_layer(1)

    def predict_processed(self, state, action, next_state, done, update_stats: bool = True, **kwargs) -> np.ndarray:
        was = self.training
        self.eval()
        try:
            with th.no_grad():
                raw = th.as_tensor(self.base.predict_processed(state, action, next_state, done, **kwargs),
                                   device=self.device)
                rew = self.normalize_output_layer(raw.reshape(-1, 1)).reshape(-1).cpu().numpy().flatten()
            if update_stats:
                with th.no_grad():
                    self.normalize_output_layer.update_stats(raw.reshape(-1, 1))
        finally:
            self.train(was)
        assert rew.shape == np.shape(state)[:1]
        return rew


class RewardNetFromDiscriminatorLogit(RewardNet):
    """`adversarial/gail.py:14-83`: generator reward `-logsigmoid(-logit)` of a discriminator-logit net."""

    def __init__(self, base: RewardNet):
        super().__init__(base.observation_space, base.action_space, base.normalize_images)
        self.base = base

    def forward(self, state, action, next_state, done):
        logits = self.base.forward(state, action, next_state, done)
        return -th.nn.functional.logsigmoid(-logits)


def bulk_relabel_ok(net: RewardNet) -> bool:
    """True when `net.predict_processed(s, a, s', d)` is, row by row, `net.predict_th` of the same rows no matter how
    the rows are split into calls -- i.e. every `predict_processed` / `predict` / `predict_th` along the wrapper chain
    is one of THIS module's stateless implementations. A user net or wrapper that overrides any of them (custom
    normalisation, clipping, counters) and `NormalizedRewardNet` (statistics updated per call) must be called once per
    environment step like the reference does (`rewards/reward_wrapper.py:110-115`)."""
    m = net
    while True:
        if isinstance(m, NormalizedRewardNet) or not isinstance(m, RewardNet):
            return False
        t = type(m)
        own = PredictProcessedWrapper if isinstance(m, PredictProcessedWrapper) else RewardNet
        if t.predict_processed is not RewardNet.predict_processed:
            return False
        if t.predict is not own.predict or t.predict_th is not own.predict_th:
            return False
        if not isinstance(m, PredictProcessedWrapper):
            return True
        m = m.base


def build_mlp(*args, **kwargs) -> Mlp:
    """`util/networks.py:204-283` signature."""
    return Mlp(*args, **kwargs)
