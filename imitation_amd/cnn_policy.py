"""Actor-critic policy for image observations behind SB3's `ActorCriticCnnPolicy` surface
([SB3 policies.ActorCriticCnnPolicy] + [SB3 torch_layers.NatureCNN]; SURVEY 8f row 4, BASELINE config 4):
NatureCNN features (Conv 8/4 - ReLU - Conv 4/2 - ReLU - Conv 3/1 - ReLU - Flatten - Linear 512 - ReLU), no
further hidden layers, a Categorical action head and a value head.

Device layout: ONE flat fp32 parameter buffer in torch `parameters()` order (cnn.0 w,b, cnn.2 w,b, cnn.4 w,b,
linear.0 w,b, action_net w,b, value_net w,b). A weight matrix [Cout, K] is the GEMM's B operand as stored;
activations are channel-last, so on the device cnn.2 / cnn.4 are kept as [Cout, KH, KW, Cin] and the columns
of `linear.0.weight` in (h, w, c) order -- both sides of every im2col / col2im copy are then contiguous runs --
and permuted to torch's layouts in `state_dict()` / `load_state_dict()`; Adam and the L2 term are
element-wise, so nothing else notices. A convolution is `ia_im2col_*` + `ia_gemm_f32`; its
weight gradient the split-K TN GEMM on the kept column buffer; its input gradient an NN GEMM + `ia_col2im_nhwc`.

Scope of this first version: what the BC step needs (`evaluate_actions`, the loss gradient, Adam) plus a
plain `predict`; Discrete action spaces (Atari). Column buffers are explicit (sized for 288 GB of HBM).
"""
from __future__ import annotations

import ctypes as C
import functools
import math
from typing import Dict, Iterator, List, Optional, Tuple

import numpy as np
import torch as th
from torch import nn

from imitation_amd import _lib as L
from imitation_amd import spaces
from imitation_amd.networks import require_device

_CONVS = ((32, 8, 4), (64, 4, 2), (64, 3, 1))   # (out channels, kernel, stride) of NatureCNN


def _is_image_space(space) -> bool:
    return (isinstance(space, spaces.Box) and len(space.shape) == 3 and space.dtype == np.uint8
            and bool(np.all(space.low == 0) and np.all(space.high == 255)))


class NatureCNN:
    """Marker / shape holder (`features_dim`), mirroring the extractor class users pass around."""

    def __init__(self, observation_space, features_dim: int = 512):
        self.features_dim = features_dim
        self.normalize = None


class ActorCriticCnnPolicy:
    def __init__(self, observation_space, action_space, lr_schedule, net_arch=None, activation_fn=nn.Tanh,
                 ortho_init: bool = True, features_extractor_class=NatureCNN, features_extractor_kwargs=None,
                 normalize_images: bool = True, optimizer_class=th.optim.Adam, optimizer_kwargs=None):
        if not _is_image_space(observation_space):
            raise ValueError("ActorCriticCnnPolicy is for uint8 image spaces [C, H, W] with bounds 0 / 255")
        if not isinstance(action_space, spaces.Discrete):
            raise NotImplementedError("the image policy implements the Categorical head (Atari) so far")
        if net_arch not in (None, [], {}):
            raise NotImplementedError("hidden layers behind NatureCNN are not implemented (SB3's default is none)")
        if features_extractor_class is not NatureCNN or not normalize_images:
            raise NotImplementedError("only NatureCNN on [0, 255] frames is implemented")
        self.observation_space, self.action_space = observation_space, action_space
        self.features_dim = int((features_extractor_kwargs or {}).get("features_dim", 512))
        self.features_extractor = NatureCNN(observation_space, self.features_dim)
        self.n_actions = int(action_space.n)
        self.training = True
        Cin, H, W = observation_space.shape
        self.geom: List[Tuple[int, int, int, int, int, int, int, int]] = []   # (Cin, H, W, Cout, K, S, OH, OW)
        for cout, k, s in _CONVS:
            oh, ow = (H - k) // s + 1, (W - k) // s + 1
            if oh < 1 or ow < 1:
                raise ValueError(f"image {observation_space.shape} is too small for the NatureCNN stack")
            self.geom.append((Cin, H, W, cout, k, s, oh, ow))
            Cin, H, W = cout, oh, ow
        self.n_flatten = Cin * H * W
        self._last_hw_c = (H, W, Cin)
        # Host construction in SB3's order so that torch's global generator is consumed identically:
        # the three convolutions, the linear layer, action_net, value_net; then orthogonal re-initialisation
        # (features extractor sqrt(2), action_net 0.01, value_net 1).
        C0 = observation_space.shape[0]
        cnn = nn.Sequential(nn.Conv2d(C0, 32, 8, 4), nn.ReLU(), nn.Conv2d(32, 64, 4, 2), nn.ReLU(),
                            nn.Conv2d(64, 64, 3, 1), nn.ReLU(), nn.Flatten())
        linear = nn.Sequential(nn.Linear(self.n_flatten, self.features_dim), nn.ReLU())
        action_net = nn.Linear(self.features_dim, self.n_actions)
        value_net = nn.Linear(self.features_dim, 1)
        if ortho_init:
            def init(m, gain):
                if isinstance(m, (nn.Linear, nn.Conv2d)):
                    nn.init.orthogonal_(m.weight, gain=gain)
                    m.bias.data.fill_(0.0)
            for mod, gain in ((cnn, np.sqrt(2)), (linear, np.sqrt(2)), (action_net, 0.01), (value_net, 1)):
                mod.apply(functools.partial(init, gain=gain))
        mods = [cnn[0], cnn[2], cnn[4], linear[0], action_net, value_net]
        self._names = ["features_extractor.cnn.0", "features_extractor.cnn.2", "features_extractor.cnn.4",
                       "features_extractor.linear.0", "action_net", "value_net"]
        self._shapes = [tuple(m.weight.shape) for m in mods]
        parts: List[th.Tensor] = []
        for i, m in enumerate(mods):
            w = m.weight.detach()
            parts += [self._to_device_layout(i, w).reshape(-1), m.bias.detach().reshape(-1)]
        self._flat = th.cat(parts).contiguous()
        self._offsets: List[Tuple[int, int, int, int]] = []   # (w offset, w numel, b offset, b numel)
        o = 0
        for shp in self._shapes:
            nw, nb = int(np.prod(shp)), int(shp[0])
            self._offsets.append((o, nw, o + nw, nb))
            o += nw + nb
        self.device = th.device("cpu")
        self._lr0 = float(lr_schedule(1))
        self._bufs: Dict[int, Dict[str, th.Tensor]] = {}

    # ---- layouts ---------------------------------------------------------------------------------------
    def _to_device_layout(self, i: int, w: th.Tensor) -> th.Tensor:
        if i in (1, 2):  # cnn.2 / cnn.4: [Cout, Cin, KH, KW] -> [Cout, KH, KW, Cin] (column order of the NHWC im2col)
            return w.reshape(self._shapes[i]).permute(0, 2, 3, 1).contiguous()
        if i == 3:  # linear.0: columns (c, h, w) -> (h, w, c)
            Hh, Ww, Cc = self._last_hw_c
            return w.reshape(w.shape[0], Cc, Hh, Ww).permute(0, 2, 3, 1).reshape(w.shape[0], -1).contiguous()
        return w.contiguous()

    def _to_torch_layout(self, i: int, w: th.Tensor) -> th.Tensor:
        if i in (1, 2):
            co, ci, kh, kw = self._shapes[i]
            return w.reshape(co, kh, kw, ci).permute(0, 3, 1, 2).contiguous()
        if i == 3:
            Hh, Ww, Cc = self._last_hw_c
            n_out = self._shapes[3][0]
            return w.reshape(n_out, Hh, Ww, Cc).permute(0, 3, 1, 2).reshape(n_out, -1).contiguous()
        return w.reshape(self._shapes[i])

    def w(self, i: int) -> th.Tensor:
        o, n, _, _ = self._offsets[i]
        return self._flat[o:o + n]

    def b(self, i: int) -> th.Tensor:
        _, _, o, n = self._offsets[i]
        return self._flat[o:o + n]

    def to(self, device):
        self.device = th.device(device)
        self._flat = self._flat.to(self.device).contiguous()
        return self

    def set_training_mode(self, mode: bool) -> None:
        self.training = bool(mode)

    def named_parameters(self) -> Iterator[Tuple[str, th.Tensor]]:
        for i, name in enumerate(self._names):
            yield f"{name}.weight", self._to_torch_layout(i, self.w(i)).reshape(self._shapes[i])
            yield f"{name}.bias", self.b(i)

    def parameters(self) -> Iterator[th.Tensor]:
        for _, p in self.named_parameters():
            yield p

    def state_dict(self) -> Dict[str, th.Tensor]:
        """SB3's keys: the shared extractor appears under three names."""
        sd: Dict[str, th.Tensor] = {}
        named = dict(self.named_parameters())
        for alias in ("features_extractor", "pi_features_extractor", "vf_features_extractor"):
            for k, v in named.items():
                if k.startswith("features_extractor."):
                    sd[alias + k[len("features_extractor"):]] = v
        for k, v in named.items():
            if not k.startswith("features_extractor."):
                sd[k] = v
        return sd

    def load_state_dict(self, sd) -> None:
        for i, name in enumerate(self._names):
            self.w(i).copy_(self._to_device_layout(i, th.as_tensor(sd[f"{name}.weight"]).to(self.device).float()).reshape(-1))
            self.b(i).copy_(th.as_tensor(sd[f"{name}.bias"]).to(self.device).float().reshape(-1))

    # ---- forward / backward ----------------------------------------------------------------------------
    def _buffers(self, B: int) -> Dict[str, th.Tensor]:
        if B not in self._bufs:
            f = lambda *s: th.empty(*s, device=self.device)
            d: Dict[str, th.Tensor] = {}
            for li, (cin, _, _, cout, k, _, oh, ow) in enumerate(self.geom):
                d[f"col{li}"] = f(B * oh * ow, cin * k * k)
                d[f"act{li}"] = f(B * oh * ow, cout)           # channel-last [B, OH, OW, Cout], post-ReLU
            d["feat"], d["logits"], d["values"] = f(B, self.features_dim), f(B, self.n_actions), f(B, 1)
            d["logp"], d["ent"], d["acts"] = f(B), f(B), f(B)
            d["dlogits"], d["dfeat"], d["dflat"] = f(B, self.n_actions), f(B, self.features_dim), f(B, self.n_flatten)
            for li in (1, 2):  # input gradients of conv 2 and 3 (conv 1's input is the image)
                cin, h, w_, _, k, _, oh, ow = self.geom[li]
                d[f"dcol{li}"] = f(B * oh * ow, cin * k * k)
                d[f"dact{li - 1}"] = f(B * h * w_, cin)
            self._bufs = {B: d}    # one batch size at a time (the column buffers are large)
        return self._bufs[B]

    @staticmethod
    def _gemm(mode, A, lda, Bm, ldb, Cm, ldc, M, N, K, bias=None, act=0, P=None, ldp=0, splits=1, dbias=None):
        L.call("ia_gemm_f32", mode, L.ptr(A), lda, L.ptr(Bm), ldb, L.ptr(Cm), ldc, M, N, K, L.ptr(bias), act,
               L.ptr(P), ldp, splits, L.ptr(dbias), L.stream())

    def _obs_u8(self, obs) -> th.Tensor:
        t = obs if isinstance(obs, th.Tensor) else th.as_tensor(np.ascontiguousarray(obs))
        if t.dtype != th.uint8:
            raise TypeError("image observations must be uint8 (the policy applies [SB3 preprocess_obs]: x / 255)")
        return t.to(self.device).reshape(-1, *self.observation_space.shape).contiguous()

    def _forward(self, obs_u8: th.Tensor) -> Dict[str, th.Tensor]:
        require_device(self.device)
        B = obs_u8.shape[0]
        d = self._buffers(B)
        C0, H0, W0 = self.observation_space.shape
        _, _, _, _, k, s, _, _ = self.geom[0]
        L.call("ia_im2col_u8_nchw", L.ptr(obs_u8), B, C0, H0, W0, k, k, s, 1.0 / 255.0, L.ptr(d["col0"]), L.stream())
        for li, (cin, h, w_, cout, k, s, oh, ow) in enumerate(self.geom):
            if li > 0:
                L.call("ia_im2col_f32_nhwc", L.ptr(d[f"act{li - 1}"]), B, h, w_, cin, k, k, s, L.ptr(d[f"col{li}"]), L.stream())
            K = cin * k * k
            self._gemm(0, d[f"col{li}"], K, self.w(li), K, d[f"act{li}"], cout, B * oh * ow, cout, K, bias=self.b(li), act=1)
        flat = d["act2"].view(B, self.n_flatten)     # (h, w, c) order: linear.0's columns are stored to match
        self._gemm(0, flat, self.n_flatten, self.w(3), self.n_flatten, d["feat"], self.features_dim, B, self.features_dim,
                   self.n_flatten, bias=self.b(3), act=1)
        F_ = self.features_dim
        self._gemm(0, d["feat"], F_, self.w(4), F_, d["logits"], self.n_actions, B, self.n_actions, F_, bias=self.b(4))
        self._gemm(0, d["feat"], F_, self.w(5), F_, d["values"], 1, B, 1, F_, bias=self.b(5))
        return d

    def evaluate_actions(self, obs, actions, logp_coef: float = 0.0, ent_coef: float = 0.0, want_grad: bool = False):
        """[SB3 evaluate_actions] -> (values [B,1], log_prob [B], entropy [B]). With `want_grad`, the gradient of
        `logp_coef * sum(log_prob) + ent_coef * sum(entropy)` w.r.t. the logits is left for `backward()`."""
        obs_u8 = self._obs_u8(obs)
        d = self._forward(obs_u8)
        B = obs_u8.shape[0]
        a = actions if isinstance(actions, th.Tensor) else th.as_tensor(np.ascontiguousarray(actions))
        d["acts"].copy_(a.to(self.device).reshape(B).float())
        L.call("ia_categorical_loss", L.ptr(d["logits"]), self.n_actions, L.ptr(d["acts"]), B, self.n_actions,
               float(logp_coef), float(ent_coef), L.ptr(d["logp"]), L.ptr(d["ent"]),
               L.ptr(d["dlogits"]) if want_grad else None, L.stream())
        return d["values"], d["logp"], d["ent"]

    def _wgrad(self, li: int, dout: th.Tensor, rows: int, n_out: int, inp: th.Tensor, K: int, grad: th.Tensor) -> None:
        """grad[w_li] += dout^T . inp ; grad[b_li] += column sums of dout   (split-K TN GEMM + ordered reduction)."""
        splits = int(min(64, max(1, rows // 2048)))
        part = th.empty(splits, n_out, K, device=self.device)
        db = th.empty(splits, n_out, device=self.device)
        self._gemm(2, dout, n_out, inp, K, part, K, n_out, K, rows, splits=splits, dbias=db)
        ow_, nw, ob_, nb = self._offsets[li]
        L.call("ia_reduce_partials", L.ptr(part), splits, nw, 1.0, 1, L.ptr(grad[ow_:ow_ + nw]), L.stream())
        L.call("ia_reduce_partials", L.ptr(db), splits, nb, 1.0, 1, L.ptr(grad[ob_:ob_ + nb]), L.stream())

    def backward(self, B: int, grad: th.Tensor) -> None:
        """Adds to `grad` (flat, device layout) the parameter gradient of the loss whose logit gradient the last
        `evaluate_actions(..., want_grad=True)` on a batch of `B` rows left behind. The value head gets none."""
        d = self._bufs[B]
        F_, A = self.features_dim, self.n_actions
        self._wgrad(4, d["dlogits"], B, A, d["feat"], F_, grad)
        self._gemm(1, d["dlogits"], A, self.w(4), F_, d["dfeat"], F_, B, F_, A, act=1, P=d["feat"], ldp=F_)
        flat = d["act2"].view(B, self.n_flatten)
        self._wgrad(3, d["dfeat"], B, F_, flat, self.n_flatten, grad)
        self._gemm(1, d["dfeat"], F_, self.w(3), self.n_flatten, d["dflat"], self.n_flatten, B, self.n_flatten, F_,
                   act=1, P=flat, ldp=self.n_flatten)
        dout = d["dflat"].view(-1, self.geom[2][3])            # [B*OH3*OW3, 64], already masked by act2 > 0
        for li in (2, 1, 0):
            cin, h, w_, cout, k, s, oh, ow = self.geom[li]
            K, rows = cin * k * k, B * oh * ow
            self._wgrad(li, dout, rows, cout, d[f"col{li}"], K, grad)
            if li == 0:
                break
            self._gemm(1, dout, cout, self.w(li), K, d[f"dcol{li}"], K, rows, K, cout)
            L.call("ia_col2im_nhwc", L.ptr(d[f"dcol{li}"]), B, h, w_, cin, k, k, s, L.ptr(d[f"act{li - 1}"]),
                   L.ptr(d[f"dact{li - 1}"]), L.stream())
            dout = d[f"dact{li - 1}"]

    # ---- acting --------------------------------------------------------------------------------------------
    def predict(self, observation, state=None, episode_start=None, deterministic: bool = False):
        """[SB3 BasePolicy.predict]. The mode is the first argmax; sampling is [SB3 CategoricalDistribution.sample]
        itself on the host (`torch.distributions.Categorical(logits).sample()`: torch.multinomial on the global
        generator, the reference's stream -- as for the MLP policies)."""
        obs = np.asarray(observation)
        vectorized = obs.shape != tuple(self.observation_space.shape)
        d = self._forward(self._obs_u8(obs))
        logits = d["logits"].float().cpu()
        if deterministic:
            acts = logits.numpy().argmax(axis=1)
        else:
            acts = th.distributions.Categorical(logits=logits).sample().numpy()
        acts = acts.astype(np.int64)
        return (acts if vectorized else acts[0]), state
