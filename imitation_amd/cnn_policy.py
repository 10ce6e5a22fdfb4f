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

Scope: the BC step (`evaluate_actions`, the loss gradient; Categorical head) and the PPO generator of GAIL on image
observations -- the same rollout / update protocol as `general_policy.GeneralTowers` (`make_act_step`,
`make_multinomial_step`, `values_rows`, `log_prob_rows`, `ppo_update`): rollout tiles hold the frames as fp32 rows
(0..255, exact), the policy turns them back into uint8 images on the device. Categorical (Discrete) and DiagGaussian
(Box) heads. Column buffers are explicit (sized for 288 GB of HBM).
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
                 ortho_init: bool = True, use_sde: bool = False, log_std_init: float = 0.0,
                 features_extractor_class=NatureCNN, features_extractor_kwargs=None, share_features_extractor: bool = True,
                 normalize_images: bool = True, optimizer_class=th.optim.Adam, optimizer_kwargs=None):
        if not _is_image_space(observation_space):
            raise ValueError("ActorCriticCnnPolicy is for uint8 image spaces [C, H, W] with bounds 0 / 255")
        self.discrete = isinstance(action_space, spaces.Discrete)
        if not self.discrete and not (isinstance(action_space, spaces.Box) and len(action_space.shape) == 1):
            raise NotImplementedError("image policies implement Discrete (Categorical) and 1-D Box (DiagGaussian) heads")
        if net_arch not in (None, [], {}):
            raise NotImplementedError("hidden layers behind NatureCNN are not implemented (SB3's default is none)")
        if features_extractor_class is not NatureCNN or not normalize_images:
            raise NotImplementedError("only NatureCNN on [0, 255] frames is implemented")
        self.observation_space, self.action_space = observation_space, action_space
        self.features_dim = int((features_extractor_kwargs or {}).get("features_dim", 512))
        self.features_extractor = NatureCNN(observation_space, self.features_dim)
        self.act_dim = int(action_space.n) if self.discrete else int(action_space.shape[0])
        self.n_actions = self.act_dim               # width of the action head (logits or Gaussian means)
        self.obs_dim = int(np.prod(observation_space.shape))
        self.fused = False                          # PPO runs this policy's own minibatch loop (`ppo_update`)
        self.discrete_sampling = "multinomial"
        self.training = True
        self.optimizer_kwargs = dict(optimizer_kwargs or {})
        self.optimizer_kwargs.setdefault("eps", 1e-5)   # SB3 ActorCriticPolicy default for Adam
        self.optimizer = None
        self._low = self._high = None
        if optimizer_class is not th.optim.Adam:
            raise NotImplementedError("the PPO step implements Adam (SB3 default)")
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
        # first layer straight from the uint8 frames (implicit GEMM, csrc/conv1_implicit.hip) when the shape is the
        # standard 4-frame stack; otherwise im2col + GEMM like the other layers. `implicit_conv1 = False` forces the
        # explicit path (tests compare the two).
        self.implicit_conv1: Optional[bool] = None   # resolved on the device (`to`)
        # layers 2 and 3: the GEMMs read their [rows, KH*KW*Cin] operand as an implicit im2col view of the channel-last
        # activations (`ia_gemm_f32_im2col`: forward and weight gradient; no column buffer) when Cin % 4 == 0 and
        # (KW*Cin) % 32 == 0 -- true for NatureCNN (4 x 32 and 3 x 64). False: explicit im2col + GEMM (tests).
        self.implicit_convs = all(g[0] % 4 == 0 and (g[4] * g[0]) % 32 == 0 for g in self.geom[1:])
        # input gradients of layers 2 / 3 as implicit transposed convolutions (`_dgrad_implicit`: no `dcol` buffers,
        # 1.1 GB less at batch 4096; on par with the NN GEMM + col2im pair in time: 2.82 against 2.87 ms per BC step).
        # False: the explicit pair (tests compare the two).
        self.implicit_dgrad = True
        self._dgrad_idx: Dict[int, th.Tensor] = {}
        # Host construction in SB3's order so that torch's global generator is consumed identically:
        # the three convolutions, the linear layer, action_net, value_net; then orthogonal re-initialisation
        # (features extractor sqrt(2), action_net 0.01, value_net 1).
        C0 = observation_space.shape[0]
        cnn = nn.Sequential(nn.Conv2d(C0, 32, 8, 4), nn.ReLU(), nn.Conv2d(32, 64, 4, 2), nn.ReLU(),
                            nn.Conv2d(64, 64, 3, 1), nn.ReLU(), nn.Flatten())
        linear = nn.Sequential(nn.Linear(self.n_flatten, self.features_dim), nn.ReLU())
        action_net = nn.Linear(self.features_dim, self.n_actions)
        log_std = None if self.discrete else th.ones(self.act_dim) * log_std_init
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
        # torch `parameters()` order: the policy's own parameter (log_std, Box heads) first, then the child modules
        parts: List[th.Tensor] = [] if self.discrete else [log_std]
        for i, m in enumerate(mods):
            w = m.weight.detach()
            parts += [self._to_device_layout(i, w).reshape(-1), m.bias.detach().reshape(-1)]
        self._flat = th.cat(parts).contiguous()
        self._offsets: List[Tuple[int, int, int, int]] = []   # (w offset, w numel, b offset, b numel)
        o = 0 if self.discrete else self.act_dim
        for shp in self._shapes:
            nw, nb = int(np.prod(shp)), int(shp[0])
            self._offsets.append((o, nw, o + nw, nb))
            o += nw + nb
        self.device = th.device("cpu")
        self._lr0 = float(lr_schedule(1))
        self._bufs: Dict[int, Dict[str, th.Tensor]] = {}
        self._pending_reduce: list = []   # (slabs, splits, n, destination piece of the flat gradient) queued by `_wgrad`

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
        from imitation_amd.networks import HipAdam
        self.device = th.device(device)
        self._flat = self._flat.to(self.device).contiguous()
        if self.device.type == "cuda":
            if self.implicit_conv1 is None:
                c0, h0, w0 = self.observation_space.shape
                self.implicit_conv1 = bool(L.load().ia_conv1_u8_implicit_ok(c0, h0, w0, 8, 8, 4, 32))
            self.optimizer = HipAdam(self._flat, th.zeros_like(self._flat), lr=self._lr0, **self.optimizer_kwargs)
            if self.discrete:
                self._low = self._high = th.zeros(self.act_dim, device=self.device)
            else:
                self._low = th.as_tensor(self.action_space.low.reshape(-1), dtype=th.float32, device=self.device)
                self._high = th.as_tensor(self.action_space.high.reshape(-1), dtype=th.float32, device=self.device)
        return self

    def set_training_mode(self, mode: bool) -> None:
        self.training = bool(mode)

    def train(self, mode: bool = True):
        self.training = bool(mode)
        return self

    def eval(self):
        return self.train(False)

    def _sync_transposed(self) -> None:
        """(protocol of the MLP policies: nothing is shadowed here)"""

    @property
    def log_std(self) -> Optional[th.Tensor]:
        return None if self.discrete else self._flat[: self.act_dim]

    @property
    def samples_on_host(self) -> bool:
        """Discrete heads sample with torch.multinomial on the host (the reference's stream)."""
        return self.discrete

    @property
    def squash_output(self) -> bool:
        return False

    def named_parameters(self) -> Iterator[Tuple[str, th.Tensor]]:
        if not self.discrete:
            yield "log_std", self._flat[: self.act_dim]
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
        if not self.discrete:
            sd["log_std"] = named.pop("log_std")
        for alias in ("features_extractor", "pi_features_extractor", "vf_features_extractor"):
            for k, v in named.items():
                if k.startswith("features_extractor."):
                    sd[alias + k[len("features_extractor"):]] = v
        for k, v in named.items():
            if not k.startswith("features_extractor."):
                sd[k] = v
        return sd

    def load_state_dict(self, sd) -> None:
        if not self.discrete:
            self._flat[: self.act_dim].copy_(th.as_tensor(sd["log_std"]).to(self.device).float().reshape(-1))
        for i, name in enumerate(self._names):
            self.w(i).copy_(self._to_device_layout(i, th.as_tensor(sd[f"{name}.weight"]).to(self.device).float()).reshape(-1))
            self.b(i).copy_(th.as_tensor(sd[f"{name}.bias"]).to(self.device).float().reshape(-1))

    # ---- forward / backward ----------------------------------------------------------------------------
    def _buffers(self, B: int) -> Dict[str, th.Tensor]:
        if B not in self._bufs:
            f = lambda *s: th.empty(*s, device=self.device)
            d: Dict[str, th.Tensor] = {}
            for li, (cin, _, _, cout, k, _, oh, ow) in enumerate(self.geom):
                if li == 0 and self.implicit_conv1:   # no column buffer: workspaces of the implicit first layer
                    d["c1_ws"] = f(128 * 64)
                    d["c1_wg"] = f(int(L.load().ia_conv1_u8_wgrad_ws_floats(B)))
                elif li > 0 and self.implicit_convs:
                    pass                               # (the GEMMs read the activations through the im2col view)
                else:
                    d[f"col{li}"] = f(B * oh * ow, cin * k * k)
                d[f"act{li}"] = f(B * oh * ow, cout)           # channel-last [B, OH, OW, Cout], post-ReLU
            d["feat"], d["logits"], d["values"] = f(B, self.features_dim), f(B, self.n_actions), f(B, 1)
            d["logp"], d["ent"], d["acts"] = f(B), f(B), f(B, 1 if self.discrete else self.act_dim)
            d["dlogits"], d["dfeat"], d["dflat"] = f(B, self.n_actions), f(B, self.features_dim), f(B, self.n_flatten)
            for li in (1, 2):  # input gradients of conv 2 and 3 (conv 1's input is the image)
                cin, h, w_, _, k, _, oh, ow = self.geom[li]
                d[f"dact{li - 1}"] = f(B * h * w_, cin)      # (`dcol{li}` only exists on the explicit path: `_dcol`)
            d["dvalues"], d["dhead"] = f(B, 1), f(2, B, self.features_dim)
            if len(self._bufs) >= 3:   # rollout step, bootstrap chunk, minibatch: more sizes evict the oldest
                self._bufs.pop(next(iter(self._bufs)))
            self._bufs[B] = d
        return self._bufs[B]

    @staticmethod
    def _gemm(mode, A, lda, Bm, ldb, Cm, ldc, M, N, K, bias=None, act=0, P=None, ldp=0, splits=1, dbias=None):
        L.call("ia_gemm_f32", mode, L.ptr(A), lda, L.ptr(Bm), ldb, L.ptr(Cm), ldc, M, N, K, L.ptr(bias), act,
               L.ptr(P), ldp, splits, L.ptr(dbias), L.stream())

    takes_uint8_frames = True   # (`PPO`: the rollout tile's host rows and their transfer are uint8 for this policy)

    # rows of a weight gradient's contraction per K split (<= 64 splits): a convolution's [64 x 512] gradient is 8 output tiles
    # -- at 2 048 rows per split a 256-frame minibatch (20 736 rows) ran on 80 workgroups of 64 K chunks each
    WGRAD_ROWS_PER_SPLIT = 512
    REDUCE_IN_ONE_LAUNCH = True
    ACT_ZERO_COPY = True   # (False: the rollout step's frames go through a device copy first -- same-box A/Bs)
    LINEAR_SPLIT_K = True   # (False: the feature layer's product unsplit at every batch size -- same-box A/Bs)

    def _linear_splits(self, B: int) -> int:
        """K splits of the `linear` layer's forward product at batch `B`: 1 once its 64 x 64 output tiles alone fill the
        device (BC's batches), else as many slabs of >= 8 K chunks as bring the launch to ~256 workgroups."""
        if not self.LINEAR_SPLIT_K or self.features_dim % 4:
            return 1
        tiles = -(-B // 64) * -(-self.features_dim // 64)
        if tiles >= 128:
            return 1
        chunks = self.n_flatten // 32
        return int(max(1, min(chunks // 8, 256 // tiles)))

    def _obs_u8(self, obs) -> th.Tensor:
        t = obs if isinstance(obs, th.Tensor) else th.as_tensor(np.ascontiguousarray(obs))
        if t.dtype != th.uint8:
            raise TypeError("image observations must be uint8 (the policy applies [SB3 preprocess_obs]: x / 255)")
        return t.to(self.device).reshape(-1, *self.observation_space.shape).contiguous()

    def _forward(self, obs_u8: th.Tensor, values_out: Optional[th.Tensor] = None) -> Dict[str, th.Tensor]:
        """`values_out` (contiguous [B] device row): the value head writes there instead of into the batch buffer."""
        require_device(self.device)
        B = obs_u8.shape[0]
        d = self._buffers(B)
        C0, H0, W0 = self.observation_space.shape
        _, _, _, _, k, s, _, _ = self.geom[0]
        d["obs_u8"] = obs_u8   # (the implicit weight gradient reads the frames again)
        if self.implicit_conv1:
            L.call("ia_conv1_u8_forward", L.ptr(obs_u8), B, H0, W0, L.ptr(self.w(0)), L.ptr(self.b(0)), 1.0 / 255.0,
                   L.ptr(d["c1_ws"]), L.ptr(d["act0"]), L.stream())
        else:
            L.call("ia_im2col_u8_nchw", L.ptr(obs_u8), B, C0, H0, W0, k, k, s, 1.0 / 255.0, L.ptr(d["col0"]), L.stream())
        for li, (cin, h, w_, cout, k, s, oh, ow) in enumerate(self.geom):
            if li == 0 and self.implicit_conv1:
                continue
            K = cin * k * k
            if li > 0 and self.implicit_convs:
                L.call("ia_gemm_f32_im2col", 0, L.ptr(d[f"act{li - 1}"]), K, L.ptr(self.w(li)), K, L.ptr(d[f"act{li}"]), cout,
                       B * oh * ow, cout, K, L.ptr(self.b(li)), 1, 1, None, h, w_, cin, k, k, s, L.stream())
                continue
            if li > 0:
                L.call("ia_im2col_f32_nhwc", L.ptr(d[f"act{li - 1}"]), B, h, w_, cin, k, k, s, L.ptr(d[f"col{li}"]), L.stream())
            self._gemm(0, d[f"col{li}"], K, self.w(li), K, d[f"act{li}"], cout, B * oh * ow, cout, K, bias=self.b(li), act=1)
        flat = d["act2"].view(B, self.n_flatten)     # (h, w, c) order: linear.0's columns are stored to match
        sk = self._linear_splits(B)
        if sk > 1:
            # few output tiles, long K (rollout steps and PPO minibatches: 8 - 32 tiles of 64 x 64 with 98 K chunks each on as
            # many of the 256 compute units): the product split along K, bias + ReLU behind the ordered sum of the slabs
            if "feat_parts" not in d:
                d["feat_parts"] = th.empty(sk, B, self.features_dim, device=self.device)
            L.call("ia_gemm_f32_nt_splitk", L.ptr(flat), self.n_flatten, L.ptr(self.w(3)), self.n_flatten,
                   L.ptr(d["feat_parts"]), L.ptr(d["feat"]), self.features_dim, B, self.features_dim, self.n_flatten,
                   L.ptr(self.b(3)), 1, sk, L.stream())
        else:
            self._gemm(0, flat, self.n_flatten, self.w(3), self.n_flatten, d["feat"], self.features_dim, B,
                       self.features_dim, self.n_flatten, bias=self.b(3), act=1)
        F_ = self.features_dim
        self._gemm(0, d["feat"], F_, self.w(4), F_, d["logits"], self.n_actions, B, self.n_actions, F_, bias=self.b(4))
        self._gemm(0, d["feat"], F_, self.w(5), F_, d["values"] if values_out is None else values_out, 1, B, 1, F_,
                   bias=self.b(5))
        return d

    def evaluate_actions(self, obs, actions, logp_coef: float = 0.0, ent_coef: float = 0.0, want_grad: bool = False):
        """[SB3 evaluate_actions] -> (values [B,1], log_prob [B], entropy [B]). With `want_grad`, the gradient of
        `logp_coef * sum(log_prob) + ent_coef * sum(entropy)` w.r.t. the logits is left for `backward()`."""
        obs_u8 = self._obs_u8(obs)
        d = self._forward(obs_u8)
        B = obs_u8.shape[0]
        a = actions if isinstance(actions, th.Tensor) else th.as_tensor(np.ascontiguousarray(actions))
        d["acts"].copy_(a.to(self.device).reshape(d["acts"].shape).float())
        if not self.discrete:
            L.call("ia_gauss_eval", L.ptr(d["logits"]), L.ptr(self._flat), L.ptr(d["acts"]), B, self.act_dim,
                   L.ptr(d["logp"]), L.ptr(d["ent"]), L.stream())
            if want_grad:   # Gaussian head: the head-loss kernel forms the gradient (see `head_grad_of_coefficients`)
                from imitation_amd.general_policy import head_grad_of_coefficients
                hw = d.setdefault("bc_ws", {})
                head_grad_of_coefficients(False, d["logits"], L.ptr(self._flat), d["values"], d["acts"], d["logp"], B,
                                          self.act_dim, logp_coef, ent_coef, hw)
                d["dlogits"].copy_(hw["d_out"])
                d["dls_pending"] = hw["dls"]
            return d["values"], d["logp"], d["ent"]
        L.call("ia_categorical_loss", L.ptr(d["logits"]), self.n_actions, L.ptr(d["acts"]), B, self.n_actions,
               float(logp_coef), float(ent_coef), L.ptr(d["logp"]), L.ptr(d["ent"]),
               L.ptr(d["dlogits"]) if want_grad else None, L.stream())
        return d["values"], d["logp"], d["ent"]

    def _wgrad(self, li: int, dout: th.Tensor, rows: int, n_out: int, inp: th.Tensor, K: int, grad: th.Tensor,
               im=None) -> None:
        """grad[w_li] += dout^T . inp ; grad[b_li] += column sums of dout   (split-K TN GEMM + ordered reduction).
        `im = (H, W, Cin, k, stride)`: `inp` is the layer's channel-last INPUT and the [rows, K] operand its implicit
        im2col view."""
        splits = int(min(64, max(1, rows // self.WGRAD_ROWS_PER_SPLIT)))
        part = th.empty(splits, n_out, K, device=self.device)
        db = th.empty(splits, n_out, device=self.device)
        if im is not None:
            h, w_, cin, k, s = im
            L.call("ia_gemm_f32_im2col", 2, L.ptr(dout), n_out, L.ptr(inp), K, L.ptr(part), K, n_out, K, rows, None, 0,
                   splits, L.ptr(db), h, w_, cin, k, k, s, L.stream())
        else:
            self._gemm(2, dout, n_out, inp, K, part, K, n_out, K, rows, splits=splits, dbias=db)
        ow_, nw, ob_, nb = self._offsets[li]
        # (the slabs' ordered sums are added to the flat gradient by ONE launch at the end of `backward`: `_flush_reduces`)
        self._pending_reduce.append((part, splits, nw, grad[ow_:ow_ + nw]))
        self._pending_reduce.append((db, splits, nb, grad[ob_:ob_ + nb]))

    def _flush_reduces(self) -> None:
        """`grad[piece] += ordered sum of the piece's slabs` for every weight / bias piece `_wgrad` queued: one launch per <= 16
        pieces (a launch per piece before: twelve per optimiser step of ~6 us each at minibatch sizes)."""
        pend, self._pending_reduce = self._pending_reduce, []
        if not self.REDUCE_IN_ONE_LAUNCH:   # (same-box A/Bs: a launch per piece, the same sums)
            for part, splits, n, dst in pend:
                L.call("ia_reduce_partials", L.ptr(part), splits, n, 1.0, 1, L.ptr(dst), L.stream())
            return
        for lo in range(0, len(pend), 16):
            seg = pend[lo:lo + 16]
            k = len(seg)
            L.call("ia_reduce_partials_multi", k, (C.c_void_p * k)(*(t[0].data_ptr() for t in seg)),
                   (C.c_int * k)(*(t[1] for t in seg)), (C.c_int64 * k)(*(t[2] for t in seg)),
                   (C.c_void_p * k)(*(t[3].data_ptr() for t in seg)), L.stream())

    def _dcol(self, d, li: int, rows: int, K: int) -> th.Tensor:
        """Column-gradient buffer of the explicit input-gradient path (allocated on first use)."""
        if f"dcol{li}" not in d:
            d[f"dcol{li}"] = th.empty(rows, K, device=self.device)
        return d[f"dcol{li}"]

    def _dgrad_implicit(self, li: int, dout: th.Tensor, B: int, d) -> None:
        """d loss / d input of convolution `li` (then the ReLU mask of the layer below) as implicit GEMMs over `dout`
        [B, OH, OW, Cout]: dX[y, x, c] = sum_{i,j,co} dout[(y-i)/S, (x-j)/S, co] W[co, i, j, c] over the taps with
        S | y-i, S | x-j. For the input pixels of one sub-pixel class (py, px) = (y % S, x % S) that is a stride-1
        convolution of `dout`, zero-padded by k/S - 1, with the k/S x k/S kernel Wd[c, ti, tj, co] =
        W[co, py + S (k/S-1-ti), px + S (k/S-1-tj), c]; `ia_gemm_f32_im2col_pad` reads its operand through that padded
        view and scatters the class's rows into the input grid. No `dcol` buffer, no col2im."""
        cin, h, w_, cout, k, s, oh, ow = self.geom[li]
        kt = k // s
        dact = d[f"dact{li - 1}"]
        Kd = kt * kt * cout
        if li not in self._dgrad_idx:   # gather index of all classes' rearranged weights (geometry only): one launch per step
            base = th.arange(cout * k * k * cin).view(cout, k, k, cin)        # device layout [Cout, KH, KW, Cin]
            blocks = []
            for py in range(s):
                for px in range(s):
                    ii = th.as_tensor([py + s * (kt - 1 - t) for t in range(kt)])
                    jj = th.as_tensor([px + s * (kt - 1 - t) for t in range(kt)])
                    blocks.append(base.index_select(1, ii).index_select(2, jj).permute(3, 1, 2, 0).reshape(-1))
            self._dgrad_idx[li] = th.cat(blocks).to(self.device)
        # every class reads the SAME padded view of `dout` (only the weights differ): one GEMM with the classes side by
        # side along the columns, rows scattered by (class, pixel) in the epilogue
        Wd_all = self.w(li).index_select(0, self._dgrad_idx[li])              # [s*s*cin, Kd], class-major rows
        gh, gw = oh + kt - 1, ow + kt - 1                                     # input pixels of one class per image
        cmap = (C.c_int * 5)(s, -1, -1, h, w_) if s > 1 else None
        if s * gh < h or s * gw < w_:
            # input pixels no window reaches ((h - k) % s or (w - k) % s left over: 9 columns under a 4 x 4 stride-2 kernel) belong
            # to no class: their gradient is zero, and nothing below writes it (the buffer is `th.empty`: the first-layer weight
            # gradient then summed whatever the allocator handed out -- found by an order-dependent failure of
            # `test_cnn_policy_forward_and_gradient_match_torch[shape2-33-18]`)
            dact.zero_()
        L.call("ia_gemm_f32_im2col_pad", 0, L.ptr(dout), Kd, L.ptr(Wd_all), Kd, L.ptr(dact), cin, B * gh * gw, s * s * cin, Kd,
               None, 0, 1, None, oh, ow, cout, kt, kt, 1, kt - 1, cmap, L.ptr(d[f"act{li - 1}"]), L.stream())

    def backward(self, B: int, grad: th.Tensor, with_values: bool = False) -> None:
        """Adds to `grad` (flat, device layout) the parameter gradient of the loss whose head gradients were left in
        the batch-`B` buffers: `dlogits` (by `evaluate_actions(..., want_grad=True)` or the PPO head loss) and, with
        `with_values`, `dvalues` (PPO's value loss; the BC loss has no value term)."""
        d = self._bufs[B]
        F_, A = self.features_dim, self.n_actions
        if d.get("dls_pending") is not None:   # log_std gradient of a Box-head BC step
            L.call("ia_reduce_partials", L.ptr(d["dls_pending"]), 1, A, 1.0, 1, L.ptr(grad), L.stream())
            d["dls_pending"] = None
        self._wgrad(4, d["dlogits"], B, A, d["feat"], F_, grad)
        if with_values:
            self._wgrad(5, d["dvalues"], B, 1, d["feat"], F_, grad)
            # d feat = relu'(feat) * (dlogits . Wa + dvalues . Wv): two NN GEMMs into one [2][B, F] slab pair,
            # summed in slab order, then the ReLU mask
            self._gemm(1, d["dlogits"], A, self.w(4), F_, d["dhead"][0], F_, B, F_, A)
            self._gemm(1, d["dvalues"], 1, self.w(5), F_, d["dhead"][1], F_, B, F_, 1)
            L.call("ia_reduce_partials", L.ptr(d["dhead"]), 2, B * F_, 1.0, 0, L.ptr(d["dhead"][0]), L.stream())
            L.call("ia_relu_backward", L.ptr(d["dhead"][0]), L.ptr(d["feat"]), B * F_, L.ptr(d["dfeat"]), L.stream())
        else:
            self._gemm(1, d["dlogits"], A, self.w(4), F_, d["dfeat"], F_, B, F_, A, act=1, P=d["feat"], ldp=F_)
        flat = d["act2"].view(B, self.n_flatten)
        self._wgrad(3, d["dfeat"], B, F_, flat, self.n_flatten, grad)
        self._gemm(1, d["dfeat"], F_, self.w(3), self.n_flatten, d["dflat"], self.n_flatten, B, self.n_flatten, F_,
                   act=1, P=flat, ldp=self.n_flatten)
        dout = d["dflat"].view(-1, self.geom[2][3])            # [B*OH3*OW3, 64], already masked by act2 > 0
        for li in (2, 1, 0):
            cin, h, w_, cout, k, s, oh, ow = self.geom[li]
            K, rows = cin * k * k, B * oh * ow
            if li == 0 and self.implicit_conv1:
                ow_, nw, ob_, nb = self._offsets[0]
                L.call("ia_conv1_u8_wgrad", L.ptr(d["obs_u8"]), B, h, w_, L.ptr(dout), 1.0 / 255.0, L.ptr(d["c1_wg"]), 1,
                       L.ptr(grad[ow_:ow_ + nw]), L.ptr(grad[ob_:ob_ + nb]), L.stream())
                break
            if li > 0 and self.implicit_convs:
                self._wgrad(li, dout, rows, cout, d[f"act{li - 1}"], K, grad, im=(h, w_, cin, k, s))
            else:
                self._wgrad(li, dout, rows, cout, d[f"col{li}"], K, grad)
            if li == 0:
                break
            if self.implicit_dgrad and k % s == 0 and ((k // s) * cout) % 32 == 0 and cout % 4 == 0:
                self._dgrad_implicit(li, dout, B, d)
            else:
                self._gemm(1, dout, cout, self.w(li), K, self._dcol(d, li, rows, K), K, rows, K, cout)
                L.call("ia_col2im_nhwc", L.ptr(d[f"dcol{li}"]), B, h, w_, cin, k, k, s, L.ptr(d[f"act{li - 1}"]),
                       L.ptr(d[f"dact{li - 1}"]), L.stream())
            dout = d[f"dact{li - 1}"]
        self._flush_reduces()

    # ---- PPO generator protocol (the surface `ppo.PPO` and the adversarial trainer drive; same as
    # `general_policy.GeneralTowers`) ---------------------------------------------------------------------------
    def _rows_u8(self, rows: th.Tensor) -> th.Tensor:
        """fp32 frame rows `[m, C*H*W]` of a rollout tile / replay table (values 0..255, exact) -> uint8 images."""
        return rows.reshape(-1, *self.observation_space.shape).to(th.uint8)

    def sample_noise(self, n: int) -> th.Tensor:
        return th.distributions.utils._standard_normal((n, self.act_dim), dtype=th.float32, device="cpu")

    def draw_noise_into(self, out: th.Tensor) -> None:
        out.normal_()

    def make_act_step(self, obs_tile: th.Tensor, noise_host: th.Tensor, acts: th.Tensor, clipped: th.Tensor,
                      val: th.Tensor, logp: th.Tensor):
        """Rollout-step launcher for Box heads (see `GeneralTowers.make_act_step`)."""
        assert not self.discrete
        n, A = obs_tile.shape[1], self.act_dim
        dev = self.device
        u8_tile = obs_tile.dtype == th.uint8   # (`RolloutBuffer(obs_u8=True)`: the frames as the environment hands them over)
        obs_d = th.empty(n, self.obs_dim, dtype=obs_tile.dtype, device=dev)
        noise_d, clip_d = th.empty(n, A, device=dev), th.empty(n, A, device=dev)
        stream_obj = th.cuda.current_stream()

        def step(t: int) -> None:
            with th.cuda.stream(stream_obj):
                obs_d.copy_(obs_tile[t], non_blocking=True)
                noise_d.copy_((noise_host[t] if noise_host.dim() == 3 else noise_host).reshape(n, A), non_blocking=True)
                d = self._forward(obs_d.view(-1, *self.observation_space.shape) if u8_tile else self._rows_u8(obs_d))
                L.call("ia_gauss_act", L.ptr(d["logits"]), L.ptr(self._flat), L.ptr(noise_d), L.ptr(self._low),
                       L.ptr(self._high), n, A, L.ptr(acts[t]), L.ptr(clip_d), L.ptr(logp[t]), L.stream())
                val[t].copy_(d["values"].reshape(n))
                clipped[t].copy_(clip_d, non_blocking=True)

        return step

    def make_multinomial_step(self, obs_tile: th.Tensor, h_logits: th.Tensor, h_clip: th.Tensor, val: th.Tensor,
                              h_logp: th.Tensor):
        """Rollout-step launcher for Discrete heads on the reference's sampling stream: logits to the pinned host
        tile, then `torch.distributions.Categorical(logits).sample()` / `.log_prob()` as SB3 composes them."""
        assert self.discrete
        launch, finish = self._multinomial_launch_finish(obs_tile, h_logits, h_clip, val, h_logp)

        def step(t: int) -> None:
            launch(t)
            finish(t)

        return step

    def make_multinomial_mailbox(self, obs_tile: th.Tensor, h_logits: th.Tensor, h_clip: th.Tensor, val: th.Tensor,
                                 h_logp: th.Tensor, T: int, timeout_s: float = 120.0):
        """`(post, wait, close)` as `ActorCriticPolicy.make_multinomial_mailbox`, with the step's launches as `post(t)` and
        the wait for them + the host-side sampling as `wait(t)`: the rollout loop posts step t + 1 as soon as its frames are in
        place and does step t's bookkeeping (the fp32 copy of the frames into the rollout tile, replay rows, callbacks) while
        the device runs the ~8 launches of NatureCNN -- which queue for workgroup slots beside the discriminator's kernels
        (`profiles/r05_image_gail.md`: 0.25 -> 1.5 ms per step, all of it host-visible before)."""
        assert self.discrete
        launch, finish = self._multinomial_launch_finish(obs_tile, h_logits, h_clip, val, h_logp)

        def wait(t: int) -> bool:
            finish(t)
            return True

        return launch, wait, (lambda: None)

    def _multinomial_launch_finish(self, obs_tile, h_logits, h_clip, val, h_logp):
        n = obs_tile.shape[1]
        # the rollout tile's pinned rows are uint8 for image observations (`RolloutBuffer(obs_u8=True)`: the frames as the
        # environment hands them over -- a quarter of an fp32 row's bytes over PCIe and no conversion launch); an fp32 tile
        # (a caller's own) is converted on the device as before
        u8_tile = obs_tile.dtype == th.uint8
        obs_d = th.empty(n, self.obs_dim, dtype=obs_tile.dtype, device=self.device)
        stream_obj = th.cuda.current_stream()

        zero_copy = self.ACT_ZERO_COPY and u8_tile and obs_tile.is_pinned() and self.implicit_conv1

        def launch(t: int) -> None:
            with th.cuda.stream(stream_obj):
                if zero_copy:
                    # the first convolution reads the step's frames straight from their pinned (device-mapped) row: no copy-engine
                    # hop of 1.8 MB in front of the step's eight launches
                    frames = obs_tile[t].view(-1, *self.observation_space.shape)
                else:
                    obs_d.copy_(obs_tile[t], non_blocking=True)
                    frames = obs_d.view(-1, *self.observation_space.shape) if u8_tile else self._rows_u8(obs_d)
                d = self._forward(frames, values_out=val[t])
                # (the action head writing the pinned tile itself instead of this copy measured ~1 ms per round SLOWER: 4-byte
                #  stores over PCIe from a GEMM epilogue)
                h_logits.copy_(d["logits"], non_blocking=True)

        h_logp_np, h_clip_np, rows_np = h_logp.numpy(), h_clip.numpy().reshape(h_clip.shape[0], n), np.arange(n)

        def finish(t: int) -> None:
            stream_obj.synchronize()
            from imitation_amd.policies import categorical_sample_into   # (local: policies imports this module)
            categorical_sample_into(h_logits, h_logp_np[t], h_clip_np[t], rows_np)

        return launch, finish

    _CHUNK = 512   # rows per forward when a whole tile is evaluated (bootstrap values, log pi of replay rows)

    def values_rows(self, obs_dev: th.Tensor, out: th.Tensor) -> None:
        m = obs_dev.shape[0]
        out = out.reshape(m)
        for lo in range(0, m, self._CHUNK):
            hi = min(m, lo + self._CHUNK)
            d = self._forward(self._rows_u8(obs_dev[lo:hi]))
            out[lo:hi].copy_(d["values"].reshape(hi - lo))

    def predict_values(self, obs) -> th.Tensor:
        d = self._forward(self._obs_u8(obs))
        return d["values"].clone()

    def log_prob_rows(self, obs_dev: th.Tensor, acts_dev: th.Tensor, out: th.Tensor, norm_snapshot=None) -> None:
        m = obs_dev.shape[0]
        a = acts_dev.float().reshape(m, -1)
        for lo in range(0, m, self._CHUNK):
            hi = min(m, lo + self._CHUNK)
            _, lp, _ = self.evaluate_actions(self._rows_u8(obs_dev[lo:hi]), a[lo:hi])
            out[lo:hi].copy_(lp)

    def ppo_update(self, rb, perm_dev: th.Tensor, n_epochs: int, batch_size: int, normalize_advantage: bool,
                   clip_range: float, ent_coef: float, vf_coef: float, max_grad_norm: float, stats: th.Tensor,
                   dp=None) -> None:
        """[SB3 PPO.train] on the rollout tile `rb` (see `GeneralTowers.ppo_update`): per minibatch gather the rows,
        forward, `ia_ppo_head_loss` (gradients w.r.t. logits / means, values, log_std + the logged statistics),
        `backward`, `clip_grad_norm_`, Adam."""
        T, n = rb.buffer_size, rb.n_envs
        total, D, A = T * n, self.obs_dim, self.act_dim
        aw = 1 if self.discrete else A
        s = L.stream()
        offs = (perm_dev % T) * n + perm_dev // T          # time-major row of every permuted index
        obs_rows, act_rows = rb.obs.reshape((T + 1) * n, D), rb.acts.reshape(total, aw)
        rows_x = getattr(rb, "_obs_x", None) if getattr(rb, "obs_u8", False) else None
        if rows_x is not None and not (rows_x.dtype == th.uint8 and rows_x.is_contiguous() and D % 4 == 0 and
                                       rows_x.numel() == (T + 1) * n * D):
            rows_x = None
        vecs = ((rb.logp.reshape(total, 1), "old"), (rb.adv.reshape(total, 1), "adv"), (rb.ret.reshape(total, 1), "ret"))
        opt = self.optimizer
        grad = opt.grad
        dev = self.device
        if getattr(self, "_clip_ws", None) is None:   # partials of the grid-wide gradient norm (long gradients)
            self._clip_ws = th.empty(int(L.load().ia_clip_grad_norm_ws_floats()), device=self.device)
        clip_ws = self._clip_ws
        for e in range(n_epochs):
            for mb, start in enumerate(range(0, total, batch_size)):
                b = min(batch_size, total - start)
                idx = offs[e, start:start + b]
                if rows_x is not None:
                    # the rollout tile's uint8 transport copy (`RolloutBuffer(obs_u8=True)`): the rows are gathered as they
                    # are -- four frame bytes per 4-byte element -- instead of as fp32 rows converted back to uint8
                    rows = th.empty(b, D, dtype=th.uint8, device=dev)
                    L.call("ia_gather_rows", L.ptr(rows_x), L.ptr(idx), b, D // 4, L.ptr(rows), s)
                    d = self._forward(rows.view(b, *self.observation_space.shape))
                else:
                    rows = th.empty(b, D, device=dev)
                    L.call("ia_gather_rows", L.ptr(obs_rows), L.ptr(idx), b, D, L.ptr(rows), s)
                    d = self._forward(self._rows_u8(rows))
                if "old" not in d:
                    d.update(old=th.empty(b, device=dev), adv=th.empty(b, device=dev), ret=th.empty(b, device=dev),
                             ms=th.empty(2, device=dev),
                             loss_ws=th.empty(int(L.load().ia_ppo_head_loss_ws_floats(b)), device=dev))
                L.call("ia_gather_rows", L.ptr(act_rows), L.ptr(idx), b, aw, L.ptr(d["acts"]), s)
                for src, key in vecs:
                    L.call("ia_gather_rows", L.ptr(src), L.ptr(idx), b, 1, L.ptr(d[key]), s)
                ms = None
                if normalize_advantage and b > 1:
                    L.call("ia_adv_moments", L.ptr(d["adv"]), b, L.ptr(d["ms"]), s)
                    ms = L.ptr(d["ms"])
                grad.zero_()
                L.call("ia_ppo_head_loss", int(self.discrete), L.ptr(d["logits"]), None if self.discrete else L.ptr(self._flat),
                       L.ptr(d["values"]), L.ptr(d["acts"]), L.ptr(d["old"]), L.ptr(d["adv"]), L.ptr(d["ret"]), ms, b, A,
                       float(clip_range), float(ent_coef), float(vf_coef), L.ptr(d["dlogits"]), L.ptr(d["dvalues"]),
                       None if self.discrete else L.ptr(grad), L.ptr(d["loss_ws"]), L.ptr(stats[e, mb]), s)
                self.backward(b, grad, with_values=True)
                if dp is not None and dp.world > 1:
                    dp.allreduce_mean_(grad)
                L.call("ia_clip_grad_norm", L.ptr(grad), grad.numel(), float(max_grad_norm), None, L.ptr(clip_ws), s)
                opt.step()

    # ---- acting --------------------------------------------------------------------------------------------
    def forward(self, obs, deterministic: bool = False):
        """[SB3 ActorCriticPolicy.forward] -> (actions, values, log_prob) device tensors."""
        d = self._forward(self._obs_u8(obs))
        n = d["values"].shape[0]
        if self.discrete:
            if deterministic:
                a_dev = th.argmax(d["logits"], dim=1)
            else:
                a_dev = th.distributions.Categorical(logits=d["logits"].cpu()).sample().to(self.device)
            d["acts"].copy_(a_dev.reshape(n, 1).float())
            L.call("ia_categorical_loss", L.ptr(d["logits"]), self.n_actions, L.ptr(d["acts"]), n, self.n_actions, 0.0, 0.0,
                   L.ptr(d["logp"]), L.ptr(d["ent"]), None, L.stream())
            return a_dev.reshape((n, *self.action_space.shape)), d["values"].clone(), d["logp"].clone()
        noise = (th.zeros(n, self.act_dim) if deterministic else self.sample_noise(n)).to(self.device)
        acts, clip, lp = th.empty(n, self.act_dim, device=self.device), th.empty(n, self.act_dim, device=self.device), th.empty(n, device=self.device)
        L.call("ia_gauss_act", L.ptr(d["logits"]), L.ptr(self._flat), L.ptr(noise), L.ptr(self._low), L.ptr(self._high), n,
               self.act_dim, L.ptr(acts), L.ptr(clip), L.ptr(lp), L.stream())
        return acts.reshape((n, *self.action_space.shape)), d["values"].clone(), lp

    __call__ = forward

    def predict(self, observation, state=None, episode_start=None, deterministic: bool = False):
        """[SB3 BasePolicy.predict]. The mode is the first argmax; sampling is [SB3 CategoricalDistribution.sample]
        itself on the host (`torch.distributions.Categorical(logits).sample()`: torch.multinomial on the global
        generator, the reference's stream -- as for the MLP policies)."""
        obs = np.asarray(observation)
        vectorized = obs.shape != tuple(self.observation_space.shape)
        if not self.discrete:
            acts = self.forward(obs.reshape((-1, *self.observation_space.shape)), deterministic=deterministic)[0]
            acts = np.clip(acts.cpu().numpy(), self.action_space.low, self.action_space.high)
            return (acts if vectorized else acts[0]), state
        d = self._forward(self._obs_u8(obs))
        logits = d["logits"].float().cpu()
        if deterministic:
            acts = logits.numpy().argmax(axis=1)
        else:
            acts = th.distributions.Categorical(logits=logits).sample().numpy()
        acts = acts.astype(np.int64)
        return (acts if vectorized else acts[0]), state
