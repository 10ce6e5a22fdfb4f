"""Behavioural cloning behind the reference's `BC` surface (`algorithms/bc.py:268-510`; SURVEY 8f row 4,
first slice: the actor-critic MLP policies of the path -- the NatureCNN policy of BASELINE config 4 is
not built yet).

One optimiser step = one pass of the policy kernels over a batch of expert rows:
  1. `ia_policy_evaluate`: log pi(a|s) and entropy of the batch (train-mode feature RunningNorm updated
     first, as `evaluate_actions` does) -> the logged metrics of `BehaviorCloningLossCalculator`
     (`bc.py:94-156`);
  2. `ia_ppo_minibatch`: the fused forward / loss / backward / Adam launch of the PPO step, fed so that its
     loss IS the BC loss: old log-probs = the current ones (ratio == 1, inside the clip range), advantages
     == 1 without normalisation  =>  d(-mean(ratio * A))/d logp = -1/B = d(-mean logp)/d logp;
     `ent_coef = ent_weight`; `vf_coef = 0` (the value tower gets no gradient, as in the reference, where
     `evaluate_actions`' values are discarded); no gradient clipping; Adam with the reference's defaults.
Index decisions stay on the host with the reference's RNG call sequence (one `DataLoader` iterator per
epoch: a base-seed draw, a sampler-seed draw and a `randperm` from torch's global generator).
The four logged means (log-prob, entropy, exp(log-prob), sum of squared parameters) and the flat-gradient
accumulation across minibatches are small torch reductions / axpys on the device tensors.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Any, Callable, Dict, Mapping, Optional

import numpy as np
import torch as th

from imitation_amd import _lib as L
from imitation_amd import data_types as dt
from imitation_amd import logger as imit_logger
from imitation_amd import policies as pol_mod
from imitation_amd.networks import require_device


class _EpochIndexStream:
    """Row indices of `DataLoader(transitions, batch_size, shuffle=True, drop_last=True)`, one epoch at a
    time, consuming torch's global generator like the loader (`algorithms/base.py:226-288`)."""

    def __init__(self, n_samples: int, batch_size: int):
        if batch_size <= 0:
            raise ValueError(f"batch_size={batch_size} must be positive.")
        if n_samples < batch_size:
            raise ValueError(f"Number of transitions in `demonstrations` {n_samples} "
                             f"is smaller than batch size {batch_size}.")
        self.n, self.batch_size = int(n_samples), int(batch_size)

    def epoch(self):
        dt.ExpertIndexStream._draw_int64()          # iter(loader): base seed
        perm = None
        for b in range(self.n // self.batch_size):
            if perm is None:                        # first next(): sampler seed + randperm
                seed = dt.ExpertIndexStream._draw_int64()
                g = th.Generator()
                g.manual_seed(seed)
                perm = th.randperm(self.n, generator=g).numpy()
            yield perm[b * self.batch_size:(b + 1) * self.batch_size]


class BC:
    def __init__(self, *, observation_space, action_space, rng: np.random.Generator, policy=None, demonstrations=None,
                 batch_size: int = 32, minibatch_size: Optional[int] = None, optimizer_cls=th.optim.Adam,
                 optimizer_kwargs: Optional[Mapping[str, Any]] = None, ent_weight: float = 1e-3,
                 l2_weight: float = 0.0, device="cuda", custom_logger=None):
        self.batch_size = batch_size
        self.minibatch_size = minibatch_size or batch_size
        if self.batch_size % self.minibatch_size != 0:
            raise ValueError("Batch size must be a multiple of minibatch size.")
        if optimizer_cls is not th.optim.Adam:
            raise NotImplementedError("the fused policy step implements Adam (the reference's default)")
        optimizer_kwargs = dict(optimizer_kwargs or {})
        if "weight_decay" in optimizer_kwargs:
            raise ValueError("Use the parameter l2_weight instead of weight_decay.")
        unknown = set(optimizer_kwargs) - {"lr", "betas", "eps"}
        if unknown:
            raise NotImplementedError(f"Adam options {sorted(unknown)} are not implemented on the HIP path")
        self._device = th.device("cuda" if device == "auto" else device)
        require_device(self._device)
        self._logger = custom_logger or imit_logger.configure()
        self._stream: Optional[_EpochIndexStream] = None
        if demonstrations is not None:
            self.set_demonstrations(demonstrations)
        self.action_space, self.observation_space, self.rng = action_space, observation_space, rng
        if policy is None:  # (constructed here, after the loader, like the reference: same draws from torch's RNG)
            policy = pol_mod.FeedForward32Policy(observation_space, action_space,
                                                 lr_schedule=lambda _: float(th.finfo(th.float32).max))
        self._policy = policy.to(self._device)
        assert self.policy.observation_space == self.observation_space
        assert self.policy.action_space == self.action_space
        self.lr = float(optimizer_kwargs.get("lr", 1e-3))
        self.betas = tuple(optimizer_kwargs.get("betas", (0.9, 0.999)))
        self.eps = float(optimizer_kwargs.get("eps", 1e-8))
        self.ent_weight, self.l2_weight = float(ent_weight), float(l2_weight)
        flat = self.policy._flat
        self._exp_avg, self._exp_avg_sq = th.zeros_like(flat), th.zeros_like(flat)
        self._steps = 0
        self._acc = th.zeros_like(flat)
        B = self.minibatch_size
        from imitation_amd.cnn_policy import ActorCriticCnnPolicy

        self._image = isinstance(self.policy, ActorCriticCnnPolicy)
        # explicit forward / backward launches (`evaluate_actions(..., want_grad=True)` + `backward`), then Adam on the
        # flat buffer: the convolution stack and MLP policies outside the fused kernels' shapes (any `net_arch`)
        self._explicit = self._image or not getattr(self.policy, "fused", True)
        if self._explicit:
            self._fused = False
        else:
            self._ws = th.zeros(int(L.load().ia_ppo_ws_floats(C.byref(self.policy.desc), B, B)), device=self._device)
            self._stats = th.zeros(8, device=self._device)
            self._ones, self._zeros = th.ones(B, device=self._device), th.zeros(B, device=self._device)
            self._rows = th.arange(B, dtype=th.int64, device=self._device)
            # gradient accumulation over minibatches and / or an L2 term: gradient and update as two launches
            # with the accumulated flat gradient in between; otherwise one fused launch per batch
            self._fused = self.minibatch_size == self.batch_size and self.l2_weight == 0.0
            self._grad_off = int(L.load().ia_ppo_grad_offset(C.byref(self.policy.desc), B))
        self._tensorboard_step = 0
        self._current_epoch = 0

    # ---- reference surface ------------------------------------------------------------------------
    @property
    def policy(self):
        return self._policy

    @property
    def logger(self):
        return self._logger

    def set_demonstrations(self, demonstrations) -> None:
        if isinstance(demonstrations, (list, tuple)) and len(demonstrations) and hasattr(demonstrations[0], "terminal"):
            demonstrations = dt.flatten_trajectories(list(demonstrations))
        obs, acts = np.asarray(demonstrations.obs), np.asarray(demonstrations.acts)
        n = len(obs)
        self._stream = _EpochIndexStream(n, self.minibatch_size)
        self._demo_host = (obs, acts)
        self._demo_obs = None      # device tables are laid out for the policy: uploaded on first use

    def _upload(self) -> None:
        obs, acts = self._demo_host
        n = len(obs)
        if self._image:   # frames stay uint8 [N, C, H, W]; the policy applies x / 255 while building its columns
            self._demo_obs = th.as_tensor(np.ascontiguousarray(obs)).to(self._device)
        else:
            self._demo_obs = th.as_tensor(np.ascontiguousarray(obs.reshape(n, -1))).to(self._device, th.float32)
            self._obs_b = th.empty(self.minibatch_size, self._demo_obs.shape[1], device=self._device)
        self._demo_acts = th.as_tensor(np.ascontiguousarray(acts.reshape(n, -1))).to(self._device, th.float32)
        self._acts_b = th.empty(self.minibatch_size, self._demo_acts.shape[1], device=self._device)

    def _gather(self, idx: np.ndarray):
        """Expert rows `idx` of the device-resident demonstration table -> (obs [B, D], acts [B, A])."""
        B = len(idx)
        if self._demo_obs is None:
            self._upload()
        i = th.as_tensor(idx).to(self._device, non_blocking=True)
        acts = self._acts_b[:B]
        if self._image:   # uint8 frames: a byte gather (data movement only)
            obs = self._demo_obs.index_select(0, i)
            L.call("ia_gather_rows", L.ptr(self._demo_acts), L.ptr(i), B, self._demo_acts.shape[1], L.ptr(acts), L.stream())
            return obs, acts
        obs = self._obs_b[:B]
        L.call("ia_gather_rows", L.ptr(self._demo_obs), L.ptr(i), B, self._demo_obs.shape[1], L.ptr(obs), L.stream())
        L.call("ia_gather_rows", L.ptr(self._demo_acts), L.ptr(i), B, self._demo_acts.shape[1], L.ptr(acts), L.stream())
        return obs, acts

    # ---- one optimiser step ---------------------------------------------------------------------------
    def _step(self, idx: np.ndarray):
        """-> (log_prob [B], entropy [B]) of the batch BEFORE the update (what the reference logs)."""
        pol = self.policy
        obs, acts = self._gather(idx)
        _, logp, ent = pol.evaluate_actions(obs, acts)
        self._steps += 1
        bc1 = 1.0 - self.betas[0] ** self._steps
        bc2 = 1.0 - self.betas[1] ** self._steps
        rn = pol.features_extractor.normalize
        B = len(idx)
        L.call("ia_ppo_minibatch", C.byref(pol.desc), L.ptr(pol._flat), L.ptr(pol._flat_t),
               L.ptr(rn.running_mean) if rn else None, L.ptr(rn.running_var) if rn else None,
               L.ptr(rn.count) if rn else None, 0, L.ptr(obs), L.ptr(acts), L.ptr(logp), L.ptr(self._ones),
               L.ptr(self._zeros), L.ptr(self._rows), B, 1, B, 0, 0.2, self.ent_weight, 0.0, 3.0e38,
               L.ptr(self._exp_avg), L.ptr(self._exp_avg_sq), self.betas[0], self.betas[1], self.eps,
               self.lr / bc1, math.sqrt(bc2), L.ptr(self._ws), L.ptr(self._stats), L.stream())
        return logp, ent

    def _accumulate(self, idx: np.ndarray):
        """Gradient of one minibatch's share of the batch loss (`loss * minibatch_size / batch_size`,
        bc.py:494-499) added to the accumulator; -> (log_prob, entropy) of the minibatch."""
        pol = self.policy
        obs, acts = self._gather(idx)
        share = len(idx) / self.batch_size
        if self._explicit:
            B = len(idx)
            _, logp, ent = pol.evaluate_actions(obs, acts, logp_coef=-share / B, ent_coef=-self.ent_weight * share / B,
                                                want_grad=True)
            pol.backward(B, self._acc)
            if self.l2_weight:
                self._acc.add_(pol._flat, alpha=self.l2_weight * share)
            return logp, ent
        _, logp, ent = pol.evaluate_actions(obs, acts)
        rn = pol.features_extractor.normalize
        B = len(idx)
        P = pol._flat.numel()
        L.call("ia_ppo_minibatch_grad", C.byref(pol.desc), L.ptr(pol._flat), L.ptr(pol._flat_t),
               L.ptr(rn.running_mean) if rn else None, L.ptr(rn.running_var) if rn else None,
               L.ptr(rn.count) if rn else None, 0, L.ptr(obs), L.ptr(acts), L.ptr(logp),
               L.ptr(self._ones * share), L.ptr(self._zeros), L.ptr(self._rows), B, 1, B, 0, 0.2,
               self.ent_weight * share, 0.0, L.ptr(self._ws), L.stream())
        self._acc += self._ws[self._grad_off:self._grad_off + P]
        if self.l2_weight:
            self._acc.add_(pol._flat, alpha=self.l2_weight * share)   # d/dw of l2_weight * sum(w^2) / 2, this share
        return logp, ent

    def _apply_accumulated(self) -> None:
        pol = self.policy
        self._steps += 1
        bc1 = 1.0 - self.betas[0] ** self._steps
        bc2 = 1.0 - self.betas[1] ** self._steps
        P = pol._flat.numel()
        if self._explicit:
            L.call("ia_adam_step", L.ptr(pol._flat), L.ptr(self._acc), L.ptr(self._exp_avg), L.ptr(self._exp_avg_sq), P,
                   self.betas[0], self.betas[1], self.eps, 0.0, self.lr / bc1, math.sqrt(bc2), L.stream())
            self._acc.zero_()
            pol._sync_transposed()   # (policies that shadow their parameters refresh the copies)
            return
        self._ws[self._grad_off:self._grad_off + P].copy_(self._acc)
        L.call("ia_ppo_minibatch_apply", C.byref(pol.desc), L.ptr(pol._flat), L.ptr(pol._flat_t), self.minibatch_size,
               self.ent_weight, 0.0, 3.0e38, L.ptr(self._exp_avg), L.ptr(self._exp_avg_sq), self.betas[0],
               self.betas[1], self.eps, self.lr / bc1, math.sqrt(bc2), L.ptr(self._ws), L.ptr(self._stats), L.stream())
        self._acc.zero_()

    def _metrics(self, logp: th.Tensor, ent: th.Tensor, l2_norm: th.Tensor) -> Dict[str, float]:
        """`BehaviorCloningLossCalculator` (bc.py:138-156) from the per-row outputs; one read-back."""
        v = th.stack([logp.mean(), ent.mean(), th.exp(logp).mean(), l2_norm]).cpu().numpy().astype(np.float32)
        log_prob, entropy, prob_true_act, l2 = (np.float32(x) for x in v)
        ent_loss = np.float32(-self.ent_weight) * entropy
        neglogp = -log_prob
        l2_loss = np.float32(self.l2_weight) * l2
        return dict(neglogp=float(neglogp), entropy=float(entropy), ent_loss=float(ent_loss),
                    prob_true_act=float(prob_true_act), l2_norm=float(l2), l2_loss=float(l2_loss),
                    loss=float(neglogp + ent_loss + l2_loss))

    def _log_batch(self, batch_num: int, batch_size: int, num_samples_so_far: int, metrics: Mapping[str, float],
                   rollout_stats: Mapping[str, float]) -> None:
        """`BCLogger.log_batch` (bc.py:223-242)."""
        self.logger.record("batch_size", batch_size)
        self.logger.record("bc/epoch", self._current_epoch)
        self.logger.record("bc/batch", batch_num)
        self.logger.record("bc/samples_so_far", num_samples_so_far)
        for k, v in metrics.items():
            self.logger.record(f"bc/{k}", v)
        for k, v in rollout_stats.items():
            if "return" in k and "monitor" not in k:
                self.logger.record("rollout/" + k, v)
        self.logger.dump(self._tensorboard_step)
        self._tensorboard_step += 1

    def train(self, *, n_epochs: Optional[int] = None, n_batches: Optional[int] = None,
              on_epoch_end: Optional[Callable[[], None]] = None, on_batch_end: Optional[Callable[[], None]] = None,
              log_interval: int = 500, log_rollouts_venv=None, log_rollouts_n_episodes: int = 5,
              progress_bar: bool = False, reset_tensorboard: bool = False) -> None:
        """`BC.train` (bc.py:381-510). `progress_bar` is accepted and ignored (no tqdm on this path)."""
        if (n_epochs is None) == (n_batches is None):
            raise ValueError("Must provide exactly one of `n_epochs` and `n_batches` arguments.")
        assert self._stream is not None, "set_demonstrations() first"
        if reset_tensorboard:
            self._tensorboard_step = 0
        self._current_epoch = 0
        num_samples_so_far = 0
        n_minibatches = n_batches * (self.batch_size // self.minibatch_size) if n_batches is not None else None
        seen = 0          # minibatches so far
        batch_num = 0
        epoch = 0
        last = None       # (log_prob, entropy, l2_norm) of the last minibatch: what the reference logs

        def process_batch():
            if not self._fused:
                self._apply_accumulated()
            if batch_num % log_interval == 0:
                stats: Mapping[str, float] = {}
                if log_rollouts_venv is not None and log_rollouts_n_episodes > 0:
                    from imitation_amd import rollout

                    trajs = rollout.generate_trajectories(self.policy, log_rollouts_venv,
                                                          rollout.make_min_episodes(log_rollouts_n_episodes),
                                                          rng=self.rng)
                    stats = rollout.rollout_stats(trajs)
                self._log_batch(batch_num, self.minibatch_size, num_samples_so_far, self._metrics(*last), stats)
            if on_batch_end is not None:
                on_batch_end()

        done = False
        while not done:
            some = False
            for idx in self._stream.epoch():
                some = True
                # metrics are those of the minibatch before the update; l2 of the parameters likewise
                batch_num = seen * self.minibatch_size // self.batch_size
                will_log = batch_num % log_interval == 0
                l2_norm = (self.policy._flat.square().sum() / 2) if will_log else None
                logp, ent = self._step(idx) if self._fused else self._accumulate(idx)
                last = (logp, ent, l2_norm)
                num_samples_so_far += len(idx)
                seen += 1
                if num_samples_so_far % self.batch_size == 0:
                    process_batch()
                if n_minibatches is not None and seen >= n_minibatches:
                    done = True
                    break
            if done:
                break
            if not some:
                raise AssertionError(f"Data loader returned no data during epoch {epoch} -- did it reset correctly?")
            self._current_epoch = epoch + 1
            if on_epoch_end is not None:
                on_epoch_end()
            epoch += 1
            if n_epochs is not None and epoch >= n_epochs:
                break
        if num_samples_so_far % self.batch_size != 0:   # an incomplete last batch still steps (bc.py:505-508)
            batch_num += 1
            if last[2] is None:
                last = (last[0], last[1], self.policy._flat.square().sum() / 2)
            process_batch()
