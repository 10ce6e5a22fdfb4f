"""`AdversarialTrainer` -- the GAIL/AIRL round on MI355X behind the reference's constructor
and `.train()` surface (`algorithms/adversarial/common.py:95-632`).

What stays on the host, with the reference's exact RNG call sequence (=> bit-exact batch
composition on identical seeds): the expert permutation (`ExpertIndexStream`), the replay
indices (`np.random.randint` via `ReplayBuffer.sample_indices`), the label layout
(`[expert_mb | gen_mb]`), ring-buffer positions, fixed-horizon check, step counters, logging.
What runs in HBM through libimitation_hip.so: batch assembly (gather + concat + one-hot),
RunningNorm updates, discriminator forward + BCE + backward (gradient accumulation over
minibatches), Adam, train statistics, the policy log-prob pass, reward relabelling.
"""
from __future__ import annotations

import abc
import contextlib
import os
import time
from typing import Callable, Dict, Iterable, Mapping, Optional, Type

import numpy as np
import torch as th

from torch import nn

from imitation_amd import _lib as L
from imitation_amd import buffer, data_types as dt
from imitation_amd import logger as imit_logger
from imitation_amd import networks, ppo, reward_nets, wrappers
from imitation_amd.networks import HipAdam, TransitionTable, require_device
from imitation_amd.policies import ActorCriticPolicy


def compute_train_stats(disc_logits_expert_is_high: th.Tensor, labels_expert_is_one: th.Tensor,
                        disc_loss: th.Tensor) -> Mapping[str, float]:
    """`adversarial/common.py:27-92` for explicit logits/labels (device or host tensors)."""
    logits = disc_logits_expert_is_high.detach().float().cpu()
    labels = labels_expert_is_one.detach().cpu()
    n = len(labels)
    gen_true = labels == 0
    gen_pred = logits < 0
    correct = gen_pred == gen_true
    n_gen = float(gen_true.sum())
    n_exp = n - n_gen
    p = th.sigmoid(logits)
    ent = th.nn.functional.binary_cross_entropy_with_logits(logits, p, reduction="none").mean() if n else float("nan")
    return _stats_dict(float(th.as_tensor(disc_loss).float().mean()), float(correct.sum()),
                       float((correct & ~gen_true).sum()), float((correct & gen_true).sum()), float(gen_pred.sum()),
                       float(ent) * n if n else float("nan"), n_exp, n_gen)


def _stats_dict(loss, n_correct, n_correct_exp, n_correct_gen, n_pred_gen, ent_sum, n_exp, n_gen) -> Dict[str, float]:
    n = n_exp + n_gen
    return {
        "disc_loss": float(loss),
        "disc_acc": float(n_correct / n) if n > 0 else float("nan"),
        "disc_acc_expert": float("nan") if n_exp < 1 else float(n_correct_exp / n_exp),
        "disc_acc_gen": float(n_correct_gen / max(1.0, n_gen)),
        "disc_entropy": float(ent_sum / n) if n > 0 else float("nan"),
        "disc_proportion_expert_true": float(n_exp / n) if n > 0 else float("nan"),
        "disc_proportion_expert_pred": float((n - n_pred_gen) / n) if n > 0 else float("nan"),
        "n_expert": float(n_exp),
        "n_generated": float(n_gen),
    }


def _upload_table(samples: Mapping, obs_shape, discrete: bool, device) -> TransitionTable:
    """Explicit `expert_samples` / `gen_samples` dicts (`common.py:317-337,564-583`) -> device rows."""
    def arr(k):
        v = samples[k]
        return v.detach().cpu().numpy() if isinstance(v, th.Tensor) else np.asarray(v)
    n = len(arr("obs"))
    f32 = lambda a: th.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).reshape(n, -1).to(device)
    acts = arr("acts")
    acts_d = (th.from_numpy(np.ascontiguousarray(acts, dtype=np.int64)).reshape(n).to(device) if discrete
              else f32(acts))
    return TransitionTable(f32(arr("obs")), acts_d, f32(arr("next_obs")),
                           th.from_numpy(np.ascontiguousarray(arr("dones"), dtype=np.uint8)).to(device), discrete)


class AdversarialTrainer(abc.ABC):
    """Base class for adversarial imitation learning algorithms like GAIL and AIRL."""

    def __init__(self, *, demonstrations, demo_batch_size: int, venv, gen_algo, reward_net: reward_nets.RewardNet,
                 demo_minibatch_size: Optional[int] = None, n_disc_updates_per_round: int = 2, log_dir="output/",
                 disc_opt_cls: Type = th.optim.Adam, disc_opt_kwargs: Optional[Mapping] = None,
                 gen_train_timesteps: Optional[int] = None, gen_replay_buffer_capacity: Optional[int] = None,
                 custom_logger: Optional[imit_logger.HierarchicalLogger] = None, init_tensorboard: bool = False,
                 init_tensorboard_graph: bool = False, debug_use_ground_truth: bool = False,
                 allow_variable_horizon: bool = False, data_parallel=None, disc_grad_penalty_coef: float = 0.0,
                 disc_grad_penalty_target: float = 1.0):
        self.demo_batch_size = demo_batch_size
        self.demo_minibatch_size = demo_minibatch_size or demo_batch_size
        if self.demo_batch_size % self.demo_minibatch_size != 0:
            raise ValueError("Batch size must be a multiple of minibatch size.")
        self._logger = custom_logger or imit_logger.configure()
        # OPT-IN extension, default off (imitation_amd/grad_penalty.py): coef * E[(|grad_x D(x_hat)| - target)^2] on
        # interpolates of the expert / generator rows, added to the BCE gradient. Not in the reference.
        self.disc_grad_penalty_coef = float(disc_grad_penalty_coef)
        self.disc_grad_penalty_target = float(disc_grad_penalty_target)
        self.last_grad_penalty: Optional[th.Tensor] = None   # mean (|grad D| - target)^2 of the last minibatch (device)
        self.allow_variable_horizon = allow_variable_horizon
        self._horizon = None
        self.venv = venv
        self.gen_algo = gen_algo
        # the replay ring's index rows of a round's discriminator updates are drawn by the PPO permutation helper's C call,
        # behind the epoch permutations, on a copy of NumPy's global generator (`_replay_rows_spec`, `_disc_round`)
        self.predraw_disc_indices = True
        self.replay_rows_predrawn = 0    # rounds whose rows came from the helper (tests, profiles)
        self._pre_replay_rows = []
        # pipelined rounds whose discriminator updates end up on the critical path (the next relabelling had to wait for
        # them: gradient penalty, many updates per round): what the round draws from torch's global CPU generator -- the
        # expert index rows, the interpolation weights -- is drawn right behind the NEXT rollout's noise, while the host
        # waits for the PPO update anyway, instead of between that update's launch and the round's own enqueue
        # (`_round_predraw`; same draws, same order). False: always in place; "always": whatever the device is doing (tests).
        self.predraw_round_draws = True
        self.round_draws_predrawn = 0    # rounds that found all their expert rows drawn ahead (tests, profiles)
        # measurement (`bench.py`): a list here collects `time.perf_counter()` at the end of every round's host iteration --
        # the spread of the rounds inside one `train()` call, on every schedule, without a callback (which would switch
        # the pipelined schedule off)
        self.round_wall_stamps: Optional[list] = None
        self._pre_expert_rows = []
        self._gp_round_pre = None
        self._disc_critical = False
        self._disc_wait_score = 0
        self._ppo_done_event = None
        if isinstance(gen_algo, ppo.PPO):
            gen_algo.randint_spec_for_round = self._replay_rows_spec
        self._device = th.device(gen_algo.device)
        require_device(self._device)
        L.load()  # fail loudly here if the HIP extension is missing
        self._discrete = hasattr(venv.action_space, "n")
        self._expert_table: Optional[TransitionTable] = None
        self._expert_stream: Optional[dt.ExpertIndexStream] = None
        self._expert_batches = None
        self.set_demonstrations(demonstrations)

        self._global_step = 0
        self._disc_step = 0
        self.n_disc_updates_per_round = n_disc_updates_per_round
        self.debug_use_ground_truth = debug_use_ground_truth
        self._reward_net = reward_net.to(self._device)
        self._log_dir = str(log_dir)
        if init_tensorboard:
            # reference `common.py:223-227,343,386-387`: a histogram of the logits every 20th update. tensorboard is not
            # part of this image: say so and train on (every scalar still goes through the logger's csv / json formats)
            import warnings
            warnings.warn("init_tensorboard=True: no tensorboard writer in this build -- the discriminator-logit "
                          "histograms are skipped, all scalars are logged as usual", RuntimeWarning)
        self._disc_opt_cls = disc_opt_cls
        self._disc_opt_kwargs = dict(disc_opt_kwargs or {})
        # An `nn.Module` reward net (imitation_amd.modules, or any user subclass of its `RewardNet`): the
        # reference's plugin contract -- autograd graph through `forward`, `loss.backward()`, a torch optimiser
        # over its parameters (`common.py:218-221,353-372`); the contractions run in the HIP custom ops.
        self._module_net = isinstance(self._reward_net, nn.Module)
        store = None if self._module_net else self._reward_net._store
        if self._module_net:
            from imitation_amd import ops
            params = [p_ for p_ in self._reward_net.parameters() if p_.requires_grad]
            self._disc_opt = (ops.HipAdam(params, **self._disc_opt_kwargs) if disc_opt_cls is th.optim.Adam
                              else disc_opt_cls(params, **self._disc_opt_kwargs))
            self._torch_opt_params = None
        elif disc_opt_cls is th.optim.Adam:
            self._disc_opt = HipAdam(store.flat, store.grad, **self._disc_opt_kwargs)
            self._torch_opt_params = None
        else:  # any other torch optimiser: steps device views of the same flat buffers
            self._torch_opt_params = [p.requires_grad_(True) for p in self._reward_net.parameters()]
            self._disc_opt = disc_opt_cls(self._torch_opt_params, **self._disc_opt_kwargs)

        self.venv_buffering = wrappers.BufferingWrapper(self.venv)
        if debug_use_ground_truth:
            self.venv_wrapped = self.venv_buffering
            self.gen_callback = None
        else:
            self.venv_wrapped = wrappers.RewardVecEnvWrapper(self.venv_buffering,
                                                             reward_fn=self.reward_train.predict_processed)
            self.gen_callback = self.venv_wrapped.make_log_callback()
        self.venv_train = self.venv_wrapped
        self.gen_algo.set_env(self.venv_train)
        self.gen_algo.set_logger(self.logger)

        if gen_train_timesteps is None:
            gen_algo_env = self.gen_algo.get_env()
            assert gen_algo_env is not None
            self.gen_train_timesteps = gen_algo_env.num_envs
            if isinstance(self.gen_algo, ppo.OnPolicyAlgorithm):
                self.gen_train_timesteps *= self.gen_algo.n_steps
        else:
            self.gen_train_timesteps = gen_train_timesteps
        if gen_replay_buffer_capacity is None:
            gen_replay_buffer_capacity = self.gen_train_timesteps
        self._gen_replay_buffer = buffer.ReplayBuffer(gen_replay_buffer_capacity, self.venv, device=self._device)

        # ---- data parallelism over env batches (extension; the reference is single-process) ----
        self._dp = data_parallel
        self._dp_many = self._dp is not None and self._dp.world > 1
        if self._dp_many:
            pol = self.gen_algo.policy
            self.gen_algo.dp = self._dp
            if self._module_net:
                # `nn.Module` reward nets: every RunningNorm merges the ranks' moments itself; parameters and buffers
                # start from rank 0's; the gradients are averaged in one flat bucket per update (`_disc_update_module`)
                from imitation_amd import modules as _modules
                norms = [m_ for m_ in self._reward_net.modules() if isinstance(m_, _modules.RunningNorm)]
                norms += [m_ for m_ in self.reward_train.modules() if isinstance(m_, _modules.RunningNorm)
                          and all(m_ is not n_ for n_ in norms)]
                heads = [p_.data for p_ in self._reward_net.parameters()]
                heads += [b_ for b_ in self._reward_net.buffers() if all(b_ is not t for n_ in norms
                                                                         for t in (n_.running_mean, n_.running_var, n_.count))]
            else:
                norms = [n for _, n in self.reward_train._named_norms()]
                norms += [s.norm for _, s in self._reward_net._named_stacks() if s.norm is not None]
                heads = [store.flat]
            if pol.features_extractor.normalize is not None:
                norms.append(pol.features_extractor.normalize)
            for nrm in norms:
                nrm.dp = self._dp
            # (EMANorm keeps two more state tensors, `inv_learning_rate` and `num_batches`: without them in the broadcast
            #  ranks that loaded different checkpoints would diverge at the first update)
            extra = [t for nrm in norms for t in (getattr(nrm, "inv_learning_rate", None), getattr(nrm, "num_batches", None))
                     if isinstance(t, th.Tensor)]
            self._dp.broadcast_(heads + [pol._flat] + [t for nrm in norms
                                                       for t in (nrm.running_mean, nrm.running_var, nrm.count)] + extra)
            pol._sync_transposed()

        # ---- GAIL only: the discriminator update never reads the policy, so inside `train()` the
        # n_disc updates run on a second stream CONCURRENTLY with the PPO update of the same round
        # (results unchanged: host RNG draws keep their program order; the policy-feature-norm
        # side effect of `evaluate_actions` is replayed on the PPO stream after
        # the PPO update, exactly where the reference executes it).
        # Data-parallel runs keep it when the PPO update needs no per-step collective (global-minibatch
        # update): then only one stream at a time has collectives in flight, in the same order on all ranks.
        dp_many = self._dp_many
        # `nn.Module` reward nets (operator boundary) take part when nothing reads the net between a round's updates and
        # the relabelling behind the next rollout's last step (`PPO.collect_rollouts` relabels the whole tile there when
        # `modules.bulk_relabel_ok`; a net that has to be called once per environment step would read parameters the
        # discriminator stream is still updating) -- forward, backward and the optimiser step of `_disc_update_module` are
        # torch operators on the current stream, here the discriminator stream. Single process only (its per-update
        # gradient all-reduce would share the communicator with the generator's collectives). Image policies
        # (`cnn_policy.ActorCriticCnnPolicy`: PPO runs their own minibatch loop) like the MLP ones.
        from imitation_amd.cnn_policy import ActorCriticCnnPolicy
        module_ok = not self._module_net
        if self._module_net and not dp_many:
            from imitation_amd import modules as _modules
            module_ok = bool(debug_use_ground_truth) or _modules.bulk_relabel_ok(self.reward_train)
        # (AIRL's updates evaluate log pi(a|s) on the discriminator stream while the next rollout's act steps run: the
        #  image policy keeps ONE set of activation buffers per batch size, so it takes part for GAIL only)
        pol_ok = (isinstance(self.policy, ActorCriticPolicy)
                  or (isinstance(self.policy, ActorCriticCnnPolicy) and not self._needs_logp and not dp_many))
        self._overlap = (isinstance(self.gen_algo, ppo.PPO) and pol_ok
                         and (not dp_many or self.gen_algo._dp_global()) and module_ok)
        # AIRL's updates read log pi(a|s) of the policy PPO has just updated: they cannot run beside that
        # PPO update, only behind the next rollout (`_train_pipelined`), with the feature statistics each
        # update's own `evaluate_actions` would have seen taken from the merge snapshots.
        self._overlap_beside_ppo = self._overlap and not self._needs_logp
        self._disc_stream = L.side_stream(self._device, "disc") if self._overlap else None
        # (Measured and dropped for the image policy, whose rollout step is a chain of ~10 ordinary launches that queue behind
        #  the reward CNN's millisecond kernels: a CU-masked discriminator stream -- hipExtStreamCreateWithCUMask, 32 or 64 CUs
        #  left free. The updates ran 30 % slower on it (23.3 -> 30.4 ms per round's updates) for 12-25 % fewer CUs and the
        #  round lost more than the act steps gained: profiles/r05_image_gail.md.)
        self._in_overlap = False
        self._gen_stored_early = False
        self._overlap_k = 0
        self._quirk_ready = None
        self._quirk_seq = None
        self._quirk_seq_merged = None
        self._disc_t0 = None
        self._disc_timing = None
        self._quirk_snap = None
        self._quirk_snap_buf = None
        self._quirk_item = 0
        # GAIL only: let round r's discriminator updates run behind round r+1's environment stepping
        # (`_train_pipelined`); off -> every round is completed before the next one starts
        self.pipeline_rounds = True
        # GAIL, pipelined rounds: True = hold the discriminator updates of round r back until PPO r has
        # finished (they then run behind rollout r+1 and the latency-bound PPO chain has the device to
        # itself); False = start them beside PPO r (it gets ~10 % slower, but nothing waits for them at the
        # end of a short rollout); None = decide per round from measurements: behind while the updates fit
        # into the rollout's host time, beside otherwise. Pure scheduling: values are identical either way.
        # AIRL's updates read the updated policy and always wait.
        self.disc_behind_ppo: Optional[bool] = None
        # pre-assembled rounds on the fused update: all n updates through one C call (False: one call per update)
        self.disc_round_one_call = True
        # pipelined rounds: write round r-1's log row behind round r's enqueue instead of ahead of it -- None: only when
        # round r-1's updates are still running at that point (waiting would delay round r's); True / False: always / never
        # (tests). The rows are the same either way.
        self.disc_log_late: Optional[bool] = None
        # pipelined rounds: enqueue round r's updates right behind the PPO launch, ahead of the iteration's host-side waits
        # and logging (False: behind them, after `learn` has returned -- the schedule before; tests compare)
        self.disc_enqueue_early = True
        self._disc_ms_behind = None   # device time of a round's updates, measured while they ran alone
        self._disc_mode_behind = True
        self._quirk_pending = []
        self._quirk_slots = []

        B = self.demo_batch_size
        nq = max(1, self.n_disc_updates_per_round)
        self._quirk_idx_host = th.zeros(nq, 2, B, dtype=th.int64).pin_memory()
        self._ring_rows_drawn = []
        self._quirk_idx_dev = th.zeros(nq, 2, B, dtype=th.int64, device=self._device)
        self._idx_host = th.zeros(2, B, dtype=th.int64).pin_memory()
        self._idx_dev = th.zeros(2, B, dtype=th.int64, device=self._device)
        self._stats_dev = th.zeros(8, device=self._device)
        # one statistics row per update of a round: `train()` enqueues all n_disc updates before it
        # reads any of them back (the host prepares update k+1 while the GPU runs update k)
        self._stats_ring = th.zeros(nq, 8, device=self._device)
        # (two host copies, used in turn: a round's rows may be read back only after the NEXT round has been enqueued)
        self._stats_ring_hosts = [th.zeros(nq, 8).pin_memory(), th.zeros(nq, 8).pin_memory()]
        self._stats_ring_turn = 0
        self._use_ring = False
        self._bce_ws = th.zeros(int(L.load().ia_bce_ws_floats(2 * self.demo_minibatch_size)), device=self._device)
        self._dlogits = th.zeros(2 * self.demo_minibatch_size, device=self._device)
        self._logp = th.zeros(2 * self.demo_minibatch_size, device=self._device)
        od = int(np.prod(venv.observation_space.shape))
        ad = 1 if self._discrete else int(np.prod(venv.action_space.shape))
        self._pol_obs = th.zeros(2 * self.demo_minibatch_size, od, device=self._device)
        self._pol_act = th.zeros(2 * self.demo_minibatch_size, max(ad, 1), device=self._device)

    # ---- properties ---------------------------------------------------------------------------
    @property
    def logger(self):
        return self._logger

    @logger.setter
    def logger(self, value):
        self._logger = value

    @property
    def policy(self):
        policy = self.gen_algo.policy
        assert policy is not None
        return policy

    @abc.abstractmethod
    def logits_expert_is_high(self, state, action, next_state, done, log_policy_act_prob=None) -> th.Tensor:
        """Discriminator logits, high = expert-like (`common.py:269-294`)."""

    @property
    @abc.abstractmethod
    def reward_train(self) -> reward_nets.RewardNet:
        """Reward used to train generator policy."""

    @property
    @abc.abstractmethod
    def reward_test(self) -> reward_nets.RewardNet:
        """Reward used to train policy at "test" time after adversarial training."""

    _needs_logp = False  # AIRL sets True

    # ---- demonstrations -------------------------------------------------------------------------
    def set_demonstrations(self, demonstrations) -> None:
        """`common.py:306-311`: device-resident expert table + the DataLoader-equivalent index stream."""
        try:
            demos = dt.as_transitions(demonstrations)
        except TypeError:
            if isinstance(demonstrations, Iterable):  # iterable of ready-made batches (`base.py:284-286`)
                self._expert_table, self._expert_stream = None, None
                self._expert_batches = _endless(demonstrations, self.demo_batch_size)
                return
            raise
        self._expert_batches = None
        self._expert_stream = dt.ExpertIndexStream(len(demos), self.demo_batch_size)
        self._expert_table = _upload_table(dict(obs=demos.obs, acts=demos.acts, next_obs=demos.next_obs,
                                                dones=demos.dones), None, self._discrete, self._device)

    def _check_fixed_horizon(self, horizons: Iterable[int]) -> None:
        """`algorithms/base.py:77-110`."""
        if self.allow_variable_horizon:
            return
        hs = set(int(h) for h in horizons)
        if self._horizon is not None:
            hs.add(self._horizon)
        if len(hs) > 1:
            raise ValueError(f"Episodes of different length detected: {hs}. Variable horizon environments are "
                             "discouraged -- termination conditions leak information about reward. If you are SURE "
                             "you want to run imitation on a variable horizon task, then please pass in the flag: "
                             "`allow_variable_horizon=True`.")
        if len(hs) == 1:
            self._horizon = hs.pop()

    def _replay_rows_spec(self, rows_in_rollout: int):
        """(high, rows, row_len) of the `np.random.randint` draws the next `_disc_round` will make (`buffer.sample_indices`:
        one row of `demo_batch_size` indices below the ring's fill per update), given that the rollout about to be collected
        adds `rows_in_rollout` rows to the ring first; None: do not predraw."""
        buf = getattr(self, "_gen_replay_buffer", None)
        if not self.predraw_disc_indices or buf is None:
            return None
        return (min(buf.size() + rows_in_rollout, buf.capacity), self.n_disc_updates_per_round, self.demo_batch_size)

    # ---- discriminator update (`common.py:317-389,521-632`) --------------------------------------
    def _batch_sources(self, expert_samples, gen_samples, upload: bool = True):
        """(expert_table, expert_idx_dev | None), (gen_table, gen_idx_dev | None) for one update. `upload=False` (ring
        rows only): the index rows stay in their pinned ring row, the caller uploads the rows of all its updates in one
        copy (`_upload_index_rows`) -- `self._ring_rows_drawn` collects the rows."""
        B = self.demo_batch_size
        e_idx = g_idx = None
        # inside an overlapped round every update gets its own (pinned, device) index rows so the
        # deferred policy-norm replay can still read them after later updates were enqueued
        ring = self._in_overlap or self._use_ring
        k = self._overlap_k % self._quirk_idx_dev.shape[0] if ring else None
        idx_host = self._quirk_idx_host[k] if k is not None else self._idx_host
        idx_dev = self._quirk_idx_dev[k] if k is not None else self._idx_dev
        if ring:
            self._overlap_k += 1
        if expert_samples is None:
            if self._expert_batches is not None:
                expert_samples = next(self._expert_batches)
            else:
                rows_e = self._pre_expert_rows.pop(0) if self._pre_expert_rows else self._expert_stream.next_indices()
                idx_host[0].copy_(th.from_numpy(rows_e))
                e_idx = idx_dev[0]
        if gen_samples is None:
            if self._gen_replay_buffer.size() == 0:
                raise RuntimeError("No generator samples for training. Call `train_gen()` first.")
            rows = self._pre_replay_rows.pop(0) if self._pre_replay_rows else self._gen_replay_buffer.sample_indices(B)
            idx_host[1].copy_(th.from_numpy(rows))
            g_idx = idx_dev[1]
        n_gen = B if gen_samples is None else len(gen_samples["obs"])
        n_exp = B if expert_samples is None else len(expert_samples["obs"])
        if not (n_gen == n_exp == B):
            raise ValueError("Need to have exactly `demo_batch_size` number of expert and generator samples, each. "
                             f"(n_gen={n_gen} n_expert={n_exp} demo_batch_size={B})")
        if e_idx is not None or g_idx is not None:
            if upload or k is None:
                idx_dev.copy_(idx_host, non_blocking=True)
            else:
                self._ring_rows_drawn.append(k)
        obs_shape = self.venv.observation_space.shape
        e_tab = self._expert_table if expert_samples is None else _upload_table(expert_samples, obs_shape,
                                                                                self._discrete, self._device)
        g_tab = self._gen_replay_buffer.table if gen_samples is None else _upload_table(gen_samples, obs_shape,
                                                                                        self._discrete, self._device)
        return (e_tab, e_idx), (g_tab, g_idx)

    def _upload_index_rows(self) -> None:
        """One host-to-device copy for the index rows of a round's updates (16 copies of 128 KB cost ~12 us of host time
        each and sit between the round's first kernels in the stream)."""
        rows, self._ring_rows_drawn = self._ring_rows_drawn, []
        if not rows:
            return
        if rows == list(range(rows[0], rows[0] + len(rows))):
            lo, hi = rows[0], rows[0] + len(rows)
            self._quirk_idx_dev[lo:hi].copy_(self._quirk_idx_host[lo:hi], non_blocking=True)
        else:
            for k in rows:
                self._quirk_idx_dev[k].copy_(self._quirk_idx_host[k], non_blocking=True)

    def _policy_pass(self, sources, mb: int, assembled: bool = False) -> Optional[th.Tensor]:
        """`common.py:606-615`: log pi(a|s) of the 2*mb rows under no_grad. For GAIL the value is
        discarded (`gail.py:157`) but the call still updates a train-mode feature RunningNorm
        (SURVEY App. C.2), so the statistics update is replayed even when logp is not needed."""
        pol = self.policy
        if not hasattr(pol, "log_prob_rows"):   # (MLP and image actor-critic policies both provide it)
            return None
        has_norm = pol.features_extractor.normalize is not None
        if not self._needs_logp and not (has_norm and pol.training):
            return None
        snapshot = None
        if self._in_overlap:
            if not self._needs_logp:  # replayed later on the generator's stream (see `train`)
                self._quirk_pending.append(list(sources))  # index views live in `_quirk_idx_dev` all round
                return None
            if self._quirk_snap is not None:  # statistics after THIS item's update (already merged, in order)
                snapshot = self._quirk_snap[self._quirk_item]
                self._quirk_item += 1
        R = 2 * mb
        row = 0
        for table, idx, n in ([] if assembled else sources):   # (assembled: `_pol_obs` / `_pol_act` already hold the rows)
            networks.gather_concat(table, idx, n, pol.obs_dim, pol.act_dim, (True, False, False, False),
                                   self._pol_obs, pol.obs_dim, row)
            if self._needs_logp:
                if table.discrete:
                    acts = table.acts if idx is None else table.acts[idx]
                    self._pol_act[row:row + n, 0].copy_(acts.float())
                else:
                    L.call("ia_gather_concat", L.ptr(table.obs), L.ptr(table.acts), None, L.ptr(table.next_obs),
                           L.ptr(table.dones), L.ptr(idx), n, pol.obs_dim, pol.act_dim, 0, 1, 0, 0,
                           L.ptr(self._pol_act), pol.act_dim, row, L.stream())
            row += n
        if self._needs_logp:
            pol.log_prob_rows(self._pol_obs[:R], self._pol_act[:R], self._logp, norm_snapshot=snapshot)
            return self._logp
        pol.features_extractor.normalize.update_stats(self._pol_obs[:R])
        return None

    def train_disc(self, *, expert_samples: Optional[Mapping] = None,
                   gen_samples: Optional[Mapping] = None) -> Mapping[str, float]:
        self._disc_update(expert_samples, gen_samples, self._stats_dev)
        s = self._stats_dev.cpu().numpy()  # the one host sync of a discriminator update
        return self._log_disc_stats(s, self._disc_step, self._global_step)

    def _log_disc_stats(self, s, disc_step: int, global_step: int) -> Mapping[str, float]:
        """`common.py:375-388`: the logging half of `train_disc` for one update's statistics row."""
        with self.logger.accumulate_means("disc"):
            train_stats = _stats_dict(*[float(x) for x in s])
            self.logger.record("global_step", global_step)
            for k, v in train_stats.items():
                self.logger.record(k, v)
            self.logger.dump(disc_step)
        return train_stats

    def _disc_round(self, prepass: bool = False, after=None, global_step: Optional[int] = None):
        """Enqueues the n_disc updates of one round without reading their statistics back; returns
        what `_finish_disc_round` needs to log them afterwards, in order. A `train_disc` replaced by
        the user (subclass or instance attribute) is honoured: then the updates run one by one."""
        n = self.n_disc_updates_per_round
        if "train_disc" in self.__dict__ or type(self).train_disc is not AdversarialTrainer.train_disc:
            for _ in range(n):
                with networks.training(self.reward_train):
                    self.train_disc()
            return None
        steps = []
        self._use_ring = True
        self._gp_block = None   # (a block left over by a round that raised is dropped, not served)
        if self._gp_round_pre is not None:   # the round's weights, drawn behind the rollout's noise (`_round_predraw`)
            th.cuda.current_stream().wait_event(self._gp_round_pre[3])   # (uploaded on a side stream)
            self._gp_block, self._gp_round_pre = self._gp_round_pre[:3], None
        if self._pre_expert_rows and len(self._pre_expert_rows) >= n:
            self.round_draws_predrawn += 1
        # the ring's index rows, if the permutation helper drew them for exactly this round and NumPy's global generator is
        # still where `PPO.train` left it (then it moves behind them: values and state of drawing in place)
        self._pre_replay_rows = []
        pre = getattr(self.gen_algo, "_predraw", None)
        ring = self._gen_replay_buffer
        plain_ring = (type(ring).sample_indices is buffer.ReplayBuffer.sample_indices
                      and "sample_indices" not in ring.__dict__)   # (a replaced sampler is asked, not bypassed)
        if self.predraw_disc_indices and plain_ring and pre is not None and hasattr(pre, "take_randint"):
            rows = pre.take_randint(self._gen_replay_buffer.size(), n, self.demo_batch_size)
            if rows is not None:
                self._pre_replay_rows = list(rows)
                self.replay_rows_predrawn += 1
        try:
            if prepass:
                # all host index draws of the round first (same order as one per update), then the
                # policy feature-norm moments of every update's batch, then the updates themselves
                self._ring_rows_drawn = []
                drawn = [self._batch_sources(None, None, upload=False) for _ in range(n)]
                self._upload_index_rows()
                round_ws = self._assemble_round(drawn)   # fused shapes: ONE launch assembles all n batches
                did = self._quirk_prepass(drawn, round_ws)
                if self._needs_logp:
                    # AIRL: the merge runs here, behind the PPO update, and leaves the statistics each
                    # update's own forward pass sees (`_policy_pass`); the next rollout waits for it
                    if after is not None:
                        th.cuda.current_stream().wait_event(after)
                    self._quirk_item = 0
                    self._replay_policy_norm_updates(snapshots=did)
                    self._quirk_ready = th.cuda.Event()
                    self._quirk_ready.record()
                else:
                    self._quirk_ready = th.cuda.Event()
                    self._quirk_ready.record()
                    if after is not None:   # the updates themselves are held back (see `_train_pipelined`)
                        th.cuda.current_stream().wait_event(after)
                    self._disc_t0 = th.cuda.Event(enable_timing=True)
                    self._disc_t0.record()
                if self._airl_round_ok(drawn, did):
                    # AIRL's fused shaped-net update: the round's updates in ONE C call too (`ia_airl_round`)
                    self._airl_round_one_call(drawn, n, did)
                    steps.extend(range(self._disc_step - n + 1, self._disc_step + 1))
                    n = 0
                elif self._round_one_call_ok(round_ws, did):
                    # every update of the round in ONE C call (`ia_disc_round_basic`): the launches of the loop below, its
                    # per-update host work (~110 us of Python each -- the round was bound by it) paid once
                    self._disc_round_one_call(round_ws, n)
                    steps.extend(range(self._disc_step - n + 1, self._disc_step + 1))
                    n = 0
                for k in range(n):
                    with networks.training(self.reward_train):
                        self._disc_update(None, None, self._stats_ring[k], drawn=drawn[k], quirk_done=did,
                                          pre=None if round_ws is None else (round_ws, k))
                    steps.append(self._disc_step)
                    if k == 0 and n > 1 and self.disc_grad_penalty_coef > 0.0:
                        # the other updates' interpolation weights as one block, drawn while the device works on the
                        # first update (whose own vector came through the per-update ring: nothing delays the round's
                        # first kernels)
                        self._gp_predraw_for_round(n - 1)
            else:
                for k in range(n):
                    with networks.training(self.reward_train):
                        self._disc_update(None, None, self._stats_ring[k])
                    steps.append(self._disc_step)
        finally:
            self._use_ring = False
            self._pre_replay_rows = []
        self._stats_ring_turn ^= 1
        host_rows = self._stats_ring_hosts[self._stats_ring_turn]
        host_rows.copy_(self._stats_ring, non_blocking=True)
        done = th.cuda.Event()
        done.record()
        self._gp_block_done(done)
        return done, steps, self._global_step if global_step is None else global_step, host_rows

    def _round_one_call_ok(self, round_ws, quirk_done: bool) -> bool:
        """The conditions under which `_disc_update` would take, for EVERY update of a pre-assembled round, the one-call
        fused update with the optimiser step inside and no host work between the updates."""
        if round_ws is None or not self.disc_round_one_call or self._dp is not None or self._module_net:
            return False   # (`_assemble_round` has checked the net, the optimiser, minibatch == batch and the penalty's shape)
        basic = self._reward_net
        while isinstance(basic, reward_nets.PredictProcessedWrapper):
            basic = basic.base
        if basic.mlp.norm is not None and not basic.mlp.norm.is_chan:
            return False
        pol = self.policy
        prn = pol.features_extractor.normalize if isinstance(pol, ActorCriticPolicy) else None
        return prn is None or not pol.training or bool(quirk_done)   # (else: a policy pass / moment copy per update)

    def _airl_round_ok(self, drawn, quirk_done: bool) -> bool:
        """The conditions under which `_disc_update` would take, for EVERY update of the round, AIRL's fused shaped-net update
        (`fused_prepare` + `log_prob_rows` + `fused_finish` with the optimiser step inside) with nothing else in between."""
        if (not self._needs_logp or not self.disc_round_one_call or self._dp is not None or self._module_net
                or self._torch_opt_params is not None
                or self.demo_minibatch_size != self.demo_batch_size or not isinstance(self._disc_opt, HipAdam)):
            return False
        basic = self._reward_net
        while isinstance(basic, reward_nets.PredictProcessedWrapper):
            basic = basic.base
        pol = self.policy
        if (not isinstance(basic, reward_nets.ShapedRewardNet) or not basic.fused_step_ok()
                or not isinstance(pol, ActorCriticPolicy) or not getattr(pol, "fused", False)):
            return False
        if any(e_idx is None or g_idx is None for (_, e_idx), (_, g_idx) in drawn):
            return False
        rn = pol.features_extractor.normalize
        # a train-mode feature norm: every update's log pi(a|s) needs ITS snapshot of the statistics (the merge launch's)
        return rn is None or not pol.training or (bool(quirk_done) and self._quirk_snap is not None)

    def _airl_round_one_call(self, drawn, n: int, quirk_done: bool) -> None:
        basic = self._reward_net
        while isinstance(basic, reward_nets.PredictProcessedWrapper):
            basic = basic.base
        pol = self.policy
        rn = pol.features_extractor.normalize
        snaps = None
        if rn is not None and pol.training:
            snaps = self._quirk_snap[self._quirk_item:self._quirk_item + n]
            self._quirk_item += n
        gp = None
        if self.disc_grad_penalty_coef > 0.0:
            # interpolation weights: torch's global CPU generator, one `th.rand(mb)` per update in update order (the draws of
            # the per-update loop: the first through the ring, the rest as one block)
            mb = self.demo_batch_size
            es = [self._gp_weights(mb)]
            if n > 1:
                self._gp_predraw_for_round(n - 1)
                es += [self._gp_weights(mb) for _ in range(n - 1)]
            gp = (es, self.disc_grad_penalty_coef, self.disc_grad_penalty_target)
        with networks.training(self.reward_train):
            logits = basic.airl_round_c(drawn, self.demo_batch_size, pol, self._pol_obs, self._pol_act, self._logp, snaps,
                                        1.0, self._stats_ring, self._disc_opt, gp=gp)
        if gp is not None:
            self.last_grad_penalty = basic._fused[0]["gp"]["pen"][0]
        self._disc_step += n
        self._last_disc_logits = logits

    def _disc_round_one_call(self, round_ws, n: int) -> None:
        basic = self._reward_net
        while isinstance(basic, reward_nets.PredictProcessedWrapper):
            basic = basic.base
        mb = self.demo_batch_size
        gp = None
        if self.disc_grad_penalty_coef > 0.0:
            # interpolation weights: torch's global CPU generator, one `th.rand(mb)` per update in update order (the first
            # through the per-update ring, the rest as one block -- the draws of the per-update loop)
            es = [self._gp_weights(mb)]
            if n > 1:
                self._gp_predraw_for_round(n - 1)
                es += [self._gp_weights(mb) for _ in range(n - 1)]
            gp = (es, self.disc_grad_penalty_coef, self.disc_grad_penalty_target)
        with networks.training(self.reward_train):
            ws = basic.disc_round_c(n, mb, 1.0, self._stats_ring, self._bce_ws, self._disc_opt, round_ws, gp=gp)
        if gp is not None:
            self.last_grad_penalty = ws["gp_out"][0]
        self._disc_step += n
        self._last_disc_logits = ws["out"].reshape(-1)

    def _finish_disc_round(self, pending) -> None:
        if pending is None:
            return
        done, steps, global_step, host_rows = pending
        done.synchronize()
        rows = host_rows.numpy()
        for k, step in enumerate(steps):
            self._log_disc_stats(rows[k], step, global_step)

    def _assemble_round(self, drawn):
        """Round-level batch assembly for the fused discriminator path (`BasicRewardNet.assemble_round`): every
        update's rows gathered by one launch, the input-norm updates applied in order by one more. None when it
        does not apply (other nets / shapes, gradient accumulation, explicit samples). Under data parallelism every
        rank assembles its own batches and the round's slab moments cross the ranks in one all-gather."""
        net = self._reward_net
        while isinstance(net, reward_nets.PredictProcessedWrapper):
            net = net.base
        if (self._module_net or not isinstance(net, reward_nets.BasicRewardNet) or self._needs_logp
                or (self.disc_grad_penalty_coef > 0.0 and net.fused_gp_ws(self.demo_batch_size) is None)
                or self._torch_opt_params is not None or self.demo_minibatch_size != self.demo_batch_size
                or not isinstance(self._disc_opt, HipAdam) or len(drawn) > self._quirk_idx_dev.shape[0]):
            return None
        (e_tab, e_idx), (g_tab, g_idx) = drawn[0]
        if e_idx is None or g_idx is None:
            return None
        with networks.training(self.reward_train):
            return net.assemble_round(e_tab, g_tab, self._quirk_idx_dev[:len(drawn)], len(drawn), self.demo_batch_size,
                                      dp=self._dp if self._dp_many else None)

    def _disc_update(self, expert_samples, gen_samples, stats_dev: th.Tensor, drawn=None,
                     quirk_done: bool = False, pre=None) -> None:
        """Device half of `train_disc` (`common.py:317-374`); the 8 statistics land in `stats_dev`.
        `drawn`: batch sources already drawn by `_batch_sources`; `quirk_done`: the policy feature-norm
        side effect of this update was already captured by `_quirk_prepass`."""
        (e_tab, e_idx), (g_tab, g_idx) = drawn if drawn is not None else self._batch_sources(expert_samples,
                                                                                               gen_samples)
        if self._module_net:
            self._disc_update_module((e_tab, e_idx), (g_tab, g_idx), stats_dev, quirk_done)
            return
        B, mb = self.demo_batch_size, self.demo_minibatch_size
        scale = mb / B
        net = self._reward_net
        first = True
        fused_step = False
        # single minibatch, HIP Adam, no cross-rank exchange: reduce + Adam in one launch
        single = not self._dp_many
        fuse_adam = self._disc_opt if (isinstance(self._disc_opt, HipAdam) and single) else None
        basic = net
        while isinstance(basic, reward_nets.PredictProcessedWrapper):
            basic = basic.base
        # data parallelism: the one-call update runs on batches the round-level assembly prepared (`pre`: rows gathered,
        # input statistics merged over all ranks); its slab reduction leaves the rank's gradient, ONE all-reduce and
        # the Adam + weight-image launch follow below. Updates outside such a round keep the general path.
        c_path = (isinstance(basic, reward_nets.BasicRewardNet) and not self._needs_logp
                  and (single or pre is not None) and self._torch_opt_params is None
                  and (basic.mlp.norm is None or basic.mlp.norm.is_chan))   # (EMANorm: stack-by-stack path)
        dp_fused = c_path and not single
        pol = self.policy
        prn = pol.features_extractor.normalize if isinstance(pol, ActorCriticPolicy) else None
        for start in range(0, B, mb):
            sl = lambda idx: None if idx is None else idx[start:start + mb]
            e_src = (e_tab if e_idx is not None else _slice_table(e_tab, start, mb), sl(e_idx), mb)
            g_src = (g_tab if g_idx is not None else _slice_table(g_tab, start, mb), sl(g_idx), mb)
            sources = [e_src, g_src]
            last = start + mb >= B
            if c_path:
                # policy feature-norm side effect (App. C.2): when the discriminator's own input norm
                # updates on a state-first batch, its slab moments cover the observation columns and
                # are reused; otherwise the explicit pass below gathers the observations.
                reuse = (prn is not None and pol.training and basic.use_state and basic.mlp.norm is not None
                         and basic.mlp.training and not quirk_done)
                if prn is not None and pol.training and not reuse and not quirk_done:
                    self._policy_pass(sources, mb)
                inline = reuse and not self._in_overlap
                gp = self.disc_grad_penalty_coef > 0.0
                # 128 / 256-wide fused update: the penalty's three tile passes + one split-K product ride in the same call
                # and share its slab reduction (+ Adam); other shapes add it stack by stack afterwards
                gp_in = gp and basic.fused_gp_ws(mb) is not None
                gp_arg = ((self._gp_weights(mb), self.disc_grad_penalty_coef * scale, self.disc_grad_penalty_target)
                          if gp_in else None)   # interpolation weights: torch's global CPU generator (only when enabled)
                ws = basic.disc_step_c(sources, mb, scale, stats_dev, self._bce_ws, accumulate=not first,
                                       adam=fuse_adam if (last and (not gp or gp_in)) else None,
                                       pnorm=prn if inline else None, pnorm_dim=pol.obs_dim if inline else 0, pre=pre,
                                       gp=gp_arg)
                if gp_in:
                    self.last_grad_penalty = ws["gp_out"][0]
                    gp = False   # (nothing left to add; the optimiser step below follows the ordinary rules)
                elif gp:
                    self._add_grad_penalty(basic.mlp, ws["X_used"], ws["norm_used"], mb, scale)
                if reuse and self._in_overlap:  # replayed on the generator stream after the PPO update
                    slot = self._quirk_moment_slot(ws["rn_ws"].numel())
                    slot.copy_(ws["rn_ws"])
                    self._quirk_pending.append((slot, 2 * mb, basic.mlp.dims[0]))
                logits = ws["out"].reshape(-1)
                fused_step = fused_step or (last and fuse_adam is not None and not gp)
            else:
                gp = self.disc_grad_penalty_coef > 0.0
                if gp and not isinstance(basic, reward_nets.ShapedRewardNet):
                    raise NotImplementedError("disc_grad_penalty_coef > 0 is implemented for BasicRewardNet (GAIL; "
                                              "state-holder or imitation_amd.modules net) and BasicShapedRewardNet "
                                              "(AIRL; state-holder) discriminators")
                if (self._needs_logp and isinstance(basic, reward_nets.ShapedRewardNet) and B == mb
                        and isinstance(self._disc_opt, HipAdam) and self._torch_opt_params is None
                        and basic.fused_step_ok() and hasattr(pol, "log_prob_rows")):
                    # AIRL's default shaped net: one assembly launch for the net's and the policy's rows (+ one for the
                    # input statistics), log pi(a|s), then forward, BCE, backward, reduction + Adam in five launches.
                    # Data-parallel: the slab moments of the ranks are all-gathered between assembly and merge, the
                    # reduced gradient is all-reduced below and the optimiser steps after it.
                    basic.fused_prepare(sources, self._pol_obs, self._pol_act, dp=None if single else self._dp)
                    logp = self._policy_pass(sources, mb, assembled=True)
                    logits = basic.fused_finish(logp, scale, stats_dev, None if gp else fuse_adam)
                    fused_step = not gp and fuse_adam is not None
                    if gp:   # penalty gradient on top of the reduced BCE gradient, then the optimiser step below
                        e = self._gp_weights(mb)   # interpolation weights: torch's global CPU generator
                        self.last_grad_penalty = basic.fused_grad_penalty(
                            e, self.disc_grad_penalty_coef * scale, self.disc_grad_penalty_target)[0]
                    first = False
                    continue
                logp = None if (quirk_done and not self._needs_logp) else self._policy_pass(sources, mb)
                logits = net.disc_forward(sources, mb, logp)
                L.call("ia_bce_logits", L.ptr(logits), 2 * mb, mb, scale, L.ptr(self._dlogits),
                       L.ptr(stats_dev), L.ptr(self._bce_ws), L.stream())
                fused_step = bool(net.disc_backward(self._dlogits, accumulate=not first,
                                                    adam=fuse_adam if (B == mb and not gp) else None))
                if gp:
                    self._add_shaped_grad_penalty(basic, mb, scale)
            first = False
        if dp_fused and isinstance(self._disc_opt, HipAdam) and not gp:
            # fused update under data parallelism: sum over the ranks, then 1 / world, Adam and the refresh of the weight
            # images the next pre-assembled update's tile kernels read in ONE launch
            self._dp.allreduce_sum_(net._store.grad)
            basic.fused_adam_step(self._disc_opt, 2 * mb, 1.0 / self._dp.world)
            fused_step = True
        elif self._dp is not None:  # one flat-bucket all-reduce per discriminator step
            self._dp.allreduce_mean_(net._store.grad)
        if self._torch_opt_params is not None:
            for p, (_, gview) in zip(self._torch_opt_params, _named_grads(net)):
                p.grad = gview
        if not fused_step:
            self._disc_opt.step()
        self._disc_step += 1
        self._last_disc_logits = logits

    def _gp_weights(self, mb: int) -> th.Tensor:
        """`th.rand(mb)` from torch's global CPU generator, on the device WITHOUT blocking the host: a plain
        `.to(device)` of pageable memory is a synchronous copy in stream order, i.e. the host waits for every update
        enqueued before it (250 us of host time per update measured with the penalty on). The draw goes into a ring of
        pinned buffers and is copied asynchronously; a slot is reused only after its copy has been consumed."""
        ring = getattr(self, "_gp_ring", None)
        if ring is None or ring[0].shape[1] != mb:
            n = 32
            ring = self._gp_ring = (th.empty(n, mb).pin_memory(), th.empty(n, mb, device=self._device),
                                    [None] * n, [0])
        blk = getattr(self, "_gp_block", None)
        if blk is not None and blk[2] < blk[0].shape[0] and blk[0].shape[1] == mb:
            blk[2] += 1                     # a round's weights, drawn and uploaded as ONE block (`_gp_predraw`)
            return blk[0][blk[2] - 1]
        host, dev, events, counter = ring
        k = counter[0] % host.shape[0]
        counter[0] += 1
        if events[k] is not None:
            events[k].synchronize()
        th.rand(mb, out=host[k])
        self._h2d_off_stream(dev[k], host[k])
        events[k] = th.cuda.Event()
        events[k].record()
        return dev[k]

    def _h2d_off_stream(self, dst: th.Tensor, src: th.Tensor) -> None:
        """Pinned -> device copy on the upload side stream, the current stream waiting for it -- instead of a copy IN the
        current stream. AIRL's discriminator stream waits for the PPO update (its log pi needs the updated policy); a copy
        enqueued behind that wait sits in the copy engine's queue until the PPO kernel ends, and every later copy of ANY
        stream queues behind it: with the penalty on, the read-back of the previous round's generator statistics
        (`PPO.finalize_train`, its own side stream) took 1.28 ms instead of 0.14 and the next rollout began that much
        later (`tools/drain_probe.py`, `profiles/r06_airl_gp.md`)."""
        up, cur = L.side_stream(self._device, "upload"), th.cuda.current_stream()
        if up == cur:
            dst.copy_(src, non_blocking=True)
            return
        with th.cuda.stream(up):
            dst.copy_(src, non_blocking=True)
            ev = th.cuda.Event()
            ev.record()
        cur.wait_event(ev)

    def _gp_predraw_for_round(self, n_updates: int) -> None:
        """Pre-draws the interpolation weights of the round's remaining updates when every update asks for exactly one
        vector: a single minibatch per update (every penalty path -- fused GAIL / AIRL, stack by stack -- draws one
        `th.rand(mb)` per minibatch; `_gp_block_done` checks that the block was consumed whole)."""
        mb = self.demo_minibatch_size
        blk = getattr(self, "_gp_block", None)
        if blk is not None and blk[2] < blk[0].shape[0]:
            return   # (the whole round's block is being served already: `_round_predraw`)
        if mb == self.demo_batch_size:
            self._gp_predraw(n_updates, mb)

    def _gp_predraw(self, count: int, mb: int) -> None:
        """The next `count` interpolation-weight vectors in ONE draw and ONE upload: `th.rand(count, mb)` consumes torch's
        CPU generator exactly as `count` draws of `th.rand(mb)` do (one 32-bit draw per element, in order), nothing else
        reads that generator while a round's updates are enqueued, and a copy per update inside the discriminator stream is
        a ~20 us bubble between two updates' kernels plus ~25 us of host time. Four pinned / device blocks in rotation; a
        block is reused only after the round that consumed it has completed (`_gp_block_done`)."""
        sets = getattr(self, "_gp_blocks", None)
        if sets is None:
            sets = self._gp_blocks = {}
        if (count, mb) not in sets:
            if len(sets) >= 4:
                sets.clear()
            sets[(count, mb)] = [[[th.empty(count, mb).pin_memory(), th.empty(count, mb, device=self._device), None]
                                  for _ in range(4)], 0]
        entry = sets[(count, mb)]
        b = entry[0][entry[1] % 4]
        entry[1] += 1
        if b[2] is not None:
            b[2].synchronize()
        th.rand(count, mb, out=b[0])
        self._h2d_off_stream(b[1], b[0])
        self._gp_block = [b[1], b, 0]

    def _round_predraw(self, last_rollout: bool = True) -> None:
        """`PPO.after_noise_predraw` of the pipelined schedule (`last_rollout`: False behind any but the last rollout of a
        round of several PPO iterations, `gen_train_timesteps > n_steps * n_envs` -- the sequential schedule draws the
        noise of the remaining rollouts first, so nothing is taken here): the coming discriminator round's draws from torch's global
        CPU generator -- its n expert index rows (`ExpertIndexStream`: two 64-bit draws per epoch of the expert table), then
        its n interpolation-weight vectors as one block -- taken HERE, right behind the rollout's noise, in the order the
        round takes them (`_disc_round`, prepass form: all index rows first, then the weights). Nothing else reads that
        generator in between (the condition under which the whole rollout's noise is one draw). Only when a previous
        round's updates were still running when the relabelling needed them (`_disc_critical`), and only while the
        previous PPO update is still running -- the host would wait for it anyway; what has not been drawn by then is
        drawn in place as before, continuing the same sequence."""
        ev = self._ppo_done_event
        force = self.predraw_round_draws == "always"   # (tests: regardless of what the device is doing)
        if not last_rollout:
            return
        if (not self.predraw_round_draws or not (self._disc_critical or force) or ev is None or self._expert_stream is None
                or self._pre_expert_rows or self._gp_round_pre is not None):
            return
        busy = (lambda: True) if force else (lambda: not ev.query())
        n = self.n_disc_updates_per_round
        while len(self._pre_expert_rows) < n and busy():
            self._pre_expert_rows.append(self._expert_stream.next_indices())
        mb = self.demo_minibatch_size
        if (len(self._pre_expert_rows) == n and self.disc_grad_penalty_coef > 0.0 and mb == self.demo_batch_size
                and busy()):
            up = L.side_stream(self._device, "upload")
            with th.cuda.stream(up):
                self._gp_predraw(n, mb)
                uploaded = th.cuda.Event()
                uploaded.record()
            self._gp_round_pre, self._gp_block = self._gp_block + [uploaded], None

    def _gp_block_done(self, event) -> None:
        blk = getattr(self, "_gp_block", None)
        if blk is not None:
            assert blk[2] in (0, blk[0].shape[0]), "a round's pre-drawn interpolation weights were only partly consumed"
            blk[1][2] = event
            self._gp_block = None

    def _add_grad_penalty(self, mlp, X: th.Tensor, norm, mb: int, scale: float) -> None:
        """Adds the gradient-penalty parameter gradient of one minibatch (`X[2*mb, ldx]` = [expert | generator] rows
        as assembled, `norm` = the (mean, var, eps) its forward normalised with, or None) to the flat gradient."""
        from imitation_amd import grad_penalty
        e = self._gp_weights(mb)      # interpolation weights: torch's global CPU generator (only when enabled)
        mean, var, eps = norm if norm is not None else (None, None, 0.0)
        pen, g = grad_penalty.penalty_and_param_grad(mlp.flat, mlp.dims, mlp.desc.hidden_act, X, mlp.ldx, mb, e, mean,
                                                     var, eps, self.disc_grad_penalty_coef * scale,
                                                     self.disc_grad_penalty_target)
        L.call("ia_reduce_partials", L.ptr(g), 1, g.numel(), 1.0, 1, L.ptr(mlp.grad), L.stream())
        self.last_grad_penalty = pen

    def _add_shaped_grad_penalty(self, shaped, mb: int, scale: float, batches=None) -> None:
        """The same for AIRL's shaped net, on the batches its forward of this minibatch assembled
        (`grad_penalty.shaped_penalty_and_param_grad`; input statistics as that forward left them, frozen)."""
        from imitation_amd import grad_penalty
        if batches is None:
            wg, wn, wc, aux, R = shaped._last
            batches = (wg["X"], shaped._base.mlp.ldx, wn["X"], wc["X"], shaped.potential._potential_net.ldx, aux["dones"])
        Xb, ldb, Sn, Sc, ldp, dones = batches
        base, pot = shaped._base, shaped.potential._potential_net
        bm = base.mlp
        if pot.desc.hidden_act != bm.desc.hidden_act:
            raise NotImplementedError("the gradient penalty needs the same activation in both stacks")
        e = self._gp_weights(mb)      # interpolation weights: torch's global CPU generator (only when enabled)
        stats = lambda n: None if n is None else (n.running_mean, n.running_var, n.eps)
        pen, gb, gpot = grad_penalty.shaped_penalty_and_param_grad(
            bm.flat, bm.dims, pot.flat, pot.dims, bm.desc.hidden_act, Xb, ldb, Sn, Sc, ldp, dones, mb, e, base.obs_dim, base.act_dim, base.flags, stats(bm.norm), stats(pot.norm),
            shaped.discount_factor, self.disc_grad_penalty_coef * scale, self.disc_grad_penalty_target)
        L.call("ia_reduce_partials", L.ptr(gb), 1, gb.numel(), 1.0, 1, L.ptr(bm.grad), L.stream())
        L.call("ia_reduce_partials", L.ptr(gpot), 1, gpot.numel(), 1.0, 1, L.ptr(pot.grad), L.stream())
        self.last_grad_penalty = pen

    def _module_batch(self, sources):
        """`common.py:564-603` for the autograd path: [expert | generator] rows of one minibatch as preprocessed
        device tensors (observations in their space's shape, one-hot actions for Discrete spaces, done as fp32)."""
        from imitation_amd import ops
        cols = {"obs": [], "acts": [], "next_obs": [], "dones": []}
        for table, idx, n in sources:
            for k, t in (("obs", table.obs), ("next_obs", table.next_obs)):
                cols[k].append(ops.gather_rows(t, idx) if idx is not None else t[:n])
            a = table.acts[idx] if idx is not None else table.acts[:n]
            if table.discrete:
                a = th.nn.functional.one_hot(a.long(), num_classes=self.venv.action_space.n).float()
            cols["acts"].append(a.float())
            d = table.dones[idx] if idx is not None else table.dones[:n]
            cols["dones"].append(d.to(th.float32))
        cat = {k: th.cat(v) for k, v in cols.items()}
        osp = self.venv.observation_space
        norm = getattr(self._reward_net, "normalize_images", True)
        # [SB3 preprocess_obs] (`rewards/reward_nets.py:90-110`): image spaces are scaled by 1 / 255, others are float
        img = lambda t: reward_nets.preprocess_space(t.reshape(-1, *osp.shape), osp, norm)
        return img(cat["obs"]), cat["acts"], img(cat["next_obs"]), cat["dones"]

    def _disc_update_module(self, e_src, g_src, stats_dev: th.Tensor, quirk_done: bool) -> None:
        """`train_disc` (`common.py:317-374`) for an `nn.Module` reward net: zero_grad, per minibatch
        `logits_expert_is_high` -> BCE-with-logits (scaled by minibatch / batch) -> `backward()`, then the
        optimiser step. Forward, backward, loss and Adam all run in the HIP custom ops."""
        from imitation_amd import ops
        (e_tab, e_idx), (g_tab, g_idx) = e_src, g_src
        B, mb = self.demo_batch_size, self.demo_minibatch_size
        self._disc_opt.zero_grad()
        stats = None
        for start in range(0, B, mb):
            sl = lambda idx: None if idx is None else idx[start:start + mb]
            sources = [(e_tab if e_idx is not None else _slice_table(e_tab, start, mb), sl(e_idx), mb),
                       (g_tab if g_idx is not None else _slice_table(g_tab, start, mb), sl(g_idx), mb)]
            logp = None if (quirk_done and not self._needs_logp) else self._policy_pass(sources, mb)
            state, action, next_state, done = self._module_batch(sources)
            logits = self.logits_expert_is_high(state, action, next_state, done,
                                                None if logp is None else logp[:2 * mb].clone())
            loss, stats = ops.bce_expert_first(logits, mb, mb / B)
            loss.backward()
            if self.disc_grad_penalty_coef > 0.0:
                self._module_grad_penalty(state, action, next_state, done, mb, mb / B)
        if self._dp_many:   # one flat bucket: the rank mean of the mean-reduced gradients = the gradient of the global mean
            with th.no_grad():
                grads = [p_.grad for g_ in self._disc_opt.param_groups for p_ in g_["params"] if p_.grad is not None]
                flat = th.cat([g_.reshape(-1) for g_ in grads])
                self._dp.allreduce_mean_(flat)
                o = 0
                for g_ in grads:
                    g_.copy_(flat[o:o + g_.numel()].view_as(g_))
                    o += g_.numel()
        self._disc_opt.step()
        stats_dev.copy_(stats)
        self._disc_step += 1
        self._last_disc_logits = logits.detach()

    def _module_grad_penalty(self, state, action, next_state, done, mb: int, scale: float) -> None:
        from imitation_amd import grad_penalty, modules
        net = self._reward_net
        if not (isinstance(net, modules.BasicRewardNet) and not self._needs_logp):
            raise NotImplementedError("disc_grad_penalty_coef > 0 is implemented for BasicRewardNet discriminators")
        mlp = net.mlp
        with th.no_grad():
            X = net.concat_inputs(state, action, next_state, done).contiguous()
            nrm = getattr(mlp, mlp._norm_name) if mlp._norm_name is not None else None
            e = self._gp_weights(mb)
            pen, g = grad_penalty.penalty_and_param_grad(
                mlp.flat_parameters().contiguous(), mlp.dims, mlp.act, X, X.shape[1], mb, e,
                None if nrm is None else nrm.running_mean, None if nrm is None else nrm.running_var,
                0.0 if nrm is None else nrm.eps, self.disc_grad_penalty_coef * scale, self.disc_grad_penalty_target)
            mlp.add_flat_grad(g)
        self.last_grad_penalty = pen

    # ---- generator update (`common.py:391-425`) -------------------------------------------------
    def train_gen(self, total_timesteps: Optional[int] = None, learn_kwargs: Optional[Mapping] = None) -> None:
        if total_timesteps is None:
            total_timesteps = self.gen_train_timesteps
        if learn_kwargs is None:
            learn_kwargs = {}
        with self.logger.accumulate_means("gen"):
            self.gen_algo.learn(total_timesteps=total_timesteps, reset_num_timesteps=False,
                                callback=self.gen_callback, **learn_kwargs)
            self._global_step += 1
        if self._gen_stored_early:   # (pipelined rounds: done behind the PPO launch, see `_train_pipelined`)
            self._gen_stored_early = False
            return
        self._store_gen_rollout()

    def _store_gen_rollout(self) -> None:
        """The tail of `train_gen` (`common.py:407-420`): the rollout's transitions go to the replay buffer."""
        rb = getattr(self.gen_algo, "rollout_buffer", None)
        device_rows = (isinstance(self.gen_algo, ppo.PPO) and rb is not None and rb.full
                       and self.venv_buffering.n_transitions == rb.buffer_size * rb.n_envs)
        store_ctx = th.cuda.stream(self._disc_stream) if self._in_overlap else contextlib.nullcontext()
        if device_rows:
            # the rollout tile is already in HBM: only the 8-byte row order crosses PCIe
            order, ep_lens, _ = self.venv_buffering.pop_order_and_lens()
            self._check_fixed_horizon(ep_lens)
            if self._in_overlap:  # the gather reads the tile the generator stream finished BEFORE its PPO update
                self._disc_stream.wait_event(self.gen_algo.rollout_done_event)
            with store_ctx:
                self._gen_replay_buffer.store_from_rollout(rb, order, infos=self.venv_buffering.last_infos)
            return
        gen_samples, ep_lens = self.venv_buffering.pop_transitions_and_lens()
        self._check_fixed_horizon(ep_lens)
        if gen_samples is not None:
            with store_ctx:  # in overlapped rounds the ring is only touched by the discriminator stream
                self._gen_replay_buffer.store(gen_samples)

    def _quirk_moment_slot(self, numel: int) -> th.Tensor:
        """Persistent per-update buffer for the slab moments replayed after the PPO update."""
        k = len(self._quirk_pending)
        while len(self._quirk_slots) <= k:
            self._quirk_slots.append(th.empty(numel, device=self._device))
        if self._quirk_slots[k].numel() != numel:
            self._quirk_slots[k] = th.empty(numel, device=self._device)
        return self._quirk_slots[k]

    def _quirk_prepass(self, drawn, round_ws=None) -> bool:
        """Slab moments of the observation columns of every (update, minibatch) batch of a round, on the
        current (discriminator) stream, queued for `_replay_policy_norm_updates`. They depend on the
        sampled rows only, so they can be taken before -- and independently of -- the updates, which
        lets the generator stream run ahead of the discriminator stream (see `train`). Returns whether
        the side effect applies at all (train-mode policy with a feature RunningNorm)."""
        pol = self.policy
        if not isinstance(pol, ActorCriticPolicy):
            return False
        rn = pol.features_extractor.normalize
        if rn is None or not pol.training:
            return False
        B, mb = self.demo_batch_size, self.demo_minibatch_size
        basic = self._reward_net
        while isinstance(basic, reward_nets.PredictProcessedWrapper):
            basic = basic.base
        if (round_ws is not None and round_ws.get("has_moments") and getattr(basic, "use_state", False)
                and pol.obs_dim <= basic.mlp.dims[0]):
            # the discriminator batches start with the observation columns and their slab moments were just taken
            # by the round assembly: the policy-feature-norm side effect (App. C.2) merges the same moments
            # (first obs_dim of D columns) -- no second gather
            self._quirk_seq_merged = round_ws["rn_seq"]   # (all ranks' moments under data parallelism)
            self._quirk_pending.append(("seq", len(drawn), round_ws["rn_stride"], round_ws["rn_groups"], 2 * mb,
                                        basic.mlp.dims[0]))
            return True
        need = int(L.load().ia_running_norm_ws_floats(2 * mb, pol.obs_dim))
        n_items = len(drawn) * len(range(0, B, mb))
        if self._quirk_seq is None or self._quirk_seq.shape != (n_items, need):
            self._quirk_seq = th.empty(n_items, need, device=self._device)
        k = 0
        if self._quirk_prepass_one_launch(drawn, need):
            drawn = ()   # (every batch's moments are in `_quirk_seq`: nothing left for the per-update launches below)
        for (e_tab, e_idx), (g_tab, g_idx) in drawn:
            for start in range(0, B, mb):
                row = 0
                for tab, idx in ((e_tab, e_idx), (g_tab, g_idx)):
                    t = tab if idx is not None else _slice_table(tab, start, mb)
                    i = None if idx is None else idx[start:start + mb]
                    networks.gather_concat(t, i, mb, pol.obs_dim, pol.act_dim, (True, False, False, False),
                                           self._pol_obs, pol.obs_dim, row)
                    row += mb
                L.call("ia_running_norm_partial", L.ptr(self._pol_obs), pol.obs_dim, 2 * mb, pol.obs_dim,
                       L.ptr(self._quirk_seq[k]), L.stream())
                k += 1
        groups, seq, stride = 1, self._quirk_seq, need
        if self._dp is not None and self._dp.world > 1:
            # every rank contributes its own batches: ONE all-gather of the round's moments, laid out
            # [update][rank][moments] so that each update merges world x 2*mb rows like a single process
            groups = self._dp.world
            allm = self._dp.all_gather_flat(self._quirk_seq.reshape(-1))
            seq = allm.view(groups, n_items, need).permute(1, 0, 2).contiguous()
            stride = groups * need
        self._quirk_seq_merged = seq
        self._quirk_pending.append(("seq", n_items, stride, groups, 2 * mb, pol.obs_dim))
        return True

    def _quirk_prepass_one_launch(self, drawn, need: int) -> bool:
        """The observation-column moments of ALL the round's batches in one launch (`ia_obs_moments_round`) instead of two
        gathers + one moment launch per update -- when every update is one minibatch on the same two tables and its index
        rows are the round's contiguous ring rows (the pipelined schedule). Same layout, same arithmetic."""
        B, mb = self.demo_batch_size, self.demo_minibatch_size
        if mb != B or not drawn:
            return False
        (e_tab, e0), (g_tab, g0) = drawn[0]
        if e0 is None or g0 is None:
            return False
        base, step = e0.data_ptr(), 2 * B * 8
        for k, ((et, ei), (gt, gi)) in enumerate(drawn):
            if (et is not e_tab or gt is not g_tab or ei is None or gi is None or ei.data_ptr() != base + k * step
                    or gi.data_ptr() != base + k * step + B * 8):
                return False
        od = self.policy.obs_dim
        if (e_tab.obs.dtype != th.float32 or g_tab.obs.dtype != th.float32 or e_tab.obs.shape[1:] != (od,)
                or g_tab.obs.shape[1:] != (od,) or not e_tab.obs.is_contiguous() or not g_tab.obs.is_contiguous()):
            return False
        n, ldx = len(drawn), (od + 3) // 4 * 4
        x = getattr(self, "_quirk_rows", None)
        if x is None or x.numel() < n * 2 * B * ldx:
            x = self._quirk_rows = th.empty(n * 2 * B * ldx, device=self._device)
        L.call("ia_obs_moments_round", L.ptr(e_tab.obs), base, B, L.ptr(g_tab.obs), base + B * 8, B, od, n, 2 * B, L.ptr(x), ldx,
               L.ptr(self._quirk_seq), need, L.stream())
        return True

    def _replay_policy_norm_updates(self, snapshots: bool = False) -> None:
        """Deferred `_policy_pass` side effects, in order, on the current stream. `snapshots`: also keep
        the statistics after every single update (`_quirk_snap`, read by AIRL's `_policy_pass`)."""
        pol = self.policy
        self._quirk_snap = None
        for item in self._quirk_pending:
            if isinstance(item, tuple) and item[0] == "seq":  # all updates of a round, one launch, in order
                _, n_items, stride, groups, rows, ld = item
                rn = pol.features_extractor.normalize
                snap = None
                if snapshots:
                    if self._quirk_snap_buf is None or self._quirk_snap_buf.shape[0] != n_items:
                        self._quirk_snap_buf = th.empty(n_items, 2, pol.obs_dim, device=self._device)
                    snap = self._quirk_snap = self._quirk_snap_buf
                L.call("ia_running_norm_merge_seq", L.ptr(self._quirk_seq_merged), n_items, stride, groups, rows,
                       pol.obs_dim, ld,
                       L.ptr(rn.running_mean), L.ptr(rn.running_var), L.ptr(rn.count), L.ptr(snap), L.stream())
                continue
            if isinstance(item, tuple):  # (slab moments of the batch, rows, moment column count)
                slot, rows, ld = item
                rn = pol.features_extractor.normalize
                L.call("ia_running_norm_merge", L.ptr(slot), 1, rows, pol.obs_dim, ld, L.ptr(rn.running_mean),
                       L.ptr(rn.running_var), L.ptr(rn.count), L.stream())
                continue
            sources = item
            row = 0
            for table, idx, n in sources:
                networks.gather_concat(table, idx, n, pol.obs_dim, pol.act_dim, (True, False, False, False),
                                       self._pol_obs, pol.obs_dim, row)
                row += n
            pol.features_extractor.normalize.update_stats(self._pol_obs[:row])
        self._quirk_pending = []

    def train(self, total_timesteps: int, callback: Optional[Callable[[int], None]] = None) -> None:
        """`common.py:427-461`."""
        n_rounds = total_timesteps // self.gen_train_timesteps
        assert n_rounds >= 1, ("No updates (need at least "
                               f"{self.gen_train_timesteps} timesteps, have only "
                               f"total_timesteps={total_timesteps})!")
        own_disc = "train_disc" not in self.__dict__ and type(self).train_disc is AdversarialTrainer.train_disc
        if self._overlap and callback is None and self.pipeline_rounds and own_disc:
            self._train_pipelined(n_rounds)
            return
        for r in range(n_rounds):
            # data-parallel GAIL: also the sequential schedules assemble a round's batches at once -- that is where the
            # ranks' input statistics are exchanged for the fused updates (AIRL's fused update exchanges per update)
            pre_dp = self._dp_many and not self._needs_logp
            if not self._overlap_beside_ppo:
                self.train_gen(self.gen_train_timesteps)
                self._overlap_k = 0
                pending = self._disc_round(prepass=pre_dp)
                if pre_dp:
                    self._replay_policy_norm_updates()
                self._finish_disc_round(pending)
            else:
                main = th.cuda.current_stream()
                self._in_overlap, self.gen_algo.defer_train_stats, self._overlap_k = True, True, 0
                try:
                    self._disc_stream.wait_stream(main)
                    self.train_gen(self.gen_train_timesteps)       # rollout; PPO update enqueued on `main`
                    with th.cuda.stream(self._disc_stream):         # ... while the disc updates run here
                        pending = self._disc_round(prepass=pre_dp)
                    main.wait_stream(self._disc_stream)
                    self._replay_policy_norm_updates()              # after the PPO update, in stream order
                    self._finish_disc_round(pending)
                    with self.logger.accumulate_means("gen"):
                        self.gen_algo.finalize_train()
                finally:
                    self._in_overlap, self.gen_algo.defer_train_stats = False, False
            if callback:
                callback(r)
            self.logger.dump(self._global_step)
            if self.round_wall_stamps is not None:
                self.round_wall_stamps.append(time.perf_counter())

    def _choose_disc_behind_ppo(self) -> bool:
        if self._needs_logp:
            return True
        if self.disc_behind_ppo is not None:
            return bool(self.disc_behind_ppo)
        self._disc_choose_calls = calls = getattr(self, "_disc_choose_calls", 0) + 1
        t = getattr(self, "_disc_timing", None)
        # a finished round measured while alone -- but not one of a trainer's first rounds: they carry one-off costs (code
        # objects loaded on first launch, first-touch allocations). A cold measurement used to put the image variant on
        # "beside" for good -- the updates' time is only measured in "behind" rounds -- where its round is 63-68 ms
        # instead of 56-58 (`profiles/r06_image_gail.md`).
        if t is not None and t[0] is not None and t[2] and t[1].query() and calls > 3:
            self._disc_ms_behind = t[0].elapsed_time(t[1])
        window = self.gen_algo.rollout_window_ms
        if self._disc_ms_behind is not None and window is not None:
            # hysteresis: switching costs nothing, but do not flap around the break-even point
            fits = self._disc_ms_behind < (0.95 if self._disc_mode_behind else 0.8) * window
            if not fits and self._disc_mode_behind:
                # (the probing gap doubles every time a probe ends on "beside" again -- 32, 64, ... 1 024 rounds: a probing round
                #  costs config P ~0.5 ms, and its answer there does not change)
                self._disc_probe_gap = min(1024, 2 * getattr(self, "_disc_probe_gap", 16))
                self._disc_probe_at = calls + self._disc_probe_gap
            self._disc_mode_behind = fits
        if not self._disc_mode_behind and calls >= getattr(self, "_disc_probe_at", 1 << 62):
            # after a while of "beside": one round "behind", to measure the updates alone again (both schedules compute the
            # same values: `test_pipelined_rounds_are_bit_identical`)
            self._disc_probe_at = 1 << 62
            self._disc_ms_behind = None
            self._disc_mode_behind = True
        return self._disc_mode_behind

    def _train_pipelined(self, n_rounds: int) -> None:
        """GAIL rounds with the discriminator updates of round r running BEHIND the environment stepping
        of round r+1 (the GPU is almost idle while the host steps the environments):

            main : rollout r | PPO r ............ | norm replay | rollout r+1 (act kernels) ... | relabel r+1 | PPO r+1
            disc :           | ring store, moment pre-pass r |  | 16 updates r ................ |

        Dependencies are exactly the reference's: the policy of rollout r+1 needs PPO r and the
        feature-norm side effects of round r's updates (the pre-pass + replay); the first use of the
        discriminator in round r+1 is the reward relabelling after the last env step, whose kernels wait
        for round r's updates (`PPO.before_relabel`). Round r's statistics are read back and logged right
        after PPO r+1 has been enqueued (`PPO.enqueue_first` / `after_enqueue`: nothing on the host delays
        a launch) -- into the log rows of round r, via `logger.replaying` -- or at the end of `train()`.
        Every value, every RNG draw and every log row equals the strictly sequential schedule
        (`test_pipelined_rounds_are_bit_identical`)."""
        algo = self.gen_algo
        main = th.cuda.current_stream()
        self._disc_stream.wait_stream(main)
        previous = []  # at most one round in flight: (disc handle, done event, root-log stash, global step)

        def gate():  # device side only: the relabelling kernels wait for the previous round's updates
            if previous:
                # are the updates what the round waits for (`_round_predraw`)? A score with hysteresis: rounds that are only
                # now and then a little late (config P: the updates end ~0.3 ms ahead of the relabelling) stay as they are
                waiting = not previous[-1][1].query()
                self._disc_wait_score = min(8, self._disc_wait_score + 2) if waiting else max(0, self._disc_wait_score - 1)
                if self._disc_wait_score >= 6:
                    self._disc_critical = True
                elif self._disc_wait_score <= 1:
                    self._disc_critical = False
                main.wait_event(previous[-1][1])

        late = []   # root-level records of the previous round's generator statistics, when its log row is written late

        def drain(final: bool = False):  # host side: read the previous round's statistics back and write its log rows
            if not previous:
                return
            pend, done, stash, gstep, train_rec = previous[-1]
            root = self.logger.default_logger
            defer = (not done.query()) if self.disc_log_late is None else bool(self.disc_log_late)
            if not final and not late and defer:
                # The previous round's updates are still running (long rounds, e.g. with the gradient penalty): waiting
                # here would hold back THIS round's updates by as long -- they are enqueued right after this hook. Only
                # what this iteration's own log row needs is done now: the generator's train statistics of the previous
                # round go to the raw/gen logger; their root-level means are kept and entered, in the same order,
                # when the round's row is written behind this round's enqueue (`drain_late`). Same rows either way.
                root.record_mean = lambda key, value, exclude=None: late.append((key, value, exclude))
                try:
                    with self.logger.replaying(stash), self.logger.accumulate_means("gen"):
                        algo._pending_train, keep = train_rec, algo._pending_train
                        algo.finalize_train()
                        algo._pending_train = keep
                finally:
                    del root.record_mean
                late.append(None)   # (marks "taken", also when nothing was recorded)
                return
            previous.pop()
            done.synchronize()
            with self.logger.replaying(stash):
                self._finish_disc_round(pend)
                if late:
                    for rec in late[:-1]:
                        root.record_mean(*rec)
                    late.clear()
                else:
                    with self.logger.accumulate_means("gen"):
                        algo._pending_train, keep = train_rec, algo._pending_train
                        algo.finalize_train()
                        algo._pending_train = keep
                self.logger.dump(gstep)

        early = []   # (pend, done) of the round enqueued from inside `learn`

        def enqueue_disc_round(global_step: int):
            ppo_done = th.cuda.Event()
            ppo_done.record()
            self._ppo_done_event = ppo_done
            with th.cuda.stream(self._disc_stream):
                # ring store and moment pre-pass run beside the PPO update; the 16 updates wait for it,
                # so they execute while the host steps the environments of the next round instead of
                # competing with the latency-bound PPO chain for the memory system
                behind = self._choose_disc_behind_ppo()
                pend = self._disc_round(prepass=True, after=ppo_done if behind else None, global_step=global_step)
                done = th.cuda.Event(enable_timing=True)
                done.record()
                self._disc_timing = (self._disc_t0, done, behind)
            return pend, done

        calls = [0]

        def after_train_enqueued(last: bool):
            # Right behind the PPO launch, AHEAD of the iteration's host-side waits (the reward bookkeeping waits for the
            # relabelled rewards, i.e. for the previous round's updates; then the previous round's rows are logged): the
            # rollout goes to the replay buffer and this round's updates are enqueued -- what `train_gen`'s tail and the
            # loop below do otherwise, in the same order among themselves (same index draws, same launches). Only for the
            # single-iteration `learn` of a round.
            calls[0] += 1
            if not last or calls[0] != 1:
                return
            self._store_gen_rollout()
            self._gen_stored_early = True
            early.append(enqueue_disc_round(self._global_step + 1))

        self._in_overlap, algo.defer_train_stats = True, True
        algo.before_relabel, algo.after_enqueue, algo.enqueue_first = gate, drain, True
        algo.after_train_enqueued = after_train_enqueued if self.disc_enqueue_early else None
        algo.after_noise_predraw = self._round_predraw
        try:
            for _ in range(n_rounds):
                self._overlap_k = 0
                calls[0] = 0
                self.train_gen(self.gen_train_timesteps)          # drains the previous round before relabelling
                pend, done = early.pop() if early else enqueue_disc_round(self._global_step)
                main.wait_event(self._quirk_ready)
                self._replay_policy_norm_updates()                 # behind the PPO update, ahead of the next rollout
                if late:   # the previous round's row, held back above
                    drain(final=True)
                train_rec, algo._pending_train = algo._pending_train, None
                previous.append((pend, done, self.logger.detach_pending(), self._global_step, train_rec))
                if self.round_wall_stamps is not None:
                    self.round_wall_stamps.append(time.perf_counter())
            drain(final=True)
            main.wait_stream(self._disc_stream)
        finally:
            self._in_overlap, algo.defer_train_stats = False, False
            algo.before_relabel, algo.after_enqueue, algo.enqueue_first = None, None, False
            algo.after_train_enqueued = None
            algo.after_noise_predraw = None
            self._ppo_done_event = None
            self._gen_stored_early = False


def _slice_table(t: TransitionTable, start: int, n: int) -> TransitionTable:
    return TransitionTable(t.obs[start:start + n], t.acts[start:start + n], t.next_obs[start:start + n],
                           t.dones[start:start + n], t.discrete)


def _named_grads(net):
    for prefix, s in net._named_stacks():
        yield from s.named_grads(prefix)


def _endless(batches, expected: int):
    """`util.endless_iter` over user-supplied batch iterables with the size check of
    `algorithms/base.py:185-223`."""
    if iter(batches) is batches:
        raise ValueError("endless_iter needs a non-iterator Iterable.")
    while True:
        for b in batches:
            if len(b["obs"]) != expected or len(b["acts"]) != expected:
                raise ValueError(f"Expected batch size {expected} != {len(b['obs'])} = len(batch['obs'])")
            yield b
