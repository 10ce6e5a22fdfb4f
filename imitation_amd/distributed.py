"""Data parallelism over env batches: one process per GPU, `torch.distributed` over RCCL/xGMI.

The reference has no distributed code at all; this is the MI355X-native
scale-out of the same round: every rank owns `n_envs` environments, its own expert
index stream and replay ring, and a full replica of policy + discriminator. Exchanges:

* one flat-bucket all-reduce (mean) of the discriminator gradient per `train_disc`
  (289 KB at 256x256) and of the policy gradient per PPO minibatch (14 KB) -- latency-bound
  messages, so ONE collective per optimiser step on a persistent flat buffer;
* an all-gather of RunningNorm slab moments whenever a normalisation layer updates, followed by
  the same Chan merge on every rank (identical statistics everywhere, exactly the update a
  single process would have made on the concatenated batch);
* a parameter broadcast from rank 0 at construction.
Replicas therefore stay bit-identical; global-norm clipping happens after the all-reduce.
"""
from __future__ import annotations

from typing import List, Optional

import torch as th
import torch.distributed as dist


class DataParallel:
    def __init__(self, group: Optional[dist.ProcessGroup] = None):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed must be initialised (backend 'nccl' == RCCL on ROCm)")
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        # `gloo` (CPU tests, or several ranks sharing one GPU in the single-GPU test) moves device
        # buffers through the host; `nccl` (= RCCL) works on them in place over xGMI.
        self._stage = dist.get_backend(group) == "gloo"
        # collectives are issued from this world size on (a single rank needs none; tests lower it to 1 so that
        # the RCCL branches execute on a one-GPU box)
        self._min_world = 2

    def allreduce_mean_(self, flat: th.Tensor) -> th.Tensor:
        """In-place mean over ranks of one flat bucket."""
        if self.world >= self._min_world:
            if self._stage and flat.is_cuda:
                h = flat.cpu()
                dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
                flat.copy_(h)
            else:
                dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
            flat.mul_(1.0 / self.world)
        return flat

    def allreduce_sum_(self, flat: th.Tensor) -> th.Tensor:
        """In-place sum over ranks of one flat bucket (the caller folds 1 / world into its next kernel)."""
        if self.world >= self._min_world:
            if self._stage and flat.is_cuda:
                h = flat.cpu()
                dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
                flat.copy_(h)
            else:
                dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
        return flat

    def broadcast_(self, tensors: List[th.Tensor], src: int = 0) -> None:
        if self.world >= self._min_world:
            for t in tensors:
                if self._stage and t.is_cuda:
                    h = t.cpu()
                    dist.broadcast(h, src=src, group=self.group)
                    t.copy_(h)
                else:
                    dist.broadcast(t, src=src, group=self.group)

    def shared_seed(self) -> int:
        """One random 31-bit integer agreed on by every rank (drawn by rank 0). The value travels in a tensor on
        the backend's own device -- host memory for gloo, device memory for RCCL (which has no CPU collectives)."""
        t = th.randint(0, 2 ** 31 - 1, (1,), dtype=th.int64)
        if self.world >= self._min_world:
            buf = t if self._stage else t.cuda()
            dist.broadcast(buf, src=0, group=self.group)
            t = buf.cpu()
        return int(t.item())

    def all_gather_flat(self, local: th.Tensor) -> th.Tensor:
        """Concatenation of every rank's (equally sized) 1-D buffer, in rank order."""
        if self.world < self._min_world:
            return local
        if self._stage:
            parts = [th.empty(local.numel(), dtype=local.dtype) for _ in range(self.world)]
            dist.all_gather(parts, local.detach().cpu().contiguous(), group=self.group)
            return th.cat(parts).to(local.device)
        out = th.empty(self.world * local.numel(), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous(), group=self.group)
        return out


def merge_moments_reference(means: th.Tensor, m2s: th.Tensor, counts: th.Tensor):
    """Host restatement of the slab merge the device performs (`rn_merge_kernel`): sequential Chan
    combination of (count, mean, M2) triples in slab order -> (mean, biased variance) of the union.
    Used by the CPU tests of the data-parallel path."""
    n_acc = th.zeros((), dtype=means.dtype)
    m_acc = th.zeros_like(means[0])
    M2 = th.zeros_like(means[0])
    for mb, qb, nb in zip(means, m2s, counts):
        nb = nb.to(means.dtype)
        tot = n_acc + nb
        dlt = mb - m_acc
        M2 = M2 + qb + dlt * dlt * n_acc * nb / tot
        m_acc = m_acc + dlt * nb / tot
        n_acc = tot
    return m_acc, M2 / n_acc
