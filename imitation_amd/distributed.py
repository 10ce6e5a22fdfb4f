"""Data parallelism over env batches: one process per GPU, `torch.distributed` over RCCL/xGMI.

The reference has no distributed code at all; this is the MI355X-native
scale-out of the same round: every rank owns `n_envs` environments, its own expert
index stream and replay ring, and a full replica of policy + discriminator. Exchanges:

* discriminator: one flat-bucket all-reduce of the gradient per `train_disc` (289 KB at 256x256) between the fused
  update's reduction and its Adam step; the RunningNorm slab moments of ALL the round's updates cross the ranks in one
  all-gather and are merged in order, with per-update snapshots, by one launch (exactly the statistics a single
  process would have formed on the concatenated batches);
* PPO update (`ppo.PPO._train_dp_global`): ONE all-gather of the round's rollout shards, then the persistent update on
  the global tile -- either row-sharded (each rank its rows of every global minibatch; one record per optimiser step
  crosses the ranks INSIDE the kernels through peer-mapped memory, `PeerExchange` below: no collective on the chain) or
  replicated (every rank the whole global minibatch, no per-step exchange at all); which of the two runs is MEASURED on
  the node during the first updates (`PPO.dp_update_form = "auto"`, `PPO.dp_choice`). Policies outside the persistent
  kernel fall back to one all-reduce of the flat policy gradient per optimiser step (`_train_data_parallel`);
* a parameter broadcast from rank 0 at construction.
Replicas therefore stay bit-identical; global-norm clipping happens after the all-reduce.
"""
from __future__ import annotations

import os
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


class PeerExchange:
    """Peer-mapped exchange areas of the row-sharded data-parallel PPO update (`ia_ppo_update_sharded`).

    Every rank owns ONE block of device memory -- receive area `[2][world][P4 + 8]` 8-byte (value, sequence) words,
    16 handshake words -- that all ranks write into from inside their persistent PPO kernels: xGMI between
    GPUs, the same protocol between processes sharing a GPU (how it is tested on one-GPU boxes). The ranks exchange
    hipIpc handles of their blocks over `torch.distributed`, map each other's blocks, and prove the path with a
    handshake kernel (every rank writes a token into every rank's block and waits for everybody's token in its own);
    `ok` is the ranks' common verdict -- False sends the trainer back to the replicated update. `seq` counts the
    optimiser steps exchanged so far (flags carry it; it never goes back).

    `PeerExchange.loopback(world, ...)`: a single process standing in for `world` ranks (every "peer" is the own
    block; the record is written `world` times and summed `world` times) -- the cost model of `tools/dp_overhead.py`."""

    HS_WORDS = 16

    def __init__(self, dp: Optional[DataParallel], desc, world: Optional[int] = None, rank: int = 0,
                 handshake_timeout_s: float = 20.0):
        import ctypes as C
        from imitation_amd import _lib as L
        lib = L.load()
        self.dp = dp
        self.loop = dp is None or world is not None
        self.world = int(world if world is not None else dp.world)
        self.rank = int(rank if self.loop else dp.rank)
        self.recv_bytes = int(lib.ia_ppo_shard_recv_bytes(C.byref(desc), self.world))
        self.base, self.fine_grained, self.seq, self._opened = 0, False, 0, []
        # Every rank walks through the SAME collectives whatever fails locally (a rank that skipped one would hang the
        # others); `good` collects the local verdict and the last all-reduce makes it common.
        good = self.recv_bytes > 0
        nbytes = max(self.recv_bytes, 0) + 4 * self.HS_WORDS
        if good:
            base, fine = C.c_void_p(), C.c_int(0)
            good = lib.ia_peer_alloc(nbytes, C.byref(base), C.byref(fine)) == 0 and bool(base.value)
            if good:
                self.base, self.fine_grained = int(base.value), bool(fine.value)
        # test-only switch: the rank named by IA_PEER_FAIL_RANK reports that it could not export / map peer memory (what a
        # node without peer access between two of its GPUs looks like) -- every rank must land on the replicated update
        self.forced_failure = (not self.loop) and os.environ.get("IA_PEER_FAIL_RANK", "") == str(self.rank)
        if self.forced_failure:
            good = False
        bases = [self.base] * self.world
        if not self.loop:
            handle = (C.c_ubyte * 64)()
            if good:
                good = lib.ia_peer_ipc_export(self.base, handle) == 0
            mine = th.tensor(list(handle) + [1 if good else 0], dtype=th.uint8)
            allh = dp.all_gather_flat(mine if dp._stage else mine.cuda()).cpu().reshape(self.world, 65)
            good = good and bool(allh[:, 64].all())
            for r in range(self.world):
                if r == self.rank or not good:
                    continue
                buf = (C.c_ubyte * 64)(*allh[r, :64].tolist())
                out = C.c_void_p()
                if lib.ia_peer_ipc_open(buf, C.byref(out)) != 0 or not out.value:
                    good = False
                    continue
                bases[r] = int(out.value)
                self._opened.append(bases[r])
        off_hs = self.recv_bytes
        arr = C.c_void_p * 8
        self.recv = self.base
        self.peer_recv = arr(*([b for b in bases] + [None] * (8 - self.world)))
        self._peer_hs = arr(*([b + off_hs for b in bases] + [None] * (8 - self.world)))
        self._hs = self.base + off_hs
        self.ok = self._handshake(handshake_timeout_s, good)

    @classmethod
    def loopback(cls, world: int, desc) -> "PeerExchange":
        return cls(None, desc, world=world, rank=0)

    def _handshake(self, timeout_s: float, good: bool) -> bool:
        from imitation_amd import _lib as L
        if self.loop:
            return good
        import torch.distributed as dist_
        # (1) everybody has mapped everybody (or given up) before anybody writes into a peer's block
        ok = th.tensor([1 if good else 0], dtype=th.int32)
        buf = ok if self.dp._stage else ok.cuda()
        dist_.all_reduce(buf, op=dist_.ReduceOp.MIN, group=self.dp.group)
        if int(buf.cpu()[0]) != 1:
            return False
        # (2) the handshake kernel on every rank, then the common verdict
        res = th.zeros(1, dtype=th.int32).pin_memory()
        rc = L.load().ia_peer_handshake(self.world, self.rank, 1, self._hs, self._peer_hs, float(timeout_s),
                                        res.data_ptr(), L.stream())
        th.cuda.current_stream().synchronize()
        ok = th.tensor([1 if (rc == 0 and int(res[0]) == 1) else 0], dtype=th.int32)
        buf = ok if self.dp._stage else ok.cuda()
        dist_.all_reduce(buf, op=dist_.ReduceOp.MIN, group=self.dp.group)
        return int(buf.cpu()[0]) == 1

    def take_steps(self, n: int) -> int:
        """Sequence base of a launch of `n` optimiser steps (every rank makes the same launches)."""
        base = self.seq
        self.seq += int(n)
        return base

    def close(self) -> None:
        from imitation_amd import _lib as L
        lib = L.load()
        for p in self._opened:
            lib.ia_peer_ipc_close(p)
        self._opened = []
        if self.base:
            lib.ia_peer_free(self.base)
            self.base = 0
