"""Whole-round cost of the data-parallel schedule on ONE GPU, collectives stubbed out: a stand-in DataParallel of
world W whose all-reduce is the identity and whose all-gather repeats the local buffer. What remains is the kernel and
host work the data-parallel round adds over the single-GPU round: the replicated persistent PPO update on the W x
minibatch, the fused discriminator updates split into slab reduce | (all-reduce) | Adam + weight images, the tile
packing around the rollout all-gather. Then the latency of the collectives themselves as far as a one-GPU box can show
it: `all_reduce` of the discriminator bucket (289 KB) and `all_gather_into_tensor` of a rollout shard (1.8 MB) through
RCCL with a world of one (launch + kernel overhead of the call; no xGMI hop).

Usage: python tools/dp_overhead.py [rounds]      (writes a markdown table to stdout)"""
import os
import sys
import time

import torch as th

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


class NullDP:
    rank = 0
    _stage = False

    def __init__(self, world):
        self.world = world
        self.calls = {"allreduce": 0, "allgather": 0}

    def allreduce_mean_(self, flat):
        self.calls["allreduce"] += 1
        return flat

    def allreduce_sum_(self, flat):
        self.calls["allreduce"] += 1
        return flat.mul_(float(self.world))   # the sum of identical replicas: the 1 / world that follows restores the mean

    def broadcast_(self, tensors, src=0):
        pass

    def all_gather_flat(self, local):
        self.calls["allgather"] += 1
        return th.cat([local] * self.world)

    def shared_seed(self):
        return 1234

    def make_peer_exchange(self, desc):
        """Row-sharded PPO update: this process stands in for every rank (its record is written and summed `world`
        times through its own block, `PeerExchange.loopback`)."""
        from imitation_amd.distributed import PeerExchange
        return PeerExchange.loopback(self.world, desc)


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    th.set_num_threads(1)
    cfg = dict(bench.CFG_P)
    per_round = cfg["n_envs"] * cfg["n_steps"]
    rows, base = [], None
    for world in (1, 2, 4, 8):
        dp = None if world == 1 else NullDP(world)
        tr = bench.build_trainer(bench.hip_namespace(), cfg, "cuda", seed=0, dp=dp)
        # (the stub world measures ONE form per run: IA_DP_ROW_SHARDED=0 the replicated one, else the row-sharded one -- the
        #  product's default, "auto", times both on the real node and keeps the faster)
        tr.gen_algo.dp_update_form = "replicated" if os.environ.get("IA_DP_ROW_SHARDED", "1") == "0" else "sharded"
        tr.train(4 * per_round)
        th.cuda.synchronize()
        if dp is not None:
            dp.calls = {"allreduce": 0, "allgather": 0}
        algo = tr.gen_algo
        algo.update_events = (th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True))
        t = time.perf_counter()
        tr.train(rounds * per_round)
        th.cuda.synchronize()
        dt = (time.perf_counter() - t) / rounds
        ppo_ms = algo.update_events[0].elapsed_time(algo.update_events[1])
        algo.update_events = None
        disc = bench.disc_update_timing(tr, cfg)
        base = base or dt
        calls = dp.calls if dp is not None else {"allreduce": 0, "allgather": 0}
        rows.append((world, 1e3 * dt, base / dt, ppo_ms, 1e3 * ppo_ms / (algo.n_epochs * algo._n_mb), disc["us"],
                     disc["path"].split(" ")[0], calls["allreduce"] / rounds, calls["allgather"] / rounds))
        del tr
        th.cuda.empty_cache()
    print("| world (stub collectives) | ms / round | weak-scaling efficiency bound | PPO update ms | us / optimiser step | "
          "disc update us (alone) | disc path | all-reduces / round | all-gathers / round |")
    print("|---|---|---|---|---|---|---|---|---|")
    for r in rows:
        print(f"| {r[0]} | {r[1]:.2f} | {r[2]:.3f} ({r[2] * r[0]:.2f}x) | {r[3]:.2f} | {r[4]:.1f} | {r[5]:.1f} | {r[6]} | "
              f"{r[7]:.0f} | {r[8]:.0f} |")

    # ---- RCCL call latency with a world of one (what a one-GPU box can measure)
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29571")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=th.device("cuda", 0))
    bucket = th.zeros(72193, device="cuda")                       # discriminator gradient bucket, 289 KB
    shard = th.zeros(16 * 1024 * (17 + 6 + 3), device="cuda")      # rollout shard, 1.8 MB
    out = th.empty_like(shard)
    print()
    print("| RCCL call (world 1) | bytes | host us / call | device us / call (events, back to back) |")
    print("|---|---|---|---|")
    for name, fn, nbytes in (("all_reduce(sum), disc bucket", lambda: dist.all_reduce(bucket), bucket.numel() * 4),
                             ("all_gather_into_tensor, rollout shard", lambda: dist.all_gather_into_tensor(out, shard),
                              shard.numel() * 4)):
        for _ in range(10):
            fn()
        th.cuda.synchronize()
        e0, e1 = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
        n = 200
        e0.record()
        t = time.perf_counter()
        for _ in range(n):
            fn()
        host = time.perf_counter() - t
        e1.record()
        th.cuda.synchronize()
        print(f"| {name} | {nbytes} | {1e6 * host / n:.1f} | {1e3 * e0.elapsed_time(e1) / n:.1f} |")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
