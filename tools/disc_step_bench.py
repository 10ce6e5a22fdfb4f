"""One discriminator update (C-ABI `ia_disc_step_basic`) at config-P shapes: the fused five-launch path
against the general path and against torch autograd (float64) on the same inputs, then both timed
back-to-back with HIP events on the launch stream.
Usage: [OD=27 AD=8] python tools/disc_step_bench.py [iters] [H] [R]"""
import os
import sys

import numpy as np
import torch as th

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import imitation_amd as P  # noqa: E402
from imitation_amd import _lib as L, reward_nets, spaces  # noqa: E402
from imitation_amd.networks import HipAdam, TransitionTable  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
H = int(sys.argv[2]) if len(sys.argv) > 2 else 256
R = int(sys.argv[3]) if len(sys.argv) > 3 else 16384
OD, AD, NE, NG = int(os.environ.get("OD", 17)), int(os.environ.get("AD", 6)), 64000, 16384   # OD=27 AD=8: Ant width (35 inputs)
dev = "cuda"
mb = R // 2


def tables(seed):
    g = th.Generator().manual_seed(seed)
    mk = lambda n: TransitionTable(th.randn(n, OD, generator=g).to(dev) * 2 + 0.5, th.rand(n, AD, generator=g).to(dev) * 2 - 1,
                                   th.randn(n, OD, generator=g).to(dev), (th.rand(n, generator=g) < 0.1).to(th.uint8).to(dev), False)
    return mk(NE), mk(NG)


def build(fused):
    reward_nets.FUSED_DISC_STEP = fused
    th.manual_seed(0)
    net = P.BasicRewardNet(spaces.Box(-np.inf, np.inf, (OD,)), spaces.Box(-1, 1, (AD,)), hid_sizes=(H, H),
                           normalize_input_layer=P.RunningNorm).to(dev)
    opt = HipAdam(net._store.flat, net._store.grad)
    return net, opt


def step(net, opt, e, g, ie, ig, stats, bce_ws, adam=True):
    with P.networks.training(net):
        return net.disc_step_c([(e, ie, mb), (g, ig, mb)], mb, 1.0, stats, bce_ws, accumulate=False,
                               adam=opt if adam else None)


if os.environ.get("TILE_ROWS"):
    L.load().ia_disc_fused_tile_rows(int(os.environ["TILE_ROWS"]))
SPLIT = os.environ.get("SPLIT") == "1"     # forward and backward tile passes as two launches (the form before round 3)
L.load().ia_disc_fused_split_tiles(int(SPLIT))
if os.environ.get("SIDE"):   # SIDE=0: the whole closing reduction in its own launch behind the split-K product
    L.load().ia_disc_fused_side_reduce(int(os.environ["SIDE"]))
e, g = tables(1)
gi = th.Generator().manual_seed(2)
ie = th.randint(0, NE, (mb,), generator=gi).to(dev)
ig = th.randint(0, NG, (mb,), generator=gi).to(dev)
bce_ws = th.zeros(int(L.load().ia_bce_ws_floats(R)), device=dev)
res = {}
for fused in (False, True):
    net, opt = build(fused)
    stats = th.zeros(8, device=dev)
    p0 = net._store.flat.clone()
    outs = []
    for k in range(3):  # three consecutive updates: norm statistics, Adam state and parameters evolve
        ws = step(net, opt, e, g, ie, ig, stats, bce_ws)
        th.cuda.synchronize()
        outs.append(dict(logits=ws["out"].reshape(-1).clone(), stats=stats.clone(), grad=net._store.grad.clone(),
                         params=net._store.flat.clone(), mean=net.mlp.norm.running_mean.clone(),
                         var=net.mlp.norm.running_var.clone(), count=int(net.mlp.norm.count), rn_ws=ws["rn_ws"].clone()))
    res[fused] = (outs, p0, ws)
    assert (ws.get("fused_ws") is not None) == fused, "fused path selection"

# ---- torch float64 reference of the first update
p0 = res[True][1].double().cpu()
X = th.cat([th.cat([e.obs[ie], e.acts[ie]], 1), th.cat([g.obs[ig], g.acts[ig]], 1)]).double().cpu()
mu, var = X.mean(0), X.var(0, unbiased=False)
Xn = (X - mu) / th.sqrt(var + 1e-5)
D = OD + AD
o = 0
Ws = []
for i, j in ((D, H), (H, H), (H, 1)):
    Ws.append(p0[o:o + i * j].view(j, i).clone().requires_grad_(True)); o += i * j
    Ws.append(p0[o:o + j].clone().requires_grad_(True)); o += j
h = th.relu(Xn @ Ws[0].T + Ws[1])
h = th.relu(h @ Ws[2].T + Ws[3])
logit = (h @ Ws[4].T + Ws[5]).reshape(-1)
y = th.cat([th.ones(mb), th.zeros(mb)]).double()
loss = th.nn.functional.binary_cross_entropy_with_logits(logit, y)
loss.backward()
gref = th.cat([w.grad.reshape(-1) for w in Ws])


def err(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max()), float((a - b).abs().max() / (b.abs().max() + 1e-30))


for fused in (False, True):
    o0 = res[fused][0][0]
    print(f"{'fused  ' if fused else 'general'} vs torch f64: logits {err(o0['logits'], logit.detach())}, grad {err(o0['grad'], gref)}, "
          f"loss {float(o0['stats'][0]):.7f} vs {float(loss):.7f}, mean {err(o0['mean'], mu)}, var {err(o0['var'], var)}")
for k in range(3):
    a, b = res[True][0][k], res[False][0][k]
    print(f"update {k}: fused vs general: " + ", ".join(f"{n} {err(a[n], b[n])[0]:.2e}" for n in ("logits", "grad", "params", "mean", "var", "stats")) +
          f", count {a['count']} / {b['count']}, slab moments bit-equal {bool(th.equal(a['rn_ws'], b['rn_ws']))}, "
          f"mean/var bit-equal {bool(th.equal(a['mean'], b['mean']) and th.equal(a['var'], b['var']))}")

# ---- timing
flops = R * (2 * (D * H + H * H + H) * 2 + 2 * (H * H + H))
for fused in (False, True):
    net, opt = build(fused)
    stats = th.zeros(8, device=dev)
    for _ in range(5):
        step(net, opt, e, g, ie, ig, stats, bce_ws)
    th.cuda.synchronize()
    ms = 0.0
    for _ in range(max(1, iters // 16)):  # 16 updates back to back = one round's worth; never a deep launch queue
        e0, e1 = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(16):
            step(net, opt, e, g, ie, ig, stats, bce_ws)
        e1.record()
        th.cuda.synchronize()
        ms += e0.elapsed_time(e1)
    us = 1e3 * ms / (16 * max(1, iters // 16))
    print(f"{'fused  ' if fused else 'general'}: {us:8.2f} us per update  {flops / us / 1e6:7.2f} TFLOP/s  "
          f"({100 * flops / us / 1e6 / 157.3:5.1f}% of the fp32 MFMA peak), R={R} H={H}")

# ---- the schedule of a pipelined round: ONE assembly launch for 16 updates, then the pre-assembled updates
net, opt = build(True)
stats = th.zeros(16, 8, device=dev)
idx_all = th.stack([th.stack([ie, ig])] * 16).contiguous()
ms = 0.0
reps = max(1, iters // 16)
for it in range(reps + 1):
    e0, e1 = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
    e0.record()
    with P.networks.training(net):
        rw = net.assemble_round(e, g, idx_all, 16, mb)
        for k in range(16):
            net.disc_step_c([(e, idx_all[k, 0], mb), (g, idx_all[k, 1], mb)], mb, 1.0, stats[k], bce_ws, accumulate=False,
                            adam=opt, pre=(rw, k))
    e1.record()
    th.cuda.synchronize()
    if it:   # (the first pass warms up)
        ms += e0.elapsed_time(e1)
us = 1e3 * ms / (16 * reps)
print(f"fused, round schedule (1 assembly launch per 16 updates + 3 launches per update): {us:8.2f} us per update  "
      f"{flops / us / 1e6:7.2f} TFLOP/s ({100 * flops / us / 1e6 / 157.3:5.1f}% of the fp32 MFMA peak)")

# ---- phase clocks of block 0 (fused tile kernels), shader cycles
reward_nets.FUSED_DISC_STEP = True
net, opt = build(True)
stats = th.zeros(8, device=dev)
buf = th.zeros(16, dtype=th.int64, device=dev)
L.load().ia_disc_fused_debug_timing(buf.data_ptr())
for _ in range(3):
    step(net, opt, e, g, ie, ig, stats, bce_ws)
th.cuda.synchronize()
L.load().ia_disc_fused_debug_timing(None)
t = buf.cpu().numpy()
names = ("prologue", "layer-1 MFMAs", "layer-1 epilogue", "layer-2 loop", "logit reduce", "BCE", "dh2/dW3 epilogue")
print("fwd tile kernel, block 0 (cycles): " + ", ".join(f"{n} {int(t[i + 1] - t[i])}" for i, n in enumerate(names)) + f", total {int(t[7] - t[0])}")
if SPLIT:
    names = ("prologue", "dgrad loop", "mask -> LDS", "dW1/db1")
    print("bwd tile kernel, block 0 (cycles): " + ", ".join(f"{n} {int(t[9 + i] - t[8 + i])}" for i, n in enumerate(names)) + f", total {int(t[12] - t[8])}")
else:
    print(f"... the same workgroup's backward half (cycles): dgrad loop {int(t[10] - t[7])}, mask -> LDS {int(t[11] - t[10])}, "
          f"dW1/db1 {int(t[12] - t[11])}; whole tile pass {int(t[12] - t[0])}")
