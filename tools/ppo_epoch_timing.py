"""Phase ticks of the one-launch-per-epoch PPO kernel (64-wide towers): workgroup 0 accumulates 100 MHz ticks per phase
-- the word-exchange kernel: {gradient incl. the parameter poll, slab poll + sum, sum of squares published, poll of the sums
of squares, Adam + publish}; IA_EPOCH_SPLIT=3 (the grid-barrier kernel): {gradient, barrier, slab sum, barrier, norm + Adam,
barrier}. Usage: [IA_EPOCH_SPLIT=3] python tools/ppo_epoch_timing.py [bench variant]"""
import os
import sys

import torch as th

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from imitation_amd import _lib as L  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "P_mlp64_1024x16"
th.set_num_threads(1)
mode = int(os.environ.get("IA_EPOCH_SPLIT", "0"))
L.load().ia_ppo_epoch_split(mode)
tr, per = bench.build_variant(name)
tr.train(3 * per)
th.cuda.synchronize()
buf = th.zeros(64, dtype=th.int64, device="cuda")
L.load().ia_ppo_epoch_debug_timing(buf.data_ptr())
rounds = 4
tr.train(rounds * per)
th.cuda.synchronize()
L.load().ia_ppo_epoch_debug_timing(None)
algo = tr.gen_algo
steps = rounds * algo.n_epochs * algo._n_mb
t = buf.cpu().numpy()[:6] / 100.0 / steps
chain2 = mode in (0, 6, 7) and tr.gen_algo.policy.obs_dim <= 32   # round 5's gradient phase (ppo_epoch_ll2_kernel)
names = (("gradient", "barrier", "slab sum + partial norm", "barrier", "norm + Adam + statistics", "barrier") if mode == 3 else
         ("gradient (incl. parameter poll)", "slab poll + sum", "partial norm published", "poll of the partial norms",
          "norm + Adam + publish + statistics", "-"))
print(f"{name}: per optimiser step (workgroup 0): " + ", ".join(f"{n} {v:.2f} us" for n, v in zip(names, t)) +
      f"; sum {t.sum():.2f} us")

# shader-clock stamps of the LAST step's gradient phase (row block 0; [16..] policy-tower workgroup, [32..] value-tower
# workgroup of the one-tower form; the whole-block form writes [16..] only)
if chain2:
    full = buf.cpu().numpy()
    lab = ("layer 1", "layer 2", "head", "loss", "dz2 / dz1", "barrier + park", "dW2 + db2", "dW1 + db1", "head gradient + sums",
           "closing barrier")
    for base, who in ((16, "policy tower"), (32, "value tower")):
        st = [int(full[base + i]) for i in range(11)]
        print(f"  {who} (row block 0, last step): " + ", ".join(f"{n} {st[i + 1] - st[i]}" for i, n in enumerate(lab)) +
              f"; total {st[-1] - st[0]} cycles")
    sys.exit(0)
order = (0, 9, 10, 11, 1, 2, 3, 4, 5, 6, 7, 8)
labels = (("row loads issued", "parameter copy issued", "rows staged (+ barrier)") if mode == 3 else
          ("row loads issued", "rows normalised + parameter poll + images", "block barrier")) + ("fragments", "layer 1", "layer 2", "heads",
          "loss", "head gradients / dz2", "dW2 / dz1", "dW1")
full = buf.cpu().numpy()
for base, who in ((16, "tower 0 / whole block"), (32, "tower 1")):
    st = [int(full[base + i]) for i in order]
    if st[0] == 0:
        continue
    print(f"  {who}: " + ", ".join(f"{n} {st[i + 1] - st[i]}" for i, n in enumerate(labels)) + f"; total {st[-1] - st[0]} cycles")
    f11, f12, f13, f1 = (int(full[base + i]) for i in (11, 12, 13, 1))
    print(f"    fragments in detail: W1^T / W2^T / head rows LDS -> VGPR + selects {f12 - f11}, biases + value-head row "
          f"{f13 - f12}, Gaussian constants {f1 - f13}")
