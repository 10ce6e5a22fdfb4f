"""Bit fingerprint of a few training rounds (config P or a bench variant): sha256 over the generator's and the reward net's state after `rounds` rounds from the bench's fixed seeds. Two libraries (IA_LIB=...) that print the same line compute
the same bits -- the check behind every "same operations, moved" kernel change (tools/ab_libs.sh builds the libraries).
Usage: python tools/ppo_bits.py [rounds] [bench variant]"""
import hashlib
import os
import sys

import torch as th

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
th.set_num_threads(1)
if len(sys.argv) > 2:
    tr, per_round = bench.build_variant(sys.argv[2])
else:
    cfg = dict(bench.CFG_P)
    tr = bench.build_trainer(bench.hip_namespace(), cfg, "cuda")
    per_round = cfg["n_envs"] * cfg["n_steps"]
tr.train(rounds * per_round)
th.cuda.synchronize()
h = hashlib.sha256()
pol = tr.gen_algo.policy
for name, t in sorted(pol.state_dict().items()):
    h.update(name.encode())
    h.update(t.detach().cpu().contiguous().numpy().tobytes())
hr = hashlib.sha256()
for name, t in sorted(tr._reward_net.state_dict().items()):   # the discriminator's parameters and running statistics
    hr.update(name.encode())
    hr.update(t.detach().cpu().contiguous().numpy().tobytes())
print((sys.argv[2] if len(sys.argv) > 2 else "P"), f"rounds={rounds}", "policy sha256", h.hexdigest()[:24],
      "reward net sha256", hr.hexdigest()[:24])
