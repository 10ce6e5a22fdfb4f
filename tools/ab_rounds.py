"""Same-box A/B of a scheduling / kernel-choice switch on whole rounds: builds the trainer afresh for each setting,
alternating, and times `rounds` rounds of `train()` after a warm-up.
Usage: python tools/ab_rounds.py <bench variant | P> <attribute>=<a>,<b>[,<c>] [rounds] [repeats]
  attribute: an attribute of the trainer (`disc_behind_ppo`, `disc_round_one_call`, `disc_enqueue_early`) or, with the prefix
  `gen.`, of its PPO (`gen.epochs_one_call`), or `lib.<toggle>`: an integer switch of the library
  (`lib.ia_ppo_update_xcd_pack`, `lib.ia_ppo_epoch_split`, `lib.ia_disc_fused_side_reduce`); values: None / True / False / numbers.
  e.g.  python tools/ab_rounds.py P disc_behind_ppo=None,True,False 100
        python tools/ab_rounds.py P_mlp64_1024x16 gen.epochs_one_call=True,False 60 3"""
import ast
import os
import sys
import time

import torch as th

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

th.set_num_threads(1)
name = sys.argv[1]
attr, vals = sys.argv[2].split("=")
vals = [ast.literal_eval(v) for v in vals.split(",")]
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 60
repeats = int(sys.argv[4]) if len(sys.argv) > 4 else 2
for v in vals * repeats:
    if name == "P":
        cfg = dict(bench.CFG_P)
        tr, per = bench.build_trainer(bench.hip_namespace(), cfg, "cuda"), cfg["n_envs"] * cfg["n_steps"]
    else:
        tr, per = bench.build_variant(name)
    if attr.startswith("lib."):
        from imitation_amd import _lib as L
        getattr(L.load(), attr[4:])(int(v))
    else:
        obj, a = (tr.gen_algo, attr[4:]) if attr.startswith("gen.") else (tr, attr)
        assert hasattr(obj, a), attr
        setattr(obj, a, v)
    tr.train(5 * per)
    th.cuda.synchronize()
    t = time.perf_counter()
    tr.train(rounds * per)
    th.cuda.synchronize()
    dt = time.perf_counter() - t
    print(f"{name}: {attr}={v}: {1e3 * dt / rounds:.3f} ms/round = {rounds * per / dt / 1e6:.3f} M env-steps/s "
          f"(discriminator updates {'behind' if tr._disc_mode_behind else 'beside'} the PPO update at the end)", flush=True)
