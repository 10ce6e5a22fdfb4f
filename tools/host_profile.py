"""cProfile of the host side of a few steady-state rounds (config P or a bench variant): where the Python time of a round
goes. Usage: python tools/host_profile.py [variant] [rounds]"""
import cProfile
import os
import pstats
import sys

import torch as th

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

th.set_num_threads(1)
name = sys.argv[1] if len(sys.argv) > 1 else "P"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 10
if name == "P":
    cfg = dict(bench.CFG_P)
    tr = bench.build_trainer(bench.hip_namespace(), cfg, "cuda")
    per = cfg["n_envs"] * cfg["n_steps"]
else:
    tr, per = bench.build_variant(name)
tr.train(4 * per)
th.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
tr.train(rounds * per)
th.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative")
print(f"{name}: {rounds} rounds; times below are totals over them (divide by {rounds})")
st.print_stats(45)
st.sort_stats("tottime")
st.print_stats(25)
