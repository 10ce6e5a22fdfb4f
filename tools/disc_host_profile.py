"""cProfile of the host side of `train_disc` for one bench.py variant. Usage: python tools/disc_host_profile.py [variant]"""
import cProfile
import os
import pstats
import sys

import torch as th

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from imitation_amd import networks  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "3_airl_ant_1024x16"
th.set_num_threads(1)
tr, per_round = bench.build_variant(name)
tr.train(3 * per_round)
th.cuda.synchronize()
pr = cProfile.Profile()
with networks.training(tr.reward_train):
    for _ in range(8):
        tr.train_disc()
    th.cuda.synchronize()
    pr.enable()
    for _ in range(64):
        tr.train_disc()
    pr.disable()
th.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
