"""Runs one bench.py variant (default: the AIRL Ant-shaped one) for a few rounds -- for rocprofv3 --kernel-trace."""
import os
import sys

import torch as th

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

th.set_num_threads(1)
name = sys.argv[1] if len(sys.argv) > 1 else "3_airl_ant_1024x16_mb1024"
print(name, bench.run_variant(name, rounds=int(sys.argv[2]) if len(sys.argv) > 2 else 6))
