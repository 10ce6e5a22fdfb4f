"""Host cost of one SyntheticVecEnv step at config-P shapes with and without the helper-thread
generator prefetch, with a stand-in for the rest of the rollout step between env steps (some Python
work holding the GIL, then a GIL-releasing wait like the device synchronisation).
Usage: python tools/env_step_profile.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imitation_amd.vec_env import SyntheticVecEnv  # noqa: E402

rng = np.random.default_rng(0)
t0 = time.perf_counter()
for _ in range(200):
    rng.standard_normal((1024, 17))
print(f"generator fill 1024x17: {1e6 * (time.perf_counter() - t0) / 200:.1f} us")
acts = np.random.default_rng(5).uniform(-1, 1, (1024, 6)).astype(np.float32)


def spin(us):
    end = time.perf_counter() + us * 1e-6
    while time.perf_counter() < end:
        pass


for pref in (False, True):
    for gil_us, wait_us in ((40, 40), (80, 0), (0, 80)):
        env = SyntheticVecEnv(1024, 17, 6, 1000, 0, prefetch_noise=pref)
        env.reset()
        ts = []
        for i in range(300):
            t = time.perf_counter()
            env.step_async(acts)
            env.step_wait_arrays()
            ts.append(time.perf_counter() - t)
            spin(gil_us)
            if wait_us:
                time.sleep(wait_us * 1e-6)
        print(f"prefetch={pref!s:5s} between steps: {gil_us:3d} us Python + {wait_us:3d} us GIL-free wait -> "
              f"env step median {1e6 * np.median(ts):6.1f} us")
