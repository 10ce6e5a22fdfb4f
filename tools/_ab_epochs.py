import os, sys, time
import torch as th
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
th.set_num_threads(1)
name = sys.argv[1] if len(sys.argv) > 1 else "P_mlp64_1024x16"
for one in (True, False) * 3:
    tr, per = bench.build_variant(name)
    tr.gen_algo.epochs_one_call = one
    tr.train(5 * per); th.cuda.synchronize()
    t = time.perf_counter(); tr.train(60 * per); th.cuda.synchronize()
    dt = time.perf_counter() - t
    print(f"{name}: epochs_one_call={one}: {1e3 * dt / 60:.3f} ms/round = {60 * per / dt / 1e6:.3f} M; behind={tr._disc_mode_behind}", flush=True)
