"""cProfile of PPO.collect_rollouts + pop/store at config P (host side of the rollout)."""
import cProfile, os, pstats, sys, time
import torch as th
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
th.set_num_threads(1)
cfg = dict(bench.CFG_P)
tr = bench.build_trainer(bench.hip_namespace(), cfg, "cuda")
tr.train(2 * 16384)
algo = tr.gen_algo
cb = algo._init_callback(tr.gen_callback)
def one():
    algo.collect_rollouts(algo.env, cb, algo.rollout_buffer, algo.n_steps)
    gs, lens = tr.venv_buffering.pop_transitions_and_lens()
    tr._gen_replay_buffer.store(gs)
for _ in range(3): one()
th.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(10): one()
th.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
