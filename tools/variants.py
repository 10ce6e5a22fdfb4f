"""Runs the other BASELINE.json / SURVEY 8d configurations through the HIP trainer for a few rounds:
sanity (finite statistics, counters) + throughput. Usage: python tools/variants.py [rounds] [names]"""
import os, sys, tempfile, time
import numpy as np, torch as th
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import imitation_amd as p
from imitation_amd.vec_env import SyntheticVecEnv

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
only = set(sys.argv[2].split(",")) if len(sys.argv) > 2 else None   # e.g. "P,3b"
th.set_num_threads(1)


def demos(n, od, ad, discrete, seed=1):
    rng = np.random.default_rng(seed)
    obs = rng.standard_normal((n, od)).astype(np.float32)
    acts = rng.integers(0, ad, n).astype(np.int64) if discrete else rng.uniform(-1, 1, (n, ad)).astype(np.float32)
    return p.Transitions(obs=obs, acts=acts, next_obs=(0.9 * obs).astype(np.float32), dones=np.zeros(n, bool))


def run(name, algo_name, n_envs, n_steps, od, ad, horizon, ppo_batch, n_epochs, demo_batch, n_disc, capacity,
        disc_kw, discrete=False, gamma=0.99, gae=0.95, clip=0.2, minib=None, normalize_output=False):
    if only is not None and name.split()[0] not in only:
        return True
    th.manual_seed(0); np.random.seed(0)
    venv = SyntheticVecEnv(num_envs=n_envs, obs_dim=od, act_dim=ad, horizon=horizon, seed=0,
                           n_discrete=ad if discrete else None)
    pk = dict(features_extractor_class=p.NormalizeFeaturesExtractor, features_extractor_kwargs=dict(normalize_class=p.RunningNorm))
    algo = p.PPO(p.FeedForward32Policy, venv, n_steps=n_steps, batch_size=ppo_batch, n_epochs=n_epochs, ent_coef=0.01,
                 gamma=gamma, gae_lambda=gae, clip_range=clip, seed=0, policy_kwargs=pk, device="cuda")
    if algo_name == "gail":
        net = p.BasicRewardNet(venv.observation_space, venv.action_space, normalize_input_layer=p.RunningNorm, **disc_kw)
        cls = p.GAIL
    else:
        net = p.BasicShapedRewardNet(venv.observation_space, venv.action_space, normalize_input_layer=p.RunningNorm, **disc_kw)
        if normalize_output:
            net = p.NormalizedRewardNet(net, p.RunningNorm)
        cls = p.AIRL
    tr = cls(demonstrations=demos(max(4 * demo_batch, 20000), od, ad, discrete), demo_batch_size=demo_batch,
             demo_minibatch_size=minib, venv=venv, gen_algo=algo, reward_net=net, n_disc_updates_per_round=n_disc,
             gen_replay_buffer_capacity=capacity, custom_logger=p.configure_logger(tempfile.mkdtemp(), []))
    per = n_envs * n_steps
    tr.train(3 * per)  # warm-up rounds (lazily created pinned buffers / streams appear during the first three)
    th.cuda.synchronize(); t0 = time.perf_counter()
    tr.train(rounds * per)   # (no hook on train_disc: an overridden train_disc switches round pipelining off)
    th.cuda.synchronize(); dt = time.perf_counter() - t0
    ok = tr._disc_step == (rounds + 3) * n_disc
    last = tr.train_disc()
    ok = ok and all(np.isfinite(v) for k, v in last.items())
    sd = tr.gen_algo.policy.state_dict()
    ok = ok and all(bool(th.isfinite(v.float()).all()) for v in sd.values())
    print(f"{name:34s} {rounds * per / dt / 1e3:9.1f} k env-steps/s  {1e3 * dt / rounds:8.2f} ms/round  "
          f"disc_loss={last['disc_loss']:.4f} acc={last['disc_acc']:.3f}  {'OK' if ok else 'FAIL'}")
    return ok


allok = True
allok &= run("P   GAIL HalfCheetah 1024x16", "gail", 1024, 16, 17, 6, 1000, 1024, 10, 8192, 16, 16384, dict(hid_sizes=(256, 256)))
allok &= run("H   GAIL horizon 1024x1000", "gail", 1024, 1000, 17, 6, 1000, 1024, 2, 8192, 4, 16384, dict(hid_sizes=(256, 256)))
allok &= run("T   GAIL tuned 1024x4 mb64", "gail", 1024, 4, 17, 6, 1000, 64, 5, 8192, 8, 512, dict(hid_sizes=(256, 256)), gamma=0.95, clip=0.1)
allok &= run("3   AIRL Ant 1024x8 mb16", "airl", 1024, 8, 27, 8, 1000, 16, 2, 8192, 16, 8192,
             dict(reward_hid_sizes=(32,), potential_hid_sizes=(32, 32)), gamma=0.995, gae=0.8, normalize_output=True)
allok &= run("3b  AIRL Ant 1024x16 mb1024", "airl", 1024, 16, 27, 8, 1000, 1024, 10, 8192, 16, 16384,
             dict(reward_hid_sizes=(32,), potential_hid_sizes=(32, 32)), normalize_output=True)
allok &= run("1   GAIL CartPole-shaped 8x256", "gail", 8, 256, 4, 2, 500, 64, 5, 1024, 4, 2048, dict(hid_sizes=(32, 32)), discrete=True)
allok &= run("acc GAIL minibatch 2048 of 8192", "gail", 1024, 16, 17, 6, 1000, 1024, 10, 8192, 4, 16384, dict(hid_sizes=(256, 256)), minib=2048)
print("ALL OK" if allok else "SOME FAILED")
