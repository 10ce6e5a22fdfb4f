"""Runs one bench.py variant (default: the AIRL Ant-shaped one) for a few rounds -- for rocprofv3 --kernel-trace."""
import os
import sys

import torch as th

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

th.set_num_threads(1)
if os.environ.get("IA_EPOCH_SPLIT"):   # A/B of the 64-wide epoch kernels (include/imitation_hip.h: ia_ppo_epoch_split)
    from imitation_amd import _lib as L
    L.load().ia_ppo_epoch_split(int(os.environ["IA_EPOCH_SPLIT"]))
if os.environ.get("IA_GRAPH") == "0":   # A/B of the graph-replayed module-policy updates
    from imitation_amd import general_policy
    general_policy.GRAPH_UPDATES = False
if os.environ.get("IA_IMG_OLD"):   # A/B of the NatureCNN policy's launch shapes (round 5's: unsplit linear layer, 2 048-row splits)
    from imitation_amd import cnn_policy
    cnn_policy.ActorCriticCnnPolicy.LINEAR_SPLIT_K = False
    cnn_policy.ActorCriticCnnPolicy.WGRAD_ROWS_PER_SPLIT = 2048
if os.environ.get("IA_RELABEL_LATE"):   # A/B: module reward nets relabel the whole rollout behind its last step
    from imitation_amd import ppo as _ppo
    _init = _ppo.PPO.__init__
    def _late(self, *a, **k):
        _init(self, *a, **k)
        self.relabel_early = False
    _ppo.PPO.__init__ = _late
if os.environ.get("IA_ACT_COPY"):   # A/B: the image act step's frames through a device copy (before: zero-copy reads of the pinned row)
    from imitation_amd import cnn_policy as _cp
    _cp.ActorCriticCnnPolicy.ACT_ZERO_COPY = False
if os.environ.get("IA_REDUCE_EACH"):   # A/B: the NatureCNN policy's slab reductions one launch per piece
    from imitation_amd import cnn_policy as _cp2
    _cp2.ActorCriticCnnPolicy.REDUCE_IN_ONE_LAUNCH = False
name = sys.argv[1] if len(sys.argv) > 1 else "3_airl_ant_1024x16_mb1024"
print(name, bench.run_variant(name, rounds=int(sys.argv[2]) if len(sys.argv) > 2 else 6))
