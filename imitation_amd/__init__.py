"""imitation_amd -- the GAIL/AIRL adversarial round of HumanCompatibleAI/imitation, rebuilt
MI355X-first: hand-written gfx950 HIP kernels behind a C ABI (`include/imitation_hip.h`,
`imitation_amd/libimitation_hip.so`) driven by a thin Python host layer that keeps the
reference's `AdversarialTrainer / GAIL / AIRL` constructor and `.train()` surface.
"""
from imitation_amd.spaces import Box, Discrete  # noqa: F401
from imitation_amd.vec_env import CountingVecEnv, SyntheticVecEnv, VecEnv, VecEnvWrapper  # noqa: F401
from imitation_amd.data_types import (ExpertIndexStream, Transitions, TransitionsWithRew,  # noqa: F401
                                      TrajectoryWithRew, flatten_trajectories, segment_order,
                                      trajectories_from_legacy_npz)
from imitation_amd.logger import HierarchicalLogger, configure as _configure_logger  # noqa: F401
from imitation_amd.networks import EMANorm, RunningNorm, evaluating, training  # noqa: F401
from imitation_amd.reward_nets import (BasicPotentialMLP, BasicRewardNet, BasicShapedRewardNet,  # noqa: F401
                                       ForwardWrapper, NormalizedRewardNet, PredictProcessedWrapper, RewardNet,
                                       RewardNetWrapper, ShapedRewardNet)
from imitation_amd.policies import (ActorCriticPolicy, FeedForward32Policy, FlattenExtractor,  # noqa: F401
                                    NormalizeFeaturesExtractor)
from imitation_amd.ppo import PPO, OnPolicyAlgorithm, set_random_seed  # noqa: F401
from imitation_amd.buffer import ReplayBuffer  # noqa: F401
from imitation_amd.wrappers import BufferingWrapper, RewardVecEnvWrapper  # noqa: F401
from imitation_amd.adversarial.common import AdversarialTrainer, compute_train_stats  # noqa: F401
from imitation_amd.adversarial.gail import GAIL, RewardNetFromDiscriminatorLogit  # noqa: F401
from imitation_amd.adversarial.airl import AIRL  # noqa: F401
from imitation_amd import bc, checkpoint, cnn_policy, modules, ops, rollout, serialize  # noqa: F401


def configure_logger(folder=None, format_strs=None):
    """`util/logger.py:387-417` `configure`."""
    return _configure_logger(folder, format_strs)
