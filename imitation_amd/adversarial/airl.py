"""Adversarial Inverse Reinforcement Learning (`algorithms/adversarial/airl.py`)."""
from __future__ import annotations

from typing import Optional

import torch as th

from imitation_amd import reward_nets
from imitation_amd.adversarial import common
from imitation_amd.cnn_policy import ActorCriticCnnPolicy
from imitation_amd.policies import ActorCriticPolicy

STOCHASTIC_POLICIES = (ActorCriticPolicy, ActorCriticCnnPolicy)   # (`airl.py:11`: SAC / actor-critic policies)


class AIRL(common.AdversarialTrainer):
    """`airl.py:15-132`: logits = f_theta(s,a,s') - log pi(a|s)."""

    _needs_logp = True

    def __init__(self, *, demonstrations, demo_batch_size: int, venv, gen_algo, reward_net: reward_nets.RewardNet,
                 **kwargs):
        super().__init__(demonstrations=demonstrations, demo_batch_size=demo_batch_size, venv=venv,
                         gen_algo=gen_algo, reward_net=reward_net, **kwargs)
        if not isinstance(self.gen_algo.policy, STOCHASTIC_POLICIES):
            raise TypeError("AIRL needs a stochastic policy to compute the discriminator output.")

    def logits_expert_is_high(self, state, action, next_state, done,
                              log_policy_act_prob: Optional[th.Tensor] = None) -> th.Tensor:
        if log_policy_act_prob is None:
            raise TypeError("Non-None `log_policy_act_prob` is required for this method.")
        return self._reward_net(state, action, next_state, done) - log_policy_act_prob.to(self._device)

    @property
    def reward_train(self) -> reward_nets.RewardNet:
        return self._reward_net

    @property
    def reward_test(self) -> reward_nets.RewardNet:
        net = self._reward_net
        while isinstance(net, reward_nets.RewardNetWrapper) or (isinstance(net, th.nn.Module) and hasattr(net, "base")):
            net = net.base
        return net
