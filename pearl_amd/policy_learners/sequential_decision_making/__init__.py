from .actor_critic_base import ActorCriticBase
from .deep_q_learning import DeepQLearning
from .deep_sarsa import DeepSARSA
from .ddpg import DeepDeterministicPolicyGradient
from .double_dqn import DoubleDQN
from .implicit_q_learning import ImplicitQLearning
from .ppo import PPOReplayBuffer, PPOTransitionBatch, ProximalPolicyOptimization
from .soft_actor_critic import SoftActorCritic
from .soft_actor_critic_continuous import ContinuousSoftActorCritic
from .td3 import TD3

__all__ = ["ActorCriticBase", "DeepDeterministicPolicyGradient", "TD3", "DeepQLearning", "DeepSARSA", "DoubleDQN", "ImplicitQLearning", "PPOReplayBuffer", "PPOTransitionBatch",
           "ProximalPolicyOptimization", "ContinuousSoftActorCritic", "SoftActorCritic"]
