from .actor_critic_base import ActorCriticBase
from .deep_q_learning import DeepQLearning
from .double_dqn import DoubleDQN
from .ppo import PPOReplayBuffer, PPOTransitionBatch, ProximalPolicyOptimization
from .soft_actor_critic_continuous import ContinuousSoftActorCritic

__all__ = ["ActorCriticBase", "DeepQLearning", "DoubleDQN", "PPOReplayBuffer", "PPOTransitionBatch",
           "ProximalPolicyOptimization", "ContinuousSoftActorCritic"]
