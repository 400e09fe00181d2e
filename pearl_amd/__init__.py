"""pearl_amd — MI355X-native learner/replay core behind Pearl's plugin API.

Scope: ``BasicReplayBuffer.sample()`` and ``PolicyLearner.learn()`` for DQN, PPO and
continuous SAC as
hand-written HIP kernels (libpearl_amd.so, C ABI in include/pearl_amd.h) behind Python classes
that mirror the reference's ``ReplayBuffer`` / ``PolicyLearner`` / ``PearlAgent`` interfaces.
There is no CPU or PyTorch fallback for that path: it fails loudly without the library / a GPU.
"""
from .pearl_agent import PearlAgent  # noqa: F401
from .replay_buffers import (BasicReplayBuffer, BootstrapReplayBuffer,  # noqa: F401
                             HindsightExperienceReplayBuffer, SARSAReplayBuffer, TransitionBatch,
                             TransitionWithBootstrapMaskBatch, filter_batch_by_bootstrap_mask)
from .policy_learners.sequential_decision_making import (TD3,  # noqa: F401
                                                         ContinuousSoftActorCritic,
                                                         DeepDeterministicPolicyGradient,
                                                         DeepQLearning, DeepSARSA, DoubleDQN,
                                                         ImplicitQLearning,
                                                         PPOReplayBuffer,
                                                         ProximalPolicyOptimization,
                                                         SoftActorCritic)
from .policy_learners.contextual_bandits import (NeuralLinearBandit, SquareCBExploration,  # noqa: F401
                                                 UCBExploration)
from .action_representation_modules import OneHotActionTensorRepresentationModule  # noqa: F401
from .utils.instantiations.spaces import BoxActionSpace, DiscreteActionSpace  # noqa: F401
from .vector_env import BatchedActionResult, BatchedEnvironment, VectorEnvFeeder  # noqa: F401

__version__ = "0.1.0"
