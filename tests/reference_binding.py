"""INTEGRATION.md §2, executable: the reference-side binding a Pearl maintainer would add.

Two classes that SUBCLASS THE REFERENCE'S OWN ABCs — `pearl.replay_buffers.ReplayBuffer`
(replay_buffer.py:18-91) and `pearl.policy_learners.policy_learner.PolicyLearner`
(policy_learner.py:40-229) — and forward the hot path to libpearl_amd through the pearl_amd host
classes, so that the REAL `pearl.pearl_agent.PearlAgent` drives the HIP replay arena and the HIP
DQN learner without knowing.  Imported only by tests/test_reference_binding.py, and only where the
reference is importable (it does not travel to the GPU box).
"""
from typing import Any, Optional

import torch
from pearl.policy_learners.exploration_modules.common.epsilon_greedy_exploration import EGreedyExploration
from pearl.policy_learners.policy_learner import PolicyLearner
from pearl.replay_buffers.replay_buffer import ReplayBuffer
from pearl.replay_buffers.transition import TransitionBatch

import pearl_amd

_FIELDS = ("state", "action", "reward", "terminated", "truncated", "next_state", "next_action",
           "curr_available_actions", "curr_unavailable_actions_mask", "next_available_actions",
           "next_unavailable_actions_mask", "weight", "cost")


class HipReplayBuffer(ReplayBuffer):
    """`ReplayBuffer` of the reference, stored in the MI355X arena."""

    def __init__(self, capacity: int, sampler: str = "python") -> None:
        super().__init__()
        self.impl = pearl_amd.BasicReplayBuffer(capacity, sampler=sampler)

    @property
    def device_for_batches(self) -> torch.device:
        return self.impl.device_for_batches

    @device_for_batches.setter
    def device_for_batches(self, new_device_for_batches: torch.device) -> None:
        self.impl.device_for_batches = new_device_for_batches

    def push(self, state, action, reward, terminated, truncated, curr_available_actions=None,
             next_state=None, next_available_actions=None, max_number_actions=None, cost=None) -> None:
        self.impl._is_action_continuous = self._is_action_continuous
        # the reference's DiscreteActionSpace already offers .n / .action_dim / .actions_batch
        self.impl.push(state, action, reward, terminated, truncated, curr_available_actions,
                       next_state, next_available_actions, max_number_actions, cost)

    def sample(self, batch_size: int) -> TransitionBatch:
        b = self.impl.sample(batch_size)          # ValueError when batch_size > len, like the reference
        return TransitionBatch(**{k: getattr(b, k) for k in _FIELDS if getattr(b, k, None) is not None})

    def clear(self) -> None:
        self.impl.clear()

    def __len__(self) -> int:
        return len(self.impl)


class HipDeepQLearning(PolicyLearner):
    """`PolicyLearner` of the reference whose learn() / learn_batch() run in libpearl_amd."""

    def __init__(self, state_dim: int, action_space: Any, hidden_dims, training_rounds: int = 10,
                 batch_size: int = 128, action_representation_module: Any = None, **kw: Any) -> None:
        super().__init__(training_rounds=training_rounds, batch_size=batch_size,
                         exploration_module=EGreedyExploration(0.05), on_policy=False,
                         is_action_continuous=False,
                         action_representation_module=action_representation_module,
                         action_space=action_space)
        n = action_representation_module.max_number_actions
        self.impl = pearl_amd.DeepQLearning(
            state_dim=state_dim, action_space=action_space, hidden_dims=hidden_dims,
            training_rounds=training_rounds, batch_size=batch_size,
            action_representation_module=pearl_amd.OneHotActionTensorRepresentationModule(n), **kw)

    def set_history_summarization_module(self, value: torch.nn.Module) -> None:
        self._history_summarization_module = value
        self.impl.set_history_summarization_module(value)

    def reset(self, action_space: Any) -> None:
        self.impl.reset(action_space)

    def act(self, subjective_state, available_action_space, exploit: bool = False):
        return self.impl.act(subjective_state, available_action_space, exploit=True)

    def learn(self, replay_buffer: ReplayBuffer) -> dict:
        inner = replay_buffer.impl if isinstance(replay_buffer, HipReplayBuffer) else replay_buffer
        report = self.impl.learn(inner)           # the fused pa_dqn_learn loop on an arena buffer
        self._training_steps = self.impl._training_steps
        return report

    def learn_batch(self, batch: TransitionBatch) -> dict:
        mine = pearl_amd.TransitionBatch(**{k: getattr(batch, k) for k in _FIELDS
                                            if getattr(batch, k, None) is not None})
        return self.impl.learn_batch(mine)
