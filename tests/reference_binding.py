"""INTEGRATION.md §2, executable: the reference-side binding a Pearl maintainer would add.

Classes that SUBCLASS THE REFERENCE'S OWN ABCs — `pearl.replay_buffers.ReplayBuffer`
(replay_buffer.py:18-91) and `pearl.policy_learners.policy_learner.PolicyLearner`
(policy_learner.py:40-229) — and forward the hot path to libpearl_amd through the pearl_amd host
classes, so that the REAL `pearl.pearl_agent.PearlAgent` drives the HIP replay arena and the HIP
learners without knowing: `HipReplayBuffer` + `HipDeepQLearning` (config 2), `HipPPOReplayBuffer` +
`HipPPO` (config 4: on-policy — the agent clears the rollout after learn(), pearl_agent.py:217-218,
and PPO's walk over `replay_buffer.memory`, ppo.py:211-293, is the arena's rollout pass) and
`HipContinuousSAC` (config 3, `ActorCriticBase.learn_batch`, actor_critic_base.py:309-366).
Imported only by tests/test_reference_binding.py, and only where the reference is importable
(/root/reference in the build container, oracle/_ref on the GPU box).
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

    impl_type = pearl_amd.BasicReplayBuffer

    def __init__(self, capacity: int, sampler: str = "python") -> None:
        super().__init__()
        self.impl = self.impl_type(capacity, sampler=sampler)

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
        return self.impl.act(subjective_state, available_action_space, exploit=exploit)

    def learn(self, replay_buffer: ReplayBuffer) -> dict:
        inner = replay_buffer.impl if isinstance(replay_buffer, HipReplayBuffer) else replay_buffer
        report = self.impl.learn(inner)           # the fused pa_dqn_learn loop on an arena buffer
        self._training_steps = self.impl._training_steps
        return report

    def learn_batch(self, batch: TransitionBatch) -> dict:
        mine = pearl_amd.TransitionBatch(**{k: getattr(batch, k) for k in _FIELDS
                                            if getattr(batch, k, None) is not None})
        return self.impl.learn_batch(mine)


class HipPPOReplayBuffer(HipReplayBuffer):
    """The reference's on-policy rollout buffer (ppo.py:85, replay_buffer_utils.py:37-128) in the
    arena: gae / lam_return / action_probs are side columns filled by HipPPO.learn's rollout pass."""
    impl_type = pearl_amd.PPOReplayBuffer


class _HipLearner(PolicyLearner):
    """Shared forwarding of a reference-side PolicyLearner whose work runs in `self.impl`."""
    impl: Any

    def set_history_summarization_module(self, value: torch.nn.Module) -> None:
        self._history_summarization_module = value
        self.impl.set_history_summarization_module(value)

    def reset(self, action_space: Any) -> None:
        self.impl.reset(action_space)

    def act(self, subjective_state, available_action_space, exploit: bool = False):
        return self.impl.act(subjective_state, available_action_space, exploit=exploit)

    def learn(self, replay_buffer: ReplayBuffer) -> dict:
        inner = replay_buffer.impl if isinstance(replay_buffer, HipReplayBuffer) else replay_buffer
        report = self.impl.learn(inner)
        self._training_steps = self.impl._training_steps
        return report

    def learn_batch(self, batch: TransitionBatch) -> dict:
        mine = pearl_amd.TransitionBatch(**{k: getattr(batch, k) for k in _FIELDS
                                            if getattr(batch, k, None) is not None})
        return self.impl.learn_batch(mine)


class HipPPO(_HipLearner):
    """`ProximalPolicyOptimization` (ppo.py:88-293) behind the reference's PolicyLearner ABC:
    learn() = rollout pass (two whole-rollout forwards + pa_ppo_gae) + pa_ppo_learn."""

    def __init__(self, state_dim: int, action_space: Any, actor_hidden_dims, critic_hidden_dims,
                 training_rounds: int = 10, batch_size: int = 128, epsilon: float = 0.0,
                 action_representation_module: Any = None, **kw: Any) -> None:
        from pearl.policy_learners.exploration_modules.common.propensity_exploration import (
            PropensityExploration)
        super().__init__(training_rounds=training_rounds, batch_size=batch_size,
                         exploration_module=PropensityExploration(), on_policy=True,
                         is_action_continuous=False,
                         action_representation_module=action_representation_module,
                         action_space=action_space)
        n = action_representation_module.max_number_actions
        self.impl = pearl_amd.ProximalPolicyOptimization(
            state_dim=state_dim, action_space=action_space, actor_hidden_dims=actor_hidden_dims,
            critic_hidden_dims=critic_hidden_dims, training_rounds=training_rounds,
            batch_size=batch_size, epsilon=epsilon,
            action_representation_module=pearl_amd.OneHotActionTensorRepresentationModule(n), **kw)


class HipContinuousSAC(_HipLearner):
    """`ContinuousSoftActorCritic` (soft_actor_critic_continuous.py:50-231) behind the reference's
    PolicyLearner ABC: learn() = pa_sac_learn, learn_batch() = pa_sac_step."""

    def __init__(self, state_dim: int, action_space: Any, actor_hidden_dims, critic_hidden_dims,
                 training_rounds: int = 10, batch_size: int = 128, **kw: Any) -> None:
        from pearl.policy_learners.exploration_modules.common.no_exploration import NoExploration
        super().__init__(training_rounds=training_rounds, batch_size=batch_size,
                         exploration_module=NoExploration(), on_policy=False,
                         is_action_continuous=True, action_space=action_space)
        self.impl = pearl_amd.ContinuousSoftActorCritic(
            state_dim=state_dim, action_space=action_space, actor_hidden_dims=actor_hidden_dims,
            critic_hidden_dims=critic_hidden_dims, training_rounds=training_rounds,
            batch_size=batch_size, **kw)
