"""Act-time exploration used by the value-based learners (outside the learner hot path).

``EGreedyExploration`` follows pearl/policy_learners/exploration_modules/common/
epsilon_greedy_exploration.py:28-102: optional linear epsilon warm-up, one ``random.random()``
draw per ``act`` from Python's global stream (shared with the replay buffer's python sampler,
SURVEY.md §7 "sampling parity").
"""
from __future__ import annotations

import random
from typing import Any, Optional

import torch
import torch.nn as nn


class ExplorationModule(nn.Module):
    def reset(self) -> None:
        pass

    def act(self, subjective_state: Any, action_space: Any, exploit_action: Any = None,
            values: Optional[torch.Tensor] = None, **kwargs: Any) -> Any:
        raise NotImplementedError

    def learn(self, replay_buffer: Any) -> None:
        pass

    def compare(self, other: "ExplorationModule") -> str:
        return "" if type(self) is type(other) else (
            f"exploration module types differ: {type(self).__name__} vs {type(other).__name__}")


class NoExploration(ExplorationModule):
    def act(self, subjective_state: Any, action_space: Any, exploit_action: Any = None,
            values: Optional[torch.Tensor] = None, **kwargs: Any) -> Any:
        if exploit_action is not None:
            return exploit_action
        assert values is not None
        return action_space.actions[int(torch.argmax(values))]


class PropensityExploration(ExplorationModule):
    """Sample an action from the policy's own distribution
    (pearl/policy_learners/exploration_modules/common/propensity_exploration.py:25-52)."""

    def act(self, subjective_state: Any, action_space: Any, exploit_action: Any = None,
            values: Optional[torch.Tensor] = None, **kwargs: Any) -> Any:
        assert values is not None, "PropensityExploration needs the action probabilities"
        idx = int(torch.distributions.Categorical(values).sample())
        return action_space.actions[idx]


class NormalDistributionExploration(ExplorationModule):
    """Gaussian noise on a continuous action, scaled to the action box and clipped
    (pearl/policy_learners/exploration_modules/common/normal_distribution_exploration.py:24-78)."""

    def __init__(self, mean: float = 0.0, std_dev: float = 1.0) -> None:
        super().__init__()
        self._mean, self._std_dev = mean, std_dev

    def act(self, subjective_state: Any = None, action_space: Any = None, exploit_action: Any = None,
            values: Optional[torch.Tensor] = None, **kwargs: Any) -> Any:
        assert exploit_action is not None and hasattr(action_space, "low")
        dev = exploit_action.device
        low = action_space.low.detach().clone().to(dev)
        high = action_space.high.detach().clone().to(dev)
        assert torch.all(exploit_action >= low) and torch.all(exploit_action <= high)
        noise = torch.normal(mean=self._mean, std=self._std_dev, size=exploit_action.size(),
                             device=dev)
        return torch.clamp(exploit_action + ((high - low) / 2) * noise, low, high)

    def compare(self, other: ExplorationModule) -> str:
        if not isinstance(other, NormalDistributionExploration):
            return "other is not an instance of NormalDistributionExploration"
        return "\n".join(f"{k} is different: {getattr(self, k)} vs {getattr(other, k)}"
                         for k in ("_mean", "_std_dev") if getattr(self, k) != getattr(other, k))


class EGreedyExploration(ExplorationModule):
    def __init__(self, epsilon: float, start_epsilon: Optional[float] = None,
                 end_epsilon: Optional[float] = None, warmup_steps: Optional[int] = None) -> None:
        super().__init__()
        self.start_epsilon, self.end_epsilon, self.warmup_steps = (start_epsilon, end_epsilon,
                                                                   warmup_steps)
        self.time_step = 0
        self._epsilon_scheduling = None not in (start_epsilon, end_epsilon, warmup_steps)
        self.curr_epsilon: float = start_epsilon if self._epsilon_scheduling else epsilon

    def act(self, subjective_state: Any, action_space: Any, exploit_action: Any = None,
            values: Optional[torch.Tensor] = None, action_availability_mask: Any = None,
            **kwargs: Any) -> Any:
        if self._epsilon_scheduling and self.time_step < self.warmup_steps:
            frac = self.time_step / self.warmup_steps
            self.curr_epsilon = self.start_epsilon + (self.end_epsilon - self.start_epsilon) * frac
        self.time_step += 1
        if exploit_action is None:
            raise ValueError("exploit_action cannot be None for epsilon-greedy exploration")
        if not hasattr(action_space, "actions_batch"):
            raise TypeError("action space must be discrete")
        if random.random() < self.curr_epsilon:
            return action_space.sample(action_availability_mask).to(exploit_action.device)
        return exploit_action

    def get_extra_state(self) -> dict:
        return {k: getattr(self, k) for k in ("start_epsilon", "curr_epsilon", "end_epsilon",
                                              "time_step", "warmup_steps")} | {
            "epsilon_scheduling": self._epsilon_scheduling}

    def set_extra_state(self, state: Any) -> None:
        for k in ("start_epsilon", "curr_epsilon", "end_epsilon", "time_step", "warmup_steps"):
            setattr(self, k, state[k])
        self._epsilon_scheduling = state["epsilon_scheduling"]

    def compare(self, other: ExplorationModule) -> str:
        if not isinstance(other, EGreedyExploration):
            return "other is not an instance of EGreedyExploration"
        keys = ("start_epsilon", "end_epsilon", "time_step", "_epsilon_scheduling", "warmup_steps",
                "curr_epsilon")
        return "\n".join(f"{k} is different: {getattr(self, k)} vs {getattr(other, k)}"
                         for k in keys if getattr(self, k) != getattr(other, k))
