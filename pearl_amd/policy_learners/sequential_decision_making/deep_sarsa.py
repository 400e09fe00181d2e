"""DeepSARSA (reference: pearl/policy_learners/sequential_decision_making/deep_sarsa.py:30-97).

On-policy TD learning: the next state is valued by ``Q_target(s', a')`` for the next action the
agent actually committed to (``batch.next_action``, provided by ``SARSAReplayBuffer``), instead of a
max over the available actions.  Everything else is DeepTDLearning's: loss, AdamW(amsgrad), target
soft updates.

Here that is rule 2 of the native learner (``pa_dqn_desc.double_q == 2``): the fused target kernel
runs with ONE action per transition — the committed one.  ``learn()`` goes through the reference's
generic ``sample -> preprocess_batch -> learn_batch`` loop (``pa_dqn_step``): the committed next
action is a column of the SARSA buffer, not of the arena the fused ``pa_dqn_learn`` loop gathers
from.
"""
from __future__ import annotations

from typing import Any

from ..policy_learner import PolicyLearner
from .deep_q_learning import DeepQLearning


class DeepSARSA(DeepQLearning):
    _double_q = 2

    def __init__(self, state_dim: Any = None, action_space: Any = None,
                 exploration_module: Any = None, action_representation_module: Any = None,
                 optimizer: Any = None, **kwargs: Any) -> None:
        # DeepSARSA takes DeepTDLearning's defaults, not DeepQLearning's
        # (deep_td_learning.py:58-63: training_rounds 100, batch_size 128, soft_update_tau 0.1)
        kwargs.setdefault("training_rounds", 100)
        kwargs.setdefault("batch_size", 128)
        kwargs.setdefault("soft_update_tau", 0.1)
        super().__init__(state_dim=state_dim, action_space=action_space,
                         exploration_module=exploration_module,
                         action_representation_module=action_representation_module,
                         optimizer=optimizer, **kwargs)
        self.on_policy = True       # deep_sarsa.py:52

    def compare(self, other: PolicyLearner) -> str:
        diffs = [super().compare(other)]
        if not isinstance(other, DeepSARSA):
            diffs.append("other is not an instance of DeepSARSA")
        return "\n".join(d for d in diffs if d)
