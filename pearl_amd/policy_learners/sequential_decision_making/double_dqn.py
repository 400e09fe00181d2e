"""DoubleDQN (reference: pearl/policy_learners/sequential_decision_making/double_dqn.py:20-76).

The reference subclass overrides exactly one method, ``get_next_state_values``: the next action
a' is the ONLINE network's argmax over the available next actions (unavailable ones at -inf,
``torch.max(1)[1]`` = first maximum) and its value comes from the TARGET network,
Q_target(s', a').  Everything else — replay, preprocessing, loss, AdamW(amsgrad), target soft
updates, the fused ``learn()`` loop and its data-parallel form — is DeepQLearning's.

Here the rule is a field of the native learner's descriptor (``pa_dqn_desc.double_q``,
include/pearl_amd.h): ``pa_dqn_learn`` / ``pa_dqn_step`` / ``pa_dqn_qvalues`` then run the
all-actions pass of ``target_fused_kernel`` on the online parameters to pick a', and one row per
transition through the target network to value it.  The choice depends on the online parameters,
which move every round, so this learner's ``learn()`` cannot batch the pass over a window of rounds
or overlap it with the online chain the way DeepQLearning's does.
"""
from __future__ import annotations

from ..policy_learner import PolicyLearner
from .deep_q_learning import DeepQLearning


class DoubleDQN(DeepQLearning):
    _double_q = 1

    def compare(self, other: PolicyLearner) -> str:
        diffs = [super().compare(other)]
        if not isinstance(other, DoubleDQN):
            diffs.append("other is not an instance of DoubleDQN")
        return "\n".join(d for d in diffs if d)
