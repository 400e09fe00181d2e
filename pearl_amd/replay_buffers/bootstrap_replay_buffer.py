"""BootstrapReplayBuffer
(reference: pearl/replay_buffers/sequential_decision_making/bootstrap_replay_buffer.py:23-114;
Osband et al. 2016, Bootstrapped DQN, Bernoulli(p) masking distribution).

Every push draws one mask ``w ~ Bernoulli(p)^ensemble_size`` (``w_k = 1``: ensemble member k
trains on this transition) and stores it with the transition; ``sample`` returns a
``TransitionWithBootstrapMaskBatch`` whose ``bootstrap_mask`` is (batch_size, ensemble_size).

The draw is the reference's own — ``torch.bernoulli(torch.tensor(p).repeat(1, K))`` on torch's
global CPU generator, once per push (:64-66) — so a seeded run stores bit-identical masks.  The
transition goes to the HBM arena like any ``BasicReplayBuffer`` row; the mask is one side column in
HBM (``SideRing``: the arena's FIFO arithmetic, gathered by ``pa_gather_rows`` with the indices the
arena's own sampler drew).  Like the reference (:86-114) the returned batch carries no ``cost``.
"""
from __future__ import annotations

from typing import Any, Dict, Optional

import torch

from .basic_replay_buffer import BasicReplayBuffer, SideRing
from .transition import TransitionWithBootstrapMaskBatch


class BootstrapReplayBuffer(BasicReplayBuffer):
    def __init__(self, capacity: int, p: float, ensemble_size: int, sampler: str = "device",
                 staging_rows: int = 0) -> None:
        super().__init__(capacity, sampler=sampler, staging_rows=staging_rows)
        self.p = p
        self.ensemble_size = ensemble_size
        self._masks = SideRing(capacity)

    def push(self, state: Any, action: Any, reward: Any, terminated: bool, truncated: bool,
             curr_available_actions: Any = None, next_state: Any = None,
             next_available_actions: Any = None, max_number_actions: Optional[int] = None,
             cost: Optional[float] = None) -> None:
        super().push(state, action, reward, terminated, truncated, curr_available_actions,
                     next_state, next_available_actions, max_number_actions, cost)
        # the reference draws inside _store_transition, i.e. after the tensorisation of the push
        # arguments and before the append (:64-66): same position in the global RNG stream
        probs = torch.tensor(self.p).repeat(1, self.ensemble_size)
        assert self._arena is not None
        self._masks.append(torch.bernoulli(probs), self._arena.device)

    def clear(self) -> None:
        super().clear()
        self._masks.clear()

    def sample(self, batch_size: int) -> TransitionWithBootstrapMaskBatch:
        b = super().sample(batch_size)       # ValueError when batch_size > len (:87-91)
        mask = self._masks.gather(self.last_indices).to(b.state.device)
        return TransitionWithBootstrapMaskBatch(
            state=b.state, action=b.action, reward=b.reward, terminated=b.terminated,
            truncated=b.truncated, next_state=b.next_state,
            curr_available_actions=b.curr_available_actions,
            curr_unavailable_actions_mask=b.curr_unavailable_actions_mask,
            next_available_actions=b.next_available_actions,
            next_unavailable_actions_mask=b.next_unavailable_actions_mask, bootstrap_mask=mask)

    def state_dict(self) -> Dict[str, Any]:
        sd = super().state_dict()
        sd["bootstrap_mask"] = self._masks.logical()
        sd["p"], sd["ensemble_size"] = self.p, self.ensemble_size
        return sd

    def load_state_dict(self, sd: Dict[str, Any]) -> None:
        super().load_state_dict(sd)
        m = sd.get("bootstrap_mask")
        if m is not None:
            assert self._arena is not None and m.shape[1] == self.ensemble_size
            self._masks.load(m, self._arena.device)
            assert self._masks.size == len(self)
