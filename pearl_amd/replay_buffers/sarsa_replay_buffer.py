"""SARSAReplayBuffer
(reference: pearl/replay_buffers/sequential_decision_making/sarsa_replay_buffer.py:22-101).

Delays every push until the NEXT action is known: a transition is cached, and stored — with the
following push's action as ``next_action`` — when that push's state equals the cached
``next_state``; terminal / truncated pushes are stored at once with their own action as a dummy
``next_action`` (:88-101).  A cached transition whose successor never arrives is dropped, and the
cache is not cleared by a terminal push or by ``clear()`` — all as in the reference.

Storage is the HBM arena of ``BasicReplayBuffer`` plus one side column for ``next_action`` kept in
the arena's slot order (a ring mirrored on the host: same head / size arithmetic), gathered with
``pa_gather_rows`` at ``sample`` time.
"""
from __future__ import annotations

from typing import Any, Dict, Optional

import torch
from torch import Tensor

from .. import _native as N
from .basic_replay_buffer import BasicReplayBuffer, SideRing
from .transition import TransitionBatch


class SARSAReplayBuffer(BasicReplayBuffer):
    def __init__(self, capacity: int, sampler: str = "device", staging_rows: int = 0) -> None:
        super().__init__(capacity, sampler=sampler, staging_rows=staging_rows)
        self.cache: Optional[Dict[str, Any]] = None
        self._na = SideRing(capacity)      # next actions of the stored rows

    # -- storage of one complete (s, a, r, s', a') -------------------------------------------
    def _store(self, args: Dict[str, Any], next_action: Tensor) -> None:
        BasicReplayBuffer.push(self, **args)
        assert self._arena is not None
        self._na.append(next_action, self._arena.device)

    @staticmethod
    def _as_state(x: Any) -> Tensor:
        return torch.as_tensor(x).detach().to("cpu", torch.float32).reshape(-1)

    def push(self, state: Any, action: Any, reward: Any, terminated: bool, truncated: bool,
             curr_available_actions: Any = None, next_state: Any = None,
             next_available_actions: Any = None, max_number_actions: Optional[int] = None,
             cost: Optional[float] = None) -> None:
        cur_state = self._as_state(state)
        cur_action = torch.as_tensor(action).detach().to("cpu")
        if self.cache is not None and torch.equal(self.cache["_next_state_key"], cur_state):
            done = {k: v for k, v in self.cache.items() if not k.startswith("_")}
            self._store(done, cur_action)                         # a complete SARSA (:55-71)
        args = dict(state=state, action=action, reward=reward, terminated=terminated,
                    truncated=truncated, curr_available_actions=curr_available_actions,
                    next_state=next_state, next_available_actions=next_available_actions,
                    max_number_actions=max_number_actions, cost=cost)
        if not (terminated or truncated):
            self.cache = dict(args, _next_state_key=self._as_state(next_state))   # (:72-86)
        else:
            self._store(args, cur_action)     # terminal: the next action is a dummy (:87-101)

    def clear(self) -> None:
        super().clear()
        self._na.clear()

    def sample(self, batch_size: int) -> TransitionBatch:
        batch = super().sample(batch_size)
        out = self._na.gather(self.last_indices)
        batch.next_action = out.reshape(tuple(batch.action.shape)).to(batch.action.device)
        return batch

    # -- checkpoint / resume: the base class saves the arena columns; the next actions of the
    # stored rows and the pending (not yet stored) transition belong to the state as well
    def state_dict(self) -> Dict[str, Any]:
        sd = super().state_dict()
        sd["next_action"] = self._na.logical()
        sd["sarsa_cache"] = self.cache
        return sd

    def load_state_dict(self, sd: Dict[str, Any]) -> None:
        super().load_state_dict(sd)
        na = sd.get("next_action")
        if na is not None:
            assert self._arena is not None
            self._na.load(na, self._arena.device)
            assert self._na.size == len(self), "next_action column and arena differ in length"
        self.cache = sd.get("sarsa_cache")
