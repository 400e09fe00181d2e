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
from .basic_replay_buffer import BasicReplayBuffer
from .transition import TransitionBatch


class SARSAReplayBuffer(BasicReplayBuffer):
    def __init__(self, capacity: int, sampler: str = "device", staging_rows: int = 0) -> None:
        super().__init__(capacity, sampler=sampler, staging_rows=staging_rows)
        self.cache: Optional[Dict[str, Any]] = None
        self._na: Optional[Tensor] = None      # [capacity, action_elems] next actions, slot order
        self._na_head = 0
        self._na_size = 0

    # -- storage of one complete (s, a, r, s', a') -------------------------------------------
    def _store(self, args: Dict[str, Any], next_action: Tensor) -> None:
        BasicReplayBuffer.push(self, **args)
        arena = self._arena
        assert arena is not None
        flat = next_action.detach().reshape(-1)
        if self._na is None:
            self._na = torch.zeros(self.capacity, flat.numel(), dtype=flat.dtype, device=arena.device)
        if self._na_size < self.capacity:
            slot = (self._na_head + self._na_size) % self.capacity
            self._na_size += 1
        else:                                   # FIFO eviction of the oldest row
            slot = self._na_head
            self._na_head = (self._na_head + 1) % self.capacity
        self._na[slot].copy_(flat.to(self._na.dtype))

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
        self._na_head = self._na_size = 0

    def sample(self, batch_size: int) -> TransitionBatch:
        batch = super().sample(batch_size)
        assert self._na is not None
        idx = self.last_indices
        phys = ((idx + self._na_head) % self.capacity).contiguous()
        width = int(self._na.shape[1])
        out = torch.empty(int(batch_size), width, dtype=self._na.dtype, device=self._na.device)
        N.check(N.lib().pa_gather_rows(self._na.data_ptr(), width * self._na.element_size(),
                                       phys.data_ptr(), int(batch_size), out.data_ptr(),
                                       N.stream_ptr(self._na.device)))
        batch.next_action = out.reshape(tuple(batch.action.shape)).to(batch.action.device)
        return batch
