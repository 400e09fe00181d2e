"""Batched ``observe``: E environments feeding one replay arena (second half).

The reference steps ONE environment and pushes ONE transition per ``PearlAgent.observe``
(pearl/pearl_agent.py:169-211; the episode loop is
pearl/utils/functional_utils/train_and_eval/online_learning.py:235-320: ``agent.act`` ->
``env.step`` -> ``agent.observe`` [-> ``agent.learn``]).  At 15 us of Python per push that is
68 k transitions/s in front of a learner that consumes 28 M/s.  ``VectorEnvFeeder`` keeps the
reference's per-environment semantics — the rows that reach the buffer are exactly the rows E
``PearlAgent``s sharing one learner and one buffer would push, in environment order — and turns a
vector step into

* ONE batched action selection (``policy_learner.act_many``: a single (E, A, S + AD) forward, then
  the exploration module row by row, i.e. the same draws from Python's ``random`` stream as E
  successive ``act`` calls), and
* ONE ``replay_buffer.push_many`` of E rows (one scatter launch into the arena).

Two kinds of environments are accepted:

* a sequence of reference-style environments (``reset() -> (observation, action_space)``,
  ``step(action) -> ActionResult``; pearl/api/environment.py:20-56).  Their transitions are staged in
  pinned host tensors that are allocated once — no per-transition tensor objects — and go to the
  device in the one asynchronous copy ``push_many`` makes;
* ONE ``BatchedEnvironment`` whose observations already are ``(E, S)`` tensors, typically on the
  GPU (a simulator that lives in HBM): states, actions, rewards and flags then go from the
  simulator to the arena without ever visiting the host.

Environments that finish an episode are reset in place (``env.reset()``), the terminal transition
carries the terminal observation as ``next_state`` and the next transition of that environment
starts from the reset observation — what ``run_episode`` called in a loop does.

Dynamic action spaces (an ``ActionResult.available_action_space`` that differs between rows) have
no batched form in the arena (``push_many`` stores one static space); such steps fall back to E
``push`` calls, same rows, same order.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Dict, List, Optional, Sequence, Tuple

import torch
from torch import Tensor

from .pearl_agent import ActionResult, PearlAgent


@dataclass
class BatchedActionResult:
    """One vector step of a ``BatchedEnvironment``: leading dimension E everywhere."""
    observation: Tensor                     # (E, S): what each environment shows AFTER the step
    reward: Tensor                          # (E,)
    terminated: Tensor                      # (E,) bool
    truncated: Tensor                       # (E,) bool
    # (E, S) or None: for rows with terminated | truncated, the first observation of the NEXT
    # episode (the environment has already reset itself); other rows are ignored.  None: no row
    # finished, or the environment never terminates.
    reset_observation: Optional[Tensor] = None
    cost: Optional[Tensor] = None           # (E,) or None


class BatchedEnvironment:
    """E lock-stepped environments behind tensors.  ``reset() -> ((E, S) observations,
    action_space)`` with ONE static action space for all of them; ``step((E, *action_shape)
    actions) -> BatchedActionResult``."""

    def reset(self, seed: Optional[int] = None) -> Tuple[Tensor, Any]:
        raise NotImplementedError

    def step(self, actions: Tensor) -> BatchedActionResult:
        raise NotImplementedError


def act_many(policy_learner: Any, states: Tensor, action_space: Any, exploit: bool) -> List[Any]:
    """E actions for E states: the learner's batched ``act_many`` when it has one, else E ``act``
    calls.  Either way row e is what ``act(states[e], action_space, exploit)`` returns, and the
    exploration draws happen in row order."""
    fn = getattr(policy_learner, "act_many", None)
    if fn is not None:
        return fn(states, action_space, exploit=exploit)
    return [policy_learner.act(states[e], action_space, exploit=exploit)
            for e in range(int(states.shape[0]))]


class VectorEnvFeeder:
    def __init__(self, agent: PearlAgent, envs: Any, pin_memory: Optional[bool] = None) -> None:
        self.agent = agent
        self.batched = isinstance(envs, BatchedEnvironment)
        self.envs: Any = envs if self.batched else list(envs)
        assert self.batched or len(self.envs) > 0, "VectorEnvFeeder needs at least one environment"
        self._pin = torch.cuda.is_available() if pin_memory is None else bool(pin_memory)
        self.num_envs = 0
        self._obs: Any = None                 # (E, S) tensor (batched) or list of E observations
        self._spaces: List[Any] = []          # per environment (one shared object when static)
        self._stage: Optional[Dict[str, Tensor]] = None
        self.steps = 0                        # vector steps taken
        self.transitions = 0                  # rows pushed
        self.episodes = 0                     # episodes finished
        self.episode_returns: List[float] = []
        self._returns: Optional[Tensor] = None

    # ------------------------------------------------------------------ reset
    def reset(self, seed: Optional[int] = None) -> None:
        """``env.reset()`` everywhere and ``policy_learner.reset(action_space)`` once
        (pearl_agent.py:260-271 per agent; the learner is shared)."""
        if self.batched:
            obs, space = self.envs.reset(seed=seed)
            self._obs = obs
            self.num_envs = int(obs.shape[0])
            self._spaces = [space] * self.num_envs
        else:
            self._obs, self._spaces = [], []
            for e, env in enumerate(self.envs):
                obs, space = env.reset(seed=None if seed is None else seed + e)
                self._obs.append(obs)
                self._spaces.append(space)
            self.num_envs = len(self.envs)
        self._returns = torch.zeros(self.num_envs, dtype=torch.float64)
        self.agent.policy_learner.reset(self._spaces[0])

    # ------------------------------------------------------------------ helpers
    def _static_space(self) -> Tuple[bool, Any]:
        """(every environment shows the same action-space object, that object) — the object may be
        None (continuous control without an explicit space)."""
        s0 = self._spaces[0]
        return all(s is s0 for s in self._spaces), s0

    def _states_tensor(self) -> Tensor:
        dev = self.agent.device
        if self.batched:
            return self._obs.to(dev)
        return torch.stack([torch.as_tensor(o, dtype=torch.float32) for o in self._obs]).to(dev)

    def _ensure_stage(self, action: Tensor, obs_shape: Tuple[int, ...], reward_dtype: torch.dtype,
                      with_cost: bool) -> Dict[str, Tensor]:
        if self._stage is not None:
            return self._stage
        E = self.num_envs

        def new(shape: Tuple[int, ...], dtype: torch.dtype) -> Tensor:
            t = torch.zeros((E,) + tuple(shape), dtype=dtype)
            return t.pin_memory() if self._pin else t

        self._stage = {"state": new(obs_shape, torch.float32),
                       "next_state": new(obs_shape, torch.float32),
                       "action": new(tuple(action.shape), action.dtype),
                       "reward": new((), reward_dtype),
                       "terminated": new((), torch.bool), "truncated": new((), torch.bool)}
        if with_cost:
            self._stage["cost"] = new((), torch.float32)
        return self._stage

    def _max_actions(self) -> Optional[int]:
        pl = self.agent.policy_learner
        return (None if pl._is_action_continuous
                else pl.action_representation_module.max_number_actions)

    # ------------------------------------------------------------------ one vector step
    def step(self, exploit: bool = False) -> Any:
        """act -> step -> push for every environment; returns what the environments returned
        (a ``BatchedActionResult`` or the list of E ``ActionResult``s)."""
        assert self.num_envs > 0, "call reset() first"
        out = self._step_batched(exploit) if self.batched else self._step_list(exploit)
        self.steps += 1
        self.transitions += self.num_envs
        return out

    def _step_batched(self, exploit: bool) -> BatchedActionResult:
        pl, rb = self.agent.policy_learner, self.agent.replay_buffer
        space = self._spaces[0]
        states = self._states_tensor()
        actions = act_many(pl, states, space, exploit)
        act_t = actions if isinstance(actions, Tensor) else torch.stack(
            [torch.as_tensor(a) for a in actions])
        res = self.envs.step(act_t)
        done = res.terminated | res.truncated
        rb.push_many(state=states, action=act_t.to(states.device), reward=res.reward,
                     terminated=res.terminated, truncated=res.truncated,
                     next_state=res.observation,
                     curr_available_actions=None if pl._is_action_continuous else space,
                     next_available_actions=None if pl._is_action_continuous else space,
                     max_number_actions=self._max_actions(), cost=res.cost)
        nxt = res.observation
        if res.reset_observation is not None:
            nxt = torch.where(done.reshape((-1,) + (1,) * (nxt.ndim - 1)).to(nxt.device),
                              res.reset_observation, nxt)
        self._obs = nxt
        self._account(res.reward, done)
        return res

    def _step_list(self, exploit: bool) -> List[ActionResult]:
        pl, rb = self.agent.policy_learner, self.agent.replay_buffer
        E = self.num_envs
        is_static, static = self._static_space()
        if is_static:
            actions = act_many(pl, self._states_tensor(), static, exploit)
        else:
            dev = self.agent.device
            actions = [pl.act(torch.as_tensor(self._obs[e]).to(dev), self._spaces[e], exploit=exploit)
                       for e in range(E)]
        results: List[ActionResult] = []
        next_spaces: List[Any] = []
        for e, env in enumerate(self.envs):
            r = env.step(actions[e])
            results.append(r)
            next_spaces.append(self._spaces[e] if r.available_action_space is None
                               else r.available_action_space)
        batchable = is_static and all(s is static for s in next_spaces)
        if batchable:
            from .replay_buffers.basic_replay_buffer import _torch_dtype_of_value
            a0 = torch.as_tensor(actions[0])
            rdt = _torch_dtype_of_value(results[0].reward)
            rdt = torch.int64 if rdt == torch.bool else rdt
            o0 = torch.as_tensor(self._obs[0])
            st = self._ensure_stage(a0, tuple(o0.shape), rdt, results[0].cost is not None)
            for e, r in enumerate(results):
                st["state"][e] = torch.as_tensor(self._obs[e], dtype=torch.float32)
                st["next_state"][e] = torch.as_tensor(r.observation, dtype=torch.float32)
                st["action"][e] = torch.as_tensor(actions[e]).to("cpu")
                st["reward"][e] = r.reward
                st["terminated"][e] = bool(r.terminated)
                st["truncated"][e] = bool(r.truncated)
                if "cost" in st:
                    st["cost"][e] = r.cost
            disc = not pl._is_action_continuous
            rb.push_many(state=st["state"], action=st["action"], reward=st["reward"],
                         terminated=st["terminated"], truncated=st["truncated"],
                         next_state=st["next_state"],
                         curr_available_actions=static if disc else None,
                         next_available_actions=static if disc else None,
                         max_number_actions=self._max_actions(), cost=st.get("cost"))
        else:
            for e, r in enumerate(results):     # dynamic action spaces: the reference's push, row by row
                rb.push(state=self._obs[e], action=actions[e], reward=r.reward,
                        next_state=r.observation, curr_available_actions=self._spaces[e],
                        next_available_actions=next_spaces[e], terminated=r.terminated,
                        truncated=r.truncated, max_number_actions=self._max_actions(), cost=r.cost)
        rewards = torch.tensor([float(r.reward) for r in results], dtype=torch.float64)
        done = torch.tensor([bool(r.terminated or r.truncated) for r in results])
        for e, r in enumerate(results):
            if done[e]:
                self._obs[e], self._spaces[e] = self.envs[e].reset()
            else:
                self._obs[e], self._spaces[e] = r.observation, next_spaces[e]
        self._account(rewards, done)
        return results

    def _account(self, reward: Tensor, done: Tensor) -> None:
        if self.batched and reward.is_cuda:
            return          # (episode statistics would be a device-to-host copy per step: left to the env)
        self._returns += reward.detach().to("cpu", torch.float64).reshape(-1)
        idx = torch.nonzero(done.to("cpu").reshape(-1)).reshape(-1).tolist()
        for e in idx:
            self.episode_returns.append(float(self._returns[e]))
            self._returns[e] = 0.0
        self.episodes += len(idx)

    # ------------------------------------------------------------------ loop
    def run(self, vector_steps: int, learn_every: int = 0, exploit: bool = False,
            learning_start_step: int = 0) -> Dict[str, Any]:
        """``vector_steps`` steps of every environment; ``agent.learn()`` after every
        ``learn_every``-th vector step once ``learning_start_step`` steps have been taken
        (online_learning.py:235-320's learn / learn_every_k_steps / learning_start_step)."""
        if self.num_envs == 0:
            self.reset()
        reports: List[Dict[str, Any]] = []
        for _ in range(int(vector_steps)):
            self.step(exploit=exploit)
            if learn_every > 0 and self.steps >= learning_start_step and self.steps % learn_every == 0:
                reports.append(self.agent.learn())
        return {"vector_steps": int(vector_steps), "transitions": self.num_envs * int(vector_steps),
                "episodes": self.episodes, "learn_reports": reports}
