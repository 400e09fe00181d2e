"""Batched observe (pearl_amd/vector_env.py): the rows a VectorEnvFeeder pushes are the rows E
reference-style agents would push one by one (pearl_agent.py:169-211), in environment order, and
``act_many`` is E ``act`` calls.  Host logic on the CPU with a recording buffer; the GPU tests drive
the real arena."""
import random

import pytest
import torch

from pearl_amd import (BasicReplayBuffer, BatchedActionResult, BatchedEnvironment, DeepQLearning,
                       DiscreteActionSpace, OneHotActionTensorRepresentationModule, PearlAgent,
                       VectorEnvFeeder)
from pearl_amd.pearl_agent import ActionResult
from pearl_amd.policy_learners.exploration import EGreedyExploration

S, A = 6, 4


def space(seed=0):
    return DiscreteActionSpace([torch.tensor([k]) for k in range(A)], seed=seed)


def learner(eps=0.0, seed=0):
    torch.manual_seed(seed)
    return DeepQLearning(state_dim=S, action_space=space(), hidden_dims=[16, 16], training_rounds=1,
                         batch_size=8, exploration_module=EGreedyExploration(eps),
                         action_representation_module=OneHotActionTensorRepresentationModule(A))


class ToyEnv:
    """Deterministic: the observation is a function of (env id, step count); an episode ends every
    `horizon` steps (terminated on even env ids, truncated on odd ones)."""

    def __init__(self, eid, sp, horizon=3, dynamic=False):
        self.eid, self.sp, self.horizon, self.dynamic = eid, sp, horizon, dynamic
        self.t = 0
        self.episode = 0

    def _obs(self):
        g = torch.Generator().manual_seed(1000 * self.eid + 10 * self.episode + self.t)
        return torch.randn(S, generator=g)

    def reset(self, seed=None):
        self.t = 0
        self.episode += 1
        return self._obs(), self.sp

    def step(self, action):
        self.t += 1
        done = self.t >= self.horizon
        nsp = DiscreteActionSpace([torch.tensor([k]) for k in range(A - 1)]) if self.dynamic else None
        return ActionResult(observation=self._obs(), reward=float(int(action) + self.eid),
                            terminated=done and self.eid % 2 == 0,
                            truncated=done and self.eid % 2 == 1, available_action_space=nsp)


class Recorder:
    """Stands in for the replay buffer: keeps what push / push_many were given, as rows."""

    def __init__(self):
        self._is_action_continuous = False
        self.device_for_batches = torch.device("cpu")
        self.rows, self.calls = [], []

    def _row(self, state, action, reward, terminated, truncated, next_state):
        return (torch.as_tensor(state, dtype=torch.float32).cpu().clone(),
                int(torch.as_tensor(action).reshape(-1)[0]), float(reward), bool(terminated),
                bool(truncated), torch.as_tensor(next_state, dtype=torch.float32).cpu().clone())

    def push(self, state, action, reward, terminated, truncated, curr_available_actions=None,
             next_state=None, next_available_actions=None, max_number_actions=None, cost=None):
        self.calls.append("push")
        self.rows.append(self._row(state, action, reward, terminated, truncated, next_state))

    def push_many(self, state, action, reward, terminated, truncated, next_state=None,
                  curr_available_actions=None, next_available_actions=None, max_number_actions=None,
                  cost=None):
        self.calls.append("push_many")
        assert max_number_actions == A and curr_available_actions is next_available_actions
        for e in range(state.shape[0]):
            self.rows.append(self._row(state[e], action[e], reward[e], terminated[e], truncated[e],
                                       next_state[e]))

    def __len__(self):
        return len(self.rows)


def same_rows(a, b):
    assert len(a) == len(b)
    for x, y in zip(a, b):
        assert torch.equal(x[0], y[0]) and x[1:5] == y[1:5] and torch.equal(x[5], y[5])


def sequential_agents(pl, envs, steps, exploit):
    """The reference's loop, one agent state per environment, one shared learner and buffer:
    round-robin over the environments, reset on done (online_learning.py:235-320)."""
    rec = Recorder()
    agents = [PearlAgent(pl, replay_buffer=rec) for _ in envs]
    for ag, env in zip(agents, envs):
        ag.reset(*env.reset())
    for _ in range(steps):
        acts = [ag.act(exploit=exploit) for ag in agents]
        for ag, env, a in zip(agents, envs, acts):
            r = env.step(a)
            ag.observe(r)
            if r.done:
                obs, sp = env.reset()
                ag._subjective_state, ag._action_space = obs, sp
    return rec


def test_act_many_is_e_act_calls_with_the_same_exploration_draws():
    pl = learner(eps=0.4)
    states = torch.randn(32, S)
    random.seed(7)
    sp = space(seed=3)
    one_by_one = torch.stack([pl.act(states[e], sp, exploit=False) for e in range(32)])
    random.seed(7)
    sp = space(seed=3)
    pl.exploration_module.time_step = 0
    batched = pl.act_many(states, sp, exploit=False)
    assert torch.equal(one_by_one, batched)
    assert torch.equal(pl.act_many(states, sp, exploit=True),
                       torch.stack([pl.act(states[e], sp, exploit=True) for e in range(32)]))


def test_act_many_follows_the_epsilon_warm_up_of_sequential_calls():
    """Epsilon scheduling (epsilon_greedy_exploration.py:61-75) moves epsilon with every act():
    row e of act_many sees the epsilon the e-th sequential call would, across the end of the
    warm-up as well."""
    def fresh():
        pl = learner()
        pl.exploration_module = EGreedyExploration(0.05, start_epsilon=0.9, end_epsilon=0.1, warmup_steps=20)
        return pl
    states = torch.randn(48, S)
    pl = fresh()
    random.seed(11)
    sp = space(seed=5)
    want = torch.stack([pl.act(states[e], sp, exploit=False) for e in range(48)])
    eps_want, t_want = pl.exploration_module.curr_epsilon, pl.exploration_module.time_step
    pl = fresh()
    random.seed(11)
    sp = space(seed=5)
    got = torch.cat([pl.act_many(states[:30], sp), pl.act_many(states[30:], sp)])
    assert torch.equal(want, got)
    assert (pl.exploration_module.curr_epsilon, pl.exploration_module.time_step) == (eps_want, t_want)


def test_feeder_pushes_what_e_sequential_agents_push():
    pl = learner()
    sp = space()
    ref = sequential_agents(pl, [ToyEnv(e, sp) for e in range(5)], steps=8, exploit=True)
    rec = Recorder()
    feeder = VectorEnvFeeder(PearlAgent(pl, replay_buffer=rec), [ToyEnv(e, sp) for e in range(5)],
                             pin_memory=False)
    feeder.reset()
    out = feeder.run(8, exploit=True)
    assert out["transitions"] == 40 and rec.calls == ["push_many"] * 8
    same_rows(ref.rows, rec.rows)
    # 8 steps with a horizon of 3: every environment finished two episodes and was reset in place
    assert feeder.episodes == 10 and len(feeder.episode_returns) == 10
    assert sum(r[3] for r in rec.rows) == 6 and sum(r[4] for r in rec.rows) == 4   # terminated | truncated


def test_dynamic_action_spaces_fall_back_to_per_row_push():
    pl = learner()
    sp = space()
    ref = sequential_agents(pl, [ToyEnv(e, sp, dynamic=True) for e in range(3)], steps=2, exploit=True)
    rec = Recorder()
    feeder = VectorEnvFeeder(PearlAgent(pl, replay_buffer=rec),
                             [ToyEnv(e, sp, dynamic=True) for e in range(3)], pin_memory=False)
    feeder.reset()
    feeder.run(2, exploit=True)
    assert rec.calls == ["push"] * 6
    same_rows(ref.rows, rec.rows)


class _ContinuousLearner(torch.nn.Module):
    """The attributes the agent / feeder read from a policy learner, with a deterministic
    continuous policy: a = tanh(mean(state)) in every dimension."""

    def __init__(self, dim=2):
        super().__init__()
        self._is_action_continuous = True
        self.on_policy = False
        self.dim = dim
        self.action_representation_module = None
        self.resets = 0

    def reset(self, action_space):
        self.resets += 1

    def act(self, state, action_space, exploit=False):
        return torch.tanh(torch.as_tensor(state).float().mean()).repeat(self.dim)


class _BoxEnv:
    def __init__(self, eid):
        self.eid, self.t = eid, 0

    def reset(self, seed=None):
        self.t = 0
        return torch.full((S,), float(self.eid)), None

    def step(self, action):
        self.t += 1
        return ActionResult(observation=torch.full((S,), float(self.eid) + 0.1 * self.t),
                            reward=float(action.sum()), terminated=False, truncated=self.t % 2 == 0)


def test_continuous_actions_go_through_act_row_by_row_and_one_push_many():
    """A learner without act_many (continuous control): E act() calls, still ONE push_many per
    vector step with no action tables (pearl_agent.py:196-201: max_number_actions is None)."""
    pl = _ContinuousLearner()

    class Rec(Recorder):
        def push_many(self, state, action, reward, terminated, truncated, next_state=None,
                      curr_available_actions=None, next_available_actions=None,
                      max_number_actions=None, cost=None):
            self.calls.append("push_many")
            assert max_number_actions is None and curr_available_actions is None
            assert action.shape == (3, 2) and action.dtype == torch.float32
            self.rows.extend((state[e].clone(), action[e].clone(), float(reward[e]), bool(truncated[e]))
                             for e in range(3))

    rec = Rec()
    rec._is_action_continuous = True
    feeder = VectorEnvFeeder(PearlAgent(pl, replay_buffer=rec), [_BoxEnv(e) for e in range(3)],
                             pin_memory=False)
    feeder.reset()
    feeder.run(4)
    assert rec.calls == ["push_many"] * 4 and pl.resets == 1
    # row order = environment order; environment 2's action is tanh(2) / tanh(2.1) / reset -> tanh(2)
    a = [rec.rows[i][1][0].item() for i in (2, 5, 8, 11)]
    want = [torch.tanh(torch.tensor(2.0)).item(), torch.tanh(torch.tensor(2.0) + 0.1).item()]
    assert a == pytest.approx(want + want)
    assert sum(r[3] for r in rec.rows) == 6 and feeder.episodes == 6


class DeviceSim(BatchedEnvironment):
    """E environments as tensors on one device: x' = roll(x) + 0.1 onehot(action); reward = x'[a];
    every `horizon`-th step terminates all rows and restarts them from fresh states."""

    def __init__(self, E, device, horizon=4):
        self.E, self.dev, self.horizon = E, torch.device(device), horizon
        self.sp = space()
        self.t = 0
        self.gen = torch.Generator(device="cpu").manual_seed(5)

    def _fresh(self):
        return torch.randn(self.E, S, generator=self.gen).to(self.dev)

    def reset(self, seed=None):
        self.x = self._fresh()
        self.t = 0
        return self.x, self.sp

    def step(self, actions):
        a = actions.reshape(self.E).to(self.dev)
        nxt = torch.roll(self.x, 1, dims=1) + 0.1 * torch.nn.functional.one_hot(a, S).float()
        reward = nxt.gather(1, a.reshape(-1, 1)).reshape(-1)
        self.t += 1
        done = torch.full((self.E,), self.t % self.horizon == 0, device=self.dev)
        res = BatchedActionResult(observation=nxt, reward=reward, terminated=done,
                                  truncated=torch.zeros_like(done),
                                  reset_observation=self._fresh() if bool(done[0]) else None)
        self.x = torch.where(done.reshape(-1, 1), res.reset_observation, nxt) \
            if res.reset_observation is not None else nxt
        return res


def test_batched_environment_rows_on_the_cpu():
    pl = learner()
    rec = Recorder()
    sim = DeviceSim(7, "cpu")
    feeder = VectorEnvFeeder(PearlAgent(pl, replay_buffer=rec), sim, pin_memory=False)
    feeder.reset()
    x0 = sim.x.cpu().clone()
    feeder.run(5, exploit=True)
    assert rec.calls == ["push_many"] * 5 and len(rec) == 35
    assert torch.equal(torch.stack([r[0] for r in rec.rows[:7]]), x0)
    # the step after the terminal one starts from the reset observation, not from the terminal one
    term_next = torch.stack([r[5] for r in rec.rows[21:28]])
    after = torch.stack([r[0] for r in rec.rows[28:35]])
    assert all(r[3] for r in rec.rows[21:28]) and not torch.equal(term_next, after)
    assert feeder.episodes == 7


@pytest.mark.gpu
def test_feeder_fills_the_arena_like_sequential_observe_and_the_learner_trains_on_it():
    dev = torch.device("cuda", 0)
    sp = space()
    pl = learner()
    a_seq = PearlAgent(pl, replay_buffer=BasicReplayBuffer(1000), device_id=0)
    agents = [a_seq] + [PearlAgent(pl, replay_buffer=a_seq.replay_buffer, device_id=0) for _ in range(4)]
    envs = [ToyEnv(e, sp) for e in range(5)]
    for ag, env in zip(agents, envs):
        ag.reset(*env.reset())
    for _ in range(8):
        acts = [ag.act(exploit=True) for ag in agents]
        for ag, env, a in zip(agents, envs, acts):
            r = env.step(a)
            ag.observe(r)
            if r.done:
                ag._subjective_state, ag._action_space = env.reset()
    rb = BasicReplayBuffer(1000)
    feeder = VectorEnvFeeder(PearlAgent(pl, replay_buffer=rb, device_id=0),
                             [ToyEnv(e, sp) for e in range(5)])
    feeder.reset()
    feeder.run(8, exploit=True)
    want, got = a_seq.replay_buffer.state_dict(), rb.state_dict()
    assert want["size"] == got["size"] == 40
    for k, v in want["columns"].items():
        if v is None:
            assert got["columns"][k] is None
        else:
            assert torch.equal(v, got["columns"][k]), k
    assert rb.shared_action_table
    # a simulator that lives on the device: no host copy between environment and arena
    rb2 = BasicReplayBuffer(4096)
    pl2 = learner()
    f2 = VectorEnvFeeder(PearlAgent(pl2, replay_buffer=rb2, device_id=0), DeviceSim(256, dev))
    f2.reset()
    out = f2.run(12, learn_every=4, exploit=False, learning_start_step=4)
    assert len(rb2) == 256 * 12 and len(out["learn_reports"]) == 3
    assert all(x == x for rep in out["learn_reports"] for x in rep["loss"])
