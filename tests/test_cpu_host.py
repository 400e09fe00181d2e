"""CPU: host logic, the C-ABI surface, and loud failure without a GPU.  No kernel runs here."""
import ctypes as C
import os
import random
import re
import subprocess
import sys

import pytest

import torch

from conftest import GOLDEN_NAMES, REPO


# ---------------------------------------------------------------------------- C ABI
def _free_port() -> int:
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from helpers import free_port
    return free_port()


def _header_functions():
    text = open(os.path.join(REPO, "include", "pearl_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pa_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from pearl_amd import _native as N
    declared = _header_functions()
    assert len(declared) >= 30
    assert sorted(N.SIGNATURES) == declared          # the ctypes table mirrors the header
    lib = N.lib()
    for name in declared:
        assert hasattr(lib, name), name              # ... and the .so exports all of it
    assert lib.pa_abi_version() == 1
    assert lib.pa_device_count() >= 0


def test_ctypes_structs_match_c_layout(tmp_path):
    from pearl_amd import _native as N
    exe = tmp_path / "abi_probe"
    subprocess.run(["gcc", "-std=c99", "-o", str(exe), os.path.join(REPO, "tests", "abi_probe.c")],
                   check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    facts = dict(line.rsplit(" ", 1) for line in out.strip().splitlines())
    pairs = {"pa_arena_desc": N.ArenaDesc, "pa_transition": N.Transition, "pa_columns": N.Columns,
             "pa_batch_out": N.BatchOut, "pa_dqn_desc": N.DqnDesc, "pa_dqn_buffers": N.DqnBuffers,
             "pa_dqn_batch": N.DqnBatch, "pa_learn_args": N.LearnArgs, "pa_mlp_desc": N.MlpDesc,
             "pa_mlp_buffers": N.MlpBuffers, "pa_sac_step_args": N.SacStepArgs,
             "pa_ddpg_step_args": N.DdpgStepArgs, "pa_ac_loop_args": N.AcLoopArgs,
             "pa_bandit_step_args": N.BanditStepArgs, "pa_ppo_learn_args": N.PpoLearnArgs,
             "pa_dsac_step_args": N.DsacStepArgs, "pa_iql_step_args": N.IqlStepArgs}
    for cname, ctype in pairs.items():
        assert C.sizeof(ctype) == int(facts[cname]), cname
    for key, value in facts.items():
        if "." in key:
            cname, member = key.split(".")
            assert getattr(pairs[cname], member).offset == int(value), key


def test_param_layout_matches_reference_network():
    from pearl_amd import _native as N
    lib = N.lib()
    offs = (C.c_int64 * 6)()
    assert lib.pa_dqn_param_offsets(128, 16, 256, 256, offs) == 0
    assert list(offs) == [0, 36864, 37120, 102656, 102912, 103168]
    assert lib.pa_dqn_param_count(128, 16, 256, 256) == 103172       # 103169 params + tail pad
    assert lib.pa_dqn_param_offsets(5, 5, 24, 16, offs) == 0
    assert all(o % 4 == 0 for o in offs)


# ---------------------------------------------------------------------------- loud failure
@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU behaviour")
def test_no_gpu_no_fallback():
    from pearl_amd import (BasicReplayBuffer, DeepQLearning, DiscreteActionSpace,
                           OneHotActionTensorRepresentationModule, TransitionBatch, _native as N)
    sp = DiscreteActionSpace([torch.tensor([k]) for k in range(3)])
    rb = BasicReplayBuffer(10)
    with pytest.raises(N.NativeError, match="no HIP device"):
        rb.push(torch.zeros(4), torch.tensor([1]), 1.0, False, False, sp, torch.zeros(4), sp, 3)
    pl = DeepQLearning(state_dim=4, action_space=sp, hidden_dims=[8, 8],
                       action_representation_module=OneHotActionTensorRepresentationModule(3))
    batch = TransitionBatch(state=torch.zeros(2, 4), action=torch.zeros(2, 3), reward=torch.zeros(2),
                            next_state=torch.zeros(2, 4))
    with pytest.raises(N.NativeError, match="no HIP device|no CPU"):
        pl.learn_batch(batch)
    # the raw ABI refuses too (no hidden host path below the python layer)
    h = C.c_void_p()
    desc = N.ArenaDesc(capacity=4, device=0, state_dim=4, action_elems=1, action_dtype=N.PA_I64,
                       reward_dtype=N.PA_F32, max_actions=0, avail_dim=0, has_next_state=1)
    assert N.lib().pa_arena_create(C.byref(h), C.byref(desc)) == N.PA_ERR_HIP
    assert "no CPU fallback" in N.last_error()


def test_unsupported_configurations_fail_loudly():
    """What the HIP engines cannot compute is refused at construction — never trained as if it
    were something else (a non-ReLU network instance used to be trained as ReLU)."""
    import torch.nn as nn
    from pearl_amd import DeepQLearning, DiscreteActionSpace, OneHotActionTensorRepresentationModule
    from pearl_amd.neural_networks.sequential_decision_making.q_value_networks import (
        DuelingQValueNetwork, VanillaQValueMultiHeadNetwork, VanillaQValueNetwork)
    sp = DiscreteActionSpace([torch.tensor([k]) for k in range(3)])
    rep = OneHotActionTensorRepresentationModule(3)
    kw = dict(state_dim=4, action_space=sp, action_representation_module=rep)
    # built: other depths / widths and the multi-head / dueling architectures (generic engine)
    assert DeepQLearning(hidden_dims=[8, 8], **kw)._fused
    for nt, hd in ((VanillaQValueNetwork, [8, 8, 8]), (VanillaQValueNetwork, [300, 8]),
                   (VanillaQValueNetwork, [8]), (VanillaQValueMultiHeadNetwork, [8, 8]),
                   (DuelingQValueNetwork, [8, 6])):
        assert not DeepQLearning(hidden_dims=hd, network_type=nt, **kw)._fused
    # the CQL term: VanillaQValueNetwork of any depth / form (fused shape: test_conservative_q_learning;
    # beyond it the generic engine, qnet_cql_* fixtures) and, since round 6, multi-head and dueling
    # networks (qnet_cql_multihead_* / qnet_cql_dueling_*: loss_fn_utils.py:17-72 takes any QValueNetwork)
    DeepQLearning(hidden_dims=[8, 8], is_conservative=True, **kw)
    assert not DeepQLearning(hidden_dims=[8, 8, 8], is_conservative=True, **kw)._fused
    for nt in (VanillaQValueMultiHeadNetwork, DuelingQValueNetwork):
        assert not DeepQLearning(hidden_dims=[8, 8], network_type=nt, is_conservative=True, **kw)._fused
    # built (round 5): mlp_block's LayerNorm and its other hidden activations — through the generic
    # engine, never the fused step (common/utils.py:75-152)
    from pearl_amd.neural_networks.common.utils import mlp_block
    net = VanillaQValueNetwork(state_dim=4, action_dim=3, hidden_dims=[8, 8], output_dim=1, use_layer_norm=True)
    assert isinstance(net._model[0][1], nn.LayerNorm)
    assert not DeepQLearning(network_instance=net, **kw)._fused
    for act in ("leaky_relu", "tanh", "softplus", "sigmoid", "linear"):
        net = VanillaQValueNetwork(state_dim=4, action_dim=3, hidden_dims=[8, 8], output_dim=1)
        net._model = mlp_block(7, [8, 8], 1, hidden_activation=act, use_layer_norm=(act == "tanh"))
        assert not DeepQLearning(network_instance=net, **kw)._fused
    # the learners whose fused kernels hard-wire Linear + ReLU refuse such a network instead of
    # training it as ReLU
    with pytest.raises(NotImplementedError, match="not Linear \\+ ReLU"):
        net.linear_layers()
    # refused: hidden layers of MIXED form (mlp_block gives every hidden layer the same one),
    # activations without a kernel
    net = VanillaQValueNetwork(state_dim=4, action_dim=3, hidden_dims=[8, 8], output_dim=1)
    net._model[0][1] = nn.Tanh()
    with pytest.raises(NotImplementedError, match="not an mlp_block the HIP engine computes"):
        DeepQLearning(network_instance=net, **kw)
    net = VanillaQValueNetwork(state_dim=4, action_dim=3, hidden_dims=[8, 8], output_dim=1)
    net._model[1] = nn.Sequential(nn.Linear(8, 8), nn.LayerNorm(8), nn.ReLU())
    with pytest.raises(NotImplementedError, match="not an mlp_block the HIP engine computes"):
        DeepQLearning(network_instance=net, **kw)
    net = VanillaQValueNetwork(state_dim=4, action_dim=3, hidden_dims=[8, 8], output_dim=1)
    net._model[0] = nn.Sequential(nn.Linear(7, 8), nn.ReLU(), nn.BatchNorm1d(8))
    with pytest.raises(NotImplementedError, match="not an mlp_block the HIP engine computes"):
        DeepQLearning(network_instance=net, **kw)
    with pytest.raises(NotImplementedError):
        mlp_block(7, [8, 8], 1, hidden_activation="normalized_softplus")
    # built (round 6): mlp_block's batch norm, dropout and skip connections (utils.py:113-131, :142-150)
    # — the module tree and state_dict keys are the reference's
    from pearl_amd.neural_networks.common.residual_wrapper import ResidualWrapper
    from pearl_amd.policy_learners.sequential_decision_making.generic_q import mlp_spec
    m = mlp_block(8, [8, 6], 6, use_batch_norm=True, use_layer_norm=True, dropout_ratio=0.25,
                  use_skip_connections=True)
    assert isinstance(m[0], ResidualWrapper) and not isinstance(m[1], ResidualWrapper) \
        and isinstance(m[2], ResidualWrapper)                   # 8 -> 8 wrapped, 8 -> 6 not, 6 -> 6 wrapped
    assert [type(x).__name__ for x in m[0].module] == ["Linear", "LayerNorm", "Dropout", "ReLU", "BatchNorm1d"]
    assert list(m.state_dict())[:9] == [
        "0.module.0.weight", "0.module.0.bias", "0.module.1.weight", "0.module.1.bias", "0.module.4.weight",
        "0.module.4.bias", "0.module.4.running_mean", "0.module.4.running_var", "0.module.4.num_batches_tracked"]
    spec = mlp_spec(m)
    assert spec["residual"] == 0b101 and spec["dropout"] == 0.25 and not spec["plain"]
    assert all(isinstance(b, nn.BatchNorm1d) for b in spec["bnorms"]) and len(spec["norms"]) == 2
    assert mlp_spec(mlp_block(7, [8, 8], 1))["plain"]
    # a skip connection on the last layer must not slip through the plain-form learners
    from pearl_amd.neural_networks.common.value_networks import VanillaValueNetwork
    with pytest.raises(NotImplementedError, match="plain form"):
        VanillaValueNetwork(4, [8, 8], 8, use_skip_connections=True).linear_layers()
    # BatchNorm1d in a VanillaQValueNetwork: the reference's own forward raises (its (B, A, S + AD)
    # input is not BatchNorm1d's (N, C) / (N, C, L)), so the learner refuses it; multi-head networks
    # (2-D state input) take it
    net = VanillaQValueNetwork(state_dim=4, action_dim=3, hidden_dims=[8, 8], output_dim=1)
    net._model = mlp_block(7, [8, 8], 1, use_batch_norm=True)
    with pytest.raises(NotImplementedError, match="BatchNorm1d"):
        DeepQLearning(network_instance=net, **kw)
    mh = VanillaQValueMultiHeadNetwork(state_dim=4, action_dim=3, hidden_dims=[8, 8], output_dim=3)
    mh._model = mlp_block(4, [8, 8], 3, use_batch_norm=True, dropout_ratio=0.1, use_skip_connections=True)
    assert not DeepQLearning(network_instance=mh, **kw)._fused
    with pytest.raises(NotImplementedError):
        DeepQLearning(hidden_dims=[8, 8], optimizer=torch.optim.SGD(net.parameters(), lr=0.1), **kw)


# ---------------------------------------------------------------------------- batch contract
def test_transition_batch_defaults_and_checks():
    """test/unit/with_pytorch/test_transition.py semantics."""
    from pearl_amd import TransitionBatch
    b = TransitionBatch(state=torch.zeros(3, 2), action=torch.zeros(3, 1), reward=torch.zeros(3))
    assert b.terminated.dtype == torch.bool and b.terminated.all() and b.terminated.shape == (3,)
    assert b.truncated.dtype == torch.bool and not b.truncated.any()
    assert len(b) == 3 and b.device.type == "cpu"
    with pytest.raises(AssertionError):
        TransitionBatch(state=torch.zeros(3), action=torch.zeros(3), reward=torch.zeros(3))
    with pytest.raises(AssertionError):
        TransitionBatch(state=torch.zeros(3, 2), action=torch.zeros(3), reward=torch.zeros(4))
    with pytest.raises(AssertionError):
        TransitionBatch(state=torch.zeros(3, 2), action=torch.zeros(3), reward=torch.zeros(3),
                        terminated=torch.zeros(2, dtype=torch.bool))
    TransitionBatch(state=torch.zeros(3, 2), action=torch.zeros(3), reward=torch.zeros(3),
                    terminated=torch.zeros(3, 1, dtype=torch.bool))
    assert b.to(torch.device("cpu")) is b


def test_padded_action_table_and_mask():
    """test/unit/with_pytorch/test_dynamic_action_space.py:28-157: actions {0,2,4} of 5."""
    from pearl_amd import DiscreteActionSpace
    from pearl_amd.replay_buffers.basic_replay_buffer import create_action_tensor_and_mask
    sp = DiscreteActionSpace([torch.tensor([0]), torch.tensor([2]), torch.tensor([4])])
    table, mask = create_action_tensor_and_mask(5, sp)
    assert torch.equal(table, torch.tensor([[0.], [2.], [4.], [0.], [0.]]))
    assert torch.equal(mask, torch.tensor([False, False, False, True, True]))
    assert create_action_tensor_and_mask(None, sp) == (None, None)
    assert create_action_tensor_and_mask(5, None) == (None, None)


def test_one_hot_module_on_host_matches_reference_formula():
    from pearl_amd import OneHotActionTensorRepresentationModule
    rep = OneHotActionTensorRepresentationModule(5)
    x = torch.tensor([[[0.], [2.], [4.], [0.], [0.]]])
    want = torch.tensor([[[1., 0, 0, 0, 0], [0, 0, 1, 0, 0], [0, 0, 0, 0, 1], [1, 0, 0, 0, 0],
                          [1, 0, 0, 0, 0]]])
    assert torch.equal(rep(x), want)
    assert torch.equal(rep(torch.tensor([3, 1])), torch.eye(5)[[3, 1]])
    assert rep.representation_dim == rep.max_number_actions == 5
    assert rep.compare(OneHotActionTensorRepresentationModule(5)) == ""
    assert rep.compare(OneHotActionTensorRepresentationModule(4)) != ""


# ---------------------------------------------------------------------------- learner host logic
@pytest.mark.parametrize("name", GOLDEN_NAMES)
def test_same_seed_same_initial_parameters_as_reference(golden, name):
    """Construction consumes torch's RNG like the reference, and state_dict keys are the
    reference's (`_model.{0,1,2}.0.{weight,bias}`)."""
    from pearl_amd import DeepQLearning, DiscreteActionSpace, OneHotActionTensorRepresentationModule
    fx = golden(name)
    cfg = fx["config"]
    torch.manual_seed(7)
    pl = DeepQLearning(state_dim=cfg["S"],
                       action_space=DiscreteActionSpace([torch.tensor([k]) for k in range(cfg["A"])]),
                       hidden_dims=cfg["hidden"], training_rounds=cfg["rounds"], batch_size=cfg["B"],
                       action_representation_module=OneHotActionTensorRepresentationModule(cfg["A"]))
    sd = pl._Q.state_dict()
    assert list(sd) == list(fx["params0"])
    for k in sd:
        assert torch.equal(sd[k], fx["params0"][k]), k
        assert torch.equal(pl._Q_target.state_dict()[k], fx["target0"][k]), k
    g = pl._optimizer.param_groups[0]
    assert (g["lr"], g["amsgrad"], g["weight_decay"], g["betas"], g["eps"]) == (
        1e-3, True, 0.01, (0.9, 0.999), 1e-8)
    assert (pl._discount_factor, pl._target_update_freq, pl._soft_update_tau) == (0.99, 10, 0.75)
    assert pl.on_policy is False and pl._is_action_continuous is False


class _FakeBuffer:
    """A ReplayBuffer that is NOT an HBM arena -> exercises the generic learn() loop."""

    def __init__(self, n):
        self.n, self.asked = n, []

    def __len__(self):
        return self.n

    def sample(self, batch_size):
        from pearl_amd import TransitionBatch
        self.asked.append(batch_size)
        return TransitionBatch(state=torch.zeros(batch_size, 2), action=torch.zeros(batch_size, 1),
                               reward=torch.zeros(batch_size), next_state=torch.zeros(batch_size, 2))


def _counting_learner(**kw):
    from pearl_amd.policy_learners.policy_learner import PolicyLearner

    class L(PolicyLearner):
        def __init__(self, **kw):
            super().__init__(on_policy=False, is_action_continuous=True, **kw)
            self.seen = []

        def set_history_summarization_module(self, value):
            self._history_summarization_module = value

        def act(self, *a, **k):
            raise NotImplementedError

        def learn_batch(self, batch):
            self.seen.append((self._training_steps, len(batch)))
            return {"loss": float(len(self.seen)), "aux": 1}

    return L(**kw)


def test_generic_learn_loop_semantics():
    """policy_learner.py:162-195: {} on empty buffer, batch-size clamp, dict of lists."""
    pl = _counting_learner(training_rounds=3, batch_size=8)
    assert pl.learn(_FakeBuffer(0)) == {}
    buf = _FakeBuffer(5)
    rep = pl.learn(buf)
    assert buf.asked == [5, 5, 5]                      # len < batch_size -> whole buffer
    assert rep == {"loss": [1.0, 2.0, 3.0], "aux": [1, 1, 1]}
    assert [s for s, _ in pl.seen] == [1, 2, 3] and pl._training_steps == 3
    pl2 = _counting_learner(training_rounds=2, batch_size=-1)
    buf = _FakeBuffer(7)
    pl2.learn(buf)
    assert buf.asked == [7, 7]
    pl3 = _counting_learner(training_rounds=1, batch_size=4)
    buf = _FakeBuffer(9)
    pl3.learn(buf)
    assert buf.asked == [4]


def test_target_update_schedule():
    """forward() soft-updates when (_training_steps + 1) % freq == 0 (deep_td_learning.py:283)."""
    from pearl_amd import DeepQLearning, DiscreteActionSpace, OneHotActionTensorRepresentationModule
    sp = DiscreteActionSpace([torch.tensor([k]) for k in range(2)])
    pl = DeepQLearning(state_dim=3, action_space=sp, hidden_dims=[4, 4], target_update_freq=10,
                       action_representation_module=OneHotActionTensorRepresentationModule(2))
    due = []
    for ts in range(1, 31):
        pl._training_steps = ts
        if pl._target_update_due():
            due.append(ts)
    assert due == [9, 19, 29]


# ---------------------------------------------------------------------------- data parallel (gloo)
def _dp_worker(rank, world, port, q):
    import torch.distributed as dist
    from pearl_amd.policy_learners.sequential_decision_making.deep_q_learning import (
        allreduce_sum_, world_size)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank,
                            world_size=world)
    assert world_size() == world
    # what pa_dqn_step leaves in the flat buffer on each rank: local gradient / world
    local = torch.arange(8, dtype=torch.float32) * (rank + 1) / world
    out = allreduce_sum_(local.clone())
    q.put((rank, out.tolist()))
    dist.destroy_process_group()


def _dp_mlp_worker(rank, world, port, q):
    import torch.distributed as dist
    from pearl_amd.policy_learners.sequential_decision_making.flat_mlp import reduce_gradient_
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank,
                            world_size=world)
    g = torch.arange(6, dtype=torch.float32) * (rank + 1)      # rank r holds (r + 1) * g
    mean = reduce_gradient_(g.clone(), "mean").tolist()
    total = reduce_gradient_(g.clone(), "sum").tolist()
    q.put((rank, (mean, total)))
    dist.destroy_process_group()


def test_actor_critic_gradient_reduction_mean_and_sum():
    """FlatMlp.adam's data-parallel step (PPO / SAC / bandit trunk): "mean" for mean-reduced losses,
    "sum" for PPO's summed surrogate; identity without a process group."""
    import torch.multiprocessing as mp
    from pearl_amd.policy_learners.sequential_decision_making.flat_mlp import reduce_gradient_
    g = torch.arange(6, dtype=torch.float32)
    assert reduce_gradient_(g.clone(), "mean").tolist() == g.tolist()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_mlp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want_mean = (g * (1 + 2) / 2).tolist()
    want_sum = (g * (1 + 2)).tolist()
    for r in (0, 1):
        assert results[r] == (want_mean, want_sum)


def test_gradient_allreduce_is_a_mean_over_ranks():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = (torch.arange(8, dtype=torch.float32) * (1 + 2) / 2).tolist()   # mean of g and 2g
    assert results[0] == want and results[1] == want


def _native_loop_worker(rank, world, port, q):
    import torch.distributed as dist
    from pearl_amd import TD3, BoxActionSpace, ContinuousSoftActorCritic
    space = BoxActionSpace(-torch.ones(2), torch.ones(2))
    kw = dict(action_space=space, state_dim=5, actor_hidden_dims=[8, 8], critic_hidden_dims=[8, 8],
              batch_size=4, training_rounds=3)
    learners = [ContinuousSoftActorCritic(**kw), TD3(**kw)]
    alone = [pl._one_call_ok() for pl in learners]
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank,
                            world_size=world)
    # (the hook returns before it looks at the replay buffer)
    together = [(pl._one_call_ok(), pl._learn_native_loop(object(), 4)) for pl in learners]
    q.put((rank, (alone, together)))
    dist.destroy_process_group()


def test_native_learn_loops_step_aside_under_data_parallelism():
    """pa_sac_learn / pa_ddpg_learn (and the one-call steps under them) are single-process paths: a
    data-parallel step all-reduces between backward and AdamW, so with a process group of more
    than one rank learn() keeps the per-round, per-stage loop on every rank."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_native_loop_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in (0, 1):
        alone, together = results[r]
        assert alone == [True, True]
        assert together == [(False, None), (False, None)]


def test_td3_loop_report_repeats_the_last_actor_loss():
    """TD3's report carries the LAST actor loss on rounds without an actor step (td3.py:106-141);
    the one-call learn() reconstructs that from the per-round losses and the step counter, across
    calls."""
    from pearl_amd import TD3, BoxActionSpace
    pl = TD3(action_space=BoxActionSpace(-torch.ones(2), torch.ones(2)), state_dim=5,
             actor_hidden_dims=[8, 8], critic_hidden_dims=[8, 8], batch_size=4, training_rounds=5,
             actor_update_freq=2)
    pl._last_actor_loss = 7.0
    # training steps 1..5: actor steps on 2 and 4 (losses written there; zeros elsewhere)
    rep = pl._loop_report([[0.0, 1.5, 0.0, 2.5, 0.0], [10.0, 11.0, 12.0, 13.0, 14.0]], step0=0, freq=2)
    assert rep == {"actor_loss": [7.0, 1.5, 1.5, 2.5, 2.5], "critic_loss": [10.0, 11.0, 12.0, 13.0, 14.0]}
    assert pl._last_actor_loss == 2.5
    # next call starts after training step 5: steps 6 and 8 are due, 7 is not
    rep = pl._loop_report([[3.5, 0.0, 4.5], [1.0, 2.0, 3.0]], step0=5, freq=2)
    assert rep["actor_loss"] == [3.5, 3.5, 4.5]
    # a device scalar left by the per-round path is read once
    pl._last_actor_loss = torch.tensor(4.0)
    assert pl._loop_report([[0.0], [1.0]], step0=0, freq=2)["actor_loss"] == [4.0]


# ---------------------------------------------------------------------------- constructor parity
def _load_fx(name):
    return torch.load(os.path.join(REPO, "tests", "golden", f"{name}.pt"), map_location="cpu",
                      weights_only=False)


def _same_sd(module, want, what):
    sd = module.state_dict()
    assert list(sd) == list(want), what
    for k in sd:
        assert torch.equal(sd[k], want[k]), f"{what}.{k}"


def _box(fx):
    from pearl_amd import BoxActionSpace
    return BoxActionSpace(fx["low"], fx["high"])


def _disc(n):
    from pearl_amd import DiscreteActionSpace
    return DiscreteActionSpace([torch.tensor([k]) for k in range(n)])


@pytest.mark.parametrize("name,seed", [("sac_tiny", 6), ("sac_cfg3_shape_small", 6),
                                       ("ddpg_tiny", 16), ("td3_cfg3_shape_small", 16),
                                       ("dsac_tiny", 26), ("dsac_shape_small", 26),
                                       ("iql_continuous_tiny", 36), ("iql_discrete_tiny", 36),
                                       ("sarsa_tiny", 7), ("ppo_tiny", 5)])
def test_actor_critic_family_constructors_match_reference(name, seed):
    """Same torch seed -> the same initial networks as the reference's constructors (same modules
    created in the same order with the same initialisers), the same state_dict keys, the same
    optimizer hyper-parameters and learner defaults.  Runs on the CPU: no kernel is involved."""
    import pearl_amd as P
    from pearl_amd.neural_networks.sequential_decision_making.actor_networks import (
        VanillaActorNetwork, VanillaContinuousActorNetwork)
    fx = _load_fx(name)
    cfg = fx["config"]
    torch.manual_seed(seed)
    kind = name.split("_")[0]
    if kind == "sac":
        pl = P.ContinuousSoftActorCritic(action_space=_box(fx), state_dim=cfg["S"],
                                         actor_hidden_dims=cfg["hidden"],
                                         critic_hidden_dims=cfg["hidden"], batch_size=cfg["B"])
        assert (pl._actor_learning_rate, pl._critic_soft_update_tau, pl.on_policy) == (1e-3, 0.005, False)
    elif kind in ("ddpg", "td3"):
        cls = P.TD3 if kind == "td3" else P.DeepDeterministicPolicyGradient
        pl = cls(action_space=_box(fx), state_dim=cfg["S"], actor_hidden_dims=cfg["hidden"],
                 critic_hidden_dims=cfg["hidden"], batch_size=cfg["B"])
        assert pl._use_actor_target and pl._actor_soft_update_tau == 0.005
        # the fixture's targets were perturbed after construction; fresh targets copy the online nets
        _same_sd(pl._actor_target, fx["actor0"], "actor_target")
        if kind == "td3":
            assert (pl._actor_update_freq, pl._actor_update_noise, pl._actor_update_noise_clip) == (2, 0.2, 0.5)
    elif kind == "dsac":
        pl = P.SoftActorCritic(action_space=_disc(cfg["A"]), state_dim=cfg["S"],
                               actor_hidden_dims=cfg["hidden"], critic_hidden_dims=cfg["hidden"],
                               batch_size=cfg["B"],
                               action_representation_module=P.OneHotActionTensorRepresentationModule(cfg["A"]))
        g = pl._entropy_optimizer.param_groups[0]
        assert type(pl._entropy_optimizer).__name__ == "Adam" and (g["lr"], g["eps"]) == (1e-4, 1e-4)
        want = -0.89 * torch.log(1.0 / torch.tensor(cfg["A"]))
        assert torch.equal(pl._target_entropy, want)
    elif kind == "iql":
        kw = dict(state_dim=cfg["S"], actor_hidden_dims=cfg["hidden"], critic_hidden_dims=cfg["hidden"],
                  value_critic_hidden_dims=cfg["hidden"], batch_size=cfg["B"], expectile=cfg["expectile"])
        if cfg["continuous"]:
            pl = P.ImplicitQLearning(action_space=_box(fx),
                                     actor_network_type=VanillaContinuousActorNetwork, **kw)
        else:
            pl = P.ImplicitQLearning(action_space=_disc(cfg["A"]), actor_network_type=VanillaActorNetwork,
                                     action_representation_module=P.OneHotActionTensorRepresentationModule(cfg["A"]),
                                     **kw)
        _same_sd(pl._value_network, fx["value0"], "value")
        assert pl._critic_soft_update_tau == 0.05 and pl._is_action_continuous == cfg["continuous"]
    elif kind == "sarsa":
        pl = P.DeepSARSA(state_dim=cfg["S"], action_space=_disc(cfg["A"]), hidden_dims=cfg["hidden"],
                         training_rounds=cfg["rounds"], batch_size=cfg["B"],
                         action_representation_module=P.OneHotActionTensorRepresentationModule(cfg["A"]))
        _same_sd(pl._Q, fx["params0"], "Q")
        _same_sd(pl._Q_target, fx["target0"], "Q_target")
        assert pl.on_policy is True and pl._soft_update_tau == 0.1
        d = P.DeepSARSA(state_dim=3, action_space=_disc(2), hidden_dims=[4, 4],
                        action_representation_module=P.OneHotActionTensorRepresentationModule(2))
        assert (d._training_rounds, d._batch_size) == (100, 128)     # DeepTDLearning's defaults
        return
    else:
        pl = P.ProximalPolicyOptimization(
            action_space=_disc(cfg["A"]), state_dim=cfg["S"], actor_hidden_dims=cfg["hidden"],
            critic_hidden_dims=cfg["hidden"], training_rounds=cfg["rounds"], batch_size=cfg["B"],
            epsilon=cfg["epsilon"],
            action_representation_module=P.OneHotActionTensorRepresentationModule(cfg["A"]))
        assert pl.on_policy is True
    _same_sd(pl._actor, fx["actor0"], "actor")
    _same_sd(pl._critic, fx["critic0"], "critic")
    for opt in (pl._actor_optimizer, pl._critic_optimizer):
        g = opt.param_groups[0]
        assert type(opt).__name__ == "AdamW" and g["amsgrad"] is True


def _family(kind, seed):
    import pearl_amd as P
    from pearl_amd.neural_networks.sequential_decision_making.actor_networks import (
        VanillaContinuousActorNetwork)
    torch.manual_seed(seed)
    box = P.BoxActionSpace(-torch.ones(2), torch.ones(2))
    rep = P.OneHotActionTensorRepresentationModule(3)
    h = dict(actor_hidden_dims=[8, 8], critic_hidden_dims=[8, 8])
    return {
        "dqn": lambda: P.DeepQLearning(state_dim=4, action_space=_disc(3), hidden_dims=[8, 8],
                                       action_representation_module=rep),
        "ddqn": lambda: P.DoubleDQN(state_dim=4, action_space=_disc(3), hidden_dims=[8, 8],
                                    action_representation_module=rep),
        "sarsa": lambda: P.DeepSARSA(state_dim=4, action_space=_disc(3), hidden_dims=[8, 8],
                                     action_representation_module=rep),
        "ppo": lambda: P.ProximalPolicyOptimization(action_space=_disc(3), state_dim=4,
                                                    action_representation_module=rep, **h),
        "sac": lambda: P.ContinuousSoftActorCritic(action_space=box, state_dim=4, **h),
        "dsac": lambda: P.SoftActorCritic(action_space=_disc(3), state_dim=4,
                                          action_representation_module=rep, **h),
        "ddpg": lambda: P.DeepDeterministicPolicyGradient(action_space=box, state_dim=4, **h),
        "td3": lambda: P.TD3(action_space=box, state_dim=4, **h),
        "iql": lambda: P.ImplicitQLearning(action_space=box, state_dim=4,
                                           value_critic_hidden_dims=[8, 8],
                                           actor_network_type=VanillaContinuousActorNetwork, **h),
    }[kind]()


@pytest.mark.parametrize("kind", ["dqn", "ddqn", "sarsa", "ppo", "sac", "dsac", "ddpg", "td3", "iql"])
def test_state_dict_round_trip_and_compare(kind):
    """The reference's serialization contract (README.md:23-46, test_serialization.py:23-43,
    test_compare.py): differently initialised learners compare as different; torch.save ->
    torch.load -> load_state_dict(strict=True) makes compare() return ""; a learner of another class
    never compares equal."""
    import io
    from pearl_amd import BasicReplayBuffer, PearlAgent
    a = PearlAgent(_family(kind, 1), replay_buffer=BasicReplayBuffer(8))
    b = PearlAgent(_family(kind, 2), replay_buffer=BasicReplayBuffer(8))
    assert a.policy_learner.compare(b.policy_learner) != ""
    buf = io.BytesIO()
    torch.save(a.state_dict(), buf)
    buf.seek(0)
    b.load_state_dict(torch.load(buf, weights_only=False), strict=True)
    assert a.policy_learner.compare(b.policy_learner) == ""
    other = _family("dqn" if kind != "dqn" else "ppo", 3)
    assert a.policy_learner.compare(other) != ""


def test_padded_action_tables_survive_recycled_space_ids():
    """dynamic action spaces are rebuilt every step, CPython hands the freed
    address to the next one, and a cache keyed on id(space) alone returned the PREVIOUS space's
    table.  Freshly built spaces of equal n but different actions must each get their own table."""
    import gc
    from pearl_amd import BasicReplayBuffer, DiscreteActionSpace
    rb = BasicReplayBuffer(8)
    seen_ids = set()
    for step in range(60):
        acts = [torch.tensor([float(step * 10 + k)]) for k in range(3)]
        sp = DiscreteActionSpace(acts)
        seen_ids.add(id(sp))
        table, mask = rb._padded_tables(5, sp)
        assert table[:3, 0].tolist() == [step * 10.0, step * 10.0 + 1, step * 10.0 + 2], step
        assert mask.tolist() == [0, 0, 0, 1, 1]
        del sp
        gc.collect()
    # the same live object still hits the cache
    sp = DiscreteActionSpace([torch.tensor([1.0]), torch.tensor([2.0])])
    a = rb._padded_tables(4, sp)
    b = rb._padded_tables(4, sp)
    assert a[0] is b[0]


def test_optimizer_argument_is_honoured_or_refused():
    """deep_td_learning.py:183-185 / actor_critic_base.py:159-211 take a caller's optimizer.  The HIP
    step has torch.optim.AdamW's arithmetic with the hyper-parameters of the optimizer's group, so an
    AdamW (any lr / betas / eps / weight_decay / amsgrad) or an Adam without weight decay over the
    network's own parameters is used as handed over; anything else is refused, loudly."""
    from pearl_amd import (ContinuousSoftActorCritic, BoxActionSpace, DeepQLearning, DiscreteActionSpace,
                           OneHotActionTensorRepresentationModule)
    from pearl_amd.neural_networks.sequential_decision_making.q_value_networks import VanillaQValueNetwork
    sp = DiscreteActionSpace([torch.tensor([k]) for k in range(3)])
    kw = dict(action_space=sp, action_representation_module=OneHotActionTensorRepresentationModule(3))
    net = VanillaQValueNetwork(state_dim=4, action_dim=3, hidden_dims=[8, 8], output_dim=1)
    opt = torch.optim.AdamW(net.parameters(), lr=3e-4, betas=(0.8, 0.99), eps=1e-6, weight_decay=0.05)
    pl = DeepQLearning(network_instance=net, optimizer=opt, **kw)
    assert pl.optimizer is opt and pl._fused
    g = pl.optimizer.param_groups[0]
    assert (g["lr"], tuple(g["betas"]), g["eps"], g["weight_decay"], g["amsgrad"]) == (3e-4, (0.8, 0.99), 1e-6, 0.05, False)
    net = VanillaQValueNetwork(state_dim=4, action_dim=3, hidden_dims=[8, 8], output_dim=1)
    DeepQLearning(network_instance=net, optimizer=torch.optim.Adam(net.parameters(), lr=1e-3), **kw)
    for bad in (torch.optim.SGD(net.parameters(), lr=0.1),
                torch.optim.Adam(net.parameters(), lr=1e-3, weight_decay=0.01),
                torch.optim.AdamW(list(net.parameters())[:2], lr=1e-3),
                torch.optim.AdamW([{"params": list(net.parameters())[:2]}, {"params": list(net.parameters())[2:]}]),
                torch.optim.AdamW(net.parameters(), lr=1e-3, maximize=True)):
        with pytest.raises(NotImplementedError):
            DeepQLearning(network_instance=net, optimizer=bad, **kw)
    # actor-critic: the actor's and the critic's optimizers
    from pearl_amd.neural_networks.sequential_decision_making.actor_networks import GaussianActorNetwork
    from pearl_amd.neural_networks.sequential_decision_making.twin_critic import TwinCritic
    space = BoxActionSpace(-torch.ones(2), torch.ones(2))
    actor = GaussianActorNetwork(input_dim=5, hidden_dims=[8, 8], output_dim=2, action_space=space)
    critic = TwinCritic(state_dim=5, action_dim=2, hidden_dims=[8, 8])
    ao = torch.optim.AdamW(actor.parameters(), lr=2e-4, amsgrad=True)
    co = torch.optim.AdamW(critic.parameters(), lr=5e-4, weight_decay=0.0)
    sac = ContinuousSoftActorCritic(action_space=space, actor_network_instance=actor,
                                    critic_network_instance=critic, actor_optimizer=ao, critic_optimizer=co)
    assert sac._actor_optimizer is ao and sac._critic_optimizer is co
    assert sac._actor_learning_rate == 2e-4 and sac._critic_learning_rate == 5e-4
    with pytest.raises(NotImplementedError):
        ContinuousSoftActorCritic(action_space=space, actor_network_instance=actor,
                                  critic_network_instance=critic,
                                  actor_optimizer=torch.optim.RMSprop(actor.parameters()))


def test_force_pinv_is_honoured():
    """linear_regression.py:138-157: force_pinv inverts A + lambda I with torch.linalg.pinv.  Every
    such system runs pa_linreg_pinv (`uses_pinv`), whatever lambda is — torch's cut-off drops
    eigen-directions near lambda once max eig(A) > lambda / (D eps), where an SPD inverse keeps
    1 / lambda (bandit_pinv_tiny / bandit_pinv_singular_* pin it against the reference).
    The kernel holds systems up to order 72 — beyond that the learner says so, for any lambda."""
    from pearl_amd import NeuralLinearBandit
    from pearl_amd.neural_networks.contextual_bandit.linear_regression import LinearRegression
    assert LinearRegression(feature_dim=4, force_pinv=True).force_pinv
    assert LinearRegression(feature_dim=4, force_pinv=True).uses_pinv
    lr = NeuralLinearBandit(feature_dim=5, hidden_dims=[8, 4], force_pinv=True,
                            l2_reg_lambda_linear=0.0).model._linear_regression_layer
    assert lr.force_pinv and lr.uses_pinv
    assert not LinearRegression(feature_dim=4, l2_reg_lambda=0.0).uses_pinv
    for lam in (0.0, 1.0):
        with pytest.raises(NotImplementedError, match="feature_dim"):
            LinearRegression(feature_dim=72, l2_reg_lambda=lam, force_pinv=True)


def test_neural_linear_regression_without_e2e_head_predicts_from_the_regression():
    """NeuralLinearRegression(nn_e2e=False) (neural_linear_regression.py:100-105, :140-147): mu is the
    LinUCB regression's [1 | features] coefs, not linear_layer_e2e's output — the act-time torch
    expression of the mode whose learner step the HIP engine runs with a frozen last layer."""
    from pearl_amd.neural_networks.contextual_bandit.linear_regression import NeuralLinearRegression
    torch.manual_seed(3)
    model = NeuralLinearRegression(feature_dim=7, hidden_dims=[12, 6], nn_e2e=False,
                                   output_activation_name="sigmoid")
    model._linear_regression_layer._coefs.copy_(torch.randn(7))
    x = torch.randn(5, 7)
    out = model.forward_with_intermediate_values(x)
    feats = out["nn_output"]
    want = torch.cat((torch.ones(5, 1), feats), dim=1) @ model._linear_regression_layer._coefs
    torch.testing.assert_close(out["pred_label_pre_activation"].view(-1), want)
    torch.testing.assert_close(out["pred_label"].view(-1), torch.sigmoid(want))
    e2e = NeuralLinearRegression(feature_dim=7, hidden_dims=[12, 6])
    out2 = e2e.forward_with_intermediate_values(x)
    torch.testing.assert_close(out2["pred_label_pre_activation"], e2e.linear_layer_e2e(out2["nn_output"]))
