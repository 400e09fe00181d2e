"""The reference's own unit tests of the learners on the path, on the HIP learners.

test/unit/with_pytorch/test_deep_td_learning.py:29-105 (DoubleDQN / DeepSARSA next-state values on a network
with ONE hidden layer of width 3) and

test/unit/with_pytorch/test_neural_linear_bandits.py:41-245 (NeuralLinearBandit), restated for pearl_amd's classes on ``cuda:0``:
state-dict exactness (rtol = atol = 0), 1000 ``learn_batch`` calls on y = sum(x) reaching the reference's loss
thresholds for the MSE / MAE / cross-entropy criteria — with the reference's one-row, zero-weight "null batch"
as the second call and its dropout ratio of 1e-4 —, the ``get_scores`` / ``act`` shapes under UCB exploration,
and discounting of the regression moments.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
NUM_EPOCHS = 1000


def _batch(state, reward):
    from pearl_amd import TransitionBatch
    return TransitionBatch(state=state, action=torch.zeros_like(state[:, -1:]), reward=reward,
                           weight=torch.ones(state.shape[0], 1, device=state.device))


def _null_batch(feature_dim):
    from pearl_amd import TransitionBatch
    return TransitionBatch(state=torch.zeros(1, feature_dim, device=DEV),
                           action=torch.empty(1, 0, device=DEV),
                           reward=torch.zeros(1, 1, device=DEV), weight=torch.zeros(1, 1, device=DEV))


def _learner(**kw):
    from pearl_amd import NeuralLinearBandit, UCBExploration
    kw.setdefault("feature_dim", 15)
    kw.setdefault("hidden_dims", [32, 32])
    return NeuralLinearBandit(learning_rate=0.01, exploration_module=UCBExploration(alpha=0.1), **kw).to(DEV)


def test_state_dict_round_trip_is_exact():
    """:41-92"""
    torch.manual_seed(0)
    pl = _learner()
    state = torch.randn(60, 15, device=DEV)
    pl.learn_batch(_batch(state, state.sum(-1, keepdim=True)))
    cp = _learner()
    cp.load_state_dict(pl.state_dict())
    a, b = cp.model._linear_regression_layer, pl.model._linear_regression_layer
    assert torch.equal(a._A, b._A) and torch.equal(a._b, b._b)
    assert a._A.abs().sum() > 0
    for p1, p2 in zip(cp.model._nn_layers.parameters(), pl.model._nn_layers.parameters()):
        assert torch.equal(p1, p2)
    # ... and the copy continues like the original
    nxt = _batch(state, state.sum(-1, keepdim=True))
    assert float(pl.learn_batch(nxt)["loss"]) == float(cp.learn_batch(nxt)["loss"])


@pytest.mark.parametrize("loss_type,activation", [("mse", "linear"), ("mae", "linear"),
                                                  ("cross_entropy", "sigmoid")])
def test_neural_linucb_learns_the_linear_reward(loss_type, activation):
    """:96-208"""
    from pearl_amd import DiscreteActionSpace
    torch.manual_seed(0)
    feature_dim, batch_size = 15, 60
    pl = _learner(dropout_ratio=0.0001, loss_type=loss_type, output_activation_name=activation)
    assert pl.feature_dim == feature_dim
    state = torch.randn(batch_size, feature_dim, device=DEV)
    reward = state.sum(-1, keepdim=True)
    if activation == "sigmoid":
        reward = torch.sigmoid(reward)
    batch = _batch(state, reward)
    losses = []
    for i in range(NUM_EPOCHS):
        b = _null_batch(feature_dim) if i == 1 else batch     # "can happen from DisjointBandit"
        losses.append(float(pl.learn_batch(b)["loss"]))
    assert all(x == x for x in losses)
    print(f"\n[bandit {loss_type}] loss {losses[0]:.4g} -> {losses[-1]:.4g}")
    if loss_type == "mse":
        assert losses[-1] < 1e-1
    elif loss_type == "mae":
        assert losses[-1] ** 2 < 1e-1
    else:
        assert losses[-1] < losses[0]
    space = DiscreteActionSpace(actions=list(batch.action.cpu()))
    scores = pl.get_scores(subjective_state=batch.state, action_space_to_score=space)
    assert scores.shape == (batch_size, batch_size)
    action = pl.act(subjective_state=state[0], available_action_space=space)
    assert int(action.reshape(-1)[0]) in range(batch_size)
    action = pl.act(subjective_state=state, available_action_space=space)
    assert action.shape == (batch_size, 1)


def test_ucb_scores_are_values_plus_alpha_sigma():
    """ucb_exploration.py:58-95 on the HIP head: score - value = alpha * sqrt(x^T A^-1 x) of the trunk's
    features, checked against float64 from the learner's own A."""
    from pearl_amd import DiscreteActionSpace
    torch.manual_seed(1)
    pl = _learner()
    state = torch.randn(60, 15, device=DEV)
    for _ in range(3):
        pl.learn_batch(_batch(state, state.sum(-1, keepdim=True)))
    space = DiscreteActionSpace([torch.tensor([0.0]), torch.tensor([1.0]), torch.tensor([2.0])])
    scores = pl.get_scores(subjective_state=state, action_space_to_score=space)
    assert scores.shape == (60, 3)
    with torch.no_grad():
        ret = pl.model.forward_with_intermediate_values(state)
    feats = torch.cat([torch.ones(60, 1, device=DEV), ret["nn_output"]], dim=1).double()
    A = pl.model._linear_regression_layer.A.double()
    sigma = torch.sqrt(torch.einsum("bi,ij,bj->b", feats, torch.linalg.inv(A), feats))
    want = ret["pred_label_pre_activation"].reshape(-1).double() + 0.1 * sigma
    torch.testing.assert_close(scores[:, 0].double(), want, rtol=1e-4, atol=1e-5)
    assert torch.equal(scores[:, 0], scores[:, 1])      # state features only: every arm scores alike


def test_discounting_shrinks_the_moments():
    """:210-245: gamma = 0.95 applied every 100 units of weight, a trunk with skip connections."""
    torch.manual_seed(0)
    pl = _learner(feature_dim=10, hidden_dims=[16, 16], use_skip_connections=True, gamma=0.95,
                  apply_discounting_interval=100.0)
    state = torch.randn(100, 10, device=DEV)
    batch = _batch(state, torch.exp(state.sum(-1, keepdim=True)))
    for _ in range(100):
        pl.learn_batch(batch)
    lr = pl.model._linear_regression_layer
    assert float(lr.A[0, 0]) < 100 * float(batch.weight.sum())
    assert float(lr._b[0]) < 100 * float((batch.reward * batch.weight).sum())
    assert float(lr.A[0, 0]) > 0


# ---- test_deep_td_learning.py ---------------------------------------------------------------------
def _td_batch():
    from pearl_amd import BasicReplayBuffer, DiscreteActionSpace
    space = DiscreteActionSpace(actions=list(torch.arange(3).view(-1, 1)), seed=0)
    rb = BasicReplayBuffer(24, sampler="python")
    rb.device_for_batches = torch.device(DEV)
    for _ in range(24):
        rb.push(state=torch.randn(10), action=space.sample(), reward=torch.randint(1, (1,)),
                next_state=torch.randn(10), curr_available_actions=space, next_available_actions=space,
                terminated=False, truncated=False, max_number_actions=3)
    return space, rb


def test_double_dqn_next_state_values_differ_from_dqn_with_the_same_weights():
    """:56-91 — hidden_dims=[3]: one hidden layer, three units (the generic TD engine's smallest case)."""
    import copy
    import random
    from pearl_amd import DeepQLearning, DoubleDQN, OneHotActionTensorRepresentationModule
    torch.manual_seed(0)
    random.seed(0)
    space, rb = _td_batch()
    batch = rb.sample(24)
    mk = lambda cls: cls(state_dim=10, action_space=space, hidden_dims=[3], training_rounds=1,
                         action_representation_module=OneHotActionTensorRepresentationModule(
                             max_number_actions=3)).to(DEV)
    ddqn, dqn = mk(DoubleDQN), mk(DeepQLearning)
    differ = False
    for _ in range(10):
        for net in (ddqn._Q, ddqn._Q_target):
            for m in net.modules():
                if isinstance(m, torch.nn.Linear):
                    torch.nn.init.xavier_normal_(m.weight)
        double_value = ddqn.get_next_state_values(ddqn.preprocess_batch(copy.deepcopy(batch)), 24)
        dqn._Q.load_state_dict(ddqn._Q.state_dict())
        dqn._Q_target.load_state_dict(ddqn._Q_target.state_dict())
        vanilla_value = dqn.get_next_state_values(dqn.preprocess_batch(copy.deepcopy(batch)), 24)
        assert double_value.shape == vanilla_value.shape == (24,)
        # Q_target(s', argmax_a Q(s', a)) <= max_a Q_target(s', a), row by row
        assert bool(torch.all(double_value <= vanilla_value + 1e-6))
        differ = bool(torch.any(double_value != vanilla_value))
        if differ:
            break
    assert differ


def test_sarsa_next_state_values_shape():
    """:93-105"""
    import random
    from pearl_amd import DeepSARSA, OneHotActionTensorRepresentationModule
    from pearl_amd.policy_learners.exploration import EGreedyExploration
    torch.manual_seed(0)
    random.seed(0)
    space, rb = _td_batch()
    batch = rb.sample(24)
    batch.next_action = batch.action
    sarsa = DeepSARSA(state_dim=10, action_space=space, hidden_dims=[3], training_rounds=1,
                      exploration_module=EGreedyExploration(0.05),
                      action_representation_module=OneHotActionTensorRepresentationModule(
                          max_number_actions=3)).to(DEV)
    v = sarsa.get_next_state_values(batch=sarsa.preprocess_batch(batch), batch_size=24)
    assert v.shape == (24,)
