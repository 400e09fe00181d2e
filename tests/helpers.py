"""Shared builders for the parity tests (oracle side and pearl_amd side) from a golden fixture."""
import torch

from oracle.pearl_oracle import DqnOracle, ReplayOracle


def fill_oracle_replay(fx) -> ReplayOracle:
    cfg, rows, states = fx["config"], fx["rows"], fx["states"]
    rb = ReplayOracle(cfg["N"] + 10)
    for i in range(cfg["N"]):
        rb.push(states[i], torch.tensor([int(rows["action"][i])]), float(rows["reward"][i]),
                bool(rows["terminated"][i]), bool(rows["truncated"][i]), int(rows["n_curr"][i]),
                states[i + 1], int(rows["n_next"][i]), cfg["A"])
    return rb


def oracle_learner(fx) -> DqnOracle:
    return DqnOracle(fx["params0"], fx["target0"],
                     double_q=fx["config"].get("learner") == "double")


def batch_pre_as_oracle_dict(fx):
    return {k: v for k, v in fx["batch_pre"].items()}


# the pure (state, action) functions oracle/make_golden.py gave the reference's HER buffer
def her_reward(state, action):
    g = state.shape[0] // 2
    return 0.0 if float((state[:g] - state[g:]).abs().sum()) < 1.5 else -1.0


def her_terminated(state, action):
    g = state.shape[0] // 2
    return bool(float((state[:g] - state[g:]).abs().sum()) < 0.75)
