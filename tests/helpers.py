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


def assert_adam_trajectory_close(got, want, lr, steps, rtol=1e-3, atol=2e-5, max_outlier_frac=2e-3,
                                 msg=""):
    """Parameters after `steps` AdamW steps against the reference run.

    AdamW moves a parameter by lr * m / (sqrt(v) + eps): for an element whose gradient is at the
    level of fp32 summation noise (dead-ish ReLU paths, near-cancelling sums over a large batch)
    that ratio is a coin flip of size ~lr per step whatever the arithmetic — MKL's blocked sums
    and the MFMA k-ordered chain legitimately disagree there, and so do two BLAS builds.  So:
    at most `max_outlier_frac` of the elements may miss (rtol, atol), and none of those by more
    than the 2.5 * lr * steps such flips can accumulate."""
    import torch
    got, want = got.detach().cpu().float(), want.detach().cpu().float()
    if got.numel() == 0:
        return
    bad = ~torch.isclose(got, want, rtol=rtol, atol=atol)
    frac = float(bad.float().mean())
    assert frac <= max_outlier_frac, f"{msg}: {frac:.4%} of the elements outside rtol={rtol}, atol={atol}"
    if bad.any():
        worst = float((got - want).abs()[bad].max())
        assert worst <= 2.5 * lr * steps, f"{msg}: outlier of {worst:.3e} > 2.5 * lr * steps"
