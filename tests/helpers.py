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


def assert_linear_solve_close(coefs, A, b, lam, want_coefs, max_backward=2e-6, msg=""):
    """`coefs = inv(A + lam I) b` of the LinUCB layer (linear_regression.py:252-257) against the
    reference's.  A linear solve has no meaningful elementwise tolerance: the reference inverts in
    fp32 (LAPACK getrf/getri) and is itself off by cond(A + lam I) * 2^-24 from the exact solution
    (1.8e-4 normwise at BASELINE config 5's batch, cond 1.5e4).  So:
      (1) the result under test must solve ITS OWN system: normwise backward error
          |M x - b|_inf / (|M|_inf |x|_inf + |b|_inf) <= max_backward, evaluated in fp64 — tight and
          independent of conditioning (2e-6: an fp32-rounded exact solution gives ~1e-7);
      (2) against the reference: normwise forward difference <= 4 cond_2(M) 2^-24 — what the
          conditioning gives two correct fp32-level solvers on inputs equal to rounding."""
    x = coefs.detach().cpu().double().view(-1)
    M = A.detach().cpu().double() + lam * torch.eye(A.shape[0], dtype=torch.float64)
    bb = b.detach().cpu().double().view(-1)
    eta = float((M @ x - bb).abs().max() / (M.abs().sum(1).max() * x.abs().max() + bb.abs().max()))
    assert eta <= max_backward, f"{msg}: backward error {eta:.3e} > {max_backward:.1e}"
    want = want_coefs.detach().cpu().double().view(-1)
    cond = float(torch.linalg.cond(M))
    fwd = float((x - want).abs().max() / want.abs().max())
    bound = 4.0 * cond * 2.0 ** -24
    assert fwd <= bound, f"{msg}: forward difference {fwd:.3e} > 4 cond eps = {bound:.3e} (cond {cond:.3g})"
    return eta, fwd, cond


def free_port() -> int:
    """A TCP port nothing listens on right now (bind to 0 and read it back): rendezvous ports picked
    with ``random`` collide once an earlier test has seeded the global generator."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return int(s.getsockname()[1])
