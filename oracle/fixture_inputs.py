"""Portable seeded inputs for the LARGE reference-minted fixtures.

TEST INFRASTRUCTURE ONLY (imported by oracle/make_golden*.py when the fixtures are minted from
/root/reference, and by tests/ when they are checked — never by pearl_amd/).

A fixture at BASELINE.json's own sizes (bandit: 4096 x 512 contexts per step; PPO: a 65 536 x 256
rollout) would put tens of megabytes of *inputs* into tests/golden/.  Instead the inputs are
regenerated wherever the fixture is used — which needs a generator whose output does not depend on
the host's SIMD level or BLAS build (torch.randn's vectorised Box-Muller and MKL's blocked sums are
not that): everything below is integer hashing (numpy uint64, wrap-around) followed by a fixed
sequence of elementwise IEEE float32 operations, bit-identical on every machine.  The fixture
stores `checksum(x)` of what the reference was fed, and the tests assert it on what they rebuilt.
"""
from __future__ import annotations

import numpy as np
import torch


def _splitmix64(idx: np.ndarray, seed: int) -> np.ndarray:
    """splitmix64 of (seed, index) — uint64 in, uint64 out, wrap-around arithmetic."""
    with np.errstate(over="ignore"):
        z = idx.astype(np.uint64) + np.uint64((seed * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def uniform(shape, seed: int) -> torch.Tensor:
    """float32 in [0, 1) with 24 random bits per element (exactly representable)."""
    n = int(np.prod(shape))
    h = _splitmix64(np.arange(n, dtype=np.uint64), seed)
    u = (h >> np.uint64(40)).astype(np.float32) * np.float32(2.0 ** -24)
    return torch.from_numpy(u.reshape(shape))


def normalish(shape, seed: int) -> torch.Tensor:
    """Zero-mean, unit-variance, bell-shaped float32 values: (u1 + u2 + u3 + u4 - 2) * sqrt(3) from
    four independent 24-bit uniforms, summed left to right in float32."""
    n = int(np.prod(shape))
    idx = np.arange(n, dtype=np.uint64)
    acc = np.zeros(n, dtype=np.float32)
    for k in range(4):
        h = _splitmix64(idx, seed * 4 + k + 1)
        acc = acc + (h >> np.uint64(40)).astype(np.float32) * np.float32(2.0 ** -24)
    out = (acc - np.float32(2.0)) * np.float32(1.7320508075688772)
    return torch.from_numpy(out.reshape(shape))


def integers(shape, seed: int, high: int) -> torch.Tensor:
    """int64 in [0, high)."""
    n = int(np.prod(shape))
    h = _splitmix64(np.arange(n, dtype=np.uint64), seed)
    return torch.from_numpy(((h >> np.uint64(33)) % np.uint64(high)).astype(np.int64).reshape(shape))


def checksum(x: torch.Tensor) -> int:
    """Order-independent 64-bit checksum of the raw bits (float32 / int64 / bool tensors)."""
    a = x.detach().cpu().contiguous().numpy()
    if a.dtype == np.float32:
        bits = a.view(np.uint32).astype(np.uint64).ravel()
    else:
        bits = a.astype(np.int64).view(np.uint64).ravel()
    with np.errstate(over="ignore"):
        mixed = _splitmix64(bits + np.arange(bits.size, dtype=np.uint64) * np.uint64(0x100000001B3), 17)
        return int(np.bitwise_xor.reduce(mixed)) if mixed.size else 0


# ----------------------------------------------------------------------------------------------
# the inputs of the seeded fixtures
# ----------------------------------------------------------------------------------------------
def bandit_contexts(cfg, step: int) -> torch.Tensor:
    """(B, F) contexts of learn_batch call `step` of a seeded NeuralLinearBandit fixture."""
    return normalish((cfg["B"], cfg["F"]), cfg["input_seed"] * 100 + step)


def ppo_rollout(cfg):
    """The rollout of a seeded PPO fixture: states (N + 1, S), actions (N,), rewards (N,),
    terminated / truncated (N,) — episode ends every 97th / 131st transition."""
    N, S, A, seed = cfg["N"], cfg["S"], cfg["A"], cfg["input_seed"]
    states = normalish((N + 1, S), seed * 100 + 1)
    actions = integers((N,), seed * 100 + 2, A)
    rewards = normalish((N,), seed * 100 + 3)
    i = torch.arange(N)
    return states, actions, rewards, (i % 97 == 96), (i % 131 == 57)


def sac_box(cfg):
    """The action box of a seeded continuous-SAC fixture (asymmetric, per-dimension bounds)."""
    A = cfg["A"]
    return -torch.ones(A) * torch.linspace(1.0, 2.0, A), torch.ones(A) * torch.linspace(1.5, 1.0, A)


def sac_transitions(cfg):
    """The transitions of a seeded continuous-SAC fixture: states (N + 1, S), actions (N, A) inside
    the box, rewards (N,), terminated (N,) — every 11th transition ends an episode."""
    N, S, A, seed = cfg["N"], cfg["S"], cfg["A"], cfg["input_seed"]
    low, high = sac_box(cfg)
    states = normalish((N + 1, S), seed * 100 + 1)
    actions = low + (high - low) * uniform((N, A), seed * 100 + 2)
    rewards = normalish((N,), seed * 100 + 3)
    return states, actions, rewards, (torch.arange(N) % 11 == 10)


def divergence(got, want):
    """Per-tensor statistics of how far two parameter sets are apart: the yardstick the long-run
    fixtures carry (twin vs reference) and the tests recompute (HIP vs reference)."""
    out = {}
    for k, w in want.items():
        g = got[k].detach().cpu().double().view(-1)
        w = w.detach().cpu().double().view(-1)
        if w.numel() == 0:
            continue
        d = (g - w).abs()
        out[k] = dict(max_abs=float(d.max()), mean_abs=float(d.mean()),
                      rms=float(d.pow(2).mean().sqrt()), ref_rms=float(w.pow(2).mean().sqrt()),
                      ref_max=float(w.abs().max()))
    return out
