"""One RCCL communicator per process for the native all-reduce hooks (``pa_comm_*``, comm.hip).

Data parallelism is not in the reference: every rank owns a replay shard and a
local batch, parameters stay replicated, and the only exchange is the flat gradient buffer once per
step.  ``native_comm`` brings the communicator up once (rank 0 mints the RCCL unique id,
``torch.distributed`` broadcasts its 128 bytes, every rank agrees on the outcome) and hands the
same handle to every learner of the process; ``allreduce_sum_`` reduces a flat fp32 tensor in place
on torch's current stream through it, or through ``torch.distributed.all_reduce`` when RCCL cannot
be loaded, the backend is gloo (CPU tests, two ranks on one GPU) or
``PEARL_AMD_TORCH_ALLREDUCE=1``.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch
import torch.distributed as dist

from . import _native as N

_state = {"handle": None, "failed": False, "device": None, "kind": None}


def world_size() -> int:
    return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1


def native_comm(dev: torch.device) -> Optional[C.c_void_p]:
    if os.environ.get("PEARL_AMD_TORCH_ALLREDUCE") == "1":
        return None
    if not (dist.is_available() and dist.is_initialized()):
        return None
    if _state["handle"] is not None:
        return _state["handle"]
    if _state["failed"]:
        return None     # RCCL bring-up failed on some rank: torch.distributed instead
    lib = N.lib()
    if os.environ.get("PEARL_AMD_P2P") == "1":
        return _p2p_comm(dev)
    if dist.get_backend() != "nccl" or not lib.pa_comm_available():
        return None
    rank, world = dist.get_rank(), dist.get_world_size()
    ident = torch.zeros(128, dtype=torch.uint8)
    if rank == 0:
        buf = (C.c_char * 128)()
        N.check(lib.pa_comm_unique_id(buf))
        ident = torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8).clone()
    ident = ident.to(dev)
    dist.broadcast(ident, src=0)
    raw = bytes(ident.cpu().numpy().tobytes())
    handle = C.c_void_p()
    torch.cuda.synchronize(dev)
    rc = lib.pa_comm_create(C.byref(handle), dev.index, world, rank, raw)
    # every rank must take the same path: agree on the outcome before using the communicator
    ok = torch.tensor([1 if rc == 0 else 0], dtype=torch.int32, device=dev)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if int(ok.item()) == 0:
        if rc == 0:
            lib.pa_comm_destroy(handle)
        _state["failed"] = True
        return None
    # RCCL sets a communicator up on its first collective (channels, proxies: up to seconds with
    # 8 ranks): do that here, not inside the first learner step
    warm = torch.zeros(256, dtype=torch.float32, device=dev)
    N.check(lib.pa_comm_allreduce_start(handle, warm.data_ptr(), warm.numel(), N.stream_ptr(dev)))
    N.check(lib.pa_comm_allreduce_wait(handle, N.stream_ptr(dev)))
    torch.cuda.synchronize(dev)
    _state["handle"], _state["device"], _state["kind"] = handle, dev, "rccl"
    return handle


def _p2p_comm(dev: torch.device) -> Optional[C.c_void_p]:
    """PEARL_AMD_P2P=1: the one-shot peer-to-peer gradient exchange (comm.hip) behind
    the same hooks — every rank allocates its exchange buffer, the 64-byte hipIpc handles travel
    through torch.distributed (any backend), every rank maps its peers' buffers, and all ranks
    agree on the outcome before the first exchange.  Messages of up to PEARL_AMD_P2P_FLOATS floats
    (default 2 M = 8 MB per slot); world size <= 8 (one node)."""
    lib = N.lib()
    rank, world = dist.get_rank(), dist.get_world_size()
    handle = C.c_void_p()
    ok_local = world <= 8
    if ok_local:
        floats = int(os.environ.get("PEARL_AMD_P2P_FLOATS", str(2 << 20)))
        ok_local = lib.pa_comm_create_p2p(C.byref(handle), dev.index, world, rank, floats) == 0
    mine = (C.c_char * 64)()
    if ok_local:
        ok_local = lib.pa_comm_p2p_handle(handle, mine) == 0
    everyone = [None] * world
    dist.all_gather_object(everyone, (bool(ok_local), bytes(mine.raw)))
    ok = all(e[0] for e in everyone)
    if ok:
        for r, (_, raw) in enumerate(everyone):
            if r != rank and lib.pa_comm_p2p_open(handle, r, raw) != 0:
                ok = False
    verdict = [None] * world
    dist.all_gather_object(verdict, bool(ok))
    if not all(verdict):
        if handle:
            lib.pa_comm_destroy(handle)
        _state["failed"] = True
        return None
    # one exchange outside any learner step: first-use costs, and a check that every peer answers
    warm = torch.ones(256, dtype=torch.float32, device=dev)
    N.check(lib.pa_comm_allreduce_start(handle, warm.data_ptr(), warm.numel(), N.stream_ptr(dev)))
    torch.cuda.synchronize(dev)
    N.check(lib.pa_comm_p2p_check(handle))
    assert float(warm[0]) == float(world), "P2P exchange: the warm-up sum is wrong"
    _state["handle"], _state["device"], _state["kind"] = handle, dev, "p2p"
    return handle


def _forget_comm() -> None:
    """tests: drop (and destroy) the process's communicator so that the next call builds a new one"""
    h = _state["handle"]
    if h is not None:
        N.lib().pa_comm_destroy(h)
    _state.update(handle=None, failed=False, device=None, kind=None)


def comm_info() -> dict:
    """What the gradient exchange of this process runs on (bench lines report it): the rank count
    RCCL itself reports for the communicator, not the one that was asked for."""
    info = {"ranks_requested": world_size()}
    h = _state["handle"]
    if h is None:
        info["library"] = "torch.distributed all_reduce" if world_size() > 1 or (
            dist.is_available() and dist.is_initialized()) else "none (single process)"
        return info
    n_seen, me = C.c_int32(-1), C.c_int32(-1)
    N.check(N.lib().pa_comm_info(h, C.byref(n_seen), C.byref(me)))
    lib_name = ("one-shot P2P over hipIpc-mapped peer buffers (native pa_comm_* hooks)"
                if _state["kind"] == "p2p" else "rccl (native pa_comm_* hooks)")
    info.update(library=lib_name, ranks_observed=n_seen.value, rank=me.value)
    return info


def check_exchange() -> None:
    """Raise if the process's native communicator has been poisoned (the P2P exchange's bounded
    wait for a peer expired: that round's gradient is NaN, the ranks are out of lock-step).  One
    read of a pinned host word; the learners call it after the host sync that ends a data-parallel
    ``learn()`` / step, so a lagging or dead peer is an exception at the call that hit it, not
    silently diverging replicas.  RCCL communicators never report: RCCL blocks instead."""
    h = _state["handle"]
    if h is not None:
        N.check(N.lib().pa_comm_check(h))


def check_exchange_after_sync() -> None:
    """``check_exchange`` for the per-step data-parallel paths (FlatMlp.adam -> allreduce_sum_): a
    no-op unless an exchange went through the native communicator since the last check.  Call it
    AFTER a host synchronisation that covers that exchange (a loss ``.item()``, the ``tolist()`` at
    the end of learn()): the poison word is written by the device, and only a completed step can
    have set it.  Without a following sync the check happens before the next exchange at the
    latest (``allreduce_sum_`` below)."""
    if _state.get("unchecked"):
        _state["unchecked"] = False
        check_exchange()


def allreduce_sum_(flat: torch.Tensor, force: bool = False) -> torch.Tensor:
    """SUM over the data-parallel group, in place.  Identity with a single rank unless ``force``
    (a 1-rank communicator still goes through RCCL: the bench's readiness run).  ONE message through
    RCCL; the P2P exchange takes messages of up to its slot size (``pa_comm_max_floats``) at 16-byte
    aligned addresses: longer ones go in slot-sized pieces, misaligned ones through
    ``torch.distributed.all_reduce`` — sizes and offsets are the same on every rank, so every rank
    takes the same route."""
    if world_size() <= 1 and not force:
        return flat
    if not (dist.is_available() and dist.is_initialized()):
        return flat
    h = native_comm(flat.device) if flat.is_cuda else None
    if h is not None and flat.dtype == torch.float32 and flat.is_contiguous():
        lib = N.lib()
        N.check(lib.pa_comm_check(h))      # an earlier exchange's expired wait surfaces here at the latest
        s = N.stream_ptr(flat.device)
        cap = int(lib.pa_comm_max_floats(h))
        n, ptr = flat.numel(), flat.data_ptr()
        if cap <= 0 or n <= cap:
            pieces = [(0, n)]
        else:
            pieces = [(o, min(cap, n - o)) for o in range(0, n, cap)]     # (cap is a multiple of 64 floats)
        if cap <= 0 or ptr % 16 == 0:
            _state["unchecked"] = True
            for off, cnt in pieces:
                N.check(lib.pa_comm_allreduce_start(h, ptr + 4 * off, cnt, s))
                N.check(lib.pa_comm_allreduce_wait(h, s))
            return flat
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    return flat
