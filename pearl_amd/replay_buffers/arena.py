"""Thin object wrapper over the ``pa_arena_*`` C ABI (include/pearl_amd.h).

The arena is the MI355X replacement of the reference's ``deque`` of per-transition CPU
tensors (pearl/replay_buffers/tensor_based_replay_buffer.py:25-36): a structure-of-arrays
ring in HBM with a pinned staging ring for ``push``.  All compute goes through
libpearl_amd.so; nothing here has a CPU path.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch

from .. import _native as N


@dataclass(frozen=True)
class ArenaLayout:
    """What one stored transition looks like (fixed by the first push)."""
    state_shape: tuple
    action_shape: tuple       # shape of one pushed action tensor; () for 0-d actions
    action_dtype: torch.dtype
    reward_dtype: torch.dtype
    max_actions: int          # 0 = no available-action tables (continuous actions)
    avail_dim: int
    has_next_state: bool
    has_cost: bool

    @property
    def state_dim(self) -> int:
        return int(np.prod(self.state_shape)) if len(self.state_shape) else 1

    @property
    def action_elems(self) -> int:
        return int(np.prod(self.action_shape)) if len(self.action_shape) else 1


class HbmArena:
    def __init__(self, capacity: int, layout: ArenaLayout, device: torch.device,
                 staging_rows: int = 0) -> None:
        N.require_gpu()
        if device.type != "cuda":
            raise N.NativeError(
                f"pearl_amd: the replay arena lives in HBM; device_for_batches={device} is not a "
                "HIP device and there is no CPU fallback")
        self.capacity = int(capacity)
        self.layout = layout
        self.device = torch.device("cuda", device.index if device.index is not None
                                   else torch.cuda.current_device())
        desc = N.ArenaDesc(
            capacity=self.capacity, device=self.device.index, state_dim=layout.state_dim,
            action_elems=layout.action_elems, action_dtype=N.pa_dtype_of(layout.action_dtype),
            reward_dtype=N.pa_dtype_of(layout.reward_dtype), max_actions=layout.max_actions,
            avail_dim=layout.avail_dim, has_next_state=int(layout.has_next_state),
            has_cost=int(layout.has_cost), staging_rows=int(staging_rows))
        handle = C.c_void_p()
        N.check(N.lib().pa_arena_create(C.byref(handle), C.byref(desc)))
        self._h = handle
        self._idx_scratch: Optional[torch.Tensor] = None

    # -- lifetime ------------------------------------------------------------
    def close(self) -> None:
        h, self._h = getattr(self, "_h", None), None
        if h:
            N.lib().pa_arena_destroy(h)

    def __del__(self) -> None:  # pragma: no cover - interpreter shutdown order
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self) -> C.c_void_p:
        assert self._h, "arena was closed"
        return self._h

    def __len__(self) -> int:
        return int(N.lib().pa_arena_len(self.handle))

    @property
    def head(self) -> int:
        return int(N.lib().pa_arena_head(self.handle))

    def clear(self) -> None:
        N.check(N.lib().pa_arena_clear(self.handle))

    def _stream(self) -> Optional[int]:
        return N.stream_ptr(self.device)

    # -- ingest ----------------------------------------------------------------
    def push_row(self, t: N.Transition) -> None:
        N.check(N.lib().pa_arena_push(self.handle, C.byref(t)))

    def push_columns(self, n: int, cols: N.Columns, on_device: bool) -> None:
        if on_device:
            N.check(N.lib().pa_arena_push_many_device(self.handle, n, C.byref(cols), self._stream()))
        else:
            N.check(N.lib().pa_arena_push_many(self.handle, n, C.byref(cols)))

    def flush(self) -> None:
        N.check(N.lib().pa_arena_flush(self.handle, self._stream()))

    # -- sampling ----------------------------------------------------------------
    def _scratch(self, B: int) -> torch.Tensor:
        if self._idx_scratch is None or self._idx_scratch.numel() < B:
            self._idx_scratch = torch.empty(max(B, 1024), dtype=torch.int64, device=self.device)
        return self._idx_scratch

    def gather(self, logical_idx: np.ndarray, out: N.BatchOut) -> None:
        """Parity mode: the caller drew the logical indices (0 = oldest transition)."""
        idx = np.ascontiguousarray(logical_idx, dtype=np.int64)
        B = int(idx.shape[0])
        N.check(N.lib().pa_arena_gather(self.handle, idx.ctypes.data, B, C.byref(out),
                                        self._scratch(B).data_ptr(), self._stream()))

    def gather_device(self, logical_idx: torch.Tensor, out: N.BatchOut) -> None:
        assert logical_idx.dtype == torch.int64 and logical_idx.is_cuda
        N.check(N.lib().pa_arena_gather_device(self.handle, logical_idx.data_ptr(),
                                               int(logical_idx.numel()), C.byref(out),
                                               self._stream()))

    def sample(self, seed: int, offset: int, B: int, out: N.BatchOut) -> torch.Tensor:
        """Fast mode: Philox draw without replacement on the device; returns the indices."""
        idx = torch.empty(max(B, 1), dtype=torch.int64, device=self.device)
        N.check(N.lib().pa_arena_sample(self.handle, seed & (2**64 - 1), offset & (2**64 - 1), B,
                                        C.byref(out), idx.data_ptr(), self._stream()))
        return idx[:B]
