"""``BasicReplayBuffer`` on the HBM arena.

Drop-in for pearl/replay_buffers/basic_replay_buffer.py:17-48 +
tensor_based_replay_buffer.py:25-400: same constructor, ``push`` signature, ``sample``
contract (TransitionBatch shapes/dtypes, ``ValueError`` when ``batch_size > len``), FIFO
eviction, ``clear``, ``len``.  Storage and the collate step are HIP kernels:

* ``push`` tensorises exactly like the reference (:143-177, :179-251) but packs the row into
  a pinned staging ring; rows reach the HBM ring in one H2D copy + one unpack kernel;
* ``sample`` draws indices and runs ONE gather kernel instead of ~10 ``torch.cat`` over B
  one-row tensors (:290-400).

Index streams.  ``sampler="python"`` draws ``random.sample(range(len(self)), B)``: the same
positions, and the same consumption of Python's global ``random`` state, as the reference's
``random.sample(self.memory, B)`` (:276) — batches are bit-identical to the reference's.
``sampler="device"`` (default) draws without replacement on the GPU with Philox keyed by 64
bits taken from Python's ``random`` per call: reproducible under ``random.seed`` but a
different (equally uniform) stream.
"""
from __future__ import annotations

import random
from typing import Any, Dict, Optional, Tuple

import numpy as np
import torch
from torch import Tensor

from .. import _native as N
from .arena import ArenaLayout, HbmArena
from .replay_buffer import ReplayBuffer
from .transition import TransitionBatch


def _default_batch_device() -> torch.device:
    if torch.cuda.is_available():
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def _as_host_array(x: Any, dtype: Optional[np.dtype] = None) -> np.ndarray:
    """CPU, contiguous numpy view of a tensor / array / python number."""
    if isinstance(x, Tensor):
        x = x.detach()
        if x.device.type != "cpu":
            x = x.cpu()  # the reference also lands pushes on the CPU (:153-157)
        arr = x.contiguous().numpy()
    else:
        arr = np.asarray(x)
    if dtype is not None and arr.dtype != dtype:
        arr = arr.astype(dtype)
    return np.ascontiguousarray(arr)


def _torch_dtype_of_value(x: Any) -> torch.dtype:
    """dtype ``torch.tensor(x)`` would pick (reference :159-166)."""
    if isinstance(x, Tensor):
        return x.dtype
    if isinstance(x, np.generic):  # numpy scalars keep their own width
        return torch.as_tensor(x).dtype
    if isinstance(x, bool):
        return torch.bool
    if isinstance(x, int):
        return torch.int64
    if isinstance(x, float):
        return torch.float32
    return torch.as_tensor(x).dtype


_NP_OF_TORCH = {torch.float32: np.float32, torch.float64: np.float64, torch.int64: np.int64,
                torch.int32: np.int32, torch.uint8: np.uint8, torch.bool: np.uint8}


def create_action_tensor_and_mask(max_number_actions: Optional[int], available_action_space: Any
                                  ) -> Tuple[Optional[Tensor], Optional[Tensor]]:
    """Zero-padded (max_number_actions, action_dim) float32 table of the available actions and
    the (max_number_actions,) bool mask that is True beyond ``space.n``
    (tensor_based_replay_buffer.py:179-251)."""
    if max_number_actions is None or available_action_space is None:
        return (None, None)
    n = int(available_action_space.n)
    table = torch.zeros((max_number_actions, int(available_action_space.action_dim)),
                        dtype=torch.float32)
    table[:n, :] = available_action_space.actions_batch
    mask = torch.zeros((max_number_actions,), dtype=torch.bool)
    mask[n:] = True
    return table, mask


class SideRing:
    """One extra per-transition column kept NEXT TO the arena in HBM (``next_action`` of the SARSA
    buffer, the Bernoulli masks of the bootstrap buffer): a ``[capacity, width]`` device ring with
    the arena's FIFO arithmetic — logical index i (0 = oldest stored row) lives in slot
    ``(head + i) % capacity`` — mirrored on the host, gathered with ``pa_gather_rows``."""

    def __init__(self, capacity: int) -> None:
        self.capacity = int(capacity)
        self.data: Optional[Tensor] = None
        self.head = 0
        self.size = 0

    def clear(self) -> None:
        self.head = self.size = 0

    def _ensure(self, width: int, dtype: torch.dtype, device: torch.device) -> Tensor:
        if self.data is None:
            self.data = torch.zeros(self.capacity, width, dtype=dtype, device=device)
        assert self.data.shape[1] == width, "side column changed its width"
        return self.data

    def append(self, row: Tensor, device: torch.device) -> None:
        """Store one row behind the newest one; a full ring drops its oldest row (deque(maxlen))."""
        flat = row.detach().reshape(-1)
        data = self._ensure(int(flat.numel()), flat.dtype, device)
        if self.size < self.capacity:
            slot = (self.head + self.size) % self.capacity
            self.size += 1
        else:
            slot = self.head
            self.head = (self.head + 1) % self.capacity
        data[slot].copy_(flat.to(data.dtype))

    def gather(self, logical_idx: Tensor) -> Tensor:
        assert self.data is not None, "side column is empty"
        B = int(logical_idx.numel())
        phys = ((logical_idx + self.head) % self.capacity).contiguous()
        width = int(self.data.shape[1])
        out = torch.empty(B, width, dtype=self.data.dtype, device=self.data.device)
        if B:
            N.check(N.lib().pa_gather_rows(self.data.data_ptr(), width * self.data.element_size(),
                                           phys.data_ptr(), B, out.data_ptr(),
                                           N.stream_ptr(self.data.device)))
        return out

    def logical(self) -> Optional[Tensor]:
        """Every stored row, oldest first, on the CPU (checkpoints)."""
        if self.data is None:
            return None
        idx = (torch.arange(self.size, device=self.data.device) + self.head) % self.capacity
        return self.data[idx].cpu()

    def load(self, rows: Optional[Tensor], device: torch.device) -> None:
        """Replace the contents with `rows` (oldest first); a smaller ring keeps the newest."""
        self.clear()
        if rows is None or rows.shape[0] == 0:
            return
        rows = rows[-self.capacity:]
        data = self._ensure(int(rows.shape[1]), rows.dtype, device)
        data[: rows.shape[0]].copy_(rows.to(device))
        self.size = int(rows.shape[0])


class TensorBasedReplayBuffer(ReplayBuffer):
    """Arena-backed counterpart of tensor_based_replay_buffer.py:25-400."""

    def __init__(self, capacity: int, sampler: str = "device", staging_rows: int = 0) -> None:
        super().__init__()
        if sampler not in ("device", "python"):
            raise ValueError(f"sampler must be 'device' or 'python', got {sampler!r}")
        self.capacity = int(capacity)
        self.sampler = sampler
        self._staging_rows = int(staging_rows)
        self._device_for_batches: torch.device = _default_batch_device()
        self._arena: Optional[HbmArena] = None
        self._layout: Optional[ArenaLayout] = None
        self._has_curr_avail = False
        self._has_next_avail = False
        self._space_cache: dict = {}
        self._last_idx: Optional[Tensor] = None
        self._presampled: Optional[Tuple[Tensor, int, int]] = None   # (lists [R, B], next row, len)
        self._pregather_bytes = 0
        self._pregathered: Optional[dict] = None    # batches of G presampled rounds, gathered at once
        self._pg_slot: Optional[int] = None         # slot of the last sample() inside _pregathered

    # -- ReplayBuffer interface -------------------------------------------------
    @property
    def device_for_batches(self) -> torch.device:
        return self._device_for_batches

    @device_for_batches.setter
    def device_for_batches(self, new_device_for_batches: torch.device) -> None:
        new = torch.device(new_device_for_batches)
        if self._arena is not None and new.type == "cuda":
            idx = new.index if new.index is not None else torch.cuda.current_device()
            if idx != self._arena.device.index:
                raise N.NativeError("pearl_amd: cannot move a populated HBM arena across devices")
        self._device_for_batches = new

    def __len__(self) -> int:
        return 0 if self._arena is None else len(self._arena)

    def clear(self) -> None:
        self._presampled = None
        self._pregathered = None
        if self._arena is not None:
            self._arena.clear()

    @property
    def arena(self) -> Optional[HbmArena]:
        return self._arena

    @property
    def shared_action_table(self) -> bool:
        """Every stored row carries the same padded next-action table and mask (a static action
        space): DeepQLearning.learn then feeds the target pass ONE table instead of gathering
        (B, A, rep) rows per window (pa_arena_shared_next_table)."""
        return self._arena is not None and bool(
            N.lib().pa_arena_shared_next_table(self._arena.handle))

    # -- push ------------------------------------------------------------------
    def _padded_tables(self, max_number_actions: int, space: Any) -> Tuple[np.ndarray, np.ndarray]:
        if space is None:
            z = self._layout
            A = max_number_actions
            d = z.avail_dim if z is not None else 1
            return np.zeros((A, d), np.float32), np.zeros((A,), np.uint8)
        # Keyed on the space OBJECT, validated by identity: the entry holds a strong reference to
        # the space it was built from, so the id cannot be recycled for another space while the
        # entry lives (dynamic action spaces are rebuilt every step and CPython reuses addresses
        # at once).  A hit additionally requires `hit_space is space`.
        key = (id(space), int(space.n), max_number_actions)
        hit = self._space_cache.get(key)
        if hit is None or hit[2] is not space:
            table, mask = create_action_tensor_and_mask(max_number_actions, space)
            hit = (_as_host_array(table, np.float32), _as_host_array(mask.to(torch.uint8)), space)
            if len(self._space_cache) > 64:
                self._space_cache.clear()
            self._space_cache[key] = hit
        return hit[0], hit[1]

    def _ensure_arena(self, layout: ArenaLayout) -> None:
        if self._arena is None:
            self._layout = layout
            self._arena = HbmArena(self.capacity, layout, self._device_for_batches,
                                   self._staging_rows)
            return
        assert layout == self._layout, (
            f"transition layout changed after the first push: {layout} vs {self._layout}")

    def _coerce_dtypes(self, act_dtype: torch.dtype, rew_dtype: torch.dtype):
        """The arena's columns have ONE dtype each, fixed by the first push; the reference stores
        per-transition tensors and lets ``torch.cat`` promote (:290-400).  Later pushes are cast to
        the stored dtype where that is what the promotion would give anyway (python ints /
        numpy scalars into a float32 reward column, int32 / int64 actions into each other); a float
        into an integer column would silently truncate and is refused."""
        z = self._layout
        if z is None:
            return act_dtype, rew_dtype
        for name, new, stored in (("reward", rew_dtype, z.reward_dtype), ("action", act_dtype, z.action_dtype)):
            if new != stored and new.is_floating_point and not stored.is_floating_point:
                raise TypeError(
                    f"pearl_amd replay arena: {name} column was created as {stored} by the first "
                    f"push and cannot hold a {new} value; push floats from the start")
        return z.action_dtype, z.reward_dtype

    def push(self, state: Any, action: Any, reward: Any, terminated: bool, truncated: bool,
             curr_available_actions: Any = None, next_state: Any = None,
             next_available_actions: Any = None, max_number_actions: Optional[int] = None,
             cost: Optional[float] = None) -> None:
        # a push changes what a logical index means (FIFO eviction shifts every row of a full
        # buffer while len() stays put): index lists / batches drawn before it are stale
        if self._presampled is not None or self._pregathered is not None:
            self.drop_presampled()
        A = 0
        curr_tab = curr_mask = next_tab = next_mask = None
        avail_dim = 0
        if not self._is_action_continuous:
            # static action space unless the caller says otherwise (:79-85)
            if max_number_actions is None:
                assert curr_available_actions is not None and hasattr(curr_available_actions, "n"), \
                    "discrete replay buffer needs curr_available_actions or max_number_actions"
                max_number_actions = int(curr_available_actions.n)
            A = int(max_number_actions)
            has_curr = curr_available_actions is not None
            has_next = next_available_actions is not None
            if self._arena is None:
                self._has_curr_avail, self._has_next_avail = has_curr, has_next
            else:
                assert (has_curr, has_next) == (self._has_curr_avail, self._has_next_avail), \
                    "availability of curr/next action spaces changed after the first push"
            some = curr_available_actions if has_curr else next_available_actions
            avail_dim = int(some.action_dim) if some is not None else 1
            if not (has_curr or has_next):
                A = 0
            else:
                curr_tab, curr_mask = self._padded_tables(A, curr_available_actions)
                next_tab, next_mask = self._padded_tables(A, next_available_actions)

        st = _as_host_array(state)
        st32 = st.astype(np.float32, copy=False) if st.dtype != np.float32 else st
        if isinstance(action, Tensor):
            act_dtype = action.dtype
        else:
            act_dtype = torch.as_tensor(action).dtype
        rew_dtype = _torch_dtype_of_value(reward)
        if rew_dtype == torch.bool:
            rew_dtype = torch.int64
        act_dtype, rew_dtype = self._coerce_dtypes(act_dtype, rew_dtype)
        act = _as_host_array(action, _NP_OF_TORCH[act_dtype])
        rew = _as_host_array(reward, _NP_OF_TORCH[rew_dtype]).reshape(-1)[:1]
        layout = ArenaLayout(
            state_shape=tuple(st.shape), action_shape=tuple(act.shape), action_dtype=act_dtype,
            reward_dtype=rew_dtype, max_actions=A, avail_dim=avail_dim if A else 0,
            has_next_state=next_state is not None, has_cost=cost is not None)
        self._ensure_arena(layout)

        t = N.Transition()
        keep = [st32, act, rew]  # keep the numpy buffers alive across the C call
        t.state = st32.ctypes.data
        t.action = act.ctypes.data
        t.reward = rew.ctypes.data
        if next_state is not None:
            ns = _as_host_array(next_state)
            ns = ns.astype(np.float32, copy=False) if ns.dtype != np.float32 else ns
            assert ns.size == st32.size, "next_state and state differ in size"
            keep.append(ns)
            t.next_state = ns.ctypes.data
        if A:
            t.curr_avail, t.curr_mask = curr_tab.ctypes.data, curr_mask.ctypes.data
            t.next_avail, t.next_mask = next_tab.ctypes.data, next_mask.ctypes.data
        if cost is not None:
            c = np.asarray([cost], np.float32)
            keep.append(c)
            t.cost = c.ctypes.data
        t.terminated = 1 if bool(terminated) else 0
        t.truncated = 1 if bool(truncated) else 0
        self._arena.push_row(t)

    def push_many(self, state: Tensor, action: Tensor, reward: Tensor, terminated: Tensor,
                  truncated: Tensor, next_state: Optional[Tensor] = None,
                  curr_available_actions: Any = None, next_available_actions: Any = None,
                  max_number_actions: Optional[int] = None, cost: Optional[Tensor] = None) -> None:
        """Batched ingest of n transitions (leading dim n) — SURVEY.md §8f rank 1.

        Same per-row semantics as n successive ``push`` calls with one (static) action space.
        Tensors may live on the arena's device (no host round trip) or on the CPU.
        """
        if self._presampled is not None or self._pregathered is not None:
            self.drop_presampled()       # (see push)
        n = int(state.shape[0])
        A = 0
        avail_dim = 0
        tabs = None
        if not self._is_action_continuous:
            if max_number_actions is None:
                assert curr_available_actions is not None
                max_number_actions = int(curr_available_actions.n)
            A = int(max_number_actions)
            has_curr = curr_available_actions is not None
            has_next = next_available_actions is not None
            if self._arena is None:
                self._has_curr_avail, self._has_next_avail = has_curr, has_next
            some = curr_available_actions if has_curr else next_available_actions
            avail_dim = int(some.action_dim) if some is not None else 1
            if not (has_curr or has_next):
                A = 0
            else:
                tabs = (*self._padded_tables(A, curr_available_actions),
                        *self._padded_tables(A, next_available_actions))
        rew_dtype = reward.dtype
        layout = ArenaLayout(
            state_shape=tuple(state.shape[1:]), action_shape=tuple(action.shape[1:]),
            action_dtype=action.dtype, reward_dtype=rew_dtype, max_actions=A,
            avail_dim=avail_dim if A else 0, has_next_state=next_state is not None,
            has_cost=cost is not None)
        self._ensure_arena(layout)
        dev = self._arena.device
        on_device = state.is_cuda
        target = dev if on_device else torch.device("cpu")

        def col(x: Optional[Tensor], dtype: Optional[torch.dtype] = None) -> Optional[Tensor]:
            if x is None:
                return None
            x = x.detach().to(device=target, dtype=dtype if dtype is not None else x.dtype)
            return x.contiguous()

        keep = [col(state, torch.float32), col(action), col(reward), col(terminated, torch.uint8),
                col(truncated, torch.uint8), col(next_state, torch.float32),
                col(cost, torch.float32)]
        cols = N.Columns()
        cols.state, cols.action, cols.reward = (N.ptr(keep[0]), N.ptr(keep[1]), N.ptr(keep[2]))
        cols.terminated, cols.truncated = N.ptr(keep[3]), N.ptr(keep[4])
        cols.next_state, cols.cost = N.ptr(keep[5]), N.ptr(keep[6])
        if A:
            tt = [torch.from_numpy(a).to(target) for a in tabs]
            keep.extend(tt)
            cols.curr_avail, cols.curr_mask = tt[0].data_ptr(), tt[1].data_ptr()
            cols.next_avail, cols.next_mask = tt[2].data_ptr(), tt[3].data_ptr()
            cols.avail_bcast = 1
        self._arena.push_columns(n, cols, on_device)
        if on_device:
            # the source columns must outlive the enqueued scatter kernel
            torch.cuda.current_stream(dev).synchronize()

    # -- checkpoint / resume (the reference does not checkpoint its buffer) --
    _CKPT_CHUNK = 262_144   # rows per gather / scatter launch (bounds the device temporaries)
    DEVICE_SAMPLER_MAX_B = 8192   # SAMPLE_MAX_B of arena.hip

    def state_dict(self) -> Dict[str, Any]:
        """Everything stored, oldest transition first, as CPU tensors (``torch.save``-able):
        the columns ``sample()`` returns — state, action, reward, terminated, truncated, next_state,
        per-row available-action tables and masks, cost — plus the layout.  The FIFO position is
        implicit: ``load_state_dict`` re-inserts the rows in the same logical order."""
        n = len(self)
        sd: Dict[str, Any] = {"capacity": self.capacity, "size": n,
                              "is_action_continuous": bool(self._is_action_continuous),
                              "has_curr_avail": self._has_curr_avail,
                              "has_next_avail": self._has_next_avail, "layout": None,
                              "columns": {}}
        if self._arena is None or self._layout is None:
            return sd
        z = self._layout
        sd["layout"] = dict(state_shape=tuple(z.state_shape), action_shape=tuple(z.action_shape),
                            action_dtype=z.action_dtype, reward_dtype=z.reward_dtype,
                            max_actions=z.max_actions, avail_dim=z.avail_dim,
                            has_next_state=z.has_next_state, has_cost=z.has_cost)
        names = ("state", "action", "reward", "terminated", "truncated", "next_state",
                 "curr_available_actions", "curr_unavailable_actions_mask",
                 "next_available_actions", "next_unavailable_actions_mask", "cost")
        parts: Dict[str, list] = {k: [] for k in names}
        keep_idx, keep_dev = self._last_idx, self._device_for_batches
        self._device_for_batches = self._arena.device
        try:
            for lo in range(0, n, self._CKPT_CHUNK):
                hi = min(n, lo + self._CKPT_CHUNK)
                b = self._gather_batch(torch.arange(lo, hi, device=self._arena.device))
                for k in names:
                    v = getattr(b, k)
                    if v is not None:
                        parts[k].append(v.cpu())
        finally:
            self._last_idx, self._device_for_batches = keep_idx, keep_dev
        sd["columns"] = {k: torch.cat(v) for k, v in parts.items() if v}
        return sd

    def load_state_dict(self, sd: Dict[str, Any]) -> None:
        """Replace the contents with a ``state_dict()``.  A smaller capacity keeps the newest
        rows, exactly as pushing them one by one would."""
        self.clear()
        self._is_action_continuous = bool(sd["is_action_continuous"])
        if sd["layout"] is None or sd["size"] == 0:
            return
        layout = ArenaLayout(**sd["layout"])
        if self._arena is None:
            self._has_curr_avail, self._has_next_avail = sd["has_curr_avail"], sd["has_next_avail"]
        else:
            assert (self._has_curr_avail, self._has_next_avail) == \
                (sd["has_curr_avail"], sd["has_next_avail"]), "available-action columns differ"
        self._ensure_arena(layout)
        c, n = sd["columns"], int(sd["size"])
        A = layout.max_actions
        for lo in range(0, n, self._CKPT_CHUNK):
            hi = min(n, lo + self._CKPT_CHUNK)

            def col(name: str, dtype: Optional[torch.dtype] = None) -> Optional[Tensor]:
                v = c.get(name)
                if v is None:
                    return None
                v = v[lo:hi]
                return v.to(dtype if dtype is not None else v.dtype).contiguous()

            keep = [col("state", torch.float32), col("action"), col("reward"),
                    col("terminated", torch.uint8), col("truncated", torch.uint8),
                    col("next_state", torch.float32), col("cost", torch.float32)]
            cols = N.Columns()
            cols.state, cols.action, cols.reward = N.ptr(keep[0]), N.ptr(keep[1]), N.ptr(keep[2])
            cols.terminated, cols.truncated = N.ptr(keep[3]), N.ptr(keep[4])
            cols.next_state, cols.cost = N.ptr(keep[5]), N.ptr(keep[6])
            if A:
                zeros_a = torch.zeros(hi - lo, A, layout.avail_dim, dtype=torch.float32)
                zeros_m = torch.zeros(hi - lo, A, dtype=torch.uint8)
                tabs = [col("curr_available_actions", torch.float32),
                        col("curr_unavailable_actions_mask", torch.uint8),
                        col("next_available_actions", torch.float32),
                        col("next_unavailable_actions_mask", torch.uint8)]
                tabs = [t if t is not None else (zeros_a if i % 2 == 0 else zeros_m)
                        for i, t in enumerate(tabs)]
                keep.extend(tabs)
                cols.curr_avail, cols.curr_mask = tabs[0].data_ptr(), tabs[1].data_ptr()
                cols.next_avail, cols.next_mask = tabs[2].data_ptr(), tabs[3].data_ptr()
                cols.avail_bcast = 0
            self._arena.push_columns(hi - lo, cols, False)
        self._arena.flush()

    # -- sample ----------------------------------------------------------------
    def _draw_host_indices(self, batch_size: int) -> np.ndarray:
        return np.fromiter(random.sample(range(len(self)), batch_size), dtype=np.int64,
                           count=batch_size)

    def presample(self, rounds: int, batch_size: int, pregather_bytes: int = 0) -> bool:
        """Draw the index lists of the next ``rounds`` ``sample(batch_size)`` calls at once — device
        sampler: ONE launch (``sample_indices_kernel`` runs one workgroup per list, so a learner
        loop does not pay a single-workgroup kernel per round); python sampler: ``rounds`` draws of
        ``random.sample`` uploaded as one tensor — the following ``sample`` calls of that size
        consume them in order.  Pushing, clearing or a different batch size drops what
        is left.  Consumes Python's ``random`` once (the Philox key), like one ``sample``.

        ``pregather_bytes`` > 0: those ``sample`` calls also share their gather — one launch fills
        the batches of as many consecutive rounds as fit in that many bytes, and each call returns
        its rows of it (views; the same values a gather of its own would produce)."""
        self._presampled = None
        self._pregathered = None
        self._pregather_bytes = 0
        n = len(self)
        if self._arena is None or rounds <= 0 or not 0 < batch_size <= n:
            return False
        dev = self._arena.device
        if self.sampler == "python":
            # parity mode: the lists `rounds` consecutive sample() calls would draw, drawn now — the
            # same consumption of Python's MT19937 stream as the reference's learn loop, whose
            # rounds touch `random` nowhere else (tensor_based_replay_buffer.py:276)
            self._arena.flush()
            host = np.stack([self._draw_host_indices(int(batch_size)) for _ in range(int(rounds))])
            lists = torch.from_numpy(host).to(dev)
        elif batch_size > self.DEVICE_SAMPLER_MAX_B:
            return False
        else:
            self._arena.flush()
            lists = torch.empty(int(rounds), int(batch_size), dtype=torch.int64, device=dev)
            N.check(N.lib().pa_sample_indices_rounds(n, random.getrandbits(64), 0, int(batch_size),
                                                     int(rounds), lists.data_ptr(), dev.index,
                                                     N.stream_ptr(dev)))
        self._presampled = (lists, 0, n)
        if pregather_bytes > 0 and self._device_for_batches == dev:
            self._pregather_bytes = int(pregather_bytes)
        return True

    def drop_presampled(self) -> None:
        self._presampled = None
        self._pregathered = None
        self._pregather_bytes = 0

    def _row_bytes(self) -> int:
        """Bytes of one gathered transition (upper bound, for sizing a shared gather)."""
        z = self._layout
        n = 4 * z.state_dim * (2 if z.has_next_state else 1) + 8 * z.action_elems + 8 + 2 + 4
        if z.max_actions and not self._is_action_continuous:
            n += (int(self._has_curr_avail) + int(self._has_next_avail)) * z.max_actions * (4 * z.avail_dim + 1)
        return n

    def _pregathered_batch(self, lists: Tensor, row: int) -> TransitionBatch:
        B = int(lists.shape[1])
        pg = self._pregathered
        if pg is None or not pg["row0"] <= row < pg["row0"] + pg["G"]:
            G = max(1, min(int(lists.shape[0]) - row, self._pregather_bytes // (self._row_bytes() * B),
                           len(self) // B))
            idx_flat = lists[row:row + G].reshape(-1)
            pg = {"row0": row, "G": G, "idx_flat": idx_flat, "big": self._gather_batch(idx_flat),
                  "extras": {}}
            self._pregathered = pg
        k = row - pg["row0"]
        self._pg_slot = k
        self._last_idx = lists[row]
        big = pg["big"]
        fields = {}
        for name in big._fields:
            v = getattr(big, name)
            fields[name] = v[k * B:(k + 1) * B] if isinstance(v, Tensor) and v.dim() > 0 \
                and v.shape[0] == pg["G"] * B else v
        return type(big)(**fields)

    def sample(self, batch_size: int) -> TransitionBatch:
        """Uniform sample without replacement -> ``TransitionBatch`` on ``device_for_batches``
        (tensor_based_replay_buffer.py:253-282, :290-400)."""
        if batch_size > len(self):
            raise ValueError(
                f"Can't get a batch of size {batch_size} from a replay buffer with "
                f"only {len(self)} elements")
        pre = self._presampled
        self._pg_slot = None
        if pre is not None:
            lists, row, n = pre
            if int(batch_size) == lists.shape[1] and n == len(self) and row < lists.shape[0]:
                self._presampled = (lists, row + 1, n)
                if self._pregather_bytes > 0:
                    return self._pregathered_batch(lists, row)
                return self._gather_batch(lists[row])
            self.drop_presampled()
        return self._gather_batch(None, int(batch_size))

    @property
    def last_indices(self) -> Tensor:
        """Logical indices (device int64) of the most recent ``sample`` / gather."""
        assert self._last_idx is not None
        return self._last_idx

    def _gather_batch(self, idx_dev: Optional[Tensor], batch_size: Optional[int] = None
                      ) -> TransitionBatch:
        """One gather launch.  idx_dev: caller-chosen logical indices on the device, or None to
        draw ``batch_size`` of them with the configured sampler."""
        z, arena = self._layout, self._arena
        assert z is not None and arena is not None
        dev = arena.device
        B = int(idx_dev.numel()) if idx_dev is not None else int(batch_size)
        A = z.max_actions

        def new(shape, dtype):
            return torch.empty(shape, dtype=dtype, device=dev)

        state = new((B, z.state_dim), torch.float32)
        action = new((B,) + z.action_shape, z.action_dtype)
        reward = new((B,), z.reward_dtype)
        term, trunc = new((B,), torch.bool), new((B,), torch.bool)
        next_state = new((B, z.state_dim), torch.float32) if z.has_next_state else None
        cost = new((B,), torch.float32) if z.has_cost else None
        ca = cm = na = nm = None
        if A and not self._is_action_continuous:
            if self._has_curr_avail:
                ca, cm = new((B, A, z.avail_dim), torch.float32), new((B, A), torch.bool)
            if self._has_next_avail:
                na, nm = new((B, A, z.avail_dim), torch.float32), new((B, A), torch.bool)
        out = N.BatchOut()
        out.state, out.action, out.reward = state.data_ptr(), action.data_ptr(), reward.data_ptr()
        out.terminated, out.truncated = term.data_ptr(), trunc.data_ptr()
        out.next_state, out.cost = N.ptr(next_state), N.ptr(cost)
        out.curr_avail, out.curr_mask = N.ptr(ca), N.ptr(cm)
        out.next_avail, out.next_mask = N.ptr(na), N.ptr(nm)
        if B > 0:
            if idx_dev is not None:
                arena.gather_device(idx_dev, out)
                self._last_idx = idx_dev
            elif self.sampler == "python" or B > self.DEVICE_SAMPLER_MAX_B:
                # (the device sampler draws one list per workgroup and holds at most 8192 indices;
                # whole-buffer batches — batch_size=-1, large PPO rollouts — take Python's sampler)
                arena.gather(self._draw_host_indices(B), out)
                self._last_idx = arena._scratch(B)[:B].clone()
            else:
                self._last_idx = arena.sample(random.getrandbits(64), 0, B, out)
        shape_s = (B,) + z.state_shape if len(z.state_shape) else (B, 1)
        batch = TransitionBatch(
            state=state.view(shape_s), action=action, reward=reward, terminated=term,
            truncated=trunc,
            next_state=None if next_state is None else next_state.view(shape_s),
            curr_available_actions=ca, curr_unavailable_actions_mask=cm,
            next_available_actions=na, next_unavailable_actions_mask=nm, cost=cost)
        if self._device_for_batches != dev and self._device_for_batches.type != "cuda":
            batch = batch.to(self._device_for_batches)
        return batch


class BasicReplayBuffer(TensorBasedReplayBuffer):
    """Plain FIFO replay buffer (pearl/replay_buffers/basic_replay_buffer.py:17-48)."""

    def __init__(self, capacity: int, sampler: str = "device", staging_rows: int = 0) -> None:
        super().__init__(capacity, sampler=sampler, staging_rows=staging_rows)
