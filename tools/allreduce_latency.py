"""Latency of the learn loop's one collective — the flat-gradient all-reduce (103 185 floats = 413 KB
for the benchmark network) — through the native RCCL hooks (pa_comm_allreduce_start / _wait), timed
with HIP events on the learner stream.  One process per GPU; with a single rank this measures the
fixed cost RCCL adds to every round (launch + its own kernel), which is what a 1-GPU box can show.

  python tools/allreduce_latency.py                      # 1 rank
  torchrun --nproc-per-node N tools/allreduce_latency.py   # N ranks on one node
"""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    from pearl_amd import _native as N
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    lib = N.lib()
    assert lib.pa_comm_available(), "RCCL not loadable"
    ident = torch.zeros(128, dtype=torch.uint8, device=dev)
    if rank == 0:
        buf = (C.c_char * 128)()
        N.check(lib.pa_comm_unique_id(buf))
        ident = torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8).clone().to(dev)
    dist.broadcast(ident, src=0)
    comm = C.c_void_p()
    torch.cuda.synchronize()
    N.check(lib.pa_comm_create(C.byref(comm), local, world, rank, bytes(ident.cpu().numpy().tobytes())))
    seen, me = C.c_int32(-1), C.c_int32(-1)
    N.check(lib.pa_comm_info(comm, C.byref(seen), C.byref(me)))
    out = {"ranks_requested": world, "ranks_observed": seen.value, "sizes": []}
    stream = N.stream_ptr(dev)
    for n in (1024, 103_185, 1_048_576):
        g = torch.ones(n, device=dev)
        for _ in range(20):
            N.check(lib.pa_comm_allreduce_start(comm, g.data_ptr(), n, stream))
            N.check(lib.pa_comm_allreduce_wait(comm, stream))
            g.fill_(1.0)
        torch.cuda.synchronize()
        reps = 200
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            N.check(lib.pa_comm_allreduce_start(comm, g.data_ptr(), n, stream))
            N.check(lib.pa_comm_allreduce_wait(comm, stream))
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        out["sizes"].append({"floats": n, "bytes": 4 * n, "us_per_allreduce": round(us, 2),
                             "algbw_GBps": round(4 * n / us / 1e3, 2)})
    if rank == 0:
        print(json.dumps(out))
    lib.pa_comm_destroy(comm)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
