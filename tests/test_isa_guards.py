"""Compiler-output guards for the hand-written kernels (CPU-only: hipcc cross-compiles gfx950).

Round 6 found that the generic engine's row kernels had lost most of their overlap to two things
hipcc did silently (profiles/r06_j_rowstep_lds_flat_and_ring.txt):

* an LDS tile reached through a pointer picked at run time was read with FLAT loads, whose wait is
  `s_waitcnt vmcnt(0) lgkmcnt(0)` — the weight ring drained in front of every k-step;
* a `vmcnt(0)` in front of an MFMA group inside the GEMM loops (ring rotation at the back edge, a
  pending store on the way into the loop, ring slots issued out of order).

Neither changes a result, so no numerical test can see a regression; this one reads the ISA."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pearl_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

# kernels of the generic engine whose operands live in run-time-selected LDS tiles
ROW_KERNELS = ("mlp_rowfwd_kernel", "mlp_rowbwd_kernel", "mlp_rowstep_kernel")


@pytest.fixture(scope="module")
def mlp_isa(tmp_path_factory):
    if not (os.path.exists(HIPCC) or shutil.which("hipcc")):
        pytest.skip("hipcc not available")
    out = tmp_path_factory.mktemp("isa") / "mlp.s"
    cmd = [HIPCC if os.path.exists(HIPCC) else "hipcc", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
           "--offload-arch=gfx950", "-w", "-S", "--cuda-device-only", "-o", str(out),
           os.path.join(CSRC, "mlp.hip")]
    subprocess.run(cmd, check=True, cwd=CSRC, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    kernels, cur = {}, None
    for ln in open(out, errors="replace"):
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            cur = m.group(1)
            kernels[cur] = []
        elif ".amdhsa_kernel" in ln:
            cur = None
        elif cur is not None and not ln.lstrip().startswith(";"):
            kernels[cur].append(ln.strip())
    return kernels


def _row_kernels(isa):
    got = {k: v for k, v in isa.items() if any(n in k for n in ROW_KERNELS)}
    assert len(got) >= 5, sorted(got)      # rowfwd, rowbwd and three row-step instantiations
    return got


def test_row_kernels_reach_lds_with_ds_instructions(mlp_isa):
    for name, body in _row_kernels(mlp_isa).items():
        flat = [ln for ln in body if ln.startswith(("flat_load", "flat_store"))]
        assert not flat, f"{name}: {len(flat)} flat accesses (an LDS pointer lost its address space): {flat[:3]}"
        assert any(ln.startswith("ds_read_b128") for ln in body), name


def test_row_kernel_gemm_loops_wait_partially(mlp_isa):
    """In front of an MFMA group, a wait for global loads names how many may stay in flight."""
    for name, body in _row_kernels(mlp_isa).items():
        drains = 0
        groups = 0
        for i, ln in enumerate(body):
            if ln.startswith("v_mfma") and not body[i - 1].startswith("v_mfma"):
                groups += 1
                # the waits between the last non-wait instruction and this group
                j = i - 1
                while j >= 0 and body[j].startswith(("s_waitcnt", "s_nop")):
                    if re.search(r"vmcnt\(0\)", body[j]):
                        drains += 1
                    j -= 1
        assert groups >= 8, (name, groups)
        assert drains == 0, f"{name}: {drains} of {groups} MFMA groups sit behind s_waitcnt vmcnt(0)"
