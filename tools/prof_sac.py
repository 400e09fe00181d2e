"""Where the time goes inside the fused SAC row kernels (sac_rows.hpp): medians over workgroups of
the in-kernel wall-clock stamps (100 MHz), one learn_batch of the cfg3 learner.
    python tools/prof_sac.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

sys.argv = [sys.argv[0]]
from host_bound import make_sac  # noqa: E402

NAMES_A = {0: "start", 1: "staged", 2: "actor fwd (head in LDS)", 3: "sampled", 4: "c1 L1", 5: "c1 L2",
           6: "c1 G", 7: "c1 gx", 8: "c2 L1", 9: "c2 L2", 10: "c2 G", 11: "c2 gx", 12: "head grad",
           13: "dz2 (d head W3)", 14: "dz1", 15: "end"}
NAMES_A_SPLIT = {0: "start", 1: "staged", 2: "actor fwd (head in LDS)", 3: "sampled + sent", 4: "c1 L1",
                 5: "c1 L2", 6: "c1 G", 7: "c1 gx", 11: "q2, gx2 received", 12: "head grad",
                 13: "dz2 (d head W3)", 14: "dz1", 15: "end"}
NAMES_H = {0: "start", 1: "staged", 3: "action received", 4: "c2 L1", 5: "c2 L2", 6: "c2 G", 15: "sent, end"}
NAMES_C = {0: "start", 1: "staged", 4: "L1", 5: "L2", 6: "G", 15: "end"}
NAMES_B_SPLIT = {0: "start", 1: "staged", 2: "actor fwd", 3: "sampled + sent", 4: "t1 L1", 5: "t1 L2",
                 11: "q2' received", 12: "y, dq", 13: "scaled", 15: "end"}
NAMES_BH = {0: "start", 1: "staged", 3: "action received", 8: "t2 L1", 9: "t2 L2", 15: "sent, end"}
NAMES_B = {0: "start", 1: "staged", 2: "actor fwd", 3: "sampled", 4: "t1 L1", 5: "t1 L2", 8: "t2 L1",
           9: "t2 L2", 12: "y, dq", 13: "scaled", 15: "end"}


def table(title, st, names):
    # st: [wgs][8][32]; relative to the earliest stamp of the launch
    t0 = st[:, :, 0].min()
    print(f"== {title}: {st.shape[0]} workgroups; us since the first wave started")
    print(f"{'phase':28s}{'min':>8s}{'median':>8s}{'max':>8s}")
    for i, nm in names.items():
        v = (st[:, :, i] - t0) / 100.0
        print(f"{nm:28s}{v.min():8.2f}{np.median(v):8.2f}{v.max():8.2f}")


def main():
    from pearl_amd import _native as N
    B = 1024
    tiles = B // 16
    learn = make_sac(8)
    learn()
    torch.cuda.synchronize()
    split = os.environ.get("PEARL_AMD_SAC_SPLIT", "1") != "0"
    pa = torch.zeros(4 * tiles * 8 * 32, dtype=torch.int64, device="cuda:0")
    pb = torch.zeros(2 * tiles * 8 * 32, dtype=torch.int64, device="cuda:0")
    N.check(N.lib().pa_debug_sac_prof(pa.data_ptr(), pb.data_ptr()))
    learn()
    torch.cuda.synchronize()
    N.check(N.lib().pa_debug_sac_prof(None, None))
    a = pa.cpu().numpy().reshape(4, tiles, 8, 32)[: 4 if split else 3]
    b2 = pb.cpu().numpy().reshape(2, tiles, 8, 32)
    b = b2[0]
    table("sac_rows_a, actor rows", a[0], NAMES_A_SPLIT if split else NAMES_A)
    table("sac_rows_a, critic rows (both)", a[1:3].reshape(2 * tiles, 8, 32), NAMES_C)
    if split:
        # the helper's clock origin is the launch's, like every table here
        table("sac_rows_a, helper rows (critic 2 at the fresh action)", a[3], NAMES_H)
    t0 = a[:, :, :, 0].min()
    print(f"whole launch: {(a[:, :, :, 15].max() - t0) / 100.0:.2f} us "
          f"(actor rows end {(a[0, :, :, 15].max() - t0) / 100.0:.2f}, critic rows end "
          f"{(a[1:, :, :, 15].max() - t0) / 100.0:.2f})")
    table("sac_rows_b", b, NAMES_B_SPLIT if split else NAMES_B)
    if split:
        table("sac_rows_b, helper rows (target critic 2)", b2[1], NAMES_BH)


if __name__ == "__main__":
    main()
