"""Debug aid: one SAC learn_batch from identical state through the sequenced and the fused form;
compares q(xq), y and losses read back from the scratch buffers (layouts: sac_step.hip)."""
import copy
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch  # noqa: E402

DEV = "cuda:0"


def a4(n):
    return (n + 3) & ~3


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "cfg3_fullbatch"
    fx = torch.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", f"sac_{name}.pt"),
                    weights_only=False)
    from test_gpu_actor_critic import make_sac, sac_batch
    cfg = fx["config"]
    B, S, A = cfg["B"], cfg["S"], cfg["A"]
    W = S + A
    out = {}
    for form in ("seq", "fused", "fused2"):
        os.environ["PEARL_AMD_SAC_ONE_CALL"] = "1"
        os.environ["PEARL_AMD_SAC_FUSED"] = "0" if form == "seq" else "1"
        pl = make_sac(fx)
        na, nc = fx["noises"][0]
        seq = iter([na, nc])
        pl.noise_source = lambda B_, A_, dev: next(seq)
        rep = pl.learn_batch(pl.preprocess_batch(sac_batch(fx)))
        torch.cuda.synchronize()
        sc = pl._flat["one_call"]["scratch"].cpu()
        if form == "seq":
            sizes = [B * W, B * 2 * A, B, B, B, B, B, B * W, B * W, B * 2 * A, B * W, B * 2 * A, B, B, B, B,
                     B * W, B, B, B, B]
            names = ["xa", "head", "logp", "q1", "q2", "dq1", "dq2", "dx1", "dx2", "d_head", "xn", "head_n",
                     "nlogp", "nq1", "nq2", "y", "xq", "qa", "qb", "dqa", "dqb"]
        else:
            sizes = [B * 2 * A, B, B * W, B, B, B, B]
            names = ["d_head", "logp", "xq", "qa", "qb", "dqa", "dqb"]
        o, d = 0, {}
        for n, sz in zip(names, sizes):
            d[n] = sc[o:o + sz].clone()
            o += a4(sz)
        d["rep"] = {k: float(v) for k, v in rep.items()}
        d["y"] = d.get("y", d["qa"] - B * d["dqa"])
        out[form] = d
        print(form, d["rep"], "fixture", fx["reports"][0])
    for k in ("qa", "qb", "dqa", "dqb", "y", "d_head", "xq"):
        for f in ("fused", "fused2"):
            x, y = out["seq"][k], out[f][k]
            err = (x - y).abs()
            print(f"{k:8s} seq vs {f}: max abs {float(err.max()):.3e} at {int(err.argmax())} "
                  f"(|x| max {float(x.abs().max()):.3e})")


if __name__ == "__main__":
    main()
