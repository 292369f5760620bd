#!/usr/bin/env python3
"""A/B of the rank pre-pass of the rank-quantised path on one GPU: grouped (one launch, no transposed intermediate) vs
transpose + rank kernels, per tree count.  Prints pre-pass / scoring / end-to-end ms (HIP events of the library,
"kernel_timing") and checks a prefix of the scores against the oracle.  Usage: tools/prepass_ab.py [--rows N] [--trees 1000,500]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "distributed-decisiontrees_amd"))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import ddt  # noqa: E402
from oracle import oracle as O  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=100_000_000)
    ap.add_argument("--trees", default="1000,500,250,125")
    ap.add_argument("--features", type=int, default=32)
    ap.add_argument("--reps", type=int, default=4)
    ap.add_argument("--opts", default="q16_prepass_groups=0;q16_prepass_groups=1;q16_prepass_groups=2;q16_prepass_groups=4;q16_prepass_groups=8;q16_fused_prepass=0,q16_grouped_prepass=0")
    a = ap.parse_args()
    D, F, N = 8, a.features, a.rows
    eng = ddt.Engine(0)
    eng.set_option("kernel_timing", 1)
    d = eng.synth_tuples_device(0, N, F)
    out = torch.empty(N, dtype=torch.float32, device="cuda")
    for T in [int(t) for t in a.trees.split(",")]:
        w, f = ddt.synth_model(T, D, F)
        m = O.Model(O.make_params(T, D, F), w, f)
        xs = d[:4096].cpu().numpy().view(np.uint32)
        want = O.score(m, xs)
        for opts in a.opts.split(";"):
            for kv in ["q16_fused_prepass=1", "q16_grouped_prepass=1", "q16_prepass_groups=0"] + list(filter(None, opts.split(","))):
                k, v = kv.split("=")
                eng.set_option(k, int(v))
            eng.load_model(ddt.make_params(T, D, F), w, f)
            eng.score_device(d, out=out)
            torch.cuda.synchronize()
            ok = bool(np.array_equal(out[:4096].cpu().numpy().view(np.uint32), want.view(np.uint32)))
            tail_ok = bool(torch.isfinite(out[-4096:]).all())
            best = None
            for _ in range(a.reps):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                eng.score_device(d, out=out)
                e1.record()
                torch.cuda.synchronize()
                st = eng.stats()
                t = (e0.elapsed_time(e1), st.last_prepass_ms, st.last_score_ms)
                best = t if best is None or t[0] < best[0] else best
            print(f"T={T:5d} {opts:28s} {eng.info().variant_name.decode():14s} G={eng.info().prepass_groups} ok={ok and tail_ok}  total {best[0]:8.3f} ms  prepass {best[1]:7.3f} ms  "
                  f"score {best[2]:8.3f} ms  {N / best[0] / 1e3:8.1f} Mtuples/s", flush=True)


if __name__ == "__main__":
    main()
