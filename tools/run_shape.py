#!/usr/bin/env python3
"""Score one model shape a few times on one GPU (a target for rocprofv3): perfect-tree models by T x D x F, sparse
forests with --sparse.  Prints kernel name, ms per launch (HIP events), Mtuples/s, algorithmic GB/s."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "distributed-decisiontrees_amd"))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import ddt  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trees", type=int, default=8)
    ap.add_argument("--levels", type=int, default=4)
    ap.add_argument("--features", type=int, default=16)
    ap.add_argument("--rows", type=int, default=100_000_000)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--sparse", action="store_true")
    ap.add_argument("--full-levels", type=int, default=10)
    ap.add_argument("--permille", type=int, default=700)
    ap.add_argument("--bins", type=int, default=0, help="sparse: snap every threshold to one of N values per feature (histogram-trained models)")
    ap.add_argument("--wide-features", type=int, default=0, help="perfect trees: the model tests --features features out of this many (feature compaction A/B)")
    ap.add_argument("--sum-mode", type=int, default=0)
    ap.add_argument("--variant", default="", help="kernel variant name")
    ap.add_argument("--opt", default="", help="engine options, comma separated key=value")
    a = ap.parse_args()
    T, D, F, N = a.trees, a.levels, a.features, a.rows
    eng = ddt.Engine(0)
    for kv in filter(None, a.opt.split(",")):
        k, v = kv.split("=")
        eng.set_option(k, int(v))
    if a.variant:
        eng.set_option("variant", ddt.variant_names().index(a.variant))
    if a.sparse:
        lines, first = ddt.synth_sparse_model(T, D, F, a.full_levels, a.permille, 0)
        if a.bins:  # thresholds of dist 0 are uniform in [0, 1)
            import numpy as np
            ln = np.asarray(lines).view(np.uint32).reshape(-1, 4)
            thr = ln[:, 0].view(np.float32)
            ln[:, 0] = (np.floor(thr * a.bins) / np.float32(a.bins)).astype(np.float32).view(np.uint32)
        eng.load_model_sparse(ddt.make_sparse_params(T, D, F), lines, first)
    else:
        w, f = ddt.synth_model(T, D, F, 0)
        if a.wide_features > F:  # feature j of the model becomes column cols[j] of a tuple of --wide-features features
            import numpy as np
            cols = np.concatenate([[0], np.sort(np.random.default_rng(1).choice(np.arange(1, a.wide_features), F - 1, replace=False))]).astype(np.uint16)
            f = np.ascontiguousarray(f).view(np.uint16).reshape(-1)
            f = (f & np.uint16(0xF800)) | cols[f & np.uint16(0x7FF)]
            F = a.wide_features
        eng.load_model(ddt.make_params(T, D, F, sum_mode=a.sum_mode), w, f)
    info = eng.info()
    d = eng.synth_tuples_device(0, N, F)
    out = torch.empty(N, dtype=torch.float32, device="cuda")
    eng.score_device(d, out=out)
    torch.cuda.synchronize()
    if not a.sparse and N >= 4096 and a.sum_mode == 0:  # parity on a prefix, so that an experiment cannot be fast and wrong
        import numpy as np
        from oracle import oracle as O
        m = O.Model(O.make_params(T, D, F), w, f)
        xs = d[:4096].cpu().numpy().view(np.uint32)
        assert np.array_equal(out[:4096].cpu().numpy().view(np.uint32), O.score(m, xs).view(np.uint32)), "parity"
    ts = []
    for _ in range(a.reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        eng.score_device(d, out=out)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ms = min(ts)
    alg = N * (4 * F + 4) + info.model_bytes_unpadded
    print(f"{info.variant_name.decode()}: {T} trees x depth {D} x {F} features, {N} rows: {ms:.3f} ms/launch, {N / ms / 1e3:.1f} Mtuples/s, "
          f"{alg / ms / 1e6:.1f} GB/s algorithmic ({alg} bytes/launch)", flush=True)
    eng.close()


if __name__ == "__main__":
    main()
