#!/usr/bin/env python3
"""Sparse-forest kernel sweep on one GPU (BASELINE config 4 shape): times score_sparse_kernel for every K (levels
staged in LDS) that fits and both deep-record orders, checks each against the sparse oracle on a prefix.
Output: table + JSON (gpurun_out/sparse_sweep.json)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "distributed-decisiontrees_amd"))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import ddt  # noqa: E402
from oracle import oracle as O  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trees", type=int, default=512)
    ap.add_argument("--depth", type=int, default=16)
    ap.add_argument("--features", type=int, default=64)
    ap.add_argument("--full-levels", type=int, default=10)
    ap.add_argument("--permille", type=int, default=700)
    ap.add_argument("--rows", type=int, default=2_000_000)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--only", default="", help="comma separated substrings of sparse variant names (default: all)")
    ap.add_argument("--orders", default="0")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "sparse_sweep.json"))
    a = ap.parse_args()
    T, D, F, N = a.trees, a.depth, a.features, a.rows
    t0 = time.time()
    lines, first = ddt.synth_sparse_model(T, D, F, a.full_levels, a.permille, 0)
    print(f"model: {T} trees, {lines.shape[0]} internal nodes ({lines.shape[0] / T:.0f}/tree, {lines.nbytes / 1e6:.1f} MB) in {time.time() - t0:.1f}s", flush=True)
    eng = ddt.Engine(0)
    d = eng.synth_tuples_device(0, N, F)
    xs = d[:1024].cpu().numpy().view(np.uint32)
    s = O.SparseModel(O.make_sparse_params(T, D, F), lines, first)
    want, gold = O.score_sparse(s, xs, want_gold=True)
    depth = O.sparse_mean_depth(s, xs[:256])
    print(f"mean visits per (tuple, tree): {depth:.2f}", flush=True)
    out = torch.empty(N, dtype=torch.float32, device="cuda")
    p = ddt.make_sparse_params(T, D, F)
    res = []
    ref_out = None
    names = ddt.variant_names()
    only = [v for v in a.only.split(",") if v]
    for vid, name in enumerate(names):
        if not name.startswith("sparse_") or (only and not any(o in name for o in only)):
            continue
        for order in [int(v) for v in a.orders.split(",")]:
            try:
                eng.set_option("sparse_deep_order", order)
                eng.set_option("variant", vid)
                eng.load_model_sparse(p, lines, first)
            except ddt.DDTError as ex:
                print(f"{name} order={order}: {ex}")
                continue
            info = eng.info()
            eng.score_device(d, out=out)
            torch.cuda.synchronize()
            ok = bool(np.array_equal(out[:1024].cpu().numpy().view(np.uint32), want.view(np.uint32)))
            if ref_out is None:
                ref_out = out.clone()   # every later variant's scores of ALL rows against the first variant's, bit for bit
            same = bool(torch.equal(out.view(torch.int32), ref_out.view(torch.int32)))
            ts = []
            for _ in range(a.reps):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                eng.score_device(d, out=out)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            ms = min(ts)
            K = int(name.split("_k")[1].split("_")[0])
            r = {"order": order, "variant": info.variant_name.decode(), "visits_per_s": round(N * T * depth / ms * 1e3, 1),
                 "deep_gathers_per_s": round(N * T * max(0.0, depth - K) / ms * 1e3, 1), "lds_bytes": info.lds_bytes, "image_MB": info.image_bytes / 1e6,
                 "ms": round(ms, 3), "Mtuples_s": round(N / ms / 1e3, 2), "bit_exact": ok, "all_rows_equal_first_variant": same}
            print(r, flush=True)
            res.append(r)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    eng.set_option("variant", -1)
    json.dump({"shape": vars(a), "nodes": int(lines.shape[0]), "mean_visits_per_tuple_and_tree": depth, "results": res}, open(a.out, "w"), indent=1)
    eng.close()


if __name__ == "__main__":
    main()
