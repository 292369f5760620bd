#!/usr/bin/env python3
"""BASELINE config 5 on one GPU: 10 classes x 100 trees, depth 8, 32 features -- per-class scores + argmax
(device-resident tuples, HIP-event timing).  Compares the engine's choice with forced kernel variants."""
import argparse
import json
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
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--classes", type=int, default=10)
    ap.add_argument("--trees", type=int, default=1000)
    ap.add_argument("--variants", default="auto,d8_t1024_r1_c4_u4_dma_f")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "classify_bench.json"))
    a = ap.parse_args()
    T, D, F, K, N = a.trees, 8, 32, a.classes, a.rows
    w, f = ddt.synth_model(T, D, F)
    C = ddt.default_clusters(T // K)
    e = ddt.Engine(0)
    d = e.synth_tuples_device(0, N, F)
    m = O.Model(O.make_params(T, D, F, clusters=C), w, f)
    want_l, _ = O.classify(m, d[:4096].cpu().numpy().view(np.uint32), K, interleaved=True)
    names = ddt.variant_names()
    res = []
    for vn in a.variants.split(","):
        e.set_option("variant", -1 if vn == "auto" else names.index(vn))
        e.load_model_multiclass(ddt.make_params(T, D, F, clusters=C), w, f, K, True)
        labels, _ = e.classify_device(d)
        torch.cuda.synchronize()
        ok = bool(np.array_equal(labels[:4096].cpu().numpy(), want_l))
        ts = []
        for _ in range(a.reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            e.classify_device(d)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        ts.sort()
        r = {"variant": e.info().variant_name.decode(), "requested": vn, "ok": ok, "min_ms": ts[0], "med_ms": ts[len(ts) // 2],
             "mtuples_s": N / ts[0] / 1e3, "classes": K, "trees": T, "rows": N}
        res.append(r)
        print(f"{K}x{T // K} trees d8 F32 {N} rows  {r['variant']:<28} ok={ok} min {ts[0]:9.3f} ms  {r['mtuples_s']:9.1f} Mtuples/s", flush=True)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(res, open(a.out, "w"), indent=1)
    e.close()


if __name__ == "__main__":
    main()
