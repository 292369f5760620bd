#!/usr/bin/env python3
"""Kernel-variant sweep on one GPU: times every compiled variant that fits a shape (HIP events, min/median
of several launches) and checks each against the oracle on a small prefix.  Output: table + JSON."""
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
    ap.add_argument("--shapes", default="1000x8x32x8000000,100x6x28x10000000,8x4x16x10000000")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--only", default="", help="substring filter on variant names (comma separated: any of them)")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "sweep.json"))
    ap.add_argument("--opt", default="", help="engine options, comma separated key=value (e.g. q16_fused_prepass=0)")
    a = ap.parse_args()
    eng = ddt.Engine(0)
    for kv in filter(None, a.opt.split(",")):
        k, v = kv.split("=")
        eng.set_option(k, int(v))
    names = ddt.variant_names()
    res = []
    for shp in a.shapes.split(","):
        T, D, F, N = [int(v) for v in shp.split("x")]
        w, f = ddt.synth_model(T, D, F)
        m = O.Model(O.make_params(T, D, F), w, f)
        d = eng.synth_tuples_device(0, N, F)
        xs = d[:2048].cpu().numpy().view(np.uint32)
        want = O.score(m, xs)
        out = torch.empty(N, dtype=torch.float32, device="cuda")
        for v, name in enumerate(names):
            if a.only and not any(k in name for k in a.only.split(",")):
                continue
            if name == "generic" and N * T * D > 2e12:
                continue
            try:
                eng.set_option("variant", v)
                eng.load_model(ddt.make_params(T, D, F), w, f)
            except ddt.DDTError:
                continue
            eng.score_device(d, out=out)
            torch.cuda.synchronize()
            ok = bool(np.array_equal(out[:2048].cpu().numpy().view(np.uint32), want.view(np.uint32)))
            ts = []
            for _ in range(a.reps):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                eng.score_device(d, out=out)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            ts.sort()
            best, med = ts[0], ts[len(ts) // 2]
            info = eng.info()
            r = {"shape": shp, "variant": name, "ok": ok, "ms_min": round(best, 4), "ms_med": round(med, 4),
                 "Mtuples_s": round(N / best / 1e3, 1), "Tvisits_s": round(N * T * D / best / 1e9, 3),
                 "hbm_GBs": round((N * (4 * F + 4)) / best / 1e6, 1), "lds_bytes": info.lds_bytes,
                 "tile": info.tile_tuples}
            res.append(r)
            print(f"{shp:>24} {name:<28} ok={ok} min {best:9.3f} ms  med {med:9.3f} ms  {r['Mtuples_s']:>10.1f} Mtuples/s "
                  f"{r['Tvisits_s']:6.3f} Tvisits/s  {r['hbm_GBs']:8.1f} GB/s  lds {info.lds_bytes}", flush=True)
        del d, out
        torch.cuda.empty_cache()
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(res, open(a.out, "w"), indent=1)
    eng.close()


if __name__ == "__main__":
    main()
