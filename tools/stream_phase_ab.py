#!/usr/bin/env python3
"""A/B of the stream kernel's phased result stores (BASELINE config 1: 8 trees x depth 4 x 16 features): blocks per CU x score slots
per wave x write window, each timed (HIP events, best of reps) and compared WORD FOR WORD with the direct-store output of the same
engine (and a prefix with the oracle).  One JSON object per line."""
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
    ap.add_argument("--trees", type=int, default=8)
    ap.add_argument("--levels", type=int, default=4)
    ap.add_argument("--features", type=int, default=16)
    ap.add_argument("--rows", type=int, default=200_000_000)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--grid", default="0:1:0,0:0:0,5:0:0,5:0:2500,5:0:4096,6:0:2500,6:0:4096,4:0:3000,5:8:3000,6:4:3000",
                    help="comma separated blocks_per_cu:res_tiles:window_ticks (0 = the default of each)")
    a = ap.parse_args()
    T, D, F, N = a.trees, a.levels, a.features, a.rows
    eng = ddt.Engine(0)
    w, f = ddt.synth_model(T, D, F, 0)
    eng.load_model(ddt.make_params(T, D, F), w, f)
    d = eng.synth_tuples_device(0, N, F)
    ref = torch.empty(N, dtype=torch.float32, device="cuda")
    eng.set_option("stream_res_tiles", 1)
    eng.score_device(d, out=ref)
    torch.cuda.synchronize()
    m = O.Model(O.make_params(T, D, F), w, f)
    k = min(N, 65536)
    xs = d[:k].cpu().numpy().view(np.uint32)
    oracle_ok = bool(np.array_equal(ref[:k].cpu().numpy().view(np.uint32), O.score(m, xs).view(np.uint32)))
    out = torch.empty(N, dtype=torch.float32, device="cuda")
    for cell in a.grid.split(","):
        bpc, nb, win = (int(x) for x in cell.split(":"))
        eng.set_option("stream_blocks_per_cu", bpc)
        eng.set_option("stream_res_tiles", nb)
        eng.set_option("stream_window_ticks", win)
        out.fill_(float("nan"))
        eng.score_device(d, out=out)
        torch.cuda.synchronize()
        same = bool(torch.equal(out.view(torch.int32), ref.view(torch.int32)))
        ts = []
        for _ in range(a.reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            eng.score_device(d, out=out)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        ms = min(ts)
        print(json.dumps({"kernel": eng.info().variant_name.decode(), "rows": N, "blocks_per_cu": bpc, "res_tiles": nb, "window_ticks": win, "ms": round(ms, 4), "ms_mean": round(sum(ts) / len(ts), 4), "ms_max": round(max(ts), 4),
                          "gtuples_per_s": round(N / ms / 1e6, 2), "TB_per_s_algorithmic": round(N * (4 * F + 4) / ms / 1e9, 3),
                          "equals_direct_stores": same, "direct_equals_oracle_prefix": oracle_ok}), flush=True)
    eng.close()


if __name__ == "__main__":
    main()
