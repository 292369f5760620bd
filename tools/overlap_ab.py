#!/usr/bin/env python3
"""A/B of the pre-pass / scoring overlap (engine option "prepass_overlap_rows") on one GPU: the rank pre-pass of piece k+1 on the
engine's own stream against the scoring kernel of piece k on the caller's.  Per shape: one launch (0), then pieces of the given
sizes, with and without a high-priority pre-pass stream; wall time per pass from events on the caller's stream, the pieces' kernel
times from the library ("kernel_timing", summed over the pieces), a prefix and the tail of the scores against the oracle / the
one-launch run (bit-exact).  Usage: tools/overlap_ab.py [--shapes cfg3,shard8,cfg2] [--out FILE.json]"""
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

SHAPES = {  # name: (trees, depth, features, rows, shard (index, count), piece sizes)
    "cfg3": (1000, 8, 32, 100_000_000, (0, 1), (12_500_000, 6_250_000, 25_000_000, 3_125_000)),
    "shard8": (1000, 8, 32, 100_000_000, (3, 8), (12_500_000, 6_250_000, 25_000_000, 3_125_000)),   # what one of 8 ranks computes
    "cfg2": (100, 6, 28, 10_000_000, (0, 1), (2_500_000, 1_250_000, 625_000, 5_000_000)),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="cfg3,shard8,cfg2")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    res = []
    for name in a.shapes.split(","):
        T, D, F, N, shard, pieces = SHAPES[name]
        eng = ddt.Engine(0)
        d = eng.synth_tuples_device(0, N, F)
        out = torch.empty(N, dtype=torch.float32, device="cuda")
        w, f = ddt.synth_model(T, D, F)
        eng.load_model(ddt.make_params(T, D, F), w, f, *shard)
        b, en = ddt.shard_bounds(T, shard[1])[shard[0]]
        m = O.Model(O.make_params(T, D, F), w, f)
        xs = d[:4096].cpu().numpy().view(np.uint32)
        want = O.score_shard(m, xs, b, en, sum_mode=O.SUM_REF_NATIVE)
        base = None
        for prio, rows in [(0, 0)] + [(0, p) for p in pieces] + [(1, pieces[0]), (1, pieces[1]), (0, -1), (0, 0)]:
            eng.set_option("prepass_overlap_priority", prio)
            eng.set_option("prepass_overlap_rows", rows)
            eng.set_option("kernel_timing", 0)
            out.zero_()
            eng.score_device(d, out=out)
            torch.cuda.synchronize()
            ok = bool(np.array_equal(out[:4096].cpu().numpy().view(np.uint32), want.view(np.uint32)))
            if base is None:
                base = out.clone()
            same = bool(torch.equal(out.view(torch.int32), base.view(torch.int32)))   # every row, bit for bit, against the one-launch run
            best = 1e30
            for _ in range(a.reps):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                eng.score_device(d, out=out)
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1))
            eng.set_option("kernel_timing", 1)
            eng.score_device(d, out=out)
            torch.cuda.synchronize()
            st = eng.stats()
            r = {"shape": name, "trees_on_engine": int(en - b), "rows": N, "kernel": eng.info().variant_name.decode(), "piece_rows": rows, "prepass_stream_priority": prio,
                 "ms": round(best, 4), "mtuples_per_s": round(N / best / 1e3, 1), "prepass_ms_sum": round(st.last_prepass_ms, 4), "score_ms_sum": round(st.last_score_ms, 4),
                 "bit_exact_vs_oracle_prefix": ok, "bit_exact_vs_one_launch_all_rows": same}
            res.append(r)
            print(json.dumps(r), flush=True)
        eng.close()
        del d, out, base
        torch.cuda.empty_cache()
    if a.out:
        json.dump({"what": "tools/overlap_ab.py: pre-pass / scoring overlap, best of %d passes per row" % a.reps, "runs": res}, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
