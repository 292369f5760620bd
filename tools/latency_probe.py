#!/usr/bin/env python3
"""Small-batch latency of one scoring call through the C-ABI (device-resident tuples, one `ddt_score_device` + a stream sync per call).

bench.py measures throughput at BASELINE's row counts; this is the other end: what ONE call costs when the batch is a few rows to a few
hundred thousand -- the regime a serving caller sees.  Per model and batch size: median and 90th percentile of `reps` calls in
microseconds, the rows per second that is, the kernel, and a bit-for-bit check of the smallest and the largest batch against the oracle.

Usage: latency_probe.py [--configs 3,2,6,4] [--rows 1,64,1024,...] [--reps 40] [--opt key=value ...] [--json out.json]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "distributed-decisiontrees_amd"))
sys.path.insert(0, ROOT)

SHAPES = {3: (1000, 8, 32), 2: (100, 6, 28), 6: (512, 12, 32), 4: (512, 16, 64), 1: (8, 4, 16), 5: (1000, 8, 32),   # 5: 10 classes x 100 trees, classified
          # not BASELINE configs: 1000 trees at XGBoost's default depth, and shallow / odd depths
          106: (1000, 6, 28), 107: (500, 7, 32), 104: (2000, 4, 16)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="3,2,6,4")
    ap.add_argument("--rows", default="1,64,1024,4096,16384,65536,262144,1048576")
    ap.add_argument("--reps", type=int, default=40)
    ap.add_argument("--opt", action="append", default=[])
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--graph", action="store_true",
                    help="also capture the call into a HIP graph (torch.cuda.CUDAGraph around ddt_score_device on the capture stream, after the warm-up calls "
                         "that size the workspaces) and time its replays: us_graph_median / us_graph_host_median, and the replayed result against the oracle")
    ap.add_argument("--json", default=None)
    args = ap.parse_args()
    import numpy as np
    import torch

    import ddt

    res = []
    for cfg in [int(c) for c in args.configs.split(",")]:
        T, D, F = SHAPES[cfg]
        eng = ddt.Engine(0)
        for kv in args.opt:
            k, _, v = kv.partition("=")
            eng.set_option(k, int(v))
        if cfg == 4:
            lines, first = ddt.synth_sparse_model(T, D, F, 8, 700, 0)
            params = ddt.make_sparse_params(T, D, F)
            eng.load_model_sparse(params, lines, first)
        elif cfg == 5:
            w, f = ddt.synth_model(T, D, F, 0)
            eng.load_model_multiclass(ddt.make_params(T, D, F, clusters=ddt.default_clusters(T // 10)), w, f, 10, True)
        else:
            w, f = ddt.synth_model(T, D, F, 0)
            params = ddt.make_params(T, D, F)
            eng.load_model(params, w, f)
        rows = [int(r) for r in args.rows.split(",")]
        nmax = max(rows)
        tuples = eng.synth_tuples_device(0, nmax, F, 0)
        W = ddt.tuple_words(F)
        out = torch.empty(nmax, dtype=torch.float32, device=tuples.device)
        ref = None
        if not args.no_check:
            from oracle import oracle as O  # the checker, behind the timed calls

            ncheck = min(nmax, 16384)
            x = tuples[:ncheck].cpu().numpy().view(np.uint32)
            if cfg == 5:
                ref = None   # (labels and class sums: tests/test_q16_cluster_split.py)
            elif cfg == 4:
                ref = O.score_sparse_fast(O.SparseModel(O.make_sparse_params(T, D, F), lines, first), x)
            else:
                ref = O.score_fast(O.Model(O.make_params(T, D, F), w, f), x)
        cls = torch.empty((10, nmax), dtype=torch.float32, device=tuples.device) if cfg == 5 else None
        lab = torch.empty(nmax, dtype=torch.int32, device=tuples.device) if cfg == 5 else None

        def call(t_in, o, n):
            if cfg == 5:
                eng.classify_device(t_in, class_scores=cls.view(-1)[: 10 * n].view(10, n), labels=lab[:n])
            else:
                eng.score_device(t_in, out=o)

        for n in rows:
            t_in = tuples[:n]
            o = out[:n]
            for _ in range(5):
                call(t_in, o, n)
            torch.cuda.synchronize()
            ts = []
            for _ in range(args.reps):
                t0 = time.perf_counter()
                call(t_in, o, n)
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t0) * 1e6)
            ts.sort()
            med, p90 = ts[len(ts) // 2], ts[int(len(ts) * 0.9)]
            hs = []   # the host's share: the call returns when its launches are queued (no sync inside the timed part)
            for _ in range(args.reps):
                t0 = time.perf_counter()
                call(t_in, o, n)
                hs.append((time.perf_counter() - t0) * 1e6)
                torch.cuda.synchronize()
            hs.sort()
            ok = None
            if ref is not None and n <= len(ref):
                ok = bool(np.array_equal(o.cpu().numpy().view(np.uint32), np.asarray(ref[:n], dtype=np.float32).view(np.uint32)))
            gr = {}
            if args.graph:
                # the library's launches of one call (memset of the tile flags, rank pre-pass, scoring kernel(s), combine) as ONE graph launch: legal once
                # the workspaces have their size (a call that grows one synchronises the device and reallocates, which a capture refuses)
                try:
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g):
                        call(t_in, o, n)
                    o.zero_()
                    g.replay()
                    torch.cuda.synchronize()
                    if ref is not None and n <= len(ref):
                        gr["graph_bit_exact"] = bool(np.array_equal(o.cpu().numpy().view(np.uint32), np.asarray(ref[:n], dtype=np.float32).view(np.uint32)))
                    gs, gh = [], []
                    for _ in range(args.reps):
                        t0 = time.perf_counter()
                        g.replay()
                        t1 = time.perf_counter()
                        torch.cuda.synchronize()
                        gs.append((time.perf_counter() - t0) * 1e6)
                        gh.append((t1 - t0) * 1e6)
                    gs.sort()
                    gh.sort()
                    gr["us_graph_median"] = round(gs[len(gs) // 2], 1)
                    gr["us_graph_host_median"] = round(gh[len(gh) // 2], 1)
                    del g
                except Exception as ex:  # a capture the runtime refused: said, not hidden
                    gr["graph_error"] = repr(ex)[:300]
                    torch.cuda.synchronize()
            info = eng.info()
            r = {"config": cfg, "trees": T, "depth": D, "features": F, "rows": n, "us_median": round(med, 1), "us_p90": round(p90, 1), "us_host_call_median": round(hs[len(hs) // 2], 1),
                 "mtuples_per_s": round(n / med, 3), "kernel": info.variant_name.decode(), "bit_exact": ok}
            r.update(gr)
            res.append(r)
            print(json.dumps(r), flush=True)
        eng.close()
        del tuples, out
    if args.json:
        json.dump(res, open(args.json, "w"), indent=1)


if __name__ == "__main__":
    main()
