#!/usr/bin/env python3
"""Time-bounded randomized soak of the scoring path against the CPU oracle (test infrastructure, run on the GPU box through gpurun).

tests/test_fuzz_gpu.py holds FIXED seeded cases, so that a failure there is a regression.  This tool draws FRESH cases from --seed for --seconds:
perfect-tree ensembles of every depth 1..15, sparse forests (both rank widths), one-vs-all models and tree shards, with batch sizes around
the tile boundaries, in the small-batch regime (cut launches) and beyond it, every sum mode, both compare modes, every cluster count, missing
values, host and device buffers -- the SAME engine scoring several batches of different sizes one after the other (workspaces grow, shrink,
are reused).  Every row of every batch is compared with the oracle bit for bit.  A mismatch is printed as one JSON line that reproduces it
(`--replay '<json>'`); the exit status is the number of failed cases (capped at 100).
"""
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

# engine sum_mode -> the oracle's: 0 reference order / IEEE adds, 1 fp64 in stream order, 2 reference order / the reference's adder
ORC_SUM = {0: O.SUM_REF_NATIVE, 1: O.SUM_F64_SEQ, 2: O.SUM_REF_FLOPOCO}
EDGE_ROWS = [1, 2, 63, 64, 65, 255, 256, 257, 511, 512, 513, 1023, 1024, 1025, 2047, 2048, 2049, 4096, 16_383, 16_384, 16_385, 65_536, 65_537,
             131_071, 131_072, 131_073, 163_840, 163_841]


def draw_rows(rng, cap):
    out = []
    for _ in range(int(rng.integers(1, 5))):
        k = rng.random()
        if k < 0.35:
            n = int(rng.choice(EDGE_ROWS))
        elif k < 0.7:
            n = int(rng.integers(1, 6000))
        else:
            n = int(rng.integers(6000, 400_000))
        out.append(max(1, min(n, cap)))
    return out


def draw_case(rng):
    kind = str(rng.choice(["perfect", "perfect", "perfect", "sparse", "sparse", "classes", "shard", "sparse_classes"]))
    c = {"kind": kind, "cmp_mode": int(rng.integers(0, 2)), "clusters": int(rng.choice([1, 2, 4, 8])), "sum_mode": int(rng.choice([0, 0, 2, 1])),
         "dist": int(rng.integers(0, 2)), "device": bool(rng.integers(0, 2)), "tseed": int(rng.integers(0, 1 << 30))}
    if kind in ("perfect", "shard", "classes"):
        D = int(rng.choice([1, 2, 3, 4, 5, 6, 7, 8, 8, 8, 9, 10, 11, 12, 13, 14, 15]))
        tmax = 1200 if D <= 8 else 300 if D <= 10 else 80 if D <= 12 else 24 if D <= 13 else 10
        T = int(rng.integers(1, tmax + 1))
        F = int(rng.integers(1, 33)) if rng.random() < 0.6 else int(rng.integers(33, 65)) if rng.random() < 0.7 else int(rng.integers(65, 260))
        c.update(T=T, D=D, F=F)
        if kind == "shard":
            G = min(int(rng.choice([2, 3, 4, 8])), T)  # (the library refuses more shards than trees: ddt_load_model_shard)
            c.update(G=G, g=int(rng.integers(0, G)))
        if kind == "classes":
            K = int(rng.integers(2, 13))
            c.update(K=K, T=max(K, (T // K) * K), interleaved=bool(rng.integers(0, 2)), sum_mode=int(rng.choice([0, 2])))
        cap = int(max(1, min(400_000, 6e9 // (c["T"] * D))))
    else:
        D = int(rng.integers(1, 21))
        T = int(rng.integers(1, 520)) if D >= 13 else int(rng.integers(1, 200))
        F = int(rng.integers(1, 130))
        full = int(rng.integers(0, min(D, 10) + 1))
        c.update(T=T, D=D, F=F, full=full, pm=int(rng.integers(300, 900)), bins=int(rng.choice([0, 0, 0, 255, 64])))
        if kind == "sparse_classes":
            K = int(rng.integers(2, 9))
            c.update(K=K, T=max(K, (T // K) * K), interleaved=bool(rng.integers(0, 2)), sum_mode=int(rng.choice([0, 2])))
        cap = int(max(1, min(300_000, 3e9 // (c["T"] * max(4, D)))))
    c["rows"] = draw_rows(rng, cap)
    return c


def snap_bins(s, bins, F):
    """Sparse forest with at most `bins` distinct thresholds per feature (a histogram-trained model): fp32 keys k / bins."""
    lines = s.node_lines.copy()
    thr = lines[:, 0].view(np.float32)
    ok = np.isfinite(thr)
    q = np.clip(np.round(np.where(ok, thr, 0.0) * bins), 0, bins - 1) / np.float32(bins)
    lines[:, 0] = np.where(ok, q.astype(np.float32), thr).view(np.uint32)
    return O.SparseModel(s.params, lines, s.first)


def first_bad(got, want):
    bad = np.flatnonzero(np.ascontiguousarray(got).view(np.uint32).reshape(-1) != np.ascontiguousarray(want).view(np.uint32).reshape(-1))
    return None if bad.size == 0 else (int(bad.size), [int(b) for b in bad[:4]])


def run_case(c):
    """-> (variant name, None) or (variant name, description of the first mismatch)"""
    kind, T, D, F = c["kind"], c["T"], c["D"], c["F"]
    sm, osm = c["sum_mode"], ORC_SUM[c["sum_mode"]]
    e = ddt.Engine(0)
    try:
        if kind in ("perfect", "shard", "classes"):
            m = O.gen_model(T, D, F, dist=c["dist"], cmp_mode=c["cmp_mode"], clusters=c["clusters"])
            p = m.params
            params = ddt.make_params(p.num_trees, p.num_levels, p.num_features, p.missing_bits, p.cmp_mode, p.clusters_per_tuple, sm)
            if kind == "classes":
                e.load_model_multiclass(params, m.wlines, m.flines, c["K"], c["interleaved"])
            elif kind == "shard":
                e.load_model(params, m.wlines, m.flines, c["g"], c["G"])
            else:
                e.load_model(params, m.wlines, m.flines)
            miss = p.missing_bits
        else:
            s = O.gen_sparse_model(T, D, F, c["full"], c["pm"], c["dist"], cmp_mode=c["cmp_mode"], clusters=c["clusters"])
            if c.get("bins"):
                s = snap_bins(s, c["bins"], F)
            q = s.params
            params = ddt.make_sparse_params(q.num_trees, q.num_levels, q.num_features, q.missing_bits, q.cmp_mode, q.clusters_per_tuple, sm)
            if kind == "sparse_classes":
                e.load_model_sparse(params, s.node_lines, s.first, 0, 1, c["K"], c["interleaved"])
            else:
                e.load_model_sparse(params, s.node_lines, s.first)
            miss = q.missing_bits
        name = e.info().variant_name.decode()
        for i, n in enumerate(c["rows"]):
            x = O.gen_tuples(c["tseed"] + 7919 * i, n, F, dist=1, missing_bits=miss)
            if kind in ("classes", "sparse_classes"):
                if kind == "classes":
                    wl, ws = O.classify_fast(m, x, c["K"], c["interleaved"], sum_mode=osm)
                else:
                    wl, ws = O.classify_sparse(s, x, c["K"], c["interleaved"], sum_mode=osm)
                if c["device"]:
                    gl, gs = e.classify_device(torch.from_numpy(x.view(np.int32)).cuda())
                    torch.cuda.synchronize()
                    gl, gs = gl.cpu().numpy(), gs.cpu().numpy()
                else:
                    gl, gs = e.classify(x, want_scores=True)
                b = first_bad(gs, ws)
                if b:
                    return name, f"batch {i} ({n} rows): {b[0]} class sums differ, first {b[1]}"
                if not np.array_equal(gl, wl):
                    return name, f"batch {i} ({n} rows): {int((gl != wl).sum())} labels differ"
                continue
            if kind == "shard":
                lo, hi = ddt.shard_bounds(T, c["G"])[c["g"]]  # (an empty shard scores +0: ceil(T / G) trees per shard)
                want = O.score_shard(m, x, lo, hi, sum_mode=osm)
            elif kind == "perfect":
                want = O.score_fast(m, x, sum_mode=osm) if sm != 1 else O.score(m, x, sum_mode=osm)
            else:
                want = O.score_sparse_fast(s, x, sum_mode=osm) if sm != 1 else O.score_sparse(s, x, sum_mode=osm)
            if c["device"]:
                got = e.score_device(torch.from_numpy(x.view(np.int32)).cuda())
                torch.cuda.synchronize()
                got = got.cpu().numpy()
            else:
                got = e.score(x)
            b = first_bad(got, want)
            if b:
                return name, f"batch {i} ({n} rows): {b[0]} rows differ, first {b[1]}"
        return name, None
    finally:
        e.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--seconds", type=float, default=300.0)
    ap.add_argument("--max-cases", type=int, default=1_000_000)
    ap.add_argument("--replay", default="", help="one case as printed by a failing run")
    ap.add_argument("--verbose", action="store_true")
    a = ap.parse_args()
    if a.replay:
        name, err = run_case(json.loads(a.replay))
        print(name, "OK" if err is None else "FAIL: " + err)
        return 1 if err else 0
    rng = np.random.default_rng(a.seed)
    t0 = time.time()
    done = failed = 0
    kernels = {}
    while time.time() - t0 < a.seconds and done < a.max_cases:
        c = draw_case(rng)
        try:
            name, err = run_case(c)
        except ddt.DDTError as ex:
            if ex.code == -5:  # DDT_EUNSUPPORTED: a shape the library refuses (documented limits) is not a failure
                name, err = "refused", None
            else:
                name, err = "error", repr(ex)
        except Exception as ex:  # the oracle's own argument checks included
            name, err = "error", repr(ex)
        done += 1
        kernels[name] = kernels.get(name, 0) + 1
        if err is not None:
            failed += 1
            print("FAIL", name, err, json.dumps(c), flush=True)
        elif a.verbose:
            print("ok", name, json.dumps(c), flush=True)
    print(json.dumps({"seed": a.seed, "cases": done, "failed": failed, "seconds": round(time.time() - t0, 1),
                      "kernels": dict(sorted(kernels.items(), key=lambda kv: -kv[1]))}))
    return min(failed, 100)


if __name__ == "__main__":
    sys.exit(main())
