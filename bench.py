#!/usr/bin/env python3
"""bench.py -- headline benchmark of the scoring hot path (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run, one rank per GPU)

Workload (BASELINE.json `metric` / configs[2]): 1000 trees, depth 8, 32 fp32 features, 100 M synthetic
tuples (SURVEY.md 8(d) generator, resident in HBM before the timed region).  One "step" = one pass of the
hot path over the whole batch.  N=1: one engine holds all 1000 trees.  N>1: the ensemble is sharded
tree-wise (rank g holds trees [g*T/N, (g+1)*T/N)), every rank scores all tuples against its shard and the
per-tuple fp32 partial scores are combined with an RCCL all-reduce over xGMI, chunk-pipelined with the
scoring (total work fixed => "strong" scaling).  `value` = tuples scored by the whole job / wall time.

Extra objects on the JSON line:
  roofline     algorithmic HBM bytes (4F+4 per tuple + model once) / mean kernel time measured
               with HIP events on the launch stream, against the 8 TB/s HBM3E peak.  NOTE: this shape is
               LDS-gather bound (8000 dependent node visits per 132 compulsory bytes); visits/s is reported too.
  cpu_baseline the CPU oracle (a port of the reference RTL semantics; the reference has no CPU scorer) timed
               on a bounded prefix of the same batch on this box's host cores (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "distributed-decisiontrees_amd"))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--rows", type=int, default=100_000_000, help="tuples per step (BASELINE: 100 M)")
    ap.add_argument("--trees", type=int, default=1000)
    ap.add_argument("--levels", type=int, default=8)
    ap.add_argument("--features", type=int, default=32)
    ap.add_argument("--combine", default="allreduce", choices=["allreduce", "chain"])
    ap.add_argument("--shard", default="trees", choices=["trees", "rows"],
                    help="N>1: 'trees' = the headline mode (ensemble sharded tree-wise, partial scores all-reduced); "
                         "'rows' = the reference's other mode (replicated ensemble, tuples partitioned, scores all-gathered)")
    ap.add_argument("--chunk-rows", type=int, default=12_500_000, help="rows per pipelined collective (N>1)")
    ap.add_argument("--variant", type=int, default=-1, help="kernel variant id (-1 = engine's choice)")
    ap.add_argument("--sum-mode", type=int, default=0)
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl (= RCCL over xGMI) is the product path; gloo lets N ranks share one GPU for a functional test")
    ap.add_argument("--force-collectives", action="store_true",
                    help="N=1 only: run the multi-GPU chunk pipeline and the collectives in a one-rank group (overhead / sanity run)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time of the baseline sample")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    import ddt

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit(f"--gpus {args.gpus} needs one rank per GPU: launch with python -m torch.distributed.run "
                     f"--nnodes=1 --nproc-per-node {args.gpus} --master-addr 127.0.0.1 bench.py --gpus {args.gpus} ...")
        sys.exit(f"--gpus {args.gpus} != WORLD_SIZE {world}")
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (the scoring path has no CPU fallback)")
    local = local % torch.cuda.device_count() if args.backend == "gloo" else local
    torch.cuda.set_device(local)
    if world == 1 and args.force_collectives:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    if world > 1 or args.force_collectives:
        if args.backend == "nccl":
            try:
                dist.init_process_group("nccl", device_id=torch.device("cuda", local))
            except TypeError:  # older torch: no device_id keyword
                dist.init_process_group("nccl")
        else:
            dist.init_process_group("gloo")

    T, D, F, N = args.trees, args.levels, args.features, args.rows
    W = ddt.tuple_words(F)
    eng = ddt.Engine(local)
    eng.set_option("variant", args.variant)
    w, f = ddt.synth_model(T, D, F, 0)
    params = ddt.make_params(T, D, F, sum_mode=args.sum_mode)
    rows_mode = world > 1 and args.shard == "rows"
    eng.load_model(params, w, f, 0 if rows_mode else rank, 1 if rows_mode else world)
    info = eng.info()

    tuples = eng.synth_tuples_device(0, N, F, 0)          # resident in HBM before the timed region
    out = torch.empty(N, dtype=torch.float32, device=tuples.device)
    scorer = None
    if world > 1 or args.force_collectives:
        scorer = (ddt.RowShardedScorer(eng) if rows_mode else
                  ddt.ShardedScorer.from_engine(eng, mode=args.combine, chunk_rows=args.chunk_rows,
                                                force_collectives=args.force_collectives))

    # per-launch HIP-event times of the pass, taken by the library on the launch stream ("kernel_timing"):
    # pre-pass kernels (rank-quantised path only) and the scoring kernel proper
    kernel_ms = []
    if scorer is None:
        eng.set_option("kernel_timing", 1)

    def step(record: bool):
        if scorer is None:
            eng.score_device(tuples, out=out)
            if record:
                st = eng.stats()  # waits for this launch's end event
                kernel_ms.append((st.last_prepass_ms, st.last_score_ms))
        else:
            scorer.score(tuples, out=out)

    def fence():
        torch.cuda.synchronize()
        if world > 1 or args.force_collectives:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(False)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=tuples.device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    ms_per_step = dt / max(1, args.steps) * 1e3
    mtuples = N / (dt / max(1, args.steps)) / 1e6

    # ---- roofline of the dominant kernel (the per-shard scoring kernel) ----------------------------
    alg_bytes_per_launch = N * (4 * F + 4) + int(info.model_bytes_unpadded)  # tuples in, scores out, model once
    roofline = None
    if world == 1 and kernel_ms:
        pre_ms = sum(a for a, _ in kernel_ms) / len(kernel_ms)
        k_ms = sum(b for _, b in kernel_ms) / len(kernel_ms)  # the dominant (scoring) kernel
        ach = alg_bytes_per_launch / (k_ms * 1e-3) / 1e9
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")  # HBM bytes per launch from rocprofv3 --pmc, if collected
        if os.path.exists(pmc):
            try:
                pj = json.load(open(pmc))
                if pj.get("rows") == N and pj.get("trees") == T:
                    traffic = pj.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        t_local = info.tree_end - info.tree_begin
        roofline = {"bound": "hbm", "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": traffic,
                    "kernel": info.variant_name.decode(), "kernel_ms": round(k_ms, 4),
                    "prepass_ms": round(pre_ms, 4),  # transpose + rank kernels of the rank-quantised path (0 otherwise)
                    "alg_bytes_per_launch": alg_bytes_per_launch,
                    "binding_resource": "LDS gather pipe (2 DS ops per node visit) + VALU issue, not HBM",
                    "node_visits_per_s": round(N * t_local * D / (k_ms * 1e-3), 1),
                    "lds_ceiling_visits_per_s": 256 * 2.4e9 * 64 / 4}

    # ---- CPU baseline (oracle = port of the reference RTL semantics), rank 0 / N=1 only ---------------
    cpu = None
    parity = None
    if world == 1 and rank == 0 and not args.no_cpu_baseline:
        from oracle import oracle as O

        m = O.Model(O.make_params(T, D, F), w, f)
        O.score(m, tuples[: min(N, 4096)].cpu().numpy().view(np.uint32), sum_mode=O.SUM_REF_NATIVE)  # thread-pool warm-up
        probe = min(N, 262_144)
        xs = tuples[:probe].cpu().numpy().view(np.uint32)
        t1 = time.perf_counter()
        ref = O.score(m, xs, sum_mode=O.SUM_REF_NATIVE)
        rate = probe / max(1e-9, time.perf_counter() - t1)
        rows = int(max(probe, min(N, 16_000_000, rate * args.cpu_seconds)))
        xs = tuples[:rows].cpu().numpy().view(np.uint32)
        t1 = time.perf_counter()
        ref = O.score(m, xs, sum_mode=O.SUM_REF_NATIVE if args.sum_mode == 0 else O.SUM_F64_SEQ)
        cdt = time.perf_counter() - t1
        cpu = {"value": round(rows / cdt / 1e6, 4), "unit": "Mtuples/s", "cores": O.hw_threads(), "kind": "port",
               "sample": f"first {rows} rows of the same synthetic batch, all {T} trees, OpenMP over rows, "
                         f"{cdt:.1f} s (oracle/ddt_oracle.c: CPU restatement of the reference RTL semantics)"}
        got = out[:rows].cpu().numpy()
        parity = {"rows_checked": rows, "bit_exact": bool(np.array_equal(got.view(np.uint32), ref.view(np.uint32)))}

    if rank == 0:
        line = {
            "metric": "Mtuples/s scored, 1000 trees depth-8 / 32 feat" if (T, D, F) == (1000, 8, 32)
            else f"Mtuples/s scored, {T} trees depth-{D} / {F} feat",
            "value": round(mtuples, 3), "unit": "Mtuples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{T} trees x depth {D} x {F} fp32 features, {N} tuples/step, "
                                   + ("single engine" if world == 1 else
                                      f"row-sharded {world}x (replicas) + RCCL all-gather" if rows_mode else
                                      f"tree-sharded {world}x + RCCL {args.combine}"),
                       "trees": T, "levels": D, "features": F, "rows": N, "parallelism": f"row-shard{world}" if rows_mode else f"tree-shard{world}",
                       "combine": args.combine if world > 1 else None,
                       "collective_backend": (args.backend if world > 1 else None), "kernel": info.variant_name.decode(),
                       "sum_mode": "reference-order fp32" if args.sum_mode == 0 else "fp64 accumulate",
                       "device": info.device_name.decode()},
        }
        if roofline:
            line["roofline"] = roofline
        if cpu:
            line["cpu_baseline"] = cpu
        if parity:
            line["parity"] = parity
        print(json.dumps(line), flush=True)
    if world > 1 or args.force_collectives:
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
