#!/usr/bin/env python3
"""bench.py -- headline benchmark of the scoring hot path (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W        (N>1: one rank per GPU -- under torch.distributed.run as the driver starts it,
                                                        or on its own: without WORLD_SIZE it launches the ranks itself)

Default workload (BASELINE.json `metric` / configs[2]): 1000 trees, depth 8, 32 fp32 features, 100 M synthetic
tuples (SURVEY.md 8(d) generator, resident in HBM before the timed region).  One "step" = one pass of the hot path
over the whole batch.  N=1: one engine holds all 1000 trees.  N>1: the ensemble is sharded tree-wise (rank g holds
trees [g*ceil(T/N), ...)), every rank scores all tuples against its shard and the per-tuple fp32 partial scores are
combined over RCCL -- issued from C++ behind the C-ABI (ddt_comm_* / ddt_score_sharded_device, csrc/ddt_comm.cpp),
chunk-pipelined with the scoring (total work fixed => "strong" scaling).  `value` = tuples scored by the whole job /
wall time (max over ranks).

  --config 4   BASELINE configs[3]: a random-forest-like SPARSE model (512 trees, depth <= 16, 64 features; synthetic
               growth of include/ddt.h ddt_synth_sparse_model, ~10^4 internal nodes per tree), 10 M tuples per step,
               scored by the explicit-children kernel (csrc/ddt_sparse.hip).

Extra objects on the JSON line:
  roofline     algorithmic HBM bytes (4F+4 per tuple + model once) / mean kernel time measured with HIP
               events on the launch stream, against the 8 TB/s HBM3E peak.  NOTE: these shapes are not HBM bound
               (config 3: 8000 dependent node visits per 132 compulsory bytes -> LDS gather pipe; config 4: vector-memory
               gathers); node visits/s and the pipe ceilings, derived from the device properties, are reported too.
  cpu_baseline the CPU oracle (a port of the reference RTL semantics; the reference has no CPU scorer) timed on a
               bounded prefix of the same batch on this box's host cores (rank 0, N=1 only).
  streamed     N=1: the same model fed from HOST memory through the pinned double-buffered hipMemcpyAsync feeder
               (PCIe-inclusive rate on a bounded sample; never `value`).
  scaling_detail / other_modes
               N>1 (or --force-collectives), measured BEHIND the timed region and never `value`: the rank's shard scored with no
               collective (what the combine costs is then on the line), and the same batch through the library's other ways of
               running the job -- chain combine, untapered all-reduce, half / double chunk_rows, a high-priority comm stream, the
               row-sharded replicas (whole ensemble per GPU, tuples partitioned) and the hybrids (tree groups of 2 / 4 ranks x row groups).  A watchdog prints the headline line and
               exits if one of these legs -- none has run with real peers before the driver's scaling run -- does not return.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "distributed-decisiontrees_amd"))
sys.path.insert(0, ROOT)

# multi-process GPU work on this platform needs dmabuf IPC (RCCL / tensor sharing fail with the legacy mode); the launcher
# normally exports it already -- set before the HIP runtime is loaded, never overriding an explicit choice
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling


def self_launch_command(n_gpus, argv=None, port=None):
    """The command line `python bench.py --gpus N ...` turns itself into when no launcher set WORLD_SIZE."""
    if port is None:
        import socket

        with socket.socket() as sk:  # a free rendezvous port (two benches on one box must not collide on a fixed one)
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(sys.argv[1:] if argv is None else argv)


# BASELINE.json configs (1: at 200 M rows, the HBM-bound shape; 6: the reference's own example configuration, profiler/profiler.cpp:32-38)
CONFIG_SHAPES = {1: (8, 4, 16, 200_000_000), 2: (100, 6, 28, 10_000_000), 3: (1000, 8, 32, 100_000_000),
                 4: (512, 16, 64, 10_000_000), 5: (1000, 8, 32, 10_000_000), 6: (512, 12, 32, 10_000_000)}


def hybrid_tree_groups(world):
    """Tree-group sizes Gt of the hybrid legs: proper divisors of the job that leave at least two row groups."""
    return [Gt for Gt in (2, 4) if world % Gt == 0 and world // Gt >= 2]


def self_launch(n_gpus):
    import subprocess

    cmd = self_launch_command(n_gpus)
    print("bench.py: --gpus %d without WORLD_SIZE: launching %s" % (n_gpus, " ".join(cmd)), file=sys.stderr, flush=True)
    return subprocess.call(cmd)


def collect_other_configs(device_index, budget_s, runner=None, configs=(1, 2, 5, 6, 4)):
    """`other_configs` of the default command's line: every other BASELINE config as a short run (run_side_config), within a time budget;
    a leg that fails or would start past the budget says so instead of costing the line."""
    runner = runner or run_side_config
    out = {}
    t0 = time.perf_counter()
    for cfg in configs:
        spent = time.perf_counter() - t0
        if spent > budget_s:
            out[str(cfg)] = {"skipped": f"the legs before it took {spent:.0f} s of the {budget_s:.0f} s budget"}
            continue
        try:
            out[str(cfg)] = runner(cfg, device_index) if runner is not run_side_config else runner(cfg, device_index, deadline=t0 + budget_s)
        except Exception as ex:  # diagnostics must never cost the headline line
            out[str(cfg)] = {"error": repr(ex)}
    out["note"] = ("BASELINE configs 1, 2, 4, 5 and 6 (= the reference's own example, 512 x d12 x 32) on the same GPU behind the timed region: full row "
                   "counts, a few steps each, HIP-event kernel time -> roofline.frac, a prefix of the result bit for bit against the oracle; never `value`")
    out["seconds"] = round(time.perf_counter() - t0, 1)
    return out


def collect_rank_proxies(device_index, tuples, headline_ms, shape, model, check_rows=262_144, runner=None):
    """`other_modes.per_rank_proxies` of the default command's line: what ONE rank of an 8-GPU job does, run on this one GPU
    behind the timed region, so that the driver's line carries a scaling figure even when no 8-GPU node is available to it -- never `value`:
      shard_of_8           the named mode (tree-sharded, PCIeReceiver.sv:241-264): shard 3 of 8 (125 trees) over ALL the step's tuples
      hybrid_rank_of_2x4   the hybrid of 4 row groups x 2 tree shards: shard 1 of 2 (500 trees) over a quarter of the tuples
      replica_of_8         row-sharded replicas (PCIeReceiver.sv:289-312): the whole ensemble over an eighth of the tuples
    each with its kernel, ms per pass, `compute_only_x` = the headline's ms per step / that (an UPPER bound of the job's speed-up: no collective,
    no CUs held by one) and a prefix of its partial scores bit for bit against the oracle's model of the shard (O.score_shard); beside them the
    analytic model's 8-GPU prediction WITH collectives (ddt/perf_model.py), whose RCCL CU count is an assumption and says so."""
    import numpy as np
    import torch

    import ddt
    from ddt import perf_model as PM
    from oracle import oracle as O

    t_begin = time.perf_counter()
    T, D, F = shape
    w, f = model
    N = tuples.shape[0]
    out = {}
    m = O.Model(O.make_params(T, D, F), w, f)
    legs = (("shard_of_8", 3, 8, N), ("hybrid_rank_of_2x4", 1, 2, N // 4 // 1024 * 1024), ("replica_of_8", 0, 1, N // 8 // 1024 * 1024))
    for name, idx, cnt, rows in legs:
        try:
            if runner is not None:
                out[name] = runner(name, idx, cnt, rows)
                continue
            eng = ddt.Engine(device_index)
            try:
                eng.load_model(ddt.make_params(T, D, F), w, f, idx, cnt)
                info = eng.info()
                d = tuples[:rows]
                o = torch.empty(rows, dtype=torch.float32, device=tuples.device)
                eng.score_device(d, out=o)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(2):
                    eng.score_device(d, out=o)
                torch.cuda.synchronize()
                ms = (time.perf_counter() - t0) / 2 * 1e3
                chk = min(rows, check_rows)
                xs = d[:chk].cpu().numpy().view(np.uint32)
                ref = O.score_shard(m, xs, int(info.tree_begin), int(info.tree_end), sum_mode=O.SUM_REF_NATIVE)
                out[name] = {"trees": int(info.tree_end - info.tree_begin), "rows": int(rows), "ms": round(ms, 4), "kernel": info.variant_name.decode(),
                             "compute_only_x": round(headline_ms / ms, 3) if ms > 0 else None,
                             "bit_exact": bool(np.array_equal(o[:chk].cpu().numpy().view(np.uint32), ref.view(np.uint32))), "rows_checked": int(chk)}
            finally:
                eng.close()
        except Exception as ex:  # diagnostics must never cost the headline line
            out[name] = {"error": repr(ex)}
    try:
        g = PM.Mi355x()
        one = PM.tree_sharded_ms(T, 1, D, float(N))["ms"]
        pred = {"tree_sharded_8": PM.tree_sharded_ms(T, 8, D, float(N), g), "hybrid_tree2_x_rows4_gathered": PM.hybrid_ms(T, 2, 4, D, float(N), g, gather=True),
                "replicas_8": PM.row_sharded_ms(T, 8, D, float(N), g)}
        out["model_8gpu"] = {k: {"ms": round(v["ms"], 3), "x_over_model_1gpu": round(one / v["ms"], 2)} for k, v in pred.items()}
        out["model_8gpu"]["assumptions"] = (f"ddt/perf_model.py: RCCL holds {g.rccl_cus} CUs while a collective runs (an ASSUMPTION: never observed, no multi-GPU "
                                            f"run exists), ring all-reduce at {g.allreduce_alg_bytes_per_s / 1e9:.0f} GB/s algorithmic over xGMI (NOT measured); "
                                            "engine costs fitted to one-GPU measurements within 1.3 %")
    except Exception as ex:
        out["model_8gpu"] = {"error": repr(ex)}
    out["note"] = ("one rank's workload of three 8-GPU jobs on ONE GPU, 2 passes each behind the timed region, no collective: compute_only_x bounds the job's "
                   "speed-up from above; north_star asks >= 6x for the tree-sharded mode")
    out["seconds"] = round(time.perf_counter() - t_begin, 2)
    return out


def collect_small_batches(eng, tuples, out, ref_bits, rows_list=(1024, 16_384, 131_072), reps=15):
    """`other_modes.small_batches` of the default command's line: what ONE call of the headline model costs on a batch of a few tiles -- the
    regime a serving caller sees; `value` is measured at 100 M tuples per call.  Per batch size: median microseconds of a call (launch + stream
    sync, device-resident tuples) with the launch cut into slices of the image (csrc/ddt_engine.cpp cluster_split_of: one block per (tile, PU
    group run / cluster), the adds in the reference's order behind it) and uncut (one block per tile walks all the trees), and the cut call's
    scores bit for bit against the timed region's own (`ref_bits`: the headline's output for the same rows, itself checked against the oracle)."""
    import numpy as np
    import torch

    res = {}
    for rows in rows_list:
        d, o = tuples[:rows], out[:rows]
        leg = {}
        for name, opt in (("cut", -1), ("uncut", 0)):
            eng.set_option("q16_cluster_split", opt)
            for _ in range(3):
                eng.score_device(d, out=o)
            torch.cuda.synchronize()
            ts = []
            for _ in range(reps):
                t0 = time.perf_counter()
                eng.score_device(d, out=o)
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t0) * 1e6)
            leg[f"us_{name}"] = round(sorted(ts)[len(ts) // 2], 1)
            if name == "cut":
                leg["bit_exact_vs_timed_result"] = bool(np.array_equal(o.cpu().numpy().view(np.uint32), ref_bits[:rows]))
        leg["x"] = round(leg["us_uncut"] / leg["us_cut"], 2) if leg["us_cut"] > 0 else None
        res[str(rows)] = leg
    eng.set_option("q16_cluster_split", -1)
    res["note"] = ("median wall microseconds of one ddt_score_device call + stream sync from Python (host overhead ~20-30 us included); the cut launch is the "
                   "library's automatic choice for batches of up to 384 tiles; never `value`")
    return res


def run_side_config(cfg, device_index, check_rows=262_144, rows=None, deadline=None):
    """One SHORT run of another BASELINE config on the same GPU, behind the default command's timed region (`other_configs` on the line): the
    config's model and full row count, a few steps, HIP-event kernel times from the library, and a prefix of the result against the oracle.
    `deadline` (time.perf_counter() value): checked between the leg's phases -- model synthesis, warm-up, the timed steps, the oracle --
    a leg that runs past it returns what it has instead of costing the line; the row count shrinks to what the device has free."""
    import numpy as np
    import torch

    import ddt
    from oracle import oracle as O

    t_begin = time.perf_counter()
    T, D, F, N = CONFIG_SHAPES[cfg]
    N = rows or N   # (tests: the plumbing on a few thousand rows)
    sparse, classes = cfg == 4, (10 if cfg == 5 else 1)
    late = lambda: deadline is not None and time.perf_counter() > deadline
    try:  # tuples + scores + the rank workspace of this leg must fit what the device has free (config 1: 13.6 GB)
        free_b = torch.cuda.mem_get_info(device_index)[0]
        per_row = 4 * ((F + 3) // 4 * 4) * 2 + 16 * (classes + 1)
        if N * per_row > 0.8 * free_b:
            N = max(1024, int(0.8 * free_b / per_row) // 1024 * 1024)
    except Exception:
        pass
    steps, warm = {1: (30, 5), 2: (30, 5), 3: (3, 1), 4: (3, 1), 5: (5, 2), 6: (3, 1)}[cfg]
    eng = ddt.Engine(device_index)
    try:
        if sparse:
            lines, first = ddt.synth_sparse_model(T, D, F, 10, 700, 0)
            eng.load_model_sparse(ddt.make_sparse_params(T, D, F), lines, first)
        elif classes > 1:
            w, f = ddt.synth_model(T, D, F, 0)
            eng.load_model_multiclass(ddt.make_params(T, D, F, clusters=ddt.default_clusters(T // classes)), w, f, classes, True)
        else:
            w, f = ddt.synth_model(T, D, F, 0)
            eng.load_model(ddt.make_params(T, D, F), w, f)
        info = eng.info()
        if late():
            return {"skipped": "past the budget after the model load", "kernel": info.variant_name.decode()}
        tuples = eng.synth_tuples_device(0, N, F, 0)
        out = torch.empty(N, dtype=torch.float32, device=tuples.device)
        labels = cls = None
        if classes > 1:
            labels = torch.empty(N, dtype=torch.int32, device=tuples.device)
            cls = torch.empty((classes, N), dtype=torch.float32, device=tuples.device)
        else:
            eng.set_option("kernel_timing", 1)

        def step():
            if classes > 1:
                eng.classify_device(tuples, class_scores=cls, labels=labels)
            else:
                eng.score_device(tuples, out=out)

        for _ in range(warm):
            step()
        torch.cuda.synchronize()
        if late():
            return {"skipped": "past the budget after the warm-up", "kernel": info.variant_name.decode()}
        st0 = eng.stats()
        if cfg in (1, 2):   # steps of 1-3 ms: one timed batch
            t0 = time.perf_counter()
            for _ in range(steps):
                step()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / steps * 1e3
        else:               # steps of 10-100 ms: each step timed by itself, the MEDIAN reported (one host hiccup -- the CPU leg's threads winding down -- cost a
            per = []        # 3-step mean 3 ms per step on one box: gpurun_out r06_s8 against r06_s9)
            for _ in range(steps):
                t0 = time.perf_counter()
                step()
                torch.cuda.synchronize()
                per.append((time.perf_counter() - t0) * 1e3)
            ms = sorted(per)[len(per) // 2]
        st1 = eng.stats()
        k = st1.timed_launches - st0.timed_launches
        k_ms = (st1.sum_score_ms - st0.sum_score_ms) / k if k > 0 else ms
        if k_ms <= 0:   # (events that could not be resolved: the step's wall time instead of a division by zero)
            k_ms = ms
        pre_ms = (st1.sum_prepass_ms - st0.sum_prepass_ms) / k if k > 0 else 0.0
        alg = N * (4 * F + 4 * (classes + 1 if classes > 1 else 1)) + int(info.model_bytes_unpadded)
        rows = min(N, check_rows if not sparse else min(check_rows, 32_768))
        if late():
            rows = min(rows, 4096)   # (the value is measured: keep it, with a token check)
        xs = tuples[:rows].cpu().numpy().view(np.uint32)
        if sparse:
            ref = O.score_sparse_fast(O.SparseModel(O.make_sparse_params(T, D, F), lines, first), xs)
            same = bool(np.array_equal(out[:rows].cpu().numpy().view(np.uint32), ref.view(np.uint32)))
        elif classes > 1:
            ref_l, ref_cs = O.classify_fast(O.Model(O.make_params(T, D, F, clusters=ddt.default_clusters(T // classes)), w, f), xs, classes, True)
            same = bool(np.array_equal(labels[:rows].cpu().numpy(), ref_l) and
                        np.array_equal(cls[:, :rows].cpu().numpy().view(np.uint32), ref_cs.view(np.uint32)))
        else:
            ref = O.score_fast(O.Model(O.make_params(T, D, F), w, f), xs)
            same = bool(np.array_equal(out[:rows].cpu().numpy().view(np.uint32), ref.view(np.uint32)))
        return {"workload": f"{T} {'sparse ' if sparse else ''}trees x depth {'<= ' if sparse else ''}{D} x {F} features"
                            + (f", {classes} classes" if classes > 1 else "") + f", {N} tuples/step",
                "value": round(N / ms / 1e3, 3), "unit": "Mtuples/s", "ms_per_step": round(ms, 4), "steps": steps, "warmup": warm,
                "kernel": info.variant_name.decode(), "fallback_kernel": bool(info.fallback_kernel),
                "roofline": {"bound": "hbm", "kernel_ms": round(k_ms, 4), "prepass_ms": round(pre_ms, 4), "alg_bytes_per_launch": alg,
                             "achieved": round(alg / (k_ms * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(alg / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                             "step_frac": round(alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)},
                "parity": {"rows_checked": rows, "bit_exact": same, "what": "int32 labels + fp32 class sums" if classes > 1 else "fp32 scores"},
                "seconds": round(time.perf_counter() - t_begin, 2)}
    finally:
        eng.close()


def main(argv=None, inproc_env=None):
    """inproc_env (tests only): a dict standing in for the launcher's environment (RANK / LOCAL_RANK / WORLD_SIZE) -- several ranks then run
    as THREADS of one process (tests/test_bench_multirank_mock.py: the whole N > 1 orchestration below against the CPU model of the host side),
    stdout is left alone, no watchdog is armed, and rank 0 RETURNS the line instead of printing it."""
    env = os.environ if inproc_env is None else inproc_env
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default 5; configs 1 and 2, whose step is 1-3 ms: 30)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed steps before them (default 2; configs 1 and 2: 5)")
    ap.add_argument("--config", type=int, default=3, choices=[1, 2, 3, 4, 5, 6],
                    help="BASELINE.json config: 3 = headline (default); 1 = 8 trees x d4 x 16 features (HBM-bound shape, 200 M rows "
                         "instead of the 1 K rows of the CPU plumbing case); 2 = 100 x d6 x 28, 10 M rows; 4 = sparse random forest; "
                         "5 = 10-class one-vs-all, 100 trees/class, d8, 32 features, 10 M rows (value = tuples classified); "
                         "6 = the REFERENCE's own example configuration (profiler/profiler.cpp:32-38): 512 trees x depth 12 x 32 features, perfect format, 10 M rows")
    ap.add_argument("--rows", type=int, default=0, help="tuples per step (default: the config's)")
    ap.add_argument("--trees", type=int, default=0)
    ap.add_argument("--levels", type=int, default=0)
    ap.add_argument("--features", type=int, default=0)
    ap.add_argument("--full-levels", type=int, default=10, help="config 4: levels grown completely")
    ap.add_argument("--permille", type=int, default=700, help="config 4: split probability below the full levels, in 1/1000")
    ap.add_argument("--combine", default="allreduce", choices=["allreduce", "chain"])
    ap.add_argument("--shard", default="trees", choices=["trees", "rows", "hybrid"],
                    help="N>1: 'trees' = the headline mode (ensemble sharded tree-wise, partial scores all-reduced); "
                         "'rows' = the reference's other mode (replicated ensemble, tuples partitioned, every step of scores handed to all peers while the next is scored); "
                         "'hybrid' = the two composed (ddt_comm_create_hybrid): row groups of --tree-ranks consecutive ranks, each a tree-sharded job on its slice of the rows")
    ap.add_argument("--tree-ranks", type=int, default=2, help="--shard hybrid: ranks per row group (= tree shards); must divide N")
    ap.add_argument("--no-gather", action="store_true", help="--shard hybrid: scores stay with their row group (no hand-over to the other row groups)")
    ap.add_argument("--shard-of", type=int, default=0,
                    help="N=1 only: load shard 3 (or the last) of a G-way tree-sharded job of the configured model and score it with no "
                         "collective -- exactly what one of G ranks computes (same cluster count, same kernel choice); no CPU / host-feeder legs")
    ap.add_argument("--chunk-rows", type=int, default=12_500_000, help="rows per pipelined collective (N>1)")
    ap.add_argument("--taper", type=int, default=-1, choices=[-1, 0, 1],
                    help="N>1 / --force-collectives: cut the last chunk into 1/2, 1/4, 1/4 so that the exposed collective is a quarter "
                         "chunk (-1 = the library's default: on when the communicator has more than one rank)")
    ap.add_argument("--collectives", default="cabi", choices=["cabi", "torch"],
                    help="cabi = RCCL called from C++ behind the C-ABI (the product path); torch = the same pipeline "
                         "driven from Python through torch.distributed (needed for --backend gloo)")
    ap.add_argument("--variant", type=int, default=-1, help="kernel variant id (-1 = engine's choice)")
    ap.add_argument("--sum-mode", type=int, default=0)
    ap.add_argument("--opt", action="append", default=[], metavar="KEY=VALUE",
                    help="engine option set before the model is loaded (ddt_set_option; A/B switches such as q16_persistent=1, q16_prepass_nt=1); repeatable")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend used for the barrier / timing reduction (and for --collectives torch); "
                         "gloo lets N ranks share one GPU for a functional test")
    ap.add_argument("--force-collectives", action="store_true",
                    help="N=1 only: run the multi-GPU chunk pipeline and the collectives in a one-rank group (overhead / sanity run)")
    ap.add_argument("--no-other-modes", action="store_true",
                    help="N>1 / --force-collectives: skip the extra legs behind the timed region (chain combine, untapered all-reduce, "
                         "row-sharded replicas) that land in `other_modes` on the line")
    ap.add_argument("--other-modes-timeout", type=float, default=120.0,
                    help="seconds after which a stuck extra leg is abandoned: the headline line is printed and the process exits")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-streamed", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time of the baseline sample")
    ap.add_argument("--check-rows", type=int, default=4_194_304,
                    help="rows of the timed job's result that are checked against the oracle where the check is a leg of its own (N>1 tolerance check, sum_mode 2)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="N=1 default workload only: skip the short runs of the other BASELINE configs behind the timed region (`other_configs` on the line)")
    ap.add_argument("--other-configs-budget", type=float, default=40.0,
                    help="seconds the `other_configs` legs may take in all: a config that would start past the budget is skipped and says so")
    args = ap.parse_args(argv)
    # a step of configs 1 / 2 takes 1-3 ms: five of them time the first launches after the allocations (~4 % slower) more than the kernel
    short = args.config in (1, 2)
    if args.steps is None:
        args.steps = 30 if short else 5
    if args.warmup is None:
        args.warmup = 5 if short else 2

    # `python bench.py --gpus N` without a launcher around it: become the launcher (one rank per GPU through torch.distributed.run on
    # 127.0.0.1, a free port), pass the ranks' stdout through -- rank 0 prints the ONE JSON line -- and exit with their status
    if args.gpus > 1 and "WORLD_SIZE" not in env:
        sys.exit(self_launch(args.gpus))

    # stdout discipline: the driver wants ONE JSON line.  Native libraries (RCCL prints a version banner through C stdio)
    # must not add lines to it: everything but the final print goes to stderr.
    saved_stdout = None
    if inproc_env is None:
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)

    import numpy as np
    import torch
    import torch.distributed as dist

    import ddt

    sparse = args.config == 4
    classes = 10 if args.config == 5 else 1
    shape = CONFIG_SHAPES[args.config]
    T, D, F, N = (args.trees or shape[0], args.levels or shape[1], args.features or shape[2], args.rows or shape[3])

    world = int(env.get("WORLD_SIZE", "1"))
    rank = int(env.get("RANK", "0"))
    local = int(env.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        sys.exit(f"--gpus {args.gpus} != WORLD_SIZE {world}")
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (the scoring path has no CPU fallback)")
    if args.backend == "gloo" and args.collectives == "cabi" and world > 1:
        args.collectives = "torch"  # several ranks on one GPU cannot form an RCCL communicator
    local = local % torch.cuda.device_count() if args.backend == "gloo" else local
    torch.cuda.set_device(local)
    multi = world > 1 or args.force_collectives
    if world == 1 and args.force_collectives:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    if multi:
        if args.backend == "nccl":
            try:
                dist.init_process_group("nccl", device_id=torch.device("cuda", local))
            except TypeError:  # older torch: no device_id keyword
                dist.init_process_group("nccl")
        else:
            dist.init_process_group("gloo")

    W = ddt.tuple_words(F)
    eng = ddt.Engine(local)
    eng.set_option("variant", args.variant)
    for kv in args.opt:
        key, _, val = kv.partition("=")
        eng.set_option(key, int(val))
    rows_mode = world > 1 and args.shard == "rows"
    hybrid_mode = multi and args.shard == "hybrid"
    if hybrid_mode and (world % args.tree_ranks or args.collectives != "cabi"):
        sys.exit("--shard hybrid: --tree-ranks must divide the number of ranks; C-ABI collectives only")
    shard = (0, 1) if rows_mode else (rank % args.tree_ranks, args.tree_ranks) if hybrid_mode else (rank, world)
    if args.shard_of > 1:
        if world > 1 or args.force_collectives or sparse or classes > 1:
            sys.exit("--shard-of: one plain engine on one GPU")
        shard = (min(3, args.shard_of - 1), args.shard_of)
        args.no_cpu_baseline = args.no_streamed = True   # the CPU legs below score the whole ensemble
    if sparse:
        lines, first = ddt.synth_sparse_model(T, D, F, args.full_levels, args.permille, 0)
        params = ddt.make_sparse_params(T, D, F, sum_mode=args.sum_mode)
        eng.load_model_sparse(params, lines, first, *shard)
    elif classes > 1:
        w, f = ddt.synth_model(T, D, F, 0)
        params = ddt.make_params(T, D, F, clusters=ddt.default_clusters(T // classes), sum_mode=args.sum_mode)
        eng.load_model_multiclass(params, w, f, classes, True, *shard)  # tree i belongs to class i % 10 (XGBoost multi:softprob order)
    else:
        w, f = ddt.synth_model(T, D, F, 0)
        params = ddt.make_params(T, D, F, sum_mode=args.sum_mode)
        eng.load_model(params, w, f, *shard)
    info = eng.info()

    tuples = eng.synth_tuples_device(0, N, F, 0)          # resident in HBM before the timed region
    out = torch.empty(N, dtype=torch.float32, device=tuples.device)
    labels = cls_scores = None
    if classes > 1:
        labels = torch.empty(N, dtype=torch.int32, device=tuples.device)
        cls_scores = torch.empty((classes, N), dtype=torch.float32, device=tuples.device)
        if rows_mode or args.collectives != "cabi":
            sys.exit("config 5 across GPUs: tree-sharded through the C-ABI collectives only")
    comm = scorer = None
    combine = ddt.COMBINE_CHAIN if args.combine == "chain" else ddt.COMBINE_ALLREDUCE
    if multi and args.collectives == "cabi":
        # the communicator id travels over the launcher's channel (torch.distributed's store); everything after that
        # -- ncclCommInitRank, the chunk pipeline, every collective -- happens in C++ behind the C-ABI
        box = [ddt.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        comm = ddt.Comm(eng, rank, world, box[0], tree_ranks=args.tree_ranks if hybrid_mode else 0)
        info = eng.info()   # (inside a multi-rank job the engine may have switched to the persistent kernel: csrc/ddt_engine.cpp engine_enter_collective_job)
        comm.set_option("chunk_rows", max(1024, args.chunk_rows // (world // args.tree_ranks)) if hybrid_mode else args.chunk_rows)
        comm.set_option("taper_tail", args.taper)
    elif multi:
        # --collectives torch: the Python mirror of the pipeline (test infrastructure; needed for gloo ranks sharing a GPU), loaded by path
        # so that no installed package called `tests` can shadow it
        import importlib.util

        spec = importlib.util.spec_from_file_location("ddt_sharded_ref", os.path.join(ROOT, "tests", "sharded_ref.py"))
        sharded_ref = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(sharded_ref)

        scorer = (sharded_ref.RowShardedScorer(eng) if rows_mode else
                  sharded_ref.ShardedScorer.from_engine(eng, mode=args.combine, chunk_rows=args.chunk_rows,
                                                force_collectives=args.force_collectives))

    # per-launch HIP-event times of the pass, taken by the library on the launch stream ("kernel_timing"):
    # pre-pass kernels (rank-quantised path only) and the scoring kernel proper
    kernel_ms = []
    launches_per_step = 1.0
    if not multi and classes == 1:  # (the per-launch events would pin the classes of config 5 to one stream)
        eng.set_option("kernel_timing", 1)

    def step(record: bool):
        if classes > 1:
            if comm is not None:
                comm.classify_sharded(tuples, combine=combine, class_scores=cls_scores, labels=labels)
            else:
                eng.classify_device(tuples, class_scores=cls_scores, labels=labels)
        elif comm is not None:
            if hybrid_mode:
                comm.score_hybrid(tuples, out=out, combine=combine, gather=not args.no_gather)
            elif rows_mode:
                comm.score_rowsharded(tuples, out=out)
            else:
                comm.score_sharded(tuples, out=out, combine=combine)
        elif scorer is not None:
            scorer.score(tuples, out=out)
        else:
            eng.score_device(tuples, out=out)

    def fence():
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(False)
    fence()
    timed = comm is None and scorer is None and classes == 1   # the library times every launch with events on the launch stream
    st0 = eng.stats() if timed else None                       # (folds the warm-up launches' events into its sums)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)                                             # no host synchronisation between the steps
    fence()
    dt = time.perf_counter() - t0
    if timed:
        st1 = eng.stats()
        k = st1.timed_launches - st0.timed_launches
        if k > 0:                                              # the timed region's launches, averaged (the library keeps 64 launches' events)
            kernel_ms.append(((st1.sum_prepass_ms - st0.sum_prepass_ms) / k, (st1.sum_score_ms - st0.sum_score_ms) / k))
        launches_per_step = (st1.kernel_launches - st0.kernel_launches) / max(1, args.steps)
    elif comm is None and scorer is None:
        kernel_ms.append((0.0, 0.0))                           # classes: the roofline below takes the whole step (K scoring launches + argmax)
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=tuples.device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    ms_per_step = dt / max(1, args.steps) * 1e3
    mtuples = N / (dt / max(1, args.steps)) / 1e6
    # the combined result of the TIMED job, kept on rank 0 for the parity leg below: the passes behind the timed region (scaling_detail,
    # other_modes) overwrite `out`
    timed_result = None
    if multi and rank == 0 and not args.no_cpu_baseline:
        timed_result = (labels.clone(), cls_scores.clone()) if classes > 1 else out.clone()

    # ---- N>1 (or --force-collectives): the same shard scored WITHOUT the collectives, so that the line itself shows what the
    # combine costs on top of the per-rank compute (max over ranks, 2 steps, outside the timed region) -----------------------
    scaling_detail = None
    if comm is not None and classes == 1 and not rows_mode and not hybrid_mode:
        try:
            fence()
            t1 = time.perf_counter()
            for _ in range(2):
                eng.score_device(tuples, out=out)
            fence()
            cdt = (time.perf_counter() - t1) / 2
            if world > 1:
                tmax = torch.tensor([cdt], dtype=torch.float64, device=tuples.device)
                dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
                cdt = float(tmax.item())
            scaling_detail = {"shard_compute_only_ms": round(cdt * 1e3, 4), "combine_overhead_ms": round(ms_per_step - cdt * 1e3, 4),
                              "trees_on_this_rank": int(info.tree_end - info.tree_begin), "chunk_rows": args.chunk_rows,
                              "note": "ms_per_step minus one pass of the rank's shard over all tuples with no collective (max over ranks)"}
            # rank 0's kernels in such a pass, timed by the library with HIP events on the launch stream: the `roofline` object of an
            # N>1 line (the dominant kernel = this rank's shard of the scoring kernel; one more pass, outside the timed region)
            eng.set_option("kernel_timing", 1)
            eng.score_device(tuples, out=out)
            st = eng.stats()
            kernel_ms.append((st.last_prepass_ms, st.last_score_ms))
            eng.set_option("kernel_timing", 0)
            fence()
        except Exception as ex:  # diagnostics must never cost the headline line
            scaling_detail = {"error": repr(ex)}

    # ---- roofline of the dominant kernel (the per-shard scoring kernel) ----------------------------
    # tuples in, scores out, model once; config 5 writes the K per-class sums and the label (argmax not fused)
    alg_bytes_per_launch = N * (4 * F + 4 * (classes + 1 if classes > 1 else 1)) + int(info.model_bytes_unpadded)
    roofline = None
    if kernel_ms:
        pre_ms = sum(a for a, _ in kernel_ms) / len(kernel_ms)
        k_ms = sum(b for _, b in kernel_ms) / len(kernel_ms)  # the dominant (scoring) kernel
        if classes > 1:  # the library times the LAST class's launch: the K launches are alike, the pre-pass ran once with the first
            k_ms, pre_ms = ms_per_step, 0.0
    if kernel_ms and k_ms > 0:  # (an event that could not be resolved leaves 0: no roofline object rather than a division by zero)
        ach = alg_bytes_per_launch / (k_ms * 1e-3) / 1e9
        traffic, traffic_source = None, None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic_cfg4.json" if sparse else "pmc_traffic_cfg1.json" if args.config == 1 else
                           "pmc_traffic_cfg6.json" if args.config == 6 else "pmc_traffic.json")
        if os.path.exists(pmc) and not multi and args.shard_of <= 1:  # HBM bytes per launch of this kernel: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE of THIS command, collected in its own run
            try:
                pj = json.load(open(pmc))
                if pj.get("rows") == N and pj.get("trees") == T:
                    traffic = pj.get("hbm_bytes_per_launch")
                    traffic_source = f"profiles/{os.path.basename(pmc)}: rocprofv3 --pmc of this command (separate run, gfx950 FETCH_SIZE x2 correction applied); not measured in this run"
            except Exception:
                traffic = None
        clock_hz = info.clock_khz * 1e3
        roofline = {"bound": "hbm", "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_source,
                    "kernel": info.variant_name.decode(), "kernel_ms": round(k_ms, 4),
                    "prepass_ms": round(pre_ms, 4),  # rank pre-pass of the rank-quantised path (0 otherwise)
                    "prepass_groups": info.prepass_groups,  # feature groups of the LDS-resident pre-pass (0: transpose + rank kernels / n.a.)
                    "alg_bytes_per_launch": alg_bytes_per_launch,
                    # an ensemble with more than 38848 distinct thresholds on a feature is scored in PARTS (one rank pre-pass + one scoring launch
                    # each, the reference-order sum handed on): kernel_ms is then everything behind the FIRST part's pre-pass
                    "scoring_launches_per_step": round(launches_per_step, 2),
                    "device": {"cus": info.num_cus, "clock_mhz": round(clock_hz / 1e6, 1), "lds_bytes_per_cu": info.lds_bytes_per_cu}}
        # the whole step (pre-pass + scoring kernel(s)) against the same algorithmic bytes; traffic = the step's HBM bytes from the same PMC file
        step_traffic = None
        if traffic is not None:
            try:
                step_traffic = json.load(open(pmc)).get("step_hbm_bytes_x2_corrected")
            except Exception:
                step_traffic = None
        if not multi and args.shard_of <= 1:
            step_ach = alg_bytes_per_launch / (ms_per_step * 1e-3) / 1e9
            roofline["step"] = {"ms": round(ms_per_step, 4), "achieved": round(step_ach, 2), "frac": round(step_ach / HBM_PEAK_GBS, 5), "traffic": step_traffic,
                                "note": "one whole step of the timed region (rank pre-pass + scoring; wall time / steps) against the same algorithmic bytes"}
        if multi:
            roofline["scope"] = (f"rank 0's shard ({int(info.tree_end - info.tree_begin)} of {T} trees, all {N} tuples): one pass with no collective, "
                                 "behind the timed region; achieved / frac are per GPU")
        if sparse:
            # leaf depth of the model = node visits per tuple and tree: measured on a sample with the oracle below
            roofline["binding_resource"] = (
                "sparse_r_*: two levels per 16-byte gather below the top image; at 4 gathers per tree and wave the kernel is latency-bound at two waves per "
                "SIMD (LDS: 80 KiB per block of four waves; VALU issue and vector-memory pipe each ~55-60 % busy: profiles/r06_sparse_r32.md), not on HBM"
                if info.variant_name.decode().startswith("sparse_r_") else
                "vector-memory gathers of the deep phase: one 16-byte load per lane and visit (the VMEM address pipe takes about one lane per cycle and CU), not HBM")
            roofline["vmem_ceiling_gathers_per_s"] = info.num_cus * clock_hz
        else:
            t_local = info.tree_end - info.tree_begin
            if ach / HBM_PEAK_GBS > 0.3:
                roofline["binding_resource"] = "HBM: the tuple stream (each tuple read once, each score written once)"
            else:
                roofline["binding_resource"] = "LDS gather pipe (2 DS ops per node visit) + VALU issue, not HBM"
            roofline["node_visits_per_s"] = round(N * t_local * D / (k_ms * 1e-3), 1)
            # 2 conflict-free DS wave-instructions per 64 visits at 2 LDS cycles each (MI355X_MICROARCH.md, LDS table)
            roofline["lds_ceiling_visits_per_s"] = info.num_cus * clock_hz * 64 / 4
            # VALU issue bound of the lane = tuple mapping: a wave instruction takes ~4 cycles on its SIMD (16 lanes per cycle; tools/ubench valu,
            # profiles/archive/r03_ubench_valu.json) and the shipped depth-8 walk spends 4.36 of them per node visit (profiles/archive/r03_pmc_q16_gl_s2.md)
            roofline["valu_issue_bound_visits_per_s"] = round(info.num_cus * 4 * clock_hz / (4.0 * 4.36) * 64, 1)
            # the BARE depth-8 walk of round 3 (no DMA, no barriers, model resident in LDS; hipcc's own read order: two chains in flight per lane),
            # measured on an MI355X and committed -- round 4's pinned read order (four chains in flight) runs the PRODUCT kernel above it
            try:
                ub = json.load(open(os.path.join(ROOT, "profiles", "r03_ubench_walk.json")))
                best = max(w["T_visits_per_s"] for w in ub["walk"] if w["bit_exact_vs_cpu"])
                roofline["r03_bare_walk_compiler_order_visits_per_s"] = best * 1e12
            except Exception:
                pass

    # ---- CPU baseline + parity (oracle = a CPU restatement of the reference RTL semantics), on rank 0.  N > 1 (and --force-collectives): the
    # combined result of the TIMED job -- kept above -- is checked against the oracle's model of the job as it ran (the chain of
    # ResultsCombiner.sv:292-311,359-369 over n_dev contiguous shards, PCIeReceiver.sv:241-264; rows mode / one rank: one device), and the
    # other ranks wait behind a barrier while rank 0 times the oracle, so the sample has the box's cores to itself.
    cpu = parity = streamed = None
    sum_ref = None
    if rank == 0 and not args.no_cpu_baseline and (not multi or timed_result is not None):
        from oracle import oracle as O

        sum_ref = {0: O.SUM_REF_NATIVE, 1: O.SUM_F64_SEQ, 2: O.SUM_REF_FLOPOCO}[args.sum_mode]
        n_dev = 1 if (not multi or rows_mode) else (args.tree_ranks if hybrid_mode else world)
        valid_rows = N
        if hybrid_mode and args.no_gather:                       # rank 0 holds the rows of its own row group only
            valid_rows = ddt.hybrid_rows(N, world // args.tree_ranks, 0)[1]
        if sparse:
            m = O.SparseModel(O.make_sparse_params(T, D, F), lines, first)
            if n_dev == 1:
                def cpu_score(xs):
                    return O.score_sparse_fast(m, xs, sum_mode=sum_ref)
                what = "oracle/ddt_oracle.c orc_score_sparse_fast: one tree at a time over a 1024-row block, 8 rows in flight per thread"
            else:
                def cpu_score(xs):
                    return O.score_sparse(m, xs, sum_mode=sum_ref, n_devices=n_dev)
                what = f"oracle/ddt_oracle.c orc_score_sparse, {n_dev}-device chain model"
        elif classes > 1:
            m = O.Model(O.make_params(T, D, F, clusters=ddt.default_clusters(T // classes)), w, f)

            def cpu_score(xs):
                return O.classify_fast(m, xs, classes, True, sum_mode=sum_ref, n_devices=n_dev)
            what = "oracle/ddt_oracle.c orc_score_fast_ex: per-class reference-order sums" + (f" over a {n_dev}-device chain" if n_dev > 1 else "") + ", argmax"
        else:
            m = O.Model(O.make_params(T, D, F), w, f)

            def cpu_score(xs):
                return O.score_fast(m, xs, sum_mode=sum_ref, n_devices=n_dev)
            what = ("oracle/ddt_oracle.c orc_score_fast: cache-blocked 8-byte nodes, 8 walks in flight per thread" if n_dev == 1 and args.sum_mode != 2 else
                    f"oracle/ddt_oracle.c orc_score_fast_ex: the same walk, {n_dev}-device chain model" + (", the reference's adder" if args.sum_mode == 2 else ""))
        cpu_score(tuples[: min(N, 4096)].cpu().numpy().view(np.uint32))  # thread-pool warm-up
        probe = min(valid_rows, 262_144)
        xs = tuples[:probe].cpu().numpy().view(np.uint32)
        t1 = time.perf_counter()
        cpu_score(xs)
        rate = probe / max(1e-9, time.perf_counter() - t1)
        rows = int(max(probe, min(valid_rows, 64_000_000, rate * args.cpu_seconds)))
        xs = tuples[:rows].cpu().numpy().view(np.uint32)
        t1 = time.perf_counter()
        ref = cpu_score(xs)
        cdt = time.perf_counter() - t1
        cpu = {"value": round(rows / cdt / 1e6, 4), "unit": "Mtuples/s", "cores": O.hw_threads(), "kind": "port",
               "sample": f"first {rows} rows of the same synthetic batch, all {T} trees, OpenMP over row blocks on {O.hw_threads()} threads "
                         f"(= the CPUs this process may use: affinity mask and cgroup quota, of {os.cpu_count()} logical CPUs on the box), "
                         f"{cdt:.1f} s ({what}; a CPU restatement of the reference RTL semantics, the reference has no CPU scorer)"
                         + ("; the other ranks wait behind a barrier meanwhile" if world > 1 else "")}
        res = timed_result if timed_result is not None else ((labels, cls_scores) if classes > 1 else out)
        if classes > 1:
            got_l, got_cs = res[0][:rows].cpu().numpy(), res[1][:, :rows].cpu().numpy()
            ref_l, ref_cs = ref
            same_l = bool(np.array_equal(got_l, ref_l))
            same_cs = bool(np.array_equal(got_cs.view(np.uint32), ref_cs.view(np.uint32)))
            parity = {"rows_checked": rows, "bit_exact": same_l and (same_cs or not multi), "what": "int32 class labels" + (" and fp32 class sums" if multi else ""),
                      "labels_that_differ": int((got_l != ref_l).sum()), "class_sums_bit_exact": same_cs}
        else:
            got = res[:rows].cpu().numpy()
            parity = {"rows_checked": rows, "bit_exact": bool(np.array_equal(got.view(np.uint32), ref.view(np.uint32))), "what": "fp32 scores"}
        if multi:
            # the job as it ran: which combine, against which model.  The chain IS the reference's order (bit-exact required); RCCL's
            # all-reduce adds the partials in its own order: north_star's 1e-6 is shown against the fp64 sum of the leaves, and the rows
            # that differ from the chain's result are counted
            parity["combine"] = "none (replicas)" if rows_mode else args.combine
            parity["oracle"] = f"{n_dev}-device chain of the reference (ResultsCombiner.sv:292-311), reference-order sums per device"
            chk = min(rows, args.check_rows)
            if classes > 1:
                _, cs_ref, gold, gabs = O.classify_fast(m, xs[:chk], classes, True, sum_mode=sum_ref, n_devices=n_dev, want_gold=True)
                err = np.abs(res[1][:, :chk].cpu().numpy().astype(np.float64) - gold)
                parity["rows_that_differ_from_chain_oracle"] = int((got_cs.view(np.uint32) != ref_cs.view(np.uint32)).any(axis=0).sum())
            elif not sparse:
                _, gold, gabs = O.score_fast(m, xs[:chk], sum_mode=sum_ref if args.sum_mode != 1 else O.SUM_REF_NATIVE, n_devices=n_dev, want_gold=True)
                err = np.abs(got[:chk].astype(np.float64) - gold)
                parity["rows_that_differ_from_chain_oracle"] = int((got.view(np.uint32) != ref.view(np.uint32)).sum())
            else:
                chk = min(chk, 65_536)
                _, gold = O.score_sparse(m, xs[:chk], sum_mode=sum_ref, n_devices=n_dev, want_gold=True)
                # the sum of the leaves' magnitudes: the same forest with |leaf| in every leaf field (entry bits 14 / 15 flag them), same walks
                la = np.array(lines, np.uint32, copy=True).reshape(-1, 4)
                for side in (0, 1):
                    is_leaf = ((la[:, 1] >> (14 + side)) & 1) != 0
                    la[is_leaf, 2 + side] &= 0x7FFFFFFF
                _, gabs = O.score_sparse(O.SparseModel(O.make_sparse_params(T, D, F), la, first), xs[:chk], sum_mode=sum_ref, n_devices=n_dev, want_gold=True)
                err = np.abs(got[:chk].astype(np.float64) - gold)
                parity["rows_that_differ_from_chain_oracle"] = int((got.view(np.uint32) != ref.view(np.uint32)).sum())
            tol = 1e-6 * np.maximum(np.abs(gold), gabs)
            parity["within_tolerance"] = bool((err <= tol).all())
            parity["tolerance"] = "|score - fp64 sum of the row's leaves| <= 1e-6 * max(|that sum|, sum of the leaves' magnitudes)"
            parity["tolerance_rows_checked"] = int(chk)
            parity["max_err_over_tolerance"] = float((err / np.maximum(tol, 1e-300)).max()) if err.size else 0.0
            if args.combine == "chain" or rows_mode or n_dev == 1:
                parity["required"] = "bit_exact"
            else:
                parity["required"] = "within_tolerance"
        if sparse and roofline is not None:
            depth = O.sparse_mean_depth(m, xs[:2048])  # node visits per (tuple, tree) on a sample
            k_s = roofline["kernel_ms"] * 1e-3
            roofline["model"] = {"internal_nodes": int(lines.shape[0]), "nodes_per_tree": round(lines.shape[0] / T, 1),
                                 "mean_visits_per_tuple_and_tree": round(depth, 3)}
            roofline["node_visits_per_s"] = round(N * T * depth / k_s, 1)
            roofline["deep_gathers_per_s"] = round(N * T * max(0.0, depth - int(info.variant_name.decode().split("_k")[1].split("_")[0])) / k_s, 1)
    if world > 1 and not args.no_cpu_baseline:
        fence()   # the other ranks wait here while rank 0 runs the CPU leg

    # ---- PCIe-inclusive "streamed" mode (SURVEY 8(d) timing protocol): host buffers through the pinned feeder ----
    if world == 1 and rank == 0 and not args.no_streamed and not multi and classes == 1:
        srows = min(N, 16_000_000)
        host = tuples[:srows].cpu().numpy().view(np.uint32)  # pageable host memory, as a caller would hold it
        eng.score(host[: min(srows, 4 << 20)])                # feeder buffers allocated and touched (all three slots), staging threads started
        sdt = 1e30
        for _ in range(2):  # the second pass over the same pageable buffer is the steady state (first: page / TLB warm-up of 2 GB)
            t1 = time.perf_counter()
            hs = eng.score(host)
            sdt = min(sdt, time.perf_counter() - t1)
        want_bits = out[:srows].cpu().numpy().view(np.uint32)
        streamed = {"value": round(srows / sdt / 1e6, 2), "unit": "Mtuples/s", "rows": srows,
                    "link_GBs": round(srows * (4 * W + 4) / sdt / 1e9, 2),
                    "note": "pageable host tuples -> staging threads -> pinned buffers (3 slots) -> hipMemcpyAsync H2D -> kernels -> D2H, PCIe-inclusive; never `value`",
                    "bit_exact_vs_resident": bool(np.array_equal(hs.view(np.uint32), want_bits))}
        # the same job on buffers the caller pinned once (ddt_host_register): the DMA engine reads / writes them directly
        hs2 = np.empty(srows, np.float32)
        t1 = time.perf_counter()
        eng.host_register(host)
        eng.host_register(hs2)
        reg_s = time.perf_counter() - t1
        eng.score(host[: min(srows, 4 << 20)], out=hs2[: min(srows, 4 << 20)])
        t1 = time.perf_counter()
        eng.score(host, out=hs2)
        pdt = time.perf_counter() - t1
        eng.host_unregister(host)
        eng.host_unregister(hs2)
        streamed["pinned"] = {"value": round(srows / pdt / 1e6, 2), "link_GBs": round(srows * (4 * W + 4) / pdt / 1e9, 2),
                              "register_ms": round(reg_s * 1e3, 1),
                              "note": "caller's buffers pinned once with ddt_host_register (time given, outside the rate): no staging copy",
                              "bit_exact_vs_resident": bool(np.array_equal(hs2.view(np.uint32), want_bits))}

    # ---- the RTL-exact sums (sum_mode 2: the reference's FloPoCo adder itself, FPAdder_2cycles_latency.v:325-326) on the same batch, measured at
    # HEAD behind the timed region: the strongest parity claim the library makes gets its own throughput figure on the line ----------------------
    sum2 = None
    if world == 1 and rank == 0 and not multi and classes == 1 and not sparse and args.sum_mode == 0 and args.shard_of <= 1 and not args.no_cpu_baseline:
        try:
            eng2 = ddt.Engine(local)
            eng2.set_option("variant", args.variant)
            for kv in args.opt:
                key, _, val = kv.partition("=")
                eng2.set_option(key, int(val))
            eng2.load_model(ddt.make_params(T, D, F, sum_mode=2), w, f)
            out2 = torch.empty(N, dtype=torch.float32, device=tuples.device)
            eng2.score_device(tuples, out=out2)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(3):
                eng2.score_device(tuples, out=out2)
            torch.cuda.synchronize()
            s2_ms = (time.perf_counter() - t1) / 3 * 1e3
            chk = min(N, args.check_rows)   # (the oracle's cache-blocked scorer with the reference adder: ddt_oracle.c section 8b)
            ref2 = O.score_fast(m, tuples[:chk].cpu().numpy().view(np.uint32), sum_mode=O.SUM_REF_FLOPOCO)
            got2 = out2[:chk].cpu().numpy()
            sum2 = {"value": round(N / s2_ms / 1e3, 3), "unit": "Mtuples/s", "ms_per_step": round(s2_ms, 4), "kernel": eng2.info().variant_name.decode(),
                    "bit_exact_vs_reference_adder": bool(np.array_equal(got2.view(np.uint32), ref2.view(np.uint32))), "rows_checked": chk,
                    "rows_that_differ_from_sum_mode0": int((out2[:chk] != out[:chk]).sum().item()),
                    "note": "sum_mode 2 = reference order with the reference's own adder (bit-exact with oracle SUM_REF_FLOPOCO on the checked prefix); "
                            "3 steps behind the timed region; never `value`"}
            eng2.close()
        except Exception as ex:  # diagnostics must never cost the headline line
            sum2 = {"error": repr(ex)}

    # ---- one rank's workload of the 8-GPU jobs on this one GPU, behind the timed region of the DEFAULT command (collect_rank_proxies) ----------
    proxies = small = None

    # ---- the other BASELINE configs, each as a short run behind the timed region of the DEFAULT command (N = 1, config 3, no overrides): the
    # driver's line then carries a value, the dominant kernel's roofline fraction and an oracle check for every config, not for the headline alone
    other_configs = None
    default_cmd = (args.config == 3 and not (args.rows or args.trees or args.levels or args.features) and args.variant < 0 and not args.opt
                   and args.sum_mode == 0 and args.shard_of <= 1)
    if world == 1 and rank == 0 and not multi and comm is None and default_cmd and not args.no_other_modes and not args.no_cpu_baseline:
        try:
            proxies = collect_rank_proxies(local, tuples, ms_per_step, (T, D, F), (w, f))
        except Exception as ex:
            proxies = {"error": repr(ex)}
        try:   # (before anything else writes `out`: its first rows are the timed region's result)
            eng.set_option("kernel_timing", 0)
            small = collect_small_batches(eng, tuples, torch.empty(131_072, dtype=torch.float32, device=tuples.device), out[:131_072].cpu().numpy().view(np.uint32))
        except Exception as ex:
            small = {"error": repr(ex)}
    if world == 1 and rank == 0 and not multi and comm is None and default_cmd and not args.no_other_configs and not args.no_cpu_baseline:
        # (config 1 alone allocates 13.6 GB next to the headline's 12.8 GB of tuples: the headline's buffers go first)
        # ... on a device that is short of memory.  On MI355X (288 GB) they stay: a side leg's buffers then come out of fresh memory either way, and
        # config 1 measured 81.2 vs 84.2 Gtuples/s with the headline's 19 GB handed back to the driver first (gpurun_out r06_s9: allocation placement)
        try:
            short = torch.cuda.mem_get_info(local)[0] < (48 << 30)
        except Exception:
            short = False
        if short:
            del tuples, out
            torch.cuda.empty_cache()
        other_configs = collect_other_configs(local, args.other_configs_budget)

    if rank == 0:
        par = (f"shard {shard[0]} of a {shard[1]}-way tree-sharded job ({int(info.tree_end - info.tree_begin)} trees) on one GPU, no collective" if args.shard_of > 1 else
               "single engine") if world == 1 else (
            f"row-sharded {world}x (replicas) + {'RCCL send/recv all-gather, pipelined' if comm is not None else 'gloo all-gather'}" if rows_mode else
            f"hybrid: {world // args.tree_ranks} row groups x {args.tree_ranks} tree shards, RCCL {args.combine} inside a row group, "
            f"{'scores stay with their row group' if args.no_gather else 'pieces handed to the other row groups while the next is scored'}" if hybrid_mode else
            f"tree-sharded {world}x + {'RCCL' if args.backend == 'nccl' else 'gloo (functional test)'} {args.combine}")
        shape = (f"{T} sparse trees (depth <= {D}, {lines.shape[0]} internal nodes) x {F} fp32 features" if sparse
                 else f"{classes}-class one-vs-all, {T // classes} trees/class x depth {D} x {F} fp32 features, argmax labels" if classes > 1
                 else f"{T} trees x depth {D} x {F} fp32 features")
        line = {
            "metric": "Mtuples/s scored, 1000 trees depth-8 / 32 feat" if (T, D, F, sparse, classes) == (1000, 8, 32, False, 1)
            else f"Mtuples/s {'classified' if classes > 1 else 'scored'}, {T} trees depth-{D} / {F} feat"
                 + (" (sparse random forest, BASELINE config 4)" if sparse else f" (the reference's own example configuration, profiler/profiler.cpp:32-38)" if args.config == 6 else f" (BASELINE config {args.config})" if args.config != 3 else ""),
            "value": round(mtuples, 3), "unit": "Mtuples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{shape}, {N} tuples/step, {par}",
                       "trees": T, "levels": D, "features": F, "rows": N,
                       "parallelism": f"row-shard{world}" if rows_mode else f"hybrid-tree{args.tree_ranks}-x-rows{world // args.tree_ranks}" if hybrid_mode else (
                           f"one-of-tree-shard{args.shard_of}" if args.shard_of > 1 else f"tree-shard{world}"),
                       "combine": args.combine if multi else None,
                       "tapered_tail": ((args.taper == 1 or (args.taper < 0 and world > 1)) if comm is not None else None),
                       "collectives": (("C-ABI ddt_comm (csrc/ddt_comm.cpp)" if comm is not None else "torch.distributed") if multi else None),
                       "collective_backend": (args.backend if multi else None), "kernel": info.variant_name.decode(),
                       "sum_mode": {0: "reference order, IEEE fp32 adds", 1: "fp64 accumulate", 2: "reference order, reference (FloPoCo) adder"}[args.sum_mode],
                       "device": info.device_name.decode()},
        }
        if args.shard_of > 1:
            line["metric"] += f" -- one shard of {args.shard_of} only (not the job's rate)"
        if roofline:
            line["roofline"] = roofline
        if cpu:
            line["cpu_baseline"] = cpu
        if parity:
            line["parity"] = parity
        if streamed:
            line["streamed"] = streamed
        if scaling_detail:
            line["scaling_detail"] = scaling_detail
        if sum2:
            line["other_modes"] = {"sum_mode2": sum2}
        if proxies:
            line.setdefault("other_modes", {})["per_rank_proxies"] = proxies
        if small:
            line.setdefault("other_modes", {})["small_batches"] = small
        if other_configs:
            line["other_configs"] = other_configs
        line["config"]["fallback_kernel"] = bool(info.fallback_kernel)
        if info.fallback_kernel:
            print(f"bench.py: WARNING: this model runs on the fallback kernel '{info.variant_name.decode()}' (no tuned kernel for this shape)", file=sys.stderr, flush=True)
    # ---- N>1 (or --force-collectives): the OTHER ways this library can run the same multi-GPU job, measured behind the timed
    # region so that the driver's scaling run records them too (never `value`).  These collectives have run in one-rank
    # communicators and in the CPU model of tests/test_comm_mock.py only: a watchdog keeps a stuck leg from costing the line.
    wd = None
    if comm is not None and classes == 1 and not sparse and not rows_mode and not hybrid_mode and not args.no_other_modes:
        import threading

        other = {}

        def bail():
            try:
                if rank == 0:
                    other["status"] = f"abandoned after {args.other_modes_timeout:.0f} s (an extra leg or the teardown behind it did not return)"
                    line["other_modes"] = other
                    os.dup2(saved_stdout, 1)
                    os.write(1, (json.dumps(line) + "\n").encode())
            finally:
                os._exit(0)

        if inproc_env is None:
            wd = threading.Timer(args.other_modes_timeout, bail)
            wd.daemon = True
            wd.start()
        try:
            def leg(fn, steps=2):
                fn()
                fence()
                t1 = time.perf_counter()
                for _ in range(steps):
                    fn()
                fence()
                ldt = (time.perf_counter() - t1) / steps
                if world > 1:
                    tm = torch.tensor([ldt], dtype=torch.float64, device=tuples.device)
                    dist.all_reduce(tm, op=dist.ReduceOp.MAX)
                    ldt = float(tm.item())
                return round(ldt * 1e3, 4)

            other["tree_sharded_chain_ms"] = leg(lambda: comm.score_sharded(tuples, out=out, combine=ddt.COMBINE_CHAIN))
            comm.set_option("taper_tail", 0)
            other["tree_sharded_allreduce_untapered_ms"] = leg(lambda: comm.score_sharded(tuples, out=out, combine=ddt.COMBINE_ALLREDUCE))
            comm.set_option("taper_tail", args.taper)
            for rows_per_chunk in (args.chunk_rows // 2, args.chunk_rows * 2):
                comm.set_option("chunk_rows", max(1024, rows_per_chunk))
                other[f"tree_sharded_allreduce_chunk_{max(1024, rows_per_chunk)}_ms"] = leg(
                    lambda: comm.score_sharded(tuples, out=out, combine=ddt.COMBINE_ALLREDUCE))
            comm.set_option("chunk_rows", args.chunk_rows)
            comm.set_option("comm_stream_priority", 1)
            other["tree_sharded_allreduce_comm_priority_ms"] = leg(lambda: comm.score_sharded(tuples, out=out, combine=ddt.COMBINE_ALLREDUCE))
            comm.set_option("comm_stream_priority", 0)
            tree_scores = out.clone()                            # combined scores of the tree-sharded job (all-reduce order)
            eng2 = ddt.Engine(local)                             # row-sharded replicas: every rank holds the WHOLE ensemble
            eng2.load_model(params, w, f, 0, 1)
            box2 = [ddt.comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(box2, src=0)
            comm2 = ddt.Comm(eng2, rank, world, box2[0])
            comm2.set_option("chunk_rows", args.chunk_rows)
            other["row_sharded_ms"] = leg(lambda: comm2.score_rowsharded(tuples, out=out))
            other["row_sharded_mtuples_per_s"] = round(N / other["row_sharded_ms"] / 1e3, 3)
            scale = float(tree_scores.abs().max().item()) or 1.0
            other["row_vs_tree_max_abs_diff_rel"] = float((out - tree_scores).abs().max().item()) / scale   # summation order differs
            # hybrid (ddt_comm_create_hybrid, csrc/ddt_comm.cpp): Gr row groups of Gt consecutive ranks, each a tree-sharded job on its slice of
            # the rows -- a rank ranks and scores N/Gr tuples against T/Gt trees (the replicated pre-pass shrinks by Gr), the all-reduce runs
            # inside a row group (a communicator split off the world communicator in C++).  Two forms: the scores stay with their row group (the
            # way the reference returns a device's rows from that device) / every finished piece is handed to the other row groups while the
            # next one is scored (all rows on every rank, like the other modes).
            for Gt in hybrid_tree_groups(world):
                Gr, tg = world // Gt, rank % Gt
                engh = ddt.Engine(local)
                engh.load_model(params, w, f, tg, Gt)
                boxh = [ddt.comm_unique_id() if rank == 0 else None]
                dist.broadcast_object_list(boxh, src=0)
                commh = ddt.Comm(engh, rank, world, boxh[0], tree_ranks=Gt)
                lo, hi = ddt.hybrid_rows(N, Gr, rank // Gt)
                commh.set_option("chunk_rows", max(1024, args.chunk_rows // Gr))
                key = f"hybrid_tree{Gt}_x_rows{Gr}"
                other[key + "_ms"] = leg(lambda: commh.score_hybrid(tuples, out=out, combine=ddt.COMBINE_ALLREDUCE, gather=False))
                other[key + "_mtuples_per_s"] = round(N / other[key + "_ms"] / 1e3, 3)
                diff = torch.tensor([float((out[lo:hi] - tree_scores[lo:hi]).abs().max().item()) / scale if hi > lo else 0.0], dtype=torch.float64, device=tuples.device)
                dist.all_reduce(diff, op=dist.ReduceOp.MAX)
                other[key + "_vs_tree_max_abs_diff_rel"] = float(diff.item())              # summation order differs (fewer partials per tuple)
                other[key + "_gathered_ms"] = leg(lambda: commh.score_hybrid(tuples, out=out, combine=ddt.COMBINE_ALLREDUCE, gather=True))
                other[key + "_gathered_mtuples_per_s"] = round(N / other[key + "_gathered_ms"] / 1e3, 3)
                other[key + "_gathered_vs_tree_max_abs_diff_rel"] = float((out - tree_scores).abs().max().item()) / scale
                commh.close()
                engh.close()
            # host buffers through the tree-sharded job (ddt_comm_score): tuples over PCIe once + xGMI hand-over, or G full copies
            srows = min(N, 8_000_000)
            host_t = tuples[:srows].cpu().numpy().view(np.uint32)
            for bc in (1, 0):
                comm.set_option("tuple_broadcast", bc)
                ms = leg(lambda: comm.score(host_t), steps=1)
                other[f"host_buffers_tuple_broadcast_{bc}_mtuples_per_s"] = round(srows / ms / 1e3, 4)
            comm.set_option("tuple_broadcast", -1)
            other["note"] = ("ms per step, max over ranks, 2 steps each after one warm-up, outside the timed region; row-sharded = replicas "
                             "only (whole ensemble per GPU, tuples partitioned, every step of scores handed to all peers), exact reference-order sums; "
                             "hybrid_treeA_x_rowsB = ddt_comm_create_hybrid with tree_ranks A: B row groups of A consecutive ranks, each a tree-sharded job "
                             "on its slice of the rows, all-reduce inside the row group (scores stay with their row group; _gathered: every piece handed to the "
                             "other row groups while the next is scored)")
            comm2.close()
            eng2.close()
        except Exception as ex:  # never at the price of the headline line
            other["error"] = repr(ex)
        if rank == 0:
            line["other_modes"] = other
    if comm is not None:
        comm.close()
    if multi:
        dist.destroy_process_group()
    eng.close()
    if wd is not None:  # the watchdog also covers the teardown: a rank that failed a leg alone must not leave the others waiting
        wd.cancel()
    if inproc_env is not None:
        return line if rank == 0 else None
    import ctypes
    ctypes.CDLL(None).fflush(None)  # C stdio buffers of native libraries -> stderr, before stdout comes back
    sys.stdout.flush()
    os.dup2(saved_stdout, 1)
    os.close(saved_stdout)
    if rank == 0:
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
