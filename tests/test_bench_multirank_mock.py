"""bench.py's N > 1 orchestration, run end to end on the CPU: 2, 4 and 8 ranks as THREADS, each calling bench.main() exactly as a rank
process would (same argv the driver passes, RANK / LOCAL_RANK / WORLD_SIZE handed in), against the REAL csrc/*.cpp host side built on the
deferred-execution model of HIP streams + RCCL (tests/mock_hip/) and a stand-in for the few torch calls bench.py makes
(tests/fake_torch.py).  What this covers that nothing else on the CPU does: the communicator-id exchange, the timed tree-sharded job,
scaling_detail, EVERY leg of other_modes (chain, untapered, chunk sizes, comm priority, row-sharded replicas, both hybrid forms for each
split, host buffers with and without tuple broadcast), the --shard hybrid / --shard rows headline paths, and the teardown order -- the
path the driver runs at round end on 8 GPUs, which no single-GPU box can exercise."""
import importlib.util
import os
import sys
import threading

import numpy as np
import pytest

import ddt
from ddt import _lib
from oracle import oracle as O
from tests import fake_torch
from tests.test_engine_mock import _build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench_module():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _run(monkeypatch, world_size, extra, rows=6000, trees=96):
    L = _build("libddt_host_mock.so")
    L.mock_reset(2, 11, 8)
    monkeypatch.setattr(_lib, "_lib", L)
    world = fake_torch.World(world_size)
    ft = fake_torch.make(world, L)
    monkeypatch.setitem(sys.modules, "torch", ft)
    monkeypatch.setitem(sys.modules, "torch.distributed", ft.distributed)
    monkeypatch.setitem(sys.modules, "torch.cuda", ft.cuda)
    bench = _bench_module()
    argv = ["--gpus", str(world_size), "--steps", "2", "--warmup", "1", "--rows", str(rows), "--trees", str(trees), "--chunk-rows", "2048"] + extra
    res, errs = [None] * world_size, [None] * world_size

    def rank_main(r):
        world.local.rank = r
        try:
            res[r] = bench.main(argv, inproc_env={"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(world_size)})
        except BaseException as ex:  # noqa: BLE001  (SystemExit of a rank is a failure here too)
            errs[r] = ex
            world.barrier.abort()

    ths = [threading.Thread(target=rank_main, args=(r,)) for r in range(world_size)]
    for t in ths:
        t.start()
    for t in ths:
        t.join(600)
    assert not any(t.is_alive() for t in ths), "a rank hung"
    assert errs == [None] * world_size, errs
    assert all(r is None for r in res[1:]) and res[0] is not None
    return res[0], L


@pytest.mark.parametrize("world_size", [2, 4, 8])
def test_tree_sharded_line_and_every_other_mode(monkeypatch, world_size):
    line, L = _run(monkeypatch, world_size, [])
    assert line["n_gpus"] == world_size and line["config"]["parallelism"] == f"tree-shard{world_size}" and line["value"] > 0
    assert line["config"]["collectives"].startswith("C-ABI ddt_comm")
    sd = line["scaling_detail"]
    assert "error" not in sd and sd["trees_on_this_rank"] == 96 // world_size
    assert "roofline" not in line or "scope" in line["roofline"]     # (the model's events measure no time: no roofline object rather than a division by zero)
    om = line["other_modes"]
    assert "error" not in om and "status" not in om, om
    want = ["tree_sharded_chain_ms", "tree_sharded_allreduce_untapered_ms", "tree_sharded_allreduce_chunk_1024_ms", "tree_sharded_allreduce_chunk_4096_ms",
            "tree_sharded_allreduce_comm_priority_ms", "row_sharded_ms", "host_buffers_tuple_broadcast_1_mtuples_per_s", "host_buffers_tuple_broadcast_0_mtuples_per_s"]
    splits = bench_splits(world_size)
    for Gt in splits:
        k = f"hybrid_tree{Gt}_x_rows{world_size // Gt}"
        want += [k + "_ms", k + "_gathered_ms"]
        assert om[k + "_vs_tree_max_abs_diff_rel"] < 1e-5 and om[k + "_gathered_vs_tree_max_abs_diff_rel"] < 1e-5   # summation order only
    assert world_size == 2 or splits, "no hybrid split exercised"
    for k in want:
        assert k in om and om[k] > 0, (k, om)
    assert om["row_vs_tree_max_abs_diff_rel"] < 1e-5
    assert L.mock_errors() == 0
    _check_parity_and_cpu(line, "within_tolerance", world_size)


def _check_parity_and_cpu(line, required, world_size, rows=6000):
    """every N > 1 line carries `parity` (the TIMED job's combined result against the oracle's model of the job) and `cpu_baseline`"""
    par, cpu = line["parity"], line["cpu_baseline"]
    assert par["required"] == required and par[required] is True, par
    assert par["within_tolerance"] is True and par["max_err_over_tolerance"] <= 1.0, par   # north_star's 1e-6, whatever the combine
    assert 0 < par["rows_checked"] <= rows and par["tolerance_rows_checked"] == par["rows_checked"]
    assert par["rows_that_differ_from_chain_oracle"] == 0 or required != "bit_exact"
    assert cpu["value"] > 0 and cpu["cores"] >= 1 and cpu["kind"] == "port" and "wait behind a barrier" in cpu["sample"]


def bench_splits(world_size):
    return _bench_module().hybrid_tree_groups(world_size)


@pytest.mark.parametrize("extra,par", [(["--shard", "hybrid", "--tree-ranks", "2"], "hybrid-tree2-x-rows4"),
                                       (["--shard", "hybrid", "--tree-ranks", "4", "--no-gather"], "hybrid-tree4-x-rows2"),
                                       (["--shard", "rows"], "row-shard8"),
                                       (["--combine", "chain"], "tree-shard8")])
def test_other_headline_shardings_on_eight_ranks(monkeypatch, extra, par):
    line, L = _run(monkeypatch, 8, extra + ["--no-other-modes"])
    assert line["config"]["parallelism"] == par and line["n_gpus"] == 8 and line["value"] > 0
    assert L.mock_errors() == 0
    # chain combine and replicas: the reference's own order, bit for bit; all-reduce inside a row group: north_star's tolerance
    _check_parity_and_cpu(line, "bit_exact" if ("chain" in extra or "rows" in extra) else "within_tolerance", 8)


@pytest.mark.parametrize("world_size,extra", [(2, ["--combine", "chain"]), (4, ["--combine", "chain"]), (4, ["--shard", "hybrid", "--tree-ranks", "2", "--combine", "chain"]),
                                              (2, ["--shard", "rows"]), (4, ["--shard", "hybrid", "--tree-ranks", "2", "--no-gather"])])
def test_parity_of_the_timed_job_on_fewer_ranks(monkeypatch, world_size, extra):
    line, L = _run(monkeypatch, world_size, extra + ["--no-other-modes"], rows=5000, trees=72)
    required = "bit_exact" if ("chain" in extra or "rows" in extra) else "within_tolerance"
    _check_parity_and_cpu(line, required, world_size, rows=5000)
    assert L.mock_errors() == 0


def test_a_wrong_combined_score_fails_the_parity_leg(monkeypatch):
    """the check has teeth: one score of the timed job's result disturbed by one ulp -> bit_exact false (and counted)"""
    orig_clone = fake_torch.Tensor.clone

    def bad_clone(self):
        t = orig_clone(self)
        if t.a.dtype == np.float32 and t.a.ndim == 1 and t.a.size == 5000:
            t.a.view(np.uint32)[17] ^= 1
        return t

    monkeypatch.setattr(fake_torch.Tensor, "clone", bad_clone)
    line, _ = _run(monkeypatch, 2, ["--combine", "chain", "--no-other-modes"], rows=5000, trees=72)
    par = line["parity"]
    assert par["required"] == "bit_exact" and par["bit_exact"] is False and par["rows_that_differ_from_chain_oracle"] == 1


def test_the_scores_of_the_timed_job_are_the_oracles(monkeypatch):
    """the timed job of an 8-rank bench run (here the hybrid 2 x 4 job with its pieces handed round: nothing overwrites `out` behind it, the
    tree-sharded line's scaling_detail pass does) leaves, on every rank, the oracle's scores for all rows within the summation-order
    tolerance -- read back through a hook on the stand-in's buffers.  The tree-sharded job's scores are tied to these by the
    *_vs_tree_max_abs_diff_rel figures asserted above."""
    kept = {}
    real_make = fake_torch.make

    def spy(world, L):
        ft = real_make(world, L)
        inner = ft.empty

        def empty(shape, dtype=fake_torch.float32, device=None):
            t = inner(shape, dtype, device)
            if dtype == fake_torch.float32 and not isinstance(shape, tuple):
                kept.setdefault(world.local.rank, t)        # the first fp32 vector a rank allocates is bench.py's `out`
            return t

        ft.empty = empty
        return ft

    monkeypatch.setattr(fake_torch, "make", spy)
    line, L = _run(monkeypatch, 8, ["--no-other-modes", "--shard", "hybrid", "--tree-ranks", "4"], rows=5000, trees=80)
    m, x = O.gen_model(80, 8, 32, 0), O.gen_tuples(0, 5000, 32, 0)
    gold = O.score(m, x, want_gold=True)[1]
    for r in range(8):
        assert np.allclose(kept[r].a, gold, rtol=1e-5, atol=1e-5), r
    assert line["config"]["trees"] == 80


# ---------------------------------------------------------------------------------------------- other_configs (N = 1, the default command)
def test_other_configs_plumbing_on_the_cpu_model(monkeypatch):
    """`other_configs` of the driver's line: run_side_config for the dense configs end to end on the CPU model of the host
    side (a few thousand rows instead of BASELINE's 10^7..10^8): model load, the engine's own kernel choice -- config 6, the reference's 512 x
    depth 12, on the deep kernel in parts -- the timed steps, the roofline arithmetic and the oracle check of a prefix."""
    L = _build("libddt_host_mock.so")
    L.mock_reset(2, 5, 8)
    monkeypatch.setattr(_lib, "_lib", L)
    world = fake_torch.World(1)
    ft = fake_torch.make(world, L)
    monkeypatch.setitem(sys.modules, "torch", ft)
    monkeypatch.setitem(sys.modules, "torch.distributed", ft.distributed)
    monkeypatch.setitem(sys.modules, "torch.cuda", ft.cuda)
    world.local.rank = 0
    bench = _bench_module()
    for cfg, kernel in ((2, "q16_d6_c16_u4_s2"), (5, "q16_d8_c8_u4_gl_s2_cm_p"), (6, "q16d_d12_k9_c4_u4_cm")):
        r = bench.run_side_config(cfg, 0, check_rows=1500, rows=2100)
        assert r["kernel"] == kernel and r["fallback_kernel"] is False, r
        assert r["parity"]["bit_exact"] is True and r["parity"]["rows_checked"] == 1500 and r["value"] > 0 and r["steps"] >= 3, r
        assert r["roofline"]["alg_bytes_per_launch"] > 2100 * 4 * 28 and r["roofline"]["peak"] == 8000.0, r
    assert L.mock_errors() == 0


def test_per_rank_proxies_plumbing_on_the_cpu_model(monkeypatch):
    """`other_modes.per_rank_proxies` of the driver's line: one rank's workload of the three 8-GPU jobs through the real host
    side on the CPU model -- shard 3 of 8, shard 1 of 2 on a quarter of the rows, the whole ensemble on an eighth -- each checked against the
    oracle's model of the shard, plus the analytic model's 8-GPU prediction with its assumption spelled out; a failing leg costs only its entry."""
    import numpy as np

    import ddt
    L = _build("libddt_host_mock.so")
    L.mock_reset(2, 7, 8)
    monkeypatch.setattr(_lib, "_lib", L)
    world = fake_torch.World(1)
    ft = fake_torch.make(world, L)
    monkeypatch.setitem(sys.modules, "torch", ft)
    monkeypatch.setitem(sys.modules, "torch.distributed", ft.distributed)
    monkeypatch.setitem(sys.modules, "torch.cuda", ft.cuda)
    world.local.rank = 0
    bench = _bench_module()
    T, D, F, N = 1000, 8, 32, 16 * 1024
    w, f = ddt.synth_model(T, D, F, 0)
    eng = ddt.Engine(0)
    tuples = eng.synth_tuples_device(0, N, F, 0)
    p = bench.collect_rank_proxies(0, tuples, 100.0, (T, D, F), (w, f), check_rows=1024)
    # `other_modes.small_batches`: one call on a batch of a few tiles, cut into slices of the image against uncut, through the real host side
    eng.load_model(ddt.make_params(T, D, F), w, f)
    full = eng.score_device(tuples)
    ft.cuda.synchronize()
    sb = bench.collect_small_batches(eng, tuples, ft.empty(N, dtype=ft.float32, device=tuples.device), full.cpu().numpy().view(np.uint32), rows_list=(1024, 5000), reps=2)
    assert sb["1024"]["bit_exact_vs_timed_result"] is True and sb["5000"]["bit_exact_vs_timed_result"] is True and sb["5000"]["us_uncut"] > 0 and "note" in sb, sb
    eng.close()
    for name, trees, rows in (("shard_of_8", 125, N), ("hybrid_rank_of_2x4", 500, N // 4), ("replica_of_8", 1000, N // 8)):
        r = p[name]
        assert "error" not in r, r
        assert r["trees"] == trees and r["rows"] == rows and r["bit_exact"] is True and r["rows_checked"] == 1024 and r["compute_only_x"] > 0, (name, r)
    m8 = p["model_8gpu"]
    assert set(m8) >= {"tree_sharded_8", "hybrid_tree2_x_rows4_gathered", "replicas_8", "assumptions"} and "ASSUMPTION" in m8["assumptions"]
    assert 4.0 < m8["tree_sharded_8"]["x_over_model_1gpu"] < 8.0 and m8["replicas_8"]["ms"] > 0   # (16 k rows: the fixed terms dominate the other two)
    assert L.mock_errors() == 0

    def runner(name, idx, cnt, rows):
        if name == "hybrid_rank_of_2x4":
            raise RuntimeError("boom")
        return {"ms": 1.0}

    p = bench.collect_rank_proxies(0, tuples, 100.0, (T, D, F), (w, f), runner=runner)
    assert p["shard_of_8"] == {"ms": 1.0} and "boom" in p["hybrid_rank_of_2x4"]["error"] and p["replica_of_8"] == {"ms": 1.0} and "note" in p


def test_other_configs_budget_and_failures_never_cost_the_line():
    bench = _bench_module()
    calls = []

    def runner(cfg, dev):
        calls.append(cfg)
        if cfg == 2:
            raise RuntimeError("boom")
        import time as _t
        _t.sleep(0.3)
        return {"value": 1.0}

    oc = bench.collect_other_configs(0, 0.5, runner=runner)
    assert oc["1"] == {"value": 1.0} and "boom" in oc["2"]["error"] and oc["5"] == {"value": 1.0}
    assert "skipped" in oc["6"] and "skipped" in oc["4"] and calls == [1, 2, 5] and oc["seconds"] >= 0.5
