"""the port of the reference's analytic profiler, checked against the outputs of the
reference programs themselves (tests/golden/profiler*_ref.txt, produced by oracle/_ref -- the only
reference code that compiles), and the MI355X re-parameterisation against this repo's measurements."""
import os
import re

from ddt import perf_model as P

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_profiler_cpp_known_answers():
    txt = open(os.path.join(G, "profiler_ref.txt")).read()
    val = lambda k: float(re.search(re.escape(k) + r"\s*=\s*([0-9.e+]+)", txt).group(1))
    s = P.sizing(P.FpgaPlatform(), 512, 12)
    assert s["max_trees_size_in_fpga"] == val("max_trees_size_in_fpga")
    assert s["user_desired_tree_size"] == val("user_desired_tree_size")
    assert s["engine_tuples_per_s"] == val("ret_val")
    assert s["min_fpgas"] == val("Minimum no.of.fpgas_needed") and s["max_fpgas"] == val("Maximum no.of.fpgas_needed")
    thr = [float(v) for v in re.findall(r"Corresponding Throughput = ([0-9.e+]+)", txt)]
    assert abs(s["min_throughput"] - thr[0]) <= 1e-6 * thr[0] and abs(s["max_throughput"] - thr[1]) <= 1e-6 * thr[1]


def test_profiler_performance_model_sweep():
    rows = [l.split() for l in open(os.path.join(G, "profiler_performance_model_ref.txt")) if re.match(r"^\d+\s", l)]
    assert len(rows) == 40
    for n, d, t in rows:
        got = P.system_throughput(P.FpgaPlatform(), int(n), int(d), 512)
        assert abs(got - float(t)) <= 6e-6 * float(t), (n, d, t, got)  # the reference prints 6 significant digits


def test_reference_model_on_baseline_configs():
    # 1000 trees, depth 8 -> 0.6 Mtuples/s per FPGA with the model's 32 PEs, 1.2 M with the RTL's 64
    assert abs(P.engine_throughput(P.FpgaPlatform(), 8, 1000) - 0.6e6) < 1
    assert abs(P.engine_throughput(P.FpgaPlatform(n_cu=8), 8, 1000) - 1.2e6) < 1


def test_mi355x_model_matches_measurements_within_20_percent():
    g = P.Mi355x()
    measured = {  # profiles/r01_*: Mtuples/s on one MI355X
        (1000, 8, 32): 664.7, (100, 6, 28): 7222.0, (8, 4, 16): 70253.6}
    for (T, D, F), m in measured.items():
        p = P.predict(g, T, D, F)["mtuples_per_s"]
        assert 0.8 < p / m < 1.25, (T, D, F, p, m)
    assert P.predict(g, 8, 4, 16)["bound"] == "hbm" and P.predict(g, 1000, 8, 32)["bound"] == "lds"
    s8 = P.predict(g, 1000, 8, 32, 8)["mtuples_per_s"] / P.predict(g, 1000, 8, 32, 1)["mtuples_per_s"]
    assert 6.0 < s8 <= 8.0  # the >= 6x aggregate target of the north star is plausible with overlapped all-reduce


def test_engine_cost_model_matches_the_measured_shard_regime():
    # per-rank scoring time of the headline job's shards, measured on one MI355X (profiles/r01_*), ms per 100 M tuples
    measured = {1000: 104.5, 125: 17.58}  # profiles/r03_bench_cfg3.log, r03_bench_shard_of_8.log (q16_d8_c8_u4_gl_s2_cm)
    for trees, ms in measured.items():
        e = P.engine_ms(trees)
        assert e["path"] == "q16"
        assert abs(e["ms"] - ms) <= 0.04 * ms, (trees, e, ms)
    assert P.engine_ms(100, depth=6)["path"] == "fp32"                       # config 2 stays on the fp32 tile kernel
    assert abs(P.engine_ms(125)["fp32_ms"] - 21.4) < 1.5                     # what the 8-way shard cost before the fused pre-pass
    s = {n: P.tree_sharded_ms(1000, n)["mtuples_per_s"] for n in (1, 2, 4, 8)}
    assert 930 < s[1] < 990 and 5.4 < s[8] / s[1] < 6.2                      # the replicated rank pre-pass holds tree sharding just below 6x
    # two tree groups x four row groups: the pre-pass runs on a quarter of the rows per rank
    h = {(gt, 8 // gt): P.hybrid_ms(1000, gt, 8 // gt)["mtuples_per_s"] / s[1] for gt in (1, 2, 4, 8)}
    assert h[(1, 8)] > h[(2, 4)] > h[(4, 2)] > h[(8, 1)] and h[(2, 4)] > 7.0 and abs(h[(8, 1)] - s[8] / s[1]) < 0.05


def test_sparse_forest_model_matches_the_config4_measurement():
    """profiles/r03_sparse_dense_level_k.json: 512 sparse trees, 12.06 visits per tuple and tree, K = 8 in two blocks per CU -> 256.6
    Mtuples/s measured (round 2, one phase-locked block, two-phase deep rounds: 213.7)."""
    r = P.predict_sparse(512, 12.059, top_levels=8)
    assert abs(r["mtuples_per_s"] - 256.6) / 256.6 < 0.1
    r2 = P.predict_sparse(512, 12.059, top_levels=8, c=P.SparseCosts(efficiency_k8=0.72, top_overlapped=False))
    assert abs(r2["mtuples_per_s"] - 213.7) / 213.7 < 0.2
    assert 280 < r["ceiling_mtuples_per_s"] < 310  # the lane-address ceiling of this model at K = 8
    assert P.predict_sparse(512, 12.059, top_levels=9)["ceiling_mtuples_per_s"] > r["ceiling_mtuples_per_s"]


def test_row_sharded_model_beats_tree_sharding_when_the_ensemble_fits_one_gpu():
    from ddt.perf_model import row_sharded_ms, tree_sharded_ms

    one = row_sharded_ms(1000, 1)["ms"]
    assert abs(one - tree_sharded_ms(1000, 1)["ms"]) < 1e-9
    for g in (2, 4, 8):
        r, t = row_sharded_ms(1000, g), tree_sharded_ms(1000, g)
        assert r["ms"] < t["ms"] and one / r["ms"] > 0.95 * g and r["exposed_comm_ms"] < 0.5
