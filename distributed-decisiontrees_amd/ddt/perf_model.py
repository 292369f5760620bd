"""Analytic performance model.

Part 1 restates the reference's offline sizing / throughput model (`profiler/profiler.cpp:51-118`,
`profiler/profiler_performance_model.cpp:50-110`): engine throughput = f * Ncu * Npe / (depth * Ntrees)
tuples/s, a system is capped by the PCIe tuple rate and (for more than one FPGA) the network tuple rate.
Part 2 re-parameterises the same idea for MI355X with the constants measured in profiles/: the per-GPU rate is
min(LDS-pipe node-visit ceiling x achieved efficiency, HBM rate), tree-sharding divides the visits per GPU and
adds an all-reduce of 4 bytes per tuple that overlaps with scoring.
"""
from __future__ import annotations

from dataclasses import dataclass


# ---- part 1: the reference's model, verbatim semantics ----------------------------------------------------
@dataclass
class FpgaPlatform:
    freq_mhz: int = 150            # profiler.cpp:35
    n_cu: int = 4                  # :36  (the published RTL instantiates 8 clusters, DTEngine_Types.sv:26)
    n_pe: int = 8                  # :37
    max_nodes_per_pe: int = 8192   # :38
    pcie_gbps: float = 2.2         # :40
    network_gbps: float = 3.0      # :41
    tuple_bytes: int = 128         # :34


def engine_throughput(p: FpgaPlatform, depth: int, n_trees: int) -> float:
    """profiler.cpp:97-102 -- tuples/s of one FPGA."""
    return float(p.freq_mhz * 1e6 * p.n_cu * p.n_pe) / float(depth * n_trees)


def sizing(p: FpgaPlatform, n_trees: int, depth: int) -> dict:
    """profiler.cpp:51-86 -- min/max useful FPGA count and the modelled throughput at each."""
    cap = p.n_cu * p.n_pe * p.max_nodes_per_pe
    want = n_trees * (2 ** depth)
    te = engine_throughput(p, depth, n_trees)
    out = {"max_trees_size_in_fpga": cap, "user_desired_tree_size": want, "engine_tuples_per_s": te}
    if want <= cap:
        out.update(max_fpgas=1, max_throughput=te)
    else:
        out["max_fpgas"] = int(p.pcie_gbps * 1e9 / (te * p.tuple_bytes))  # C truncation of the double
        out["min_fpgas"] = int(want / cap)
        out["min_throughput"] = out["min_fpgas"] * te
        out["max_throughput"] = out["max_fpgas"] * te
    return out


def system_throughput(p: FpgaPlatform, n_fpgas: int, depth_incl_leaf_level: int, n_trees: int) -> float:
    """profiler_performance_model.cpp:70-75,101-110 -- note it counts depth-1 compare levels."""
    te = n_fpgas * engine_throughput(p, depth_incl_leaf_level - 1, n_trees)
    t_mem = p.pcie_gbps * 1e9 / p.tuple_bytes
    t_net = p.network_gbps * 1e9 / p.tuple_bytes
    return min(te, t_mem, t_net) if n_fpgas > 1 else min(te, t_mem)


# ---- part 2: MI355X ----------------------------------------------------------------------------------------
@dataclass
class Mi355x:
    cus: int = 256
    clock_hz: float = 2.4e9
    hbm_bytes_per_s: float = 8.0e12        # spec peak; ~6.3e12 achievable (MI355X_MICROARCH.md)
    ds_ops_per_visit: float = 1.75         # the shipped _gl_s2 walk (profiles/archive/r03_pmc_q16_gl_s2.md: 1.75 DS + 4.36 VALU per visit)
    lds_cycles_per_ds_op: float = 2.48     # measured incl. bank conflicts
    lds_efficiency: float = 0.93           # achieved / LDS-pipe ceiling: 8.4-8.5 T visits/s of 9.06 T (round 4, pinned read order: four chains in
                                           # flight per lane); the VALU issue bound, 4 cycles x 4.36 instructions per visit = 9.0 T/s, is as near
    hbm_efficiency: float = 0.715          # streaming kernel with phased result stores, measured on config 1 (5.73 TB/s of 8; round 3, direct stores: 0.60)
    prepass_hbm_efficiency: float = 0.60   # the rank pre-pass's 2 : 1 read : write mix (4.7-5.0 TB/s: profiles/r04_stream_phased_stores.md section 4)
    allreduce_alg_bytes_per_s: float = 87e9  # ring over xGMI: ~153 GB/s link x 8/14; NOT measured: no multi-GPU box
    rccl_cus: int = 32                       # CUs a collective's kernels hold while it runs -- an ASSUMPTION (RCCL's channel count on this
                                             # node is unknown until the driver's 8-GPU run); what k CUs cost is measured: collective_cu_slowdown()


def lds_visit_ceiling(g: Mi355x) -> float:
    """node visits per second if the LDS pipe were 100 % busy"""
    return g.cus * g.clock_hz * 64.0 / (g.ds_ops_per_visit * g.lds_cycles_per_ds_op)


# VALU instructions per node visit of the shipped rank-quantised walks (rocprofv3 SQ_INSTS_VALU / visits: profiles/archive/r03_pmc_q16_gl_s2.md depth 8,
# r03_pmc_cfg2_q16.md depth 6); other depths: the walk's 4 + per-tree work (leaf read, adder tree, scalar levels) spread over D visits
VALU_PER_VISIT = {8: 4.36, 6: 5.61}


def valu_visit_ceiling(g: Mi355x, depth: int) -> float:
    """node visits per second at the VALU issue bound: a wave instruction takes 4 cycles on its SIMD (16 lanes per cycle, 4 SIMDs per CU)"""
    return g.cus * 4 * g.clock_hz * 64.0 / (4.0 * VALU_PER_VISIT.get(depth, 4.0 + 9.0 / depth))


def predict(g: Mi355x, n_trees: int, depth: int, n_features: int, n_gpus: int = 1, rows: float = 1e8) -> dict:
    """Predicted whole-job Mtuples/s for the tree-sharded mode (trees / n_gpus per GPU, all tuples on every GPU): the walk at 95 % of
    its VALU issue bound (the LDS pipe is the second wall), or the tuple stream at the measured HBM efficiency; ensembles on the
    rank-quantised path (>= 480 tree-levels) also pay the HBM-bound rank pre-pass (4F bytes read + 2F written per tuple)."""
    visits = (n_trees / n_gpus) * depth
    t_valu = rows * visits / (valu_visit_ceiling(g, depth) * 0.95)
    t_lds = rows * visits / (lds_visit_ceiling(g) * g.lds_efficiency)
    t_hbm = rows * (4 * n_features + 4) / (g.hbm_bytes_per_s * g.hbm_efficiency)
    t_walk = max(t_valu, t_lds)
    t_pre = rows * 6 * n_features / (g.hbm_bytes_per_s * g.prepass_hbm_efficiency) if visits >= 480 else 0.0
    t_score = max(t_walk, t_hbm) + t_pre
    t_comm = 0.0 if n_gpus == 1 else rows * 4 / g.allreduce_alg_bytes_per_s
    t = max(t_score, t_comm) + (0.0 if n_gpus == 1 else min(t_score, t_comm) / 8.0)  # 8 pipelined chunks: one is exposed
    return {"seconds": t, "mtuples_per_s": rows / t / 1e6, "bound": ("valu" if t_valu >= t_lds else "lds") if t_walk >= t_hbm else "hbm",
            "t_score": t_score, "t_comm": t_comm}


# ---- part 3: the engine's own cost model (what ddt_choice.cpp's auto_variant encodes), per GPU ----------------
@dataclass
class PathCosts:
    """Measured on one MI355X in round 4, milliseconds per 100 M tuples of 32 fp32 features, depth-8 trees (gpurun_out/r04_s3, r04_s4, r04_s5 =
    profiles/r04_q16_pinned_persistent.md: `bench.py [--shard-of G]` with the pinned-read-order kernel q16_d8_c8_u4_gl_s2_cm_x; two boxes, ~1 % apart)."""
    q16_ms_per_chunk: float = 0.751       # per chunk of 8 trees (EMPTY padding included): 125 chunks 94.4-95.1, 63: 48.5, 32: 24.8-24.9, 16: 12.85-13.04 => 8.5 T node visits/s
    fp32_ms_per_tree: float = 0.147       # score_tile_kernel: 5.4 T node visits/s
    q16_fixed: float = 0.82               # per-tile fixed cost of the q16 scoring kernel (block turn-over: tile DMA, first chunk, barriers)
    fp32_fixed: float = 3.2               # per-tile fixed cost of the fp32 tile kernel (tuple load phase)
    prepass_two_kernel: float = 9.8       # transpose_kernel + rank_kernel (tables too big for 8 feature groups, or > 32 tuple words)
    # LDS-resident pre-pass in 1 / 2 / 4 / 8 feature groups (fused / grouped_rank_kernel): floor + ms per probe (log2 P probes).  Round-4 lines:
    # 125 trees (2 groups) 4.07-4.13, 250 trees (2 groups) 4.55-4.60, 500 trees (4 groups) 4.77, 1000 trees (8 groups) 5.6-5.74
    prepass_base: tuple = (3.28, 3.40, 3.67, 4.60)
    prepass_per_probe: tuple = (0.35, 0.35, 0.275, 0.275)
    keys_per_group: int = 32_000          # distinct thresholds whose tables (+ pads, bucket starts) fit 160 KiB of LDS


def prepass_ms(keys: int, c: "PathCosts") -> float:
    """The engine's choice (ddt_image.cpp build_prepass_image): cheapest feasible number of feature groups.  Bucket
    budget -> probes: the LDS left after the tables holds ~2 bytes per bucket; P = power of two above the fullest bucket,
    about 4x the mean occupancy for thresholds uniform in value."""
    best = c.prepass_two_kernel
    for i, g in enumerate((1, 2, 4, 8)):
        per_group = keys / g
        left = 160 * 1024 - per_group * 4.3
        if left <= 4096:
            continue
        occupancy = per_group / (left / 2.0)
        probes = 1
        while (1 << probes) <= 4.0 * max(occupancy, 0.25) + 1.0:
            probes += 1
        best = min(best, c.prepass_base[i] + c.prepass_per_probe[i] * probes)
    return best


def engine_ms(trees: int, depth: int = 8, rows: float = 1e8, c: PathCosts = PathCosts()) -> dict:
    """Predicted time of one scoring call on one GPU and the path the engine picks (thresholds of auto_variant)."""
    scale = rows / 1e8 * depth / 8.0
    keys = trees * (2 ** depth - 1)  # upper bound: every node a distinct threshold
    pre = prepass_ms(keys, c)
    chunks = -(-trees // 8)
    score = (c.q16_fixed + c.q16_ms_per_chunk * chunks) * scale
    q16 = pre * rows / 1e8 + score
    fp32 = (c.fp32_fixed + c.fp32_ms_per_tree * trees) * scale
    q16_ok = trees >= 224 or (keys <= 8 * c.keys_per_group and trees * depth >= 640)  # ddt_choice.cpp kQ16MinTreeLevels
    path = "q16" if q16_ok else "fp32"
    return {"path": path, "ms": q16 if path == "q16" else fp32, "q16_ms": q16, "fp32_ms": fp32, "prepass_ms": pre * rows / 1e8, "score_ms": score}


def collective_cu_slowdown(score_ms: float, prepass_ms_: float, busy_fraction: float, g: "Mi355x") -> float:
    """What the collectives' kernels cost the scoring: while a collective runs its workgroups hold `g.rccl_cus` CUs (dealt over the XCDs like
    any launch).  Measured with CU-masked streams on one GPU (profiles/r04_cu_mask_probe.md, a 125-tree shard, 100 M tuples): k CUs off, k / 8
    in every XCD: the step takes 1.12x at k = 16 and 32, 1.28x at 64 -- the VALU-bound scoring part scales with 256 / (256 - k), the HBM-bound
    pre-pass does not.  (All k in ONE XCD: 1.34x at 8, 1.97x at 16 for the plain launch -- the dispatcher deals its blocks round-robin over
    the XCDs -- and 1.09x / 1.25x for the persistent kernel with its ticket counter, which is why engines inside a job take that kernel.)
    `busy_fraction` = share of the scoring time during which a collective is in flight.  Returns extra milliseconds."""
    lost = g.cus / (g.cus - g.rccl_cus) - 1.0
    return score_ms * busy_fraction * lost


def tree_sharded_ms(trees: int, n_gpus: int, depth: int = 8, rows: float = 1e8, g: Mi355x = Mi355x(), chunks: int = 8,
                    taper: bool = True) -> dict:
    """Whole-job time of the tree-sharded mode: per-rank scoring of ceil(T/G) trees, slowed by the CUs the overlapped all-reduces hold,
    + the exposed part of the chunk pipeline (the last piece: a chunk, or a quarter of one with the tapered tail of csrc/ddt_comm.cpp)."""
    per = -(-trees // n_gpus)
    e = engine_ms(per, depth, rows)
    if n_gpus == 1:
        return {"ms": e["ms"], "mtuples_per_s": rows / e["ms"] / 1e3, "path": e["path"], "score_ms": e["ms"], "exposed_comm_ms": 0.0, "cu_share_ms": 0.0}
    comm_total = rows * 4 / g.allreduce_alg_bytes_per_s * 1e3            # all chunks' all-reduces, back to back
    exposed = comm_total / chunks / (4.0 if taper else 1.0)
    busy = min(1.0, (comm_total - exposed) / e["ms"])                      # collectives in flight while the rank scores
    cu = collective_cu_slowdown(e["score_ms"], e["prepass_ms"], busy, g)
    ms = e["ms"] + cu + exposed
    return {"ms": ms, "mtuples_per_s": rows / ms / 1e3, "path": e["path"], "score_ms": e["ms"], "exposed_comm_ms": exposed, "cu_share_ms": cu}


def row_sharded_ms(trees: int, n_gpus: int, depth: int = 8, rows: float = 1e8, g: Mi355x = Mi355x(), chunks: int = 8) -> dict:
    """Whole-job time of the row-sharded mode ("replicas only", csrc/ddt_comm.cpp ddt_score_rowsharded_device): every rank scores
    rows / G tuples against the WHOLE ensemble; each finished step goes to all peers while the next one is scored, so only the
    last step's messages (rows / chunks scores to G - 1 peers, one xGMI link each) are exposed."""
    e = engine_ms(trees, depth, rows / n_gpus)
    link = 153e9 / 2                       # one xGMI link, one direction
    comm = 0.0 if n_gpus == 1 else (rows / chunks / n_gpus) * 4 / link * 1e3 + 0.05
    ms = e["ms"] + comm
    return {"ms": ms, "mtuples_per_s": rows / ms / 1e3, "path": e["path"], "score_ms": e["ms"], "exposed_comm_ms": comm}


def hybrid_ms(trees: int, tree_ranks: int, row_groups: int, depth: int = 8, rows: float = 1e8, g: Mi355x = Mi355x(), chunks: int = 8,
              gather: bool = False) -> dict:
    """Whole-job time of the hybrid job (csrc/ddt_comm.cpp ddt_comm_create_hybrid / ddt_score_hybrid_device; bench.py other_modes
    "hybrid_tree{Gt}_x_rows{Gr}[_gathered]"): `row_groups` row groups of `tree_ranks` consecutive ranks; a rank ranks and scores
    rows / Gr tuples against ceil(T / Gt) trees, so the replicated rank pre-pass of the tree-sharded mode shrinks by Gr; the all-reduce stays
    inside a row group (its last quarter piece exposed).  gather: every finished piece also goes to the Gr - 1 ranks with the same tree
    shard in the other row groups (one xGMI link each) while the next piece is scored; the last piece's hand-over is exposed."""
    per = -(-trees // tree_ranks)
    r = rows / row_groups
    e = engine_ms(per, depth, r)
    comm_total = 0.0 if tree_ranks == 1 else r * 4 / g.allreduce_alg_bytes_per_s * 1e3
    exposed = comm_total / chunks / 4.0
    busy = min(1.0, (comm_total - exposed) / e["ms"]) if tree_ranks > 1 else 0.0
    cu = collective_cu_slowdown(e["score_ms"], e["prepass_ms"], busy, g)
    hand_over = (r / chunks / 4.0) * 4 / (153e9 / 2) * 1e3 + 0.05 if gather and row_groups > 1 else 0.0
    ms = e["ms"] + cu + exposed + hand_over
    return {"ms": ms, "mtuples_per_s": rows / ms / 1e3, "path": e["path"], "score_ms": e["ms"], "exposed_comm_ms": exposed + hand_over, "cu_share_ms": cu}


# ---- part 4: sparse forests (config 4): the vector-memory lane-address ceiling ---------------------------------
@dataclass
class SparseCosts:
    """Measured on one MI355X (profiles/archive/r03_sparse_dense_level_k.json, r03_sparse_schedules_and_blocks.json; round 2:
    r02_pmc_sparse_k8_t512.md): the deep phase gathers one 16-byte record per lane and visit; the vector-memory path takes about ONE
    lane address per cycle and CU."""
    lane_addresses_per_cycle_per_cu: float = 1.0
    efficiency_k8: float = 0.87          # 0.53 T gathers/s at K = 8 (two blocks per CU, rotating deep phase) against CUs x clock = 0.61 T/s
    top_overlapped: bool = True          # two blocks per CU: one block's top phase runs under the other's deep phase (round 2: 0.72, serial)
    lds_visit_rate: float = 7.0e12       # top-phase visits run at the LDS rate of the dense kernels; a minor term


def predict_sparse(trees: int, mean_visits_per_tree: float, top_levels: int = 8, rows: float = 1e7, g: Mi355x = Mi355x(),
                   c: SparseCosts = SparseCosts()) -> dict:
    """Predicted Mtuples/s of score_sparse_kernel: `mean_visits_per_tree` node visits per tuple and tree (the leaf depth
    actually walked), the first `top_levels` of them out of LDS, the rest as gathers at the lane-address ceiling."""
    deep = max(0.0, mean_visits_per_tree - top_levels)
    ceiling = g.cus * g.clock_hz * c.lane_addresses_per_cycle_per_cu
    t_deep = rows * trees * deep / (ceiling * c.efficiency_k8)
    t_top = rows * trees * min(mean_visits_per_tree, top_levels) / c.lds_visit_rate
    t = max(t_deep, t_top) if c.top_overlapped else t_deep + t_top
    return {"seconds": t, "mtuples_per_s": rows / t / 1e6, "gather_ceiling_per_s": ceiling, "deep_visits_per_tree": deep,
            "ceiling_mtuples_per_s": (ceiling / (trees * deep) / 1e6) if deep else float("inf")}
