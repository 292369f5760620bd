#!/usr/bin/env python3
"""Regenerate profiles/pmc_traffic.json (config 3) and profiles/pmc_traffic_cfg4.json from the FETCH_SIZE / WRITE_SIZE passes of
tools/gpu_evidence.sh:  python tools/update_pmc_traffic.py gpurun_out/<tag>.  bench.py reads `hbm_bytes_per_launch` of these files
into roofline.traffic (labelled `traffic_source`: measured by rocprofv3 on the same command, in separate runs)."""
import glob
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def pmc(path, like):
    out = {}
    for db in glob.glob(path + "/**/*.db", recursive=True):
        cur = sqlite3.connect(db).cursor()
        q = "select counter_name, sum(value), count(*) from counters_collection where kernel_name like ? group by counter_name"
        for cn, v, n in cur.execute(q, (f"%{like}%",)):
            out[cn] = v / n
    return out


def launches(path, like):
    """launches of the kernels whose name contains `like` in the PMC run under `path` (0 if none)"""
    n = 0
    for db in glob.glob(path + "/**/*.db", recursive=True):
        cur = sqlite3.connect(db).cursor()
        for (k,) in cur.execute("select count(*) from (select distinct dispatch_id from counters_collection where kernel_name like ?)", (f"%{like}%",)):
            n += int(k)
    return n


def kib(d, key):
    return int(d[key] * 1024) if key in d else None


def main():
    ev = sys.argv[1].rstrip("/")
    cmd = "python bench.py --config {} --steps 3 --warmup 1 --no-cpu-baseline --no-streamed"
    # ---- config 3: rank-quantised path -----------------------------------------------------------------------------
    f, w = pmc(ev + "/fetch_cfg3", "score_q16"), pmc(ev + "/write_cfg3", "score_q16")
    fr, wr = kib(f, "FETCH_SIZE"), kib(w, "WRITE_SIZE")
    pre = {}
    for name, like in (("grouped_rank", "grouped_rank_kernel"), ("fused_rank", "fused_rank_kernel"), ("rank", "ddt::rank_kernel"),
                       ("transpose", "transpose_kernel")):
        f2, w2 = pmc(ev + "/fetch_cfg3", like), pmc(ev + "/write_cfg3", like)
        if f2 or w2:
            pre[name] = {"FETCH_SIZE": kib(f2, "FETCH_SIZE"), "WRITE_SIZE": kib(w2, "WRITE_SIZE")}
    step = 2 * fr + wr + sum(2 * v["FETCH_SIZE"] + v["WRITE_SIZE"] for v in pre.values())
    j = {"rows": 100000000, "trees": 1000, "kernel": "score_q16_kernel<8,8,4,23> (q16_d8_c8_u4_gl_s2_cm_x)",
         "fetch_bytes_raw_per_launch": fr, "fetch_bytes_x2_corrected_per_launch": 2 * fr, "write_bytes_per_launch": wr,
         "hbm_bytes_per_launch": 2 * fr + wr,
         "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes), averaged over the launches of the scoring kernel; FETCH_SIZE "
                 "doubled as MI355X_MICROARCH.md prescribes for wide coalesced reads on gfx950 (the q tiles arrive by 16 B/lane global->LDS DMA). "
                 "The scoring kernel reads the u16 rank tiles (6.4 GB) + the model image's L2 misses and writes 0.4 GB of scores; the fp32 tuples "
                 "(12.8 GB) are read once, by the rank pre-pass (`prepass`: raw counter bytes per launch of its kernel(s)).",
         "prepass": pre, "step_hbm_bytes_x2_corrected": step, "round": 4,
         "source": f"{ev} (tools/gpu_evidence.sh): `{cmd.format(3)}`"}
    json.dump(j, open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w"), indent=1)
    print("config 3: scoring kernel", j["hbm_bytes_per_launch"], "B/launch; whole step", step, "B;", pre)
    # ---- config 1: stream kernel (each tuple read once, each score written once) ----------------------------------------
    f, w = pmc(ev + "/fetch_cfg1", "score_stream"), pmc(ev + "/write_cfg1", "score_stream")
    if f and w:
        fr1, wr1 = kib(f, "FETCH_SIZE"), kib(w, "WRITE_SIZE")
        j1 = {"rows": 200000000, "trees": 8, "kernel": "score_stream_kernel<4,4,4> (stream_d4_u4_l4, phased result stores)",
              "fetch_bytes_raw_per_launch": fr1, "fetch_bytes_x2_corrected_per_launch": 2 * fr1, "write_bytes_per_launch": wr1,
              "hbm_bytes_per_launch": 2 * fr1 + wr1, "step_hbm_bytes_x2_corrected": 2 * fr1 + wr1,
              "algorithmic_bytes_per_launch": 200000000 * 68,
              "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) of the config-1 bench command, averaged over the launches of "
                      "the stream kernel; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for wide coalesced reads on gfx950 (16 B/lane nontemporal loads).",
              "round": 4, "source": f"{ev} (tools/gpu_evidence_slim.sh): `python bench.py --config 1 --steps 10 --warmup 3 --no-cpu-baseline --no-streamed`"}
        json.dump(j1, open(os.path.join(ROOT, "profiles", "pmc_traffic_cfg1.json"), "w"), indent=1)
        print("config 1:", j1["hbm_bytes_per_launch"], "B/launch against", j1["algorithmic_bytes_per_launch"], "algorithmic")
    # ---- config 4: sparse forest (round 6: `sparse_r_k9_u8_t256` -- a rank pre-pass + the scoring kernel over the 32-bit rank tile) -------------
    p4 = os.path.join(ROOT, "profiles", "pmc_traffic_cfg4.json")
    f, w = pmc(ev + "/fetch_cfg4", "score_sparse"), pmc(ev + "/write_cfg4", "score_sparse")
    if f and w:
        fr4, wr4 = kib(f, "FETCH_SIZE"), kib(w, "WRITE_SIZE")
        tile = 10000000 * 64 * 4  # the rank tile the scoring kernel DMAs in (wide coalesced stream: counted at half on gfx950)
        pre4 = {}
        for name, like in (("rank32", "rank32_kernel"), ("transpose", "transpose_kernel")):
            f2, w2 = pmc(ev + "/fetch_cfg4", like), pmc(ev + "/write_cfg4", like)
            if f2 or w2:
                pre4[name] = {"FETCH_SIZE": kib(f2, "FETCH_SIZE"), "WRITE_SIZE": kib(w2, "WRITE_SIZE")}
        scoring = fr4 + tile // 2 + wr4
        c4 = {"rows": 10000000, "trees": 512, "kernel": "score_sparse_r_kernel<9,8,256> (sparse_r_k9_u8_t256)", "round": 6,
              "fetch_bytes_raw_per_launch": fr4, "write_bytes_per_launch": wr4, "tuple_stream_bytes": tile, "hbm_bytes_per_launch": scoring,
              "prepass": pre4, "step_hbm_bytes_x2_corrected": scoring + sum(2 * (v["FETCH_SIZE"] or 0) + (v["WRITE_SIZE"] or 0) for v in pre4.values()),
              "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) on the config-4 bench command, averaged over the launches. "
                      "FETCH_SIZE counts the L2's fabric-side read requests (Infinity-Cache hits included, MI355X_MICROARCH.md).  Scoring kernel: the 2.56 GB "
                      "rank tile arrives by wide global->LDS DMA (counted at half on gfx950: + tile / 2 added here); the rest are the L2 misses of the "
                      "pair-record gathers into the model image (16-byte records pulled as whole lines; the image fits the 256 MiB Infinity Cache, so most "
                      "of those requests do not reach HBM) -- the x2 correction is calibrated for wide streams only and is NOT applied to them.  `prepass`: raw "
                      "counter bytes per launch of the transpose and the rank32 kernel; the step figure doubles their FETCH (streams) -- an upper bound for "
                      "rank32, whose key-block gathers are not wide streams.",
              "source": f"{ev}: `{cmd.format(4)}`"}
        json.dump(c4, open(p4, "w"), indent=1)
        print("config 4:", c4["hbm_bytes_per_launch"], "B/launch (scoring); step", c4["step_hbm_bytes_x2_corrected"])
    # ---- config 6: the reference's own 512 x d12 x 32 on the deep kernel, in parts (round 5) -------------------------------------------
    f, w = pmc(ev + "/fetch_cfg6", "score_q16d"), pmc(ev + "/write_cfg6", "score_q16d")
    if f and w:
        fr6, wr6 = kib(f, "FETCH_SIZE"), kib(w, "WRITE_SIZE")
        # parts and pre-pass launches per step come from the RUN (they were constants, silently wrong once the part plan changes): the
        # command runs `steps_run` steps (3 timed + 1 warm-up; the transposed tuples are shared by a step's parts: one transpose per step)
        n_tr = launches(ev + "/fetch_cfg6", "transpose_kernel")
        steps_run = n_tr if n_tr else 4
        parts = max(1, round(launches(ev + "/fetch_cfg6", "score_q16d") / steps_run))
        pre6 = {}
        for name, like in (("rank", "ddt::rank_kernel"), ("transpose", "transpose_kernel"), ("fused_rank", "fused_rank_kernel"), ("grouped_rank", "grouped_rank_kernel")):
            f2, w2 = pmc(ev + "/fetch_cfg6", like), pmc(ev + "/write_cfg6", like)
            if f2 or w2:
                pre6[name] = {"FETCH_SIZE": kib(f2, "FETCH_SIZE"), "WRITE_SIZE": kib(w2, "WRITE_SIZE"),
                              "launches_per_step": max(1, round(launches(ev + "/fetch_cfg6", like) / steps_run))}
        scoring = parts * (2 * fr6 + wr6)
        step6 = scoring + sum(v["launches_per_step"] * (2 * (v["FETCH_SIZE"] or 0) + (v["WRITE_SIZE"] or 0)) for v in pre6.values())
        j6 = {"rows": 10000000, "trees": 512, "kernel": f"score_q16d_kernel<12,9,4> (q16d_d12_k9_c4_u4_cm), {parts} parts per step", "parts_per_step": parts,
              "fetch_bytes_raw_per_launch": fr6, "fetch_bytes_x2_corrected_per_launch": 2 * fr6, "write_bytes_per_launch": wr6,
              "hbm_bytes_per_launch": scoring,
              "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of the config-6 bench command.  The ensemble is scored in parts (`parts_per_step`, counted in the run): "
                      "`hbm_bytes_per_launch` is the sum over the scoring launches of one step (2 x FETCH + WRITE each; the u16 rank tiles, the "
                      "stage records' L2 misses, the sum's state between the parts), which is what the line's `kernel_ms` spans together with two "
                      "of the pre-passes; `prepass`: raw counter bytes per launch of the pre-pass kernels and their launches per step.",
              "prepass": pre6, "step_hbm_bytes_x2_corrected": step6, "algorithmic_bytes_per_step": 10000000 * 132 + 512 * (4 * 8191 + 2 * 4095),
              "round": 5, "source": f"{ev} (tools/gpu_evidence_r05.sh): `{cmd.format(6)}`"}
        json.dump(j6, open(os.path.join(ROOT, "profiles", "pmc_traffic_cfg6.json"), "w"), indent=1)
        print("config 6: scoring launches", scoring, "B/step; whole step", step6, "B")


if __name__ == "__main__":
    main()
