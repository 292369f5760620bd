#!/usr/bin/env python3
"""Summarise rocprofv3 outputs (rocpd sqlite .db files under gpurun_out/) into profiles/<tag>.json + .md.

  python tools/prof_summary.py <tag> [--stats gpurun_out/prof_stats] [--pmc gpurun_out/pmc1 gpurun_out/pmc2 ...]
                               [--kernel score_tile] [--rows N --trees T --levels D --features F]

Kernel-trace stats: per-kernel calls / total / average duration (the `rocprofv3 --kernel-trace --stats` view).
PMC: per-launch counter values for the dominant kernel, plus derived figures (LDS-pipe busy fraction,
cycles per DS instruction, HBM bytes per launch with the guide's FETCH_SIZE correction stated).
"""
import argparse
import glob
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def dbs(path):
    return sorted(glob.glob(os.path.join(path, "**", "*.db"), recursive=True)) if os.path.isdir(path) else [path]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("tag")
    ap.add_argument("--stats", default=None)
    ap.add_argument("--pmc", nargs="*", default=[])
    ap.add_argument("--kernel", default="score_tile")
    ap.add_argument("--rows", type=int, default=0)
    ap.add_argument("--trees", type=int, default=1000)
    ap.add_argument("--levels", type=int, default=8)
    ap.add_argument("--features", type=int, default=32)
    ap.add_argument("--cmd", default="")
    a = ap.parse_args()
    out = {"tag": a.tag, "command": a.cmd}
    md = [f"# rocprofv3 summary `{a.tag}`", "", f"command: `{a.cmd}`" if a.cmd else ""]
    if a.stats:
        rows = []
        for db in dbs(a.stats):
            cur = sqlite3.connect(db).cursor()
            for r in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
                rows.append({"kernel": r[0], "calls": r[1], "total_us": round(r[2], 1), "avg_us": round(r[3], 1),
                             "pct": round(r[4], 2)})
        out["kernel_stats"] = rows
        md += ["", "## kernel-trace --stats", "", "| kernel | calls | total (us) | avg (us) | % |", "|---|---|---|---|---|"]
        md += [f"| `{r['kernel']}` | {r['calls']} | {r['total_us']} | {r['avg_us']} | {r['pct']} |" for r in rows]
    if a.pmc:
        ctr, launches, dur = {}, 0, []
        kname = None
        for p in a.pmc:
            for db in dbs(p):
                cur = sqlite3.connect(db).cursor()
                q = ("select kernel_name, counter_name, sum(value), count(*) from counters_collection "
                     "where kernel_name like ? group by kernel_name, counter_name")
                for kn, cn, v, n in cur.execute(q, (f"%{a.kernel}%",)):
                    ctr[cn] = v / n
                    launches, kname = n, kn
        out["pmc_kernel"] = kname
        out["pmc_per_launch"] = {k: round(v, 1) for k, v in sorted(ctr.items())}
        d = {}
        CUS, XCD = 256, 8
        if "GRBM_GUI_ACTIVE" in ctr:
            d["kernel_cycles"] = ctr["GRBM_GUI_ACTIVE"] / XCD  # the counter is summed over the 8 XCDs
        if "SQ_LDS_IDX_ACTIVE" in ctr and "SQ_INSTS_LDS" in ctr:
            d["lds_cycles_per_ds_instr"] = ctr["SQ_LDS_IDX_ACTIVE"] / ctr["SQ_INSTS_LDS"]
            if "SQ_LDS_BANK_CONFLICT" in ctr:
                d["bank_conflict_cycles_per_ds_instr"] = ctr["SQ_LDS_BANK_CONFLICT"] / ctr["SQ_INSTS_LDS"]
                d["bank_conflict_share_of_lds_cycles"] = ctr["SQ_LDS_BANK_CONFLICT"] / ctr["SQ_LDS_IDX_ACTIVE"]
            if "kernel_cycles" in d:
                d["lds_pipe_busy_frac"] = ctr["SQ_LDS_IDX_ACTIVE"] / CUS / d["kernel_cycles"]
        if "SQ_INSTS_VALU" in ctr and "kernel_cycles" in d:
            d["valu_issue_frac"] = ctr["SQ_INSTS_VALU"] / (CUS * d["kernel_cycles"] * 2.0)  # 4 SIMDs x 1 wave64 op / 2 clk
        if a.rows:
            visits = a.rows * a.trees * a.levels
            for k in ("SQ_INSTS_LDS", "SQ_INSTS_VALU", "SQ_INSTS_SALU"):
                if k in ctr:
                    d[k.lower() + "_per_64_visits"] = ctr[k] / (visits / 64.0)
            alg = a.rows * (4 * a.features + 4)
            d["algorithmic_bytes_per_launch"] = alg
            if "FETCH_SIZE" in ctr:
                d["fetch_bytes_raw"] = ctr["FETCH_SIZE"] * 1024
                d["fetch_bytes_x2_guide_correction"] = ctr["FETCH_SIZE"] * 2048
            if "WRITE_SIZE" in ctr:
                d["write_bytes"] = ctr["WRITE_SIZE"] * 1024
        if "SQ_WAVE_CYCLES" in ctr:
            for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_LDS",
                      "SQ_ACTIVE_INST_VALU"):
                if k in ctr:
                    d[k.lower() + "_frac_of_wave_cycles"] = ctr[k] / ctr["SQ_WAVE_CYCLES"]
        out["derived"] = {k: (round(v, 4) if v < 1e6 else round(v, 0)) for k, v in d.items()}
        md += ["", f"## PMC, per launch of `{kname}` ({launches} launches averaged)", "", "| counter | value |", "|---|---|"]
        md += [f"| {k} | {v:,.0f} |" for k, v in sorted(ctr.items())]
        md += ["", "### derived", "", "| quantity | value |", "|---|---|"]
        md += [f"| {k} | {v:,.4f} |" if v < 1e6 else f"| {k} | {v:,.0f} |" for k, v in d.items()]
        md += ["", "Notes: SQ_* counters are summed over all SEs/CUs; GRBM_GUI_ACTIVE is summed over the 8 XCDs; "
               "SQ_WAVE_CYCLES/SQ_WAIT_* count quad-cycles; FETCH_SIZE/WRITE_SIZE are KiB.  "
               "MI355X_MICROARCH.md: FETCH_SIZE under-reports wide coalesced streams by 2x on gfx950 -- both the raw and "
               "the x2 figure are given; the tuple reads here are 16 B/lane at a 128 B stride, not the calibrated pattern."]
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "profiles", a.tag + ".json"), "w"), indent=1)
    open(os.path.join(ROOT, "profiles", a.tag + ".md"), "w").write("\n".join(md) + "\n")
    print("\n".join(md))


if __name__ == "__main__":
    main()
