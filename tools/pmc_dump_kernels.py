#!/usr/bin/env python3
"""Per kernel and counter: the average value per launch over the rocprofv3 --pmc passes under the given directories (rocpd sqlite).
usage: pmc_dump_kernels.py <dir> [<dir> ...] [--like score_sparse]"""
import glob
import sqlite3
import sys


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    like = sys.argv[sys.argv.index("--like") + 1] if "--like" in sys.argv else ""
    args = [a for a in args if a != like]
    rows = {}
    for d in args:
        for db in glob.glob(d + "/**/*.db", recursive=True):
            cur = sqlite3.connect(db).cursor()
            q = "select kernel_name, counter_name, sum(value), count(distinct dispatch_id) from counters_collection where kernel_name like ? group by kernel_name, counter_name"
            for kn, cn, v, n in cur.execute(q, (f"%{like}%",)):
                rows[(kn.split("(")[0][-70:], cn)] = (v / max(1, n), n)
    for (kn, cn), (v, n) in sorted(rows.items()):
        print(f"{kn:72s} {cn:34s} {v:20,.0f}  ({n} launches)")


if __name__ == "__main__":
    main()
