#!/usr/bin/env python3
"""Timeline of the last calls in a rocprofv3 --kernel-trace output (rocpd sqlite): per kernel of a call its duration and the gap to the kernel
before it, averaged over the last `--calls` calls of `--per-call` kernels each.  usage: ktimeline.py <dir> --per-call 3 [--calls 20]"""
import glob
import sqlite3
import sys


def main():
    d = sys.argv[1]
    per = int(sys.argv[sys.argv.index("--per-call") + 1])
    calls = int(sys.argv[sys.argv.index("--calls") + 1]) if "--calls" in sys.argv else 20
    for db in sorted(glob.glob(d + "/**/*.db", recursive=True)):
        cur = sqlite3.connect(db).cursor()
        rows = [r for r in cur.execute("select name,start,end from kernels order by start") if "rocclr" not in r[0] and "synth" not in r[0]]
        rows = rows[-per * calls:]
        for k in range(per):
            sel = rows[k::per]
            dur = sum(r[2] - r[1] for r in sel) / len(sel) / 1e3
            gaps = [rows[i][1] - rows[i - 1][2] for i in range(k, len(rows), per) if i > 0]
            gap = sum(gaps) / max(1, len(gaps)) / 1e3
            print(f"{sel[0][0].split('(')[0][-48:]:50s} {dur:8.2f} us   gap before {gap:8.2f} us")
        span = [rows[i + per - 1][2] - rows[i][1] for i in range(0, len(rows) - per + 1, per)]
        print(f"first kernel's start to last kernel's end: {sum(span) / len(span) / 1e3:.2f} us over {len(span)} calls")


if __name__ == "__main__":
    main()
