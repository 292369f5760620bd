#!/usr/bin/env python3
"""Print the kernel table (name, calls, average us) of the rocprofv3 --kernel-trace --stats outputs (rocpd sqlite) under a directory.
usage: kstats.py <dir> [--like score]"""
import glob
import sqlite3
import sys


def main():
    d = sys.argv[1]
    like = sys.argv[sys.argv.index("--like") + 1] if "--like" in sys.argv else ""
    for db in sorted(glob.glob(d + "/**/*.db", recursive=True)):
        cur = sqlite3.connect(db).cursor()
        for name, calls, avg in cur.execute("select name,total_calls,average from top_kernels"):
            if like in name and "rocclr" not in name and "synth" not in name:
                print(f"{name.split('(')[0][-60:]:62s} {calls:4d} x {avg / 1e3:9.3f} ms")


if __name__ == "__main__":
    main()
