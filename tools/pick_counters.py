#!/usr/bin/env python3
"""Pick, from a wish list, the counters this box's rocprofv3 offers, and print them in passes of at most N (one line per pass).
Usage: rocprofv3 -L > avail.txt; tools/pick_counters.py avail.txt 3 CTR_A CTR_B ..."""
import re
import sys

avail = open(sys.argv[1], errors="replace").read()
names = set(re.findall(r"\b[A-Z][A-Za-z0-9_]{3,}\b", avail))
per = int(sys.argv[2])
have = [c for c in sys.argv[3:] if c in names]
missing = [c for c in sys.argv[3:] if c not in names]
if missing:
    print("not offered here: " + " ".join(missing), file=sys.stderr)
for i in range(0, len(have), per):
    print(" ".join(have[i:i + per]))
