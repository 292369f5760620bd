#!/usr/bin/env python3
"""rocprofv3 --pmc result databases under the given directories -> one JSON {kernel: {counter: average per launch, ...}} on stdout
(run on the GPU box, so that only the summary has to travel back)."""
import glob
import json
import os
import sqlite3
import sys

out = {}
for d in sys.argv[1:]:
    for db in sorted(glob.glob(os.path.join(d, "**", "*.db"), recursive=True)):
        cur = sqlite3.connect(db).cursor()
        q = ("select kernel_name, counter_name, sum(value), count(distinct dispatch_id), avg(duration) from counters_collection "
             "group by kernel_name, counter_name")
        for kn, cn, v, n, dur in cur.execute(q):
            k = out.setdefault(kn, {})
            k[cn] = v / max(1, n)
            k["launches"] = n
            k.setdefault("avg_ns_under_pmc", {})[cn] = dur
json.dump(out, sys.stdout, indent=1)
