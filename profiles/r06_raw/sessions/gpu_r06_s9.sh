#!/bin/bash
# Round 6, session 9: why config 6's side leg of the default command lost 3 ms per step (wall, not events) in session 8.
set -u
tag=${1:-r06_s9}
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$tag
rm -rf "$OUT"; mkdir -p "$OUT"
show() { python - "$1" <<'PY'
import json,sys
l=[x for x in open(sys.argv[1]) if x.startswith('{')]
d=json.loads(l[-1])
oc=d.get("other_configs",{})
print(sys.argv[1].split('/')[-1], d["value"], {k:(v.get("value"), v.get("ms_per_step"), v.get("roofline",{}).get("kernel_ms"), v.get("roofline",{}).get("prepass_ms")) for k,v in oc.items() if isinstance(v,dict)})
PY
}
( timeout 600 python bench.py --cpu-seconds 2 --no-streamed ) > $OUT/a_default.log 2>&1; show $OUT/a_default.log
( timeout 600 env DDT_BENCH_KEEP_HEADLINE=1 python bench.py --cpu-seconds 2 --no-streamed ) > $OUT/b_keep.log 2>&1; show $OUT/b_keep.log
( timeout 600 python bench.py --cpu-seconds 2 --no-streamed --no-other-modes ) > $OUT/c_noproxies.log 2>&1; show $OUT/c_noproxies.log
( timeout 600 python bench.py --cpu-seconds 2 --no-streamed ) > $OUT/d_default.log 2>&1; show $OUT/d_default.log
