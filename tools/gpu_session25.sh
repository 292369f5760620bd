#!/bin/bash
mkdir -p gpurun_out && cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
rm -rf $OUT/pmcp1
SW="python $GRAFT_REPO_ROOT/tools/sweep.py --shapes 1000x8x32x8000000 --only q16_d8 --reps 2 --out /tmp/sw.json"
( cd /tmp && timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE -d $OUT/pmcp1 -o pmc -- $SW ) > $OUT/pmcp1.log 2>&1; echo "pmc rc=$?"
python - <<'PY'
import sqlite3, glob, os
for db in glob.glob(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "pmcp1", "**", "*.db"), recursive=True):
    cur = sqlite3.connect(db).cursor()
    for kn, cn, v, n in cur.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection where kernel_name like '%score_q16%' group by 1,2"):
        print(cn, v / n, n)
PY
