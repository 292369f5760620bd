#!/bin/bash
# where does the per-tile fixed cost of the fp32 tile kernel go?  (shard regime: 125 trees)
# exp1 = tuple loads hit L2 (every tile re-reads tile 0), exp2 = tuple loads are a broadcast of row 0
mkdir -p gpurun_out && cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
L=distributed-decisiontrees_amd/lib
SH=125x8x32x100000000
cp $L/libddt.so /tmp/libddt_orig.so
for x in 1 2; do
  cp $L/libddt_exp$x.so $L/libddt.so
  timeout 300 python tools/sweep.py --shapes $SH --only d8_t1024_r1_c4_u4_dma_f --reps 3 --out $OUT/sweep_exp$x.json > $OUT/exp$x.log 2>&1
  echo "== exp$x"; grep "d8_t1024" $OUT/exp$x.log
done
cp /tmp/libddt_orig.so $L/libddt.so
SW="python tools/sweep.py --shapes 125x8x32x20000000 --only d8_t1024_r1_c4_u4_dma_f --reps 1 --out $OUT/sweep_pmc.json"
( cd /tmp && timeout 300 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum -d $OUT/pmcL1 -o pmc -- bash -c "cd $GRAFT_REPO_ROOT && $SW" ) > $OUT/pmcL1.log 2>&1; echo "pmcL1 rc=$?"
( cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum -d $OUT/pmcL2 -o pmc -- bash -c "cd $GRAFT_REPO_ROOT && $SW" ) > $OUT/pmcL2.log 2>&1; echo "pmcL2 rc=$?"
tail -3 $OUT/pmcL1.log $OUT/pmcL2.log
find $OUT/pmcL1 $OUT/pmcL2 -name "*counter_collection.csv" | head
python - <<'PY'
import csv, glob, collections, os
for d in ("pmcL1", "pmcL2"):
    for f in glob.glob(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", d, "**", "*counter_collection.csv"), recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:60]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[(k, r["Counter_Name"])] += 1
        for k, v in acc.items():
            if "score" in k or "tile" in k:
                print(d, k, {c: (x, cnt[(k, c)]) for c, x in v.items()})
PY
