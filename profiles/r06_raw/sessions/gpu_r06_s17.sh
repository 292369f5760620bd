#!/bin/bash
# Round 6, session 15: what bounds the LDS-resident rank pre-passes (grouped_rank_kernel at 1000 trees, fused_rank_kernel at 125 trees and on config 2).
set -u
tag=${1:-r06_s17}
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$tag
rm -rf "$OUT"; mkdir -p "$OUT"
i=0
for shape in "--trees 1000 --levels 8 --features 32 --rows 20000000" "--trees 125 --levels 8 --features 32 --rows 20000000" "--trees 100 --levels 6 --features 28 --rows 10000000"; do
  i=$((i + 1))
  CMD="python $GRAFT_REPO_ROOT/tools/run_shape.py $shape --reps 2"
  bash tools/pmc_session.sh $tag/p$i "$CMD" "GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_BUSY_avr" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "FETCH_SIZE" "WRITE_SIZE" > $OUT/session_$i.log 2>&1
  python tools/pmc_dump_kernels.py $OUT/p$i/pmc1 $OUT/p$i/pmc2 $OUT/p$i/pmc3 $OUT/p$i/pmc4 $OUT/p$i/pmc5 $OUT/p$i/pmc6 $OUT/p$i/pmc7 --like rank > $OUT/counters_$i.txt 2>&1
  echo "=== $shape"; cut -c1-150 $OUT/counters_$i.txt
done
find $OUT -name "*.db" -size +3M -delete
