#!/bin/bash
# Round 6, session 51: counters of rank_kernel (the transposed rank pre-pass) after the short search, on config 6's shape (two parts of ~37 k keys per feature)
set -u
tag=${1:-r06_s51}
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$tag
rm -rf "$OUT"; mkdir -p "$OUT"
CMD="python $GRAFT_REPO_ROOT/tools/run_shape.py --trees 512 --levels 12 --features 32 --rows 10000000 --reps 2"
bash tools/pmc_session.sh $tag/p1 "$CMD" "GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_BUSY_avr" "FETCH_SIZE" "WRITE_SIZE" > $OUT/session_1.log 2>&1
python tools/pmc_dump_kernels.py $OUT/p1/pmc1 $OUT/p1/pmc2 $OUT/p1/pmc3 $OUT/p1/pmc4 $OUT/p1/pmc5 $OUT/p1/pmc6 --like rank > $OUT/counters_rank.txt 2>&1
cut -c1-170 $OUT/counters_rank.txt
python tools/kstats.py $OUT/p1/stats | head -6
find $OUT -name "*.db" -size +3M -delete
