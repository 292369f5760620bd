#!/bin/bash
# Round 6, session 28: counters of the FINAL sparse_r scoring kernel (lagged walk, wave-private rows) and its pre-pass on BASELINE config 4's forest (4 M tuples per launch), beside the fp32 pair-record kernel.
set -u
tag=${1:-r06_s28}
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$tag
rm -rf "$OUT"; mkdir -p "$OUT"
for o in 1; do
  CMD="python $GRAFT_REPO_ROOT/tools/run_shape.py --sparse --trees 512 --levels 16 --features 64 --rows 4000000 --reps 2 --opt sparse_r32=$o"
  bash tools/pmc_session.sh $tag/r32_$o "$CMD" "TA_BUSY_avr TA_BUFFER_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_WAVES" "FETCH_SIZE" "WRITE_SIZE" > $OUT/session_$o.log 2>&1
  python tools/pmc_dump_kernels.py $OUT/r32_$o/pmc1 $OUT/r32_$o/pmc2 $OUT/r32_$o/pmc3 $OUT/r32_$o/pmc4 $OUT/r32_$o/pmc5 $OUT/r32_$o/pmc6 $OUT/r32_$o/pmc7 > $OUT/counters_r32_$o.txt 2>&1
  grep -v synth $OUT/counters_r32_$o.txt | cut -c1-200
done
find $OUT -name "*.db" -size +3M -delete
