#!/bin/bash
# round 3, GPU session 1: microbenchmarks (ceilings), same-box baseline bench, SQ/LDS/TA counters of the shipped q16_gl kernel
set -u
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_s1
rm -rf "$OUT"; mkdir -p "$OUT"
( timeout 300 tools/ubench/ubench all ) > $OUT/ubench.json 2> $OUT/ubench.err; echo "ubench rc=$?"; head -c 6000 $OUT/ubench.json
( timeout 600 python bench.py --no-streamed ) > $OUT/bench_cfg3.log 2> $OUT/bench_cfg3.err; tail -1 $OUT/bench_cfg3.log | cut -c1-400
S="python $GRAFT_REPO_ROOT/tools/sweep.py --shapes 1000x8x32x8000000 --only q16_d8_c8_u4_gl"
tools/pmc_session.sh r03_s1/pmc_q16_gl "$S" \
  "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
  "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVES GRBM_GUI_ACTIVE" \
  "TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum"
