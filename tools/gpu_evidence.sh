#!/bin/bash
# Round evidence on the GPU box (through gpurun): GPU tests, the bench lines (config 3 = headline, config 4 = sparse
# forest), rocprofv3 kernel-trace stats and HBM-traffic PMC passes of the SAME bench commands, smoke().
# Usage: tools/gpu_evidence.sh <tag>    -> gpurun_out/<tag>/...
set -u
tag=${1:-ev}
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$tag
rm -rf "$OUT"; mkdir -p "$OUT"
( timeout 1500 python -m pytest tests -q -m gpu 2>&1 | grep -v "Extension modules" ) > $OUT/gpu_tests.log; grep -n "passed\|failed" $OUT/gpu_tests.log | tail -2
( timeout 300 python __graft_entry__.py smoke ) > $OUT/smoke.log 2>&1; grep "smoke ok" $OUT/smoke.log
( timeout 900 python bench.py ) > $OUT/bench_cfg3.log 2> $OUT/bench_cfg3.err; tail -1 $OUT/bench_cfg3.log | cut -c1-300
( timeout 900 python bench.py --config 4 ) > $OUT/bench_cfg4.log 2> $OUT/bench_cfg4.err; tail -1 $OUT/bench_cfg4.log | cut -c1-300
for cfg in 1 2 5; do
  ( timeout 900 python bench.py --config $cfg ) > $OUT/bench_cfg$cfg.log 2> $OUT/bench_cfg$cfg.err; tail -1 $OUT/bench_cfg$cfg.log | cut -c1-300
done
for cfg in 3 4; do
  B="python $GRAFT_REPO_ROOT/bench.py --config $cfg --steps 3 --warmup 1 --no-cpu-baseline --no-streamed"
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/stats_cfg$cfg -o bench -- $B ) > $OUT/stats_cfg$cfg.log 2>&1; echo "cfg$cfg stats rc=$?"
  ( cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch_cfg$cfg -o pmc -- $B ) > $OUT/fetch_cfg$cfg.log 2>&1; echo "cfg$cfg fetch rc=$?"
  ( cd /tmp && timeout 600 rocprofv3 --pmc WRITE_SIZE -d $OUT/write_cfg$cfg -o pmc -- $B ) > $OUT/write_cfg$cfg.log 2>&1; echo "cfg$cfg write rc=$?"
done
( timeout 600 python bench.py --steps 3 --warmup 1 --force-collectives --no-cpu-baseline --no-streamed ) > $OUT/bench_force_allreduce.log 2>/dev/null
( timeout 600 python bench.py --steps 3 --warmup 1 --force-collectives --combine chain --no-cpu-baseline --no-streamed ) > $OUT/bench_force_chain.log 2>/dev/null
( timeout 600 python bench.py --steps 3 --warmup 1 --trees 125 --force-collectives --no-cpu-baseline --no-streamed ) > $OUT/bench_force_t125.log 2>/dev/null
( timeout 600 python bench.py --steps 3 --warmup 1 --trees 125 --no-cpu-baseline --no-streamed ) > $OUT/bench_t125.log 2>/dev/null
# the tapered tail of the multi-GPU pipeline in a one-rank communicator: what the two extra launches cost before any peer exists
( timeout 600 python bench.py --steps 3 --warmup 1 --trees 125 --force-collectives --taper 1 --no-cpu-baseline --no-streamed ) > $OUT/bench_force_t125_taper.log 2>/dev/null
( timeout 600 python bench.py --steps 3 --warmup 1 --force-collectives --taper 1 --no-cpu-baseline --no-streamed ) > $OUT/bench_force_allreduce_taper.log 2>/dev/null
tail -qn1 $OUT/bench_force_*.log $OUT/bench_t125.log | cut -c1-160
# SQ / LDS / TA counters of the shipped depth-8 kernel (separate passes, 8 M rows)
S="python $GRAFT_REPO_ROOT/tools/sweep.py --shapes 1000x8x32x8000000 --only q16_d8_c8_u4_gl_s2_cm"
tools/pmc_session.sh $tag/pmc_q16 "$S" \
  "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
  "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVES GRBM_GUI_ACTIVE" \
  "TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" | tail -4
