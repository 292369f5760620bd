#!/bin/bash
# Round evidence on the GPU box (through gpurun), trimmed to ~15 GPU-minutes: GPU tests, smoke, the five bench lines (config 3 and 4
# in full, 1 / 2 / 5 without the CPU and host-feeder legs), rocprofv3 kernel-trace stats + FETCH / WRITE passes of the config-3 and
# config-4 bench commands, the shard-regime lines, SQ / LDS / TA counters of the shipped depth-8 kernel and of the sparse kernel.
# Usage: tools/gpu_evidence_slim.sh <tag>    -> gpurun_out/<tag>/...   (then tools/refresh_profiles.sh <tag>)
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
  ( timeout 600 python bench.py --config $cfg --no-cpu-baseline --no-streamed ) > $OUT/bench_cfg$cfg.log 2> $OUT/bench_cfg$cfg.err; tail -1 $OUT/bench_cfg$cfg.log | cut -c1-300
done
for cfg in 3 4 1; do
  B="python $GRAFT_REPO_ROOT/bench.py --config $cfg --steps 3 --warmup 1 --no-cpu-baseline --no-streamed"
  [ $cfg = 1 ] && B="python $GRAFT_REPO_ROOT/bench.py --config 1 --steps 10 --warmup 3 --no-cpu-baseline --no-streamed"
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/stats_cfg$cfg -o bench -- $B ) > $OUT/stats_cfg$cfg.log 2>&1; echo "cfg$cfg stats rc=$?"
  ( cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch_cfg$cfg -o pmc -- $B ) > $OUT/fetch_cfg$cfg.log 2>&1; echo "cfg$cfg fetch rc=$?"
  ( cd /tmp && timeout 600 rocprofv3 --pmc WRITE_SIZE -d $OUT/write_cfg$cfg -o pmc -- $B ) > $OUT/write_cfg$cfg.log 2>&1; echo "cfg$cfg write rc=$?"
done
( timeout 600 python bench.py --steps 3 --warmup 1 --force-collectives --no-cpu-baseline --no-streamed ) > $OUT/bench_force_allreduce.log 2>/dev/null
( timeout 600 python bench.py --steps 3 --warmup 1 --force-collectives --combine chain --no-cpu-baseline --no-streamed ) > $OUT/bench_force_chain.log 2>/dev/null
( timeout 600 python bench.py --steps 3 --warmup 1 --trees 125 --force-collectives --no-cpu-baseline --no-streamed ) > $OUT/bench_force_t125.log 2>/dev/null
( timeout 600 python bench.py --steps 3 --warmup 1 --trees 125 --no-cpu-baseline --no-streamed ) > $OUT/bench_t125.log 2>/dev/null
( timeout 600 python bench.py --steps 3 --warmup 1 --shard-of 8 --no-cpu-baseline --no-streamed ) > $OUT/bench_shard_of_8.log 2>/dev/null
( timeout 600 python bench.py --steps 3 --warmup 1 --force-collectives --shard hybrid --tree-ranks 1 --no-cpu-baseline --no-streamed ) > $OUT/bench_force_hybrid.log 2>/dev/null
tail -qn1 $OUT/bench_force_*.log $OUT/bench_t125.log $OUT/bench_shard_of_8.log | cut -c1-160
S="python $GRAFT_REPO_ROOT/tools/sweep.py --shapes 1000x8x32x8000000 --only q16_d8_c8_u4_gl_s2_cm_x"
tools/pmc_session.sh $tag/pmc_q16 "$S" \
  "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
  "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVES GRBM_GUI_ACTIVE" \
  "TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" | tail -4
P="python $GRAFT_REPO_ROOT/tools/run_shape.py --sparse --trees 512 --levels 16 --features 64 --rows 4000000 --reps 2"
tools/pmc_session.sh $tag/pmc_sparse "$P" \
  "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM" \
  "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE" \
  "TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum TA_BUFFER_READ_WAVEFRONTS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" | tail -4
