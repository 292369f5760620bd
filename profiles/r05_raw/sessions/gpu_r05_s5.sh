#!/bin/bash
# Round 5, GPU session 5: sparse forests with dense mid levels -- tests, A/B per M, counters.
set -u
tag=${1:-r05_s5}
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$tag
rm -rf "$OUT"; mkdir -p "$OUT"
( timeout 900 python -m pytest tests/test_sparse_dm.py tests/test_sparse.py -x -q -m gpu 2>&1 | grep -v "Extension modules" ) > $OUT/tests_sparse.log; tail -3 $OUT/tests_sparse.log
for rep in 1 2; do
  for dm in 0 1 2 3; do
    ( timeout 300 python bench.py --config 4 --no-streamed --no-cpu-baseline --opt sparse_dm=$dm ) > $OUT/bench_cfg4_dm${dm}_$rep.log 2>/dev/null; tail -1 $OUT/bench_cfg4_dm${dm}_$rep.log | cut -c1-240
  done
done
( timeout 300 python bench.py --config 4 --no-streamed ) > $OUT/bench_cfg4.log 2> $OUT/bench_cfg4.err; tail -1 $OUT/bench_cfg4.log | cut -c1-300
P="python $GRAFT_REPO_ROOT/tools/run_shape.py --sparse --trees 512 --levels 16 --features 64 --rows 4000000 --reps 2"
tools/pmc_session.sh $tag/pmc_sparse "$P" \
  "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" \
  "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_BUSY_avr TA_BUFFER_READ_WAVEFRONTS_sum" \
  "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES" | tail -3
python tools/pmc_dump.py $OUT/pmc_sparse/pmc1 $OUT/pmc_sparse/pmc2 $OUT/pmc_sparse/pmc3 > $OUT/pmc_sparse_dm.json 2>/dev/null
rm -rf $OUT/pmc_sparse/pmc*/ $OUT/pmc_sparse/stats 2>/dev/null
( timeout 600 python -m pytest tests/test_full_size_gpu.py -x -q -m gpu -k "config4" 2>&1 | grep -v "Extension modules" ) > $OUT/tests_full4.log; tail -2 $OUT/tests_full4.log
