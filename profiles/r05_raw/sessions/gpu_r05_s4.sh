#!/bin/bash
# Round 5, GPU session 4: the sparse kernels with finished walkers' gathers sent out of the buffer's range (sparse_idle_oob) -- tests, A/B, counters.
set -u
tag=${1:-r05_s4}
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$tag
rm -rf "$OUT"; mkdir -p "$OUT"
( timeout 900 python -m pytest tests/test_sparse.py tests/test_full_size_gpu.py -x -q -m gpu -k "sparse or config4" 2>&1 | grep -v "Extension modules" ) > $OUT/tests_sparse.log; tail -3 $OUT/tests_sparse.log
for rep in 1 2; do
  ( timeout 300 python bench.py --config 4 --no-streamed --no-cpu-baseline --opt sparse_idle_oob=1 ) > $OUT/bench_cfg4_oob1_$rep.log 2>/dev/null; tail -1 $OUT/bench_cfg4_oob1_$rep.log | cut -c1-260
  ( timeout 300 python bench.py --config 4 --no-streamed --no-cpu-baseline --opt sparse_idle_oob=0 ) > $OUT/bench_cfg4_oob0_$rep.log 2>/dev/null; tail -1 $OUT/bench_cfg4_oob0_$rep.log | cut -c1-260
done
( timeout 300 python bench.py --config 4 --no-streamed ) > $OUT/bench_cfg4.log 2> $OUT/bench_cfg4.err; tail -1 $OUT/bench_cfg4.log | cut -c1-300
python tools/sparse_sweep.py --help > /dev/null 2>&1
P="python $GRAFT_REPO_ROOT/tools/run_shape.py --sparse --trees 512 --levels 16 --features 64 --rows 4000000 --reps 2"
tools/pmc_session.sh $tag/pmc_sparse "$P" \
  "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" \
  "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_BUSY_avr TA_BUFFER_READ_WAVEFRONTS_sum" \
  "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES" | tail -3
python tools/pmc_dump.py $OUT/pmc_sparse/pmc1 $OUT/pmc_sparse/pmc2 $OUT/pmc_sparse/pmc3 > $OUT/pmc_sparse_oob1.json 2>/dev/null
rm -rf $OUT/pmc_sparse/pmc*/ $OUT/pmc_sparse/stats 2>/dev/null
# other sparse shapes: the sklearn-like forest of the tests is covered by pytest; K sweep of the synthetic forest
( timeout 600 python tools/sparse_sweep.py --rows 4000000 --only sparse_dk_k,sparse_k8_u8_t512,sparse_k7_u8_t256 --out $OUT/sparse_sweep.json ) > $OUT/sparse_sweep.log 2>&1; tail -15 $OUT/sparse_sweep.log
