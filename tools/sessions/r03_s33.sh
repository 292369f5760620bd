#!/bin/bash
# dense level K as shipped (picker: two blocks per CU count one level): config-4 bench line + its rocprofv3 passes, the rank-quantised
# kernels with dense level K on the 255-bin model, the whole GPU suite
set -u
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_s33
rm -rf "$OUT"; mkdir -p "$OUT"
( timeout 900 python bench.py --config 4 ) > $OUT/bench_cfg4.log 2> $OUT/bench_cfg4.err; tail -1 $OUT/bench_cfg4.log | cut -c1-300
for v in sparse_q_k8_u8_t1024 sparse_qd_k8_u8_t1024 sparse_qd_k9_u8_t1024 auto; do
  echo "== bins 255 $v"
  if [ $v = auto ]; then V=""; else V="--variant $v"; fi
  ( timeout 200 python tools/run_shape.py --sparse --trees 512 --levels 16 --features 64 --rows 10000000 --reps 3 --bins 255 $V ) 2>&1 | grep -v "^W\|amdgpu.ids" | tail -1
done | tee $OUT/cfg4_bins.log
( timeout 1200 python -m pytest tests -q -m gpu 2>&1 | grep -v "Extension modules" ) > $OUT/gpu_tests.log; grep -n "passed\|failed" $OUT/gpu_tests.log | tail -2
B="python $GRAFT_REPO_ROOT/bench.py --config 4 --steps 3 --warmup 1 --no-cpu-baseline --no-streamed"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/stats_cfg4 -o bench -- $B ) > $OUT/stats_cfg4.log 2>&1; echo "cfg4 stats rc=$?"
( cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch_cfg4 -o pmc -- $B ) > $OUT/fetch_cfg4.log 2>&1; echo "cfg4 fetch rc=$?"
( cd /tmp && timeout 600 rocprofv3 --pmc WRITE_SIZE -d $OUT/write_cfg4 -o pmc -- $B ) > $OUT/write_cfg4.log 2>&1; echo "cfg4 write rc=$?"
P="python $GRAFT_REPO_ROOT/tools/run_shape.py --sparse --trees 512 --levels 16 --features 64 --rows 4000000 --reps 2"
tools/pmc_session.sh r03_s33/pmc_sparse "$P" \
  "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM" \
  "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE" \
  "TA_BUSY_avr TA_BUFFER_READ_WAVEFRONTS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" | tail -4
