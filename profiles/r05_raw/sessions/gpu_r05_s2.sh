#!/bin/bash
# Round 5, GPU session 2: the full GPU suite; A/B of the deep kernels against the one-block q16_d9 / q16_d10 forms they replace; the wide kernels
# (33..64 words) against the fp32 tile kernels / the generic kernel; counters: TCC hit rate of config 4's gathers, pipes of config 2's kernel.
set -u
tag=${1:-r05_s2}
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$tag
rm -rf "$OUT"; mkdir -p "$OUT"
( timeout 1800 python -m pytest tests -q -m gpu 2>&1 | grep -v "Extension modules" ) > $OUT/gpu_tests.log; tail -4 $OUT/gpu_tests.log
S="python tools/sweep.py --reps 3"
( timeout 300 $S --shapes 256x9x32x4000000,1000x9x32x4000000 --only q16_d9_c4_u4,q16d_d9 --out $OUT/ab_d9.json ) > $OUT/ab_d9.log 2>&1; tail -6 $OUT/ab_d9.log
( timeout 300 $S --shapes 256x10x32x4000000,1000x10x32x4000000 --only q16_d10_c4_u4,q16d_d10 --out $OUT/ab_d10.json ) > $OUT/ab_d10.log 2>&1; tail -6 $OUT/ab_d10.log
( timeout 300 $S --shapes 1000x8x64x20000000,1000x8x33x20000000,1000x8x48x20000000 --only q16w_d8,d8_t512,d8_t256 --out $OUT/ab_wide_d8.json ) > $OUT/ab_wide_d8.log 2>&1; tail -12 $OUT/ab_wide_d8.log
( timeout 300 $S --shapes 300x6x40x10000000,300x6x64x10000000 --only q16w_d6,d6_t512,d6_t256 --out $OUT/ab_wide_d6.json ) > $OUT/ab_wide_d6.log 2>&1; tail -8 $OUT/ab_wide_d6.log
( timeout 300 $S --shapes 512x12x64x4000000,300x9x64x4000000,300x10x40x4000000 --only q16dw,generic --out $OUT/ab_wide_deep.json ) > $OUT/ab_wide_deep.log 2>&1; tail -10 $OUT/ab_wide_deep.log
# counters
P="python $GRAFT_REPO_ROOT/tools/run_shape.py --sparse --trees 512 --levels 16 --features 64 --rows 4000000 --reps 2"
tools/pmc_session.sh $tag/pmc_sparse "$P" \
  "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" \
  "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_BUSY_avr TA_BUFFER_READ_WAVEFRONTS_sum" \
  "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum FETCH_SIZE" | tail -4
python tools/pmc_dump.py $OUT/pmc_sparse/pmc1 $OUT/pmc_sparse/pmc2 $OUT/pmc_sparse/pmc3 > $OUT/pmc_sparse.json 2>/dev/null
C2="python $GRAFT_REPO_ROOT/tools/sweep.py --reps 3 --shapes 100x6x28x10000000 --only q16_d6_c16_u4_s2 --out /tmp/x.json"
tools/pmc_session.sh $tag/pmc_cfg2 "$C2" \
  "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
  "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVES GRBM_GUI_ACTIVE" | tail -3
python tools/pmc_dump.py $OUT/pmc_cfg2/pmc1 $OUT/pmc_cfg2/pmc2 > $OUT/pmc_cfg2.json 2>/dev/null
C6="python $GRAFT_REPO_ROOT/tools/sweep.py --reps 3 --shapes 512x12x32x10000000 --only q16d_d12 --out /tmp/y.json"
tools/pmc_session.sh $tag/pmc_cfg6 "$C6" \
  "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
  "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVES GRBM_GUI_ACTIVE" \
  "TA_BUSY_avr TA_BUFFER_READ_WAVEFRONTS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" \
  "FETCH_SIZE" "WRITE_SIZE" | tail -3
python tools/pmc_dump.py $OUT/pmc_cfg6/pmc1 $OUT/pmc_cfg6/pmc2 $OUT/pmc_cfg6/pmc3 $OUT/pmc_cfg6/pmc4 $OUT/pmc_cfg6/pmc5 > $OUT/pmc_cfg6.json 2>/dev/null
( timeout 300 python bench.py --config 6 --no-streamed ) > $OUT/bench_cfg6.log 2> $OUT/bench_cfg6.err; tail -1 $OUT/bench_cfg6.log | cut -c1-300
rm -rf $OUT/pmc_*/pmc*/ $OUT/pmc_*/stats/*.db 2>/dev/null; du -sh $OUT | tail -1
