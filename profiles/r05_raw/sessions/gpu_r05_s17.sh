#!/bin/bash
# Round 5, session 17: counters of the config-4 kernel after the round's last two changes (no gather behind the last visits; dense pair records):
# the same passes as session 2's (profiles/r05_pmc_cfg2_cfg4_cfg6.md section 1) + the instruction mix.
set -u
tag=${1:-r05_s17}
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$tag
rm -rf "$OUT"; mkdir -p "$OUT"
P="python $GRAFT_REPO_ROOT/tools/run_shape.py --sparse --trees 512 --levels 16 --features 64 --rows 4000000 --reps 2"
tools/pmc_session.sh $tag/pmc_sparse "$P" \
  "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" \
  "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_BUSY_avr TA_BUFFER_READ_WAVEFRONTS_sum" \
  "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum FETCH_SIZE" \
  "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU" \
  "SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_WAIT_ANY" | tail -8
python tools/pmc_dump.py $OUT/pmc_sparse/pmc1 $OUT/pmc_sparse/pmc2 $OUT/pmc_sparse/pmc3 $OUT/pmc_sparse/pmc4 $OUT/pmc_sparse/pmc5 > $OUT/pmc_sparse.json 2>/dev/null
head -c 1500 $OUT/pmc_sparse.json
