#!/bin/bash
# config 1, second look: result stores that do not pass the TA -> TCP write path (no-return atomic swap in the L2; scalar stores)
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_s11; rm -rf "$OUT"; mkdir -p "$OUT"
U="$GRAFT_REPO_ROOT/tools/ubench/ubench storepol"
( timeout 120 $U ) > $OUT/storepol.json 2> $OUT/storepol.err; grep result_store $OUT/storepol.json | cut -c1-260
i=0
for ctrs in "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum" "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_CYCLE_sum TCC_ATOMIC_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_LEVEL_sum TCC_WRITE_sum"; do
  i=$((i + 1))
  ( cd /tmp && timeout 300 rocprofv3 --pmc $ctrs -d $OUT/u_pmc$i -o pmc -- $U ) > $OUT/u_pmc$i.log 2>&1; echo "ubench pass $i ($ctrs) rc=$?"
done
python tools/pmc_dump.py $OUT/u_pmc* > $OUT/ubench_counters.json
find $OUT -name "*.db" -delete
