#!/bin/bash
# config 1: which queue does the 4-byte result store saturate?  L2 <-> fabric (EA) counters of the stream kernel's
# memory pattern with and without the store, the store's cache-policy bits, and the shipped stream kernel itself.
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_s10; rm -rf "$OUT"; mkdir -p "$OUT"
( cd /tmp && timeout 120 rocprofv3 -L ) > $OUT/avail.txt 2>&1 || ( cd /tmp && timeout 120 rocprofv3 --list-avail ) > $OUT/avail.txt 2>&1
grep -o "TCC_[A-Za-z0-9_]*" $OUT/avail.txt | sort -u > $OUT/avail_tcc.txt; wc -l $OUT/avail_tcc.txt
U="$GRAFT_REPO_ROOT/tools/ubench/ubench storepol"
( timeout 120 $U ) > $OUT/storepol.json 2> $OUT/storepol.err; grep result_store $OUT/storepol.json | cut -c1-200
WISH="TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_TAG_STALL_sum TCC_BUSY_sum TCC_CYCLE_sum
 TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_GMI_CREDIT_STALL_sum TCC_EA0_WRREQ_IO_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum
 TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_GMI_CREDIT_STALL_sum TCC_EA0_RDREQ_IO_CREDIT_STALL_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_sum
 TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum TCC_HIT_sum TCC_MISS_sum TCC_WRITEBACK_sum TCC_NORMAL_WRITEBACK_sum TCC_NORMAL_EVICT_sum TCC_EA0_WR_UNCACHED_32B_sum TCC_STREAMING_REQ_sum TCC_NC_REQ_sum TCC_UC_REQ_sum TCC_CC_REQ_sum TCC_RW_REQ_sum TCC_PROBE_sum
 TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_WRITE_TAGCONFLICT_STALL_CYCLES_sum GRBM_GUI_ACTIVE"
python tools/pick_counters.py $OUT/avail.txt 4 $WISH > $OUT/passes.txt 2> $OUT/passes.err; cat $OUT/passes.err; wc -l $OUT/passes.txt
i=0
while read -r ctrs; do
  i=$((i + 1))
  ( cd /tmp && timeout 300 rocprofv3 --pmc $ctrs -d $OUT/u_pmc$i -o pmc -- $U ) > $OUT/u_pmc$i.log 2>&1; echo "ubench pass $i ($ctrs) rc=$?"
done < $OUT/passes.txt
S="python $GRAFT_REPO_ROOT/tools/run_shape.py --trees 8 --levels 4 --features 16 --rows 200000000 --reps 3"
( timeout 300 $S ) > $OUT/stream_plain.log 2>&1; tail -2 $OUT/stream_plain.log
i=0
while read -r ctrs; do
  i=$((i + 1))
  ( cd /tmp && timeout 300 rocprofv3 --pmc $ctrs -d $OUT/s_pmc$i -o pmc -- $S ) > $OUT/s_pmc$i.log 2>&1; echo "stream pass $i rc=$?"
done < $OUT/passes.txt
python tools/pmc_dump.py $OUT/u_pmc* > $OUT/ubench_counters.json
python tools/pmc_dump.py $OUT/s_pmc* > $OUT/stream_counters.json
find $OUT -name "*.db" -delete
du -sh $OUT
