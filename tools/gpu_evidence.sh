#!/bin/bash
# round evidence: GPU tests, bench line, rocprofv3 kernel-trace stats and HBM-traffic PMC passes of the SAME bench command
mkdir -p gpurun_out && cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
rm -rf $OUT/ev_stats $OUT/ev_fetch $OUT/ev_write
( timeout 1500 python -m pytest tests -q -m gpu 2>&1 | grep -v "Extension modules" ) > $OUT/ev_tests.log; tail -2 $OUT/ev_tests.log
( timeout 900 python bench.py ) > $OUT/ev_bench.log 2>&1; tail -1 $OUT/ev_bench.log | cut -c1-400
B="python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/ev_stats -o bench -- $B ) > $OUT/ev_stats.log 2>&1; echo "stats rc=$?"
( cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE -d $OUT/ev_fetch -o pmc -- $B ) > $OUT/ev_fetch.log 2>&1; echo "fetch rc=$?"
( cd /tmp && timeout 600 rocprofv3 --pmc WRITE_SIZE -d $OUT/ev_write -o pmc -- $B ) > $OUT/ev_write.log 2>&1; echo "write rc=$?"
ls -la $OUT/ev_stats $OUT/ev_fetch $OUT/ev_write 2>/dev/null | head -20
