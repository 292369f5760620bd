#!/bin/bash
# session r06_s49: rank_kernel / rank32_kernel with byte-address probes on linear tables (compare + select + add per probe; 8 VALU before), tables of
# up to 38848 keys.  GPU suite, config 6 (new limit / the old one), its kernel stats, config 4 (rank32_kernel), the compacted-feature shape, a soak.
set -u
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06_s49; mkdir -p $O
( timeout 1500 python -m pytest tests -q -m gpu -x -p no:cacheprovider 2>&1 | grep -v "Extension modules" ) > $O/gpu_tests.log; grep -n "passed\|failed" $O/gpu_tests.log | tail -2
P='import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(sys.argv[1], d["value"], d["ms_per_step"], r["scoring_launches_per_step"], r["prepass_ms"], r["kernel_ms"], r["kernel"])'
for i in 1 2; do
  ( timeout 600 python bench.py --config 6 --no-cpu-baseline --no-streamed ) > $O/cfg6_new_$i.log 2> $O/cfg6_new_$i.err; tail -1 $O/cfg6_new_$i.log | python -c "$P" cfg6_new
  ( timeout 600 python bench.py --config 6 --no-cpu-baseline --no-streamed --opt q16_max_table=32767 ) > $O/cfg6_old_$i.log 2> $O/cfg6_old_$i.err; tail -1 $O/cfg6_old_$i.log | python -c "$P" cfg6_oldlimit
  ( timeout 600 python bench.py --config 4 --no-cpu-baseline --no-streamed ) > $O/cfg4_$i.log 2> $O/cfg4_$i.err; tail -1 $O/cfg4_$i.log | python -c "$P" cfg4
done
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/stats_cfg6 -o bench -- python $GRAFT_REPO_ROOT/bench.py --config 6 --steps 3 --warmup 1 --no-cpu-baseline --no-streamed ) > $O/stats_cfg6.log 2>&1; echo "stats rc=$?"
python tools/kstats.py $O/stats_cfg6 2>/dev/null | head -8
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/stats_cfg4 -o bench -- python $GRAFT_REPO_ROOT/bench.py --config 4 --steps 3 --warmup 1 --no-cpu-baseline --no-streamed ) > $O/stats_cfg4.log 2>&1; echo "stats rc=$?"
python tools/kstats.py $O/stats_cfg4 2>/dev/null | head -8
( timeout 300 python tools/run_shape.py --trees 512 --levels 12 --features 60 --wide-features 200 --rows 4000000 ) > $O/wide_compaction.log 2>&1; tail -2 $O/wide_compaction.log
( timeout 300 python tools/run_shape.py --trees 4000 --levels 8 --features 16 --rows 4000000 ) > $O/t4000.log 2>&1; tail -2 $O/t4000.log
( timeout 900 python bench.py ) > $O/bench_cfg3.log 2> $O/bench_cfg3.err; tail -1 $O/bench_cfg3.log | cut -c1-200
timeout 400 python tools/soak_fuzz.py --seed 505 --seconds 300 > $O/soak_505.log 2>&1; echo "soak rc=$?"; tail -1 $O/soak_505.log | cut -c1-200
