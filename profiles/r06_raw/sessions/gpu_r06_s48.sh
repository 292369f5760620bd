#!/bin/bash
# session r06_s48: rank tables of up to 37727 keys (one block's LDS in rank_kernel; 32767 before): config 6 in two parts instead of three.
# GPU suite, config 6 A/B against the old limit on one box (alternating), the other configs' lines (nothing else may move), graph-replay latency probe
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06_s48; mkdir -p $O
( timeout 1500 python -m pytest tests -q -m gpu -x -p no:cacheprovider 2>&1 | grep -v "Extension modules" ) > $O/gpu_tests.log; grep -n "passed\|failed" $O/gpu_tests.log | tail -2
for i in 1 2; do
  ( timeout 600 python bench.py --config 6 --no-cpu-baseline --no-streamed ) > $O/cfg6_new_$i.log 2> $O/cfg6_new_$i.err; tail -1 $O/cfg6_new_$i.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg6 new', d['value'], d['ms_per_step'], d['roofline']['scoring_launches_per_step'], d['roofline']['prepass_ms'], d['roofline']['kernel_ms'])"
  ( timeout 600 python bench.py --config 6 --no-cpu-baseline --no-streamed --opt q16_max_table=32767 ) > $O/cfg6_old_$i.log 2> $O/cfg6_old_$i.err; tail -1 $O/cfg6_old_$i.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg6 old', d['value'], d['ms_per_step'], d['roofline']['scoring_launches_per_step'], d['roofline']['prepass_ms'], d['roofline']['kernel_ms'])"
done
( timeout 900 python bench.py ) > $O/bench_cfg3.log 2> $O/bench_cfg3.err; tail -1 $O/bench_cfg3.log | cut -c1-200
for cfg in 3 6 4 2; do
  ( timeout 300 python tools/latency_probe.py --configs $cfg --rows 1,1024,16384,131072 --reps 30 --graph ) > $O/graph_cfg$cfg.log 2>&1; cut -c1-400 $O/graph_cfg$cfg.log | tail -4
done
( timeout 300 python tools/latency_probe.py --configs 5 --rows 1024,16384 --reps 30 --graph --no-check ) > $O/graph_cfg5.log 2>&1; cut -c1-400 $O/graph_cfg5.log | tail -2
timeout 400 python tools/soak_fuzz.py --seed 404 --seconds 300 > $O/soak_404.log 2>&1; echo "soak rc=$?"; tail -1 $O/soak_404.log | cut -c1-300
