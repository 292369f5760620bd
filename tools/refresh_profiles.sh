#!/bin/bash
# Copy the judged summaries of an evidence run (tools/gpu_evidence.sh <tag>, + bench_cfg{1,2,5}.log) from gpurun_out/<tag>/ into
# profiles/ (round-2 names) and regenerate the PMC traffic files.  Usage: tools/refresh_profiles.sh <tag>
set -eu
cd "$(dirname "$0")/.."
E=gpurun_out/$1
for f in bench_cfg1 bench_cfg2 bench_cfg3 bench_cfg4 bench_cfg5 bench_force_allreduce bench_force_chain bench_force_t125 bench_t125 gpu_tests smoke; do
  [ -f $E/$f.log ] && cp $E/$f.log profiles/r02_$f.log
done
python tools/update_pmc_traffic.py $E
python tools/prof_summary.py r02_final_bench_cfg3 --stats $E/stats_cfg3 --pmc $E/fetch_cfg3 $E/write_cfg3 --kernel score_q16 --rows 100000000 --trees 1000 \
  --levels 8 --features 32 --cmd "python bench.py --config 3 --steps 3 --warmup 1 --no-cpu-baseline --no-streamed" > /dev/null
python tools/prof_summary.py r02_final_bench_cfg4 --stats $E/stats_cfg4 --pmc $E/fetch_cfg4 $E/write_cfg4 --kernel score_sparse --rows 10000000 --trees 512 \
  --levels 12 --features 64 --cmd "python bench.py --config 4 --steps 3 --warmup 1 --no-cpu-baseline --no-streamed" > /dev/null
grep "score_q16\|rank_kernel\|transpose_k\|score_sparse" profiles/r02_final_bench_cfg3.md profiles/r02_final_bench_cfg4.md | cut -c1-200 | head
