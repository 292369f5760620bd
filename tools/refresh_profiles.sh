#!/bin/bash
# Copy the judged summaries of an evidence run (tools/gpu_evidence.sh <tag>, + bench_cfg{1,2,5}.log) from gpurun_out/<tag>/ into
# profiles/ (names prefixed with the round, default r04) and regenerate the PMC traffic files.  Usage: tools/refresh_profiles.sh <tag> [rNN]
set -eu
cd "$(dirname "$0")/.."
E=gpurun_out/$1
R=${2:-r04}
for f in bench_cfg1 bench_cfg2 bench_cfg3 bench_cfg4 bench_cfg5 bench_cfg6 bench_cfg6_shard_of_8 bench_force_allreduce bench_force_chain bench_force_t125 bench_force_hybrid bench_shard_of_8 bench_t125 gpu_tests smoke; do
  [ -f $E/$f.log ] && cp $E/$f.log profiles/${R}_$f.log || true
done
python tools/update_pmc_traffic.py $E
python tools/prof_summary.py ${R}_final_bench_cfg3 --stats $E/stats_cfg3 --pmc $E/fetch_cfg3 $E/write_cfg3 --kernel score_q16 --rows 100000000 --trees 1000 \
  --levels 8 --features 32 --cmd "python bench.py --config 3 --steps 3 --warmup 1 --no-cpu-baseline --no-streamed" > /dev/null
python tools/prof_summary.py ${R}_final_bench_cfg4 --stats $E/stats_cfg4 --pmc $E/fetch_cfg4 $E/write_cfg4 --kernel score_sparse --rows 10000000 --trees 512 \
  --levels 12 --features 64 --cmd "python bench.py --config 4 --steps 3 --warmup 1 --no-cpu-baseline --no-streamed" > /dev/null
if [ -d $E/stats_cfg6 ]; then
  python tools/prof_summary.py ${R}_final_bench_cfg6 --stats $E/stats_cfg6 --pmc $E/fetch_cfg6 $E/write_cfg6 --kernel score_q16d --rows 10000000 --trees 512 \
    --levels 12 --features 32 --cmd "python bench.py --config 6 --steps 3 --warmup 1 --no-cpu-baseline --no-streamed" > /dev/null
fi
if [ -d $E/stats_cfg1 ]; then
  python tools/prof_summary.py ${R}_final_bench_cfg1 --stats $E/stats_cfg1 --pmc $E/fetch_cfg1 $E/write_cfg1 --kernel score_stream --rows 200000000 --trees 8 \
    --levels 4 --features 16 --cmd "python bench.py --config 1 --steps 10 --warmup 3 --no-cpu-baseline --no-streamed" > /dev/null
fi
if [ -d $E/stats_cfg2 ]; then
  python tools/prof_summary.py ${R}_final_bench_cfg2 --stats $E/stats_cfg2 --pmc $E/fetch_cfg2 $E/write_cfg2 --kernel score_q16 --rows 10000000 --trees 100 \
    --levels 6 --features 28 --cmd "python bench.py --config 2 --steps 10 --warmup 3 --no-cpu-baseline --no-streamed" > /dev/null
fi
if [ -d $E/stats_cfg5 ]; then
  python tools/prof_summary.py ${R}_final_bench_cfg5 --stats $E/stats_cfg5 --pmc $E/fetch_cfg5 $E/write_cfg5 --kernel score_q16p --rows 10000000 --trees 1000 \
    --levels 8 --features 32 --cmd "python bench.py --config 5 --steps 3 --warmup 1 --no-cpu-baseline --no-streamed" > /dev/null
fi
grep "score_q16\|rank_kernel\|transpose_k\|score_sparse" profiles/${R}_final_bench_cfg3.md profiles/${R}_final_bench_cfg4.md | cut -c1-200 | head
if [ -d $E/pmc_q16 ]; then
  python tools/prof_summary.py ${R}_pmc_q16_x --stats $E/pmc_q16/stats --pmc $E/pmc_q16/pmc1 $E/pmc_q16/pmc2 $E/pmc_q16/pmc3 --kernel score_q16 --rows 8000000 \
    --cmd "rocprofv3 --pmc <set> -- python tools/sweep.py --shapes 1000x8x32x8000000 --only q16_d8_c8_u4_gl_s2_cm_x (3 separate passes)" > /dev/null
fi
