#!/bin/bash
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
for mode in allreduce chain; do
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --force-collectives --combine $mode > gpurun_out/s32_$mode.log 2>&1
tail -1 gpurun_out/s32_$mode.log | cut -c1-260
done
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --force-collectives --trees 125 > gpurun_out/s32_t125.log 2>&1; tail -1 gpurun_out/s32_t125.log | cut -c1-260
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --trees 125 > gpurun_out/s32_t125_plain.log 2>&1; tail -1 gpurun_out/s32_t125_plain.log | cut -c1-260
