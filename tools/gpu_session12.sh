#!/bin/bash
# quad-coalesced tuple loads (+ persistent form): parity, then shard-regime sweep
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/s12_tests.log 2>&1
tail -5 gpurun_out/s12_tests.log
timeout 600 python tools/sweep.py --shapes 125x8x32x100000000,1000x8x32x100000000 --only d8_t1024_r1_c4_u4_dma_f --reps 3 --out gpurun_out/sweep_r.json > gpurun_out/s12_sweep.log 2>&1
timeout 300 python tools/sweep.py --shapes 100x6x28x10000000 --only d6_ --reps 5 --out gpurun_out/sweep_r6.json >> gpurun_out/s12_sweep.log 2>&1
timeout 300 python tools/sweep.py --shapes 64x4x32x10000000,300x8x20x10000000 --only _dma --reps 5 --out gpurun_out/sweep_r4.json >> gpurun_out/s12_sweep.log 2>&1
grep -v "^/opt" gpurun_out/s12_sweep.log | tail -40
