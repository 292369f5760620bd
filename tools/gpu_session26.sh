#!/bin/bash
mkdir -p gpurun_out
timeout 900 python tools/sweep.py --shapes 300x4x32x20000000,300x6x28x20000000,1000x6x32x20000000,1000x4x16x20000000 --reps 3 --out gpurun_out/sweep_mid.json > gpurun_out/s26_sweep.log 2>&1
grep -v "^/opt" gpurun_out/s26_sweep.log | grep -v generic | sort -k1,1 -k6,6n | awk '{print $1,$2,$5,$6}' | tail -60
python - <<'PY'
import sys, os
sys.path.insert(0, "distributed-decisiontrees_amd")
import ddt
e = ddt.Engine(0)
for T, D, F in ((300, 4, 32), (300, 6, 28), (1000, 6, 32), (1000, 4, 16)):
    w, f = ddt.synth_model(T, D, F)
    e.load_model(ddt.make_params(T, D, F), w, f)
    print("auto", T, D, F, e.info().variant_name.decode())
PY
