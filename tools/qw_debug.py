#!/usr/bin/env python3
"""TEMP: iteration / cycle counters of the queued-walker kernels (DDT_QW_DEBUG=1..4: iterations, cycles in rounds, cycles before rounds, cycles of the loop)"""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:
    sys.path.insert(0, os.path.join(ROOT, "distributed-decisiontrees_amd")); sys.path.insert(0, ROOT)
    import numpy as np, torch, ddt
    name, N = sys.argv[1], 1 << 20
    lines, first = ddt.synth_sparse_model(512, 16, 64, 10, 700, 0)
    eng = ddt.Engine(0)
    d = eng.synth_tuples_device(0, N, 64)
    out = torch.empty(N, dtype=torch.float32, device="cuda")
    eng.set_option("variant", ddt.variant_names().index(name))
    eng.load_model_sparse(ddt.make_sparse_params(512, 16, 64), lines, first)
    eng.score_device(d, out=out); torch.cuda.synchronize()
    o = out.cpu().numpy().reshape(-1, 64)[:, 0]   # one value per wave
    print(name, "DEBUG", os.environ.get("DDT_QW_DEBUG"), "mean %.1f min %.1f max %.1f" % (o.mean(), o.min(), o.max()), "per group %.2f" % (o.mean() / 64), flush=True)
else:
    for name in ("sparse_qw2_k8_u8_t256", "sparse_qw3_k8_u8_t256", "sparse_qw4_k8_u8_t256"):
        for dbg in "1234":
            subprocess.run([sys.executable, __file__, name], env=dict(os.environ, DDT_QW_DEBUG=dbg))
