import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "distributed-decisiontrees_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch, ddt
from oracle import oracle as O
eng = ddt.Engine(0)
N, F, D = 20000, 32, 8
d = eng.synth_tuples_device(0, N, F); torch.cuda.synchronize(); print("synth ok", flush=True)
for T in (125, 250, 1000):
    w, f = ddt.synth_model(T, D, F)
    m = O.Model(O.make_params(T, D, F), w, f)
    want = O.score(m, d.cpu().numpy().view(np.uint32))
    for G in (1, 2, 4, 8, 0):
        eng.set_option("q16_prepass_groups", G)
        eng.load_model(ddt.make_params(T, D, F), w, f)
        print("T", T, "forced", G, "plan", eng.info().prepass_groups, eng.info().variant_name.decode(), flush=True)
        out = eng.score_device(d); torch.cuda.synchronize()
        print("   scored, ok =", bool(np.array_equal(out.cpu().numpy().view(np.uint32), want.view(np.uint32))), flush=True)
