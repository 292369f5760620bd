#!/usr/bin/env python3
"""A ddt_score_device call captured by torch.cuda.CUDAGraph and replayed on new tuples in the same buffer, per kernel family and pre-pass form:
for each replay, which of the three batches the scores match (the diagnosis behind launch_zero_words, csrc/ddt_internal.h; the torch-free
counterpart is tools/ubench/graph_capi.cpp)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "distributed-decisiontrees_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch, ddt
from oracle import oracle as O

def run(T, D, F, n, opts, label):
    e = ddt.Engine(0)
    for k, v in opts.items(): e.set_option(k, v)
    m = O.gen_model(T, D, F, dist=0)
    p = m.params
    e.load_model(ddt.make_params(p.num_trees, p.num_levels, p.num_features, p.missing_bits, p.cmp_mode, p.clusters_per_tuple, 0), m.wlines, m.flines)
    xs = [O.gen_tuples(1000 * i + 5, n, F, dist=0) for i in range(3)]
    want = [O.score_fast(m, x) for x in xs]
    d = torch.from_numpy(xs[0].view(np.int32)).cuda()
    out = torch.zeros(n, dtype=torch.float32, device="cuda")
    e.score_device(d, out=out); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        e.score_device(d, out=out)
    res = []
    for i in (1, 2, 0):
        d.copy_(torch.from_numpy(xs[i].view(np.int32))); out.zero_(); torch.cuda.synchronize()
        g.replay(); torch.cuda.synchronize()
        got = out.cpu().numpy()
        res.append([int(np.array_equal(got.view(np.uint32), w.view(np.uint32))) for w in want])
    # direct calls for comparison
    d.copy_(torch.from_numpy(xs[1].view(np.int32))); e.score_device(d, out=out); torch.cuda.synchronize()
    direct = int(np.array_equal(out.cpu().numpy().view(np.uint32), want[1].view(np.uint32)))
    print(label, e.info().variant_name.decode(), "prepass_groups", e.info().prepass_groups, "replays(match batch0,1,2):", res, "direct:", direct, flush=True)
    e.close()

run(300, 8, 32, 2000, {}, "d8 small auto")
run(300, 8, 32, 2000, {"q16_cluster_split": 0}, "d8 small uncut")
run(300, 8, 32, 2000, {"q16_fused_prepass": 0, "q16_grouped_prepass": 0}, "d8 small transposed prepass")
run(300, 8, 32, 200000, {}, "d8 large")
run(300, 8, 32, 200000, {"q16_fused_prepass": 0, "q16_grouped_prepass": 0}, "d8 large transposed prepass")
run(8, 4, 16, 5000, {}, "stream d4")
run(30, 6, 16, 5000, {}, "d6")
