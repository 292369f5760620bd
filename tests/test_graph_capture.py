"""A scoring call inside a HIP graph: once a call of the same size has sized the engine's workspaces, `ddt_score_device` / `ddt_classify_device`
only enqueue work on the caller's stream (memset of the tile flags, rank pre-pass, scoring kernel(s), combine) -- no allocation, no
synchronisation -- so a caller may capture them (torch.cuda.CUDAGraph here) and replay the graph on new data in the same buffers.  The replays
must give the oracle's bits for the data that is in the buffers AT REPLAY TIME (nothing about the batch may be baked in at capture time: the
missing-value flags, the cut launch's partial sums, the ticket counters are all rebuilt on the stream)."""
import numpy as np
import pytest

from oracle import oracle as O
import ddt

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape", ["d8_small_batch_cut", "d8_large", "d12_parts", "sparse_r", "classes"])
def test_a_captured_call_replays_on_new_data(shape):
    import torch

    e = ddt.Engine(0)
    K = 1
    if shape == "sparse_r":
        T, D, F, n = 64, 16, 40, 3000
        s = O.gen_sparse_model(T, D, F, 8, 700, 0)
        q = s.params
        e.load_model_sparse(ddt.make_sparse_params(q.num_trees, q.num_levels, q.num_features, q.missing_bits, q.cmp_mode, q.clusters_per_tuple, 0), s.node_lines, s.first)
        ref = lambda x: O.score_sparse_fast(s, x)
    else:
        T, D, F, n = {"d8_small_batch_cut": (300, 8, 32, 2000), "d8_large": (120, 8, 20, 300_000), "d12_parts": (80, 12, 4, 5000), "classes": (300, 8, 32, 2500)}[shape]
        K = 10 if shape == "classes" else 1
        m = O.gen_model(T, D, F, dist=0, clusters=ddt.default_clusters(T // K))
        p = m.params
        params = ddt.make_params(p.num_trees, p.num_levels, p.num_features, p.missing_bits, p.cmp_mode, p.clusters_per_tuple, 0)
        if K > 1:
            e.load_model_multiclass(params, m.wlines, m.flines, K, True)
            ref = lambda x: O.classify_fast(m, x, K, True)
        else:
            e.load_model(params, m.wlines, m.flines)
            ref = lambda x: O.score_fast(m, x)
    batches = [O.gen_tuples(100 + i, n, F, dist=0) for i in range(3)]
    batches[1][7, 0] = 0x7FC00000                                       # the second batch holds a missing value: its tile's flag is set on the stream at replay
    d = torch.from_numpy(batches[0].view(np.int32)).cuda()
    out = torch.zeros(n, dtype=torch.float32, device="cuda")
    cls = torch.zeros((K, n), dtype=torch.float32, device="cuda") if K > 1 else None
    lab = torch.zeros(n, dtype=torch.int32, device="cuda") if K > 1 else None

    def call():
        if K > 1:
            e.classify_device(d, class_scores=cls, labels=lab)
        else:
            e.score_device(d, out=out)

    call()                                                               # sizes the workspaces (a call that grows one synchronises and reallocates: not capturable)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        call()
    for x in batches[1:] + batches[:1]:
        d.copy_(torch.from_numpy(x.view(np.int32)))
        out.zero_()
        g.replay()
        torch.cuda.synchronize()
        if K > 1:
            wl, ws = ref(x)
            assert np.array_equal(cls.cpu().numpy().view(np.uint32), ws.view(np.uint32)) and np.array_equal(lab.cpu().numpy(), wl)
        else:
            assert np.array_equal(out.cpu().numpy().view(np.uint32), ref(x).view(np.uint32)), shape
    del g
    e.close()
