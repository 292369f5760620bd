"""GPU, BASELINE.json's FULL row counts: every row of every config held to the oracle through a size-independent property.

The oracle scores a few hundred thousand rows in seconds, not 10^8.  So the batch is made PERIODIC: row i = base[i mod P] with P a
prime near 10^6 / 3 * 10^5 (never a multiple of a kernel's tile of 64..1024 tuples: over the batch every base row meets every lane
and tile position, the ragged last tile, the tile that straddles two periods).  Then
  * rows [0, P) equal the oracle bit for bit (the first period IS an ordinary parity test), and
  * out[i] == out[i mod P] for EVERY i (compared on the GPU as 32-bit words, so NaN patterns count too),
which together say: all N results are the oracle's.  A kernel that mishandles a tile position, a chunk boundary, the phased result
stores' slot ring (config 1), the persistent kernels' ticket order (config 5) or a batch-size dependent launch plan shows up as a
row that differs from its twin one period earlier.  Reference semantics: SURVEY.md 8(a) A7-A14 (DTPU.sv:579-760 walk,
FPAddersReduceTree.sv:94-141 + FPAggregator.v:79-131 sum order); configs: BASELINE.json `configs`."""
import numpy as np
import pytest

from oracle import oracle as O
import ddt

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    e = ddt.Engine(0)
    yield e
    e.close()


def _periodic(base, n):
    """[n, W] int32 on the GPU: the rows of `base` ([P, W] uint32 host array) repeated."""
    import torch

    b = torch.from_numpy(np.ascontiguousarray(base).view(np.int32)).cuda()
    reps = (n + b.shape[0] - 1) // b.shape[0]
    return b.repeat(reps, 1)[:n]


def _assert_periodic(words, P, what):
    """words: 1-D int32 CUDA tensor; every element equals the one P places earlier."""
    import torch

    n = words.numel()
    full = n // P
    if full > 1:
        grid = words[: full * P].view(full, P)
        bad = (grid[1:] != grid[0]).any(dim=1)
        assert not bool(bad.any()), f"{what}: period(s) {[int(i) + 1 for i in torch.nonzero(bad)[:4, 0]]} differ from the first"
    tail = n - full * P
    if tail and full:
        assert torch.equal(words[full * P:], words[:tail]), f"{what}: the ragged last period differs from the first"


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


# (config, T, D, F, rows, P, dist, sum_mode): the bench's own synthetic models (ddt.synth_model seed 0) and shapes; sum_mode 2 = the
# reference's own adder (FPAdder_2cycles_latency.v:210-387) against the oracle's bit-level model of it (slower: a shorter period)
DENSE = [
    pytest.param(1, 8, 4, 16, 200_000_000, 1_000_003, 0, 0, id="config1-200M"),
    pytest.param(2, 100, 6, 28, 10_000_000, 1_000_003, 0, 0, id="config2-10M"),
    pytest.param(3, 1000, 8, 32, 100_000_000, 1_000_003, 0, 0, id="config3-100M"),
    pytest.param(3, 1000, 8, 32, 20_000_000, 300_007, 1, 0, id="config3-20M-missing-values"),
    pytest.param(1, 8, 4, 16, 50_000_000, 300_007, 1, 0, id="config1-50M-missing-values"),
    pytest.param(3, 1000, 8, 32, 100_000_000, 300_007, 0, 2, id="config3-100M-reference-adder"),
    pytest.param(2, 100, 6, 28, 10_000_000, 300_007, 1, 2, id="config2-10M-reference-adder-missing-values"),
    # round 5: the reference's own example configuration (profiler/profiler.cpp:32-38) on the deep kernel, in parts; with missing values and
    # the reference adder; and tuples of 64 words on the wide kernel
    pytest.param(6, 512, 12, 32, 10_000_000, 300_007, 0, 0, id="config6-10M-512xd12"),
    pytest.param(6, 512, 12, 32, 10_000_000, 100_003, 1, 2, id="config6-10M-reference-adder-missing-values"),
    pytest.param(3, 1000, 8, 64, 20_000_000, 300_007, 1, 0, id="wide-1000xd8x64-20M-missing-values"),
]


@pytest.mark.parametrize("config,T,D,F,rows,P,dist,sum_mode", DENSE)
def test_every_row_of_the_full_batch_is_the_oracles(eng, config, T, D, F, rows, P, dist, sum_mode):
    import torch

    if dist == 0:
        w, f = ddt.synth_model(T, D, F, 0)
        m = O.Model(O.make_params(T, D, F), w, f)
    else:
        m = O.gen_model(T, D, F, dist=1)
    p = m.params
    eng.set_option("variant", -1)
    eng.load_model(ddt.make_params(p.num_trees, p.num_levels, p.num_features, p.missing_bits, p.cmp_mode, p.clusters_per_tuple, sum_mode), m.wlines, m.flines)
    base = O.gen_tuples(17, P, F, dist=dist, missing_bits=p.missing_bits)
    d = _periodic(base, rows)
    out = torch.full((rows,), float("nan"), dtype=torch.float32, device="cuda")
    eng.score_device(d, out=out)
    torch.cuda.synchronize()
    want = O.score_fast(m, base, sum_mode=O.SUM_REF_FLOPOCO if sum_mode == 2 else O.SUM_REF_NATIVE)
    assert np.array_equal(_bits(out[:P].cpu().numpy()), _bits(want)), f"config {config}: the first period differs from the oracle"
    _assert_periodic(out.view(torch.int32), P, f"config {config} ({eng.info().variant_name.decode()})")
    del d, out
    torch.cuda.empty_cache()


def test_config5_every_label_and_class_sum(eng):
    import torch

    T, D, F, K, rows, P = 1000, 8, 32, 10, 10_000_000, 100_003
    w, f = ddt.synth_model(T, D, F, 0)
    clusters = ddt.default_clusters(T // K)
    eng.set_option("variant", -1)
    eng.load_model_multiclass(ddt.make_params(T, D, F, clusters=clusters), w, f, K, True)
    m = O.Model(O.make_params(T, D, F, clusters=clusters), w, f)
    base = O.gen_tuples(23, P, F)
    d = _periodic(base, rows)
    labels, cs = eng.classify_device(d)
    torch.cuda.synchronize()
    want_l, want_cs = O.classify(m, base, K, True, sum_mode=O.SUM_REF_NATIVE)
    assert np.array_equal(labels[:P].cpu().numpy(), want_l)
    assert np.array_equal(_bits(cs[:, :P].cpu().numpy()), _bits(want_cs))
    _assert_periodic(labels, P, "config 5 labels")
    for k in range(K):
        _assert_periodic(cs[k].view(torch.int32), P, f"config 5 class {k} sums")
    del d, labels, cs
    torch.cuda.empty_cache()


def test_config4_every_row_of_the_sparse_forest(eng):
    import torch

    T, D, F, rows, P = 512, 16, 64, 10_000_000, 100_003
    lines, first = ddt.synth_sparse_model(T, D, F, 10, 700, 0)
    eng.set_option("variant", -1)
    eng.load_model_sparse(ddt.make_sparse_params(T, D, F), lines, first)
    m = O.SparseModel(O.make_sparse_params(T, D, F), lines, first)
    base = O.gen_tuples(29, P, F)
    d = _periodic(base, rows)
    out = torch.full((rows,), float("nan"), dtype=torch.float32, device="cuda")
    eng.score_device(d, out=out)
    torch.cuda.synchronize()
    want = O.score_sparse_fast(m, base, sum_mode=O.SUM_REF_NATIVE)
    assert np.array_equal(_bits(out[:P].cpu().numpy()), _bits(want))
    _assert_periodic(out.view(torch.int32), P, f"config 4 ({eng.info().variant_name.decode()})")
    del d, out
    torch.cuda.empty_cache()
