"""GPU: the plain rank-quantised kernels do not walk the EMPTY trees that pad an image to whole chunks (Q16Aux::walk_subgroups,
csrc/ddt_kernels.hip TAILSKIP; depths <= 6, where a chunk holds 16..128 trees) -- scores equal, bit for bit, to the same kernel
walking them (option "q16_walk_padding" 1) and to the oracle, for tree counts that end in every position of a sub-group and of a
chunk, with missing values (the slow image), every cluster count and both reference-order adders.  The reference has no padding to
skip: an unused PU slot holds an EMPTY tree whose leaves are +0 (DTPU.sv model store) and adds +0 to its group."""
import numpy as np
import pytest

from oracle import oracle as O
import ddt

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    e = ddt.Engine(0)
    yield e
    e.set_option("q16_walk_padding", 0)
    e.set_option("variant", -1)
    e.close()


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def _q16_variants(D):
    return [(i, n) for i, n in enumerate(ddt.variant_names()) if n.startswith(f"q16_d{D}_") and not n.endswith("_p")]


def _score(eng, x):
    import torch

    d = torch.from_numpy(x.view(np.int32)).cuda()
    out = eng.score_device(d)
    torch.cuda.synchronize()
    return out.cpu().numpy()


# (T, D, F, clusters): T chosen against the chunk sizes 16 (d6) / 32 (d5) / 64 (d4, U = 8) / 128 (d3, U = 8)
CASES = [
    (100, 6, 28, 1),   # BASELINE config 2: 4 real trees in the last chunk of 16
    (100, 6, 28, 8),
    (3, 6, 12, 1),     # fewer trees than one sub-group
    (17, 6, 20, 2),    # one tree into the second chunk
    (29, 6, 16, 4),    # a chunk's last sub-group partly filled
    (33, 5, 32, 1),
    (70, 5, 16, 8),
    (5, 5, 8, 2),
    (40, 4, 16, 1),    # U = 8: sub-groups are whole PU groups
    (65, 4, 24, 4),
    (9, 4, 8, 8),
    (200, 3, 12, 1),
    (130, 3, 8, 2),
    (7, 3, 4, 1),
]


@pytest.mark.parametrize("T,D,F,clusters", CASES)
def test_padding_skip_is_bit_exact(eng, T, D, F, clusters):
    rows = 2500
    m = O.gen_model(T, D, F, dist=1, clusters=clusters)
    x = O.gen_tuples(5, rows, F, dist=1)  # dist 1: missing values -> some tiles walk the slow image
    x[:1024] = O.gen_tuples(6, 1024, F, dist=0)  # ... and the first tile none
    p = m.params
    seen = 0
    for sum_mode, osum in ((0, O.SUM_REF_NATIVE), (2, O.SUM_REF_FLOPOCO)):
        want = O.score(m, x, sum_mode=osum)
        for vid, name in _q16_variants(D):
            eng.set_option("variant", -1)
            eng.set_option("q16_walk_padding", 0)
            try:
                eng.load_model(ddt.make_params(p.num_trees, p.num_levels, p.num_features, p.missing_bits, p.cmp_mode, p.clusters_per_tuple, sum_mode),
                               m.wlines, m.flines)
                eng.set_option("variant", vid)
            except ddt.DDTError as ex:
                assert ex.code == -5
                continue
            skipped = _score(eng, x)
            eng.set_option("q16_walk_padding", 1)
            walked = _score(eng, x)
            assert np.array_equal(_bits(skipped), _bits(walked)), (name, sum_mode)
            assert np.array_equal(_bits(skipped), _bits(want)), (name, sum_mode)
            seen += 1
    assert seen >= 2, "no rank-quantised kernel took this shape"
