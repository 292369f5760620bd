"""GPU: the stream kernel's phased result stores (csrc/ddt_kernels.hip stream_body: scores parked in LDS, written by all waves in the
same window of the device's 100 MHz clock) against its direct stores and the oracle -- every row of batches large enough for the
phased form to be chosen, ragged sizes, windows so short that every tile ends one, slot counts so small that the buffer-full path
runs, missing values (the slow walk), the fp64 and the reference-adder sums."""
import numpy as np
import pytest

from oracle import oracle as O
import ddt

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    e = ddt.Engine(0)
    yield e
    for k in ("stream_res_tiles", "stream_window_ticks", "stream_blocks_per_cu"):
        e.set_option(k, 0)
    e.close()


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def _score(eng, x):
    import torch

    d = torch.from_numpy(x.view(np.int32)).cuda()
    out = torch.full((x.shape[0],), float("nan"), dtype=torch.float32, device="cuda")
    eng.score_device(d, out=out)
    torch.cuda.synchronize()
    return out.cpu().numpy()


# (T, D, F, rows, dist, sum_mode)
CASES = [
    (8, 4, 16, 2_700_003, 0, 0),   # config 1's model, a ragged last tile
    (8, 4, 16, 2_700_003, 1, 0),   # missing values: tiles on the slow walk
    (8, 4, 16, 2_650_000, 0, 2),   # the reference adder
    (8, 4, 16, 2_650_000, 1, 1),   # fp64 sum
    (5, 4, 13, 2_700_001, 1, 0),   # EMPTY slots, F not a multiple of 4
    (12, 6, 16, 2_700_000, 0, 0),  # more than 64 visits per tuple: the resident block count stays
]
# (blocks_per_cu, res_tiles, window_ticks): defaults; a window every tile; buffer-full flushes (4 slots, a window that never comes); odd mixes
KNOBS = [(0, 0, 0), (0, 0, 100), (0, 4, 10_000_000), (3, 9, 700), (8, 0, 0)]


@pytest.mark.parametrize("T,D,F,rows,dist,sum_mode", CASES)
def test_phased_stores_equal_direct_stores_and_oracle(eng, T, D, F, rows, dist, sum_mode):
    m = O.gen_model(T, D, F, dist)
    x = O.gen_tuples(11, rows, F, dist)
    osum = {0: O.SUM_REF_NATIVE, 1: O.SUM_F64_SEQ, 2: O.SUM_REF_FLOPOCO}[sum_mode]
    eng.set_option("variant", -1)
    p = m.params
    eng.load_model(ddt.make_params(p.num_trees, p.num_levels, p.num_features, p.missing_bits, p.cmp_mode, p.clusters_per_tuple, sum_mode), m.wlines, m.flines)
    assert eng.info().variant_name.decode().startswith("stream_"), "this test is about the stream kernel"
    eng.set_option("stream_blocks_per_cu", 0)
    eng.set_option("stream_res_tiles", 1)
    direct = _score(eng, x)
    k = 200_000
    for sl in (slice(0, k), slice(rows - 4096, rows)):
        gold = O.score(m, x[sl], sum_mode=osum)
        if sum_mode == 1:  # fp64 accumulate: within the north-star tolerance of the oracle's fp64 sum
            assert np.allclose(direct[sl], gold, rtol=1e-6, atol=1e-6)
        else:
            assert np.array_equal(_bits(direct[sl]), _bits(gold))
    for bpc, nb, win in KNOBS:
        eng.set_option("stream_blocks_per_cu", bpc)
        eng.set_option("stream_res_tiles", nb)
        eng.set_option("stream_window_ticks", win)
        got = _score(eng, x)
        assert np.array_equal(_bits(got), _bits(direct)), (bpc, nb, win)
    for key in ("stream_res_tiles", "stream_window_ticks", "stream_blocks_per_cu"):
        eng.set_option(key, 0)


def test_option_ranges(eng):
    for key, bad in (("stream_res_tiles", -1), ("stream_res_tiles", 65), ("stream_window_ticks", 50), ("stream_window_ticks", -3)):
        with pytest.raises(ddt.DDTError):
            eng.set_option(key, bad)
    eng.set_option("stream_res_tiles", 0)
    eng.set_option("stream_window_ticks", 0)
