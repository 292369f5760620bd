"""The rank-quantised path (q16 kernels): exactness on adversarial values, fallback rules, timing counters.

q16 replaces every feature by its rank among the model's thresholds for that feature, so values that sit exactly
on, just below and just above thresholds are the interesting inputs; the generic parity suite
(test_gpu_parity.py) already runs every q16 variant on every shape it accepts."""
import numpy as np
import pytest

from oracle import oracle as O
import ddt

pytestmark = pytest.mark.gpu


def _variant(name):
    return ddt.variant_names().index(name)


def _params(m, sum_mode=0):
    p = m.params
    return ddt.make_params(p.num_trees, p.num_levels, p.num_features, p.missing_bits, p.cmp_mode, p.clusters_per_tuple, sum_mode)


@pytest.mark.parametrize("cmp_mode", [0, 1])
def test_values_on_and_next_to_thresholds(cmp_mode):
    T, D, F, n = 200, 8, 32, 4096
    m = O.gen_model(T, D, F, dist=1, cmp_mode=cmp_mode)
    rng = np.random.default_rng(0)
    thr = m.wlines.reshape(T, -1)[:, :255].reshape(-1)             # every threshold bit pattern of the model
    x = O.gen_tuples(0, n, F, dist=1)
    pick = thr[rng.integers(0, thr.size, (n, F))]
    delta = rng.integers(-1, 2, (n, F)).astype(np.int64)            # exactly on / one ulp either side
    near = (pick.astype(np.int64) + delta).astype(np.uint32)
    mask = rng.random((n, F)) < 0.6
    x[:, :F] = np.where(mask, near, x[:, :F])
    x[0, :F] = [0x80000000, 0x00000000, 0x7F800000, 0xFF800000, 0x7FC00001, 0x7FFFFFFF, 0x80000001, 0x00000001] * 4
    e = ddt.Engine(0)
    e.set_option("variant", _variant("q16_d8_c4_u4"))
    want = O.score(m, x)
    for sum_mode in (0, 1):
        e.load_model(_params(m, sum_mode), m.wlines, m.flines)
        assert e.info().variant_name.decode() == "q16_d8_c4_u4"
        got = e.score(x)
        ref = want if sum_mode == 0 else O.score(m, x, sum_mode=O.SUM_F64_SEQ)
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    e.close()


def test_auto_selection_and_fallbacks():
    e = ddt.Engine(0)
    w, f = ddt.synth_model(1000, 8, 32)
    e.load_model(ddt.make_params(1000, 8, 32), w, f)
    assert e.info().variant_name.decode() == "q16_d8_c4_u4"        # many trees: the pre-pass pays off
    e.load_model(ddt.make_params(1000, 8, 32), w, f, 3, 8)          # 125 trees per engine (8-way shard): fp32 tile kernel
    assert e.info().variant_name.decode() == "d8_t1024_r1_c4_u4_dma_f"
    # too many distinct thresholds on one feature for 16-bit ranks: 2000 trees x 255 nodes on 4 features
    w, f = ddt.synth_model(2000, 8, 4)
    e.set_option("variant", _variant("q16_d8_c4_u4"))
    with pytest.raises(ddt.DDTError) as ei:
        e.load_model(ddt.make_params(2000, 8, 4), w, f)
    assert ei.value.code == -5
    e.set_option("variant", -1)
    e.load_model(ddt.make_params(2000, 8, 4), w, f)
    assert not e.info().variant_name.decode().startswith("q16")
    x = O.gen_tuples(0, 1500, 4)
    m = O.Model(O.make_params(2000, 8, 4), w, f)
    assert np.array_equal(e.score(x).view(np.uint32), O.score(m, x).view(np.uint32))
    e.close()


def test_kernel_timing_counters():
    import torch

    e = ddt.Engine(0)
    w, f = ddt.synth_model(1000, 8, 32)
    e.load_model(ddt.make_params(1000, 8, 32), w, f)
    d = e.synth_tuples_device(0, 1 << 20, 32)
    e.set_option("kernel_timing", 1)
    e.score_device(d)
    st = e.stats()
    assert st.last_score_ms > 0.2 and 0.0 < st.last_prepass_ms < st.last_score_ms
    e.set_option("variant", _variant("d8_t1024_r1_c4_u4_dma_f"))
    e.score_device(d)
    st = e.stats()
    assert st.last_score_ms > 0.2 and st.last_prepass_ms < 0.05    # fp32 kernel: no pre-pass
    torch.cuda.synchronize()
    e.close()
