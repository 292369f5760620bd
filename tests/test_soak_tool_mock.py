"""tools/soak_fuzz.py (the randomized soak that runs on the GPU box) against the CPU model of the host side: its case generator and its checker
stay runnable -- a handful of small drawn cases through the real csrc/*.cpp with the kernels' CPU stand-ins (tests/mock_hip/), every row against
the oracle, and a planted mismatch must be reported."""
import importlib.util
import os

import numpy as np
import pytest

import ddt
from ddt import _lib
from tests.test_engine_mock import _build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture()
def soak(monkeypatch):
    L = _build("libddt_host_mock.so")
    L.mock_reset(2, 3, 8)
    monkeypatch.setattr(_lib, "_lib", L)
    spec = importlib.util.spec_from_file_location("soak_fuzz", os.path.join(ROOT, "tools", "soak_fuzz.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_drawn_cases_pass_and_a_planted_mismatch_is_reported(soak, monkeypatch):
    rng = np.random.default_rng(11)
    kinds, done = set(), 0
    while done < 14:
        c = soak.draw_case(rng)
        if c["T"] * c["D"] > 1500:
            continue
        c["device"] = False                                      # (host buffers: no torch tensors on a box without a GPU)
        c["rows"] = [min(r, 2500) for r in c["rows"]]
        try:
            name, err = soak.run_case(c)
        except ddt.DDTError as ex:
            assert ex.code == -5, (c, ex)                        # a documented refusal is not a failure
            continue
        assert err is None, (name, err, c)
        kinds.add(c["kind"])
        done += 1
    assert {"perfect", "sparse"} <= kinds
    # the checker itself: the oracle's answer off by one ulp in one row must come back as a mismatch of that batch
    c = {"kind": "perfect", "cmp_mode": 0, "clusters": 2, "sum_mode": 0, "dist": 0, "device": False, "tseed": 5, "T": 20, "D": 5, "F": 7, "rows": [300]}
    real = soak.O.score_fast

    def off_by_one(m, x, **kw):
        out = real(m, x, **kw)
        out.view(np.uint32)[17] ^= 1
        return out

    monkeypatch.setattr(soak.O, "score_fast", off_by_one)
    name, err = soak.run_case(c)
    assert err is not None and "1 rows differ" in err and "[17]" in err
