"""pytest configuration: registers the `gpu` marker and puts the repo root on sys.path.

`-m "not gpu"` : oracle vs known-answer vectors, host logic, C-ABI symbol checks (no GPU needed).
`-m gpu`       : parity tests proper -- the HIP path, called through the C-ABI, vs the CPU oracle.
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
PKG = os.path.join(ROOT, "distributed-decisiontrees_amd")
if PKG not in sys.path:
    sys.path.insert(0, PKG)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu() -> bool:
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # A gpu-marked test on a box without a GPU is skipped (it is never silently "passed" on CPU).
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)
