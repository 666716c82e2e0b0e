"""Shared fixtures.  `-m gpu` tests need a real MI355X; everything else runs on CPU.

The oracle (oracle/) is test infrastructure: it is imported HERE and in the test modules only.
"""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def _ensure_built():
    """A fresh checkout has no .so files (they are git-ignored): build them once (hipcc cross-compiles)."""
    import glob
    import importlib.util
    pkg = os.path.join(ROOT, "bioseq_amd")
    if os.path.exists(os.path.join(pkg, "libbsq_hip.so")) and glob.glob(os.path.join(pkg, "cbioseq*.so")):
        return
    spec = importlib.util.spec_from_file_location("bsq_build", os.path.join(pkg, "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    b.build_all()


def pytest_configure(config):
    _ensure_built()
    config.addinivalue_line("markers", "gpu: needs a HIP device (run on the MI355X box with -m gpu)")
    config.addinivalue_line("markers", "slow: full BASELINE-size case")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def alphabets_golden():
    with open(os.path.join(GOLDEN, "alphabets.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def kats():
    with open(os.path.join(GOLDEN, "kats.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def small_cases():
    with open(os.path.join(GOLDEN, "small_cases.json")) as f:
        index = json.load(f)
    arrays = np.load(os.path.join(GOLDEN, "small_cases.npz"))
    return index, arrays


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def bsq():
    import bioseq_amd
    return bioseq_amd


@pytest.fixture(scope="session")
def gpu(bsq):
    if bsq.device_count() < 1:
        pytest.fail("GPU test selected but no HIP device is visible to libbsq_hip.so "
                    "(the product has no CPU fallback)")
    import torch
    assert torch.cuda.is_available()
    return torch.device("cuda:0")
