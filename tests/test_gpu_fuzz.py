"""GPU: a short run of the randomised differential harness (tests/fuzz_gpu.py) -- every kernel path,
packed / list entry points, masks, misaligned device views, both one-hot layouts -- against the oracle."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_random_configurations_bit_exact(gpu):
    spec = importlib.util.spec_from_file_location("fuzz_gpu", os.path.join(ROOT, "tests", "fuzz_gpu.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    n, ran = fz.run(budget=20.0, seed=2024)
    assert n > 40 and ran["host lists"] > 0   # (60-100 configurations in 20 s, by box)
