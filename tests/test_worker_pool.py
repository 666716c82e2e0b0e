"""The pybind11 layer's host worker pool (bioseq_amd/csrc/bsq_worker_pool.h) under ThreadSanitizer: 20 000 short jobs of 2 ... 18
tasks back to back -- the pattern of one list call in pieces (scan, then a pack per piece) -- with workers polling, asleep or
late for the previous job.  CPU only; skipped when g++ has no TSan runtime."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("flags", [["-O1", "-g", "-fsanitize=thread"], ["-O2"]], ids=["tsan", "O2"])
def test_worker_pool_stress(tmp_path, flags):
    if not shutil.which("g++"):
        pytest.skip("no g++")
    exe = str(tmp_path / "pool")
    cmd = ["g++", "-std=c++17", "-pthread", *flags, "-I", os.path.join(ROOT, "bioseq_amd", "csrc"),
           os.path.join(ROOT, "tests", "native", "worker_pool_stress.cpp"), "-o", exe]
    b = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    if b.returncode != 0 and "-fsanitize=thread" in flags:
        pytest.skip("g++ cannot link the ThreadSanitizer runtime here: " + b.stderr[-300:])
    assert b.returncode == 0, b.stderr[-2000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "WORKER_POOL_OK" in r.stdout and "WARNING: ThreadSanitizer" not in r.stderr, r.stdout[-500:] + r.stderr[-3000:]
