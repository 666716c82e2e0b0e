"""CPU: every host translation unit of the product under a sanitizer (GPU sanitizers do not exist on this pool).

* scripts/asan_host.sh: bsq_host.cpp, bsq_alphabet.cpp, bsq_fastx.cpp and the pybind11 layer rebuilt with g++ -fsanitize=address,undefined,
  the host-only test modules run against them (incl. the FASTX differentials against the compiled reference: adversarial texts,
  16 383 ... 131 072-byte lines, truncated gzip members).
* tests/native/host_ring_stress.cpp: the staging ring, encodes in pieces (with an injected failing piece), staged batches with fetched
  results and the pipelined download of bsq_host.cpp under ThreadSanitizer on a mock HIP runtime whose streams are real threads.
Skipped where g++ lacks the sanitizer runtimes."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROCM_INC = "/opt/rocm/include"


def _can_link(flag, tmp_path):
    src = tmp_path / "probe.cpp"
    src.write_text("int main() { return 0; }\n")
    return shutil.which("g++") and subprocess.run(["g++", flag, str(src), "-o", str(tmp_path / "probe")], capture_output=True).returncode == 0


def test_staging_ring_under_thread_sanitizer(tmp_path):
    if not os.path.isdir(os.path.join(ROCM_INC, "hip")):
        pytest.skip("no HIP headers")
    if not _can_link("-fsanitize=thread", tmp_path):
        pytest.skip("g++ cannot link the ThreadSanitizer runtime here")
    exe = str(tmp_path / "host_ring")
    csrc = os.path.join(ROOT, "bioseq_amd", "csrc")
    b = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-pthread", "-fsanitize=thread", "-D__HIP_PLATFORM_AMD__", "-I" + ROCM_INC,
                        "-I" + os.path.join(ROOT, "include"), "-I" + csrc, os.path.join(ROOT, "tests", "native", "host_ring_stress.cpp"),
                        os.path.join(csrc, "bsq_host.cpp"), os.path.join(csrc, "bsq_alphabet.cpp"), "-o", exe], capture_output=True, text=True, timeout=600)
    assert b.returncode == 0, b.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "HOST_RING_OK" in r.stdout and "WARNING: ThreadSanitizer" not in r.stderr, r.stdout[-1500:] + r.stderr[-4000:]


def test_host_code_under_address_and_ub_sanitizers(tmp_path):
    if not _can_link("-fsanitize=address,undefined", tmp_path):
        pytest.skip("g++ cannot link the ASan / UBSan runtimes here")
    if not os.path.isdir(os.path.join(ROCM_INC, "hip")) or not os.path.exists("/opt/rocm/lib/libamdhip64.so"):
        pytest.skip("no ROCm runtime to link the scratch library against")
    r = subprocess.run(["bash", os.path.join(ROOT, "scripts", "asan_host.sh")], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0 and "sanitizer run clean" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error:" not in r.stderr, r.stderr[-4000:]
