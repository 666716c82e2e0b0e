"""Make libbsq_hip.so and PyTorch share ONE HIP runtime.

PyTorch-ROCm wheels bundle their own libamdhip64.so (SONAME libamdhip64.so.7) under torch/lib and
load it by the unversioned file name; libbsq_hip.so needs "libamdhip64.so.7".  The dynamic loader
merges the two only when the bundled copy is loaded FIRST (its SONAME then satisfies our NEEDED
entry).  Loaded the other way round the process ends up with two HIP/HSA runtimes, torch reports no
devices, and stream / memory handles cannot be exchanged.  So: before the extension is loaded,
preload torch's bundled runtime if torch is installed (located without importing torch).
"""
import ctypes
import importlib.util
import os
import sys

_done = False


def preload() -> None:
    global _done
    if _done:
        return
    _done = True
    if "torch" in sys.modules:  # torch already loaded its runtime
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    libdir = os.path.join(os.path.dirname(spec.origin), "lib")
    path = os.path.join(libdir, "libamdhip64.so")
    if os.path.exists(path):
        try:
            ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
        except OSError:
            pass  # fall back to the system ROCm runtime
