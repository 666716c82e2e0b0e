"""In-tree build of the native pieces (no setuptools, no JIT cache):

    bioseq_amd/libbsq_hip.so                      C-ABI library: HIP kernels for gfx950 + host staging
    bioseq_amd/cbioseq.cpython-*.so               pybind11 host layer (links libbsq_hip.so, rpath $ORIGIN)

`python bioseq_amd/build.py` or `bioseq_amd.build.build_all()`.  hipcc cross-compiles gfx950 code
objects without a GPU, so this runs in the build container; the .so files travel to the GPU box.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(ROOT, "include")
LIB = os.path.join(HERE, "libbsq_hip.so")
EXT = os.path.join(HERE, "cbioseq" + sysconfig.get_config_var("EXT_SUFFIX"))

LIB_SRCS = ["bsq_onehot.hip", "bsq_tokens.hip", "bsq_generic.hip", "bsq_tokens8.hip", "bsq_decode.hip", "bsq_augment.hip", "bsq_gather.hip", "bsq_diag.hip", "bsq_host.cpp", "bsq_alphabet.cpp", "bsq_fastx.cpp"]
EXT_SRCS = ["cbioseq_module.cpp"]


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _run(cmd):
    print("+", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)


def hipcc():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def build_lib(force=False):
    """One object per source under csrc/_obj/ (compiled in parallel, rebuilt only when the source or a header
    changed), then one link."""
    from concurrent.futures import ThreadPoolExecutor
    # every header of csrc/ and include/ is a dependency of every object (a handful of files: an exact depfile graph would
    # save nothing, and a header missing from a hand-kept list once left two kernels disagreeing about a table layout)
    import glob
    # (csrc/labs/ is history, not a dependency: nothing there is compiled and the build id does not depend on it)
    headers = sorted(glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(INCLUDE, "*.h")))
    extra = os.environ.get("BSQ_EXTRA_HIPCC_FLAGS", "").split()
    objdir = os.path.join(CSRC, "_obj")
    os.makedirs(objdir, exist_ok=True)
    stamp = os.path.join(objdir, "flags.txt")
    if not os.path.exists(stamp) or open(stamp).read() != " ".join(extra):
        force = True
    # build id (bsq_build_id(), include/bsq_diag.h): sources + headers + flags; compiled into bsq_alphabet.cpp only
    import hashlib
    h = hashlib.sha256(" ".join(extra).encode())
    for path in [os.path.join(CSRC, f) for f in LIB_SRCS] + headers:
        h.update(os.path.basename(path).encode() + b"\0" + open(path, "rb").read())
    build_id = h.hexdigest()[:16]
    id_stamp = os.path.join(objdir, "build_id.txt")
    id_changed = not os.path.exists(id_stamp) or open(id_stamp).read() != build_id
    # objects of sources that are no longer part of the library (a split or renamed translation unit) must not survive in _obj/: whoever
    # links `_obj/*.o` (scripts/asan_host.sh did) would get duplicate symbols or stale kernels
    keep = {f + ".o" for f in LIB_SRCS}
    for stale in os.listdir(objdir):
        if stale.endswith(".o") and stale not in keep:
            os.remove(os.path.join(objdir, stale))
    jobs, objs = [], []
    for f in LIB_SRCS:
        src, obj = os.path.join(CSRC, f), os.path.join(objdir, f + ".o")
        objs.append(obj)
        if force or _newer(obj, [src] + headers) or (f == "bsq_alphabet.cpp" and id_changed):
            # gfx950 hands the first 14 kernel-argument dwords to a wave in SGPRs (no s_load round trip before its first
            # vector load): the fast token kernel's signature is laid out for it (bsq_tokens8.hip)
            lang = ["-x", "hip", "-mllvm", "-amdgpu-kernarg-preload-count=14"] if f.endswith(".hip") else []
            jobs.append([hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wextra", "-pthread",
                         "-I" + INCLUDE, "-I" + CSRC] + extra + lang + (['-DBSQ_BUILD_ID="%s"' % build_id] if f == "bsq_alphabet.cpp" else []) +
                        ["-c", src, "-o", obj])
    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), int(os.environ.get("BSQ_BUILD_JOBS", "4")))) as ex:
            list(ex.map(_run, jobs))
        with open(stamp, "w") as fh:
            fh.write(" ".join(extra))
        with open(id_stamp, "w") as fh:
            fh.write(build_id)
    if jobs or not os.path.exists(LIB):
        _run([hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread", "-o", LIB] + objs + ["-lz"])
    return LIB


def build_ext(force=False):
    import pybind11
    import glob
    deps = [os.path.join(CSRC, f) for f in EXT_SRCS] + sorted(glob.glob(os.path.join(INCLUDE, "*.h"))) + [LIB]
    if force or _newer(EXT, deps):
        _run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", "-Wall", "-Wextra",
              "-I" + INCLUDE, "-I" + pybind11.get_include(), "-I" + sysconfig.get_paths()["include"],
              "-o", EXT] + [os.path.join(CSRC, f) for f in EXT_SRCS] +
             ["-L" + HERE, "-lbsq_hip", "-Wl,-rpath,$ORIGIN"])
    return EXT


def kernel_objects():
    """The hipcc-built objects of the library's .hip sources, in link order (scripts/asan_host.sh links exactly these)."""
    return [os.path.join(CSRC, "_obj", f + ".o") for f in LIB_SRCS if f.endswith(".hip")]


def build_all(force=False):
    return build_lib(force), build_ext(force)


if __name__ == "__main__":
    if "--kernel-objects" in sys.argv:
        print(" ".join(kernel_objects()))
    else:
        build_all(force="--force" in sys.argv)
