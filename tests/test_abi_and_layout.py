"""CPU: the C-ABI library loads and exports every symbol include/bsq.h and include/bsq_diag.h declare; host-only ABI calls
behave; repository layout rules (the product never touches the oracle; no reference sources)."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from bioseq_amd import capi
    lib = capi.load()
    names = capi.declared_symbols()
    assert len(names) >= 25 and "bsq_onehot_device" in names and "bsq_tokenize_host" in names
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    assert lib.bsq_abi_version() == 7
    product = capi.declared_symbols(capi.HEADER_PATH)  # the drop-in surface holds no knobs / yardsticks / probes
    assert not [n for n in product if 'tuning' in n or 'fill' in n or 'selftest' in n or 'xcd' in n]


def test_host_only_abi_calls(alphabets_golden):
    from bioseq_amd import capi
    lib = capi.load()
    assert lib.bsq_num_keys() == 20
    assert [lib.bsq_key_name(i).decode() for i in range(20)] == sorted(alphabets_golden["keys"])
    lut = (ctypes.c_int8 * 256)()
    n = ctypes.c_int32(0)
    assert lib.bsq_lut_get(b"seb14", lut, ctypes.byref(n)) == capi.OK and n.value == 14
    assert list(lut) == alphabets_golden["luts"]["SEB14"]
    assert lib.bsq_lut_get(b"zzz", lut, ctypes.byref(n)) == capi.ERR_INVALID_KEY
    d = capi.make_desc("DNA", eos=True, bos=True, padchar=True)
    assert (lib.bsq_bos_id(ctypes.byref(d)), lib.bsq_eos_id(ctypes.byref(d)), lib.bsq_pad_id(ctypes.byref(d)),
            lib.bsq_alphabet_size(ctypes.byref(d))) == (4, 5, 6, 7)
    dt = ctypes.c_int(-1)
    for ch, code in ((b"B", capi.I8), (b"h", capi.I16), (b"I", capi.I32), (b"L", capi.U64), (b"q", capi.U64),
                     (b"f", capi.F32), (b"D", capi.F64)):
        assert lib.bsq_dtype_from_destchar(ch, ctypes.byref(dt)) == capi.OK and dt.value == code
    assert lib.bsq_dtype_from_destchar(b"u", ctypes.byref(dt)) == capi.ERR_DTYPE
    assert [lib.bsq_dtype_size(i) for i in range(6)] == [1, 2, 4, 8, 4, 8]
    offs = np.array([0, 4, 9, 9], dtype=np.int64)
    bad = ctypes.c_int64(0)
    assert lib.bsq_validate_lengths(offs.ctypes.data, 3, 7, 1, 1, ctypes.byref(bad)) == capi.OK and bad.value == -1
    assert lib.bsq_validate_lengths(offs.ctypes.data, 3, 6, 1, 1, ctypes.byref(bad)) == capi.ERR_SEQ_TOO_LONG and bad.value == 1
    assert lib.bsq_strerror(capi.ERR_SEQ_TOO_LONG) == b"seq len + bos + eos > padlen"
    assert lib.bsq_tuning_set(b"no_such_knob", 1) == capi.ERR_INVALID_ARG
    # result-changing ablations and the experiment kernels that lost are not in the library at all (round 6: csrc/labs/README.md): their
    # knob names are unknown to it, and no environment variable reaches anything
    for labs_knob in (b"tokens8_abl", b"expand_mode", b"xcd_claim", b"chunk_math", b"tokenize_nch", b"chunks_cpw", b"augment_mode"):
        assert lib.bsq_tuning_set(labs_knob, 1) == capi.ERR_INVALID_ARG, labs_knob
        assert lib.bsq_tuning_get(labs_knob) == 0
    assert lib.bsq_tuning_set(b"onehot_path", 0) == capi.OK and lib.bsq_tuning_get(b"onehot_path") == 0
    assert lib.bsq_onehot_kernel_name(ctypes.byref(capi.make_desc("AMINO20")), 65536, 1024, capi.F32) == b"k_tokens_pb8_fast<raw>+k_expand_chunks"
    assert lib.bsq_onehot_kernel_name(ctypes.byref(capi.make_desc("AMINO20")), 8192, 1024, capi.F32) == b"k_onehot_chunks"
    # (7-byte rows of SHORT reads: nibble ids + the LDS-free expansion since the end of round 5; long reads stay tiled)
    assert lib.bsq_onehot_kernel_name(ctypes.byref(capi.make_desc("DNA4", 1, 1, 1)), 1000000, 160, capi.I8) == b"k_tokens_pb8_fast<raw, nibbles>+k_expand_rows1<nibbles>"
    assert lib.bsq_onehot_kernel_name(ctypes.byref(capi.make_desc("DNA5")), 131072, 1024, capi.I8) == b"k_tokens_pb8_fast<raw, nibbles>+k_expand_rows1<nibbles>"   # (5-byte rows, long reads, chunk-aligned pitch: tiled until round 6's on-box check)
    assert lib.bsq_onehot_kernel_name(ctypes.byref(capi.make_desc("DNA5")), 1000000, 160, capi.I8) == b"k_tokens_pb8_fast<raw, nibbles>+k_expand_rows1<nibbles>"
    # (28-byte rows: ids as nibbles, round 5)
    assert lib.bsq_onehot_kernel_name(ctypes.byref(capi.make_desc("DNA4", 1, 1, 1)), 1000000, 160, capi.F32) == b"k_tokens_pb8_fast<raw, nibbles>+k_expand_chunks<nibbles>"
    assert lib.bsq_onehot_kernel_name(ctypes.byref(capi.make_desc("BYTES", 1, 1, 1)), 1000, 160, capi.I16) == b"k_onehot_generic"
    # the token entry points' dispatch, from the library itself (bench.py labels its lines with it)
    tk = lambda key, flags, B, P, bf, t, aug: lib.bsq_tokenize_kernel_name(ctypes.byref(capi.make_desc(key, *flags)), B, P, bf, t, aug)
    assert tk("AMINO20", (0, 0, 0), 65536, 1024, 1, capi.I8, 0) == b"k_tokens_bp8_fast" and tk("AMINO20", (0, 0, 0), 65536, 1024, 0, capi.I8, 0) == b"k_tokens_pb8_fast"
    assert tk("SEB8", (0, 0, 0), 262144, 512, 1, capi.I8, 1) == b"k_augment_tokens_fused(k_augment_groups -> k_tokens_bp8_fast)"      # 32 768 chunks: the flag form
    assert tk("SEB8", (0, 0, 0), 131072, 512, 1, capi.I8, 1) == b"k_augment_tokens_nowait(k_augment_groups || k_tokens_bp8_fast)+k_patch_tokens"  # 16 384 chunks
    assert tk("SEB8", (0, 0, 0), 131073, 512, 1, capi.I8, 1).startswith(b"k_augment_tokens_fused")
    assert tk("SEB8", (0, 0, 0), 4096, 500, 1, capi.I8, 1) == b"k_augment_groups+k_tokens_bp8" and tk("SEB8", (0, 0, 0), 4096, 100, 1, capi.I8, 0) == b"k_tokenize_chunks"
    assert tk("BYTES", (1, 1, 1), 1000, 160, 1, capi.I8, 0) == b"k_tokenize_generic"


def test_compute_entry_points_refuse_without_a_device():
    from bioseq_amd import capi
    lib = capi.load()
    if lib.bsq_device_count() > 0:
        pytest.skip("a HIP device is visible")
    d = capi.make_desc("DNA")
    chars = np.frombuffer(b"ACGT", dtype=np.uint8).copy()
    offs = np.array([0, 4], dtype=np.int64)
    out = np.zeros(8, dtype=np.int8)
    bad = ctypes.c_int64(0)
    st = lib.bsq_tokenize_host(ctypes.byref(d), chars.ctypes.data, offs.ctypes.data, 1, 8, 1, capi.I8,
                               out.ctypes.data, capi.SPACE_HOST, None, ctypes.byref(bad))
    assert st == capi.ERR_NO_DEVICE and b"no HIP device" in lib.bsq_last_error()
    assert (out == 0).all()


def _py_files(sub):
    for dp, _, fs in os.walk(os.path.join(ROOT, sub)):
        for f in fs:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                yield os.path.join(dp, f)


def test_product_never_touches_the_oracle():
    pat = re.compile(r"\boracle\b|bsq_oracle|bsqo_|_ref\b")
    for path in list(_py_files("bioseq_amd")) + list(_py_files("include")):
        text = open(path, errors="replace").read()
        hits = [l for l in text.splitlines() if pat.search(l)]
        assert not hits, (path, hits[:3])


def _non_doc_strings(path):
    """String constants of a Python file that are not docstrings."""
    import ast
    tree = ast.parse(open(path).read())
    doc_ids = set()
    for node in ast.walk(tree):
        if isinstance(node, (ast.Module, ast.ClassDef, ast.FunctionDef, ast.AsyncFunctionDef)) and node.body:
            first = node.body[0]
            if isinstance(first, ast.Expr) and isinstance(first.value, ast.Constant) and isinstance(first.value.value, str):
                doc_ids.add(id(first.value))
    for node in ast.walk(tree):
        if isinstance(node, ast.Constant) and isinstance(node.value, str) and id(node) not in doc_ids:
            yield node.value


def test_nothing_reads_the_reference_tree_at_run_time():
    """/root/reference does not exist on the GPU box: only doc citations may mention it (make_golden.py
    and __graft_entry__.build() are build-container tools and are exempt)."""
    files = [os.path.join(ROOT, "bench.py")] + [p for p in _py_files("bioseq_amd") if p.endswith(".py")] + \
            [p for p in _py_files("tests") if p.endswith(".py") and not p.endswith("make_golden.py")] + \
            [p for p in _py_files("oracle") if p.endswith(".py")] + [p for p in _py_files("scripts") if p.endswith(".py")]
    for path in files:
        bad = [v for v in _non_doc_strings(path) if "/root/reference" in v and os.path.basename(path) != "test_abi_and_layout.py"]
        assert not bad, (path, bad)


def test_repo_layout():
    for f in ("bench.py", "__graft_entry__.py", "DESIGN.md", "INTEGRATION.md", "include/bsq.h", "oracle/bsq_oracle.c",
              "oracle/Makefile", "tests/golden/make_golden.py", "profiles/README.md"):
        assert os.path.exists(os.path.join(ROOT, f)), f
    gi = open(os.path.join(ROOT, ".gitignore")).read()
    assert "oracle/_ref/" in gi
    gri = open(os.path.join(ROOT, ".gpurunignore")).read()
    assert "_ref" not in gri and ".so" not in gri  # built artefacts must travel to the GPU box


def test_index_math_selftest():
    """The kernels divide by reciprocal multiplies (fast_div: n < 2^31, div_by: n < 2^52); the library runs the same
    inline functions on the host against integer division -- 1.3 million cases incl. powers of two, 2^k +- 1,
    exact multiples and the top of both ranges."""
    from bioseq_amd import capi
    assert capi.load().bsq_selftest_index_math() == 0


def test_environment_cannot_reach_result_changing_knobs():
    """BSQ_TOKENS8_ABL / BSQ_EXPAND_MODE used to select ablation kernels whose output is wrong on purpose (ADVICE round 2,
    medium): in the product build they are not knobs at all; ordinary knobs still take their initial value from the
    environment."""
    import subprocess
    import sys
    code = ("import ctypes; from bioseq_amd import capi; L = capi.load(); "
            "print(L.bsq_tuning_get(b'tokens8_abl'), L.bsq_tuning_get(b'expand_mode'), L.bsq_tuning_get(b'onehot_path'), L.bsq_tuning_get(b'nt_stores'))")
    env = dict(os.environ, BSQ_TOKENS8_ABL="4", BSQ_EXPAND_MODE="9", BSQ_ONEHOT_PATH="2")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, cwd=ROOT, check=True).stdout.split()
    assert out == ["0", "0", "2", "1"], out


def test_onehot_path_rule_for_one_byte_rows():
    """The automatic kernel choice for int8 one-hots with rows of 3 ... 15 bytes (end of round 5; no device needed: the rule is host code).
    From 192 MB on: 3- ... 15-byte rows take nibble ids + the LDS-free expansion (round 5 kept 3- ... 6-byte rows of long reads at a chunk-aligned
    pitch tiled; round 6's on-box check found the tiled kernel 15-33 % behind there: profiles/r06/dispatch_check.txt); below 192 MB aligned
    tensors stay tiled, tensors whose position rows are not 64-byte aligned go two-pass from 32 MB.  Round 6 also: rows of 16 ... 23 bytes up to
    40 MB take the one-launch chunk-owner kernel, rows >= 16 bytes go two-pass from 128 MB, 2-byte elements with 14- / 15-byte rows from 192 MB."""
    from bioseq_amd import capi
    lib = capi.load()
    two = b"k_tokens_pb8_fast<raw, nibbles>+k_expand_rows1<nibbles>"
    name = lambda key, flags, B, P: lib.bsq_onehot_kernel_name(ctypes.byref(capi.make_desc(key, *flags)), B, P, capi.I8)
    for key, flags, B, P, want in [("DNA4", (1, 1, 1), 1000000, 160, two), ("DNA4", (1, 1, 1), 262144, 512, two), ("DNA4", (1, 1, 1), 65536, 2048, two),
                                   ("DNA4", (0, 0, 0), 1000000, 160, two), ("DNA5", (0, 0, 0), 1048576, 160, two), ("DNA5", (0, 0, 0), 299968, 600, two),
                                   ("DNA5", (0, 0, 0), 131072, 1024, two), ("DNA4", (1, 1, 0), 262144, 512, two),
                                   ("DNA4", (1, 1, 1), 131072, 160, b"k_onehot_tile"), ("DNA4", (1, 1, 1), 125000, 160, two),
                                   ("SEB14", (0, 0, 0), 131072, 512, two), ("SEB8", (1, 1, 1), 262144, 512, two), ("SEB14", (0, 0, 0), 16384, 512, b"k_onehot_tile")]:
        assert name(key, flags, B, P) == want, (key, flags, B, P, name(key, flags, B, P))
    nm = lambda key, flags, B, P, t: lib.bsq_onehot_kernel_name(ctypes.byref(capi.make_desc(key, *flags)), B, P, t)
    assert nm("DNA", (0, 0, 0), 4096, 512, capi.F32) == b"k_onehot_chunks" and nm("DNA", (0, 0, 0), 8192, 512, capi.F32) == b"k_onehot_tile"
    assert nm("DNA", (0, 0, 0), 16384, 512, capi.F32) == b"k_tokens_pb8_fast<raw>+k_expand_chunks"
    assert nm("DNA4", (1, 1, 1), 1000000, 160, capi.I16).startswith(b"k_tokens_pb8_fast<raw") and nm("DNA4", (0, 0, 0), 1000000, 160, capi.I16) == b"k_onehot_tile"
