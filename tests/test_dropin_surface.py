"""CPU: the drop-in surface against facts read off the compiled reference (tests/golden/surface.json, made by
tests/golden/make_surface.py from oracle/_ref): every reference parameter of every Tokenizer method is here under the same name, in
the same position, with the same default (`nthreads = 1`, src/tokenize.cpp:81,98); additive parameters are keyword-only; the module
exports every reference name; iterating a FlatFile yields FlatFileIterator objects with `.seq` / `.sequence`
(fxstats.cpp:136-160,177)."""
import json
import os
import re

import pytest


@pytest.fixture(scope="module")
def surface(golden_dir):
    with open(os.path.join(golden_dir, "surface.json")) as f:
        return json.load(f)


def _overloads(doc):
    """[(positional [(name, default)], keyword_only [(name, default)])] per overload of a pybind11 docstring."""
    out = []
    for m in re.finditer(r"^\s*(?:\d+\.\s*)?\w+\((self: [^,)]+(?:, )?)(.*)\) -> ", doc, flags=re.M):
        pos, kw, bucket = [], [], None
        bucket = pos
        for part in re.split(r", (?=\w+: |\*)", m.group(2)) if m.group(2) else []:
            if part.strip() == "*":
                bucket = kw
                continue
            bucket.append([part.split(":", 1)[0].strip(), part.rsplit(" = ", 1)[1] if " = " in part else None])
        out.append((pos, kw))
    return out


def test_every_reference_parameter_keeps_its_name_position_and_default(bsq, surface):
    for name, ref_overloads in surface["signatures"].items():
        mine = _overloads(getattr(bsq.Tokenizer, name).__doc__)
        assert len(mine) == len(ref_overloads), name
        for (pos, kw), ref in zip(mine, ref_overloads):
            assert pos == ref, (name, pos, ref)        # the reference's parameters, nothing in between, same defaults
            assert all(d is not None for _, d in kw)   # what this build adds is keyword-only and optional
    pos, _ = _overloads(bsq.Tokenizer.batch_tokenize.__doc__)[0]
    assert ["nthreads", "1"] in pos
    pos, _ = _overloads(bsq.Tokenizer.batch_onehot_encode.__doc__)[0]
    assert ["nthreads", "1"] in pos and pos[-1] == ["mask", "None"]


def test_every_reference_name_is_exported(bsq, surface):
    for n in surface["module_names"]:
        assert hasattr(bsq, n), n
    for n in surface["tokenizer_names"]:
        assert hasattr(bsq.Tokenizer, n), n


def test_host_thread_policy_is_module_level(bsq):
    """The reference's default nthreads = 1 defers to set_host_threads / BSQ_HOST_THREADS (0 = automatic)."""
    was = bsq.get_host_threads()
    try:
        bsq.set_host_threads(3)
        assert bsq.get_host_threads() == 3
        bsq.set_host_threads(-5)
        assert bsq.get_host_threads() == 0
    finally:
        bsq.set_host_threads(was)


def test_flatfile_iteration_protocol(bsq, surface, golden_dir):
    from bioseq_amd.flatfile import FlatFile, FlatFileIterator
    g = surface["flatfile_iter"]
    ff = FlatFile(os.path.join(golden_dir, "flatfile_small.ff"))
    items = list(ff)
    assert type(items[0]).__name__ == g["yielded_type"] == "FlatFileIterator" and isinstance(items[0], FlatFileIterator)
    assert type(items[0].seq).__name__ == g["seq_type"] == "bytearray"
    every = [bytes(ff[i]).decode() for i in range(len(ff))]
    assert [bytes(x.seq).decode() for x in items] == every == [bytes(x.sequence).decode() for x in items]  # every sequence, the first too
    assert every[1:] == g["seq"] == g["sequence"] and len(ff) == g["n"]                                   # the reference drops sequence 0 ...
    it = iter(ff)
    a, b = next(it), next(it)
    assert a is not b and bytes(a.seq).decode() == every[0] and bytes(b.seq).decode() == every[1]  # snapshots, as in the reference
    assert isinstance(iter(a), FlatFileIterator)
    try:  # ... which the switch reproduces bit for bit
        FlatFile.ITER_SKIPS_FIRST = True
        assert [bytes(x.seq).decode() for x in ff] == g["seq"]
    finally:
        FlatFile.ITER_SKIPS_FIRST = False
