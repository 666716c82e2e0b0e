#!/usr/bin/env python3
"""Golden facts about the reference's PYTHON SURFACE for the hot path, read off the compiled reference (oracle/_ref; build container
only): the pybind11 signature text of every Tokenizer method (parameter names, order and defaults -- `nthreads = 1`,
src/tokenize.cpp:81,98), the names the module exports, and what iterating a FlatFile yields (FlatFileIterator objects with
.seq / .sequence, never sequence 0: fxstats.cpp:136-160,177).  Writes tests/golden/surface.json (data only)."""
import json
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(ROOT, "oracle", "_ref"))
import cbioseq  # noqa: E402  (the compiled reference)


def params(doc):
    """[(name, default or None), ...] per overload, from a pybind11 docstring."""
    out = []
    for m in re.finditer(r"^\s*(?:\d+\.\s*)?\w+\((self: [^,)]+(?:, )?)(.*)\) -> ", doc, flags=re.M):
        ps = []
        for part in re.split(r", (?=\w+: )", m.group(2)) if m.group(2) else []:
            name = part.split(":", 1)[0].strip()
            default = part.rsplit(" = ", 1)[1] if " = " in part else None
            ps.append([name, default])
        out.append(ps)
    return out


def main():
    T = cbioseq.Tokenizer
    out = {"module_names": sorted(n for n in dir(cbioseq) if not n.startswith("_")),
           "tokenizer_names": sorted(n for n in dir(T) if not n.startswith("_")),
           "signatures": {n: params(getattr(T, n).__doc__) for n in ("__init__", "batch_tokenize", "batch_onehot_encode", "onehot_encode", "decode_tokens")}}
    ff = cbioseq.FlatFile(os.path.join(HERE, "flatfile_small.ff"))
    items = list(ff)
    out["flatfile_iter"] = {"n": len(ff), "yielded_type": type(items[0]).__name__,
                            "seq": [bytes(x.seq).decode() for x in items], "sequence": [bytes(x.sequence).decode() for x in items],
                            "seq_type": type(items[0].seq).__name__,
                            "note": "the reference never yields sequence 0 (pre-increment in __next__)"}
    with open(os.path.join(HERE, "surface.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print(json.dumps(out, indent=1)[:1500])


if __name__ == "__main__":
    main()
