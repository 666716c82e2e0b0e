"""Build container only (needs oracle/_ref, the reference compiled in place): the FASTX -> FlatFile differential of tests/test_flatfile.py
on as many random adversarial texts as asked for.   python tests/stress/cpu_stress_fastx.py 20000 777"""
import gzip, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # tests/stress/ -> repo root
sys.path.insert(0, ROOT)
import numpy as np
import bioseq_amd
from bioseq_amd.flatfile import FlatFile
from oracle import oracle as O
import importlib.util
spec = importlib.util.spec_from_file_location("tf", os.path.join(ROOT, "tests", "test_flatfile.py")); tf = importlib.util.module_from_spec(spec); spec.loader.exec_module(tf)
ref = O.load_reference()
assert ref is not None
td = tempfile.mkdtemp()
n = int(sys.argv[1]); seed = int(sys.argv[2])
rng = np.random.default_rng(seed)
for case in range(n):
    text = tf._random_fastx(rng)
    src = os.path.join(td, "c.fx") + (".gz" if case % 5 == 0 else "")
    with (gzip.open if src.endswith(".gz") else open)(src, "wb") as f:
        f.write(text)
    want = ref.getstats([src])[0]; got = bioseq_amd.getstats([src])[0]
    assert got.tolist() == want.tolist(), (case, text)
    a, b = os.path.join(td, "ref.ff"), os.path.join(td, "mine.ff")
    ref.FlatFile(src, a); FlatFile(src, b)
    assert open(b, "rb").read() == open(a, "rb").read(), (case, text)
    os.remove(src)
print("fastx differential ok: %d texts, seed %d" % (n, seed))
