"""The dispatch of bsq_onehot_device against every forced path (scripts/check_dispatch.py) on the `gate` subset of
scripts/dispatch_shapes.json: every forced path must give the automatic path's output BIT FOR BIT (a failure); how far the automatic choice
is behind the best forced one is only REPORTED (boxes differ by a few percent; the full table with its 5 % gate runs in gpu_evidence.sh and
its last output is under profiles/)."""
import importlib.util
import io
import json
import os

import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.slow]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_forced_paths_equal_the_automatic_one_and_report_speed(gpu, capsys):
    spec = importlib.util.spec_from_file_location("check_dispatch", os.path.join(ROOT, "scripts", "check_dispatch.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    shapes = [s for s in json.load(open(os.path.join(ROOT, "scripts", "dispatch_shapes.json")))["shapes"] if s.get("gate")]
    assert 8 <= len(shapes) <= 12
    buf = io.StringIO()
    results, bad = m.run(shapes, tolerance=0.05, report_only=True, out=buf)
    with capsys.disabled():
        print("\n" + buf.getvalue())
    assert not bad, bad
    assert len(results) == len(shapes) and all(r["us"]["auto"] > 0 for r in results)
