"""CPU: the pieces bench.py's untimed correctness checks stand on.

* `bench.fold_device` (torch integer arithmetic, what runs on the GPU) computes the same three 64-bit folds as the numpy definition in
  tests/golden/make_bench_folds.py, on sizes that are not multiples of 8 and across its chunk boundary;
* the committed folds (made from the compiled REFERENCE's outputs) are reproduced by the C oracle on the workloads the oracle
  finishes in seconds -- the fixture is not a number nobody can recompute;
* every workload bench.py puts into the driver's line has its folds."""
import importlib.util
import json
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.fixture(scope="module")
def bench():
    return _load("bsq_bench", os.path.join(ROOT, "bench.py"))


def _np_fold(arr):
    M64 = (1 << 64) - 1
    b = np.ascontiguousarray(arr).view(np.uint8).reshape(-1)
    if b.size % 8:
        b = np.concatenate([b, np.zeros(8 - b.size % 8, dtype=np.uint8)])
    w = b.view("<u8")
    x = int(np.bitwise_xor.reduce(w)) if w.size else 0
    s = int(sum(int(v) for v in w.tolist())) & M64 if w.size < 4096 else None
    with np.errstate(over="ignore"):
        s2 = int(w.sum(dtype=np.uint64))
        ws = int((w * (np.arange(w.size, dtype=np.uint64) * np.uint64(2) + np.uint64(1))).sum(dtype=np.uint64))
    assert s is None or s == s2
    return "%016x" % x, "%016x" % s2, "%016x" % ws


def test_fold_device_equals_the_numpy_definition(bench):
    import torch
    rng = np.random.default_rng(3)
    for n in (1, 7, 8, 9, 1000, 4093, (1 << 16) + 5):
        a = rng.integers(0, 256, size=n, dtype=np.uint8)
        assert bench.fold_device(torch.from_numpy(a)) == _np_fold(a), n
    f = rng.standard_normal((33, 7, 5)).astype(np.float32)
    assert bench.fold_device(torch.from_numpy(f)) == _np_fold(f)
    t = torch.from_numpy(rng.integers(-100, 100, size=(50, 40), dtype=np.int64))
    assert bench.fold_device(t.t()) == _np_fold(np.ascontiguousarray(t.numpy().T))  # non-contiguous input: folded in C order


def test_committed_folds_are_reproduced_by_the_oracle(bench, oracle):
    from bioseq_amd import synth
    folds = json.load(open(os.path.join(ROOT, "tests", "golden", "bench_folds.json")))
    for w in bench.DEFAULT_CONFIGS + ["cfg3", "cfg1oh"]:
        assert w in folds and {"xor", "sum", "wsum", "nbytes"} <= set(folds[w]), w
    assert {"mutated_chars", "mutated_sequences"} <= set(folds["cfg5aug"])
    for w in ("cfg1oh", "cfg2", "cfg2sf", "cfg5"):
        cfg_name, op, destchar, batch_first = bench.WORKLOADS[w]
        c = synth.CONFIGS[cfg_name]
        chars, offs = synth.synth_packed(c["seed"], c["n"], c["lo"], c["hi"], c["letters"])
        ora = oracle.OracleTokenizer(c["key"], c["eos"], c["bos"], c["padchar"])
        out = ora.onehot_packed(chars, offs, c["padlen"], destchar, 8) if op == "onehot" else ora.tokenize_packed(chars, offs, c["padlen"], destchar, batch_first, 8)
        x, s, ws = _np_fold(out)
        assert (x, s, ws, out.nbytes) == (folds[w]["xor"], folds[w]["sum"], folds[w]["wsum"], folds[w]["nbytes"]), w


def test_summarizer_keeps_exactly_the_timed_dispatches(tmp_path):
    """scripts/summarize_prof.py on a synthetic rocprofv3 trace: a step kernel dispatched twice per step (40 steps, the timed ones
    are steps 10 .. 29 and take 100 ns, all others 1000 ns), a yardstick kernel and an odd one.  Only the timed dispatches enter
    kernel_timed_avg_us, and algorithmic bytes / that time / 8 TB/s is the recomputed roofline fraction."""
    import csv
    import subprocess
    import sys
    out = tmp_path / "prof"
    (out / "trace" / "x").mkdir(parents=True)
    rows, t = [], 0
    for step in range(40):
        for rep in range(2):
            d = 100 if 10 <= step < 30 else 1000
            rows.append(dict(Kernel_Name="k_step", Start_Timestamp=t, End_Timestamp=t + d, VGPR_Count=64, SGPR_Count=32, LDS_Block_Size=0,
                             Scratch_Size=0, Workgroup_Size=256, Grid_Size=1024))
            t += 2000
        if step % 7 == 0:
            rows.append(dict(Kernel_Name="k_fill", Start_Timestamp=t, End_Timestamp=t + 5000, VGPR_Count=8, SGPR_Count=8, LDS_Block_Size=0,
                             Scratch_Size=0, Workgroup_Size=256, Grid_Size=64))
            t += 6000
    with open(out / "trace" / "x" / "1_kernel_trace.csv", "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0]))
        w.writeheader()
        w.writerows(rows)
    line = {"roofline": {"algorithmic_bytes_per_launch": 1600, "frac": 0.99}, "timed": {"first_step_index": 10, "steps": 20, "step_calls_total": 40}}
    (out / "bench_trace.json").write_text("noise\n" + json.dumps(line) + "\n")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "summarize_prof.py"), str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    s = json.load(open(out / "summary.json"))
    k = s["kernel_trace"]["k_step"]
    assert k["calls"] == 80 and k["launches_per_step"] == 2 and k["n_timed"] == 40
    assert abs(k["kernel_timed_avg_us"] - 0.2) < 1e-9 and abs(k["kernel_median_us"] - 0.1) < 1e-9     # 2 launches x 100 ns per step
    assert "kernel_timed_avg_us" not in s["kernel_trace"]["k_fill"]
    assert abs(s["roofline_check"]["frac_from_trace"] - 1600 / 0.2e-6 / 8e12) < 1e-12 and s["roofline_check"]["frac_printed"] == 0.99


def test_build_id_and_traffic_staleness(bench, tmp_path, monkeypatch):
    """bsq_build_id() is a 16-hex-digit id of sources + flags; bench.py calls the committed counters stale when they were measured
    on another build, and every shard workload of the driver's line has reference-made folds."""
    import re
    from bioseq_amd import capi
    lib = capi.load()
    bid = lib.bsq_build_id().decode()
    assert re.fullmatch(r"[0-9a-f]{16}", bid), bid
    prof = tmp_path / "profiles"
    prof.mkdir()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    (prof / "traffic.json").write_text(json.dumps({"cfg3": {"hbm_bytes_per_launch": 5, "build_id": bid}, "cfg2": {"hbm_bytes_per_launch": 7, "build_id": "0" * 16},
                                                   "cfg5": {"hbm_bytes_per_launch": 9}}))
    assert bench.traffic_stale("cfg3", lib) is False and bench.traffic_stale("cfg2", lib) is True
    assert bench.traffic_stale("cfg5", lib) is True        # counters of an unknown build
    assert bench.traffic_stale("cfg4f", lib) is None       # no counters at all
    monkeypatch.undo()
    folds = json.load(open(os.path.join(ROOT, "tests", "golden", "bench_folds.json")))
    for w in bench.SHARD_CONFIGS:
        f = folds["%s_shard8" % w]
        assert {"xor", "sum", "wsum", "nbytes", "sequences"} <= set(f), w


def test_shard_folds_are_reproduced_by_the_oracle(bench, oracle):
    """The reference-made folds of rank 0's cfg4b shard (125 000 sequences, 140 MB) against the C oracle."""
    from bioseq_amd import synth
    folds = json.load(open(os.path.join(ROOT, "tests", "golden", "bench_folds.json")))["cfg4b_shard8"]
    c = synth.CONFIGS["cfg4"]
    chars, offs = synth.synth_packed(c["seed"], folds["sequences"], c["lo"], c["hi"], c["letters"])
    out = oracle.OracleTokenizer(c["key"], c["eos"], c["bos"], c["padchar"]).onehot_packed(chars, offs, c["padlen"], "B", 8)
    x, s, ws = _np_fold(out)
    assert (x, s, ws, out.nbytes) == (folds["xor"], folds["sum"], folds["wsum"], folds["nbytes"])


def test_driver_line_is_compact_and_round_trips(bench, tmp_path, monkeypatch, capsys):
    """round 5's 22-KB line was not parsed by the driver (BENCH_r05.json: parsed null).  The line bench.py
    prints is built from the full result by `compact_line`: at most 4096 bytes, the contract's keys + roofline + cpu_baseline, whatever
    was measured -- here from round 5's own full result (profiles/r05/bench_default.json) and from a grotesquely inflated one."""
    full = json.loads(open(os.path.join(ROOT, "profiles", "r05", "bench_default.json")).read().strip().splitlines()[-1])
    assert len(json.dumps(full)) > 20000   # the object that broke the reader
    text = bench.compact_line(full)
    assert len(text.encode()) < 4096 and "\n" not in text
    line = json.loads(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"):
        assert k in line, k
    assert line["metric"] == full["metric"] and line["n_gpus"] == 1 and line["vs_baseline"] is None
    assert abs(line["value"] - full["value"]) < 1e-5 * full["value"] and abs(line["ms_per_step"] - full["ms_per_step"]) < 1e-5 * full["ms_per_step"]
    assert line["config"]["workload"].startswith("cfg3:") and "model" not in line["config"]
    r = line["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s" and r["traffic"] == full["roofline"]["traffic"]
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4 and abs(r["frac"] - full["roofline"]["frac"]) < 1e-5
    assert r["algorithmic_bytes_per_launch"] == 5404354664 and r["kernel"] and r["kernel_avg_ms"] > 0
    c = line["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and c["sample"] and c["single_thread"]["cores"] == 1
    assert line["check"] == {"ok": True} and line["build_id"] == full["build_id"]
    assert set(line["configs"]) == set(full["configs"])
    for name, e in line["configs"].items():
        assert set(e) <= {"ms", "frac", "cold", "ok", "multi4"} and e["ok"] is True
        assert abs(e["frac"] - full["configs"][name]["frac"]) < 6e-4
    # whatever a later round measures, the line stays under the limit: optional parts are dropped, the contract's keys never
    fat = json.loads(json.dumps(full))
    for i in range(300):
        fat["configs"]["extra_workload_with_a_long_name_%d" % i] = dict(full["configs"]["cfg2"])
    fat["config"]["sharding"] = "x" * 5000
    fat["cpu_baseline"]["sample"] = "y" * 5000
    text = bench.compact_line(fat)
    assert len(text.encode()) <= 4096
    line = json.loads(text)
    assert "roofline" in line and "cpu_baseline" in line and "configs" not in line and line["value"] > 0
    # NaN / inf never reach the line (json.loads of the driver would choke on them)
    bad = json.loads(json.dumps(full))
    bad["roofline"]["frac_of_copy_mix"] = float("nan")
    bad["configs"]["cfg2"]["ms_per_step"] = float("inf")
    assert "NaN" not in bench.compact_line(bad) and "Infinity" not in bench.compact_line(bad)
    # an N > 1 run has no configs / cpu_baseline / e2e / loader (rank 0 prints the job's line): the same function, nothing missing that the contract names
    multi = {k: v for k, v in json.loads(json.dumps(full)).items() if k not in ("configs", "cpu_baseline", "e2e", "sustained", "loader")}
    multi["n_gpus"] = 8
    multi["config"]["rccl_world_size"] = 8
    multi["gather"] = {"rccl_world_size": 8, "rccl_version": "2.26.6", "forms": {"all_gather": {"ms": 1.5}, "direct_root": {"ms": 0.9}}}
    multi["config"]["backend"] = "gloo: nccl failed (DistBackendError: NCCL error in: ... unhandled system error); barrier + timing reductions over gloo"
    assert len(bench.compact_line(multi).encode()) < 4096
    line8 = json.loads(bench.compact_line(multi))
    assert line8["config"]["backend"].startswith("gloo: nccl failed")   # (a run that went on without RCCL says so in the line)
    assert line8["n_gpus"] == 8 and line8["config"]["rccl_world_size"] == 8 and line8["roofline"]["frac"] > 0 and "cpu_baseline" not in line8
    assert line8["gather"]["ms"] == {"all_gather": 1.5, "direct_root": 0.9} and line8["gather"]["rccl_world_size"] == 8
    # emit(): stdout is exactly that one line; the full object goes to bench_full.json
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    bench.emit(full)
    out = capsys.readouterr().out
    assert out.count("\n") == 1 and json.loads(out) == json.loads(bench.compact_line(full))
    assert json.load(open(tmp_path / "bench_full.json")) == full
