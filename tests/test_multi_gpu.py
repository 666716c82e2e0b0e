"""The N > 1 path on the GPU box.

* `bench.py --gpus 2` exactly as the driver's SCALE run calls it -- self-launching, strong and weak scaling, every whole-batch
  assembly timed AND compared by content with the single-rank encode -- with two ranks SHARING this box's GPU over gloo
  (BSQ_BENCH_BACKEND / BSQ_BENCH_SHARE_GPU: a code-path test, never a measurement).
* With two or more devices visible (an 8-GPU GPUTEST box; skipped on today's 1-GPU boxes): the same two commands over nccl = RCCL,
  one GPU per rank, and tests/nccl_worker.py -- every assembly form of bioseq_amd.sharding on device tensors, equal / ragged / empty
  shards, bit-exact against the CPU oracle."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(extra_env, *args):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **extra_env)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--full-line"] + list(args)
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]          # ONE JSON line, from rank 0
    return json.loads(lines[0])


def _check_line(d, scaling, backend):
    assert d["n_gpus"] == 2 and d["config"]["rccl_world_size"] == 2 and d["scaling"] == scaling
    assert d["config"]["backend"].startswith(backend)
    assert d["value"] > 0 and d["ms_per_step"] > 0 and d["roofline"]["frac"] > 0
    assert d["check"]["ok"] is True
    n = d["config"]["sequences_per_gpu"]
    assert d["config"]["job_sequences"] == (1000 if scaling == "strong" else 2 * n)


@pytest.mark.parametrize("scaling", ["strong", "weak"])
def test_bench_two_ranks_sharing_the_gpu_over_gloo(gpu, scaling):
    d = _bench({"BSQ_BENCH_BACKEND": "gloo", "BSQ_BENCH_SHARE_GPU": "1"}, "--workload", "cfg1oh", "--scaling", scaling, "--gather", "1")
    _check_line(d, scaling, "gloo")
    forms = d["gather"]["forms"]
    for name in ("all_gather", "direct_all", "direct_root", "store_into_root"):  # (via_tokens needs device collectives: nccl only)
        assert name in forms and forms[name]["ms"] == forms[name]["ms"] and forms[name]["ms"] > 0, (name, forms.get(name))
    assert "folds" in d["gather"]["check"]


def _need_two_devices():
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two HIP devices (one rank per GPU over RCCL)")


@pytest.mark.parametrize("scaling,workload", [("strong", "cfg1oh"), ("weak", "cfg1oh"), ("strong", "cfg2")])
def test_bench_two_ranks_over_rccl(gpu, scaling, workload):
    _need_two_devices()
    d = _bench({}, "--workload", workload, "--scaling", scaling, "--gather", "1")
    assert d["n_gpus"] == 2 and d["config"]["rccl_world_size"] == 2 and d["config"]["backend"].startswith("nccl")
    forms = d["gather"]["forms"]
    for name, f in forms.items():
        assert f["ms"] == f["ms"] and f["ms"] > 0, (name, f)   # every form finite (each assembled batch was compared by content)
    assert "store_into_root" in forms and ("via_tokens" in forms) == (workload == "cfg1oh")


def test_every_assembly_form_rehearsal_two_ranks_one_gpu(gpu):
    """tests/nccl_worker.py with its ranks sharing this box's GPU and gloo as the transport (host tensors cross the process boundary):
    the worker's own logic -- shard sizes incl. empty ones, every assembly form, the oracle comparison -- runs on every GPUTEST box,
    so that the RCCL run below cannot fail on a mistake in the test itself."""
    _run_worker(2, {"BSQ_TEST_BACKEND": "gloo"})


def test_every_assembly_form_over_rccl_vs_oracle(gpu):
    _need_two_devices()
    import torch
    _run_worker(min(torch.cuda.device_count(), 4), {})


def _run_worker(n, extra_env):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **extra_env)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "nccl_worker.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1200, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "MULTI_GPU_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
