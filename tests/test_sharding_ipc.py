"""SURVEY.md section 8e option 3 on the device: the encode kernels of every rank store straight into the ROOT's buffer through
IPC-mapped memory (sharding.open_root_buffer / encode_into_root) -- no data-path collective.  Two ranks that share the box's
GPU (gloo for the handle broadcast and the barrier): the mapping and the slab arithmetic are what is tested here; over xGMI
the same stores cross the links (8-GPU runs are the driver's)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_encode_into_the_roots_buffer(gpu):
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "ipc_worker.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "IPC_ROOT_BUFFER_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
