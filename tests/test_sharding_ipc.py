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


@pytest.mark.parametrize("key,flags,dtype", [("AMINO20", (1, 1, 1), "f"), ("DNA", (1, 1, 1), "b"), ("SEB8", (0, 1, 0), "h"), ("DNA5", (1, 0, 1), "d")])
def test_onehot_written_as_a_column_block_of_a_larger_tensor(gpu, oracle, key, flags, dtype):
    """bsq_onehot_block_device: three shards written side by side into ONE (P, B, C) tensor == the oracle's one-hot of the
    whole batch; bytes outside the tensor untouched; ragged shard sizes, unaligned column offsets."""
    import ctypes
    import numpy as np
    import torch
    from bioseq_amd import capi, synth
    lib = capi.load()
    desc = capi.make_desc(key, *flags)
    ora = oracle.OracleTokenizer(key, *flags)
    C = ora.alphabet_size()
    dt = ctypes.c_int(0)
    capi.check(lib.bsq_dtype_from_destchar(dtype.encode(), ctypes.byref(dt)))
    tdt = {0: torch.int8, 1: torch.int16, 4: torch.float32, 5: torch.float64}[dt.value]
    for B, P, cuts in ((1000, 130, (0, 333, 334, 1000)), (257, 64, (0, 1, 256, 257)), (70000, 40, (0, 30000, 30001, 70000))):
        chars, offs = synth.synth_packed(B + P, B, 0, P - 2, synth.DIRTY)
        exp = ora.onehot_packed(chars, offs, P, dtype)
        guard = 64
        blocks = list(zip(cuts[:-1], cuts[1:]))
        for order in (blocks, blocks[::-1]):  # (in both orders: a block that writes a byte of its neighbour is then caught either way)
            buf = torch.full((P * B * C + 2 * guard,), 7, dtype=tdt, device=gpu)
            full = buf[guard:guard + P * B * C].view(P, B, C)
            for b0, b1 in order:
                c = torch.from_numpy(chars[offs[b0]:offs[b1]].copy()).to(gpu)
                o = torch.from_numpy(offs[b0:b1 + 1] - offs[b0]).to(gpu)
                capi.check(lib.bsq_onehot_block_device(ctypes.byref(desc), c.data_ptr(), o.data_ptr(), None, b1 - b0, P, dt,
                                                       full[:, b0:b1].data_ptr(), B, None))
            torch.cuda.synchronize()
            host = buf.cpu().numpy()
            assert (host[:guard] == 7).all() and (host[-guard:] == 7).all(), "wrote outside the tensor"
            assert host[guard:-guard].tobytes() == exp.tobytes(), (key, B, P)
    assert lib.bsq_onehot_block_device(ctypes.byref(desc), c.data_ptr(), o.data_ptr(), None, 10, P, dt, full.data_ptr(), 5, None) != 0   # row_seqs < B


@pytest.mark.parametrize("key,flags,dtype", [("AMINO20", (0, 0, 0), "f"), ("DNA", (1, 1, 1), "f"), ("SEB8", (0, 1, 0), "h"), ("AMINO20", (1, 1, 1), "b"),
                                             ("BYTES", (1, 0, 1), "h")])  # (BYTES: ids > 250 -> the generic kernel, also as a block)
def test_column_blocks_made_of_whole_chunks(gpu, oracle, key, flags, dtype):
    """Blocks of k x 4096 sequences at 4096-sequence boundaries of a 4-KiB aligned tensor go through the two-pass stream with a
    row gap (rows >= 16 bytes), the ragged last block through the tiles; the offsets are the whole batch's, advanced to the
    block's first sequence (what the host path in pieces hands over).  Also with the tiled kernel forced."""
    import ctypes
    import numpy as np
    import torch
    from bioseq_amd import capi, synth
    lib = capi.load()
    desc = capi.make_desc(key, *flags)
    ora = oracle.OracleTokenizer(key, *flags)
    C = ora.alphabet_size()
    dt = ctypes.c_int(0)
    capi.check(lib.bsq_dtype_from_destchar(dtype.encode(), ctypes.byref(dt)))
    tdt = {0: torch.int8, 1: torch.int16, 4: torch.float32, 5: torch.float64}[dt.value]
    for B, P, cuts in ((12288, 70, (0, 4096, 12288)), (20000, 33, (0, 8192, 16384, 20000)), (8192, 128, (0, 4096, 8192))):
        chars, offs = synth.synth_packed(B + P, B, 0, P - 2, synth.DIRTY)
        exp = ora.onehot_packed(chars, offs, P, dtype)
        dchars, doffs = torch.from_numpy(chars).to(gpu), torch.from_numpy(offs).to(gpu)
        for path in (0, 1):
            capi.check(lib.bsq_tuning_set(b"onehot_path", path))
            try:
                guard = 4096 // tdt.itemsize
                raw = torch.full((P * B * C + 3 * guard,), 7, dtype=tdt, device=gpu)
                buf = raw[(-raw.data_ptr() % 4096) // tdt.itemsize:][:P * B * C + 2 * guard]  # the caching allocator aligns to 512 only
                assert buf.data_ptr() % 4096 == 0
                full = buf[guard:guard + P * B * C].view(P, B, C)
                for b0, b1 in zip(cuts[:-1], cuts[1:]):
                    capi.check(lib.bsq_onehot_block_device(ctypes.byref(desc), dchars.data_ptr(), doffs[b0:].data_ptr(), None, b1 - b0, P, dt,
                                                           full[:, b0:b1].data_ptr(), B, None))
                torch.cuda.synchronize()
                host = buf.cpu().numpy()
            finally:
                capi.check(lib.bsq_tuning_set(b"onehot_path", 0))
            assert (host[:guard] == 7).all() and (host[-guard:] == 7).all(), "wrote outside the tensor"
            assert host[guard:-guard].tobytes() == exp.tobytes(), (key, B, P, path)


@pytest.mark.parametrize("mis", [8, 2560, 0])
def test_large_ragged_blocks_split_into_head_chunks_tail(gpu, oracle, mis):
    """A large block that neither starts nor ends on a 4-KiB chunk of the result (a ragged shard; a tensor torch aligned to 512 bytes)
    is encoded as head + run of whole chunks + tail; blocks written in both orders, guard bytes around the tensor."""
    import ctypes
    import numpy as np
    import torch
    from bioseq_amd import capi, synth
    lib = capi.load()
    key, flags, P, B = "AMINO20", (1, 1, 0), 64, 40000
    desc = capi.make_desc(key, *flags)
    ora = oracle.OracleTokenizer(key, *flags)
    C = ora.alphabet_size()  # 22: rows of 88 bytes, period 512 sequences
    chars, offs = synth.synth_packed(11, B, 0, P - 2, synth.DIRTY)
    exp = ora.onehot_packed(chars, offs, P, "f")
    dchars, doffs = torch.from_numpy(chars).to(gpu), torch.from_numpy(offs).to(gpu)
    nbytes = exp.nbytes
    raw = torch.empty((nbytes + 3 * 4096,), dtype=torch.uint8, device=gpu)
    start = (-raw.data_ptr()) % 4096 + 4096 + mis
    blocks = [(0, 13001), (13001, 13002), (13002, 40000)]
    for order in (blocks, blocks[::-1]):
        raw.fill_(7)
        for b0, b1 in order:
            capi.check(lib.bsq_onehot_block_device(ctypes.byref(desc), dchars.data_ptr(), doffs[b0:].data_ptr(), None, b1 - b0, P, 4,
                                                   raw.data_ptr() + start + b0 * C * 4, B, None))
        torch.cuda.synchronize()
        host = raw.cpu().numpy()
        assert (host[:start] == 7).all() and (host[start + nbytes:] == 7).all(), "wrote outside the tensor"
        assert host[start:start + nbytes].tobytes() == exp.tobytes(), (mis, order[0])


@pytest.mark.parametrize("key,flags,dtype", [("AMINO20", (1, 1, 1), "b"), ("DNA", (0, 1, 0), "h"), ("SEB8", (1, 0, 1), "q"), ("AMINO20", (0, 0, 0), "i"),
                                             ("DNA5", (1, 1, 0), "f"), ("BYTES", (1, 1, 1), "h")])
def test_seq_first_tokens_written_as_column_blocks(gpu, oracle, key, flags, dtype):
    """bsq_tokenize_block_device: shards written side by side into ONE (P, B) token matrix == the oracle's matrix of the whole batch;
    ragged and unaligned block widths, both orders, guard bytes; the fast types (b, h, q) and the generic fallback (i, f, BYTES)."""
    import ctypes
    import numpy as np
    import torch
    from bioseq_amd import capi, synth
    lib = capi.load()
    desc = capi.make_desc(key, *flags)
    ora = oracle.OracleTokenizer(key, *flags)
    dt = ctypes.c_int(0)
    capi.check(lib.bsq_dtype_from_destchar(dtype.encode(), ctypes.byref(dt)))
    tdt = {0: torch.int8, 1: torch.int16, 2: torch.int32, 3: torch.int64, 4: torch.float32, 5: torch.float64}[dt.value]
    for B, P, cuts in ((1000, 130, (0, 333, 334, 1000)), (70000, 40, (0, 4096, 30000, 30002, 70000)), (8192, 64, (0, 4096, 8192))):
        chars, offs = synth.synth_packed(B + P, B, 0, P - 2, synth.DIRTY)
        exp = ora.tokenize_packed(chars, offs, P, dtype, False)
        dchars, doffs = torch.from_numpy(chars).to(gpu), torch.from_numpy(offs).to(gpu)
        guard = 64
        blocks = list(zip(cuts[:-1], cuts[1:]))
        for order in (blocks, blocks[::-1]):
            buf = torch.full((P * B + 2 * guard,), 7, dtype=tdt, device=gpu)
            full = buf[guard:guard + P * B].view(P, B)
            for b0, b1 in order:
                capi.check(lib.bsq_tokenize_block_device(ctypes.byref(desc), dchars.data_ptr(), doffs[b0:].data_ptr(), b1 - b0, P, dt,
                                                         full[:, b0:b1].data_ptr(), B, None))
            torch.cuda.synchronize()
            host = buf.cpu().numpy()
            assert (host[:guard] == 7).all() and (host[-guard:] == 7).all(), "wrote outside the matrix"
            assert host[guard:-guard].tobytes() == exp.tobytes(), (key, dtype, B, P, order[0])
