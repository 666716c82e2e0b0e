"""Every visible HIP device, not just cuda:0.

The library keeps per-device state (XCD probe cache, two-pass scratch per (device, stream), pinned staging and the copy
stream of the host entry points, the augmentation table): each smoke-level case below runs on EVERY device of the box, and
one test drives two devices from ONE process on two streams -- the reference's only multi-GPU idiom
(training/cnnpretrain.py:85-86: one process, several devices).  On a 1-GPU box this is cuda:0 once and the two-device
test is skipped; on an 8-GPU box GPUTEST covers cuda:1...7.  Expected values: the README vector and the CPU oracle."""
import ctypes

import numpy as np
import pytest

from bioseq_amd import synth

pytestmark = pytest.mark.gpu


def _ndev():
    try:
        import torch
        return max(1, torch.cuda.device_count())   # counting devices does not initialise the GPU
    except Exception:
        return 1


DEVICES = list(range(_ndev()))


@pytest.fixture(params=DEVICES, ids=["cuda:%d" % d for d in DEVICES])
def dev(request, gpu):
    import torch
    if request.param >= torch.cuda.device_count():
        pytest.skip("device %d not visible" % request.param)
    return torch.device("cuda", request.param)


def same(a, b):
    assert a.dtype == b.dtype and a.shape == b.shape, (a.dtype, a.shape, b.dtype, b.shape)
    assert a.tobytes() == b.tobytes()


def test_readme_vector_on_each_device(dev, bsq, kats):
    tok = bsq.pbeos_tokenizers["DNA"]
    d = tok.batch_tokenize(["ACGT", "GGGG"], padlen=7, batch_first=True, device=dev)
    assert d.device == dev and d.cpu().tolist() == kats["readme"]["tokens"]
    assert tok.decode_tokens(d) == kats["readme"]["decoded"]
    oh = tok.batch_onehot_encode(["ACGT", "GGGG"], padlen=7, destchar="f", device=dev)
    assert oh.device == dev and tok.decode_logits(oh, batch_first=False) == kats["readme"]["decoded"]


@pytest.mark.parametrize("path", [1, 2, 3], ids=["tiled", "two-pass", "chunk-owner"])
def test_each_onehot_kernel_family_on_each_device(dev, bsq, oracle, path):
    from bioseq_amd import capi
    lib = capi.load()
    chars, offs = synth.synth_packed(4242 + dev.index, 700, 0, 180, synth.DIRTY)
    seqs = synth.unpack(chars, offs)
    tok, ora = bsq.Tokenizer("AMINO20", 1, 1, 1), oracle.OracleTokenizer("AMINO20", 1, 1, 1)
    exp = ora.batch_onehot_encode(seqs, padlen=182, destchar="f")
    capi.check(lib.bsq_tuning_set(b"onehot_path", path))
    try:
        for _ in range(2):    # the second call reuses the scratch this (device, stream) pair cached
            got = tok.batch_onehot_encode(seqs, padlen=182, destchar="f", device=dev)
            assert got.device == dev
            same(got.cpu().numpy(), exp)
    finally:
        capi.check(lib.bsq_tuning_set(b"onehot_path", 0))


def test_token_kernels_and_layouts_on_each_device(dev, bsq, oracle):
    import torch
    chars, offs = synth.synth_packed(99 + dev.index, 1500, 0, 250, synth.DIRTY)
    dch, dof = torch.from_numpy(chars).to(dev), torch.from_numpy(offs).to(dev)
    tok, ora = bsq.Tokenizer("SEB8", 1, 0, 1), oracle.OracleTokenizer("SEB8", 1, 0, 1)
    for P in (256, 251):                     # k_tokens_bp8 aligned form / row-piece form
        for d in ("b", "i", "q"):
            for bf in (True, False):
                got = tok.tokenize_packed(dch, dof, P, d, bf)
                assert got.device == dev
                exp = ora.tokenize_packed(chars, offs, P, d, bf)
                same(got.cpu().numpy().view(exp.dtype), exp)
    got = tok.onehot_packed(dch, dof, 256, "f", layout="bcl")          # channels-first (loader layout)
    exp = ora.onehot_packed(chars, offs, 256, "f")
    same(got.cpu().numpy(), np.ascontiguousarray(exp.transpose(1, 2, 0)))


def test_host_entry_points_on_each_device(dev, bsq, oracle):
    """list -> numpy (the reference's default return): pinned staging, the library's copy stream and the pipelined
    download all live on the CURRENT device."""
    import torch
    chars, offs = synth.synth_packed(7 + dev.index, 5000, 0, 400, synth.AA)
    seqs = synth.unpack(chars, offs)
    tok, ora = bsq.Tokenizer("AMINO20", 1, 1, 1), oracle.OracleTokenizer("AMINO20", 1, 1, 1)
    with torch.cuda.device(dev):
        got = tok.batch_onehot_encode(seqs, padlen=402, destchar="f", nthreads=4)          # 185 MB: pipelined download
        assert isinstance(got, np.ndarray)
        same(got, ora.batch_onehot_encode(seqs, padlen=402, destchar="f", nthreads=4))
        got = tok.batch_tokenize(seqs, padlen=402, batch_first=True)
        same(got, ora.batch_tokenize(seqs, padlen=402, batch_first=True))
        got = tok.tokenize_packed(chars, offs, 402, "h", False)
        same(got, ora.tokenize_packed(chars, offs, 402, "h", False))


def test_flatfile_store_augmentation_and_decode_on_each_device(dev, bsq, oracle, tmp_path):
    import torch
    from bioseq_amd import blosum
    from bioseq_amd.flatfile import FlatFile, write_flatfile
    chars, offs = synth.synth_packed(31 + dev.index, 900, 0, 120, synth.AA)
    seqs = synth.unpack(chars, offs)
    ff = FlatFile(write_flatfile(seqs, str(tmp_path / "store.ff")))
    dch, dof = ff.to_device(dev)
    assert dch.device == dev and dof.device == dev
    tok, ora = bsq.Tokenizer("AMINO20", 1, 1, 0), oracle.OracleTokenizer("AMINO20", 1, 1, 0)
    got = ff.batch_tokenize(tok, 100, 700, padlen=122, destchar="B", batch_first=True, device=dev)
    exp = ora.batch_tokenize(seqs[100:700], padlen=122, batch_first=True)
    same(got.cpu().numpy().view(exp.dtype), exp)
    # augmentation in place on this device, then tokens == oracle tokens of the mutated bytes
    c2 = dch.clone()
    blosum.augment_packed(c2, dof, 1, 1.0, 11)
    mutated = c2.cpu().numpy()
    assert ((mutated != chars).sum()) == int((np.diff(offs) > 0).sum())
    got = tok.tokenize_packed(c2, dof, 122, "B", True)
    same(got.cpu().numpy().view(np.int8), ora.tokenize_packed(mutated, offs, 122, "b", True))


def test_c_abi_with_explicit_stream_on_each_device(dev, oracle):
    """Raw C ABI on a non-default stream of device k (no torch device guard inside the library: the caller's current
    device is the contract, include/bsq.h)."""
    import torch
    from bioseq_amd import capi
    lib = capi.load()
    B, P = 3000, 200
    chars, offs = synth.synth_packed(5 + dev.index, B, 0, P - 2, synth.DIRTY)
    desc = capi.make_desc("AMINO20", 1, 1, 1)
    ora = oracle.OracleTokenizer("AMINO20", 1, 1, 1)
    with torch.cuda.device(dev):
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            dch, dof = torch.from_numpy(chars).to(dev), torch.from_numpy(offs).to(dev)
            out = torch.empty((P, B, 23), dtype=torch.float32, device=dev)
            tk = torch.empty((B, P), dtype=torch.int8, device=dev)
            capi.check(lib.bsq_onehot_device(ctypes.byref(desc), dch.data_ptr(), dof.data_ptr(), None, B, P, capi.F32,
                                             out.data_ptr(), ctypes.c_void_p(st.cuda_stream)))
            capi.check(lib.bsq_tokenize_device(ctypes.byref(desc), dch.data_ptr(), dof.data_ptr(), B, P, 1, capi.I8,
                                               tk.data_ptr(), ctypes.c_void_p(st.cuda_stream)))
        st.synchronize()
        assert lib.bsq_xcd_round_robin() in (0, 1)
    same(out.cpu().numpy(), ora.onehot_packed(chars, offs, P, "f"))
    same(tk.cpu().numpy(), ora.tokenize_packed(chars, offs, P, "b", True))


def test_two_devices_from_one_process_on_two_streams(gpu, bsq, oracle):
    """training/cnnpretrain.py:85-86 drives several devices from one process.  Two devices, one stream each, launches
    interleaved from one host thread and from two host threads."""
    import threading
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two visible devices")
    tok, ora = bsq.Tokenizer("AMINO20", 1, 1, 1), oracle.OracleTokenizer("AMINO20", 1, 1, 1)
    devs = [torch.device("cuda", 0), torch.device("cuda", torch.cuda.device_count() - 1)]
    batches = [synth.synth_packed(1000 + i, 4000 + 333 * i, 0, 300, synth.DIRTY) for i in range(2)]
    exp = [ora.onehot_packed(c, o, 302, "f") for c, o in batches]
    res = [None, None]
    streams = []
    for d in devs:
        with torch.cuda.device(d):
            streams.append(torch.cuda.Stream())
    dev_in = [(torch.from_numpy(c).to(d), torch.from_numpy(o).to(d)) for (c, o), d in zip(batches, devs)]
    for rep in range(3):        # interleaved from one host thread
        for k, d in enumerate(devs):
            with torch.cuda.device(d), torch.cuda.stream(streams[k]):
                res[k] = tok.onehot_packed(dev_in[k][0], dev_in[k][1], 302, "f")
    for k, d in enumerate(devs):
        streams[k].synchronize()
        assert res[k].device == d
        same(res[k].cpu().numpy(), exp[k])

    def work(k):
        with torch.cuda.device(devs[k]), torch.cuda.stream(streams[k]):
            for _ in range(10):
                res[k] = tok.onehot_packed(dev_in[k][0], dev_in[k][1], 302, "f")
            streams[k].synchronize()
    th = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for k in range(2):
        same(res[k].cpu().numpy(), exp[k])
