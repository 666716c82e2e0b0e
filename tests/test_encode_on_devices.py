"""`sharding.encode_on_devices` (round 6): ONE process, N devices -- the host packs the list once into pinned memory, device g receives
its `shard_bounds` slice on its own copy stream and encodes it on its own stream (SURVEY.md section 8e "host packs once; GPU g receives its
slice"; the reference's nn.DataParallel consumer, /root/reference/training/cnnpretrain.py:85-94, over the by-sequence partition of
src/tokenize.h:339-342).  On a 1-GPU box the devices are repeats of cuda:0 (separate stream pairs); with more devices visible the same
tests use them all.  Expected values: the CPU oracle on the same sequences, bit-exact."""
import numpy as np
import pytest

from bioseq_amd import synth

pytestmark = pytest.mark.gpu


def _devices(n):
    import torch
    have = torch.cuda.device_count()
    return ["cuda:%d" % (g % have) for g in range(n)]


def _seqs(seed, n, lo, hi):
    chars, offs = synth.synth_packed(seed, n, lo, hi, synth.AA)
    return synth.unpack(chars, offs), chars, offs


@pytest.mark.parametrize("n,G", [(1000, 2), (1001, 3), (7, 3), (2, 3), (0, 2), (4096, 1), (513, 8)])
def test_shards_equal_the_oracle(gpu, bsq, oracle, n, G):
    """per-device shards (no root): tokens in both layouts, one-hot in both layouts, every shard == the oracle on its sequences;
    ragged (n % G != 0), tiny (fewer sequences than devices: EMPTY shards) and empty batches"""
    import torch
    from bioseq_amd import sharding
    P = 160
    seqs, _, _ = _seqs(31 + n, n, 0, P - 2)
    tok, ora = bsq.Tokenizer("PROTEIN", 1, 1, 1), oracle.OracleTokenizer("PROTEIN", 1, 1, 1)
    devs = _devices(G)
    for op, kw, ax in (("tokenize", {"batch_first": True}, 0), ("tokenize", {"batch_first": False}, 1),
                       ("onehot", {"layout": "tbc", "destchar": "f"}, 1), ("onehot", {"layout": "bcl", "destchar": "f"}, 0)):
        got = sharding.encode_on_devices(tok, seqs, P, devices=devs, op=op, **kw)
        assert len(got) == G
        for g in range(G):
            b0, b1 = sharding.shard_bounds(n, G, g)
            assert got[g].device == torch.device(devs[g]) and got[g].shape[ax] == b1 - b0
            if b1 == b0:
                continue
            part = seqs[b0:b1]
            if op == "tokenize":
                want = ora.batch_tokenize(part, padlen=P, batch_first=kw["batch_first"])
            else:
                want = ora.batch_onehot_encode(part, padlen=P, destchar="f")
                if kw["layout"] == "bcl":
                    want = np.ascontiguousarray(want.transpose(1, 2, 0))
            assert got[g].cpu().numpy().tobytes() == want.tobytes(), (op, kw, g)


@pytest.mark.parametrize("n,G", [(1000, 2), (1001, 3), (5, 4), (30000, 4)])
def test_root_tensor_equals_the_oracle(gpu, bsq, oracle, n, G):
    """root=: every device stores its shard straight into the whole-batch tensor on the root device -- column blocks for the seq-first
    layouts (any first sequence: 1001 / 3 sequences are not chunk-aligned), row slabs for the batch-first ones"""
    import torch
    from bioseq_amd import sharding
    P = 128
    seqs, chars, offs = _seqs(77 + n, n, 0, P - 1)
    tok, ora = bsq.Tokenizer("DNA", 0, 1, 1), oracle.OracleTokenizer("DNA", 0, 1, 1)
    devs = _devices(G)
    for op, kw in (("tokenize", {"batch_first": True}), ("tokenize", {"batch_first": False}), ("onehot", {"layout": "tbc", "destchar": "f"}),
                   ("onehot", {"layout": "tbc", "destchar": "B"}), ("onehot", {"layout": "bcl", "destchar": "f"})):
        full = sharding.encode_on_devices(tok, seqs, P, devices=devs, op=op, root=devs[-1], **kw)
        assert full.device == torch.device(devs[-1])
        if op == "tokenize":
            want = ora.tokenize_packed(chars, offs, P, "B", kw["batch_first"])
        else:
            want = ora.onehot_packed(chars, offs, P, kw["destchar"], 4)
            if kw["layout"] == "bcl":
                want = np.ascontiguousarray(want.transpose(1, 2, 0))
        assert tuple(full.shape) == want.shape and full.cpu().numpy().tobytes() == want.tobytes(), (op, kw)


def test_packed_input_item_types_and_errors(gpu, bsq, oracle):
    """an already packed host batch (numpy) goes the same way; mixed item types pack like the list API; an over-long sequence raises the
    reference's errors (RuntimeError from the token path, ValueError from the one-hot path) before anything is uploaded"""
    import torch
    from bioseq_amd import sharding
    P = 64
    tok, ora = bsq.Tokenizer("AMINO20", 1, 0, 1), oracle.OracleTokenizer("AMINO20", 1, 0, 1)
    seqs, chars, offs = _seqs(5, 300, 0, P - 1)
    devs = _devices(2)
    a = sharding.encode_on_devices(tok, (chars, offs), P, devices=devs, op="tokenize", batch_first=True, root=devs[0])
    assert a.cpu().numpy().tobytes() == ora.tokenize_packed(chars, offs, P, "B", True).tobytes()
    mixed = [s.decode() if i % 3 == 0 else (bytearray(s) if i % 3 == 1 else np.frombuffer(s, dtype=np.uint8)) for i, s in enumerate(seqs)]
    b = sharding.encode_on_devices(tok, mixed, P, devices=devs, op="tokenize", batch_first=True, root=devs[0])
    assert torch.equal(a, b)
    bad = list(seqs)
    bad[150] = b"A" * P
    with pytest.raises(RuntimeError, match="seq len \\+ bos \\+ eos > padlen: %d, vs padlen %d" % (P + 1, P)):
        sharding.encode_on_devices(tok, bad, P, devices=devs, op="tokenize")
    with pytest.raises(ValueError, match="seq len"):
        sharding.encode_on_devices(tok, bad, P, devices=devs, op="onehot")
    with pytest.raises(ValueError):
        sharding.encode_on_devices(tok, seqs, P, devices=[], op="onehot")
    with pytest.raises(ValueError):
        sharding.encode_on_devices(tok, seqs, P, devices=devs, op="decode")


def test_a_consumer_on_its_own_stream_sees_finished_shards(gpu, bsq, oracle):
    """the hand-over is by events: a consumer that reads every shard right away on the device's current stream (here a side stream)
    gets the finished bytes without any host synchronisation in between; repeated calls reuse the stream pairs"""
    import torch
    from bioseq_amd import sharding
    P = 256
    seqs, chars, offs = _seqs(9, 20000, 0, P - 2)
    tok, ora = bsq.Tokenizer("SEB8", 1, 1, 1), oracle.OracleTokenizer("SEB8", 1, 1, 1)
    want = ora.tokenize_packed(chars, offs, P, "B", True)
    devs = _devices(4)
    mine = torch.cuda.Stream(device=gpu)
    for rep in range(3):
        with torch.cuda.stream(mine):
            shards = sharding.encode_on_devices(tok, seqs, P, devices=devs, op="tokenize", batch_first=True)
            sums = [s.to(torch.int64).sum() for s in shards]
        mine.synchronize()
        got = np.concatenate([s.cpu().numpy() for s in shards])
        assert got.tobytes() == want.tobytes()
        assert int(sum(int(x) for x in sums)) == int(want.astype(np.int64).sum())


def test_facade_devices_keyword(gpu, bsq, oracle):
    """bioseq.onehot_encode(..., to_pytorch=True, devices=[...]): the list of per-device (padlen, B_g, C) shards -- (B_g, padlen, C) views
    with batch_first -- equal to the reference call's result cut by sequence"""
    from bioseq_amd import sharding
    seqs, _, _ = _seqs(3, 999, 0, 60)
    tok, ora = bsq.pbeos_tokenizers["DNA"], oracle.OracleTokenizer("DNA", 1, 1, 1)
    want = ora.batch_onehot_encode(seqs, padlen=64, destchar="f")
    devs = _devices(3)
    got = bsq.onehot_encode(tok, seqs, padlen=64, destchar="f", to_pytorch=True, devices=devs)
    gotb = bsq.f_encode(seqs, key="DNA", bos=True, eos=True, padchar=True, padlen=64, destchar="f", batch_first=True, to_pytorch=True, devices=devs)
    for g in range(3):
        b0, b1 = sharding.shard_bounds(999, 3, g)
        assert got[g].cpu().numpy().tobytes() == np.ascontiguousarray(want[:, b0:b1]).tobytes()
        assert gotb[g].shape == (b1 - b0, 64, want.shape[2])
        assert gotb[g].cpu().numpy().tobytes() == np.ascontiguousarray(want[:, b0:b1].transpose(1, 0, 2)).tobytes()
    with pytest.raises(ValueError):
        bsq.onehot_encode(tok, seqs, to_pytorch=True, devices=devs)   # padlen must be explicit


def test_tokenizer_devices_keyword(gpu, bsq, oracle):
    """Tokenizer.batch_tokenize / batch_onehot_encode(..., devices=[...]) (keyword-only, beside device=): the per-device shards of the
    reference call's result cut by sequence; with device= (the root) the whole-batch tensor there; a mask is refused, and the checks of
    the one-device call (padlen, layout, destchar, an item that is too long) come in the same order with the same text"""
    import torch
    from bioseq_amd import sharding
    seqs, _, _ = _seqs(17, 1501, 0, 94)
    tok, ora = bsq.Tokenizer("PROTEIN", 1, 1, 1), oracle.OracleTokenizer("PROTEIN", 1, 1, 1)
    devs = _devices(3)
    wt = ora.batch_tokenize(seqs, padlen=96)
    wo = ora.batch_onehot_encode(seqs, padlen=96, destchar="f")
    gt = tok.batch_tokenize(seqs, padlen=96, devices=devs)
    gtb = tok.batch_tokenize(seqs, 96, "B", True, devices=devs)
    go = tok.batch_onehot_encode(seqs, padlen=96, destchar="f", devices=devs)
    gob = tok.batch_onehot_encode(seqs, padlen=96, destchar="f", devices=devs, layout="bcl")
    for g in range(3):
        b0, b1 = sharding.shard_bounds(len(seqs), 3, g)
        assert gt[g].cpu().numpy().tobytes() == np.ascontiguousarray(wt[:, b0:b1]).tobytes()
        assert gtb[g].cpu().numpy().tobytes() == np.ascontiguousarray(wt[:, b0:b1].T).tobytes()
        assert go[g].cpu().numpy().tobytes() == np.ascontiguousarray(wo[:, b0:b1]).tobytes()
        assert gob[g].cpu().numpy().tobytes() == np.ascontiguousarray(wo[:, b0:b1].transpose(1, 2, 0)).tobytes()
    whole = tok.batch_tokenize(seqs, padlen=96, device="cuda:0", devices=devs)
    assert isinstance(whole, torch.Tensor) and whole.cpu().numpy().tobytes() == wt.tobytes()
    whole = tok.batch_onehot_encode(seqs, padlen=96, destchar="f", device="cuda:0", devices=devs)
    assert whole.cpu().numpy().tobytes() == wo.tobytes()
    # nthreads > 1 reaches the host pack as given
    gt2 = tok.batch_tokenize(seqs, padlen=96, nthreads=4, devices=devs)
    assert all(a.cpu().numpy().tobytes() == b.cpu().numpy().tobytes() for a, b in zip(gt, gt2))
    with pytest.raises(ValueError, match="padlen"):
        tok.batch_tokenize(seqs, devices=devs)
    with pytest.raises(ValueError, match="mask"):
        tok.batch_onehot_encode(seqs, padlen=96, mask=[1] * len(seqs), devices=devs)
    with pytest.raises(ValueError, match="layout"):
        tok.batch_onehot_encode(seqs, padlen=96, layout="lbc", devices=devs)
    with pytest.raises(RuntimeError):
        tok.batch_tokenize(seqs, padlen=40, devices=devs)          # (tokenize.h:456-459: runtime_error)
    with pytest.raises(ValueError):
        tok.batch_onehot_encode(seqs, padlen=40, devices=devs)     # (tokenize.h:359-362: invalid_argument)
