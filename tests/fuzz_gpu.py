#!/usr/bin/env python3
"""Randomised differential test on the GPU: product (all kernel paths, all entry points) vs the CPU oracle.
   python tests/fuzz_gpu.py [seconds] [seed]      -- prints a one-line summary; exit 1 on the first mismatch."""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bioseq_amd as bsq
from bioseq_amd import capi, synth
from oracle import oracle as O

def run(budget=120.0, seed=1):
    rng = np.random.default_rng(seed)
    lib = capi.load()
    KEYS = O.keys()
    ALPH = [synth.DIRTY, synth.AA, "ACGT", "ACGTNacgtn", synth.DIRTY + "".join(chr(c) for c in range(33, 64))]
    dev = torch.device("cuda:0")
    n = 0
    ran = {"multi-batch": 0, "gather": 0, "encode_on_devices": 0, "device decode": 0, "loader epoch": 0, "host lists": 0}
    t_end = time.time() + budget
    while time.time() < t_end:
        key = KEYS[rng.integers(len(KEYS))]
        eos, bos, pad = (int(x) for x in rng.integers(0, 2, 3))
        B = int(rng.choice([1, 2, 7, 63, 64, 65, 200, 255, 256, 257, 1000, 1001, 4096, 5000, 20000, 20001, 33000]))
        P = int(rng.choice([1, 3, 15, 16, 17, 63, 64, 65, 100, 128, 255, 300, 512])) + eos + bos
        if rng.random() < 0.35:  # 128 and up: the (B,P) int8 kernel k_tokens_bp8 applies (its row-piece form when P % 16 != 0)
            P = int(rng.choice([128, 129, 135, 144, 160, 250, 272, 500, 512, 1000, 1001, 1024, 2048, 4111, 4112]))
        hi = P - eos - bos
        lo = int(rng.integers(0, hi + 1))
        if B * P > 6_000_000:
            B = max(1, 6_000_000 // P)
        d = "bhiqfd"[rng.integers(6)]
        if key == "BYTES" and d == "b":
            d = "h"
        chars, offs = synth.synth_packed(int(rng.integers(1 << 30)), B, lo, hi, ALPH[rng.integers(len(ALPH))])
        use_mask = rng.random() < 0.3
        mask = (rng.random(chars.size) < 0.7).astype(np.uint8) if use_mask else None
        path = int(rng.integers(0, 4))
        capi.check(lib.bsq_tuning_set(b"onehot_path", path))
        capi.check(lib.bsq_tuning_set(b"tokenize_path", int(rng.integers(0, 3))))
        capi.check(lib.bsq_tuning_set(b"tokenize_tb", int(rng.choice([0, 64, 128, 256]))))
        knobs = (int(rng.integers(0, 6)), int(rng.choice([0, 0, 1, 2])), int(rng.integers(0, 3)), int(rng.integers(0, 3)))
        capi.check(lib.bsq_tuning_set(b"tile_order", knobs[0]))
        capi.check(lib.bsq_tuning_set(b"tokens8_lookup", knobs[2]))
        capi.check(lib.bsq_tuning_set(b"tokens8", knobs[3]))
        capi.check(lib.bsq_tuning_set(b"tokens8_fast", int(rng.integers(0, 2))))
        capi.check(lib.bsq_tuning_set(b"raw_mode", int(rng.choice([0, 1, 4]))))
        capi.check(lib.bsq_tuning_set(b"tokens_pb8", int(rng.choice([0, 0, 0, 1, 2, 3]))))
        capi.check(lib.bsq_tuning_set(b"bcl_path", int(rng.integers(0, 4))))
        capi.check(lib.bsq_tuning_set(b"expand_rows1", int(rng.choice([0, 0, 1, 2]))))
        capi.check(lib.bsq_tuning_set(b"raw_nibbles", int(rng.choice([0, 1, 2, 2]))))
        capi.check(lib.bsq_tuning_set(b"two_pass_slice_mb", int(rng.choice([0, 0, 1, -1]))))
        capi.check(lib.bsq_tuning_set(b"tokens_pb8_pair", int(rng.choice([0, 0, 1]))))
        tok, ora = bsq.Tokenizer(key, eos, bos, pad), O.OracleTokenizer(key, eos, bos, pad)
        shift = int(rng.integers(0, 4))  # misaligned device views of the inputs
        dch = torch.from_numpy(np.concatenate([np.zeros(shift, np.uint8), chars])).to(dev)[shift:]
        dof = torch.from_numpy(offs).to(dev)
        dm = None if mask is None else torch.from_numpy(np.concatenate([np.zeros(5 - shift, np.uint8), mask])).to(dev)[5 - shift:]
        desc = (key, eos, bos, pad, B, P, lo, hi, d, path, use_mask, shift, knobs)
        with open("/tmp/fuzz_last.txt", "w") as f:  # survives a GPU fault that kills the process
            f.write(repr(desc) + " total_chars=%d\n" % chars.size)
        try:
            bf = bool(rng.integers(0, 2))
            e = ora.tokenize_packed(chars, offs, P, d, bf)
            u64 = (lambda a: a.view(np.uint64) if d in "lq" else a)  # 'q' device tensors are torch.int64 (same bits)
            g = u64(tok.tokenize_packed(dch, dof, P, d, bf).cpu().numpy())
            assert g.dtype == e.dtype and g.tobytes() == e.tobytes(), ("tokenize", bf)
            e = ora.onehot_packed(chars, offs, P, d, mask=mask)
            g = u64(tok.onehot_packed(dch, dof, P, d, mask=dm).cpu().numpy())
            assert g.dtype == e.dtype and g.shape == e.shape and g.tobytes() == e.tobytes(), "onehot"
            g = tok.onehot_packed(dch, dof, P, d, mask=dm, layout="bcl").cpu().numpy()
            assert g.tobytes() == np.ascontiguousarray(e.transpose(1, 2, 0)).tobytes(), "onehot bcl"
            if mask is None and rng.random() < 0.35:  # the seq-first one-hot written as COLUMN BLOCKS of the whole tensor, any cuts, any alignment
                dt = ctypes.c_int(0)
                capi.check(lib.bsq_dtype_from_destchar(d.encode(), ctypes.byref(dt)))
                cdesc = capi.make_desc(key, eos, bos, pad)
                rowb = e.shape[2] * e.dtype.itemsize
                cuts = sorted(set([0, B] + [int(x) for x in rng.integers(0, B + 1, int(rng.integers(1, 4)))]))
                root = torch.full((e.nbytes + 32768,), 0x5A, dtype=torch.uint8, device=dev)   # (room for the largest shift: 2560 x 8 bytes)
                base = (-root.data_ptr()) % 4096 + int(rng.choice([0, 0, 512, 2560, 16, 1])) * e.dtype.itemsize
                for b0, b1 in zip(cuts[:-1], cuts[1:]):
                    sub = dof[b0:b1 + 1].contiguous()   # (absolute offsets into dch)
                    capi.check(lib.bsq_onehot_block_device(ctypes.byref(cdesc), dch.data_ptr(), sub.data_ptr(), None, b1 - b0, P, dt,
                                                           root.data_ptr() + base + b0 * rowb, B, None))
                h = root.cpu().numpy()
                if h[base:base + e.nbytes].tobytes() != e.tobytes():   # which block, where, under which knobs
                    got, exp = h[base:base + e.nbytes].reshape(e.shape[0], -1), np.frombuffer(e.tobytes(), np.uint8).reshape(e.shape[0], -1)
                    where = [(b0, b1, np.argwhere(got[:, b0 * rowb:b1 * rowb] != exp[:, b0 * rowb:b1 * rowb])[[0, -1]].tolist(),
                              int((got[:, b0 * rowb:b1 * rowb] != exp[:, b0 * rowb:b1 * rowb]).sum()))
                             for b0, b1 in zip(cuts[:-1], cuts[1:]) if not np.array_equal(got[:, b0 * rowb:b1 * rowb], exp[:, b0 * rowb:b1 * rowb])]
                    state = {k.decode(): lib.bsq_tuning_get(k) for k in (b"onehot_path", b"tile_order", b"raw_mode", b"tokens_pb8", b"expand_rows1", b"raw_nibbles", b"two_pass_slice_mb", b"tokens8_lookup")}
                    raise AssertionError(("column blocks", cuts, base % 4096, rowb, where, state))
                assert (h[:base] == 0x5A).all() and (h[base + e.nbytes:] == 0x5A).all(), ("column blocks wrote outside the root", cuts)
            if rng.random() < 0.3:  # augmentation + tokens in one call == the two calls (fused launch or not, any shape / type / layout)
                from bioseq_amd import blosum
                cl, fr, sd = int(rng.integers(0, 4)), float(rng.choice([0.3, 0.5, 1.0])), int(rng.integers(1 << 30))
                capi.check(lib.bsq_tuning_set(b"augment_fused", int(rng.integers(0, 5))))
                a1, a2 = dch.clone(), dch.clone()
                blosum.augment_packed(a1, dof, cl, fr, sd)
                t1 = tok.tokenize_packed(a1, dof, P, d, bf, validate=False)
                t2 = blosum.augment_tokenize_packed(tok, a2, dof, P, d, bf, chain_len=cl, augment_frac=fr, seed=sd)
                assert torch.equal(a1, a2) and u64(t1.cpu().numpy()).tobytes() == u64(t2.cpu().numpy()).tobytes(), ("augment+tokenize", cl, fr, bf)
            if d in "bh" and rng.random() < 0.3:  # round 6: the batch cut into 1 ... 10 pieces, all of them in ONE multi-batch call (multi kernels
                from bioseq_amd import multi            # where every piece qualifies, single launches otherwise; empty pieces allowed)
                ran["multi-batch"] += 1
                cuts = sorted([0, B] + [int(x) for x in rng.integers(0, B + 1, int(rng.integers(0, 10)))])
                if rng.random() < 0.5:  # multiples of 64 sequences: what the (P,B) multi kernel takes
                    cuts = sorted(set([0, B] + [c // 64 * 64 for c in cuts]))
                parts = [(dch[int(offs[b0]):int(offs[b1])] if offs[b1] > offs[b0] else dch[:0], (dof[b0:b1 + 1] - dof[b0]).contiguous())
                         for b0, b1 in zip(cuts[:-1], cuts[1:])]
                got = multi.tokenize_packed_multi(tok, parts, P, d, bf)
                whole = ora.tokenize_packed(chars, offs, P, d, bf)
                for (b0, b1), g in zip(zip(cuts[:-1], cuts[1:]), got):
                    w = whole[b0:b1] if bf else whole[:, b0:b1]
                    assert g.cpu().numpy().tobytes() == np.ascontiguousarray(w).tobytes(), ("multi tokens", bf, cuts, b0, b1)
                if d == "b" and bf and rng.random() < 0.5:  # ... and with augmentation: == the per-piece calls with the same seeds
                    from bioseq_amd import blosum
                    seeds = [int(x) for x in rng.integers(1 << 30, size=len(parts))]
                    p1 = [(c.clone(), o) for c, o in parts]
                    p2 = [(c.clone(), o) for c, o in parts]
                    want = [blosum.augment_tokenize_packed(tok, c, o, P, "b", True, chain_len=1, augment_frac=0.5, seed=sd_) if o.numel() > 1 else None
                            for (c, o), sd_ in zip(p1, seeds)]
                    got = multi.augment_tokenize_packed_multi(tok, p2, P, "b", True, chain_len=1, augment_frac=0.5, seeds=seeds, validate=False)
                    for (c1, _), (c2, _), w, g in zip(p1, p2, want, got):
                        assert torch.equal(c1, c2) and (w is None or torch.equal(w, g)), ("multi augment + tokens", cuts)
            if rng.random() < 0.15:  # round 6: index batches of every size class of the gather (one launch <= 4096, two beyond, three under the knob)
                nidx = int(rng.choice([1, 100, 4096, 4097, 9000, 20000, 66000]))
                ran["gather"] += 1
                idx = rng.integers(0, B, size=nidx).astype(np.int64)
                capi.check(lib.bsq_tuning_set(b"gather_small", int(rng.choice([0, 0, 1]))))
                lens_ = np.diff(offs)
                woffs = np.concatenate([[0], np.cumsum(lens_[idx])]).astype(np.int64)
                cap = int(woffs[-1])
                oc = torch.full((cap + 64,), 0xEE, dtype=torch.uint8, device=dev)
                oo = torch.empty(nidx + 1, dtype=torch.int64, device=dev)
                stt = torch.empty(1, dtype=torch.int64, device=dev)
                capi.check(lib.bsq_gather_packed_device(dch.data_ptr(), dof.data_ptr(), B, torch.from_numpy(idx).to(dev).data_ptr(), nidx, oc.data_ptr(), cap,
                                                        oo.data_ptr(), stt.data_ptr(), None))
                capi.check(lib.bsq_tuning_set(b"gather_small", 0))
                assert int(stt.item()) == -1 and (oo.cpu().numpy() == woffs).all(), ("gather offsets", nidx)
                pos = np.repeat(offs[:-1][idx] - woffs[:-1], lens_[idx]) + np.arange(cap)
                hc = oc.cpu().numpy()
                assert (hc[:cap] == chars[pos]).all() and (hc[cap:] == 0xEE).all(), ("gather characters", nidx)
            if B <= 20001 and rng.random() < 0.12:  # round 6: ONE process, several device entries (stream pairs of this GPU): shards and a root tensor
                from bioseq_amd import sharding
                ran["encode_on_devices"] += 1
                seqs = synth.unpack(chars, offs)
                G = int(rng.integers(1, 6))
                devs = ["cuda:0"] * G
                op = "tokenize" if rng.random() < 0.5 else "onehot"
                lay = "bcl" if rng.random() < 0.5 else "tbc"
                e0 = e if mask is None else ora.onehot_packed(chars, offs, P, d)  # (no mask on this path)
                want = ora.tokenize_packed(chars, offs, P, d, bf) if op == "tokenize" else (np.ascontiguousarray(e0.transpose(1, 2, 0)) if lay == "bcl" else e0)
                full = sharding.encode_on_devices(tok, seqs, P, d, devices=devs, op=op, batch_first=bf, layout=lay, root="cuda:0")
                torch.cuda.synchronize()
                assert u64(full.cpu().numpy()).tobytes() == want.tobytes(), ("encode_on_devices root", op, lay, bf, G)
                shards = sharding.encode_on_devices(tok, seqs, P, d, devices=devs, op=op, batch_first=bf, layout=lay)
                torch.cuda.synchronize()
                ax = (0 if bf else 1) if op == "tokenize" else (0 if lay == "bcl" else 1)
                cat = np.concatenate([u64(x.cpu().numpy()) for x in shards], axis=ax)
                assert np.ascontiguousarray(cat).tobytes() == want.tobytes(), ("encode_on_devices shards", op, lay, bf, G)
            if d in "bhiq" and B * P <= 2_000_000 and rng.random() < 0.1:  # device decode (any strides) == the host decode of the same tokens
                ran["device decode"] += 1
                ed = ora.tokenize_packed(chars, offs, P, d, bf)
                ed = ed if d != "q" else ed.view(np.int64)
                dt_ = torch.from_numpy(ed).to(dev)
                assert tok.decode_tokens(dt_) == tok.decode_tokens(ed), ("decode on the device", bf)
                assert tok.decode_tokens(dt_.T) == tok.decode_tokens(np.ascontiguousarray(ed.T)), ("decode on the device, transposed view", bf)
                if ed.shape[1] > 1:
                    assert tok.decode_tokens(dt_[:, ::2]) == tok.decode_tokens(np.ascontiguousarray(ed[:, ::2])), ("decode on the device, strided view", bf)
            if 2 <= B <= 5000 and int(np.diff(offs).max()) + eos + bos >= 1 and rng.random() < 0.08:
                # a loader epoch out of a FlatFile resident in HBM: any batch size, shuffled or not, drop_last, groups of batches, prefetch
                import tempfile
                ran["loader epoch"] += 1
                from bioseq_amd.flatfile import FlatFile, write_flatfile
                from bioseq_amd.loaders import FlatFileDataset
                seqs = synth.unpack(chars, offs)
                with tempfile.TemporaryDirectory() as td:
                    ff = FlatFile(write_flatfile(seqs, os.path.join(td, "s.ff")))
                    bs = int(rng.choice([1, 7, 64, 100, 256, 1000]))
                    bs = max(bs, -(-B // 150))  # (at most 150 batches per epoch: every one is read back and checked)
                    cnn = rng.random() < 0.3 and bs * (int(np.diff(offs).max()) + 2) * tok.alphabet_size() * 4 <= (64 << 20)
                    tdt = "q" if (key == "BYTES" or rng.random() < 0.5) else "b"
                    ds = FlatFileDataset(ff, tok, cnn=cnn, device=dev, token_dtype=tdt)
                    PP = ds.max_seq_len
                    shuffle, drop_last = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
                    grp, pf, gs = int(rng.integers(1, 6)), int(rng.choice([0, 0, 1, 3])), int(rng.integers(1 << 30))
                    perm = (torch.randperm(B, device=dev, generator=torch.Generator(device=dev).manual_seed(gs)).cpu().tolist() if shuffle else list(range(B)))
                    k = 0
                    for batch in ds.batches(bs, shuffle=shuffle, drop_last=drop_last, generator=torch.Generator(device=dev).manual_seed(gs), prefetch=pf, group=grp):
                        part = [seqs[i] for i in perm[k * bs:(k + 1) * bs]]
                        if cnn:
                            want = np.ascontiguousarray(ora.batch_onehot_encode(part, padlen=PP, destchar="f").transpose(1, 2, 0))
                        else:
                            want = ora.batch_tokenize(part, padlen=PP, destchar=tdt, batch_first=True)
                        assert batch.cpu().numpy().tobytes() == want.tobytes(), ("loader epoch", cnn, tdt, bs, shuffle, drop_last, grp, pf, k)
                        k += 1
                    assert k == (B // bs if drop_last else -(-B // bs)), ("loader epoch: number of batches", bs, drop_last, grp, pf, k)
                    del ds, ff
            if B * P < 400_000:  # host entry points (list of bytes -> numpy)
                ran["host lists"] += 1
                seqs = synth.unpack(chars, offs)
                ml = None if mask is None else [mask[offs[i]:offs[i + 1]].copy() for i in range(B)]
                g = tok.batch_onehot_encode(seqs, padlen=P, destchar=d, mask=ml)
                assert g.tobytes() == e.tobytes(), "host onehot"
            if B >= 16384 and rng.random() < 0.6:  # list -> DEVICE result: the staged path (scan + pack + upload + encode in pieces)
                seqs = synth.unpack(chars, offs, as_str=bool(rng.integers(0, 2)))  # (every fuzz alphabet is ASCII)
                ml = None if mask is None else [mask[offs[i]:offs[i + 1]].copy() for i in range(B)]
                capi.check(lib.bsq_tuning_set(b"host_pieces", int(rng.choice([0, 1, 2, 3, 4, 8]))))
                if rng.random() < 0.5:
                    torch.cuda.synchronize()  # (an idle stream: the automatic rule splits)
                g = u64(tok.batch_onehot_encode(seqs, padlen=P, destchar=d, mask=ml, device="cuda").cpu().numpy())
                assert g.tobytes() == e.tobytes(), "list -> device onehot"
                g = tok.batch_onehot_encode(seqs, padlen=P, destchar=d, mask=ml, device="cuda", layout="bcl").cpu().numpy()
                assert g.tobytes() == np.ascontiguousarray(e.transpose(1, 2, 0)).tobytes(), "list -> device onehot bcl"
                et = ora.tokenize_packed(chars, offs, P, d, bf)
                g = u64(tok.batch_tokenize(seqs, padlen=P, destchar=d, batch_first=bf, device="cuda").cpu().numpy())
                assert g.tobytes() == et.tobytes(), ("list -> device tokens", bf)
                # numpy results (<= 256 MB: pieces fetched back while the next ones go up)
                g = tok.batch_tokenize(seqs, padlen=P, destchar=d, batch_first=bf)
                assert g.dtype == et.dtype and g.shape == et.shape and g.tobytes() == et.tobytes(), ("list -> numpy tokens", bf)
                g = tok.batch_onehot_encode(seqs, padlen=P, destchar=d, mask=ml, layout="bcl" if bf else "tbc")
                assert g.tobytes() == (np.ascontiguousarray(e.transpose(1, 2, 0)) if bf else e).tobytes(), ("list -> numpy onehot", bf)
        except AssertionError as ex:
            raise AssertionError("MISMATCH %s %r" % (ex, desc))
        except Exception as ex:  # (an entry point that REFUSES a valid configuration is a finding too: name the configuration)
            raise AssertionError("ERROR %s: %s %r (configuration %d of seed %d)" % (type(ex).__name__, ex, desc, n, seed)) from ex
        n += 1
    torch.cuda.synchronize()
    capi.check(lib.bsq_fused_status(None))  # no token wave of a fused augmentation launch gave up waiting
    for name in (b"onehot_path", b"tokenize_path", b"tile_order", b"tokens8_lookup", b"tokens8", b"tokens8_fast", b"raw_mode", b"tokens_pb8", b"augment_fused", b"bcl_path", b"tokenize_tb", b"host_pieces", b"expand_rows1", b"raw_nibbles", b"two_pass_slice_mb", b"gather_small", b"tokens_pb8_pair"):
        capi.check(lib.bsq_tuning_set(name, 0))
    return n, ran


if __name__ == "__main__":
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    try:
        n, ran = run(budget, seed)
    except AssertionError as ex:
        print(ex, flush=True)
        sys.exit(1)
    print("fuzz ok: %d random configurations, %.0f s, all bit-exact vs the oracle" % (n, budget))
    print("  of which also: " + ", ".join("%s %d" % kv for kv in ran.items()))
