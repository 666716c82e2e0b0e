"""list[bytes] -> seq-first one-hot tensor on the device (cfg3 batch), synchronous and back to back, for values of the host_pieces
knob (1 = the whole-batch path of rounds 1-3: scan, pack, one upload, one encode; N = N pieces; 0 = automatic), interleaved over
REPS rounds so that box drift hits every setting alike.  Prints medians over the rounds; run on the GPU box:
    python3 scripts/host_pieces_lab.py > gpurun_out/r04/host_pieces_lab.txt"""
import statistics
import sys
import time

import torch

sys.path.insert(0, ".")
import bioseq_amd as bsq  # noqa: E402
from bioseq_amd import capi, synth  # noqa: E402

lib = capi.load()
B, P = 65536, 1024
REPS = 7
chars, offs = synth.synth_packed(1, B, 50, 1024, synth.AA)
items = [bytes(chars[offs[i]:offs[i + 1]]) for i in range(B)]
tok = bsq.Tokenizer("AMINO20")


def sync_ms(call, n=9):
    ts = []
    r = None
    for _ in range(n):
        del r
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = call()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    return statistics.median(ts), r


def pipelined_ms(call, n=20):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        r = call()
        del r
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / n


CALLS = {
    "one-hot f32 (P,B,C)": lambda: tok.batch_onehot_encode(items, padlen=P, destchar="f", device="cuda"),
    "one-hot int8 (P,B,C)": lambda: tok.batch_onehot_encode(items, padlen=P, destchar="b", device="cuda"),
    "one-hot f32 (B,C,P)": lambda: tok.batch_onehot_encode(items, padlen=P, destchar="f", device="cuda", layout="bcl"),
    "tokens int8 (B,P)": lambda: tok.batch_tokenize(items, padlen=P, destchar="b", batch_first=True, device="cuda"),
    "tokens int8 (P,B)": lambda: tok.batch_tokenize(items, padlen=P, destchar="b", device="cuda"),
}
KNOBS = (1, 4, 8, 0)
for name, call in CALLS.items():
    res = {k: ([], []) for k in KNOBS}
    ref = None
    for rep in range(REPS):
        for knob in KNOBS:
            capi.check(lib.bsq_tuning_set(b"host_pieces", knob))
            for _ in range(2):
                call()
            med, r = sync_ms(call)
            fold = int(r.view(torch.uint8).to(torch.int64).sum().item())
            ref = fold if ref is None else ref
            assert fold == ref, "results differ between piece counts"
            del r
            res[knob][0].append(med)
            res[knob][1].append(pipelined_ms(call))
    for knob in KNOBS:
        s, p = res[knob]
        print(f"{name:22s} host_pieces={knob}: synchronous {statistics.median(s):.3f} ms (min {min(s):.3f}), "
              f"back to back x20 {statistics.median(p):.3f} ms/batch (min {min(p):.3f})", flush=True)
capi.check(lib.bsq_tuning_set(b"host_pieces", 0))
