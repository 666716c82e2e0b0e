"""Host time and GPU-complete time of blosum.augment_tokenize_packed on small batches (the loader's per-batch call with augmentation)."""
import statistics
import sys
import time

import torch

sys.path.insert(0, ".")
import bioseq_amd as bsq  # noqa: E402
from bioseq_amd import blosum, capi, synth  # noqa: E402

lib = capi.load()
tok = bsq.Tokenizer("SEB8", True, True, True)
for B in (64, 256, 512, 1024, 4096, 16384):
    chars, offs = synth.synth_packed(3, B, 30, 512, synth.AA)
    dch, dof = torch.from_numpy(chars).cuda(), torch.from_numpy(offs).cuda()
    for fused in (0, 1):
        capi.check(lib.bsq_tuning_set(b"augment_fused", fused))
        for sync in (False, True):
            for _ in range(20):
                blosum.augment_tokenize_packed(tok, dch, dof, 514, "b", True, chain_len=1, augment_frac=0.5, seed=1)
            torch.cuda.synchronize()
            ts = []
            for i in range(300):
                t0 = time.perf_counter()
                r = blosum.augment_tokenize_packed(tok, dch, dof, 514, "b", True, chain_len=1, augment_frac=0.5, seed=i)
                if sync:
                    torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
            torch.cuda.synchronize()
            print(f"B={B:6d} augment_fused={fused} {'call + sync' if sync else 'call       '}: {statistics.median(ts) * 1e6:7.1f} us", flush=True)
capi.check(lib.bsq_tuning_set(b"augment_fused", 0))
