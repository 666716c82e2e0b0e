"""cfg5 + augmentation, four cold batches per call (bsq_augment_tokenize_device_multi): the knobs of the augmentation role, and its pieces."""
import os, sys, json, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import importlib.util
spec = importlib.util.spec_from_file_location("b", os.path.join(ROOT, "bench.py")); m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
import torch
from bioseq_amd import capi
lib = capi.load()
dev = torch.device("cuda:0")
stream = torch.cuda.current_stream()
for ak in (0, 2, 1):
    capi.check(lib.bsq_tuning_set(b"augment_k", ak))
    b = m.Batch("cfg5aug", lib, dev, stream)
    r = m.cold_regime(b, 50, 0.5, stream)
    print("augment_k", ak, "one batch per call %.2f us (frac %.3f)" % (r["ms_per_step"] * 1e3, r["frac"]), "| four per call %.2f us per batch (frac %.3f)" % (r["multi4"]["ms_per_step"] * 1e3, r["multi4"]["frac"]))
    # the augmentation alone, cold, 4 batches per launch and 1
    del b
    torch.cuda.empty_cache()
capi.check(lib.bsq_tuning_set(b"augment_k", 0))
