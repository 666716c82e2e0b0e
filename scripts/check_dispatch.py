#!/usr/bin/env python3
"""Is the automatic one-hot path within 5 % of the best forced path?  (the dispatch of bsq_onehot_device is a
hand-grown threshold tree -- bsq_onehot.hip, choose_onehot_path / two_pass_plan -- justified by sweeps on individual boxes.)

    python3 scripts/check_dispatch.py [--gate] [--tolerance 0.05] [--shapes scripts/dispatch_shapes.json] [--report-only] [--json out.json]

For every shape of the table: the batch is synthesised (bioseq_amd.synth), encoded through the C ABI with the automatic choice and with
every forced setting of the knobs `onehot_path` (1 tiled, 2 two-pass, 3 chunk-owner) and, for the two-pass form, `raw_nibbles` (1 never,
2 whenever they apply) -- on FRESH inputs (K distinct copies of the batch in turn, K * input >= 512 MiB, 2 <= K <= 6: the paths differ most
when the characters come from HBM), HIP events on the launch stream, mean of 3 K launches after 2 K warm-up launches, two rounds over all
settings (the second in reverse order; a setting's time is its better round; 30 ms of launches first lift the clocks out of idle).  Every forced output must
equal the automatic one bit for bit (they are all checked against the oracle elsewhere; here a forced path that differs is a FAILURE whatever
its speed).  Exit code 1 when the automatic choice is more than `tolerance` behind the best forced setting on any shape (boxes differ by
+-3 %, small kernels by 8 %: --report-only never fails on speed; the -m gpu test runs the `gate` subset that way)."""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

VARIANTS = [("auto", {}), ("tile", {"onehot_path": 1}), ("two_pass", {"onehot_path": 2}), ("two_pass_bytes", {"onehot_path": 2, "raw_nibbles": 1}),
            ("two_pass_nibbles", {"onehot_path": 2, "raw_nibbles": 2}), ("chunks", {"onehot_path": 3})]


def run(shapes, tolerance=0.05, report_only=False, out=sys.stdout):
    import numpy as np
    import torch
    from bioseq_amd import capi, synth
    lib = capi.load()
    dev = torch.device("cuda:0")
    stream = torch.cuda.current_stream()
    sh = ctypes.c_void_p(stream.cuda_stream)
    results, bad = [], []
    for s in shapes:
        desc = capi.make_desc(s["key"], *s["flags"])
        C = lib.bsq_alphabet_size(ctypes.byref(desc))
        dt = ctypes.c_int(0)
        capi.check(lib.bsq_dtype_from_destchar(s["destchar"].encode(), ctypes.byref(dt)))
        sz = lib.bsq_dtype_size(dt)
        tdt = {0: torch.int8, 1: torch.int16, 2: torch.int32, 3: torch.int64, 4: torch.float32, 5: torch.float64}[dt.value]
        letters = synth.AA if s["key"][:3] not in ("DNA",) else "ACGT"
        B, P = s["B"], s["P"]
        chars, offs = synth.synth_packed(1234, B, s["lo"], min(s["hi"], P - s["flags"][0] - s["flags"][1]), letters)
        in_bytes = chars.size + offs.size * 8
        K = max(2, min(6, -(-(512 << 20) // max(in_bytes, 1))))
        d_offs = torch.from_numpy(offs).to(dev)
        base = torch.from_numpy(chars).to(dev)
        copies = [base] + [base.clone() for _ in range(K - 1)]
        out_t = torch.empty((P, B, C), dtype=tdt, device=dev)
        ref = None
        rowbytes, total = C * sz, P * B * C * sz
        algo = in_bytes + total
        times, names = {}, {}
        it = [0]

        def step():
            st = lib.bsq_onehot_device(ctypes.byref(desc), copies[it[0] % K].data_ptr(), d_offs.data_ptr(), None, B, P, dt, out_t.data_ptr(), sh)
            if st:
                capi.check(st)
            it[0] += 1

        todo = []
        for name, knobs in VARIANTS:
            if name == "chunks" and rowbytes < 16 and total > (256 << 20):
                continue  # (per-position gathers of tiny rows: 13x behind at cfg4b, profiles/r05/onehot_paths_small_rows.txt -- not worth the box time)
            if name.startswith("two_pass_") and C > 15:
                continue  # (nibble ids: alphabets of at most 15 classes)
            todo.append((name, knobs))
        # clocks out of idle before anything is timed (the first variant of a shape otherwise reads 5-10 % slow)
        t_end = time.perf_counter() + 0.03
        while time.perf_counter() < t_end:
            for _ in range(K):
                step()
            torch.cuda.synchronize()
        for rnd in range(2):  # two rounds over all variants, the second in reverse order; a variant's time = its better round
            for name, knobs in (todo if rnd == 0 else todo[::-1]):
                for k, v in knobs.items():
                    capi.check(lib.bsq_tuning_set(k.encode(), v))
                try:
                    if rnd == 0:
                        names[name] = lib.bsq_onehot_kernel_name(ctypes.byref(desc), B, P, dt).decode()
                        out_t.fill_(3)
                        step()
                        torch.cuda.synchronize()
                        if name == "auto":
                            ref = out_t.clone() if total <= (2 << 30) else None
                            ref_sum = float(out_t.sum(dtype=torch.float64))
                        else:
                            same = torch.equal(out_t, ref) if ref is not None else float(out_t.sum(dtype=torch.float64)) == ref_sum
                            if not same:
                                bad.append("%s: forced %s differs from the automatic path's output" % (s["name"], name))
                    for _ in range(2 * K):
                        step()
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record(stream)
                    for _ in range(3 * K):
                        step()
                    b.record(stream)
                    torch.cuda.synchronize()
                    t = a.elapsed_time(b) / (3 * K) * 1e3
                    times[name] = min(times.get(name, t), t)
                finally:
                    for k in knobs:
                        capi.check(lib.bsq_tuning_set(k.encode(), 0))
        best = min(times, key=times.get)
        behind = times["auto"] / times[best] - 1.0
        if times["auto"] - times[best] < 2.0:
            behind = min(behind, tolerance)  # (launches of a few microseconds differ by 1-2 us from run to run: no verdict below 2 us)
        verdict = "ok" if behind <= tolerance else "BEHIND"
        if behind > tolerance:
            bad.append("%s: automatic (%s, %.1f us) is %.1f %% behind forced %s (%s, %.1f us)" % (
                s["name"], names["auto"], times["auto"], behind * 100, best, names.get(best), times[best])) if not report_only else None
        results.append({"shape": s, "row_bytes": rowbytes, "output_bytes": total, "auto_kernel": names["auto"], "us": times, "best": best, "behind": behind,
                        "frac_auto": algo / times["auto"] / 1e3 / 8000.0})
        print("%-15s %-8s B=%-8d P=%-5d row %3d B out %7.2f GB | auto %8.1f us (%.3f) %-58s | %s | best %-16s %+5.1f %% %s" % (
            s["name"], s["key"], B, P, rowbytes, total / 1e9, times["auto"], results[-1]["frac_auto"], names["auto"][:58],
            " ".join("%s %.1f" % (k, v) for k, v in times.items() if k != "auto"), best, -behind * 100, verdict), file=out, flush=True)
        del copies, base, out_t, ref
        torch.cuda.empty_cache()
    return results, bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default=os.path.join(ROOT, "scripts", "dispatch_shapes.json"))
    ap.add_argument("--gate", action="store_true", help="only the shapes marked `gate` (what the -m gpu test runs)")
    ap.add_argument("--tolerance", type=float, default=0.05)
    ap.add_argument("--report-only", action="store_true")
    ap.add_argument("--json", default=None)
    args = ap.parse_args()
    shapes = json.load(open(args.shapes))["shapes"]
    if args.gate:
        shapes = [s for s in shapes if s.get("gate")]
    results, bad = run(shapes, args.tolerance, args.report_only)
    if args.json:
        json.dump(results, open(args.json, "w"), indent=1)
    worst = max(results, key=lambda r: r["behind"])
    print("# %d shapes; automatic choice behind the best forced path by at most %.1f %% (%s)" % (len(results), worst["behind"] * 100, worst["shape"]["name"]))
    for b in bad:
        print("FAIL:", b)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
