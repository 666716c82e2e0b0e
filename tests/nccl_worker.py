"""Worker of tests/test_multi_gpu.py: one rank of an N-rank job, ONE GPU PER RANK, backend nccl (= RCCL over xGMI).  Every whole-batch
assembly of bioseq_amd.sharding on DEVICE tensors produced by the HIP kernels -- all_gather forms, grouped point-to-point
(`gather_direct`, to a root and to every rank), token matrices + local expansion (`onehot_gathered`), and the encode kernels storing
straight into the root's IPC-mapped buffer (`store_shard_into_root`, three layouts) -- for equal, ragged and EMPTY shards, each result
compared bit for bit with the CPU oracle's encode of the whole batch.  Prints MULTI_GPU_OK on rank 0."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist
    import bioseq_amd
    from bioseq_amd import sharding, synth
    from oracle import oracle as O
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    # BSQ_TEST_BACKEND=gloo: the same script with the ranks SHARING one GPU and the shards crossing the process boundary as host
    # tensors (gloo moves no device memory) -- a rehearsal of this worker's own logic on a 1-GPU box, not a test of RCCL
    nccl = os.environ.get("BSQ_TEST_BACKEND", "nccl") == "nccl"
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")) % (torch.cuda.device_count() if not nccl else 1 << 30))
    torch.cuda.set_device(dev)
    if nccl:
        dist.init_process_group("nccl", device_id=dev)
    else:
        dist.init_process_group("gloo")
    ship = (lambda t: t) if nccl else (lambda t: t.cpu())
    ok = True
    report = {"backend": "nccl" if nccl else "gloo", "world_size": dist.get_world_size(), "rccl_version": None,
              "ipc_mode_legacy": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"), "gb_per_s_into_each_rank": {}}
    if nccl:
        try:
            report["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception as ex:  # noqa: BLE001
            report["rccl_version"] = repr(ex)

    def rate(name, fn, nbytes):
        """one assembly form once more, timed between two barriers (MAX over ranks is taken by rank 0's clock around the barriers)"""
        import time
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = fn()
        torch.cuda.synchronize()
        dist.barrier()
        dt = time.perf_counter() - t0
        report["gb_per_s_into_each_rank"][name] = nbytes / dt / 1e9
        return r
    tok, ora = bioseq_amd.Tokenizer("AMINO20", 1, 1, 1), O.OracleTokenizer("AMINO20", 1, 1, 1)
    # B: equal shards, ragged shards, fewer sequences than ranks (empty shards), one big batch
    for B, hi in ((world * 64, 120), (world * 50 + 1, 250), (max(1, world - 1), 60), (20011, 300)):
        chars, offs = synth.synth_packed(1000 + B, B, 0, hi, synth.DIRTY)
        P = (hi + 2 + 15) // 16 * 16
        full_oh = ora.onehot_packed(chars, offs, P, "f")
        full_bf = ora.tokenize_packed(chars, offs, P, "b", True)
        full_sf = ora.tokenize_packed(chars, offs, P, "b", False)
        enc_oh = lambda c, o: ship(tok.onehot_packed(torch.as_tensor(np.ascontiguousarray(c)).to(dev), torch.as_tensor(np.ascontiguousarray(o)).to(dev), P, "f"))
        enc_bf = lambda c, o: ship(tok.tokenize_packed(torch.as_tensor(np.ascontiguousarray(c)).to(dev), torch.as_tensor(np.ascontiguousarray(o)).to(dev), P, "b", True))
        enc_sf = lambda c, o: ship(tok.tokenize_packed(torch.as_tensor(np.ascontiguousarray(c)).to(dev), torch.as_tensor(np.ascontiguousarray(o)).to(dev), P, "b", False))

        def same(t, want):
            return t is not None and (t.is_cuda or not nccl) and t.cpu().numpy().tobytes() == want.tobytes()

        for gather, enc, want in (("onehot", enc_oh, full_oh), ("tokens_bf", enc_bf, full_bf), ("tokens_sf", enc_sf, full_sf),
                                  ("direct_onehot", enc_oh, full_oh), ("direct_tokens_bf", enc_bf, full_bf), ("direct_tokens_sf", enc_sf, full_sf)):
            ok = ok and same(sharding.encode_sharded(enc, chars, offs, gather=gather), want)
        keep_oh = sharding.encode_sharded(enc_oh, chars, offs)
        keep_bf = sharding.encode_sharded(enc_bf, chars, offs)
        for root in range(world):
            r = sharding.gather_direct(keep_oh.contiguous(), 1, B, root)
            ok = ok and (same(r, full_oh) if rank == root else r is None)
            r = sharding.gather_direct(keep_bf.contiguous(), 0, B, root)
            ok = ok and (same(r, full_bf) if rank == root else r is None)
            for dc, layout, want in (("b", "tokens_bf", full_bf), ("b", "tokens_sf", full_sf), ("f", "bcl", np.ascontiguousarray(full_oh.transpose(1, 2, 0))),
                                     ("f", "tbc", full_oh)):
                got = sharding.encode_into_root(tok, chars, offs, P, dc, layout, dev, root=root)
                ok = ok and (same(got, want) if rank == root else got is None)
                del got
                dist.barrier()
        raw_tokens, expand = sharding.device_passes(tok, P, "f", dev)
        if nccl:
            ok = ok and same(sharding.onehot_gathered(raw_tokens, expand, chars, offs), full_oh)
        else:
            ok = ok and same(sharding.onehot_gathered(lambda c, o: raw_tokens(c, o).cpu(), lambda t: expand(t.to(dev)), chars, offs), full_oh)
    # BASELINE config 4's split at 1/64 scale ("1M reads, 1 vs 8 GPU shard + RCCL gather": 15 625 reads of 150, DNA4 + BOS / EOS / PAD,
    # padlen 160, f32 -- 28-byte rows, shards that are not a multiple of anything): the point-to-point gather with its staging cut into
    # row groups (a cap far below the shard: dozens of messages per peer in flight group after group), to a root and to every rank; the
    # token-matrix gather; the kernels storing straight into the root's buffer -- and then the same store with ONE rank handed a
    # sequence that does not fit: the MIN-reduced ok flag must raise on EVERY rank, none may be left waiting
    tok4, ora4 = bioseq_amd.Tokenizer("DNA4", 1, 1, 1), O.OracleTokenizer("DNA4", 1, 1, 1)
    B4, P4 = 15625, 160
    chars4, offs4 = synth.synth_packed(404, B4, 150, 150, "ACGT")
    full4 = ora4.onehot_packed(chars4, offs4, P4, "f")
    enc4 = lambda c, o: ship(tok4.onehot_packed(torch.as_tensor(np.ascontiguousarray(c)).to(dev), torch.as_tensor(np.ascontiguousarray(o)).to(dev), P4, "f"))
    keep4 = sharding.encode_sharded(enc4, chars4, offs4).contiguous()
    for root in (None, 0, world - 1):
        for cap in (0, 1 << 20, 200000):
            r = sharding.gather_direct(keep4, 1, B4, root, stage_bytes=cap)
            want_here = root is None or rank == root
            ok = ok and ((r is not None and r.cpu().numpy().tobytes() == full4.tobytes()) if want_here else r is None)
            del r
    got = sharding.encode_into_root(tok4, chars4, offs4, P4, "f", "tbc", dev, root=0)
    ok = ok and ((got is not None and got.cpu().numpy().tobytes() == full4.tobytes()) if rank == 0 else got is None)
    del got
    dist.barrier()
    # per-form rates on this 70-MB tensor (bytes a rank RECEIVES / time between two barriers): small, but the first numbers RCCL gives this code
    recv = full4.nbytes * (world - 1) / world
    rate("all_gather", lambda: sharding.encode_sharded(enc4, chars4, offs4, gather="onehot"), recv)
    rate("direct_all", lambda: sharding.gather_direct(keep4, 1, B4, None), recv)
    rate("direct_root", lambda: sharding.gather_direct(keep4, 1, B4, 0), recv)
    rate("store_into_root", lambda: sharding.encode_into_root(tok4, chars4, offs4, P4, "f", "tbc", dev, root=0), recv)
    # ranks that disagree on stage_bytes must get an error on EVERY rank, not a hang
    differs = False
    try:
        sharding.gather_direct(keep4, 1, B4, None, stage_bytes=(3 << 20) + rank)
    except ValueError:
        differs = True
    ok = ok and (differs or world == 1)
    dist.barrier()
    bad_rank = world - 1
    c_bad, o_bad = sharding.shard_packed(chars4, offs4, world, rank)
    b0_bad = sharding.shard_bounds(B4, world, rank)[0]
    if rank == bad_rank:  # this rank's shard gets one sequence longer than padlen - 2: its validation fails before any launch
        o_bad = o_bad.copy()
        c_bad = np.concatenate([c_bad, np.full(200, ord("A"), np.uint8)])
        o_bad[-1] += 200
    raised = False
    try:
        sharding.store_shard_into_root(tok4, c_bad, o_bad, b0_bad, B4, P4, "f", "tbc", dev, 0, None, True)
    except Exception:
        raised = True
    ok = ok and raised            # on EVERY rank: the bad one with its own error, the others with "another rank failed"
    dist.barrier()
    flag = torch.tensor([1 if ok else 0], device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if rank == 0:
        import json
        print(json.dumps(report), flush=True)
        print("MULTI_GPU_OK" if int(flag.item()) == 1 else "MULTI_GPU_MISMATCH", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
