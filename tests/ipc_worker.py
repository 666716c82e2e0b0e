"""Worker of tests/test_sharding_ipc.py: one rank of a 2-rank gloo job whose ranks SHARE the box's GPU.  Every rank encodes
its shard straight into the root's buffer (sharding.encode_into_root: IPC-mapped memory, SURVEY.md section 8e option 3);
the root compares the assembled batch with the CPU oracle and prints the verdict."""
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
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    ok = True
    for B, hi in ((1001, 250), (64, 130), (3, 200)):
        chars, offs = synth.synth_packed(77 + B, B, 0, hi, synth.DIRTY)
        P = hi + 6 - (hi + 6) % 16 + 16
        tok, ora = bioseq_amd.Tokenizer("AMINO20", 1, 1, 1), O.OracleTokenizer("AMINO20", 1, 1, 1)
        for root in range(world):
            got = sharding.encode_into_root(tok, chars, offs, P, "b", "tokens_bf", dev, root=root)
            if rank == root:
                ok = ok and got.cpu().numpy().tobytes() == ora.tokenize_packed(chars, offs, P, "b", True).tobytes()
            else:
                ok = ok and got is None
            got = sharding.encode_into_root(tok, chars, offs, P, "b", "tokens_sf", dev, root=root)  # (P, B): column blocks
            if rank == root:
                ok = ok and got.cpu().numpy().tobytes() == ora.tokenize_packed(chars, offs, P, "b", False).tobytes()
            else:
                ok = ok and got is None
            got = sharding.encode_into_root(tok, chars, offs, P, "f", "bcl", dev, root=root)
            if rank == root:
                exp = np.ascontiguousarray(ora.onehot_packed(chars, offs, P, "f").transpose(1, 2, 0))
                ok = ok and got.cpu().numpy().tobytes() == exp.tobytes()
            del got
            got = sharding.encode_into_root(tok, chars, offs, P, "f", "tbc", dev, root=root)     # seq-first (P, B, C): column blocks
            if rank == root:
                ok = ok and got.cpu().numpy().tobytes() == ora.onehot_packed(chars, offs, P, "f").tobytes()
            del got
            dist.barrier()
    # ONE rank holds a broken shard (offsets that run past its characters): EVERY rank raises -- the failing one its own error, the
    # others "another rank failed" -- and nobody is left waiting in a barrier
    B = 64
    chars, offs = synth.synth_packed(5, B, 0, 100, synth.DIRTY)
    tok = bioseq_amd.Tokenizer("AMINO20", 1, 1, 1)
    b0, _ = sharding.shard_bounds(B, world, rank)
    c, o = sharding.shard_packed(chars, offs, world, rank)
    if rank == world - 1:
        o = np.array(o, copy=True)
        o[-1] += 10 ** 6
    try:
        sharding.store_shard_into_root(tok, c, o, b0, B, 112, "b", "tokens_bf", dev, 0)
        ok = False
    except RuntimeError as ex:
        ok = ok and (("another rank failed" in str(ex)) == (rank != world - 1))
    dist.barrier()
    flag = torch.tensor([1 if ok else 0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if rank == 0:
        print("IPC_ROOT_BUFFER_OK" if int(flag.item()) == 1 else "IPC_ROOT_BUFFER_MISMATCH", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
