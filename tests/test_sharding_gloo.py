"""CPU, world_size 2 over gloo: the multi-process path -- shard-by-sequence bounds, packed-batch
slicing, and whole-batch assembly of seq-first / batch-first shards (the per-rank encode is done by the
oracle here, since the product itself needs a GPU; the GPU box runs the same code with nccl = RCCL)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_bounds_cover_and_balance():
    from bioseq_amd.sharding import shard_bounds
    for B in (0, 1, 7, 8, 9, 65536, 1000003):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(B, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == B
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)


def test_shard_packed_rebases_offsets():
    from bioseq_amd import synth
    from bioseq_amd.sharding import shard_packed
    chars, offs = synth.synth_packed(3, 11, 0, 9, "ACGT")
    seqs = synth.unpack(chars, offs)
    got = []
    for r in range(3):
        c, o = shard_packed(chars, offs, 3, r)
        assert o[0] == 0 and o[-1] == c.size
        got += synth.unpack(c, o)
    assert got == seqs


def _worker(rank, world, port, B, P, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    from bioseq_amd import sharding, synth
    from oracle import oracle as O
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        chars, offs = synth.synth_packed(99, B, 0, P - 2, synth.DIRTY)
        tok = O.OracleTokenizer("AMINO20", 1, 1, 1)
        full_oh = tok.onehot_packed(chars, offs, P, "f")
        full_bf = tok.tokenize_packed(chars, offs, P, "B", True)
        full_sf = tok.tokenize_packed(chars, offs, P, "i", False)
        oh = sharding.encode_sharded(lambda c, o: tok.onehot_packed(c, o, P, "f"), chars, offs, gather="onehot")
        bf = sharding.encode_sharded(lambda c, o: tok.tokenize_packed(c, o, P, "B", True), chars, offs, gather="tokens_bf")
        sf = sharding.encode_sharded(lambda c, o: tok.tokenize_packed(c, o, P, "i", False), chars, offs, gather="tokens_sf")
        keep = sharding.encode_sharded(lambda c, o: tok.onehot_packed(c, o, P, "f"), chars, offs)
        # point-to-point assembly (gather_direct): the SAME function the GPU box runs over RCCL, here over gloo, fed by
        # the oracle encoder -- whole batch on every rank, and to a root only, for all three layouts
        d_oh = sharding.encode_sharded(lambda c, o: tok.onehot_packed(c, o, P, "f"), chars, offs, gather="direct_onehot")
        d_bf = sharding.encode_sharded(lambda c, o: tok.tokenize_packed(c, o, P, "B", True), chars, offs, gather="direct_tokens_bf")
        d_sf = sharding.encode_sharded(lambda c, o: tok.tokenize_packed(c, o, P, "i", False), chars, offs, gather="direct_tokens_sf")
        direct_ok = (d_oh.numpy().tobytes() == full_oh.tobytes() and d_bf.numpy().tobytes() == full_bf.tobytes()
                     and d_sf.numpy().tobytes() == full_sf.tobytes())
        for root in range(world):
            r_oh = sharding.gather_direct(torch.from_numpy(keep), 1, B, root, rows_per_call=5)     # one message per (peer, row)
            r_oh1 = sharding.gather_direct(torch.from_numpy(keep), 1, B, root)                     # one message per peer (default)
            direct_ok = direct_ok and ((r_oh1 is None) if rank != root else r_oh1.numpy().tobytes() == full_oh.tobytes())
            # the staging cap: row groups of a few KB here (the 256-MB default never splits a test-sized shard) -- several messages per
            # peer, every rank cutting the same groups, ragged and empty shards included
            for cap in (1, 4096, 20000):
                r_g = sharding.gather_direct(torch.from_numpy(keep), 1, B, root, stage_bytes=cap)
                direct_ok = direct_ok and ((r_g is None) if rank != root else r_g.numpy().tobytes() == full_oh.tobytes())
            r_bf = sharding.gather_direct(torch.from_numpy(np.ascontiguousarray(full_bf[slice(*sharding.shard_bounds(B, world, rank))])), 0, B, root)
            if rank == root:
                direct_ok = direct_ok and r_oh.numpy().tobytes() == full_oh.tobytes() and r_bf.numpy().tobytes() == full_bf.tobytes()
            else:
                direct_ok = direct_ok and r_oh is None and r_bf is None

        # ranks that pass DIFFERENT stage_bytes would cut different row groups and hang: the first call with a value checks it with
        # one all-reduce and raises on every rank
        try:
            sharding.gather_direct(torch.from_numpy(keep), 1, B, None, stage_bytes=777 + rank)
            direct_ok = direct_ok and world == 1
        except ValueError as ex:
            direct_ok = direct_ok and "stage_bytes differs" in str(ex)

        # token-gather assembly (onehot_gathered) with CPU stand-ins for the two device passes
        def raw_tokens(c, o):
            oh_ = tok.onehot_packed(c, o, P, "b")
            ids = oh_.argmax(axis=2).astype(np.uint8)
            ids[oh_.sum(axis=2) == 0] = 255
            return torch.from_numpy(ids)

        def expand(tokens):
            t_ = tokens.numpy()
            return torch.from_numpy((t_[:, :, None] == np.arange(full_oh.shape[2], dtype=np.uint8)[None, None, :]).astype(np.float32))

        via_tokens = sharding.onehot_gathered(raw_tokens, expand, chars, offs)
        b0, b1 = sharding.shard_bounds(B, world, rank)
        ok = (oh.numpy().tobytes() == full_oh.tobytes() and bf.numpy().tobytes() == full_bf.tobytes()
              and sf.numpy().tobytes() == full_sf.tobytes()
              and via_tokens.numpy().tobytes() == full_oh.tobytes()
              and keep.tobytes() == np.ascontiguousarray(full_oh[:, b0:b1]).tobytes() and direct_ok)
        # bench.py-style timing reduction: MAX over ranks
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ok = ok and t.item() == world
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,B", [(2, 10), (2, 11), (3, 2), (3, 13)])
def test_gather_matches_single_process(world, B):
    """world 2 with equal / ragged shards; world 3 with an EMPTY shard (B = 2) and ragged ones."""
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, B, 23, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res == [(r, True) for r in range(world)]
