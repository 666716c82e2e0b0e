"""Multi-GPU use of the hot path: shard a ragged batch BY SEQUENCE, one process per GPU.

Sequences are independent (the reference's OpenMP loop partitions exactly like this on threads,
/root/reference/src/tokenize.h:340-342), and every sequence costs the same output bytes
(P*C'*sizeof(T)), so equal-count contiguous ranges balance the dominant (write) cost.  A
data-parallel consumer needs NO collective: each rank keeps its shard.  Only when one rank needs the
whole batch do the shards travel -- `gather_tokens` / `gather_onehot` (torch.distributed; backend
"nccl" is RCCL over xGMI on ROCm, "gloo" for CPU tests).

Whole-batch assembly:
  * batch-first tokens (B_g, P): shards are contiguous slabs of the result -> one all_gather.
  * seq-first tokens (P, B_g) and one-hot (P, B_g, C): a shard is a COLUMN block of every position row;
    shards are all-gathered into per-rank staging slabs and concatenated along the batch axis
    (one extra read+write of the tensor on the receiver; SURVEY.md section 8e option 2).
Ragged shard counts (B % world != 0) are padded to the largest shard for the collective and trimmed.

`gather_direct` is the point-to-point form (SURVEY.md section 8e option 1): no ring.
xGMI is a full mesh of point-to-point links (7 links x ~153 GB/s per GPU), so every rank posts ONE grouped
batch of isend / irecv with ONE message per peer -- peer k of the group is rank +- k, i.e. every link of the mesh
carries exactly one shard in each direction at the same time: batch-first slabs land straight in the destination
tensor, seq-first / one-hot column blocks in a per-peer staging buffer that one strided device copy lays into the
result (an HBM pass, a few percent of the link time).  `root=r` is the gather `north_star` names (only rank r ends
up with the whole batch, the others only send); `root=None` leaves the whole batch on every rank like all_gather does.

`store_shard_into_root` / `encode_into_root` are SURVEY.md section 8e option 3: the root's buffer is mapped into every rank
through an IPC handle (`open_root_buffer`) and the encode kernels store straight into it over xGMI -- the only form with no
data-path collective and no second pass over the bytes.  Batch-first tokens and the channels-first one-hot are contiguous
slabs of the result; the seq-first (P, B, C) one-hot is written as a column block by `bsq_onehot_block_device`.

`onehot_gathered` is the xGMI-friendly form of the whole-batch one-hot: the shards that travel are the raw
uint8 TOKEN matrices (P, B_g) -- 1/(C*sizeof(T)) of the one-hot's bytes, 1/80 at cfg3 -- and every rank expands
the assembled (P, B) token matrix into the (P, B, C) tensor locally at HBM speed (the second pass of the
library's own two-pass kernel, `bsq_onehot_from_raw_tokens_device`).  Gathering the 5.4 GB f32 one-hot of cfg3
from 8 ranks is link-bound (>= 4.4 ms at 153 GB/s per xGMI link); its tokens are 67 MB.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import numpy as np


def shard_bounds(B: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous range [b0, b1) of rank `rank`; sizes differ by at most one, larger shards first."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad world/rank")
    base, extra = divmod(B, world)
    b0 = rank * base + min(rank, extra)
    return b0, b0 + base + (1 if rank < extra else 0)


def shard_packed(chars, offsets, world: int, rank: int):
    """Slice a packed batch (numpy or torch) to this rank's sequences; offsets are rebased to 0."""
    B = int(offsets.shape[0]) - 1
    b0, b1 = shard_bounds(B, world, rank)
    c0, c1 = int(offsets[b0]), int(offsets[b1])
    return chars[c0:c1], offsets[b0:b1 + 1] - offsets[b0]


def _dist():
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized():
        raise RuntimeError("torch.distributed is not initialised")
    return dist


def _all_gather_padded(local, batch_axis: int, B: int, group=None):
    """all_gather of shards that differ by at most one along `batch_axis`; returns the list of trimmed shards."""
    import torch
    dist = _dist()
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    sizes = [shard_bounds(B, world, r)[1] - shard_bounds(B, world, r)[0] for r in range(world)]
    if local.shape[batch_axis] != sizes[rank]:
        raise ValueError("local shard has %d sequences, expected %d" % (local.shape[batch_axis], sizes[rank]))
    mx = max(sizes)
    send = local
    if sizes[rank] != mx:
        pad_shape = list(local.shape)
        pad_shape[batch_axis] = mx - sizes[rank]
        send = torch.cat([local, local.new_zeros(pad_shape)], dim=batch_axis)
    send = send.contiguous()
    bufs = [torch.empty_like(send) for _ in range(world)]
    dist.all_gather(bufs, send, group=group)
    return [b.narrow(batch_axis, 0, sizes[r]) for r, b in enumerate(bufs)]


def _gather(local, batch_axis: int, B: int, group=None):
    import torch
    dist = _dist()
    world = dist.get_world_size(group)
    if B % world == 0 and local.is_contiguous():
        # equal shards: one all_gather straight into the result (batch-first slabs) or into a (G, ...) staging
        # tensor that is permuted into place (seq-first column blocks)
        try:
            if batch_axis == 0:
                full = local.new_empty((B,) + tuple(local.shape[1:]))
                dist.all_gather_into_tensor(full, local, group=group)
                return full
            stage = local.new_empty((world,) + tuple(local.shape))
            dist.all_gather_into_tensor(stage, local, group=group)
            return torch.cat(list(stage.unbind(0)), dim=batch_axis)
        except (RuntimeError, NotImplementedError):  # backend without the flat all-gather: generic path below
            pass
    return torch.cat(_all_gather_padded(local, batch_axis, B, group), dim=batch_axis)


_checked_same = set()


def _check_same_on_every_rank(value: int, what: str, group, like):
    """One MIN / MAX all-reduce the first time a (group, value) is seen in this process: raises ValueError on EVERY rank when the ranks
    disagree (a disagreement on something that decides how many exchanges a rank posts would otherwise hang the job)."""
    import torch
    dist = _dist()
    key = (id(group) if group is not None else 0, what, int(value))
    if key in _checked_same:
        return
    dev = like.device if str(dist.get_backend(group)) == "nccl" else "cpu"
    v = torch.tensor([int(value), -int(value)], dtype=torch.int64, device=dev)
    dist.all_reduce(v, op=dist.ReduceOp.MIN, group=group)
    lo, hi = int(v[0].item()), -int(v[1].item())
    if lo != hi:
        raise ValueError("%s differs between the ranks of the group (min %d, max %d): it must be identical everywhere" % (what, lo, hi))
    _checked_same.add(key)


def gather_direct(local, batch_axis: int, B: int, root: Optional[int] = None, group=None, rows_per_call: int = 0, stage_bytes: int = 0,
                  check_stage_bytes: bool = True):
    """Whole batch from per-rank shards by grouped point-to-point transfers: ONE message per peer.

    local: this rank's shard, contiguous; batch_axis 0 = (B_g, ...) slabs, 1 = (P, B_g, ...) column blocks.
    root: None -> every rank returns the whole batch; r -> only (group) rank r does, the others return None.
    Batch-first slabs are received straight into the result.  Column blocks (seq-first tokens, the (P,B,C) one-hot)
    arrive as one contiguous message per peer in a staging buffer and are laid into the result by one strided device
    copy each -- an HBM pass (5.4 GB at cfg3: ~2 ms) against >= 35 ms of xGMI time for the same bytes.  (Until round 3
    every position row of every peer was its own message: P x (world - 1) sends and as many receives per rank, 7168 +
    7168 at cfg3 on 8 ranks -- per-operation overhead, not link bandwidth, would have bounded it.  rows_per_call > 0
    keeps that form, in calls of that many rows, for measurement.)
    stage_bytes: cap of one peer's staging buffer for column blocks (default 256 MB); a shard above it travels in ROW GROUPS, one grouped
    exchange per group.  It MUST BE THE SAME ON EVERY RANK of the group: the ranks cut their groups from it, and ranks that cut different
    groups post different numbers of exchanges -- the job would hang.  The first call of a process with a given (group, value) checks that
    with one MIN / MAX all-reduce of two integers and raises ValueError on every rank when they differ (`check_stage_bytes=False` skips it).
    """
    dist = _dist()
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if batch_axis == 1 and rows_per_call <= 0 and check_stage_bytes:
        _check_same_on_every_rank(int(stage_bytes) if stage_bytes and stage_bytes > 0 else (256 << 20), "stage_bytes", group, local)
    if batch_axis not in (0, 1) or (batch_axis == 1 and local.dim() < 2):
        raise ValueError("batch_axis must be 0, or 1 for (P, B_g, ...) shards")
    if root is not None and not 0 <= root < world:
        raise ValueError("bad root")
    bounds = [shard_bounds(B, world, r) for r in range(world)]
    b0, b1 = bounds[rank]
    if local.shape[batch_axis] != b1 - b0:
        raise ValueError("local shard has %d sequences, expected %d" % (local.shape[batch_axis], b1 - b0))
    local = local.contiguous()
    peer = (lambda r: dist.get_global_rank(group, r)) if group is not None else (lambda r: r)
    receives = root is None or rank == root
    full = None
    if receives:
        shape = list(local.shape)
        shape[batch_axis] = B
        full = local.new_empty(shape)
        full.narrow(batch_axis, b0, b1 - b0).copy_(local)
    # k-th exchange: send to rank + k, receive from rank - k -- all world - 1 exchanges are in flight together,
    # each on its own link of the mesh
    sends = [(rank + k) % world for k in range(1, world)] if root is None else ([root] if rank != root else [])
    recvs = [(rank - k) % world for k in range(1, world)] if receives else []
    recvs = [r for r in recvs if bounds[r][1] > bounds[r][0]]
    reqs = []
    if b1 == b0:
        sends = []
    if batch_axis == 0:
        ops = [dist.P2POp(dist.irecv, full.narrow(0, bounds[r][0], bounds[r][1] - bounds[r][0]), peer(r), group) for r in recvs]
        ops += [dist.P2POp(dist.isend, local, peer(r), group) for r in sends]
        if ops:
            reqs += dist.batch_isend_irecv(ops)
        for q in reqs:
            q.wait()
    elif rows_per_call <= 0:
        # Column blocks: one contiguous message per peer and ROW GROUP into a staging buffer, then one strided copy per peer into the
        # result.  A row group holds at most `stage_bytes` (default 256 MB) of the largest shard, so the staging beside the result is
        # (world - 1) x 256 MB however large the shards are -- at cfg3 on 8 ranks 7 x 671 MB were staged per receiving rank until
        # round 4; one group when the shards are smaller than that (one message per peer, as before).
        # Every rank cuts the SAME groups (they depend on P, the largest shard and `stage_bytes` -- checked above to be the same on every
        # rank), so sends and receives pair up group by group.
        P = int(local.shape[0])
        inner = 1
        for d in local.shape[2:]:
            inner *= int(d)
        widest = max(b[1] - b[0] for b in bounds)
        row_bytes = max(1, widest * inner * local.element_size())
        cap = int(stage_bytes) if stage_bytes and stage_bytes > 0 else (256 << 20)
        rows_per_group = max(1, min(P, cap // row_bytes))
        for t0 in range(0, P, rows_per_group):
            t1 = min(P, t0 + rows_per_group)
            stage = {}
            for r in recvs:
                shape = list(local.shape)
                shape[0], shape[1] = t1 - t0, bounds[r][1] - bounds[r][0]
                stage[r] = local.new_empty(shape)
            ops = [dist.P2POp(dist.irecv, stage[r], peer(r), group) for r in recvs]
            ops += [dist.P2POp(dist.isend, local[t0:t1], peer(r), group) for r in sends]  # (rows t0 .. t1 of a contiguous shard: contiguous)
            reqs = dist.batch_isend_irecv(ops) if ops else []
            for q in reqs:
                q.wait()
            for r in recvs:  # column block of the group's position rows: one strided copy per peer
                full[t0:t1].narrow(1, bounds[r][0], bounds[r][1] - bounds[r][0]).copy_(stage[r])
            del stage
    else:
        P = int(local.shape[0])
        for t0 in range(0, P, rows_per_call):
            ops = []
            for t in range(t0, min(P, t0 + rows_per_call)):
                ops += [dist.P2POp(dist.irecv, full[t].narrow(0, bounds[r][0], bounds[r][1] - bounds[r][0]), peer(r), group)
                        for r in recvs]
                ops += [dist.P2POp(dist.isend, local[t], peer(r), group) for r in sends]
            if ops:
                reqs += dist.batch_isend_irecv(ops)
        for q in reqs:
            q.wait()
    return full


def gather_tokens(local, B: int, batch_first: bool, group=None):
    """Whole-batch token matrix on every rank from per-rank shards ((B_g,P) or (P,B_g));
    also used for channels-first one-hot (B_g, C, P) with batch_first=True."""
    return _gather(local, 0 if batch_first else 1, B, group)


def gather_onehot(local, B: int, group=None):
    """Whole-batch (P, B, C) one-hot on every rank from per-rank (P, B_g, C) shards."""
    return _gather(local, 1, B, group)


def onehot_gathered(raw_tokens: Callable, expand: Callable, chars, offsets, group=None):
    """Whole-batch (P, B, C) one-hot on every rank, moving only token matrices between ranks.

    raw_tokens(chars_shard, offsets_shard) -> (P, B_g) uint8 torch tensor of raw ids (255 = all-zero row);
    expand(tokens (P, B)) -> the (P, B, C) one-hot.  `device_passes` builds both from a Tokenizer.
    """
    dist = _dist()
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    B = int(offsets.shape[0]) - 1
    c, o = shard_packed(chars, offsets, world, rank)
    return expand(gather_tokens(raw_tokens(c, o), B, False, group).contiguous())


def device_passes(tokenizer, padlen: int, destchar: str, device):
    """(raw_tokens, expand) for `onehot_gathered`, running the library's two passes on `device` through the C ABI."""
    import ctypes

    import torch

    from . import capi
    lib = capi.load()
    desc = capi.make_desc(tokenizer.key, tokenizer.includes_eos(), tokenizer.includes_bos(), tokenizer.is_padded())
    C = int(tokenizer.alphabet_size())
    dt = ctypes.c_int(0)
    capi.check(lib.bsq_dtype_from_destchar(destchar.encode(), ctypes.byref(dt)))
    tdt = {0: torch.int8, 1: torch.int16, 2: torch.int32, 3: torch.int64, 4: torch.float32, 5: torch.float64}[dt.value]
    dev = torch.device(device)

    def raw_tokens(chars, offsets):
        ch = torch.as_tensor(chars).to(dev)
        of = torch.as_tensor(offsets).to(dev).to(torch.int64).contiguous()
        Bg = int(of.shape[0]) - 1
        out = torch.empty((padlen, Bg), dtype=torch.uint8, device=dev)
        if Bg == 0:
            return out
        with capi.on_device(dev):
            capi.check(lib.bsq_raw_tokens_device(ctypes.byref(desc), ch.data_ptr(), of.data_ptr(), None, Bg, padlen,
                                                 out.data_ptr(), Bg, capi.raw_stream()))
        return out

    def expand(tokens):
        P, B = int(tokens.shape[0]), int(tokens.shape[1])
        out = torch.empty((P, B, C), dtype=tdt, device=tokens.device)
        with capi.on_device(tokens.device):
            capi.check(lib.bsq_onehot_from_raw_tokens_device(tokens.data_ptr(), B, B, P, C, dt, out.data_ptr(),
                                                             capi.raw_stream()))
        return out

    return raw_tokens, expand


def require_dmabuf_ipc():
    """Sharing device memory between processes (`open_root_buffer`, RCCL's own intra-node transports) needs the dmabuf IPC path on this
    driver: HSA_ENABLE_IPC_MODE_LEGACY=0 in the environment BEFORE the process makes its first HIP call -- the runtime reads it once; with the
    legacy mode hipIpcGetMemHandle fails with 'invalid argument' in the middle of a collective.  If no HIP call has been made yet the
    variable is set here; otherwise a process that started without it gets a clear error instead of that failure."""
    import os
    import torch
    if os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY") == "0":
        return
    if not torch.cuda.is_initialized():
        os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
        return
    raise RuntimeError("HSA_ENABLE_IPC_MODE_LEGACY=0 must be in the environment before the first HIP call of the process (it is %r): device "
                       "memory cannot be shared between processes otherwise (hipIpcGetMemHandle: invalid argument).  Export it in the launcher "
                       "(bench.py and the tests do), or call bioseq_amd.sharding.require_dmabuf_ipc() before anything touches the GPU."
                       % os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"))


def open_root_buffer(shape, dtype, device, root: int = 0, group=None):
    """SURVEY.md section 8e option 3, step 1 (collective): rank `root` allocates the whole-batch tensor in ITS HBM, every
    other rank maps the same memory into its own address space through an IPC handle (torch.multiprocessing's
    hipIpcGetMemHandle / hipIpcOpenMemHandle path; needs HSA_ENABLE_IPC_MODE_LEGACY=0 on this driver).  A kernel on a peer
    GPU that stores through the mapped tensor writes over xGMI straight into the root's memory: no collective, no
    staging, no copy.  Returns the tensor (the real one on `root`, the mapped view elsewhere); keep it alive on `root`
    until every rank is done with it."""
    import torch
    from torch.multiprocessing.reductions import reduce_tensor
    dist = _dist()
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if not 0 <= root < world:
        raise ValueError("bad root")
    require_dmabuf_ipc()
    src = dist.get_global_rank(group, root) if group is not None else root
    payload = [None]
    full = None
    if rank == root:
        full = torch.empty(tuple(shape), dtype=dtype, device=device)
        payload = [reduce_tensor(full)]
    dist.broadcast_object_list(payload, src=src, group=group)
    if rank != root:
        rebuild, args = payload[0]
        full = rebuild(*args)
    return full


def store_shard_into_root(tokenizer, shard_chars, shard_offsets, b0: int, B: int, padlen: int, destchar: str, layout: str,
                          device, root: int = 0, group=None, validate: bool = True):
    """SURVEY.md section 8e option 3: this rank encodes ITS shard (sequences [b0, b0 + B_g) of a B-sequence batch, packed,
    on `device`) with the ordinary kernels, but the output pointer is its slab of the ROOT's buffer (peer-mapped,
    `open_root_buffer`) -- the stores of the encode kernels are the gather.  No data-path collective: one small object
    broadcast (the handle) before, one barrier after.  Collective call (every rank of the group).

    layout: 'tokens_bf' -> (B, padlen) tokens and 'bcl' -> (B, C, padlen) one-hot: a rank's shard is one CONTIGUOUS slab of
    the result, written by the streaming kernels; 'tokens_sf' -> the (padlen, B) token matrix (batch_tokenize's default layout):
    a column block through `bsq_tokenize_block_device`; 'tbc' -> the seq-first (padlen, B, C) one-hot: the shard is a column block
    of every position row and goes through `bsq_onehot_block_device` (the two-pass stream with a gap after every position
    row, every row of the shard cut at the 4-KiB boundaries of the root's memory -- any first sequence, any pitch --; the tiled
    kernel with the root tensor's row pitch for shards below 128 MB; the xGMI links, not HBM, bound a remote store).  Returns the whole-batch tensor on `root`,
    None elsewhere."""
    import ctypes

    import torch

    from . import capi
    dist = _dist()
    rank = dist.get_rank(group)
    if layout not in ("tokens_bf", "tokens_sf", "bcl", "tbc"):
        raise ValueError("layout must be 'tokens_bf', 'tokens_sf', 'bcl' or 'tbc'")
    lib = capi.load()
    desc = capi.make_desc(tokenizer.key, tokenizer.includes_eos(), tokenizer.includes_bos(), tokenizer.is_padded())
    C = int(tokenizer.alphabet_size())
    dt = ctypes.c_int(0)
    capi.check(lib.bsq_dtype_from_destchar(destchar.encode(), ctypes.byref(dt)))
    tdt = {0: torch.int8, 1: torch.int16, 2: torch.int32, 3: torch.int64, 4: torch.float32, 5: torch.float64}[dt.value]
    dev = torch.device(device)
    shape = {"tokens_bf": (int(B), padlen), "tokens_sf": (padlen, int(B)), "bcl": (int(B), C, padlen), "tbc": (padlen, int(B), C)}[layout]
    # Everything a rank can fail at on its own (mapping the root's memory, an invalid shard, an encode error) is caught and
    # agreed on by ALL ranks below: one bad shard must raise everywhere, not leave the others waiting in a barrier.
    failure = None
    full = None
    try:
        full = open_root_buffer(shape, tdt, dev, root, group)
    except Exception as ex:  # (the broadcast inside has completed or failed on every rank alike; the mapping is per rank)
        failure = ex
    try:
        if failure is None:
            _store_shard(lib, desc, dt, full, shard_chars, shard_offsets, b0, B, padlen, layout, dev, validate)
        torch.cuda.synchronize(dev)   # this rank's stores have left its GPU ...
    except Exception as ex:
        failure = failure or ex
    # ... and every rank's have -- or some rank failed: MIN over an ok flag (also the barrier that orders the stores)
    ok = torch.tensor([0 if failure is not None else 1], dtype=torch.int32,
                      device=dev if str(dist.get_backend(group)) == "nccl" else "cpu")
    dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
    if int(ok.item()) == 0:
        del full
        if failure is not None:
            raise failure
        raise RuntimeError("store_shard_into_root: another rank failed to encode or store its shard; the root's buffer is incomplete")
    if rank != root:
        del full
        return None
    return full


def _store_shard(lib, desc, dt, full, shard_chars, shard_offsets, b0, B, padlen, layout, dev, validate):
    """This rank's part of `store_shard_into_root`: the ordinary encode kernels with the root's (mapped) slab as their output."""
    import ctypes

    import torch

    from . import capi
    ch = torch.as_tensor(shard_chars).to(dev)
    of = torch.as_tensor(shard_offsets).to(dev).to(torch.int64).contiguous()
    nb = int(of.shape[0]) - 1
    if nb > 0:
        # 'tbc': this rank's sequences are a COLUMN BLOCK of every position row of the (P, B, C) tensor
        slab = full[:, b0:b0 + nb] if layout in ("tbc", "tokens_sf") else full[b0:b0 + nb]
        assert layout in ("tbc", "tokens_sf") or (slab.is_contiguous() and slab.shape[0] == nb)
        with capi.on_device(dev):
            stream = ctypes.c_void_p(capi.raw_stream())
            if validate:
                bad = ctypes.c_int64(-1)
                capi.check(lib.bsq_validate_packed_device(of.data_ptr(), nb, padlen, desc.bos, desc.eos, ch.numel(), ctypes.byref(bad), stream))
            if layout == "tokens_bf":
                capi.check(lib.bsq_tokenize_device(ctypes.byref(desc), ch.data_ptr(), of.data_ptr(), nb, padlen, 1, dt,
                                                   slab.data_ptr(), stream))
            elif layout == "tokens_sf":  # batch_tokenize's default layout: a column block of the root's (P, B) matrix
                capi.check(lib.bsq_tokenize_block_device(ctypes.byref(desc), ch.data_ptr(), of.data_ptr(), nb, padlen, dt,
                                                         slab.data_ptr(), int(B), stream))
            elif layout == "bcl":
                capi.check(lib.bsq_onehot_bcl_device(ctypes.byref(desc), ch.data_ptr(), of.data_ptr(), None, nb, padlen, dt,
                                                     slab.data_ptr(), stream))
            else:  # a column block of the root tensor (bsq_onehot_block_device)
                capi.check(lib.bsq_onehot_block_device(ctypes.byref(desc), ch.data_ptr(), of.data_ptr(), None, nb, padlen, dt,
                                                       slab.data_ptr(), int(B), stream))


def encode_into_root(tokenizer, chars, offsets, padlen: int, destchar: str, layout: str, device, root: int = 0, group=None):
    """`store_shard_into_root` for a WHOLE packed batch (host arrays or tensors) that every rank holds: sharded here with
    `shard_bounds`."""
    dist = _dist()
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    B = int(offsets.shape[0]) - 1
    b0, _ = shard_bounds(B, world, rank)
    c, o = shard_packed(chars, offsets, world, rank)
    return store_shard_into_root(tokenizer, c, o, b0, B, padlen, destchar, layout, device, root, group)


def encode_sharded(encode: Callable, chars, offsets, gather: Optional[str] = None, group=None):
    """Run `encode(chars_shard, offsets_shard)` on this rank's sequences.

    encode: e.g. ``lambda c, o: tok.onehot_packed(c, o, padlen, 'f', device=dev)``.
    gather: None (data-parallel consumer: keep the shard), 'onehot', 'tokens_bf' or 'tokens_sf' (all_gather forms),
            'direct_onehot', 'direct_tokens_bf', 'direct_tokens_sf' (point-to-point form, `gather_direct`).
    """
    dist = _dist()
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    B = int(offsets.shape[0]) - 1
    c, o = shard_packed(chars, offsets, world, rank)
    local = encode(c, o)
    if gather is None:
        return local
    import torch
    if isinstance(local, np.ndarray):
        local = torch.from_numpy(local)
    if gather == "onehot":
        return gather_onehot(local, B, group)
    if gather == "tokens_bf":
        return gather_tokens(local, B, True, group)
    if gather == "tokens_sf":
        return gather_tokens(local, B, False, group)
    if gather in ("direct_onehot", "direct_tokens_sf"):
        return gather_direct(local, 1, B, None, group)
    if gather == "direct_tokens_bf":
        return gather_direct(local, 0, B, None, group)
    raise ValueError("gather must be None, 'onehot', 'tokens_bf', 'tokens_sf' or one of their 'direct_' forms")


# ---------------------------------------------------------------------------------------------------------------------------------
# ONE process, N devices: "host packs once; GPU g receives its slice" (SURVEY.md section 8e).  The reference's only multi-GPU consumer is
# single-process: nn.DataParallel over the batch `load_next` made on the host (/root/reference/training/cnnpretrain.py:85-94), whose
# OpenMP encode partitions the batch by sequence (src/tokenize.h:339-342).  Everything above is the one-process-per-GPU form of the
# same partition; this is the form a DataParallel-style caller uses.

def _pinned_alloc(nbytes):
    """A writable uint8 numpy view of a PINNED torch tensor of `nbytes` (what `cbioseq._pack_list_into` packs into)."""
    import torch
    t = torch.empty(int(nbytes), dtype=torch.uint8, pin_memory=True)
    a = t.numpy()
    _pinned_alloc.keep = t  # (the numpy view holds a reference to the tensor's storage; this only documents who owns the pages)
    return a


def pack_once(tokenizer, batch, padlen: int, nthreads: int = 0, onehot: bool = False):
    """ONE GIL-held scan + ONE pack of a Python list of str / bytes / bytearray / 8-bit arrays into PINNED host memory:
    (chars uint8[total (+16 spare)], offsets int64[B + 1]) as torch CPU tensors (pinned: their slices go up as asynchronous copies).
    A sequence longer than padlen - bos - eos raises the reference's error (tokenize.h:359-362 / :456-459; `onehot` picks its type) before anything is packed.
    Also accepts an already packed (chars, offsets) pair of numpy arrays / CPU tensors (pinned as it is, or after one copy)."""
    import torch
    from . import cbioseq
    room = int(padlen) - int(tokenizer.includes_bos()) - int(tokenizer.includes_eos())
    if isinstance(batch, tuple) and len(batch) == 2 and not isinstance(batch[0], (str, bytes, bytearray)):
        chars = torch.as_tensor(batch[0]).reshape(-1).view(torch.uint8) if not isinstance(batch[0], torch.Tensor) else batch[0]
        offsets = torch.as_tensor(batch[1]).to(torch.int64).contiguous()
        if chars.is_cuda or offsets.is_cuda:
            raise ValueError("pack_once: a packed batch must live in host memory (device batches are sharded with shard_packed)")
        lens = offsets[1:] - offsets[:-1]
        if lens.numel() and int(lens.max()) > room:
            first = int((lens > room).nonzero()[0])
            _raise_too_long(tokenizer, int(lens[first]), padlen, onehot)
        if not chars.is_pinned():
            chars = chars.contiguous().pin_memory()
        return chars, offsets
    offs, buf, bad = cbioseq._pack_list_into(batch, max(room, 0), int(nthreads), _pinned_alloc)
    if bad >= 0:
        _raise_too_long(tokenizer, int(offs[bad + 1] - offs[bad]), padlen, onehot)
    return torch.from_numpy(buf), torch.from_numpy(offs)


def _raise_too_long(tokenizer, length, padlen, onehot):
    """The reference's error for an over-long sequence, type and text (RuntimeError from batch_tokenize, tokenize.h:456-459;
    ValueError from batch_onehot_encode, :359-362) -- as `Tokenizer.batch_tokenize` / `batch_onehot_encode` raise it here."""
    tl = int(length) + int(tokenizer.includes_bos()) + int(tokenizer.includes_eos())
    msg = "seq len + bos + eos > padlen: %d, vs padlen %d" % (tl, int(padlen))
    raise (ValueError if onehot else RuntimeError)(msg)


class _DeviceSlot:
    """What one entry of `devices` owns (a device may appear more than once: every entry gets its own set): a copy stream, an encode stream,
    two events that are re-recorded every call, and a device input buffer (rebased offsets | characters) that is kept between calls."""

    def __init__(self, device):
        import torch
        self.device = torch.device(device)
        with torch.cuda.device(self.device):
            self.copy = torch.cuda.Stream()
            self.encode = torch.cuda.Stream()
            self.uploaded = torch.cuda.Event()
            self.encoded = torch.cuda.Event()
        self.buf = None
        self.used = False
        self.keep = None  # the caller's pinned arrays of the last call (an asynchronous copy may still be reading them)

    def input_buffer(self, nbytes):
        import torch
        if self.buf is None or self.buf.numel() < nbytes:
            if self.buf is not None:
                # the buffer being dropped may still be read by the previous call's copies / kernels, and it was allocated on whatever stream
                # was current THEN: tell the allocator, or it could hand the block to that stream's next allocation at once
                self.buf.record_stream(self.copy)
                self.buf.record_stream(self.encode)
            with torch.cuda.device(self.device):
                self.buf = torch.empty(max(int(nbytes * 1.25) + 4096, 1 << 16), dtype=torch.uint8, device=self.device)
            self.used = False  # (a fresh allocation: nothing in flight reads it)
        return self.buf


_slots = {}


class _PinnedRing:
    """Pinned host buffers of `encode_on_devices`, kept between calls (a fresh 35-MB hipHostMalloc per call costs ~0.4 ms): two take turns, and a
    buffer is refilled only after the uploads that last read it have completed (one event per device entry, re-recorded every use)."""

    def __init__(self):
        self.turn = 0
        self.slots = [None, None]

    def get(self, nbytes):
        import torch
        self.turn ^= 1
        slot = self.slots[self.turn]
        if slot is None or slot["t"].numel() < nbytes:
            slot = self.slots[self.turn] = {"t": torch.empty(max(int(nbytes * 1.25) + 4096, 1 << 16), dtype=torch.uint8, pin_memory=True), "events": {}, "live": []}
        for key in slot["live"]:
            slot["events"][key].synchronize()
        slot["live"] = []
        return slot

    @staticmethod
    def event(slot, key):
        import torch
        ev = slot["events"].get(key)
        if ev is None:
            ev = slot["events"][key] = torch.cuda.Event()
        slot["live"].append(key)
        return ev


_pinned = _PinnedRing()
_encode_lock = None  # serialises the ISSUE phase of encode_on_devices between Python threads (stream pairs and pinned buffers are shared)


def encode_on_devices(tokenizer, batch, padlen: int, destchar: str = "B", devices=None, op: str = "onehot", batch_first: bool = False,
                      layout: str = "tbc", root=None, nthreads: int = 0):
    """Sharded encode from ONE process: the host scans `batch` once under the GIL (`cbioseq._ListScan`: pointer + length of every item, the
    offsets) and packs it once into pinned memory, slice by slice; device g of `devices` gets sequences `shard_bounds(B, len(devices), g)` --
    its slice of the characters and its rebased offsets uploaded on ITS copy stream as soon as that slice is packed (while the next device's
    is being packed), encoded on ITS encode stream by the ordinary kernels -- with all devices in flight together (N PCIe links, N GPUs);
    no collective, nothing synchronises.

    op 'tokenize' (batch_first as in `batch_tokenize`) or 'onehot' (layout 'tbc' = the reference's (padlen, B, C), or 'bcl' = (B, C, padlen)).
    devices: list of devices; an entry may repeat (`['cuda:0', 'cuda:0']`: two stream pairs of one GPU -- how this pool tests it).
    Returns the list of per-device shards (what `nn.parallel.parallel_apply` consumes), each safe to use on its device's current stream.
    root = a device: the whole-batch tensor is allocated THERE and every device stores its shard straight into it (column blocks through
    `bsq_onehot_block_device` / `bsq_tokenize_block_device`, row slabs for batch-first layouts) over peer access; returns that one tensor."""
    import threading
    global _encode_lock
    if op not in ("tokenize", "onehot"):
        raise ValueError("op must be 'tokenize' or 'onehot'")
    if op == "onehot" and layout not in ("tbc", "bcl"):
        raise ValueError("layout must be 'tbc' or 'bcl'")
    if _encode_lock is None:
        _encode_lock = threading.Lock()
    with _encode_lock:  # (the call only ENQUEUES work: held for about a millisecond)
        return _encode_on_devices(tokenizer, batch, padlen, destchar, devices, op, batch_first, layout, root, nthreads)


def _encode_on_devices(tokenizer, batch, padlen, destchar, devices, op, batch_first, layout, root, nthreads):
    import ctypes

    import torch

    from . import capi
    if op not in ("tokenize", "onehot"):
        raise ValueError("op must be 'tokenize' or 'onehot'")
    if op == "onehot" and layout not in ("tbc", "bcl"):
        raise ValueError("layout must be 'tbc' or 'bcl'")
    devs = [torch.device(d) for d in (devices if devices is not None else ["cuda:%d" % i for i in range(torch.cuda.device_count())])]
    if not devs or any(d.type != "cuda" for d in devs):
        raise ValueError("encode_on_devices needs a non-empty list of HIP devices")
    devs = [torch.device("cuda", torch.cuda.current_device() if d.index is None else d.index) for d in devs]
    lib = capi.load()
    desc = capi.make_desc(tokenizer.key, tokenizer.includes_eos(), tokenizer.includes_bos(), tokenizer.is_padded())
    C = int(tokenizer.alphabet_size())
    dt = ctypes.c_int(0)
    capi.check(lib.bsq_dtype_from_destchar(destchar.encode(), ctypes.byref(dt)))
    tdt = {0: torch.int8, 1: torch.int16, 2: torch.int32, 3: torch.int64, 4: torch.float32, 5: torch.float64}[dt.value]
    seq_first = (op == "onehot" and layout == "tbc") or (op == "tokenize" and not batch_first)
    G = len(devs)
    scan = user_chars = None
    if isinstance(batch, tuple) and len(batch) == 2 and not isinstance(batch[0], (str, bytes, bytearray)):
        user_chars, offsets = pack_once(tokenizer, batch, padlen, nthreads, onehot=(op == "onehot"))  # already packed: pinned as it is / after one copy
    else:
        # ONE scan under the GIL (pointer + length of every item, the offsets); the bytes are packed into pinned memory SLICE BY SLICE below,
        # device g's slice on its way over PCIe while device g + 1's is being packed
        from . import cbioseq
        room = int(padlen) - int(tokenizer.includes_bos()) - int(tokenizer.includes_eos())
        scan = cbioseq._ListScan(batch, max(room, 0), int(nthreads))
        offsets = torch.from_numpy(scan.offsets)
        if scan.bad >= 0:
            _raise_too_long(tokenizer, int(offsets[scan.bad + 1] - offsets[scan.bad]), padlen, op == "onehot")
    B = int(offsets.shape[0]) - 1
    o_np = offsets.numpy()
    total = int(o_np[-1])
    # ONE pinned buffer per call (two take turns between calls): every device's REBASED offsets (device g: entries [b0 + g, b1 + g + 1)), then the characters
    off_bytes = ((B + G) * 8 + 63) // 64 * 64
    ring = _pinned.get(off_bytes + (total + 16 if scan is not None else 0))
    pin_np = ring["t"].numpy()
    r_np = pin_np[:(B + G) * 8].view(np.int64)
    reb = ring["t"][:(B + G) * 8].view(torch.int64)
    if scan is not None:
        chars = ring["t"][off_bytes:off_bytes + total + 16]
        chars_np = pin_np[off_bytes:off_bytes + total + 16]
    else:
        chars = user_chars
    bounds = [shard_bounds(B, G, g) for g in range(G)]
    for g, (b0, b1) in enumerate(bounds):
        np.subtract(o_np[b0:b1 + 1], o_np[b0], out=r_np[b0 + g:b1 + g + 1])

    def shape_of(n):
        if op == "tokenize":
            return (n, padlen) if batch_first else (padlen, n)
        return (padlen, n, C) if layout == "tbc" else (n, C, padlen)

    full = None
    root_dev = None
    root_ready = None
    if root is not None:
        root_dev = torch.device(root)
        root_dev = torch.device("cuda", torch.cuda.current_device() if root_dev.index is None else root_dev.index)
        with capi.on_device(root_dev):
            full = torch.empty(shape_of(B), dtype=tdt, device=root_dev)
            root_ready = torch.cuda.Event()
            root_ready.record(torch.cuda.current_stream())  # (the allocator may hand out memory that this stream's queued work still uses)
        for d in devs:
            capi.check(lib.bsq_enable_peer_access(d.index, root_dev.index))
    set_stream = torch.cuda.set_stream  # (cheaper than entering a `torch.cuda.stream` context twice per device)
    outs, slots = [], []
    for g, dev in enumerate(devs):
        b0, b1 = bounds[g]
        nb = b1 - b0
        c0, c1 = int(o_np[b0]), int(o_np[b1])
        slot = _slots.get((g, dev.index))
        if slot is None:
            slot = _slots[(g, dev.index)] = _DeviceSlot(dev)
        slots.append(slot)
        if scan is not None and nb > 0:
            scan.pack(b0, b1, chars_np)  # this device's characters (the pool's threads copy; the GIL stays held: the items must not change)
        nchar = min(c1 + 16, chars.numel()) - c0  # (+16 spare bytes ride along when the buffer has them: the kernels' unaligned 16-byte loads stay inside)
        dof_bytes = ((nb + 1) * 8 + 255) // 256 * 256
        buf = slot.input_buffer(dof_bytes + nchar + 16)
        d_offs = buf[:(nb + 1) * 8].view(torch.int64)
        d_chars = buf[dof_bytes:dof_bytes + nchar]
        with capi.on_device(dev):
            back = torch.cuda.current_stream()
            try:
                set_stream(slot.copy)
                if slot.used:
                    slot.copy.wait_event(slot.encoded)  # the previous call's kernels have read this slot's device buffer
                d_chars.copy_(chars[c0:c0 + nchar], non_blocking=True)
                d_offs.copy_(reb[b0 + g:b1 + g + 1], non_blocking=True)
                slot.uploaded.record(slot.copy)
                _PinnedRing.event(ring, g).record(slot.copy)  # the pinned buffer may be refilled once these copies have completed
                set_stream(slot.encode)
                slot.encode.wait_event(slot.uploaded)
                if full is None:
                    out = torch.empty(shape_of(nb), dtype=tdt, device=dev)
                    dst, pitch = out, nb
                else:
                    slot.encode.wait_event(root_ready)
                    out = None
                    dst, pitch = (full[:, b0:b1] if seq_first else full[b0:b1]), B
                if nb > 0:
                    stream = ctypes.c_void_p(slot.encode.cuda_stream)
                    a = (ctypes.byref(desc), d_chars.data_ptr(), d_offs.data_ptr())
                    if op == "tokenize" and batch_first:
                        st = lib.bsq_tokenize_device(*a, nb, padlen, 1, dt, dst.data_ptr(), stream)
                    elif op == "tokenize":
                        st = lib.bsq_tokenize_block_device(*a, nb, padlen, dt, dst.data_ptr(), pitch, stream)
                    elif layout == "bcl":
                        st = lib.bsq_onehot_bcl_device(*a, None, nb, padlen, dt, dst.data_ptr(), stream)
                    else:
                        st = lib.bsq_onehot_block_device(*a, None, nb, padlen, dt, dst.data_ptr(), pitch, stream)
                    capi.check(st)
                slot.encoded.record(slot.encode)
                slot.used = True
                slot.keep = user_chars
                # hand-over: this device's CURRENT stream waits for its encode stream (an event; the host never blocks)
                back.wait_event(slot.encoded)
                if out is not None:
                    out.record_stream(back)
            finally:
                set_stream(back)
        outs.append(out)
    if full is not None:
        with capi.on_device(root_dev):
            cur = torch.cuda.current_stream()
            for g in range(G):  # the root's stream waits for every device's stores
                cur.wait_event(slots[g].encoded)
        return full
    return outs
