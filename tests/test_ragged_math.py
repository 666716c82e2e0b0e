"""CPU: the slot arithmetic of the ragged block form (EParams::ragged / chunk_coord in bioseq_amd/csrc/bsq_onehot.hip), restated in Python and
held to its invariants for random blocks: the pieces of a position row partition the row's bytes exactly, every piece lies inside ONE
4-KiB chunk of memory, at most the first and the last piece of a row are partial, and the memory chunk a slot writes has the slot's class
(chunk index mod 8 == slot index mod 8 -- the XCD pinning of the flat stream).  The kernels themselves are checked on the GPU
(tests/test_ragged_blocks.py); this pins the formula they implement."""
import numpy as np

CH = 4096


def slots_of_row(out_addr, t, pitch, gap):
    """[(k, lo, hi)] for the live slots of row t: bytes [lo, hi) relative to the row's first byte (the same steps as chunk_coord)."""
    npr8 = ((pitch + 2 * CH - 2) // CH + 7) // 8 * 8
    a_t = out_addr + t * (pitch + gap)
    h = a_t & (CH - 1)
    d = (8 - ((a_t >> 12) & 7)) & 7
    live = []
    for sl in range(npr8):
        j = (sl & ~7) + ((sl + d) & 7)
        lo, hi = max(j * CH - h, 0), min(j * CH - h + CH, pitch)
        if hi > lo:
            live.append((t * npr8 + sl, lo, hi, a_t))
    return npr8, live


def test_ragged_slots_partition_every_row_and_keep_their_class():
    rng = np.random.default_rng(7)
    for _ in range(300):
        rb = int(rng.choice([3, 7, 16, 20, 28, 80, 92]))
        B = int(rng.integers(1, 5000))
        row_seqs = B + int(rng.integers(1, 9000))
        pitch, gap = B * rb, (row_seqs - B) * rb
        out = int(rng.integers(1 << 20, 1 << 40)) * int(rng.choice([1, 2, 4, 8]))
        for t in (0, 1, int(rng.integers(2, 500))):
            npr8, live = slots_of_row(out, t, pitch, gap)
            assert npr8 % 8 == 0 and npr8 * CH >= pitch + CH - 1
            covered = np.zeros(pitch, np.int32)
            partial = 0
            for k, lo, hi, a_t in live:
                covered[lo:hi] += 1
                first_chunk, last_chunk = (a_t + lo) >> 12, (a_t + hi - 1) >> 12
                assert first_chunk == last_chunk, "a piece straddles a chunk boundary of memory"
                assert first_chunk % 8 == k % 8, "the slot's class is not its memory chunk's"
                partial += (hi - lo) != CH
                if hi - lo == CH:
                    assert (a_t + lo) % CH == 0
            assert (covered == 1).all(), "the pieces do not partition the row"
            assert partial <= 2
