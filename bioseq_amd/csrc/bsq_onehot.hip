// The ONE-HOT kernels of libbsq_hip.so (batch_onehot_encode; /root/reference/src/tokenize.h:283-371), written for gfx950 (MI355X / CDNA4) only.
// (Until round 5 this file, as bsq_kernels.hip, also held the token kernels -- now bsq_tokens.hip -- and the element kernels -- bsq_generic.hip;
//  what both units share is bsq_tiles.h.)
//
// The path is byte-LUT + streaming stores: HBM-bound, no MFMA.  Traffic per call is
//     sum(L) chars + 8(B+1) offsets  read once,   P*B*C'*sizeof(T) output  written once
// and the one-hot output is ~150x the input, so everything is organised around the WRITE side:
// every output element is produced exactly once by a 16-byte store, there is no memset pass and no
// scattered store to global memory (the reference's structure, /root/reference/src/tokenize.h:332 +
// :342-369, is memset + one 4-byte scattered store per residue at stride B*C*sizeof(T)).
//
// What the write side needs on this part (measured): naturally aligned 4-KiB
// chunks, each XCD writing its own residue class (chunk id % 8 == blockIdx % 8 under round-robin
// dispatch), one chunk per wave, non-temporal stores, about 12 waves resident per CU (occupancy capped with
// unused LDS) and therefore short instruction streams (reciprocal multiplies instead of integer divisions).
// Speed only -- results never depend on it.
//
// Kernel inventory
//   k_tokens_raw +     two-pass (P,B,C) one-hot for any pitch: raw uint8 tokens through tiled, coalesced
//   k_expand_chunks    character reads into a scratch matrix, then the chunk-wise flat expansion of it.
//                      Large outputs; cfg3: 92 % of HBM peak (the expansion alone writes at 7.5-7.8 TB/s).
//   k_expand_small     the expansion for rows of 16..63 bytes (hundreds of rows per chunk): four rows per lane from
//                      one unaligned dword of tokens, all of a chunk's token loads in flight together.
//   k_onehot_chunks    (P,B,C) one-hot, chunk-owner form, one launch: a wave gathers the characters of the
//                      ~4096/rowbytes rows of its chunk (consecutive sequences at one position), LUT from a
//                      wave-private LDS table, scatters the ones into a 4-KiB LDS image, streams it out.
//                      Outputs below 4 GiB with rows >= 48 B; cfg3 (forced): 89 %.
//   k_onehot_tile      (P,B,C) one-hot, tiled: workgroup = TB sequences x 64 positions, per-wave LDS row
//                      images; small rows / small outputs.
//   k_tokenize_chunks  (B,P) tokens and the channels-first (B,C,P) one-hot: flat chunk stream, a lane owns 16
//                      output bytes of one sequence row, unaligned vector loads of its characters.
//                      (The (B,P) int8 matrix has its own kernel: k_tokens_bp8, bsq_tokens8.hip.)
//   k_tokenize_rows    (B,P) tokens for odd padlen / unaligned bases.
//   k_tokens_raw<value>, k_tokenize_tile   (P,B) tokens (int8 / wider types): tiled transpose through LDS.
//   (k_*_generic -- one thread per output element, any shape / alignment / alphabet -- and the device-side length validation
//    k_first_too_long live in bsq_generic.hip since round 5.)
// (The write-bandwidth yardsticks, store-pattern diagnostics and probes of include/bsq_diag.h live in bsq_diag.hip.)
#include "bsq_tiles.h"

namespace {

// ------------------------------------------------------------------------------------------
// One-hot, tiled.  Dynamic LDS layout:
//   [0, off_bytes)              int64 offsets of the tile's sequences (+1)
//   [.., +256)                  alphabet LUT
//   [.., +TB*kTokStride)        token tile
//   [.., +4*row_pad)            one row image per wave (row_pad = TB*C*sizeof(ST) rounded to 16)
// ------------------------------------------------------------------------------------------
template <typename ST, int TB, bool NT>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(TB == 256 ? 5 : 4))) void k_onehot_tile(const KParams p) {
    extern __shared__ __align__(16) uint8_t smem[];
    SeqSpan *s_span = reinterpret_cast<SeqSpan *>(smem);
    uint8_t *s_lut = smem + tile_off_bytes<TB>();
    uint8_t *s_tok = s_lut + 256;
    uint8_t *s_rows = s_tok + TB * kTokStride;

    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    int32_t tb, tt;
    tile_of_block(p, tb, tt);
    if (tb >= p.ntb) return;  // (order 2 rounds the sequence tiles up to a multiple of 8)
    const int64_t b0 = static_cast<int64_t>(tb) * TB;
    const int32_t t0 = tt * kTT;

    const int32_t C = p.C;
    const int32_t row_pad = (TB * C * static_cast<int32_t>(sizeof(ST)) + 15) & ~15;
    uint8_t *row = s_rows + wave * row_pad;
    // zero this wave's row image while the tile's characters are in flight
    for (int32_t o = lane * 16; o < row_pad; o += 64 * 16) *reinterpret_cast<uint4 *>(row + o) = uint4{0, 0, 0, 0};

    build_token_tile<TB>(p, b0, t0, s_lut, s_span, s_tok);

    const int64_t nb64 = p.B - b0;
    const int32_t nb = nb64 < TB ? static_cast<int32_t>(nb64) : TB;
    const int32_t seg = nb * C * static_cast<int32_t>(sizeof(ST));  // bytes of one output row segment
    const ST one = static_cast<ST>(p.one_bits);
    const int64_t row_pitch = p.row_seqs * C * static_cast<int64_t>(sizeof(ST));
    uint8_t *gtile = static_cast<uint8_t *>(p.out) + b0 * C * static_cast<int64_t>(sizeof(ST));

    constexpr int kRowsPerWave = kTT / 4;
    constexpr int kSeqPerLane = TB / 64;
    for (int r = 0; r < kRowsPerWave; ++r) {
        const int32_t tl = wave * kRowsPerWave + r;
        const int64_t t = static_cast<int64_t>(t0) + tl;
        if (t >= p.P) break;  // wave-uniform
        // 1. scatter the ones of this row into the wave's LDS image.  All token reads first: the compiler cannot prove
        // that the image does not alias the token tile, so a read placed after a write waits for it (four serial LDS
        // round trips per row before round 2).  (Unpredicated writes with a spare slot for the lanes without a token
        // were tried: one shared slot serialises those lanes, a slot per lane cost cfg4 int8 2 % -- ab_tile_scatter*.txt.)
        int32_t hot[kSeqPerLane];
        uint32_t tks[kSeqPerLane];
#pragma unroll
        for (int q = 0; q < kSeqPerLane; ++q) tks[q] = s_tok[(lane + 64 * q) * kTokStride + tl];
#pragma unroll
        for (int q = 0; q < kSeqPerLane; ++q) {
            const int32_t sb = lane + 64 * q;
            hot[q] = (tks[q] != kNone) ? (sb * C + static_cast<int32_t>(tks[q])) * static_cast<int32_t>(sizeof(ST)) : -1;
        }
#pragma unroll
        for (int q = 0; q < kSeqPerLane; ++q)
            if (hot[q] >= 0) *reinterpret_cast<ST *>(row + hot[q]) = one;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // 2. stream the image to global memory
        uint8_t *grow = gtile + t * row_pitch;
        if (p.aligned) {
            for (int32_t o = lane * 16; o < seg; o += 4 * 1024) {  // up to 4 x 1 KiB per wave per step
                const bool c1 = o + 1024 < seg, c2 = o + 2048 < seg, c3 = o + 3072 < seg;
                const uint4 z{0, 0, 0, 0};
                const uint4 v0 = *reinterpret_cast<const uint4 *>(row + o);
                const uint4 v1 = c1 ? *reinterpret_cast<const uint4 *>(row + o + 1024) : z;
                const uint4 v2 = c2 ? *reinterpret_cast<const uint4 *>(row + o + 2048) : z;
                const uint4 v3 = c3 ? *reinterpret_cast<const uint4 *>(row + o + 3072) : z;
                store16<NT>(grow + o, v0);
                if (c1) store16<NT>(grow + o + 1024, v1);
                if (c2) store16<NT>(grow + o + 2048, v2);
                if (c3) store16<NT>(grow + o + 3072, v3);
            }
        } else {
            for (int32_t o = lane * static_cast<int32_t>(sizeof(ST)); o < seg; o += 64 * static_cast<int32_t>(sizeof(ST)))
                *reinterpret_cast<ST *>(grow + o) = *reinterpret_cast<const ST *>(row + o);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // 3. clear the ones again
#pragma unroll
        for (int q = 0; q < kSeqPerLane; ++q)
            if (hot[q] >= 0) *reinterpret_cast<ST *>(row + hot[q]) = ST(0);
    }
}

// ------------------------------------------------------------------------------------------
// One-hot, two-pass form.  The (P,B,C) one-hot tensor is the flat expansion of the flat (P,B) token
// matrix: out[r*C + c] = (tok[r] == c), r = t*B + b.  k_expand_chunks streams that expansion in units
// of naturally aligned 4-KiB CHUNKS of the output, and every workgroup only writes chunks of ONE
// residue class mod 8: blocks are dealt round-robin over the 8 XCDs, so block b (class b % 8) keeps
// "XCD x writes the chunks with (chunk id % 8) == x" -- measured on MI355X: 7.1 TB/s for that
// assignment vs 5.8 TB/s when the classes are mixed across XCDs (profiles/r01/sweep_pattern.txt, sweep_perm.txt).
// The mapping only affects speed, never results.
//
// One wave = one chunk at a time: load the ~4096/(C*sizeof(T)) tokens whose rows intersect the chunk
// (coalesced bytes), scatter their ones into the wave's private 4-KiB LDS image, stream the image
// out with 4 x (ds_read_b128 -> global_store_dwordx4), clear the ones.  No barriers.
// ------------------------------------------------------------------------------------------
struct EParams {
    const uint8_t *tok;  // raw tokens (kNone = no one), row t at tok + t*Bp (Bp = B rounded up to 256: every
                         // row of the scratch is 256-byte aligned whatever B is)
    int64_t B, Bp;
    uint8_t *out;        // output base (any alignment that is a multiple of sizeof(ST))
    int64_t total;       // output bytes
    int64_t nchunks;     // chunks intersecting [out, out + total)
    int32_t head;        // out & 4095
    int32_t C;
    uint64_t one_bits;
    double inv_rowbytes, inv_B;  // reciprocals for div_by()
    uint32_t rb_magic, rb_shift, rb_pow2;  // fast_div() constants of rowbytes
    int32_t force4;                        // experiment knob "expand_slots" = 4: always four token slots per step
    int64_t pitch;                         // B * rowbytes: bytes of one position row of the output
    Div64 dv_pitch, dv_rb;                 // div64() constants of pitch and rowbytes (scalar chunk arithmetic)
    int32_t nib;                           // 1: `tok` holds NIBBLES (k_expand_chunks<.., NIB>; never with the other expansion kernels)
    int64_t row_gap;                       // column block of a wider tensor: bytes between the end of one position row of the block and
                                           // the start of the next (0 = the whole tensor).  Without `ragged`: only when pitch % 4096 == 0 and
                                           // head == 0 -- no chunk then straddles two position rows
    // ragged (end of round 5): a column block whose position rows do not start on chunk boundaries of MEMORY (any first sequence, any
    // tensor pitch).  Every row gets npr8 chunk slots (a multiple of 8 >= the pieces a row can be cut into); slot s of row t is the
    // memory chunk (A_t >> 12) + j(s), A_t the row's address, clipped to the row -- its first and last pieces are partial, all others
    // whole naturally aligned chunks exactly as in the flat stream; j(s) rotates the slots of every group of eight so that
    // (memory chunk index) % 8 == (slot index) % 8 == the class of the workgroup that owns it: the XCD pinning of the flat stream holds.
    int32_t ragged;
    int64_t npr8;
    double inv_npr8;
};

// Where chunk k of the flat output lies: byte range [lo, lo + len) relative to `out`, first row r_lo = t_lo * B +
// b_lo intersecting it, `skip` bytes of that row before the chunk, `nr` rows intersecting it.  k is WAVE-UNIFORM.
// MATH 1: 64-bit integer reciprocal multiplies -- scalar-ALU work on uniform operands; MATH 0: the double
// reciprocals of round 1 (div_by; always vector-ALU work at the FP64 rate).
struct ChunkCoord {
    int64_t lo, t_lo, b_lo;
    int32_t len, skip, nr;
    bool live;
};
template <int MATH>
__device__ __forceinline__ ChunkCoord chunk_coord(const EParams &p, int64_t k, int32_t rowbytes) {
    ChunkCoord c;
    if (p.ragged) {
        int64_t sl;
        const int64_t t = div_by(k, p.npr8, p.inv_npr8, &sl);
        const uint64_t a_t = reinterpret_cast<uintptr_t>(p.out) + static_cast<uint64_t>(t) * static_cast<uint64_t>(p.pitch + p.row_gap);
        const int64_t h = static_cast<int64_t>(a_t & (kChunk - 1));
        const int64_t d = (8 - static_cast<int64_t>((a_t >> 12) & 7u)) & 7;
        const int64_t j = (sl & ~int64_t(7)) + ((sl + d) & 7);
        int64_t lo_r = j * kChunk - h, hi_r = lo_r + kChunk;  // byte range relative to the row's first byte
        if (lo_r < 0) lo_r = 0;
        if (hi_r > p.pitch) hi_r = p.pitch;
        c.live = k < p.nchunks && hi_r > lo_r;
        c.lo = t * p.pitch + lo_r;  // (the kernels add t_lo * row_gap)
        c.len = c.live ? static_cast<int32_t>(hi_r - lo_r) : 0;
        c.t_lo = t;
        c.b_lo = 0;
        c.skip = c.nr = 0;
        if (!c.live) return c;
        int64_t skip64;
        c.b_lo = div_by(lo_r, rowbytes, p.inv_rowbytes, &skip64);
        c.skip = static_cast<int32_t>(skip64);
        c.nr = static_cast<int32_t>(fast_div(static_cast<uint32_t>(c.skip + c.len + rowbytes - 1), p.rb_magic, p.rb_shift, p.rb_pow2));
        return c;
    }
    int64_t lo = k * kChunk - p.head, hi = lo + kChunk;  // byte range relative to `out`
    if (lo < 0) lo = 0;
    if (hi > p.total) hi = p.total;
    c.live = k < p.nchunks && hi > lo;
    c.lo = lo;
    c.len = c.live ? static_cast<int32_t>(hi - lo) : 0;
    c.t_lo = c.b_lo = 0;
    c.skip = c.nr = 0;
    if (!c.live) return c;
    if constexpr (MATH == 1) {
        const uint64_t t = div64(static_cast<uint64_t>(lo), p.dv_pitch);      // position
        const uint64_t rem = static_cast<uint64_t>(lo) - t * static_cast<uint64_t>(p.pitch);
        const uint64_t b = div64(rem, p.dv_rb);                               // sequence
        c.t_lo = static_cast<int64_t>(t);
        c.b_lo = static_cast<int64_t>(b);
        c.skip = static_cast<int32_t>(rem - b * static_cast<uint64_t>(rowbytes));
    } else {
        int64_t skip64, b_lo;
        const int64_t r_lo = div_by(lo, rowbytes, p.inv_rowbytes, &skip64);
        c.skip = static_cast<int32_t>(skip64);
        c.t_lo = div_by(r_lo, p.B, p.inv_B, &b_lo);
        c.b_lo = b_lo;
    }
    c.nr = static_cast<int32_t>(fast_div(static_cast<uint32_t>(c.skip + c.len + rowbytes - 1), p.rb_magic, p.rb_shift, p.rb_pow2));
    return c;
}

// (A variant with the four waves of a workgroup sharing one chunk -- the shape of the fastest plain fill --
// measured 1.6x slower: every wave then pays the token-load latency for a single 1-KiB store.)
// (Experiments that lost -- a placement-independent chunk claim, scalar 64-bit chunk arithmetic, dword token loads -- are history: csrc/labs/.)
// GATE (rows of 24 ... 63 bytes; knob "expand_gate"): every wave first issues ONE agent-scope load -- of the head of the token scratch, a
// line that is always at the memory side -- and makes its token loads depend on it.  The load means nothing; what it does is pace the
// waves: the small-row expansion runs at 5 workgroups per CU, all of whose waves otherwise reach their token loads and their 4 KiB of
// stores in step.  Measured over 24 shapes (profiles/r04/expand_gate_sweep.txt): 28-byte rows (DNA f32: cfg4) +1-2 %, 32-byte rows +5.5 %,
// 56-byte rows +4.7 %; rows of 64 bytes and more lose 5-7 % (cfg3 0.724 -> 0.774 ms), rows of 20 bytes and less lose 1-4 %: those
// do not get it.  (Found as a by-product of the one-launch experiment, profiles/r04/onehot_fused_one_launch_lost.txt.)
// NIB (round 5): the id matrix holds NIBBLES -- the id of (t, b) in bits 4 (b & 1) ... of byte (t * Bp + b) / 2 (Bp is even), 15 = no one:
// the scratch of alphabets with at most 15 classes at half its bytes (see two_pass_nibbles).  A template flag: as a runtime one its
// shift / mask arithmetic cost the BYTE form 13 % on cfg4 f32 (670 -> 759 us).
template <typename ST, bool NT, int MATH, bool GATE = false, bool NIB = false>
__global__ __launch_bounds__(kThreads) void k_expand_chunks(const EParams p) {
    constexpr int PIECE = kChunk;             // bytes per wave
    constexpr int NS = PIECE / 1024;          // 16-byte stores per lane
    __shared__ __align__(16) uint8_t s_img[4][PIECE];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint8_t *img = s_img[wave];
#pragma unroll
    for (int u = 0; u < NS; ++u) *reinterpret_cast<uint4 *>(img + u * 1024 + lane * 16) = uint4{0, 0, 0, 0};

    // chunk of this wave: class = blockIdx % 8 (pinned to the XCD the block lands on).  The wave index goes through
    // readfirstlane so that the chunk arithmetic below is scalar-ALU work.
    const int wave_s = MATH == 1 ? __builtin_amdgcn_readfirstlane(wave) : wave;
    int64_t group = static_cast<int64_t>(blockIdx.x >> 3);
    int32_t cls = static_cast<int32_t>(blockIdx.x & 7u);
    const int64_t slot = group * 4 + wave_s;
    const int64_t k = static_cast<int64_t>(cls) + 8 * slot;
    if (k >= p.nchunks) return;
    uint32_t gate = 0;
    if constexpr (GATE) gate = __hip_atomic_load(reinterpret_cast<const uint32_t *>(p.tok) + (blockIdx.x & 63u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int32_t rowbytes = p.C * static_cast<int32_t>(sizeof(ST));
    const ST one = static_cast<ST>(p.one_bits);
    const ChunkCoord cc = chunk_coord<MATH>(p, k, rowbytes);
    if (!cc.live) return;
    const int64_t lo = cc.lo, b_lo = cc.b_lo, t_lo = cc.t_lo;
    const int32_t len = cc.len, skip = cc.skip, nr = cc.nr;
    // NIB: the byte that holds the chunk's first id, and that id's half (Bp is even: the parity of t_lo * Bp + b_lo is b_lo's)
    const uint8_t *tok = NIB ? p.tok + ((t_lo * p.Bp + b_lo) >> 1) : p.tok + t_lo * p.Bp + b_lo;
    const int32_t par0 = NIB ? static_cast<int32_t>(b_lo & 1) : 0;
    if constexpr (GATE) tok += (gate == 0xFEFEFEFDu && p.nchunks < 0) ? 1 : 0;  // never taken: the token loads wait for the gate load
    constexpr uint32_t nmask = NIB ? 0xFu : kNone;  // "no one"
    const int64_t wrap_at = p.B - b_lo;  // rows i >= wrap_at belong to position t_lo + 1 (or later)
    // scatter: row r_lo + i has its one at image byte i*rowbytes - skip + tok*sizeof(ST).
    // NS (1..4) coalesced token loads in flight per step, straight-line per NS: the number of 64-row slots a
    // step needs and the (rare) row-wrap case are wave-uniform, so they are scalar branches.
    const int32_t nr_s = __builtin_amdgcn_readfirstlane(nr);
    const bool wraps = wrap_at < nr_s;  // the piece runs over the end of position t_lo's rows
    auto step = [&](auto ns_tag, int32_t i0) {
        constexpr int NS = decltype(ns_tag)::value;
        uint32_t tk[NS];
#pragma unroll
        for (int q = 0; q < NS; ++q) {
            const int32_t i = i0 + 64 * q + lane;
            int64_t a = i;
            if (wraps && i >= wrap_at) {
                const int64_t w = (i - wrap_at) / p.B + 1;
                a = i + w * (p.Bp - p.B);
            }
            if constexpr (NIB) {
                const int64_t ia = a + par0;
                tk[q] = i < nr_s ? (static_cast<uint32_t>(tok[ia >> 1]) >> ((static_cast<uint32_t>(ia) & 1u) << 2)) & 0xFu : nmask;
            } else {
                tk[q] = i < nr_s ? static_cast<uint32_t>(tok[a]) : kNone;
            }
        }
#pragma unroll
        for (int q = 0; q < NS; ++q) {
            const int32_t i = i0 + 64 * q + lane;
            const int32_t pos = i * rowbytes - skip + static_cast<int32_t>(tk[q]) * static_cast<int32_t>(sizeof(ST));
            if (tk[q] != nmask && pos >= 0 && pos < len) *reinterpret_cast<ST *>(img + pos) = one;
        }
    };
    for (int32_t i0 = 0; i0 < nr_s; i0 += 256) {
        const int32_t left = p.force4 ? 256 : nr_s - i0;
        if (left > 192) step(std::integral_constant<int, 4>{}, i0);
        else if (left > 128) step(std::integral_constant<int, 3>{}, i0);
        else if (left > 64) step(std::integral_constant<int, 2>{}, i0);
        else step(std::integral_constant<int, 1>{}, i0);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    uint8_t *g = p.out + lo + t_lo * p.row_gap;
    if (len == PIECE && (reinterpret_cast<uintptr_t>(g) & 15) == 0) {
        uint4 v[NS];
#pragma unroll
        for (int u = 0; u < NS; ++u) v[u] = *reinterpret_cast<const uint4 *>(img + u * 1024 + lane * 16);
#pragma unroll
        for (int u = 0; u < NS; ++u) store16<NT>(g + u * 1024 + lane * 16, v[u]);
    } else {  // clipped first / last piece of the tensor
        for (int32_t o = lane * static_cast<int32_t>(sizeof(ST)); o < len; o += 64 * static_cast<int32_t>(sizeof(ST)))
            *reinterpret_cast<ST *>(g + o) = *reinterpret_cast<const ST *>(img + o);
    }
}

// k_expand_rows1 (round 5): the same chunk stream for ONE-BYTE elements with rows of 3 ... 15 bytes (int8 one-hots of DNA-sized
// alphabets: BASELINE config 4's default dtype, 7-byte rows) WITHOUT the LDS image.  k_expand_chunks scatters one byte per row into
// its image: 585 rows per chunk at 7 bytes = ten byte loads and ten ds_write_b8 per lane, more than half of its LDS cycles bank
// conflicts (profiles/r04/cfg4b_sq_tcc_counters.txt), 4.4 TB/s.  Here a lane BUILDS its 16 output bytes in registers: they cover
// at most NR = (rb + 14) / rb + 1 consecutive rows, whose ids come as one unaligned 8-byte load from the (P,B) id matrix; row j's
// one-hot is the rb-bit string 1 << id (255 = no one: bit 31, masked off); the rows' strings concatenated, shifted right by the
// lane's phase inside its first row, are the lane's 16 bytes as 16 BITS, and a nibble becomes four 0 / 1 bytes by one 24-bit
// multiply: ((n * 0x204081) & 0x01010101).  ~35 vector instructions per 16 bytes, no LDS, no barrier; one wave = one aligned 4-KiB
// chunk, class = blockIdx % 8, exactly as above.  Chunks that are clipped (first / last of the tensor), misaligned, or that run
// over the end of a position row of a padded id matrix (Bp != B) take the byte loop at the end -- a few hundred chunks of a million.
// NIB (end of round 5): the id matrix holds nibbles as in k_expand_chunks<..., NIB> -- a lane's <= 6 ids and the parity of its first one
// are one unaligned 4-byte window.
template <bool NT, int NR, bool NIB = false>
__global__ __launch_bounds__(kThreads) void k_expand_rows1(const EParams p) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    const int64_t slot = static_cast<int64_t>(blockIdx.x >> 3) * 4 + wave_s;
    const int64_t k = static_cast<int64_t>(blockIdx.x & 7u) + 8 * slot;
    if (k >= p.nchunks) return;
    const int32_t rb = p.C;  // bytes per row (one-byte elements)
    const ChunkCoord cc = chunk_coord<0>(p, k, rb);
    if (!cc.live) return;
    const int32_t len = cc.len, skip = cc.skip;
    const int32_t nr_s = __builtin_amdgcn_readfirstlane(cc.nr);
    // NIB: the byte that holds the chunk's first id, and that id's half (Bp is even: the parity of t_lo * Bp + b_lo is b_lo's)
    const uint8_t *tok = NIB ? p.tok + ((cc.t_lo * p.Bp + cc.b_lo) >> 1) : p.tok + cc.t_lo * p.Bp + cc.b_lo;
    const uint32_t par0 = NIB ? static_cast<uint32_t>(cc.b_lo & 1) : 0u;
    constexpr uint32_t kNo = NIB ? 0xFu : kNone;  // "no one"
    const int64_t wrap_at = p.B - cc.b_lo;  // rows i >= wrap_at belong to position t_lo + 1 (or later)
    const bool wraps = p.Bp != p.B && wrap_at < nr_s;  // (wave-uniform) the chunk runs over the end of a position row of a PADDED id matrix
    uint8_t *g = p.out + cc.lo + cc.t_lo * p.row_gap;
    const uint32_t one = static_cast<uint32_t>(p.one_bits) & 0xFFu;
    // the id of row i of the chunk (rows behind wrap_at lie in later position rows of the matrix, Bp - B ids further each)
    auto id_of = [&](uint32_t i) -> uint32_t {
        int64_t at = i;
        if (static_cast<int64_t>(i) >= wrap_at) at += ((static_cast<int64_t>(i) - wrap_at) / p.B + 1) * (p.Bp - p.B);
        if constexpr (NIB) {
            at += par0;
            return (static_cast<uint32_t>(tok[at >> 1]) >> ((static_cast<uint32_t>(at) & 1u) << 2)) & 0xFu;
        } else {
            return static_cast<uint32_t>(tok[at]);
        }
    };
    constexpr int IDB = NIB ? 4 : 8;  // bits per id in a lane's window
    if (len == kChunk && (reinterpret_cast<uintptr_t>(g) & 15) == 0 && nr_s >= 8) {
        uint64_t ids[4];
        uint32_t ph[4];
        if (!wraps && NIB) {
            typedef uint32_t u32u __attribute__((aligned(1)));
            uint32_t w[4], sh[4];
            const uint32_t lb = (par0 + static_cast<uint32_t>(nr_s) - 1u) >> 1;  // the last byte of ids this chunk owns (>= 3: nr_s >= 8)
#pragma unroll
            for (int u = 0; u < 4; ++u) {  // the four id windows of the lane in flight together: NR <= 6 nibbles + a parity fit four bytes
                const uint32_t o = static_cast<uint32_t>(skip) + static_cast<uint32_t>(u * 1024 + lane * 16);
                const uint32_t q = fast_div(o, p.rb_magic, p.rb_shift, p.rb_pow2);
                ph[u] = o - q * static_cast<uint32_t>(rb);
                const uint32_t n0 = q + par0, bo = n0 >> 1;
                const uint32_t off = bo + 3u <= lb ? bo : lb - 3u;  // pulled back to END at the chunk's last id byte (rows < nr_s lie in bo ... lb)
                sh[u] = (bo - off) * 8u + ((n0 & 1u) << 2);
                w[u] = *reinterpret_cast<const u32u *>(tok + off);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) ids[u] = w[u] >> sh[u];
        } else if (!wraps) {
            typedef uint32_t u32x2u __attribute__((ext_vector_type(2), aligned(1)));
            u32x2u w[4];
            uint32_t sh[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {  // the four id windows of the lane in flight together
                const uint32_t o = static_cast<uint32_t>(skip) + static_cast<uint32_t>(u * 1024 + lane * 16);
                const uint32_t q = fast_div(o, p.rb_magic, p.rb_shift, p.rb_pow2);
                ph[u] = o - q * static_cast<uint32_t>(rb);
                // the last lanes' windows are pulled back to END at the chunk's last row: never a byte beyond the ids this chunk owns
                const uint32_t off = q + 8u <= static_cast<uint32_t>(nr_s) ? q : static_cast<uint32_t>(nr_s) - 8u;
                sh[u] = (q - off) * 8u;
                w[u] = *reinterpret_cast<const u32x2u *>(tok + off);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) ids[u] = ((static_cast<uint64_t>(w[u].y) << 32) | w[u].x) >> sh[u];
        } else {  // one chunk per position row: the NR ids of a window one by one, across the padding (all 4 NR byte loads in flight together)
            uint32_t b[4][NR];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t o = static_cast<uint32_t>(skip) + static_cast<uint32_t>(u * 1024 + lane * 16);
                const uint32_t q = fast_div(o, p.rb_magic, p.rb_shift, p.rb_pow2);
                ph[u] = o - q * static_cast<uint32_t>(rb);
#pragma unroll
                for (int j = 0; j < NR; ++j) b[u][j] = q + j < static_cast<uint32_t>(nr_s) ? id_of(q + j) : kNo;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                ids[u] = 0;
#pragma unroll
                for (int j = 0; j < NR; ++j) ids[u] |= static_cast<uint64_t>(b[u][j]) << (IDB * j);
            }
        }
        const uint32_t cmask = (1u << rb) - 1u;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            uint32_t bits = 0;
#pragma unroll
            for (int j = 0; j < NR; ++j) {
                const uint32_t id = static_cast<uint32_t>(ids[u] >> (IDB * j)) & (NIB ? 0xFu : 0xFFu);
                const uint32_t m = (1u << (id & 31u)) & cmask;  // id 255 (no one) -> bit 31 -> 0; nibble 15 -> bit 15 -> 0 (rows of <= 15 bytes)
                const int32_t at = j * rb;                      // wave-uniform; strings that start at bit >= 32 lie beyond the lane's bits
                if (at < 32) bits |= m << at;
            }
            bits >>= ph[u];
            uint4 o;
            o.x = (((bits >> 0) & 15u) * 0x204081u) & 0x01010101u;
            o.y = (((bits >> 4) & 15u) * 0x204081u) & 0x01010101u;
            o.z = (((bits >> 8) & 15u) * 0x204081u) & 0x01010101u;
            o.w = (((bits >> 12) & 15u) * 0x204081u) & 0x01010101u;
            if (one != 1u) {
                o.x *= one;
                o.y *= one;
                o.z *= one;
                o.w *= one;
            }
            store16<NT>(g + u * 1024 + lane * 16, o);
        }
        return;
    }
    // the clipped first / last chunk of the tensor, a result that is not 16-byte aligned: byte by byte, eight independent ids at a time
    for (int32_t o0 = lane; o0 < len; o0 += 64 * 8) {
        uint32_t idv[8], cv[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const int32_t o = o0 + 64 * m;
            const uint32_t a = static_cast<uint32_t>(o + skip);
            const uint32_t i = fast_div(a, p.rb_magic, p.rb_shift, p.rb_pow2);
            cv[m] = a - i * static_cast<uint32_t>(rb);
            idv[m] = o < len ? id_of(i) : kNo;
        }
#pragma unroll
        for (int m = 0; m < 8; ++m)
            if (o0 + 64 * m < len) g[o0 + 64 * m] = idv[m] == cv[m] ? static_cast<uint8_t>(one) : uint8_t(0);
    }
}


// ------------------------------------------------------------------------------------------
// Channels-first one-hot (B, C, P), two-pass form: raw (B, P) uint8 ids from k_tokens_bp8 (bsq_tokens8.hip), then
// this expansion.  The output is the flat (B*C, P) matrix, row b*C + c = (tok[b, :] == c); one wave = one aligned
// 4-KiB chunk (class pinned to the XCD), a lane owns 16 output bytes = EPL consecutive positions of one row, i.e. ONE
// aligned EPL-byte load of tokens, EPL compares, one nt store.  No LDS, no dependent second load: the kernel is a pure
// write stream (capped at 3 workgroups per CU like the (P,B,C) expansion) that re-reads each token row C times out of
// the caches.  Needs P % EPL == 0 and a 16-byte aligned output.
// ------------------------------------------------------------------------------------------
struct BParams {
    const uint8_t *tok;  // (B, P) raw ids, kNone = no one
    uint8_t *out;
    int64_t total, nchunks, nrows;  // output bytes, chunks, B * C
    int64_t P;
    int32_t C;
    uint64_t one_bits;
    uint32_t magic, shift, pow2;        // fast_div constants of P
    uint32_t magic_c, shift_c, pow2_c;  // ... of C
    double inv_P;                       // div_by constant (outputs of 2^31 elements and more)
};

template <typename T, bool NT>
__global__ __launch_bounds__(kThreads) void k_expand_bcl(const BParams p) {
    constexpr int SZ = static_cast<int>(sizeof(T));
    constexpr int EPL = 16 / SZ;
    const int lane = threadIdx.x & 63;
    const int wave_s = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
    const int64_t k = static_cast<int64_t>(blockIdx.x & 7u) + 8 * (static_cast<int64_t>(blockIdx.x >> 3) * 4 + wave_s);
    if (k >= p.nchunks) return;
    const int64_t lo = k * kChunk;
    const int64_t ec = lo / SZ;  // first element of the chunk (wave-uniform)
    const bool small = p.nrows * p.P < (int64_t(1) << 31);
    const uint32_t Pu = static_cast<uint32_t>(p.P);
    int64_t rc;   // row of the chunk's first element
    uint32_t tc;  // its position
    if (small) {
        const uint32_t q = fast_div(static_cast<uint32_t>(ec), p.magic, p.shift, p.pow2);
        rc = q;
        tc = static_cast<uint32_t>(ec) - q * Pu;
    } else {
        int64_t rem;
        rc = div_by(ec, p.P, p.inv_P, &rem);
        tc = static_cast<uint32_t>(rem);
    }
    const T one = static_cast<T>(p.one_bits);
    // the four stores of the lane: element ec + u * (1024 / SZ) + lane * EPL
    uint32_t tok[4][EPL >= 4 ? EPL / 4 : 1];
    uint32_t chan[4];
    bool live[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const uint32_t tl = tc + static_cast<uint32_t>(u * (1024 / SZ) + lane * EPL);  // < P + 4096
        const uint32_t ql = fast_div(tl, p.magic, p.shift, p.pow2);
        const int64_t r = rc + ql;
        const uint32_t t = tl - ql * Pu;
        live[u] = r < p.nrows;
        const int64_t rr = live[u] ? r : 0;
        int64_t b;
        if (p.nrows < (int64_t(1) << 31))
            b = fast_div(static_cast<uint32_t>(rr), p.magic_c, p.shift_c, p.pow2_c);
        else
            b = rr / p.C;
        chan[u] = static_cast<uint32_t>(rr - b * p.C);
        const uint8_t *src = p.tok + b * p.P + t;  // EPL-byte aligned: P % EPL == 0, t % EPL == 0
        if constexpr (EPL == 16) {
            const uint4 v = *reinterpret_cast<const uint4 *>(src);
            tok[u][0] = v.x, tok[u][1] = v.y, tok[u][2] = v.z, tok[u][3] = v.w;
        } else if constexpr (EPL == 8) {
            const uint2 v = *reinterpret_cast<const uint2 *>(src);
            tok[u][0] = v.x, tok[u][1] = v.y;
        } else if constexpr (EPL == 4) {
            tok[u][0] = *reinterpret_cast<const uint32_t *>(src);
        } else {
            tok[u][0] = *reinterpret_cast<const uint16_t *>(src);
        }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        if (!live[u]) continue;
        alignas(16) T vals[EPL];
#pragma unroll
        for (int i = 0; i < EPL; ++i) {
            const uint32_t tk = (tok[u][i >> 2] >> (8 * (i & 3))) & 0xFFu;
            vals[i] = tk == chan[u] ? one : T(0);
        }
        store16<NT>(p.out + lo + u * 1024 + lane * 16, *reinterpret_cast<const uint4 *>(vals));
    }
}

template <typename ST, int TB>
bsq_status launch_onehot_tile(const KParams &k, hipStream_t s) {
    const int row_pad = (TB * k.C * int(sizeof(ST)) + 15) & ~15;
    const size_t smem = tile_fixed_bytes<TB>() + 4 * size_t(row_pad);
    const int64_t ntt = (k.P + kTT - 1) / kTT;
    const int64_t grid = tile_grid(k, ntt);
    if (bsq_internal::nontemporal_stores())
        hipLaunchKernelGGL((k_onehot_tile<ST, TB, true>), dim3(unsigned(grid)), dim3(kThreads), smem, s, k);
    else
        hipLaunchKernelGGL((k_onehot_tile<ST, TB, false>), dim3(unsigned(grid)), dim3(kThreads), smem, s, k);
    return check_launch("k_onehot_tile");
}

template <typename ST>
bsq_status dispatch_onehot_tile(KParams &k, hipStream_t s) {
    // Row segment = TB*C*sizeof(ST) bytes of contiguous output per (tile,row): keep it >= 2 KiB.
    const int seg64 = 64 * k.C * int(sizeof(ST));
    int tb = seg64 >= 2048 ? 64 : 256;
    const int forced = bsq_internal::tuning().onehot_tb;
    if ((forced == 64 || forced == 128 || forced == 256) &&
        4 * (forced * k.C * int(sizeof(ST)) + 16) + tile_fixed_bytes<256>() <= 64 * 1024)
        tb = forced;
    // automatic tile order: XCD-aware only for the 256-sequence tiles of tiny rows (cfg4 int8: 258 -> 233 us); the
    // 64-sequence tiles of wide rows stream 5-9 % faster with the sequence-tile index fastest (sweep_shapes_r02.txt
    // vs profiles/r01/sweep_shapes5.txt, column p1)
    if (bsq_internal::tuning().tile_order == 0 && tb != 256) k.order = 0;
    k.ntb = int32_t((k.B + tb - 1) / tb);
    if (tb == 64) return launch_onehot_tile<ST, 64>(k, s);
    if (tb == 128) return launch_onehot_tile<ST, 128>(k, s);
    return launch_onehot_tile<ST, 256>(k, s);
}

// ------------------------------------------------------------------------------------------
// One-hot, single pass, CHUNK-OWNER form: one wave = one naturally aligned 4-KiB chunk of the output,
// chunk classes (id % 8) pinned to XCDs exactly as in k_expand_chunks, but the tokens of the chunk's
// rows are resolved on the fly: rows r_lo.. are consecutive sequences b at (mostly) one position t,
// so each lane loads offsets[b], offsets[b+1] (coalesced) and gathers ONE character per sequence.
// The gathered lines are re-used by the next positions from L2 (each XCD keeps to its own chunk
// columns when the row pitch is a multiple of 32 KiB), so HBM sees the characters about once.
// Any shape / pitch / alignment; best when a row (C*sizeof(T) bytes) is >= ~32 bytes.
// (Tried in round 2: the rounds of 64 rows of a small-row chunk batched 2 / 4 at a time -- all offsets in flight, then all
// characters -- to pay the two dependent round trips once: slower everywhere but cfg3 (1000 x 256 DNA f32 6 -> 7 us,
// 8192 x 1024 AMINO20 f32 0.10 -> 0.11 ms, profiles/r02/ab_owner_rounds.txt): the extra registers cost the occupancy
// this kernel lives on.)
// ------------------------------------------------------------------------------------------
struct CParams {
    int8_t lut[256];
    const uint8_t *chars;
    const int64_t *offsets;
    const uint8_t *mask;
    uint8_t *out;
    int64_t total;    // output bytes
    int64_t nchunks;
    int64_t B, P;
    int32_t head;     // out & 4095
    int32_t C;
    int32_t bos;
    uint32_t bos_id, at_len_id, fill_id;
    int32_t room;     // P - bos - eos (length clamp)
    int32_t cpw;
    uint64_t one_bits;
    double inv_rowbytes, inv_B;  // reciprocals for div_by()
    uint32_t rb_magic, rb_shift, rb_pow2;  // fast_div() constants of rowbytes
};

// Fewer resident workgroups stream faster (see launch_chunks): the launch caps the occupancy at 4 per CU.
template <typename ST, bool NT>
__global__ __launch_bounds__(kThreads) void k_onehot_chunks(const CParams p) {
    __shared__ __align__(16) uint8_t s_img[4][kChunk];
    __shared__ __align__(16) uint8_t s_lut4[4][256];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint8_t *img = s_img[wave];
    uint8_t *lut = s_lut4[wave];
#pragma unroll
    for (int u = 0; u < 4; ++u) *reinterpret_cast<uint4 *>(img + u * 1024 + lane * 16) = uint4{0, 0, 0, 0};
    {   // wave-private copy of the alphabet table (unmapped / >= 0x80 -> kNone): no workgroup barrier needed
        uint32_t w = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int idx = lane * 4 + q;
            const int8_t v = p.lut[idx];
            const uint32_t e = (idx < 128 && v >= 0) ? static_cast<uint32_t>(v) : kNone;
            w |= e << (8 * q);
        }
        reinterpret_cast<uint32_t *>(lut)[lane] = w;
    }
    const int32_t cls = static_cast<int32_t>(blockIdx.x & 7u);
    const int64_t group = static_cast<int64_t>(blockIdx.x >> 3);
    const int32_t rowbytes = p.C * static_cast<int32_t>(sizeof(ST));
    const ST one = static_cast<ST>(p.one_bits);
    int64_t j = (group * 4 + wave) * p.cpw;
    for (int32_t it = 0; it < p.cpw; ++it, ++j) {
        const int64_t k = cls + 8 * j;
        if (k >= p.nchunks) break;  // wave-uniform
        int64_t lo = k * kChunk - p.head, hi = lo + kChunk;
        if (lo < 0) lo = 0;
        if (hi > p.total) hi = p.total;
        const int32_t len = static_cast<int32_t>(hi - lo);
        int64_t skip64, b_lo;
        const int64_t r_lo = div_by(lo, rowbytes, p.inv_rowbytes, &skip64);
        const int32_t skip = static_cast<int32_t>(skip64);
        const int32_t nr = static_cast<int32_t>(fast_div(static_cast<uint32_t>(skip + len + rowbytes - 1), p.rb_magic,
                                                         p.rb_shift, p.rb_pow2));
        const int64_t t_lo = div_by(r_lo, p.B, p.inv_B, &b_lo);
        for (int32_t i0 = 0; i0 < nr; i0 += 64) {
            const int32_t i = i0 + lane;
            if (i < nr) {
                int64_t b = b_lo + i, t = t_lo;
                if (b >= p.B) {  // the chunk runs over the end of position t's rows
                    const int64_t q = b / p.B;
                    t += q;
                    b -= q * p.B;
                }
                uint32_t tk;
                const int64_t start = p.offsets[b];
                int64_t L = p.offsets[b + 1] - start;
                L = L > p.room ? p.room : L;
                const int64_t jj = t - p.bos;
                if (jj < 0) {
                    tk = p.bos_id;
                } else if (jj < L) {
                    tk = lut[p.chars[start + jj]];
                    if (p.mask && p.mask[start + jj] == 0) tk = kNone;
                } else {
                    tk = (jj == L) ? p.at_len_id : p.fill_id;
                }
                const int32_t pos = i * rowbytes - skip + static_cast<int32_t>(tk) * static_cast<int32_t>(sizeof(ST));
                if (tk != kNone && pos >= 0 && pos < len) *reinterpret_cast<ST *>(img + pos) = one;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        uint8_t *g = p.out + lo;
        if (len == kChunk) {
            const uint4 v0 = *reinterpret_cast<const uint4 *>(img + lane * 16);
            const uint4 v1 = *reinterpret_cast<const uint4 *>(img + 1024 + lane * 16);
            const uint4 v2 = *reinterpret_cast<const uint4 *>(img + 2048 + lane * 16);
            const uint4 v3 = *reinterpret_cast<const uint4 *>(img + 3072 + lane * 16);
            store16<NT>(g + lane * 16, v0);
            store16<NT>(g + 1024 + lane * 16, v1);
            store16<NT>(g + 2048 + lane * 16, v2);
            store16<NT>(g + 3072 + lane * 16, v3);
        } else {
            for (int32_t o = lane * static_cast<int32_t>(sizeof(ST)); o < len; o += 64 * static_cast<int32_t>(sizeof(ST)))
                *reinterpret_cast<ST *>(g + o) = *reinterpret_cast<const ST *>(img + o);
        }
        if (it + 1 < p.cpw) {  // image is reused: wipe it (wave-private, in-order LDS)
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int u = 0; u < 4; ++u) *reinterpret_cast<uint4 *>(img + u * 1024 + lane * 16) = uint4{0, 0, 0, 0};
        }
    }
}

template <typename ST>
bsq_status launch_chunks(const CParams &c, hipStream_t s) {
    const int64_t per_class = (c.nchunks + 7) / 8;
    const int64_t groups = (per_class + int64_t(4) * c.cpw - 1) / (int64_t(4) * c.cpw);
    const dim3 grid(unsigned(groups * 8));
    // Occupancy cap through unused dynamic LDS: 4 workgroups per CU (17 KiB + 22 KiB each) stream at 7.2 TB/s
    // on cfg3; 5 (the VGPR limit) at 6.9, 3 at 6.6, 2 at 4.8 (profiles/r01/chunks_occupancy.txt).  The same
    // holds for a plain fill: 6.8 TB/s at 8 workgroups per CU, 7.4 at 3.  Knob "chunks_pad" overrides (bytes).
    const int padv = bsq_internal::tuning().chunks_pad;
    const size_t pad = padv > 0 ? size_t(padv) : (padv < 0 ? size_t(0) : size_t(22528));
    if (bsq_internal::nontemporal_stores())
        hipLaunchKernelGGL((k_onehot_chunks<ST, true>), grid, dim3(kThreads), pad, s, c);
    else
        hipLaunchKernelGGL((k_onehot_chunks<ST, false>), grid, dim3(kThreads), pad, s, c);
    return check_launch("k_onehot_chunks");
}

bsq_status onehot_chunk_owner(const KParams &k, size_t sz, hipStream_t s) {
    CParams c;
    for (int i = 0; i < 256; ++i) c.lut[i] = k.lut[i];
    c.chars = k.chars;
    c.offsets = k.offsets;
    c.mask = k.mask;
    c.out = static_cast<uint8_t *>(k.out);
    c.B = k.B;
    c.P = k.P;
    c.total = k.P * k.B * k.C * int64_t(sz);
    c.head = int32_t(reinterpret_cast<uintptr_t>(k.out) & (kChunk - 1));
    c.nchunks = (c.head + c.total + kChunk - 1) / kChunk;
    c.C = k.C;
    c.bos = k.bos;
    c.bos_id = uint32_t(k.bos_id);
    c.fill_id = uint32_t(k.fill_id);
    c.at_len_id = k.eos ? uint32_t(k.eos_id) : c.fill_id;
    const int64_t room = k.P - k.bos - k.eos;
    c.room = int32_t(room < 0 ? 0 : room);
    c.cpw = 1;  // chunks per wave (1 is fastest: 0.75 / 0.86 / 0.93 ms for 1 / 2 / 4 on cfg3; the other forms are labs/ history)
    c.one_bits = k.one_bits;
    c.inv_rowbytes = 1.0 / double(k.C * int64_t(sz));
    c.inv_B = 1.0 / double(k.B);
    div_constants(uint32_t(k.C * int64_t(sz)), &c.rb_magic, &c.rb_shift, &c.rb_pow2);
    if ((c.nchunks + 7) / 8 / (4 * c.cpw) + 1 >= (int64_t(1) << 28)) return bsq_internal::set_error(BSQ_ERR_INVALID_ARG, "output too large");
    switch (sz) {
    case 1: return launch_chunks<uint8_t>(c, s);
    case 2: return launch_chunks<uint16_t>(c, s);
    case 4: return launch_chunks<uint32_t>(c, s);
    default: return launch_chunks<uint64_t>(c, s);
    }
}

template <typename ST>
bsq_status launch_expand(const EParams &e, hipStream_t s) {
    const int64_t per_class = (e.nchunks + 7) / 8;
    const int64_t groups = (per_class + 3) / 4;
    if (groups * 8 >= (int64_t(1) << 31)) return bsq_internal::set_error(BSQ_ERR_INVALID_ARG, "output too large");
    const dim3 grid(unsigned(groups * 8));
    // Occupancy cap through unused dynamic LDS (3 x (16 KiB image + 36 KiB) = 156 KiB <= 160 KiB; 37 KiB already
    // rounds up to 2 per CU).  Rows >= 64 B (one token load per lane and chunk): 3 workgroups per CU stream
    // cfg3 at 7.5-7.8 TB/s, 8 at 6.3, 4 at 6.7, 2 at 5.5.  Smaller rows: 16 KiB + the 16-KiB image -- FOUR workgroups per CU, not the five
    // the arithmetic suggests (round 5, profiles/r05/nibble_ids_lab.txt: the step from four to five lies between pads of 16 384 and 15 360
    // bytes; byte ids 673 us at four, 690 at five, 703 at six on cfg4 f32).
    // In round 1 a cap HURT the 1M x 160 x 28-byte batch (0.90 ms at 5 per CU vs 0.78 uncapped) -- because the token pass
    // then fetched every character three times and pushed its own scratch out of the Infinity Cache; with the XCD-aware
    // tile order the token loads of the expansion are cache hits and the cap pays: 0.687 ms at 5, 0.689 at 4, 0.725 at
    // 3, 0.731 uncapped (profiles/r02/pad_lab2.txt, pad_lab3.txt).  12 waves per CU is
    // the optimum also with 2-wave workgroups (16 / 14 / 12 / 10 waves: 0.83 / 0.81 / 0.76 / 0.93 ms), and 3 x 4 waves
    // (0.73 ms) beats 6 x 2.
    // Knob "expand_pad": 0 = this rule, > 0 = that many bytes, < 0 = none.
    // (k_expand_small -- dword token loads -- lost: once the token scratch is written in XCD-aware tile order the byte-load kernel under
    // an occupancy cap is 1-2 % ahead of it, and two or four chunks per wave were 20-40 % slower: profiles/r02/pad_lab*.txt, expand_lab3.txt;
    // its source is csrc/labs/bsq_expand_small.inc.)
    const int padv = bsq_internal::tuning().expand_pad;
    const bool big_rows = e.C * int64_t(sizeof(ST)) >= 64;
    // (nibble ids, rows below 64 bytes: 12 KiB = FIVE workgroups per CU -- 16 KiB + the 16-KiB image + the kernel's few bytes of static LDS
    //  round up to four; cfg4 f32 with nibbles 714 us at four, 648 at five, 662 at six; byte ids 673 / 690 / 703: profiles/r05/nibble_ids_lab.txt)
    const size_t pad = padv > 0 ? size_t(padv) : (padv < 0 ? size_t(0) : (big_rows ? size_t(36864) : (e.nib ? size_t(12288) : size_t(16384))));
    // (Scalar 64-bit reciprocal multiplies instead of the double reciprocals for the chunk arithmetic: ~90 SALU instructions instead of
    // ~130 VALU ones, ahead where a wave's latency is exposed (2 workgroups per CU: 0.938 vs 0.969 ms on cfg3) and 1 % BEHIND at the
    // bandwidth optimum (3 per CU: 0.741 vs 0.734 ms) -- profiles/r02/math_lab1.txt; not built any more.)
    // one-byte elements, rows of 3 ... 15 bytes: the LDS-free form (knob "expand_rows1": 0 automatic, 1 never, 2 whenever it applies)
    if constexpr (sizeof(ST) == 1) {
        const int rk = bsq_internal::tuning().expand_rows1;
        const int rb = e.C;
        if (rk != 1 && rb >= 3 && rb <= 15) {
            const int nrows = (rb + 14) / rb + 1;  // rows a lane's 16 bytes can touch: 6, 5, 4, 4, 4, 3 ... 3, 2
            const size_t rpad = padv > 0 ? size_t(padv) : (padv < 0 ? size_t(0) : size_t(32768));  // (no static LDS here: 5 workgroups per CU)
            const bool nt = bsq_internal::nontemporal_stores();
#define BSQ_ROWS1(NRV)                                                                                       \
    case NRV:                                                                                                \
        if (e.nib) {                                                                                         \
            if (nt) hipLaunchKernelGGL((k_expand_rows1<true, NRV, true>), grid, dim3(kThreads), rpad, s, e); \
            else hipLaunchKernelGGL((k_expand_rows1<false, NRV, true>), grid, dim3(kThreads), rpad, s, e);   \
        } else {                                                                                             \
            if (nt) hipLaunchKernelGGL((k_expand_rows1<true, NRV>), grid, dim3(kThreads), rpad, s, e);       \
            else hipLaunchKernelGGL((k_expand_rows1<false, NRV>), grid, dim3(kThreads), rpad, s, e);         \
        }                                                                                                    \
        break;
            switch (nrows) {
                BSQ_ROWS1(2) BSQ_ROWS1(3) BSQ_ROWS1(4) BSQ_ROWS1(5) BSQ_ROWS1(6)
            default: return bsq_internal::set_error(BSQ_ERR_INVALID_ARG, "k_expand_rows1: rows per lane");
            }
#undef BSQ_ROWS1
            return check_launch(e.nib ? "k_expand_rows1<nibbles>" : "k_expand_rows1");
        }
    }
    // knob "expand_gate": 0 automatic (rows of 24 ... 63 bytes), 1 never, 2 always (the scratch holds at least 256 bytes: Bp >= 256)
    const int gk = bsq_internal::tuning().expand_gate;
    const int64_t rowb = e.C * int64_t(sizeof(ST));
    const bool gated = (gk == 2 || (gk == 0 && rowb >= 24 && rowb < 64)) && e.Bp >= 256;
    if constexpr (sizeof(ST) >= 2) {
        if (e.nib) {  // ids as nibbles (two_pass_nibbles)
#define BSQ_EXPN(NTV, GV) hipLaunchKernelGGL((k_expand_chunks<ST, NTV, 0, GV, true>), grid, dim3(kThreads), pad, s, e)
            if (bsq_internal::nontemporal_stores()) { if (gated) BSQ_EXPN(true, true); else BSQ_EXPN(true, false); }
            else { if (gated) BSQ_EXPN(false, true); else BSQ_EXPN(false, false); }
#undef BSQ_EXPN
            return check_launch("k_expand_chunks<nibbles>");
        }
    }
    if (e.nib) return bsq_internal::set_error(BSQ_ERR_INVALID_ARG, "nibble ids: elements of 2 bytes and more, or one-byte rows of 3 ... 15 bytes");
    if (bsq_internal::nontemporal_stores()) {
        if (gated) hipLaunchKernelGGL((k_expand_chunks<ST, true, 0, true>), grid, dim3(kThreads), pad, s, e);
        else hipLaunchKernelGGL((k_expand_chunks<ST, true, 0>), grid, dim3(kThreads), pad, s, e);
    } else {
        if (gated) hipLaunchKernelGGL((k_expand_chunks<ST, false, 0, true>), grid, dim3(kThreads), pad, s, e);
        else hipLaunchKernelGGL((k_expand_chunks<ST, false, 0>), grid, dim3(kThreads), pad, s, e);
    }
    return check_launch("k_expand_chunks");
}

// Two-pass one-hot: raw (P,B) tokens into `workspace` (P*B bytes), then the chunk expansion.
int64_t two_pass_pitch(int64_t B) { return (B + kRawTB - 1) / kRawTB * kRawTB; }

// Pass 1: raw tokens (kNone = no token) of the batch into a (P, pitch) uint8 matrix.
bsq_status launch_tokens_raw(KParams &k, void *tokens, int64_t pitch, hipStream_t s, bool nib = false, int64_t tt0 = 0, int64_t ntt_count = 0) {
    k.out = tokens;
    k.out_pitch = pitch;
    k.aligned = reinterpret_cast<uintptr_t>(tokens) % 16 == 0 && pitch % 16 == 0;  // every row starts 16-byte aligned
    k.vw = k.aligned ? 16 : 1;
    k.ntb = int32_t((k.B + kRawTB - 1) / kRawTB);
    // no mask, 16-byte aligned rows: the register-transposed tiles of k_tokens_pb8_fast in raw-id mode (round 3; knob
    // tokens_pb8 = 1 or any raw_mode != 0: k_tokens_raw)
    if (!k.mask && k.desc && bsq_internal::tuning().raw_mode == 0 && bsq_internal::tokens_pb8_applicable(k.desc, k.B, k.P, tokens, pitch))
        return bsq_internal::launch_tokens_pb8(k.desc, k.chars, k.offsets, k.B, k.P, tokens, pitch, s, true, BSQ_I8, nib, tt0, ntt_count);
    if (nib || tt0 != 0 || ntt_count != 0) return bsq_internal::set_error(BSQ_ERR_INVALID_ARG, "nibble ids / slices need the register-transposed raw pass");
    const dim3 grid(unsigned(tile_grid(k, k.ntt)));
    // knob "raw_mode": 0 / 1 k_tokens_raw.  (k_tokens_raw2 -- a register transpose through v_permlane swaps -- LOST: profiles/r02/raw_lab.txt,
    // cfg2 as (P,B) int8 tokens 24.0 us -> 25.8 / 27.4; its source is csrc/labs/bsq_tokens_raw2.inc.)
    const int rm = bsq_internal::tuning().raw_mode;
    if (!k.mask && rm == 4 && k.P <= (int64_t(1) << 20)) {  // measurement: the wide tile of the (P,B) int8 token matrix
        k.ntb = int32_t((k.B + kWideTB - 1) / kWideTB);
        k.ntt = int32_t((k.P + kWideTT - 1) / kWideTT);
        if ((int64_t(k.ntb) + 8 * int64_t(k.group)) * int64_t(k.ntt) >= (int64_t(1) << 31)) k.order = 0;
        hipLaunchKernelGGL((k_tokens_raw<false, true, kWideTB, kWideTT>), dim3(unsigned(tile_grid(k, k.ntt))),
                           dim3(kThreads), 0, s, k);
        return check_launch("k_tokens_raw<wide>");
    }
    if (k.mask) hipLaunchKernelGGL(k_tokens_raw<true>, grid, dim3(kThreads), 0, s, k);
    else hipLaunchKernelGGL(k_tokens_raw<false>, grid, dim3(kThreads), 0, s, k);
    return check_launch("k_tokens_raw");
}

// Pass 2: the (P, B, C) one-hot as the flat expansion of a (P, pitch) raw token matrix.
bsq_status launch_expansion(const uint8_t *tokens, int64_t pitch, int64_t B, int64_t P, int32_t C, size_t sz,
                            uint64_t one_bits, void *out, hipStream_t s, int64_t row_gap = 0, bool nib = false) {
    EParams e;
    e.row_gap = row_gap;
    e.nib = nib ? 1 : 0;  // (k_expand_chunks only: see two_pass_nibbles)
    e.tok = tokens;
    e.B = B;
    e.Bp = pitch;
    e.out = static_cast<uint8_t *>(out);
    e.total = P * B * C * int64_t(sz);
    e.head = int32_t(reinterpret_cast<uintptr_t>(out) & (kChunk - 1));
    e.nchunks = (e.head + e.total + kChunk - 1) / kChunk;
    // a column block whose rows do not all start on chunk boundaries of memory: the ragged form (see EParams)
    const int64_t block_row = B * C * int64_t(sz);
    e.ragged = (row_gap != 0 && (e.head != 0 || block_row % kChunk != 0 || (block_row + row_gap) % kChunk != 0)) ? 1 : 0;
    e.npr8 = ((block_row + 2 * int64_t(kChunk) - 2) / kChunk + 7) / 8 * 8;
    e.inv_npr8 = 1.0 / double(e.npr8);
    if (e.ragged) {
        e.head = 0;
        e.nchunks = P * e.npr8;
    }
    e.C = C;
    e.one_bits = one_bits;
    e.inv_rowbytes = 1.0 / double(C * int64_t(sz));
    e.inv_B = 1.0 / double(B);
    e.pitch = B * C * int64_t(sz);
    e.dv_pitch = div64_constants(uint64_t(e.pitch));
    e.dv_rb = div64_constants(uint64_t(C * int64_t(sz)));
    div_constants(uint32_t(C * int64_t(sz)), &e.rb_magic, &e.rb_shift, &e.rb_pow2);
    e.force4 = bsq_internal::tuning().expand_slots == 4;
    switch (sz) {
    case 1: return launch_expand<uint8_t>(e, s);
    case 2: return launch_expand<uint16_t>(e, s);
    case 4: return launch_expand<uint32_t>(e, s);
    default: return launch_expand<uint64_t>(e, s);
    }
}

// How a two-pass one-hot is run (round 5).
//  * nib: ids as NIBBLES in the scratch -- alphabets of at most 15 classes (DNA, the reduced amino alphabets; 15 = no one) whose raw pass is
//    k_tokens_pb8_fast and whose expansion is k_expand_chunks (elements of 2 bytes and more) or k_expand_rows1 (one-byte rows of 3 ... 15
//    bytes).  The scratch is written and re-read at half its bytes.  Automatic for rows of 24 ... 31 bytes, where it wins on every shape
//    tried (cfg4 f32 673 -> 648 us; 0.5-4 % elsewhere; other widths of k_expand_chunks lose 1-3 % to the byte form:
//    profiles/r05/nibble_ids_lab.txt), and for every one-byte row (2-9 %: rows1_nib_sweep.txt).  Knob "raw_nibbles": 1 never, 2 whenever they apply.
//  * tiles_per_slice: the matrix in SLICES of 64-position tiles, raw pass and expansion of one slice after the other through ONE scratch of
//    a slice's size.  The expansion runs at the write roof only while the ids it reads come out of the Infinity Cache: with 268 MB of ids
//    (262 144 x 1024 AMINO20) the int8 one-hot fell from 0.85 to 0.68 of the roof and the f32 one from 0.94 to 0.72, 2M x 160 DNA f32 to 0.59.
//    Slices of <= 96 MB keep every size at the small batches' rate, for two more launches per slice.  Knob "two_pass_slice_mb".
struct TwoPassPlan {
    int64_t pitch;            // ids per scratch row
    bool nib, nib_ok;         // ids as nibbles; whether they could be
    bool wants_slices;        // the id matrix is too large for one piece (it may still BE one piece: a single position tile cannot be cut)
    int64_t ntt, tiles_per_slice;
    size_t ws_bytes;
};
TwoPassPlan two_pass_plan(const KParams &k, size_t sz) {
    const auto &tn = bsq_internal::tuning();
    TwoPassPlan pl;
    pl.pitch = two_pass_pitch(k.B);  // padded: every scratch row is aligned, full-width vector stores
    pl.ntt = (k.P + kTT - 1) / kTT;
    const int64_t rb = k.C * int64_t(sz);
    // the register-transposed raw pass (no mask, foldable or LDS table, ids < 251): what nibbles and slices are built on
    const bool pb8 = !k.mask && k.desc && tn.raw_mode == 0 &&
                     bsq_internal::tokens_pb8_applicable(k.desc, k.B, k.P, reinterpret_cast<const void *>(uintptr_t(256)), pl.pitch);
    const bool rows1 = sz == 1 && rb >= 3 && rb <= 15 && tn.expand_rows1 != 1;  // (launch_expand: one-byte rows expand through k_expand_rows1)
    const bool nib_ok = pb8 && k.C <= 15 && (sz >= 2 || rows1);
    // (automatic: rows of 24 ... 31 bytes, and every id matrix beyond 128 MB as bytes -- half the scratch to keep resident, slices of twice the rows)
    pl.nib_ok = nib_ok && tn.raw_nibbles != 1;
    // (one-byte rows through k_expand_rows1<nibbles>: ahead of byte ids on every shape tried, 2-9 % -- profiles/r05/rows1_nib_sweep.txt)
    pl.nib = nib_ok && (tn.raw_nibbles == 2 || (tn.raw_nibbles == 0 && (rows1 || (rb >= 24 && rb < 32) || pl.pitch * k.P > (int64_t(128) << 20))));
    const int64_t row_bytes = pl.pitch >> (pl.nib ? 1 : 0), all_bytes = row_bytes * k.P;
    pl.tiles_per_slice = pl.ntt;
    pl.wants_slices = false;
    const int64_t mb = tn.two_pass_slice_mb;
    // (slices of a column block that do not start on a chunk boundary take the ragged form, see EParams: no condition on the gap)
    if (pb8 && mb >= 0) {
        const int64_t slice_bytes = (mb > 0 ? mb : 96) << 20;
        if ((mb > 0 && all_bytes > slice_bytes) || all_bytes > (int64_t(128) << 20)) {
            pl.wants_slices = true;
            const int64_t tile_bytes = row_bytes * kTT;
            const int64_t max_tiles = slice_bytes / tile_bytes > 1 ? slice_bytes / tile_bytes : 1;  // (a single tile may exceed the target: B beyond 1.5 M)
            const int64_t nslices = (pl.ntt + max_tiles - 1) / max_tiles;
            pl.tiles_per_slice = (pl.ntt + nslices - 1) / nslices;  // balanced, none above max_tiles
        }
    }
    const int64_t rows = pl.tiles_per_slice * kTT < k.P ? pl.tiles_per_slice * kTT : k.P;
    pl.ws_bytes = size_t(pl.tiles_per_slice == pl.ntt ? pl.pitch * k.P : row_bytes * rows);  // (one slice: the byte-sized scratch as before)
    return pl;
}

// Position slices thinner than four tiles are a poor cut: every 64-position tile's raw pass fetches the 128-byte lines its neighbour
// fetched (short reads: 150 characters = three tiles, each touching ~1.5 lines per sequence), and with B beyond 1.5 M a single tile is
// too much id matrix already -- cfg4 f32 at 2 M / 4 M reads ran at 0.80 / 0.63 of the roof that way.  Such batches are cut by SEQUENCES
// instead: column blocks of nb sequences (a multiple of 4096 / gcd(row bytes, 4096), so that every block's rows are whole chunks), each a
// two-pass stream of its own with a gap after every position row (bsq_onehot_block_device) whose ids fit the slice target in one piece.
// Returns nb, or 0 when the batch is not to be cut this way.
int64_t two_pass_sequence_block(const KParams &k, size_t sz, const void *out) {
    const TwoPassPlan pl = two_pass_plan(k, sz);
    // (fat position slices are fine; thin ones, and a matrix of a single tile -- padlen <= 64 -- that is too large all the same, are cut here)
    // (any alignment of the result since the ragged block form: a tensor torch placed 2560 bytes off a chunk is cut like an aligned one)
    (void)out;
    if (!pl.wants_slices || pl.tiles_per_slice >= 4) return 0;
    const int64_t mb = bsq_internal::tuning().two_pass_slice_mb;
    const int64_t rb = k.C * int64_t(sz);
    int64_t g = rb, h = kChunk;
    while (h) {
        const int64_t r = g % h;
        g = h;
        h = r;
    }
    const int64_t m = kChunk / g;  // sequences per period of (b * rb) mod 4096
    const int64_t ids = (((mb > 0 ? mb : 96) << 20)) << (pl.nib_ok ? 1 : 0);  // a block's ids in one slice: 96 MB as bytes or as nibbles
    int64_t nb = ids / k.P / m * m;
    if (nb < m) nb = m;
    if (nb >= k.B) return 0;
    // blocks of equal size (2M reads as 1M + 1M, not 1.26M + 0.74M: the last, short block's launches are no cheaper per sequence)
    const int64_t nblocks = (k.B + nb - 1) / nb;
    const int64_t even = ((k.B + nblocks - 1) / nblocks + m - 1) / m * m;
    return even < nb ? even : nb;
}

// The caller holds nothing: the scratch is acquired here (shared by the calls of one stream -- workspace cache --, so the launches of a
// call are enqueued back to back under the workspace mutex).
bsq_status onehot_two_pass(KParams &k, size_t sz, hipStream_t s, int64_t row_gap = 0) {
    const TwoPassPlan pl = two_pass_plan(k, sz);
    std::lock_guard<std::mutex> two_pass_turn(bsq_internal::workspace_mutex());
    void *ws = nullptr;
    bsq_status st = bsq_internal::workspace_acquire(pl.ws_bytes, s, &ws);
    if (st != BSQ_OK) return st;
    uint8_t *out = static_cast<uint8_t *>(k.out);
    const int64_t out_row = k.B * k.C * int64_t(sz) + row_gap;  // bytes from one position row of the result to the next
    if (pl.tiles_per_slice >= pl.ntt) {
        st = launch_tokens_raw(k, ws, pl.pitch, s, pl.nib);
        if (st == BSQ_OK) st = launch_expansion(static_cast<const uint8_t *>(ws), pl.pitch, k.B, k.P, k.C, sz, k.one_bits, out, s, row_gap, pl.nib);
    } else {
        for (int64_t tt0 = 0; tt0 < pl.ntt && st == BSQ_OK; tt0 += pl.tiles_per_slice) {
            const int64_t cnt = pl.ntt - tt0 < pl.tiles_per_slice ? pl.ntt - tt0 : pl.tiles_per_slice;
            const int64_t p0 = tt0 * kTT, p1 = (tt0 + cnt) * kTT < k.P ? (tt0 + cnt) * kTT : k.P;
            st = launch_tokens_raw(k, ws, pl.pitch, s, pl.nib, tt0, cnt);  // rows p0 .. p1 - 1 of the id matrix into the scratch
            if (st == BSQ_OK)
                st = launch_expansion(static_cast<const uint8_t *>(ws), pl.pitch, k.B, p1 - p0, k.C, sz, k.one_bits, out + p0 * out_row, s, row_gap, pl.nib);
        }
    }
    bsq_internal::workspace_release(ws, s);
    return st;
}

template <typename T>
bsq_status launch_expand_bcl(const uint8_t *tokens, int64_t B, int64_t P, int32_t C, uint64_t one_bits, void *out, hipStream_t s) {
    BParams b;
    b.tok = tokens;
    b.out = static_cast<uint8_t *>(out);
    b.nrows = B * C;
    b.P = P;
    b.C = C;
    b.total = b.nrows * P * int64_t(sizeof(T));
    b.nchunks = (b.total + kChunk - 1) / kChunk;
    b.one_bits = one_bits;
    b.inv_P = 1.0 / double(P);
    div_constants(uint32_t(P), &b.magic, &b.shift, &b.pow2);
    div_constants(uint32_t(C), &b.magic_c, &b.shift_c, &b.pow2_c);
    const int64_t groups = ((b.nchunks + 7) / 8 + 3) / 4;
    if (groups * 8 >= (int64_t(1) << 31)) return bsq_internal::set_error(BSQ_ERR_INVALID_ARG, "output too large");
    const int padv = bsq_internal::tuning().bcl_pad;  // unused dynamic LDS = occupancy cap: 0 -> 3 workgroups per CU
    const size_t pad = padv > 0 ? size_t(padv) : (padv < 0 ? size_t(0) : size_t(53248));
    if (bsq_internal::nontemporal_stores())
        hipLaunchKernelGGL((k_expand_bcl<T, true>), dim3(unsigned(groups * 8)), dim3(kThreads), pad, s, b);
    else
        hipLaunchKernelGGL((k_expand_bcl<T, false>), dim3(unsigned(groups * 8)), dim3(kThreads), pad, s, b);
    return check_launch("k_expand_bcl");
}

}  // namespace

extern "C" {

// 0 generic, 1 tiled, 2 two-pass, 3 chunk-owner
static int choose_onehot_path(int32_t C, size_t sz, int64_t B, int64_t P, bool misaligned_out = false, bool masked = false) {
    const int64_t ntiles = ((B + 63) / 64) * ((P + kTT - 1) / kTT);
    const bool tiled_ok = C <= 250 && 4 * (64 * C * int64_t(sz) + 16) + tile_fixed_bytes<64>() <= 60 * 1024 &&
                          ntiles < (int64_t(1) << 31) && B < (int64_t(1) << 31) - 256 && P <= kMaxTiledP;
    if (!tiled_ok) return 0;
    int path = bsq_internal::tuning().onehot_path;
    if (path == 0) {
        // Measured on MI355X over 22 shapes (profiles/r01/sweep_shapes5.txt, sweep_occupancy2.txt):
        //  2 two-pass   : the fastest streamer once the output is large -- 7.3-7.4 TB/s at 3 workgroups per CU
        //                 when rows are >= 64 B, 6-7 TB/s for smaller rows, any pitch; needs rows >= 16 B and an
        //                 output that amortises the token pass and the second launch;
        //  3 chunk-owner: one launch, no scratch: ahead below ~1.5 GiB of output (round 1 measured it ahead up to 2.7 GB:
        //                 0.383 vs 0.390 ms; 5.4 GB: 0.757 vs 0.733) when a row is >= 48 B and its
        //                 per-position gather set stays L2-resident, i.e. the pitch is a multiple of 32 KiB (each
        //                 XCD keeps to its own chunk columns) and B <= 128k, or B <= 16k whatever the pitch;
        //                 Small batches (profiles/r02/path_small.txt): the tile kernel has a ~15 us floor, the
        //                 chunk-owner none -- 1000 x 256 DNA f32 6 vs 19 us, 1024 x 512 DNA int8 9 vs 16 us,
        //                 4096 x 512 DNA f32 17 vs 20 us -- so it also takes every output <= 8 MB and rows >= 24 B
        //                 up to 128 MB;
        //  1 tiled      : the rest (tiny rows such as int8 DNA once the batch is not small, 64-byte aligned position rows).
        const int64_t rowbytes = C * int64_t(sz), pitch = B * rowbytes, total = pitch * P;
        const bool pinned_columns = pitch % (8 * kChunk) == 0;
        const bool owner_big = rowbytes >= 48 && ((pinned_columns && B <= 131072) || B <= 16384);
        // (round 6, scripts/check_dispatch.py: rows of 16 ... 23 bytes up to 40 MB too -- 4096 x 512 DNA f32 10.2 us against 14.4 tiled, 2048 x 512
        //  6.3 against 14.2; at 67 MB and beyond the tiled kernel is level or ahead: profiles/r06/dispatch_check.txt)
        const bool owner_small = total <= (int64_t(8) << 20) || (rowbytes >= 24 && total <= (int64_t(128) << 20)) ||
                                 (rowbytes >= 16 && total <= (int64_t(40) << 20));
        // (end of round 2: two-pass is 3-4 % ahead from 2 GB on -- 32768 x 1024 AMINO20 f32 0.376 vs 0.389 ms --, level at
        // 1.3 GB and behind below: profiles/r02/sweep_shapes_final.txt)
        if ((owner_big || owner_small) && total < (int64_t(3) << 29))
            path = 3;
        else if ((rowbytes >= 16 && total >= (int64_t(128) << 20)) || (sz <= 2 && rowbytes >= (sz == 1 ? 8 : 14) && total >= (int64_t(192) << 20)) ||
                 ((pitch % 64 != 0 || misaligned_out) && total >= (int64_t(32) << 20)))
            path = 2;  // (second case: position rows that are not 64-byte aligned -- the tiles would share memory sectors or
                       // fall to element stores: 250001 x 256 int8 DNA 187 -> 120 us, profiles/r02/path_unaligned.txt; round 5: -> 94 us.
                       // Round 5: one-byte rows of 8 ... 15 bytes too -- their expansion is k_expand_rows1: SEB14 131072 x 512 int8
                       // 187 -> 156 us, SEB8 + BOS / EOS / PAD (11-byte rows) 281 -> 248 us; rows of 3 ... 7 bytes on BYTE ids: cfg4 int8
                       // 225 us tiled, 240 two-pass -- profiles/r05/rows1_lab.txt; on nibble ids: the next rule)
        else if (sz == 1 && rowbytes >= 3 && rowbytes < 8 && total >= (int64_t(192) << 20) &&
                 !masked && bsq_internal::tuning().expand_rows1 != 1 && bsq_internal::tuning().raw_nibbles != 1 && bsq_internal::tuning().raw_mode == 0 &&
                 bsq_internal::tuning().tokens_pb8 != 1)
            path = 2;  // (end of round 5) rows of 3 ... 7 bytes -- BASELINE config 4's default dtype: 1M x 160 DNA int8 -- once the ids are NIBBLES
                       // and expand through k_expand_rows1<nibbles>: 227 -> 207 us on a resident batch, 257-268 -> 222 us on fresh batches (0.60 ->
                       // 0.72 of the roof); 4-, 5-, 6-byte rows 188 -> 153, 204 -> 174, 214 -> 198 us cold.  Over 35 shapes
                       // (profiles/r05/rows1_nib_sweep.txt) the pair runs at 0.73-0.80 whatever the shape, the tiled kernel at 0.62-0.72 -- each of
                       // its tiles fetches the character lines its neighbours fetch, which fresh inputs pay at HBM -- except long reads at a
                       // chunk-aligned pitch (0.75-0.84: 262144 x 512, 131072 x 1024), where 7-byte rows still tie (165 vs 166 us) and narrower
                       // ones lost 6-10 % (DNA5 131072 x 1024: 124 vs 131 us) -- round 5 kept those tiled.  Round 6's on-box check
                       // (scripts/check_dispatch.py, fresh inputs, full-length and ragged reads) finds the tiled kernel 15-33 % BEHIND on exactly
                       // those shapes (DNA5 131072 x 1024: 170-183 us against 137; DNA4 C = 4: 145-157 against 119; 6-byte rows 262144 x 512:
                       // 187-197 against 154-163): the exception is gone (profiles/r06/dispatch_check.txt).
                       // Rules of round 6 next to it: rows >= 16 B go two-pass from 128 MB (16384 x 512 DNA f32: 28.0 us against 30.7 tiled);
                       // 2-byte elements with rows of 14 / 15 bytes from 192 MB (1M x 160 DNA int16: 412 against 445; 262144 x 512: 332 against 388 --
                       // rows of 8 ... 12 bytes of 2-byte elements stay tiled: 270 against 280-309, 316 against 328-354, 184 against 196)
        else
            path = 1;
    }
    return path;
}

const char *bsq_onehot_kernel_name(const bsq_desc *d, int64_t B, int64_t P, bsq_dtype t) {
    if (!d) return "";
    switch (choose_onehot_path(bsq_alphabet_size(d), bsq_dtype_size(t), B, P)) {
    case 1: return "k_onehot_tile";
    case 2: {
        const int64_t rb = bsq_alphabet_size(d) * int64_t(bsq_dtype_size(t));
        // (unmasked: the raw-id pass runs in k_tokens_pb8_fast unless a knob keeps it in k_tokens_raw -- see launch_tokens_raw)
        const bool rows1 = bsq_dtype_size(t) == 1 && rb >= 3 && rb <= 15 && bsq_internal::tuning().expand_rows1 != 1;
        if (bsq_internal::tuning().raw_mode == 0 && bsq_internal::tuning().tokens_pb8 != 1) {
            KParams kp;  // (what two_pass_plan looks at)
            kp.desc = d;
            kp.mask = nullptr;
            kp.B = B;
            kp.P = P;
            kp.C = bsq_alphabet_size(d);
            if (two_pass_plan(kp, bsq_dtype_size(t)).nib)
                return rows1 ? "k_tokens_pb8_fast<raw, nibbles>+k_expand_rows1<nibbles>" : "k_tokens_pb8_fast<raw, nibbles>+k_expand_chunks<nibbles>";
            return rows1 ? "k_tokens_pb8_fast<raw>+k_expand_rows1" : "k_tokens_pb8_fast<raw>+k_expand_chunks";
        }
        return rows1 ? "k_tokens_raw+k_expand_rows1" : "k_tokens_raw+k_expand_chunks";
    }
    case 3: return "k_onehot_chunks";
    default: return "k_onehot_generic";
    }
}

bsq_status bsq_onehot_device(const bsq_desc *d, const uint8_t *chars, const int64_t *offsets,
                             const uint8_t *mask_or_null, int64_t B, int64_t P, bsq_dtype t, void *out,
                             void *hip_stream) {
    KParams k;
    bsq_status st = fill_common(k, d, chars, offsets, mask_or_null, B, P, out);
    if (st != BSQ_OK) return st;
    if (B == 0) return BSQ_OK;
    const size_t sz = bsq_dtype_size(t);
    if (sz == 0) return bsq_internal::set_error(BSQ_ERR_DTYPE, "bad bsq_dtype");
    // Limits of the LDS kernels: 8-bit token ids, 32-bit tile arithmetic, row images must fit in LDS.
    // (a result that does not start on a 64-byte boundary -- a view into a larger tensor -- counts as misaligned: the tiled kernel's
    //  row segments then straddle memory sectors, cfg4 int8 16 bytes off: 225 -> 330 us tiled, 242 us two-pass)
    const int path = choose_onehot_path(k.C, sz, B, P, reinterpret_cast<uintptr_t>(out) % 64 != 0, k.mask != nullptr);
    if (path == 0) return bsq_onehot_device_generic(d, chars, offsets, mask_or_null, B, P, t, out, hip_stream);
    k.one_bits = one_bits_of(t);
    const int64_t pitch = B * k.C * int64_t(sz);
    k.aligned = (reinterpret_cast<uintptr_t>(out) % 16 == 0) && (pitch % 16 == 0);
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    // Path selection (tuning knob "onehot_path": 0 auto, 1 tiled, 2 two-pass, 3 chunk-owner).
    // Measured on MI355X (profiles/r01/sweep_shapes.txt): the chunk-owner kernel streams at ~7 TB/s when a
    // row is >= 48 bytes and its per-position gather set stays L2-resident -- i.e. the row pitch is a
    // multiple of 32 KiB (each XCD then keeps to its own chunk columns) and B is moderate, or B is small.
    if (path == 3) return onehot_chunk_owner(k, sz, s);
    if (path == 2) {
        // (the scratch is shared by the calls of one stream -- workspace cache --: onehot_two_pass enqueues its launches back to back
        //  under the workspace mutex; concurrent host threads take turns there: enqueueing takes microseconds, the GPU work overlaps)
        if (const int64_t nb = two_pass_sequence_block(k, sz, out)) {  // short reads, very many of them: column blocks of nb sequences
            const int64_t rb = k.C * int64_t(sz);
            for (int64_t b0 = 0; b0 < B; b0 += nb) {
                const int64_t n = B - b0 < nb ? B - b0 : nb;
                st = bsq_onehot_block_device(d, chars, offsets + b0, mask_or_null, n, P, t, static_cast<uint8_t *>(out) + b0 * rb, B, hip_stream);
                if (st != BSQ_OK) return st;
            }
            return BSQ_OK;
        }
        return onehot_two_pass(k, sz, s);
    }
    switch (sz) {
    case 1: return dispatch_onehot_tile<uint8_t>(k, s);
    case 2: return dispatch_onehot_tile<uint16_t>(k, s);
    case 4: return dispatch_onehot_tile<uint32_t>(k, s);
    default: return dispatch_onehot_tile<uint64_t>(k, s);
    }
}

bsq_status bsq_onehot_block_device(const bsq_desc *d, const uint8_t *chars, const int64_t *offsets,
                                   const uint8_t *mask_or_null, int64_t B, int64_t P, bsq_dtype t, void *out, int64_t row_seqs,
                                   void *hip_stream) {
    if (row_seqs < B) return bsq_internal::set_error(BSQ_ERR_INVALID_ARG, "row_seqs < B");
    if (row_seqs == B) return bsq_onehot_device(d, chars, offsets, mask_or_null, B, P, t, out, hip_stream);
    KParams k;
    bsq_status st = fill_common(k, d, chars, offsets, mask_or_null, B, P, out);
    if (st != BSQ_OK) return st;
    if (B == 0) return BSQ_OK;
    const size_t sz = bsq_dtype_size(t);
    if (sz == 0) return bsq_internal::set_error(BSQ_ERR_DTYPE, "bad bsq_dtype");
    if (reinterpret_cast<uintptr_t>(out) % sz) return bsq_internal::set_error(BSQ_ERR_INVALID_ARG, "output is not aligned to its element size");
    const int block_path = choose_onehot_path(k.C, sz, B, P, false);
    if (block_path == 0) return bsq_internal::onehot_generic_block(d, chars, offsets, mask_or_null, B, P, t, out, row_seqs, hip_stream);
    k.one_bits = one_bits_of(t);
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    const int64_t block_pitch = B * k.C * int64_t(sz), rb = k.C * int64_t(sz);
    // (rows that the chunk stream expands well: 16 bytes and more, and -- end of round 5 -- one-byte rows of 3 ... 15 bytes through
    //  k_expand_rows1; unmasked only: a masked raw pass is k_tokens_raw with byte ids -- such blocks stay with the tiled kernel as before)
    const bool stream_rows = rb >= 16 || (sz == 1 && rb >= 3 && !k.mask && bsq_internal::tuning().expand_rows1 != 1);
    const bool whole_chunks = block_pitch % kChunk == 0 && reinterpret_cast<uintptr_t>(out) % kChunk == 0;
    // A block whose position rows are whole 4-KiB chunks of memory (B * C * sizeof(T) and the address of its first element multiples of
    // 4096: e.g. any multiple of 4096 sequences at a 4096-sequence boundary of an aligned tensor) is the two-pass stream with a gap after
    // every row: no chunk straddles two rows (16 384-sequence blocks of cfg3: 4 x 0.19 ms against 4 x 0.25 ms for the tiles).  Any other block of 128 MB and more -- ragged shards of a sharded job
    // (sharding.store_shard_into_root), the last piece of a host batch, a tensor whose pitch is no multiple of 4 KiB -- takes the RAGGED
    // form of the same stream (EParams::ragged: every row cut at the chunk boundaries of memory, its first and last piece partial).
    // Rounds 4-5 cut such blocks in three calls instead (the sequences up to the first chunk boundary, the run of whole chunks, the
    // rest): two ~15-us side launches, and only row 0 of the middle run aligned when the tensor's pitch was not.
    // (knob onehot_path = 2: whatever the size -- tests)
    if (block_path != 1 && stream_rows && (whole_chunks || block_pitch * P >= (int64_t(128) << 20) || bsq_internal::tuning().onehot_path == 2)) {
        if (const int64_t nb = two_pass_sequence_block(k, sz, out)) {  // (a block of very many short reads: sub-blocks, see bsq_onehot_device)
            for (int64_t b0 = 0; b0 < B; b0 += nb) {
                const int64_t n = B - b0 < nb ? B - b0 : nb;
                st = bsq_onehot_block_device(d, chars, offsets + b0, mask_or_null, n, P, t, static_cast<uint8_t *>(out) + b0 * rb, row_seqs, hip_stream);
                if (st != BSQ_OK) return st;
            }
            return BSQ_OK;
        }
        return onehot_two_pass(k, sz, s, (row_seqs - B) * k.C * int64_t(sz));
    }
    // otherwise the tiled kernel: a workgroup owns (sequence tile x 64 positions) and writes one row SEGMENT per position, so a row
    // pitch other than B * C is just another stride
    k.row_seqs = row_seqs;
    // 16-byte stores: the block's first element, the row pitch AND the block's own row segment must be multiples of 16 (the last tile's
    // segment ends where the block ends; with the whole tensor that is the pitch, here it is not: a block that started aligned and
    // ended 8 bytes short of a line wrote those 8 bytes of its neighbour -- found in round 4 by the misaligned-result test)
    k.aligned = (reinterpret_cast<uintptr_t>(out) % 16 == 0) && ((row_seqs * k.C * int64_t(sz)) % 16 == 0) && (block_pitch % 16 == 0);
    switch (sz) {
    case 1: return dispatch_onehot_tile<uint8_t>(k, s);
    case 2: return dispatch_onehot_tile<uint16_t>(k, s);
    case 4: return dispatch_onehot_tile<uint32_t>(k, s);
    default: return dispatch_onehot_tile<uint64_t>(k, s);
    }
}

bsq_status bsq_raw_tokens_device(const bsq_desc *d, const uint8_t *chars, const int64_t *offsets,
                                 const uint8_t *mask_or_null, int64_t B, int64_t P, uint8_t *tokens, int64_t pitch,
                                 void *hip_stream) {
    KParams k;
    bsq_status st = fill_common(k, d, chars, offsets, mask_or_null, B, P, tokens);
    if (st != BSQ_OK) return st;
    if (pitch < B) return bsq_internal::set_error(BSQ_ERR_INVALID_ARG, "pitch < B");
    if (B == 0) return BSQ_OK;
    if (k.C > 250 || P > kMaxTiledP || B >= (int64_t(1) << 31) - 256 || ((B + kRawTB - 1) / kRawTB) * k.ntt >= (int64_t(1) << 31))
        return bsq_internal::set_error(BSQ_ERR_INVALID_ARG, "raw tokens need ids < 251, padlen <= 2^22 and < 2^31 tiles");
    return launch_tokens_raw(k, tokens, pitch, static_cast<hipStream_t>(hip_stream));
}

bsq_status bsq_onehot_from_raw_tokens_device(const uint8_t *tokens, int64_t pitch, int64_t B, int64_t P, int32_t C,
                                             bsq_dtype t, void *out, void *hip_stream) {
    const size_t sz = bsq_dtype_size(t);
    if (sz == 0) return bsq_internal::set_error(BSQ_ERR_DTYPE, "bad bsq_dtype");
    if (B < 0 || P <= 0 || C <= 0 || C > 250 || pitch < B || (B > 0 && (!tokens || !out)))
        return bsq_internal::set_error(BSQ_ERR_INVALID_ARG, "null pointer, bad shape or pitch < B");
    if (B == 0) return BSQ_OK;
    if (reinterpret_cast<uintptr_t>(out) % sz) return bsq_internal::set_error(BSQ_ERR_INVALID_ARG, "output is not aligned to its element size");
    return launch_expansion(tokens, pitch, B, P, C, sz, one_bits_of(t), out, static_cast<hipStream_t>(hip_stream));
}

// Channels-first one-hot (B, C, P).  Fast path: chunk kernel (needs P % (16/sizeof(T)) == 0, a 16-byte
// aligned base and 8-bit ids); otherwise the generic element kernel.
bsq_status bsq_onehot_bcl_device(const bsq_desc *d, const uint8_t *chars, const int64_t *offsets,
                                 const uint8_t *mask_or_null, int64_t B, int64_t P, bsq_dtype t, void *out,
                                 void *hip_stream) {
    KParams k;
    bsq_status st = fill_common(k, d, chars, offsets, mask_or_null, B, P, out);
    if (st != BSQ_OK) return st;
    if (B == 0) return BSQ_OK;
    const size_t sz = bsq_dtype_size(t);
    if (sz == 0) return bsq_internal::set_error(BSQ_ERR_DTYPE, "bad bsq_dtype");
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    k.one_bits = one_bits_of(t);
    // Two-pass form (raw (B,P) ids, then k_expand_bcl) for large outputs, masked or not; knob "bcl_path": 0 automatic,
    // 1 never, 2 whenever it applies.
    const int bcl_path = bsq_internal::tuning().bcl_path;
    const int64_t total_bytes = B * int64_t(k.C) * P * int64_t(sz);
    if (bcl_path != 1 && (bcl_path == 2 || total_bytes >= (int64_t(256) << 20)) && k.C <= 250 &&
        reinterpret_cast<uintptr_t>(out) % 16 == 0 && P % 16 == 0 && B * int64_t(k.C) * P < (int64_t(1) << 51) &&
        bsq_internal::tokens_bp8_applicable(d, B, P, out)) {
        // In SLICES of sequences once the (B,P) id matrix exceeds 128 MB (round 5: like the seq-first two-pass form, the expansion holds the
        // roof only while its ids come out of the Infinity Cache -- 262 144 x 1024 f32: 0.60 of the roof in one piece).  A sequence's rows
        // are contiguous in the (B,C,P) result, so a slice is simply a smaller batch: sequences [b0, b0 + nb) with offsets + b0 (the
        // offsets are absolute), nb a multiple of 256 (every slice then starts on a 4-KiB boundary of the result when the result does).
        // Knob "two_pass_slice_mb" as in onehot_two_pass.
        const int64_t mb = bsq_internal::tuning().two_pass_slice_mb;
        int64_t per_slice = B;
        if (mb >= 0 && (mb > 0 || B * P > (int64_t(128) << 20))) {
            per_slice = (((mb > 0 ? mb : 96) << 20) / P) / 256 * 256;
            if (per_slice < 256) per_slice = 256;
        }
        std::lock_guard<std::mutex> two_pass_turn(bsq_internal::workspace_mutex());  // see onehot_two_pass
        void *ws = nullptr;
        bsq_status wst = bsq_internal::workspace_acquire(size_t(per_slice < B ? per_slice : B) * size_t(P), s, &ws);
        if (wst != BSQ_OK) return wst;
        const uint8_t *tk = static_cast<const uint8_t *>(ws);
        for (int64_t b0 = 0; b0 < B && wst == BSQ_OK; b0 += per_slice) {
            const int64_t nb = B - b0 < per_slice ? B - b0 : per_slice;
            uint8_t *dst = static_cast<uint8_t *>(out) + b0 * int64_t(k.C) * P * int64_t(sz);
            wst = bsq_internal::launch_tokens_bp8(d, chars, offsets + b0, nb, P, ws, s, true, mask_or_null);
            if (wst != BSQ_OK) break;
            switch (sz) {
            case 1: wst = launch_expand_bcl<uint8_t>(tk, nb, P, k.C, k.one_bits, dst, s); break;
            case 2: wst = launch_expand_bcl<uint16_t>(tk, nb, P, k.C, k.one_bits, dst, s); break;
            case 4: wst = launch_expand_bcl<uint32_t>(tk, nb, P, k.C, k.one_bits, dst, s); break;
            default: wst = launch_expand_bcl<uint64_t>(tk, nb, P, k.C, k.one_bits, dst, s); break;
            }
        }
        bsq_internal::workspace_release(ws, s);
        return wst;
    }
    if (k.C <= 250 && reinterpret_cast<uintptr_t>(out) % sz == 0 &&
        ((reinterpret_cast<uintptr_t>(out) % 16 == 0 && P % int64_t(16 / sz) == 0) || bcl_path != 3)) {  // knob 3: aligned only
        return bsq_internal::launch_onehot_bcl_chunks(d, chars, offsets, mask_or_null, B, P, t, out, s);  // (bsq_tokens.hip: k_tokenize_chunks<HOT>)
    }
    return bsq_internal::onehot_generic_bcl(d, chars, offsets, mask_or_null, B, P, t, out, hip_stream);
}

}  // extern "C"
