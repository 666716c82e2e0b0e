// The TOKEN kernels of libbsq_hip.so (batch_tokenize; /root/reference/src/tokenize.h:381-485) for every element type and both layouts, gfx950
// only -- except the two fast int8 / byte-transposing kernels, which have their own unit (bsq_tokens8.hip):
//   k_tokenize_chunks  (B,P) tokens and, with HOT, the channels-first (B,C,P) one-hot: flat chunk stream, a lane owns 16 output bytes of
//                      one sequence row, unaligned vector loads of its characters; row-piece form for any padlen / alignment.
//   k_tokenize_rows    (B,P) tokens, one wave per sequence (knob tokenize_path = 1 / 2: the round-1 form).
//   k_tokenize_tile    (P,B) tokens of 2- / 4- / 8-byte elements: tiled transpose through LDS.
//   k_tokens_raw<value> (bsq_tiles.h)  (P,B) int8 tokens where k_tokens_pb8_fast does not apply.
// Split out of bsq_kernels.hip in round 5; the shared tile machinery is bsq_tiles.h, the one-hot kernels bsq_onehot.hip.
#include "bsq_tiles.h"

namespace {

// ------------------------------------------------------------------------------------------
// Tokens, (P,B) layout, tiled: phase 1 as above, then a transposed read of the token tile.
// ------------------------------------------------------------------------------------------
// Token id (< 256) as a value of type T.  double goes through the 2^52 trick -- bits(2^52 + k) = 0x4330000000000000 | k,
// minus 2^52 is exact -- because v_cvt_f64_u32 is slow on this part (f64 token matrices ran 1.7x slower than int64
// ones with the plain cast; the f32 cast is full rate).
template <typename T>
__device__ __forceinline__ T id_as(uint32_t tk) {
    if constexpr (std::is_same<T, double>::value)
        return __hiloint2double(0x43300000, static_cast<int>(tk)) - 4503599627370496.0;
    else
        return static_cast<T>(tk);
}

template <typename T>
__device__ __forceinline__ T token_value(uint32_t tk) {
    return tk == kNone ? T(0) : id_as<T>(tk);  // unmapped / unpadded positions keep the memset 0
}

template <typename T, int TB>
__global__ __launch_bounds__(kThreads) void k_tokenize_tile(const KParams p) {
    extern __shared__ __align__(16) uint8_t smem[];
    SeqSpan *s_span = reinterpret_cast<SeqSpan *>(smem);
    uint8_t *s_lut = smem + tile_off_bytes<TB>();
    uint8_t *s_tok = s_lut + 256;

    const int tid = threadIdx.x;
    int32_t tb, tt;
    tile_of_block(p, tb, tt);
    if (tb >= p.ntb) return;
    const int64_t b0 = static_cast<int64_t>(tb) * TB;
    const int32_t t0 = tt * kTT;
    build_token_tile<TB>(p, b0, t0, s_lut, s_span, s_tok);

    constexpr int EPC = 16 / static_cast<int>(sizeof(T));  // elements per 16-byte chunk
    constexpr int CPR = TB / EPC;                          // chunks per row segment
    T *out = static_cast<T *>(p.out);
    if (p.vw == 2) {
        // Rows that are only element-aligned (odd batch sizes, offset outputs): the row segment of the tile is cut at the
        // 16-byte lines of the OUTPUT -- slot 0 = the head (the 0 .. EPC-1 elements up to the first line), then whole
        // aligned pieces (nt stores as in the aligned case), the last one the tail; heads and tails as element stores.
        const int64_t nb64 = p.B - b0;
        const int32_t nb = nb64 < TB ? static_cast<int32_t>(nb64) : TB;  // sequences of the tile
        const uint32_t a0e = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(p.out) & 15u) / static_cast<uint32_t>(sizeof(T));
        for (int f = tid; f < kTT * (CPR + 1); f += kThreads) {
            const int32_t tl = f / (CPR + 1), slot = f % (CPR + 1);
            const int64_t t = static_cast<int64_t>(t0) + tl;
            if (t >= p.P) continue;
            const int64_t e0 = t * p.B + b0;  // element index of the segment's first element
            const int32_t h = static_cast<int32_t>((EPC - ((a0e + static_cast<uint32_t>(e0)) & (EPC - 1))) & (EPC - 1));
            const int32_t sb0 = slot == 0 ? 0 : h + (slot - 1) * EPC;
            const int32_t left = nb - sb0;
            const int32_t cnt = slot == 0 ? (h < left ? h : left) : (left > EPC ? EPC : left);
            if (cnt <= 0) continue;
            alignas(16) T vals[EPC];
#pragma unroll
            for (int i = 0; i < EPC; ++i) {
                const int32_t sb = sb0 + i < TB ? sb0 + i : TB - 1;
                vals[i] = token_value<T>(s_tok[sb * kTokStride + tl]);
            }
            T *dst = out + e0 + sb0;
            if (cnt == EPC) {
                store16<true>(dst, *reinterpret_cast<const uint4 *>(vals));
            } else {
#pragma unroll
                for (int i = 0; i < EPC - 1; ++i)
                    if (i < cnt) dst[i] = vals[i];
            }
        }
        return;
    }
    for (int f = tid; f < kTT * CPR; f += kThreads) {
        const int32_t tl = f / CPR, q = f % CPR;
        const int64_t t = static_cast<int64_t>(t0) + tl;
        if (t >= p.P) continue;
        const int32_t sb0 = q * EPC;
        alignas(16) T vals[EPC];
#pragma unroll
        for (int i = 0; i < EPC; ++i) {
            const uint32_t tk = s_tok[(sb0 + i) * kTokStride + tl];
            vals[i] = token_value<T>(tk);
        }
        T *dst = out + t * p.B + b0 + sb0;
        if (p.aligned && b0 + sb0 + EPC <= p.B) {
            store16<true>(dst, *reinterpret_cast<const uint4 *>(vals));  // streamed once, never re-read: non-temporal
        } else {
#pragma unroll
            for (int i = 0; i < EPC; ++i)
                if (b0 + sb0 + i < p.B) dst[i] = vals[i];
        }
    }
}

// ------------------------------------------------------------------------------------------
// Tokens, (B,P) layout: one wave per sequence, 4 positions per lane per step, no transpose.
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kThreads) void k_tokenize_rows(const KParams p) {
    __shared__ __align__(16) uint8_t s_lut[256];
    stage_lut(p, s_lut);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int64_t b = static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6);
    if (b >= p.B) return;
    const TokenRule rule = make_rule(p, b, 1);  // window = this wave's sequence
    const uint32_t start = 0;
    const int32_t L = clamp_len(p, p.offsets[b + 1] - rule.off0);
    T *orow = static_cast<T *>(p.out) + b * p.P;
    const int32_t P = static_cast<int32_t>(p.P);
    for (int32_t tpos = 4 * lane; tpos < P; tpos += 256) {
        const uint32_t packed = resolve4(rule, s_lut, start, L, tpos);
        alignas(16) T vals[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) vals[i] = token_value<T>((packed >> (8 * i)) & 0xFFu);
        if (p.aligned && tpos + 4 <= P) {
            if constexpr (sizeof(T) == 1) {
                *reinterpret_cast<uint32_t *>(orow + tpos) = *reinterpret_cast<const uint32_t *>(vals);
            } else if constexpr (sizeof(T) == 2) {
                *reinterpret_cast<uint2 *>(orow + tpos) = *reinterpret_cast<const uint2 *>(vals);
            } else if constexpr (sizeof(T) == 4) {
                store16<true>(orow + tpos, *reinterpret_cast<const uint4 *>(vals));
            } else {
                store16<true>(orow + tpos, reinterpret_cast<const uint4 *>(vals)[0]);
                store16<true>(orow + tpos + 2, reinterpret_cast<const uint4 *>(vals)[1]);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (tpos + i < P) orow[tpos + i] = vals[i];
        }
    }
}


// ------------------------------------------------------------------------------------------
// Tokens, (B,P) layout, CHUNK form: the (B,P) matrix is a flat stream of B*P elements; one wave
// produces one naturally aligned 4-KiB chunk of it (chunk classes pinned to XCDs as in the one-hot
// chunk kernels).  A lane owns 16 output bytes = EPL = 16/sizeof(T) consecutive positions of one
// sequence per store (needs P % EPL == 0 and a 16-byte aligned base): one unaligned vector load of
// its EPL characters, EPL LUT lookups from a wave-private LDS table, one 16-byte store.
// ------------------------------------------------------------------------------------------
struct TParams {
    int8_t lut[256];
    const uint8_t *chars;
    const int64_t *offsets;
    uint8_t *out;
    int64_t total;    // output bytes
    int64_t nchunks;
    int64_t B, P;
    int32_t bos;
    uint32_t bos_id, at_len_id, fill_id;
    int32_t room;
    // one-hot (B,C,P) mode only:
    const uint8_t *mask;
    int32_t C;
    uint64_t one_bits;
    // index arithmetic in 16-byte PIECES (ppr = ceil(P / EPL) per row, the last one partial when P % EPL != 0),
    // without divisions: n / ppr for n < 2^31 is mulhi(n, magic) >> shift (pow2: n >> shift);
    // step_q / step_r = 64 / ppr and % ppr: row / piece advance between two stores of a lane
    uint32_t ppr, magic, shift, pow2, step_q, step_r;
    uint32_t a0e, pmod;  // RG: (out mod 16) / sizeof(T) and P mod EPL -- where in its 16-byte line a row starts
    int32_t wide_index;  // knob "wide_index": the 64-bit index arithmetic whatever the size (tests)
    uint32_t magic_c, shift_c, pow2_c;  // the same for / C (one-hot mode: row -> sequence, channel)
};


// N characters held as whole words (bytes are extracted only where they are consumed, so the loads
// stay in flight); alignment 1: gfx950 does unaligned vector loads in hardware.
template <int N>
struct __attribute__((packed, aligned(1))) UBytes {
    uint32_t w[N / 4];
    __device__ __forceinline__ uint32_t byte(int i) const { return (w[i >> 2] >> (8 * (i & 3))) & 0xFFu; }
    __device__ __forceinline__ void set_byte(int i, uint32_t v) {
        w[i >> 2] = (w[i >> 2] & ~(0xFFu << (8 * (i & 3)))) | ((v & 0xFFu) << (8 * (i & 3)));
    }
    __device__ __forceinline__ void clear() {
#pragma unroll
        for (int q = 0; q < N / 4; ++q) w[q] = 0;
    }
};
template <>
struct __attribute__((packed, aligned(1))) UBytes<2> {
    uint16_t h;
    __device__ __forceinline__ uint32_t byte(int i) const { return (h >> (8 * i)) & 0xFFu; }
    __device__ __forceinline__ void set_byte(int i, uint32_t v) {
        h = static_cast<uint16_t>((h & ~(0xFFu << (8 * i))) | ((v & 0xFFu) << (8 * i)));
    }
    __device__ __forceinline__ void clear() { h = 0; }
};

// HOT = false: (B,P) tokens of type T.  HOT = true: the "channels-first" one-hot (B,C,P) that conv nets
// consume (the reference gets it with einops.rearrange('length batch emb -> batch emb length') + .float(),
// bioseq/loaders.py:74): row = b*C + c of the flat (B*C, P) matrix holds (token(b,t) == c) -- the same
// character-row reader, compared against the row's channel instead of converted to a value.
// NCH = chunks per wave, software-pipelined: the offsets of chunk j + 2 and the characters of chunk j + 1 are in
// flight while chunk j is looked up and stored, so a wave pays the offsets -> characters -> store chain of
// dependent memory round trips once instead of once per chunk.  Measured: NCH = 4 is slower than 1 (VALU-bound
// kernel, lower occupancy), so 1 is what runs; 4 stays selectable for experiments.
template <typename T, bool HOT>
struct ChunkState {  // one 4-KiB chunk in flight: the lane's four 16-byte stores
    static constexpr int EPL = 16 / static_cast<int>(sizeof(T));
    int64_t lo;       // byte offset of the chunk in the output
    int64_t bc;       // first row of the chunk (wave-uniform)
    bool valid;       // chunk index < nchunks (wave-uniform)
    bool live[4];
    int32_t t0[4], L[4];
    uint32_t chan[4];
    int64_t row[4];   // RG: row of the lane's piece
    int32_t cnt[4];   // RG: elements of the piece (EPL, or fewer for the head / tail piece of a row)
    int64_t start[4], stop[4];
    UBytes<EPL> cw[4], mw[4];
    bool slow[4];
};

// RG ("ragged"): any P and any element-aligned output.  Pieces are counted per ROW, so a lane's elements always lie in
// one row, and they are cut at the 16-byte lines of the OUTPUT: slot 0 of a row is its head (the h elements up to the
// first 16-byte boundary, h = 0 .. EPL - 1 depending on where the row starts), slots 1 .. are whole aligned 16-byte
// pieces, the last one the tail; ceil(P / EPL) + 1 slots per row, at most one of them empty.  Whole pieces are the same
// aligned nt stores as in the plain form; heads and tails go out as 8 / 4 / 2 / 1-byte stores.  (A first version kept the
// pieces row-relative and stored them with unaligned 16-byte stores: 4-byte aligned dwordx4 stores cost 45-55 % --
// int32 65536 x 1001 63 us against 43 us aligned, profiles/r02/cliff_lab4.txt.)
template <typename T, bool NT, bool HOT, int NCH, bool RG = false>
__global__ __launch_bounds__(kThreads) void k_tokenize_chunks(const TParams p) {
    __shared__ __align__(16) uint8_t s_lut4[4][256];
    constexpr int SZ = static_cast<int>(sizeof(T));
    constexpr int EPL = 16 / SZ;  // elements (= characters) per lane per store
    using State = ChunkState<T, HOT>;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint8_t *lut = s_lut4[wave];
    {   // wave-private table: token VALUES (unmapped / >= 0x80 -> 0, the memset value of tokenize.h:427),
        // or raw ids with kNone for the one-hot mode
        uint32_t w = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int idx = lane * 4 + q;
            const int8_t v = p.lut[idx];
            const uint32_t e = (idx < 128 && v >= 0) ? static_cast<uint32_t>(v) : (HOT ? kNone : 0u);
            w |= e << (8 * q);
        }
        reinterpret_cast<uint32_t *>(lut)[lane] = w;
    }
    const int64_t nrows = HOT ? p.B * p.C : p.B;
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);  // scalar: chunk-level arithmetic stays on the SALU
    // chunks of this wave: class = blockIdx % 8 (pinned to the XCD), NCH consecutive slots of that class
    const int64_t slot0 = (static_cast<int64_t>(blockIdx.x >> 3) * 4 + wave_s) * NCH;
    const int64_t k0 = static_cast<int64_t>(blockIdx.x & 7u) + 8 * slot0;
    if (k0 >= p.nchunks) return;
    const int64_t total_chars = p.offsets[p.B];
    const uint32_t Pu = static_cast<uint32_t>(p.P), PPR = p.ppr;
    const bool has_mask = HOT && p.mask != nullptr;
    const bool small = nrows * int64_t(PPR) < (int64_t(1) << 31) && !p.wide_index;  // 32-bit piece indices: divide by reciprocal

    // stage A: (row, position) of the lane's four stores -- element e0 + u*EPS with e0 = lo/SZ + lane*EPL -- without a
    // per-lane division (the chunk's first element is wave-uniform, the lane's share adds < 1024 positions, the
    // stores advance by (step_q, step_r)) -- and the offsets of their sequences (8 independent loads).
    auto stage_a = [&](State &c, int64_t k) {
        c.valid = k < p.nchunks;
        if (!c.valid) return;
        c.lo = k * kChunk;  // !RG: chunks are relative to `out` (16-byte aligned)
        const int64_t g0 = k * (kChunk / 16);  // first piece of the chunk
        uint32_t tc;
        if (small) {
            const uint32_t q = fast_div(static_cast<uint32_t>(g0), p.magic, p.shift, p.pow2);
            c.bc = q;
            tc = static_cast<uint32_t>(g0) - q * PPR;
        } else {
            c.bc = g0 / PPR;
            tc = static_cast<uint32_t>(g0 - c.bc * PPR);
        }
        const uint32_t tl = tc + static_cast<uint32_t>(lane);  // < ppr + 64
        const uint32_t ql = fast_div(tl, p.magic, p.shift, p.pow2);
        int64_t bu = c.bc + ql;
        uint32_t tu = tl - ql * PPR;  // piece of the row
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            int64_t b = bu;  // row of the flat matrix
            c.t0[u] = static_cast<int32_t>(tu * EPL);
            c.row[u] = b;
            c.live[u] = b < nrows;
            if constexpr (RG) {
                // the row starts `sm` elements into a 16-byte line of the output: head = the elements up to the next line
                const uint32_t sm = (p.a0e + (static_cast<uint32_t>(b) & (EPL - 1)) * p.pmod) & (EPL - 1);
                const int32_t h = static_cast<int32_t>((EPL - sm) & (EPL - 1));
                const int32_t t0 = tu == 0 ? 0 : h + static_cast<int32_t>(tu - 1) * EPL;
                const int32_t left = static_cast<int32_t>(Pu) - t0;
                c.t0[u] = t0;
                c.cnt[u] = tu == 0 ? (h < left ? h : left) : (left > EPL ? EPL : left);
                c.live[u] = c.live[u] && c.cnt[u] > 0;
            }
            b = c.live[u] ? b : nrows - 1;
            c.chan[u] = 0;
            if constexpr (HOT) {  // row = sequence * C + channel
                int64_t seq;
                if (nrows < (int64_t(1) << 31) && !p.wide_index)  // wave-uniform
                    seq = fast_div(static_cast<uint32_t>(b), p.magic_c, p.shift_c, p.pow2_c);
                else
                    seq = b / p.C;
                c.chan[u] = static_cast<uint32_t>(b - seq * p.C);
                b = seq;
            }
            c.start[u] = p.offsets[b];
            c.stop[u] = p.offsets[b + 1];
            bu += p.step_q;
            tu += p.step_r;
            if (tu >= PPR) {
                tu -= PPR;
                bu += 1;
            }
        }
    };

    // stage B: the characters.  Loads are UNCONDITIONAL (lanes that must not touch their own address read the
    // first bytes of the window instead) so that all four are in flight together.  Addresses are 32-bit
    // offsets from a wave-uniform base: the rows of a chunk are consecutive sequences, their characters lie
    // within 2^31 bytes of the first one's (off0), so "is [a, a + EPL) inside the buffer" is one unsigned
    // compare of (rel - lo_b) against span, and the load takes the scalar-base + 32-bit-offset form.
    auto stage_b = [&](State &c) {
        if (!c.valid) return;
#pragma unroll
        for (int u = 0; u < 4; ++u) {  // lengths from the low words (a valid length is < 2^31; else clamped to `room`)
            const uint32_t len = static_cast<uint32_t>(c.stop[u]) - static_cast<uint32_t>(c.start[u]);
            c.L[u] = static_cast<int32_t>(len > static_cast<uint32_t>(p.room) ? static_cast<uint32_t>(p.room) : len);
        }
        int64_t seq0 = c.bc;
        if constexpr (HOT)
            seq0 = (nrows < (int64_t(1) << 31) && !p.wide_index) ? int64_t(fast_div(static_cast<uint32_t>(c.bc), p.magic_c, p.shift_c, p.pow2_c))
                                                : c.bc / p.C;
        const int64_t off0 = p.offsets[seq0];
        const int64_t lo_b64 = -off0, hi_b64 = total_chars - off0 - EPL;  // valid range of a vector's first byte, relative to off0
        const bool can_vec = hi_b64 >= lo_b64;                            // wave-uniform (the buffer holds >= EPL bytes)
        const int32_t lo_b = lo_b64 < INT32_MIN ? INT32_MIN : static_cast<int32_t>(lo_b64);
        const int32_t hi_b = hi_b64 > INT32_MAX ? INT32_MAX : (hi_b64 < lo_b ? lo_b : static_cast<int32_t>(hi_b64));
        const uint32_t span = static_cast<uint32_t>(hi_b) - static_cast<uint32_t>(lo_b);
        uint32_t uoff[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int32_t j0 = c.t0[u] - p.bos;
            const uint32_t rel = static_cast<uint32_t>(c.start[u]) - static_cast<uint32_t>(off0) + static_cast<uint32_t>(j0);
            const uint32_t d = rel - static_cast<uint32_t>(lo_b);  // offset from the lowest valid address
            const bool need = c.live[u] && j0 < c.L[u] && j0 + EPL > 0;
            const bool fast = can_vec && need && d <= span;
            c.slow[u] = need && !fast;
            uoff[u] = fast ? d : 0u;
        }
        if (can_vec) {
            const uint8_t *cbase = p.chars + (off0 + lo_b);  // wave-uniform, inside the buffer
#pragma unroll
            for (int u = 0; u < 4; ++u) c.cw[u] = *reinterpret_cast<const UBytes<EPL> *>(cbase + uoff[u]);
            if (has_mask) {
                const uint8_t *mbase = p.mask + (off0 + lo_b);
#pragma unroll
                for (int u = 0; u < 4; ++u) c.mw[u] = *reinterpret_cast<const UBytes<EPL> *>(mbase + uoff[u]);
            }
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) c.cw[u].clear();
        }
    };

    // stage C: the rare byte-wise edge reads, then LUT lookups packed 4 (or 2) per word with BOS / EOS / PAD folded in
    // as word masks, and the four 16-byte stores.
    constexpr int WB = EPL >= 4 ? 4 : EPL;  // characters per word
    const uint32_t ones = WB == 4 ? 0x01010101u : 0x0101u;
    const uint32_t fill_v = (!HOT && p.fill_id == kNone) ? 0u : p.fill_id;
    const uint32_t at_len_v = (!HOT && p.at_len_id == kNone) ? 0u : p.at_len_id;
    const uint32_t fill_w = fill_v * ones, at_len_w = at_len_v * ones;
    const T hot_one = static_cast<T>(p.one_bits);
    auto stage_c = [&](State &c) {
        if (!c.valid) return;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (c.slow[u]) {  // first / last bytes of the buffer: never read outside it
                const int32_t j0 = c.t0[u] - p.bos;
                c.cw[u].clear();
                if (has_mask) c.mw[u].clear();
#pragma unroll
                for (int i = 0; i < EPL; ++i)
                    if (j0 + i >= 0 && j0 + i < c.L[u]) {
                        c.cw[u].set_byte(i, p.chars[c.start[u] + j0 + i]);
                        if (has_mask) c.mw[u].set_byte(i, p.mask[c.start[u] + j0 + i]);
                    }
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (!c.live[u]) continue;
            const int32_t j0 = c.t0[u] - p.bos;
            alignas(16) T vals[EPL];
            uint32_t packed[EPL / WB];
#pragma unroll
            for (int q = 0; q < EPL / WB; ++q) {
                uint32_t w = 0;
#pragma unroll
                for (int i = 0; i < WB; ++i) {
                    uint32_t tk = lut[c.cw[u].byte(q * WB + i)];
                    if (has_mask && c.mw[u].byte(q * WB + i) == 0) tk = kNone;
                    w |= tk << (8 * i);
                }
                const int32_t jf = j0 + q * WB;  // character index of the word's first byte
                const int32_t nv = c.L[u] - jf;  // characters of the sequence left from there
                // Branch-free (nearly every wave holds a lane that straddles or lies beyond L): the first nvc bytes stay,
                // the rest is fill, byte nv (if it is one of this word's) is the token at position bos + L.
                const int32_t nvc = nv < 0 ? 0 : (nv > WB ? WB : nv);
                const uint32_t keep = static_cast<uint32_t>(uint64_t(1) << (8 * nvc)) - 1u;  // low nvc bytes (nvc = 4: all)
                w = (w & keep) | (fill_w & ~keep);
                const uint32_t at = static_cast<uint32_t>(nv) < static_cast<uint32_t>(WB) ? (0xFFu << (8 * nvc)) : 0u;
                w = (w & ~at) | (at_len_w & at);
                if (q == 0 && jf < 0) w = (w & ~0xFFu) | p.bos_id;  // position 0 with BOS (j0 >= -1: only the first word)
                packed[q] = w;
                if constexpr (HOT || SZ != 1) {
#pragma unroll
                    for (int i = 0; i < WB; ++i) {
                        const uint32_t tk = (w >> (8 * i)) & 0xFFu;
                        if constexpr (HOT)
                            vals[q * WB + i] = tk == c.chan[u] ? hot_one : T(0);
                        else
                            vals[q * WB + i] = id_as<T>(tk);
                    }
                }
            }
            uint4 o;
            if constexpr (!HOT && SZ == 1)  // 8-bit tokens: the packed words ARE the 16 output bytes
                o = uint4{packed[0], packed[1], packed[2], packed[3]};
            else
                o = *reinterpret_cast<const uint4 *>(vals);
            if constexpr (!RG) {
                store16<NT>(p.out + c.lo + u * 1024 + lane * 16, o);
            } else {
                uint8_t *dst = p.out + (c.row[u] * p.P + c.t0[u]) * SZ;
                // (skipping the partial stores when no lane of the wave holds a head / tail -- a ballot -- measured 2-7 %
                // slower: profiles/r02/cliff_lab5.txt vs cliff_lab6.txt)
                if (c.cnt[u] == EPL) store16<NT>(dst, o);  // a whole piece: 16-byte aligned by construction
                else store_head_bytes_var<SZ>(dst, o, static_cast<uint32_t>(c.cnt[u]) * SZ);
            }
        }
    };

    State st[NCH];
    stage_a(st[0], k0);
    if constexpr (NCH > 1) stage_a(st[1], k0 + 8);
    stage_b(st[0]);
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        if (j + 1 < NCH) stage_b(st[j + 1]);
        if (j + 2 < NCH) stage_a(st[j + 2], k0 + 8 * (j + 2));
        stage_c(st[j]);
    }
}

template <typename T, int TB>
bsq_status launch_tokenize_tile(KParams &k, hipStream_t s) {
    // automatic tile order: sequence-tile index fastest.  This kernel writes TB * sizeof(T) = 256..512-byte row segments;
    // with the XCD-aware order their neighbours in a row are written far apart in time and the (P,B) int32 / f32 matrix of
    // the cfg2 batch takes 62 us instead of 48 (profiles/r02/tokens_dtypes.txt); its character re-reads are small beside that.
    if (bsq_internal::tuning().tile_order == 0 && k.order != 4) k.order = 0;  // (4: chosen by the caller for unaligned rows)
    k.ntb = int32_t((k.B + TB - 1) / TB);
    const int64_t ntt = (k.P + kTT - 1) / kTT;
    const size_t smem = tile_fixed_bytes<TB>();
    hipLaunchKernelGGL((k_tokenize_tile<T, TB>), dim3(unsigned(tile_grid(k, ntt))), dim3(kThreads), smem, s, k);
    return check_launch("k_tokenize_tile");
}

template <typename T, bool HOT>
bsq_status launch_tokenize_chunks(const KParams &k, hipStream_t s) {
    TParams c;
    for (int i = 0; i < 256; ++i) c.lut[i] = k.lut[i];
    c.chars = k.chars;
    c.offsets = k.offsets;
    c.mask = HOT ? k.mask : nullptr;
    c.out = static_cast<uint8_t *>(k.out);
    c.B = k.B;
    c.P = k.P;
    c.C = k.C;
    c.one_bits = k.one_bits;
    constexpr uint32_t EPL = 16u / uint32_t(sizeof(T));
    const bool ragged = k.P % EPL != 0 || reinterpret_cast<uintptr_t>(k.out) % 16 != 0;
    c.total = k.B * k.P * int64_t(sizeof(T)) * (HOT ? k.C : 1);
    c.ppr = uint32_t((k.P + EPL - 1) / EPL) + (ragged ? 1u : 0u);  // ragged: + the head slot
    c.a0e = uint32_t(reinterpret_cast<uintptr_t>(k.out) % 16) / uint32_t(sizeof(T));
    c.pmod = uint32_t(k.P % EPL);
    c.wide_index = bsq_internal::tuning().wide_index;
    c.nchunks = (k.B * (HOT ? int64_t(k.C) : 1) * int64_t(c.ppr) + kChunk / 16 - 1) / (kChunk / 16);  // 256 pieces per wave
    c.bos = k.bos;
    c.bos_id = uint32_t(k.bos_id);
    c.fill_id = uint32_t(k.fill_id);
    c.at_len_id = k.eos ? uint32_t(k.eos_id) : c.fill_id;
    const int64_t room = k.P - k.bos - k.eos;
    c.room = int32_t(room < 0 ? 0 : room);
    div_constants(c.ppr, &c.magic, &c.shift, &c.pow2);
    div_constants(uint32_t(k.C > 0 ? k.C : 1), &c.magic_c, &c.shift_c, &c.pow2_c);
    c.step_q = 64u / c.ppr;
    c.step_r = 64u % c.ppr;
    // Chunks per wave: 1.  The software-pipelined 4-chunk form (round 2; not built any more) was 15-20 % SLOWER on cfg2 /
    // cfg5: the kernel is bound by its ~550 VALU instructions per chunk, not by memory latency, and four chunks
    // per wave cost occupancy (102 VGPRs).
    const int nch = 1;
    const int64_t groups = ((c.nchunks + 7) / 8 + int64_t(4) * nch - 1) / (int64_t(4) * nch);
    if (groups * 8 >= (int64_t(1) << 31)) return bsq_internal::set_error(BSQ_ERR_INVALID_ARG, "output too large");
    const dim3 grid(unsigned(groups * 8));
    const int padv = bsq_internal::tuning().tokenize_pad;  // unused dynamic LDS = occupancy cap (experiments)
    const size_t pad = padv > 0 ? size_t(padv) : 0;
    const bool nt = bsq_internal::nontemporal_stores();
    if (ragged) {
        if (nt) hipLaunchKernelGGL((k_tokenize_chunks<T, true, HOT, 1, true>), grid, dim3(kThreads), pad, s, c);
        else hipLaunchKernelGGL((k_tokenize_chunks<T, false, HOT, 1, true>), grid, dim3(kThreads), pad, s, c);
    } else {
        if (nt) hipLaunchKernelGGL((k_tokenize_chunks<T, true, HOT, 1>), grid, dim3(kThreads), pad, s, c);
        else hipLaunchKernelGGL((k_tokenize_chunks<T, false, HOT, 1>), grid, dim3(kThreads), pad, s, c);
    }
    return check_launch(HOT ? "k_tokenize_chunks<onehot bcl>" : "k_tokenize_chunks");
}

}  // namespace

namespace bsq_internal {

// the channels-first (B, C, P) one-hot through k_tokenize_chunks<T, HOT> (called by bsq_onehot_bcl_device, bsq_onehot.hip)
bsq_status launch_onehot_bcl_chunks(const bsq_desc *d, const uint8_t *chars, const int64_t *offsets, const uint8_t *mask_or_null, int64_t B,
                                    int64_t P, bsq_dtype t, void *out, hipStream_t s) {
    KParams k;
    const bsq_status st = fill_common(k, d, chars, offsets, mask_or_null, B, P, out);
    if (st != BSQ_OK) return st;
    k.one_bits = one_bits_of(t);
    switch (bsq_dtype_size(t)) {
    case 1: return launch_tokenize_chunks<uint8_t, true>(k, s);
    case 2: return launch_tokenize_chunks<uint16_t, true>(k, s);
    case 4: return launch_tokenize_chunks<uint32_t, true>(k, s);
    default: return launch_tokenize_chunks<uint64_t, true>(k, s);
    }
}

}  // namespace bsq_internal

extern "C" {

bsq_status bsq_tokenize_device(const bsq_desc *d, const uint8_t *chars, const int64_t *offsets, int64_t B,
                               int64_t P, int32_t batch_first, bsq_dtype t, void *out, void *hip_stream) {
    KParams k;
    bsq_status st = fill_common(k, d, chars, offsets, nullptr, B, P, out);
    if (st != BSQ_OK) return st;
    if (B == 0) return BSQ_OK;
    const size_t sz = bsq_dtype_size(t);
    if (sz == 0) return bsq_internal::set_error(BSQ_ERR_DTYPE, "bad bsq_dtype");
    if (k.C > 250 || B >= (int64_t(1) << 31) - 1024 || (!batch_first && P > kMaxTiledP))
        return bsq_tokenize_device_generic(d, chars, offsets, B, P, batch_first, t, out, hip_stream);
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    const uintptr_t addr = reinterpret_cast<uintptr_t>(out);
    // int8 (B,P): k_tokens_bp8 takes any padlen >= 128 and any alignment (its row-piece form when P % 16 != 0)
    if (batch_first && t == BSQ_I8 && bsq_internal::tuning().tokenize_path != 1 && bsq_internal::tuning().tokens8 != 1 &&
        bsq_internal::tokens_bp8_applicable(d, B, P, out) &&
        ((addr % 16 == 0 && P % 16 == 0) || bsq_internal::tuning().tokens8 != 2))  // knob 2: aligned shapes only (round-2 state)
        return bsq_internal::launch_tokens_bp8(d, chars, offsets, B, P, out, s);
    // chunk kernel: a lane's 16 output bytes lie inside one row (its row-piece form when P % (16 / sz) != 0 or the output
    // is not 16-byte aligned; knob "tokenize_path" 2: aligned shapes only, the rest falls to k_tokenize_rows as in round 1)
    if (batch_first && bsq_internal::tuning().tokenize_path != 1 && addr % sz == 0 &&
        ((addr % 16 == 0 && P % int64_t(16 / sz) == 0) || bsq_internal::tuning().tokenize_path != 2)) {
        switch (t) {
        case BSQ_I8: return launch_tokenize_chunks<int8_t, false>(k, s);
        case BSQ_I16: return launch_tokenize_chunks<int16_t, false>(k, s);
        case BSQ_I32: return launch_tokenize_chunks<int32_t, false>(k, s);
        case BSQ_U64: return launch_tokenize_chunks<uint64_t, false>(k, s);
        case BSQ_F32: return launch_tokenize_chunks<float, false>(k, s);
        case BSQ_F64: return launch_tokenize_chunks<double, false>(k, s);
        }
    }
    if (batch_first) {
        const size_t vec = sz * 4 > 16 ? 16 : sz * 4;  // widest store used by k_tokenize_rows
        k.aligned = (addr % vec == 0) && ((P * int64_t(sz)) % int64_t(vec) == 0);
        const unsigned grid = unsigned((B + 3) / 4);
#define BSQ_ROWS(T) hipLaunchKernelGGL((k_tokenize_rows<T>), dim3(grid), dim3(kThreads), 0, s, k)
        switch (t) {
        case BSQ_I8: BSQ_ROWS(int8_t); break;
        case BSQ_I16: BSQ_ROWS(int16_t); break;
        case BSQ_I32: BSQ_ROWS(int32_t); break;
        case BSQ_U64: BSQ_ROWS(uint64_t); break;
        case BSQ_F32: BSQ_ROWS(float); break;
        case BSQ_F64: BSQ_ROWS(double); break;
        }
#undef BSQ_ROWS
        return check_launch("k_tokenize_rows");
    }
    k.aligned = (addr % 16 == 0) && ((B * int64_t(sz)) % 16 == 0);
    k.vw = (!k.aligned && addr % sz == 0 && bsq_internal::tuning().tokenize_path != 2) ? 2 : 1;  // k_tokenize_tile: 2 = line-aligned slots
    // (P,B) with 16-byte aligned rows, any element type: register-transposed 256 x 64 tiles (bsq_tokens8.hip; knob tokens_pb8 = 1: never)
    if (bsq_internal::tuning().tokenize_path != 1 && bsq_internal::tokens_pb8_applicable(d, B, P, out, B, t))
        return bsq_internal::launch_tokens_pb8(d, chars, offsets, B, P, out, B, s, false, t);
    if (t == BSQ_I8 && bsq_internal::tuning().tokenize_path != 1) {  // int8 (P,B): the raw-token kernel in value mode
        const uint64_t al = uint64_t(addr) | uint64_t(B);  // every row starts at out + t * B
        k.vw = al % 16 == 0 ? 16 : (al % 8 == 0 ? 8 : (al % 4 == 0 ? 4 : 1));
        // tile order 5 (XCD-contiguous ranges of sequence tiles, their position tiles back to back): the 256-byte row
        // segments of neighbouring tiles meet in one L2 -- cfg5 43.0 -> 41.3 us, 100000 x 256 DNA 13.8 -> 12.7,
        // 65000 x 1001 29.1 -> 27.3, cfg2 24.7 -> 24.0 (profiles/r02/seqfirst_orders2.txt)
        if (bsq_internal::tuning().tile_order == 0) k.order = 5;
        const int rm = bsq_internal::tuning().raw_mode;
        // knob "raw_mode": 0 automatic, 1 the 256 x 64 tile, 4 the wide 1024 x 16 tile
        const bool wide_ok = P <= (int64_t(1) << 20);  // 1024 sequences x padlen in 32-bit window offsets
        if (wide_ok && rm == 4) {  // measurement only: 35 vs 24 us on cfg2 (profiles/r02/seqfirst_lab1.txt)
            k.ntb = int32_t((k.B + kWideTB - 1) / kWideTB);
            k.ntt = int32_t((P + kWideTT - 1) / kWideTT);
            if ((int64_t(k.ntb) + 8 * int64_t(k.group)) * int64_t(k.ntt) >= (int64_t(1) << 31)) k.order = 0;
            hipLaunchKernelGGL((k_tokens_raw<false, false, kWideTB, kWideTT>), dim3(unsigned(tile_grid(k, k.ntt))),
                               dim3(kThreads), 0, s, k);
            return check_launch("k_tokens_raw<value, wide>");
        }
        k.ntb = int32_t((k.B + kRawTB - 1) / kRawTB);
        const dim3 vgrid(unsigned(tile_grid(k, k.ntt)));
        if (vgrid.x <= 2048u)  // about one round of workgroups: the latency form
            hipLaunchKernelGGL((k_tokens_raw<false, false, kRawTB, kTT, true>), vgrid, dim3(kThreads), 0, s, k);
        else
            hipLaunchKernelGGL((k_tokens_raw<false, false>), vgrid, dim3(kThreads), 0, s, k);
        return check_launch("k_tokens_raw<value>");
    }
    // Sequences per tile (knob "tokenize_tb": 0 automatic, 64 / 128 / 256).  A tile writes TB * sz-byte row segments;
    // when the rows are not 64-byte aligned (B * sz % 64 != 0) neighbouring tiles share memory sectors, and longer
    // segments share fewer of them: 65000 x 1024 int32 99 -> 82 us, int16 83 -> 55 us with 256 sequences
    // (profiles/r02/tile_tb_lab.txt); aligned batches and small ones keep the short tiles (more workgroups).
    // With such rows the 2- / 4-byte types also take tile order 4 (every XCD walks its own contiguous range of sequence
    // tiles, so the sectors two tiles share are merged in ONE L2): int16 55 -> 42 us, int32 78 -> 69 us on 65000 x 1024
    // (profiles/r02/tile_tb_lab2.txt).  8-byte elements gain from neither (151-161 us whatever the tile).
    const bool shared_sectors = (B * int64_t(sz)) % 64 != 0 && B >= 16384 && sz < 8;
    const int tbk = shared_sectors && bsq_internal::tuning().tokenize_tb == 0 ? 256 : bsq_internal::tuning().tokenize_tb;
    if (shared_sectors && bsq_internal::tuning().tile_order == 0) k.order = 4;
#define BSQ_TILE(T, AUTO)                                                        \
    switch (tbk ? tbk : AUTO) {                                                  \
    case 64: return launch_tokenize_tile<T, 64>(k, s);                           \
    case 128: return launch_tokenize_tile<T, 128>(k, s);                         \
    default: return launch_tokenize_tile<T, 256>(k, s);                          \
    }
    switch (t) {
    case BSQ_I8: return launch_tokenize_tile<int8_t, 256>(k, s);
    case BSQ_I16: BSQ_TILE(int16_t, 128)
    case BSQ_I32: BSQ_TILE(int32_t, 64)
    case BSQ_U64: BSQ_TILE(uint64_t, 64)
    case BSQ_F32: BSQ_TILE(float, 64)
    case BSQ_F64: BSQ_TILE(double, 64)
    }
#undef BSQ_TILE
    return bsq_internal::set_error(BSQ_ERR_DTYPE, "bad bsq_dtype");
}

bsq_status bsq_augment_tokenize_device(const bsq_desc *d, uint8_t *chars, const int64_t *offsets, int64_t B, int64_t P,
                                       int32_t batch_first, bsq_dtype t, void *out, int32_t chain_len, double frac, uint64_t seed,
                                       void *hip_stream) {
    KParams k;
    bsq_status st = fill_common(k, d, chars, offsets, nullptr, B, P, out);
    if (st != BSQ_OK) return st;
    if (chain_len < 0) return bsq_internal::set_error(BSQ_ERR_INVALID_ARG, "bad augment arguments");
    // a token wave of an EARLIER fused launch gave up waiting for its rows' augmentation (its chunk is poisoned): sticky until
    // bsq_fused_status_clear() -- no later call succeeds silently on top of it
    if (bsq_internal::fused_failures() != 0)
        return bsq_internal::set_error(BSQ_ERR_FUSED_WAIT, "an earlier fused augmentation + token launch gave up waiting (output poisoned); see bsq_fused_status()");
    if (B == 0) return BSQ_OK;
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    // one launch where bsq_tokenize_device would take the fast form of k_tokens_bp8 (same conditions as there)
    if (chain_len > 0 && frac > 0.0 && chars && batch_first && t == BSQ_I8 && k.C <= 250 && B < (int64_t(1) << 31) - 1024 &&
        bsq_internal::tuning().tokenize_path != 1 && bsq_internal::tuning().tokens8 != 1 && bsq_internal::tokens_bp8_applicable(d, B, P, out)) {
        bsq_internal::FusedAugRequest fr{chars, chain_len, frac, seed};
        bool taken = false;
        st = bsq_internal::launch_tokens_bp8(d, chars, offsets, B, P, out, s, false, nullptr, &fr, &taken);
        if (st != BSQ_OK) return st;
        if (taken) return BSQ_OK;
    }
    // every other shape and layout (the (P,B) matrix fused the same way gained 2.6 %: round 3; dropped with its coherent loads in round 4)
    st = bsq_augment_device(chars, offsets, B, chain_len, frac, seed, hip_stream);
    if (st != BSQ_OK) return st;
    return bsq_tokenize_device(d, chars, offsets, B, P, batch_first, t, out, hip_stream);
}

bsq_status bsq_tokenize_device_multi(const bsq_desc *d, int32_t n, const bsq_batch *batches, int64_t P, int32_t batch_first, bsq_dtype t,
                                     void *hip_stream) {
    if (!d || n < 0 || (n > 0 && !batches) || P <= 0) return bsq_internal::set_error(BSQ_ERR_INVALID_ARG, "null pointer, n < 0 or padlen <= 0");
    if (bsq_dtype_size(t) == 0) return bsq_internal::set_error(BSQ_ERR_DTYPE, "bad bsq_dtype");
    for (int32_t i = 0; i < n; ++i)
        if (batches[i].B < 0 || (batches[i].B > 0 && (!batches[i].offsets || !batches[i].out)))
            return bsq_internal::set_error(BSQ_ERR_INVALID_ARG, "a batch with a null pointer or B < 0");
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    bsq_batch grp[8];
    int32_t i = 0;
    while (i < n) {  // groups of up to eight non-empty batches
        int32_t g = 0;
        while (i < n && g < 8) {
            if (batches[i].B > 0) grp[g++] = batches[i];
            ++i;
        }
        if (g == 0) break;
        bool taken = false;
        if (g > 1) {
            const bsq_status st = bsq_internal::launch_tokens_multi(d, g, grp, P, batch_first != 0, t, s, &taken);
            if (st != BSQ_OK) return st;
        }
        if (!taken)
            for (int32_t k = 0; k < g; ++k) {
                const bsq_status st = bsq_tokenize_device(d, grp[k].chars, grp[k].offsets, grp[k].B, P, batch_first, t, grp[k].out, hip_stream);
                if (st != BSQ_OK) return st;
            }
    }
    return BSQ_OK;
}

bsq_status bsq_augment_tokenize_device_multi(const bsq_desc *d, int32_t n, const bsq_batch *batches, int64_t P, int32_t batch_first, bsq_dtype t,
                                             int32_t chain_len, double frac, const uint64_t *seeds, void *hip_stream) {
    if (!d || n < 0 || (n > 0 && (!batches || !seeds)) || P <= 0 || chain_len < 0)
        return bsq_internal::set_error(BSQ_ERR_INVALID_ARG, "null pointer, n < 0, padlen <= 0 or chain_len < 0");
    if (bsq_dtype_size(t) == 0) return bsq_internal::set_error(BSQ_ERR_DTYPE, "bad bsq_dtype");
    for (int32_t i = 0; i < n; ++i)
        if (batches[i].B < 0 || (batches[i].B > 0 && (!batches[i].offsets || !batches[i].out || !batches[i].chars)))
            return bsq_internal::set_error(BSQ_ERR_INVALID_ARG, "a batch with a null pointer or B < 0");
    // Two launches per eight batches: every batch's augmentation (one concatenated grid), then every batch's tokens (one concatenated grid
    // where the fast kernel of the layout applies) -- nobody waits inside a kernel, no write-through, every character streamed once by plain
    // loads.  (Round 6 also built the augmentation of the NEXT group beside the tokens of THIS group in one launch, in front of the token
    // blocks and interleaved with them: the same 51 us per batch -- the augmentation's cost is its ~0.5 M random gathers from HBM per batch,
    // not a latency that could hide: profiles/r06/cfg5aug_multi.txt, scripts/probes/augment_next_tokens_pipeline.inc.)
    bsq_status st = bsq_augment_device_multi(n, batches, chain_len, frac, seeds, hip_stream);
    if (st != BSQ_OK) return st;
    return bsq_tokenize_device_multi(d, n, batches, P, batch_first, t, hip_stream);
}

// Name of the kernel(s) bsq_tokenize_device (augment = 0) / bsq_augment_tokenize_device (augment = chain_len > 0) launch for this shape, for a
// 16-byte aligned contiguous output: the dispatch above, as a function of the shape alone (profiling / bench labels).
const char *bsq_tokenize_kernel_name(const bsq_desc *d, int64_t B, int64_t P, int32_t batch_first, bsq_dtype t, int32_t augment) {
    if (!d || B <= 0 || P <= 0) return "";
    const size_t sz = bsq_dtype_size(t);
    if (sz == 0) return "";
    const int C = bsq_alphabet_size(d);
    const auto &tn = bsq_internal::tuning();
    if (C > 250 || B >= (int64_t(1) << 31) - 1024 || (!batch_first && P > kMaxTiledP)) return augment ? "k_augment_groups+k_tokenize_generic" : "k_tokenize_generic";
    if (batch_first) {
        const bool bp8 = t == BSQ_I8 && tn.tokenize_path != 1 && tn.tokens8 != 1 && bsq_internal::tokens_bp8_applicable(d, B, P, nullptr) &&
                         (P % 16 == 0 || tn.tokens8 != 2);
        if (bp8) {
            const bool fast = bsq_internal::tokens_bp8_fast_form(d, B, P, true);
            if (augment && fast && tn.augment_fused != 1)
                return bsq_internal::tokens_bp8_nowait_form(bsq_internal::tokens_bp8_chunks(B, P))
                           ? "k_augment_tokens_nowait(k_augment_groups || k_tokens_bp8_fast)+k_patch_tokens"
                           : "k_augment_tokens_fused(k_augment_groups -> k_tokens_bp8_fast)";
            if (augment) return fast ? "k_augment_groups+k_tokens_bp8_fast" : "k_augment_groups+k_tokens_bp8";
            return fast ? "k_tokens_bp8_fast" : "k_tokens_bp8";
        }
        const bool chunks = tn.tokenize_path != 1 && (P % int64_t(16 / sz) == 0 || tn.tokenize_path != 2);
        if (augment) return chunks ? "k_augment_groups+k_tokenize_chunks" : "k_augment_groups+k_tokenize_rows";
        return chunks ? "k_tokenize_chunks" : "k_tokenize_rows";
    }
    const char *name;
    alignas(64) static const char aligned_dummy[64] = {};
    if (tn.tokenize_path != 1 && bsq_internal::tokens_pb8_applicable(d, B, P, aligned_dummy, B, t)) name = "k_tokens_pb8_fast";
    else if (t == BSQ_I8 && tn.tokenize_path != 1) name = "k_tokens_raw<value>";
    else name = "k_tokenize_tile";
    if (!augment) return name;
    return name == std::string("k_tokens_pb8_fast") ? "k_augment_groups+k_tokens_pb8_fast"
           : (name == std::string("k_tokens_raw<value>") ? "k_augment_groups+k_tokens_raw<value>" : "k_augment_groups+k_tokenize_tile");
}

bsq_status bsq_fused_status(uint32_t *failures) {
    const uint32_t n = bsq_internal::fused_failures();
    if (failures) *failures = n;
    if (n != 0) return bsq_internal::set_error(BSQ_ERR_FUSED_WAIT, "a fused augmentation + token launch gave up waiting for its rows' augmentation (output poisoned)");
    return BSQ_OK;
}
void bsq_fused_status_clear(void) { bsq_internal::fused_failures_clear(); }

// The (P, B) token matrix as a COLUMN BLOCK of a wider (P, row_seqs) matrix: `out` points at element (0, b0) of it.  The block form of
// batch_tokenize's default layout -- pieces of a host batch (staged batches), a rank's shard stored into another GPU's matrix.
// 1-, 2- and 8-byte types of alphabets with ids < 251 run through k_tokens_pb8_fast at the speed of the whole matrix; the rest
// through the generic kernel (correct, slow: callers that care split only the fast types).
bsq_status bsq_tokenize_block_device(const bsq_desc *d, const uint8_t *chars, const int64_t *offsets, int64_t B, int64_t P,
                                     bsq_dtype t, void *out, int64_t row_seqs, void *hip_stream) {
    if (row_seqs < B) return bsq_internal::set_error(BSQ_ERR_INVALID_ARG, "row_seqs < B");
    if (row_seqs == B) return bsq_tokenize_device(d, chars, offsets, B, P, 0, t, out, hip_stream);
    if (!d || B < 0 || P <= 0 || (B > 0 && (!offsets || !out)))
        return bsq_internal::set_error(BSQ_ERR_INVALID_ARG, "null pointer, B < 0 or padlen <= 0");
    if (B == 0) return BSQ_OK;
    const size_t sz = bsq_dtype_size(t);
    if (sz == 0) return bsq_internal::set_error(BSQ_ERR_DTYPE, "bad bsq_dtype");
    if (reinterpret_cast<uintptr_t>(out) % sz) return bsq_internal::set_error(BSQ_ERR_INVALID_ARG, "output is not aligned to its element size");
    if (row_seqs < (int64_t(1) << 31) && bsq_internal::tokens_pb8_applicable(d, B, P, out, row_seqs, t))
        return bsq_internal::launch_tokens_pb8(d, chars, offsets, B, P, out, row_seqs, static_cast<hipStream_t>(hip_stream), false, t);
    return bsq_internal::tokenize_generic_block(d, chars, offsets, B, P, 0, t, out, row_seqs, hip_stream);
}

}  // extern "C"
