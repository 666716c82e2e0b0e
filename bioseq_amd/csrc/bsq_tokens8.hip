// k_tokens_bp8: the (B,P) int8 token matrix of batch_tokenize(batch_first=True, destchar='b') -- BASELINE
// configs 2 and 5 -- for gfx950 (MI355X / CDNA4) only.  Semantics: /root/reference/src/tokenize.h:454-479.
//
// Same streaming shape as the other chunk kernels (one wave = one naturally aligned 4-KiB chunk of the flat
// output, chunk classes pinned to XCDs, one unaligned 16-byte character load and one 16-byte nt store per lane
// and KiB), but with the per-character work cut to what an 8-bit stream can afford (the generic
// k_tokenize_chunks spends ~30 VALU + 4 LDS instructions per 4 characters and runs at 0.60-0.69 of the HBM
// roof on these shapes, bound by instruction issue and by the serial ds_read_u8 -> pack chains):
//   * alphabet lookup out of REGISTERS: every alphabet but BYTES maps letters only and both cases alike, so
//     the table is 32 bytes indexed by (c & 31) -- eight dwords in VGPRs, looked up four characters at a time
//     with v_perm_b32 (4 + 2 + 1 perms over bits 2:0, 3, 4 of each byte).  Non-letters (rare) are detected on
//     the packed words and fixed on a wave-uniform slow path.  LK = 0 keeps the LDS byte table instead.
//   * BOS / EOS / PAD rules from a per-wave LDS table: the 16 bytes of a lane start `d` = L - j0 characters
//     before the end of its sequence; entry clamp(d, -1, 16) holds the 16-byte keep mask and the 16 constant
//     bytes (EOS at byte d, PAD behind it), so the rules cost two ds_read_b128 per store and ONE v_and_or per
//     word instead of ~10 VALU per word.
//   * the offsets of the <= 63 sequences a chunk touches are loaded ONCE per wave (lane i: sequence bc + i) and
//     handed to the lanes with ds_bpermute, instead of eight 8-byte loads per lane.
#include <hip/hip_runtime.h>

#include <climits>
#include <cstdint>
#include <cstring>
#include <mutex>

#include "bsq.h"
#include "bsq_augment_dev.h"
#include "bsq_device.h"
#include "bsq_diag.h"
#include "bsq_internal.h"

namespace {

using namespace bsq_dev;

struct T8Params {
    int8_t lut[256];   // LK == 0 only
    uint32_t tab[8];   // LK == 1: token value of letter (c & 31), unmapped = 0 (the memset value of tokenize.h:427)
    const uint8_t *chars;
    const uint8_t *mask;   // MASK: one byte per character, 0 = the position gets none_v (one-hot: an all-zero row)
    const int64_t *offsets;
    uint8_t *out;
    int64_t total;     // output bytes = B * P
    int64_t nchunks;
    int64_t B;
    double inv_ppr;    // div_by() constant of ppr (outputs of 2^31 pieces and more)
    uint32_t P, ppr;   // padlen; 16-byte PIECES per row = ceil(P / 16) (the last one is partial when P % 16 != 0)
    uint32_t magic, shift, pow2;  // fast_div() constants of ppr
    int32_t bos, room;
    uint32_t bos_id, at_len_v, fill_v;  // token VALUES at position 0, bos + L and beyond (0 where the reference leaves the memset)
    int32_t abl;       // ablation experiments (diagnostic builds of the kernel only)
    int32_t wide_index;  // knob "wide_index": take the 64-bit chunk arithmetic whatever the size (tests: that path otherwise
                         // needs > 32 GB of output)
    uint32_t none_v;   // value of an unmapped character: 0 (token VALUES, tokenize.h:427) or 0xFF (raw ids for the one-hot expansion)
};

typedef uint32_t u32x4u __attribute__((ext_vector_type(4), aligned(1)));

__device__ __forceinline__ uint32_t lookup4_perm(uint32_t cw, const uint32_t (&T)[8]) {
    const uint32_t sel = cw & 0x07070707u;
    const uint32_t r0 = __builtin_amdgcn_perm(T[1], T[0], sel);
    const uint32_t r1 = __builtin_amdgcn_perm(T[3], T[2], sel);
    const uint32_t r2 = __builtin_amdgcn_perm(T[5], T[4], sel);
    const uint32_t r3 = __builtin_amdgcn_perm(T[7], T[6], sel);
    const uint32_t s3 = ((cw >> 1) & 0x04040404u) | 0x03020100u;  // byte i: i + 4 * bit 3 of character i
    const uint32_t lo = __builtin_amdgcn_perm(r1, r0, s3);
    const uint32_t hi = __builtin_amdgcn_perm(r3, r2, s3);
    const uint32_t s4 = ((cw >> 2) & 0x04040404u) | 0x03020100u;  // ... bit 4
    return __builtin_amdgcn_perm(hi, lo, s4);
}

// The same lookup with the table in VECTOR registers and the selector constant 0x03020100 in one (round 3): gfx9 VOP3
// encodings take neither a literal nor two scalar operands, so with the table in SGPRs every v_perm_b32 above costs a
// v_mov_b32 of one table word first (4 per word of four tokens) and every selector a separate v_and + v_or: 18 vector
// instructions per word.  Here 1 v_and + 4 v_perm + (v_lshrrev + v_and_or) x 2 + 3 v_perm = 12.
__device__ __forceinline__ uint32_t lookup4_perm_v(uint32_t cw, const uint32_t (&Tv)[8], uint32_t k3210) {
    const uint32_t sel = cw & 0x07070707u;
    const uint32_t r0 = __builtin_amdgcn_perm(Tv[1], Tv[0], sel);
    const uint32_t r1 = __builtin_amdgcn_perm(Tv[3], Tv[2], sel);
    const uint32_t r2 = __builtin_amdgcn_perm(Tv[5], Tv[4], sel);
    const uint32_t r3 = __builtin_amdgcn_perm(Tv[7], Tv[6], sel);
    const uint32_t s3 = ((cw >> 1) & 0x04040404u) | k3210;  // byte i: i + 4 * bit 3 of character i
    const uint32_t lo = __builtin_amdgcn_perm(r1, r0, s3);
    const uint32_t hi = __builtin_amdgcn_perm(r3, r2, s3);
    const uint32_t s4 = ((cw >> 2) & 0x04040404u) | k3210;  // ... bit 4
    return __builtin_amdgcn_perm(hi, lo, s4);
}

// 0xFF in every byte of cw that is NOT a letter position (0x40..0x7F): those bytes are unmapped.
__device__ __forceinline__ uint32_t nonletter_mask(uint32_t cw) {
    const uint32_t x = (cw ^ 0x40404040u) & 0xC0C0C0C0u;
    const uint32_t f = ((x >> 6) | (x >> 7)) & 0x01010101u;
    return (f << 8) - f;  // f * 0xFF
}

// ABL (diagnostic instantiations only): 0 the kernel; 1 no alphabet lookup (characters stored as they are);
// 2 no character loads; 3 no offsets loads either (synthetic sequence spans); 4 stores only; 5 character loads at
// synthetic addresses that do not depend on the offsets.
struct OffStage {   // stage A of one chunk: where it lies, the offsets of its rows in flight
    bool valid;     // wave-uniform
    int64_t lo, bc; // byte offset of the chunk in the output (RG: unused), its first row
    uint32_t tc;    // PIECE index inside row bc of the chunk's first piece
    int64_t o0, o1; // lane i: offsets[bc + i], offsets[bc + i + 1]
};
struct CharStage {  // stage B: the lane's four 16-byte character vectors in flight + what stage C needs
    bool valid;
    int64_t lo, bc;
    uint32_t tc;
    u32x4u cw[4];
    u32x4u mw[4];   // MASK: the mask bytes of the same characters
    int32_t j0[4], L[4];
    uint32_t q[4];  // RG: row of the lane's piece relative to bc
    bool live[4], slow[4];
};

// RG ("ragged"): any padlen >= 128 and any output alignment.  The unit is still 256 consecutive 16-byte pieces per
// wave, but pieces are counted per ROW (ceil(P / 16) of them, the last one partial), so that a lane's bytes always lie
// inside one row: piece m of row b goes to out + b * P + 16 m with an unaligned 16-byte store, the partial piece with
// 8 / 4 / 2 / 1-byte stores.  With P % 16 == 0 and a 16-byte aligned output this is the same mapping as RG = false.

// MASK (raw-id mode of the channels-first one-hot only): p.mask holds one byte per character; a character whose mask byte
// is 0 gives none_v (tokenize.h:346-348: the masked position's one-hot row stays zero).  Four more 16-byte loads per
// lane, one zero-byte test per word.
template <bool NT, int LK, int ABL, bool RG = false, bool MASK = false>
__global__ __launch_bounds__(kThreads) void k_tokens_bp8(const T8Params p) {
    __shared__ __align__(16) uint4 s_rule[4][2][18];
    __shared__ __align__(16) uint8_t s_lut4[LK == 0 ? 4 : 1][256];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);

    // chunk of this wave: class = blockIdx % 8 (pinned to the XCD the block lands on)
    const int64_t k0 = static_cast<int64_t>(blockIdx.x & 7u) + 8 * (static_cast<int64_t>(blockIdx.x >> 3) * 4 + wave_s);
    if (k0 >= p.nchunks) return;

    if constexpr (ABL == 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (k0 * kChunk + u * 1024 + lane * 16 < p.total)
                store16<NT>(p.out + k0 * kChunk + u * 1024 + lane * 16, uint4{p.fill_v, 0, 0, 0});
        return;
    }

    // ---- per-wave tables ----
    const uint32_t fill_w = p.fill_v * 0x01010101u, at_w = p.at_len_v * 0x01010101u;
    if (lane < 18) {  // rule entry `lane`: n = lane - 1 bytes kept, byte n (if any) = token at bos + L, the rest fill
        const int n = lane - 1;
        uint32_t keep[4], cst[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int nv = n - 4 * q;
            const int nvc = nv < 0 ? 0 : (nv > 4 ? 4 : nv);
            const uint32_t kq = nvc == 4 ? 0xFFFFFFFFu : ((1u << (8 * nvc)) - 1u);
            const uint32_t at = (nv >= 0 && nv < 4) ? (0xFFu << (8 * nv)) : 0u;
            keep[q] = kq;
            cst[q] = (fill_w & ~kq & ~at) | (at_w & at);
        }
        s_rule[wave][0][lane] = uint4{keep[0], keep[1], keep[2], keep[3]};
        s_rule[wave][1][lane] = uint4{cst[0], cst[1], cst[2], cst[3]};
    }
    uint32_t T[8];
    if constexpr (LK == 1) {
#pragma unroll
        for (int i = 0; i < 8; ++i) T[i] = p.tab[i];
    } else {  // wave-private byte table of token VALUES (unmapped / >= 0x80 -> 0)
        uint32_t w = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int idx = lane * 4 + q;
            const int8_t v = p.lut[idx];
            w |= ((idx < 128 && v >= 0) ? static_cast<uint32_t>(v) : p.none_v) << (8 * q);
        }
        reinterpret_cast<uint32_t *>(s_lut4[wave])[lane] = w;
    }
    const uint8_t *lut = s_lut4[LK == 0 ? wave : 0];
    // the tables are wave-private: LDS operations of one wave execute in order, only the compiler must not reorder them
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    const bool small = p.nchunks < (int64_t(1) << 23) && !p.wide_index;  // piece indices below 2^31
    const uint32_t P = p.P, PPR = p.ppr;
    auto div_p = [&](uint32_t n) { return fast_div(n, p.magic, p.shift, p.pow2); };
    constexpr bool kLoadOffsets = ABL != 3 && ABL != 5;
    constexpr bool kLoadChars = ABL < 2 || ABL >= 5;
    const int64_t total_chars = kLoadOffsets ? p.offsets[p.B] : int64_t(1) << 24;

    // ---- stage A: rows of the chunk, their offsets (ONE coalesced pair of loads per wave) ----
    auto stage_a = [&](int64_t k) {
        OffStage a;
        a.valid = k < p.nchunks;  // wave-uniform
        a.o0 = a.o1 = 0;
        a.lo = a.bc = 0;
        a.tc = 0;
        if (!a.valid) return a;
        a.lo = k * kChunk;
        const int64_t g0 = k * (kChunk / 16);  // first piece of the chunk
        if (small) {
            const uint32_t q = div_p(static_cast<uint32_t>(g0));
            a.bc = q;
            a.tc = static_cast<uint32_t>(g0) - q * PPR;
        } else {
            int64_t rem;
            a.bc = div_by(g0, PPR, p.inv_ppr, &rem);
            a.tc = static_cast<uint32_t>(rem);
        }
        const uint32_t nr = div_p(a.tc + (kChunk / 16 - 1));  // rows bc .. bc + nr intersect the chunk (nr <= 32: P >= 128)
        if (kLoadOffsets && static_cast<uint32_t>(lane) <= nr) {
            const int64_t i0 = a.bc + lane, i1 = i0 + 1;
            a.o0 = p.offsets[i0 < p.B ? i0 : p.B];
            a.o1 = p.offsets[i1 < p.B ? i1 : p.B];
        }
        return a;
    };

    // ---- stage B: the characters (4 unconditional unaligned 16-byte loads, all in flight together).  Lanes that
    // must not touch their own address read the first bytes of the window instead.  Addresses are 32-bit offsets
    // from a wave-uniform base: the rows of a chunk are consecutive sequences, their characters lie within 2^31
    // bytes of the first one's (off0), so "is [a, a + 16) inside the buffer" is one unsigned compare. ----
    auto stage_b = [&](const OffStage &a) {
        CharStage b;
        b.valid = a.valid;
        b.lo = a.lo;
        b.bc = a.bc;
        b.tc = a.tc;
        if (!a.valid) return b;
        int64_t off0;
        uint32_t rel;
        int32_t Lr;
        if constexpr (kLoadOffsets) {
            off0 = (static_cast<int64_t>(__builtin_amdgcn_readfirstlane(static_cast<int32_t>(a.o0 >> 32))) << 32) |
                   static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int32_t>(a.o0)));
            rel = static_cast<uint32_t>(a.o0) - static_cast<uint32_t>(off0);
            const uint32_t len = static_cast<uint32_t>(a.o1) - static_cast<uint32_t>(a.o0);
            Lr = static_cast<int32_t>(len > static_cast<uint32_t>(p.room) ? static_cast<uint32_t>(p.room) : len);
        } else {
            off0 = (a.lo >> 1) & ((int64_t(1) << 23) - 1);  // synthetic spans: ~half of the positions hold characters
            rel = lane * (P / 2) + 3;
            Lr = static_cast<int32_t>(P / 2) < p.room ? static_cast<int32_t>(P / 2) : p.room;
        }
        const int64_t lo_b64 = -off0, hi_b64 = total_chars - off0 - 16;  // valid range of a vector's first byte, relative to off0
        const bool can_vec = hi_b64 >= lo_b64;                           // wave-uniform (the buffer holds >= 16 bytes)
        const int32_t lo_b = lo_b64 < INT32_MIN ? INT32_MIN : static_cast<int32_t>(lo_b64);
        const int32_t hi_b = hi_b64 > INT32_MAX ? INT32_MAX : (hi_b64 < lo_b ? lo_b : static_cast<int32_t>(hi_b64));
        const uint32_t span = static_cast<uint32_t>(hi_b) - static_cast<uint32_t>(lo_b);
        const int64_t rows_left64 = p.B - a.bc;
        const uint32_t rows_left = rows_left64 > 64 ? 64u : static_cast<uint32_t>(rows_left64);
        uint32_t uoff[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t tl = a.tc + static_cast<uint32_t>(u * 64 + lane);
            const uint32_t q = div_p(tl);
            const int32_t t0 = static_cast<int32_t>((tl - q * PPR) << 4);
            b.q[u] = q;
            b.live[u] = q < rows_left;
            const uint32_t rs = static_cast<uint32_t>(__builtin_amdgcn_ds_bpermute(static_cast<int>(q << 2), static_cast<int>(rel)));
            b.L[u] = __builtin_amdgcn_ds_bpermute(static_cast<int>(q << 2), Lr);
            const int32_t j0 = t0 - p.bos;  // >= -1
            b.j0[u] = j0;
            const bool need = b.live[u] && j0 < b.L[u];
            const uint32_t d = rs + static_cast<uint32_t>(j0) - static_cast<uint32_t>(lo_b);  // offset from the lowest valid address
            const bool fast = can_vec && need && d <= span;
            b.slow[u] = need && !fast;
            uoff[u] = fast ? d : 0u;
        }
        if (kLoadChars && can_vec) {
            const uint8_t *cbase = p.chars + (off0 + lo_b);  // wave-uniform, inside the buffer
#pragma unroll
            for (int u = 0; u < 4; ++u) b.cw[u] = *reinterpret_cast<const u32x4u *>(cbase + uoff[u]);
            if constexpr (MASK) {
                const uint8_t *mbase = p.mask + (off0 + lo_b);
#pragma unroll
                for (int u = 0; u < 4; ++u) b.mw[u] = *reinterpret_cast<const u32x4u *>(mbase + uoff[u]);
            }
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                b.cw[u] = u32x4u{0x41434447u, 0x61636474u, 0x4B4C4D4Eu, 0x50515253u};
                if constexpr (MASK) b.mw[u] = u32x4u{~0u, ~0u, ~0u, ~0u};
            }
        }
        return b;
    };

    // ---- stage C: lookups, rules, stores ----
    auto stage_c = [&](CharStage &b) {
        if (!b.valid) return;

#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (kLoadChars && ABL != 5 && b.slow[u]) {  // first / last bytes of the buffer: never read outside it
                const uint32_t tl = b.tc + static_cast<uint32_t>(u * 64 + lane);
                const int64_t row = b.bc + div_p(tl);
                const int64_t start = p.offsets[row];
                uint32_t w[4] = {0, 0, 0, 0}, mk[4] = {~0u, ~0u, ~0u, ~0u};
#pragma unroll 1
                for (int i = 0; i < 16; ++i)
                    if (b.j0[u] + i >= 0 && b.j0[u] + i < b.L[u]) {
                        w[i >> 2] |= static_cast<uint32_t>(p.chars[start + b.j0[u] + i]) << (8 * (i & 3));
                        if constexpr (MASK)
                            if (p.mask[start + b.j0[u] + i] == 0) mk[i >> 2] &= ~(0xFFu << (8 * (i & 3)));
                    }
                b.cw[u] = u32x4u{w[0], w[1], w[2], w[3]};
                if constexpr (MASK) b.mw[u] = u32x4u{mk[0], mk[1], mk[2], mk[3]};
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int32_t dd = b.L[u] - b.j0[u];
            const int32_t idx = (dd < -1 ? -1 : (dd > 16 ? 16 : dd)) + 1;
            const uint4 keep = s_rule[wave][0][idx];
            const uint4 cst = s_rule[wave][1][idx];
            uint32_t w[4];
            const uint32_t in[4] = {b.cw[u].x, b.cw[u].y, b.cw[u].z, b.cw[u].w};
            if constexpr (ABL == 1) {
#pragma unroll
                for (int q = 0; q < 4; ++q) w[q] = in[q];
            } else if constexpr (LK == 1) {
                uint32_t bad = 0;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    w[q] = lookup4_perm(in[q], T);
                    bad |= (in[q] ^ 0x40404040u) & 0xC0C0C0C0u;
                }
                if (__builtin_amdgcn_ballot_w64(bad != 0) != 0) {  // some lane holds a non-letter: exact masks (rare)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const uint32_t m = nonletter_mask(in[q]);
                        w[q] = (w[q] & ~m) | (m & (p.none_v * 0x01010101u));
                    }
                }
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const uint32_t x = in[q];
                    w[q] = static_cast<uint32_t>(lut[x & 0xFFu]) | (static_cast<uint32_t>(lut[(x >> 8) & 0xFFu]) << 8) |
                           (static_cast<uint32_t>(lut[(x >> 16) & 0xFFu]) << 16) | (static_cast<uint32_t>(lut[x >> 24]) << 24);
                }
            }
            if constexpr (MASK) {  // masked characters -> none_v (before the position rules: those overwrite what lies behind L)
                const uint32_t mi[4] = {b.mw[u].x, b.mw[u].y, b.mw[u].z, b.mw[u].w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    uint32_t z = (mi[q] & 0x7F7F7F7Fu) + 0x7F7F7F7Fu;  // exact zero-byte detection
                    z = ~(z | mi[q] | 0x7F7F7F7Fu);                     // 0x80 in every byte of mi that is zero
                    const uint32_t zm = (z >> 7) * 0xFFu;
                    w[q] = (w[q] & ~zm) | ((p.none_v * 0x01010101u) & zm);
                }
            }
            uint4 o;
            o.x = (w[0] & keep.x) | cst.x;
            o.y = (w[1] & keep.y) | cst.y;
            o.z = (w[2] & keep.z) | cst.z;
            o.w = (w[3] & keep.w) | cst.w;
            if (b.j0[u] < 0) o.x = (o.x & ~0xFFu) | p.bos_id;  // position 0 with BOS
            if constexpr (!RG) {
                if (b.live[u]) store16<NT>(p.out + b.lo + u * 1024 + lane * 16, o);
            } else {
                const int32_t t0 = b.j0[u] + p.bos;
                uint8_t *dst = p.out + (b.bc + b.q[u]) * static_cast<int64_t>(P) + t0;
                const uint32_t nb = P - static_cast<uint32_t>(t0);  // bytes of the row from this piece on
                if (b.live[u] && nb >= 16) store16_unaligned<NT>(dst, o);
                const uint32_t r = P & 15u;  // wave-uniform: the partial piece of every row holds r bytes
                if (r != 0) store_head_bytes(dst, o, r, b.live[u] && nb < 16);
            }
        }
    };

    const OffStage a = stage_a(k0);
    CharStage b = stage_b(a);
    stage_c(b);
}

// ---------------------------------------------------------------------------------------------------------------------
// k_tokens_bp8_fast: the aligned, unmasked, register-table case of k_tokens_bp8 (cfg2, cfg5 and every other foldable
// alphabet with padlen % 16 == 0 and < 2^23 chunks) with the wave's CRITICAL PATH cut down.  Same lane mapping, same
// loads and stores, same results; what changed (round 3, profiles/r03/mix_lab1.txt: a plain read + write stream of this
// shape with two dependent load steps takes 15.7 / 30.3 us on cfg2 / cfg5, k_tokens_bp8 17.0-17.4 / 31.8-33):
//   * the scalars the first loads depend on are KERNEL ARGUMENTS IN SGPRs at wave start (14 dwords, gfx950 kernarg
//     preload: -mllvm -amdgpu-kernarg-preload-count=14 in build.py) instead of five serialised rounds of s_load +
//     s_waitcnt out of a 400-byte by-value struct;
//   * the offsets loads are issued FIRST; the per-wave rule table (host-built, two vector loads out of the kernel
//     arguments + 2 ds_write_b128), the register alphabet table (one s_load_dwordx8) and offsets[B] arrive while they are in flight;
//   * one division constant pair for every row width (powers of two as magic = 2^(32-s), shift 0: no branch per divide).
struct T8Tab {
    uint32_t t[8];
};
// The BOS / EOS / PAD rule table of a launch (entry n + 1: n bytes kept, byte n = the token at bos + L, the rest fill), built
// on the HOST and handed over in the kernel-argument segment: every wave copies its 576 bytes into LDS with two vector loads
// instead of rebuilding them with ~45 vector instructions (of ~490 per wave).
struct T8Rules {
    uint4 keep[18];
    uint4 cst[18];
};
// (the body as a device function of a VIRTUAL block index: the fused augmentation + token launch below runs it behind its
//  augmentation blocks; FLAGS: wait for the augmentation of the chunk's rows first, see k_augment_tokens_fused)
struct FusedWait {            // what a token wave of the fused launch needs from the augmentation role
    const uint32_t *flags;    // one word per augmentation wave (64 sequences): the launch's epoch once their characters are visible
    uint32_t *failures;       // host-mapped, sticky: token waves that gave up waiting (their chunk is poisoned; the API reports it)
    uint32_t epoch;
    uint32_t spins, naps;     // polls before a wave gives up; s_sleep(2) per poll
    uint32_t chunks_per_aug_wave;  // same-XCD form (FLAGS == 2): an augmentation wave mutates the rows of this many chunks of ITS class
};
// EOSV: the token at position bos + L (EOS) differs from the fill behind it.  Without it (cfg2, cfg5: no EOS) a store is
// bfi(keep, tokens, fill) -- no constant half of the rule entry: 16 registers and four LDS reads less per lane, 66 -> <= 64 VGPRs =
// EIGHT waves per SIMD instead of seven, which is what the cold-input regime (every character from HBM) is short of.
template <bool NT, int FLAGS, bool EOSV = true>
__device__ __forceinline__ void tokens_fast_body(uint32_t vblock, const int64_t *__restrict__ offsets, const uint8_t *__restrict__ chars,
                                                 uint8_t *__restrict__ out, uint32_t nchunks, uint32_t B, uint32_t PPR, uint32_t magic,
                                                 uint32_t shift, int32_t room, uint32_t packed, const T8Tab &tab, const T8Rules &rules,
                                                 const FusedWait &fw) {
    __shared__ __align__(16) uint4 s_rule[4][2][18];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t wave_s = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(wave));
    const uint32_t k0 = (vblock & 7u) + 8u * ((vblock >> 3) * 4u + wave_s);  // class = block % 8 (XCD-pinned)
    if (k0 >= nchunks) return;
    if constexpr (FLAGS != 0) {
        // The rows of this chunk: bc .. bc + nr (nr <= 32).  Their sequences were augmented by the waves (row / 64) of the
        // augmentation blocks, which are dispatched BEFORE every token block (lower block indices: the dispatcher hands blocks out
        // in order, as the decoupled look-back scans of rocPRIM rely on) and never wait for anything.  Protocol (hand-rolled release /
        // acquire over exactly the data it orders): an augmentation wave stores its mutated characters at AGENT scope (written
        // through to the memory side, where the XCDs agree), waits for their acknowledgement (s_waitcnt vmcnt(0)), then stores the
        // launch's epoch into its flag; a token wave loads the flag at agent scope until it reads the epoch and only THEN issues its
        // character loads, also at agent scope (sc1: not served from an L2 / L1 line that predates the mutation).  A cache-wide
        // release / acquire pair (buffer_wbl2 / buffer_inv) does the same for ALL memory and took 326 us instead of 42.
        // The wait is BOUNDED, and a wave that gives up never encodes stale characters under BSQ_OK: it
        // POISONS its chunk -- every byte 0xFF, which no token matrix contains -- and counts itself in host-mapped memory that
        // bsq_augment_tokenize_device (at its next entry) and bsq_fused_status report.
        const uint32_t g0f = k0 * (kChunk / 16);
        const uint32_t bcf = __umulhi(g0f, magic) >> shift;
        const uint32_t tcf = g0f - bcf * PPR;
        const uint32_t nrf = __umulhi(tcf + (kChunk / 16 - 1), magic) >> shift;
        const uint32_t lastrow = bcf + nrf < B ? bcf + nrf : B - 1;
        uint32_t f0 = bcf / 64u, f1 = lastrow / 64u;  // f1 - f0 <= 1
        uint32_t want = (fw.epoch << 4) | 15u;  // (15: published with written-through stores, valid on every XCD)
        if constexpr (FLAGS == 2) {
            // Same-XCD form (round 5): the rows of this chunk -- whole rows, 4096 / padlen of them -- were mutated by ONE augmentation
            // wave of this chunk's own class (= this XCD under round-robin placement): plain stores into the L2 this wave's loads go
            // through, no write-through, no second fetch from the memory side.  The flag carries the XCD the augmentation wave ran
            // on; on any other XCD the data is not visible here, so a mismatch fails LOUDLY (poison + count) like an expired wait.
            const uint32_t m = (k0 >> 3) / fw.chunks_per_aug_wave;               // augmentation wave of class k0 % 8
            f0 = f1 = (((m >> 2) * 8u + (k0 & 7u)) << 2) + (m & 3u);             // block (m / 4) * 8 + class, wave m % 4
            uint32_t xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            want = (fw.epoch << 4) | (xcc & 15u);
        }
        bool ok = true;
        if (static_cast<uint32_t>(lane) <= f1 - f0) {
            ok = false;
            for (uint32_t spin = 0; spin < fw.spins; ++spin) {  // (default 2^18 polls: ~1 s)
                const uint32_t seen = __hip_atomic_load(fw.flags + f0 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (seen == want) {
                    ok = true;
                    break;
                }
                if (FLAGS == 2 && (seen >> 4) == (want >> 4)) break;  // published from ANOTHER XCD: not ok, no point in waiting
                // ~3.5 us between polls: 8192 resident chunk waves polling every ~60 ns slowed the augmentation's own memory
                // operations down (cfg5aug 45.2 us; 60 naps: 42.8-43.1; one long nap, then short ones: 44-45 -- profiles/r03/augment_fused_flags_ab.txt)
                for (uint32_t z = 0; z < fw.naps; ++z) __builtin_amdgcn_s_sleep(2);
            }
        }
        if (__builtin_amdgcn_ballot_w64(!ok) != 0) {  // wave-uniform: gave up
            if (lane == 0) __hip_atomic_fetch_add(fw.failures, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            const uint32_t left = B - bcf > 64u ? 64u : B - bcf;
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if ((__umulhi(tcf + static_cast<uint32_t>(u * 64 + lane), magic) >> shift) < left)  // (pieces behind the last row are not written)
                    store16<NT>(out + static_cast<int64_t>(k0) * kChunk + lane * 16 + u * 1024, uint4{~0u, ~0u, ~0u, ~0u});
            return;
        }
        // (no cache-wide acquire: the characters are read with agent-scope loads below, which find what those waves wrote through)
        __builtin_amdgcn_wave_barrier();
    }
    auto div_p = [&](uint32_t n) { return __umulhi(n, magic) >> shift; };
    const uint32_t bos = packed & 1u, none_v = (packed & 2u) ? 0xFFu : 0u;
    const uint32_t bos_id = (packed >> 8) & 0xFFu;

    // ---- rows of the chunk; their offsets go out first ----
    const uint32_t g0 = k0 * (kChunk / 16);  // first piece (nchunks < 2^23)
    const uint32_t bc = div_p(g0);
    const uint32_t tc = g0 - bc * PPR;
    const uint32_t nr = div_p(tc + (kChunk / 16 - 1));  // rows bc .. bc + nr intersect the chunk (nr <= 32: P >= 128)
    int64_t o0 = 0, o1 = 0;
    if (static_cast<uint32_t>(lane) <= nr) {
        const uint32_t i0 = bc + lane, i1 = i0 + 1;
        o0 = offsets[i0 < B ? i0 : B];
        o1 = offsets[i1 < B ? i1 : B];
    }
    int64_t total_chars = offsets[B];  // scalar load, in flight beside them
    uint32_t T[8];  // the register alphabet table: fetched HERE, beside the offsets (the compiler would sink the s_load to its use,
                    // behind the character loads)
#pragma unroll
    for (int i = 0; i < 8; ++i) T[i] = tab.t[i];
    asm volatile("" : "+s"(T[0]), "+s"(T[1]), "+s"(T[2]), "+s"(T[3]), "+s"(T[4]), "+s"(T[5]), "+s"(T[6]), "+s"(T[7]), "+s"(total_chars));

    // ---- per-wave rule table: copied out of the kernel arguments while the offsets are in flight ----
    {
        if (lane < 18) {
            s_rule[wave][0][lane] = rules.keep[lane];
            if constexpr (EOSV) s_rule[wave][1][lane] = rules.cst[lane];
        }
        // wave-private: LDS operations of one wave execute in order, only the compiler must not reorder them
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }

    // ---- the characters: 4 unconditional unaligned 16-byte loads, all in flight together (see k_tokens_bp8) ----
    const int64_t off0 = (static_cast<int64_t>(__builtin_amdgcn_readfirstlane(static_cast<int32_t>(o0 >> 32))) << 32) |
                         static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int32_t>(o0)));
    const uint32_t rel = static_cast<uint32_t>(o0) - static_cast<uint32_t>(off0);
    const uint32_t len = static_cast<uint32_t>(o1) - static_cast<uint32_t>(o0);
    const int32_t Lr = static_cast<int32_t>(len > static_cast<uint32_t>(room) ? static_cast<uint32_t>(room) : len);
    const int64_t lo_b64 = -off0, hi_b64 = total_chars - off0 - 16;  // valid range of a vector's first byte, relative to off0
    const bool can_vec = hi_b64 >= lo_b64;                           // wave-uniform (the buffer holds >= 16 bytes)
    const int32_t lo_b = lo_b64 < INT32_MIN ? INT32_MIN : static_cast<int32_t>(lo_b64);
    const int32_t hi_b = hi_b64 > INT32_MAX ? INT32_MAX : (hi_b64 < lo_b ? lo_b : static_cast<int32_t>(hi_b64));
    const uint32_t span = static_cast<uint32_t>(hi_b) - static_cast<uint32_t>(lo_b);
    const uint32_t rows_left = B - bc > 64u ? 64u : B - bc;
    uint32_t uoff[4], qrow[4], rs[4];
    int32_t j0[4], L[4];
    bool live[4], slow[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {  // all eight cross-lane reads in flight together
        const uint32_t tl = tc + static_cast<uint32_t>(u * 64 + lane);
        const uint32_t q = div_p(tl);
        qrow[u] = q;
        j0[u] = static_cast<int32_t>((tl - q * PPR) << 4) - static_cast<int32_t>(bos);  // >= -1
        rs[u] = static_cast<uint32_t>(__builtin_amdgcn_ds_bpermute(static_cast<int>(q << 2), static_cast<int>(rel)));
        L[u] = __builtin_amdgcn_ds_bpermute(static_cast<int>(q << 2), Lr);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        live[u] = qrow[u] < rows_left;
        const bool need = live[u] && j0[u] < L[u];
        const uint32_t d = rs[u] + static_cast<uint32_t>(j0[u]) - static_cast<uint32_t>(lo_b);  // offset from the lowest valid address
        const bool fast = can_vec && need && d <= span;
        slow[u] = need && !fast;
        uoff[u] = fast ? d : 0u;
    }
    u32x4u cw[4];
    if (can_vec) {
        const uint8_t *cbase = chars + (off0 + lo_b);  // wave-uniform, inside the buffer
        if constexpr (FLAGS != 0) {
            // agent-scope loads (sc1: not served from this XCD's L2 or the CU's L1, where a line may predate the mutation of a
            // NEIGHBOURING sequence group).  One asm block with its own wait: the compiler must not touch the registers in between.
            asm volatile("global_load_dwordx4 %0, %4, off sc1\n\t"
                         "global_load_dwordx4 %1, %5, off sc1\n\t"
                         "global_load_dwordx4 %2, %6, off sc1\n\t"
                         "global_load_dwordx4 %3, %7, off sc1\n\t"
                         "s_waitcnt vmcnt(0)"
                         : "=&v"(cw[0]), "=&v"(cw[1]), "=&v"(cw[2]), "=&v"(cw[3])
                         : "v"(cbase + uoff[0]), "v"(cbase + uoff[1]), "v"(cbase + uoff[2]), "v"(cbase + uoff[3])
                         : "memory");
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) cw[u] = *reinterpret_cast<const u32x4u *>(cbase + uoff[u]);
        }
    } else {
#pragma unroll
        for (int u = 0; u < 4; ++u) cw[u] = u32x4u{0x41434447u, 0x61636474u, 0x4B4C4D4Eu, 0x50515253u};
    }
    // rule entries of the four stores: LDS reads that run beside the character loads
    uint4 keep[4], cst[EOSV ? 4 : 1];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int32_t dd = L[u] - j0[u];
        const int32_t idx = (dd < -1 ? -1 : (dd > 16 ? 16 : dd)) + 1;
        keep[u] = s_rule[wave][0][idx];
        if constexpr (EOSV) cst[u] = s_rule[wave][1][idx];
    }
    const uint32_t fill_w = (packed >> 24) * 0x01010101u;  // !EOSV
    if (__builtin_amdgcn_ballot_w64(slow[0] | slow[1] | slow[2] | slow[3]) != 0) {  // first / last bytes of the buffer: never read outside it
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (slow[u]) {
                const int64_t start = offsets[bc + qrow[u]];
                uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll 1
                for (int i = 0; i < 16; ++i)
                    if (j0[u] + i >= 0 && j0[u] + i < L[u])
                        w[i >> 2] |= static_cast<uint32_t>(FLAGS != 0 ? __hip_atomic_load(chars + start + j0[u] + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                                                 : chars[start + j0[u] + i]) << (8 * (i & 3));
                cw[u] = u32x4u{w[0], w[1], w[2], w[3]};
            }
        }
    }
    uint8_t *dst = out + static_cast<int64_t>(k0) * kChunk + lane * 16;
    // VT: the alphabet table and the selector constant in VECTOR registers (lookup4_perm_v: 12 instead of 18 vector instructions per
    // word) -- where the no-EOS form has the registers to spare
    constexpr bool VT = !EOSV && FLAGS == 0;
    uint32_t Tv[8], k3210 = 0x03020100u;
    if constexpr (VT) {
#pragma unroll
        for (int q = 0; q < 8; ++q) Tv[q] = T[q];
        asm volatile("" : "+v"(Tv[0]), "+v"(Tv[1]), "+v"(Tv[2]), "+v"(Tv[3]), "+v"(Tv[4]), "+v"(Tv[5]), "+v"(Tv[6]), "+v"(Tv[7]), "+v"(k3210));
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const uint32_t in[4] = {cw[u].x, cw[u].y, cw[u].z, cw[u].w};
        uint32_t w[4], bad = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if constexpr (VT) w[q] = lookup4_perm_v(in[q], Tv, k3210);
            else w[q] = lookup4_perm(in[q], T);
            bad |= (in[q] ^ 0x40404040u) & 0xC0C0C0C0u;
        }
        if (__builtin_amdgcn_ballot_w64(bad != 0) != 0) {  // some lane holds a non-letter: exact masks (rare)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t m = nonletter_mask(in[q]);
                w[q] = (w[q] & ~m) | (m & (none_v * 0x01010101u));
            }
        }
        uint4 o;
        if constexpr (EOSV) {
            o.x = (w[0] & keep[u].x) | cst[u].x;
            o.y = (w[1] & keep[u].y) | cst[u].y;
            o.z = (w[2] & keep[u].z) | cst[u].z;
            o.w = (w[3] & keep[u].w) | cst[u].w;
        } else {  // v_bfi_b32: kept bytes from the tokens, the rest fill
            o.x = (w[0] & keep[u].x) | (fill_w & ~keep[u].x);
            o.y = (w[1] & keep[u].y) | (fill_w & ~keep[u].y);
            o.z = (w[2] & keep[u].z) | (fill_w & ~keep[u].z);
            o.w = (w[3] & keep[u].w) | (fill_w & ~keep[u].w);
        }
        if (j0[u] < 0) o.x = (o.x & ~0xFFu) | bos_id;  // position 0 with BOS
        if (live[u]) store16<NT>(dst + u * 1024, o);
    }
}

template <bool NT, bool EOSV>
__global__ __launch_bounds__(kThreads) void k_tokens_bp8_fast(const int64_t *__restrict__ offsets, const uint8_t *__restrict__ chars,
                                                              uint8_t *__restrict__ out, uint32_t nchunks, uint32_t B, uint32_t PPR,
                                                              uint32_t magic, uint32_t shift, int32_t room, uint32_t packed,
                                                              T8Tab tab, T8Rules rules) {
    tokens_fast_body<NT, 0, EOSV>(blockIdx.x, offsets, chars, out, nchunks, B, PPR, magic, shift, room, packed, tab, rules, FusedWait{});
}

// SEVERAL INDEPENDENT BATCHES IN ONE LAUNCH (round 6, bsq_tokenize_device_multi).  A cold 16-40-us token launch spends a visible part of
// its life filling the chip and draining it again, and on one in-order stream the next batch's launch cannot start under the tail of
// this one (round 5: two alternating streams lifted cfg2 from 0.58 to 0.71 of the roof).  Here the grid is the CONCATENATION of up to
// kMultiMax batches' grids -- one ramp-up and one drain per launch instead of per batch.  Every batch's block range starts at a multiple
// of 8, so (block % 8) stays the chunk class of the batch's own stream and the XCD pinning holds; what differs per batch (its three
// pointers, its chunk count and its sequence count) is a table in the kernel-argument segment, read with scalar loads after a
// compare chain on the block index; what the batches share -- padlen, alphabet, BOS / EOS / PAD rules -- is as in the one-batch kernel.
constexpr int kMultiMax = 8;
struct T8Multi {
    const int64_t *offsets[kMultiMax];
    const uint8_t *chars[kMultiMax];
    uint8_t *out[kMultiMax];
    uint32_t first_block[kMultiMax];  // multiples of 8, ascending; entries behind the last batch: 0xFFFFFFFF
    uint32_t units[kMultiMax];        // chunks of the (B,P) matrix / sequence tiles of the (P,B) matrix
    uint32_t B[kMultiMax];
};
__device__ __forceinline__ uint32_t multi_batch_of(const T8Multi &m, uint32_t blk) {
    uint32_t i = 0;
#pragma unroll
    for (int k = 1; k < kMultiMax; ++k) i += blk >= m.first_block[k] ? 1u : 0u;
    return i;
}

template <bool NT, bool EOSV>
__global__ __launch_bounds__(kThreads) void k_tokens_bp8_fast_multi(uint32_t PPR, uint32_t magic, uint32_t shift, int32_t room, uint32_t packed,
                                                                    T8Tab tab, T8Rules rules, T8Multi m) {
    const uint32_t blk = blockIdx.x;
    const uint32_t i = multi_batch_of(m, blk);
    tokens_fast_body<NT, 0, EOSV>(blk - m.first_block[i], m.offsets[i], m.chars[i], m.out[i], m.units[i], m.B[i], PPR, magic, shift, room, packed,
                                  tab, rules, FusedWait{});
}


// BASELINE config 5 as ONE launch (round 3): BLOSUM62 augmentation (bsq_augment.hip) and the (B,P) int8 token matrix.  The first
// `aug_blocks` workgroups (a multiple of 8, so that the token role's chunk classes stay pinned to their XCDs) are k_augment_groups
// on 256 sequences each; every one of their waves, once its 64 sequences are mutated, publishes the launch's epoch in flags[wave].
// The workgroups behind them are k_tokens_bp8_fast; a chunk wave waits for the (one or two) flags of its rows and runs as usual
// (protocol, bounded wait and what happens when it expires: tokens_fast_body).  The augmentation is one generation of waves that
// lives ~7 us; as its own launch it cost 16.4 us on cfg5 -- launch latency and completion of a kernel this small are exposed, and
// the token kernel could not start before them (profiles/r03/augment_timeline.txt).  Results are those of the two launches, bit for bit.
// Round 4 measured three other shapes of this launch and kept this one (profiles/r04/aug_fused_ab.txt): the mutations computed inside
// the token workgroups (no wait at all; 74-105 us), token waves that stream at once and patch the mutated positions from a side list
// after a late wait (46.5 us) or after an early one (47.9 us) -- against 41-42 us here and 46.2-46.9 for the two launches.
struct FusedAug {
    uint8_t *chars;  // the same buffer the token role reads
    int64_t B;
    const bsq_aug::AugTable *tab;
    double frac;
    uint64_t seed;
    int32_t chain_len;
    uint32_t aug_blocks;
    uint32_t *flags;          // aug_blocks * 4 words (one per augmentation wave)
    FusedWait wait;
    uint2 *rec;               // no-wait form: B * chain_len mutation records (bsq_augment_dev.h, RECORD)
};
template <bool NT, int K, bool SAMEXCD = false>
__global__ __launch_bounds__(kThreads) void k_augment_tokens_fused(const int64_t *__restrict__ offsets, const uint8_t *chars,
                                                                   uint8_t *__restrict__ out, uint32_t nchunks, uint32_t B, uint32_t PPR,
                                                                   uint32_t magic, uint32_t shift, int32_t room, uint32_t packed,
                                                                   T8Tab tab, T8Rules rules, FusedAug fa) {
    if (blockIdx.x < fa.aug_blocks) {
        if constexpr (SAMEXCD) {
            // block a = (m / 4) * 8 + class, wave m % 4: the rows of chunks class + 8 * (m * CW + i), i < CW = 64 / R chunks of R = 256 / PPR
            // whole rows each -- lane i * R + r takes row r of chunk i.  Plain stores: the only readers are token waves of the same class.
            const uint32_t R = 256u / PPR, CW = fa.wait.chunks_per_aug_wave;
            const uint32_t m = (blockIdx.x >> 3) * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
            const uint32_t i = lane / R, r = lane - i * R;
            const uint64_t k = (blockIdx.x & 7u) + 8ull * (static_cast<uint64_t>(m) * CW + i);
            const int64_t b = k < nchunks ? static_cast<int64_t>(k * R + r) : fa.B;
            bsq_aug::augment_groups_body<K, false, true>(blockIdx.x, fa.chars, offsets, fa.B, fa.chain_len, fa.frac, fa.seed, fa.tab, b < fa.B ? b : fa.B);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's stores have reached its XCD's L2
            if ((threadIdx.x & 63) == 0) {
                uint32_t xcc;
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
                __hip_atomic_store(fa.flags + blockIdx.x * 4u + (threadIdx.x >> 6), (fa.wait.epoch << 4) | (xcc & 15u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            return;
        }
        bsq_aug::augment_groups_body<K, true>(blockIdx.x, fa.chars, offsets, fa.B, fa.chain_len, fa.frac, fa.seed, fa.tab);
        // this wave's character stores (agent scope: written through) have been acknowledged ... (no cache-wide release: a
        // buffer_wbl2 per wave made the launch take 326 us instead of 49)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if ((threadIdx.x & 63) == 0)                          // ... before its flag is published
            __hip_atomic_store(fa.flags + blockIdx.x * 4u + (threadIdx.x >> 6), (fa.wait.epoch << 4) | 15u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    tokens_fast_body<NT, SAMEXCD ? 2 : 1>(blockIdx.x - fa.aug_blocks, offsets, chars, out, nchunks, B, PPR, magic, shift, room, packed, tab, rules, fa.wait);
}

// The NO-WAIT form (round 5) -- cfg5's default entry since then.  Nobody waits for anybody inside the launch: the augmentation role mutates the
// characters in place (single BYTE stores at agent scope -- augment_groups_body<K, COHERENT = true>, as compiled: a one-byte store is atomic
// whatever its scope, the scope only says where it becomes visible) and writes every mutation down (position, new byte); the token role is k_tokens_bp8_fast as it is, at
// once, on whatever it finds -- for a mutated position that is the old residue or the new one, depending on who came first.  A second, tiny
// launch (k_patch_tokens, one thread per sequence, in chain order) then stores the token of every recorded new residue at its place in the
// matrix: whichever version the token role saw, the matrix ends up as the tokens of the mutated batch, bit for bit (a mutation never
// changes a length, so BOS / EOS / PAD are where they were).  Against the flag form (above; still there under knob augment_fused = 2):
// the ~9 us the token stream used to wait for the augmentation are gone, so are the second fetch of the characters through sc1 loads, the
// polling, the epoch bookkeeping -- and the failure mode (a wait that could expire).  The token role READS `chars` through ordinary (restrict-
// qualified, cacheable) loads while other workgroups of the same launch store single bytes into it: by the letter of the language that is a
// data race; what the hardware does with it is read either the old or the new BYTE of a position (never a torn one), and the patch launch
// overwrites exactly those positions -- the tolerated race is the design, stated here so that nobody takes the qualifier for a proof
// (ADVICE round 5).  What it costs is
// the patch launch: 1 MB of records read, one byte store per mutation.  Round 4 had tried the side list with the wait kept INSIDE the
// launch (46.5 us: the late waits stalled the stream); profiles/r05/aug_nowait_patch_ab.txt.
template <bool NT, int K, bool EOSV>
__global__ __launch_bounds__(kThreads) void k_augment_tokens_nowait(const int64_t *__restrict__ offsets, const uint8_t *chars,
                                                                    uint8_t *__restrict__ out, uint32_t nchunks, uint32_t B, uint32_t PPR,
                                                                    uint32_t magic, uint32_t shift, int32_t room, uint32_t packed,
                                                                    T8Tab tab, T8Rules rules, FusedAug fa) {
    if (blockIdx.x < fa.aug_blocks) {
        // the augmentation's waves share their SIMDs with seven token waves each: with priority their 64-bit multiplies and FP64 draw are not
        // an eighth of the issue slots -- 32 768 sequences 16.2 -> 13.0 us; no effect from 131 072 on (profiles/r05/aug_nowait_patch_ab.txt)
        __builtin_amdgcn_s_setprio(3);
        bsq_aug::augment_groups_body<K, true, false, true>(blockIdx.x, fa.chars, offsets, fa.B, fa.chain_len, fa.frac, fa.seed, fa.tab, 0, fa.rec);
        return;
    }
    tokens_fast_body<NT, 0, EOSV>(blockIdx.x - fa.aug_blocks, offsets, chars, out, nchunks, B, PPR, magic, shift, room, packed, tab, rules, FusedWait{});
}

struct PatchLut {
    uint8_t v[128];  // token VALUE of every 7-bit byte (unmapped: 0, the memset value of tokenize.h:427)
};
__global__ __launch_bounds__(kThreads) void k_patch_tokens(const uint2 *__restrict__ rec, int64_t B, int32_t chain_len, int64_t P, int32_t bos,
                                                           int32_t room, PatchLut lut, uint8_t *__restrict__ out) {
    const int64_t b = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
    if (b >= B) return;
    for (int32_t m = 0; m < chain_len; ++m) {  // in chain order: a later mutation of the same position wins, as in the characters
        const uint2 r = rec[b * chain_len + m];
        if (r.x != 0xFFFFFFFFu && r.x < static_cast<uint32_t>(room)) out[b * P + bos + r.x] = (r.y < 128u) ? lut.v[r.y] : uint8_t(0);
    }
}

// k_tokens_pb8_fast: the (P,B) token matrix -- batch_tokenize's DEFAULT layout (batch_first=False, tokenize.cpp:82-98) --
// of 1-, 2- and 8-byte elements without a mask (SZ / FLT below; UA: rows that are only element-aligned), and the raw-id pass of
// the two-pass one-hot.  Round 3.
// k_tokens_raw (bsq_tiles.h) spends ~40 vector and ~10.6 LDS instructions per word of four tokens (byte lookups
// in an LDS table, four transposed ds_write_b8 per word, spans staged through LDS) and sits at 0.59 of the HBM roof on
// cfg2's shape, bound by instruction issue in BOTH pipes (profiles/r02/cfg2sf_sq_tcc_counters.txt).  Here, on a tile of
// TB sequences x 64 positions:
//   * lane (R, a, b) of a wave -- 16-lane row R, a = lane / 4 % 4, b = lane % 4 -- fetches the unaligned 16-byte piece b
//     (3 - b in odd a) of sequence 4 (4 wave + R) + a of the pass: four adjacent lanes read 64 contiguous characters (a first form with
//     one sequence per lane was bound by its 64-line load instructions: profiles/r03/pb8_lane_per_sequence_walk_lab.txt);
//     the pieces of all TB / 64 passes are in flight together;
//   * lookups out of the register table (LK = 1; foldable alphabets) or the LDS byte table (LK = 0), BOS / EOS / PAD
//     from the host-built rule table, exactly as k_tokens_bp8_fast;
//   * the byte transpose happens IN REGISTERS between the four lanes b, b + 4, b + 8, b + 12 of a row (the same piece of
//     four consecutive sequences): v_mov_dpp row_half_mirror + v_perm_b32, then row_ror:8 + v_perm_b32 leave lane (R, a, b)
//     with position 16 piece + 4 k + a of the row's four sequences -- ONE ds_write_b32 per word;
//   * LDS tile: position p lives in physical row 4 (p % 16) + p / 16, row stride TB + 8 bytes (2 banks): the 32 writes of a
//     half-wave fall on 32 banks; the rows leave as 8-byte LDS reads -> 16-byte stores of TB-byte segments.
// XCD-aware order: block b -> class b % 8 walks its own sequence tiles, the position tiles of a sequence tile back to back
// (rows that are not 64-byte aligned, and 4- / 8-byte elements: XCD-contiguous ranges, sequence tile fastest -- see the kernel).
struct T8Lut {
    uint32_t w[64];  // LK = 0: token VALUE of every byte (unmapped: none_v)
};
// SZ / FLT: the element type of the matrix -- 1, 2, 4, 8-byte integers (ids zero-extended: < 251) or, FLT, float / double.
// The tile is built in bytes whatever the type; only the last phase differs: a thread turns 16 / SZ token bytes of a tile row into
// one 16-byte store, so a tile row leaves as TB * SZ contiguous bytes.
template <int SZ, bool FLT>
__device__ __forceinline__ uint4 widen_tokens(const uint8_t *src) {
    if constexpr (SZ == 1) {
        const uint2 lo = *reinterpret_cast<const uint2 *>(src), hi = *reinterpret_cast<const uint2 *>(src + 8);
        return uint4{lo.x, lo.y, hi.x, hi.y};
    } else if constexpr (SZ == 2) {
        const uint2 w = *reinterpret_cast<const uint2 *>(src);  // 8 tokens; 0x0c in a selector byte = constant 0
        return uint4{__builtin_amdgcn_perm(0u, w.x, 0x0c010c00u), __builtin_amdgcn_perm(0u, w.x, 0x0c030c02u),
                     __builtin_amdgcn_perm(0u, w.y, 0x0c010c00u), __builtin_amdgcn_perm(0u, w.y, 0x0c030c02u)};
    } else if constexpr (SZ == 4) {
        const uint32_t w = *reinterpret_cast<const uint32_t *>(src);  // 4 tokens
        if constexpr (FLT)
            return uint4{__float_as_uint(static_cast<float>(w & 0xFFu)), __float_as_uint(static_cast<float>((w >> 8) & 0xFFu)),
                         __float_as_uint(static_cast<float>((w >> 16) & 0xFFu)), __float_as_uint(static_cast<float>(w >> 24))};
        else
            return uint4{w & 0xFFu, (w >> 8) & 0xFFu, (w >> 16) & 0xFFu, w >> 24};
    } else {
        const uint32_t w = *reinterpret_cast<const uint16_t *>(src);  // 2 tokens
        if constexpr (FLT) {
            const uint64_t a = static_cast<uint64_t>(__double_as_longlong(static_cast<double>(w & 0xFFu)));
            const uint64_t b = static_cast<uint64_t>(__double_as_longlong(static_cast<double>(w >> 8)));
            return uint4{static_cast<uint32_t>(a), static_cast<uint32_t>(a >> 32), static_cast<uint32_t>(b), static_cast<uint32_t>(b >> 32)};
        } else {
            return uint4{w & 0xFFu, 0u, w >> 8, 0u};
        }
    }
}

// UA: rows (pitch * SZ bytes) or the output are only element-aligned -- any batch size.  The 16-byte stores go out unaligned
// (gfx950 splits them in hardware), the pieces that cross the end of a row as single elements, and every XCD walks its own
// CONTIGUOUS range of sequence tiles: the memory sectors that two neighbouring tiles share are then written through ONE L2.
// NIB (round 5, raw ids of alphabets with at most 15 classes -- DNA and the reduced amino alphabets): the id matrix leaves as NIBBLES, two
// sequences per byte (sequence 2 m in the low half of byte m of a position row, 255 -> 15 = no one), a row of the tile as TB / 2 bytes: the
// scratch of the two-pass one-hot is written and re-read at half its bytes (cfg4 f32: 160 -> 80 MB of 318 in the raw pass).
template <bool NT, int TB, int LK, int SZ = 1, bool FLT = false, bool UA = false, bool NIB = false>
__device__ __forceinline__ void tokens_pb8_body(const uint32_t vblock, const int64_t *__restrict__ offsets, const uint8_t *__restrict__ chars,
                                                uint8_t *__restrict__ out, int64_t pitch, uint32_t B, uint32_t P, uint32_t ntb, uint32_t ntt,
                                                uint32_t magic, uint32_t shift, int32_t room, uint32_t packed, uint32_t tt0, const T8Tab &tab,
                                                const T8Rules &rules, const T8Lut &lut) {
    // (tt0: the launch covers the position tiles tt0 .. tt0 + ntt - 1 -- a SLICE of the matrix; `out` is then the address row 0 would
    //  have, so that row t of the slice lands at out + t * pitch as everywhere below)
    constexpr int TT = 64, STRIDE = TB + 8, PASSES = TB / 64;
    static_assert(TB % 64 == 0 && (STRIDE / 4) % 32 == 2, "tile shape");
    __shared__ __align__(16) uint4 s_rule[2][18];
    __shared__ __align__(16) uint8_t s_lut[LK == 0 ? 256 : 16];
    __shared__ __align__(16) uint8_t s_t[TT * STRIDE];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const uint32_t cls = vblock & 7u, i = vblock >> 3;
    const uint32_t quo = (magic ? __umulhi(i, magic) : i) >> shift;  // i / ntt (magic 0: a power of two); contiguous form: i / per
    uint32_t tt = i - quo * ntt, tb = quo * 8u + cls;
    if (UA || (packed & 4u)) {
        // Rows that are not 64-byte aligned: neighbouring sequence tiles share memory sectors.  Every XCD walks its own CONTIGUOUS
        // range of `per` sequence tiles, sequence tile fastest: the two halves of a shared sector are written back to back through
        // ONE L2 (the host passes the division constants of `per` instead of ntt).
        const uint32_t per = (ntb + 7u) / 8u;
        tt = quo;
        tb = cls * per + (i - quo * per);
        if (tt >= ntt) return;
    }
    if (tb >= ntb) return;
    // PAIRED TAIL (round 6; packed bit 3, aligned class-pinned form only): the matrix's last position tile holds at most 32 positions (padlen 160 =
    // 2.5 tiles: BASELINE config 4), so half of the tile's pieces -- and of its lookups, transposes and LDS traffic -- would belong to positions that
    // do not exist; the kernel is bound by vector issue (569 vector instructions per wave at ~90 % VALU activity: profiles/r06/cfg4b_sq_tcc_counters.txt).
    // The tail tile of sequence tile tb (group quo even) takes the tail of tb + 8 as well (the same chunk class = the same XCD): lanes whose piece
    // would be 2 or 3 work on pieces 0 / 1 of the PARTNER's sequences, the partner's own tail workgroup returns at once.
    const bool tail_pair = (packed & 8u) != 0u && (tt + tt0) == (P - 1u) / TT;  // wave-uniform
    // the LATER workgroup of the two (odd group) does both tails: its partner's lines were fetched by the partner's other tiles just before
    const bool later = (quo & 1u) != 0u;
    if (tail_pair && !later && tb + 8u < ntb) return;
    const uint32_t bos = packed & 1u, none_v = (packed & 2u) ? 0xFFu : 0u;
    const uint32_t bos_id = (packed >> 8) & 0xFFu;
    const int32_t t0 = static_cast<int32_t>(tt + tt0) * TT;
    const int la = (lane >> 2) & 3, lb = lane & 3;
    const int pc = (la & 1) ? 3 - lb : lb;  // the lane's piece: reversed in odd a, so that row_half_mirror pairs a with a ^ 1 on the SAME piece
    const uint32_t second = (tail_pair && pc >= 2) ? 1u : 0u;  // this lane works for the partner tile
    const uint32_t tbl = later ? tb - 8u * second : tb + 8u * second;  // (an even group without a partner: tb + 8 >= ntb, those lanes find no sequences)

    // ---- the spans first (the four lanes of a sequence read the same two words) ----
    int64_t o0[PASSES], o1[PASSES];
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
        const uint32_t b = tbl * TB + ps * 64 + (tid >> 2);  // the lane's sequence of this pass
        o0[ps] = offsets[b < B ? b : B];
        o1[ps] = offsets[b + 1 < B ? b + 1 : B];
    }
    int64_t total_chars = offsets[B];
    uint32_t T[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) T[q] = tab.t[q];
    asm volatile("" : "+s"(T[0]), "+s"(T[1]), "+s"(T[2]), "+s"(T[3]), "+s"(T[4]), "+s"(T[5]), "+s"(T[6]), "+s"(T[7]), "+s"(total_chars));
    uint32_t Tv[8], k3210 = 0x03020100u;  // the table and the selector constant in vector registers (see lookup4_perm_v)
#pragma unroll
    for (int q = 0; q < 8; ++q) Tv[q] = T[q];
    asm volatile("" : "+v"(Tv[0]), "+v"(Tv[1]), "+v"(Tv[2]), "+v"(Tv[3]), "+v"(Tv[4]), "+v"(Tv[5]), "+v"(Tv[6]), "+v"(Tv[7]), "+v"(k3210));
    if (tid < 18) {
        s_rule[0][tid] = rules.keep[tid];
        s_rule[1][tid] = rules.cst[tid];
    }
    if constexpr (LK == 0) {
        if (tid >= 64 && tid < 128) reinterpret_cast<uint32_t *>(s_lut)[tid - 64] = lut.w[tid - 64];
    }
    __syncthreads();

    // ---- the characters: the piece of every pass in flight together ----
    const int32_t j0 = t0 + 16 * (pc - 2 * static_cast<int>(second)) - static_cast<int32_t>(bos);  // character index of the piece's first byte (>= -1)
    u32x4u cw[PASSES];
    int32_t Lr[PASSES];
    bool slow_any = false;
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
        const uint32_t len = static_cast<uint32_t>(o1[ps] - o0[ps]);
        Lr[ps] = static_cast<int32_t>(len > static_cast<uint32_t>(room) ? static_cast<uint32_t>(room) : len);
        const int64_t a = o0[ps] + j0;
        const bool need = j0 < Lr[ps];
        const bool fast = need && a >= 0 && a + 16 <= total_chars;  // never read outside the buffer
        slow_any |= need && !fast;
        cw[ps] = u32x4u{0x41414141u, 0x41414141u, 0x41414141u, 0x41414141u};
        if (fast) cw[ps] = *reinterpret_cast<const u32x4u *>(chars + a);
    }
    if (__builtin_amdgcn_ballot_w64(slow_any) != 0) {  // first / last bytes of the buffer (a handful of lanes per launch)
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps) {
            const int64_t a = o0[ps] + j0;
            if (j0 < Lr[ps] && !(a >= 0 && a + 16 <= total_chars)) {
                uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll 1
                for (int k = 0; k < 16; ++k)
                    if (j0 + k >= 0 && j0 + k < Lr[ps])
                        w[k >> 2] |= static_cast<uint32_t>(chars[a + k]) << (8 * (k & 3));
                cw[ps] = u32x4u{w[0], w[1], w[2], w[3]};
            }
        }
    }

    // ---- lookups, rules, transposes between the lanes a = 0..3 of a piece, one ds_write_b32 per word ----
    const uint32_t sel1 = (la & 1) ? 0x03070105u : 0x06020400u;  // even a: (w0, p0, w2, p2); odd a: (p1, w1, p3, w3)
    const uint32_t sel2 = (la & 2) ? 0x03020706u : 0x05040100u;  // a = 0, 1: (x.lo, y.lo); a = 2, 3: (y.hi, x.hi)
    // word k of the lane goes to physical row 4 (4 k + a) + piece, dword column 16 pass + 4 wave + R
    uint8_t *wbase = s_t + (4 * la + pc) * STRIDE + wave * 16 + (lane >> 4) * 4;
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
        const int32_t dd = Lr[ps] - j0;
        const int32_t idx = (dd < -1 ? -1 : (dd > 16 ? 16 : dd)) + 1;
        const uint4 keep = s_rule[0][idx], cst = s_rule[1][idx];
        const uint32_t in[4] = {cw[ps].x, cw[ps].y, cw[ps].z, cw[ps].w};
        uint32_t w[4];
        if constexpr (LK == 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                w[q] = static_cast<uint32_t>(s_lut[in[q] & 0xFFu]) | (static_cast<uint32_t>(s_lut[(in[q] >> 8) & 0xFFu]) << 8) |
                       (static_cast<uint32_t>(s_lut[(in[q] >> 16) & 0xFFu]) << 16) | (static_cast<uint32_t>(s_lut[in[q] >> 24]) << 24);
        } else {
            uint32_t bad = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                w[q] = lookup4_perm_v(in[q], Tv, k3210);
                bad |= (in[q] ^ 0x40404040u) & 0xC0C0C0C0u;
            }
            if (__builtin_amdgcn_ballot_w64(bad != 0) != 0) {  // some lane holds a non-letter: exact masks (rare)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const uint32_t m = nonletter_mask(in[q]);
                    w[q] = (w[q] & ~m) | (m & (none_v * 0x01010101u));
                }
            }
        }
        w[0] = (w[0] & keep.x) | cst.x;
        w[1] = (w[1] & keep.y) | cst.y;
        w[2] = (w[2] & keep.z) | cst.z;
        w[3] = (w[3] & keep.w) | cst.w;
        if (j0 < 0) w[0] = (w[0] & ~0xFFu) | bos_id;  // position 0 with BOS
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            // the word of lane a ^ 1 on the same piece: row_half_mirror (lane i <-> 7 - i of each half row)
            const int p1 = __builtin_amdgcn_update_dpp(0, static_cast<int>(w[q]), 0x141, 0xF, 0xF, false);
            const uint32_t x = __builtin_amdgcn_perm(static_cast<uint32_t>(p1), w[q], sel1);
            const uint32_t y = static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), 0x128, 0xF, 0xF, false));  // lane a ^ 2
            const uint32_t tr = __builtin_amdgcn_perm(y, x, sel2);  // position 16 piece + 4 q + a of the row's four sequences
            *reinterpret_cast<uint32_t *>(wbase + 16 * q * STRIDE + ps * 64) = tr;
        }
    }
    __syncthreads();

    // ---- the tile's position rows: TB * SZ contiguous bytes of the output each (pitch in ELEMENTS) ----
    // columns that exist: the scratch of the two-pass one-hot is written up to its padded pitch; a token matrix has B of them --
    // its row stride may be larger: a COLUMN BLOCK of a wider (P, pitch) matrix (bsq_tokenize_block_device)
    const int64_t ncols = (packed & 2u) ? pitch : static_cast<int64_t>(B);
    constexpr int N = 16 / SZ;     // tokens per 16-byte store
    constexpr int PPR = TB / N;    // stores per tile row
    static_assert((TT * PPR) % kThreads == 0 && (PPR & (PPR - 1)) == 0, "row walk");
    if constexpr (NIB) {
        static_assert(!NIB || (SZ == 1 && !UA && !FLT), "nibble ids: the raw byte matrix");
        constexpr int PPRN = TB / 32;  // 16-byte stores per tile row: 32 ids each
        static_assert((TT * PPRN) % kThreads == 0 && (PPRN & (PPRN - 1)) == 0, "row walk");
#pragma unroll
        for (int f0 = 0; f0 < TT * PPRN; f0 += kThreads) {
            const int f = f0 + tid;
            const int rr = f / PPRN, piece = f % PPRN;
            const bool part = tail_pair && (rr & 3) >= 2;  // (paired tail: physical rows of pieces 2 / 3 hold the partner tile's positions 0 .. 31)
            const int32_t t = t0 + 16 * ((rr & 3) - (part ? 2 : 0)) + (rr >> 2);
            const int64_t col = static_cast<int64_t>(part ? (later ? tb - 8u : tb + 8u) : tb) * TB + piece * 32;
            const uint2 *src = reinterpret_cast<const uint2 *>(s_t + rr * STRIDE + piece * 32);
            uint32_t o[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint2 d = src[q];
                uint32_t lo = d.x & 0x0F0F0F0Fu, hi = d.y & 0x0F0F0F0Fu;
                lo |= lo >> 4;  // byte 0: id 0 | id 1 << 4, byte 2: id 2 | id 3 << 4
                hi |= hi >> 4;
                o[q] = __builtin_amdgcn_perm(hi, lo, 0x06040200u);
            }
            if (t < static_cast<int32_t>(P) && col < ncols)
                store16<NT>(out + ((static_cast<int64_t>(t) * pitch + col) >> 1), uint4{o[0], o[1], o[2], o[3]});
        }
    } else if constexpr (!UA) {
#pragma unroll
        for (int f0 = 0; f0 < TT * PPR; f0 += kThreads) {
            const int f = f0 + tid;
            const int rr = f / PPR, piece = f % PPR;           // physical row rr holds position 16 (rr % 4) + rr / 4
            const bool part = tail_pair && (rr & 3) >= 2;      // (paired tail: the partner tile's positions 0 .. 31)
            const int32_t t = t0 + 16 * ((rr & 3) - (part ? 2 : 0)) + (rr >> 2);
            const int64_t col = static_cast<int64_t>(part ? (later ? tb - 8u : tb + 8u) : tb) * TB + piece * N;
            if (t < static_cast<int32_t>(P) && col < ncols)
                store16<NT>(out + (static_cast<int64_t>(t) * pitch + col) * SZ, widen_tokens<SZ, FLT>(s_t + rr * STRIDE + piece * N));
        }
    } else {
        // Rows that are only element-aligned: the row segment of the tile is cut at the 16-byte lines of the OUTPUT.  Slot 0 = the
        // head (the elements up to the first line: a neighbouring tile writes the rest of that line), slots 1 .. PPR = whole aligned
        // lines -- 16-byte nt stores as in the aligned case, their tokens read from the tile at a BYTE offset (three 8-byte LDS
        // reads + v_alignbyte) --, the last one the tail; heads and tails leave as single elements.
        static_assert(!UA || (SZ <= 2 && !FLT), "unaligned rows: 1- and 2-byte integers");
        constexpr int SLOTS = PPR + 1;
        const int64_t c_tile = static_cast<int64_t>(tb) * TB;
        const int32_t nb = ncols - c_tile < TB ? static_cast<int32_t>(ncols - c_tile) : TB;  // elements of the tile's row segment
        for (int f = tid; f < TT * SLOTS; f += kThreads) {
            const int rr = f / SLOTS, slot = f % SLOTS;
            const int32_t t = t0 + 16 * (rr & 3) + (rr >> 2);
            if (t >= static_cast<int32_t>(P)) continue;
            const int64_t e0 = static_cast<int64_t>(t) * pitch + c_tile;  // element index of the segment's first element
            const uint32_t mis = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(out) + static_cast<uint64_t>(e0) * SZ) & 15u;
            const int32_t head = static_cast<int32_t>(((16u - mis) & 15u) / SZ);  // elements up to the first 16-byte line
            const int32_t c0 = slot == 0 ? 0 : head + (slot - 1) * N;
            const int32_t left = nb - c0;
            const int32_t cnt = slot == 0 ? (head < left ? head : left) : (left > N ? N : left);
            if (cnt <= 0) continue;
            const uint8_t *src = s_t + rr * STRIDE + c0;
            uint8_t *dst = out + (e0 + c0) * SZ;
            if (cnt == N) {  // a whole line: N token bytes from the byte offset c0 of the tile row (rows are 8-byte aligned, 8 bytes of padding behind)
                const uint32_t sh = static_cast<uint32_t>(c0) & 7u;
                const uint2 *q = reinterpret_cast<const uint2 *>(src - sh);
                const uint2 q0 = q[0], q1 = q[1];
                uint32_t d0 = q0.x, d1 = q0.y, d2 = q1.x, d3 = q1.y, d4 = 0;
                if constexpr (SZ == 1) {
                    const uint2 q2 = q[2];
                    d4 = q2.x;
                    if (sh & 4u) d0 = d1, d1 = d2, d2 = d3, d3 = d4, d4 = q2.y;
                } else {
                    if (sh & 4u) d0 = d1, d1 = d2, d2 = d3;
                }
                const uint32_t w0 = __builtin_amdgcn_alignbyte(d1, d0, sh & 3u), w1 = __builtin_amdgcn_alignbyte(d2, d1, sh & 3u);
                if constexpr (SZ == 1) {
                    const uint32_t w2 = __builtin_amdgcn_alignbyte(d3, d2, sh & 3u), w3 = __builtin_amdgcn_alignbyte(d4, d3, sh & 3u);
                    store16<NT>(dst, uint4{w0, w1, w2, w3});
                } else {
                    store16<NT>(dst, uint4{__builtin_amdgcn_perm(0u, w0, 0x0c010c00u), __builtin_amdgcn_perm(0u, w0, 0x0c030c02u),
                                           __builtin_amdgcn_perm(0u, w1, 0x0c010c00u), __builtin_amdgcn_perm(0u, w1, 0x0c030c02u)});
                }
            } else {
                for (int k = 0; k < cnt; ++k) {
                    if constexpr (SZ == 1) dst[k] = src[k];
                    else reinterpret_cast<uint16_t *>(dst)[k] = src[k];
                }
            }
        }
    }
}

template <bool NT, int TB, int LK, int SZ = 1, bool FLT = false, bool UA = false, bool NIB = false>
__global__ __launch_bounds__(kThreads) void k_tokens_pb8_fast(const int64_t *__restrict__ offsets, const uint8_t *__restrict__ chars,
                                                              uint8_t *__restrict__ out, int64_t pitch, uint32_t B, uint32_t P,
                                                              uint32_t ntb, uint32_t ntt, uint32_t magic, uint32_t shift, int32_t room,
                                                              uint32_t packed, uint32_t tt0, T8Tab tab, T8Rules rules, T8Lut lut) {
    tokens_pb8_body<NT, TB, LK, SZ, FLT, UA, NIB>(blockIdx.x, offsets, chars, out, pitch, B, P, ntb, ntt, magic, shift, room, packed, tt0, tab, rules, lut);
}

// Several independent (P,B) matrices of ONE tokenizer and padlen in one launch (see k_tokens_bp8_fast_multi): the 64-byte aligned form only
// (block -> class block % 8, position tiles of a sequence tile back to back: the divisor of the block index is ntt, which the batches
// share); a batch's matrix is its own (pitch = B).
template <bool NT, int LK, int SZ>
__global__ __launch_bounds__(kThreads) void k_tokens_pb8_fast_multi(uint32_t P, uint32_t ntt, uint32_t magic, uint32_t shift, int32_t room, uint32_t packed,
                                                                    T8Tab tab, T8Rules rules, T8Lut lut, T8Multi m) {
    const uint32_t blk = blockIdx.x;
    const uint32_t i = multi_batch_of(m, blk);
    const uint32_t Bi = m.B[i];
    tokens_pb8_body<NT, 256, LK, SZ>(blk - m.first_block[i], m.offsets[i], m.chars[i], m.out[i], static_cast<int64_t>(Bi), Bi, P, m.units[i], ntt, magic,
                                     shift, room, packed, 0u, tab, rules, lut);
}

template <bool NT, int LK>
void launch_variant(const T8Params &c, dim3 grid, size_t pad, hipStream_t s) {
    if (c.mask)  // raw ids for the masked channels-first one-hot (aligned shapes only: see launch_tokens_bp8)
        hipLaunchKernelGGL((k_tokens_bp8<NT, LK, 0, false, true>), grid, dim3(kThreads), pad, s, c);
    else if (c.P % 16 != 0 || reinterpret_cast<uintptr_t>(c.out) % 16 != 0)
        hipLaunchKernelGGL((k_tokens_bp8<NT, LK, 0, true>), grid, dim3(kThreads), pad, s, c);
    else
        hipLaunchKernelGGL((k_tokens_bp8<NT, LK, 0>), grid, dim3(kThreads), pad, s, c);
}

}  // namespace

namespace bsq_internal {

// True when every mapped byte of `lut` is a letter position (0x40..0x7F) and both cases map alike: the
// 32-entry folded table represents it exactly (all reference alphabets but BYTES, alphabet.h:39,44).
static bool fold_table(const int8_t lut[256], uint32_t tab[8], uint32_t none_v) {
    for (int i = 0; i < 8; ++i) tab[i] = none_v * 0x01010101u;
    for (int c = 0; c < 256; ++c) {
        const bool mapped = c < 128 && lut[c] >= 0;
        if (!mapped) continue;
        if (c < 0x40) return false;
        const int other = c ^ 0x20;
        if (lut[other] != lut[c]) return false;
        uint32_t &t = tab[(c & 31) >> 2];
        t = (t & ~(0xFFu << (8 * (c & 3)))) | (static_cast<uint32_t>(static_cast<uint8_t>(lut[c])) << (8 * (c & 3)));
    }
    // a letter position that is unmapped in one case must be unmapped in the other (checked above for mapped ones)
    for (int c = 0x40; c < 0x80; ++c)
        if ((lut[c] >= 0) != (lut[c ^ 0x20] >= 0)) return false;
    return true;
}

// The BOS / EOS / PAD rule table of a launch (see T8Rules): entry e holds n = e - 1 kept bytes.
static void build_rules(uint32_t fill_v, uint32_t at_len_v, T8Rules &rules) {
    const uint32_t fill_w = (fill_v & 0xFFu) * 0x01010101u, at_w = (at_len_v & 0xFFu) * 0x01010101u;
    for (int e = 0; e < 18; ++e) {  // entry e: n = e - 1 bytes kept, byte n (if any) = token at bos + L, the rest fill
        const int n = e - 1;
        uint32_t keep[4], cst[4];
        for (int q = 0; q < 4; ++q) {
            const int nv = n - 4 * q;
            const int nvc = nv < 0 ? 0 : (nv > 4 ? 4 : nv);
            const uint32_t kq = nvc == 4 ? 0xFFFFFFFFu : ((1u << (8 * nvc)) - 1u);
            const uint32_t at = (nv >= 0 && nv < 4) ? (0xFFu << (8 * nv)) : 0u;
            keep[q] = kq;
            cst[q] = (fill_w & ~kq & ~at) | (at_w & at);
        }
        rules.keep[e] = uint4{keep[0], keep[1], keep[2], keep[3]};
        rules.cst[e] = uint4{cst[0], cst[1], cst[2], cst[3]};
    }
}

// The decisions of launch_tokens_bp8 as functions of the shape alone -- shared by the launcher and by bsq_tokenize_kernel_name (ADVICE round 5:
// bench.py used to re-implement them and could report a kernel that was not the one launched).
int64_t tokens_bp8_chunks(int64_t B, int64_t P) { return (B * ((P + 15) / 16) + kChunk / 16 - 1) / (kChunk / 16); }
bool tokens_bp8_fast_form(const bsq_desc *d, int64_t B, int64_t P, bool aligned_out) {
    uint32_t tab[8];
    const Tuning &tn = tuning();
    int lk = tn.tokens8_lookup;
    const bool foldable = fold_table(d->lut, tab, 0u);
    if (lk == 0) lk = foldable ? 2 : 1;
    if (lk == 2 && !foldable) lk = 1;
    return lk == 2 && P % 16 == 0 && aligned_out && !tn.wide_index && tokens_bp8_chunks(B, P) < (int64_t(1) << 23) && B < (int64_t(1) << 31) &&
           tn.tokens8_fast != 1;
}
// fused augmentation + tokens: the no-wait form (k_augment_tokens_nowait + k_patch_tokens) or the flag form (k_augment_tokens_fused)
bool tokens_bp8_nowait_form(int64_t nchunks) { return (tuning().augment_fused == 0 && nchunks <= 16384) || tuning().augment_fused == 4; }

bool tokens_bp8_applicable(const bsq_desc *d, int64_t B, int64_t P, const void *out) {
    (void)d;
    (void)out;  // any alignment: P % 16 != 0 or a misaligned output take the kernel's row-piece form (RG)
    return B > 0 && P >= 128 && P <= (int64_t(1) << 30) && bsq_alphabet_size(d) <= 250 && B * P < (int64_t(1) << 51);
}

// Per (device, stream) state of the fused augmentation + token launch: flag words (zeroed once, then every launch publishes its own
// epoch).  kFusedSlots (device, stream) pairs are kept; one more evicts the least recently used.
struct FusedSlot {
    int dev;
    hipStream_t stream;
    uint32_t *buf;
    size_t flag_words;
    uint32_t epoch;
    uint64_t last_use;
};
constexpr int kFusedSlots = 16;
static FusedSlot g_fused[kFusedSlots] = {};
static uint64_t g_fused_clock = 0;
static std::mutex g_fused_mu;
// Token waves of fused launches that gave up waiting, counted by the device in HOST memory (pinned, mapped, coherent): the count
// survives every reallocation above, is readable without a synchronisation, and is sticky until fused_status_clear().
static uint32_t *g_fused_failures = nullptr;
static bool g_fused_failures_tried = false;

static uint32_t *fused_failures_word() {  // (g_fused_mu held)
    if (!g_fused_failures_tried) {
        g_fused_failures_tried = true;
        void *p = nullptr;
        if (hipHostMalloc(&p, 64, hipHostMallocPortable | hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess && p) {
            std::memset(p, 0, 64);
            g_fused_failures = static_cast<uint32_t *>(p);
        } else {
            (void)hipGetLastError();
        }
    }
    return g_fused_failures;
}

// *buf = nullptr: no fused launch this time (allocation refused, e.g. under another thread's global-mode stream capture): the caller
// runs the two launches.
static bsq_status fused_acquire(hipStream_t s, size_t flag_words, uint32_t **flags, uint32_t **failures, uint32_t *epoch) {
    std::lock_guard<std::mutex> lock(g_fused_mu);
    *flags = *failures = nullptr;
    uint32_t *fw = fused_failures_word();
    if (!fw) return BSQ_OK;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return set_hip_error("hipGetDevice", e);
    FusedSlot *slot = nullptr;
    for (FusedSlot &f : g_fused)
        if (f.buf && f.dev == dev && f.stream == s) slot = &f;
    if (!slot)
        for (FusedSlot &f : g_fused)
            if (!f.buf) { slot = &f; break; }
    if (!slot) {  // every slot holds another (device, stream): the least recently used one goes
        slot = &g_fused[0];
        for (FusedSlot &f : g_fused)
            if (f.last_use < slot->last_use) slot = &f;
        int cur = dev;
        if (slot->dev != cur) (void)hipSetDevice(slot->dev);
        (void)hipFree(slot->buf);  // (synchronises that device: nothing in flight still uses it)
        if (slot->dev != cur) (void)hipSetDevice(cur);
        slot->buf = nullptr;
    }
    if (!slot->buf || slot->flag_words < flag_words || slot->epoch >= 0x0FFFFFFEu) {  // (a flag word is epoch << 4 | who-published: 28 bits of epoch)
        if (slot->buf) (void)hipFree(slot->buf);  // (synchronises the device: nothing in flight still reads it)
        slot->buf = nullptr;
        const size_t fcap = flag_words < 4096 ? 4096 : flag_words + flag_words / 2;
        e = hipMalloc(reinterpret_cast<void **>(&slot->buf), fcap * sizeof(uint32_t));
        if (e == hipSuccess) {
            e = hipMemsetAsync(slot->buf, 0, fcap * sizeof(uint32_t), s);  // on the launch's stream: ordered before its first use
            if (e != hipSuccess) {
                (void)hipFree(slot->buf);
                slot->buf = nullptr;
            }
        } else {
            slot->buf = nullptr;
        }
        if (e != hipSuccess) {
            (void)hipGetLastError();
            slot->flag_words = 0;
            return BSQ_OK;
        }
        slot->dev = dev;
        slot->stream = s;
        slot->flag_words = fcap;
        slot->epoch = 0;
    }
    slot->last_use = ++g_fused_clock;
    *flags = slot->buf;
    *failures = fw;
    *epoch = ++slot->epoch;
    return BSQ_OK;
}

uint32_t fused_failures() {  // no synchronisation: what the device has counted so far
    std::lock_guard<std::mutex> lock(g_fused_mu);
    return g_fused_failures ? __atomic_load_n(g_fused_failures, __ATOMIC_RELAXED) : 0u;
}
void fused_failures_clear() {
    std::lock_guard<std::mutex> lock(g_fused_mu);
    if (g_fused_failures) __atomic_store_n(g_fused_failures, 0u, __ATOMIC_RELAXED);
}

bsq_status launch_tokens_bp8(const bsq_desc *d, const uint8_t *chars, const int64_t *offsets, int64_t B, int64_t P,
                             void *out, hipStream_t s, bool raw, const uint8_t *mask, const FusedAugRequest *fuse, bool *fused_taken) {
    if (fused_taken) *fused_taken = false;
    if (mask && (!raw || P % 16 != 0 || reinterpret_cast<uintptr_t>(out) % 16 != 0))
        return set_error(BSQ_ERR_INVALID_ARG, "k_tokens_bp8: a mask needs raw-id mode and an aligned shape");
    T8Params c;
    c.mask = mask;
    for (int i = 0; i < 256; ++i) c.lut[i] = d->lut[i];
    c.none_v = raw ? 0xFFu : 0u;  // raw: ids with BSQ_NO_TOKEN where the one-hot row is all zero (plain stores: re-read at once)
    const bool foldable = fold_table(d->lut, c.tab, c.none_v);
    c.chars = chars;
    c.offsets = offsets;
    c.out = static_cast<uint8_t *>(out);
    c.B = B;
    c.P = uint32_t(P);
    c.total = B * P;
    c.ppr = uint32_t((P + 15) / 16);
    c.nchunks = (B * int64_t(c.ppr) + kChunk / 16 - 1) / (kChunk / 16);  // 256 pieces per wave
    c.inv_ppr = 1.0 / double(c.ppr);
    div_constants(c.ppr, &c.magic, &c.shift, &c.pow2);
    c.bos = d->bos;
    const int64_t room = P - d->bos - d->eos;
    c.room = int32_t(room < 0 ? 0 : room);
    c.bos_id = uint32_t(bsq_bos_id(d)) & 0xFFu;
    const uint32_t fill = d->padchar ? uint32_t(bsq_pad_id(d)) : c.none_v;  // no padchar: the memset 0 of tokenize.h:427 stays
    c.fill_v = fill;
    c.at_len_v = d->eos ? uint32_t(bsq_eos_id(d)) : fill;
    c.abl = 0;
    c.wide_index = tuning().wide_index;
    int lk = tuning().tokens8_lookup;  // 0 automatic (registers when the table folds), 1 LDS byte table, 2 registers
    if (lk == 0) lk = foldable ? 2 : 1;
    if (lk == 2 && !foldable) lk = 1;
    const int64_t groups = ((c.nchunks + 7) / 8 + 3) / 4;
    if (groups * 8 >= (int64_t(1) << 31)) return set_error(BSQ_ERR_INVALID_ARG, "output too large");
    const dim3 grid(unsigned(groups * 8));
    const int padv = tuning().tokens8_pad;  // unused dynamic LDS = occupancy cap (experiments)
    const size_t pad = padv > 0 ? size_t(padv) : 0;
    const bool nt = nontemporal_stores() && !raw;
    // the fast form: register table, aligned rows, no mask, 32-bit piece indices (knob tokens8_fast = 1: never)
    if (lk == 2 && !mask && c.abl == 0 && P % 16 == 0 && reinterpret_cast<uintptr_t>(out) % 16 == 0 && !c.wide_index &&
        c.nchunks < (int64_t(1) << 23) && B < (int64_t(1) << 31) && tuning().tokens8_fast != 1) {
        uint32_t magic = c.magic, shift = c.shift;
        if (c.pow2) {  // d = 2^s, s >= 3: mulhi(n, 2^(32 - s)) == n >> s
            magic = uint32_t(1) << (32 - c.shift);
            shift = 0;
        }
        const uint32_t packed = uint32_t(c.bos != 0) | (raw ? 2u : 0u) | (c.bos_id << 8) | ((c.at_len_v & 0xFFu) << 16) | ((c.fill_v & 0xFFu) << 24);
        T8Tab tab;
        for (int i = 0; i < 8; ++i) tab.t[i] = c.tab[i];
        T8Rules rules;
        build_rules(c.fill_v, c.at_len_v, rules);
        if (fuse && fused_taken && tuning().augment_fused != 1 && pad == 0) {  // augmentation in the same launch (see k_augment_tokens_fused)
            hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
            (void)hipStreamIsCapturing(s, &cap);
            const int64_t aug_blocks = ((B + 255) / 256 + 7) / 8 * 8;
            if (cap == hipStreamCaptureStatusNone && 2 * aug_blocks + int64_t(grid.x) < (int64_t(1) << 31)) {  // (a replayed graph would replay the epoch)
                const void *atab = nullptr;
                bsq_status st = augment_device_table(&atab);
                if (st != BSQ_OK) return st;
                // The no-wait form (round 5, see k_augment_tokens_nowait; knob "augment_fused" 2 / 3: the flag forms instead)
                // Automatic (knob 0): up to 16 384 chunks (65 536 sequences of cfg5's shape: 15.7 vs 18.6 us; the 1/8 shard 13.7 vs 16.8; a
                // loader batch of 4096: 12.5 vs 14.1) -- beyond that the mutations' own work no longer hides beside the stream and the patch
                // launch comes on top (262 144 sequences: 40.5 + 6.4 us against 41 for the flag form, which cfg5 itself therefore keeps).
                // Knob 4: whatever the size.
                if (tokens_bp8_nowait_form(c.nchunks)) {
                    std::lock_guard<std::mutex> scratch_turn(workspace_mutex());  // (the scratch is shared by the calls of a stream: both launches back to back)
                    void *ws = nullptr;
                    const int32_t chain = fuse->chain_len > 0 ? fuse->chain_len : 1;
                    st = workspace_acquire(size_t(B) * size_t(chain) * sizeof(uint2), s, &ws);
                    if (st != BSQ_OK) return st;
                    FusedAug fa;
                    fa.chars = fuse->chars;
                    fa.B = B;
                    fa.tab = static_cast<const bsq_aug::AugTable *>(atab);
                    fa.frac = fuse->frac;
                    fa.seed = fuse->seed;
                    fa.chain_len = fuse->chain_len;
                    fa.aug_blocks = uint32_t(aug_blocks);
                    fa.flags = nullptr;
                    fa.wait = FusedWait{};
                    fa.rec = static_cast<uint2 *>(ws);
                    const bool eosv_f = (c.at_len_v & 0xFFu) != (c.fill_v & 0xFFu);
                    const dim3 fgrid(unsigned(aug_blocks + int64_t(grid.x)));
#define BSQ_NOWAIT(NTV, EV)                                                                                                              \
    hipLaunchKernelGGL((k_augment_tokens_nowait<NTV, 4, EV>), fgrid, dim3(kThreads), 0, s, offsets, chars, c.out, uint32_t(c.nchunks),   \
                       uint32_t(B), c.ppr, magic, shift, c.room, packed, tab, rules, fa)
                    if (nt) { if (eosv_f) BSQ_NOWAIT(true, true); else BSQ_NOWAIT(true, false); }
                    else { if (eosv_f) BSQ_NOWAIT(false, true); else BSQ_NOWAIT(false, false); }
#undef BSQ_NOWAIT
                    hipError_t en = hipGetLastError();
                    if (en == hipSuccess && fuse->chain_len > 0) {
                        PatchLut pl;
                        for (int i = 0; i < 128; ++i) pl.v[i] = d->lut[i] >= 0 ? uint8_t(d->lut[i]) : uint8_t(c.none_v);
                        hipLaunchKernelGGL(k_patch_tokens, dim3(unsigned((B + kThreads - 1) / kThreads)), dim3(kThreads), 0, s, static_cast<const uint2 *>(ws), B,
                                           fuse->chain_len, P, int32_t(c.bos), c.room, pl, c.out);
                        en = hipGetLastError();
                    }
                    workspace_release(ws, s);
                    if (en != hipSuccess) return set_hip_error("k_augment_tokens_nowait / k_patch_tokens", en);
                    *fused_taken = true;
                    return BSQ_OK;
                }
                // Same-XCD form (round 5; knob "augment_fused" = 2: never): padlen divides 4096 (a chunk is 4096 / padlen WHOLE rows) and
                // the dispatcher deals blocks round-robin over the XCDs (probed once per device): every chunk's rows are mutated by an
                // augmentation wave of the chunk's own class, the hand-off stays inside one L2 (see tokens_fast_body, FLAGS == 2)
                const uint32_t rows_per_chunk = uint32_t(P) <= 4096u && 4096u % uint32_t(P) == 0 ? 4096u / uint32_t(P) : 0u;
                // (measured, profiles/r05/aug_same_xcd_lab.txt: 1 us ahead up to 32 768 sequences of cfg5's shape, 2.6 us BEHIND at 262 144 -- the
                //  augmentation role's rows are then eight runs of eight sequences per wave and its offsets loads gathers; knob 3: whenever it applies)
                const bool same_xcd = rows_per_chunk >= 1 && rows_per_chunk <= 64 && tuning().augment_fused == 3 && bsq_xcd_round_robin() == 1;
                uint32_t cw = 0;
                int64_t aug_blocks_x = aug_blocks;
                if (same_xcd) {
                    cw = 64u / rows_per_chunk;                                        // chunks per augmentation wave
                    const int64_t per_class = (c.nchunks + 7) / 8;
                    const int64_t aug_waves = (per_class + cw - 1) / cw;             // per class
                    aug_blocks_x = (aug_waves + 3) / 4 * 8;
                }
                uint32_t *flags = nullptr, *failures = nullptr, epoch = 0;
                st = fused_acquire(s, size_t(aug_blocks_x) * 4, &flags, &failures, &epoch);
                if (st != BSQ_OK) return st;
                if (!flags) return BSQ_OK;  // (fused_taken stays false)
                FusedAug fa;
                fa.chars = fuse->chars;
                fa.B = B;
                fa.tab = static_cast<const bsq_aug::AugTable *>(atab);
                fa.frac = fuse->frac;
                fa.seed = fuse->seed;
                fa.chain_len = fuse->chain_len;
                fa.aug_blocks = uint32_t(aug_blocks_x);
                fa.rec = nullptr;
                fa.flags = flags;
                fa.wait.flags = flags;
                fa.wait.failures = failures;
                fa.wait.epoch = epoch;
                fa.wait.chunks_per_aug_wave = cw;
                // knob "fused_spins" (fault injection for the tests): polls before a token wave gives up; 0 = 2^18 (~1 s)
                fa.wait.spins = tuning().fused_spins > 0 ? uint32_t(tuning().fused_spins) : (1u << 18);
                // s_sleep(2) naps between two polls of a waiting token wave: 60 (~3.5 us) when 8192 chunk waves poll at once (cfg5), fewer
                // for smaller launches -- the same polling traffic chip-wide, and a 1/8 shard (4096 chunks) no longer naps through half of its run
                const int64_t nap = c.nchunks / 136;
                fa.wait.naps = uint32_t(nap < 6 ? 6 : (nap > 60 ? 60 : nap));
                const dim3 fgrid(unsigned(aug_blocks_x + int64_t(grid.x)));
#define BSQ_FUSED(NTV, SX)                                                                                                          \
    hipLaunchKernelGGL((k_augment_tokens_fused<NTV, 4, SX>), fgrid, dim3(kThreads), 0, s, offsets, chars, c.out, uint32_t(c.nchunks), \
                       uint32_t(B), c.ppr, magic, shift, c.room, packed, tab, rules, fa)
                if (nt) { if (same_xcd) BSQ_FUSED(true, true); else BSQ_FUSED(true, false); }
                else { if (same_xcd) BSQ_FUSED(false, true); else BSQ_FUSED(false, false); }
#undef BSQ_FUSED
                const hipError_t eff = hipGetLastError();
                if (eff != hipSuccess) return set_hip_error("k_augment_tokens_fused", eff);
                *fused_taken = true;
                return BSQ_OK;
            }
        }
        if (fuse) return BSQ_OK;  // the caller runs the two launches (fused_taken stays false; nothing was launched)
        const bool eosv = (c.at_len_v & 0xFFu) != (c.fill_v & 0xFFu);
#define BSQ_T8F(NTV, EV)                                                                                                                 \
    hipLaunchKernelGGL((k_tokens_bp8_fast<NTV, EV>), grid, dim3(kThreads), pad, s, offsets, chars, c.out, uint32_t(c.nchunks), uint32_t(B), \
                       c.ppr, magic, shift, c.room, packed, tab, rules)
        if (nt) { if (eosv) BSQ_T8F(true, true); else BSQ_T8F(true, false); }
        else { if (eosv) BSQ_T8F(false, true); else BSQ_T8F(false, false); }
#undef BSQ_T8F
        const hipError_t ef = hipGetLastError();
        if (ef != hipSuccess) return set_hip_error("k_tokens_bp8_fast", ef);
        return BSQ_OK;
    }
    if (fuse) return BSQ_OK;  // no fused form for this shape: the caller runs the two launches
#define BSQ_T8(NTV, LKV) launch_variant<NTV, LKV>(c, grid, pad, s)
    if (lk == 2) { if (nt) BSQ_T8(true, 1); else BSQ_T8(false, 1); }
    else { if (nt) BSQ_T8(true, 0); else BSQ_T8(false, 0); }
#undef BSQ_T8
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_hip_error("k_tokens_bp8", e);
    return BSQ_OK;
}

// (P,B) tokens of any element type (pitch = ELEMENTS between two position rows; the final matrix: pitch = B) through
// k_tokens_pb8_fast: rows and output 16-byte aligned.
bool tokens_pb8_applicable(const bsq_desc *d, int64_t B, int64_t P, const void *out, int64_t pitch, bsq_dtype t) {
    if (tuning().tokens_pb8 == 1) return false;
    const int64_t sz = int64_t(bsq_dtype_size(t));
    if (sz == 0) return false;
    // 4-byte elements: measured slower than k_tokenize_tile (cfg2: 51 vs 48-50 us), 8-byte elements 3-4 % faster (100 vs 103.5 us) once the
    // blocks walk sequence tiles fastest (profiles/r03/pb8_wider_types.txt) -- 4-byte types only under knob tokens_pb8 = 2 (tests, measurement)
    if (sz == 4 && tuning().tokens_pb8 != 2) return false;
    // rows / outputs that are only element-aligned: the UA form (1- and 2-byte integers; its stores are cut at the output's
    // 16-byte lines).  65 537 x 1024 int8 42.8 -> 27.5 us, 65 000 x 1024 int8 30.2 -> 25.0, 100 001 x 512 int16 34.0 -> 28.8
    // (profiles/r03/pb8_unaligned_rows.txt).  Knob tokens_pb8 = 3: aligned shapes only.
    // (a column block -- pitch > B, final matrix -- must also END on a 16-byte line for the aligned form: its last store is whole)
    const bool aligned = (pitch * sz) % 16 == 0 && reinterpret_cast<uintptr_t>(out) % 16 == 0 && (B * sz) % 16 == 0;
    if (!aligned && (sz > 2 || tuning().tokens_pb8 == 3 || reinterpret_cast<uintptr_t>(out) % uintptr_t(sz) != 0)) return false;
    return B > 0 && P >= 1 && P <= (int64_t(1) << 30) && B < (int64_t(1) << 31) - 4096 && pitch >= B && pitch < (int64_t(1) << 31) &&
           bsq_alphabet_size(d) <= 250;
}

bsq_status launch_tokens_pb8(const bsq_desc *d, const uint8_t *chars, const int64_t *offsets, int64_t B, int64_t P, void *out,
                             int64_t pitch, hipStream_t s, bool raw, bsq_dtype t, bool nib, int64_t tt0, int64_t ntt_count) {
    const uint32_t none_v = raw ? 0xFFu : 0u;
    if (raw && t != BSQ_I8) return set_error(BSQ_ERR_INVALID_ARG, "raw ids are bytes");
    if (nib && (!raw || pitch % 32 != 0 || reinterpret_cast<uintptr_t>(out) % 16 != 0 || bsq_alphabet_size(d) > 15))
        return set_error(BSQ_ERR_INVALID_ARG, "nibble ids: raw mode, at most 15 classes, rows of a multiple of 32 ids");
    T8Tab tab;
    const bool foldable = fold_table(d->lut, tab.t, none_v);
    // knob "tokens8_lookup": 0 automatic (registers when the table folds), 1 LDS byte table, 2 registers
    int lk = tuning().tokens8_lookup;
    if (lk == 0) lk = foldable ? 2 : 1;
    if (lk == 2 && !foldable) lk = 1;
    T8Lut lut;
    for (int w = 0; w < 64; ++w) {
        uint32_t v = 0;
        for (int k = 0; k < 4; ++k) {
            const int c = 4 * w + k;
            const uint32_t tkn = (c < 128 && d->lut[c] >= 0) ? uint32_t(uint8_t(d->lut[c])) : none_v;
            v |= tkn << (8 * k);
        }
        lut.w[w] = v;
    }
    const uint32_t fill = d->padchar ? uint32_t(bsq_pad_id(d)) : none_v;  // no padchar: the memset 0 of tokenize.h:427 stays
    const uint32_t at_len = d->eos ? uint32_t(bsq_eos_id(d)) : fill;
    const uint32_t bos_id = uint32_t(bsq_bos_id(d)) & 0xFFu;
    T8Rules rules;
    build_rules(fill, at_len, rules);
    const int64_t room64 = P - d->bos - d->eos;
    const int32_t room = int32_t(room64 < 0 ? 0 : room64);
    // bit 2: every XCD walks its own contiguous range of sequence tiles (rows that are not 64-byte aligned: the sectors two
    // neighbouring tiles share are then merged in ONE L2 -- 65 000 x 1024 int16: 49.6 -> 38.9 us, profiles/r03/pb8_unaligned_rows.txt)
    // (also for 4- / 8-byte elements: with the position tiles of a sequence tile back to back they took 53 / 111 us instead of 51 / 100)
    const bool contig = (pitch * int64_t(bsq_dtype_size(t))) % 64 != 0 || reinterpret_cast<uintptr_t>(out) % 64 != 0 || bsq_dtype_size(t) >= 4;
    uint32_t packed = uint32_t(d->bos != 0) | (raw ? 2u : 0u) | (contig ? 4u : 0u) | (bos_id << 8) | ((at_len & 0xFFu) << 16) | ((fill & 0xFFu) << 24);
    const bool nt = nontemporal_stores() && !raw;  // the raw matrix is re-read by the expansion right away
    // one tile shape: 256 sequences x 64 positions (512 x 64 and 512 x 32 were built and measured slower:
    // profiles/r03/pb8_coalesced_ab.txt, pb8_512x32_tile_lost.txt)
    const int TB = 256, TT = 64;
    const int64_t ntb = (B + TB - 1) / TB, ntt_all = (P + TT - 1) / TT;
    if (tt0 < 0 || tt0 >= ntt_all || ntt_count < 0 || tt0 + ntt_count > ntt_all) return set_error(BSQ_ERR_INVALID_ARG, "position tiles of the slice");
    const int64_t ntt = ntt_count > 0 ? ntt_count : ntt_all - tt0;  // the launch's position tiles: tt0 .. tt0 + ntt - 1 (`out` = the slice's first row)
    // the address row 0 of the matrix would have (the kernel indexes rows by their absolute position)
    out = static_cast<uint8_t *>(out) - ((tt0 * TT * pitch * int64_t(bsq_dtype_size(t))) >> (nib ? 1 : 0));
    const int64_t blocks = (ntb + 7) / 8 * 8 * ntt;
    if (blocks >= (int64_t(1) << 31)) return set_error(BSQ_ERR_INVALID_ARG, "output too large");
    const bool ua = !((pitch * int64_t(bsq_dtype_size(t))) % 16 == 0 && reinterpret_cast<uintptr_t>(out) % 16 == 0 &&
                      (raw || (B * int64_t(bsq_dtype_size(t))) % 16 == 0));
    uint32_t magic = 0, shift = 0, pow2 = 0;
    div_constants((contig || ua) ? uint32_t((ntb + 7) / 8) : uint32_t(ntt), &magic, &shift, &pow2);  // the divisor of the block index
    if (pow2) magic = 0;  // the kernel shifts (magic 0 marks a power of two)
    // paired tail (see the kernel): the last position tile holds 1 ... 32 positions, the aligned class-pinned form, more than one group of 8 sequence
    // tiles (knob "tokens_pb8_pair" = 1: never)
    if (!contig && !ua && P % TT != 0 && P % TT <= 32 && ntb > 8 && tuning().tokens_pb8_pair != 1) packed |= 8u;
#define BSQ_PB8U(NTV, LKV, SZV, FLTV, UAV)                                                                                             \
    hipLaunchKernelGGL((k_tokens_pb8_fast<NTV, 256, LKV, SZV, FLTV, UAV>), dim3(unsigned(blocks)), dim3(kThreads), 0, s, offsets, chars,  \
                       static_cast<uint8_t *>(out), pitch, uint32_t(B), uint32_t(P), uint32_t(ntb), uint32_t(ntt), magic, shift,    \
                       room, packed, uint32_t(tt0), tab, rules, lut)
#define BSQ_PB8(NTV, LKV, SZV, FLTV)                                       \
    do {                                                                   \
        if constexpr (SZV <= 2 && !FLTV) {                                 \
            if (ua) BSQ_PB8U(NTV, LKV, SZV, FLTV, true);                   \
            else BSQ_PB8U(NTV, LKV, SZV, FLTV, false);                     \
        } else {                                                           \
            BSQ_PB8U(NTV, LKV, SZV, FLTV, false);                          \
        }                                                                  \
    } while (0)
#define BSQ_PB8_T(SZV, FLTV)                                                                   \
    do {                                                                                       \
        if (lk == 2) { if (nt) BSQ_PB8(true, 1, SZV, FLTV); else BSQ_PB8(false, 1, SZV, FLTV); } \
        else { if (nt) BSQ_PB8(true, 0, SZV, FLTV); else BSQ_PB8(false, 0, SZV, FLTV); }         \
    } while (0)
    if (nib) {  // (raw: plain stores, aligned rows)
#define BSQ_PB8N(LKV)                                                                                                                     \
    hipLaunchKernelGGL((k_tokens_pb8_fast<false, 256, LKV, 1, false, false, true>), dim3(unsigned(blocks)), dim3(kThreads), 0, s, offsets, \
                       chars, static_cast<uint8_t *>(out), pitch, uint32_t(B), uint32_t(P), uint32_t(ntb), uint32_t(ntt), magic, shift,   \
                       room, packed, uint32_t(tt0), tab, rules, lut)
        if (lk == 2) BSQ_PB8N(1); else BSQ_PB8N(0);
#undef BSQ_PB8N
        const hipError_t en = hipGetLastError();
        if (en != hipSuccess) return set_hip_error("k_tokens_pb8_fast<nibbles>", en);
        return BSQ_OK;
    }
    switch (t) {
    case BSQ_I8: BSQ_PB8_T(1, false); break;
    case BSQ_I16: BSQ_PB8_T(2, false); break;
    case BSQ_I32: BSQ_PB8_T(4, false); break;
    case BSQ_F32: BSQ_PB8_T(4, true); break;
    case BSQ_U64: BSQ_PB8_T(8, false); break;
    case BSQ_F64: BSQ_PB8_T(8, true); break;
    default: return set_error(BSQ_ERR_DTYPE, "bad bsq_dtype");
    }
#undef BSQ_PB8_T
#undef BSQ_PB8
#undef BSQ_PB8U
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_hip_error("k_tokens_pb8_fast", e);
    return BSQ_OK;
}

// What the batches of a multi-batch (B,P) int8 launch share (k_tokens_bp8_fast's scalar arguments), and their per-batch table.
struct Bp8Shared {
    uint32_t ppr, magic, shift, packed;
    int32_t room;
    T8Tab tab;
    T8Rules rules;
    bool eosv, nt;
};
// false: the fast (B,P) int8 kernel does not take this tokenizer / padlen (or a knob says no)
static bool bp8_shared_setup(const bsq_desc *d, int64_t P, Bp8Shared &c) {
    const Tuning &tn = tuning();
    if (bsq_alphabet_size(d) > 250 || P < 128 || P % 16 != 0 || P > (int64_t(1) << 30)) return false;
    if (tn.tokenize_path == 1 || tn.wide_index || tn.tokens8_pad > 0 || tn.tokens8 == 1 || tn.tokens8_fast == 1) return false;
    const bool foldable = fold_table(d->lut, c.tab.t, 0u);
    if (!foldable || tn.tokens8_lookup == 1) return false;  // (the LDS byte-table form has no fast kernel)
    const uint32_t fill = d->padchar ? uint32_t(bsq_pad_id(d)) : 0u;
    const uint32_t at_len = d->eos ? uint32_t(bsq_eos_id(d)) : fill;
    const uint32_t bos_id = uint32_t(bsq_bos_id(d)) & 0xFFu;
    build_rules(fill, at_len, c.rules);
    const int64_t room64 = P - d->bos - d->eos;
    c.room = int32_t(room64 < 0 ? 0 : room64);
    c.nt = nontemporal_stores();
    c.ppr = uint32_t(P / 16);
    uint32_t pow2 = 0;
    div_constants(c.ppr, &c.magic, &c.shift, &pow2);
    if (pow2) {  // d = 2^s, s >= 3: mulhi(n, 2^(32 - s)) == n >> s
        c.magic = uint32_t(1) << (32 - c.shift);
        c.shift = 0;
    }
    c.packed = uint32_t(d->bos != 0) | (bos_id << 8) | ((at_len & 0xFFu) << 16) | ((fill & 0xFFu) << 24);
    c.eosv = (at_len & 0xFFu) != (fill & 0xFFu);
    return true;
}
static void multi_clear(T8Multi &m) {
    for (int i = 0; i < kMultiMax; ++i) {
        m.offsets[i] = nullptr, m.chars[i] = nullptr, m.out[i] = nullptr;
        m.first_block[i] = 0xFFFFFFFFu, m.units[i] = 0, m.B[i] = 0;
    }
}
// false: some batch does not qualify (alignment, size); *blocks = the grid of the n batches' chunk streams
static bool bp8_multi_table(const bsq_desc *d, int32_t n, const bsq_batch *bt, int64_t P, uint32_t ppr, T8Multi &m, int64_t *blocks) {
    multi_clear(m);
    *blocks = 0;
    for (int i = 0; i < n; ++i) {
        const int64_t B = bt[i].B;
        if (B <= 0 || B >= (int64_t(1) << 31) || !tokens_bp8_applicable(d, B, P, bt[i].out) || reinterpret_cast<uintptr_t>(bt[i].out) % 16 != 0) return false;
        const int64_t nchunks = (B * int64_t(ppr) + kChunk / 16 - 1) / (kChunk / 16);
        if (nchunks >= (int64_t(1) << 23)) return false;
        m.offsets[i] = bt[i].offsets, m.chars[i] = bt[i].chars, m.out[i] = static_cast<uint8_t *>(bt[i].out);
        m.first_block[i] = uint32_t(*blocks), m.units[i] = uint32_t(nchunks), m.B[i] = uint32_t(B);
        *blocks += ((nchunks + 7) / 8 + 3) / 4 * 8;
    }
    return *blocks < (int64_t(1) << 31);
}
static bsq_status launch_bp8_multi(const Bp8Shared &c, const T8Multi &m, int64_t blocks, hipStream_t s) {
#define BSQ_T8M(NTV, EV) \
    hipLaunchKernelGGL((k_tokens_bp8_fast_multi<NTV, EV>), dim3(unsigned(blocks)), dim3(kThreads), 0, s, c.ppr, c.magic, c.shift, c.room, c.packed, c.tab, c.rules, m)
    if (c.nt) { if (c.eosv) BSQ_T8M(true, true); else BSQ_T8M(true, false); }
    else { if (c.eosv) BSQ_T8M(false, true); else BSQ_T8M(false, false); }
#undef BSQ_T8M
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_hip_error("k_tokens_bp8_fast_multi", e);
    return BSQ_OK;
}

// n <= kMultiMax independent batches of one tokenizer, padlen, layout and element type in ONE launch (bsq_tokenize_device_multi).  *taken =
// false and nothing launched when some batch does not qualify for the fast kernel of its layout (the caller then issues the batches
// one after the other): (B,P): int8, padlen % 16 == 0 and >= 128, 16-byte aligned outputs, a foldable alphabet; (P,B): 1- / 2-byte
// integers, B * sz % 64 == 0 and 64-byte aligned outputs (the class-pinned tile order whose block-index divisor the batches share).
bsq_status launch_tokens_multi(const bsq_desc *d, int32_t n, const bsq_batch *bt, int64_t P, bool batch_first, bsq_dtype t, hipStream_t s,
                               bool *taken) {
    *taken = false;
    if (n < 1 || n > kMultiMax || bsq_alphabet_size(d) > 250 || P <= 0 || P > (int64_t(1) << 30)) return BSQ_OK;
    const Tuning &tn = tuning();
    if (tn.tokenize_path == 1 || tn.wide_index || tn.tokens8_pad > 0) return BSQ_OK;
    const size_t sz = bsq_dtype_size(t);
    if (batch_first) {
        Bp8Shared c;
        T8Multi m;
        int64_t blocks = 0;
        if (t != BSQ_I8 || !bp8_shared_setup(d, P, c) || !bp8_multi_table(d, n, bt, P, c.ppr, m, &blocks)) return BSQ_OK;
        const bsq_status st = launch_bp8_multi(c, m, blocks, s);
        if (st != BSQ_OK) return st;
        *taken = true;
        return BSQ_OK;
    }
    const uint32_t none_v = 0u;
    T8Tab tab;
    const bool foldable = fold_table(d->lut, tab.t, none_v);
    int lk = tn.tokens8_lookup;  // 0 automatic (registers when the table folds), 1 LDS byte table, 2 registers
    if (lk == 0) lk = foldable ? 2 : 1;
    if (lk == 2 && !foldable) lk = 1;
    const uint32_t fill = d->padchar ? uint32_t(bsq_pad_id(d)) : none_v;
    const uint32_t at_len = d->eos ? uint32_t(bsq_eos_id(d)) : fill;
    const uint32_t bos_id = uint32_t(bsq_bos_id(d)) & 0xFFu;
    T8Rules rules;
    build_rules(fill, at_len, rules);
    const int64_t room64 = P - d->bos - d->eos;
    const int32_t room = int32_t(room64 < 0 ? 0 : room64);
    const bool nt = nontemporal_stores();
    T8Multi m;
    multi_clear(m);
    int64_t blocks = 0;
    if (sz > 2 || (t != BSQ_I8 && t != BSQ_I16) || tn.tokens_pb8 == 1 || tn.tokens_pb8 == 3) return BSQ_OK;
    const int TB = 256, TT = 64;
    const int64_t ntt = (P + TT - 1) / TT;
    for (int i = 0; i < n; ++i) {
        const int64_t B = bt[i].B;
        if (B <= 0 || !tokens_pb8_applicable(d, B, P, bt[i].out, B, t) || (B * int64_t(sz)) % 64 != 0 || reinterpret_cast<uintptr_t>(bt[i].out) % 64 != 0)
            return BSQ_OK;
        const int64_t ntb = (B + TB - 1) / TB;
        m.offsets[i] = bt[i].offsets, m.chars[i] = bt[i].chars, m.out[i] = static_cast<uint8_t *>(bt[i].out);
        m.first_block[i] = uint32_t(blocks), m.units[i] = uint32_t(ntb), m.B[i] = uint32_t(B);
        blocks += (ntb + 7) / 8 * 8 * ntt;
    }
    if (blocks >= (int64_t(1) << 31)) return BSQ_OK;
    T8Lut lut;
    for (int w = 0; w < 64; ++w) {
        uint32_t v = 0;
        for (int k = 0; k < 4; ++k) {
            const int c = 4 * w + k;
            v |= ((c < 128 && d->lut[c] >= 0) ? uint32_t(uint8_t(d->lut[c])) : none_v) << (8 * k);
        }
        lut.w[w] = v;
    }
    uint32_t magic = 0, shift = 0, pow2 = 0;
    div_constants(uint32_t(ntt), &magic, &shift, &pow2);
    if (pow2) magic = 0;  // the kernel shifts (magic 0 marks a power of two)
    const uint32_t packed = uint32_t(d->bos != 0) | (bos_id << 8) | ((at_len & 0xFFu) << 16) | ((fill & 0xFFu) << 24);
#define BSQ_PB8M(NTV, LKV, SZV)                                                                                                            \
    hipLaunchKernelGGL((k_tokens_pb8_fast_multi<NTV, LKV, SZV>), dim3(unsigned(blocks)), dim3(kThreads), 0, s, uint32_t(P), uint32_t(ntt), magic, \
                       shift, room, packed, tab, rules, lut, m)
#define BSQ_PB8M_T(SZV)                                                              \
    do {                                                                             \
        if (lk == 2) { if (nt) BSQ_PB8M(true, 1, SZV); else BSQ_PB8M(false, 1, SZV); } \
        else { if (nt) BSQ_PB8M(true, 0, SZV); else BSQ_PB8M(false, 0, SZV); }         \
    } while (0)
    if (sz == 1) BSQ_PB8M_T(1); else BSQ_PB8M_T(2);
#undef BSQ_PB8M_T
#undef BSQ_PB8M
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_hip_error("k_tokens_pb8_fast_multi", e);
    *taken = true;
    return BSQ_OK;
}

}  // namespace bsq_internal
