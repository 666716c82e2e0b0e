// Shared by the one-hot and the token translation units of libbsq_hip.so (bsq_onehot.hip, bsq_tokens.hip; gfx950 only): the launch
// parameters of the TILED kernels, the pieces every tile kernel is made of (tile order, BOS / EOS / PAD rule, character fetch + lookup,
// sequence spans, the LDS token tile), the raw-id / token-value tile kernel both units launch (k_tokens_raw), and the host-side launch
// helpers.  Everything sits in an anonymous namespace: each unit compiles its own copy of what it uses.  Split out of bsq_kernels.hip in
// round 5 (one 2 769-line translation unit).
#pragma once
#include <hip/hip_runtime.h>

#include <climits>
#include <cstdint>
#include <mutex>
#include <type_traits>

#include "bsq.h"
#include "bsq_diag.h"
#include "bsq_device.h"
#include "bsq_internal.h"

namespace {


using namespace bsq_dev;  // kThreads, kNone, kChunk, store16, fast_div, div_constants, div_by

constexpr int kTT = 64;               // positions per tile
constexpr int kTokStride = kTT + 4;   // bytes per sequence row of the LDS token tile (17 dwords:
                                      // odd dword stride -> column reads hit 32 distinct banks)
constexpr int64_t kMaxTiledP = int64_t(1) << 22;  // tiled kernels: 256 sequences x padlen must fit 32-bit offsets

struct KParams {
    int8_t lut[256];
    const uint8_t *chars;
    const int64_t *offsets;
    const uint8_t *mask;  // may be null
    void *out;
    int64_t B;
    int64_t P;
    int32_t C;        // one-hot channels
    int32_t bos;      // 0/1
    int32_t eos;      // 0/1
    int32_t bos_id;
    int32_t eos_id;
    int32_t fill_id;  // token of positions >= L+bos+eos: pad id, or kNone without padchar
    int32_t ntb;      // number of sequence tiles
    int32_t aligned;  // 1: every output row segment is 16-byte aligned -> vector stores
    int32_t vw;       // k_tokens_raw: bytes per store that the alignment of its output rows allows (16, 8, 4 or 1);
                      // k_tokenize_tile: 2 = rows only element-aligned, segments cut at the output's 16-byte lines
    int32_t ntt;      // number of position tiles
    int32_t order;    // 0: sequence-tile index fastest over blockIdx, 1: position-tile index fastest, 2: XCD-aware
    int32_t group;    // order 2: sequence tiles per XCD and group (see tile_of_block)
    int64_t out_pitch;  // k_tokens_raw only: bytes between two position rows of its output
    int64_t row_seqs;  // k_onehot_tile: sequences per position row of the DESTINATION tensor (= B unless the batch is a column block of a larger one)
    uint64_t one_bits;
    uint32_t tab_raw[8], tab_val[8];  // 32-entry folded alphabet (index c & 31): ids with kNone / values with 0 for unmapped
    int32_t foldable;                 // the folded tables represent lut[] exactly (letters only, both cases alike)
    const bsq_desc *desc;             // HOST only: the descriptor this was filled from (launchers that hand the work to bsq_tokens8.hip)
};

// order 2 (XCD-aware): the position tiles of ONE sequence tile go to blocks b, b + 8, b + 16, ... -- one XCD under
// round-robin placement, dispatched back to back -- so the character lines that neighbouring position tiles share
// (a 128-byte line holds the characters of two 64-position tiles) are fetched into that XCD's L2 once.  The grid
// is tile_grid() blocks; blocks whose sequence tile lies beyond the batch exit.
__device__ __forceinline__ void tile_of_block(const KParams &p, int32_t &tb, int32_t &tt) {
    if (p.order == 2) {
        // groups of 8 * G sequence tiles: inside a group all tiles of position tile 0 first, then position tile 1, ...
        // (G = 1: the plain XCD-aware order).  A larger G keeps the rows that are written at the same time together
        // (DRAM locality of the store stream) while the group's characters still sit in the XCDs' L2s.
        const uint32_t G = static_cast<uint32_t>(p.group);
        const uint32_t per = 8u * G * static_cast<uint32_t>(p.ntt);
        const uint32_t g = blockIdx.x / per, r = blockIdx.x % per;
        tt = static_cast<int32_t>(r / (8u * G));
        const uint32_t q = r % (8u * G);
        tb = static_cast<int32_t>((g * G + (q >> 3)) * 8u + (q & 7u));
    } else if (p.order == 5) {
        // as 4, but all position tiles of a sequence tile back to back (the character lines they share stay in that L2)
        const uint32_t per = (static_cast<uint32_t>(p.ntb) + 7u) / 8u;
        const uint32_t xcd = blockIdx.x & 7u, i = blockIdx.x >> 3;
        tb = static_cast<int32_t>(xcd * per + i / static_cast<uint32_t>(p.ntt));
        tt = static_cast<int32_t>(i % static_cast<uint32_t>(p.ntt));
    } else if (p.order == 4) {
        // every XCD walks its own contiguous range of sequence tiles (position tile by position tile): the row segments
        // of neighbouring sequence tiles are written through the SAME L2, close in time -- when the rows are not
        // 64-byte aligned, the memory sectors that two tiles share are merged there instead of being written twice,
        // partially, from two XCDs.
        const uint32_t per = (static_cast<uint32_t>(p.ntb) + 7u) / 8u;
        const uint32_t xcd = blockIdx.x & 7u, i = blockIdx.x >> 3;
        tb = static_cast<int32_t>(xcd * per + i % per);
        tt = static_cast<int32_t>(i / per);
    } else if (p.order == 0) {
        tb = static_cast<int32_t>(blockIdx.x % static_cast<uint32_t>(p.ntb));
        tt = static_cast<int32_t>(blockIdx.x / static_cast<uint32_t>(p.ntb));
    } else {
        tt = static_cast<int32_t>(blockIdx.x % static_cast<uint32_t>(p.ntt));
        tb = static_cast<int32_t>(blockIdx.x / static_cast<uint32_t>(p.ntt));
    }
}

// Aligned dword loads go through the GLOBAL address space (a pointer rebuilt from an integer would
// otherwise be a flat pointer and cost a flat_load + lgkmcnt wait).
typedef const __attribute__((address_space(1))) uint32_t *global_u32_ptr;

// Tokens of positions tpos..tpos+3 of one sequence, packed little-endian into a dword.
// s_lut holds the alphabet table with unmapped == kNone.  Semantics follow
// /root/reference/src/tokenize.h:342-369 (one-hot) and :454-479 (tokens):
//   pos 0 -> BOS (if bos); pos bos+j -> lut[s[j]] (mask==0 or unmapped -> none);
//   pos bos+L -> EOS (if eos); later positions -> PAD id (if padchar) or none.
//
// Addressing: a workgroup reads the characters of a WINDOW of consecutive sequences [b_first, b_first + n).
// Inside the window every address is a 32-bit byte offset from a 4-byte aligned, wave-uniform base
// (chars + offsets[b_first], rounded down), so a fetch costs a handful of 32-bit VALU operations and its
// loads take the scalar base + 32-bit offset form.  The window spans at most n * padlen characters
// (n <= 256, padlen <= 2^22 on these paths).
struct TokenRule {
    global_u32_ptr chars_al, mask_al;  // aligned window bases (mask_al: null without a mask)
    uint32_t mis, mis_m;               // (unaligned base) & 3
    int32_t last, last_m;              // offset of the last dword that holds a byte of the BUFFER (clamp bound)
    int64_t off0;                      // offsets[b_first]: sequence starts are stored relative to it
    bool nonempty;                     // the window holds at least one character (else nothing may be read)
    int32_t bos;
    uint32_t bos_id, at_len_id, fill_id;  // ids at position 0 (BOS), bos+L (EOS or fill) and beyond
};
__device__ __forceinline__ TokenRule make_rule(const KParams &k, int64_t b_first, int32_t n) {
    TokenRule r;
    const int64_t bf = b_first < k.B ? b_first : k.B;
    const int64_t bl = b_first + n < k.B ? b_first + n : k.B;
    const int64_t total = k.offsets[k.B];
    r.off0 = k.offsets[bf];
    r.nonempty = k.offsets[bl] > r.off0;
    const uintptr_t base = reinterpret_cast<uintptr_t>(k.chars) + static_cast<uintptr_t>(r.off0);
    const uintptr_t end_w = (reinterpret_cast<uintptr_t>(k.chars) + static_cast<uintptr_t>(total) - 1) & ~uintptr_t(3);
    const int64_t span = static_cast<int64_t>(end_w) - static_cast<int64_t>(base & ~uintptr_t(3));  // >= 0 when nonempty
    r.chars_al = reinterpret_cast<global_u32_ptr>(base & ~uintptr_t(3));
    r.mis = static_cast<uint32_t>(base & 3);
    r.last = static_cast<int32_t>(span < 0 ? 0 : (span > 0x7FFFFFF8 ? 0x7FFFFFF8 : span));
    r.mask_al = nullptr;
    r.mis_m = 0;
    r.last_m = 0;
    if (k.mask) {  // same offsets in the mask array, its own alignment
        const uintptr_t mbase = reinterpret_cast<uintptr_t>(k.mask) + static_cast<uintptr_t>(r.off0);
        const uintptr_t mend_w = (reinterpret_cast<uintptr_t>(k.mask) + static_cast<uintptr_t>(total) - 1) & ~uintptr_t(3);
        const int64_t mspan = static_cast<int64_t>(mend_w) - static_cast<int64_t>(mbase & ~uintptr_t(3));
        r.mask_al = reinterpret_cast<global_u32_ptr>(mbase & ~uintptr_t(3));
        r.mis_m = static_cast<uint32_t>(mbase & 3);
        r.last_m = static_cast<int32_t>(mspan < 0 ? 0 : (mspan > 0x7FFFFFF8 ? 0x7FFFFFF8 : mspan));
    }
    r.bos = k.bos;
    r.bos_id = static_cast<uint32_t>(k.bos_id);
    r.fill_id = static_cast<uint32_t>(k.fill_id);
    r.at_len_id = k.eos ? static_cast<uint32_t>(k.eos_id) : r.fill_id;
    return r;
}

// The raw words behind 4 consecutive characters (and their mask bytes).  fetch4 issues its loads
// UNCONDITIONALLY so that callers can keep many fetch4's in flight before the first finish4 consumes one:
// the two dword offsets are CLAMPED into the buffer instead of predicated.  A dword that holds a needed
// character is never moved by the clamp (it lies inside the buffer); any other dword only supplies bytes
// that the position rules of finish4 overwrite.  No word wholly outside the buffer is ever touched.
struct Raw4 {
    uint32_t a, b, ma, mb, sh;
};

typedef const __attribute__((address_space(1))) uint8_t *global_u8_ptr;
// Dword at byte offset min(off, last) from `base` (one v_min_u32 + a scalar-base load).  A "negative" offset
// (the dword before the window: only ever behind the BOS position of the window's first sequence) wraps
// to a huge unsigned value and is clamped to `last` like any offset past the end.
__device__ __forceinline__ uint32_t load_clamped(global_u32_ptr base, uint32_t off, int32_t last) {
    const uint32_t o = off < static_cast<uint32_t>(last) ? off : static_cast<uint32_t>(last);
    return *reinterpret_cast<global_u32_ptr>(reinterpret_cast<global_u8_ptr>(base) + o);
}

// `start` = offsets[b] - rule.off0 (window-relative), tpos = first of the four positions.
// CHECK = false: the caller has hoisted the (workgroup-uniform) `nonempty` test out of its loop -- inside it, the
// branch keeps the compiler from batching the span reads and the loads of several fetches.
// MASK: 0 no mask; 1 test p.mask_al at run time (the tiled kernels serve both cases); 2 the caller knows there is one
// (like CHECK, the wave-uniform test inside the fetch keeps the compiler from batching the loads of several fetches).
template <int MASK = 1, bool CHECK = true>
__device__ __forceinline__ Raw4 fetch4(const TokenRule p, uint32_t start, int32_t tpos) {
    if (CHECK && !p.nonempty) return Raw4{0, 0, ~0u, ~0u, 0};  // wave-uniform: nothing to read
    const uint32_t j = start + static_cast<uint32_t>(tpos - p.bos);  // may be "-1" (BOS position of the first sequence)
    const uint32_t rel = j + p.mis;
    const uint32_t w0 = rel & ~3u;
    Raw4 r;
    r.sh = rel & 3u;
    r.a = load_clamped(p.chars_al, w0, p.last);
    r.b = load_clamped(p.chars_al, w0 + 4u, p.last);
    r.ma = r.mb = 0xFFFFFFFFu;
    if (MASK == 2 || (MASK == 1 && p.mask_al)) {  // wave-uniform
        const uint32_t relm = j + p.mis_m;
        const uint32_t m0 = relm & ~3u;
        r.ma = load_clamped(p.mask_al, m0, p.last_m);
        r.mb = load_clamped(p.mask_al, m0 + 4u, p.last_m);
        r.sh |= (relm & 3u) << 8;
    }
    return r;
}

// Tokens of positions tpos..tpos+3 from the fetched words: four LUT lookups packed into one word, then
// mask / PAD / EOS / BOS applied to the packed word.  Bytes fetched from outside [0, L) are garbage but
// every such position is overwritten by the rules below.
template <int MASK = 1>
__device__ __forceinline__ uint32_t finish4(const TokenRule p, const uint8_t *s_lut, const Raw4 r, int32_t L,
                                            int32_t tpos) {
    const int32_t j0 = tpos - p.bos;
    const uint32_t cw = __builtin_amdgcn_alignbyte(r.b, r.a, r.sh & 3u);
    uint32_t w = static_cast<uint32_t>(s_lut[cw & 0xFFu]) | (static_cast<uint32_t>(s_lut[(cw >> 8) & 0xFFu]) << 8) |
                 (static_cast<uint32_t>(s_lut[(cw >> 16) & 0xFFu]) << 16) | (static_cast<uint32_t>(s_lut[cw >> 24]) << 24);
    if (MASK == 2 || (MASK == 1 && p.mask_al)) {
        const uint32_t mw = __builtin_amdgcn_alignbyte(r.mb, r.ma, (r.sh >> 8) & 3u);
        uint32_t z = (mw & 0x7F7F7F7Fu) + 0x7F7F7F7Fu;  // exact zero-byte detection:
        z = ~(z | mw | 0x7F7F7F7Fu);                     // 0x80 in every byte of mw that is zero
        w |= (z >> 7) * 0xFFu;                           // masked position -> kNone
    }
    // Branch-free position rules (bytes are positions tpos..tpos+3, nv = characters left from the first one):
    //   byte >= nv      -> fill (PAD id / none)      byte == nv -> EOS (or fill)      j0 < 0: byte 0 -> BOS
    const int32_t nv = L - j0;
    const int32_t nvc = nv < 0 ? 0 : (nv > 4 ? 4 : nv);
    const uint32_t keep = static_cast<uint32_t>((uint64_t(1) << (8 * nvc)) - 1u);  // low nvc bytes
    w = (w & keep) | ((p.fill_id * 0x01010101u) & ~keep);
    const uint32_t at = (nv >= 0 && nv < 4) ? (0xFFu << (8 * nvc)) : 0u;
    w = (w & ~at) | ((p.at_len_id * 0x01010101u) & at);
    const uint32_t first = j0 < 0 ? 0xFFu : 0u;
    w = (w & ~first) | (p.bos_id & first);
    return w;
}


__device__ __forceinline__ uint32_t resolve4(const TokenRule p, const uint8_t *s_lut, uint32_t start, int32_t L,
                                             int32_t tpos) {
    return finish4(p, s_lut, fetch4(p, start, tpos), L, tpos);
}

__device__ __forceinline__ void stage_lut(const KParams &p, uint8_t *s_lut) {
    // bytes >= 0x80 and negative table entries are unmapped
    const int i = threadIdx.x;
    if (i < 256) {
        const int8_t v = p.lut[i];
        s_lut[i] = (i < 128 && v >= 0) ? static_cast<uint8_t>(v) : static_cast<uint8_t>(kNone);
    }
}

__device__ __forceinline__ int32_t clamp_len(const KParams &p, int64_t len) {
    const int64_t room = p.P - p.bos - p.eos;  // memory safety only; callers validate beforehand
    return static_cast<int32_t>(len < 0 ? 0 : (len > room ? (room < 0 ? 0 : room) : len));
}

// (window-relative start, clamped length) of the n sequences of a window -> LDS, one 8-byte entry each.
// Sequences past the end of the batch get length 0.
struct SeqSpan {
    uint32_t start;
    int32_t len;
};
__device__ __forceinline__ void stage_spans(const KParams &p, const TokenRule &rule, int64_t b0, int n, SeqSpan *s_span) {
    for (int i = threadIdx.x; i < n; i += kThreads) {
        const int64_t b = b0 + i;
        const int64_t lo = p.offsets[b <= p.B ? b : p.B], hi = p.offsets[b + 1 <= p.B ? b + 1 : p.B];
        s_span[i] = SeqSpan{static_cast<uint32_t>(lo - rule.off0), clamp_len(p, hi - lo)};
    }
}

// Phase 1 shared by the tiled kernels: s_tok[sb * kTokStride + tl] = token of sequence b0+sb at
// position t0+tl, for sb < TB, tl < 64.  Sequences past the end of the batch get kNone.
template <int TB>
__device__ __forceinline__ void build_token_tile(const KParams &p, int64_t b0, int32_t t0, uint8_t *s_lut,
                                                 SeqSpan *s_span, uint8_t *s_tok) {
    const int tid = threadIdx.x;
    const TokenRule rule = make_rule(p, b0, TB);
    stage_lut(p, s_lut);
    stage_spans(p, rule, b0, TB, s_span);
    __syncthreads();
    const int g = tid & 15;  // 16 lanes x 4 characters cover the 64 positions of one sequence
    constexpr int NI = TB / 16;                  // sequences per thread
    constexpr int BATCH = NI < 4 ? NI : 4;       // fetches kept in flight (more costs occupancy: 129 VGPRs at 8)
    auto run = [&](auto nonempty) {  // the window holds characters (workgroup-uniform): hoisted out of the fetches
#pragma unroll 1
        for (int i0 = 0; i0 < NI; i0 += BATCH) {
            Raw4 raw[BATCH];
            int32_t len[BATCH];
#pragma unroll
            for (int k = 0; k < BATCH; ++k) {
                const int sb = (tid >> 4) + 16 * (i0 + k);
                const SeqSpan sp = s_span[sb];
                len[k] = sp.len;
                raw[k] = decltype(nonempty)::value ? fetch4<1, false>(rule, sp.start, t0 + 4 * g) : Raw4{0, 0, ~0u, ~0u, 0};
            }
#pragma unroll
            for (int k = 0; k < BATCH; ++k) {
                const int sb = (tid >> 4) + 16 * (i0 + k);
                const uint32_t w = finish4(rule, s_lut, raw[k], len[k], t0 + 4 * g);
                *reinterpret_cast<uint32_t *>(s_tok + sb * kTokStride + 4 * g) = (b0 + sb < p.B) ? w : kNone * 0x01010101u;
            }
        }
    };
    if (rule.nonempty) run(std::true_type{});
    else run(std::false_type{});
    __syncthreads();
}

// Fixed part of the tiled kernels' dynamic LDS (k_onehot_tile, k_tokenize_tile): the tile's offsets (+1), the alphabet LUT, the token tile.
template <int TB>
__host__ __device__ constexpr int tile_off_bytes() {
    return ((TB + 1) * 8 + 15) & ~15;
}
template <int TB>
__host__ __device__ constexpr int tile_fixed_bytes() {
    return tile_off_bytes<TB>() + 256 + TB * kTokStride;
}

// amdgpu_waves_per_eu(5) for the 256-sequence tile (the smaller ones would spill a few registers): 94 instead of 127 VGPRs, so that the 5 workgroups per CU that the LDS allows also fit the
// register file (4 before): 65536 x 256 int8 DNA 31 -> 28 us, 8192 x 512 int8 AMINO20 22 -> 19 us, cfg4 int8 -1..3 %
// (profiles/r02/ab_tile_occ5.txt).
// Raw (P,B) uint8 tokens (kNone kept) for k_expand_chunks.  Workgroup = 256 sequences x 64 positions.
// Phase 1: 4 characters per lane (two aligned words + alignbyte, 8 fetches of a thread in flight together),
// 4 LUT lookups packed into a word, mask / PAD / EOS / BOS applied to the packed word, bytes written
// TRANSPOSED into LDS (row = position; the four sequences a wave handles per step are 4 apart so the 64
// byte-writes of an instruction fall on 32 banks).  Phase 2: each position row of the tile is 256 contiguous
// bytes of the output: dword LDS reads -> 16-byte stores.
// (Measured alternatives that did NOT help: a tile-major scratch written as whole 4-KiB chunks, 16 fetches in
// flight per thread -- the kernel is bound by its ~17 VALU instructions per token, not by memory.)
constexpr int kRawTB = 256;
constexpr int kRawStride = kRawTB + 4;  // 65 dwords: odd stride
// EXPERIMENT (knob "raw_mode" 4, lost): a WIDE tile of 1024 sequences x 16 positions, so that a position row of the tile
// is 1 KiB of the output (one full-wave 16-byte store) instead of 256 bytes.  Stores alone on the cfg2 geometry (1024
// rows x 64 KiB; profiles/r02/pattern_cfg2sf.txt) take 17.6 us in 256-byte segments and 11-12 us in 1-KiB segments,
// but the kernel got SLOWER (cfg2 24.3 -> 35.0 us, cfg5 42 -> 59, cfg4 57 -> 71; profiles/r02/seqfirst_lab1.txt):
// every sequence then contributes 16 characters per tile, so a 128-byte character line is fetched by eight tiles and
// the two dwords behind a lane's four characters are rarely shared -- the character side, not the store pattern, is
// what the tile pays for.  (A 512 x 32 tile measured within +-4 % of 256 x 64: profiles/r02/seqfirst_lab3.txt.)
constexpr int kWideTB = 1024, kWideTT = 16;
// LDS row stride of a TB-sequence tile: the byte-writes of one instruction (TT/4 position groups x 64/(TT/4) sequences
// 4 apart) must spread over the 32 banks twice -- 65 dwords for 16 x 4, 260 dwords (= 4 mod 32, 16-byte rows) for 4 x 16.
template <int TB>
__host__ __device__ constexpr int raw_stride() {
    return TB == 256 ? kRawStride : TB + 16;
}

// RAW = false: the same kernel produces the final int8 (P,B) token matrix of batch_tokenize(batch_first=False)
// (unmapped / unpadded positions are 0 instead of kNone).
// HOIST: the (workgroup-uniform) "window holds characters" test is taken out of the fetches, so the compiler batches
// the span reads and loads of the 8 fetches of a step (70 VGPRs instead of 42, 7 instead of 8 workgroups per CU):
// 12-15 % faster while the grid is about one round of workgroups (latency), 2-5 % slower on large grids
// (profiles/r02/ab_hoist.txt, ab_hoist2.txt) -- the launcher picks it for small (P,B) int8 token matrices.
template <bool MASK, bool RAW = true, int TB = kRawTB, int TT = kTT, bool HOIST = false>
__global__ __launch_bounds__(kThreads) void k_tokens_raw(const KParams p) {
    constexpr int STRIDE = raw_stride<TB>();
    constexpr int LPS = TT / 4;          // lanes per sequence (4 characters each)
    constexpr int SPP = kThreads / LPS;  // sequences per step of the workgroup
    static_assert(TB * TT == kRawTB * kTT && TB % SPP == 0 && SPP % 16 == 0, "tile shape");
    __shared__ __align__(16) uint8_t s_lut[256];
    __shared__ __align__(16) SeqSpan s_span[TB];
    __shared__ __align__(16) uint8_t s_t[TT * STRIDE];
    const int tid = threadIdx.x;
    int32_t tb, tt;
    tile_of_block(p, tb, tt);
    if (tb >= p.ntb) return;
    const int64_t b0 = static_cast<int64_t>(tb) * TB;
    const int32_t t0 = tt * TT;
    stage_lut(p, s_lut);
    TokenRule rule = make_rule(p, b0, TB);
    stage_spans(p, rule, b0, TB, s_span);
    __syncthreads();
    if (!RAW) {  // value space: "no token" is the memset 0 of tokenize.h:427
        if (tid < 64) {  // s_lut was staged with kNone markers: rewrite them (one dword per lane)
            uint32_t w = reinterpret_cast<uint32_t *>(s_lut)[tid];
            uint32_t z = (~w & 0x7F7F7F7Fu) + 0x7F7F7F7Fu;   // bytes equal to 0xFF <=> ~byte == 0
            z = ~(z | ~w | 0x7F7F7F7Fu);
            reinterpret_cast<uint32_t *>(s_lut)[tid] = w & ~((z >> 7) * 0xFFu);
        }
        __syncthreads();
        if (rule.fill_id == kNone) rule.fill_id = 0;
        if (rule.at_len_id == kNone) rule.at_len_id = 0;
    }
    const int g = tid % LPS;
    const int32_t tpos = t0 + 4 * g;
    constexpr int NI = TB / SPP, BATCH = 8;
    // Sequence of (thread group tg = tid / LPS, step k): 4 * (tg % (SPP/4)) + tg / (SPP/4) + SPP * k -- the sequences
    // of one wave are 4 apart.
    const int tg = tid / LPS;
    const int sb0 = 4 * (tg % (SPP / 4)) + tg / (SPP / 4);
    auto run = [&](auto mode) {
        constexpr int M = decltype(mode)::value;
#pragma unroll 1
        for (int i0 = 0; i0 < NI; i0 += BATCH) {
            Raw4 raw[BATCH];
            int32_t len[BATCH];
#pragma unroll
            for (int k = 0; k < BATCH; ++k) {
                const int sb = sb0 + SPP * (i0 + k);
                const SeqSpan sp = s_span[sb];
                len[k] = sp.len;
                if constexpr (M == 0) raw[k] = Raw4{0, 0, ~0u, ~0u, 0};
                else if constexpr (M == 1) raw[k] = fetch4<MASK ? 2 : 0, false>(rule, sp.start, tpos);
                else raw[k] = fetch4<MASK ? 2 : 0, true>(rule, sp.start, tpos);
            }
#pragma unroll
            for (int k = 0; k < BATCH; ++k) {
                const int sb = sb0 + SPP * (i0 + k);
                const uint32_t w = finish4<MASK ? 2 : 0>(rule, s_lut, raw[k], len[k], tpos);  // columns >= B are never read
                uint8_t *col = s_t + (4 * g) * STRIDE + sb;
                col[0] = static_cast<uint8_t>(w);
                col[STRIDE] = static_cast<uint8_t>(w >> 8);
                col[2 * STRIDE] = static_cast<uint8_t>(w >> 16);
                col[3 * STRIDE] = static_cast<uint8_t>(w >> 24);
            }
        }
    };
    if constexpr (!HOIST) run(std::integral_constant<int, 2>{});  // the test inside every fetch
    else if (rule.nonempty) run(std::integral_constant<int, 1>{});
    else run(std::integral_constant<int, 0>{});
    __syncthreads();
    uint8_t *out = static_cast<uint8_t *>(p.out);
    if (p.vw == 16) {
        for (int f = tid; f < TT * (TB / 16); f += kThreads) {
            const int32_t tl = f / (TB / 16), q = f % (TB / 16);
            const int64_t t = static_cast<int64_t>(t0) + tl;
            if (t >= p.P) continue;
            const uint8_t *src = s_t + tl * STRIDE + q * 16;
            uint8_t *dst = out + t * p.out_pitch + b0 + q * 16;
            if (b0 + q * 16 + 16 <= p.out_pitch) {
                uint4 v;
                if constexpr (STRIDE % 16 == 0) {
                    v = *reinterpret_cast<const uint4 *>(src);
                } else {  // LDS rows are only 4-byte aligned (stride 260): four dword reads
                    v.x = *reinterpret_cast<const uint32_t *>(src);
                    v.y = *reinterpret_cast<const uint32_t *>(src + 4);
                    v.z = *reinterpret_cast<const uint32_t *>(src + 8);
                    v.w = *reinterpret_cast<const uint32_t *>(src + 12);
                }
                if constexpr (RAW)
                    *reinterpret_cast<uint4 *>(dst) = v;  // scratch: re-read by the expansion pass right away
                else
                    store16<true>(dst, v);               // final token matrix: streamed once
            } else {
                for (int i = 0; i < 16; ++i)
                    if (b0 + q * 16 + i < p.B) dst[i] = src[i];
            }
        }
    } else if (p.vw >= 4) {  // rows (batch size) only 8- or 4-byte aligned: 8- / 4-byte stores, still one row per wave step
        const int lg = p.vw == 8 ? 3 : 2, ppr = TB >> lg;  // pieces per tile row
        for (int f = tid; f < TT * ppr; f += kThreads) {
            const int32_t tl = f / ppr, q = f % ppr;
            const int64_t t = static_cast<int64_t>(t0) + tl;
            if (t >= p.P) continue;
            const uint8_t *src = s_t + tl * STRIDE + (q << lg);
            uint8_t *dst = out + t * p.out_pitch + b0 + (q << lg);
            if (b0 + (q << lg) + p.vw <= p.B) {
                const uint32_t lo = *reinterpret_cast<const uint32_t *>(src);
                if (lg == 3)
                    *reinterpret_cast<uint2 *>(dst) = uint2{lo, *reinterpret_cast<const uint32_t *>(src + 4)};
                else
                    *reinterpret_cast<uint32_t *>(dst) = lo;
            } else {
                for (int i = 0; i < p.vw; ++i)
                    if (b0 + (q << lg) + i < p.B) dst[i] = src[i];
            }
        }
    } else {  // any batch size: byte stores, consecutive lanes on consecutive bytes
        for (int f = tid; f < TT * TB; f += kThreads) {
            const int32_t tl = f / TB, c = f % TB;
            const int64_t t = static_cast<int64_t>(t0) + tl;
            if (t < p.P && b0 + c < p.B) out[t * p.out_pitch + b0 + c] = s_t[tl * STRIDE + c];
        }
    }
}


// ------------------------------------------------------------------------------------------
// Launch helpers
// ------------------------------------------------------------------------------------------
// Blocks of a tiled launch (see tile_of_block).
int64_t tile_grid(const KParams &k, int64_t ntt) {
    const int64_t unit = 8 * int64_t(k.group);
    if (k.order == 4 || k.order == 5) return (int64_t(k.ntb) + 7) / 8 * 8 * ntt;
    return (k.order == 2 ? (int64_t(k.ntb) + unit - 1) / unit * unit : int64_t(k.ntb)) * ntt;
}

bsq_status check_launch(const char *what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return bsq_internal::set_hip_error(what, e);
    return BSQ_OK;
}

uint64_t one_bits_of(bsq_dtype t) {
    switch (t) {
    case BSQ_F32: return 0x3F800000ull;
    case BSQ_F64: return 0x3FF0000000000000ull;
    default: return 1ull;
    }
}

bsq_status fill_common(KParams &k, const bsq_desc *d, const uint8_t *chars, const int64_t *offsets,
                       const uint8_t *mask, int64_t B, int64_t P, void *out) {
    if (!d || B < 0 || P <= 0 || (B > 0 && (!offsets || !out)))  // (an EMPTY batch -- a rank without sequences -- has nothing to point at)
        return bsq_internal::set_error(BSQ_ERR_INVALID_ARG, "null pointer, B < 0 or padlen <= 0");
    if (P > (int64_t(1) << 30)) return bsq_internal::set_error(BSQ_ERR_INVALID_ARG, "padlen > 2^30 is not supported");
    for (int i = 0; i < 256; ++i) k.lut[i] = d->lut[i];
    k.desc = d;
    k.chars = chars;
    k.offsets = offsets;
    k.mask = mask;
    k.out = out;
    k.B = B;
    k.P = P;
    k.C = bsq_alphabet_size(d);
    k.bos = d->bos;
    k.eos = d->eos;
    k.bos_id = bsq_bos_id(d);
    k.eos_id = bsq_eos_id(d);
    k.fill_id = d->padchar ? bsq_pad_id(d) : static_cast<int32_t>(kNone);
    k.ntb = 1;
    k.ntt = int32_t((P + kTT - 1) / kTT);
    // knob "tile_order": 0 automatic (XCD-aware), 1 position-tile index fastest, 2 XCD-aware, 3 sequence-tile index fastest.
    // XCD-aware placement fetches the characters once instead of ~3 times on the 1M x 160 DNA batch (FETCH_SIZE 234 ->
    // 78 MB; k_tokens_raw 97 -> 66 us, k_onehot_tile 258 -> 233 us: profiles/r02/order_lab.txt).
    const int order_knob = bsq_internal::tuning().tile_order;
    k.order = order_knob == 1 ? 1 : (order_knob == 3 ? 0 : (order_knob == 4 ? 4 : (order_knob == 5 ? 5 : 2)));
    const int group_knob = bsq_internal::tuning().tile_group;
    k.group = group_knob > 0 && group_knob <= 4096 ? group_knob : 1;
    if (k.order == 2 && (B / 64 + 8 * int64_t(k.group)) * int64_t(k.ntt) >= (int64_t(1) << 31)) k.order = 0;  // keep the rounded-up grid in 32 bits
    k.aligned = 0;
    k.vw = 1;
    k.out_pitch = B;
    k.row_seqs = B;
    k.one_bits = 1;
    {   // folded tables (see k_tokens_raw2): exact iff only letter positions are mapped and both cases map alike
        bool ok = true;
        for (int i = 0; i < 8; ++i) k.tab_raw[i] = 0xFFFFFFFFu, k.tab_val[i] = 0;
        for (int c = 0; c < 256 && ok; ++c) {
            const bool mapped = c < 128 && d->lut[c] >= 0;
            if (!mapped) continue;
            if (c < 0x40 || d->lut[c ^ 0x20] != d->lut[c]) {
                ok = false;
                break;
            }
            const uint32_t id = uint32_t(uint8_t(d->lut[c])), sh = 8 * (c & 3);
            uint32_t &tr = k.tab_raw[(c & 31) >> 2], &tv = k.tab_val[(c & 31) >> 2];
            tr = (tr & ~(0xFFu << sh)) | (id << sh);
            tv = (tv & ~(0xFFu << sh)) | (id << sh);
        }
        k.foldable = ok;
    }
    return BSQ_OK;
}

}  // namespace
