// HIP kernels of libbsq_hip.so, written for gfx950 (MI355X / CDNA4) only.
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

#ifdef BSQ_LABS
#include "labs/bsq_tokens_raw2_helpers.inc"  // helpers of k_tokens_raw2 (4 x 4 byte transposes in registers
#endif

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

// ------------------------------------------------------------------------------------------
// One-hot, tiled.  Dynamic LDS layout:
//   [0, off_bytes)              int64 offsets of the tile's sequences (+1)
//   [.., +256)                  alphabet LUT
//   [.., +TB*kTokStride)        token tile
//   [.., +4*row_pad)            one row image per wave (row_pad = TB*C*sizeof(ST) rounded to 16)
// ------------------------------------------------------------------------------------------
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
    int32_t mode;                          // k_expand_small: 9 = no token loads (ablation)
    int64_t pitch;                         // B * rowbytes: bytes of one position row of the output
    unsigned int *claim;                   // CLAIM: 8 counters, 128 bytes apart, zeroed before the launch
    int64_t groups_per_class;              // CLAIM: slots-of-4 per class
    Div64 dv_pitch, dv_rb;                 // div64() constants of pitch and rowbytes (scalar chunk arithmetic)
    int64_t row_gap;                       // column block of a wider tensor: bytes between the end of one position row of the block and
                                           // the start of the next (0 = the whole tensor).  Non-zero only when pitch % 4096 == 0 and
                                           // head == 0: no chunk then straddles two position rows
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
// (-DBSQ_LABS builds add a CLAIM parameter: labs/bsq_expand_claim.inc.)
// GATE (rows of 24 ... 63 bytes; knob "expand_gate"): every wave first issues ONE agent-scope load -- of the head of the token scratch, a
// line that is always at the memory side -- and makes its token loads depend on it.  The load means nothing; what it does is pace the
// waves: the small-row expansion runs at 5 workgroups per CU, all of whose waves otherwise reach their token loads and their 4 KiB of
// stores in step.  Measured over 24 shapes (profiles/r04/expand_gate_sweep.txt): 28-byte rows (DNA f32: cfg4) +1-2 %, 32-byte rows +5.5 %,
// 56-byte rows +4.7 %; rows of 64 bytes and more lose 5-7 % (cfg3 0.724 -> 0.774 ms), rows of 20 bytes and less lose 1-4 %: those
// do not get it.  (Found as a by-product of the one-launch experiment, profiles/r04/onehot_fused_one_launch_lost.txt.)
#ifdef BSQ_LABS
template <typename ST, bool NT, int MATH, bool GATE = false, int CLAIM = 0>
#else
template <typename ST, bool NT, int MATH, bool GATE = false>
#endif
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
#ifdef BSQ_LABS
#include "labs/bsq_expand_claim.inc"  // CLAIM == 1: placement-independent chunk classes (measurement only)
#endif
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
    const uint8_t *tok = p.tok + t_lo * p.Bp + b_lo;
    if constexpr (GATE) tok += (gate == 0xFEFEFEFDu && p.nchunks < 0) ? 1 : 0;  // never taken: the token loads wait for the gate load
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
            tk[q] = i < nr_s ? static_cast<uint32_t>(tok[a]) : kNone;
        }
#pragma unroll
        for (int q = 0; q < NS; ++q) {
            const int32_t i = i0 + 64 * q + lane;
            const int32_t pos = i * rowbytes - skip + static_cast<int32_t>(tk[q]) * static_cast<int32_t>(sizeof(ST));
            if (tk[q] != kNone && pos >= 0 && pos < len) *reinterpret_cast<ST *>(img + pos) = one;
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
template <bool NT, int NR>
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
    const uint8_t *tok = p.tok + cc.t_lo * p.Bp + cc.b_lo;
    const int64_t wrap_at = p.B - cc.b_lo;  // rows i >= wrap_at belong to position t_lo + 1 (or later)
    const bool wraps = p.Bp != p.B && wrap_at < nr_s;  // (wave-uniform) the chunk runs over the end of a position row of a PADDED id matrix
    uint8_t *g = p.out + cc.lo + cc.t_lo * p.row_gap;
    const uint32_t one = static_cast<uint32_t>(p.one_bits) & 0xFFu;
    // address of the id of row i of the chunk (rows behind wrap_at lie in later position rows of the matrix, Bp - B bytes further each)
    auto id_at = [&](uint32_t i) -> int64_t {
        int64_t at = i;
        if (static_cast<int64_t>(i) >= wrap_at) at += ((static_cast<int64_t>(i) - wrap_at) / p.B + 1) * (p.Bp - p.B);
        return at;
    };
    if (len == kChunk && (reinterpret_cast<uintptr_t>(g) & 15) == 0 && nr_s >= 8) {
        uint64_t ids[4];
        uint32_t ph[4];
        if (!wraps) {
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
                for (int j = 0; j < NR; ++j) b[u][j] = q + j < static_cast<uint32_t>(nr_s) ? static_cast<uint32_t>(tok[id_at(q + j)]) : kNone;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                ids[u] = 0;
#pragma unroll
                for (int j = 0; j < NR; ++j) ids[u] |= static_cast<uint64_t>(b[u][j]) << (8 * j);
            }
        }
        const uint32_t cmask = (1u << rb) - 1u;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            uint32_t bits = 0;
#pragma unroll
            for (int j = 0; j < NR; ++j) {
                const uint32_t id = static_cast<uint32_t>(ids[u] >> (8 * j)) & 0xFFu;
                const uint32_t m = (1u << (id & 31u)) & cmask;  // id 255 (no one) -> bit 31 -> 0
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
            idv[m] = o < len ? static_cast<uint32_t>(tok[id_at(i)]) : kNone;
        }
#pragma unroll
        for (int m = 0; m < 8; ++m)
            if (o0 + 64 * m < len) g[o0 + 64 * m] = idv[m] == cv[m] ? static_cast<uint8_t>(one) : uint8_t(0);
    }
}

#ifdef BSQ_LABS
#include "labs/bsq_expand_small.inc"  // k_expand_small
#endif

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

#ifdef BSQ_LABS
#include "labs/bsq_tokens_raw2.inc"  // k_tokens_raw2
#endif

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
    const int cpw = bsq_internal::tuning().chunks_cpw;  // chunks per wave (1 is fastest: 0.75 / 0.86 / 0.93 ms for 1 / 2 / 4 on cfg3)
    c.cpw = cpw > 0 ? cpw : 1;
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
    // cfg3 at 7.5-7.8 TB/s, 8 at 6.3, 4 at 6.7, 2 at 5.5.  Smaller rows: 5 workgroups per CU (16 KiB + 16 KiB each).
    // In round 1 a cap HURT the 1M x 160 x 28-byte batch (0.90 ms at 5 per CU vs 0.78 uncapped) -- because the token pass
    // then fetched every character three times and pushed its own scratch out of the Infinity Cache; with the XCD-aware
    // tile order the token loads of the expansion are cache hits and the cap pays: 0.687 ms at 5, 0.689 at 4, 0.725 at
    // 3, 0.731 uncapped (profiles/r02/pad_lab2.txt, pad_lab3.txt).  12 waves per CU is
    // the optimum also with 2-wave workgroups (16 / 14 / 12 / 10 waves: 0.83 / 0.81 / 0.76 / 0.93 ms), and 3 x 4 waves
    // (0.73 ms) beats 6 x 2.
    // Knob "expand_pad": 0 = this rule, > 0 = that many bytes, < 0 = none.
    // Knob "expand_mode": 0 / 1 k_expand_chunks; 2 k_expand_small (dword token loads), 9 the same without token loads
    // (ablation).  k_expand_small is an experiment that lost: once the token scratch is written in XCD-aware tile
    // order the byte-load kernel under an occupancy cap is 1-2 % ahead of it, and two or four chunks per wave were
    // 20-40 % slower (profiles/r02/pad_lab*.txt, expand_lab3.txt; those instantiations are no longer built).
#ifdef BSQ_LABS
    const int mode = e.mode;
    const int64_t rb = e.C * int64_t(sizeof(ST));
    if (rb >= 4 && (mode == 2 || mode == 9)) {
        const int padv2 = bsq_internal::tuning().expand_pad;
        const size_t pad2 = padv2 > 0 ? size_t(padv2) : 0;
        if (bsq_internal::nontemporal_stores())
            hipLaunchKernelGGL((k_expand_small<ST, true, 1, 0>), grid, dim3(kThreads), pad2, s, e);
        else
            hipLaunchKernelGGL((k_expand_small<ST, false, 1, 0>), grid, dim3(kThreads), pad2, s, e);
        return check_launch("k_expand_small");
    }
#endif
    const int padv = bsq_internal::tuning().expand_pad;
    const bool big_rows = e.C * int64_t(sizeof(ST)) >= 64;
    const size_t pad = padv > 0 ? size_t(padv) : (padv < 0 ? size_t(0) : (big_rows ? size_t(36864) : size_t(16384)));
#ifdef BSQ_LABS
    if (bsq_internal::tuning().xcd_claim == 1) {  // measurement only: placement-independent chunk classes (see the kernel)
        static unsigned int *counters[16] = {};
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return bsq_internal::set_error(BSQ_ERR_HIP, "hipGetDevice");
        if (!counters[dev] && hipMalloc(reinterpret_cast<void **>(&counters[dev]), 8 * 128) != hipSuccess)
            return bsq_internal::set_error(BSQ_ERR_ALLOC, "claim counters");
        if (hipMemsetAsync(counters[dev], 0, 8 * 128, s) != hipSuccess) return bsq_internal::set_error(BSQ_ERR_HIP, "hipMemsetAsync");
        EParams ec = e;
        ec.claim = counters[dev];
        ec.groups_per_class = groups;
        if (bsq_internal::nontemporal_stores())
            hipLaunchKernelGGL((k_expand_chunks<ST, true, 0, false, 1>), grid, dim3(kThreads), pad, s, ec);
        else
            hipLaunchKernelGGL((k_expand_chunks<ST, false, 0, false, 1>), grid, dim3(kThreads), pad, s, ec);
        return check_launch("k_expand_chunks<claim>");
    }
#endif
    // knob "chunk_math": 2 = scalar 64-bit reciprocal multiplies (div64) instead of the double reciprocals (div_by).
    // Measured (profiles/r02/math_lab1.txt): the scalar prologue is ~90 SALU instructions instead of ~130 VALU ones
    // (22 of them FP64) and wins where a wave's latency is exposed (2 workgroups per CU: 0.938 vs 0.969 ms on cfg3),
    // but at the bandwidth optimum (3 per CU) the double form is 1 % FASTER (0.734 vs 0.741 ms): the stream is paced by
    // the memory system there, not by the prologue.  So the double form stays the default.
#ifdef BSQ_LABS
    if (bsq_internal::tuning().chunk_math == 2) {
        if (bsq_internal::nontemporal_stores())
            hipLaunchKernelGGL((k_expand_chunks<ST, true, 1>), grid, dim3(kThreads), pad, s, e);
        else
            hipLaunchKernelGGL((k_expand_chunks<ST, false, 1>), grid, dim3(kThreads), pad, s, e);
        return check_launch("k_expand_chunks<div64>");
    }
#endif
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
        if (nt) hipLaunchKernelGGL((k_expand_rows1<true, NRV>), grid, dim3(kThreads), rpad, s, e);           \
        else hipLaunchKernelGGL((k_expand_rows1<false, NRV>), grid, dim3(kThreads), rpad, s, e);             \
        break;
            switch (nrows) {
                BSQ_ROWS1(2) BSQ_ROWS1(3) BSQ_ROWS1(4) BSQ_ROWS1(5) BSQ_ROWS1(6)
            default: return bsq_internal::set_error(BSQ_ERR_INVALID_ARG, "k_expand_rows1: rows per lane");
            }
#undef BSQ_ROWS1
            return check_launch("k_expand_rows1");
        }
    }
    // knob "expand_gate": 0 automatic (rows of 24 ... 63 bytes), 1 never, 2 always (the scratch holds at least 256 bytes: Bp >= 256)
    const int gk = bsq_internal::tuning().expand_gate;
    const int64_t rowb = e.C * int64_t(sizeof(ST));
    const bool gated = (gk == 2 || (gk == 0 && rowb >= 24 && rowb < 64)) && e.Bp >= 256;
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
size_t two_pass_workspace_bytes(int64_t B, int64_t P) { return size_t(two_pass_pitch(B)) * size_t(P); }

// Pass 1: raw tokens (kNone = no token) of the batch into a (P, pitch) uint8 matrix.
bsq_status launch_tokens_raw(KParams &k, void *tokens, int64_t pitch, hipStream_t s) {
    k.out = tokens;
    k.out_pitch = pitch;
    k.aligned = reinterpret_cast<uintptr_t>(tokens) % 16 == 0 && pitch % 16 == 0;  // every row starts 16-byte aligned
    k.vw = k.aligned ? 16 : 1;
    k.ntb = int32_t((k.B + kRawTB - 1) / kRawTB);
    // no mask, 16-byte aligned rows: the register-transposed tiles of k_tokens_pb8_fast in raw-id mode (round 3; knob
    // tokens_pb8 = 1 or any raw_mode != 0: k_tokens_raw)
    if (!k.mask && k.desc && bsq_internal::tuning().raw_mode == 0 && bsq_internal::tokens_pb8_applicable(k.desc, k.B, k.P, tokens, pitch))
        return bsq_internal::launch_tokens_pb8(k.desc, k.chars, k.offsets, k.B, k.P, tokens, pitch, s, true);
    const dim3 grid(unsigned(tile_grid(k, k.ntt)));
    // knob "raw_mode": 0 / 1 k_tokens_raw; 2 k_tokens_raw2 (register transpose) with the LDS byte table, 3 with the
    // register table.  k_tokens_raw2 is an experiment that LOST (profiles/r02/raw_lab.txt: cfg2 as (P,B) int8 tokens
    // 24.0 us -> 25.8 (2) / 27.4 (3); cfg3 / cfg4 f32 steps +0.1 / +0.3 %): the tile is bound by vector instructions at
    // least as much as by LDS traffic, and the transpose trades 3 LDS writes for 6 vector instructions per word.
    const int rm = bsq_internal::tuning().raw_mode;
    if (!k.mask && rm == 4 && k.P <= (int64_t(1) << 20)) {  // measurement: the wide tile of the (P,B) int8 token matrix
        k.ntb = int32_t((k.B + kWideTB - 1) / kWideTB);
        k.ntt = int32_t((k.P + kWideTT - 1) / kWideTT);
        if ((int64_t(k.ntb) + 8 * int64_t(k.group)) * int64_t(k.ntt) >= (int64_t(1) << 31)) k.order = 0;
        hipLaunchKernelGGL((k_tokens_raw<false, true, kWideTB, kWideTT>), dim3(unsigned(tile_grid(k, k.ntt))),
                           dim3(kThreads), 0, s, k);
        return check_launch("k_tokens_raw<wide>");
    }
#ifdef BSQ_LABS
    if (!k.mask && (rm == 2 || rm == 3)) {
        if (rm == 3 && k.foldable) hipLaunchKernelGGL((k_tokens_raw2<true, true>), grid, dim3(kThreads), 0, s, k);
        else hipLaunchKernelGGL((k_tokens_raw2<true, false>), grid, dim3(kThreads), 0, s, k);
        return check_launch("k_tokens_raw2");
    }
#endif
    if (k.mask) hipLaunchKernelGGL(k_tokens_raw<true>, grid, dim3(kThreads), 0, s, k);
    else hipLaunchKernelGGL(k_tokens_raw<false>, grid, dim3(kThreads), 0, s, k);
    return check_launch("k_tokens_raw");
}

// Pass 2: the (P, B, C) one-hot as the flat expansion of a (P, pitch) raw token matrix.
bsq_status launch_expansion(const uint8_t *tokens, int64_t pitch, int64_t B, int64_t P, int32_t C, size_t sz,
                            uint64_t one_bits, void *out, hipStream_t s, int64_t row_gap = 0) {
    EParams e;
    e.row_gap = row_gap;
    e.tok = tokens;
    e.B = B;
    e.Bp = pitch;
    e.out = static_cast<uint8_t *>(out);
    e.total = P * B * C * int64_t(sz);
    e.head = int32_t(reinterpret_cast<uintptr_t>(out) & (kChunk - 1));
    e.nchunks = (e.head + e.total + kChunk - 1) / kChunk;
    e.C = C;
    e.one_bits = one_bits;
    e.inv_rowbytes = 1.0 / double(C * int64_t(sz));
    e.inv_B = 1.0 / double(B);
    e.pitch = B * C * int64_t(sz);
    e.dv_pitch = div64_constants(uint64_t(e.pitch));
    e.dv_rb = div64_constants(uint64_t(C * int64_t(sz)));
    e.claim = nullptr;
    e.groups_per_class = 0;
    div_constants(uint32_t(C * int64_t(sz)), &e.rb_magic, &e.rb_shift, &e.rb_pow2);
    e.force4 = bsq_internal::tuning().expand_slots == 4;
    e.mode = bsq_internal::tuning().expand_mode;
    switch (sz) {
    case 1: return launch_expand<uint8_t>(e, s);
    case 2: return launch_expand<uint16_t>(e, s);
    case 4: return launch_expand<uint32_t>(e, s);
    default: return launch_expand<uint64_t>(e, s);
    }
}

bsq_status onehot_two_pass(KParams &k, size_t sz, void *workspace, hipStream_t s, int64_t row_gap = 0) {
    void *out = k.out;
    const int64_t pitch = two_pass_pitch(k.B);  // padded: every scratch row is aligned, full-width vector stores
    const bsq_status st = launch_tokens_raw(k, workspace, pitch, s);
    if (st != BSQ_OK) return st;
    return launch_expansion(static_cast<const uint8_t *>(workspace), pitch, k.B, k.P, k.C, sz, k.one_bits, out, s, row_gap);
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
    // Chunks per wave: 1.  The software-pipelined 4-chunk form (knob "tokenize_nch" = 4) is 15-20 % SLOWER on cfg2 /
    // cfg5: the kernel is bound by its ~550 VALU instructions per chunk, not by memory latency, and four chunks
    // per wave cost occupancy (102 VGPRs).
#ifdef BSQ_LABS
    const int nch = bsq_internal::tuning().tokenize_nch == 4 && !ragged ? 4 : 1;
#else
    const int nch = 1;
#endif
    const int64_t groups = ((c.nchunks + 7) / 8 + int64_t(4) * nch - 1) / (int64_t(4) * nch);
    if (groups * 8 >= (int64_t(1) << 31)) return bsq_internal::set_error(BSQ_ERR_INVALID_ARG, "output too large");
    const dim3 grid(unsigned(groups * 8));
    const int padv = bsq_internal::tuning().tokenize_pad;  // unused dynamic LDS = occupancy cap (experiments)
    const size_t pad = padv > 0 ? size_t(padv) : 0;
    const bool nt = bsq_internal::nontemporal_stores();
    if (ragged) {
        if (nt) hipLaunchKernelGGL((k_tokenize_chunks<T, true, HOT, 1, true>), grid, dim3(kThreads), pad, s, c);
        else hipLaunchKernelGGL((k_tokenize_chunks<T, false, HOT, 1, true>), grid, dim3(kThreads), pad, s, c);
#ifdef BSQ_LABS
    } else if (nch == 4) {
        if (nt) hipLaunchKernelGGL((k_tokenize_chunks<T, true, HOT, 4>), grid, dim3(kThreads), pad, s, c);
        else hipLaunchKernelGGL((k_tokenize_chunks<T, false, HOT, 4>), grid, dim3(kThreads), pad, s, c);
#endif
    } else {
        if (nt) hipLaunchKernelGGL((k_tokenize_chunks<T, true, HOT, 1>), grid, dim3(kThreads), pad, s, c);
        else hipLaunchKernelGGL((k_tokenize_chunks<T, false, HOT, 1>), grid, dim3(kThreads), pad, s, c);
    }
    return check_launch(HOT ? "k_tokenize_chunks<onehot bcl>" : "k_tokenize_chunks");
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
static int choose_onehot_path(int32_t C, size_t sz, int64_t B, int64_t P, bool misaligned_out = false) {
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
        const bool owner_small = total <= (int64_t(8) << 20) || (rowbytes >= 24 && total <= (int64_t(128) << 20));
        // (end of round 2: two-pass is 3-4 % ahead from 2 GB on -- 32768 x 1024 AMINO20 f32 0.376 vs 0.389 ms --, level at
        // 1.3 GB and behind below: profiles/r02/sweep_shapes_final.txt)
        if ((owner_big || owner_small) && total < (int64_t(3) << 29))
            path = 3;
        else if ((rowbytes >= (sz == 1 ? 8 : 16) && total >= (int64_t(192) << 20)) || ((pitch % 64 != 0 || misaligned_out) && total >= (int64_t(32) << 20)))
            path = 2;  // (second case: position rows that are not 64-byte aligned -- the tiles would share memory sectors or
                       // fall to element stores: 250001 x 256 int8 DNA 187 -> 120 us, profiles/r02/path_unaligned.txt; round 5: -> 94 us.
                       // Round 5: one-byte rows of 8 ... 15 bytes too -- their expansion is k_expand_rows1: SEB14 131072 x 512 int8
                       // 187 -> 156 us, SEB8 + BOS / EOS / PAD (11-byte rows) 281 -> 248 us; rows of 3 ... 7 bytes stay tiled unless
                       // misaligned: cfg4 int8 225 us tiled, 240 two-pass -- profiles/r05/rows1_lab.txt)
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
#ifdef BSQ_LABS
        const int mode = bsq_internal::tuning().expand_mode;
        if (rb >= 4 && (mode == 2 || mode == 9)) return "k_tokens_raw+k_expand_small";
#endif
        // (unmasked: the raw-id pass runs in k_tokens_pb8_fast unless a knob keeps it in k_tokens_raw -- see launch_tokens_raw)
        const bool rows1 = bsq_dtype_size(t) == 1 && rb >= 3 && rb <= 15 && bsq_internal::tuning().expand_rows1 != 1;
        if (bsq_internal::tuning().raw_mode == 0 && bsq_internal::tuning().tokens_pb8 != 1)
            return rows1 ? "k_tokens_pb8_fast<raw>+k_expand_rows1" : "k_tokens_pb8_fast<raw>+k_expand_chunks";
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
    const int path = choose_onehot_path(k.C, sz, B, P, reinterpret_cast<uintptr_t>(out) % 64 != 0);
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
        // The scratch is shared by the calls of one stream (workspace cache): the two launches of a call must be enqueued
        // back to back, so concurrent host threads take turns here (enqueueing takes microseconds; the GPU work overlaps).
        std::lock_guard<std::mutex> two_pass_turn(bsq_internal::workspace_mutex());
        void *ws = nullptr;
        bsq_status wst = bsq_internal::workspace_acquire(two_pass_workspace_bytes(B, P), s, &ws);
        if (wst != BSQ_OK) return wst;
        wst = onehot_two_pass(k, sz, ws, s);
        bsq_internal::workspace_release(ws, s);
        return wst;
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
    // A block whose position rows are whole 4-KiB chunks (B * C * sizeof(T) and the address of its first element multiples of 4096: e.g.
    // any multiple of 4096 sequences at a 4096-sequence boundary of an aligned tensor) is the two-pass stream with a gap after every
    // row: no chunk straddles two rows.  16 384-sequence blocks of cfg3: 4 x 0.19 ms against 4 x 0.25 ms for the tiles.
    const int64_t block_pitch = B * k.C * int64_t(sz), rb = k.C * int64_t(sz);
    if (block_path != 1 && rb >= 16 && !(block_pitch % kChunk == 0 && reinterpret_cast<uintptr_t>(out) % kChunk == 0)) {
        // Any other large block: the sequences up to the first one that starts a chunk of the result, the run of whole chunks behind
        // it (a multiple of 4096 / gcd(row bytes, 4096) sequences), the rest -- three calls, the middle one the fast stream; worth it
        // once the middle is large (the two short ones cost a ~15-us tile launch each).  Ragged shards of a sharded job
        // (sharding.store_shard_into_root), the last piece of a host batch, results that torch aligned to 512 bytes only.
        const int64_t mis = int64_t(reinterpret_cast<uintptr_t>(out) % kChunk);
        int64_t g = rb, h = kChunk;
        while (h) {
            const int64_t r = g % h;
            g = h;
            h = r;
        }
        const int64_t m = kChunk / g;  // sequences per period of (b * rb) mod 4096
        int64_t lead = -1;
        for (int64_t b = 0; b < m && lead < 0; ++b)
            if ((mis + b * rb) % kChunk == 0) lead = b;
        const int64_t main = lead >= 0 && lead < B ? (B - lead) / m * m : 0;
        if (main > 0 && main * rb * P >= (int64_t(128) << 20)) {
            uint8_t *o = static_cast<uint8_t *>(out);
            // (the two short pieces -- fewer than 4096 / gcd sequences each -- through the element kernel when they are small: the tiled
            //  kernel has a ~15-us floor, and two of them in front of and behind a 90-us stream cost the 1/8 shard of cfg4 f32 a third of
            //  its time, 127 vs 91 us: profiles/r05/bench_default.json, cfg4f_shard8.into_root)
            auto side = [&](const int64_t *offs, int64_t n, uint8_t *dst) {
                if (n * P * k.C <= (int64_t(1) << 21)) return bsq_internal::onehot_generic_block(d, chars, offs, mask_or_null, n, P, t, dst, row_seqs, hip_stream);
                return bsq_onehot_block_device(d, chars, offs, mask_or_null, n, P, t, dst, row_seqs, hip_stream);
            };
            if (lead > 0) {
                st = side(offsets, lead, o);
                if (st != BSQ_OK) return st;
            }
            st = bsq_onehot_block_device(d, chars, offsets + lead, mask_or_null, main, P, t, o + lead * rb, row_seqs, hip_stream);
            if (st != BSQ_OK || lead + main == B) return st;
            return side(offsets + lead + main, B - lead - main, o + (lead + main) * rb);
        }
    }
    if (block_path != 1 && rb >= 16 && block_pitch % kChunk == 0 && reinterpret_cast<uintptr_t>(out) % kChunk == 0) {
        std::lock_guard<std::mutex> two_pass_turn(bsq_internal::workspace_mutex());
        void *ws = nullptr;
        bsq_status wst = bsq_internal::workspace_acquire(two_pass_workspace_bytes(B, P), s, &ws);
        if (wst != BSQ_OK) return wst;
        wst = onehot_two_pass(k, sz, ws, s, (row_seqs - B) * k.C * int64_t(sz));
        bsq_internal::workspace_release(ws, s);
        return wst;
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
        // knob "raw_mode": 0 automatic, 1 the 256 x 64 tile, 4 the wide 1024 x 16 tile, 2 / 3 the round-2 experiments
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
#ifdef BSQ_LABS
        if (rm == 2 || rm == 3) {
            if (rm == 3 && k.foldable) hipLaunchKernelGGL((k_tokens_raw2<false, true>), vgrid, dim3(kThreads), 0, s, k);
            else hipLaunchKernelGGL((k_tokens_raw2<false, false>), vgrid, dim3(kThreads), 0, s, k);
            return check_launch("k_tokens_raw2<value>");
        }
#endif
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
        std::lock_guard<std::mutex> two_pass_turn(bsq_internal::workspace_mutex());  // see bsq_onehot_device
        void *ws = nullptr;
        bsq_status wst = bsq_internal::workspace_acquire(size_t(B) * size_t(P), s, &ws);
        if (wst != BSQ_OK) return wst;
        wst = bsq_internal::launch_tokens_bp8(d, chars, offsets, B, P, ws, s, true, mask_or_null);
        if (wst == BSQ_OK) {
            const uint8_t *tk = static_cast<const uint8_t *>(ws);
            switch (sz) {
            case 1: wst = launch_expand_bcl<uint8_t>(tk, B, P, k.C, k.one_bits, out, s); break;
            case 2: wst = launch_expand_bcl<uint16_t>(tk, B, P, k.C, k.one_bits, out, s); break;
            case 4: wst = launch_expand_bcl<uint32_t>(tk, B, P, k.C, k.one_bits, out, s); break;
            default: wst = launch_expand_bcl<uint64_t>(tk, B, P, k.C, k.one_bits, out, s); break;
            }
        }
        bsq_internal::workspace_release(ws, s);
        return wst;
    }
    if (k.C <= 250 && reinterpret_cast<uintptr_t>(out) % sz == 0 &&
        ((reinterpret_cast<uintptr_t>(out) % 16 == 0 && P % int64_t(16 / sz) == 0) || bcl_path != 3)) {  // knob 3: aligned only
        switch (sz) {
        case 1: return launch_tokenize_chunks<uint8_t, true>(k, s);
        case 2: return launch_tokenize_chunks<uint16_t, true>(k, s);
        case 4: return launch_tokenize_chunks<uint32_t, true>(k, s);
        default: return launch_tokenize_chunks<uint64_t, true>(k, s);
        }
    }
    return bsq_internal::onehot_generic_bcl(d, chars, offsets, mask_or_null, B, P, t, out, hip_stream);
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
