// Device-side helpers shared by the kernel translation units of libbsq_hip.so (gfx950 only; not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace bsq_dev {

constexpr int kThreads = 256;      // 4 waves of 64
constexpr uint32_t kNone = 0xFFu;  // "no token": all-zero one-hot row / token value 0
constexpr int kChunk = 4096;       // the unit of the streaming kernels: one naturally aligned 4-KiB piece of the output

// One 16-byte store, non-temporal if NT (global_store_dwordx4 ... nt).  The value goes through ONE vector-typed
// nontemporal store: four scalar ones only stay `nt` if the compiler happens to merge them unchanged (it dropped
// the flag for the 8-byte element types, which cost the f64 token matrix 40 % of its bandwidth).
// (Write-through `sc1` stores were tried in round 2: a pure 64-128 MiB store stream runs 15-20 % faster with them,
// every kernel that also reads runs the same or slower, multi-GB outputs much slower: profiles/r02/store_kinds_bench.txt, tokens8_lab4.txt.)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <bool NT>
__device__ __forceinline__ void store16(void *dst, const uint4 &v) {
    if constexpr (NT) {
        const u32x4 x = {v.x, v.y, v.z, v.w};
        __builtin_nontemporal_store(x, reinterpret_cast<u32x4 *>(dst));
    } else {
        *reinterpret_cast<uint4 *>(dst) = v;
    }
}

// The same at ANY byte alignment (gfx950 splits an unaligned vector access in hardware), and the first `nbytes`
// (wave-uniform, < 16) bytes of v as 8- / 4- / 2- / 1-byte stores from the lanes with `on` set: the row-piece forms of
// the (B,P) chunk kernels (padlen not a multiple of a lane's 16 bytes, misaligned outputs).
typedef uint32_t u32x4_unaligned __attribute__((ext_vector_type(4), aligned(1)));
typedef uint64_t u64_unaligned __attribute__((aligned(1)));
typedef uint32_t u32_unaligned __attribute__((aligned(1)));
typedef uint16_t u16_unaligned __attribute__((aligned(1)));
template <bool NT>
__device__ __forceinline__ void store16_unaligned(void *dst, const uint4 &v) {
    const u32x4 x = {v.x, v.y, v.z, v.w};
    if constexpr (NT) __builtin_nontemporal_store(x, reinterpret_cast<u32x4_unaligned *>(dst));
    else *reinterpret_cast<u32x4_unaligned *>(dst) = x;
}
__device__ __forceinline__ void store_head_bytes(void *dst, const uint4 &v, uint32_t nbytes, bool on) {
    uint8_t *d = static_cast<uint8_t *>(dst);
    uint64_t w = (static_cast<uint64_t>(v.y) << 32) | v.x;
    if (nbytes & 8u) {
        if (on) *reinterpret_cast<u64_unaligned *>(d) = w;
        w = (static_cast<uint64_t>(v.w) << 32) | v.z;
        d += 8;
    }
    if (nbytes & 4u) {
        if (on) *reinterpret_cast<u32_unaligned *>(d) = static_cast<uint32_t>(w);
        w >>= 32;
        d += 4;
    }
    if (nbytes & 2u) {
        if (on) *reinterpret_cast<u16_unaligned *>(d) = static_cast<uint16_t>(w);
        w >>= 16;
        d += 2;
    }
    if ((nbytes & 1u) && on) *d = static_cast<uint8_t>(w);
}

// The same with a LANE-VARYING byte count (0 .. 15, a multiple of the element size SZ; no branches, up to four
// predicated stores).
template <int SZ>
__device__ __forceinline__ void store_head_bytes_var(void *dst, const uint4 &v, uint32_t nbytes) {
    uint8_t *d = static_cast<uint8_t *>(dst);
    uint64_t w = (static_cast<uint64_t>(v.y) << 32) | v.x;
    const bool c8 = (nbytes & 8u) != 0, c4 = SZ <= 4 && (nbytes & 4u) != 0, c2 = SZ <= 2 && (nbytes & 2u) != 0,
               c1 = SZ <= 1 && (nbytes & 1u) != 0;
    if (c8) *reinterpret_cast<u64_unaligned *>(d) = w;
    w = c8 ? ((static_cast<uint64_t>(v.w) << 32) | v.z) : w;
    d += c8 ? 8 : 0;
    if (c4) *reinterpret_cast<u32_unaligned *>(d) = static_cast<uint32_t>(w);
    w = c4 ? (w >> 32) : w;
    d += c4 ? 4 : 0;
    if (c2) *reinterpret_cast<u16_unaligned *>(d) = static_cast<uint16_t>(w);
    w = c2 ? (w >> 16) : w;
    d += c2 ? 2 : 0;
    if (c1) *d = static_cast<uint8_t>(w);
}

// floor(n / d) for n < 2^31 with the constants of div_constants() (round-up method: exact below 2^31).
// (host + device: bsq_selftest_index_math() runs the very same code on the CPU)
__host__ __device__ __forceinline__ uint32_t fast_div(uint32_t n, uint32_t magic, uint32_t shift, uint32_t pow2) {
    return pow2 ? n >> shift : static_cast<uint32_t>((static_cast<uint64_t>(n) * magic) >> 32) >> shift;  // v_mul_hi_u32
}

// Constants of fast_div(): floor(n / d) == mulhi(n, magic) >> shift for every n < 2^31 (round-up method:
// magic = floor(2^(32+shift) / d) + 1 with shift = floor(log2 d); the error term n / 2^(32+shift) stays
// below 1/d because d < 2^(shift+1)); powers of two are plain shifts.  1 <= d <= 2^30.
inline void div_constants(uint32_t d, uint32_t *magic, uint32_t *shift, uint32_t *pow2) {
    uint32_t sh = 0;
    while ((uint64_t(2) << sh) <= d) ++sh;
    *shift = sh;
    *pow2 = (d & (d - 1)) == 0;
    *magic = *pow2 ? 0u : uint32_t((uint64_t(1) << (32 + sh)) / d + 1);
}

// floor(n / d) and the remainder for 0 <= n < 2^52, 1 <= d < 2^31 through one double multiply with
// inv = 1.0 / d (computed on the host): the product is within 1 of n / d, one correction step makes it
// exact.  Replaces the ~120-instruction 64-bit integer division the chunk kernels would otherwise run
// twice per wave (byte offset -> row, row -> position).
__host__ __device__ __forceinline__ int64_t div_by(int64_t n, int64_t d, double inv, int64_t *rem) {
    int64_t q = static_cast<int64_t>(static_cast<double>(n) * inv);
    int64_t r = n - q * d;
    if (r < 0) {
        q -= 1;
        r += d;
    } else if (r >= d) {
        q += 1;
        r -= d;
    }
    *rem = r;
    return q;
}

// floor(n / d) for 0 <= n < 2^63 and 1 <= d < 2^63 by ONE 64 x 64 -> high-64 multiply with a precomputed magic
// (round-up method: magic = floor(2^(64+s) / d) + 1 with s = floor(log2 d) lies in (2^63, 2^64]; the error term
// n * e / 2^(64+s), e <= 1, stays below 1/d while n < 2^63); powers of two are shifts.  On WAVE-UNIFORM operands the
// whole thing is scalar-ALU work (s_mul_hi_u32 / s_mul_i32), unlike div_by() whose double arithmetic always runs on
// the vector ALU at the FP64 rate -- at 12 resident waves per CU that chain was ~15 % of a chunk wave's lifetime.
struct Div64 {
    uint64_t magic;
    uint32_t shift, pow2;
};
inline Div64 div64_constants(uint64_t d) {
    Div64 c;
    uint32_t sh = 0;
    while (sh < 62 && (uint64_t(2) << sh) <= d) ++sh;
    c.shift = sh;
    c.pow2 = (d & (d - 1)) == 0;
    c.magic = c.pow2 ? 0 : uint64_t((static_cast<unsigned __int128>(1) << (64 + sh)) / d + 1);
    return c;
}
__host__ __device__ __forceinline__ uint64_t div64(uint64_t n, const Div64 &c) {
#if defined(__HIP_DEVICE_COMPILE__)
    return c.pow2 ? n >> c.shift : __umul64hi(n, c.magic) >> c.shift;
#else
    return c.pow2 ? n >> c.shift : uint64_t((static_cast<unsigned __int128>(n) * c.magic) >> 64) >> c.shift;
#endif
}

}  // namespace bsq_dev
