// Measurement-only entry points of include/bsq_diag.h: write-bandwidth yardsticks (k_fill*), the store-pattern lab
// (k_fill_pattern), the XCD placement probe and the host-side self-test of the kernels' index arithmetic.  gfx950 only.
// Nothing here is on the encode path.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <mutex>

#include "bsq.h"
#include "bsq_diag.h"
#include "bsq_device.h"
#include "bsq_internal.h"

namespace {

using namespace bsq_dev;

bsq_status check_launch(const char *what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return bsq_internal::set_hip_error(what, e);
    return BSQ_OK;
}

__global__ __launch_bounds__(kThreads) void k_fill(uint4 *dst, size_t n16, uint32_t pattern) {
    const size_t stride = static_cast<size_t>(gridDim.x) * kThreads;
    const uint4 v{pattern, pattern, pattern, pattern};
    for (size_t i = static_cast<size_t>(blockIdx.x) * kThreads + threadIdx.x; i < n16; i += stride) dst[i] = v;
}

// Write-pattern experiments (fill_mode 1..4); mode 0 is k_fill above.
//  1/3: one 16-byte store per thread, one block per 4 KiB (plain / nt)
//  2/4: one block per 16 KiB, 4 stores per thread 4 KiB apart (plain / nt)
template <int PER_THREAD, bool NT>
__global__ __launch_bounds__(kThreads) void k_fill_blocks(uint4 *dst, size_t n16, uint32_t pattern) {
    const uint4 v{pattern, pattern, pattern, pattern};
    const size_t base = static_cast<size_t>(blockIdx.x) * (kThreads * PER_THREAD) + threadIdx.x;
#pragma unroll
    for (int k = 0; k < PER_THREAD; ++k) {
        const size_t i = base + static_cast<size_t>(k) * kThreads;
        if (i < n16) store16<NT>(dst + i, v);
    }
}

// Diagnostic: the WRITE PATTERN of the tiled one-hot kernel without any of its work.  The output is a
// (rows, pitch) byte matrix; block (cb, rb) owns columns [cb*seg, (cb+1)*seg) of 4*rpw rows; each of its
// 4 waves writes rpw rows (contiguous rows if !interleave), `seg` contiguous bytes per row.
template <bool NT>
__global__ __launch_bounds__(kThreads) void k_fill_pattern(uint8_t *dst, int64_t rows, int64_t pitch, int32_t seg,
                                                           int32_t rpw, int32_t ncb, int32_t nrb, int32_t order,
                                                           int32_t interleave, int32_t wait) {
    int32_t cb, rb;
    if (order >= 2) {  // permutations of the block -> column-chunk map (ncb must be a multiple of 64)
        const uint32_t b = blockIdx.x % ncb;
        rb = blockIdx.x / ncb;
        uint32_t c = b;
        if (order == 2) c = (b & ~7u) | ((b + (b >> 3)) & 7u);            // rotate the mod-8 class per group of 8
        if (order == 3) c = b ^ 1u;                                        // swap neighbours
        if (order == 4) c = (b & ~63u) | (__brev(b & 63u) >> 26);          // bit-reverse inside 64-chunk windows
        if (order == 5) c = (b & ~7u) | ((b + 1u) & 7u);                   // constant rotation of the class
        if (order == 6) c = (b & ~63u) | (((b & 7u) << 3) | ((b >> 3) & 7u));  // transpose 8x8 inside windows
        cb = static_cast<int32_t>(c);
    } else if (order == 0) {
        cb = blockIdx.x % ncb;
        rb = blockIdx.x / ncb;
    } else {
        rb = blockIdx.x % nrb;
        cb = blockIdx.x / nrb;
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint4 v{1, 2, 3, 4};
    if (interleave == 2) {  // row-wise: the 4 waves write 4 adjacent segments of the SAME row
        for (int r = 0; r < rpw; ++r) {
            const int64_t row = static_cast<int64_t>(rb) * rpw + r;
            if (row >= rows) break;
            uint8_t *g = dst + row * pitch + (static_cast<int64_t>(cb) * 4 + wave) * seg;
            for (int32_t o = lane * 16; o < seg; o += 1024) store16<NT>(g + o, v);
        }
        return;
    }
    for (int r = 0; r < rpw; ++r) {
        const int64_t row = static_cast<int64_t>(rb) * 4 * rpw + (interleave ? r * 4 + wave : wave * rpw + r);
        if (row >= rows) break;
        uint8_t *g = dst + row * pitch + static_cast<int64_t>(cb) * seg;
        for (int32_t o = lane * 16; o < seg; o += 1024) store16<NT>(g + o, v);
        // knob "pattern_wait" n > 0: at most n - 1 (0 / 1 / 2 / 4) stores of the wave in flight before its next row
        if (wait == 1) __builtin_amdgcn_s_waitcnt(0x0F70);
        else if (wait == 2) __builtin_amdgcn_s_waitcnt(0x0F71);
        else if (wait == 3) __builtin_amdgcn_s_waitcnt(0x0F72);
        else if (wait == 5) __builtin_amdgcn_s_waitcnt(0x0F74);
    }
}

// Diagnostic: a read + write stream of the (B,P) int8 token kernel's SHAPE with none of its work -- the yardstick the
// token kernels of cfg2 / cfg5 are quoted against next to the plain fill.  Wave k writes the aligned 4-KiB chunk k of
// dst (class k % 8 pinned to its XCD as in k_tokens_bp8) and reads its share of src (ceil(nsrc16 / nchunks) 16-byte
// pieces, coalesced).  mode 0: loads, then stores of the loaded data (one dependent step); mode 1: an 8-byte load of
// src first, whose (zeroed) value is added to the load addresses: two dependent steps, like offsets -> characters;
// mode 2: the stores do not wait for the loads (their data only reaches a never-taken store at the end); mode 3: the loads only.
template <bool NT>
__global__ __launch_bounds__(kThreads) void k_copy_mix(uint4 *dst, int64_t nchunks, const uint4 *src, int64_t nsrc16,
                                                       int32_t per_chunk, int32_t mode) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int64_t k = static_cast<int64_t>(blockIdx.x & 7u) + 8 * (static_cast<int64_t>(blockIdx.x >> 3) * 4 + wave);
    if (k >= nchunks) return;
    int64_t base = k * per_chunk;
    if (mode == 1) {
        const uint64_t dep = reinterpret_cast<const uint64_t *>(src)[(k * 2) % (nsrc16 * 2)];
        base += dep == 0x123456789ABCDEF1ull ? 1 : 0;  // practically always 0, but the addresses now wait for the load
    }
    uint4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int64_t i = base + u * 64 + lane;
        v[u] = (u * 64 + lane < per_chunk && i < nsrc16) ? src[i] : uint4{1, 2, 3, 4};
    }
    uint4 *d = dst + k * (kChunk / 16) + lane;
    if (mode == 3) {  // loads ONLY (the read half of the stream by itself: what a read phase without any store traffic runs at)
        const uint32_t x = v[0].x & v[1].y & v[2].z & v[3].w;
        if (x == 0x9E3779B9u) store16<NT>(d, v[0]);  // keeps the loads alive; practically never taken
        return;
    }
    if (mode == 2) {
        const uint4 c{5, 6, 7, 8};
#pragma unroll
        for (int u = 0; u < 4; ++u) store16<NT>(d + u * 64, c);
        const uint32_t x = v[0].x & v[1].y & v[2].z & v[3].w;
        if (x == 0x9E3779B9u) store16<NT>(d, v[0]);  // keeps the loads alive
        return;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) store16<NT>(d + u * 64, uint4{v[u].x | v[(u + 1) & 3].x, v[u].y, v[u].z, v[u].w});
}

// The same stream with PERSISTENT workgroups (round 5; VERDICT round 4, item 1: the two forms the cold-input regime had not seen).
// mode 4 -- register pipeline: a wave walks its XCD class's chunks (k, k + 8 * stride, ...) and issues the loads of its NEXT chunk
//   before the stores of the current one (the loads are older than the stores in the wave's vmcnt order, so the next iteration's
//   wait does not wait for the stores).
// mode 5 -- LDS-DMA loader wave: wave 0 of a workgroup stages the source pieces of its three consumer waves' chunks into an LDS ring
//   with global_load_lds_dwordx4 (no VGPR destination: D steps = 3 * D chunks in flight per workgroup whatever the consumers do),
//   publishes a step with a counted s_waitcnt vmcnt + s_barrier; consumers ds_read_b128 their 4 KiB and stream it out.  Ring of
//   D + 1 slots of 12 KiB.
// Both keep the chunk class = blockIdx % 8.  `wgs` workgroups per XCD.
template <bool NT>
__global__ __launch_bounds__(kThreads) void k_copy_persist(uint4 *dst, int64_t nchunks, const uint4 *src, int64_t nsrc16, int32_t per_chunk,
                                                           int32_t wgs) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int64_t cls = blockIdx.x & 7u, stride = static_cast<int64_t>(wgs) * 4;
    int64_t slot = static_cast<int64_t>(blockIdx.x >> 3) * 4 + wave;
    if (cls + 8 * slot >= nchunks) return;
    // (unconditional loads at clamped indices: predicated ones put every load in a basic block of its own, and the compiler then waits
    //  vmcnt(0) at the joins -- no load of the next chunk would survive the stores of this one)
    const int32_t piece = lane < per_chunk ? lane : per_chunk - 1;
    auto load = [&](int64_t k, uint4 (&v)[4]) {
        const int64_t base = k * per_chunk;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            int64_t i = base + (u * 64 + piece < per_chunk ? u * 64 + piece : per_chunk - 1);
            i = i < nsrc16 ? i : nsrc16 - 1;
            v[u] = src[i];
        }
    };
    auto store = [&](int64_t k, const uint4 (&v)[4]) {
        uint4 *d = dst + k * (kChunk / 16) + lane;
#pragma unroll
        for (int u = 0; u < 4; ++u) store16<NT>(d + u * 64, uint4{v[u].x | v[(u + 1) & 3].x, v[u].y, v[u].z, v[u].w});
    };
    // two register sets taking turns (no copies: a copy at the end of an iteration would wait for the loads it has just issued)
    uint4 a[4], b[4];
    load(cls + 8 * slot, a);
    for (;;) {
        const int64_t k = cls + 8 * slot, k1 = cls + 8 * (slot + stride), k2 = cls + 8 * (slot + 2 * stride);
        const bool more1 = k1 < nchunks, more2 = k2 < nchunks;  // wave-uniform
        load(more1 ? k1 : k, b);
        store(k, a);
        if (!more1) break;
        load(more2 ? k2 : k, a);
        store(k1, b);
        if (!more2) break;
        slot += 2 * stride;
    }
}

typedef __attribute__((address_space(3))) void bsq_lds_void;
typedef __attribute__((address_space(1))) const void bsq_glb_void;
template <bool NT, int D>
__global__ __launch_bounds__(kThreads) void k_copy_lds(uint4 *dst, int64_t nchunks, const uint4 *src, int64_t nsrc16, int32_t per_chunk,
                                                       int32_t wgs) {
    constexpr int R = D + 1;
    __shared__ __align__(16) uint4 ring[R][3][256];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int64_t cls = blockIdx.x & 7u, g = blockIdx.x >> 3;
    const int64_t per_class = (nchunks - cls + 7) / 8;                  // slots of this class
    const int64_t step_slots = static_cast<int64_t>(wgs) * 3;           // slots all workgroups of the class take per step
    const int32_t nsteps = static_cast<int32_t>((per_class - g * 3 + step_slots - 1) / step_slots);  // (<= 0: nothing to do)
    if (nsteps <= 0) return;
    auto chunk_of = [&](int32_t s, int c) { return cls + 8 * ((g + static_cast<int64_t>(s) * wgs) * 3 + c); };
    if (wave == 0) {  // ---- loader ----
        auto issue = [&](int32_t s) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int64_t k = chunk_of(s, c);
                const int64_t base = k * per_chunk;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    int64_t i = base + u * 64 + lane;
                    if (!(k < nchunks && u * 64 + lane < per_chunk && i < nsrc16)) i = 0;  // (always 12 DMA instructions per step: the waits count them)
                    __builtin_amdgcn_global_load_lds((bsq_glb_void *)(src + i), (bsq_lds_void *)(&ring[s % R][c][u * 64]), 16, 0, 0);
                }
            }
        };
        for (int32_t s = 0; s < D && s < nsteps; ++s) issue(s);
        for (int32_t s = 0; s < nsteps; ++s) {
            // steps s .. min(s + D, nsteps) - 1 are in flight: step s has landed once all but the younger ones' 12 instructions each are done
            const int32_t younger = (s + D <= nsteps ? D : nsteps - s) - 1;
            if (younger >= 2) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
            else if (younger == 1) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();  // step s published; every consumer has finished reading step s - 1
            if (s + D < nsteps) issue(s + D);  // into the slot of step s - 1
        }
    } else {          // ---- consumers ----
        const int c = wave - 1;
        for (int32_t s = 0; s < nsteps; ++s) {
            __builtin_amdgcn_s_barrier();
            const int64_t k = chunk_of(s, c);
            if (k < nchunks) {
                uint4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = ring[s % R][c][u * 64 + lane];
                uint4 *d = dst + k * (kChunk / 16) + lane;
#pragma unroll
                for (int u = 0; u < 4; ++u) store16<NT>(d + u * 64, uint4{v[u].x | v[(u + 1) & 3].x, v[u].y, v[u].z, v[u].w});
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the reads of this slot are done before the next barrier
        }
    }
}

// Diagnostic: the XCD every block of a 1-D grid ran on (HW_REG_XCC_ID, 0..7).  The chunk kernels rely -- for
// speed only -- on blocks b and b + 8 sharing an XCD; this records what the dispatcher actually did.
__global__ __launch_bounds__(kThreads) void k_xcd_probe(int32_t *xcd) {
    uint32_t id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    if (threadIdx.x == 0) xcd[blockIdx.x] = static_cast<int32_t>(id & 0xFu);
}

}  // namespace

extern "C" {

bsq_status bsq_fill_device(void *dst, size_t nbytes, uint32_t pattern, void *hip_stream) {
    if (!dst || nbytes % 16 || reinterpret_cast<uintptr_t>(dst) % 16)
        return bsq_internal::set_error(BSQ_ERR_INVALID_ARG, "fill needs a 16-byte aligned pointer and size");
    if (nbytes == 0) return BSQ_OK;
    const size_t n16 = nbytes / 16;
    const size_t blocks = (n16 + kThreads - 1) / kThreads;
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    uint4 *d4 = static_cast<uint4 *>(dst);
    const int mode = bsq_internal::tuning().fill_mode;
    if (blocks >= (size_t(1) << 31)) return bsq_internal::set_error(BSQ_ERR_INVALID_ARG, "fill too large");
    switch (mode) {
    case 1: hipLaunchKernelGGL((k_fill_blocks<1, false>), dim3(unsigned(blocks)), dim3(kThreads),
                               size_t(bsq_internal::tuning().fill_pad), s, d4, n16, pattern); break;
    case 2: hipLaunchKernelGGL((k_fill_blocks<4, false>), dim3(unsigned((blocks + 3) / 4)), dim3(kThreads), 0, s, d4, n16, pattern); break;
    case 3: hipLaunchKernelGGL((k_fill_blocks<1, true>), dim3(unsigned(blocks)), dim3(kThreads),
                               size_t(bsq_internal::tuning().fill_pad), s, d4, n16, pattern); break;
    case 4: hipLaunchKernelGGL((k_fill_blocks<4, true>), dim3(unsigned((blocks + 3) / 4)), dim3(kThreads), 0, s, d4, n16, pattern); break;
    default: {
        const unsigned grid = unsigned(blocks > 256 * 16 ? 256 * 16 : blocks);
        hipLaunchKernelGGL(k_fill, dim3(grid), dim3(kThreads), 0, s, d4, n16, pattern);
    }
    }
    return check_launch("k_fill");
}

bsq_status bsq_fill_pattern_device(void *dst, int64_t rows, int64_t pitch, int32_t seg, int32_t rows_per_wave,
                                   int32_t order, int32_t interleave, int32_t nt, void *hip_stream) {
    if (!dst || rows <= 0 || pitch <= 0 || seg <= 0 || seg % 16 || pitch % seg || rows_per_wave <= 0)
        return bsq_internal::set_error(BSQ_ERR_INVALID_ARG, "bad fill pattern");
    int32_t ncb = int32_t(pitch / seg);
    int32_t nrb = int32_t((rows + 4 * rows_per_wave - 1) / (4 * rows_per_wave));
    if (interleave == 2) {
        if (ncb % 4) return bsq_internal::set_error(BSQ_ERR_INVALID_ARG, "row-wise pattern needs pitch % (4*seg) == 0");
        ncb /= 4;
        nrb = int32_t((rows + rows_per_wave - 1) / rows_per_wave);
    }
    const dim3 grid(unsigned(int64_t(ncb) * nrb));
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    const size_t pad = size_t(bsq_internal::tuning().fill_pad);  // unused dynamic LDS: caps the workgroups per CU
    const int32_t wait = bsq_internal::tuning().pattern_wait;
    if (nt)
        hipLaunchKernelGGL((k_fill_pattern<true>), grid, dim3(kThreads), pad, s, static_cast<uint8_t *>(dst), rows, pitch,
                           seg, rows_per_wave, ncb, nrb, order, interleave, wait);
    else
        hipLaunchKernelGGL((k_fill_pattern<false>), grid, dim3(kThreads), pad, s, static_cast<uint8_t *>(dst), rows, pitch,
                           seg, rows_per_wave, ncb, nrb, order, interleave, wait);
    return check_launch("k_fill_pattern");
}

bsq_status bsq_copy_mix_device(void *dst, size_t dst_bytes, const void *src, size_t src_bytes, int32_t mode, int32_t nt,
                               void *hip_stream) {
    if (!dst || !src || dst_bytes % kChunk || src_bytes % 16 || src_bytes < 16 || reinterpret_cast<uintptr_t>(dst) % 16 ||
        reinterpret_cast<uintptr_t>(src) % 16)
        return bsq_internal::set_error(BSQ_ERR_INVALID_ARG, "copy mix: dst a multiple of 4 KiB, src of 16 bytes, both 16-byte aligned");
    if (dst_bytes == 0) return BSQ_OK;
    const int64_t nchunks = int64_t(dst_bytes / kChunk), nsrc16 = int64_t(src_bytes / 16);
    const int64_t per_chunk = (nsrc16 + nchunks - 1) / nchunks;
    if (per_chunk > 256) return bsq_internal::set_error(BSQ_ERR_INVALID_ARG, "copy mix: src must not exceed dst");
    const int64_t groups = ((nchunks + 7) / 8 + 3) / 4;
    if (groups * 8 >= (int64_t(1) << 31)) return bsq_internal::set_error(BSQ_ERR_INVALID_ARG, "copy mix too large");
    const dim3 grid(unsigned(groups * 8));
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    const size_t pad = size_t(bsq_internal::tuning().fill_pad);
    if (mode == 4 || (mode >= 5 && mode <= 7)) {  // persistent forms: knob fill_mode = workgroups per CU (default 4); mode 5 / 6 / 7: D = 1 / 2 / 3
        const int per_cu = bsq_internal::tuning().fill_mode > 0 ? bsq_internal::tuning().fill_mode : 4;
        const int32_t wgs = 32 * per_cu;  // per XCD (32 CUs)
        const dim3 pgrid(unsigned(wgs) * 8u);
        uint4 *d4 = static_cast<uint4 *>(dst);
        const uint4 *s4 = static_cast<const uint4 *>(src);
#define BSQ_PL(KERNEL) hipLaunchKernelGGL(KERNEL, pgrid, dim3(kThreads), pad, s, d4, nchunks, s4, nsrc16, int32_t(per_chunk), wgs)
        if (mode == 4) { if (nt) BSQ_PL((k_copy_persist<true>)); else BSQ_PL((k_copy_persist<false>)); }
        else if (mode == 5) { if (nt) BSQ_PL((k_copy_lds<true, 1>)); else BSQ_PL((k_copy_lds<false, 1>)); }
        else if (mode == 6) { if (nt) BSQ_PL((k_copy_lds<true, 2>)); else BSQ_PL((k_copy_lds<false, 2>)); }
        else { if (nt) BSQ_PL((k_copy_lds<true, 3>)); else BSQ_PL((k_copy_lds<false, 3>)); }
#undef BSQ_PL
        return check_launch("k_copy_persist / k_copy_lds");
    }
    if (nt)
        hipLaunchKernelGGL((k_copy_mix<true>), grid, dim3(kThreads), pad, s, static_cast<uint4 *>(dst), nchunks,
                           static_cast<const uint4 *>(src), nsrc16, int32_t(per_chunk), mode);
    else
        hipLaunchKernelGGL((k_copy_mix<false>), grid, dim3(kThreads), pad, s, static_cast<uint4 *>(dst), nchunks,
                           static_cast<const uint4 *>(src), nsrc16, int32_t(per_chunk), mode);
    return check_launch("k_copy_mix");
}

// Host-side self-test of the kernels' division-free index arithmetic (the same inline functions the device runs):
// fast_div against n / d for dividends below 2^31, div_by against the integer quotient / remainder below 2^52.
// Returns 0, or a non-zero code that identifies the first failing case.
int64_t bsq_selftest_index_math(void) {
    uint64_t x = 0x9E3779B97F4A7C15ull;
    auto next = [&x]() {  // splitmix64
        x += 0x9E3779B97F4A7C15ull;
        uint64_t z = x;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    };
    for (int it = 0; it < 20000; ++it) {
        uint32_t d;
        switch (it % 5) {
        case 0: d = uint32_t(it / 5 + 1); break;                                    // 1, 2, 3, ...
        case 1: d = uint32_t(1) << (it / 5 % 31); break;                            // powers of two
        case 2: d = (uint32_t(1) << (it / 5 % 30 + 1)) - 1; break;                  // 2^k - 1
        case 3: d = (uint32_t(1) << (it / 5 % 29 + 1)) + 1; break;                  // 2^k + 1
        default: d = uint32_t(next() % (uint64_t(1) << 30)) + 1; break;             // random <= 2^30
        }
        if (d > (uint32_t(1) << 30)) d = uint32_t(1) << 30;
        uint32_t magic, shift, pow2;
        div_constants(d, &magic, &shift, &pow2);
        const double inv = 1.0 / double(d);
        for (int j = 0; j < 64; ++j) {
            uint32_t n;
            const uint64_t qmax = ((uint64_t(1) << 31) - 1) / d;
            switch (j % 4) {
            case 0: n = uint32_t(next() & 0x7FFFFFFFu); break;
            case 1: n = uint32_t((next() % (qmax + 1)) * d); break;                 // exact multiples
            case 2: { const uint64_t m = (next() % (qmax + 1)) * d; n = uint32_t(m ? m - 1 : 0); break; }  // one below
            default: n = uint32_t(0x7FFFFFFFu - uint32_t(j)); break;                // the top of the range
            }
            if (fast_div(n, magic, shift, pow2) != n / d) return 1000000 + it;
            int64_t big = int64_t(next() >> 12);                                    // < 2^52
            if (j % 8 == 3) big = (big / d) * int64_t(d);
            if (j % 8 == 5) big = (int64_t(1) << 52) - 1 - j;
            int64_t rem = -1;
            const int64_t q = div_by(big, int64_t(d), inv, &rem);
            if (q != big / int64_t(d) || rem != big % int64_t(d)) return 2000000 + it;
            // div64: dividends up to 2^63 - 1, divisors up to 2^42 (a position row of 2^31 sequences x 2000 bytes)
            const uint64_t d64 = (j % 3 == 0) ? uint64_t(d) : ((j % 3 == 1) ? uint64_t(d) * 2000u : (next() >> 22) + 1);
            const Div64 dc = div64_constants(d64);
            uint64_t n64 = next() >> 1;
            if (j % 8 == 2) n64 = (n64 / d64) * d64;
            if (j % 8 == 6) n64 = (n64 / d64) * d64 + d64 - 1;
            if (j % 8 == 7) n64 = (uint64_t(1) << 63) - 1 - uint64_t(j);
            if (div64(n64, dc) != n64 / d64) return 3000000 + it;
        }
    }
    return 0;
}

bsq_status bsq_xcd_of_blocks_device(int32_t *xcd_dev, int32_t nblocks, void *hip_stream) {
    if (!xcd_dev || nblocks <= 0) return bsq_internal::set_error(BSQ_ERR_INVALID_ARG, "null pointer or nblocks <= 0");
    hipLaunchKernelGGL(k_xcd_probe, dim3(unsigned(nblocks)), dim3(kThreads), 0, static_cast<hipStream_t>(hip_stream), xcd_dev);
    return check_launch("k_xcd_probe");
}

int32_t bsq_xcd_round_robin(void) {
    static int cached[16] = {};  // 0 unknown, 1 no, 2 yes
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) {
        (void)hipGetLastError();
        return -1;
    }
    if (cached[dev]) return cached[dev] - 1;
    constexpr int n = 256;
    int32_t *d = nullptr;
    int32_t h[n];
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&d), n * sizeof(int32_t));
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_xcd_probe, dim3(n), dim3(kThreads), 0, nullptr, d);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        (void)hipFree(d);
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return -1;
    }
    bool ok = true;
    for (int b = 0; b + 8 < n; ++b) ok = ok && h[b] == h[b + 8];
    uint32_t seen = 0;
    for (int b = 0; b < 8; ++b) seen |= 1u << (h[b] & 15);
    ok = ok && __builtin_popcount(seen) == 8;
    cached[dev] = ok ? 2 : 1;
    return ok ? 1 : 0;
}

}  // extern "C"
