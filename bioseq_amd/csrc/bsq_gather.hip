// Index-list batches out of a packed store that lives in HBM (SURVEY.md section 8 row f-3 under a shuffling sampler).
//
// The reference's FlatFileDataset.__getitem__ (/root/reference/bioseq/loaders.py:76-104) fetches ONE sequence per call from
// the memory-mapped file, tokenises it on the host and lets the DataLoader stack the results; a shuffled epoch therefore
// touches the host for every sample.  Here the whole FlatFile is uploaded once (FlatFile.to_device) and a batch of
// arbitrary, repeated or empty sequence indices is rebuilt ON THE DEVICE as a packed batch (chars, offsets) that the
// encode kernels consume as they are: ONE launch for loader-sized batches (k_gather_small), two beyond 4096 indices (k_gather_lengths2 +
// k_gather_place, round 6), no host round trip, no H2D copy.  The three launches of rounds 2-5 stay behind knob "gather_small" = 1:
//   k_gather_lengths  out_offsets[i + 1] <- length of sequence index[i]   (bad indices: length 0, position recorded)
//   k_gather_scan     in-place inclusive prefix sum -> out_offsets[i + 1] = end of output sequence i  (one workgroup;
//                     a batch is 10^3 .. 10^6 sequences: 8 MB at most, microseconds)
//   k_gather_chars    16 lanes per output sequence copy its characters with unaligned 16-byte loads / stores
// gfx950 only.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <mutex>

#include "bsq.h"
#include "bsq_internal.h"

namespace {

constexpr int kThreads = 256;
typedef uint32_t g_u32x4u __attribute__((ext_vector_type(4), aligned(1)));

__global__ __launch_bounds__(kThreads) void k_gather_lengths(const int64_t *offsets, int64_t n_store, const int64_t *index,
                                                             int64_t n, int64_t *out_offsets, unsigned long long *first_bad) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
    if (i == 0) out_offsets[0] = 0;
    if (i >= n) return;
    const int64_t j = index[i];
    int64_t len = 0;
    if (j >= 0 && j < n_store) {
        len = offsets[j + 1] - offsets[j];
        if (len < 0) len = 0;
    } else if (first_bad) {
        atomicMin(first_bad, static_cast<unsigned long long>(i));
    }
    out_offsets[i + 1] = len;
}

// in-place inclusive prefix sum of v[0 .. n) by ONE workgroup of 1024 threads (pieces of 1024 with a running carry)
__global__ __launch_bounds__(1024) void k_gather_scan(int64_t *v, int64_t n) {
    __shared__ int64_t s_wave[16];
    __shared__ int64_t s_carry;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (int64_t base = 0; base < n; base += 1024) {
        const int64_t i = base + threadIdx.x;
        int64_t x = i < n ? v[i] : 0;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int64_t o = __shfl_up(x, d, 64);
            if (lane >= d) x += o;
        }
        if (lane == 63) s_wave[wave] = x;
        __syncthreads();
        int64_t before = s_carry;
        for (int w = 0; w < wave; ++w) before += s_wave[w];
        if (i < n) v[i] = x + before;
        __syncthreads();
        if (threadIdx.x == 1023) s_carry = x + before;
        __syncthreads();
    }
}

// 16 lanes per output sequence.  capacity: bytes of out_chars; a batch that does not fit is cut there (never a write
// past the buffer) and recorded in first_bad as position n + i.
__global__ __launch_bounds__(kThreads) void k_gather_chars(const uint8_t *chars, const int64_t *offsets, int64_t n_store,
                                                           const int64_t *index, int64_t n, const int64_t *out_offsets,
                                                           uint8_t *out_chars, int64_t capacity, unsigned long long *first_bad) {
    const int sub = threadIdx.x & 15;
    const int64_t i = static_cast<int64_t>(blockIdx.x) * (kThreads / 16) + (threadIdx.x >> 4);
    if (i >= n) return;
    const int64_t j = index[i];
    if (j < 0 || j >= n_store) return;
    const int64_t src0 = offsets[j], d0 = out_offsets[i];
    int64_t len = out_offsets[i + 1] - d0;
    if (d0 + len > capacity) {
        if (sub == 0 && first_bad) atomicMin(first_bad, static_cast<unsigned long long>(n + i));
        len = capacity > d0 ? capacity - d0 : 0;
    }
    const uint8_t *src = chars + src0;
    uint8_t *dst = out_chars + d0;
    const int64_t body = len & ~int64_t(15);
    for (int64_t p = sub * 16; p < body; p += 256)
        *reinterpret_cast<g_u32x4u *>(dst + p) = *reinterpret_cast<const g_u32x4u *>(src + p);
    if (sub < (len & 15)) dst[body + sub] = src[body + sub];
}

// Loader-sized batches (n <= kSmallN: what a training step asks for) in ONE launch instead of memset + three: a workgroup owns 16
// output sequences (16 lanes each, as k_gather_chars), sums the lengths of every sequence in front of its own -- all 256 threads,
// every load of the sum independent of the others -- and then does lengths, offsets and characters of its 16.  n^2 / 16 length
// look-ups in total (1 M at n = 4096, L2 hits) buy three launches' worth of latency per batch: the step of a shuffled epoch is
// host- and launch-bound (profiles/r04/loader_step_lab.txt).  first_bad may be null (a caller that vouches for its indices).
constexpr int64_t kSmallN = 4096;
__global__ __launch_bounds__(kThreads) void k_gather_small(const uint8_t *chars, const int64_t *offsets, int64_t n_store, const int64_t *index,
                                                           int64_t n, int64_t *out_offsets, uint8_t *out_chars, int64_t capacity,
                                                           unsigned long long *first_bad) {
    __shared__ int64_t s_part[kThreads / 64];
    __shared__ int64_t s_len[16];
    const int tid = threadIdx.x, sub = tid & 15, g = tid >> 4;
    const int64_t first = static_cast<int64_t>(blockIdx.x) * 16;
    int64_t acc = 0;
#pragma unroll
    for (int k = 0; k < int(kSmallN / kThreads); ++k) {
        const int64_t j = tid + kThreads * k;
        if (j < first) {
            const int64_t src = index[j];
            if (src >= 0 && src < n_store) {
                const int64_t l = offsets[src + 1] - offsets[src];
                acc += l > 0 ? l : 0;
            }
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d, 64);
    if ((tid & 63) == 0) s_part[tid >> 6] = acc;
    const int64_t i = first + g;
    const bool live = i < n;
    const int64_t src = live ? index[i] : -1;
    const bool ok = live && src >= 0 && src < n_store;
    int64_t src0 = 0, len = 0;
    if (ok) {
        src0 = offsets[src];
        len = offsets[src + 1] - src0;
        if (len < 0) len = 0;
    }
    if (sub == 0) {
        s_len[g] = len;
        if (live && !ok && first_bad) atomicMin(first_bad, static_cast<unsigned long long>(i));
    }
    __syncthreads();
    int64_t d0 = s_part[0] + s_part[1] + s_part[2] + s_part[3];
    for (int q = 0; q < g; ++q) d0 += s_len[q];
    if (tid == 0 && blockIdx.x == 0) out_offsets[0] = 0;
    if (!live) return;
    if (sub == 0) out_offsets[i + 1] = d0 + len;
    if (!ok || !chars || !out_chars) return;
    if (d0 + len > capacity) {
        if (sub == 0 && first_bad) atomicMin(first_bad, static_cast<unsigned long long>(n + i));
        len = capacity > d0 ? capacity - d0 : 0;
    }
    const uint8_t *sp = chars + src0;
    uint8_t *dst = out_chars + d0;
    const int64_t body = len & ~int64_t(15);
    for (int64_t p = sub * 16; p < body; p += 256)
        *reinterpret_cast<g_u32x4u *>(dst + p) = *reinterpret_cast<const g_u32x4u *>(sp + p);
    if (sub < (len & 15)) dst[body + sub] = sp[body + sub];
}

// Lists beyond kSmallN in TWO launches (round 6; rounds 2-5: three, the middle one a prefix sum by ONE workgroup walking the list in
// pieces of 1024 -- 16 384 indices of BASELINE config 5's store took 31 us, twice the encode that follows them):
//   k_gather_lengths2  out_offsets[i + 1] <- length of sequence index[i], and the SUM of every 64 of them (one wave) -> wave_sums[i / 64]
//   k_gather_place     a workgroup owns 64 output sequences: the sums of the groups in front of it (coalesced loads, all threads: n / 64
//                      of them at most -- lists of up to 2^20 indices) + a wave scan of its own 64 lengths give their offsets; 16 lanes
//                      per sequence copy the characters, four rounds.  A workgroup rewrites only its OWN entries of out_offsets.
constexpr int kPlaceS = 64;
__global__ __launch_bounds__(kThreads) void k_gather_lengths2(const int64_t *offsets, int64_t n_store, const int64_t *index, int64_t n,
                                                              int64_t *out_offsets, int64_t *wave_sums, unsigned long long *first_bad) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
    if (i == 0) out_offsets[0] = 0;
    int64_t len = 0;
    if (i < n) {
        const int64_t j = index[i];
        if (j >= 0 && j < n_store) {
            len = offsets[j + 1] - offsets[j];
            if (len < 0) len = 0;
        } else if (first_bad) {
            atomicMin(first_bad, static_cast<unsigned long long>(i));
        }
        out_offsets[i + 1] = len;
    }
    int64_t acc = len;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d, 64);
    if ((threadIdx.x & 63) == 0 && i < n) wave_sums[i >> 6] = acc;
}

__global__ __launch_bounds__(kThreads) void k_gather_place(const uint8_t *chars, const int64_t *offsets, int64_t n_store, const int64_t *index,
                                                           int64_t n, int64_t *out_offsets, const int64_t *wave_sums, uint8_t *out_chars,
                                                           int64_t capacity, unsigned long long *first_bad) {
    __shared__ int64_t s_part[kThreads / 64];
    __shared__ int64_t s_src0[kPlaceS], s_d0[kPlaceS], s_len[kPlaceS];
    const int tid = threadIdx.x;
    const int64_t first = static_cast<int64_t>(blockIdx.x) * kPlaceS;
    int64_t acc = 0;  // the sums of the 64-sequence groups in front of this one: coalesced, every thread
    for (int64_t k = tid; k < static_cast<int64_t>(blockIdx.x); k += kThreads) acc += wave_sums[k];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d, 64);
    if ((tid & 63) == 0) s_part[tid >> 6] = acc;
    int64_t len = 0, x = 0;
    if (tid < kPlaceS) {  // (one wave)
        const int64_t i = first + tid;
        if (i < n) len = out_offsets[i + 1];
        const int64_t src = i < n ? index[i] : -1;
        s_src0[tid] = (src >= 0 && src < n_store) ? offsets[src] : -1;  // -1: nothing to copy
        s_len[tid] = len;
        x = len;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int64_t o = __shfl_up(x, d, 64);
            if (tid >= d) x += o;
        }
    }
    __syncthreads();  // (every read of the lengths in out_offsets is done: this workgroup's own entries are overwritten below, nobody else's)
    if (tid < kPlaceS) {
        const int64_t end = s_part[0] + s_part[1] + s_part[2] + s_part[3] + x;
        s_d0[tid] = end - len;
        if (first + tid < n) out_offsets[first + tid + 1] = end;
    }
    __syncthreads();
    if (!chars || !out_chars) return;
    const int sub = tid & 15;
#pragma unroll
    for (int r = 0; r < kPlaceS / (kThreads / 16); ++r) {
        const int g = r * (kThreads / 16) + (tid >> 4);
        const int64_t i = first + g;
        const int64_t src0 = s_src0[g];
        if (i >= n || src0 < 0) continue;
        const int64_t d0 = s_d0[g];
        int64_t l = s_len[g];
        if (d0 + l > capacity) {
            if (sub == 0 && first_bad) atomicMin(first_bad, static_cast<unsigned long long>(n + i));
            l = capacity > d0 ? capacity - d0 : 0;
        }
        const uint8_t *sp = chars + src0;
        uint8_t *dst = out_chars + d0;
        const int64_t body = l & ~int64_t(15);
        for (int64_t p = sub * 16; p < body; p += 256)
            *reinterpret_cast<g_u32x4u *>(dst + p) = *reinterpret_cast<const g_u32x4u *>(sp + p);
        if (sub < (l & 15)) dst[body + sub] = sp[body + sub];
    }
}

}  // namespace

extern "C" {

bsq_status bsq_gather_packed_device(const uint8_t *chars, const int64_t *offsets, int64_t n_store, const int64_t *index,
                                    int64_t n, uint8_t *out_chars, int64_t out_capacity, int64_t *out_offsets,
                                    int64_t *status_dev, void *hip_stream) {
    if (!offsets || !out_offsets || n_store < 0 || n < 0 || out_capacity < 0 || (n > 0 && !index))
        return bsq_internal::set_error(BSQ_ERR_INVALID_ARG, "bsq_gather_packed_device: null pointer or negative size");
    if ((n + kThreads - 1) / kThreads >= (int64_t(1) << 31) || (n + 15) / 16 >= (int64_t(1) << 31) || (n + kPlaceS - 1) / kPlaceS >= (int64_t(1) << 31))
        return bsq_internal::set_error(BSQ_ERR_INVALID_ARG, "index list too long");
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    hipError_t e = hipSuccess;
    // status_dev == NULL: the caller vouches for its indices and its capacity (bad indices still read nothing, an overflow is still cut)
    if (status_dev) e = hipMemsetAsync(status_dev, 0xFF, sizeof(int64_t), s);  // -1 = every index valid, everything fitted
    if (e != hipSuccess) return bsq_internal::set_hip_error("hipMemsetAsync(gather status)", e);
    unsigned long long *bad = reinterpret_cast<unsigned long long *>(status_dev);
    if (n == 0) {
        e = hipMemsetAsync(out_offsets, 0, sizeof(int64_t), s);
        if (e != hipSuccess) return bsq_internal::set_hip_error("hipMemsetAsync(gather offsets)", e);
        return BSQ_OK;
    }
    if (n <= kSmallN && bsq_internal::tuning().gather_small != 1) {
        hipLaunchKernelGGL(k_gather_small, dim3(unsigned((n + 15) / 16)), dim3(kThreads), 0, s, chars, offsets, n_store, index, n, out_offsets,
                           out_chars, out_chars ? out_capacity : 0, bad);
        e = hipGetLastError();
        if (e != hipSuccess) return bsq_internal::set_hip_error("k_gather_small", e);
        return BSQ_OK;
    }
    if (bsq_internal::tuning().gather_small != 1 && n <= (int64_t(1) << 20)) {  // two launches; the sums of 64 lengths travel through the stream's scratch
        const int64_t nblk = (n + kThreads - 1) / kThreads;
        std::lock_guard<std::mutex> scratch_turn(bsq_internal::workspace_mutex());
        void *ws = nullptr;
        const bsq_status st = bsq_internal::workspace_acquire(size_t((n + 63) / 64) * sizeof(int64_t), s, &ws);
        if (st != BSQ_OK) return st;
        hipLaunchKernelGGL(k_gather_lengths2, dim3(unsigned(nblk)), dim3(kThreads), 0, s, offsets, n_store, index, n, out_offsets,
                           static_cast<int64_t *>(ws), bad);
        hipLaunchKernelGGL(k_gather_place, dim3(unsigned((n + kPlaceS - 1) / kPlaceS)), dim3(kThreads), 0, s, chars, offsets, n_store, index, n,
                           out_offsets, static_cast<const int64_t *>(ws), out_capacity > 0 ? out_chars : nullptr, out_capacity, bad);
        e = hipGetLastError();
        bsq_internal::workspace_release(ws, s);
        if (e != hipSuccess) return bsq_internal::set_hip_error("k_gather_lengths2 / k_gather_place", e);
        return BSQ_OK;
    }
    hipLaunchKernelGGL(k_gather_lengths, dim3(unsigned((n + kThreads - 1) / kThreads)), dim3(kThreads), 0, s, offsets, n_store,
                       index, n, out_offsets, bad);
    hipLaunchKernelGGL(k_gather_scan, dim3(1), dim3(1024), 0, s, out_offsets + 1, n);
    if (chars && out_chars && out_capacity > 0)
        hipLaunchKernelGGL(k_gather_chars, dim3(unsigned((n + 15) / 16)), dim3(kThreads), 0, s, chars, offsets, n_store, index, n,
                           out_offsets, out_chars, out_capacity, bad);
    e = hipGetLastError();
    if (e != hipSuccess) return bsq_internal::set_hip_error("k_gather_*", e);
    return BSQ_OK;
}

}  // extern "C"
