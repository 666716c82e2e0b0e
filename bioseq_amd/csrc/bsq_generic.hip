// The element kernels of libbsq_hip.so (gfx950 only): one thread per OUTPUT element, any shape, alignment and alphabet (BYTES has ids > 255,
// which the LDS kernels' 8-bit ids cannot hold) -- the fallback of every fast path and the in-library cross-check of the fast kernels
// (bsq_*_device_generic) -- and the device-side length / offsets validation.  Split out of bsq_kernels.hip in round 5.
// Semantics: /root/reference/src/tokenize.h:342-369 (one-hot), :454-479 (tokens).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <mutex>

#include "bsq.h"
#include "bsq_device.h"
#include "bsq_internal.h"

namespace {

using namespace bsq_dev;  // kThreads

struct GParams {
    int8_t lut[256];
    const uint8_t *chars;
    const int64_t *offsets;
    const uint8_t *mask;
    void *out;
    int64_t B, P;
    int32_t C, bos, eos, bos_id, eos_id, pad_id, padchar, batch_first;
    int64_t row_seqs;  // (P, B, C) one-hot: sequences per position row of the destination (= B unless the batch is a column block)
};

__device__ __forceinline__ int32_t token_at(const GParams &p, int64_t b, int64_t t) {
    const int64_t start = p.offsets[b];
    int64_t L = p.offsets[b + 1] - start;
    const int64_t room = p.P - p.bos - p.eos;
    if (L > room) L = room;
    if (L < 0) L = 0;
    if (p.bos && t == 0) return p.bos_id;
    const int64_t j = t - p.bos;
    if (j < L) {
        if (p.mask && p.mask[start + j] == 0) return -1;
        const uint8_t c = p.chars[start + j];
        return c < 128 ? static_cast<int32_t>(p.lut[c]) : -1;  // negative == unmapped
    }
    if (p.eos && j == L) return p.eos_id;
    return p.padchar ? p.pad_id : -1;
}

template <typename T>
__global__ __launch_bounds__(kThreads) void k_onehot_generic(const GParams p) {
    const int64_t n = p.P * p.B * p.C;
    const int64_t stride = static_cast<int64_t>(gridDim.x) * kThreads;
    T *out = static_cast<T *>(p.out);
    for (int64_t e = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; e < n; e += stride) {
        int64_t t, b;
        int32_t c;
        if (p.batch_first == 2) {  // (B, C, P)
            const int64_t r = e / p.P;
            t = e - r * p.P;
            b = r / p.C;
            c = static_cast<int32_t>(r - b * p.C);
        } else {  // (P, B, C)
            const int64_t r = e / p.C;
            c = static_cast<int32_t>(e - r * p.C);
            t = r / p.B;
            b = r - t * p.B;
        }
        const int32_t tk = token_at(p, b, t);
        out[p.batch_first == 2 ? e : (t * p.row_seqs + b) * p.C + c] = (tk == c) ? T(1) : T(0);
    }
}

template <typename T>
__global__ __launch_bounds__(kThreads) void k_tokenize_generic(const GParams p) {
    const int64_t n = p.P * p.B;
    const int64_t stride = static_cast<int64_t>(gridDim.x) * kThreads;
    T *out = static_cast<T *>(p.out);
    for (int64_t e = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; e < n; e += stride) {
        int64_t b, t;
        if (p.batch_first) {
            b = e / p.P;
            t = e - b * p.P;
        } else {
            t = e / p.B;
            b = e - t * p.B;
        }
        const int32_t tk = token_at(p, b, t);
        out[p.batch_first ? e : t * p.row_seqs + b] = tk >= 0 ? static_cast<T>(tk) : T(0);
    }
}

// first_bad[0]: first sequence longer than `room`; first_bad[1]: first entry i with offsets[i] > offsets[i + 1],
// offsets[0] < 0 (reported as 0) or offsets[B] > nchars (reported as B) -- only checked when nchars >= 0.
// first_bad[2] counts the workgroups that are done: the LAST one copies the two minima to `report` -- host-mapped memory the caller
// reads after synchronising the stream -- and puts the device words back to "none" for the next call: one launch per validation,
// no memset in front of it and no device -> host copy behind it (round 3: 43 us per validated call, 31 of them these three stream
// operations; a loader epoch paid them per batch).
__global__ __launch_bounds__(kThreads) void k_first_too_long(const int64_t *offsets, int64_t B, int64_t room, int64_t nchars,
                                                             unsigned long long *first_bad, unsigned long long *report) {
    const int64_t stride = static_cast<int64_t>(gridDim.x) * kThreads;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; i < B; i += stride) {
        const int64_t lo = offsets[i], hi = offsets[i + 1];
        if (hi - lo > room) atomicMin(first_bad, static_cast<unsigned long long>(i));
        if (nchars >= 0) {
            if (hi < lo || (i == 0 && lo < 0)) atomicMin(first_bad + 1, static_cast<unsigned long long>(i));
            if (i == B - 1 && hi > nchars) atomicMin(first_bad + 1, static_cast<unsigned long long>(B));
        }
    }
    __syncthreads();  // (every atomicMin of this workgroup has been issued; device-scope atomics are ordered at the L2)
    if (threadIdx.x == 0) {
        __threadfence();
        if (atomicAdd(first_bad + 2, 1ull) + 1 == gridDim.x) {
            __threadfence();
            const unsigned long long a = atomicExch(first_bad, ~0ull), b = atomicExch(first_bad + 1, ~0ull);
            atomicExch(first_bad + 2, 0ull);
            __hip_atomic_store(report, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(report + 1, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

bsq_status check_launch(const char *what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return bsq_internal::set_hip_error(what, e);
    return BSQ_OK;
}

void fill_generic(GParams &g, const bsq_desc *d, const uint8_t *chars, const int64_t *offsets, const uint8_t *mask,
                  int64_t B, int64_t P, int batch_first, void *out) {
    for (int i = 0; i < 256; ++i) g.lut[i] = d->lut[i];
    g.chars = chars;
    g.offsets = offsets;
    g.mask = mask;
    g.out = out;
    g.B = B;
    g.P = P;
    g.C = bsq_alphabet_size(d);
    g.bos = d->bos;
    g.eos = d->eos;
    g.bos_id = bsq_bos_id(d);
    g.eos_id = bsq_eos_id(d);
    g.pad_id = bsq_pad_id(d);
    g.padchar = d->padchar;
    g.batch_first = batch_first;
    g.row_seqs = B;
}

unsigned generic_grid(int64_t n) {
    const int64_t blocks = (n + kThreads - 1) / kThreads;
    const int64_t cap = 256 * 32;
    return static_cast<unsigned>(blocks < 1 ? 1 : (blocks > cap ? cap : blocks));
}

}  // namespace

namespace bsq_internal {

bsq_status onehot_generic_block(const bsq_desc *d, const uint8_t *chars, const int64_t *offsets, const uint8_t *mask_or_null, int64_t B,
                                       int64_t P, bsq_dtype t, void *out, int64_t row_seqs, void *hip_stream) {
    if (!d || B < 0 || P <= 0 || (B > 0 && (!offsets || !out)))  // (an EMPTY batch -- a rank without sequences -- has nothing to point at)
        return bsq_internal::set_error(BSQ_ERR_INVALID_ARG, "null pointer, B < 0 or padlen <= 0");
    if (B == 0) return BSQ_OK;
    GParams g;
    fill_generic(g, d, chars, offsets, mask_or_null, B, P, 0, out);
    g.row_seqs = row_seqs;
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    const unsigned grid = generic_grid(P * B * g.C);
#define BSQ_GEN(T) hipLaunchKernelGGL((k_onehot_generic<T>), dim3(grid), dim3(kThreads), 0, s, g)
    switch (t) {
    case BSQ_I8: BSQ_GEN(int8_t); break;
    case BSQ_I16: BSQ_GEN(int16_t); break;
    case BSQ_I32: BSQ_GEN(int32_t); break;
    case BSQ_U64: BSQ_GEN(uint64_t); break;
    case BSQ_F32: BSQ_GEN(float); break;
    case BSQ_F64: BSQ_GEN(double); break;
    default: return bsq_internal::set_error(BSQ_ERR_DTYPE, "bad bsq_dtype");
    }
#undef BSQ_GEN
    return check_launch("k_onehot_generic");
}

// the channels-first (B, C, P) one-hot, element by element: bsq_onehot_bcl_device's fallback (ids > 250, misaligned results under knob bcl_path 3)
bsq_status onehot_generic_bcl(const bsq_desc *d, const uint8_t *chars, const int64_t *offsets, const uint8_t *mask_or_null, int64_t B, int64_t P,
                              bsq_dtype t, void *out, void *hip_stream) {
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    GParams g;
    fill_generic(g, d, chars, offsets, mask_or_null, B, P, 2, out);
    const unsigned grid = generic_grid(P * B * g.C);
#define BSQ_GEN(T) hipLaunchKernelGGL((k_onehot_generic<T>), dim3(grid), dim3(kThreads), 0, s, g)
    switch (t) {
    case BSQ_I8: BSQ_GEN(int8_t); break;
    case BSQ_I16: BSQ_GEN(int16_t); break;
    case BSQ_I32: BSQ_GEN(int32_t); break;
    case BSQ_U64: BSQ_GEN(uint64_t); break;
    case BSQ_F32: BSQ_GEN(float); break;
    case BSQ_F64: BSQ_GEN(double); break;
    }
#undef BSQ_GEN
    return check_launch("k_onehot_generic<bcl>");
}

bsq_status tokenize_generic_block(const bsq_desc *d, const uint8_t *chars, const int64_t *offsets, int64_t B, int64_t P,
                                         int32_t batch_first, bsq_dtype t, void *out, int64_t row_seqs, void *hip_stream) {
    if (!d || B < 0 || P <= 0 || (B > 0 && (!offsets || !out)))  // (an EMPTY batch -- a rank without sequences -- has nothing to point at)
        return bsq_internal::set_error(BSQ_ERR_INVALID_ARG, "null pointer, B < 0 or padlen <= 0");
    if (B == 0) return BSQ_OK;
    GParams g;
    fill_generic(g, d, chars, offsets, nullptr, B, P, batch_first != 0, out);
    g.row_seqs = row_seqs;
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    const unsigned grid = generic_grid(P * B);
#define BSQ_GEN(T) hipLaunchKernelGGL((k_tokenize_generic<T>), dim3(grid), dim3(kThreads), 0, s, g)
    switch (t) {
    case BSQ_I8: BSQ_GEN(int8_t); break;
    case BSQ_I16: BSQ_GEN(int16_t); break;
    case BSQ_I32: BSQ_GEN(int32_t); break;
    case BSQ_U64: BSQ_GEN(uint64_t); break;
    case BSQ_F32: BSQ_GEN(float); break;
    case BSQ_F64: BSQ_GEN(double); break;
    default: return bsq_internal::set_error(BSQ_ERR_DTYPE, "bad bsq_dtype");
    }
#undef BSQ_GEN
    return check_launch("k_tokenize_generic");
}

}  // namespace bsq_internal

extern "C" {

bsq_status bsq_onehot_device_generic(const bsq_desc *d, const uint8_t *chars, const int64_t *offsets,
                                     const uint8_t *mask_or_null, int64_t B, int64_t P, bsq_dtype t, void *out,
                                     void *hip_stream) {
    return bsq_internal::onehot_generic_block(d, chars, offsets, mask_or_null, B, P, t, out, B, hip_stream);
}

bsq_status bsq_tokenize_device_generic(const bsq_desc *d, const uint8_t *chars, const int64_t *offsets,
                                       int64_t B, int64_t P, int32_t batch_first, bsq_dtype t, void *out,
                                       void *hip_stream) {
    return bsq_internal::tokenize_generic_block(d, chars, offsets, B, P, batch_first, t, out, B, hip_stream);
}

bsq_status bsq_validate_lengths_device(const int64_t *offsets_dev, int64_t B, int64_t P, int32_t bos,
                                       int32_t eos, int64_t *first_bad, void *hip_stream) {
    return bsq_validate_packed_device(offsets_dev, B, P, bos, eos, -1, first_bad, hip_stream);
}

bsq_status bsq_validate_packed_device(const int64_t *offsets_dev, int64_t B, int64_t P, int32_t bos, int32_t eos,
                                      int64_t nchars, int64_t *first_bad, void *hip_stream) {
    if (!offsets_dev || B < 0 || P <= 0 || !first_bad) return bsq_internal::set_error(BSQ_ERR_INVALID_ARG, "null pointer, B < 0 or padlen <= 0");
    *first_bad = -1;
    if (B == 0) return BSQ_OK;
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    // per device, allocated once: three device words (two minima + the count of finished workgroups; "none", "none", 0 between calls)
    // and two host-mapped words the kernel's last workgroup reports into
    static unsigned long long *flags[16] = {}, *reports[16] = {};
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);  // the words are shared: one validation at a time per process
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return bsq_internal::set_hip_error("hipGetDevice", e);
    if (dev < 0 || dev >= 16) return bsq_internal::set_error(BSQ_ERR_INVALID_ARG, "device ordinal out of range");
    if (!flags[dev]) {
        unsigned long long *f = nullptr, *r = nullptr;
        const unsigned long long init[3] = {~0ull, ~0ull, 0ull};
        e = hipMalloc(reinterpret_cast<void **>(&f), sizeof(init));
        if (e == hipSuccess) e = hipMemcpy(f, init, sizeof(init), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void **>(&r), 2 * sizeof(unsigned long long), hipHostMallocPortable | hipHostMallocMapped | hipHostMallocCoherent);
        if (e != hipSuccess) {
            if (f) (void)hipFree(f);
            return bsq_internal::set_hip_error("bsq_validate_packed_device: flags", e);
        }
        flags[dev] = f;
        reports[dev] = r;
    }
    volatile unsigned long long *host = reports[dev];
    host[0] = host[1] = ~0ull;
    hipLaunchKernelGGL(k_first_too_long, dim3(generic_grid(B)), dim3(kThreads), 0, s, offsets_dev, B, P - (bos != 0) - (eos != 0), nchars,
                       flags[dev], reports[dev]);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) return bsq_internal::set_hip_error("bsq_validate_packed_device", e);
    if (host[1] != ~0ull) {
        *first_bad = static_cast<int64_t>(host[1]);
        return bsq_internal::set_error(BSQ_ERR_INVALID_ARG, "offsets are negative, decreasing or run past the end of chars");
    }
    if (host[0] != ~0ull) {
        *first_bad = static_cast<int64_t>(host[0]);
        return BSQ_ERR_SEQ_TOO_LONG;
    }
    return BSQ_OK;
}

}  // extern "C"
