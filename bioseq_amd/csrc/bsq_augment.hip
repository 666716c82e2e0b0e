// BLOSUM62 point-mutation augmentation on the device.
//
// Replaces the per-sequence pure-Python chain of /root/reference/bioseq/blosum.py:63-87
// (`augment_seq`) and its table construction (:36-48 `normrows`): callers apply it to the byte strings
// right before batch_tokenize / batch_onehot_encode (bioseq/loaders.py:83,102;
// training/cnnpretrain.py:115-117).  Here it runs in place on the packed batch already in HBM.
//
// Distribution per sequence (the reference's): with probability `frac` mutate the sequence;
// repeat chain_len times { repeat { idx = uniform position; new = draw from normrows[row(seq[idx])] }
// until new != seq[idx]; seq[idx] = new }.  Unknown residues (anything outside the 20 letters, incl.
// lower case) use the 'X' row, as probdict.get(c, default_transitions) does.  The kernel samples the same
// joint distribution in two steps (accept the position with probability 1 - row[old], then draw the new
// residue from the row without its own entry): see k_augment.
// The random STREAM is our own (counter-based splitmix64 keyed by seed / sequence / mutation /
// attempt; the reference uses a module-global numpy PCG64 that also depends on import order), so
// parity is defined on the table (bit-exact), the invariants and the substitution statistics.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <type_traits>

#include "bsq.h"
#include "bsq_augment_dev.h"
#include "bsq_internal.h"

namespace {

using namespace bsq_aug;  // kRows, kCols, AugTable, mix64, unit, mulhi_64x32, kMaxAttempts, kSeqPerWave, augment_groups_body

constexpr char kLetters[] = "ARNDCQEGHILKMFPSTWYV";  // column order; row order is the same + 'X'

// BLOSUM62 scores, rows ARNDCQEGHILKMFPSTWYV + X, columns ARNDCQEGHILKMFPSTWYV (public NCBI matrix).
constexpr int8_t kBlosum62[kRows][kCols] = {
    /*A*/ {4, -1, -2, -2, 0, -1, -1, 0, -2, -1, -1, -1, -1, -2, -1, 1, 0, -3, -2, 0},
    /*R*/ {-1, 5, 0, -2, -3, 1, 0, -2, 0, -3, -2, 2, -1, -3, -2, -1, -1, -3, -2, -3},
    /*N*/ {-2, 0, 6, 1, -3, 0, 0, 0, 1, -3, -3, 0, -2, -3, -2, 1, 0, -4, -2, -3},
    /*D*/ {-2, -2, 1, 6, -3, 0, 2, -1, -1, -3, -4, -1, -3, -3, -1, 0, -1, -4, -3, -3},
    /*C*/ {0, -3, -3, -3, 9, -3, -4, -3, -3, -1, -1, -3, -1, -2, -3, -1, -1, -2, -2, -1},
    /*Q*/ {-1, 1, 0, 0, -3, 5, 2, -2, 0, -3, -2, 1, 0, -3, -1, 0, -1, -2, -1, -2},
    /*E*/ {-1, 0, 0, 2, -4, 2, 5, -2, 0, -3, -3, 1, -2, -3, -1, 0, -1, -3, -2, -2},
    /*G*/ {0, -2, 0, -1, -3, -2, -2, 6, -2, -4, -4, -2, -3, -3, -2, 0, -2, -2, -3, -3},
    /*H*/ {-2, 0, 1, -1, -3, 0, 0, -2, 8, -3, -3, -1, -2, -1, -2, -1, -2, -2, 2, -3},
    /*I*/ {-1, -3, -3, -3, -1, -3, -3, -4, -3, 4, 2, -3, 1, 0, -3, -2, -1, -3, -1, 3},
    /*L*/ {-1, -2, -3, -4, -1, -2, -3, -4, -3, 2, 4, -2, 2, 0, -3, -2, -1, -2, -1, 1},
    /*K*/ {-1, 2, 0, -1, -3, 1, 1, -2, -1, -3, -2, 5, -1, -3, -1, 0, -1, -3, -2, -2},
    /*M*/ {-1, -1, -2, -3, -1, 0, -2, -3, -2, 1, 2, -1, 5, 0, -2, -1, -1, -1, -1, 1},
    /*F*/ {-2, -3, -3, -3, -2, -3, -3, -3, -1, 0, 0, -3, 0, 6, -4, -2, -2, 1, 3, -1},
    /*P*/ {-1, -2, -2, -1, -3, -1, -1, -2, -2, -3, -3, -1, -2, -4, 7, -1, -1, -4, -3, -2},
    /*S*/ {1, -1, 1, 0, -1, 0, 0, 0, -1, -2, -2, 0, -1, -2, -1, 4, 1, -3, -2, -2},
    /*T*/ {0, -1, 0, -1, -1, -1, -1, -2, -2, -1, -1, -1, -1, -2, -1, 1, 5, -2, -2, 0},
    /*W*/ {-3, -3, -4, -4, -2, -2, -3, -2, -2, -3, -2, -3, -1, 1, -4, -3, -2, 11, 2, -3},
    /*Y*/ {-2, -2, -2, -3, -2, -1, -2, -3, 2, -1, -1, -2, -1, 3, -3, -2, -2, 2, 7, -1},
    /*V*/ {0, -3, -3, -3, -1, -2, -2, -3, -3, 3, 1, -2, 1, -1, -2, -2, 0, -3, -1, 4},
    /*X*/ {0, -1, -1, -1, -2, -1, -1, -1, -1, -1, -1, -1, -1, -1, -2, 0, 0, -2, -1, -1},
};

// normrows[a][:] = 2^score / sum(2^score)  (blosum.py:41-45).  Every 2^score is a power of two in
// [2^-4, 2^11], so the row sum is exact in double whatever the summation order, and the quotient is one
// correctly rounded IEEE division: bit-identical to numpy's result.
void make_normrows(double out[kRows * kCols]) {
    for (int r = 0; r < kRows; ++r) {
        double sum = 0.0;
        for (int c = 0; c < kCols; ++c) sum += std::ldexp(1.0, kBlosum62[r][c]);
        for (int c = 0; c < kCols; ++c) out[r * kCols + c] = std::ldexp(1.0, kBlosum62[r][c]) / sum;
    }
}

// The position of a residue with self-probability p_self is accepted iff double(lo32) * 2^-32 < 1.0 - p_self (the twin's test).
// Both sides are exact doubles (lo32 < 2^32, a power-of-two scaling), so with t = 1.0 - p_self:
//   lo * 2^-32 < t  <=>  lo < t * 2^32  <=>  lo <= ceil(t * 2^32) - 1      (t > 0 always)
// -- one integer compare against a per-row constant instead of a convert, a multiply, a subtract and a compare in FP64.
uint32_t accept_threshold(double pself) {
    const double c = std::ceil(std::ldexp(1.0 - pself, 32));
    return c >= 4294967296.0 ? 0xFFFFFFFFu : static_cast<uint32_t>(static_cast<uint64_t>(c) - 1);
}



template <int K>
__global__ __launch_bounds__(256) void k_augment_groups(uint8_t *chars, const int64_t *offsets, int64_t B, int32_t chain_len,
                                                        double frac, uint64_t seed, const AugTable *tab) {
    augment_groups_body<K>(blockIdx.x, chars, offsets, B, chain_len, frac, seed, tab);
}

// Several independent batches in ONE launch (round 6, bsq_augment_device_multi): the augmentation of a batch is one short-lived, latency-
// bound generation of waves (~7-16 us as its own launch whatever the batch: launch latency and completion are exposed); the grids of up to
// eight batches concatenated cost about as much as one.  The per-batch pointers / counts / seeds are a table in the kernel arguments.
constexpr int kAugMultiMax = 8;
struct AugMulti {
    uint8_t *chars[kAugMultiMax];
    const int64_t *offsets[kAugMultiMax];
    int64_t B[kAugMultiMax];
    uint64_t seed[kAugMultiMax];
    uint32_t first_block[kAugMultiMax];  // ascending; entries behind the last batch: 0xFFFFFFFF
};
template <int K>
__global__ __launch_bounds__(256) void k_augment_groups_multi(AugMulti m, int32_t chain_len, double frac, const AugTable *tab) {
    const uint32_t blk = blockIdx.x;
    uint32_t i = 0;
#pragma unroll
    for (int k = 1; k < kAugMultiMax; ++k) i += blk >= m.first_block[k] ? 1u : 0u;
    augment_groups_body<K>(blk - m.first_block[i], m.chars[i], m.offsets[i], m.B[i], chain_len, frac, m.seed[i], tab);
}

AugTable *g_dev_table[16] = {};
std::mutex g_table_mu;

bsq_status device_table(AugTable **out) {
    std::lock_guard<std::mutex> lock(g_table_mu);
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return bsq_internal::set_hip_error("hipGetDevice", e);
    if (dev < 0 || dev >= 16) return bsq_internal::set_error(BSQ_ERR_INVALID_ARG, "device ordinal out of range");
    if (!g_dev_table[dev]) {
        AugTable h;
        double nr[kRows * kCols];
        make_normrows(nr);
        for (int r = 0; r < kRows; ++r) {
            double acc = 0.0;
            for (int c = 0; c < kCols; ++c) {
                acc += nr[r * kCols + c];
                h.cdf[r][c] = acc;
            }
            h.self[r] = r < kCols ? nr[r * kCols + r] : 0.0;
            h.accept_le[r] = accept_threshold(h.self[r]);
        }
        h.pad_[0] = h.pad_[1] = h.pad_[2] = h.pad_[3] = 0;
        std::memset(h.row_of, kRows - 1, sizeof(h.row_of));
        for (int c = 0; c < kCols; ++c) {
            h.row_of[static_cast<unsigned char>(kLetters[c])] = static_cast<uint8_t>(c);
            h.letter[c] = static_cast<uint8_t>(kLetters[c]);
        }
        AugTable *d = nullptr;
        e = hipMalloc(reinterpret_cast<void **>(&d), sizeof(AugTable));
        if (e != hipSuccess) return bsq_internal::set_hip_error("hipMalloc(augment table)", e);
        e = hipMemcpy(d, &h, sizeof(AugTable), hipMemcpyHostToDevice);
        if (e != hipSuccess) return bsq_internal::set_hip_error("hipMemcpy(augment table)", e);
        g_dev_table[dev] = d;
    }
    *out = g_dev_table[dev];
    return BSQ_OK;
}

}  // namespace

namespace bsq_internal {
bsq_status augment_device_table(const void **table) {
    AugTable *t = nullptr;
    const bsq_status st = device_table(&t);
    *table = t;
    return st;
}
}  // namespace bsq_internal

extern "C" {

bsq_status bsq_blosum62_accept_thresholds(uint32_t *out21) {
    if (!out21) return BSQ_ERR_INVALID_ARG;
    double nr[kRows * kCols];
    make_normrows(nr);
    for (int r = 0; r < kRows; ++r) out21[r] = accept_threshold(r < kCols ? nr[r * kCols + r] : 0.0);
    return BSQ_OK;
}

bsq_status bsq_blosum62_normrows(double *out21x20) {
    if (!out21x20) return BSQ_ERR_INVALID_ARG;
    make_normrows(out21x20);
    return BSQ_OK;
}

bsq_status bsq_augment_device(uint8_t *chars, const int64_t *offsets, int64_t B, int32_t chain_len, double frac,
                              uint64_t seed, void *hip_stream) {
    if (!offsets || B < 0 || chain_len < 0) return bsq_internal::set_error(BSQ_ERR_INVALID_ARG, "bad augment arguments");
    if (B == 0 || chain_len == 0 || !(frac > 0.0)) return BSQ_OK;
    if (!chars) return bsq_internal::set_error(BSQ_ERR_INVALID_ARG, "chars is null");
    AugTable *tab = nullptr;
    const bsq_status st = device_table(&tab);
    if (st != BSQ_OK) return st;
    const int64_t blocks = (B + 255) / 256;
    if (blocks >= (int64_t(1) << 31)) return bsq_internal::set_error(BSQ_ERR_INVALID_ARG, "batch too large");
    // knob "augment_k": attempts per lane and round (0 automatic = 4; 1 = the round-2 form; 2) -- speed only
    const int ak = bsq_internal::tuning().augment_k;
    if (ak == 1)
        hipLaunchKernelGGL(k_augment_groups<1>, dim3(unsigned(blocks)), dim3(256), 0, static_cast<hipStream_t>(hip_stream), chars,
                           offsets, B, chain_len, frac, seed, tab);
    else if (ak == 2)
        hipLaunchKernelGGL(k_augment_groups<2>, dim3(unsigned(blocks)), dim3(256), 0, static_cast<hipStream_t>(hip_stream), chars,
                           offsets, B, chain_len, frac, seed, tab);
    else
        hipLaunchKernelGGL(k_augment_groups<4>, dim3(unsigned(blocks)), dim3(256), 0, static_cast<hipStream_t>(hip_stream), chars,
                           offsets, B, chain_len, frac, seed, tab);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return bsq_internal::set_hip_error("k_augment", e);
    return BSQ_OK;
}

bsq_status bsq_augment_device_multi(int32_t n, const bsq_batch *batches, int32_t chain_len, double frac, const uint64_t *seeds, void *hip_stream) {
    if (n < 0 || (n > 0 && (!batches || !seeds)) || chain_len < 0) return bsq_internal::set_error(BSQ_ERR_INVALID_ARG, "bad augment arguments");
    for (int32_t i = 0; i < n; ++i)
        if (batches[i].B < 0 || (batches[i].B > 0 && (!batches[i].offsets || !batches[i].chars)))
            return bsq_internal::set_error(BSQ_ERR_INVALID_ARG, "a batch with a null pointer or B < 0");
    if (n == 0 || chain_len == 0 || !(frac > 0.0)) return BSQ_OK;
    AugTable *tab = nullptr;
    const bsq_status st = device_table(&tab);
    if (st != BSQ_OK) return st;
    int32_t i = 0;
    while (i < n) {  // groups of up to eight non-empty batches
        AugMulti m;
        for (int k = 0; k < kAugMultiMax; ++k) {
            m.chars[k] = nullptr, m.offsets[k] = nullptr, m.B[k] = 0, m.seed[k] = 0;
            m.first_block[k] = 0xFFFFFFFFu;
        }
        int g = 0;
        int64_t blocks = 0;
        while (i < n && g < kAugMultiMax) {
            if (batches[i].B > 0) {
                m.chars[g] = const_cast<uint8_t *>(batches[i].chars);  // (mutated in place: the caller passes writable memory)
                m.offsets[g] = batches[i].offsets;
                m.B[g] = batches[i].B;
                m.seed[g] = seeds[i];
                m.first_block[g] = uint32_t(blocks);
                blocks += (batches[i].B + 255) / 256;
                ++g;
            }
            ++i;
        }
        if (g == 0) break;
        if (blocks >= (int64_t(1) << 31)) return bsq_internal::set_error(BSQ_ERR_INVALID_ARG, "batches too large");
        const int ak = bsq_internal::tuning().augment_k;  // attempts per lane and round (0 automatic = 4; results never depend on it)
        if (ak == 1) hipLaunchKernelGGL(k_augment_groups_multi<1>, dim3(unsigned(blocks)), dim3(256), 0, static_cast<hipStream_t>(hip_stream), m, chain_len, frac, tab);
        else if (ak == 2) hipLaunchKernelGGL(k_augment_groups_multi<2>, dim3(unsigned(blocks)), dim3(256), 0, static_cast<hipStream_t>(hip_stream), m, chain_len, frac, tab);
        else hipLaunchKernelGGL(k_augment_groups_multi<4>, dim3(unsigned(blocks)), dim3(256), 0, static_cast<hipStream_t>(hip_stream), m, chain_len, frac, tab);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return bsq_internal::set_hip_error("k_augment_groups_multi", e);
    }
    return BSQ_OK;
}

}  // extern "C"
