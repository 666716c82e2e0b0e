// decode_tokens on the device (/root/reference/src/tokenize.h:131-183) and the
// argmax that turns logits into tokens in front of it (README.md:48 "if you have logits, use an argmax to convert
// to tokens for decoding") -- gfx950 only.  The token matrix never travels to the host: only the decoded text does.
//
// A token decodes to ONE byte (the first byte of its alphabet group) or to a five-byte piece (<BOS>, <EOS>, <PAD>),
// so the text of a row is a variable-width expansion of its tokens.  Two passes over the tokens:
//   k_decode_sizes   one wave per row: piece widths summed with a wave reduction -> row length; the first token
//                    that is not in the tokenizer's table is recorded (the reference throws on it);
//   k_scan_rows      exclusive prefix sum of the row lengths (one workgroup, running carry) -> row offsets;
//   k_decode_write   one wave per row, 64 tokens per step: an in-wave exclusive prefix sum of the widths (DPP-free
//                    shuffle scan) gives every lane the byte offset of its piece; pieces are written byte-wise.
// k_argmax_tokens: one lane per (row, position), channels contiguous; first maximum wins (torch.argmax's rule).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>
#include <mutex>

#include "bsq.h"
#include "bsq_device.h"
#include "bsq_internal.h"

namespace {

using namespace bsq_dev;

// Decode table: entry for token value v (as int32, -128 <= v < 512) at index v + 128:
//   0xFFFF invalid, 0x0100 | byte  a single byte,  0x0200 | k  the k-th special piece (0 <BOS>, 1 <EOS>, 2 <PAD>).
constexpr int kTabLo = -128, kTabN = 640;
struct DParams {
    uint16_t tab[kTabN];
    const uint8_t *tokens;
    int64_t nrows, ncols, row_stride, col_stride;  // strides in bytes
    int32_t itemsize;
    int64_t *row_len;            // [nrows] (sizes pass) -- becomes offsets after the scan
    const int64_t *row_off;      // [nrows + 1] (write pass)
    uint8_t *out;
    unsigned long long *first_bad;  // flat index row * ncols + col of the first invalid token (atomicMin)
};

__device__ __forceinline__ uint32_t entry_of(const DParams &p, const uint16_t *s_tab, int64_t row, int64_t col) {
    const uint8_t *a = p.tokens + row * p.row_stride + col * p.col_stride;
    int32_t v;
    switch (p.itemsize) {  // tokenize.h:107-124 load_value, then the implicit uint32 -> int32 of lookup.find
    case 1: v = *a; break;
    case 2: v = *reinterpret_cast<const uint16_t *>(a); break;
    case 4: v = static_cast<int32_t>(*reinterpret_cast<const uint32_t *>(a)); break;
    default: v = static_cast<int32_t>(static_cast<uint32_t>(*reinterpret_cast<const uint64_t *>(a))); break;
    }
    const int32_t idx = v - kTabLo;
    return (idx >= 0 && idx < kTabN) ? s_tab[idx] : 0xFFFFu;
}

__device__ __forceinline__ int32_t width_of(uint32_t e) { return e == 0xFFFFu ? 0 : ((e & 0x0200u) ? 5 : 1); }

__device__ __forceinline__ void stage_tab(const DParams &p, uint16_t *s_tab) {
    for (int i = threadIdx.x; i < kTabN; i += kThreads) s_tab[i] = p.tab[i];
    __syncthreads();
}

__global__ __launch_bounds__(kThreads) void k_decode_sizes(const DParams p) {
    __shared__ uint16_t s_tab[kTabN];
    stage_tab(p, s_tab);
    const int lane = threadIdx.x & 63;
    const int64_t row = static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6);
    if (row >= p.nrows) return;
    int64_t sum = 0;
    for (int64_t c = lane; c < p.ncols; c += 64) {
        const uint32_t e = entry_of(p, s_tab, row, c);
        if (e == 0xFFFFu) atomicMin(p.first_bad, static_cast<unsigned long long>(row * p.ncols + c));
        sum += width_of(e);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) sum += __shfl_xor(sum, d, 64);
    if (lane == 0) p.row_len[row] = sum;
}

// In place: len[0..n) -> exclusive offsets, len[n] = total.  One workgroup; rows are few compared with tokens.
__global__ __launch_bounds__(1024) void k_scan_rows(int64_t *len, int64_t n) {
    __shared__ int64_t s_wave[16];
    __shared__ int64_t s_carry;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (int64_t base = 0; base < n; base += 1024) {
        const int64_t i = base + tid;
        const int64_t v = i < n ? len[i] : 0;
        int64_t x = v;  // inclusive scan inside the wave
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int64_t y = __shfl_up(x, d, 64);
            if (lane >= d) x += y;
        }
        if (lane == 63) s_wave[wave] = x;
        __syncthreads();
        int64_t before = s_carry;
        for (int w = 0; w < wave; ++w) before += s_wave[w];
        if (i < n) len[i] = before + x - v;
        __syncthreads();
        if (tid == 1023) s_carry = before + x;
        __syncthreads();
    }
    if (tid == 0) len[n] = s_carry;
}

__global__ __launch_bounds__(kThreads) void k_decode_write(const DParams p) {
    __shared__ uint16_t s_tab[kTabN];
    stage_tab(p, s_tab);
    const int lane = threadIdx.x & 63;
    const int64_t row = static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6);
    if (row >= p.nrows) return;
    uint8_t *dst = p.out + p.row_off[row];
    int64_t done = 0;  // bytes of this row written by earlier steps (wave-uniform)
    for (int64_t c0 = 0; c0 < p.ncols; c0 += 64) {
        const int64_t c = c0 + lane;
        const uint32_t e = c < p.ncols ? entry_of(p, s_tab, row, c) : 0xFFFFu;
        const int32_t w = width_of(e);
        int32_t x = w;  // inclusive prefix sum of the widths over the wave
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int32_t y = __shfl_up(x, d, 64);
            if (lane >= d) x += y;
        }
        uint8_t *o = dst + done + (x - w);
        if (w == 1) {
            o[0] = static_cast<uint8_t>(e & 0xFFu);
        } else if (w == 5) {  // "<BOS>", "<EOS>", "<PAD>"
            const uint32_t k = e & 3u;
            const uint32_t mid = k == 0 ? ('B' | 'O' << 8 | 'S' << 16) : (k == 1 ? ('E' | 'O' << 8 | 'S' << 16) : ('P' | 'A' << 8 | 'D' << 16));
            o[0] = '<';
            o[1] = static_cast<uint8_t>(mid);
            o[2] = static_cast<uint8_t>(mid >> 8);
            o[3] = static_cast<uint8_t>(mid >> 16);
            o[4] = '>';
        }
        done += __shfl(x, 63, 64);
    }
}

// tokens[r] = argmax_c logits[r * row_stride + c] (first maximum), r < n.  OT: uint8 (C <= 256) or int32.
template <typename LT, typename OT>
__global__ __launch_bounds__(kThreads) void k_argmax_tokens(const LT *logits, int64_t n, int32_t C, int64_t row_stride, OT *tokens) {
    const int64_t stride = static_cast<int64_t>(gridDim.x) * kThreads;
    for (int64_t r = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; r < n; r += stride) {
        const LT *a = logits + r * row_stride;
        float best = static_cast<float>(a[0]);
        double bestd = static_cast<double>(a[0]);
        int32_t arg = 0;
        for (int32_t c = 1; c < C; ++c) {
            if constexpr (sizeof(LT) == 8) {
                const double v = static_cast<double>(a[c]);
                if (v > bestd || (v != v && bestd == bestd)) bestd = v, arg = c;  // NaN is the maximum, first one wins (torch.argmax)
            } else {
                const float v = static_cast<float>(a[c]);
                if (v > best || (v != v && best == best)) best = v, arg = c;
            }
        }
        tokens[r] = static_cast<OT>(arg);
    }
}

bsq_status fill_table(const bsq_desc *d, uint16_t tab[kTabN]) {
    for (int i = 0; i < kTabN; ++i) tab[i] = 0xFFFFu;
    for (int i = 0; i < 256; ++i) {  // first byte of every table value (tokenize.h:83-92)
        const int32_t v = d->lut[i];
        uint16_t &e = tab[v - kTabLo];
        if (e == 0xFFFFu) e = static_cast<uint16_t>(0x0100u | unsigned(i));
    }
    const int32_t ids[3] = {d->bos ? bsq_bos_id(d) : -1000, d->eos ? bsq_eos_id(d) : -1000, d->padchar ? bsq_pad_id(d) : -1000};
    for (int k = 0; k < 3; ++k)  // tokenize.h:93-101, in this order (a later one overwrites an earlier one)
        if (ids[k] != -1000) {
            if (ids[k] - kTabLo < 0 || ids[k] - kTabLo >= kTabN) return bsq_internal::set_error(BSQ_ERR_INVALID_ARG, "token id out of range");
            tab[ids[k] - kTabLo] = static_cast<uint16_t>(0x0200u | unsigned(k));
        }
    return BSQ_OK;
}

bsq_status prepare(DParams &p, const bsq_desc *d, const void *tokens, int32_t itemsize, int64_t nrows, int64_t ncols,
                   int64_t row_stride, int64_t col_stride) {
    if (!d || !tokens || nrows < 0 || ncols < 0) return bsq_internal::set_error(BSQ_ERR_INVALID_ARG, "null pointer or negative shape");
    if (itemsize != 1 && itemsize != 2 && itemsize != 4 && itemsize != 8)
        return bsq_internal::set_error(BSQ_ERR_DTYPE, "Unexpected itemsize: expected 1, 2, 4, or 8.");
    if (row_stride % itemsize || col_stride % itemsize || reinterpret_cast<uintptr_t>(tokens) % itemsize)
        return bsq_internal::set_error(BSQ_ERR_INVALID_ARG, "token strides / base must be multiples of the item size");
    const bsq_status st = fill_table(d, p.tab);
    if (st != BSQ_OK) return st;
    p.tokens = static_cast<const uint8_t *>(tokens);
    p.nrows = nrows;
    p.ncols = ncols;
    p.row_stride = row_stride;
    p.col_stride = col_stride;
    p.itemsize = itemsize;
    p.row_len = nullptr;
    p.row_off = nullptr;
    p.out = nullptr;
    p.first_bad = nullptr;
    return BSQ_OK;
}

}  // namespace

extern "C" {

bsq_status bsq_decode_sizes_device(const bsq_desc *d, const void *tokens, int32_t itemsize, int64_t nrows, int64_t ncols,
                                   int64_t row_stride, int64_t col_stride, int64_t *row_offsets, int64_t *total,
                                   int64_t *first_bad, void *hip_stream) {
    DParams p;
    bsq_status st = prepare(p, d, tokens, itemsize, nrows, ncols, row_stride, col_stride);
    if (st != BSQ_OK) return st;
    if (!row_offsets || !total || !first_bad) return bsq_internal::set_error(BSQ_ERR_INVALID_ARG, "null pointer");
    *total = 0;
    *first_bad = -1;
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    if ((nrows + 3) / 4 >= (int64_t(1) << 31)) return bsq_internal::set_error(BSQ_ERR_INVALID_ARG, "too many rows");
    unsigned long long bad = ~0ull;
    int64_t tot = 0;
    hipError_t e;
    {
        // the flag lives in the stream's shared scratch: the lock is held until the last operation that touches it is
        // ENQUEUED (bsq_internal.h), never across the synchronisation -- stream order protects it from later users
        std::lock_guard<std::mutex> turn(bsq_internal::workspace_mutex());
        void *ws = nullptr;
        st = bsq_internal::workspace_acquire(sizeof(unsigned long long), s, &ws);
        if (st != BSQ_OK) return st;
        p.first_bad = static_cast<unsigned long long *>(ws);
        p.row_len = row_offsets;
        e = hipMemsetAsync(ws, 0xFF, sizeof(unsigned long long), s);
        if (e == hipSuccess && nrows == 0) e = hipMemsetAsync(row_offsets, 0, sizeof(int64_t), s);
        if (e == hipSuccess && nrows > 0) {
            hipLaunchKernelGGL(k_decode_sizes, dim3(unsigned((nrows + 3) / 4)), dim3(kThreads), 0, s, p);
            hipLaunchKernelGGL(k_scan_rows, dim3(1), dim3(1024), 0, s, row_offsets, nrows);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpyAsync(&bad, ws, sizeof(bad), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipMemcpyAsync(&tot, row_offsets + nrows, sizeof(tot), hipMemcpyDeviceToHost, s);
        bsq_internal::workspace_release(ws, s);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) return bsq_internal::set_hip_error("bsq_decode_sizes_device", e);
    *total = tot;
    if (bad != ~0ull) {
        *first_bad = static_cast<int64_t>(bad);
        return bsq_internal::set_error(BSQ_ERR_INVALID_ARG, "Unexpected/invalid token");
    }
    return BSQ_OK;
}

bsq_status bsq_decode_write_device(const bsq_desc *d, const void *tokens, int32_t itemsize, int64_t nrows, int64_t ncols,
                                   int64_t row_stride, int64_t col_stride, const int64_t *row_offsets, uint8_t *out_chars,
                                   void *hip_stream) {
    DParams p;
    const bsq_status st = prepare(p, d, tokens, itemsize, nrows, ncols, row_stride, col_stride);
    if (st != BSQ_OK) return st;
    if (!row_offsets || (!out_chars && nrows > 0 && ncols > 0)) return bsq_internal::set_error(BSQ_ERR_INVALID_ARG, "null pointer");
    if (nrows == 0 || ncols == 0) return BSQ_OK;
    p.row_off = row_offsets;
    p.out = out_chars;
    hipLaunchKernelGGL(k_decode_write, dim3(unsigned((nrows + 3) / 4)), dim3(kThreads), 0, static_cast<hipStream_t>(hip_stream), p);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return bsq_internal::set_hip_error("k_decode_write", e);
    return BSQ_OK;
}

bsq_status bsq_argmax_tokens_device(const void *logits, int32_t logit_kind, int64_t n, int32_t C, int64_t row_stride,
                                    void *tokens, int32_t token_itemsize, void *hip_stream) {
    if (!logits || !tokens || n < 0 || C <= 0 || row_stride < C) return bsq_internal::set_error(BSQ_ERR_INVALID_ARG, "null pointer or bad shape");
    if (token_itemsize != 1 && token_itemsize != 4) return bsq_internal::set_error(BSQ_ERR_DTYPE, "tokens must be uint8 or int32");
    if (token_itemsize == 1 && C > 256) return bsq_internal::set_error(BSQ_ERR_INVALID_ARG, "uint8 tokens need C <= 256");
    if (n == 0) return BSQ_OK;
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    const int64_t blocks = (n + kThreads - 1) / kThreads;
    const dim3 grid(unsigned(blocks > 256 * 64 ? 256 * 64 : blocks));
#define BSQ_ARGMAX(LT)                                                                                                   \
    do {                                                                                                                 \
        if (token_itemsize == 1)                                                                                         \
            hipLaunchKernelGGL((k_argmax_tokens<LT, uint8_t>), grid, dim3(kThreads), 0, s, static_cast<const LT *>(logits), n, \
                               C, row_stride, static_cast<uint8_t *>(tokens));                                           \
        else                                                                                                             \
            hipLaunchKernelGGL((k_argmax_tokens<LT, int32_t>), grid, dim3(kThreads), 0, s, static_cast<const LT *>(logits), n, \
                               C, row_stride, static_cast<int32_t *>(tokens));                                           \
    } while (0)
    switch (logit_kind) {
    case BSQ_LOGITS_F32: BSQ_ARGMAX(float); break;
    case BSQ_LOGITS_F64: BSQ_ARGMAX(double); break;
    case BSQ_LOGITS_F16: BSQ_ARGMAX(_Float16); break;
    case BSQ_LOGITS_BF16: BSQ_ARGMAX(__bf16); break;
    default: return bsq_internal::set_error(BSQ_ERR_DTYPE, "logits must be f32, f64, f16 or bf16");
    }
#undef BSQ_ARGMAX
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return bsq_internal::set_hip_error("k_argmax_tokens", e);
    return BSQ_OK;
}

}  // extern "C"
