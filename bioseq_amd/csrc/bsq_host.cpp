// Host side of libbsq_hip.so: error reporting, device queries and the *_host entry points that
// stage a packed batch from host memory through pinned + device buffers.
//
// Replaces, on the reference side, the allocate + memset + OpenMP-loop body of
// Tokenizer::transencode<T> / Tokenizer::tokenize<T> (/root/reference/src/tokenize.h:420-427,
// :326-333) and the numpy -> torch -> .to(device) hand-off of bioseq/__init__.py:58-65: the
// encoded batch is produced on the accelerator instead of being copied to it.
//
// There is deliberately NO CPU implementation here: without a HIP device every compute entry
// point returns BSQ_ERR_NO_DEVICE.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <chrono>
#include <cctype>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "bsq.h"
#include "bsq_diag.h"
#include "bsq_internal.h"

namespace {
std::atomic<uint64_t> g_upload_bytes{0};  // bytes this library copied host -> device (bsq_host_upload_bytes)
}  // namespace

namespace {

thread_local std::string t_last_error;

// Grow-only staging buffers of one process (per HIP device).  All *_host calls are serialised by g_mu.
// The input side is a ring of kInSlots (pinned pack area, device input area, "busy" event) slots, so the host can gather +
// pack batch n + 1 while the GPU still copies / encodes batch n (`busy` marks the point after which a slot may be reused: the
// kernel that read its device area has finished).  THREE slots, not two: with two, pack(n + 2) waits for kernel(n) --
// pack 0.8 ms, upload 0.7 ms, kernel 0.73 ms on cfg3 then alternate between a free and a stalled call: 1.15 ms per batch
// back to back (rounds 2-3) instead of the 0.8 ms the slowest stage allows.
struct InSlot {
    void *pinned = nullptr;
    size_t pinned_cap = 0;
    void *d_in = nullptr;  // offsets | chars | mask
    size_t d_in_cap = 0;
    hipEvent_t busy = nullptr;
    bool busy_pending = false;
};
constexpr int kInSlots = 3;
struct Staging {
    int device = -1;
    InSlot in[kInSlots];
    int next = 0;  // slot the next call packs into
    hipStream_t copy_stream = nullptr;  // uploads run here, so that H2D of batch n + 1 overlaps the encode of batch n
    hipEvent_t uploaded = nullptr;
    hipEvent_t piece[8] = {};  // "piece j of this batch is on the device" (run_host in pieces; created on first use)
    hipEvent_t fetched[8] = {};  // "fetch k of this staged batch has landed in the pinned result area" (bsq_stage_fetch)
    void *d_out = nullptr;
    size_t d_out_cap = 0;
    void *h_result = nullptr;  // pinned landing area of a staged batch's HOST result (bsq_stage_result; grow-only)
    size_t h_result_cap = 0;
    // device -> pageable host results: ring of pinned bounce slots (see download())
    void *bounce = nullptr;
    hipEvent_t slot_done[4] = {nullptr, nullptr, nullptr, nullptr};
};
constexpr int kSlots = 4;
constexpr size_t kSlotMin = size_t(8) << 20, kSlotMax = size_t(32) << 20;  // bytes of a bounce slot: see download()
constexpr int kMaxDevices = 16;
Staging g_staging[kMaxDevices];
std::mutex g_mu;

size_t round_up(size_t n, size_t a) { return (n + a - 1) / a * a; }

bsq_status wait_idle(InSlot &s) {
    if (s.busy_pending) {
        const hipError_t e = hipEventSynchronize(s.busy);
        s.busy_pending = false;
        if (e != hipSuccess) return bsq_internal::set_hip_error("hipEventSynchronize", e);
    }
    return BSQ_OK;
}

bsq_status current_staging(Staging **out) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        return bsq_internal::set_error(BSQ_ERR_NO_DEVICE, bsq_strerror(BSQ_ERR_NO_DEVICE));
    }
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return bsq_internal::set_hip_error("hipGetDevice", e);
    if (dev < 0 || dev >= kMaxDevices) return bsq_internal::set_error(BSQ_ERR_INVALID_ARG, "device ordinal out of range");
    Staging &s = g_staging[dev];
    if (s.device < 0) {
        for (InSlot &slot : s.in) {
            e = hipEventCreateWithFlags(&slot.busy, hipEventDisableTiming);
            if (e != hipSuccess) return bsq_internal::set_hip_error("hipEventCreate", e);
        }
        e = hipStreamCreateWithFlags(&s.copy_stream, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&s.uploaded, hipEventDisableTiming);
        if (e != hipSuccess) return bsq_internal::set_hip_error("copy stream", e);
        s.device = dev;
    }
    *out = &s;
    return BSQ_OK;
}

bsq_status grow_device(void **buf, size_t *cap, size_t need) {
    if (need <= *cap) return BSQ_OK;
    if (*buf) (void)hipFree(*buf);
    *buf = nullptr;
    *cap = 0;
    const size_t want = round_up(need + need / 4, size_t(1) << 20);
    hipError_t e = hipMalloc(buf, want);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        e = hipMalloc(buf, round_up(need, 4096));
        if (e != hipSuccess) return bsq_internal::set_hip_error("hipMalloc(staging)", e);
        *cap = round_up(need, 4096);
        return BSQ_OK;
    }
    *cap = want;
    return BSQ_OK;
}

bsq_status grow_pinned(InSlot &s, size_t nbytes) {
    if (nbytes <= s.pinned_cap) return BSQ_OK;
    if (s.pinned) (void)hipHostFree(s.pinned);
    s.pinned = nullptr;
    s.pinned_cap = 0;
    const size_t want = round_up(nbytes + nbytes / 4, size_t(1) << 20);
    const hipError_t e = hipHostMalloc(&s.pinned, want, hipHostMallocDefault);
    if (e != hipSuccess) return bsq_internal::set_hip_error("hipHostMalloc", e);
    s.pinned_cap = want;
    return BSQ_OK;
}

bool in_pinned(const InSlot &s, const void *p) {
    const char *c = static_cast<const char *>(p), *b = static_cast<const char *>(s.pinned);
    return b && c >= b && c < b + s.pinned_cap;
}

struct DeviceBatch {
    const uint8_t *chars = nullptr;
    const int64_t *offsets = nullptr;
    const uint8_t *mask = nullptr;
};

// Copy offsets | chars | mask to the device staging buffer on `stream`.
bsq_status upload(InSlot &s, const uint8_t *chars, const int64_t *offsets, const uint8_t *mask, int64_t B,
                  hipStream_t stream, DeviceBatch *db) {
    const size_t total = static_cast<size_t>(offsets[B]);
    const size_t off_bytes = round_up(size_t(B + 1) * 8, 256);
    const size_t chr_bytes = round_up(total + 8, 256);  // +8: slack so the tail word is always mapped
    const size_t need = off_bytes + chr_bytes + (mask ? chr_bytes : 0);
    bsq_status st = grow_device(&s.d_in, &s.d_in_cap, need);
    if (st != BSQ_OK) return st;
    char *base = static_cast<char *>(s.d_in);
    g_upload_bytes.fetch_add(uint64_t(B + 1) * 8 + uint64_t(total) * (mask ? 2 : 1), std::memory_order_relaxed);
    // A batch packed as offsets | chars | mask in ONE span of memory (what the pybind layer packs into bsq_pinned_scratch) goes up
    // as one copy, the device pointers at the same distances: a call of BASELINE config 1's size (1000 sequences) is ~70 us of
    // fixed costs, a hipMemcpyAsync 5-8 us of them.
    const char *h0 = reinterpret_cast<const char *>(offsets), *h1 = reinterpret_cast<const char *>(chars),
               *h2 = reinterpret_cast<const char *>(mask);
    const size_t d1 = size_t(h1 - h0), d2 = mask ? size_t(h2 - h0) : 0;
    const bool one_span = total && h1 >= h0 + size_t(B + 1) * 8 && d1 <= off_bytes && (!mask || (h2 >= h1 + total && d2 - d1 <= chr_bytes));
    if (one_span) {
        const size_t span = (mask ? d2 : d1) + total;
        const hipError_t e1 = hipMemcpyAsync(base, offsets, span, hipMemcpyHostToDevice, stream);
        if (e1 != hipSuccess) return bsq_internal::set_hip_error("hipMemcpyAsync(H2D)", e1);
        db->offsets = reinterpret_cast<const int64_t *>(base);
        db->chars = reinterpret_cast<const uint8_t *>(base + d1);
        db->mask = mask ? reinterpret_cast<const uint8_t *>(base + d2) : nullptr;
        return BSQ_OK;
    }
    hipError_t e = hipMemcpyAsync(base, offsets, size_t(B + 1) * 8, hipMemcpyHostToDevice, stream);
    if (e == hipSuccess && total) e = hipMemcpyAsync(base + off_bytes, chars, total, hipMemcpyHostToDevice, stream);
    if (e == hipSuccess && mask && total)
        e = hipMemcpyAsync(base + off_bytes + chr_bytes, mask, total, hipMemcpyHostToDevice, stream);
    if (e != hipSuccess) return bsq_internal::set_hip_error("hipMemcpyAsync(H2D)", e);
    db->offsets = reinterpret_cast<const int64_t *>(base);
    db->chars = reinterpret_cast<const uint8_t *>(base + off_bytes);
    db->mask = mask ? reinterpret_cast<const uint8_t *>(base + off_bytes + chr_bytes) : nullptr;
    return BSQ_OK;
}

// Result -> caller's (pageable) host buffer.  A plain hipMemcpy into pageable memory runs at ~16 GB/s on
// the MI355X box (the runtime bounces through its own staging, and the first touch of a fresh numpy
// buffer page-faults inside that single thread).  Here the copy is pipelined instead: the stream DMAs
// 8 ... 32-MiB pieces into a ring of 4 pinned slots while `nthreads` workers memcpy finished slots into the
// destination (every worker takes its 1/n of each piece, so the page faults of the destination are spread
// over the workers too).  Small results take the plain path.  Knob "host_copy_threads" (default 8, 1 = plain).
bsq_status download(Staging &s, void *out, const void *dev_out, size_t nbytes, hipStream_t stream) {
    int nthreads = bsq_internal::tuning().host_copy_threads;
    if (nthreads <= 0) nthreads = 8;
    const unsigned hw = std::thread::hardware_concurrency();
    if (hw && unsigned(nthreads) > hw) nthreads = int(hw);
    hipError_t e = hipSuccess;
    // piece size: 1/16 of the result between 8 and 32 MiB -- 5.4 GB came back in 107.9 ms through 8-MiB pieces, in 101.9 through 32-MiB
    // ones (1.34 GB: 28.7 -> 26.9 ms; 64 MiB: no further gain; profiles/r04/default_call_lab.txt)
    static const size_t slot_cap = [] {  // BSQ_D2H_SLOT_MAX_MB = 8 ... 32 (measurement)
        const char *e = std::getenv("BSQ_D2H_SLOT_MAX_MB");
        const size_t mb = e && *e ? size_t(std::atoi(e)) : 32;
        return std::min(kSlotMax, std::max(kSlotMin, mb << 20));
    }();
    const size_t kSlotBytes = std::min(slot_cap, std::max(kSlotMin, round_up(nbytes / 16, size_t(1) << 20)));
    if (nbytes < 4 * kSlotMin || nthreads < 2) {
        e = hipMemcpyAsync(out, dev_out, nbytes, hipMemcpyDeviceToHost, stream);
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
        if (e != hipSuccess) return bsq_internal::set_hip_error("D2H copy of the result", e);
        return BSQ_OK;
    }
    if (!s.bounce) {
        e = hipHostMalloc(&s.bounce, kSlots * kSlotMax, hipHostMallocDefault);
        for (int i = 0; i < kSlots && e == hipSuccess; ++i) e = hipEventCreateWithFlags(&s.slot_done[i], hipEventDisableTiming);
        if (e != hipSuccess) return bsq_internal::set_hip_error("pinned bounce buffer", e);
    }
    const int64_t npieces = int64_t((nbytes + kSlotBytes - 1) / kSlotBytes);
    std::unique_ptr<std::atomic<int>[]> copied(new std::atomic<int>[size_t(npieces)]);
    for (int64_t c = 0; c < npieces; ++c) copied[size_t(c)].store(0, std::memory_order_relaxed);
    std::atomic<int64_t> issued{0};
    std::atomic<int> failed{0};
    const int device = s.device;
    char *dst = static_cast<char *>(out);
    char *ring = static_cast<char *>(s.bounce);
    auto worker = [&](int j) {
        (void)hipSetDevice(device);
        for (int64_t c = 0; c < npieces; ++c) {
            while (issued.load(std::memory_order_acquire) <= c) {
                if (failed.load(std::memory_order_relaxed)) return;
                std::this_thread::yield();
            }
            if (hipEventSynchronize(s.slot_done[c % kSlots]) != hipSuccess) {
                failed.store(1);
                return;
            }
            const size_t base = size_t(c) * kSlotBytes;
            const size_t len = nbytes - base < kSlotBytes ? nbytes - base : kSlotBytes;
            // page-granular shares; ceil(len / nthreads) first, so that per * nthreads >= len (with the floor, a piece
            // whose floor share was already a multiple of 4096 lost its last len % nthreads bytes)
            const size_t per = ((len + size_t(nthreads) - 1) / size_t(nthreads) + 4095) & ~size_t(4095);
            const size_t lo = per * size_t(j) < len ? per * size_t(j) : len;
            const size_t hi = lo + per < len ? lo + per : len;
            if (hi > lo) std::memcpy(dst + base + lo, ring + size_t(c % kSlots) * kSlotBytes + lo, hi - lo);
            copied[size_t(c)].fetch_add(1, std::memory_order_release);
        }
    };
    std::vector<std::thread> pool;
    try {
        pool.reserve(size_t(nthreads));
        for (int j = 0; j < nthreads; ++j) pool.emplace_back(worker, j);
    } catch (...) {  // no threads to be had (never let an exception cross the C ABI): plain copy instead
        failed.store(1);
        for (std::thread &t : pool) t.join();
        e = hipMemcpyAsync(out, dev_out, nbytes, hipMemcpyDeviceToHost, stream);
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
        if (e != hipSuccess) return bsq_internal::set_hip_error("D2H copy of the result", e);
        return BSQ_OK;
    }
    for (int64_t c = 0; c < npieces && !failed.load(); ++c) {
        if (c >= kSlots)  // the slot is free once every worker has drained piece c - kSlots
            while (copied[size_t(c - kSlots)].load(std::memory_order_acquire) < nthreads && !failed.load())
                std::this_thread::yield();
        const size_t base = size_t(c) * kSlotBytes;
        const size_t len = nbytes - base < kSlotBytes ? nbytes - base : kSlotBytes;
        e = hipMemcpyAsync(ring + size_t(c % kSlots) * kSlotBytes, static_cast<const char *>(dev_out) + base, len,
                           hipMemcpyDeviceToHost, stream);
        if (e == hipSuccess) e = hipEventRecord(s.slot_done[c % kSlots], stream);
        if (e != hipSuccess) {
            failed.store(1);
            break;
        }
        issued.store(c + 1, std::memory_order_release);
    }
    for (std::thread &t : pool) t.join();
    if (e != hipSuccess) return bsq_internal::set_hip_error("D2H copy of the result", e);
    if (failed.load()) return bsq_internal::set_error(BSQ_ERR_HIP, "pipelined D2H copy failed");
    return BSQ_OK;
}

template <typename Block>
bsq_status run_pieces(Staging &s, InSlot &slot, const uint8_t *chars, const int64_t *offsets, const uint8_t *mask, int64_t B, void *out,
                      hipStream_t stream, const Block &block) {
    const size_t total = static_cast<size_t>(offsets[B]);
    const size_t off_bytes = round_up(size_t(B + 1) * 8, 256);
    const size_t chr_bytes = round_up(total + 8, 256);
    bsq_status st = grow_device(&slot.d_in, &slot.d_in_cap, off_bytes + chr_bytes + (mask ? chr_bytes : 0));
    if (st != BSQ_OK) return st;
    char *base = static_cast<char *>(slot.d_in);
    g_upload_bytes.fetch_add(uint64_t(B + 1) * 8 + uint64_t(total) * (mask ? 2 : 1), std::memory_order_relaxed);
    hipError_t e = hipMemcpyAsync(base, offsets, size_t(B + 1) * 8, hipMemcpyHostToDevice, s.copy_stream);
    constexpr int kEvents = int(sizeof(s.piece) / sizeof(s.piece[0]));
    static const bool prof = std::getenv("BSQ_PROFILE_HOST") != nullptr;  // per-piece host times on stderr
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto us = [](auto a, auto b) { return long(std::chrono::duration_cast<std::chrono::microseconds>(b - a).count()); };
    int j = 0;
    for (int64_t b0 = 0, want = block.head + block.seqs; b0 < B && e == hipSuccess; b0 += want, want = block.seqs, ++j) {
        const int64_t n = B - b0 < want ? B - b0 : want;
        const size_t c0 = size_t(offsets[b0]), c1 = size_t(offsets[b0 + n]);
        const auto t0 = now();
        const auto t1 = now();
        if (c1 > c0) e = hipMemcpyAsync(base + off_bytes + c0, chars + c0, c1 - c0, hipMemcpyHostToDevice, s.copy_stream);
        if (e == hipSuccess && mask && c1 > c0)
            e = hipMemcpyAsync(base + off_bytes + chr_bytes + c0, mask + c0, c1 - c0, hipMemcpyHostToDevice, s.copy_stream);
        const auto t2 = now();
        hipEvent_t &ev = s.piece[j % kEvents];
        if (e == hipSuccess && !ev) e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventRecord(ev, s.copy_stream);
        if (e == hipSuccess) e = hipStreamWaitEvent(stream, ev, 0);
        if (e != hipSuccess) break;
        const auto t3 = now();
        DeviceBatch db;
        db.offsets = reinterpret_cast<const int64_t *>(base) + b0;
        db.chars = reinterpret_cast<const uint8_t *>(base + off_bytes);
        db.mask = mask ? reinterpret_cast<const uint8_t *>(base + off_bytes + chr_bytes) : nullptr;
        const int64_t lead = b0 == 0 && block.head < n ? block.head : 0;  // the sequences in front of the first chunk boundary
        if (lead) {
            st = block(db, lead, out, stream);
            if (st != BSQ_OK) return st;
            db.offsets += lead;
        }
        st = block(db, n - lead, static_cast<char *>(out) + size_t(b0 + lead) * block.row_bytes, stream);
        if (st != BSQ_OK) return st;
        if (prof)
            std::fprintf(stderr, "[bsq host] piece %d: pack %ld us, hipMemcpyAsync %ld us, event + wait %ld us, launches %ld us\n", j, us(t0, t1),
                         us(t1, t2), us(t2, t3), us(t3, now()));
    }
    if (e != hipSuccess) return bsq_internal::set_hip_error("upload in pieces", e);
    return BSQ_OK;
}

// A batch in PIECES (the seq-first one-hot with a device result, when nobody is ahead of this call on its stream): the characters go up
// in `count` slices of `seqs` sequences and the encode of slice j -- a column block of the (P, B, C) tensor -- runs while slice j + 1 is
// still on the bus.  One call then costs pack + upload + the LAST block's kernel instead of pack + upload + the whole kernel
// (cfg3 list -> device tensor, synchronous: 2.0-2.1 -> see profiles/r04/host_pieces_lab.txt).  `block` encodes sequences
// [b0, b0 + n) given the device batch (offsets pointer already advanced to b0) and the address of element (0, b0, 0).
struct NoPieces {
    int64_t seqs = 0, head = 0;  // pieces of `seqs` sequences; the first one holds `head` more in front (see piece_sequences)
    size_t row_bytes = 0;
    bsq_status operator()(const struct DeviceBatch &, int64_t, void *, hipStream_t) const { return BSQ_OK; }
};

template <typename Launch, typename Block = NoPieces>
bsq_status run_host(const bsq_desc *d, const uint8_t *chars, const int64_t *offsets, const uint8_t *mask,
                    int64_t B, int64_t P, int32_t bos, int32_t eos, size_t out_bytes, void *out,
                    bsq_space out_space, void *hip_stream, int64_t *first_bad, Launch launch, Block block = Block()) {
    if (first_bad) *first_bad = -1;
    if (!d || B < 0 || (B > 0 && (!offsets || !out))) return bsq_internal::set_error(BSQ_ERR_INVALID_ARG, "null pointer or B < 0");
    if (P <= 0) return bsq_internal::set_error(BSQ_ERR_INVALID_ARG, "batch tokenize requires padlen is provded.");
    if (B == 0) return BSQ_OK;  // an empty batch (e.g. a rank without sequences): nothing to encode, nothing to write
    int64_t bad = -1;
    bsq_status st = bsq_validate_lengths(offsets, B, P, bos, eos, &bad);
    if (first_bad) *first_bad = bad;
    if (st != BSQ_OK) return bsq_internal::set_error(st, bsq_strerror(st));
    if (B > 0 && offsets[B] > 0 && !chars) return bsq_internal::set_error(BSQ_ERR_INVALID_ARG, "chars is null");

    std::lock_guard<std::mutex> lock(g_mu);
    Staging *sp = nullptr;
    st = current_staging(&sp);
    if (st != BSQ_OK) return st;
    Staging &s = *sp;
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    if (out_bytes == 0 || B == 0) return BSQ_OK;
    // Input slot: the one the caller packed this batch into (bsq_pinned_scratch() already waited for it), else the
    // next one in turn -- after waiting for the call that used it two calls ago.
    InSlot *slot = nullptr;
    for (InSlot &cand : s.in)
        if (in_pinned(cand, offsets)) slot = &cand;
    if (!slot) {
        slot = &s.in[s.next];
        s.next = (s.next + 1) % kInSlots;
        st = wait_idle(*slot);
        if (st != BSQ_OK) return st;
    }
    if (block.seqs > 0 && out_space == BSQ_SPACE_DEVICE && block.seqs < B) {
        st = run_pieces(s, *slot, chars, offsets, mask, B, out, stream, block);
        // Whatever run_pieces returned: the copies and kernels of the pieces in front of a failing one are enqueued and read this slot's
        // pinned and device areas, so the slot is marked busy on EVERY exit (an early return left it looking idle and
        // the next call could overwrite what those kernels were still reading).  After a failure the copy stream is folded in first.
        hipError_t eb = hipSuccess;
        if (st != BSQ_OK) {
            eb = hipEventRecord(s.uploaded, s.copy_stream);
            if (eb == hipSuccess) eb = hipStreamWaitEvent(stream, s.uploaded, 0);
        }
        if (eb == hipSuccess) eb = hipEventRecord(slot->busy, stream);
        if (eb == hipSuccess) {
            slot->busy_pending = true;
        } else {  // no event to wait for later: wait now
            (void)hipGetLastError();
            (void)hipStreamSynchronize(s.copy_stream);
            (void)hipStreamSynchronize(stream);
            if (st == BSQ_OK) return bsq_internal::set_hip_error("hipEventRecord", eb);
        }
        return st;
    }
    // Upload on the copy stream (the slot is idle: waited for above or in bsq_pinned_scratch), then make the
    // caller's stream wait for it: the copy overlaps whatever that stream is still running (the previous encode).
    // (A small batch goes up on the caller's stream itself: nothing worth overlapping, and the event pair costs 4-6 us of a ~50-us call.)
    DeviceBatch db;
    const bool small = size_t(offsets[B]) * (mask ? 2 : 1) + size_t(B + 1) * 8 < (size_t(256) << 10);
    st = upload(*slot, chars, offsets, mask, B, small ? stream : s.copy_stream, &db);
    if (st != BSQ_OK) return st;
    if (!small) {
        hipError_t eu = hipEventRecord(s.uploaded, s.copy_stream);
        if (eu == hipSuccess) eu = hipStreamWaitEvent(stream, s.uploaded, 0);
        if (eu != hipSuccess) return bsq_internal::set_hip_error("upload -> encode dependency", eu);
    }
    void *dev_out = out;
    if (out_space == BSQ_SPACE_HOST) {
        st = grow_device(&s.d_out, &s.d_out_cap, out_bytes);
        if (st != BSQ_OK) return st;
        dev_out = s.d_out;
    }
    st = launch(db, dev_out, stream);
    if (st != BSQ_OK) return st;
    hipError_t e = hipSuccess;
    if (out_space == BSQ_SPACE_HOST) {
        st = download(s, out, dev_out, out_bytes, stream);
        if (st != BSQ_OK) return st;
    } else {
        e = hipEventRecord(slot->busy, stream);
        if (e != hipSuccess) return bsq_internal::set_hip_error("hipEventRecord", e);
        slot->busy_pending = true;
    }
    return BSQ_OK;
}

// Sequences per piece for run_host in pieces, or 0 = one upload and one encode.  Knob "host_pieces": 0 automatic, 1 never, N >= 2 that
// many pieces.  Automatic: 2 ... 8 pieces of ~8 MB of characters when the batch is large (>= 16 384 sequences, >= 4 MB of characters), the result's rows are >= 16
// bytes (the two-pass kernels take the blocks: pieces are multiples of 4096 sequences, so that every block row is whole 4-KiB chunks of
// an aligned tensor) and the caller's stream is IDLE -- somebody who keeps the stream busy is measuring throughput, where the host's own
// work per batch is what counts and one upload is cheaper than four.
int64_t piece_sequences(int64_t B, size_t nchars, size_t row_bytes, const void *out, hipStream_t stream, int64_t *head) {
    *head = 0;
    const int knob = bsq_internal::tuning().host_pieces;
    if (knob == 1) return -1;  // (the stage API's callers read < 0 as "no staging at all": the whole-batch path of rounds 1-3)
    // row_bytes == 0: the blocks of the result are contiguous (batch-first tokens, channels-first one-hot) -- any split will do.
    // Column blocks: pieces must start where a 4-KiB chunk of the result starts.  torch's allocator aligns to 512 bytes only, so the
    // first such sequence boundary is rarely 0: the `*head` sequences in front of it travel with the first piece and are encoded by a
    // call of their own (the tiled kernel; a few hundred sequences).
    int64_t lead = 0;
    if (row_bytes != 0) {
        if (row_bytes < 16) return 0;
        const size_t mis = reinterpret_cast<uintptr_t>(out) % 4096;
        if (mis != 0) {
            for (int64_t b = 1; b <= 4096 && lead == 0; ++b)
                if ((mis + size_t(b) * row_bytes) % 4096 == 0) lead = b;
            if (lead == 0) return 0;  // (an address that is not a multiple of gcd(row_bytes, 4096): no boundary is aligned)
        }
    }
    int64_t count = knob >= 2 ? knob : int64_t(nchars >> 23);  // automatic: ~8 MB of characters per piece, 2 ... 8 pieces
    if (count < 2) count = 2;
    if (count > 8) count = 8;
    if (knob == 0) {
        if (B < 16384 || nchars < (size_t(4) << 20)) return 0;
        if (hipStreamQuery(stream) != hipSuccess) {
            (void)hipGetLastError();
            return 0;
        }
    }
    const int64_t seqs = ((B - lead) / count) / 4096 * 4096;
    if (seqs < 4096 || lead + seqs >= B) return 0;
    *head = lead;
    return seqs;
}

}  // namespace

namespace bsq_internal {

bsq_status set_error(bsq_status st, const char *msg) {
    t_last_error = msg ? msg : "";
    return st;
}

bsq_status set_hip_error(const char *what, hipError_t e) {
    t_last_error = std::string(what ? what : "HIP") + ": " + hipGetErrorString(e);
    (void)hipGetLastError();
    return (e == hipErrorNoDevice || e == hipErrorInvalidDevice) ? BSQ_ERR_NO_DEVICE : BSQ_ERR_HIP;
}

namespace {
struct KnobInfo {
    const char *name;
    int32_t Tuning::*field;
};
const KnobInfo g_knobs[] = {
#define BSQ_KNOB_ENTRY(n, d) {#n, &Tuning::n},
    BSQ_KNOB_LIST(BSQ_KNOB_ENTRY)
#undef BSQ_KNOB_ENTRY
};
std::mutex g_knob_mu;                          // writers only
std::atomic<const Tuning *> g_tuning{nullptr};  // the published snapshot (superseded ones: see set_tuning)

const Tuning *initial_tuning() {  // defaults, then BSQ_<NAME> from the environment -- once
    Tuning *t = new Tuning();
    for (const KnobInfo &k : g_knobs) {
        std::string env = "BSQ_";
        for (const char *c = k.name; *c; ++c) env.push_back(char(std::toupper(static_cast<unsigned char>(*c))));
        const char *e = std::getenv(env.c_str());
        if (e && *e) t->*(k.field) = std::atoi(e);
    }
    return t;
}
const KnobInfo *find_knob(const char *name) {
    for (const KnobInfo &k : g_knobs)
        if (name && std::strcmp(k.name, name) == 0) return &k;
    return nullptr;
}
}  // namespace

// Stream-ordered scratch.  The two-pass one-hot path needs P x B bytes per call; a hipMallocAsync / hipFreeAsync pair
// per call puts ~5 us of stream operations between two steps (0.7 % of a cfg3 step), so the scratch of the last few
// (device, stream) pairs is KEPT: work on one stream is ordered, so consecutive calls on it may share the buffer without
// any further synchronisation.  While a stream is being captured into a HIP graph the allocation is made with
// hipMallocAsync / hipFreeAsync instead, so that the graph owns its memory and never points into this cache.
namespace {
struct WsEntry {
    hipStream_t stream = nullptr;
    int device = -1;
    void *ptr = nullptr;
    size_t bytes = 0;
    uint64_t tick = 0;
};
constexpr int kWsSlots = 4;
WsEntry g_ws[kWsSlots];
uint64_t g_ws_tick = 0;
std::mutex g_ws_mu;

bool capturing(hipStream_t stream) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &st) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return st != hipStreamCaptureStatusNone;
}

void keep_pool_memory(int dev) {
    static std::once_flag once[kMaxDevices];
    if (dev < 0 || dev >= kMaxDevices) return;
    std::call_once(once[dev], [dev] {  // keep freed blocks in the pool instead of returning them to the OS
        hipMemPool_t pool = nullptr;
        if (hipDeviceGetDefaultMemPool(&pool, dev) == hipSuccess) {
            uint64_t keep = ~uint64_t(0);
            (void)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep);
        }
        (void)hipGetLastError();
    });
}
}  // namespace

constexpr size_t kWsCacheCap = size_t(2) << 30;

bsq_status workspace_acquire(size_t nbytes, hipStream_t stream, void **ptr) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return set_hip_error("hipGetDevice", e);
    keep_pool_memory(dev);
    if (nbytes == 0) nbytes = 16;
    // Scratch above kWsCacheCap is never kept: one huge call would otherwise pin its bytes (invisible to the caller's
    // allocator) per (device, stream) slot until bsq_release_staging().  cfg3 needs 64 MiB, cfg4 160 MiB.
    if (capturing(stream) || nbytes > kWsCacheCap || tuning().workspace_cache == 1) {
        e = hipMallocAsync(ptr, nbytes, stream);
        if (e != hipSuccess) return set_hip_error("hipMallocAsync(workspace)", e);
        return BSQ_OK;
    }
    std::lock_guard<std::mutex> lock(g_ws_mu);
    WsEntry *hit = nullptr, *victim = &g_ws[0];
    for (WsEntry &w : g_ws) {
        if (w.ptr && w.device == dev && w.stream == stream) hit = &w;
        if (!w.ptr || (victim->ptr && w.tick < victim->tick)) victim = &w;
    }
    if (hit && hit->bytes >= nbytes) {
        hit->tick = ++g_ws_tick;
        *ptr = hit->ptr;
        return BSQ_OK;
    }
    WsEntry *slot = hit ? hit : victim;
    if (slot->ptr) {  // too small, or the least recently used stream makes room: freed in ITS stream's order
        int prev = dev;
        if (slot->device != dev) (void)hipSetDevice(slot->device);
        if (hipFreeAsync(slot->ptr, slot->stream) != hipSuccess) {  // e.g. the stream was destroyed meanwhile
            (void)hipGetLastError();
            (void)hipFree(slot->ptr);
        }
        if (slot->device != dev) (void)hipSetDevice(prev);
        *slot = WsEntry();
    }
    e = hipMallocAsync(ptr, nbytes, stream);
    if (e != hipSuccess) return set_hip_error("hipMallocAsync(workspace)", e);
    slot->stream = stream;
    slot->device = dev;
    slot->ptr = *ptr;
    slot->bytes = nbytes;
    slot->tick = ++g_ws_tick;
    return BSQ_OK;
}

void workspace_release(void *ptr, hipStream_t stream) {
    if (!ptr) return;
    {
        std::lock_guard<std::mutex> lock(g_ws_mu);
        for (const WsEntry &w : g_ws)
            if (w.ptr == ptr) return;  // cached: stays allocated for the next call on this stream
    }
    (void)hipFreeAsync(ptr, stream);
}

std::mutex &workspace_mutex() {
    static std::mutex m;
    return m;
}

void workspace_drop_cache() {
    // workspace_mutex() first: a caller that sits between workspace_acquire() and its last launch holds it, and its
    // scratch must not be freed under it
    std::lock_guard<std::mutex> turn(workspace_mutex());
    std::lock_guard<std::mutex> lock(g_ws_mu);
    int prev = 0;
    (void)hipGetDevice(&prev);
    for (WsEntry &w : g_ws) {
        if (!w.ptr) continue;
        (void)hipSetDevice(w.device);
        (void)hipStreamSynchronize(w.stream);
        (void)hipFree(w.ptr);
        w = WsEntry();
    }
    (void)hipSetDevice(prev);
    (void)hipGetLastError();
}

const Tuning &tuning() {
    const Tuning *t = g_tuning.load(std::memory_order_acquire);
    if (t) return *t;
    std::lock_guard<std::mutex> lock(g_knob_mu);
    t = g_tuning.load(std::memory_order_acquire);
    if (!t) {
        t = initial_tuning();
        g_tuning.store(t, std::memory_order_release);
    }
    return *t;
}

int get_tuning(const char *name) {
    const KnobInfo *k = find_knob(name);
    return k ? tuning().*(k->field) : 0;
}

bool set_tuning(const char *name, int value) {
    const KnobInfo *k = find_knob(name);
    if (!k) return false;
    (void)tuning();  // (the initial snapshot exists)
    std::lock_guard<std::mutex> lock(g_knob_mu);
    const Tuning *cur = g_tuning.load(std::memory_order_acquire);
    if (cur->*(k->field) == value) return true;  // nothing to publish (the `finally: set(knob, 0)` of every test)
    Tuning *next = new Tuning(*cur);
    next->*(k->field) = value;
    g_tuning.store(next, std::memory_order_release);
    // A launcher holds its snapshot for the microseconds of one call.  Superseded snapshots are freed once kRetired newer ones have
    // been published (a randomised harness sets knobs thousands of times; round 3 leaked every one of them).
    constexpr size_t kRetired = 1024;
    static const Tuning *retired[kRetired] = {};
    static size_t head = 0;
    delete retired[head];
    retired[head] = cur;
    head = (head + 1) % kRetired;
    return true;
}

}  // namespace bsq_internal

extern "C" {

const char *bsq_last_error(void) { return t_last_error.c_str(); }

bsq_status bsq_tuning_set(const char *name, int32_t value) {
    return bsq_internal::set_tuning(name, value) ? BSQ_OK
                                                 : bsq_internal::set_error(BSQ_ERR_INVALID_ARG, "unknown tuning knob");
}
int32_t bsq_tuning_get(const char *name) { return bsq_internal::get_tuning(name); }
uint64_t bsq_host_upload_bytes(void) { return g_upload_bytes.load(std::memory_order_relaxed); }

int32_t bsq_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

bsq_status bsq_enable_peer_access(int32_t device, int32_t peer) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n < 1) {
        (void)hipGetLastError();
        return bsq_internal::set_error(BSQ_ERR_NO_DEVICE, "no HIP device");
    }
    if (device < 0 || device >= n || peer < 0 || peer >= n) return bsq_internal::set_error(BSQ_ERR_INVALID_ARG, "bsq_enable_peer_access: no such device");
    if (device == peer) return BSQ_OK;
    int can = 0;
    hipError_t e = hipDeviceCanAccessPeer(&can, device, peer);
    if (e != hipSuccess) return bsq_internal::set_hip_error("hipDeviceCanAccessPeer", e);
    if (!can) return bsq_internal::set_error(BSQ_ERR_INVALID_ARG, "bsq_enable_peer_access: the device cannot map the peer's memory");
    int cur = 0;
    e = hipGetDevice(&cur);
    if (e == hipSuccess && cur != device) e = hipSetDevice(device);
    if (e == hipSuccess) {
        e = hipDeviceEnablePeerAccess(peer, 0);
        if (e == hipErrorPeerAccessAlreadyEnabled) {  // (torch enables it lazily for its own copies)
            (void)hipGetLastError();
            e = hipSuccess;
        }
    }
    if (cur != device) (void)hipSetDevice(cur);
    if (e != hipSuccess) return bsq_internal::set_hip_error("hipDeviceEnablePeerAccess", e);
    return BSQ_OK;
}

void *bsq_pinned_scratch(size_t nbytes) {
    std::lock_guard<std::mutex> lock(g_mu);
    Staging *sp = nullptr;
    if (current_staging(&sp) != BSQ_OK) return nullptr;
    InSlot &s = sp->in[sp->next];  // the slots take turns: the previous batches may still be in flight in the others
    sp->next = (sp->next + 1) % kInSlots;
    if (wait_idle(s) != BSQ_OK) return nullptr;
    return grow_pinned(s, nbytes) == BSQ_OK ? s.pinned : nullptr;
}

/* ---- staged batches (bsq.h): the slot of the ring is owned from bsq_stage_begin to bsq_stage_end (g_mu held) */
struct bsq_stage {
    std::unique_lock<std::mutex> lock;
    Staging *s = nullptr;
    InSlot *slot = nullptr;
    hipStream_t stream = nullptr;
    int64_t max_seqs = 0;
    size_t max_chars = 0;
    bool with_mask = false;
    int64_t *h_offsets = nullptr;
    uint8_t *h_chars = nullptr, *h_mask = nullptr;
    char *d_base = nullptr;
    size_t d_off_bytes = 0, d_chr_bytes = 0;
    int64_t uploaded = 0;  // sequences [0, uploaded) are on their way
    int pieces = 0;
    int fetches = 0;
    size_t result_bytes = 0;  // of THIS batch (bsq_stage_result); the buffers themselves outlive it
};

bsq_status bsq_stage_begin(int64_t max_seqs, size_t max_chars, int32_t with_mask, void *hip_stream, bsq_stage **stage,
                           int64_t **offsets, uint8_t **chars, uint8_t **mask) {
    if (!stage || !offsets || !chars || max_seqs < 0 || (with_mask && !mask))
        return bsq_internal::set_error(BSQ_ERR_INVALID_ARG, "bsq_stage_begin: null pointer or max_seqs < 0");
    *stage = nullptr;
    std::unique_ptr<bsq_stage> st(new (std::nothrow) bsq_stage());
    if (!st) return bsq_internal::set_error(BSQ_ERR_ALLOC, "bsq_stage");
    st->lock = std::unique_lock<std::mutex>(g_mu);
    bsq_status rc = current_staging(&st->s);
    if (rc != BSQ_OK) return rc;
    st->slot = &st->s->in[st->s->next];
    st->s->next = (st->s->next + 1) % kInSlots;
    rc = wait_idle(*st->slot);
    if (rc != BSQ_OK) return rc;
    const size_t h_off = round_up(size_t(max_seqs + 1) * 8, 64), h_chr = round_up(max_chars + 8, 64);
    rc = grow_pinned(*st->slot, h_off + h_chr * (with_mask ? 2 : 1));
    if (rc != BSQ_OK) return rc;
    st->d_off_bytes = round_up(size_t(max_seqs + 1) * 8, 256);
    st->d_chr_bytes = round_up(max_chars + 8, 256);  // +8: slack so the tail word is always mapped
    rc = grow_device(&st->slot->d_in, &st->slot->d_in_cap, st->d_off_bytes + st->d_chr_bytes * (with_mask ? 2 : 1));
    if (rc != BSQ_OK) return rc;
    st->stream = static_cast<hipStream_t>(hip_stream);
    st->max_seqs = max_seqs;
    st->max_chars = max_chars;
    st->with_mask = with_mask != 0;
    char *h = static_cast<char *>(st->slot->pinned);
    st->h_offsets = reinterpret_cast<int64_t *>(h);
    st->h_chars = reinterpret_cast<uint8_t *>(h + h_off);
    st->h_mask = with_mask ? st->h_chars + h_chr : nullptr;
    st->d_base = static_cast<char *>(st->slot->d_in);
    st->h_offsets[0] = 0;
    *offsets = st->h_offsets;
    *chars = st->h_chars;
    if (mask) *mask = st->h_mask;
    *stage = st.release();
    return BSQ_OK;
}

bsq_status bsq_stage_upload(bsq_stage *st, int64_t first, int64_t last, const int64_t **d_offsets, const uint8_t **d_chars,
                            const uint8_t **d_mask) {
    if (!st || !d_offsets || !d_chars) return bsq_internal::set_error(BSQ_ERR_INVALID_ARG, "bsq_stage_upload: null pointer");
    if (first != st->uploaded || last < first || last > st->max_seqs)
        return bsq_internal::set_error(BSQ_ERR_INVALID_ARG, "bsq_stage_upload: pieces must follow one another inside [0, max_seqs]");
    const int64_t c0 = st->h_offsets[first], c1 = st->h_offsets[last];
    if (c0 < 0 || c1 < c0 || size_t(c1) > st->max_chars)
        return bsq_internal::set_error(BSQ_ERR_INVALID_ARG, "bsq_stage_upload: offsets decrease or exceed max_chars");
    Staging &s = *st->s;
    hipStream_t copy = s.copy_stream;
    // offsets[first] went up as the previous piece's last entry (entry 0 with the first piece)
    const int64_t o0 = first == 0 ? 0 : first + 1;
    hipError_t e = hipSuccess;
    if (last + 1 > o0)
        e = hipMemcpyAsync(st->d_base + size_t(o0) * 8, st->h_offsets + o0, size_t(last + 1 - o0) * 8, hipMemcpyHostToDevice, copy);
    char *dc = st->d_base + st->d_off_bytes;
    if (e == hipSuccess && c1 > c0) e = hipMemcpyAsync(dc + c0, st->h_chars + c0, size_t(c1 - c0), hipMemcpyHostToDevice, copy);
    if (e == hipSuccess && st->with_mask && c1 > c0)
        e = hipMemcpyAsync(dc + st->d_chr_bytes + c0, st->h_mask + c0, size_t(c1 - c0), hipMemcpyHostToDevice, copy);
    constexpr int kEvents = int(sizeof(s.piece) / sizeof(s.piece[0]));
    hipEvent_t &ev = s.piece[st->pieces++ % kEvents];
    if (e == hipSuccess && !ev) e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventRecord(ev, copy);
    if (e == hipSuccess) e = hipStreamWaitEvent(st->stream, ev, 0);
    if (e != hipSuccess) return bsq_internal::set_hip_error("bsq_stage_upload", e);
    g_upload_bytes.fetch_add(uint64_t(last + 1 - o0) * 8 + uint64_t(c1 - c0) * (st->with_mask ? 2 : 1), std::memory_order_relaxed);
    st->uploaded = last;
    *d_offsets = reinterpret_cast<const int64_t *>(st->d_base) + first;
    *d_chars = reinterpret_cast<const uint8_t *>(dc);
    if (d_mask) *d_mask = st->with_mask ? reinterpret_cast<const uint8_t *>(dc + st->d_chr_bytes) : nullptr;
    return BSQ_OK;
}

bsq_status bsq_stage_end(bsq_stage *st) {
    if (!st) return BSQ_OK;
    std::unique_ptr<bsq_stage> owner(st);  // (unlocks g_mu)
    if (st->pieces == 0) return BSQ_OK;    // nothing was uploaded: nothing on the stream reads the slot
    const hipError_t e = hipEventRecord(st->slot->busy, st->stream);
    if (e != hipSuccess) return bsq_internal::set_hip_error("hipEventRecord", e);
    st->slot->busy_pending = true;
    return BSQ_OK;
}

bsq_status bsq_stage_result(bsq_stage *st, size_t nbytes, void **d_result, void **h_result) {
    if (!st || !d_result || !h_result) return bsq_internal::set_error(BSQ_ERR_INVALID_ARG, "bsq_stage_result: null pointer");
    Staging &s = *st->s;
    bsq_status rc = grow_device(&s.d_out, &s.d_out_cap, nbytes);
    if (rc != BSQ_OK) return rc;
    if (nbytes > s.h_result_cap) {
        if (s.h_result) (void)hipHostFree(s.h_result);
        s.h_result = nullptr;
        s.h_result_cap = 0;
        const size_t want = round_up(nbytes + nbytes / 8, size_t(1) << 20);
        const hipError_t e = hipHostMalloc(&s.h_result, want, hipHostMallocDefault);
        if (e != hipSuccess) return bsq_internal::set_hip_error("hipHostMalloc(result)", e);
        s.h_result_cap = want;
    }
    st->result_bytes = nbytes;
    *d_result = s.d_out;
    *h_result = s.h_result;
    return BSQ_OK;
}

bsq_status bsq_stage_fetch(bsq_stage *st, size_t offset, size_t nbytes, int32_t *ticket) {
    if (ticket) *ticket = -1;
    if (!st) return bsq_internal::set_error(BSQ_ERR_INVALID_ARG, "bsq_stage_fetch: null stage");
    Staging &s = *st->s;
    if (offset > st->result_bytes || nbytes > st->result_bytes - offset)
        return bsq_internal::set_error(BSQ_ERR_INVALID_ARG, "bsq_stage_fetch: outside the result of this batch's bsq_stage_result");
    if (nbytes == 0) return BSQ_OK;
    const hipError_t e = hipMemcpyAsync(static_cast<char *>(s.h_result) + offset, static_cast<const char *>(s.d_out) + offset, nbytes,
                                        hipMemcpyDeviceToHost, st->stream);
    if (e != hipSuccess) return bsq_internal::set_hip_error("bsq_stage_fetch", e);
    constexpr int kEvents = int(sizeof(s.fetched) / sizeof(s.fetched[0]));
    if (ticket && st->fetches < kEvents) {  // (a ninth fetch of one batch gets no ticket: wait for everything)
        hipEvent_t &ev = s.fetched[st->fetches];
        hipError_t e2 = hipSuccess;
        if (!ev) e2 = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
        if (e2 == hipSuccess) e2 = hipEventRecord(ev, st->stream);
        if (e2 != hipSuccess) return bsq_internal::set_hip_error("bsq_stage_fetch: event", e2);
        *ticket = st->fetches++;
    }
    return BSQ_OK;
}

bsq_status bsq_stage_wait(bsq_stage *st, int32_t ticket) {
    if (!st) return BSQ_OK;
    constexpr int kEvents = int(sizeof(st->s->fetched) / sizeof(st->s->fetched[0]));
    const hipError_t e = ticket >= 0 && ticket < kEvents && ticket < st->fetches ? hipEventSynchronize(st->s->fetched[ticket])
                                                                               : hipStreamSynchronize(st->stream);
    if (e != hipSuccess) return bsq_internal::set_hip_error("bsq_stage_wait", e);
    return BSQ_OK;
}

int64_t bsq_stage_piece_hint(int64_t B, size_t nchars, size_t block_row_bytes, const void *out, void *hip_stream, int64_t *head_seqs) {
    int64_t head = 0;
    const int64_t seqs = piece_sequences(B, nchars, block_row_bytes, out, static_cast<hipStream_t>(hip_stream), &head);
    if (head_seqs) *head_seqs = head;
    return seqs;  // (a caller that takes no head still gets pieces: bsq_onehot_block_device splits a large misaligned block itself)
}

void bsq_release_staging(void) {
    bsq_internal::workspace_drop_cache();
    std::lock_guard<std::mutex> lock(g_mu);
    for (Staging &s : g_staging) {
        if (s.device < 0) continue;
        int prev = 0;
        (void)hipGetDevice(&prev);
        (void)hipSetDevice(s.device);
        for (InSlot &slot : s.in) {
            if (slot.busy_pending) (void)hipEventSynchronize(slot.busy);
            if (slot.pinned) (void)hipHostFree(slot.pinned);
            if (slot.d_in) (void)hipFree(slot.d_in);
            if (slot.busy) (void)hipEventDestroy(slot.busy);
        }
        if (s.copy_stream) (void)hipStreamDestroy(s.copy_stream);
        if (s.uploaded) (void)hipEventDestroy(s.uploaded);
        for (hipEvent_t ev : s.piece)
            if (ev) (void)hipEventDestroy(ev);
        for (hipEvent_t ev : s.fetched)
            if (ev) (void)hipEventDestroy(ev);
        if (s.d_out) (void)hipFree(s.d_out);
        if (s.h_result) (void)hipHostFree(s.h_result);
        if (s.bounce) (void)hipHostFree(s.bounce);
        for (hipEvent_t ev : s.slot_done)
            if (ev) (void)hipEventDestroy(ev);
        (void)hipSetDevice(prev);
        s = Staging();
    }
}

bsq_status bsq_tokenize_host(const bsq_desc *d, const uint8_t *chars, const int64_t *offsets, int64_t B,
                             int64_t P, int32_t batch_first, bsq_dtype t, void *out, bsq_space out_space,
                             void *hip_stream, int64_t *first_bad) {
    const size_t sz = bsq_dtype_size(t);
    if (sz == 0) return bsq_internal::set_error(BSQ_ERR_DTYPE, "bad bsq_dtype");
    const size_t out_bytes = (B > 0 && P > 0) ? size_t(B) * size_t(P) * sz : 0;
    return run_host(d, chars, offsets, nullptr, B, P, d ? d->bos : 0, d ? d->eos : 0, out_bytes, out, out_space,
                    hip_stream, first_bad, [&](const DeviceBatch &db, void *dev_out, hipStream_t s) {
                        return bsq_tokenize_device(d, db.chars, db.offsets, B, P, batch_first, t, dev_out, s);
                    });
}

bsq_status bsq_onehot_host(const bsq_desc *d, const uint8_t *chars, const int64_t *offsets,
                           const uint8_t *mask_or_null, int64_t B, int64_t P, bsq_dtype t, void *out,
                           bsq_space out_space, void *hip_stream, int64_t *first_bad) {
    const size_t sz = bsq_dtype_size(t);
    if (sz == 0) return bsq_internal::set_error(BSQ_ERR_DTYPE, "bad bsq_dtype");
    const size_t C = d ? size_t(bsq_alphabet_size(d)) : 0;
    const size_t out_bytes = (B > 0 && P > 0) ? size_t(P) * size_t(B) * C * sz : 0;
    struct Block {
        int64_t seqs = 0, head = 0;
        size_t row_bytes = 0;
        const bsq_desc *d;
        int64_t B, P;
        bsq_dtype t;
        bsq_status operator()(const DeviceBatch &db, int64_t n, void *out_block, hipStream_t s) const {
            return bsq_onehot_block_device(d, db.chars, db.offsets, db.mask, n, P, t, out_block, B, s);
        }
    } block;
    block.d = d;
    block.B = B;
    block.P = P;
    block.t = t;
    block.row_bytes = C * sz;
    block.seqs = out_space == BSQ_SPACE_DEVICE && d && offsets && B > 0
                     ? piece_sequences(B, size_t(offsets[B]), block.row_bytes, out, static_cast<hipStream_t>(hip_stream), &block.head)
                     : 0;
    return run_host(d, chars, offsets, mask_or_null, B, P, d ? d->bos : 0, d ? d->eos : 0, out_bytes, out, out_space, hip_stream,
                    first_bad,
                    [&](const DeviceBatch &db, void *dev_out, hipStream_t s) {
                        return bsq_onehot_device(d, db.chars, db.offsets, db.mask, B, P, t, dev_out, s);
                    },
                    block);
}

bsq_status bsq_onehot_bcl_host(const bsq_desc *d, const uint8_t *chars, const int64_t *offsets,
                               const uint8_t *mask_or_null, int64_t B, int64_t P, bsq_dtype t, void *out,
                               bsq_space out_space, void *hip_stream, int64_t *first_bad) {
    const size_t sz = bsq_dtype_size(t);
    if (sz == 0) return bsq_internal::set_error(BSQ_ERR_DTYPE, "bad bsq_dtype");
    const size_t C = d ? size_t(bsq_alphabet_size(d)) : 0;
    const size_t out_bytes = (B > 0 && P > 0) ? size_t(P) * size_t(B) * C * sz : 0;
    return run_host(d, chars, offsets, mask_or_null, B, P, d ? d->bos : 0, d ? d->eos : 0, out_bytes, out,
                    out_space, hip_stream, first_bad, [&](const DeviceBatch &db, void *dev_out, hipStream_t s) {
                        return bsq_onehot_bcl_device(d, db.chars, db.offsets, db.mask, B, P, t, dev_out, s);
                    });
}

}  // extern "C"
