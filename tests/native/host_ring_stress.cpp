// The host staging code of libbsq_hip.so (bioseq_amd/csrc/bsq_host.cpp: the ring of three pinned + device input slots, uploads on a
// copy stream, encodes in pieces, staged batches with fetched results, the pipelined download) under ThreadSanitizer WITHOUT a GPU
// (the pool has no GPU sanitizers, and without a device every entry point stops at BSQ_ERR_NO_DEVICE).
//
// This file is test infrastructure.  It links bsq_host.cpp + bsq_alphabet.cpp with
//   * a mock HIP runtime: "device" memory is malloc'ed host memory, every stream is a worker thread with an in-order task queue
//     (copies and "kernels" really run asynchronously, a little late, so that a slot reused too early IS a data race TSan reports),
//     events record / wait / synchronise with HIP's semantics;
//   * mock encode entry points (bsq_tokenize_device, bsq_onehot_device, bsq_onehot_block_device, bsq_onehot_bcl_device): tasks on
//     their stream that READ the device input area when they run and write a simple function of it in the real output layouts; one of
//     them can be told to fail (the failing-piece exit of run_pieces: ADVICE round 4).
// Scenarios: back-to-back device results through bsq_pinned_scratch (ring reuse), host results (pipelined download), batches in
// pieces (knob host_pieces) with an injected failure in the middle, the staged-batch API with fetched results, two caller threads.
// Every output is compared with the same function computed directly from the caller's batch.  Built and run by tests/test_sanitizers.py.
#include <hip/hip_runtime_api.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <mutex>
#include <random>
#include <thread>
#include <vector>

#include "bsq.h"
#include "bsq_diag.h"

// ------------------------------------------------------------------------------------------------ mock HIP runtime
namespace {
struct MockStream {
    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::function<void()>> q;
    bool busy = false, stop = false;
    std::thread th;
    unsigned seed;
    explicit MockStream(unsigned s) : seed(s) { th = std::thread([this] { run(); }); }
    void run() {
        for (;;) {
            std::function<void()> f;
            {
                std::unique_lock<std::mutex> l(mu);
                cv.wait(l, [&] { return stop || !q.empty(); });
                if (q.empty()) return;
                f = std::move(q.front());
                q.pop_front();
                busy = true;
            }
            seed = seed * 1664525u + 1013904223u;  // a little late, differently every time
            if ((seed >> 28) == 0) std::this_thread::sleep_for(std::chrono::microseconds(30 + (seed >> 20) % 200));
            else if ((seed >> 30) == 0) std::this_thread::yield();
            f();
            {
                std::lock_guard<std::mutex> l(mu);
                busy = false;
            }
            cv.notify_all();
        }
    }
    void push(std::function<void()> f) {
        {
            std::lock_guard<std::mutex> l(mu);
            q.push_back(std::move(f));
        }
        cv.notify_all();
    }
    bool idle() {
        std::lock_guard<std::mutex> l(mu);
        return q.empty() && !busy;
    }
    void sync() {
        std::unique_lock<std::mutex> l(mu);
        cv.wait(l, [&] { return q.empty() && !busy; });
    }
};
struct MockEvent {
    std::mutex mu;
    std::condition_variable cv;
    uint64_t recorded = 0, completed = 0;
};
MockStream *g_null_stream = nullptr;
std::once_flag g_null_once;
std::atomic<unsigned> g_stream_ids{1};
MockStream *S(hipStream_t s) {
    if (s) return reinterpret_cast<MockStream *>(s);
    std::call_once(g_null_once, [] { g_null_stream = new MockStream(12345u); });
    return g_null_stream;
}
MockEvent *E(hipEvent_t e) { return reinterpret_cast<MockEvent *>(e); }
}  // namespace

extern "C" {
hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
hipError_t hipGetDevice(int *d) { *d = 0; return hipSuccess; }
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipDeviceCanAccessPeer(int *can, int, int) { *can = 1; return hipSuccess; }
hipError_t hipDeviceEnablePeerAccess(int, unsigned int) { return hipSuccess; }
hipError_t hipGetLastError(void) { return hipSuccess; }
const char *hipGetErrorString(hipError_t) { return "mock HIP error"; }
hipError_t hipMalloc(void **p, size_t n) {  // (device allocations are at least page-aligned)
    *p = std::aligned_alloc(4096, (n + 4095) / 4096 * 4096 + 4096);
    return *p ? hipSuccess : hipErrorOutOfMemory;
}
hipError_t hipFree(void *p) { std::free(p); return hipSuccess; }
hipError_t hipHostMalloc(void **p, size_t n, unsigned int) { *p = std::malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipHostFree(void *p) { std::free(p); return hipSuccess; }
hipError_t hipMallocAsync(void **p, size_t n, hipStream_t) { return hipMalloc(p, n); }
hipError_t hipFreeAsync(void *p, hipStream_t s) { S(s)->push([p] { std::free(p); }); return hipSuccess; }
hipError_t hipDeviceGetDefaultMemPool(hipMemPool_t *pool, int) { *pool = reinterpret_cast<hipMemPool_t>(uintptr_t(0x10)); return hipSuccess; }
hipError_t hipMemPoolSetAttribute(hipMemPool_t, hipMemPoolAttr, void *) { return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned int) { *s = reinterpret_cast<hipStream_t>(new MockStream(g_stream_ids.fetch_add(1) * 2654435761u)); return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t s) { S(s)->sync(); return hipSuccess; }
hipError_t hipStreamQuery(hipStream_t s) { return S(s)->idle() ? hipSuccess : hipErrorNotReady; }
hipError_t hipStreamIsCapturing(hipStream_t, hipStreamCaptureStatus *st) { *st = hipStreamCaptureStatusNone; return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = reinterpret_cast<hipEvent_t>(new MockEvent()); return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s) {
    MockEvent *ev = E(e);
    uint64_t gen;
    {
        std::lock_guard<std::mutex> l(ev->mu);
        gen = ++ev->recorded;
    }
    S(s)->push([ev, gen] {
        std::lock_guard<std::mutex> l(ev->mu);  // (notify under the lock: the event may be destroyed as soon as a waiter has seen it)
        if (ev->completed < gen) ev->completed = gen;
        ev->cv.notify_all();
    });
    return hipSuccess;
}
hipError_t hipEventSynchronize(hipEvent_t e) {
    MockEvent *ev = E(e);
    std::unique_lock<std::mutex> l(ev->mu);
    const uint64_t gen = ev->recorded;
    ev->cv.wait(l, [&] { return ev->completed >= gen; });
    return hipSuccess;
}
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned int) {
    MockEvent *ev = E(e);
    uint64_t gen;
    {
        std::lock_guard<std::mutex> l(ev->mu);
        gen = ev->recorded;
    }
    S(s)->push([ev, gen] {
        std::unique_lock<std::mutex> l(ev->mu);
        ev->cv.wait(l, [&] { return ev->completed >= gen; });
    });
    return hipSuccess;
}
hipError_t hipEventDestroy(hipEvent_t e) { delete E(e); return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) {
    MockStream *m = S(s);
    m->sync();
    {
        std::lock_guard<std::mutex> l(m->mu);
        m->stop = true;
    }
    m->cv.notify_all();
    m->th.join();
    delete m;
    return hipSuccess;
}
hipError_t hipMemcpyAsync(void *dst, const void *src, size_t n, hipMemcpyKind, hipStream_t s) {
    S(s)->push([dst, src, n] { std::memcpy(dst, src, n); });
    return hipSuccess;
}
}

// ------------------------------------------------------------------------------------------------ mock encode entry points
namespace {
std::atomic<int> g_fail_in{-1};  // the n-th block / device call from now fails (once)
inline uint8_t tok_of(const uint8_t *chars, const int64_t *offs, int64_t b, int64_t j) {
    const int64_t len = offs[b + 1] - offs[b];
    // (a function of the bytes and of WHERE they lie in the batch -- not of b itself: a piece sees its sequences as 0 ... n - 1)
    return j < len ? uint8_t((chars[offs[b] + j] + 1 + (offs[b] & 7)) & 0x7F) : uint8_t(0x7E);
}
bool injected_failure() {
    int v = g_fail_in.load();
    while (v >= 0) {
        if (g_fail_in.compare_exchange_weak(v, v - 1)) return v == 0;
    }
    return false;
}
}  // namespace

extern "C" {
bsq_status bsq_tokenize_device(const bsq_desc *, const uint8_t *chars, const int64_t *offsets, int64_t B, int64_t P, int32_t batch_first, bsq_dtype,
                               void *out, void *stream) {
    if (injected_failure()) return BSQ_ERR_HIP;
    S(static_cast<hipStream_t>(stream))->push([=] {
        uint8_t *o = static_cast<uint8_t *>(out);
        for (int64_t b = 0; b < B; ++b)
            for (int64_t j = 0; j < P; ++j) o[batch_first ? b * P + j : j * B + b] = tok_of(chars, offsets, b, j);
    });
    return BSQ_OK;
}
bsq_status bsq_onehot_block_device(const bsq_desc *d, const uint8_t *chars, const int64_t *offsets, const uint8_t *, int64_t B, int64_t P, bsq_dtype,
                                   void *out, int64_t row_seqs, void *stream) {
    if (injected_failure()) return BSQ_ERR_HIP;
    const int64_t C = bsq_alphabet_size(d);
    S(static_cast<hipStream_t>(stream))->push([=] {
        uint8_t *o = static_cast<uint8_t *>(out);
        for (int64_t j = 0; j < P; ++j)
            for (int64_t b = 0; b < B; ++b)
                for (int64_t c = 0; c < C; ++c) o[(j * row_seqs + b) * C + c] = uint8_t(tok_of(chars, offsets, b, j) % C == c);
    });
    return BSQ_OK;
}
bsq_status bsq_onehot_device(const bsq_desc *d, const uint8_t *chars, const int64_t *offsets, const uint8_t *m, int64_t B, int64_t P, bsq_dtype t, void *out,
                             void *stream) {
    return bsq_onehot_block_device(d, chars, offsets, m, B, P, t, out, B, stream);
}
bsq_status bsq_onehot_bcl_device(const bsq_desc *d, const uint8_t *chars, const int64_t *offsets, const uint8_t *, int64_t B, int64_t P, bsq_dtype, void *out,
                                 void *stream) {
    if (injected_failure()) return BSQ_ERR_HIP;
    const int64_t C = bsq_alphabet_size(d);
    S(static_cast<hipStream_t>(stream))->push([=] {
        uint8_t *o = static_cast<uint8_t *>(out);
        for (int64_t b = 0; b < B; ++b)
            for (int64_t c = 0; c < C; ++c)
                for (int64_t j = 0; j < P; ++j) o[(b * C + c) * P + j] = uint8_t(tok_of(chars, offsets, b, j) % C == c);
    });
    return BSQ_OK;
}
}

// ------------------------------------------------------------------------------------------------ the scenarios
namespace {
struct HostBatch {
    std::vector<uint8_t> chars;
    std::vector<int64_t> offs;
    int64_t B;
};
HostBatch make_batch(std::mt19937 &rng, int64_t B, int maxlen) {
    HostBatch h;
    h.B = B;
    h.offs.assign(size_t(B) + 1, 0);
    for (int64_t b = 0; b < B; ++b) h.offs[size_t(b) + 1] = h.offs[size_t(b)] + int64_t(rng() % unsigned(maxlen + 1));
    h.chars.resize(size_t(h.offs[size_t(B)]) + 1);
    for (uint8_t &c : h.chars) c = uint8_t("ACGT"[rng() & 3]);
    return h;
}
std::atomic<int> g_bad{0};
std::mutex g_pack_mu;  // pinned scratch + the call that consumes it are one critical section (the pybind11 layer's PackLock)
#define CHECK(c)                                                           \
    do {                                                                   \
        if (!(c)) {                                                        \
            std::printf("CHECK FAILED line %d: %s\n", __LINE__, #c);       \
            ++g_bad;                                                       \
        }                                                                  \
    } while (0)

bool tokens_ok(const HostBatch &h, int64_t P, bool bf, const uint8_t *o) {
    for (int64_t b = 0; b < h.B; ++b)
        for (int64_t j = 0; j < P; ++j)
            if (o[bf ? b * P + j : j * h.B + b] != tok_of(h.chars.data(), h.offs.data(), b, j)) return false;
    return true;
}
bool onehot_ok(const HostBatch &h, int64_t P, int64_t C, const uint8_t *o) {
    for (int64_t j = 0; j < P; ++j)
        for (int64_t b = 0; b < h.B; ++b)
            for (int64_t c = 0; c < C; ++c)
                if (o[(j * h.B + b) * C + c] != uint8_t(tok_of(h.chars.data(), h.offs.data(), b, j) % C == c)) return false;
    return true;
}
// pack a batch into the library's pinned scratch the way the pybind11 layer does (offsets | chars), return the pointers
void pack_pinned(const HostBatch &h, const int64_t **offs, const uint8_t **chars) {
    const size_t ob = (size_t(h.B) + 1) * 8, ob_al = (ob + 255) / 256 * 256;
    char *p = static_cast<char *>(bsq_pinned_scratch(ob_al + h.chars.size() + 64));
    std::memcpy(p, h.offs.data(), ob);
    std::memcpy(p + ob_al, h.chars.data(), h.chars.size());
    *offs = reinterpret_cast<const int64_t *>(p);
    *chars = reinterpret_cast<const uint8_t *>(p + ob_al);
}

void back_to_back(unsigned seed, hipStream_t stream, int rounds) {
    std::mt19937 rng(seed);
    bsq_desc d;
    CHECK(bsq_desc_init(&d, "DNA", 0, 0, 0) == BSQ_OK);
    const int64_t P = 12;
    const int kDepth = 5;  // results checked only after kDepth more calls were enqueued: the ring (3 slots) turns under the GPU's feet
    std::vector<HostBatch> hb;
    std::vector<uint8_t *> outs;
    for (int r = 0; r < rounds; ++r) {
        hb.push_back(make_batch(rng, 200 + int64_t(rng() % 900), int(P)));
        const HostBatch &h = hb.back();
        void *o = nullptr;
        CHECK(hipMalloc(&o, size_t(h.B * P)) == hipSuccess);
        outs.push_back(static_cast<uint8_t *>(o));
        int64_t bad = -1;
        const bool bf = (r & 1) != 0;
        if (r % 3 == 0) {  // caller-packed pinned scratch (the pybind layer's whole-batch path)
            std::lock_guard<std::mutex> pack(g_pack_mu);
            const int64_t *po;
            const uint8_t *pc;
            pack_pinned(h, &po, &pc);
            CHECK(bsq_tokenize_host(&d, pc, po, h.B, P, bf, BSQ_I8, o, BSQ_SPACE_DEVICE, stream, &bad) == BSQ_OK);
        } else {           // pageable caller memory
            std::lock_guard<std::mutex> pack(g_pack_mu);
            CHECK(bsq_tokenize_host(&d, h.chars.data(), h.offs.data(), h.B, P, bf, BSQ_I8, o, BSQ_SPACE_DEVICE, stream, &bad) == BSQ_OK);
        }
        if (r >= kDepth && r % kDepth == 0) {
            CHECK(hipStreamSynchronize(stream) == hipSuccess);
            for (int q = r - kDepth; q < r; ++q) CHECK(tokens_ok(hb[size_t(q)], P, (q & 1) != 0, outs[size_t(q)]));
        }
    }
    CHECK(hipStreamSynchronize(stream) == hipSuccess);
    for (size_t q = 0; q < hb.size(); ++q) CHECK(tokens_ok(hb[q], P, (q & 1) != 0, outs[q]));
    for (uint8_t *o : outs) std::free(o);
}

void host_results(unsigned seed) {
    std::mt19937 rng(seed);
    bsq_desc d;
    CHECK(bsq_desc_init(&d, "DNA", 0, 0, 0) == BSQ_OK);
    const int64_t P = 10, C = bsq_alphabet_size(&d);
    for (int r = 0; r < 12; ++r) {
        const HostBatch h = make_batch(rng, 3000 + int64_t(rng() % 3000), int(P));
        std::vector<uint8_t> out(size_t(P * h.B * C), 0xAB);
        int64_t bad = -1;
        CHECK(bsq_onehot_host(&d, h.chars.data(), h.offs.data(), nullptr, h.B, P, BSQ_I8, out.data(), BSQ_SPACE_HOST, nullptr, &bad) == BSQ_OK);
        CHECK(onehot_ok(h, P, C, out.data()));  // (the call returns after the D2H copy)
    }
}

void in_pieces_with_a_failure(unsigned seed) {
    std::mt19937 rng(seed);
    bsq_desc d;
    CHECK(bsq_desc_init(&d, "AMINO20", 0, 0, 0) == BSQ_OK);  // (20-byte rows: column blocks need rows of >= 16 bytes)
    const int64_t P = 6, C = bsq_alphabet_size(&d);
    CHECK(bsq_tuning_set("host_pieces", 4) == BSQ_OK);
    hipStream_t stream = nullptr;
    CHECK(hipStreamCreateWithFlags(&stream, 0) == hipSuccess);
    for (int r = 0; r < 10; ++r) {
        const HostBatch h = make_batch(rng, 4 * 4096 + int64_t(rng() % 2000), int(P));
        void *o = nullptr;
        CHECK(hipMalloc(&o, size_t(P * h.B * C)) == hipSuccess);
        int64_t bad = -1;
        const bool fail = r % 3 == 1;
        if (fail) g_fail_in.store(2);  // the third block of this call: two pieces are already on their way
        const bsq_status st = bsq_onehot_host(&d, h.chars.data(), h.offs.data(), nullptr, h.B, P, BSQ_I8, o, BSQ_SPACE_DEVICE, stream, &bad);
        g_fail_in.store(-1);
        CHECK(fail ? st != BSQ_OK : st == BSQ_OK);
        // The NEXT call packs into the ring at once: after a failed call the slot whose copies and kernels are still in flight must
        // not be handed out before they are done (TSan sees the race if it is; the result check sees stale input)
        const HostBatch h2 = make_batch(rng, 4 * 4096 + 17, int(P));
        void *o2 = nullptr;
        CHECK(hipMalloc(&o2, size_t(P * h2.B * C)) == hipSuccess);
        CHECK(bsq_onehot_host(&d, h2.chars.data(), h2.offs.data(), nullptr, h2.B, P, BSQ_I8, o2, BSQ_SPACE_DEVICE, stream, &bad) == BSQ_OK);
        const HostBatch h3 = make_batch(rng, 4 * 4096 + 33, int(P));
        void *o3 = nullptr;
        CHECK(hipMalloc(&o3, size_t(P * h3.B * C)) == hipSuccess);
        CHECK(bsq_onehot_host(&d, h3.chars.data(), h3.offs.data(), nullptr, h3.B, P, BSQ_I8, o3, BSQ_SPACE_DEVICE, stream, &bad) == BSQ_OK);
        const HostBatch h4 = make_batch(rng, 4 * 4096 + 49, int(P));
        void *o4 = nullptr;
        CHECK(hipMalloc(&o4, size_t(P * h4.B * C)) == hipSuccess);
        CHECK(bsq_onehot_host(&d, h4.chars.data(), h4.offs.data(), nullptr, h4.B, P, BSQ_I8, o4, BSQ_SPACE_DEVICE, stream, &bad) == BSQ_OK);
        CHECK(hipStreamSynchronize(stream) == hipSuccess);
        if (!fail) CHECK(onehot_ok(h, P, C, static_cast<uint8_t *>(o)));
        CHECK(onehot_ok(h2, P, C, static_cast<uint8_t *>(o2)));
        CHECK(onehot_ok(h3, P, C, static_cast<uint8_t *>(o3)));
        CHECK(onehot_ok(h4, P, C, static_cast<uint8_t *>(o4)));
        std::free(o);
        std::free(o2);
        std::free(o3);
        std::free(o4);
    }
    CHECK(bsq_tuning_set("host_pieces", 0) == BSQ_OK);
}

void staged_with_fetches(unsigned seed) {
    std::mt19937 rng(seed);
    bsq_desc d;
    CHECK(bsq_desc_init(&d, "DNA", 0, 0, 0) == BSQ_OK);
    const int64_t P = 9;
    for (int r = 0; r < 15; ++r) {
        const HostBatch h = make_batch(rng, 5000 + int64_t(rng() % 4000), int(P));
        std::unique_lock<std::mutex> pack(g_pack_mu);  // (begin ... end inside the caller's pack lock, as Tokenizer::staged has it)
        bsq_stage *st = nullptr;
        int64_t *po = nullptr;
        uint8_t *pc = nullptr;
        CHECK(bsq_stage_begin(h.B, h.chars.size(), 0, nullptr, &st, &po, &pc, nullptr) == BSQ_OK);
        void *d_res = nullptr, *h_res = nullptr;
        CHECK(bsq_stage_result(st, size_t(h.B * P), &d_res, &h_res) == BSQ_OK);
        std::vector<int32_t> tickets;
        std::vector<std::pair<int64_t, int64_t>> spans;
        const int64_t piece = 1 + h.B / (3 + r % 4);
        bool failed = false;
        for (int64_t lo = 0; lo < h.B && !failed; lo += piece) {
            const int64_t hi = std::min(h.B, lo + piece);
            for (int64_t b = lo; b < hi; ++b) po[b + 1] = h.offs[size_t(b) + 1];
            std::memcpy(pc + h.offs[size_t(lo)], h.chars.data() + h.offs[size_t(lo)], size_t(h.offs[size_t(hi)] - h.offs[size_t(lo)]));
            const int64_t *d_offs = nullptr;
            const uint8_t *d_chars = nullptr;
            CHECK(bsq_stage_upload(st, lo, hi, &d_offs, &d_chars, nullptr) == BSQ_OK);
            if (r % 5 == 4 && lo > 0) {  // a caller that gives up in the middle (an over-long item, a StageOverflow): end must still guard the slot
                failed = true;
                break;
            }
            CHECK(bsq_tokenize_device(&d, d_chars, d_offs, hi - lo, P, 1, BSQ_I8, static_cast<char *>(d_res) + lo * P, nullptr) == BSQ_OK);
            int32_t ticket = -1;
            CHECK(bsq_stage_fetch(st, size_t(lo * P), size_t((hi - lo) * P), &ticket) == BSQ_OK);
            tickets.push_back(ticket);
            spans.emplace_back(lo, hi);
        }
        std::vector<uint8_t> landed(size_t(h.B * P), 0);
        for (size_t k = 0; k < tickets.size(); ++k) {
            CHECK(bsq_stage_wait(st, tickets[k]) == BSQ_OK);
            std::memcpy(landed.data() + spans[k].first * P, static_cast<const uint8_t *>(h_res) + spans[k].first * P, size_t((spans[k].second - spans[k].first) * P));
        }
        CHECK(bsq_stage_end(st) == BSQ_OK);
        pack.unlock();
        if (!failed) CHECK(tokens_ok(h, P, true, landed.data()));
    }
}
}  // namespace

int main() {
    // one caller, its own stream and the null stream
    hipStream_t s1 = nullptr;
    CHECK(hipStreamCreateWithFlags(&s1, 0) == hipSuccess);
    back_to_back(1, s1, 60);
    back_to_back(2, nullptr, 30);
    host_results(3);
    in_pieces_with_a_failure(4);
    staged_with_fetches(5);
    // two caller threads at once (the library serialises them; their streams do not)
    hipStream_t s2 = nullptr, s3 = nullptr;
    CHECK(hipStreamCreateWithFlags(&s2, 0) == hipSuccess);
    CHECK(hipStreamCreateWithFlags(&s3, 0) == hipSuccess);
    std::thread a([&] { back_to_back(6, s2, 40); }), b([&] { back_to_back(7, s3, 40); }), c([&] { staged_with_fetches(8); });
    a.join();
    b.join();
    c.join();
    bsq_release_staging();
    if (g_bad.load()) {
        std::printf("HOST_RING_FAILED %d\n", g_bad.load());
        return 1;
    }
    std::printf("HOST_RING_OK\n");
    return 0;
}
