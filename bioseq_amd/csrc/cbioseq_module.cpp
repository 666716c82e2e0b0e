// pybind11 host layer: the Python class `Tokenizer` with the surface of the reference's
// `cbioseq.Tokenizer` (/root/reference/src/tokenize.cpp:22-112) and the thread knobs of
// /root/reference/src/omp.cpp:27-32, implemented on top of the C ABI in include/bsq.h.
//
// All encoding work happens in libbsq_hip.so on the GPU.  This file only
//   * gathers the str/bytes/bytearray items of a batch (GIL held, as in tokenize.h:389-419),
//   * packs them into the library's pinned scratch as  offsets | chars | mask,
//   * allocates the result (numpy array, or a torch tensor on `device=`) and calls the C ABI,
//   * keeps the small host-only pieces of the class (ids, decode tables, pickle).
// It does not link HIP or torch; torch is reached through its Python API for device memory and
// the current stream only.
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <algorithm>
#include <atomic>
#include <cctype>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <sstream>
#include <string>
#include <thread>
#include <unordered_map>
#include <pthread.h>

#include <condition_variable>
#include <functional>
#include <memory>
#include <vector>

#include "bsq.h"
#include "bsq_worker_pool.h"
#include "bsq_diag.h"

namespace py = pybind11;

namespace {

std::atomic<int> g_threads{0};  // 0: hardware concurrency
std::mutex g_pack_mu;           // pinned scratch is one buffer per device: pack + launch atomically

// Take g_pack_mu with the GIL released (a thread that waits for the mutex while holding the GIL would
// deadlock against the owner, which re-acquires the GIL after its GPU work).
struct PackLock {
    std::unique_lock<std::mutex> l;
    PackLock() : l(g_pack_mu, std::defer_lock) {
        py::gil_scoped_release nogil;
        l.lock();
    }
};

int default_threads() {
    const int t = g_threads.load();
    if (t > 0) return t;
    const unsigned hc = std::thread::hardware_concurrency();
    return hc ? int(hc) : 1;
}

// Leaked on purpose (joining threads during interpreter shutdown can deadlock); a forked child starts with a fresh pool
// (the parent's worker threads do not exist there).
std::atomic<WorkerPool *> g_pool{nullptr};
WorkerPool &pool() {
    WorkerPool *p = g_pool.load(std::memory_order_acquire);
    if (!p) {
        static std::once_flag atfork;
        std::call_once(atfork, [] { pthread_atfork(nullptr, nullptr, [] { g_pool.store(nullptr, std::memory_order_release); }); });
        WorkerPool *fresh = new WorkerPool();
        if (g_pool.compare_exchange_strong(p, fresh, std::memory_order_acq_rel)) p = fresh;
        else delete fresh;
    }
    return *p;
}

// nthreads of a batch call.  The SIGNATURE default is the reference's 1 (src/tokenize.cpp:81,98) -- inspect.signature of both batch
// methods equals the reference's for every reference parameter -- and in the reference that 1 sizes the OpenMP team of the encode
// loops.  Here the encode runs on the GPU whatever the value; what is left for host threads is the scan + pack of the Python items,
// and that is governed by a MODULE-LEVEL policy the reference's default defers to (round 5; was a signature default of 0):
//   * nthreads > 1   an explicit request: honoured as it is;
//   * nthreads <= 1  (the reference's default, and <= 0, which the reference maps to 1: tokenize.h:384) -> set_host_threads(n) /
//                    BSQ_HOST_THREADS: n > 0 exactly n threads (1 = strictly serial), 0 (default) automatic: one thread below
//                    8192 items, min(get_num_threads(), 16) workers above.
// Results never depend on any of this (tests/test_gpu_parity.py: nthreads sweep).
std::atomic<int> g_host_threads{-1};  // -1: not set yet (the environment decides at first use), 0 automatic, > 0 fixed
int host_threads_policy() {
    int t = g_host_threads.load();
    if (t < 0) {
        const char *e = std::getenv("BSQ_HOST_THREADS");
        int v = e ? std::atoi(e) : 0;
        if (v < 0) v = 0;
        if (v > 1024) v = 1024;
        int expect = -1;
        g_host_threads.compare_exchange_strong(expect, v);
        t = g_host_threads.load();
    }
    return t;
}
int resolve_threads(int nthreads, Py_ssize_t nitems) {
    if (nthreads > 1) return nthreads;
    const int policy = host_threads_policy();
    if (policy > 0) return policy;
    if (nitems < 8192) return 1;
    const int t = default_threads();
    return t > 16 ? 16 : (t < 1 ? 1 : t);
}

[[noreturn]] void throw_status(bsq_status st, const std::string &extra = std::string()) {
    std::string msg = extra.empty() ? std::string(bsq_strerror(st)) : extra;
    const char *detail = bsq_last_error();
    if (extra.empty() && detail && *detail && msg != detail) msg += std::string(": ") + detail;
    switch (st) {
    case BSQ_ERR_INVALID_ARG:
    case BSQ_ERR_DTYPE:
    case BSQ_ERR_SEQ_TOO_LONG: throw std::invalid_argument(msg);  // -> ValueError
    default: throw std::runtime_error(msg);                       // -> RuntimeError
    }
}

struct Item {
    const char *ptr;
    size_t len;
};

struct Gathered {
    std::vector<Item> items;
    std::vector<const uint8_t *> masks;  // empty when no mask list was given
    std::vector<py::object> keep;        // temporaries that own converted buffers
    size_t total = 0;
    bool has_mask = false;
};

// Pointer + length of the three item types whose bytes can be read without calling into the interpreter:
// bytes, bytearray and compact-ASCII str (UTF-8 == the stored bytes).  Reads object memory only, so worker
// threads may run it while the calling thread holds the GIL (nothing can mutate or free the items meanwhile).
inline bool fast_item(PyObject *o, Item *it) {
    if (PyBytes_CheckExact(o)) {
        *it = {PyBytes_AS_STRING(o), size_t(PyBytes_GET_SIZE(o))};
        return true;
    }
    if (PyUnicode_CheckExact(o) && PyUnicode_IS_READY(o) && PyUnicode_IS_COMPACT_ASCII(o)) {
        *it = {reinterpret_cast<const char *>(PyUnicode_1BYTE_DATA(o)), size_t(PyUnicode_GET_LENGTH(o))};
        return true;
    }
    if (PyByteArray_CheckExact(o)) {
        *it = {PyByteArray_AS_STRING(o), size_t(PyByteArray_GET_SIZE(o))};
        return true;
    }
    return false;
}

// Item acceptance of tokenize.h:292-322 / :389-419.  (The reference means to accept 8-bit numpy
// arrays too but falls through to its error label; they are accepted here.)
// Large batches without a mask list are first scanned by `nthreads` workers (the scan is bound by the cache
// misses on 64k object headers); whatever is not a plain bytes / bytearray / ASCII str is left to the serial pass.
// scan_begin: the list itself (item count, mask list); scan_range: pointer + length of items [lo, hi) into g.items / g.masks.
// The whole-batch path scans [0, n) at once; the staged path (Tokenizer::staged) piece by piece.
struct Scan {
    PyObject **objs = nullptr;
    Py_ssize_t n = 0;
    bool mask_is_list = false;
    py::list mlist;
};

Scan scan_begin(py::sequence batch, const py::object &mask, Gathered &g, int &nthreads) {
    py::object fast = py::reinterpret_steal<py::object>(PySequence_Fast(batch.ptr(), "batch must be a sequence"));
    if (!fast) throw py::error_already_set();
    Scan sc;
    sc.n = PySequence_Fast_GET_SIZE(fast.ptr());
    nthreads = resolve_threads(nthreads, sc.n);
    sc.objs = PySequence_Fast_ITEMS(fast.ptr());
    g.keep.push_back(fast);
    sc.mask_is_list = py::isinstance<py::list>(mask);  // anything else is ignored (tokenize.h:294)
    g.items.assign(size_t(sc.n), Item{nullptr, 0});
    if (sc.mask_is_list) {
        sc.mlist = py::reinterpret_borrow<py::list>(mask);
        g.masks.assign(size_t(sc.n), nullptr);
        g.has_mask = true;
    }
    return sc;
}

void scan_range(const Scan &sc, Gathered &g, Py_ssize_t lo, Py_ssize_t hi, int nthreads) {
    PyObject **objs = sc.objs;
    const Py_ssize_t n = hi - lo;
    std::vector<uint8_t> resolved;
    if (!sc.mask_is_list && nthreads > 1 && sc.n >= 8192 && n >= 1024) {
        resolved.assign(size_t(n), 0);
        std::vector<size_t> part(size_t(nthreads), 0);
        // (Software prefetch of the object headers / first data lines of the items ahead, here and in pack_range(): built and measured in
        //  round 4 on the list in allocation order and shuffled -- no effect, profiles/r04/host_prefetch_lab.txt -- and taken out.)
        auto scan = [&](int t) {
            size_t sum = 0;
            for (Py_ssize_t i = lo + n * t / nthreads, e = lo + n * (t + 1) / nthreads; i < e; ++i)
                if (fast_item(objs[i], &g.items[size_t(i)])) {
                    resolved[size_t(i - lo)] = 1;
                    sum += g.items[size_t(i)].len;
                }
            part[size_t(t)] = sum;
        };
        pool().parallel_for(nthreads, scan);
        for (size_t v : part) g.total += v;
    }
    for (Py_ssize_t i = lo; i < hi; ++i) {
        if (!resolved.empty() && resolved[size_t(i - lo)]) continue;
        PyObject *o = objs[i];
        Item it{nullptr, 0};
        if (PyUnicode_Check(o)) {
            Py_ssize_t sz = 0;
            const char *s = PyUnicode_AsUTF8AndSize(o, &sz);
            if (!s) throw py::error_already_set();
            it = {s, size_t(sz)};
        } else if (PyBytes_Check(o)) {
            char *s = nullptr;
            Py_ssize_t sz = 0;
            if (PyBytes_AsStringAndSize(o, &s, &sz)) throw py::error_already_set();
            it = {s, size_t(sz)};
        } else if (PyByteArray_Check(o)) {
            it = {PyByteArray_AS_STRING(o), size_t(PyByteArray_GET_SIZE(o))};
        } else if (py::isinstance<py::array>(py::handle(o))) {
            py::array a = py::reinterpret_borrow<py::array>(o);
            if (a.itemsize() != 1 || (a.dtype().kind() != 'i' && a.dtype().kind() != 'u' && a.dtype().kind() != 'S'))
                throw std::invalid_argument("item was none of string, bytes, or numpy array of 8-bit integers. ");
            py::array c = py::array::ensure(a, py::array::c_style);
            g.keep.push_back(c);
            it = {static_cast<const char *>(c.data()), size_t(c.size())};
        } else {
            throw std::invalid_argument("item was none of string, bytes, or numpy array of 8-bit integers. ");
        }
        if (sc.mask_is_list) {
            const uint8_t *mp = nullptr;
            if (size_t(i) >= sc.mlist.size()) throw py::index_error("list index out of range");
            py::object m = sc.mlist[size_t(i)];
            if (py::isinstance<py::array>(m)) {  // tokenize.h:372-380; other entries: sequence unmasked
                py::array_t<uint8_t, py::array::forcecast | py::array::c_style> arr(m);
                if (size_t(arr.size()) < it.len)
                    throw std::invalid_argument("mask entry " + std::to_string(i) + " is shorter than its sequence");
                g.keep.push_back(arr);
                mp = arr.data();
            }
            g.masks[size_t(i)] = mp;
        }
        g.total += it.len;
        g.items[size_t(i)] = it;
    }
}

// Layout inside the pinned scratch.
struct Packed {
    int64_t *offsets = nullptr;
    uint8_t *chars = nullptr;
    uint8_t *mask = nullptr;
    int64_t B = 0;
    size_t cap = SIZE_MAX;  // characters the area behind `chars` holds (staged batches: sized from an ESTIMATE, see estimate_chars)
};

// A staged batch whose characters outgrow the estimate its staging area was sized for: the piece loop stops before a byte of the
// offending piece is copied and the caller falls back to the whole-batch path (which measures the batch before it allocates).
struct StageOverflow {};

// What a list batch will need in the pinned / device staging areas, from the lengths of <= 512 evenly spaced items (object headers
// only) x 1.5 + 1 MiB, never more than the longest legal batch n * maxlen.  Round 4 sized the areas for n * maxlen itself (capped at
// 1 GiB): 16 384 sequences of ~200 characters under padlen 65 536 then pinned 3 x 1.25 GiB of host memory and as much on the device
// where a few MB were needed.  An under-estimate costs a fallback (StageOverflow), never a wrong result.
size_t estimate_chars(const Scan &sc, int64_t maxlen, size_t *mean_total) {
    const size_t n = size_t(sc.n), worst = n * size_t(maxlen);
    const size_t samples = std::min<size_t>(n, 512);
    size_t sum = 0;
    for (size_t k = 0; k < samples; ++k) {
        PyObject *o = sc.objs[k * n / samples];
        Item it{nullptr, 0};
        size_t len = size_t(maxlen);  // other item types (numpy arrays, non-ASCII str, subclasses): assume the longest legal one
        if (fast_item(o, &it)) len = it.len;
        else if (PyUnicode_Check(o) && PyUnicode_IS_READY(o)) len = size_t(PyUnicode_GET_LENGTH(o)) * 2;
        sum += std::min(len, size_t(maxlen));
    }
    const size_t est = samples ? size_t(double(sum) / double(samples) * double(n)) : 0;
    if (mean_total) *mean_total = std::min(worst, est);
    return std::min(worst, est + est / 2 + (size_t(1) << 20));
}

size_t align_up(size_t n, size_t a) { return (n + a - 1) / a * a; }

// pack = pack_begin (pinned scratch + offsets) + pack_range over every sequence; batch_onehot_encode hands pack_range to the
// library instead (bsq_onehot_host_pieces), which calls it piece by piece between its uploads.
Packed pack_begin(const Gathered &g) {
    Packed p;
    p.B = int64_t(g.items.size());
    const size_t off_bytes = align_up(size_t(p.B + 1) * 8, 64);
    const size_t chr_bytes = align_up(g.total + 8, 64);
    const size_t need = off_bytes + chr_bytes * (g.has_mask ? 2 : 1);
    char *base = static_cast<char *>(bsq_pinned_scratch(need));
    if (!base) throw_status(bsq_device_count() > 0 ? BSQ_ERR_ALLOC : BSQ_ERR_NO_DEVICE);
    p.offsets = reinterpret_cast<int64_t *>(base);
    p.chars = reinterpret_cast<uint8_t *>(base + off_bytes);
    p.mask = g.has_mask ? p.chars + chr_bytes : nullptr;
    int64_t acc = 0;
    for (int64_t i = 0; i < p.B; ++i) {
        p.offsets[i] = acc;
        acc += int64_t(g.items[size_t(i)].len);
    }
    p.offsets[p.B] = acc;
    return p;
}

void pack_range(const Gathered &g, const Packed &p, int64_t first, int64_t last, int nthreads) {
    auto copy_range = [&](int64_t lo, int64_t hi) {
        for (int64_t i = lo; i < hi; ++i) {
            const Item &it = g.items[size_t(i)];
            if (it.len) std::memcpy(p.chars + p.offsets[i], it.ptr, it.len);
            if (p.mask && it.len) {
                if (g.masks[size_t(i)])
                    std::memcpy(p.mask + p.offsets[i], g.masks[size_t(i)], it.len);
                else
                    std::memset(p.mask + p.offsets[i], 1, it.len);
            }
        }
    };
    const int64_t n = last - first;
    if (nthreads <= 1 || size_t(p.offsets[last] - p.offsets[first]) < (size_t(1) << 20) || n < 2 * nthreads) {
        copy_range(first, last);
    } else {  // workers touch raw bytes only; the caller keeps the GIL so the items stay alive
        // (Handing finished pieces to the copy engine while the rest is still being packed, WITHOUT encoding the pieces as they
        // arrive, was built and measured in round 3 -- profiles/r03/e2e_lab1.txt: no gain -- and taken out again; with the
        // encode of piece j under the upload of piece j + 1 it is what Tokenizer::staged does: profiles/r04/host_pieces_lab.txt.)
        const int nt = nthreads;
        pool().parallel_for(nt, [&](int t) { copy_range(first + n * t / nt, first + n * (t + 1) / nt); });
    }
}

// Scan AND pack of items [first, last) as ONE job of the pool, for the staged path (the pinned area is sized for the longest legal
// batch, so nothing needs the total first): every thread reads pointer + length of its share of the items, all meet at a barrier,
// every thread derives the offset of its first item from the shares before it, writes its offsets and copies its bytes.  One pool
// job, no serial pass over the items.  Plain bytes / bytearray / ASCII str only: false = some item is something else (or there
// is a mask list) and nothing usable was written -- the caller takes scan_range + pack_range for this piece.  An item longer than
// maxlen stops everything before a byte is copied: *too_long = the first such item and p.offsets[*too_long + 1] holds its end.
bool scan_pack_fast(const Scan &sc, Gathered &g, const Packed &p, int64_t first, int64_t last, int nthreads, int64_t maxlen, int64_t *too_long) {
    *too_long = -1;  // (throws StageOverflow when the piece does not fit behind p.chars: nothing of it has been copied then)
    const int64_t n = last - first;
    const int nt = int(std::min<int64_t>(std::min(nthreads, 64), n / 256));  // (every task of the job needs a thread of its own: barrier)
    if (sc.mask_is_list || nt < 2) return false;
    struct Part {
        size_t sum = 0;
        int64_t odd = -1, too_long = -1;  // first item of the share that is not a fast item / that is too long
        char pad[40];
    };
    std::vector<Part> part(size_t(nt), Part{});
    std::atomic<int> arrived{0}, verdict{0};  // verdict: 1 go on, 2 stop, 3 stop: the piece does not fit (p.cap)
    PyObject **objs = sc.objs;
    pool().parallel_for(nt, [&](int t) {
        const int64_t i0 = first + n * t / nt, i1 = first + n * (t + 1) / nt;
        Part mine;
        for (int64_t i = i0; i < i1; ++i) {
            Item &it = g.items[size_t(i)];
            if (!fast_item(objs[i], &it)) {
                mine.odd = i;
                break;
            }
            if (int64_t(it.len) > maxlen && mine.too_long < 0) mine.too_long = i;
            mine.sum += it.len;
        }
        part[size_t(t)] = mine;
        if (arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == nt) {  // the last one in decides for all
            bool go = true;
            size_t end = size_t(p.offsets[first]);
            for (const Part &q : part) {
                go = go && q.odd < 0 && q.too_long < 0;
                end += q.sum;
            }
            verdict.store(!go ? 2 : (end > p.cap ? 3 : 1), std::memory_order_release);
        }
        int v;
        for (unsigned spins = 0; (v = verdict.load(std::memory_order_acquire)) == 0; ++spins) {
            if (spins < (1u << 16)) __builtin_ia32_pause();
            else std::this_thread::yield();
        }
        if (v != 1) return;
        int64_t at = p.offsets[first];
        for (int u = 0; u < t; ++u) at += int64_t(part[size_t(u)].sum);
        for (int64_t i = i0; i < i1; ++i) {
            const Item &it = g.items[size_t(i)];
            if (it.len) std::memcpy(p.chars + at, it.ptr, it.len);
            at += int64_t(it.len);
            p.offsets[i + 1] = at;  // (offsets[i0] is the end the share before writes; offsets[first] was there on entry)
        }
    });
    for (const Part &q : part)
        if (q.odd >= 0) return false;
    if (verdict.load(std::memory_order_acquire) == 3) throw StageOverflow{};
    for (const Part &q : part)
        if (q.too_long >= 0) {  // the first too-long item: lengths up to it are all known
            int64_t at = p.offsets[first];
            for (int64_t i = first; i <= q.too_long; ++i) {
                p.offsets[i] = at;
                at += int64_t(g.items[size_t(i)].len);
            }
            p.offsets[q.too_long + 1] = at;
            *too_long = q.too_long;
            return true;
        }
    return true;
}

Packed pack(const Gathered &g, int nthreads) {
    const Packed p = pack_begin(g);
    pack_range(g, p, 0, p.B, nthreads);
    return p;
}

const char *numpy_dtype_name(bsq_dtype t) {
    switch (t) {
    case BSQ_I8: return "int8";
    case BSQ_I16: return "int16";
    case BSQ_I32: return "int32";
    case BSQ_U64: return "uint64";
    case BSQ_F32: return "float32";
    default: return "float64";
    }
}

bsq_dtype parse_dtype(const std::string &dt) {
    bsq_dtype t;
    if (dt.empty() || bsq_dtype_from_destchar(dt[0], &t) != BSQ_OK)
        throw std::invalid_argument(std::string("Unsupported dtype: ") + dt);
    return t;
}

// Result buffer: a numpy array (host) or a torch tensor on `device`.
struct OutBuf {
    py::object obj;
    void *ptr = nullptr;
    bsq_space space = BSQ_SPACE_HOST;
    void *stream = nullptr;
    py::object guard;  // torch.cuda.device(...) context, entered
    ~OutBuf() {
        if (guard) {
            try {
                guard.attr("__exit__")(py::none(), py::none(), py::none());
            } catch (...) {
            }
        }
    }
};

// torch, looked up once (a call of BASELINE config 1's size is ~70 us of fixed costs; module / attribute lookups and the
// torch.cuda.device context manager were ~15 of them).  Leaked on purpose: destroying py::objects at interpreter exit is not safe.
struct TorchApi {
    py::object device, empty, current_stream, current_device, cuda_device, raw_stream, dtypes[6];  // raw_stream: None when torch has none
};
const TorchApi &torch_api() {
    static const TorchApi *api = [] {
        py::module_ torch = py::module_::import("torch");
        TorchApi *a = new TorchApi();
        a->device = torch.attr("device");
        a->empty = torch.attr("empty");
        a->current_stream = torch.attr("cuda").attr("current_stream");
        a->current_device = torch.attr("cuda").attr("current_device");
        a->cuda_device = torch.attr("cuda").attr("device");
        // torch._C._cuda_getCurrentRawStream(device_index) -> int: what torch.cuda.current_stream() wraps in ~9 us of Python
        a->raw_stream = py::getattr(torch.attr("_C"), "_cuda_getCurrentRawStream", py::none());
        const char *names[6] = {"int8", "int16", "int32", "int64", "float32", "float64"};  // (BSQ_U64 -> int64: see below)
        for (int i = 0; i < 6; ++i) a->dtypes[i] = torch.attr(names[i]);
        return a;
    }();
    return *api;
}

void make_out(OutBuf &o, const std::vector<py::ssize_t> &shape, bsq_dtype t, const py::object &device) {
    if (device.is_none()) {
        py::array a(py::dtype(numpy_dtype_name(t)), shape);
        o.ptr = a.mutable_data();
        o.obj = a;
        o.space = BSQ_SPACE_HOST;
        return;
    }
    const TorchApi &T = torch_api();
    py::object dev = T.device(device);
    if (dev.attr("type").cast<std::string>() != "cuda")
        throw std::invalid_argument("device= must be a HIP ('cuda') device; omit it for a numpy result");
    // make `device=` the current device for the staging buffers and the launch -- through torch's context manager only when it
    // is not the current one already (the common case)
    const py::object index = dev.attr("index");
    const int current = T.current_device().cast<int>(), want = index.is_none() ? current : index.cast<int>();
    if (want != current) {
        o.guard = T.cuda_device(dev);
        o.guard.attr("__enter__")();
    }
    // 'l' / 'q' results are uint64 in numpy (the reference's type); as device tensors they are torch.int64 -- same
    // bits for token ids and 0/1, and torch.uint64 does not exist before torch 2.3 and supports almost no ops after
    static const int dtype_index[6] = {0, 1, 2, 3, 4, 5};  // bsq_dtype order: I8, I16, I32, U64, F32, F64
    py::object ten = T.empty(py::cast(shape), py::arg("dtype") = T.dtypes[dtype_index[int(t)]], py::arg("device") = dev);
    o.ptr = reinterpret_cast<void *>(ten.attr("data_ptr")().cast<uintptr_t>());
    o.stream = reinterpret_cast<void *>(T.raw_stream.is_none() ? T.current_stream().attr("cuda_stream").cast<uintptr_t>()
                                                               : T.raw_stream(want).cast<uintptr_t>());
    o.obj = ten;
    o.space = BSQ_SPACE_DEVICE;
}

// A packed-batch argument: numpy array (host) or torch tensor on a HIP device.
struct ArrayArg {
    const void *ptr = nullptr;
    int64_t n = 0;
    bool on_device = false;
    py::object keep;
};

ArrayArg as_array(const py::object &o, const char *np_dtype, size_t itemsize, const char *what) {
    ArrayArg a;
    if (py::hasattr(o, "data_ptr") && py::hasattr(o, "is_cuda")) {  // torch tensor
        py::object t = o;
        if (!t.attr("is_contiguous")().cast<bool>()) t = t.attr("contiguous")();
        // one-byte arguments (chars, mask): any 1-byte integer or bool tensor is the same bytes, as the numpy branch below
        // accepts any 1-byte array; wider arguments (offsets) must be exactly the named type -- float64 offsets were the hazard
        const std::string dt = py::str(t.attr("dtype")).cast<std::string>();
        const bool same_bytes = itemsize == 1 && (dt == "torch.uint8" || dt == "torch.int8" || dt == "torch.bool");
        if (!same_bytes && dt != std::string("torch.") + np_dtype)
            throw std::invalid_argument(std::string(what) + ": expected a torch." + np_dtype + " tensor" +
                                        (itemsize == 1 ? " (or torch.int8 / torch.bool)" : ""));
        a.on_device = t.attr("is_cuda").cast<bool>();
        a.ptr = reinterpret_cast<const void *>(t.attr("data_ptr")().cast<uintptr_t>());
        a.n = t.attr("numel")().cast<int64_t>();
        a.keep = t;
        return a;
    }
    py::array arr = py::array::ensure(o, py::array::c_style);
    if (!arr) throw std::invalid_argument(std::string(what) + ": expected a numpy array or torch tensor");
    if (size_t(arr.itemsize()) != itemsize) {
        arr = py::array::ensure(arr.attr("astype")(np_dtype), py::array::c_style);
    }
    a.ptr = arr.data();
    a.n = int64_t(arr.size());
    a.keep = arr;
    return a;
}

class Tokenizer {
  public:
    bsq_desc desc{};
    std::string key;
    std::unordered_map<int32_t, std::string> lookup;     // token -> first byte / "<BOS>"...  (tokenize.h:83-99)
    std::unordered_map<int32_t, std::string> tokensets;  // token -> every byte of the group
    std::string token_map_str;

    Tokenizer(std::string key_, bool eos, bool bos, bool padchar) : key(std::move(key_)) {
        std::transform(key.begin(), key.end(), key.begin(), [](unsigned char c) { return char(std::toupper(c)); });
        const bsq_status st = bsq_desc_init(&desc, key.c_str(), eos, bos, padchar);
        if (st == BSQ_ERR_INVALID_KEY) {
            std::string options;
            for (int i = 0; i < bsq_num_keys(); ++i) options += std::string(bsq_key_name(i)) + ';';
            throw std::runtime_error(std::string("Invalid tokenizer type; select one from") + options);
        }
        if (st != BSQ_OK) throw_status(st);
        for (int32_t i = 0; i < 256; ++i) {
            const int32_t value = desc.lut[i];
            if (lookup.find(value) == lookup.end()) lookup[value] = std::string(1, char(i));
            tokensets[value] += char(i);
        }
        if (desc.bos) lookup[bsq_bos_id(&desc)] = "<BOS>";
        if (desc.eos) lookup[bsq_eos_id(&desc)] = "<EOS>";
        if (desc.padchar) lookup[bsq_pad_id(&desc)] = "<PAD>";
        for (const auto &kv : lookup) token_map_str += std::to_string(kv.first) + ':' + kv.second + ';';
        if (!token_map_str.empty()) token_map_str.pop_back();
    }

    void check_padlen(py::ssize_t padlen) const {
        if (padlen <= 0) throw std::invalid_argument("batch tokenize requires padlen is provded.");
    }

    // The reference throws std::runtime_error from batch_tokenize (tokenize.h:456-459) and std::invalid_argument from
    // batch_onehot_encode (:359-362) -- inside an OpenMP region, so its process aborts; here they reach Python as
    // RuntimeError / ValueError with the reference's text.
    [[noreturn]] void throw_too_long(const int64_t *offsets, int64_t bad, py::ssize_t padlen, bool onehot) const {
        const int64_t tl = offsets[bad + 1] - offsets[bad] + desc.bos + desc.eos;
        const std::string msg = "seq len + bos + eos > padlen: " + std::to_string(tl) + ", vs padlen " + std::to_string(padlen);
        if (onehot) throw std::invalid_argument(msg);
        throw std::runtime_error(msg);
    }

    // A list -> DEVICE result in pieces (include/bsq.h, "staged batches"): items [lo, hi) are scanned, packed into the pinned staging
    // area, sent on their way and encoded as a block of the result while the next piece is scanned and packed -- the call costs
    // one piece of host work + the upload + one piece of kernel instead of the sum of the three.  false = not applicable (small
    // batch, numpy result, knob host_pieces = 1): nothing has been scanned, the caller goes on with the whole-batch path.
    // The GIL stays held as it is during every pack (the items must stay alive and unchanged); nothing in here blocks on the
    // GPU except the wait for the staging slot used three calls ago.
    // The piece loop of both staged paths: items [lo, hi) scanned + packed (one pool job, or the general passes), uploaded, then
    // `per_piece(lo, hi, d_chars, d_offsets, d_mask)` (device pointers of the whole batch; d_offsets at sequence lo).
    template <typename Produce, typename PerPiece>
    static void piece_loop(int64_t n, bsq_stage *stage, int64_t head, int64_t seqs, Produce produce, PerPiece per_piece) {
        for (int64_t lo = 0, want = head + seqs; lo < n; lo += want, want = seqs) {
            const int64_t hi = std::min<int64_t>(n, lo + want);
            produce(lo, hi);  // offsets[lo + 1 .. hi] and the characters of [lo, hi) into the pinned staging area (may throw)
            const int64_t *d_offsets = nullptr;
            const uint8_t *d_chars = nullptr, *d_mask = nullptr;
            bsq_status st = bsq_stage_upload(stage, lo, hi, &d_offsets, &d_chars, &d_mask);
            if (st == BSQ_OK) st = per_piece(lo, hi, d_chars, d_offsets, d_mask);
            if (st != BSQ_OK) throw_status(st);
        }
    }
    template <typename PerPiece>
    void staged_pieces(const Scan &sc, Gathered &g, const Packed &p, bsq_stage *stage, int64_t head, int64_t seqs, py::ssize_t padlen, int nthreads,
                       bool onehot, PerPiece per_piece) const {
        const int64_t maxlen = int64_t(padlen) - desc.bos - desc.eos;
        piece_loop(sc.n, stage, head, seqs,
                   [&](int64_t lo, int64_t hi) {
                       int64_t bad = -1;
                       if (!scan_pack_fast(sc, g, p, lo, hi, nthreads, maxlen, &bad)) {  // mask list / other item types: the general passes
                           scan_range(sc, g, lo, hi, nthreads);
                           for (int64_t i = lo; i < hi; ++i) {
                               p.offsets[i + 1] = p.offsets[i] + int64_t(g.items[size_t(i)].len);
                               if (int64_t(g.items[size_t(i)].len) > maxlen) throw_too_long(p.offsets, i, padlen, onehot);
                           }
                           if (size_t(p.offsets[hi]) > p.cap) throw StageOverflow{};
                           pack_range(g, p, lo, hi, nthreads);
                       }
                       if (bad >= 0) throw_too_long(p.offsets, bad, padlen, onehot);
                   },
                   per_piece);
    }

    struct StageEnd {
        bsq_stage *s;
        ~StageEnd() { (void)bsq_stage_end(s); }
    };

    template <typename BlockFn>
    bool staged(const Scan &sc, Gathered &g, py::ssize_t padlen, int nthreads, const OutBuf &out, bool splittable, size_t block_row_bytes,
                bool onehot, BlockFn block) const {
        const int64_t maxlen = int64_t(padlen) - desc.bos - desc.eos;  // a longer item is an error anyway
        if (out.space != BSQ_SPACE_DEVICE || sc.n < 16384 || maxlen <= 0) return false;
        size_t likely = 0;
        const size_t max_chars = estimate_chars(sc, maxlen, &likely);  // (an estimate: see StageOverflow)
        if (max_chars > (size_t(1) << 30)) return false;
        int64_t head = 0;  // sequences in front of the first piece boundary (column blocks of a result that is not 4-KiB aligned)
        int64_t seqs = splittable ? bsq_stage_piece_hint(sc.n, likely, block_row_bytes, out.ptr, out.stream, &head) : 0;
        if (seqs < 0) return false;                  // knob host_pieces = 1: the whole-batch path of rounds 1-3
        if (seqs == 0 || seqs > sc.n) seqs = sc.n;   // one piece (busy stream, misaligned result, ...): still one scan + pack job
        bsq_stage *stage = nullptr;
        Packed p;
        p.B = sc.n;
        p.cap = max_chars;
        const bsq_status st0 = bsq_stage_begin(sc.n, max_chars, g.has_mask ? 1 : 0, out.stream, &stage, &p.offsets, &p.chars, &p.mask);
        if (st0 != BSQ_OK) throw_status(st0);
        StageEnd end{stage};
        try {
            staged_pieces(sc, g, p, stage, head, seqs, padlen, nthreads, onehot,
                          [&](int64_t lo, int64_t hi, const uint8_t *d_chars, const int64_t *d_offsets, const uint8_t *d_mask) {
                              const int64_t lead = lo == 0 && head < hi ? head : 0;
                              bsq_status st = BSQ_OK;
                              if (lead) st = block(d_chars, d_offsets, d_mask, 0, lead);
                              if (st == BSQ_OK) st = block(d_chars, d_offsets + lead, d_mask, lo + lead, hi - lo - lead);
                              return st;
                          });
        } catch (const StageOverflow &) {
            // more characters than the sampled estimate allowed for: the blocks encoded so far are valid and will simply be written
            // again, in stream order, by the whole-batch path the caller goes on with (it scans everything before it allocates)
            reset_scan(g);
            return false;
        }
        return true;
    }

    static void reset_scan(Gathered &g) {  // forget a partial scan (the pieces in front of a StageOverflow); `keep` only grows
        g.total = 0;
    }

    // The same for a NUMPY result (the reference's default return) of up to 256 MB -- token matrices, small one-hots: every piece
    // is encoded as a matrix of its own (`piece(d_chars, d_offsets, d_mask, n, dst)`: n sequences, per_seq_bytes each) in the staging
    // area's device scratch and fetched into its pinned mirror while the next piece is packed and uploaded (PCIe runs both ways);
    // at the end the pool copies the pinned pieces into the array -- straight (rows == 0: a piece is a contiguous slab of the result:
    // (B, P), (B, C, P)) or row by row (a piece is `rows` x n x col_bytes of a (rows, B, col_bytes) result: (P, B), (P, B, C)).
    // Host results of a staged batch: `loop(per_piece)` runs the piece loop; every piece is encoded by `piece(...)` as a matrix of its
    // own (n sequences of per_seq_bytes) in the staging area's device scratch and fetched into its pinned mirror; the pieces land in
    // `dst` as their fetches complete -- straight (rows == 0) or row by row (`rows` x n x col_bytes pieces of a (rows, B, col_bytes) result).
    template <typename Loop, typename PieceFn>
    static void fetch_and_land(bsq_stage *stage, int64_t B, size_t per_seq_bytes, size_t rows, size_t col_bytes, int nthreads, char *dst, Loop loop,
                               PieceFn piece, bool gil_held) {
        void *d_res = nullptr, *h_res = nullptr;
        bsq_status st = bsq_stage_result(stage, size_t(B) * per_seq_bytes, &d_res, &h_res);
        if (st != BSQ_OK) throw_status(st);
        static const bool prof = std::getenv("BSQ_PROFILE_HOST") != nullptr;
        const auto t0 = std::chrono::steady_clock::now();
        const char *src = static_cast<const char *>(h_res);
        const int nt = std::max(1, std::min(nthreads, 64));
        auto land = [&](int64_t lo, int64_t hi) {
            const size_t off = size_t(lo) * per_seq_bytes, len = size_t(hi - lo) * per_seq_bytes, n = size_t(hi - lo);
            if (rows == 0) {
                pool().parallel_for(nt, [&](int t) {
                    const size_t a = len * size_t(t) / size_t(nt) / 4096 * 4096, b = t + 1 == nt ? len : len * size_t(t + 1) / size_t(nt) / 4096 * 4096;
                    if (b > a) std::memcpy(dst + off + a, src + off + a, b - a);
                });
            } else {
                pool().parallel_for(nt, [&](int t) {
                    for (size_t r = rows * size_t(t) / size_t(nt), r1 = rows * size_t(t + 1) / size_t(nt); r < r1; ++r)
                        std::memcpy(dst + (r * size_t(B) + size_t(lo)) * col_bytes, src + off + r * n * col_bytes, n * col_bytes);
                });
            }
        };
        struct Pending {
            int64_t lo, hi;
            int32_t ticket;
        };
        std::vector<Pending> pending;
        loop([&](int64_t lo, int64_t hi, const uint8_t *d_chars, const int64_t *d_offsets, const uint8_t *d_mask) {
            bsq_status s1 = piece(d_chars, d_offsets, d_mask, hi - lo, static_cast<char *>(d_res) + size_t(lo) * per_seq_bytes);
            int32_t ticket = -1;
            if (s1 == BSQ_OK) s1 = bsq_stage_fetch(stage, size_t(lo) * per_seq_bytes, size_t(hi - lo) * per_seq_bytes, &ticket);
            pending.push_back(Pending{lo, hi, ticket});
            return s1;
        });
        const auto t1 = std::chrono::steady_clock::now();
        // the pieces land in the array as their fetches complete: the copy (and the page faults of a fresh array) of piece j runs
        // while pieces j + 1 ... are still on the bus.  No Python object is touched: other Python threads may run meanwhile.
        auto land_all = [&] {
            for (const Pending &q : pending) {
                st = bsq_stage_wait(stage, q.ticket);
                if (st != BSQ_OK) return;
                land(q.lo, q.hi);
            }
        };
        if (gil_held) {
            py::gil_scoped_release nogil;
            land_all();
        } else {
            land_all();
        }
        if (st != BSQ_OK) throw_status(st);
        if (prof) {
            auto us = [](auto x, auto y) { return (long)std::chrono::duration_cast<std::chrono::microseconds>(y - x).count(); };
            std::fprintf(stderr, "[bsq host] host result in pieces: produce + upload + launch + fetch %ld us, waits + copies into the result %ld us\n",
                         us(t0, t1), us(t1, std::chrono::steady_clock::now()));
        }
    }

    // A NUMPY result (the reference's default return) of up to 256 MB -- token matrices, small one-hots -- from a list: see fetch_and_land.
    template <typename PieceFn>
    bool staged_host(const Scan &sc, Gathered &g, py::ssize_t padlen, int nthreads, const OutBuf &out, size_t per_seq_bytes, size_t rows,
                     size_t col_bytes, bool onehot, PieceFn piece) const {
        const int64_t maxlen = int64_t(padlen) - desc.bos - desc.eos;
        const size_t total = size_t(sc.n) * per_seq_bytes;
        if (out.space != BSQ_SPACE_HOST || sc.n < 16384 || maxlen <= 0 || total > (size_t(256) << 20)) return false;
        size_t likely = 0;
        const size_t max_chars = estimate_chars(sc, maxlen, &likely);
        if (max_chars > (size_t(1) << 30)) return false;
        int64_t head = 0;
        int64_t seqs = bsq_stage_piece_hint(sc.n, likely, 0, nullptr, nullptr, &head);
        if (seqs < 0) return false;
        if (seqs == 0 || seqs > sc.n) seqs = sc.n;
        bsq_stage *stage = nullptr;
        Packed p;
        p.B = sc.n;
        p.cap = max_chars;
        const bsq_status st = bsq_stage_begin(sc.n, max_chars, g.has_mask ? 1 : 0, nullptr, &stage, &p.offsets, &p.chars, &p.mask);
        if (st != BSQ_OK) throw_status(st);
        StageEnd end{stage};
        try {
            fetch_and_land(stage, sc.n, per_seq_bytes, rows, col_bytes, nthreads, static_cast<char *>(out.ptr),
                           [&](auto per_piece) { staged_pieces(sc, g, p, stage, 0, seqs, padlen, nthreads, onehot, per_piece); }, piece, true);
        } catch (const StageOverflow &) {  // (nothing has landed in the array yet: pieces land after the loop)
            reset_scan(g);
            return false;
        }
        return true;
    }

    // The same for a PACKED batch in host memory (numpy arrays, a memory-mapped FlatFile): the pieces are copied into the pinned
    // staging area (no Python object involved: the GIL is released by the caller).  Lengths have been validated.
    template <typename PieceFn>
    bool staged_host_packed(const uint8_t *chars, const int64_t *offs, const uint8_t *mask, int64_t B, int nthreads, char *dst, size_t per_seq_bytes,
                            size_t rows, size_t col_bytes, PieceFn piece) const {
        const size_t total = size_t(B) * per_seq_bytes, nchars = size_t(offs[B] - offs[0]);
        if (B < 16384 || total > (size_t(256) << 20) || nchars > (size_t(1) << 30)) return false;
        int64_t head = 0;
        int64_t seqs = bsq_stage_piece_hint(B, nchars, 0, nullptr, nullptr, &head);
        if (seqs < 0) return false;
        if (seqs == 0 || seqs > B) seqs = B;
        bsq_stage *stage = nullptr;
        Packed p;
        p.B = B;
        const bsq_status st = bsq_stage_begin(B, nchars, mask ? 1 : 0, nullptr, &stage, &p.offsets, &p.chars, &p.mask);
        if (st != BSQ_OK) throw_status(st);
        StageEnd end{stage};
        const int nt = std::max(1, std::min(nthreads, 64));
        const int64_t base = offs[0];
        auto produce = [&](int64_t lo, int64_t hi) {
            for (int64_t i = lo; i < hi; ++i) p.offsets[i + 1] = offs[i + 1] - base;
            const size_t c0 = size_t(offs[lo] - base), len = size_t(offs[hi] - offs[lo]);
            pool().parallel_for(len >= (size_t(1) << 20) ? nt : 1, [&](int t) {
                const int k = len >= (size_t(1) << 20) ? nt : 1;
                const size_t a = len * size_t(t) / size_t(k), b = len * size_t(t + 1) / size_t(k);
                if (b > a) {
                    std::memcpy(p.chars + c0 + a, chars + size_t(offs[lo]) + a, b - a);
                    if (mask) std::memcpy(p.mask + c0 + a, mask + size_t(offs[lo]) + a, b - a);
                }
            });
        };
        fetch_and_land(stage, B, per_seq_bytes, rows, col_bytes, nthreads, dst,
                       [&](auto per_piece) { piece_loop(B, stage, 0, seqs, produce, per_piece); }, piece, false);
        return true;
    }

    // batch_tokenize (tokenize.cpp:82-98 -> tokenize.h:381-485)
    py::object batch_tokenize(py::sequence batch, py::ssize_t padlen, const std::string &dt, bool batch_first,
                              int nthreads, const py::object &device) const {
        const bsq_dtype t = parse_dtype(dt);
        check_padlen(padlen);
        Gathered g;
        const Scan sc = scan_begin(batch, py::none(), g, nthreads);  // (resolves nthreads = 0 to the automatic count)
        // numpy results: small batches scan first (item errors before the result is allocated, as ever); large ones that fit the
        // staged path (<= 256 MB of result) are scanned piece by piece in staged_host
        const size_t row_bytes = size_t(padlen) * bsq_dtype_size(t);
        const bool host_stage = device.is_none() && sc.n >= 16384 && size_t(sc.n) * row_bytes <= (size_t(256) << 20);
        bool scanned = false;
        if (device.is_none() && !host_stage) {
            scan_range(sc, g, 0, sc.n, nthreads);
            scanned = true;
        }
        PackLock lock;
        OutBuf out;  // first: it makes `device=` the current device, so the pinned scratch and the staging
                     // buffers used by pack() and by the encode call belong to the same device
        const py::ssize_t nb = py::ssize_t(sc.n);
        make_out(out, batch_first ? std::vector<py::ssize_t>{nb, padlen} : std::vector<py::ssize_t>{padlen, nb}, t, device);
        const size_t row = size_t(padlen) * bsq_dtype_size(t);
        // batch-first: a piece is rows [lo, lo + n) of (B, P); seq-first: a column block of (P, B) -- split only for the types that
        // run through k_tokens_pb8_fast as a block (1, 2, 8 bytes, ids < 251), one piece otherwise
        const size_t tsz = bsq_dtype_size(t);
        const bool sf_blocks = (tsz == 1 || tsz == 2 || (tsz == 8 && nb % 2 == 0)) && bsq_alphabet_size(&desc) <= 250;  // (8-byte: whole 16-byte lines)
        if (staged(sc, g, padlen, nthreads, out, batch_first || sf_blocks, 0, false,
                   [&](const uint8_t *chars, const int64_t *offsets, const uint8_t *, int64_t lo, int64_t n) {
                       if (batch_first)
                           return bsq_tokenize_device(&desc, chars, offsets, n, padlen, 1, t, static_cast<char *>(out.ptr) + size_t(lo) * row, out.stream);
                       return bsq_tokenize_block_device(&desc, chars, offsets, n, padlen, t, static_cast<char *>(out.ptr) + size_t(lo) * tsz, int64_t(nb),
                                                        out.stream);
                   }))
            return out.obj;
        // numpy result of up to 256 MB: the pieces are encoded on the device and fetched while the next one is packed and uploaded
        if (host_stage && staged_host(sc, g, padlen, nthreads, out, row, batch_first ? 0 : size_t(padlen), tsz, false,
                                            [&](const uint8_t *chars, const int64_t *offsets, const uint8_t *, int64_t n, void *dst) {
                                                return bsq_tokenize_device(&desc, chars, offsets, n, padlen, batch_first, t, dst, nullptr);
                                            }))
            return out.obj;
        if (!scanned) scan_range(sc, g, 0, sc.n, nthreads);
        const Packed p = pack(g, nthreads);
        int64_t bad = -1;
        bsq_status st;
        {
            py::gil_scoped_release nogil;
            st = bsq_tokenize_host(&desc, p.chars, p.offsets, p.B, padlen, batch_first, t, out.ptr, out.space,
                                   out.stream, &bad);
        }
        if (st == BSQ_ERR_SEQ_TOO_LONG) throw_too_long(p.offsets, bad, padlen, false);
        if (st != BSQ_OK) throw_status(st);
        return out.obj;
    }

    // batch_onehot_encode (tokenize.cpp:65-81 -> tokenize.h:283-371); always (P, B, C)
    static bool parse_layout(const std::string &layout) {  // true: channels-first (B, C, P)
        if (layout == "tbc" || layout == "seq_first" || layout.empty()) return false;
        if (layout == "bcl" || layout == "channels_first") return true;
        throw std::invalid_argument("layout must be 'tbc' (padlen, batch, channels) or 'bcl' (batch, channels, padlen)");
    }

    py::object batch_onehot_encode(py::sequence batch, py::ssize_t padlen, const std::string &dt, int nthreads,
                                   const py::object &mask, const py::object &device, const std::string &layout) const {
        const bool bcl = parse_layout(layout);
        const bsq_dtype t = parse_dtype(dt);
        check_padlen(padlen);
        static const bool prof = std::getenv("BSQ_PROFILE_HOST") != nullptr;  // per-phase host times on stderr
        const auto t0 = std::chrono::steady_clock::now();
        Gathered g;
        const Scan sc = scan_begin(batch, mask, g, nthreads);
        const size_t per_seq = size_t(padlen) * size_t(bsq_alphabet_size(&desc)) * bsq_dtype_size(t);
        const bool host_stage = device.is_none() && sc.n >= 16384 && size_t(sc.n) * per_seq <= (size_t(256) << 20);  // (see batch_tokenize)
        bool scanned = false;
        if (device.is_none() && !host_stage) {
            scan_range(sc, g, 0, sc.n, nthreads);
            scanned = true;
        }
        const auto t1 = std::chrono::steady_clock::now();
        PackLock lock;
        OutBuf out;  // before pack(): see batch_tokenize
        const py::ssize_t C = py::ssize_t(bsq_alphabet_size(&desc)), nb = py::ssize_t(sc.n);
        make_out(out, bcl ? std::vector<py::ssize_t>{nb, C, padlen} : std::vector<py::ssize_t>{padlen, nb, C}, t, device);
        const auto t2 = std::chrono::steady_clock::now();
        auto us = [](auto a, auto b) { return (long)std::chrono::duration_cast<std::chrono::microseconds>(b - a).count(); };
        // seq-first: piece = a column block of (P, B, C); channels-first: piece = rows [lo, lo + n) of (B, C, P)
        const size_t row = size_t(C) * bsq_dtype_size(t);
        const bool done =
            bcl ? staged(sc, g, padlen, nthreads, out, true, 0, true,
                         [&](const uint8_t *chars, const int64_t *offsets, const uint8_t *m, int64_t lo, int64_t n) {
                             return bsq_onehot_bcl_device(&desc, chars, offsets, m, n, padlen, t,
                                                          static_cast<char *>(out.ptr) + size_t(lo) * row * size_t(padlen), out.stream);
                         })
                : staged(sc, g, padlen, nthreads, out, true, row, true,
                         [&](const uint8_t *chars, const int64_t *offsets, const uint8_t *m, int64_t lo, int64_t n) {
                             return bsq_onehot_block_device(&desc, chars, offsets, m, n, padlen, t, static_cast<char *>(out.ptr) + size_t(lo) * row,
                                                            int64_t(nb), out.stream);
                         });
        if (done) {
            if (prof) std::fprintf(stderr, "[bsq host] list setup %ld us, output alloc %ld us (address %% 4096 = %lu), scan + pack + upload + launch in pieces %ld us\n",
                                   us(t0, t1), us(t1, t2), (unsigned long)(reinterpret_cast<uintptr_t>(out.ptr) % 4096), us(t2, std::chrono::steady_clock::now()));
            return out.obj;
        }
        if (host_stage && staged_host(sc, g, padlen, nthreads, out, per_seq, bcl ? 0 : size_t(padlen), row, true,
                                      [&](const uint8_t *chars, const int64_t *offsets, const uint8_t *m, int64_t n, void *dst) {
                                          return bcl ? bsq_onehot_bcl_device(&desc, chars, offsets, m, n, padlen, t, dst, nullptr)
                                                     : bsq_onehot_device(&desc, chars, offsets, m, n, padlen, t, dst, nullptr);
                                      }))
            return out.obj;
        if (!scanned) scan_range(sc, g, 0, sc.n, nthreads);
        const Packed p = pack(g, nthreads);
        const auto t3 = std::chrono::steady_clock::now();
        int64_t bad = -1;
        bsq_status st;
        {
            py::gil_scoped_release nogil;
            st = (bcl ? bsq_onehot_bcl_host : bsq_onehot_host)(&desc, p.chars, p.offsets, p.mask, p.B, padlen, t, out.ptr,
                                                               out.space, out.stream, &bad);
        }
        if (prof)
            std::fprintf(stderr, "[bsq host] gather %ld us, output alloc %ld us, scan (device results) + pack %ld us, upload+launch %ld us\n", us(t0, t1),
                         us(t1, t2), us(t2, t3), us(t3, std::chrono::steady_clock::now()));
        if (st == BSQ_ERR_SEQ_TOO_LONG) throw_too_long(p.offsets, bad, padlen, true);
        if (st != BSQ_OK) throw_status(st);
        return out.obj;
    }

    // Packed-batch entry points (additive): chars uint8[total] + offsets int64[B+1] (+ mask uint8[total]),
    // as numpy arrays (staged through the library) or torch tensors already on the device (zero copy --
    // the layout of the reference's FlatFile, fxstats.cpp:33-64).
    py::object encode_packed(bool onehot, const py::object &chars_o, const py::object &offsets_o,
                             py::ssize_t padlen, const std::string &dt, bool batch_first, const py::object &mask_o,
                             const py::object &device_o, bool validate, bool bcl = false) const {
        const bsq_dtype t = parse_dtype(dt);
        check_padlen(padlen);
        ArrayArg chars = as_array(chars_o, "uint8", 1, "chars");
        ArrayArg offsets = as_array(offsets_o, "int64", 8, "offsets");
        ArrayArg mask;
        const bool has_mask = onehot && !mask_o.is_none();
        if (has_mask) mask = as_array(mask_o, "uint8", 1, "mask");
        if (offsets.n < 1) throw std::invalid_argument("offsets must have B + 1 entries");
        const int64_t B = offsets.n - 1;
        if (chars.on_device != offsets.on_device || (has_mask && mask.on_device != chars.on_device))
            throw std::invalid_argument("chars, offsets and mask must live in the same memory space");
        if (has_mask && mask.n < chars.n) throw std::invalid_argument("mask must have one byte per character");
        py::object device = device_o;
        if (chars.on_device && device.is_none()) device = chars.keep.attr("device");
        if (chars.on_device && !device.is_none()) {
            py::module_ torch = py::module_::import("torch");
            if (!torch.attr("device")(device).equal(chars.keep.attr("device")))
                throw std::invalid_argument("device= differs from the device of the packed batch");
        }
        const py::ssize_t C = bsq_alphabet_size(&desc);
        std::vector<py::ssize_t> shape = onehot ? (bcl ? std::vector<py::ssize_t>{B, C, padlen} : std::vector<py::ssize_t>{padlen, B, C})
                                                : (batch_first ? std::vector<py::ssize_t>{B, padlen}
                                                               : std::vector<py::ssize_t>{padlen, B});
        PackLock lock;
        OutBuf out;
        make_out(out, shape, t, device);
        int64_t bad = -1;
        bsq_status st;
        std::vector<int64_t> bad_pair(2, 0);
        if (chars.on_device) {
            const int64_t *offs = static_cast<const int64_t *>(offsets.ptr);
            {
                py::gil_scoped_release nogil;
                st = BSQ_OK;
                // validate=True also checks the offsets themselves (non-negative, non-decreasing, inside chars): the
                // kernels bound their reads by offsets[B], not by the real size of the buffer.  validate=False is an
                // unchecked contract: the caller vouches for well-formed offsets and lengths <= padlen - bos - eos.
                if (validate) st = bsq_validate_packed_device(offs, B, padlen, desc.bos, desc.eos, chars.n, &bad, out.stream);
                if (st == BSQ_OK)
                    st = onehot ? (bcl ? bsq_onehot_bcl_device : bsq_onehot_device)(
                                      &desc, static_cast<const uint8_t *>(chars.ptr), offs,
                                      has_mask ? static_cast<const uint8_t *>(mask.ptr) : nullptr, B, padlen, t, out.ptr,
                                      out.stream)
                                : bsq_tokenize_device(&desc, static_cast<const uint8_t *>(chars.ptr), offs, B, padlen,
                                                      batch_first, t, out.ptr, out.stream);
            }
            if (st == BSQ_ERR_INVALID_ARG && bad >= 0)
                throw std::invalid_argument("offsets must be non-negative, non-decreasing and end inside chars (first bad entry: " +
                                            std::to_string(bad) + ")");
            if (st == BSQ_ERR_SEQ_TOO_LONG) {
                py::object pair = offsets.keep.attr("__getitem__")(py::slice(bad, bad + 2, 1)).attr("tolist")();
                bad_pair = pair.cast<std::vector<int64_t>>();
                throw_too_long(bad_pair.data(), 0, padlen, onehot);
            }
        } else {
            const int64_t *offs = static_cast<const int64_t *>(offsets.ptr);
            if (B > 0 && (offs[0] < 0 || offs[B] > chars.n)) throw std::invalid_argument("offsets exceed chars");
            for (int64_t i = 0; i < B; ++i)
                if (offs[i + 1] < offs[i]) throw std::invalid_argument("offsets must be non-decreasing");
            {
                py::gil_scoped_release nogil;
                const uint8_t *cp = static_cast<const uint8_t *>(chars.ptr), *mp = has_mask ? static_cast<const uint8_t *>(mask.ptr) : nullptr;
                st = bsq_validate_lengths(offs, B, padlen, desc.bos, desc.eos, &bad);
                bool done = false;
                // a numpy result of up to 256 MB: pieces go up, are encoded and fetched back while the next ones go up (staged_host_packed)
                if (st == BSQ_OK && out.space == BSQ_SPACE_HOST && B >= 16384) {
                    const size_t sz = bsq_dtype_size(t), per_seq = size_t(padlen) * sz * (onehot ? size_t(C) : 1);
                    const bool columns = onehot ? !bcl : !batch_first;
                    done = staged_host_packed(cp, offs, mp, B, resolve_threads(0, B), static_cast<char *>(out.ptr), per_seq, columns ? size_t(padlen) : 0,
                                              onehot ? size_t(C) * sz : sz,
                                              [&](const uint8_t *dc, const int64_t *dof, const uint8_t *dm, int64_t n, void *dst) {
                                                  if (!onehot) return bsq_tokenize_device(&desc, dc, dof, n, padlen, batch_first, t, dst, nullptr);
                                                  return bcl ? bsq_onehot_bcl_device(&desc, dc, dof, dm, n, padlen, t, dst, nullptr)
                                                             : bsq_onehot_device(&desc, dc, dof, dm, n, padlen, t, dst, nullptr);
                                              });
                }
                if (st == BSQ_OK && !done)
                    st = onehot ? (bcl ? bsq_onehot_bcl_host : bsq_onehot_host)(&desc, cp, offs, mp, B, padlen, t, out.ptr, out.space, out.stream, &bad)
                                : bsq_tokenize_host(&desc, cp, offs, B, padlen, batch_first, t, out.ptr, out.space, out.stream, &bad);
            }
            if (st == BSQ_ERR_SEQ_TOO_LONG) throw_too_long(offs, bad, padlen, onehot);
        }
        if (st != BSQ_OK) throw_status(st);
        return out.obj;
    }

    // Single-sequence one-hot (tokenize.cpp:8-48 -> tokenize.h:188-216).  NOT part of the batch hot
    // path: a few hundred bytes of host work, done here like decode_tokens.
    // Shape (max(L, padlen) + bos + eos, C); PAD rows only up to padlen; the dtype character is
    // case-masked (B -> uint8, H -> uint16, I -> uint32, F -> float32, D -> float64).  An unmapped
    // byte leaves its row all-zero (the reference writes one element before the row: tokenize.h:203-206).
    template <typename T>
    py::array onehot_single_t(const char *s, py::ssize_t L, py::ssize_t padlen) const {
        if (padlen > 0 && L > padlen) throw std::runtime_error("padlen is too short to accommodate sequence\n");
        const py::ssize_t C = bsq_alphabet_size(&desc);
        const py::ssize_t rows = std::max(L, padlen) + desc.bos + desc.eos;
        py::array_t<T> ret(std::vector<py::ssize_t>{rows, C});
        T *ptr = ret.mutable_data();
        std::fill(ptr, ptr + rows * C, T(0));
        py::ssize_t r = 0;
        if (desc.bos) ptr[bsq_bos_id(&desc)] = T(1), ++r;
        for (py::ssize_t i = 0; i < L; ++i, ++r) {
            const unsigned char c = static_cast<unsigned char>(s[i]);
            const int tr = c < 128 ? desc.lut[c] : -1;
            if (tr >= 0) ptr[r * C + tr] = T(1);
        }
        if (desc.eos) ptr[r * C + bsq_eos_id(&desc)] = T(1), ++r;
        if (desc.padchar)
            for (; r < padlen; ++r) ptr[r * C + bsq_pad_id(&desc)] = T(1);
        return ret;
    }
    // Single-sequence one-hot ON THE DEVICE (device=...): the batch kernel with B = 1 and P = rows writes BOS, the
    // residues, EOS and PAD rows; the reference pads only the rows below `padlen` (tokenize.h:208-212), so the rows
    // from max(padlen, L + bos + eos) on are zeroed again.  Returns a torch tensor (rows, C) of uint8 / int16 / int32 /
    // float32 / float64 (the 16- and 32-bit unsigned types of the host path have no usable torch dtype: same bits).
    py::object onehot_single_device(const char *s, py::ssize_t L, py::ssize_t padlen, const std::string &dt,
                                    const py::object &device) const {
        if (padlen > 0 && L > padlen) throw std::runtime_error("padlen is too short to accommodate sequence\n");
        bsq_dtype t;
        const char *tname;
        switch (dt.empty() ? 0 : (dt[0] & 223)) {
        case 'B': t = BSQ_I8, tname = "uint8"; break;
        case 'H': t = BSQ_I16, tname = "int16"; break;
        case 'I': t = BSQ_I32, tname = "int32"; break;
        case 'F': t = BSQ_F32, tname = "float32"; break;
        case 'D': t = BSQ_F64, tname = "float64"; break;
        default: throw std::invalid_argument(std::string("Unsupported dtype: ") + dt);
        }
        py::module_ torch = py::module_::import("torch");
        py::object dev = torch.attr("device")(device);
        if (dev.attr("type").cast<std::string>() != "cuda")
            throw std::invalid_argument("device= must be a HIP ('cuda') device; omit it for a numpy result");
        const py::ssize_t C = bsq_alphabet_size(&desc);
        const py::ssize_t rows = std::max(L, padlen) + desc.bos + desc.eos;
        py::object out = torch.attr("empty")(py::make_tuple(rows, C), py::arg("dtype") = torch.attr(tname), py::arg("device") = dev);
        if (rows == 0) return out;
        // packed batch of one sequence: offsets | chars in one small host tensor -> device
        py::array_t<uint8_t> host(py::ssize_t(16 + L));
        int64_t offs[2] = {0, int64_t(L)};
        std::memcpy(host.mutable_data(), offs, 16);
        if (L) std::memcpy(host.mutable_data() + 16, s, size_t(L));
        py::object packed = torch.attr("from_numpy")(host).attr("to")(dev);
        const uintptr_t base = packed.attr("data_ptr")().cast<uintptr_t>();
        py::object guard = torch.attr("cuda").attr("device")(dev);
        guard.attr("__enter__")();
        void *stream = reinterpret_cast<void *>(torch.attr("cuda").attr("current_stream")().attr("cuda_stream").cast<uintptr_t>());
        const bsq_status st = bsq_onehot_device(&desc, reinterpret_cast<const uint8_t *>(base + 16), reinterpret_cast<const int64_t *>(base),
                                                nullptr, 1, rows, t, reinterpret_cast<void *>(out.attr("data_ptr")().cast<uintptr_t>()), stream);
        guard.attr("__exit__")(py::none(), py::none(), py::none());
        if (st != BSQ_OK) throw_status(st);
        const py::ssize_t first_zero = std::max(padlen, L + desc.bos + desc.eos);
        if (desc.padchar && first_zero < rows) out.attr("__getitem__")(py::slice(first_zero, rows, 1)).attr("zero_")();
        return out;
    }
    py::object onehot_single(const char *s, py::ssize_t L, py::ssize_t padlen, const std::string &dt,
                             const py::object &device = py::none()) const {
        if (!device.is_none()) return onehot_single_device(s, L, padlen, dt, device);
        switch (dt.empty() ? 0 : (dt[0] & 223)) {
        case 'B': return onehot_single_t<uint8_t>(s, L, padlen);
        case 'H': return onehot_single_t<uint16_t>(s, L, padlen);
        case 'I': return onehot_single_t<uint32_t>(s, L, padlen);
        case 'F': return onehot_single_t<float>(s, L, padlen);
        case 'D': return onehot_single_t<double>(s, L, padlen);
        default: throw std::invalid_argument(std::string("Unsupported dtype: ") + dt);
        }
    }

    // decode_tokens (tokenize.h:131-183): 1-D -> str, 2-D -> list of str.
    // Token matrices that live on a HIP device are decoded THERE (bsq_decode.hip: per-row wave prefix sums of the
    // piece widths; only the decoded text is copied back), host arrays here.
    py::object decode_device(py::object t) const {
        py::module_ torch = py::module_::import("torch");
        t = t.attr("detach")();
        const int64_t ndim = t.attr("dim")().cast<int64_t>();
        if (ndim > 2 || ndim == 0)
            throw std::invalid_argument("Currently supported: 1 or 2 dimensions for decoding tokens.");
        if (t.attr("dtype").equal(torch.attr("bool"))) t = t.attr("to")(torch.attr("uint8"));
        const int32_t itemsize = t.attr("element_size")().cast<int32_t>();
        if (itemsize != 1 && itemsize != 2 && itemsize != 4 && itemsize != 8)
            throw std::runtime_error("Unexpected itemsize: expected 1, 2, 4, or 8. Found " + std::to_string(itemsize));
        const std::vector<int64_t> shape = t.attr("shape").cast<std::vector<int64_t>>();
        const std::vector<int64_t> stride = t.attr("stride")().cast<std::vector<int64_t>>();
        const int64_t nrows = ndim == 2 ? shape[0] : 1, ncols = ndim == 2 ? shape[1] : shape[0];
        const int64_t rs = ndim == 2 ? stride[0] * itemsize : 0, cs = (ndim == 2 ? stride[1] : stride[0]) * itemsize;
        if (nrows == 0 || ncols == 0) {  // nothing to decode (an empty tensor has no storage to point at)
            if (ndim == 1) return py::str("");
            py::list empty;
            for (int64_t r = 0; r < nrows; ++r) empty.append(py::str(""));
            return empty;
        }
        py::object dev = t.attr("device");
        py::object guard = torch.attr("cuda").attr("device")(dev);
        guard.attr("__enter__")();
        struct Exit {
            py::object g;
            ~Exit() {
                try {
                    g.attr("__exit__")(py::none(), py::none(), py::none());
                } catch (...) {
                }
            }
        } exit_guard{guard};
        void *stream = reinterpret_cast<void *>(torch.attr("cuda").attr("current_stream")().attr("cuda_stream").cast<uintptr_t>());
        const void *tok = reinterpret_cast<const void *>(t.attr("data_ptr")().cast<uintptr_t>());
        py::object offs = torch.attr("empty")(py::make_tuple(nrows + 1), py::arg("dtype") = torch.attr("int64"), py::arg("device") = dev);
        int64_t *offs_p = reinterpret_cast<int64_t *>(offs.attr("data_ptr")().cast<uintptr_t>());
        int64_t total = 0, bad = -1;
        bsq_status st;
        {
            py::gil_scoped_release nogil;
            st = bsq_decode_sizes_device(&desc, tok, itemsize, nrows, ncols, rs, cs, offs_p, &total, &bad, stream);
        }
        if (st == BSQ_ERR_INVALID_ARG && bad >= 0) {  // the reference's message carries the offending value (as uint32)
            const int64_t r = ncols ? bad / ncols : 0, c = ncols ? bad % ncols : 0;
            py::object item = (ndim == 2 ? t.attr("__getitem__")(py::make_tuple(r, c)) : t.attr("__getitem__")(c)).attr("item")();
            const uint64_t v = uint64_t(item.cast<int64_t>()) & (itemsize == 1 ? 0xFFull : itemsize == 2 ? 0xFFFFull : 0xFFFFFFFFull);
            throw std::runtime_error("Unexpected/invalid token " + std::to_string(v));
        }
        if (st != BSQ_OK) throw_status(st);
        py::object out = torch.attr("empty")(py::make_tuple(total), py::arg("dtype") = torch.attr("uint8"), py::arg("device") = dev);
        {
            uint8_t *out_p = reinterpret_cast<uint8_t *>(out.attr("data_ptr")().cast<uintptr_t>());
            py::gil_scoped_release nogil;
            st = bsq_decode_write_device(&desc, tok, itemsize, nrows, ncols, rs, cs, offs_p, out_p, stream);
        }
        if (st != BSQ_OK) throw_status(st);
        py::array_t<uint8_t> text = out.attr("cpu")().attr("numpy")().cast<py::array_t<uint8_t>>();
        py::array_t<int64_t> ho = offs.attr("cpu")().attr("numpy")().cast<py::array_t<int64_t>>();
        const char *tp = reinterpret_cast<const char *>(text.data());
        const int64_t *op = ho.data();
        if (ndim == 1) return py::str(tp + op[0], size_t(op[1] - op[0]));
        py::list ret;
        for (int64_t r = 0; r < nrows; ++r) ret.append(py::str(tp + op[r], size_t(op[r + 1] - op[r])));
        return ret;
    }

    // decode_logits (addition; README.md:48): argmax over the last (channel) axis on the device, then decode_device.
    // logits: float tensor on a HIP device, (L, C) -> str, (B, L, C) -> list of B str (batch_first) or (L, B, C).
    py::object decode_logits(py::object logits, bool batch_first) const {
        py::module_ torch = py::module_::import("torch");
        if (!py::hasattr(logits, "is_cuda") || !logits.attr("is_cuda").cast<bool>())
            throw std::invalid_argument("decode_logits expects a float tensor on a HIP ('cuda') device");
        logits = logits.attr("detach")();
        const int64_t ndim = logits.attr("dim")().cast<int64_t>();
        if (ndim != 2 && ndim != 3) throw std::invalid_argument("logits must be (L, C) or (B, L, C) / (L, B, C)");
        if (logits.attr("stride")(-1).cast<int64_t>() != 1 || !logits.attr("is_contiguous")().cast<bool>())
            logits = logits.attr("contiguous")();
        const std::string dt = py::str(logits.attr("dtype")).cast<std::string>();
        int32_t kind;
        if (dt == "torch.float32") kind = BSQ_LOGITS_F32;
        else if (dt == "torch.float64") kind = BSQ_LOGITS_F64;
        else if (dt == "torch.float16") kind = BSQ_LOGITS_F16;
        else if (dt == "torch.bfloat16") kind = BSQ_LOGITS_BF16;
        else throw std::invalid_argument("logits must be float32, float64, float16 or bfloat16, not " + dt);
        const std::vector<int64_t> shape = logits.attr("shape").cast<std::vector<int64_t>>();
        const int64_t C = shape.back();
        if (C <= 0 || C > (int64_t(1) << 30)) throw std::invalid_argument("bad channel count");
        int64_t n = 1;
        std::vector<int64_t> tshape(shape.begin(), shape.end() - 1);
        for (int64_t v : tshape) n *= v;
        py::object dev = logits.attr("device");
        py::object tokens = torch.attr("empty")(py::cast(tshape), py::arg("dtype") = torch.attr(C <= 256 ? "uint8" : "int32"),
                                                py::arg("device") = dev);
        {
            py::object guard = torch.attr("cuda").attr("device")(dev);
            guard.attr("__enter__")();
            void *stream = reinterpret_cast<void *>(torch.attr("cuda").attr("current_stream")().attr("cuda_stream").cast<uintptr_t>());
            const bsq_status st = bsq_argmax_tokens_device(reinterpret_cast<const void *>(logits.attr("data_ptr")().cast<uintptr_t>()),
                                                           kind, n, int32_t(C), C,
                                                           reinterpret_cast<void *>(tokens.attr("data_ptr")().cast<uintptr_t>()),
                                                           C <= 256 ? 1 : 4, stream);
            guard.attr("__exit__")(py::none(), py::none(), py::none());
            if (st != BSQ_OK) throw_status(st);
        }
        if (ndim == 3 && !batch_first) tokens = tokens.attr("t")();  // a strided view: the decoder takes any strides
        return decode_device(tokens);
    }

    py::object decode_tokens(py::object obj) const {
        if (py::hasattr(obj, "is_cuda") && py::hasattr(obj, "data_ptr") && obj.attr("is_cuda").cast<bool>())
            return decode_device(obj);
        if (py::hasattr(obj, "detach") && py::hasattr(obj, "cpu"))  // torch tensor in host memory
            obj = obj.attr("detach")().attr("cpu")().attr("numpy")();
        py::array array = py::array::ensure(obj);
        if (!array) throw std::invalid_argument("decode_tokens expects a numpy array or torch tensor");
        py::buffer_info info = array.request();
        if (info.ptr == nullptr) throw std::invalid_argument("Empty array cannot yield a decoded string");
        if (info.ndim > 2 || info.ndim == 0)
            throw std::invalid_argument("Currently supported: 1 or 2 dimensions for decoding tokens.");
        auto load = [&](const uint8_t *p) -> uint32_t {
            switch (info.itemsize) {
            case 1: return *p;
            case 2: { uint16_t v; std::memcpy(&v, p, 2); return v; }
            case 4: { uint32_t v; std::memcpy(&v, p, 4); return v; }
            case 8: { uint64_t v; std::memcpy(&v, p, 8); return uint32_t(v); }
            default:
                throw std::runtime_error("Unexpected itemsize: expected 1, 2, 4, or 8. Found " +
                                         std::to_string(info.itemsize));
            }
        };
        auto piece = [&](uint32_t value) -> const std::string & {
            const auto it = lookup.find(int32_t(value));
            if (it == lookup.end()) throw std::runtime_error("Unexpected/invalid token " + std::to_string(value));
            return it->second;
        };
        const uint8_t *base = static_cast<const uint8_t *>(info.ptr);
        if (info.ndim == 1) {
            std::string s;
            for (py::ssize_t i = 0; i < info.shape[0]; ++i) s += piece(load(base + i * info.strides[0]));
            return py::str(s);
        }
        py::list ret;
        for (py::ssize_t r = 0; r < info.shape[0]; ++r) {
            std::string s;
            for (py::ssize_t c = 0; c < info.shape[1]; ++c)
                s += piece(load(base + r * info.strides[0] + c * info.strides[1]));
            ret.append(py::str(s));
        }
        return ret;
    }

    std::unordered_map<int32_t, py::bytes> token_decoder() const {
        std::unordered_map<int32_t, py::bytes> ret;
        for (const auto &kv : tokensets) ret[kv.first] = py::bytes(kv.second);
        return ret;
    }
};

struct Threading {
    explicit Threading(py::ssize_t n = -1) { set(n); }
    void set(py::ssize_t n) const {
        if (n > 0) g_threads.store(int(n));
    }
    py::ssize_t get() const { return default_threads(); }
};

}  // namespace

// "Host packs once; GPU g receives its slice" (sharding.encode_on_devices): ONE scan of the list under the GIL -- pointer + length
// of every item, the offsets, the first item that is too long -- kept alive (it holds the list's items) while the caller packs RANGES of it
// into memory it allocated once the total was known: device g's slice is packed and sent on its way while device g + 1's is being packed.
struct ListScan {
    Gathered g;
    py::array_t<int64_t> offsets;
    int64_t n = 0, bad = -1;
    int nthreads = 1;
};

// devices=[...] of Tokenizer.batch_tokenize / batch_onehot_encode (keyword-only; not in the reference): the same validation, in the same
// order, as the one-device call, then sharding.encode_on_devices -- one host pack, every device its slice.  Without device= the list of
// per-device shards comes back; with device= (the root) one whole-batch tensor there, written by every device over peer access.
static py::object on_devices(const Tokenizer &t, const py::sequence &batch, py::ssize_t padlen, const std::string &dt, int nthreads,
                             const py::object &mask, const py::object &device, const py::object &devices, const char *op, bool batch_first,
                             const std::string &layout) {
    const py::object self = py::cast(&t, py::return_value_policy::reference);  // (the registered wrapper of this instance)
    const bool bcl = Tokenizer::parse_layout(layout);
    (void)parse_dtype(dt);
    t.check_padlen(padlen);
    if (!mask.is_none()) throw std::invalid_argument("devices= does not take a mask: encode the masked batch per device (device=), or drop the mask");
    return py::module_::import("bioseq_amd.sharding")
        .attr("encode_on_devices")(self, batch, padlen, dt, py::arg("devices") = devices, py::arg("op") = op, py::arg("batch_first") = batch_first,
                                   py::arg("layout") = bcl ? "bcl" : "tbc", py::arg("root") = device, py::arg("nthreads") = nthreads <= 1 ? 0 : nthreads);
}

PYBIND11_MODULE(cbioseq, m) {
    m.doc() = "bioseq_amd.cbioseq: MI355X-native drop-in for the reference's cbioseq tokenizer module";
    m.attr("abi_version") = bsq_abi_version();
    m.def("device_count", [] { return bsq_device_count(); }, "Number of visible HIP devices");
    m.def("release_staging", [] {
        // g_mu (inside the library) is held by a staged batch from bsq_stage_begin to bsq_stage_end, and that thread releases the GIL
        // while its pieces land: taking g_mu here with the GIL held would deadlock against it.  Same order as every
        // other entry: PackLock (GIL released while waiting), then the library call without the GIL.
        PackLock lock;
        py::gil_scoped_release nogil;
        bsq_release_staging();
    });
    // The host half of the staged path without a device (tests, sanitizer runs): scan + pack `batch` piece by piece exactly as
    // Tokenizer::staged does -- scan_pack_fast per piece, the general passes where it declines -- into ordinary memory.
    // Returns (offsets int64[n + 1], chars uint8[total], first item longer than maxlen or -1, pieces that took the fast job).
    m.def("_pack_list_in_pieces", [](py::sequence batch, int64_t piece, int64_t maxlen, int nthreads) {
        Gathered g;
        const Scan sc = scan_begin(batch, py::none(), g, nthreads);
        if (piece <= 0) piece = std::max<int64_t>(1, sc.n);
        std::vector<int64_t> offsets(size_t(sc.n) + 1, 0);
        std::vector<uint8_t> chars(size_t(sc.n) * size_t(std::max<int64_t>(maxlen, 0)) + 8);
        Packed p;
        p.B = sc.n;
        p.offsets = offsets.data();
        p.chars = chars.data();
        int64_t bad = -1, fast = 0;
        for (int64_t lo = 0; lo < sc.n && bad < 0; lo += piece) {
            const int64_t hi = std::min<int64_t>(sc.n, lo + piece);
            if (scan_pack_fast(sc, g, p, lo, hi, nthreads, maxlen, &bad)) {
                ++fast;
                continue;
            }
            scan_range(sc, g, lo, hi, nthreads);
            for (int64_t i = lo; i < hi && bad < 0; ++i) {
                offsets[size_t(i) + 1] = offsets[size_t(i)] + int64_t(g.items[size_t(i)].len);
                if (int64_t(g.items[size_t(i)].len) > maxlen) bad = i;
            }
            if (bad < 0) pack_range(g, p, lo, hi, nthreads);
        }
        const int64_t upto = bad >= 0 ? bad : sc.n;
        py::array_t<int64_t> o(upto + 1);
        std::memcpy(o.mutable_data(), offsets.data(), size_t(upto + 1) * 8);
        py::array_t<uint8_t> c(offsets[size_t(upto)]);
        if (offsets[size_t(upto)]) std::memcpy(c.mutable_data(), chars.data(), size_t(offsets[size_t(upto)]));
        return py::make_tuple(o, c, bad, fast);
    }, py::arg("batch"), py::arg("piece"), py::arg("maxlen"), py::arg("nthreads") = 0);
    // "Host packs once" (sharding.encode_on_devices): ONE scan of the list under the GIL, then ONE pack into memory the caller
    // allocates once the total is known -- `alloc(nbytes)` returns a writable, C-contiguous uint8 buffer of at least nbytes (a numpy view
    // of a PINNED torch tensor: the slices of the N devices then go up as asynchronous copies on N copy streams).  Returns (offsets
    // int64[n + 1], the buffer alloc returned, index of the first item longer than maxlen or -1).  Nothing is packed when an item is too long.
    m.def("_pack_list_into", [](py::sequence batch, int64_t maxlen, int nthreads, py::function alloc) -> py::tuple {
        Gathered g;
        const Scan sc = scan_begin(batch, py::none(), g, nthreads);
        scan_range(sc, g, 0, sc.n, nthreads);
        py::array_t<int64_t> offsets(sc.n + 1);
        int64_t *o = offsets.mutable_data();
        int64_t bad = -1, acc = 0;
        o[0] = 0;
        for (Py_ssize_t i = 0; i < sc.n; ++i) {
            const int64_t len = int64_t(g.items[size_t(i)].len);
            if (len > maxlen && bad < 0) bad = i;
            acc += len;
            o[i + 1] = acc;
        }
        if (bad >= 0) return py::make_tuple(offsets, py::object(py::none()), bad);
        py::object buf = alloc(py::int_(size_t(acc) + 16));  // (+16: the kernels' unaligned 16-byte loads never leave the allocation)
        py::buffer_info info = py::reinterpret_borrow<py::buffer>(buf).request(true);
        if (info.itemsize != 1 || info.ndim != 1 || size_t(info.shape[0]) < size_t(acc) || (info.strides[0] != 1 && info.shape[0] > 1))
            throw std::invalid_argument("_pack_list_into: alloc() must return a writable, contiguous buffer of bytes of the requested size");
        Packed p;
        p.B = sc.n;
        p.offsets = o;
        p.chars = static_cast<uint8_t *>(info.ptr);
        pack_range(g, p, 0, sc.n, nthreads);
        return py::make_tuple(offsets, buf, bad);
    }, py::arg("batch"), py::arg("maxlen"), py::arg("nthreads"), py::arg("alloc"));
    py::class_<ListScan>(m, "_ListScan")
        .def(py::init([](py::sequence batch, int64_t maxlen, int nthreads) {
                 auto sc = std::make_unique<ListScan>();
                 const Scan scan = scan_begin(batch, py::none(), sc->g, nthreads);
                 scan_range(scan, sc->g, 0, scan.n, nthreads);
                 sc->n = scan.n;
                 sc->nthreads = nthreads;
                 sc->offsets = py::array_t<int64_t>(scan.n + 1);
                 int64_t *o = sc->offsets.mutable_data();
                 int64_t acc = 0;
                 o[0] = 0;
                 for (Py_ssize_t i = 0; i < scan.n; ++i) {
                     const int64_t len = int64_t(sc->g.items[size_t(i)].len);
                     if (len > maxlen && sc->bad < 0) sc->bad = i;
                     acc += len;
                     o[i + 1] = acc;
                 }
                 return sc;
             }),
             py::arg("batch"), py::arg("maxlen"), py::arg("nthreads") = 0)
        .def_readonly("offsets", &ListScan::offsets)
        .def_readonly("bad", &ListScan::bad)
        .def_readonly("n", &ListScan::n)
        .def("pack", [](ListScan &sc, int64_t lo, int64_t hi, py::buffer dst) {
            // items [lo, hi) -> dst[offsets[lo] : offsets[hi]] (dst = the whole batch's buffer: writable, contiguous bytes)
            if (lo < 0 || hi < lo || hi > sc.n) throw py::index_error("_ListScan.pack: range outside the list");
            py::buffer_info info = dst.request(true);
            const int64_t *o = sc.offsets.data();
            if (info.itemsize != 1 || info.ndim != 1 || int64_t(info.shape[0]) < o[hi] || (info.strides[0] != 1 && info.shape[0] > 1))
                throw std::invalid_argument("_ListScan.pack: dst must be a writable, contiguous buffer of at least offsets[hi] bytes");
            Packed p;
            p.B = sc.n;
            p.offsets = const_cast<int64_t *>(o);
            p.chars = static_cast<uint8_t *>(info.ptr);
            pack_range(sc.g, p, lo, hi, sc.nthreads);
        }, py::arg("lo"), py::arg("hi"), py::arg("dst"));
    m.def("alphabet_keys", [] {
        std::vector<std::string> k;
        for (int i = 0; i < bsq_num_keys(); ++i) k.emplace_back(bsq_key_name(i));
        return k;
    });
    // omp.cpp:27-32 -- here the knob sizes the host packing threads
    m.def("set_num_threads", [](py::ssize_t n) {
        if (n > 0) g_threads.store(int(n));
    });
    m.def("get_num_threads", [] { return py::ssize_t(default_threads()); });
    m.def("set_host_threads", [](py::ssize_t n) { g_host_threads.store(n < 0 ? 0 : (n > 1024 ? 1024 : int(n))); }, py::arg("n"),
          "Host threads for the scan + pack of a list batch when a call leaves nthreads at the reference's default (1): n > 0 exactly n "
          "(1 = serial), 0 = automatic (the default; also BSQ_HOST_THREADS).  An explicit nthreads > 1 always wins.");
    m.def("get_host_threads", [] { return py::ssize_t(host_threads_policy()); });
    py::class_<Threading>(m, "Threading")
        .def(py::init<>())
        .def(py::init<py::ssize_t>())
        .def_property("nthreads", &Threading::get, &Threading::set)
        .def_property("p", &Threading::get, &Threading::set);

    py::class_<Tokenizer>(m, "Tokenizer")
        .def(py::init<std::string, bool, bool, bool>(), py::arg("key"), py::arg("eos") = false,
             py::arg("bos") = false, py::arg("padchar") = false)
        .def("batch_tokenize",
             [](const Tokenizer &t, py::sequence batch, py::ssize_t padlen, const std::string &dt, bool batch_first, int nthreads,
                const py::object &device, const py::object &devices) -> py::object {
                 if (devices.is_none()) return t.batch_tokenize(batch, padlen, dt, batch_first, nthreads, device);
                 return on_devices(t, batch, padlen, dt, nthreads, py::none(), device, devices, "tokenize", batch_first, "tbc");
             },
             py::arg("batch"), py::arg("padlen") = -1,
             py::arg("destchar") = "B", py::arg("batch_first") = false, py::arg("nthreads") = 1, py::kw_only(),
             py::arg("device") = py::none(), py::arg("devices") = py::none(),
             "Token matrix of a list of sequences -- (padlen, B), or (B, padlen) with batch_first -- as the reference's\n"
             "Tokenizer.batch_tokenize (src/tokenize.cpp:82-98, tokenize.h:381-485); device= keeps the result on the GPU.\n\n"
             "nthreads: in the reference the OpenMP team of the encode loop; here the encode runs on the GPU and the value governs the\n"
             "host scan + pack only.  nthreads > 1 is honoured as given.  The signature default 1 CANNOT be told from an explicit 1:\n"
             "both defer to the module policy -- set_host_threads(n) / BSQ_HOST_THREADS, where 0 (the initial value) means one thread\n"
             "below 8192 items and up to 16 above.  A caller that needs a strictly serial host path whatever the batch size\n"
             "(DataLoader workers, cgroup-limited jobs) calls set_host_threads(1) once.\n\n"
             "devices=[...] (keyword-only; not in the reference): the batch is sharded by sequence over these HIP devices from this one\n"
             "process -- one host pack, every device uploads and encodes its slice (sharding.encode_on_devices); returns the list of\n"
             "per-device shards, or with device= (the root) one whole-batch tensor there.")
        .def("batch_onehot_encode",
             [](const Tokenizer &t, py::sequence batch, py::ssize_t padlen, const std::string &dt, int nthreads, const py::object &mask,
                const py::object &device, const std::string &layout, const py::object &devices) -> py::object {
                 if (devices.is_none()) return t.batch_onehot_encode(batch, padlen, dt, nthreads, mask, device, layout);
                 return on_devices(t, batch, padlen, dt, nthreads, mask, device, devices, "onehot", false, layout);
             },
             py::arg("batch"), py::arg("padlen") = -1,
             py::arg("destchar") = "B", py::arg("nthreads") = 1, py::arg("mask") = py::none(), py::kw_only(),
             py::arg("device") = py::none(), py::arg("layout") = "tbc", py::arg("devices") = py::none(),
             "One-hot tensor (padlen, B, C) of a list of sequences, as the reference's Tokenizer.batch_onehot_encode\n"
             "(src/tokenize.cpp:65-81, tokenize.h:283-371); device= keeps the result on the GPU, layout='bcl' writes (B, C, padlen).\n\n"
             "nthreads: the host scan + pack only (the encode runs on the GPU).  nthreads > 1 is honoured as given; the signature\n"
             "default 1 cannot be told from an explicit 1 and defers to the module policy (set_host_threads / BSQ_HOST_THREADS; 0 =\n"
             "one thread below 8192 items, up to 16 above).  For a strictly serial host path call set_host_threads(1).\n\n"
             "devices=[...] (keyword-only; not in the reference): sharded by sequence over these HIP devices from this one process\n"
             "(sharding.encode_on_devices): the list of per-device shards, or with device= one whole-batch tensor there; no mask.")
        .def("tokenize_packed",
             [](const Tokenizer &t, const py::object &chars, const py::object &offsets, py::ssize_t padlen,
                const std::string &dt, bool batch_first, const py::object &device, bool validate) {
                 return t.encode_packed(false, chars, offsets, padlen, dt, batch_first, py::none(), device, validate);
             },
             py::arg("chars"), py::arg("offsets"), py::arg("padlen"), py::arg("destchar") = "B",
             py::arg("batch_first") = false, py::arg("device") = py::none(), py::arg("validate") = true)
        .def("onehot_packed",
             [](const Tokenizer &t, const py::object &chars, const py::object &offsets, py::ssize_t padlen,
                const std::string &dt, const py::object &mask, const py::object &device, bool validate,
                const std::string &layout) {
                 return t.encode_packed(true, chars, offsets, padlen, dt, false, mask, device, validate,
                                        Tokenizer::parse_layout(layout));
             },
             py::arg("chars"), py::arg("offsets"), py::arg("padlen"), py::arg("destchar") = "B",
             py::arg("mask") = py::none(), py::arg("device") = py::none(), py::arg("validate") = true,
             py::arg("layout") = "tbc")
        .def("onehot_encode",
             [](const Tokenizer &t, py::str s, py::ssize_t padlen, const std::string &dt, const py::object &device) {
                 Py_ssize_t n = 0;
                 const char *p = PyUnicode_AsUTF8AndSize(s.ptr(), &n);
                 if (!p) throw py::error_already_set();
                 return t.onehot_single(p, n, padlen, dt, device);
             },
             py::arg("str"), py::arg("padlen") = 0, py::arg("destchar") = "f", py::kw_only(), py::arg("device") = py::none())
        .def("onehot_encode",
             [](const Tokenizer &t, py::bytearray s, py::ssize_t padlen, const std::string &dt, const py::object &device) {
                 return t.onehot_single(PyByteArray_AS_STRING(s.ptr()), PyByteArray_GET_SIZE(s.ptr()), padlen, dt, device);
             },
             py::arg("bytearray"), py::arg("padlen") = 0, py::arg("destchar") = "f", py::kw_only(), py::arg("device") = py::none())
        .def("onehot_encode",
             [](const Tokenizer &t, py::bytes s, py::ssize_t padlen, const std::string &dt, const py::object &device) {
                 char *p = nullptr;
                 Py_ssize_t n = 0;
                 if (PyBytes_AsStringAndSize(s.ptr(), &p, &n)) throw py::error_already_set();
                 return t.onehot_single(p, n, padlen, dt, device);
             },
             py::arg("str"), py::arg("padlen") = 0, py::arg("destchar") = "B", py::kw_only(), py::arg("device") = py::none())
        .def("decode_tokens", &Tokenizer::decode_tokens, py::arg("tokenizer"))
        .def("decode_logits", &Tokenizer::decode_logits, py::arg("logits"), py::arg("batch_first") = true,
             "argmax over the last axis on the device, then decode (README: 'if you have logits, use an argmax ...')")
        .def("lut", [](const Tokenizer &t) { return t.lookup; })
        .def("token_map", [](const Tokenizer &t) { return t.token_map_str; })
        .def("token_decoder", &Tokenizer::token_decoder)
        .def("nchars", [](const Tokenizer &t) { return int(t.desc.nchars); })
        .def("alphabet_size", [](const Tokenizer &t) { return size_t(bsq_alphabet_size(&t.desc)); })
        .def("bos", [](const Tokenizer &t) { return int(bsq_bos_id(&t.desc)); })
        .def("eos", [](const Tokenizer &t) { return int(bsq_eos_id(&t.desc)); })
        .def("pad", [](const Tokenizer &t) { return int(bsq_pad_id(&t.desc)); })
        .def_property_readonly("key", [](const Tokenizer &t) { return t.key; })
        .def("is_padded", [](const Tokenizer &t) { return t.desc.padchar != 0; })
        .def("includes_bos", [](const Tokenizer &t) { return t.desc.bos != 0; })
        .def("includes_eos", [](const Tokenizer &t) { return t.desc.eos != 0; })
        .def("byte_table",
             [](const Tokenizer &t) {
                 py::array_t<int8_t> a(256);
                 std::memcpy(a.mutable_data(), t.desc.lut, 256);
                 return a;
             },
             "256-entry byte -> id table (-1 = unmapped)")
        .def(py::pickle(
            [](const Tokenizer &t) {
                return py::make_tuple(t.key, t.desc.eos != 0, t.desc.bos != 0, t.desc.padchar != 0);
            },
            [](py::tuple s) {
                return Tokenizer(s[0].cast<std::string>(), s[1].cast<bool>(), s[2].cast<bool>(), s[3].cast<bool>());
            }));
}
