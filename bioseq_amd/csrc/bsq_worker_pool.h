// Host worker pool of the pybind11 layer (cbioseq_module.cpp); a header of its own so that tests/test_worker_pool.py can
// compile it with -fsanitize=thread and hammer it without a Python interpreter or a GPU.
#ifndef BSQ_WORKER_POOL_H
#define BSQ_WORKER_POOL_H
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

// Persistent host workers for the two parallel phases of a list-of-objects call (item scan, pinned pack): creating and
// joining 2 x nthreads std::threads per call cost 0.2-0.4 ms of a 2.7 ms call (profiles/r02/e2e_python_api.txt: 32 threads
// were SLOWER than 8).  parallel_for(n, fn) runs fn(t) for t in [0, n) -- t = 0 on the caller -- and returns when all are
// done.  Workers only ever touch raw bytes (never the interpreter), the caller keeps the GIL meanwhile, so the items stay
// alive.  One job at a time (g_pack_mu / the GIL serialise the callers).
// A call now runs five to nine short jobs in a row (the scan, then one pack per piece of ~0.1 ms): a worker that has finished
// a job keeps polling for the next one for kSpinUs before it goes to sleep on the condition variable, and the caller polls for
// the end of a job -- waking fifteen sleepers through a futex and one mutex cost 30-60 us per job.
class WorkerPool {
  public:
    ~WorkerPool() {
        stop_.store(true);
        {
            std::lock_guard<std::mutex> l(mu_);
        }
        cv_.notify_all();
        for (std::thread &t : threads_) t.join();
    }
    void parallel_for(int n, const std::function<void(int)> &fn) {
        if (n <= 1) {
            if (n == 1) fn(0);
            return;
        }
        std::unique_lock<std::mutex> job_lock(job_mu_);  // one job at a time
        grow(n - 1);
        const uint64_t g = generation_.load(std::memory_order_relaxed) + 1;
        fn_.store(&fn, std::memory_order_relaxed);
        pending_.store(n - 1, std::memory_order_relaxed);
        slot_.store(((g & 0xFFFFFFu) << 40) | (uint64_t(n) << 20) | 1u, std::memory_order_release);
        generation_.store(g);  // (sequentially consistent, as the sleepers' count: one side always sees the other)
        if (sleepers_.load() > 0) {
            {
                std::lock_guard<std::mutex> l(mu_);  // a worker between its last look at generation_ and its wait holds mu_
            }
            cv_.notify_all();
        }
        // whatever fn(0) does on the caller -- return or throw --, the workers still hold &fn: wait for them before unwinding
        struct Drain {
            WorkerPool *p;
            ~Drain() {
                for (unsigned spins = 0; p->pending_.load(std::memory_order_acquire) != 0; ++spins) {
                    if (spins < 4096) __builtin_ia32_pause();
                    else std::this_thread::yield();
                }
                p->fn_.store(nullptr, std::memory_order_relaxed);  // (slot_ stays exhausted: a late worker finds nothing to take)
            }
        } drain{this};
        fn(0);
    }

  private:
    static constexpr long kSpinUs = 300;
    void grow(int want) {
        while (int(threads_.size()) < want && threads_.size() < 256) threads_.emplace_back([this] { run(); });
    }
    void run() {
        uint64_t seen = 0;
        for (;;) {
            // the next job: poll for kSpinUs, then sleep
            const auto t0 = std::chrono::steady_clock::now();
            for (unsigned spins = 0; generation_.load(std::memory_order_acquire) == seen && !stop_.load(std::memory_order_relaxed); ++spins) {
                __builtin_ia32_pause();
                if ((spins & 255u) == 255u &&
                    std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() > kSpinUs) {
                    std::unique_lock<std::mutex> l(mu_);
                    sleepers_.fetch_add(1);
                    cv_.wait(l, [&] { return stop_.load() || generation_.load() != seen; });
                    sleepers_.fetch_sub(1);
                }
            }
            if (stop_.load()) return;
            seen = generation_.load(std::memory_order_acquire);
            // Tasks are taken by compare-and-swap on ONE word that holds the job's id, its task count and the next task: a worker that is
            // late for job k can neither take a task of job k + 1 under job k's function nor use up one of its indices.
            uint64_t v = slot_.load(std::memory_order_acquire);
            for (;;) {
                const uint32_t limit = uint32_t(v >> 20) & 0xFFFFFu, t = uint32_t(v) & 0xFFFFFu;
                if ((v >> 40) != (seen & 0xFFFFFFu) || t >= limit) break;
                if (!slot_.compare_exchange_weak(v, v + 1, std::memory_order_acq_rel, std::memory_order_acquire)) continue;
                (*fn_.load(std::memory_order_relaxed))(int(t));
                pending_.fetch_sub(1, std::memory_order_release);
                v = slot_.load(std::memory_order_acquire);
            }
        }
    }
    std::mutex mu_, job_mu_;
    std::condition_variable cv_;
    std::vector<std::thread> threads_;  // (grown under job_mu_ only)
    std::atomic<const std::function<void(int)> *> fn_{nullptr};
    std::atomic<int> pending_{0}, sleepers_{0};
    std::atomic<uint64_t> generation_{0}, slot_{0};  // slot_: job id [63:40] | tasks of the job [39:20] | next task [19:0]
    std::atomic<bool> stop_{false};
};

#endif  // BSQ_WORKER_POOL_H
