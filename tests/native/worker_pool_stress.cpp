// Stress test of bioseq_amd/csrc/bsq_worker_pool.h (built with -fsanitize=thread by tests/test_worker_pool.py): every task of every
// job runs exactly once under that job's function, whatever the workers are doing (polling, asleep, late for the previous job);
// a job whose caller-side task throws still waits for its workers.
#include <cstdio>
#include "bsq_worker_pool.h"

int main() {
    WorkerPool pool;
    std::vector<int> hits(64);
    long total = 0;
    for (int iter = 0; iter < 20000; ++iter) {
        const int n = 2 + iter % 17;
        for (int &h : hits) h = 0;
        std::function<void(int)> fn = [&](int t) { hits[size_t(t)] += 1 + iter; };
        pool.parallel_for(n, fn);
        for (int t = 0; t < 64; ++t)
            if (hits[size_t(t)] != (t < n ? 1 + iter : 0)) {
                std::printf("BAD job %d task %d ran %d\n", iter, t, hits[size_t(t)]);
                return 1;
            }
        total += n;
        if (iter % 1000 == 999)  // let the workers fall asleep now and then (they poll for 300 us)
            std::this_thread::sleep_for(std::chrono::microseconds(iter % 3000 == 2999 ? 1500 : 100));
    }
    try {
        std::function<void(int)> fn = [&](int t) {
            if (t == 0) throw 1;
        };
        pool.parallel_for(8, fn);
        return 2;
    } catch (int) {
    }
    std::printf("WORKER_POOL_OK %ld\n", total);
    return 0;
}
