// HostThreads.hpp -- a parallel-for over independent items on the host's cores (models of a .hmm file are parsed and
// projected independently: host/phmm/PhmmPreprocessor.cpp:9-31 loops over them one by one).  std::thread only.
#ifndef HAVAC_HOST_THREADS_HPP
#define HAVAC_HOST_THREADS_HPP

#include <algorithm>
#include <atomic>
#include <cstddef>
#include <thread>
#include <vector>

// The cores this process may use, capped (a GPU box gives 16 to a GPU; more threads than that only add start-up cost).
inline unsigned havacHostThreads(size_t items, unsigned cap = 16) {
    unsigned n = std::thread::hardware_concurrency();
    if (n == 0) n = 1;
    n = std::min(n, cap);
    if (items < n) n = (unsigned)std::max<size_t>(items, 1);
    return n;
}

// fn(i) for every i in [0, n), items handed out one at a time (they differ in size: model lengths 50 ... 2000)
template <class F>
void havacParallelFor(size_t n, unsigned threads, F &&fn) {
    if (threads <= 1 || n <= 1) {
        for (size_t i = 0; i < n; i++) fn(i);
        return;
    }
    std::atomic<size_t> next{0};
    auto work = [&]() {
        for (size_t i = next.fetch_add(1, std::memory_order_relaxed); i < n; i = next.fetch_add(1, std::memory_order_relaxed)) fn(i);
    };
    std::vector<std::thread> pool;
    pool.reserve(threads - 1);
    for (unsigned t = 1; t < threads; t++) pool.emplace_back(work);
    work();
    for (auto &t : pool) t.join();
}
#endif
