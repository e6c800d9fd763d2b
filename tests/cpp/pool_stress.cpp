// Stress of bwa-mem2_amd/csrc/host_pool.h (built and run by tests/test_host_pool.py): every phase must run exactly n copies of its
// function, whatever the sequence of participant counts, from several calling threads at once; the counting sort must be a stable sort.
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <numeric>
#include "host_pool.h"

static int caller(unsigned seed, int phases, int max_n) {
    unsigned x = seed;
    for (int p = 0; p < phases; ++p) {
        x = x * 1664525u + 1013904223u;
        const int n = 1 + (int)((x >> 16) % (unsigned)max_n);
        std::atomic<int> copies(0);
        bm2_run_threads(n, [&]() { copies.fetch_add(1); volatile int sink = 0; if ((x >> 8) & 1) for (int i = 0; i < 2000; ++i) sink = sink + i; });
        if (copies.load() != n) { fprintf(stderr, "phase %d: %d copies ran, %d wanted\n", p, copies.load(), n); return 1; }
    }
    return 0;
}

int main(int argc, char **argv) {
    const int phases = argc > 1 ? atoi(argv[1]) : 20000, max_n = argc > 2 ? atoi(argv[2]) : 12;
    int bad[3] = { 0, 0, 0 };
    std::thread a([&]() { bad[0] = caller(1, phases, max_n); }), b([&]() { bad[1] = caller(2, phases, max_n); });
    bad[2] = caller(3, phases, max_n);
    a.join(); b.join();
    if (bad[0] || bad[1] || bad[2]) return 1;
    // counting sort against std::stable_sort
    const int n = 300000, n_keys = 1000;
    std::vector<int> key((size_t)n), order((size_t)n), ref((size_t)n);
    unsigned x = 7;
    for (int i = 0; i < n; ++i) { x = x * 1664525u + 1013904223u; key[(size_t)i] = (int)((x >> 12) % n_keys); }
    std::iota(ref.begin(), ref.end(), 0);
    std::stable_sort(ref.begin(), ref.end(), [&](int p, int q) { return key[(size_t)p] < key[(size_t)q]; });
    for (int threads : { 1, 3, 8 }) {
        std::fill(order.begin(), order.end(), -1);
        bm2_counting_order(n, n_keys, threads, [&](int i) { return key[(size_t)i]; }, order.data());
        if (order != ref) { fprintf(stderr, "counting order differs from stable_sort on %d threads\n", threads); return 1; }
    }
    int64_t sum = 0; std::atomic<int64_t> got(0);
    for (int i = 0; i < n; ++i) sum += key[(size_t)i];
    bm2_parallel_ranges(n, 4096, 8, [&](int64_t lo, int64_t hi) { int64_t s = 0; for (int64_t i = lo; i < hi; ++i) s += key[(size_t)i]; got += s; });
    if (got.load() != sum) { fprintf(stderr, "parallel ranges: wrong sum\n"); return 1; }
    printf("ok\n");
    return 0;
}
