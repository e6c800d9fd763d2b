// host_pool.h -- the host side's worker threads (sam_tail.cpp, fastq_io.cpp): bm2_run_threads(n, f) runs n copies of f.
#pragma once
#include <limits.h>
#include <linux/futex.h>
#include <pthread.h>
#include <sched.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <sys/syscall.h>
#include <unistd.h>
#include <atomic>
#include <functional>
#include <thread>
#include <vector>
#include <string>

// Worker threads for the host side's short parallel phases (SAM tail, FASTQ parser).  A chunk goes through a dozen short phases, so what counts is how fast ALL workers get
// going: a queue behind one mutex hands the lock from one woken thread to the next (each hand-over costs a scheduler wake-up: milliseconds
// for a few hundred threads), and spawning threads per phase cost more than the phases themselves.  Here every calling thread (a tail
// worker of the pipeline) owns its workers; a phase is published by bumping a generation word and waking every sleeper with ONE futex
// call; a worker that finished spins briefly before it sleeps, so back-to-back phases find the workers awake; completion is a counter
// the caller spins / sleeps on.  No lock anywhere.  run(n, f) runs n copies of f (the caller is one of them) and returns when all are through.
class TailPool {
    std::vector<std::thread> workers;
    std::function<void()> *job = nullptr;
    alignas(64) std::atomic<uint32_t> gen{0};                   // futex word: phase number << 12 | workers taking part in it
    alignas(64) std::atomic<uint32_t> left{0};                  // futex word: participants still inside f
    std::atomic<bool> stop{false};
    static void pause() {
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#endif
    }
    // How long a thread polls a word before it sleeps on it: ~100 us by default (BM2_POOL_SPIN_US; polling burns CPU-time quota where there is one).  The gaps between the phases of a chunk
    // are mostly shorter (prefix sums, a scan over blocks), and a sleeper's wake-up costs a scheduler round trip -- milliseconds where the
    // CPUs are virtual and halt when idle -- while a poller on an otherwise idle CPU costs nothing that anybody wanted.
    static int spin_rounds() {
        static const int r = []() { const char *e = getenv("BM2_POOL_SPIN_US"); const long us = e && *e ? atol(e) : 100; return (int)(us < 0 ? 0 : us > 100000 ? 100000 : us); }();
        return r;                                                 // rounds of ~1 us (16 pauses)
    }
    template <class W> static void wait_while_equal(std::atomic<uint32_t> *w, uint32_t seen, W still) {
        const int rounds = spin_rounds();
        for (int r = 0; r < rounds; ++r) {
            for (int k = 0; k < 16; ++k) pause();
            if (!still()) return;
        }
        while (still()) futex_wait(w, seen);
    }
    static void futex_wait(std::atomic<uint32_t> *w, uint32_t seen) { syscall(SYS_futex, (uint32_t *)w, FUTEX_WAIT_PRIVATE, seen, nullptr, nullptr, 0); }
    static void futex_wake_all(std::atomic<uint32_t> *w) { syscall(SYS_futex, (uint32_t *)w, FUTEX_WAKE_PRIVATE, INT_MAX, nullptr, nullptr, 0); }
    // A worker stays on one CPU of the PROCESS's affinity mask (read once, from the main thread: a pool made by an already pinned thread
    // must not inherit that thread's one-CPU mask): a freshly woken thread then starts where it slept instead of queueing on the
    // waker's CPU until the load balancer gets to it (milliseconds on virtualised hosts; a phase lasts a few).  One process per GPU is
    // the deployment: rank r of w (LOCAL_RANK / LOCAL_WORLD_SIZE, as torchrun exports them) hands out only the r-th of w equal slices of
    // the mask, so the ranks of a node never stack their workers on the same CPUs; without a launcher the start is hashed from the pid.
    // BM2_TAIL_PIN=0 leaves the placement to the scheduler.
    struct PinPlan { cpu_set_t all; int n = 0, base = 0, span = 0; };
    static const PinPlan &pin_plan() {
        static const PinPlan plan = []() {
            PinPlan p; CPU_ZERO(&p.all);
            if (sched_getaffinity(getpid(), sizeof p.all, &p.all) != 0) return p;      // (pid = the main thread: the process's own mask)
            p.n = CPU_COUNT(&p.all);
            const char *r = getenv("LOCAL_RANK"), *w = getenv("LOCAL_WORLD_SIZE");
            const int world = w && *w ? atoi(w) : 1, rank = r && *r ? atoi(r) : 0;
            if (world > 1 && rank >= 0 && rank < world && p.n >= 2 * world) { p.span = p.n / world; p.base = rank * p.span; }
            else { p.span = p.n; p.base = p.n > 1 ? (int)(((unsigned)getpid() * 2654435761u >> 8) % (unsigned)p.n) : 0; }
            return p;
        }();
        return plan;
    }
    static void pin_self() {
        static const bool on = []() { const char *e = getenv("BM2_TAIL_PIN"); return !(e && e[0] == '0'); }();
        if (!on) return;
        const PinPlan &p = pin_plan();
        if (p.n < 2 || p.span < 1) return;
        static std::atomic<unsigned> next_cpu{1};
        int k = (p.base + (int)(next_cpu.fetch_add(1) % (unsigned)p.span)) % p.n;
        for (int c = 0; c < CPU_SETSIZE; ++c)
            if (CPU_ISSET(c, &p.all) && k-- == 0) { cpu_set_t one; CPU_ZERO(&one); CPU_SET(c, &one); pthread_setaffinity_np(pthread_self(), sizeof one, &one); return; }
    }
    // The generation word carries the phase's participant count in its low bits: a worker decides from ONE load whether the phase it
    // saw is its business (a separate `want` could already belong to the next phase by the time a slow non-participant reads it).
    enum { WANT_BITS = 12, WANT_MASK = (1 << WANT_BITS) - 1 };
    void loop(int idx, uint32_t seen) {
        pin_self();
        for (;;) {
            wait_while_equal(&gen, seen, [&]() { return gen.load(std::memory_order_acquire) == seen; });
            const uint32_t g = gen.load(std::memory_order_acquire);      // (ONE load decides: see above)
            seen = g;
            if (stop.load(std::memory_order_acquire)) return;
            if (idx < (int)(g & WANT_MASK)) {
                (*job)();
                if (left.fetch_sub(1, std::memory_order_acq_rel) == 1) futex_wake_all(&left);
            }
        }
    }
public:
    TailPool() {}
    ~TailPool() {
        stop.store(true, std::memory_order_release);
        gen.store((((gen.load(std::memory_order_relaxed) >> WANT_BITS) + 1) << WANT_BITS), std::memory_order_release);
        futex_wake_all(&gen);
        for (auto &t : workers) t.join();
    }
    void run(int n, std::function<void()> f) {
        if (n <= 1) { f(); return; }
        if (n - 1 > WANT_MASK) n = WANT_MASK + 1;
        while ((int)workers.size() < n - 1) {                     // (a worker starts out having "seen" the current generation)
            const int idx = (int)workers.size(); const uint32_t g = gen.load(std::memory_order_relaxed);
            // (a worker is named after the thread it works for -- "<owner>-w": bench.py's per-thread CPU table of the FASTQ -> SAM leg tells the parser's pool from the tail workers' by it)
            char nm[16] = "bm2";
            if (pthread_getname_np(pthread_self(), nm, sizeof nm) != 0) { nm[0] = 'b'; nm[1] = 'm'; nm[2] = '2'; nm[3] = 0; }
            nm[13] = 0;
            const std::string wname = std::string(nm) + "-w";
            workers.emplace_back([this, idx, g, wname]() { pthread_setname_np(pthread_self(), wname.c_str()); loop(idx, g); });
        }
        job = &f;
        left.store((uint32_t)(n - 1), std::memory_order_relaxed);
        gen.store((((gen.load(std::memory_order_relaxed) >> WANT_BITS) + 1) << WANT_BITS) | (uint32_t)(n - 1), std::memory_order_release);
        futex_wake_all(&gen);
        f();                                                      // the caller takes part
        for (uint32_t l; (l = left.load(std::memory_order_acquire)) != 0;)
            wait_while_equal(&left, l, [&]() { return left.load(std::memory_order_acquire) == l; });
    }
};
// The CPUs this process can really use: the hardware threads it may run on, capped by its cgroup's CPU-time quota (cgroup v2 cpu.max,
// v1 cpu.cfs_quota_us / cpu.cfs_period_us).  A GPU slice of a shared node typically sees every hardware thread of the host but is given
// the time of a few (measured on the MI355X box of this project: 256 visible, quota 16): threads beyond the quota do not run in
// parallel, they get the whole process throttled for the rest of the scheduler period -- and spinning burns the quota too.
inline int bm2_effective_cpus() {
    static const int n = []() {
        int hw = (int)std::thread::hardware_concurrency();
        cpu_set_t all;
        if (sched_getaffinity(getpid(), sizeof all, &all) == 0 && CPU_COUNT(&all) > 0 && CPU_COUNT(&all) < hw) hw = CPU_COUNT(&all);
        if (hw < 1) hw = 1;
        long long quota = -1, period = 100000;
        if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
            char q[64] = "";
            if (fscanf(f, "%63s %lld", q, &period) >= 1 && q[0] != 'm') quota = atoll(q);
            fclose(f);
        } else {
            if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(g, "%lld", &quota) != 1) quota = -1; fclose(g); }
            if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(g, "%lld", &period) != 1) period = 100000; fclose(g); }
        }
        if (quota > 0 && period > 0) { const int c = (int)((quota + period - 1) / period); if (c >= 1 && c < hw) hw = c; }
        const char *e = getenv("BM2_HOST_CPUS");                  // (override)
        if (e && atoi(e) > 0) hw = atoi(e);
        return hw;
    }();
    return n;
}

// How many host threads the call in progress on THIS thread may use (bm2_sam_pe / bm2_sam_se set it from bm2_sam_opt::n_threads for
// the batch hooks they call, which have no such argument); 0 = all hardware threads.
inline int &bm2_host_thread_budget() { static thread_local int v = 0; return v; }
inline int bm2_host_threads() {
    const int b = bm2_host_thread_budget();
    if (b > 0) return b;
    return bm2_effective_cpus();
}

inline void bm2_run_threads(int n_threads, std::function<void()> f) {
    if (n_threads <= 1) { f(); return; }
    static thread_local TailPool pool;                            // one set of workers per calling thread
    pool.run(n_threads, std::move(f));
}

// order[0, n) = the items 0 .. n-1 by ascending key(i) in [0, n_keys), items of one key in index order (a counting sort on up to
// n_threads threads over static ranges, so the result does not depend on the thread count)
template <class K> void bm2_counting_order(int n, int n_keys, int n_threads, K key, int *order) {
    if (n <= 0) return;
    int T = n_threads < 1 ? 1 : n_threads;
    if (T > n / 16384 + 1) T = n / 16384 + 1;
    if ((int64_t)T * n_keys > (1 << 22)) T = (1 << 22) / n_keys > 1 ? (1 << 22) / n_keys : 1;
    std::vector<uint32_t> keys((size_t)n), cnt((size_t)T * (size_t)n_keys, 0);
    std::atomic<int> nx(0);
    bm2_run_threads(T, [&]() {
        for (int t; (t = nx.fetch_add(1)) < T;) {
            uint32_t *c = cnt.data() + (size_t)t * (size_t)n_keys;
            for (int64_t i = (int64_t)n * t / T, hi = (int64_t)n * (t + 1) / T; i < hi; ++i) { const uint32_t k = (uint32_t)key((int)i); keys[(size_t)i] = k; ++c[k]; }
        }
    });
    uint32_t pos = 0;
    for (int k = 0; k < n_keys; ++k)
        for (int t = 0; t < T; ++t) { uint32_t &c = cnt[(size_t)t * (size_t)n_keys + (size_t)k]; const uint32_t v = c; c = pos; pos += v; }
    nx = 0;
    bm2_run_threads(T, [&]() {
        for (int t; (t = nx.fetch_add(1)) < T;) {
            uint32_t *c = cnt.data() + (size_t)t * (size_t)n_keys;
            for (int64_t i = (int64_t)n * t / T, hi = (int64_t)n * (t + 1) / T; i < hi; ++i) order[c[keys[(size_t)i]]++] = (int)i;
        }
    });
}

// f(lo, hi) over [0, n) in pieces of `grain` items on up to n_threads threads
template <class F> void bm2_parallel_ranges(int64_t n, int64_t grain, int n_threads, F f) {
    if (n <= 0) return;
    const int64_t pieces = (n + grain - 1) / grain;
    int T = n_threads < 1 ? 1 : n_threads;
    if ((int64_t)T > pieces) T = (int)pieces;
    std::atomic<int64_t> nx(0);
    bm2_run_threads(T, [&]() { for (int64_t p; (p = nx.fetch_add(1)) < pieces;) f(p * grain, (p + 1) * grain < n ? (p + 1) * grain : n); });
}
