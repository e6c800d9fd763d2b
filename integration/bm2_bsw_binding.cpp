// integration/bm2_bsw_binding.cpp -- seam S1 of libbm2 compiled INTO bwa-mem2 (oracle/Makefile, target `bm2s1`): BASELINE.json config 2,
// "banded-SW HIP kernel only, FM-index seeding still on host".
//
// The reference's extension stage (mem_chain2aln_across_reads_V2, bwamem.cpp:2069-2890) collects the SeqPairs of a block of reads and hands
// them to BandedPairWiseSW::getScores8 / getScores16 / scalarBandedSWAWrapper (bandedSWA.h:126-135,199-211; call sites bwamem.cpp:2476,2541,
// 2544,2610,2613,2692,2754,2757,2825,2828).  This file IS those three member functions, with the reference's own signatures, for a build in
// which bandedSWA.cpp is compiled with the three names renamed (-DgetScores8=getScores8_reference ...): every call in bwamem.cpp then lands
// here and runs on the GPU -- seeding, chaining, the band retry rule and everything after stay the reference's host code.
// Nothing of the reference is modified or copied; its header is included from where it lies (the class keeps its parameters private; a
// maintainer would add accessors -- this file opens the class with the preprocessor instead).
//
// COMBINING (round 5).  The reference calls these from every worker thread at once, a few thousand pairs per call -- too few for the
// pair-per-lane kernel, and a synchronous copy in, launch, copy out each (rounds 3-4: one bm2_bsw per call behind a mutex; the program was 8 %
// SLOWER than the unmodified binary).  Now a call files a request and the first caller that finds a free device slot becomes the LEADER of a
// batch: it takes every request filed so far (same band and scoring), lays their pairs and sequences end to end in page-locked buffers of
// the slot (the pairs' offsets moved accordingly), runs ONE bm2_bsw, copies the six result fields back and wakes the callers.  While a batch
// is on the device the next requests pile up: batches grow with the load by themselves (group commit), no timer involved.  BM2_S1_CONTEXTS
// slots (default 2) so that the copies of one batch overlap the kernels of another.
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <vector>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define private public
#include "bandedSWA.h"
#undef private
#include "../include/bm2.h"

namespace {
enum { MAX_SLOT = 32 };
struct Request {
    SeqPair *pairs; const uint8_t *ref, *qer; int n, w; int64_t ref_bytes, qer_bytes; bm2_sw_params p; bool done;
};
struct Slot {                          // one device context with its staging buffers (page-locked: plain DMA, no second host copy inside the library)
    bm2_ctx *ctx = nullptr; bool busy = false;
    bm2_seqpair_t *pairs = nullptr; uint8_t *ref = nullptr, *qer = nullptr; size_t cap_pairs = 0, cap_ref = 0, cap_qer = 0;
};
Slot g_slot[MAX_SLOT];
int g_n = 0;
std::once_flag g_once;
std::mutex g_mu;
std::condition_variable g_cv;
std::vector<Request *> g_pending;
std::atomic<long long> g_pairs{0}, g_batches{0}, g_calls{0};
// where a call's time goes (nanoseconds summed over all calls / batches; printed at exit): a caller WAITS for a leader's batch or LEADS one, a leader
// gathers the requests into its page-locked staging buffers, calls bm2_bsw (H2D, device sort + kernels, D2H, the leader asleep), scatters the results
std::atomic<long long> g_ns_call{0}, g_ns_gather{0}, g_ns_bsw{0}, g_ns_scatter{0};
inline long long now_ns() { return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// The leader of a batch sleeps while the device works (its CPU goes to the threads that are seeding): the library reads BM2_BLOCKING_SYNC when a context
// is made.  Set at load time, while the program has ONE thread -- inside attach() it raced with the getenv of leaders on other slots (setenv may move environ).
const int g_env_set = (setenv("BM2_BLOCKING_SYNC", "1", 0), 0);

void attach() {
    const char *e = getenv("BM2_S1_CONTEXTS"), *dev = getenv("BM2_DEVICE");
    // (device slots: a call leads a batch as soon as a slot is free, and a batch is a round trip of ~0.5 ms whatever its size -- with 2 slots a call waited 2.05 ms
    //  for its results and the chunk took 6.8 s against the unmodified binary's 5.9-6.3; with 8: 0.87 ms and 5.67 s, profiles/r06r_s1_binding_device_slots.json)
    int n = e && *e ? atoi(e) : 8;
    g_n = n < 1 ? 1 : n > MAX_SLOT ? MAX_SLOT : n;
    for (int i = 0; i < g_n; i++) {
        g_slot[i].ctx = bm2_create(dev ? atoi(dev) : 0, nullptr);       // (no index: S1 needs none)
        if (!g_slot[i].ctx) { fprintf(stderr, "[bm2s1] bm2_create: %s\n", bm2_last_error()); exit(EXIT_FAILURE); }
    }
    fprintf(stderr, "[bm2s1] banded extension (getScores8 / getScores16 / scalarBandedSWAWrapper) runs in libbm2: the calls of all threads combined into batches on %d context(s)\n", g_n);
    atexit([]() {
        fprintf(stderr, "[bm2s1] %lld SeqPairs in %lld device batches from %lld calls\n", g_pairs.load(), g_batches.load(), g_calls.load());
        const double nb = (double)(g_batches.load() ? g_batches.load() : 1), nc = (double)(g_calls.load() ? g_calls.load() : 1);
        fprintf(stderr, "[bm2s1] per call %.3f ms (all threads' calls: %.2f s); per device batch: gather %.3f ms, bm2_bsw %.3f ms, scatter %.3f ms\n",
                g_ns_call.load() / nc * 1e-6, g_ns_call.load() * 1e-9, g_ns_gather.load() / nb * 1e-6, g_ns_bsw.load() / nb * 1e-6, g_ns_scatter.load() / nb * 1e-6);
    });
}

template <class T> void grow(T *&buf, size_t &cap, size_t need) {
    if (need <= cap) return;
    if (buf) bm2_host_free(buf);
    cap = need + need / 2 + 4096;
    buf = (T *)bm2_host_alloc(cap * sizeof(T));
    if (!buf) { fprintf(stderr, "[bm2s1] out of page-locked memory (%zu bytes)\n", cap * sizeof(T)); exit(EXIT_FAILURE); }
}

// the leader's part: requests laid end to end, one device batch, results back (no lock held)
void run_batch(Slot &s, const std::vector<Request *> &batch) {
    const long long t_0 = now_ns();
    size_t n = 0, rb = 0, qb = 0;
    for (const Request *r : batch) { n += (size_t)r->n; rb += (size_t)r->ref_bytes; qb += (size_t)r->qer_bytes; }
    grow(s.pairs, s.cap_pairs, n); grow(s.ref, s.cap_ref, rb + 8); grow(s.qer, s.cap_qer, qb + 8);
    size_t at = 0, ro = 0, qo = 0;
    for (const Request *r : batch) {
        memcpy(s.ref + ro, r->ref, (size_t)r->ref_bytes); memcpy(s.qer + qo, r->qer, (size_t)r->qer_bytes);
        static_assert(sizeof(SeqPair) == sizeof(bm2_seqpair_t), "SeqPair and bm2_seqpair_t are the same 56 bytes");
        memcpy(s.pairs + at, r->pairs, (size_t)r->n * sizeof(SeqPair));
        for (int i = 0; i < r->n; i++) { s.pairs[at + i].idr += (int64_t)ro; s.pairs[at + i].idq += (int64_t)qo; }
        at += (size_t)r->n; ro += (size_t)r->ref_bytes; qo += (size_t)r->qer_bytes;
    }
    // one batch: the six output fields of every pair, by the rule of the pair's own kernel class (bm2_bsw derives the class from len1, len2
    // and h0 exactly as sortPairsLenExt does, bwamem.cpp:1924-1950 -- the class the reference filed the pair under)
    const long long t_1 = now_ns();
    if (bm2_bsw(s.ctx, s.pairs, s.ref, (int64_t)rb, s.qer, (int64_t)qb, (int)n, batch[0]->w, &batch[0]->p)) {
        fprintf(stderr, "[bm2s1] bm2_bsw: %s\n", bm2_last_error()); exit(EXIT_FAILURE);
    }
    const long long t_2 = now_ns();
    at = 0;
    for (const Request *r : batch) {
        for (int i = 0; i < r->n; i++) {
            const bm2_seqpair_t &o = s.pairs[at + i];
            SeqPair &d = r->pairs[i];
            d.score = o.score; d.tle = o.tle; d.gtle = o.gtle; d.qle = o.qle; d.gscore = o.gscore; d.max_off = o.max_off;
        }
        at += (size_t)r->n;
    }
    g_pairs += (long long)n; ++g_batches;
    g_ns_gather += t_1 - t_0; g_ns_bsw += t_2 - t_1; g_ns_scatter += now_ns() - t_2;
}

void run(const BandedPairWiseSW *self, SeqPair *pairs, uint8_t *ref, uint8_t *qer, int n, int w) {
    if (n <= 0) return;
    std::call_once(g_once, attach);
    Request me; memset(&me.p, 0, sizeof me.p);
    me.pairs = pairs; me.ref = ref; me.qer = qer; me.n = n; me.w = w; me.done = false;
    me.p.o_del = self->o_del; me.p.e_del = self->e_del; me.p.o_ins = self->o_ins; me.p.e_ins = self->e_ins; me.p.zdrop = self->zdrop;
    me.p.end_bonus = self->end_bonus; me.p.w_match = self->w_match; me.p.w_mismatch = self->w_mismatch;
    memcpy(me.p.mat, self->mat, 25);
    me.ref_bytes = me.qer_bytes = 0;
    for (int i = 0; i < n; i++) {
        const int64_t r = (int64_t)pairs[i].idr + pairs[i].len1, q = (int64_t)pairs[i].idq + pairs[i].len2;
        if (r > me.ref_bytes) me.ref_bytes = r;
        if (q > me.qer_bytes) me.qer_bytes = q;
    }
    ++g_calls;
    const long long t_call = now_ns();
    std::unique_lock<std::mutex> lock(g_mu);
    g_pending.push_back(&me);
    for (;;) {
        if (me.done) { g_ns_call += now_ns() - t_call; return; }
        int free_slot = -1;
        for (int i = 0; i < g_n; i++) if (!g_slot[i].busy) { free_slot = i; break; }
        if (free_slot >= 0 && !g_pending.empty()) {              // lead a batch: everything filed so far that shares the first request's band and scoring
            std::vector<Request *> batch, rest;
            const Request *lead = g_pending[0];
            size_t pairs_in = 0, ref_in = 0, qer_in = 0;          // (SeqPair's offsets are 32-bit: a batch stays below 2^30 bytes of either sequence buffer)
            for (Request *r : g_pending) {
                const bool same = r->w == lead->w && !memcmp(&r->p, &lead->p, sizeof r->p);
                // (the FIRST request of a batch is rebased by zero: its own 32-bit offsets hold whatever its size; only what joins it must stay below 2^30)
                const bool fits = batch.empty() || (pairs_in + (size_t)r->n < ((size_t)1 << 28) && ref_in + (size_t)r->ref_bytes < ((size_t)1 << 30) &&
                                                    qer_in + (size_t)r->qer_bytes < ((size_t)1 << 30));
                if (same && fits) { batch.push_back(r); pairs_in += (size_t)r->n; ref_in += (size_t)r->ref_bytes; qer_in += (size_t)r->qer_bytes; } else rest.push_back(r);
            }
            g_pending.swap(rest);
            Slot &s = g_slot[free_slot];
            s.busy = true;
            lock.unlock();
            run_batch(s, batch);
            lock.lock();
            s.busy = false;
            for (Request *r : batch) r->done = true;
            g_cv.notify_all();
            continue;                                            // (mine may have been in another leader's batch, or still be pending)
        }
        g_cv.wait(lock);
    }
}
}  // namespace

void BandedPairWiseSW::getScores8(SeqPair *pairArray, uint8_t *seqBufRef, uint8_t *seqBufQer, int32_t numPairs, uint16_t, int32_t w) {
    run(this, pairArray, seqBufRef, seqBufQer, numPairs, w);
}
void BandedPairWiseSW::getScores16(SeqPair *pairArray, uint8_t *seqBufRef, uint8_t *seqBufQer, int32_t numPairs, uint16_t, int32_t w) {
    run(this, pairArray, seqBufRef, seqBufQer, numPairs, w);
}
void BandedPairWiseSW::scalarBandedSWAWrapper(SeqPair *seqPairArray, uint8_t *seqBufRef, uint8_t *seqBufQer, int numPairs, int, int32_t w) {
    run(this, seqPairArray, seqBufRef, seqBufQer, numPairs, w);
}
