// integration/bm2_bsw_binding.cpp -- seam S1 of libbm2 compiled INTO bwa-mem2 (oracle/Makefile, target `bm2s1`): BASELINE.json config 2,
// "banded-SW HIP kernel only, FM-index seeding still on host".
//
// The reference's extension stage (mem_chain2aln_across_reads_V2, bwamem.cpp:2069-2890) collects the SeqPairs of a block of reads and hands
// them to BandedPairWiseSW::getScores8 / getScores16 / scalarBandedSWAWrapper (bandedSWA.h:126-135,199-211; call sites bwamem.cpp:2476,2541,
// 2544,2610,2613,2692,2754,2757,2825,2828).  This file IS those three member functions, with the reference's own signatures, for a build in
// which bandedSWA.cpp is compiled with the three names renamed (-DgetScores8=getScores8_reference ...): every call in bwamem.cpp then lands
// here and runs as one bm2_bsw batch on the GPU -- seeding, chaining, the band retry rule and everything after stay the reference's host code.
// Nothing of the reference is modified or copied; its header is included from where it lies (the class keeps its parameters private; a
// maintainer would add accessors -- this file opens the class with the preprocessor instead).
//
// The reference calls these from every worker thread at once (one BandedPairWiseSW per call of mem_chain2aln_across_reads_V2); a bm2_ctx
// serves one host thread at a time, so the calls are spread over BM2_S1_CONTEXTS contexts (default 4) on device BM2_DEVICE, each behind a
// mutex: the copies of one thread's batch overlap the kernel of another's.
#include <atomic>
#include <mutex>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define private public
#include "bandedSWA.h"
#undef private
#include "../include/bm2.h"

namespace {
enum { MAX_CTX = 16 };
struct Slot { bm2_ctx *ctx = nullptr; std::mutex mu; };
Slot g_slot[MAX_CTX];
int g_n = 0;
std::once_flag g_once;
std::atomic<unsigned> g_next{0};
std::atomic<long long> g_pairs{0}, g_calls{0};

void attach() {
    const char *e = getenv("BM2_S1_CONTEXTS"), *dev = getenv("BM2_DEVICE");
    int n = e && *e ? atoi(e) : 4;
    g_n = n < 1 ? 1 : n > MAX_CTX ? MAX_CTX : n;
    for (int i = 0; i < g_n; i++) {
        g_slot[i].ctx = bm2_create(dev ? atoi(dev) : 0, nullptr);       // (no index: S1 needs none)
        if (!g_slot[i].ctx) { fprintf(stderr, "[bm2s1] bm2_create: %s\n", bm2_last_error()); exit(EXIT_FAILURE); }
    }
    fprintf(stderr, "[bm2s1] banded extension (getScores8 / getScores16 / scalarBandedSWAWrapper) runs in libbm2 on %d context(s)\n", g_n);
    atexit([]() { fprintf(stderr, "[bm2s1] %lld SeqPairs in %lld device batches\n", g_pairs.load(), g_calls.load()); });
}

// one batch: the six output fields of every pair, by the rule of the pair's own kernel class (bm2_bsw derives the class from len1, len2
// and h0 exactly as sortPairsLenExt does, bwamem.cpp:1924-1950 -- the class the reference filed the pair under)
void run(const BandedPairWiseSW *self, SeqPair *pairs, uint8_t *ref, uint8_t *qer, int n, int w) {
    if (n <= 0) return;
    std::call_once(g_once, attach);
    static_assert(sizeof(SeqPair) == sizeof(bm2_seqpair_t), "SeqPair and bm2_seqpair_t are the same 56 bytes");
    bm2_sw_params p; memset(&p, 0, sizeof p);
    p.o_del = self->o_del; p.e_del = self->e_del; p.o_ins = self->o_ins; p.e_ins = self->e_ins; p.zdrop = self->zdrop;
    p.end_bonus = self->end_bonus; p.w_match = self->w_match; p.w_mismatch = self->w_mismatch;
    memcpy(p.mat, self->mat, 25);
    int64_t ref_bytes = 0, qer_bytes = 0;
    for (int i = 0; i < n; i++) {
        const int64_t r = (int64_t)pairs[i].idr + pairs[i].len1, q = (int64_t)pairs[i].idq + pairs[i].len2;
        if (r > ref_bytes) ref_bytes = r;
        if (q > qer_bytes) qer_bytes = q;
    }
    Slot &s = g_slot[g_next.fetch_add(1) % (unsigned)g_n];
    std::lock_guard<std::mutex> lock(s.mu);
    if (bm2_bsw(s.ctx, (bm2_seqpair_t *)pairs, ref, ref_bytes, qer, qer_bytes, n, w, &p)) {
        fprintf(stderr, "[bm2s1] bm2_bsw: %s\n", bm2_last_error()); exit(EXIT_FAILURE);
    }
    g_pairs += n; ++g_calls;
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
