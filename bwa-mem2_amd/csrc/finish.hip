// finish.hip -- the tail of mem_kernel2_core on the device: mem_sort_dedup_patch (bwamem.cpp:292-353) with the merge test of
// mem_patch_reg (:175-225, whose score is a banded global alignment: bwa_gen_cigar2 -> ksw_global2 without backtrack,
// bwa.cpp:260-347, ksw.cpp:558-668) and the ALT flag (:1161-1169).  After this stage the device holds exactly the mem_alnreg_v
// contents the reference hands to worker_sam.
//
// Shape.  The walk over a read's hits is sequential by definition (a merge rewrites the hit the next comparison reads), short
// (1.8 hits per 150-base read), and now and then needs one global alignment whose result decides how it goes on.  So:
//   * k_fin_walk: ONE READ PER LANE, resumable.  A lane sorts its hits by reference end (klib's introsort on an index array: its
//     order among equal keys is observable), walks them, and when a pair passes the cheap colinearity tests it files a REQUEST for
//     the alignment score, stores where it stands (i, j) and leaves; at its next launch it picks up the score and continues.  When
//     the walk is through it drops excluded hits, sorts by (score, rb, qb), drops identical hits and reports its count.
//   * k_fin_dp: ONE REQUEST PER WAVEFRONT, lanes = columns of a DP row.  ksw_global2 takes the insertion state from M, not from H
//     (ksw.cpp:630-633), so F along a row is a max-plus prefix scan (6 DPP steps); H/E of the band live in a per-wave LDS ring of
//     R >= 2w+4 slots indexed by column mod R, which keeps a 30 kb query inside a few KB.  Reverse-strand hits are aligned
//     back to front by index arithmetic (the reference reverses its copies, bwa.cpp:277-282: the band is anchored at the start, so
//     the direction matters).
//   * the host alternates the two until no request is left (150-base reads: one or two rounds), then a scan and k_fin_gather lay
//     the survivors out densely in read order.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "pipeline.h"
#include "bsw_dev.h"
#include "ksort_dev.h"

#define FIN_MINUS_INF (-0x40000000)
#define FIN_SCAN_ID   (-0x60000000)              // identity of the max-scans: below every value the DP can hold
#define PATCH_MAX_R_BW 0.05f
#define PATCH_MIN_SC_RATIO 0.90f

struct FinParams {
    int32_t w, max_chain_gap, o_del, e_del, o_ins, e_ins, mat0;
    float mask_level_redun;
    int8_t mat[25]; int8_t pad[3];
};
struct FinState { int32_t i, j, phase, dp_score; };             // phase: 0 = not started, 1 = walking, 2 = waiting for a score, 3 = done
struct FinReq { int32_t read, a, b, w; };                       // hits a, b (indices into the read's work slice), input band of bwa_gen_cigar2

// the band of ksw_global2 as bwa_gen_cigar2 sets it (bwa.cpp:294-303)
static __device__ __forceinline__ int fin_band(int l_query, int rlen, int w_, const FinParams &P) {
    int max_ins = (int)((double)(((l_query + 1) >> 1) * P.mat0 - P.o_ins) / P.e_ins + 1.);
    int max_del = (int)((double)(((l_query + 1) >> 1) * P.mat0 - P.o_del) / P.e_del + 1.);
    int max_gap = max_ins > max_del ? max_ins : max_del;
    max_gap = max_gap > 1 ? max_gap : 1;
    const int dl = rlen > l_query ? rlen - l_query : l_query - rlen;
    int w = (max_gap + dl + 1) >> 1;
    w = w < w_ ? w : w_;
    const int min_w = dl + 3;
    return w > min_w ? w : min_w;
}

// the two orders of the stage as out-of-line functions (one copy of klib's introsort each, called, not inlined into the kernels)
struct FinByEnd { const bm2_alnreg_t *A; __device__ bool operator()(int32_t x, int32_t y) const { return A[x].re < A[y].re; } };
struct FinByScore {
    const bm2_alnreg_t *A;
    __device__ bool operator()(int32_t x, int32_t y) const {
        const bm2_alnreg_t &a = A[x], &b = A[y];
        return a.score > b.score || (a.score == b.score && (a.rb < b.rb || (a.rb == b.rb && a.qb < b.qb)));
    }
};
#ifdef BM2_FIN_NOINLINE
#define FIN_SORT_ATTR __noinline__
#else
#define FIN_SORT_ATTR
#endif
static __device__ FIN_SORT_ATTR void fin_sort_by_end(int n, int32_t *ord, const bm2_alnreg_t *A) { FinByEnd lt = { A }; k_introsort_flat(n, ord, lt); }
static __device__ FIN_SORT_ATTR void fin_sort_by_score(int n, int32_t *ord, const bm2_alnreg_t *A) { FinByScore lt = { A }; k_introsort_flat(n, ord, lt); }

// first kernel of the stage: the hits as mem_kernel2_core holds them at bwamem.cpp:1152 (calloc'd: every other field is 0), ordered
// by reference end ("sort by the END position", :299)
__global__ void __launch_bounds__(128)
k_fin_init(int n_reads, const bm2_reg_t *__restrict__ regs, const int64_t *__restrict__ reg_off, bm2_alnreg_t *work, int32_t *ordbuf,
           FinState *state, int32_t *n_fin) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    const int64_t base = reg_off[r];
    const int n = (int)(reg_off[r + 1] - base);
    bm2_alnreg_t *A = work + base;
    int32_t *ord = ordbuf + base;
    for (int i = 0; i < n; i++) {
        const bm2_reg_t s = regs[base + i];
        bm2_alnreg_t d; memset(&d, 0, sizeof d);
        d.rb = s.rb; d.re = s.re; d.qb = s.qb; d.qe = s.qe; d.rid = s.rid; d.score = s.score; d.truesc = s.truesc; d.w = s.w;
        d.seedcov = s.seedcov; d.seedlen0 = s.seedlen0; d.frac_rep = s.frac_rep;
        d.n_comp = n > 1 ? 1 : 0;                                // bwamem.cpp:298-300 (a single hit returns before n_comp is set)
        A[i] = d; ord[i] = i;
    }
    FinState st; st.i = 1; st.j = -2; st.phase = n <= 1 ? 3 : 1; st.dp_score = 0;
    state[r] = st;
    n_fin[r] = n <= 1 ? n : 0;
    if (n > 1) fin_sort_by_end(n, ord, A);
}

// the walk, resumable (phase 1 = walking, 2 = a score has arrived, 3 = through, 4 = walk done, final ordering pending)
__global__ void __launch_bounds__(128)
k_fin_walk(DevIndex ix, FinParams P, int n_reads, const int64_t *__restrict__ reg_off, bm2_alnreg_t *work, const int32_t *__restrict__ ordbuf,
           FinState *state, FinReq *reqs, unsigned long long *cnt /* [0] requests, [1] widest band */) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    FinState st = state[r];
    if (st.phase >= 3) return;
    const int64_t base = reg_off[r];
    const int n = (int)(reg_off[r + 1] - base);
    bm2_alnreg_t *A = work + base;
    const int32_t *ord = ordbuf + base;
    // ---- the walk, bwamem.cpp:302-335, as ONE loop over (i, j) pairs (no nested loops, no early exits: a lane that must wait for
    // an alignment score parks and falls through to the end of the kernel).  j == -2: hit i has not been looked at yet; on entry
    // with phase 2, (i, j) is the pair whose score has just arrived.
    int i = st.i, j = st.j;
    bool have_score = st.phase == 2, parked = false;
    while (i < n && !parked) {
        bm2_alnreg_t *p = &A[ord[i]];
        bool next_i = false;
        if (j == -2) {                                          // first look at hit i: anything before it close enough?
            const bm2_alnreg_t *pr = &A[ord[i - 1]];
            if (p->rid != pr->rid || p->rb >= pr->re + P.max_chain_gap) next_i = true;
            else j = i - 1;
        }
        if (!next_i) {
            bm2_alnreg_t *q = &A[ord[j >= 0 ? j : 0]];
            if (j < 0 || !(p->rid == q->rid && p->rb < q->re + P.max_chain_gap)) next_i = true;
            else if (have_score) {
                // ---- back with the score: the rest of mem_patch_reg (:214-224) and the merge (:320-332)
                have_score = false;
                int wq = (int)((q->re - p->rb) - (q->qe - p->qb));
                wq = wq > 0 ? wq : -wq;
                wq += q->w + p->w;
                wq = wq < P.w << 2 ? wq : P.w << 2;
                const int score = st.dp_score;
                // (the reference's AVX-512 / AVX2 builds contract x / y * z + .499 into one fused multiply-add; so does this)
                const int q_s = (int)fma((double)(p->qe - q->qb) / (double)((p->qe - p->qb) + (q->qe - q->qb)), (double)(p->score + q->score), .499);
                const int r_s = (int)fma((double)(p->re - q->rb) / (double)((p->re - p->rb) + (q->re - q->rb)), (double)(p->score + q->score), .499);
                if (!((double)score / (double)(q_s > r_s ? q_s : r_s) < PATCH_MIN_SC_RATIO) && score > 0) {
                    p->n_comp += q->n_comp + 1;
                    p->seedcov = p->seedcov > q->seedcov ? p->seedcov : q->seedcov;
                    p->sub = p->sub > q->sub ? p->sub : q->sub;
                    p->csub = p->csub > q->csub ? p->csub : q->csub;
                    p->qb = q->qb; p->rb = q->rb;
                    p->truesc = p->score = score;
                    p->w = wq;
                    q->qb = q->qe;
                }
                --j;
            } else if (q->qe == q->qb) --j;                     // excluded earlier
            else {
                const int64_t orr = q->re - p->rb;
                const int64_t oq = q->qb < p->qb ? q->qe - p->qb : p->qe - q->qb;
                const int64_t mr = q->re - q->rb < p->re - p->rb ? q->re - q->rb : p->re - p->rb;
                const int64_t mq = q->qe - q->qb < p->qe - p->qb ? q->qe - q->qb : p->qe - p->qb;
                if (orr > P.mask_level_redun * mr && oq > P.mask_level_redun * mq) {          // one of the two is redundant
                    if (p->score < q->score) { p->qe = p->qb; next_i = true; }
                    else { q->qe = q->qb; --j; }
                } else {
                    // mem_patch_reg(a = q, b = p) up to the alignment, bwamem.cpp:186-205
                    bool cand = q->rb < p->rb && !(q->rb < ix.l_pac && p->rb >= ix.l_pac) && !(q->qb >= p->qb || q->qe >= p->qe || q->re >= p->re);
                    int w = 0;
                    if (cand) {
                        w = (int)((q->re - p->rb) - (q->qe - p->qb));
                        w = w > 0 ? w : -w;
                        double rr = (double)(q->re - p->rb) / (double)(p->re - q->rb) - (double)(q->qe - p->qb) / (double)(p->qe - q->qb);
                        rr = rr > 0. ? rr : -rr;
                        if (q->re < p->rb || q->qe < p->qb) { if (w > P.w << 1 || rr >= PATCH_MAX_R_BW) cand = false; }
                        else if (w > P.w << 2 || rr >= PATCH_MAX_R_BW * 2) cand = false;
                    }
                    if (!cand) --j;
                    else {                                      // the score of the global alignment decides: file the request and park
                        w += q->w + p->w;
                        w = w < P.w << 2 ? w : P.w << 2;
                        const unsigned long long at = atomicAdd(&cnt[0], 1ULL);
                        FinReq rq; rq.read = r; rq.a = ord[j]; rq.b = ord[i]; rq.w = w;
                        reqs[at] = rq;                          // (at most one per read and round: the queue holds n_reads)
                        const int lq = p->qe - q->qb; const int64_t rl = p->re - q->rb;
                        const int band = rl <= 0x3fffffff ? fin_band(lq, (int)rl, w, P) : 0x3fffffff;
                        atomicMax(&cnt[1], (unsigned long long)band);
                        st.i = i; st.j = j; st.phase = 2; st.dp_score = 0;
                        parked = true;
                    }
                }
            }
        }
        if (next_i) { ++i; j = -2; }
    }
    if (!parked) st.phase = 4;
    state[r] = st;
}

// last kernel: bwamem.cpp:336-352
__global__ void __launch_bounds__(128)
k_fin_order(int n_reads, const int64_t *__restrict__ reg_off, bm2_alnreg_t *work, int32_t *ordbuf, FinState *state, int32_t *n_fin) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    FinState st = state[r];
    if (st.phase != 4) return;
    const int64_t base = reg_off[r];
    const int n = (int)(reg_off[r + 1] - base);
    bm2_alnreg_t *A = work + base;
    int32_t *ord = ordbuf + base;
    // ---- bwamem.cpp:336-352: drop the excluded, order by (score desc, rb, qb), drop identical hits
    int m = 0;
    for (int k = 0; k < n; k++) if (A[ord[k]].qe > A[ord[k]].qb) ord[m++] = ord[k];
    fin_sort_by_score(m, ord, A);
    for (int k = 1; k < m; k++) {
        bm2_alnreg_t &a = A[ord[k]]; const bm2_alnreg_t &b = A[ord[k - 1]];
        if (a.score == b.score && a.rb == b.rb && a.qb == b.qb) a.qe = a.qb;
    }
    int m2 = m > 0 ? 1 : 0;
    for (int k = 1; k < m; k++) if (A[ord[k]].qe > A[ord[k]].qb) ord[m2++] = ord[k];
    n_fin[r] = m2;
    st.phase = 3;
    state[r] = st;
}

// ---- ksw_global2 without backtrack (ksw.cpp:558-668) for one request on one wavefront.  q / t are walked with strides qs / ts.
static __device__ int fin_global_score(const uint8_t *qp, int qs, int qlen, RefPtr tp, int ts, int tlen, int w, const FinParams &P,
                                       int *RH, int *RE, int RM) {
    const int lane = threadIdx.x & 63;
    const int o_del = P.o_del, e_del = P.e_del, o_ins = P.o_ins, e_ins = P.e_ins, oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
    int maxEnd = -1;                                            // columns <= maxEnd have been stored in the ring
    int tchunk = 4;
    for (int i = 0; i < tlen; ++i) {
        if ((i & 63) == 0) { const int ti = i + lane; tchunk = ti < tlen ? (int)tp[(int64_t)ti * ts] : 4; }
        const int tb = __builtin_amdgcn_readlane(tchunk, i & 63);
        const int s0 = P.mat[tb * 5 + 0], s1 = P.mat[tb * 5 + 1], s2 = P.mat[tb * 5 + 2], s3 = P.mat[tb * 5 + 3], s4 = P.mat[tb * 5 + 4];
        const int beg = i > w ? i - w : 0;                     // :619-620
        const int end = i + w + 1 < qlen ? i + w + 1 : qlen;
        int hcarry = beg == 0 ? -(o_del + e_del * (i + 1)) : FIN_MINUS_INF;     // h1, :621
        int fc = FIN_MINUS_INF;                                 // f entering the first column of a chunk
        for (int jb = beg; jb <= end; jb += 64) {
            const int j = jb + lane;
            const bool act = j < end, st = j <= end;
            int Hd = FIN_MINUS_INF, E = FIN_MINUS_INF;
            if (st) {
                if (j <= maxEnd) { Hd = RH[j & RM]; E = RE[j & RM]; }
                else Hd = j == 0 ? 0 : (j <= w ? -(o_ins + e_ins * j) : FIN_MINUS_INF);       // first row, :587-591
            }
            const int qb = act ? (int)qp[(int64_t)j * qs] : 4;
            const int sc = qb == 0 ? s0 : qb == 1 ? s1 : qb == 2 ? s2 : qb == 3 ? s3 : s4;
            const int m = act ? Hd + sc : FIN_SCAN_ID;
            const int U = act ? m - oe_ins + lane * e_ins : FIN_SCAN_ID;
            const int Pm = wave_scan_max(U, FIN_SCAN_ID);
            const int Pprev = wave_shr1(Pm, FIN_SCAN_ID);
            const int F = imax(fc - lane * e_ins, Pprev - (lane - 1) * e_ins);
            const int h = act ? imax(imax(m, E), F) : FIN_MINUS_INF;
            const int hs = wave_shr1(h, hcarry);                // H(i, j-1): what eh[j].h holds for the next row
            const int en = act ? imax(E - e_del, m - oe_del) : FIN_MINUS_INF;
            if (st) { RH[j & RM] = hs; RE[j & RM] = en; }       // (j == end: eh[end] = { h1, -inf }, :636)
            const int nact = imin(64, end - jb);
            if (nact > 0) {
                hcarry = __builtin_amdgcn_readlane(h, nact - 1);
                fc = imax(fc - 64 * e_ins, __builtin_amdgcn_readlane(Pm, 63) - 63 * e_ins);
            }
        }
        maxEnd = imax(maxEnd, end);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    int res = qlen <= maxEnd ? RH[qlen & RM] : (qlen == 0 ? 0 : (qlen <= w ? -(o_ins + e_ins * qlen) : FIN_MINUS_INF));
    return __builtin_amdgcn_readfirstlane(res);
}

__global__ void __launch_bounds__(256)
k_fin_dp(DevIndex ix, FinParams P, const FinReq *__restrict__ reqs, const unsigned long long *__restrict__ cnt, const uint8_t *__restrict__ enc,
         const int64_t *__restrict__ off, const int64_t *__restrict__ reg_off, const bm2_alnreg_t *__restrict__ work, FinState *state, int R) {
    extern __shared__ __attribute__((aligned(16))) int fin_lds[];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int *RH = fin_lds + (size_t)wv * 2 * R, *RE = RH + R;
    const long long n_req = (long long)cnt[0];
    const long long idx = (long long)blockIdx.x * (blockDim.x >> 6) + wv;
    if (idx >= n_req) return;
    const FinReq rq = reqs[idx];
    const int64_t base = reg_off[rq.read];
    const bm2_alnreg_t a = work[base + rq.a], b = work[base + rq.b];
    // bwa_gen_cigar2(w_ = rq.w, l_query = b.qe - a.qb, query + a.qb, rb = a.rb, re = b.re), bwa.cpp:260-310
    const int lq = b.qe - a.qb;
    const int64_t rb = a.rb, re = b.re;
    int score = 0;                                              // a rejected range leaves the caller's score alone: no merge
    if (lq > 0 && rb < re && !(rb < ix.l_pac && re > ix.l_pac) && rb >= 0 && re <= (ix.l_pac << 1) && re - rb <= 0x3fffffff) {
        const int rlen = (int)(re - rb);
        const bool rev = rb >= ix.l_pac;                        // then both are walked back to front (:277-282)
        const uint8_t *q0 = enc + off[rq.read] + a.qb;
        const uint8_t *qp = rev ? q0 + lq - 1 : q0;
        const RefPtr tp = ix.ref(rev ? re - 1 : rb);
        const int sd = rev ? -1 : 1;
        if (lq == rlen && rq.w == 0) {                          // no gap possible (:283-293)
            int sc = 0;
            for (int k = lane; k < lq; k += 64) sc += P.mat[(int)tp[(int64_t)k * sd] * 5 + (int)qp[(int64_t)k * sd]];
            for (int o = 32; o > 0; o >>= 1) sc += __shfl_xor(sc, o);
            score = sc;
        } else {
            const int w = fin_band(lq, rlen, rq.w, P);
            score = fin_global_score(qp, sd, lq, tp, sd, rlen, w, P, RH, RE, R - 1);
        }
    }
    if (lane == 0) state[rq.read].dp_score = score;
}

__global__ void __launch_bounds__(256)
k_fin_gather(DevIndex ix, int n_reads, const int64_t *__restrict__ reg_off, const bm2_alnreg_t *__restrict__ work, const int32_t *__restrict__ ordbuf,
             const int32_t *__restrict__ n_fin, const int64_t *__restrict__ fin_off, bm2_alnreg_t *out) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    const int64_t base = reg_off[r], o = fin_off[r];
    const int m = n_fin[r];
    for (int k = 0; k < m; k++) {
        bm2_alnreg_t a = work[base + ordbuf[base + k]];
        if (a.rid >= 0 && ix.ann_is_alt[a.rid]) a.is_alt = 1;   // bwamem.cpp:1161-1169
        out[o + k] = a;
    }
}

// device buffers of the stage (owned by the batch, pipeline.hip)
int bm2_run_finish(bm2_ctx *c, const bm2_opt *opt, int n_reads, const uint8_t *enc, const int64_t *off, const bm2_reg_t *regs,
                   const int64_t *reg_off, int64_t n_regs, DevBuf &work, DevBuf &ordb, DevBuf &stateb, DevBuf &nfin, DevBuf &finoff, DevBuf &reqb,
                   DevBuf &cntb, DevBuf &out, DevBuf &scan_tmp, int64_t *n_out, int *rounds) {
    hipStream_t s = c->stream;
    int rc;
    *n_out = 0; if (rounds) *rounds = 0;
    if ((rc = bm2_reserve(finoff, (size_t)(n_reads + 2) * 8))) return rc;
    if (n_reads == 0) return bm2_check(hipMemsetAsync(finoff.p, 0, 16, s), "memset");
    if ((rc = bm2_reserve(work, (size_t)(n_regs + 1) * sizeof(bm2_alnreg_t)))) return rc;
    if ((rc = bm2_reserve(ordb, (size_t)(n_regs + 1) * 4))) return rc;
    if ((rc = bm2_reserve(stateb, (size_t)(n_reads + 1) * sizeof(FinState)))) return rc;
    if ((rc = bm2_reserve(nfin, (size_t)(n_reads + 1) * 4))) return rc;
    if ((rc = bm2_reserve(reqb, (size_t)(n_reads + 1) * sizeof(FinReq)))) return rc;
    if ((rc = bm2_reserve(cntb, 64))) return rc;
    FinParams P; memset(&P, 0, sizeof P);
    P.w = opt->w; P.max_chain_gap = opt->max_chain_gap; P.o_del = opt->o_del; P.e_del = opt->e_del; P.o_ins = opt->o_ins; P.e_ins = opt->e_ins;
    P.mat0 = opt->mat[0]; P.mask_level_redun = opt->mask_level_redun;
    for (int i = 0; i < 25; i++) P.mat[i] = opt->mat[i];
    const unsigned nb = (unsigned)((n_reads + 127) / 128);
    const bool verbose = getenv("BM2_FIN_VERBOSE") != nullptr;
    // (the lane-per-read kernels below with the reads in classes of hit count, as the chaining takes them: 4.7 instead of 2.5 ms per chunk of the FASTQ -> SAM
    //  leg, profiles/r06ah_* -- a read's hits are two or three records next to its neighbours'; not kept)
    hipLaunchKernelGGL(k_fin_init, dim3(nb), dim3(128), 0, s, n_reads, regs, reg_off, (bm2_alnreg_t *)work.p, (int32_t *)ordb.p, (FinState *)stateb.p,
                       (int32_t *)nfin.p);
    if (verbose) { rc = bm2_check(hipStreamSynchronize(s), "k_fin_init"); fprintf(stderr, "[finish] init + sort by end: rc %d\n", rc); if (rc) return rc; }
    for (int round = 0; ; round++) {
        if ((rc = bm2_check(hipMemsetAsync(cntb.p, 0, 16, s), "memset fin counters"))) return rc;
        hipLaunchKernelGGL(k_fin_walk, dim3(nb), dim3(128), 0, s, c->ix, P, n_reads, reg_off, (bm2_alnreg_t *)work.p, (const int32_t *)ordb.p,
                           (FinState *)stateb.p, (FinReq *)reqb.p, (unsigned long long *)cntb.p);
        unsigned long long h_cnt[2] = { 0, 0 };
        if ((rc = bm2_check(hipMemcpyAsync(h_cnt, cntb.p, 16, hipMemcpyDeviceToHost, s), "D2H fin counters"))) return rc;
        if ((rc = bm2_check(hipStreamSynchronize(s), "k_fin_walk"))) return rc;
        if (rounds) *rounds = round + 1;
        if (verbose) fprintf(stderr, "[finish] round %d: %llu alignment requests, widest band %llu\n", round, h_cnt[0], h_cnt[1]);
        if (h_cnt[0] == 0) break;
        // the alignments of this round: one per wavefront, LDS ring of R slots per wave for the widest band
        int R = 64;
        while (R < 2 * (long long)h_cnt[1] + 4 && R < (1 << 20)) R <<= 1;
        int waves = 4;
        while (waves > 1 && (size_t)waves * 2 * R * 4 > 64 * 1024) waves >>= 1;
        const size_t lds = (size_t)waves * 2 * R * 4;
        if (lds > 160 * 1024) { bm2_set_error("hit merging: a band of %llu columns needs more LDS than a CU has", h_cnt[1]); return BM2_EUNSUP; }
        if (lds > 64 * 1024 && (rc = bm2_check(hipFuncSetAttribute((const void *)k_fin_dp, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), "hipFuncSetAttribute(k_fin_dp)"))) return rc;
        const unsigned nbd = (unsigned)((h_cnt[0] + waves - 1) / waves);
        hipLaunchKernelGGL(k_fin_dp, dim3(nbd), dim3(64 * waves), lds, s, c->ix, P, (const FinReq *)reqb.p, (const unsigned long long *)cntb.p, enc, off,
                           reg_off, (const bm2_alnreg_t *)work.p, (FinState *)stateb.p, R);
        if ((rc = bm2_check(hipGetLastError(), "k_fin_dp launch"))) return rc;
    }
    hipLaunchKernelGGL(k_fin_order, dim3(nb), dim3(128), 0, s, n_reads, reg_off, (bm2_alnreg_t *)work.p, (int32_t *)ordb.p, (FinState *)stateb.p,
                       (int32_t *)nfin.p);
    if (verbose) { rc = bm2_check(hipStreamSynchronize(s), "k_fin_order"); fprintf(stderr, "[finish] final order: rc %d\n", rc); if (rc) return rc; }
    if ((rc = bm2_scan_i32(c, (const int32_t *)nfin.p, n_reads, (int64_t *)finoff.p, scan_tmp))) return rc;
    int64_t tot = 0;
    if ((rc = bm2_check(hipMemcpyAsync(&tot, (int64_t *)finoff.p + n_reads, 8, hipMemcpyDeviceToHost, s), "D2H n_fin"))) return rc;
    if ((rc = bm2_check(hipStreamSynchronize(s), "fin scan"))) return rc;
    if ((rc = bm2_reserve(out, (size_t)(tot + 1) * sizeof(bm2_alnreg_t)))) return rc;
    hipLaunchKernelGGL(k_fin_gather, dim3((n_reads + 255) / 256), dim3(256), 0, s, c->ix, n_reads, reg_off, (const bm2_alnreg_t *)work.p,
                       (const int32_t *)ordb.p, (const int32_t *)nfin.p, (const int64_t *)finoff.p, (bm2_alnreg_t *)out.p);
    if ((rc = bm2_check(hipGetLastError(), "k_fin_gather launch"))) return rc;
    *n_out = tot;
    return BM2_OK;
}
