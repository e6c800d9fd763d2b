// cigar.hip -- CIGAR / NM / MD of a batch of hits on the device: bwa_gen_cigar2 (bwa.cpp:260-347) = ksw_global2 with backtrack
// (ksw.cpp:558-668), SURVEY.md 8(f) row 2.  One task per lane: the banded global alignment of a 150-base read against its
// reference range is a few thousand cells, there are a few per read, and nothing in it crosses lanes -- each lane walks its own
// rows, keeps its direction bytes and its row of (H, E) in its own slice of a scratch buffer, backtracks, and writes its CIGAR and
// MD string into its own slice of the output.  Targets are read from the context's resident ref_string (reversed in place by index
// arithmetic for hits on the reverse strand, as the reference reverses its copies so that gaps end up leftmost on the forward
// strand).  Host oracle: gen_cigar / global_align in sam_tail.cpp (bm2_gen_cigar), pinned against the reference's bwa_gen_cigar2.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>
#include "../../include/bm2.h"
#include "bm2_ctx.h"
#include "host_tail.h"
#include "host_pool.h"
#include "pipeline.h"

#define CG_MINUS_INF (-0x40000000)

struct CigarPrm { int8_t mat[25]; int8_t pad[3]; int32_t o_del, e_del, o_ins, e_ins, a, w; int64_t l_pac; };
struct CigarTask {
    int64_t q_off, rb, re;
    int64_t z_off, eh_off, cg_off, md_off;      // this task's slices of the scratch buffers
    int32_t q_len, w, truesc, retry;            // retry = 1: w is the hit's band and the kernel runs mem_reg2aln's retry loop around the alignment
};
struct CigarRes { int32_t score, nm, n_cigar, md_len; };

// the band of ksw_global2 as bwa_gen_cigar2 sets it (bwa.cpp:289-301); host and device size and run with the same number
static __host__ __device__ inline int cigar_band(int l_query, int rlen, int w_, int mat0, int o_del, int e_del, int o_ins, int e_ins) {
    int max_ins = (int)((double)(((l_query + 1) >> 1) * mat0 - o_ins) / e_ins + 1.);
    int max_del = (int)((double)(((l_query + 1) >> 1) * mat0 - o_del) / e_del + 1.);
    int max_gap = max_ins > max_del ? max_ins : max_del;
    max_gap = max_gap > 1 ? max_gap : 1;
    const int dl = rlen > l_query ? rlen - l_query : l_query - rlen;
    int w = (max_gap + dl + 1) >> 1;
    w = w < w_ ? w : w_;
    const int min_w = dl + 3;
    return w > min_w ? w : min_w;
}
static __host__ __device__ inline bool cigar_range_ok(int64_t l_pac, int l_query, int64_t rb, int64_t re) {
    if (l_query <= 0 || rb >= re || (rb < l_pac && re > l_pac)) return false;
    return rb >= 0 && re <= (l_pac << 1);                       // bns_get_seq would clamp: a clamped range is the NULL return
}
// infer_bw, bwamem.cpp:1811-1818, and the band of mem_reg2aln's first try (:1743-1747)
static __host__ __device__ inline int cigar_infer_bw(int l1, int l2, int score, int a, int q, int r) {
    if (l1 == l2 && l1 * a - score < (q + r - a) << 1) return 0;
    int w = (int)((double)((l1 < l2 ? l1 : l2) * a - score - q) / r + 2.);
    const int d = l1 > l2 ? l1 - l2 : l2 - l1;
    return w < d ? d : w;
}
static __host__ __device__ inline int cigar_first_band(int lq, int rlen, int truesc, int w_hit, int a, int w_opt, int o_del, int e_del, int o_ins, int e_ins) {
    const int t = cigar_infer_bw(lq, rlen, truesc, a, o_del, e_del);
    int w2 = cigar_infer_bw(lq, rlen, truesc, a, o_ins, e_ins);
    w2 = w2 > t ? w2 : t;
    if (w2 > w_opt) w2 = w2 < w_hit ? w2 : w_hit;
    return w2;
}

// kputw: plain decimal of v >= 0.  Digits by CONSTANT divisors, most significant first (the digit buffer of the obvious form -- fill backwards, copy
// forwards -- is a local array under a run-time index: the compiler kept its twelve bytes in registers behind a select chain, ~60 VALU instructions per
// digit, in every inlined copy)
static __device__ int put_dec(char *s, int n, int v) {
    bool lead = false;
#define PUT_DIGIT(P) { const int d = v / (P); if (lead || d) { s[n++] = (char)('0' + d); v -= d * (P); lead = true; } }
    if (v >= 100000) { PUT_DIGIT(1000000000) PUT_DIGIT(100000000) PUT_DIGIT(10000000) PUT_DIGIT(1000000) PUT_DIGIT(100000) }      // (MD run lengths and CIGAR lengths of reads below 32768 bases never come here)
    PUT_DIGIT(10000) PUT_DIGIT(1000) PUT_DIGIT(100) PUT_DIGIT(10)
#undef PUT_DIGIT
    s[n++] = (char)('0' + v);
    return n;
}

// Four shapes of a task, four launches (the host sorts the batch by shape, then by cost):
//   FLAT   the query and its range are equally long and the band is 0 (bwa.cpp:281-288: no gap possible; three reads in four of a
//          short-read run): one pass over the bases gives score, NM and MD, the CIGAR is one M.  No DP state at all.
//   RING   banded DP whose band fits 64 columns and whose scores stay small (see CG_SMALL): the row of (H, E) lives in LDS as a RING of 64
//          columns, [column & 63][lane] -- row i only ever touches columns i - w .. i + w + 1, and every column is written (by the row
//          before: its band's last cell, or its closing store) before it is read -- so a wavefront takes 16 KB of LDS instead of the
//          (query length + 1) x 256 bytes of a full row, and twice as many wavefronts share a CU.  Only the FIRST try of mem_reg2aln's
//          retry loop runs here; a task that has to try a wider band (rare) is put on a list and goes through the ROW kernel afterwards.
//   ROW    the full row in LDS, 16 + 16 bits per column (small scores, any band); also the later tries of deferred tasks.
//   GLOBAL the row in the task's slice of a global scratch buffer, 32 + 32 bits (anything else).
// (Measured in round 6, profiles/r06n_*, r06o_*: the ring kernel -- 8 ms per million-read chunk, VALU issue in 31 % of its wave cycles -- is NOT waiting for its scattered
//  direction stores (left out: 8.7 -> 8.3 ms), nor for the LDS round trip of a cell (the row word requested a column ahead: no change), nor for wavefronts per CU
//  (rings of 16 / 32 / 64 columns by band, 9 / 13 / 21 KB per wavefront: 16.1 -> 17.4 ms per 2 M-read chunk, not kept): a lane's ~50 instructions per cell are one dependent chain.)
// In the LDS shapes "minus infinity" is -22768 instead of -2^30 without changing a single comparison (every DP value is either a real
// score or ONE sentinel plus an offset the reference has too; the two families never meet).  The query sits in LDS 4 bits per base.
// Scores are computed, not looked up: (match, mismatch, ambiguous) = (mat[0], mat[1], mat[4]) -- the bwa_fill_scmat form the host checks.
#define CG_SMALL 10000
#define CG_MINF16 (-32768 + CG_SMALL)
#define CG_RING 64
enum { CG_GLOBAL = 0, CG_ROW = 1, CG_RINGED = 2 };
enum { CG_NULL = -1, CG_DEFERRED = -3 };                          // CigarRes::n_cigar: the reference's NULL return; "try the next band in the ROW kernel"

static __device__ __forceinline__ int cg_score(int t, int q, int s_match, int s_mis, int s_amb) { return (t == q && t < 4) ? s_match : ((t > 3 || q > 3) ? s_amb : s_mis); }

// NM and MD of a finished CIGAR (bwa.cpp:311-340)
template <class QF, class RFn, class RF8n>
static __device__ __forceinline__ void cg_nm_md(const uint32_t *cg, int ncg, QF Q, RFn RF, RF8n RF8 /* (i, n): reference bases i .. i + n - 1, n <= 8, four bits each */, bool rev, char *md, CigarRes &R) {
    int x = 0, y = 0, u = 0, n_mm = 0, n_gap = 0, nmd = 0;
    const char *int2base = rev ? "TGCAN" : "ACGTN";
    for (int k = 0; k < ncg; ++k) {
        const int op = cg[k] & 0xf, len = (int)(cg[k] >> 4);
        if (op == 0) {
            for (int i0 = 0; i0 < len; i0 += 8) {                  // (eight reference bases in one load: see k_cigar_flat)
                const int nk = len - i0 < 8 ? len - i0 : 8;
                uint32_t tw = RF8(y + i0, nk);
#pragma nounroll
                for (int k = 0; k < nk; ++k, tw >>= 4) {
                    const int rb_ = (int)(tw & 15u);
                    if (Q(x + i0 + k) != rb_) { nmd = put_dec(md, nmd, u); md[nmd++] = int2base[rb_]; ++n_mm; u = 0; }
                    else ++u;
                }
            }
            x += len; y += len;
        } else if (op == 2) {
            if (k > 0 && k < ncg - 1) {
                nmd = put_dec(md, nmd, u); md[nmd++] = '^';
                for (int i = 0; i < len; ++i) md[nmd++] = int2base[RF(y + i)];
                u = 0; n_gap += len;
            }
            y += len;
        } else if (op == 1) { x += len; n_gap += len; }
    }
    nmd = put_dec(md, nmd, u);
    md[nmd] = 0;
    R.nm = n_mm + n_gap; R.n_cigar = ncg; R.md_len = nmd;
}

__global__ void __launch_bounds__(256)
k_cigar_flat(RefPtr ref, const uint8_t *__restrict__ seqs, const CigarTask *__restrict__ tasks, const int *__restrict__ order,
             int n, CigarPrm prm, uint32_t *__restrict__ cgbuf, char *__restrict__ mdbuf, CigarRes *__restrict__ res) {
    const int slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= n) return;
    const int id = order[slot];
    const CigarTask T = tasks[id];
    CigarRes R; R.score = 0; R.nm = -1; R.n_cigar = CG_NULL; R.md_len = 0;
    if (!cigar_range_ok(prm.l_pac, T.q_len, T.rb, T.re)) { res[id] = R; return; }     // (the NULL returns of the batch are filed under this shape)
    const int lq = T.q_len;
    const bool rev = T.rb >= prm.l_pac;
    const uint8_t *qp = seqs + T.q_off;
    const int s_match = prm.mat[0], s_mis = prm.mat[1], s_amb = prm.mat[4];
    char *md = mdbuf + T.md_off;
    const char *int2base = rev ? "TGCAN" : "ACGTN";
    int sc = 0, u = 0, n_mm = 0, nmd = 0;
    // both in alignment order (reversed for a hit on the reverse strand).  EIGHT positions at a time, by TWO loads (RefPtr::nib8): a lane that walks its
    // read and its reference range byte by byte fetches every line of them 64 (16) times, each a round trip of its own -- the MD bytes a mismatch stores may
    // alias anything a load reads, so the compiler keeps the loads in order -- and the lanes of a CU hold far more lines than its L1: the kernel waited in 69 %
    // of its wave cycles and issued VALU in 2 % (profiles/r06f_tail_kernels_pmc_sq.md).  The last, partial group goes position by position.
    const RefPtr qr = RefPtr::bytes(qp);
    for (int i0 = 0; i0 < lq; i0 += 8) {
        uint32_t tw = 0, qw = 0;                                // four bits per position
        const int nk = lq - i0 < 8 ? lq - i0 : 8;
        if (nk == 8) {
            tw = rev ? ref.nib8(T.re - 1 - i0, -1) : ref.nib8(T.rb + i0, 1);
            qw = rev ? qr.nib8(lq - 1 - i0, -1) : qr.nib8(i0, 1);
        } else for (int k = 0; k < nk; ++k) {
            const int i = i0 + k;
            tw |= (uint32_t)(rev ? ref[T.re - 1 - i] : ref[T.rb + i]) << (4 * k);
            qw |= (uint32_t)((rev ? qp[lq - 1 - i] : qp[i]) & 15) << (4 * k);
        }
#pragma nounroll
        for (int k = 0; k < nk; ++k, tw >>= 4, qw >>= 4) {
            const int t = (int)(tw & 15u), q = (int)(qw & 15u);
            sc += cg_score(t, q, s_match, s_mis, s_amb);
            if (q != t) { nmd = put_dec(md, nmd, u); md[nmd++] = int2base[t]; ++n_mm; u = 0; }
            else ++u;
        }
    }
    nmd = put_dec(md, nmd, u);
    md[nmd] = 0;
    cgbuf[T.cg_off] = (uint32_t)lq << 4;
    R.score = sc; R.nm = n_mm; R.n_cigar = 1; R.md_len = nmd;
    res[id] = R;
}

template <int MODE>
__global__ void __launch_bounds__(64)
k_gen_cigar(RefPtr ref, const uint8_t *__restrict__ seqs, const CigarTask *__restrict__ tasks, const int *__restrict__ order,
            int n, CigarPrm prm, uint8_t *__restrict__ zbuf, int2 *__restrict__ ehbuf, uint32_t *__restrict__ cgbuf, char *__restrict__ mdbuf,
            CigarRes *__restrict__ res, int qmax, int resume, int *__restrict__ defer_list, int *__restrict__ defer_count) {
    constexpr bool LDS = MODE != CG_GLOBAL, RING = MODE == CG_RINGED;
    extern __shared__ __attribute__((aligned(16))) uint32_t cg_lds[];
    uint32_t *EH = cg_lds;                                      // ROW: [(qmax + 1)][64]; RING: [64][64]
    uint32_t *Q4 = cg_lds + (size_t)(RING ? CG_RING : qmax + 1) * 64;        // [(qmax + 7) / 8][64]: 8 bases of 4 bits
    const int lane = threadIdx.x;
    constexpr int MINF = LDS ? CG_MINF16 : CG_MINUS_INF;
    const int slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= n) return;
    const int id = order[slot];
    const CigarTask T = tasks[id];
    CigarRes R; R.score = 0; R.nm = -1; R.n_cigar = CG_NULL; R.md_len = 0;
    if (!cigar_range_ok(prm.l_pac, T.q_len, T.rb, T.re)) { res[id] = R; return; }
    const int lq = T.q_len, rlen = (int)(T.re - T.rb);
    const bool rev = T.rb >= prm.l_pac;
    const uint8_t *qp = seqs + T.q_off;
    const int s_match = prm.mat[0], s_mis = prm.mat[1], s_amb = prm.mat[4];
    if (LDS) {                                                  // the query in alignment order (reversed for reverse-strand hits)
        const RefPtr qr = RefPtr::bytes(qp);
        for (int j0 = 0; j0 < lq; j0 += 8) {
            uint32_t wq = 0;
            if (j0 + 8 <= lq) wq = rev ? qr.nib8(lq - 1 - j0, -1) : qr.nib8(j0, 1);      // (one load: see k_cigar_flat)
            else for (int u = 0; j0 + u < lq; u++) wq |= (uint32_t)((rev ? qp[lq - 1 - (j0 + u)] : qp[j0 + u]) & 15) << (4 * u);
            Q4[(j0 >> 3) * 64 + lane] = wq;
        }
    }
    auto Q = [&](int i) -> int { return LDS ? (int)((Q4[(i >> 3) * 64 + lane] >> (4 * (i & 7))) & 15u) : (rev ? qp[lq - 1 - i] : qp[i]); };
    auto RF = [&](int i) -> int { return rev ? ref[T.re - 1 - i] : ref[T.rb + i]; };
    auto RF8 = [&](int i, int n) -> uint32_t {                   // reference bases i .. i + n - 1 (n <= 8) of the range, four bits each
        if (n == 8) return rev ? ref.nib8(T.re - 1 - i, -1) : ref.nib8(T.rb + i, 1);
        uint32_t tw = 0;
        for (int k = 0; k < n; ++k) tw |= (uint32_t)RF(i + k) << (4 * k);
        return tw;
    };
    int2 *ehg = ehbuf + T.eh_off;                               // .x = h, .y = e (global version)
    auto eh_get = [&](int j) -> int2 {
        if (!LDS) return ehg[j];
        const uint32_t w = EH[(RING ? (j & (CG_RING - 1)) : j) * 64 + lane];
        return make_int2((int)(int16_t)(w & 0xffffu), (int)w >> 16);
    };
    auto eh_set = [&](int j, int h, int e) {
        if (!LDS) ehg[j] = make_int2(h, e);
        else EH[(RING ? (j & (CG_RING - 1)) : j) * 64 + lane] = ((uint32_t)h & 0xffffu) | (uint32_t)e << 16;
    };
    uint32_t *cg = cgbuf + T.cg_off;
    int ncg = 0;
    // one alignment (bwa_gen_cigar2, bwa.cpp:281-310), or mem_reg2aln's loop around it (bwamem.cpp:1748-1766): retry with twice the band
    // while the score keeps changing, stays more than a match below the hit's score, and the band has not reached 4 w
    int w_try = T.retry ? cigar_first_band(lq, rlen, T.truesc, T.w, prm.a, prm.w, prm.o_del, prm.e_del, prm.o_ins, prm.e_ins) : T.w;
    int last_sc = -(1 << 30), attempt = 0;
    if (resume) { const CigarRes P = res[id]; last_sc = P.score; w_try = P.nm; attempt = P.md_len; }      // (a deferred task: where the RING kernel left its loop)
    for (;;) {
        if (T.retry) w_try = w_try < prm.w << 2 ? w_try : prm.w << 2;
        ncg = 0;
        if (lq == rlen && w_try == 0) {                         // no gap possible: one M (bwa.cpp:281-288)
            int sc = 0;
            for (int i = 0; i < lq; ++i) sc += cg_score(RF(i), Q(i), s_match, s_mis, s_amb);
            R.score = sc;
            cg[ncg++] = (uint32_t)lq << 4;
        } else {
            const int w = cigar_band(lq, rlen, w_try, prm.mat[0], prm.o_del, prm.e_del, prm.o_ins, prm.e_ins);
            const int oe_del = prm.o_del + prm.e_del, oe_ins = prm.o_ins + prm.e_ins, e_del = prm.e_del, e_ins = prm.e_ins;
            const int n_col = ((lq < 2 * w + 1 ? lq : 2 * w + 1) + 3) & ~3;       // row stride: whole dwords (the directions are stored four cells at a time)
            uint8_t *z = zbuf + T.z_off;
            int j;
            // row -1.  Only the columns 0 .. w are ever read from it: a column further right is first read by the row whose band reaches it,
            // and the row before that one has stored it (its closing store).  H of the LAST column travels in a register beside the row.
            int h_lq = lq <= w ? -(prm.o_ins + e_ins * lq) : MINF;
            eh_set(0, 0, MINF);
            for (j = 1; j <= lq && j <= w; ++j) eh_set(j, -(prm.o_ins + e_ins * j), MINF);
            if (!RING) for (; j <= lq; ++j) eh_set(j, MINF, MINF);
            uint32_t t_word = 0, t_ahead = RF8(0, rlen < 8 ? rlen : 8);
            for (int i = 0; i < rlen; ++i) {                    // ksw.cpp:598-637
                int f = MINF;
                if ((i & 7) == 0) {                             // the rows' reference bases eight to a load, the next eight requested eight rows ahead
                    t_word = t_ahead;
                    if (i + 8 < rlen) t_ahead = RF8(i + 8, rlen - (i + 8) < 8 ? rlen - (i + 8) : 8);
                }
                const int tb = (int)(t_word & 15u);
                t_word >>= 4;
                const int beg = i > w ? i - w : 0, end = i + w + 1 < lq ? i + w + 1 : lq;
                int h1 = beg == 0 ? -(prm.o_del + e_del * (i + 1)) : MINF;
                uint32_t *zi = (uint32_t *)(z + (int64_t)i * n_col);
                uint32_t zacc = 0;
                // LDS shapes: the row word of the NEXT column is requested before this column is computed (a column is written by its own cell only, so the
                // request may pass the store), and the query's 4-bit codes come eight to a register, shifted along -- with both reads asked for and waited
                // for inside every cell the ring kernel issued VALU in 31 % of its wave cycles and waited in 50 (profiles/r06f_tail_kernels_pmc_sq.md)
                uint32_t w_next = 0, q_word = 0;
                if (LDS && beg < end) {
                    w_next = EH[(RING ? (beg & (CG_RING - 1)) : beg) * 64 + lane];
                    q_word = Q4[(beg >> 3) * 64 + lane] >> (4 * (beg & 7));
                }
                for (j = beg; j < end; ++j) {
                    int2 p; int qj;
                    if (LDS) {
                        const uint32_t wv = w_next;
                        w_next = EH[(RING ? ((j + 1) & (CG_RING - 1)) : j + 1) * 64 + lane];      // (column `end` exists in both shapes)
                        p = make_int2((int)(int16_t)(wv & 0xffffu), (int)wv >> 16);
                        qj = (int)(q_word & 15u);
                        q_word >>= 4;
                        if (((j + 1) & 7) == 0 && j + 1 < lq) q_word = Q4[((j + 1) >> 3) * 64 + lane];
                    } else { p = eh_get(j); qj = Q(j); }
                    int m = p.x, e = p.y, h, t;
                    uint8_t d;
                    m += cg_score(tb, qj, s_match, s_mis, s_amb);
                    d = m >= e ? 0 : 1;
                    h = m >= e ? m : e;
                    d = h >= f ? d : 2;
                    h = h >= f ? h : f;
                    t = m - oe_del; e -= e_del;
                    d |= e > t ? 1 << 2 : 0;
                    e = e > t ? e : t;
                    eh_set(j, h1, e);
                    h1 = h;
                    t = m - oe_ins; f -= e_ins;
                    d |= f > t ? 2 << 4 : 0;
                    f = f > t ? f : t;
                    const int c = j - beg;
                    zacc |= (uint32_t)d << (8 * (c & 3));
                    if ((c & 3) == 3) { zi[c >> 2] = zacc; zacc = 0; }
                }
                if ((end - beg) & 3) zi[(end - beg) >> 2] = zacc;
                eh_set(end, h1, MINF);
                if (end == lq) h_lq = h1;
            }
            R.score = h_lq;
            // backtrack (ksw.cpp:640-660): ops are pushed last-to-first, merged, then reversed
            auto push = [&](int op, int len) {
                if (ncg == 0 || op != (int)(cg[ncg - 1] & 0xf)) cg[ncg++] = (uint32_t)len << 4 | (uint32_t)op;
                else cg[ncg - 1] += (uint32_t)len << 4;
            };
            int which = 0, i = rlen - 1, k = (i + w + 1 < lq ? i + w + 1 : lq) - 1;
            while (i >= 0 && k >= 0) {
                which = z[(int64_t)i * n_col + (k - (i > w ? i - w : 0))] >> (which << 1) & 3;
                if (which == 0) { push(0, 1); --i; --k; }
                else if (which == 1) { push(2, 1); --i; }
                else { push(1, 1); --k; }
            }
            if (i >= 0) push(2, i + 1);
            if (k >= 0) push(1, k + 1);
            for (int a = 0, b = ncg; a + 1 < b; ++a, --b) { const uint32_t t = cg[a]; cg[a] = cg[b - 1]; cg[b - 1] = t; }
        }
        if (!T.retry) break;
        if (R.score == last_sc || w_try == prm.w << 2) break;
        last_sc = R.score;
        w_try <<= 1;
        if (!(++attempt < 3 && R.score < T.truesc - prm.a)) break;
        if (RING) {                                             // another try, with a wider band: the ROW kernel's (rare)
            R.score = last_sc; R.nm = w_try; R.md_len = attempt; R.n_cigar = CG_DEFERRED;
            res[id] = R;
            defer_list[atomicAdd(defer_count, 1)] = id;
            return;
        }
    }
    cg_nm_md(cg, ncg, Q, RF, RF8, rev, mdbuf + T.md_off, R);
    res[id] = R;
}

// per-task slices -> dense arrays: ops of task i at cg_out[cg_pos[i] ..], its MD (with the NUL) at md_out[md_pos[i] ..]
__global__ void __launch_bounds__(256)
k_cigar_sizes(int n, const CigarRes *__restrict__ res, int32_t *n_ops, int32_t *n_md) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    n_ops[i] = res[i].n_cigar > 0 ? res[i].n_cigar : 0;
    n_md[i] = res[i].n_cigar >= 0 ? res[i].md_len + 1 : 1;
}
__global__ void __launch_bounds__(256)
k_cigar_compact(int n, const CigarTask *__restrict__ tasks, const CigarRes *__restrict__ res, const uint32_t *__restrict__ cgbuf,
                const char *__restrict__ mdbuf, const int64_t *__restrict__ cg_pos, const int64_t *__restrict__ md_pos, uint32_t *cg_out, char *md_out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const CigarRes R = res[i];
    const CigarTask T = tasks[i];
    for (int k = 0; k < R.n_cigar; ++k) cg_out[cg_pos[i] + k] = cgbuf[T.cg_off + k];
    if (R.n_cigar >= 0) for (int k = 0; k <= R.md_len; ++k) md_out[md_pos[i] + k] = mdbuf[T.md_off + k];
    else md_out[md_pos[i]] = 0;
}

// Runs the tasks (slices not yet assigned) against the context's resident reference; queries are ranges of `seqs` (uploaded here).
// Results: res[i] and the dense arrays (offsets cg_pos / md_pos have n + 1 entries).
static int cigar_run(bm2_ctx *c, const bm2_opt *opt, std::vector<CigarTask> &tasks, const uint8_t *seqs, int64_t seq_bytes,
                     std::vector<CigarRes> &h_res, std::vector<int64_t> &cg_pos, std::vector<int64_t> &md_pos, std::vector<uint32_t> &cg, std::vector<char> &md) {
    TailProf prof("gen_cigar_dev");
    const int n = (int)tasks.size();
    int rc = bm2_check(hipSetDevice(c->device), "hipSetDevice");
    if (rc) return rc;
    CigarPrm prm; memset(&prm, 0, sizeof prm);
    for (int a = 0; a < 25; ++a) prm.mat[a] = opt->mat[a];
    prm.o_del = opt->o_del; prm.e_del = opt->e_del; prm.o_ins = opt->o_ins; prm.e_ins = opt->e_ins; prm.a = opt->a; prm.w = opt->w; prm.l_pac = c->ix.l_pac;
    // slices of the scratch buffers (sized for the widest band a task can come to) and the cost class of every task: sizes per piece
    // of the task list on the host's worker threads, offsets by a scan over the pieces, then the tasks' slices
    static thread_local std::vector<int> order_tl; static thread_local std::vector<int64_t> cost_tl; static thread_local std::vector<int16_t> wb_tl;
    std::vector<int> &order = order_tl; std::vector<int64_t> &cost = cost_tl; std::vector<int16_t> &wb1 = wb_tl;
    if (order.size() < (size_t)n) { order.resize((size_t)n); cost.resize((size_t)n); wb1.resize((size_t)n); }
    for (int i = 0; i < 5; i++) for (int j = 0; j < 5; j++) {       // the kernels compute scores from (match, mismatch, ambiguous), as the hot path does
        const int want = (i == 4 || j == 4) ? opt->mat[4] : (i == j ? opt->mat[0] : opt->mat[1]);
        if (opt->mat[i * 5 + j] != want) { bm2_set_error("bm2_gen_cigar_dev: scoring matrix is not of the bwa_fill_scmat form"); return BM2_EUNSUP; }
    }
    const int host_threads = bm2_host_threads();
    const int64_t grain = 16384, pieces = ((int64_t)n + grain - 1) / grain;
    struct Sz { int64_t z, e, c, m; int wb_first; };          // wb_first: the band of the first (or only) alignment, -1 = none (flat / NULL)
    std::vector<Sz> at((size_t)pieces + 1, Sz{ 0, 0, 0, 0, -1 });
    std::atomic<int> too_long(0);
    auto sizes = [&](const CigarTask &T, Sz &s) {               // what the task takes of the four buffers; s.z = its DP area = its cost
        s = Sz{ 0, 0, 0, 0, -1 };
        if (!cigar_range_ok(prm.l_pac, T.q_len, T.rb, T.re)) return;
        const int64_t rlen = T.re - T.rb;
        if (rlen > 0x3fffffff) { too_long = 1; return; }
        const int w_first = T.retry ? cigar_first_band(T.q_len, (int)rlen, T.truesc, T.w, prm.a, prm.w, prm.o_del, prm.e_del, prm.o_ins, prm.e_ins) : T.w;
        if (!(T.q_len == rlen && w_first == 0)) {               // (a retried task never starts from band 0: the first score ends its loop)
            const int w_cap = T.retry ? (w_first < prm.w << 2 ? prm.w << 2 : w_first) : T.w;
            const int wb = cigar_band(T.q_len, (int)rlen, w_cap, prm.mat[0], prm.o_del, prm.e_del, prm.o_ins, prm.e_ins);
            const int n_col = ((T.q_len < 2 * wb + 1 ? T.q_len : 2 * wb + 1) + 3) & ~3;      // a multiple of 4: every z slice starts 4-aligned
            s.z = (int64_t)n_col * rlen; s.e = T.q_len + 1;
            const int w0 = T.retry ? (w_first < prm.w << 2 ? w_first : prm.w << 2) : T.w;      // (the kernel's clamp of the first try)
            s.wb_first = cigar_band(T.q_len, (int)rlen, w0, prm.mat[0], prm.o_del, prm.e_del, prm.o_ins, prm.e_ins);
        }
        s.c = T.q_len + rlen + 2; s.m = 2 * (T.q_len + rlen) + 16;
    };
    bm2_parallel_ranges(n, grain, host_threads, [&](int64_t lo, int64_t hi) {
        Sz sum{ 0, 0, 0, 0, -1 }, s;
        for (int64_t i = lo; i < hi; ++i) {
            sizes(tasks[(size_t)i], s);
            cost[(size_t)i] = s.z; wb1[(size_t)i] = (int16_t)(s.wb_first > 32000 ? 32000 : s.wb_first);
            sum.z += s.z; sum.e += s.e; sum.c += s.c; sum.m += s.m;
        }
        at[(size_t)(lo / grain) + 1] = sum;
    });
    if (too_long.load()) { bm2_set_error("bm2_gen_cigar_dev: reference range too long"); return BM2_EINVAL; }
    for (int64_t p = 0; p < pieces; ++p) { Sz &x = at[(size_t)p + 1]; const Sz &y = at[(size_t)p]; x.z += y.z; x.e += y.e; x.c += y.c; x.m += y.m; }
    const int64_t zo = at[(size_t)pieces].z, eo = at[(size_t)pieces].e, co = at[(size_t)pieces].c, mo = at[(size_t)pieces].m;
    bm2_parallel_ranges(n, grain, host_threads, [&](int64_t lo, int64_t hi) {
        Sz o = at[(size_t)(lo / grain)], s;
        for (int64_t i = lo; i < hi; ++i) {
            CigarTask &T = tasks[(size_t)i];
            T.z_off = o.z; T.eh_off = o.e; T.cg_off = o.c; T.md_off = o.m;
            sizes(T, s);
            o.z += s.z; o.e += s.e; o.c += s.c; o.m += s.m;
        }
    });
    prof.mark("slices");
    // lanes of a wavefront run their tasks side by side: neighbours should be of one shape (see k_gen_cigar) and cost alike.  Counting sort by
    // (shape, cost on a log scale, expensive first); the order array is the four shapes' task lists one after the other.
    enum { SH_RING = 0, SH_ROW = 1, SH_GLOBAL = 2, SH_FLAT = 3 };
    int n_shape[4] = { 0, 0, 0, 0 }, qmax = 0;
    {
        int pen = 1;
        for (int a = 0; a < 25; ++a) pen = std::max(pen, abs((int)opt->mat[a]));
        pen = std::max(pen, std::max(opt->o_del + opt->e_del, opt->o_ins + opt->e_ins));
        static const int no_lds = getenv("BM2_CIGAR_NO_LDS") ? atoi(getenv("BM2_CIGAR_NO_LDS")) : 0;
        static const int no_ring = getenv("BM2_CIGAR_NO_RING") ? atoi(getenv("BM2_CIGAR_NO_RING")) : 0;
        const int ring_by_band = bm2_knob("BM2_CIGAR_RING_BY_BAND", 1);
        auto small = [&](int i) { const CigarTask &T = tasks[(size_t)i]; return !no_lds && T.q_len <= 220 && (int64_t)(T.q_len + (T.re - T.rb)) * pen < CG_SMALL; };
        auto shape = [&](int i) {
            if (cost[(size_t)i] == 0) return (int)SH_FLAT;       // (no DP area: the one-M case, or a NULL return)
            if (!small(i)) return (int)SH_GLOBAL;
            return (!no_ring && 2 * (int)wb1[(size_t)i] + 2 <= CG_RING) ? (int)SH_RING : (int)SH_ROW;
        };
        // (the RING tasks by the band of their FIRST try, exactly: a wavefront steps through the widest band among its 64 tasks, and `cost` -- the DP area
        //  the task's slices are sized for, i.e. of the widest band its retry loop can come to -- is the same for nearly all of them)
        auto cls = [&](int i) {
            const int sh = shape(i);
            if (sh == SH_RING && ring_by_band) return sh * 64 + 31 - std::min(std::max((int)wb1[(size_t)i], 0), 31);
            const int64_t v = cost[(size_t)i]; const int k = v > 0 ? 64 - __builtin_clzll((unsigned long long)v) : 0; return sh * 64 + 63 - k;
        };
        bm2_counting_order(n, 256, host_threads, cls, order.data());
        std::atomic<int> ns[4], qm(0);
        for (auto &x : ns) x = 0;
        bm2_parallel_ranges(n, grain, host_threads, [&](int64_t lo, int64_t hi) {
            int c[4] = { 0, 0, 0, 0 }, q = 0;
            for (int64_t i = lo; i < hi; ++i) { const int sh = shape((int)i); ++c[sh]; if (sh == SH_RING || sh == SH_ROW) q = std::max(q, tasks[(size_t)i].q_len); }
            for (int k = 0; k < 4; ++k) ns[k] += c[k];
            for (int cur = qm.load(); q > cur && !qm.compare_exchange_weak(cur, q);) {}
        });
        for (int k = 0; k < 4; ++k) n_shape[k] = ns[k].load();
        qmax = qm.load();
    }
    prof.mark("order");
    c->n_bsw = 0;                                               // (these scratch buffers held the resident S1 batch, if any: it is gone)
    DevBuf &b_seq = c->b_ref, &b_task = c->b_qer, &b_res = c->b_pairs, &b_scr = c->b_misc;
    const size_t task_bytes = ((size_t)n * sizeof(CigarTask) + 15) & ~(size_t)15, ord_bytes = ((size_t)n * sizeof(int) + 15) & ~(size_t)15;
    const size_t z_bytes = ((size_t)zo + 15) & ~(size_t)15, eh_bytes = (size_t)eo * sizeof(int2), cg_bytes = ((size_t)co * 4 + 15) & ~(size_t)15, md_bytes = ((size_t)mo + 15) & ~(size_t)15;
    const size_t res_bytes = ((size_t)n * sizeof(CigarRes) + 15) & ~(size_t)15, cnt_bytes = ((size_t)(n + 2) * 4 + 15) & ~(size_t)15, pos_bytes = (size_t)(n + 2) * 8;
    const size_t defer_bytes = ((size_t)(n_shape[SH_RING] + 4) * 4 + 15) & ~(size_t)15;      // [0] = count, then the list
    if ((rc = bm2_reserve(b_seq, (size_t)seq_bytes + 64))) return rc;
    if ((rc = bm2_reserve(b_task, task_bytes + ord_bytes + 64))) return rc;
    if ((rc = bm2_reserve(b_res, res_bytes + 2 * cnt_bytes + 2 * pos_bytes + defer_bytes + 64))) return rc;
    if ((rc = bm2_reserve(b_scr, z_bytes + eh_bytes + cg_bytes + md_bytes + 64))) return rc;
    hipStream_t s = c->stream;
    CigarTask *d_task = (CigarTask *)b_task.p; int *d_order = (int *)((char *)b_task.p + task_bytes);
    CigarRes *d_res = (CigarRes *)b_res.p;
    int32_t *d_nops = (int32_t *)((char *)b_res.p + res_bytes), *d_nmd = (int32_t *)((char *)d_nops + cnt_bytes);
    int64_t *d_cgpos = (int64_t *)((char *)d_nmd + cnt_bytes), *d_mdpos = (int64_t *)((char *)d_cgpos + pos_bytes);
    int *d_defer = (int *)((char *)d_mdpos + pos_bytes);
    uint8_t *d_z = (uint8_t *)b_scr.p; int2 *d_eh = (int2 *)((char *)b_scr.p + z_bytes);
    uint32_t *d_cg = (uint32_t *)((char *)d_eh + eh_bytes); char *d_md = (char *)d_cg + cg_bytes;
    prof.mark("reserve");
    rc = bm2_copy_h2d(c, b_seq.p, seqs, (size_t)seq_bytes);        // (pageable memory: through the context's pinned staging buffers)
    if (!rc) rc = bm2_copy_h2d(c, d_task, tasks.data(), (size_t)n * sizeof(CigarTask));
    if (!rc) rc = bm2_copy_h2d(c, d_order, order.data(), (size_t)n * sizeof(int));
    if (rc) return rc;
    const size_t lds_q = (size_t)((qmax + 7) / 8) * 256, lds_row = (size_t)(qmax + 1) * 256 + lds_q, lds_ring = (size_t)CG_RING * 256 + lds_q;
    if (lds_row > 64 * 1024 && (rc = bm2_check(hipFuncSetAttribute((const void *)k_gen_cigar<CG_ROW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_row), "hipFuncSetAttribute(k_gen_cigar)"))) return rc;
    const int *o_ring = d_order, *o_row = o_ring + n_shape[SH_RING], *o_glob = o_row + n_shape[SH_ROW], *o_flat = o_glob + n_shape[SH_GLOBAL];
    if ((rc = bm2_check(hipMemsetAsync(d_defer, 0, 4, s), "memset"))) return rc;
    // (the costliest tasks lead every list; the four launches follow each other on the stream, the cheap flat tasks last)
    if (n_shape[SH_RING])
        hipLaunchKernelGGL(k_gen_cigar<CG_RINGED>, dim3((n_shape[SH_RING] + 63) / 64), dim3(64), lds_ring, s, c->ix.ref(0), (const uint8_t *)b_seq.p, d_task, o_ring,
                           n_shape[SH_RING], prm, d_z, d_eh, d_cg, d_md, d_res, qmax, 0, d_defer + 1, d_defer);
    if (n_shape[SH_ROW])
        hipLaunchKernelGGL(k_gen_cigar<CG_ROW>, dim3((n_shape[SH_ROW] + 63) / 64), dim3(64), lds_row, s, c->ix.ref(0), (const uint8_t *)b_seq.p, d_task, o_row,
                           n_shape[SH_ROW], prm, d_z, d_eh, d_cg, d_md, d_res, qmax, 0, (int *)nullptr, (int *)nullptr);
    if (n_shape[SH_GLOBAL])
        hipLaunchKernelGGL(k_gen_cigar<CG_GLOBAL>, dim3((n_shape[SH_GLOBAL] + 63) / 64), dim3(64), 0, s, c->ix.ref(0), (const uint8_t *)b_seq.p, d_task, o_glob,
                           n_shape[SH_GLOBAL], prm, d_z, d_eh, d_cg, d_md, d_res, 0, 0, (int *)nullptr, (int *)nullptr);
    if (n_shape[SH_FLAT])
        hipLaunchKernelGGL(k_cigar_flat, dim3((n_shape[SH_FLAT] + 255) / 256), dim3(256), 0, s, c->ix.ref(0), (const uint8_t *)b_seq.p, d_task, o_flat,
                           n_shape[SH_FLAT], prm, d_cg, d_md, d_res);
    int n_defer = 0;
    if (n_shape[SH_RING]) {                                      // tasks whose first try asked for a wider band: their later tries in the ROW kernel
        if ((rc = bm2_check(hipMemcpyAsync(&n_defer, d_defer, 4, hipMemcpyDeviceToHost, s), "D2H deferred"))) return rc;
        if ((rc = bm2_check(hipStreamSynchronize(s), "k_gen_cigar"))) return rc;
        if (n_defer > 0)
            hipLaunchKernelGGL(k_gen_cigar<CG_ROW>, dim3((n_defer + 63) / 64), dim3(64), lds_row, s, c->ix.ref(0), (const uint8_t *)b_seq.p, d_task, d_defer + 1,
                               n_defer, prm, d_z, d_eh, d_cg, d_md, d_res, qmax, 1, (int *)nullptr, (int *)nullptr);
    }
    if ((rc = bm2_check(hipGetLastError(), "k_gen_cigar launch"))) return rc;
    if (prof.on) {
        (void)hipStreamSynchronize(s);
        fprintf(stderr, "[tail] gen_cigar_dev  tasks by shape: ring %d (%d of them deferred to a wider band), row %d, global %d, flat %d\n", n_shape[SH_RING], n_defer,
                n_shape[SH_ROW], n_shape[SH_GLOBAL], n_shape[SH_FLAT]);
        prof.mark("H2D + kernel");
    }
    // dense output: sizes -> offsets (scan) -> gather, all on the device; then one small copy back
    hipLaunchKernelGGL(k_cigar_sizes, dim3((n + 255) / 256), dim3(256), 0, s, n, d_res, d_nops, d_nmd);
    DevBuf scan_tmp;                                            // (scratch of the scans; small)
    if ((rc = bm2_scan_i32(c, d_nops, n, d_cgpos, scan_tmp))) { bm2_release(scan_tmp); return rc; }
    if ((rc = bm2_scan_i32(c, d_nmd, n, d_mdpos, scan_tmp))) { bm2_release(scan_tmp); return rc; }
    cg_pos.resize((size_t)n + 1); md_pos.resize((size_t)n + 1); h_res.resize((size_t)n);
    rc = bm2_copy_d2h(c, cg_pos.data(), d_cgpos, (size_t)(n + 1) * 8);
    if (!rc) rc = bm2_copy_d2h(c, md_pos.data(), d_mdpos, (size_t)(n + 1) * 8);
    if (!rc) rc = bm2_copy_d2h(c, h_res.data(), d_res, (size_t)n * sizeof(CigarRes));
    bm2_release(scan_tmp);
    if (rc) return rc;
    const int64_t n_cg = cg_pos[(size_t)n], n_md = md_pos[(size_t)n];
    DevBuf &b_out = c->b_pairs2;
    const size_t ocg_bytes = ((size_t)n_cg * 4 + 15) & ~(size_t)15;
    if ((rc = bm2_reserve(b_out, ocg_bytes + (size_t)n_md + 64))) return rc;
    uint32_t *o_cg = (uint32_t *)b_out.p; char *o_md = (char *)b_out.p + ocg_bytes;
    hipLaunchKernelGGL(k_cigar_compact, dim3((n + 255) / 256), dim3(256), 0, s, n, d_task, d_res, d_cg, d_md, d_cgpos, d_mdpos, o_cg, o_md);
    if ((rc = bm2_check(hipGetLastError(), "k_cigar_compact launch"))) return rc;
    cg.resize((size_t)n_cg + 1); md.resize((size_t)n_md + 1);
    if (n_cg) rc = bm2_copy_d2h(c, cg.data(), o_cg, (size_t)n_cg * 4);
    if (!rc && n_md) rc = bm2_copy_d2h(c, md.data(), o_md, (size_t)n_md);
    if (!rc) rc = bm2_check(hipStreamSynchronize(s), "bm2_gen_cigar_dev sync");
    prof.mark("compact + D2H");
    return rc;
}

// Device twin of bm2_gen_cigar: same arguments after the context (which must hold the index), same results.
extern "C" int bm2_gen_cigar_dev(bm2_ctx *c, const bm2_opt *opt, int32_t n, const uint8_t *seqs, int64_t seq_bytes, const int64_t *q_off,
                                 const int32_t *q_len, const int64_t *rb, const int64_t *re, const int32_t *w, int32_t *score, int32_t *nm,
                                 int32_t *n_cigar, int64_t *cigar_off, uint32_t *cigar, int64_t cigar_cap, int64_t *cigar_need,
                                 int64_t *md_off, char *md, int64_t md_cap, int64_t *md_need) {
    if (!c || !opt || n < 0 || (n > 0 && (!seqs || !q_off || !q_len || !rb || !re || !w || !score || !nm || !n_cigar || !cigar_off || !md_off)) ||
        !cigar_need || !md_need) { bm2_set_error("bm2_gen_cigar_dev: bad argument"); return BM2_EINVAL; }
    if (!c->has_index || !c->ix.ref_string) { bm2_set_error("bm2_gen_cigar_dev: the context holds no index"); return BM2_EINVAL; }
    *cigar_need = 0; *md_need = 0;
    if (n == 0) return BM2_OK;
    if (opt->e_del <= 0 || opt->e_ins <= 0) { bm2_set_error("bm2_gen_cigar_dev: gap extension penalties must be > 0"); return BM2_EINVAL; }
    std::vector<CigarTask> tasks((size_t)n);
    for (int i = 0; i < n; ++i) {
        CigarTask &T = tasks[(size_t)i]; memset(&T, 0, sizeof T);
        T.q_off = q_off[i]; T.q_len = q_len[i]; T.rb = rb[i]; T.re = re[i]; T.w = w[i];
    }
    std::vector<CigarRes> res; std::vector<int64_t> cg_pos, md_pos; std::vector<uint32_t> cg; std::vector<char> mdv;
    const int rc = cigar_run(c, opt, tasks, seqs, seq_bytes, res, cg_pos, md_pos, cg, mdv);
    if (rc) return rc;
    int64_t oc = 0, om = 0;                                     // the caller's layout: NULL results take no space at all
    for (int i = 0; i < n; ++i) {
        const CigarRes &R = res[(size_t)i];
        score[i] = R.score; nm[i] = R.nm; n_cigar[i] = R.n_cigar; cigar_off[i] = oc; md_off[i] = om;
        if (R.n_cigar < 0) continue;
        if (cigar && oc + R.n_cigar <= cigar_cap) memcpy(cigar + oc, cg.data() + cg_pos[(size_t)i], (size_t)R.n_cigar * 4);
        if (md && om + R.md_len + 1 <= md_cap) memcpy(md + om, mdv.data() + md_pos[(size_t)i], (size_t)R.md_len + 1);
        oc += R.n_cigar; om += R.md_len + 1;
    }
    *cigar_need = oc; *md_need = om;
    if (oc > cigar_cap || om > md_cap || (oc && !cigar) || (om && !md)) return BM2_ECAP;
    return BM2_OK;
}

// The CIGAR batch of a SAM chunk on the device (hook of bm2h_sam_pe / bm2h_sam_se: user = the context): every hit with the retry loop
// of mem_reg2aln run by the kernel, results compacted on the device.
int bm2_dev_cigar_batch(void *user, const bm2_opt *opt, const bm2_reads *reads, int64_t enc_bytes, int32_t n, const bm2h_cg_hit *hits, bm2h_cg_out *out) {
    bm2_ctx *c = (bm2_ctx *)user;
    static thread_local std::vector<CigarTask> tasks_tl;
    std::vector<CigarTask> &tasks = tasks_tl;
    tasks.resize((size_t)n);                                     // (cigar_run takes the list's size as the task count)
    const int host_threads = bm2_host_threads();
    bm2_parallel_ranges(n, 16384, host_threads, [&](int64_t lo, int64_t hi) {
        for (int64_t i = lo; i < hi; ++i) {
            CigarTask &T = tasks[(size_t)i]; memset(&T, 0, sizeof T);
            const bm2h_cg_hit &h = hits[i];
            T.q_off = reads->off[h.read] + h.qb; T.q_len = h.qe - h.qb; T.rb = h.rb; T.re = h.re; T.w = h.w; T.truesc = h.truesc; T.retry = 1;
        }
    });
    static thread_local std::vector<CigarRes> res_tl;
    std::vector<CigarRes> &res = res_tl;
    const int rc = cigar_run(c, opt, tasks, reads->enc, enc_bytes, res, out->cigar_off, out->md_off, out->cigar, out->md);
    if (rc) return rc;
    out->score.resize((size_t)n); out->nm.resize((size_t)n); out->n_cigar.resize((size_t)n);
    bm2_parallel_ranges(n, 16384, host_threads, [&](int64_t lo, int64_t hi) {
        for (int64_t i = lo; i < hi; ++i) { out->score[(size_t)i] = res[(size_t)i].score; out->nm[(size_t)i] = res[(size_t)i].nm; out->n_cigar[(size_t)i] = res[(size_t)i].n_cigar; }
    });
    return BM2_OK;
}

extern "C" int bm2_sam_se_dev(bm2_ctx *c, const bm2_index_desc *idx, const bm2_opt *opt, const bm2_sam_opt *so, const bm2_reads *reads,
                              const bm2_read_text *txt, bm2_alnreg_t *alnregs, const int64_t *reg_off, int64_t n_processed, char *out,
                              int64_t cap, int64_t *n_out) {
    if (!c || !c->has_index || !c->ix.ref_string) { bm2_set_error("bm2_sam_se_dev: the context holds no index"); return BM2_EINVAL; }
    return bm2h_sam_se(idx, opt, so, reads, txt, alnregs, reg_off, n_processed, out, cap, n_out, bm2_dev_cigar_batch, c);
}
