// finish_regs.cpp -- host side of the seam, above the device boundary: the tail of mem_kernel2_core
// (bwamem.cpp:1154-1169) = mem_sort_dedup_patch (bwamem.cpp:292-353) + mem_patch_reg (:175-225) + the ALT flag.
// mem_patch_reg scores a candidate merge of two colinear hits with a banded global alignment
// (bwa_gen_cigar2 -> ksw_global2, bwa.cpp:260-347, ksw.cpp:558-668; score only, no backtrack).
// Plain C++ on the host: a few alignments per read, branchy, order-sensitive (klib introsort ties are observable).
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../../include/bm2.h"
#include "ksort_host.h"
#include "host_tail.h"

void bm2_set_error(const char *fmt, ...);

namespace {

// ---- ksw_global2 without backtrack (ksw.cpp:558-668, the `else` loop) -----------------------------------------------
const int MINUS_INF = -0x40000000;
int global_score(int qlen, const uint8_t *query, int tlen, const uint8_t *target, const int8_t *mat, int o_del, int e_del,
                 int o_ins, int e_ins, int w) {
    struct EH { int32_t h, e; };
    std::vector<EH> eh((size_t)qlen + 1);
    const int oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
    int i, j;
    eh[0].h = 0; eh[0].e = MINUS_INF;
    for (j = 1; j <= qlen && j <= w; ++j) { eh[j].h = -(o_ins + e_ins * j); eh[j].e = MINUS_INF; }
    for (; j <= qlen; ++j) eh[j].h = eh[j].e = MINUS_INF;
    for (i = 0; i < tlen; ++i) {
        int32_t f = MINUS_INF, h1, beg, end, t;
        const int8_t *q = &mat[target[i] * 5];
        beg = i > w ? i - w : 0;
        end = i + w + 1 < qlen ? i + w + 1 : qlen;
        h1 = beg == 0 ? -(o_del + e_del * (i + 1)) : MINUS_INF;
        for (j = beg; j < end; ++j) {
            EH *p = &eh[j];
            int32_t h, m = p->h, e = p->e;
            p->h = h1;
            m += q[query[j]];
            h = m >= e ? m : e;
            h = h >= f ? h : f;
            h1 = h;
            t = m - oe_del; e -= e_del; e = e > t ? e : t; p->e = e;
            t = m - oe_ins; f -= e_ins; f = f > t ? f : t;
        }
        eh[end].h = h1; eh[end].e = MINUS_INF;
    }
    return eh[qlen].h;
}

// ---- bwa_gen_cigar2 with n_cigar == NM == NULL (bwa.cpp:260-347): score of the global alignment of query vs [rb, re) ----
// ref_string (.0123) holds exactly what bns_get_seq unpacks from .pac (forward, then reverse complement).
bool gen_score(const int8_t mat[25], int o_del, int e_del, int o_ins, int e_ins, int w_, int64_t l_pac, const uint8_t *ref_string,
               int l_query, const uint8_t *query, int64_t rb, int64_t re, int *score) {
    if (l_query <= 0 || rb >= re || (rb < l_pac && re > l_pac)) return false;
    if (re > (l_pac << 1)) re = l_pac << 1;               // bns_get_seq clamps; a clamped range makes the caller bail out
    if (rb < 0) rb = 0;
    const int64_t rlen = re - rb;
    std::vector<uint8_t> rseq(ref_string + rb, ref_string + re), q(query, query + l_query);
    if (rb >= l_pac) {                                    // reverse both so that indels are placed leftmost (:273-278)
        for (int i = 0; i < l_query >> 1; ++i) { uint8_t t = q[i]; q[i] = q[l_query - 1 - i]; q[l_query - 1 - i] = t; }
        for (int64_t i = 0; i < rlen >> 1; ++i) { uint8_t t = rseq[i]; rseq[i] = rseq[rlen - 1 - i]; rseq[rlen - 1 - i] = t; }
    }
    if (l_query == rlen && w_ == 0) {
        int sc = 0;
        for (int i = 0; i < l_query; ++i) sc += mat[rseq[i] * 5 + q[i]];
        *score = sc;
    } else {
        int max_ins = (int)((double)(((l_query + 1) >> 1) * mat[0] - o_ins) / e_ins + 1.);
        int max_del = (int)((double)(((l_query + 1) >> 1) * mat[0] - o_del) / e_del + 1.);
        int max_gap = max_ins > max_del ? max_ins : max_del;
        max_gap = max_gap > 1 ? max_gap : 1;
        int w = (max_gap + abs((int)rlen - l_query) + 1) >> 1;
        w = w < w_ ? w : w_;
        const int min_w = abs((int)rlen - l_query) + 3;
        w = w > min_w ? w : min_w;
        *score = global_score(l_query, q.data(), (int)rlen, rseq.data(), mat, o_del, e_del, o_ins, e_ins, w);
    }
    return true;
}

#define PATCH_MAX_R_BW 0.05f
#define PATCH_MIN_SC_RATIO 0.90f

// mem_patch_reg, bwamem.cpp:175-225
int patch_reg(const bm2_opt *opt, int64_t l_pac, const uint8_t *ref_string, const uint8_t *query, const bm2_alnreg_t *a,
              const bm2_alnreg_t *b, int *_w) {
    if (!ref_string || !query) return 0;                   // bwamem.cpp:181
    int w, score = 0, q_s, r_s;
    double r;
    if (a->rb < l_pac && b->rb >= l_pac) return 0;
    if (a->qb >= b->qb || a->qe >= b->qe || a->re >= b->re) return 0;
    w = (int)((a->re - b->rb) - (a->qe - b->qb));
    w = w > 0 ? w : -w;
    r = (double)(a->re - b->rb) / (b->re - a->rb) - (double)(a->qe - b->qb) / (b->qe - a->qb);
    r = r > 0. ? r : -r;
    if (a->re < b->rb || a->qe < b->qb) {
        if (w > opt->w << 1 || r >= PATCH_MAX_R_BW) return 0;
    } else if (w > opt->w << 2 || r >= PATCH_MAX_R_BW * 2) return 0;
    w += a->w + b->w;
    w = w < opt->w << 2 ? w : opt->w << 2;
    gen_score(opt->mat, opt->o_del, opt->e_del, opt->o_ins, opt->e_ins, w, l_pac, ref_string, b->qe - a->qb, query + a->qb, a->rb, b->re, &score);
    q_s = (int)((double)(b->qe - a->qb) / ((b->qe - b->qb) + (a->qe - a->qb)) * (b->score + a->score) + .499);
    r_s = (int)((double)(b->re - a->rb) / ((b->re - b->rb) + (a->re - a->rb)) * (b->score + a->score) + .499);
    if ((double)score / (q_s > r_s ? q_s : r_s) < PATCH_MIN_SC_RATIO) return 0;
    *_w = w;
    return score;
}

// mem_sort_dedup_patch, bwamem.cpp:292-353
int sort_dedup_patch(const bm2_opt *opt, int64_t l_pac, const uint8_t *ref_string, const uint8_t *query, int n, bm2_alnreg_t *a) {
    int m, i, j;
    if (n <= 1) return n;
    k_introsort((size_t)n, a, [](const bm2_alnreg_t &x, const bm2_alnreg_t &y) { return x.re < y.re; });     // alnreg_slt2
    for (i = 0; i < n; ++i) a[i].n_comp = 1;
    for (i = 1; i < n; ++i) {
        bm2_alnreg_t *p = &a[i];
        if (p->rid != a[i - 1].rid || p->rb >= a[i - 1].re + opt->max_chain_gap) continue;
        for (j = i - 1; j >= 0 && p->rid == a[j].rid && p->rb < a[j].re + opt->max_chain_gap; --j) {
            bm2_alnreg_t *q = &a[j];
            int64_t or_, oq, mr, mq;
            int score, w;
            if (q->qe == q->qb) continue;
            or_ = q->re - p->rb;
            oq = q->qb < p->qb ? q->qe - p->qb : p->qe - q->qb;
            mr = q->re - q->rb < p->re - p->rb ? q->re - q->rb : p->re - p->rb;
            mq = q->qe - q->qb < p->qe - p->qb ? q->qe - q->qb : p->qe - p->qb;
            if (or_ > opt->mask_level_redun * mr && oq > opt->mask_level_redun * mq) {
                if (p->score < q->score) { p->qe = p->qb; break; }
                else q->qe = q->qb;
            } else if (q->rb < p->rb && (score = patch_reg(opt, l_pac, ref_string, query, q, p, &w)) > 0) {
                p->n_comp += q->n_comp + 1;
                p->seedcov = p->seedcov > q->seedcov ? p->seedcov : q->seedcov;
                p->sub = p->sub > q->sub ? p->sub : q->sub;
                p->csub = p->csub > q->csub ? p->csub : q->csub;
                p->qb = q->qb; p->rb = q->rb;
                p->truesc = p->score = score;
                p->w = w;
                q->qb = q->qe;
            }
        }
    }
    for (i = 0, m = 0; i < n; ++i)
        if (a[i].qe > a[i].qb) { if (m != i) a[m++] = a[i]; else ++m; }
    n = m;
    k_introsort((size_t)n, a, [](const bm2_alnreg_t &x, const bm2_alnreg_t &y) {                                 // alnreg_slt
        return x.score > y.score || (x.score == y.score && (x.rb < y.rb || (x.rb == y.rb && x.qb < y.qb))); });
    for (i = 1; i < n; ++i)
        if (a[i].score == a[i - 1].score && a[i].rb == a[i - 1].rb && a[i].qb == a[i - 1].qb) a[i].qe = a[i].qb;
    for (i = 1, m = 1; i < n; ++i)
        if (a[i].qe > a[i].qb) { if (m != i) a[m++] = a[i]; else ++m; }
    return m;
}

}  // namespace

int bm2h_sort_dedup_patch(const bm2_opt *opt, int64_t l_pac, const uint8_t *ref_string, const uint8_t *query, int n, bm2_alnreg_t *a) {
    return sort_dedup_patch(opt, l_pac, ref_string, query, n, a);
}

// The tail of mem_kernel2_core for a whole chunk: regs of the device boundary in, final mem_alnreg_v contents out.
extern "C" int bm2_finish_regs(const bm2_index_desc *idx, const bm2_opt *opt, const bm2_reads *reads, const bm2_reg_t *regs,
                               const int64_t *reg_off, bm2_alnreg_t *out, int64_t cap, int64_t *out_off, int64_t *n_out) {
    if (!idx || !opt || !reads || !reg_off || !out_off || !n_out || !idx->ref_string) { bm2_set_error("bm2_finish_regs: bad argument"); return BM2_EINVAL; }
    const int n = reads->n_reads;
    const int64_t n_in = reg_off[n];
    *n_out = n_in;
    if (n_in > cap) { bm2_set_error("bm2_finish_regs: capacity %ld < %ld", (long)cap, (long)n_in); return BM2_ECAP; }   // output never grows
    int64_t o = 0;
    for (int r = 0; r < n; r++) {
        const int k = (int)(reg_off[r + 1] - reg_off[r]);
        bm2_alnreg_t *a = out + o;
        for (int i = 0; i < k; i++) {
            const bm2_reg_t &s = regs[reg_off[r] + i];
            bm2_alnreg_t d; memset(&d, 0, sizeof d);
            d.rb = s.rb; d.re = s.re; d.qb = s.qb; d.qe = s.qe; d.rid = s.rid; d.score = s.score; d.truesc = s.truesc; d.w = s.w;
            d.seedcov = s.seedcov; d.seedlen0 = s.seedlen0; d.frac_rep = s.frac_rep;
            a[i] = d;
        }
        const int m = sort_dedup_patch(opt, idx->l_pac, idx->ref_string, reads->enc + reads->off[r], k, a);
        for (int i = 0; i < m; i++)                              // bwamem.cpp:1161-1169
            if (a[i].rid >= 0 && idx->ann_is_alt[a[i].rid]) a[i].is_alt = 1;
        out_off[r] = o;
        o += m;
    }
    out_off[n] = o;
    *n_out = o;
    return BM2_OK;
}
