// sam_se.cpp -- host side, one more row of SURVEY.md 8(f): the single-end branch of worker_sam (bwamem.cpp:1320-1335).
// Input: the mem_alnreg_v contents of every read (bm2_finish_regs).  Output: the SAM alignment lines, byte for byte what
// `bwa-mem2 mem` prints for single-end reads.  Plain C++ on the host: per read a sort, a few banded global alignments with
// backtrack (one per output record and per XA alternative) and text formatting -- branchy and order-sensitive.
#include <limits.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include "../../include/bm2.h"
#include "ksort_host.h"

void bm2_set_error(const char *fmt, ...);

namespace {

enum { F_ALL = 0x8, F_NO_MULTI = 0x10, F_REF_HDR = 0x100, F_SOFTCLIP = 0x200, F_PRIMARY5 = 0x800, F_KEEP_SUPP_MAPQ = 0x1000 };
const int MINUS_INF = -0x40000000;

struct Ref {                        // what bntseq_t + pac give this code
    int64_t l_pac; const uint8_t *ref_string; int n_seqs; const int64_t *off; const char *const *name; const char *const *anno;
    int64_t depos(int64_t pos, int *is_rev) const { return (*is_rev = (pos >= l_pac)) ? (l_pac << 1) - 1 - pos : pos; }   // bntseq.h:87-90
    int pos2rid(int64_t pos_f) const {                         // bntseq.cpp:378-392
        if (pos_f >= l_pac) return -1;
        int left = 0, mid = 0, right = n_seqs;
        while (left < right) {
            mid = (left + right) >> 1;
            if (pos_f >= off[mid]) {
                if (mid == n_seqs - 1) break;
                if (pos_f < off[mid + 1]) break;
                left = mid + 1;
            } else right = mid;
        }
        return mid;
    }
};

void put_int(std::string &s, long long v) { char b[32]; snprintf(b, sizeof b, "%lld", v); s += b; }       // kputw / kputl

// ---- ksw_global2 with backtrack (ksw.cpp:558-668): direction byte per cell = f<<4 | e<<2 | h ---------------------------
int global_align(int qlen, const uint8_t *query, int tlen, const uint8_t *target, const int8_t *mat, int o_del, int e_del, int o_ins,
                 int e_ins, int w, std::vector<uint32_t> &cigar) {
    struct EH { int32_t h, e; };
    const int oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
    const int n_col = qlen < 2 * w + 1 ? qlen : 2 * w + 1;
    std::vector<uint8_t> z((size_t)n_col * (size_t)tlen);
    std::vector<EH> eh((size_t)qlen + 1);
    int i, j, k;
    eh[0].h = 0; eh[0].e = MINUS_INF;
    for (j = 1; j <= qlen && j <= w; ++j) { eh[j].h = -(o_ins + e_ins * j); eh[j].e = MINUS_INF; }
    for (; j <= qlen; ++j) eh[j].h = eh[j].e = MINUS_INF;
    for (i = 0; i < tlen; ++i) {
        int32_t f = MINUS_INF, h1, beg, end, t;
        const int8_t *q = &mat[target[i] * 5];
        beg = i > w ? i - w : 0;
        end = i + w + 1 < qlen ? i + w + 1 : qlen;
        h1 = beg == 0 ? -(o_del + e_del * (i + 1)) : MINUS_INF;
        uint8_t *zi = &z[(size_t)i * n_col];
        for (j = beg; j < end; ++j) {
            EH *p = &eh[j];
            int32_t h, m = p->h, e = p->e;
            uint8_t d;
            p->h = h1;
            m += q[query[j]];
            d = m >= e ? 0 : 1;
            h = m >= e ? m : e;
            d = h >= f ? d : 2;
            h = h >= f ? h : f;
            h1 = h;
            t = m - oe_del; e -= e_del;
            d |= e > t ? 1 << 2 : 0;
            e = e > t ? e : t;
            p->e = e;
            t = m - oe_ins; f -= e_ins;
            d |= f > t ? 2 << 4 : 0;
            f = f > t ? f : t;
            zi[j - beg] = d;
        }
        eh[end].h = h1; eh[end].e = MINUS_INF;
    }
    const int score = eh[qlen].h;
    cigar.clear();
    auto push = [&](int op, int len) {                         // push_cigar, ksw.cpp:546-556
        if (cigar.empty() || op != (int)(cigar.back() & 0xf)) cigar.push_back((uint32_t)len << 4 | (uint32_t)op);
        else cigar.back() += (uint32_t)len << 4;
    };
    int which = 0;
    i = tlen - 1; k = (i + w + 1 < qlen ? i + w + 1 : qlen) - 1;
    while (i >= 0 && k >= 0) {
        which = z[(size_t)i * n_col + (k - (i > w ? i - w : 0))] >> (which << 1) & 3;
        if (which == 0) { push(0, 1); --i; --k; }
        else if (which == 1) { push(2, 1); --i; }
        else { push(1, 1); --k; }
    }
    if (i >= 0) push(2, i + 1);
    if (k >= 0) push(1, k + 1);
    for (size_t a = 0, b = cigar.size(); a + 1 < b; ++a, --b) { uint32_t t = cigar[a]; cigar[a] = cigar[b - 1]; cigar[b - 1] = t; }
    return score;
}

// ---- bwa_gen_cigar2 (bwa.cpp:260-347): CIGAR, score, NM and MD of query vs [rb, re).  false = the NULL return ----------
bool gen_cigar(const int8_t mat[25], int o_del, int e_del, int o_ins, int e_ins, int w_, const Ref &R, int l_query, const uint8_t *query,
               int64_t rb, int64_t re, int *score, std::vector<uint32_t> &cigar, int *NM, std::string &MD) {
    cigar.clear(); MD.clear(); *NM = -1;
    const int64_t l_pac = R.l_pac;
    if (l_query <= 0 || rb >= re || (rb < l_pac && re > l_pac)) return false;
    int64_t b = rb, e = re;                                     // bns_get_seq clamps (bntseq.cpp:320-345); a clamped range bails out
    if (e > (l_pac << 1)) e = l_pac << 1;
    if (b < 0) b = 0;
    if (e - b != re - rb) return false;
    const int64_t rlen = re - rb;
    std::vector<uint8_t> rseq(R.ref_string + rb, R.ref_string + re), q(query, query + l_query);
    if (rb >= l_pac) {                                          // reverse both: indels end up leftmost on the forward strand
        for (int i = 0; i < l_query >> 1; ++i) { uint8_t t = q[i]; q[i] = q[l_query - 1 - i]; q[l_query - 1 - i] = t; }
        for (int64_t i = 0; i < rlen >> 1; ++i) { uint8_t t = rseq[i]; rseq[i] = rseq[rlen - 1 - i]; rseq[rlen - 1 - i] = t; }
    }
    if (l_query == rlen && w_ == 0) {
        cigar.push_back((uint32_t)l_query << 4 | 0);
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
        *score = global_align(l_query, q.data(), (int)rlen, rseq.data(), mat, o_del, e_del, o_ins, e_ins, w, cigar);
    }
    {   // NM and MD (:311-340)
        int x = 0, y = 0, u = 0, n_mm = 0, n_gap = 0;
        const char *int2base = rb < l_pac ? "ACGTN" : "TGCAN";
        const int n = (int)cigar.size();
        for (int k = 0; k < n; ++k) {
            const int op = cigar[k] & 0xf, len = (int)(cigar[k] >> 4);
            if (op == 0) {
                for (int i = 0; i < len; ++i) {
                    if (q[x + i] != rseq[y + i]) { put_int(MD, u); MD.push_back(int2base[rseq[y + i]]); ++n_mm; u = 0; }
                    else ++u;
                }
                x += len; y += len;
            } else if (op == 2) {
                if (k > 0 && k < n - 1) {
                    put_int(MD, u); MD.push_back('^');
                    for (int i = 0; i < len; ++i) MD.push_back(int2base[rseq[y + i]]);
                    u = 0; n_gap += len;
                }
                y += len;
            } else if (op == 1) { x += len; n_gap += len; }
        }
        put_int(MD, u);
        *NM = n_mm + n_gap;
    }
    return true;
}

struct Aln {                        // mem_aln_t (bwamem.h:168-178)
    int64_t pos = -1; int rid = -1, flag = 0, is_rev = 0, is_alt = 0, mapq = 0, NM = 0;
    std::vector<uint32_t> cigar; std::string MD; const std::string *XA = nullptr;
    int score = 0, sub = 0, alt_sc = 0;
};

int infer_bw(int l1, int l2, int score, int a, int q, int r) {  // bwamem.cpp:1811-1818
    if (l1 == l2 && l1 * a - score < (q + r - a) << 1) return 0;
    int w = (int)((double)((l1 < l2 ? l1 : l2) * a - score - q) / r + 2.);
    if (w < abs(l1 - l2)) w = abs(l1 - l2);
    return w;
}

int approx_mapq_se(const bm2_opt *opt, const bm2_sam_opt *so, const bm2_alnreg_t *a) {          // bwamem.cpp:1470-1494
    int mapq, l, sub = a->sub ? a->sub : opt->min_seed_len * opt->a;
    double identity;
    sub = a->csub > sub ? a->csub : sub;
    if (sub >= a->score) return 0;
    l = a->qe - a->qb > a->re - a->rb ? a->qe - a->qb : (int)(a->re - a->rb);
    identity = 1. - (double)(l * opt->a - a->score) / (opt->a + opt->b) / l;
    if (a->score == 0) mapq = 0;
    else if (so->mapQ_coef_len > 0) {
        double tmp = l < so->mapQ_coef_len ? 1. : so->mapQ_coef_fac / log(l);
        tmp *= identity * identity;
        mapq = (int)(6.02 * (a->score - sub) / opt->a * tmp * tmp + .499);
    } else {
        mapq = (int)(30.0 * (1. - (double)sub / a->score) * log(a->seedcov) + .499);
        mapq = identity < 0.95 ? (int)(mapq * identity * identity + .499) : mapq;
    }
    if (a->sub_n > 0) mapq -= (int)(4.343 * log(a->sub_n + 1) + .499);
    if (mapq > 60) mapq = 60;
    if (mapq < 0) mapq = 0;
    mapq = (int)(mapq * (1. - a->frac_rep) + .499);
    return mapq;
}

// mem_reg2aln, bwamem.cpp:1732-1805; ar == NULL -> the unmapped record
bool reg2aln(const bm2_opt *opt, const bm2_sam_opt *so, const Ref &R, int l_query, const uint8_t *query, const bm2_alnreg_t *ar, Aln &a) {
    a = Aln();
    if (ar == 0 || ar->rb < 0 || ar->re < 0) { a.rid = -1; a.pos = -1; a.flag |= 0x4; return true; }
    const int qb = ar->qb, qe = ar->qe;
    const int64_t rb = ar->rb, re = ar->re;
    a.mapq = ar->secondary < 0 ? approx_mapq_se(opt, so, ar) : 0;
    if (ar->secondary >= 0) a.flag |= 0x100;
    int tmp = infer_bw(qe - qb, (int)(re - rb), ar->truesc, opt->a, opt->o_del, opt->e_del);
    int w2 = infer_bw(qe - qb, (int)(re - rb), ar->truesc, opt->a, opt->o_ins, opt->e_ins);
    w2 = w2 > tmp ? w2 : tmp;
    if (w2 > opt->w) w2 = w2 < ar->w ? w2 : ar->w;
    int i = 0, score = 0, NM = 0, last_sc = -(1 << 30);
    bool ok;
    do {
        w2 = w2 < opt->w << 2 ? w2 : opt->w << 2;
        ok = gen_cigar(opt->mat, opt->o_del, opt->e_del, opt->o_ins, opt->e_ins, w2, R, qe - qb, query + qb, rb, re, &score, a.cigar, &NM, a.MD);
        if (!ok) break;
        if (score == last_sc || w2 == opt->w << 2) break;
        last_sc = score;
        w2 <<= 1;
    } while (++i < 3 && score < ar->truesc - opt->a);
    if (!ok) return false;                                      // the reference asserts a.cigar != NULL here
    a.NM = NM;
    int is_rev;
    int64_t pos = R.depos(rb < R.l_pac ? rb : re - 1, &is_rev);
    a.is_rev = is_rev;
    if (!a.cigar.empty()) {                                     // squeeze out a leading or a trailing deletion
        if ((a.cigar[0] & 0xf) == 2) { pos += a.cigar[0] >> 4; a.cigar.erase(a.cigar.begin()); }
        else if ((a.cigar.back() & 0xf) == 2) a.cigar.pop_back();
    }
    if (qb != 0 || qe != l_query) {                             // clipping
        const int clip5 = is_rev ? l_query - qe : qb, clip3 = is_rev ? qb : l_query - qe;
        if (clip5) a.cigar.insert(a.cigar.begin(), (uint32_t)clip5 << 4 | 3);
        if (clip3) a.cigar.push_back((uint32_t)clip3 << 4 | 3);
    }
    a.rid = R.pos2rid(pos);
    if (a.rid < 0) return false;
    a.pos = pos - R.off[a.rid];
    a.score = ar->score; a.sub = ar->sub > ar->csub ? ar->sub : ar->csub;
    a.is_alt = ar->is_alt; a.alt_sc = ar->alt_sc;
    return true;
}

uint64_t hash_64(uint64_t key) {                                // utils.h:117-128
    key += ~(key << 32); key ^= (key >> 22); key += ~(key << 13); key ^= (key >> 8);
    key += (key << 3); key ^= (key >> 15); key += ~(key << 27); key ^= (key >> 31);
    return key;
}

// mem_mark_primary_se_core, bwamem.cpp:1392-1418
void mark_primary_core(const bm2_opt *opt, int n, bm2_alnreg_t *a, std::vector<int> &z) {
    int tmp = opt->a + opt->b;
    tmp = opt->o_del + opt->e_del > tmp ? opt->o_del + opt->e_del : tmp;
    tmp = opt->o_ins + opt->e_ins > tmp ? opt->o_ins + opt->e_ins : tmp;
    z.clear(); z.push_back(0);
    for (int i = 1; i < n; ++i) {
        size_t k;
        for (k = 0; k < z.size(); ++k) {
            const int j = z[k];
            const int b_max = a[j].qb > a[i].qb ? a[j].qb : a[i].qb, e_min = a[j].qe < a[i].qe ? a[j].qe : a[i].qe;
            if (e_min > b_max) {
                const int min_l = a[i].qe - a[i].qb < a[j].qe - a[j].qb ? a[i].qe - a[i].qb : a[j].qe - a[j].qb;
                if (e_min - b_max >= min_l * opt->mask_level) {
                    if (a[j].sub == 0) a[j].sub = a[i].score;
                    if (a[j].score - a[i].score <= tmp && (a[j].is_alt || !a[i].is_alt)) ++a[j].sub_n;
                    break;
                }
            }
        }
        if (k == z.size()) z.push_back(i);
        else a[i].secondary = z[k];
    }
}

// mem_mark_primary_se, bwamem.cpp:1420-1465
int mark_primary_se(const bm2_opt *opt, int n, bm2_alnreg_t *a, int64_t id) {
    if (n == 0) return 0;
    int n_pri = 0;
    std::vector<int> z;
    for (int i = 0; i < n; ++i) {
        a[i].sub = a[i].alt_sc = 0; a[i].secondary = a[i].secondary_all = -1; a[i].hash = hash_64((uint64_t)(id + i));
        if (!a[i].is_alt) ++n_pri;
    }
    k_introsort((size_t)n, a, [](const bm2_alnreg_t &x, const bm2_alnreg_t &y) {          // alnreg_hlt, bwamem.cpp:155
        return x.score > y.score || (x.score == y.score && (x.is_alt < y.is_alt || (x.is_alt == y.is_alt && x.hash < y.hash)));
    });
    mark_primary_core(opt, n, a, z);
    for (int i = 0; i < n; ++i) {
        bm2_alnreg_t *p = &a[i];
        p->secondary_all = i;
        if (!p->is_alt && p->secondary >= 0 && a[p->secondary].is_alt) p->alt_sc = a[p->secondary].score;
    }
    if (n_pri >= 0 && n_pri < n) {
        z.assign((size_t)n, 0);
        if (n_pri > 0)
            k_introsort((size_t)n, a, [](const bm2_alnreg_t &x, const bm2_alnreg_t &y) {  // alnreg_hlt2, bwamem.cpp:158
                return x.is_alt < y.is_alt || (x.is_alt == y.is_alt && (x.score > y.score || (x.score == y.score && x.hash < y.hash)));
            });
        for (int i = 0; i < n; ++i) z[a[i].secondary_all] = i;
        for (int i = 0; i < n; ++i) {
            if (a[i].secondary >= 0) {
                a[i].secondary_all = z[a[i].secondary];
                if (a[i].is_alt) a[i].secondary = INT_MAX;
            } else a[i].secondary_all = -1;
        }
        if (n_pri > 0) {
            for (int i = 0; i < n_pri; ++i) { a[i].sub = 0; a[i].secondary = -1; }
            mark_primary_core(opt, n_pri, a, z);
        }
    } else {
        for (int i = 0; i < n; ++i) a[i].secondary_all = a[i].secondary;
    }
    return n_pri;
}

// mem_reorder_primary5, bwamem.cpp:1496-1519
void reorder_primary5(int T, int n, bm2_alnreg_t *a) {
    int n_pri = 0, left_st = INT_MAX, left_k = -1;
    for (int k = 0; k < n; ++k) if (a[k].secondary < 0 && !a[k].is_alt && a[k].score >= T) ++n_pri;
    if (n_pri <= 1) return;
    for (int k = 0; k < n; ++k) {
        const bm2_alnreg_t *p = &a[k];
        if (p->secondary >= 0 || p->is_alt || p->score < T) continue;
        if (p->qb < left_st) { left_st = p->qb; left_k = k; }
    }
    if (left_k == 0) return;
    bm2_alnreg_t t = a[0]; a[0] = a[left_k]; a[left_k] = t;
    for (int k = 1; k < n; ++k) {
        bm2_alnreg_t *p = &a[k];
        if (p->secondary == 0) p->secondary = left_k;
        else if (p->secondary == left_k) p->secondary = 0;
        if (p->secondary_all == 0) p->secondary_all = left_k;
        else if (p->secondary_all == left_k) p->secondary_all = 0;
    }
}

void put_cigar(std::string &s, const std::vector<uint32_t> &cg, const char *ops) {
    for (uint32_t c : cg) { put_int(s, c >> 4); s.push_back(ops[c & 0xf]); }
}

// mem_gen_alt, bwamem_extra.cpp:118-183: the XA:Z value of every primary hit ("" = none)
bool gen_alt(const bm2_opt *opt, const bm2_sam_opt *so, const Ref &R, int n, const bm2_alnreg_t *a, int l_query, const uint8_t *query,
             std::vector<std::string> &XA, bool &any) {
    auto pri_idx = [&](int i) {
        const int k = a[i].secondary_all;
        return (k >= 0 && a[i].score >= a[k].score * (double)so->XA_drop_ratio) ? k : -1;      // (a double parameter in the reference)
    };
    std::vector<int> cnt((size_t)n, 0); std::vector<char> has_alt((size_t)n, 0);
    int tot = 0;
    any = false;
    for (int i = 0; i < n; ++i) {
        const int r = pri_idx(i);
        if (r >= 0) { ++cnt[r]; ++tot; if (a[i].is_alt) has_alt[r] = 1; }
    }
    if (tot == 0) return true;
    any = true;
    XA.assign((size_t)n, std::string());
    for (int i = 0; i < n; ++i) {
        const int r = pri_idx(i);
        if (r < 0) continue;
        if (cnt[r] > so->max_XA_hits_alt || (!has_alt[r] && cnt[r] > so->max_XA_hits)) continue;
        Aln t;
        if (!reg2aln(opt, so, R, l_query, query, &a[i], t)) return false;
        std::string &s = XA[r];
        s += R.name[t.rid]; s.push_back(','); s.push_back("+-"[t.is_rev]); put_int(s, t.pos + 1); s.push_back(',');
        put_cigar(s, t.cigar, "MIDSHN");
        s.push_back(','); put_int(s, t.NM); s.push_back(';');
    }
    return true;
}

// add_cigar, bwamem.cpp:1579-1590
void add_cigar(const bm2_sam_opt *so, const Aln &p, std::string &s, int which) {
    if (p.cigar.empty()) { s.push_back('*'); return; }
    for (uint32_t cg : p.cigar) {
        int c = cg & 0xf;
        if (!(so->flag & F_SOFTCLIP) && !p.is_alt && (c == 3 || c == 4)) c = which ? 4 : 3;
        put_int(s, cg >> 4); s.push_back("MIDSH"[c]);
    }
}

// mem_aln2sam without a mate (bwamem.cpp:1592-1730)
void aln2sam(const bm2_sam_opt *so, const Ref &R, std::string &s, const char *name, const char *comment, const char *qual, int l_seq,
             const uint8_t *seq, const std::vector<Aln> &list, int which, const std::vector<bool> &has_xa) {
    Aln p = list[which];
    const int n = (int)list.size();
    p.flag |= p.rid < 0 ? 0x4 : 0;
    p.flag |= p.is_rev ? 0x10 : 0;
    s += name; s.push_back('\t');
    put_int(s, (p.flag & 0xffff) | (p.flag & 0x10000 ? 0x100 : 0)); s.push_back('\t');
    if (p.rid >= 0) {
        s += R.name[p.rid]; s.push_back('\t');
        put_int(s, p.pos + 1); s.push_back('\t');
        put_int(s, p.mapq); s.push_back('\t');
        add_cigar(so, p, s, which);
    } else s += "*\t0\t0\t*";
    s.push_back('\t');
    s += "*\t0\t0";
    s.push_back('\t');
    if (p.flag & 0x100) s += "*\t*";
    else {
        int qb = 0, qe = l_seq;
        if (!p.cigar.empty() && which && !(so->flag & F_SOFTCLIP) && !p.is_alt) {
            const uint32_t c0 = p.cigar[0], c1 = p.cigar.back();
            if (!p.is_rev) {
                if ((c0 & 0xf) == 4 || (c0 & 0xf) == 3) qb += c0 >> 4;
                if ((c1 & 0xf) == 4 || (c1 & 0xf) == 3) qe -= c1 >> 4;
            } else {
                if ((c0 & 0xf) == 4 || (c0 & 0xf) == 3) qe -= c0 >> 4;
                if ((c1 & 0xf) == 4 || (c1 & 0xf) == 3) qb += c1 >> 4;
            }
        }
        if (!p.is_rev) {
            for (int i = qb; i < qe; ++i) s.push_back("ACGTN"[seq[i]]);
            s.push_back('\t');
            if (qual) s.append(qual + qb, (size_t)(qe - qb)); else s.push_back('*');
        } else {
            for (int i = qe - 1; i >= qb; --i) s.push_back("TGCAN"[seq[i]]);
            s.push_back('\t');
            if (qual) { for (int i = qe - 1; i >= qb; --i) s.push_back(qual[i]); } else s.push_back('*');
        }
    }
    if (!p.cigar.empty()) { s += "\tNM:i:"; put_int(s, p.NM); s += "\tMD:Z:"; s += p.MD; }
    if (p.score >= 0) { s += "\tAS:i:"; put_int(s, p.score); }
    if (p.sub >= 0) { s += "\tXS:i:"; put_int(s, p.sub); }
    if (so->rg_id && so->rg_id[0]) { s += "\tRG:Z:"; s += so->rg_id; }
    if (!(p.flag & 0x100)) {
        int i;
        for (i = 0; i < n; ++i) if (i != which && !(list[i].flag & 0x100)) break;
        if (i < n) {
            s += "\tSA:Z:";
            for (i = 0; i < n; ++i) {
                const Aln &r = list[i];
                if (i == which || (r.flag & 0x100)) continue;
                s += R.name[r.rid]; s.push_back(','); put_int(s, r.pos + 1); s.push_back(','); s.push_back("+-"[r.is_rev]); s.push_back(',');
                put_cigar(s, r.cigar, "MIDSH");
                s.push_back(','); put_int(s, r.mapq); s.push_back(','); put_int(s, r.NM); s.push_back(';');
            }
        }
        if (p.alt_sc > 0) { char b[64]; snprintf(b, sizeof b, "\tpa:f:%.3f", (double)p.score / p.alt_sc); s += b; }
    }
    if (has_xa[which] && p.XA) { s += "\tXA:Z:"; s += *p.XA; }
    if (comment) { s.push_back('\t'); s += comment; }
    if ((so->flag & F_REF_HDR) && p.rid >= 0 && R.anno && R.anno[p.rid] && R.anno[p.rid][0]) {
        s += "\tXR:Z:";
        for (const char *c = R.anno[p.rid]; *c; ++c) s.push_back(*c == '\t' ? ' ' : *c);
    }
    s.push_back('\n');
}

// mem_reg2sam with extra_flag == 0 and no mate (bwamem.cpp:1521-1577)
bool reg2sam(const bm2_opt *opt, const bm2_sam_opt *so, const Ref &R, std::string &out, const char *name, const char *comment,
             const char *qual, int l_seq, const uint8_t *seq, int n, const bm2_alnreg_t *a) {
    std::vector<std::string> XA; bool any_xa = false;
    if (!(so->flag & F_ALL)) { if (!gen_alt(opt, so, R, n, a, l_seq, seq, XA, any_xa)) return false; }
    std::vector<Aln> aa; std::vector<bool> has_xa;
    int l = 0;
    for (int k = 0; k < n; ++k) {
        const bm2_alnreg_t *p = &a[k];
        if (p->score < so->T) continue;
        if (p->secondary >= 0 && (p->is_alt || !(so->flag & F_ALL))) continue;
        if (p->secondary >= 0 && p->secondary < INT_MAX && p->score < a[p->secondary].score * opt->drop_ratio) continue;
        aa.emplace_back();
        Aln &q = aa.back();
        if (!reg2aln(opt, so, R, l_seq, seq, p, q)) return false;
        const bool xa = any_xa && !XA[k].empty();               // XA[k] is a NULL pointer in the reference when nothing was appended
        has_xa.push_back(xa);
        q.XA = xa ? &XA[k] : nullptr;
        if (p->secondary >= 0) q.sub = -1;
        if (l && p->secondary < 0) q.flag |= (so->flag & F_NO_MULTI) ? 0x10000 : 0x800;
        if (!(so->flag & F_KEEP_SUPP_MAPQ) && l && !p->is_alt && q.mapq > aa[0].mapq) q.mapq = aa[0].mapq;
        ++l;
    }
    if (aa.empty()) {
        aa.emplace_back();
        reg2aln(opt, so, R, l_seq, seq, 0, aa[0]);
        has_xa.assign(1, false);
        aln2sam(so, R, out, name, comment, qual, l_seq, seq, aa, 0, has_xa);
    } else {
        for (int k = 0; k < (int)aa.size(); ++k) aln2sam(so, R, out, name, comment, qual, l_seq, seq, aa, k, has_xa);
    }
    return true;
}

}  // namespace

extern "C" void bm2_sam_opt_init(bm2_sam_opt *o) {
    if (!o) return;
    memset(o, 0, sizeof *o);
    o->T = 30; o->flag = 0; o->max_XA_hits = 5; o->max_XA_hits_alt = 200; o->XA_drop_ratio = 0.80f;
    o->mapQ_coef_len = 50; o->mapQ_coef_fac = (int32_t)log(o->mapQ_coef_len);     // an int in mem_opt_t: 3
    o->rg_id = 0;
}

extern "C" int bm2_sam_se(const bm2_index_desc *idx, const bm2_opt *opt, const bm2_sam_opt *so, const bm2_reads *reads,
                          const bm2_read_text *txt, bm2_alnreg_t *alnregs, const int64_t *reg_off, int64_t n_processed, char *out,
                          int64_t cap, int64_t *n_out) {
    if (!idx || !opt || !so || !reads || !txt || !txt->name || !reg_off || !n_out || (!alnregs && reg_off[reads->n_reads] > 0)) {
        bm2_set_error("bm2_sam_se: bad argument"); return BM2_EINVAL;
    }
    if (!idx->ref_string || !idx->ann_offset || !idx->ann_name) { bm2_set_error("bm2_sam_se: the index descriptor needs ref_string and contig names"); return BM2_EINVAL; }
    Ref R = { idx->l_pac, idx->ref_string, idx->n_seqs, idx->ann_offset, idx->ann_name, idx->ann_anno };
    std::string s;
    for (int i = 0; i < reads->n_reads; ++i) {
        bm2_alnreg_t *a = alnregs + reg_off[i];
        const int n = (int)(reg_off[i + 1] - reg_off[i]);
        mark_primary_se(opt, n, a, n_processed + i);
        if (so->flag & F_PRIMARY5) reorder_primary5(so->T, n, a);
        if (!reg2sam(opt, so, R, s, txt->name[i], txt->comment ? txt->comment[i] : 0, txt->qual ? txt->qual[i] : 0, reads->len[i],
                     reads->enc + reads->off[i], n, a)) {
            bm2_set_error("bm2_sam_se: read %d has a hit whose CIGAR cannot be generated (range outside the reference)", i);
            return BM2_EINVAL;
        }
    }
    *n_out = (int64_t)s.size();
    if ((int64_t)s.size() > cap) return BM2_ECAP;
    if (out && !s.empty()) memcpy(out, s.data(), s.size());
    return BM2_OK;
}
