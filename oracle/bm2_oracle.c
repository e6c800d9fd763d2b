/* oracle/bm2_oracle.c
 *
 * TEST INFRASTRUCTURE ONLY.  A plain-C, single-threaded restatement of the bwa-mem2 v2.2.1
 * hot path (seed -> chain -> extend), written from the reference's behaviour, each function
 * citing the reference file:line it follows (paths relative to /root/reference/src).
 * It is the CHECKER for the HIP path: only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it.  The product (bwa-mem2_amd/) never links or calls it.
 *
 * Parity pin: the reference holds no golden vectors for this path, so the
 * restatement is pinned against the reference itself: oracle/refdump.cpp links the compiled
 * reference and dumps every stage (SMEM, SA coords, chains, regs); tests/test_oracle_vs_ref.py
 * compares stage by stage, and small dumps are committed under tests/golden/.
 */
#define _GNU_SOURCE
#include "bm2_oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <limits.h>

#define ORA_BLOCK 512          /* BATCH_SIZE, macro.h:48 */
#define ORA_SEEDS_PER_CHAIN 1  /* macro.h:50 (only matters for allocation in the reference) */
#define ORA_MAX_BAND_TRY 2     /* bwamem.cpp:51 */
#define ORA_H0 (-99)           /* H0_, macro.h:44 */

#define VEC(T) struct { T *a; int64_t n, m; }
#define vec_push(T, v, x) do { if ((v).n == (v).m) { (v).m = (v).m ? (v).m * 2 : 16; \
        (v).a = (T *)realloc((v).a, (size_t)(v).m * sizeof(T)); } (v).a[(v).n++] = (x); } while (0)

/* ------------------------------------------------------------------ index files */

static void *read_file(const char *fn, int64_t *size) {
    FILE *f = fopen(fn, "rb");
    if (!f) return 0;
    fseek(f, 0, SEEK_END); int64_t n = ftell(f); fseek(f, 0, SEEK_SET);
    void *p = malloc(n > 0 ? (size_t)n : 1);
    if (p && n > 0 && (int64_t)fread(p, 1, (size_t)n, f) != n) { free(p); p = 0; }
    fclose(f);
    if (size) *size = n;
    return p;
}

/* layout of <prefix>.bwt.2bit.64: FMI_search.cpp:162-164,252,275-276,296 (write) / :415-457 (read);
 * .ann text: bntseq.cpp:118-147; .alt: bntseq.cpp:201-226; .0123: fastmap.cpp:860-888 */
ora_index *ora_index_load(const char *prefix) {
    char fn[4096];
    ora_index *ix = (ora_index *)calloc(1, sizeof(*ix));
    snprintf(fn, sizeof fn, "%s.bwt.2bit.64", prefix);
    FILE *f = fopen(fn, "rb");
    if (!f) { free(ix); return 0; }
    int ok = fread(&ix->ref_len, 8, 1, f) == 1 && fread(ix->count, 8, 5, f) == 5;
    int64_t nocc = (ix->ref_len >> 6) + 1, nsa = (ix->ref_len >> 3) + 1;
    ix->cp_occ = (ora_cpocc *)malloc((size_t)nocc * sizeof(ora_cpocc));
    ix->sa_ms_byte = (int8_t *)malloc((size_t)nsa);
    ix->sa_ls_word = (uint32_t *)malloc((size_t)nsa * 4);
    ok = ok && (int64_t)fread(ix->cp_occ, sizeof(ora_cpocc), (size_t)nocc, f) == nocc;
    ok = ok && (int64_t)fread(ix->sa_ms_byte, 1, (size_t)nsa, f) == nsa;
    ok = ok && (int64_t)fread(ix->sa_ls_word, 4, (size_t)nsa, f) == nsa;
    ok = ok && fread(&ix->sentinel_index, 8, 1, f) == 1;
    fclose(f);
    for (int i = 0; i < 5; i++) ix->count[i] += 1;            /* FMI_search.cpp:433-436 */
    if (!ok) { ora_index_free(ix); return 0; }
    snprintf(fn, sizeof fn, "%s.ann", prefix);
    f = fopen(fn, "r");
    if (!f) { ora_index_free(ix); return 0; }
    long long xx; unsigned seed;
    if (fscanf(f, "%lld%d%u", &xx, &ix->n_seqs, &seed) != 3) { fclose(f); ora_index_free(ix); return 0; }
    ix->l_pac = xx;
    ix->ann_offset = (int64_t *)calloc((size_t)ix->n_seqs, 8);
    ix->ann_len = (int32_t *)calloc((size_t)ix->n_seqs, 4);
    ix->ann_is_alt = (int32_t *)calloc((size_t)ix->n_seqs, 4);
    ix->ann_name = (char **)calloc((size_t)ix->n_seqs, sizeof(char *));
    for (int i = 0; i < ix->n_seqs; i++) {
        unsigned gi; char name[8193]; int c, namb;
        if (fscanf(f, "%u%8192s", &gi, name) != 2) { fclose(f); ora_index_free(ix); return 0; }
        ix->ann_name[i] = strdup(name);
        while ((c = fgetc(f)) != '\n' && c != EOF) {}
        if (fscanf(f, "%lld%d%d", &xx, &ix->ann_len[i], &namb) != 3) { fclose(f); ora_index_free(ix); return 0; }
        ix->ann_offset[i] = xx;
    }
    fclose(f);
    snprintf(fn, sizeof fn, "%s.alt", prefix);
    if ((f = fopen(fn, "r")) != 0) {
        char line[8192];
        while (fgets(line, sizeof line, f)) {
            size_t l = strcspn(line, "\t\r\n"); line[l] = 0;
            if (line[0] == '@') continue;
            for (int i = 0; i < ix->n_seqs; i++) if (!strcmp(line, ix->ann_name[i])) ix->ann_is_alt[i] = 1;
        }
        fclose(f);
    }
    int64_t sz;
    snprintf(fn, sizeof fn, "%s.0123", prefix);
    ix->ref_string = (uint8_t *)read_file(fn, &sz);
    if (!ix->ref_string || sz != 2 * ix->l_pac) { ora_index_free(ix); return 0; }
    snprintf(fn, sizeof fn, "%s.pac", prefix);
    ix->pac = (uint8_t *)read_file(fn, &sz);
    return ix;
}

void ora_index_free(ora_index *ix) {
    if (!ix) return;
    free(ix->cp_occ); free(ix->sa_ms_byte); free(ix->sa_ls_word); free(ix->ref_string); free(ix->pac);
    free(ix->ann_offset); free(ix->ann_len); free(ix->ann_is_alt);
    if (ix->ann_name) for (int i = 0; i < ix->n_seqs; i++) free(ix->ann_name[i]);
    free(ix->ann_name); free(ix);
}

/* bwa_fill_scmat, bwa.cpp:248-257 */
void ora_opt_fill_scmat(ora_opt *o) {
    int k = 0;
    for (int i = 0; i < 4; i++) {
        for (int j = 0; j < 4; j++) o->mat[k++] = (int8_t)(i == j ? o->a : -o->b);
        o->mat[k++] = -1;
    }
    for (int j = 0; j < 5; j++) o->mat[k++] = -1;
}

/* mem_opt_init, bwamem.cpp:107-143 */
void ora_opt_init(ora_opt *o) {
    memset(o, 0, sizeof(*o));
    o->a = 1; o->b = 4; o->o_del = o->o_ins = 6; o->e_del = o->e_ins = 1;
    o->w = 100; o->zdrop = 100; o->pen_clip5 = o->pen_clip3 = 5;
    o->max_mem_intv = 20; o->min_seed_len = 19; o->split_width = 10; o->max_occ = 500;
    o->max_chain_gap = 10000; o->mask_level = 0.50f; o->drop_ratio = 0.50f; o->split_factor = 1.5f;
    o->mask_level_redun = 0.95f; o->min_chain_weight = 0; o->max_chain_extend = 1 << 30;
    ora_opt_fill_scmat(o);
}

/* ------------------------------------------------------------------ FM-index primitives */

typedef struct { int64_t k, l, s; } bi_t;     /* bi-interval part of SMEM, FMI_search.h:75-83 */
typedef struct { int64_t n_ext, n_same, n_lf, n_sa; } fm_stat;

/* GET_OCC, FMI_search.h:66-73; mask[y] = top y bits set, FMI_search.cpp:386-394 */
static inline int64_t occ_at(const ora_index *ix, int64_t pp, int c) {
    const ora_cpocc *b = &ix->cp_occ[pp >> 6];
    int y = (int)(pp & 63);
    uint64_t mask = y ? (~0ULL << (64 - y)) : 0;
    return b->cp_count[c] + __builtin_popcountll(b->bwt[c] & mask);
}

/* FMI_search::backwardExt, FMI_search.cpp:1025-1052 */
static bi_t backward_ext(const ora_index *ix, bi_t in, int a, fm_stat *st) {
    int64_t k[4], l[4], s[4];
    for (int b = 0; b < 4; b++) {
        int64_t o_sp = occ_at(ix, in.k, b), o_ep = occ_at(ix, in.k + in.s, b);
        k[b] = ix->count[b] + o_sp;
        s[b] = o_ep - o_sp;
    }
    int64_t sent = (in.k <= ix->sentinel_index && in.k + in.s > ix->sentinel_index) ? 1 : 0;
    l[3] = in.l + sent; l[2] = l[3] + s[3]; l[1] = l[2] + s[2]; l[0] = l[1] + s[1];
    if (st) { st->n_ext++; if ((in.k >> 6) == ((in.k + in.s) >> 6)) st->n_same++; }
    bi_t out = { k[a], l[a], s[a] };
    return out;
}

/* forward extension = backward extension on the swapped interval with the complement base,
 * FMI_search.cpp:546-554 */
static bi_t forward_ext(const ora_index *ix, bi_t in, int a, fm_stat *st) {
    bi_t sw = { in.l, in.k, in.s };
    bi_t r = backward_ext(ix, sw, 3 - a, st);
    bi_t out = { r.l, r.k, r.s };
    return out;
}

typedef struct { uint32_t rid, m, n; bi_t iv; } smem_t;
typedef VEC(smem_t) smem_v;

/* one (read, start position) step of FMI_search::getSMEMsOnePosOneThread, FMI_search.cpp:514-668.
 * Returns next_x. */
static int smem_one_pos(const ora_index *ix, const uint8_t *q, int len, uint32_t rid, int x, int64_t min_intv,
                        int min_seed_len, smem_v *out, smem_t *prev, fm_stat *st) {
    int next_x = x + 1;
    int a = q[x];
    if (a >= 4) return next_x;                                      /* :525, A.1 item 1 */
    smem_t sm; sm.rid = rid; sm.m = (uint32_t)x; sm.n = (uint32_t)x;
    sm.iv.k = ix->count[a]; sm.iv.l = ix->count[3 - a]; sm.iv.s = ix->count[a + 1] - ix->count[a];   /* :531-533 */
    int n_prev = 0, j;
    for (j = x + 1; j < len; j++) {                                   /* forward phase :537-575 */
        a = q[j];
        next_x = j + 1;
        if (a >= 4) break;
        smem_t ns = sm;
        ns.iv = forward_ext(ix, sm.iv, a, st);
        ns.n = (uint32_t)j;
        prev[n_prev] = sm;
        n_prev += (ns.iv.s != sm.iv.s);                               /* :556-559 */
        if (ns.iv.s < min_intv) { next_x = j; break; }                /* :560-564 */
        sm = ns;
    }
    if (sm.iv.s >= min_intv) prev[n_prev++] = sm;                     /* :576-581 */
    for (int p = 0; p < n_prev / 2; p++) { smem_t t = prev[p]; prev[p] = prev[n_prev - 1 - p]; prev[n_prev - 1 - p] = t; }
    for (j = x - 1; j >= 0; j--) {                                    /* backward phase :596-655 */
        int n_curr = 0, p;
        int32_t curr_s = -1;                                          /* an int in the reference (:599) */
        a = q[j];
        if (a > 3) break;
        for (p = 0; p < n_prev; p++) {
            smem_t s0 = prev[p], ns = s0;
            ns.iv = backward_ext(ix, s0.iv, a, st);
            ns.m = (uint32_t)j;
            if (ns.iv.s < min_intv && (int)(s0.n - s0.m + 1) >= min_seed_len) {
                vec_push(smem_t, *out, s0);
                break;
            }
            if (ns.iv.s >= min_intv && ns.iv.s != (int64_t)curr_s) {
                curr_s = (int32_t)ns.iv.s;
                prev[n_curr++] = ns;
                break;
            }
        }
        p++;
        for (; p < n_prev; p++) {
            smem_t s0 = prev[p], ns = s0;
            ns.iv = backward_ext(ix, s0.iv, a, st);
            ns.m = (uint32_t)j;
            if (ns.iv.s >= min_intv && ns.iv.s != (int64_t)curr_s) {
                curr_s = (int32_t)ns.iv.s;
                prev[n_curr++] = ns;
            }
        }
        n_prev = n_curr;
        if (n_curr == 0) break;
    }
    if (n_prev != 0) {                                                /* :656-665 */
        smem_t s0 = prev[0];
        if ((int)(s0.n - s0.m + 1) >= min_seed_len) vec_push(smem_t, *out, s0);
    }
    return next_x;
}

/* FMI_search::bwtSeedStrategyAllPosOneThread for one read, FMI_search.cpp:740-810 */
static void smem_pass3(const ora_index *ix, const uint8_t *q, int len, uint32_t rid, int64_t max_intv,
                       int min_seed_len, smem_v *out, fm_stat *st) {
    int x = 0;
    while (x < len) {
        int next_x = x + 1;
        int a = q[x];
        if (a < 4) {
            smem_t sm; sm.rid = rid; sm.m = (uint32_t)x; sm.n = (uint32_t)x;
            sm.iv.k = ix->count[a]; sm.iv.l = ix->count[3 - a]; sm.iv.s = ix->count[a + 1] - ix->count[a];
            for (int j = x + 1; j < len; j++) {
                next_x = j + 1;
                a = q[j];
                if (a >= 4) break;
                sm.iv = forward_ext(ix, sm.iv, a, st);
                sm.n = (uint32_t)j;
                if (sm.iv.s < max_intv && (int)(sm.n - sm.m + 1) >= min_seed_len) {
                    if (sm.iv.s > 0) vec_push(smem_t, *out, sm);
                    break;
                }
            }
        }
        x = next_x;
    }
}

static int smem_cmp(const void *pa, const void *pb) {   /* final order (rid, m, n): FMI_search.cpp:987-1006 + bwamem.cpp:45-46,787-799 */
    const smem_t *a = (const smem_t *)pa, *b = (const smem_t *)pb;
    if (a->rid != b->rid) return a->rid < b->rid ? -1 : 1;
    if (a->m != b->m) return a->m < b->m ? -1 : 1;
    if (a->n != b->n) return a->n < b->n ? -1 : 1;
    return 0;   /* equal (rid,m,n) => same substring => identical k,l,s */
}

/* mem_collect_smem for one block, bwamem.cpp:626-803 */
static void collect_smem_block(const ora_index *ix, const ora_opt *opt, int nseq, const uint8_t *enc,
                               const int64_t *off, const int32_t *len, smem_v *out, fm_stat *st) {
    int split_len = (int)(opt->min_seed_len * opt->split_factor + .499);   /* :639 */
    int max_len = 1;
    for (int l = 0; l < nseq; l++) if (len[l] > max_len) max_len = len[l];
    smem_t *prev = (smem_t *)malloc((size_t)(max_len + 1) * sizeof(smem_t));
    out->n = 0;
    for (int l = 0; l < nseq; l++) {                     /* pass 1: FMI_search.cpp:672-724 (all start positions) */
        int x = 0;
        while (x < len[l]) x = smem_one_pos(ix, enc + off[l], len[l], (uint32_t)l, x, 1, opt->min_seed_len, out, prev, st);
    }
    int64_t n1 = out->n;
    for (int64_t i = 0; i < n1; i++) {                   /* pass 2: bwamem.cpp:695-714, 742-753 */
        smem_t p = out->a[i];
        int start = (int)p.m, end = (int)p.n + 1;
        if (end - start < split_len || p.iv.s > opt->split_width) continue;
        smem_one_pos(ix, enc + off[p.rid], len[p.rid], p.rid, (end + start) >> 1, p.iv.s + 1, opt->min_seed_len, out, prev, st);
    }
    if (opt->max_mem_intv > 0)                           /* pass 3: bwamem.cpp:755-781 */
        for (int l = 0; l < nseq; l++)
            smem_pass3(ix, enc + off[l], len[l], (uint32_t)l, opt->max_mem_intv, opt->min_seed_len + 1, out, st);
    qsort(out->a, (size_t)out->n, sizeof(smem_t), smem_cmp);
    free(prev);
}

/* get_sa_entries_prefetch + call_one_step for ONE position, FMI_search.cpp:1202-1255 */
static int64_t sa_lookup(const ora_index *ix, int64_t pos, fm_stat *st) {
    int64_t sp = pos, offset = 0;
    if (st) st->n_sa++;
    if ((sp & 7) == 0) return ((int64_t)ix->sa_ms_byte[sp >> 3] << 32) + ix->sa_ls_word[sp >> 3];
    for (;;) {
        const ora_cpocc *blk = &ix->cp_occ[sp >> 6];
        int y = 63 - (int)(sp & 63), b;
        if ((blk->bwt[0] >> y) & 1) b = 0;
        else if ((blk->bwt[1] >> y) & 1) b = 1;
        else if ((blk->bwt[2] >> y) & 1) b = 2;
        else if ((blk->bwt[3] >> y) & 1) b = 3;
        else return 0;                                   /* sentinel: value 0 whatever the offset, :1230-1233 (A.2 item 11) */
        if (st) st->n_lf++;
        sp = ix->count[b] + occ_at(ix, sp, b);
        offset++;
        if ((sp & 7) == 0)
            return ((int64_t)ix->sa_ms_byte[sp >> 3] << 32) + ix->sa_ls_word[sp >> 3] + offset;
    }
}

/* ------------------------------------------------------------------ bntseq helpers */

/* bns_depos bntseq.h:87-90, bns_pos2rid bntseq.cpp:378-392, bns_intv2rid :394-402 */
static inline int64_t depos(const ora_index *ix, int64_t pos, int *is_rev) {
    return (*is_rev = (pos >= ix->l_pac)) ? (ix->l_pac << 1) - 1 - pos : pos;
}
static int pos2rid(const ora_index *ix, int64_t pos_f) {
    int left = 0, mid = 0, right = ix->n_seqs;
    if (pos_f >= ix->l_pac) return -1;
    while (left < right) {
        mid = (left + right) >> 1;
        if (pos_f >= ix->ann_offset[mid]) {
            if (mid == ix->n_seqs - 1) break;
            if (pos_f < ix->ann_offset[mid + 1]) break;
            left = mid + 1;
        } else right = mid;
    }
    return mid;
}
static int intv2rid(const ora_index *ix, int64_t rb, int64_t re) {
    int is_rev, rid_b, rid_e;
    if (rb < ix->l_pac && re > ix->l_pac) return -2;
    rid_b = pos2rid(ix, depos(ix, rb, &is_rev));
    rid_e = rb < re ? pos2rid(ix, depos(ix, re - 1, &is_rev)) : rid_b;
    return rid_b == rid_e ? rid_b : -1;
}

/* ------------------------------------------------------------------ klib introsort (ksort.h:185-236) */

typedef int (*lt_fn)(const void *, const void *);
static void o_swap(char *a, char *b, size_t sz) { char t[256]; memcpy(t, a, sz); memcpy(a, b, sz); memcpy(b, t, sz); }
static void o_insertsort(char *s, char *t, size_t sz, lt_fn lt) {        /* __ks_insertsort, ksort.h:155-162 */
    for (char *i = s + sz; i < t; i += sz)
        for (char *j = i; j > s && lt(j, j - sz); j -= sz) o_swap(j, j - sz, sz);
}
static void o_combsort(size_t n, char *a, size_t sz, lt_fn lt) {         /* ks_combsort, ksort.h:163-184 */
    const double shrink = 1.2473309501039786540366528676643;
    int do_swap; size_t gap = n;
    do {
        if (gap > 2) { gap = (size_t)(gap / shrink); if (gap == 9 || gap == 10) gap = 11; }
        do_swap = 0;
        for (char *i = a; i < a + (n - gap) * sz; i += sz) {
            char *j = i + gap * sz;
            if (lt(j, i)) { o_swap(i, j, sz); do_swap = 1; }
        }
    } while (do_swap || gap > 2);
    if (gap != 1) o_insertsort(a, a + n * sz, sz, lt);
}
static void ora_introsort(void *base, size_t n, size_t sz, lt_fn lt) {   /* ks_introsort, ksort.h:185-236 */
    char *a = (char *)base, rp[256];
    struct { char *left, *right; int depth; } stack[136], *top = stack;
    int d;
    if (n < 1) return;
    if (n == 2) { if (lt(a + sz, a)) o_swap(a, a + sz, sz); return; }
    for (d = 2; (1ul << d) < n; ++d) {}
    char *s = a, *t = a + (n - 1) * sz;
    d <<= 1;
    for (;;) {
        if (s < t) {
            if (--d == 0) { o_combsort((size_t)(t - s) / sz + 1, s, sz, lt); t = s; continue; }
            char *i = s, *j = t, *k = i + (((size_t)(j - i) / sz) >> 1) * sz + sz;
            if (lt(k, i)) { if (lt(k, j)) k = j; }
            else k = lt(j, i) ? i : j;
            memcpy(rp, k, sz);
            if (k != t) o_swap(k, t, sz);
            for (;;) {
                do i += sz; while (lt(i, rp));
                do j -= sz; while (i <= j && lt(rp, j));
                if (j <= i) break;
                o_swap(i, j, sz);
            }
            o_swap(i, t, sz);
            if (i - s > t - i) {
                if ((size_t)(i - s) > 16 * sz) { top->left = s; top->right = i - sz; top->depth = d; ++top; }
                s = (size_t)(t - i) > 16 * sz ? i + sz : t;
            } else {
                if ((size_t)(t - i) > 16 * sz) { top->left = i + sz; top->right = t; top->depth = d; ++top; }
                t = (size_t)(i - s) > 16 * sz ? i - sz : s;
            }
        } else {
            if (top == stack) { o_insertsort(a, a + n * sz, sz, lt); return; }
            --top; s = top->left; t = top->right; d = top->depth;
        }
    }
}

/* ------------------------------------------------------------------ chaining */

typedef struct { int64_t rbeg; int32_t qbeg, len, score, aln; } seed_t;      /* mem_seed_t, bwamem.h:113-124 */
typedef struct {                                                            /* mem_chain_t, bwamem.h:126-133 */
    int32_t seqid, n, m, first, rid, w, kept, is_alt;
    float frac_rep; int64_t pos; seed_t *seeds;
} chain_t;
typedef VEC(chain_t) chain_v;

/* klib B-tree with t = 5 (kb_init(chn, 512+8) with 48-byte keys: kbtree.h:56-74), keys = chain indices
 * ordered by chain.pos (bwamem.cpp:40-41).  Node-exact because equal keys are possible (A.4 item 27). */
#define BT_T 5
typedef struct { int is_internal, n; int key[2 * BT_T - 1]; int ptr[2 * BT_T]; } bt_node;
typedef struct { VEC(bt_node) nodes; int root, n_keys; const chain_v *ch; } btree;

static int bt_new(btree *b, int internal) {
    bt_node z; memset(&z, 0, sizeof z); z.is_internal = internal;
    vec_push(bt_node, b->nodes, z);
    return (int)b->nodes.n - 1;
}
static inline int64_t bt_pos(const btree *b, int key) { return b->ch->a[key].pos; }
/* __kb_getp_aux, kbtree.h:124-138 */
static int bt_getp_aux(const btree *b, const bt_node *x, int64_t k, int *r) {
    int tr, *rr = r ? r : &tr, begin = 0, end = x->n;
    if (x->n == 0) return -1;
    while (begin < end) {
        int mid = (begin + end) >> 1;
        if (bt_pos(b, x->key[mid]) < k) begin = mid + 1; else end = mid;
    }
    if (begin == x->n) { *rr = 1; return x->n - 1; }
    int64_t kp = bt_pos(b, x->key[begin]);
    *rr = (kp < k) - (k < kp);
    if (*rr < 0) --begin;
    return begin;
}
/* kb_intervalp (lower only), kbtree.h:158-175 */
static int bt_lower(const btree *b, int64_t k) {
    int lower = -1, x = b->root, r = 0;
    while (x >= 0) {
        const bt_node *nd = &b->nodes.a[x];
        int i = bt_getp_aux(b, nd, k, &r);
        if (i >= 0 && r == 0) return nd->key[i];
        if (i >= 0) lower = nd->key[i];
        if (!nd->is_internal) return lower;
        x = nd->ptr[i + 1];
    }
    return lower;
}
/* __kb_split, kbtree.h:179-196 */
static void bt_split(btree *b, int xi, int i, int yi) {
    int zi = bt_new(b, b->nodes.a[yi].is_internal);
    bt_node *x = &b->nodes.a[xi], *y = &b->nodes.a[yi], *z = &b->nodes.a[zi];
    z->n = BT_T - 1;
    memcpy(z->key, y->key + BT_T, sizeof(int) * (BT_T - 1));
    if (y->is_internal) memcpy(z->ptr, y->ptr + BT_T, sizeof(int) * BT_T);
    y->n = BT_T - 1;
    memmove(x->ptr + i + 2, x->ptr + i + 1, sizeof(int) * (size_t)(x->n - i));
    x->ptr[i + 1] = zi;
    memmove(x->key + i + 1, x->key + i, sizeof(int) * (size_t)(x->n - i));
    x->key[i] = y->key[BT_T - 1];
    ++x->n;
}
/* __kb_putp_aux, kbtree.h:197-215 */
static void bt_putp_aux(btree *b, int xi, int key) {
    int64_t k = bt_pos(b, key);
    bt_node *x = &b->nodes.a[xi];
    if (!x->is_internal) {
        int i = bt_getp_aux(b, x, k, 0);
        if (i != x->n - 1) memmove(x->key + i + 2, x->key + i + 1, (size_t)(x->n - i - 1) * sizeof(int));
        x->key[i + 1] = key;
        ++x->n;
    } else {
        int i = bt_getp_aux(b, x, k, 0) + 1;
        if (b->nodes.a[x->ptr[i]].n == 2 * BT_T - 1) {
            bt_split(b, xi, i, x->ptr[i]);
            x = &b->nodes.a[xi];                           /* nodes array may have moved */
            if (k > bt_pos(b, x->key[i])) ++i;
        }
        bt_putp_aux(b, x->ptr[i], key);
    }
}
/* kb_putp, kbtree.h:216-231 */
static void bt_put(btree *b, int key) {
    ++b->n_keys;
    if (b->nodes.a[b->root].n == 2 * BT_T - 1) {
        int s = bt_new(b, 1), r = b->root;
        b->root = s; b->nodes.a[s].ptr[0] = r;
        bt_split(b, s, 0, r);
    }
    bt_putp_aux(b, b->root, key);
}
static void bt_traverse(const btree *b, int x, int *out, int *n) {   /* __kb_traverse, kbtree.h:343-366 (in-order) */
    const bt_node *nd = &b->nodes.a[x];
    for (int i = 0; i < nd->n; i++) {
        if (nd->is_internal) bt_traverse(b, nd->ptr[i], out, n);
        out[(*n)++] = nd->key[i];
    }
    if (nd->is_internal) bt_traverse(b, nd->ptr[nd->n], out, n);
}

/* test_and_merge, bwamem.cpp:357-399 */
static int test_and_merge(const ora_opt *opt, int64_t l_pac, chain_t *c, const seed_t *p, int seed_rid) {
    const seed_t *last = &c->seeds[c->n - 1];
    int64_t qend = last->qbeg + last->len, rend = last->rbeg + last->len, x, y;
    if (seed_rid != c->rid) return 0;
    if (p->qbeg >= c->seeds[0].qbeg && p->qbeg + p->len <= qend && p->rbeg >= c->seeds[0].rbeg && p->rbeg + p->len <= rend)
        return 1;
    if ((last->rbeg < l_pac || c->seeds[0].rbeg < l_pac) && p->rbeg >= l_pac) return 0;
    x = p->qbeg - last->qbeg;
    y = p->rbeg - last->rbeg;
    if (y >= 0 && x - y <= opt->w && y - x <= opt->w && x - last->len < opt->max_chain_gap && y - last->len < opt->max_chain_gap) {
        if (c->n == c->m) { c->m <<= 1; c->seeds = (seed_t *)realloc(c->seeds, (size_t)c->m * sizeof(seed_t)); }
        c->seeds[c->n++] = *p;
        return 1;
    }
    return 0;
}

/* mem_chain_seeds for ONE read, bwamem.cpp:834-968.  smems = this read's SMEMs in sorted order. */
static void chain_read(const ora_index *ix, const ora_opt *opt, int seqid, int l_seq, const smem_t *sm, int64_t n_sm,
                       const int64_t *sa, chain_v *out) {
    int b = 0, e = 0, l_rep = 0;
    for (int64_t i = 0; i < n_sm; i++) {                  /* l_rep, :849-861 */
        int sb = (int)sm[i].m, se = (int)sm[i].n + 1;
        if (sm[i].iv.s <= opt->max_occ) continue;
        if (sb > e) { l_rep += e - b; b = sb; e = se; }
        else e = e > se ? e : se;
    }
    l_rep += e - b;
    chain_v ch = {0, 0, 0};
    btree bt; memset(&bt, 0, sizeof bt); bt.ch = &ch;
    bt.root = bt_new(&bt, 0);
    int64_t mypos = 0;
    for (int64_t i = 0; i < n_sm; i++) {
        const smem_t *p = &sm[i];
        int32_t slen = (int32_t)(p->n + 1 - p->m);
        int64_t step = p->iv.s > opt->max_occ ? p->iv.s / opt->max_occ : 1;
        int64_t k; int32_t count;
        for (k = count = 0; k < p->iv.s && count < opt->max_occ; k += step, ++count) {
            seed_t s; s.rbeg = sa[mypos++]; s.qbeg = (int32_t)p->m; s.score = s.len = slen; s.aln = 0;
            int rid = intv2rid(ix, s.rbeg, s.rbeg + s.len);
            if (rid < 0) continue;                        /* :915-919 */
            int to_add = 0;
            if (bt.n_keys) {
                bt.ch = &ch;
                int lower = bt_lower(&bt, s.rbeg);
                if (lower < 0 || !test_and_merge(opt, ix->l_pac, &ch.a[lower], &s, rid)) to_add = 1;
            } else to_add = 1;
            if (to_add) {                                  /* :930-951 */
                chain_t c; memset(&c, 0, sizeof c);
                c.n = 1; c.m = 4; c.seeds = (seed_t *)malloc(4 * sizeof(seed_t));
                c.seeds[0] = s; c.rid = rid; c.seqid = seqid; c.is_alt = !!ix->ann_is_alt[rid]; c.pos = s.rbeg;
                vec_push(chain_t, ch, c);
                bt.ch = &ch;
                bt_put(&bt, (int)ch.n - 1);
            }
        }
    }
    int *order = (int *)malloc((size_t)(ch.n + 1) * sizeof(int)), n_ord = 0;
    bt.ch = &ch;
    bt_traverse(&bt, bt.root, order, &n_ord);
    for (int i = 0; i < n_ord; i++) {
        chain_t c = ch.a[order[i]];
        c.frac_rep = (float)l_rep / l_seq;                 /* :965-966 */
        vec_push(chain_t, *out, c);
    }
    free(order); free(ch.a); free(bt.nodes.a);
}

/* mem_chain_weight, bwamem.cpp:429-448 */
static int chain_weight(const chain_t *c) {
    int64_t end; int j, w = 0, tmp;
    for (j = 0, end = 0; j < c->n; ++j) {
        const seed_t *s = &c->seeds[j];
        if (s->qbeg >= end) w += s->len;
        else if (s->qbeg + s->len > end) w += (int)(s->qbeg + s->len - end);
        end = end > s->qbeg + s->len ? end : s->qbeg + s->len;
    }
    tmp = w; w = 0;
    for (j = 0, end = 0; j < c->n; ++j) {
        const seed_t *s = &c->seeds[j];
        if (s->rbeg >= end) w += s->len;
        else if (s->rbeg + s->len > end) w += (int)(s->rbeg + s->len - end);
        end = end > s->rbeg + s->len ? end : s->rbeg + s->len;
    }
    w = w < tmp ? w : tmp;
    return w < 1 << 30 ? w : (1 << 30) - 1;
}
static int flt_lt(const void *a, const void *b) { return ((const chain_t *)a)->w > ((const chain_t *)b)->w; }   /* bwamem.cpp:61 */
#define chn_beg(ch) ((ch).seeds->qbeg)
#define chn_end(ch) ((ch).seeds[(ch).n - 1].qbeg + (ch).seeds[(ch).n - 1].len)

/* mem_chain_flt for the chains of ONE read, bwamem.cpp:506-624 */
static int chain_flt(const ora_opt *opt, int n_chn, chain_t *a) {
    int i, k;
    if (n_chn == 0) return 0;
    for (i = k = 0; i < n_chn; ++i) {
        chain_t *c = &a[i];
        c->first = -1; c->kept = 0;
        c->w = chain_weight(c);
        if (c->w < opt->min_chain_weight) { if (i > 0) free(c->seeds); }   /* a[0] may be resurrected below */
        else { if (k == 0 && i > 0) free(a[0].seeds); a[k++] = *c; }
    }
    n_chn = k;
    /* Quirk (bwamem.cpp:529-546): when every chain fell below min_chain_weight the reference still builds one
     * range [0,1) and so keeps the untouched a_[0] (the original first chain) with kept=3.  Restated as is. */
    if (n_chn == 0) n_chn = 1;
    ora_introsort(a, (size_t)n_chn, sizeof(chain_t), flt_lt);
    int *chains = (int *)malloc((size_t)n_chn * sizeof(int)), n_kept = 0;
    a[0].kept = 3;
    chains[n_kept++] = 0;
    for (i = 1; i < n_chn; ++i) {
        int large_ovlp = 0;
        for (k = 0; k < n_kept; ++k) {
            int j = chains[k];
            int b_max = chn_beg(a[j]) > chn_beg(a[i]) ? chn_beg(a[j]) : chn_beg(a[i]);
            int e_min = chn_end(a[j]) < chn_end(a[i]) ? chn_end(a[j]) : chn_end(a[i]);
            if (e_min > b_max && (!a[j].is_alt || a[i].is_alt)) {
                int li = chn_end(a[i]) - chn_beg(a[i]);
                int lj = chn_end(a[j]) - chn_beg(a[j]);
                int min_l = li < lj ? li : lj;
                if (e_min - b_max >= min_l * opt->mask_level && min_l < opt->max_chain_gap) {
                    large_ovlp = 1;
                    if (a[j].first < 0) a[j].first = i;
                    if (a[i].w < a[j].w * opt->drop_ratio && a[j].w - a[i].w >= opt->min_seed_len << 1) break;
                }
            }
        }
        if (k == n_kept) { chains[n_kept++] = i; a[i].kept = large_ovlp ? 2 : 3; }
    }
    for (i = 0; i < n_kept; ++i) { chain_t *c = &a[chains[i]]; if (c->first >= 0) a[c->first].kept = 1; }
    free(chains);
    for (i = k = 0; i < n_chn; ++i) {
        if (a[i].kept == 0 || a[i].kept == 3) continue;
        if (++k >= opt->max_chain_extend) break;
    }
    for (; i < n_chn; ++i) if (a[i].kept < 3) a[i].kept = 0;
    for (i = k = 0; i < n_chn; ++i) {
        chain_t *c = &a[i];
        if (c->kept == 0) free(c->seeds);
        else a[k++] = a[i];
    }
    return k;
}

/* ------------------------------------------------------------------ short-seed filter (long reads / -W) */

/* score of ksw_align2(..., KSW_XSTART) = ksw_i16 (ksw.cpp:234-338): local Smith-Waterman, affine gaps opened from H
 * (not from M as in ksw_extend2), everything clamped at 0 by the unsigned saturating subtractions.  Only the score is
 * used by the caller; Farrar's striped evaluation order gives the same maximum as this plain row-by-row DP. */
static int local_sw_score(int qlen, const uint8_t *query, int tlen, const uint8_t *target, const int8_t *mat,
                          int o_del, int e_del, int o_ins, int e_ins) {
    int *H = (int *)calloc((size_t)qlen + 1, sizeof(int)), *E = (int *)calloc((size_t)qlen + 1, sizeof(int));
    int gmax = 0;
    for (int i = 0; i < tlen; i++) {
        const int8_t *row = &mat[target[i] * 5];
        int hdiag = 0, f = 0;
        for (int j = 0; j < qlen; j++) {
            int h = hdiag + row[query[j]];
            int e = E[j], t;
            hdiag = H[j];
            h = h > e ? h : e;
            h = h > f ? h : f;
            if (h > gmax) gmax = h;
            H[j] = h;
            e -= e_del; if (e < 0) e = 0;
            t = h - (o_del + e_del); if (t < 0) t = 0;
            E[j] = e > t ? e : t;
            f -= e_ins; if (f < 0) f = 0;
            t = h - (o_ins + e_ins); if (t < 0) t = 0;
            f = f > t ? f : t;
        }
    }
    free(H); free(E);
    return gmax;
}

/* mem_seed_sw, bwamem.cpp:401-427 */
static int seed_sw(const ora_index *ix, const ora_opt *opt, int l_query, const uint8_t *query, const seed_t *s) {
    int qb, qe;
    int64_t rb, re, mid, l_pac = ix->l_pac;
    if (s->len >= 200) return -1;                       /* MEM_SHORT_LEN */
    qb = s->qbeg; qe = s->qbeg + s->len;
    rb = s->rbeg; re = s->rbeg + s->len;
    mid = (rb + re) >> 1;
    qb -= 50; qb = qb > 0 ? qb : 0;                     /* MEM_SHORT_EXT */
    qe += 50; qe = qe < l_query ? qe : l_query;
    rb -= 50; rb = rb > 0 ? rb : 0;
    re += 50; re = re < l_pac << 1 ? re : l_pac << 1;
    if (rb < l_pac && l_pac < re) { if (mid < l_pac) re = l_pac; else rb = l_pac; }
    if (qe - qb >= 200 || re - rb >= 200) return -1;
    {                                                   /* bns_fetch_seq, bntseq.cpp:454-482: clip to the contig of mid */
        int is_rev;
        int rid = pos2rid(ix, depos(ix, mid, &is_rev));
        int64_t far_beg = ix->ann_offset[rid], far_end = far_beg + ix->ann_len[rid];
        if (is_rev) { int64_t tmp = far_beg; far_beg = (l_pac << 1) - far_end; far_end = (l_pac << 1) - tmp; }
        rb = rb > far_beg ? rb : far_beg;
        re = re < far_end ? re : far_end;
    }
    return local_sw_score(qe - qb, query + qb, (int)(re - rb), ix->ref_string + rb, opt->mat, opt->o_del, opt->e_del, opt->o_ins, opt->e_ins);
}

/* mem_flt_chained_seeds for the chains of one read, bwamem.cpp:472-504 */
static void flt_chained_seeds(const ora_index *ix, const ora_opt *opt, int l_query, const uint8_t *query, int n_chn, chain_t *a) {
    double min_l = opt->min_chain_weight ? 1.1f * opt->min_chain_weight : 5.5f * log(l_query);   /* MEM_HSP_COEF, MEM_MINSC_COEF */
    int min_HSP_score = (int)(opt->a * min_l + .499);
    if (min_l > 0.05f * l_query) return;                /* MEM_SEEDSW_COEF: short reads skip this */
    for (int i = 0; i < n_chn; ++i) {
        chain_t *c = &a[i];
        int j, k;
        for (j = k = 0; j < c->n; ++j) {
            seed_t *s = &c->seeds[j];
            s->score = seed_sw(ix, opt, l_query, query, s);
            if (s->score < 0 || s->score >= min_HSP_score) {
                s->score = s->score < 0 ? s->len * opt->a : s->score;
                c->seeds[k++] = *s;
            }
        }
        c->n = k;
    }
}

/* ------------------------------------------------------------------ banded extension */

int ora_pair_class(int len1, int len2, int h0, int a) {        /* bwamem.cpp:1947-1953, 2304-2313 */
    int minval = h0 + (len1 < len2 ? len1 : len2) * a;
    if (len1 < 128 && len2 < 128 && minval < 128) return 8;
    if (len1 < 32768 && len2 < 32768 && minval < 32768) return 16;
    return 32;
}

/* band clamp: scalar bandedSWA.cpp:148-156; vector wrappers :635-653 (int8) and :1333-1353 (int16) compute the
 * same quantity in wrapping unsigned lane arithmetic and an integer division. */
int ora_band_clamp(int w, int qlen, int max_sc, int end_bonus, int o_ins, int e_ins, int o_del, int e_del, int cls) {
    int max_ins, max_del;
    if (cls == 32) {
        max_ins = (int)((double)(qlen * max_sc + end_bonus - o_ins) / e_ins + 1.);
        max_del = (int)((double)(qlen * max_sc + end_bonus - o_del) / e_del + 1.);
    } else {
        unsigned mask = cls == 8 ? 0xffu : 0xffffu;
        unsigned q = (unsigned)(qlen * max_sc) & mask;
        unsigned ti = (q + ((unsigned)(end_bonus - o_ins) & mask)) & mask;
        unsigned td = (q + ((unsigned)(end_bonus - o_del) & mask)) & mask;
        max_ins = (int)((double)(int)(ti / (unsigned)e_ins) + 1.0);
        max_del = (int)((double)(int)(td / (unsigned)e_del) + 1.0);
    }
    max_ins = max_ins > 1 ? max_ins : 1;
    w = w < max_ins ? w : max_ins;
    max_del = max_del > 1 ? max_del : 1;
    w = w < max_del ? w : max_del;
    return w;
}

/* ksw_extend2 (ksw.cpp:432-533) == BandedPairWiseSW::scalarBandedSWA (bandedSWA.cpp:116-237).
 * `w` must already be clamped by ora_band_clamp.  cls = 8 / 16: the pair runs in the reference's int8 / int16 SIMD kernel,
 * whose Z-drop test (ZSCORE8 / ZSCORE16, bandedSWA.cpp:268-281, 309-322) is NOT the scalar one: it is evaluated on every
 * row (also the row that raised the maximum, and also when zdrop <= 0), in wrapping lane arithmetic with zdrop itself
 * truncated to the lane width (-d 200 is -56 in the int8 kernel), and the diagonal offset is not multiplied by the gap
 * extension penalty.  With e_del = e_ins = 1 and 0 < zdrop < 128 the two tests agree. */
int ora_ksw_extend_cls(int qlen, const uint8_t *query, int tlen, const uint8_t *target, const int8_t *mat,
                       int o_del, int e_del, int o_ins, int e_ins, int w, int end_bonus, int zdrop, int h0,
                       int *_qle, int *_tle, int *_gtle, int *_gscore, int *_max_off, int64_t *cells, int cls) {
    typedef struct { int32_t h, e; } eh_t;
    const int m = 5;
    int i, j, oe_del = o_del + e_del, oe_ins = o_ins + e_ins, beg, end, max, max_i, max_j, max_ie, gscore, max_off;
    eh_t *eh = (eh_t *)calloc((size_t)qlen + 1, sizeof(eh_t));
    int64_t nc = 0;
    (void)end_bonus;
    /* ORA_TRACE_ROWS=<file>: the band [beg, end) of every row of every call, for tools/ext_sim.py (how the lanes of a
     * wavefront would be used by a given task-to-lane mapping); int32 header {qlen, tlen, h0, w, rows} then rows x {beg, end} int16 */
    static FILE *trace_f = NULL; static int trace_on = -1;
    if (trace_on < 0) { const char *fn = getenv("ORA_TRACE_ROWS"); trace_on = fn && *fn; if (trace_on) trace_f = fopen(fn, "wb"); if (!trace_f) trace_on = 0; }
    int16_t *trace_rows = trace_on ? (int16_t *)malloc((size_t)(tlen + 1) * 4) : NULL; int trace_n = 0;
    eh[0].h = h0; eh[1].h = h0 > oe_ins ? h0 - oe_ins : 0;
    for (j = 2; j <= qlen && eh[j - 1].h > e_ins; ++j) eh[j].h = eh[j - 1].h - e_ins;
    max = h0; max_i = max_j = -1; max_ie = -1; gscore = -1; max_off = 0;
    beg = 0; end = qlen;
    for (i = 0; i < tlen; ++i) {
        int t, f = 0, h1, mm = 0, mj = -1;
        const int8_t *q = &mat[target[i] * m];
        if (beg < i - w) beg = i - w;
        if (end > i + w + 1) end = i + w + 1;
        if (end > qlen) end = qlen;
        if (beg == 0) { h1 = h0 - (o_del + e_del * (i + 1)); if (h1 < 0) h1 = 0; }
        else h1 = 0;
        if (trace_rows) { trace_rows[2 * trace_n] = (int16_t)beg; trace_rows[2 * trace_n + 1] = (int16_t)end; trace_n++; }
        for (j = beg; j < end; ++j) {
            eh_t *p = &eh[j];
            int h, M = p->h, e = p->e;
            p->h = h1;
            M = M ? M + q[query[j]] : 0;
            h = M > e ? M : e;
            h = h > f ? h : f;
            h1 = h;
            mj = mm > h ? mj : j;
            mm = mm > h ? mm : h;
            t = M - oe_del; t = t > 0 ? t : 0;
            e -= e_del; e = e > t ? e : t;
            p->e = e;
            t = M - oe_ins; t = t > 0 ? t : 0;
            f -= e_ins; f = f > t ? f : t;
            nc++;
        }
        eh[end].h = h1; eh[end].e = 0;
        if (j == qlen) {
            max_ie = gscore > h1 ? max_ie : i;
            gscore = gscore > h1 ? gscore : h1;
        }
        if (mm == 0) break;
        if (mm > max) {
            max = mm; max_i = i; max_j = mj;
            max_off = max_off > abs(mj - i) ? max_off : abs(mj - i);
        } else if (cls == 32 && zdrop > 0) {
            if (i - max_i > mj - max_j) {
                if (max - mm - ((i - max_i) - (mj - max_j)) * e_del > zdrop) break;
            } else {
                if (max - mm - ((mj - max_j) - (i - max_i)) * e_ins > zdrop) break;
            }
        }
        if (cls != 32) {
#define ORA_WR(v) (cls == 8 ? (int)(int8_t)(v) : (int)(int16_t)(v))
            const int ti = ORA_WR(i - max_i), tj = ORA_WR(mj - max_j);
            const int diff = ti > tj ? ORA_WR(ti - tj) : ORA_WR(tj - ti);
            const int t2 = ORA_WR(ORA_WR(max - mm) - diff);
            if (t2 > ORA_WR(zdrop)) break;
#undef ORA_WR
        }
        for (j = beg; j < end && eh[j].h == 0 && eh[j].e == 0; ++j) {}
        beg = j;
        for (j = end; j >= beg && eh[j].h == 0 && eh[j].e == 0; --j) {}
        end = j + 2 < qlen ? j + 2 : qlen;
    }
    free(eh);
    if (trace_rows) {
        int32_t hdr[5] = { qlen, tlen, h0, w, trace_n };
        fwrite(hdr, 4, 5, trace_f); fwrite(trace_rows, 4, (size_t)trace_n, trace_f); fflush(trace_f);
        free(trace_rows);
    }
    if (_qle) *_qle = max_j + 1;
    if (_tle) *_tle = max_i + 1;
    if (_gtle) *_gtle = max_ie + 1;
    if (_gscore) *_gscore = gscore;
    if (_max_off) *_max_off = max_off;
    if (cells) *cells += nc;
    return max;
}

int ora_ksw_extend(int qlen, const uint8_t *query, int tlen, const uint8_t *target, const int8_t *mat,
                   int o_del, int e_del, int o_ins, int e_ins, int w, int end_bonus, int zdrop, int h0,
                   int *_qle, int *_tle, int *_gtle, int *_gscore, int *_max_off, int64_t *cells) {
    return ora_ksw_extend_cls(qlen, query, tlen, target, mat, o_del, e_del, o_ins, e_ins, w, end_bonus, zdrop, h0, _qle, _tle, _gtle,
                              _gscore, _max_off, cells, 32);
}

/* cal_max_gap, bwamem.cpp:66-76 */
static int cal_max_gap(const ora_opt *opt, int qlen) {
    int l_del = (int)((double)(qlen * opt->a - opt->o_del) / opt->e_del + 1.);
    int l_ins = (int)((double)(qlen * opt->a - opt->o_ins) / opt->e_ins + 1.);
    int l = l_del > l_ins ? l_del : l_ins;
    l = l > 1 ? l : 1;
    return l < opt->w << 1 ? l : opt->w << 1;
}

typedef struct {                                 /* the fields of mem_alnreg_t (bwamem.h:137-160) kernel2 writes */
    int64_t rb, re; int32_t qb, qe, rid, score, truesc, w, seedcov, seedlen0; float frac_rep;
    const chain_t *c;
} reg_t;
typedef VEC(reg_t) reg_v;
typedef VEC(ora_pair_rec) pair_v;

static void seedcov_update(reg_t *a) {           /* bwamem.cpp:2507-2516 and its five siblings */
    if (a->rb != ORA_H0 && a->qb != ORA_H0 && a->qe != ORA_H0 && a->re != ORA_H0) {
        a->seedcov = 0;
        for (int i = 0; i < a->c->n; ++i) {
            const seed_t *t = &a->c->seeds[i];
            if (t->qbeg >= a->qb && t->qbeg + t->len <= a->qe && t->rbeg >= a->rb && t->rbeg + t->len <= a->re)
                a->seedcov += t->len;
        }
    }
}
static int u64_lt(const void *a, const void *b) { return *(const uint64_t *)a < *(const uint64_t *)b; }

/* run one extension task with the two-try band logic of bwamem.cpp:2472-2526 (left) / :2688-2740 (right) */
static void extend_task(const ora_opt *opt, const uint8_t *qs, int len2, const uint8_t *rs, int len1, int h0,
                        int end_bonus, int prev_score, ora_pair_rec *pr, int64_t *cells) {
    int cls = ora_pair_class(len1, len2, h0, opt->a);
    int prev = prev_score;
    for (int i = 0; i < ORA_MAX_BAND_TRY; i++) {
        int w = opt->w << i, qle, tle, gtle, gscore, max_off;
        int wc = ora_band_clamp(w, len2, opt->a, end_bonus, opt->o_ins, opt->e_ins, opt->o_del, opt->e_del, cls);
        int sc = ora_ksw_extend_cls(len2, qs, len1, rs, opt->mat, opt->o_del, opt->e_del, opt->o_ins, opt->e_ins, wc,
                                    end_bonus, opt->zdrop, h0, &qle, &tle, &gtle, &gscore, &max_off, cells, cls);
        pr->score = sc; pr->qle = qle; pr->tle = tle; pr->gtle = gtle; pr->gscore = gscore; pr->max_off = max_off;
        pr->w_used = w;
        if (sc == prev || max_off < (w >> 1) + (w >> 2) || i + 1 == ORA_MAX_BAND_TRY) break;
        prev = sc;
    }
}

/* mem_chain2aln_across_reads_V2 for ONE read (bwamem.cpp:2069-2994), eager extension + post-filter */
static void chain2aln_read(const ora_index *ix, const ora_opt *opt, int read_id, const uint8_t *query, int l_query,
                           chain_t *chains, int n_chains, reg_v *av, ora_result *res, pair_v *pairs) {
    int64_t l_pac = ix->l_pac;
    int tot_seeds = 0;
    for (int j = 0; j < n_chains; j++) tot_seeds += chains[j].n;
    uint32_t *srtg = (uint32_t *)malloc((size_t)(tot_seeds + 1) * sizeof(uint32_t));
    uint8_t *qbuf = (uint8_t *)malloc((size_t)l_query + 1);
    int spos = 0;
    av->n = 0;
    for (int j = 0; j < n_chains; j++) {
        chain_t *c = &chains[j];
        if (c->n == 0) continue;
        int64_t rmax0 = l_pac << 1, rmax1 = 0;
        for (int i = 0; i < c->n; ++i) {              /* :2145-2157 */
            const seed_t *t = &c->seeds[i];
            int64_t b = t->rbeg - (t->qbeg + cal_max_gap(opt, t->qbeg));
            int64_t e = t->rbeg + t->len + ((l_query - t->qbeg - t->len) + cal_max_gap(opt, l_query - t->qbeg - t->len));
            rmax0 = rmax0 < b ? rmax0 : b;
            rmax1 = rmax1 > e ? rmax1 : e;
        }
        rmax0 = rmax0 > 0 ? rmax0 : 0;
        rmax1 = rmax1 < l_pac << 1 ? rmax1 : l_pac << 1;
        if (rmax0 < l_pac && l_pac < rmax1) {          /* :2161-2165 */
            if (c->seeds[0].rbeg < l_pac) rmax1 = l_pac; else rmax0 = l_pac;
        }
        {                                              /* bns_fetch_seq_v2, :1890-1923 */
            int is_rev;
            int rid = pos2rid(ix, depos(ix, c->seeds[0].rbeg, &is_rev));
            int64_t far_beg = ix->ann_offset[rid], far_end = far_beg + ix->ann_len[rid];
            if (is_rev) { int64_t tmp = far_beg; far_beg = (l_pac << 1) - far_end; far_end = (l_pac << 1) - tmp; }
            rmax0 = rmax0 > far_beg ? rmax0 : far_beg;
            rmax1 = rmax1 < far_end ? rmax1 : far_end;
        }
        const uint8_t *rseq = ix->ref_string + rmax0;
        uint64_t *srt = (uint64_t *)malloc((size_t)c->n * 8);
        for (int i = 0; i < c->n; ++i) srt[i] = (uint64_t)c->seeds[i].score << 32 | (uint32_t)i;     /* :2188-2192 */
        if (c->n > 1) ora_introsort(srt, (size_t)c->n, 8, u64_lt);
        for (int i = 0; i < c->n; ++i) srtg[spos++] = (uint32_t)srt[i];
        for (int k = c->n - 1; k >= 0; k--) {
            seed_t *s = &c->seeds[(uint32_t)srt[k]];
            reg_t a; memset(&a, 0, sizeof a);
            s->aln = (int32_t)av->n;
            a.w = opt->w; a.score = a.truesc = -1; a.rid = c->rid; a.frac_rep = c->frac_rep; a.seedlen0 = s->len; a.c = c;
            a.rb = a.re = ORA_H0; a.qb = a.qe = ORA_H0;
            int reg_idx = (int)av->n;
            ora_pair_rec pl, prr; int have_l = 0, have_r = 0;
            memset(&pl, 0, sizeof pl); memset(&prr, 0, sizeof prr);
            if (s->qbeg) {                              /* left extension :2229-2317 */
                int64_t tmp = s->rbeg - rmax0;
                pl.read = read_id; pl.reg = reg_idx; pl.is_right = 0; pl.len2 = s->qbeg; pl.len1 = (int32_t)tmp;
                pl.h0 = s->len * opt->a; pl.ref_pos = s->rbeg - 1; pl.q_pos = s->qbeg - 1;
                a.qb = s->qbeg; a.rb = s->rbeg;
                have_l = 1;
            } else {
                a.score = a.truesc = s->len * opt->a; a.qb = 0; a.rb = s->rbeg;
            }
            if (s->qbeg + s->len != l_query) {          /* right extension :2324-2418 */
                int64_t qe = s->qbeg + s->len, re = s->rbeg + s->len - rmax0;
                prr.read = read_id; prr.reg = reg_idx; prr.is_right = 1;
                prr.len2 = (int32_t)(l_query - qe); prr.len1 = (int32_t)(rmax1 - rmax0 - re);
                prr.ref_pos = rmax0 + re; prr.q_pos = (int32_t)qe;
                a.qe = (int32_t)qe; a.re = rmax0 + re;
                have_r = 1;
            } else {
                a.qe = l_query; a.re = s->rbeg + s->len;
                if (a.rb != ORA_H0 && a.qb != ORA_H0) seedcov_update(&a);
            }
            if (have_l) {                               /* :2472-2526 (and the int16/int8 copies) */
                uint8_t *rs = (uint8_t *)malloc((size_t)pl.len1 + 1);
                for (int i = 0; i < pl.len2; ++i) qbuf[i] = query[s->qbeg - 1 - i];
                for (int64_t i = 0; i < pl.len1; ++i) rs[i] = rseq[pl.len1 - 1 - i];
                extend_task(opt, qbuf, pl.len2, rs, pl.len1, pl.h0, opt->pen_clip5, a.score, &pl, &res->n_sw_cells);
                free(rs);
                a.score = pl.score;
                if (pl.gscore <= 0 || pl.gscore <= a.score - opt->pen_clip5) {
                    a.qb -= pl.qle; a.rb -= pl.tle; a.truesc = a.score;
                } else {
                    a.qb = 0; a.rb -= pl.gtle; a.truesc = pl.gscore;
                }
                a.w = a.w > pl.w_used ? a.w : pl.w_used;
                seedcov_update(&a);
                vec_push(ora_pair_rec, *pairs, pl);
            }
            if (have_r) {                               /* :2672-2677 then :2688-2880 */
                prr.h0 = a.score;
                extend_task(opt, query + prr.q_pos, prr.len2, ix->ref_string + prr.ref_pos, prr.len1, prr.h0,
                            opt->pen_clip3, a.score, &prr, &res->n_sw_cells);
                a.score = prr.score;
                if (prr.gscore <= 0 || prr.gscore <= a.score - opt->pen_clip3) {
                    a.qe += prr.qle; a.re += prr.tle; a.truesc += a.score - prr.h0;
                } else {
                    a.qe = l_query; a.re += prr.gtle; a.truesc += prr.gscore - prr.h0;
                }
                a.w = a.w > prr.w_used ? a.w : prr.w_used;
                seedcov_update(&a);
                vec_push(ora_pair_rec, *pairs, prr);
            }
            vec_push(reg_t, *av, a);
        }
        free(srt);
    }
    /* redundant-seed post-filter, :2895-2989 */
    static int check_chunked = -1;                 /* ORA_CHECK_CHUNKED=1: cross-check the 64-regs-at-a-time form of the walk */
    if (check_chunked < 0) check_chunked = getenv("ORA_CHECK_CHUNKED") ? 1 : 0;
    int lim = 0, s_start = 0;
    for (int j = 0; j < n_chains; j++) {
        chain_t *c = &chains[j];
        uint32_t *srt2 = srtg + s_start;
        s_start += c->n;
        for (int k = c->n - 1; k >= 0; k--) {
            seed_t *s = &c->seeds[srt2[k]];
            int i, v = 0;
            for (i = 0; i < (int)av->n && v < lim; ++i) {
                reg_t *p = &av->a[i];
                int64_t rd; int qd, w, max_gap;
                if (p->qb == -1 && p->qe == -1) continue;
                if (s->rbeg < p->rb || s->rbeg + s->len > p->re || s->qbeg < p->qb || s->qbeg + s->len > p->qe) { v++; continue; }
                if (s->len - p->seedlen0 > .1 * l_query) { v++; continue; }
                qd = s->qbeg - p->qb; rd = s->rbeg - p->rb;
                max_gap = cal_max_gap(opt, qd < rd ? qd : (int)rd);
                w = max_gap < p->w ? max_gap : p->w;
                if (qd - rd < w && rd - qd < w) break;
                qd = p->qe - (s->qbeg + s->len); rd = p->re - (s->rbeg + s->len);
                max_gap = cal_max_gap(opt, qd < rd ? qd : (int)rd);
                w = max_gap < p->w ? max_gap : p->w;
                if (qd - rd < w && rd - qd < w) break;
                v++;
            }
            if (check_chunked) {
                /* the form a wavefront would use (notes/postfilter_heavy_wip.patch): judge 64 regs at once, restore the order
                 * of the walk with masks: v = non-purged regs seen; a reg is looked at only while v < lim; the first
                 * looked-at reg that passes one of the two band tests ends the walk */
                int v2 = 0, stopped = 0;
                for (int i0 = 0; i0 < (int)av->n && v2 < lim && !stopped; i0 += 64) {
                    uint64_t lm = 0, bk = 0;
                    for (int lane = 0; lane < 64 && i0 + lane < (int)av->n; lane++) {
                        const reg_t *p = &av->a[i0 + lane];
                        if (p->qb == -1 && p->qe == -1) continue;
                        lm |= 1ULL << lane;
                        if (s->rbeg < p->rb || s->rbeg + s->len > p->re || s->qbeg < p->qb || s->qbeg + s->len > p->qe) continue;
                        if (s->len - p->seedlen0 > .1 * l_query) continue;
                        int qd = s->qbeg - p->qb; int64_t rd = s->rbeg - p->rb;
                        int max_gap = cal_max_gap(opt, qd < rd ? qd : (int)rd), w = max_gap < p->w ? max_gap : p->w;
                        if (qd - rd < w && rd - qd < w) { bk |= 1ULL << lane; continue; }
                        qd = p->qe - (s->qbeg + s->len); rd = p->re - (s->rbeg + s->len);
                        max_gap = cal_max_gap(opt, qd < rd ? qd : (int)rd); w = max_gap < p->w ? max_gap : p->w;
                        if (qd - rd < w && rd - qd < w) bk |= 1ULL << lane;
                    }
                    uint64_t looked_brk = 0;
                    for (int lane = 0; lane < 64; lane++)
                        if ((bk >> lane & 1) && v2 + __builtin_popcountll(lm & (lane ? ~0ULL >> (64 - lane) : 0ULL)) < lim) looked_brk |= 1ULL << lane;
                    if (looked_brk) {
                        const int ib = __builtin_ctzll(looked_brk);
                        v2 += __builtin_popcountll(lm & (ib ? ~0ULL >> (64 - ib) : 0ULL));
                        stopped = 1;
                    } else v2 += __builtin_popcountll(lm);
                }
                if ((stopped || v2 < lim) != (v < lim)) {
                    fprintf(stderr, "ORA_CHECK_CHUNKED: the chunked walk disagrees (v %d lim %d, chunked v %d stopped %d)\n", v, lim, v2, stopped);
                    abort();
                }
            }
            if (v < lim) {
                for (v = k + 1; v < c->n; ++v) {
                    const seed_t *t;
                    if (srt2[v] == UINT_MAX) continue;
                    t = &c->seeds[srt2[v]];
                    if (t->len < s->len * .95) continue;
                    if (s->qbeg <= t->qbeg && s->qbeg + s->len - t->qbeg >= s->len >> 2 && t->qbeg - s->qbeg != t->rbeg - s->rbeg) break;
                    if (t->qbeg <= s->qbeg && t->qbeg + t->len - s->qbeg >= s->len >> 2 && s->qbeg - t->qbeg != s->rbeg - t->rbeg) break;
                }
                if (v == c->n) {
                    reg_t *ar = &av->a[s->aln];
                    ar->qb = ar->qe = -1;
                    srt2[k] = UINT_MAX;
                    continue;
                }
            }
            lim++;
        }
    }
    free(srtg); free(qbuf);
}

/* ------------------------------------------------------------------ driver */

static void put_chain_recs(const chain_t *a, int n, int read_id, ora_chain_rec **cv, int64_t *nc, int64_t *mc,
                           ora_seed_rec **sv, int64_t *ns, int64_t *ms) {
    for (int j = 0; j < n; j++) {
        const chain_t *c = &a[j];
        if (*nc == *mc) { *mc = *mc ? *mc * 2 : 1024; *cv = (ora_chain_rec *)realloc(*cv, (size_t)*mc * sizeof(ora_chain_rec)); }
        ora_chain_rec r = { read_id, c->n, c->rid, c->is_alt, c->pos, c->frac_rep, c->w, c->kept, c->first };
        (*cv)[(*nc)++] = r;
        for (int i = 0; i < c->n; i++) {
            if (*ns == *ms) { *ms = *ms ? *ms * 2 : 4096; *sv = (ora_seed_rec *)realloc(*sv, (size_t)*ms * sizeof(ora_seed_rec)); }
            ora_seed_rec s = { c->seeds[i].rbeg, c->seeds[i].qbeg, c->seeds[i].len, c->seeds[i].score, 0 };
            (*sv)[(*ns)++] = s;
        }
    }
}

static void put_reg(const reg_t *p, int read_id, ora_reg_rec **rv, int64_t *n, int64_t *m) {
    if (*n == *m) { *m = *m ? *m * 2 : 1024; *rv = (ora_reg_rec *)realloc(*rv, (size_t)*m * sizeof(ora_reg_rec)); }
    ora_reg_rec d; memset(&d, 0, sizeof d);
    d.read = read_id; d.rb = p->rb; d.re = p->re; d.qb = p->qb; d.qe = p->qe; d.rid = p->rid; d.score = p->score;
    d.truesc = p->truesc; d.w = p->w; d.seedcov = p->seedcov; d.seedlen0 = p->seedlen0; d.frac_rep = p->frac_rep;
    (*rv)[(*n)++] = d;
}


/* ------------------------------------------------------------------------------------------------------------------
 * Tail of mem_kernel2_core (bwamem.cpp:1154-1169): mem_sort_dedup_patch (:292-353) with mem_patch_reg (:175-225), whose merge
 * test scores a banded global alignment (bwa_gen_cigar2, bwa.cpp:260-347 -> ksw_global2 without backtrack, ksw.cpp:558-668),
 * then the ALT flag.  Operates on ora_reg_rec (every field of mem_alnreg_t the later stages read). */
#define ORA_MINUS_INF (-0x40000000)
static int global_score(int qlen, const uint8_t *query, int tlen, const uint8_t *target, const int8_t *mat, int o_del, int e_del,
                        int o_ins, int e_ins, int w) {          /* ksw.cpp:558-668, n_cigar_ == 0 */
    int32_t *eh_h = (int32_t *)malloc((size_t)(qlen + 1) * 4), *eh_e = (int32_t *)malloc((size_t)(qlen + 1) * 4);
    int oe_del = o_del + e_del, oe_ins = o_ins + e_ins, i, j;
    eh_h[0] = 0; eh_e[0] = ORA_MINUS_INF;                                                        /* :587-591 */
    for (j = 1; j <= qlen && j <= w; ++j) { eh_h[j] = -(o_ins + e_ins * j); eh_e[j] = ORA_MINUS_INF; }
    for (; j <= qlen; ++j) eh_h[j] = eh_e[j] = ORA_MINUS_INF;
    for (i = 0; i < tlen; ++i) {                                                                 /* :618-637 (no backtrack matrix) */
        int32_t f = ORA_MINUS_INF, h1, beg, end, t;
        const int8_t *q = &mat[target[i] * 5];
        beg = i > w ? i - w : 0;
        end = i + w + 1 < qlen ? i + w + 1 : qlen;
        h1 = beg == 0 ? -(o_del + e_del * (i + 1)) : ORA_MINUS_INF;
        for (j = beg; j < end; ++j) {
            int32_t h, m = eh_h[j], e = eh_e[j];
            eh_h[j] = h1;
            m += q[query[j]];
            h = m >= e ? m : e;
            h = h >= f ? h : f;
            h1 = h;
            t = m - oe_del; e -= e_del; e = e > t ? e : t; eh_e[j] = e;
            t = m - oe_ins; f -= e_ins; f = f > t ? f : t;
        }
        eh_h[end] = h1; eh_e[end] = ORA_MINUS_INF;
    }
    i = eh_h[qlen];
    free(eh_h); free(eh_e);
    return i;
}
/* bwa_gen_cigar2 with n_cigar == NM == NULL (bwa.cpp:260-347); 0 = rejected range (*score untouched) */
static int gen_score(const ora_index *ix, const ora_opt *opt, int w_, int l_query, const uint8_t *query, int64_t rb, int64_t re, int *score) {
    int64_t l_pac = ix->l_pac, rlen;
    if (l_query <= 0 || rb >= re || (rb < l_pac && re > l_pac)) return 0;                       /* :274 */
    if (rb < 0 || re > (l_pac << 1)) return 0;                                                   /* bns_get_seq clamps -> :276 */
    rlen = re - rb;
    uint8_t *rseq = (uint8_t *)malloc((size_t)rlen), *q = (uint8_t *)malloc((size_t)l_query);
    memcpy(rseq, ix->ref_string + rb, (size_t)rlen); memcpy(q, query, (size_t)l_query);
    if (rb >= l_pac) {                                                                           /* :277-282 */
        for (int i = 0; i < l_query >> 1; ++i) { uint8_t t = q[i]; q[i] = q[l_query - 1 - i]; q[l_query - 1 - i] = t; }
        for (int64_t i = 0; i < rlen >> 1; ++i) { uint8_t t = rseq[i]; rseq[i] = rseq[rlen - 1 - i]; rseq[rlen - 1 - i] = t; }
    }
    if (l_query == rlen && w_ == 0) {                                                            /* :283-293 */
        int sc = 0;
        for (int i = 0; i < l_query; ++i) sc += opt->mat[rseq[i] * 5 + q[i]];
        *score = sc;
    } else {                                                                                     /* :294-310 */
        int max_ins = (int)((double)(((l_query + 1) >> 1) * opt->mat[0] - opt->o_ins) / opt->e_ins + 1.);
        int max_del = (int)((double)(((l_query + 1) >> 1) * opt->mat[0] - opt->o_del) / opt->e_del + 1.);
        int max_gap = max_ins > max_del ? max_ins : max_del, w, min_w;
        max_gap = max_gap > 1 ? max_gap : 1;
        w = (max_gap + abs((int)rlen - l_query) + 1) >> 1;
        w = w < w_ ? w : w_;
        min_w = abs((int)rlen - l_query) + 3;
        w = w > min_w ? w : min_w;
        *score = global_score(l_query, q, (int)rlen, rseq, opt->mat, opt->o_del, opt->e_del, opt->o_ins, opt->e_ins, w);
    }
    free(rseq); free(q);
    return 1;
}
#define ORA_PATCH_MAX_R_BW 0.05f
#define ORA_PATCH_MIN_SC_RATIO 0.90f
static int patch_reg(const ora_index *ix, const ora_opt *opt, const uint8_t *query, const ora_reg_rec *a, const ora_reg_rec *b, int *_w) {
    int w, score = 0, q_s, r_s;                                                                  /* bwamem.cpp:175-225 */
    double r;
    if (query == 0) return 0;
    if (a->rb < ix->l_pac && b->rb >= ix->l_pac) return 0;
    if (a->qb >= b->qb || a->qe >= b->qe || a->re >= b->re) return 0;
    w = (int)((a->re - b->rb) - (a->qe - b->qb));
    w = w > 0 ? w : -w;
    r = (double)(a->re - b->rb) / (b->re - a->rb) - (double)(a->qe - b->qb) / (b->qe - a->qb);
    r = r > 0. ? r : -r;
    if (a->re < b->rb || a->qe < b->qb) {
        if (w > opt->w << 1 || r >= ORA_PATCH_MAX_R_BW) return 0;
    } else if (w > opt->w << 2 || r >= ORA_PATCH_MAX_R_BW * 2) return 0;
    w += a->w + b->w;
    w = w < opt->w << 2 ? w : opt->w << 2;
    gen_score(ix, opt, w, b->qe - a->qb, query + a->qb, a->rb, b->re, &score);
    q_s = (int)((double)(b->qe - a->qb) / ((b->qe - b->qb) + (a->qe - a->qb)) * (b->score + a->score) + .499);
    r_s = (int)((double)(b->re - a->rb) / ((b->re - b->rb) + (a->re - a->rb)) * (b->score + a->score) + .499);
    if ((double)score / (q_s > r_s ? q_s : r_s) < ORA_PATCH_MIN_SC_RATIO) return 0;
    *_w = w;
    return score;
}
static int reg_slt2(const void *a, const void *b) { return ((const ora_reg_rec *)a)->re < ((const ora_reg_rec *)b)->re; }    /* alnreg_slt2, bwamem.cpp:262 */
static int reg_slt(const void *pa, const void *pb) {                                                                             /* alnreg_slt, :265 */
    const ora_reg_rec *a = (const ora_reg_rec *)pa, *b = (const ora_reg_rec *)pb;
    return a->score > b->score || (a->score == b->score && (a->rb < b->rb || (a->rb == b->rb && a->qb < b->qb)));
}
/* mem_sort_dedup_patch, bwamem.cpp:292-353 */
static int sort_dedup_patch(const ora_index *ix, const ora_opt *opt, const uint8_t *query, int n, ora_reg_rec *a) {
    int m, i, j;
    if (n <= 1) return n;
    ora_introsort(a, (size_t)n, sizeof *a, reg_slt2);
    for (i = 0; i < n; ++i) a[i].n_comp = 1;
    for (i = 1; i < n; ++i) {
        ora_reg_rec *p = &a[i];
        if (p->rid != a[i - 1].rid || p->rb >= a[i - 1].re + opt->max_chain_gap) continue;
        for (j = i - 1; j >= 0 && p->rid == a[j].rid && p->rb < a[j].re + opt->max_chain_gap; --j) {
            ora_reg_rec *q = &a[j];
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
            } else if (q->rb < p->rb && (score = patch_reg(ix, opt, query, q, p, &w)) > 0) {
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
    for (i = 0, m = 0; i < n; ++i) if (a[i].qe > a[i].qb) { if (m != i) a[m++] = a[i]; else ++m; }
    n = m;
    ora_introsort(a, (size_t)n, sizeof *a, reg_slt);
    for (i = 1; i < n; ++i)
        if (a[i].score == a[i - 1].score && a[i].rb == a[i - 1].rb && a[i].qb == a[i - 1].qb) a[i].qe = a[i].qb;
    for (i = 1, m = 1; i < n; ++i) if (a[i].qe > a[i].qb) { if (m != i) a[m++] = a[i]; else ++m; }
    return m;
}
/* regs of the device boundary (any producer: this oracle's REGPRG, refdump's, the HIP path's), grouped by read in read order ->
 * the mem_alnreg_v contents worker_sam receives.  out has room for n_in records; returns the number written. */
int64_t ora_finish_regs(const ora_index *ix, const ora_opt *opt, int32_t n_reads, const uint8_t *enc, const int64_t *off,
                        const ora_reg_rec *in, int64_t n_in, ora_reg_rec *out) {
    int64_t i = 0, o = 0;
    (void)n_reads;
    while (i < n_in) {
        int64_t j = i;
        const int r = in[i].read;
        while (j < n_in && in[j].read == r) j++;
        ora_reg_rec *a = out + o;
        memcpy(a, in + i, (size_t)(j - i) * sizeof *a);
        const int m = sort_dedup_patch(ix, opt, enc + off[r], (int)(j - i), a);
        for (int k = 0; k < m; k++) if (a[k].rid >= 0 && ix->ann_is_alt[a[k].rid]) a[k].is_alt = 1;     /* bwamem.cpp:1161-1169 */
        o += m;
        i = j;
    }
    return o;
}

int ora_run(const ora_index *ix, const ora_opt *opt, int32_t n_reads, const uint8_t *enc, const int64_t *off,
            const int32_t *len, ora_result *res, int stop_after_seeding) {
    memset(res, 0, sizeof *res);
    res->sa_cnt = (int32_t *)calloc((size_t)n_reads + 1, 4);
    int64_t m_smem = 0, m_sa = 0, m_c0 = 0, m_s0 = 0, m_c1 = 0, m_s1 = 0, m_rr = 0, m_rp = 0;
    pair_v pairs = {0, 0, 0};
    smem_v sv = {0, 0, 0};
    fm_stat st = {0, 0, 0, 0};
    for (int base = 0; base < n_reads; base += ORA_BLOCK) {
        int nseq = base + ORA_BLOCK < n_reads ? ORA_BLOCK : n_reads - base;
        /* block-local offsets */
        int64_t *boff = (int64_t *)malloc((size_t)nseq * 8);
        for (int l = 0; l < nseq; l++) boff[l] = off[base + l];
        collect_smem_block(ix, opt, nseq, enc, boff, len + base, &sv, &st);
        for (int64_t i = 0; i < sv.n; i++) {
            if (res->n_smem == m_smem) { m_smem = m_smem ? m_smem * 2 : 4096; res->smem = (ora_smem *)realloc(res->smem, (size_t)m_smem * sizeof(ora_smem)); }
            ora_smem d = { base + (int32_t)sv.a[i].rid, (int32_t)sv.a[i].m, (int32_t)sv.a[i].n, 0, sv.a[i].iv.k, sv.a[i].iv.l, sv.a[i].iv.s };
            res->smem[res->n_smem++] = d;
        }
        /* per read: SA lookup, chaining, filtering, extension.  The loop bound `pos < num_smem - 1` of
         * mem_chain_seeds (bwamem.cpp:834) drops every chain of a block whose total
         * SMEM count is <= 1. */
        int block_has_chains = sv.n > 1;
        int64_t i = 0;
        for (int l = 0; l < nseq; l++) {
            int64_t j = i;
            while (j < sv.n && (int)sv.a[j].rid == l) j++;
            int64_t n_sm = j - i;
            int64_t sa_beg = res->n_sa;
            for (int64_t t = i; t < j; t++) {                         /* FMI_search.cpp:1280-1290 */
                const smem_t *p = &sv.a[t];
                int64_t hi = p->iv.k + p->iv.s, step = p->iv.s > opt->max_occ ? p->iv.s / opt->max_occ : 1, c = 0;
                for (int64_t pos = p->iv.k; pos < hi && c < opt->max_occ; pos += step, c++) {
                    if (res->n_sa == m_sa) { m_sa = m_sa ? m_sa * 2 : 8192; res->sa_coord = (int64_t *)realloc(res->sa_coord, (size_t)m_sa * 8); }
                    res->sa_coord[res->n_sa++] = sa_lookup(ix, pos, &st);
                }
            }
            res->sa_cnt[base + l] = (int32_t)(res->n_sa - sa_beg);
            if (!stop_after_seeding) {
                chain_v ch = {0, 0, 0};
                if (block_has_chains && n_sm > 0 && len[base + l] >= opt->min_seed_len)
                    chain_read(ix, opt, l, len[base + l], sv.a + i, n_sm, res->sa_coord + sa_beg, &ch);
                put_chain_recs(ch.a, (int)ch.n, base + l, &res->chn0, &res->n_chn0, &m_c0, &res->seed0, &res->n_seed0, &m_s0);
                ch.n = chain_flt(opt, (int)ch.n, ch.a);
                flt_chained_seeds(ix, opt, len[base + l], enc + off[base + l], (int)ch.n, ch.a);
                put_chain_recs(ch.a, (int)ch.n, base + l, &res->chn1, &res->n_chn1, &m_c1, &res->seed1, &res->n_seed1, &m_s1);
                reg_v av = {0, 0, 0};
                chain2aln_read(ix, opt, base + l, enc + off[base + l], len[base + l], ch.a, (int)ch.n, &av, res, &pairs);
                for (int64_t r = 0; r < av.n; r++) put_reg(&av.a[r], base + l, &res->regraw, &res->n_regraw, &m_rr);
                for (int64_t r = 0; r < av.n; r++)               /* bwamem.cpp:1141-1152 */
                    if (av.a[r].qe > av.a[r].qb) put_reg(&av.a[r], base + l, &res->regprg, &res->n_regprg, &m_rp);
                for (int64_t c = 0; c < ch.n; c++) free(ch.a[c].seeds);
                free(ch.a); free(av.a);
            }
            i = j;
        }
        free(boff);
    }
    if (!stop_after_seeding && res->n_regprg) {
        res->regfin = (ora_reg_rec *)malloc((size_t)res->n_regprg * sizeof(ora_reg_rec));
        res->n_regfin = ora_finish_regs(ix, opt, n_reads, enc, off, res->regprg, res->n_regprg, res->regfin);
    }
    res->pair = pairs.a; res->n_pair = pairs.n;
    res->n_ext = st.n_ext; res->n_ext_sameblk = st.n_same; res->n_lf = st.n_lf; res->n_sa_lookup = st.n_sa;
    free(sv.a);
    return 0;
}

void ora_result_free(ora_result *r) {
    free(r->smem); free(r->sa_coord); free(r->sa_cnt); free(r->chn0); free(r->seed0); free(r->chn1); free(r->seed1);
    free(r->regraw); free(r->regprg); free(r->regfin); free(r->pair);
    memset(r, 0, sizeof *r);
}
