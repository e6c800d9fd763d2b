// oracle/refdump.cpp -- TEST INFRASTRUCTURE ONLY.
//
// A small driver of OUR OWN that links against the reference's objects (compiled where
// they lie under /root/reference by oracle/Makefile) and calls the reference's hot-path
// functions through their own signatures, dumping every intermediate so that the C
// restatement (oracle/bm2_oracle.c) and the HIP path can be pinned stage by stage:
//
//   mem_collect_smem            (bwamem.cpp:626)   -> SMEM    (sorted SMEMs per block)
//   FMI_search::get_sa_entries_prefetch (FMI_search.cpp:1257) -> SACOORD
//   mem_chain_seeds             (bwamem.cpp:806)   -> CHN0/SEED0
//   mem_chain_flt + mem_flt_chained_seeds (:506,:472) -> CHN1/SEED1
//   mem_chain2aln_across_reads_V2 (bwamem.cpp:2069) -> REGRAW (incl. purged regs)
//   tail of mem_kernel2_core    (bwamem.cpp:1141-1169) -> REGFIN
//
// The glue below follows the call order of mem_kernel1_core (bwamem.cpp:976-1091) and
// mem_kernel2_core (:1093-1173); to make sure the glue itself is faithful, the driver
// ALSO runs the real mem_kernel1_core + mem_kernel2_core on a second copy of the reads
// and aborts if the final regs differ.
//
// Blocks are BATCH_SIZE(512) reads, exactly as kt_for hands them out (kthread.cpp:53-78).
//
// Output: a sequence of sections  [8-byte tag][int64 nbytes][payload]  (little endian).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <ctype.h>
#include <string>
#include <vector>
#include <thread>
#include <time.h>
#include <unistd.h>

#include "bwamem.h"
#include "bwa.h"
#include "bntseq.h"
#include "utils.h"
#include "ksw.h"
#include "FMI_search.h"
#include "fastmap.h"
#include "bandedSWA.h"

uint64_t proc_freq, tprof[LIM_R][LIM_C], prof[LIM_R];

// non-static functions of bwamem.cpp that have no prototype in the headers
SMEM *mem_collect_smem(FMI_search *fmi, const mem_opt_t *opt, const bseq1_t *seq_, int nseq,
                       SMEM *matchArray, int32_t *min_intv_ar, int16_t *query_pos_ar,
                       uint8_t *enc_qdb, int32_t *rid, mem_cache *mmc, int64_t &tot_smem, int tid);
void mem_chain_seeds(FMI_search *fmi, const mem_opt_t *opt, const bntseq_t *bns, const bseq1_t *seq_,
                     int nseq, int tid, mem_chain_v *chain_ar, mem_seed_t *seedBuf,
                     int64_t seedBufSize, SMEM *matchArray, int64_t num_smem);
int mem_chain_flt(const mem_opt_t *opt, int n_chn_, mem_chain_t *a_, int tid);
void mem_flt_chained_seeds(const mem_opt_t *opt, const bntseq_t *bns, const uint8_t *pac,
                           bseq1_t *seq_, int n_chn, mem_chain_t *a);
int mem_sort_dedup_patch(const mem_opt_t *opt, const bntseq_t *bns, const uint8_t *pac,
                         uint8_t *query, int n, mem_alnreg_t *a);
int mem_kernel1_core(FMI_search *fmi, const mem_opt_t *opt, bseq1_t *seq_, int nseq,
                     mem_chain_v *chain_ar, mem_seed_t *seedBuf, int64_t seedBufSize,
                     mem_cache *mmc, int tid);
int mem_kernel2_core(FMI_search *fmi, const mem_opt_t *opt, bseq1_t *seq_, mem_alnreg_v *regs,
                     int nseq, mem_chain_v *chain_ar, mem_cache *mmc, uint8_t *ref_string, int tid);

struct Section { std::string tag; std::vector<uint8_t> data; };
static std::vector<Section*> g_sections;
static Section *sec(const char *tag) {
    for (auto s : g_sections) if (s->tag == tag) return s;
    Section *s = new Section; s->tag = tag; g_sections.push_back(s); return s;
}
template <class T> static void put(Section *s, const T &v) {
    const uint8_t *p = (const uint8_t *)&v; s->data.insert(s->data.end(), p, p + sizeof(T));
}

#pragma pack(push, 1)
struct DSmem  { int32_t read, m, n, pad; int64_t k, l, s; };
struct DChain { int32_t read, n, rid, is_alt; int64_t pos; float frac_rep; int32_t w, kept, first; };
struct DSeed  { int64_t rbeg; int32_t qbeg, len, score, pad; };
struct DReg   { int32_t read, pad; int64_t rb, re; int32_t qb, qe, rid, score, truesc, sub, alt_sc, csub,
                sub_n, w, seedcov, secondary, secondary_all, seedlen0, n_comp, is_alt; float frac_rep; int32_t pad2; };
#pragma pack(pop)

static void alloc_cache(mem_cache &mmc) {   // sizes as in memoryAlloc (fastmap.cpp:100-187), one thread
    int64_t wsize = (int64_t)BATCH_SIZE * SEEDS_PER_READ;
    mmc.seqBufLeftRef[0]  = (uint8_t *)_mm_malloc(wsize * MAX_SEQ_LEN_REF + MAX_LINE_LEN, 64);
    mmc.seqBufLeftQer[0]  = (uint8_t *)_mm_malloc(wsize * MAX_SEQ_LEN_QER + MAX_LINE_LEN, 64);
    mmc.seqBufRightRef[0] = (uint8_t *)_mm_malloc(wsize * MAX_SEQ_LEN_REF + MAX_LINE_LEN, 64);
    mmc.seqBufRightQer[0] = (uint8_t *)_mm_malloc(wsize * MAX_SEQ_LEN_QER + MAX_LINE_LEN, 64);
    mmc.wsize_buf_ref[0] = wsize * MAX_SEQ_LEN_REF;
    mmc.wsize_buf_qer[0] = wsize * MAX_SEQ_LEN_QER;
    mmc.seqPairArrayAux[0]      = (SeqPair *)malloc((wsize + MAX_LINE_LEN) * sizeof(SeqPair));
    mmc.seqPairArrayLeft128[0]  = (SeqPair *)malloc((wsize + MAX_LINE_LEN) * sizeof(SeqPair));
    mmc.seqPairArrayRight128[0] = (SeqPair *)malloc((wsize + MAX_LINE_LEN) * sizeof(SeqPair));
    mmc.wsize[0] = wsize;
    int64_t wm = (int64_t)BATCH_MUL * BATCH_SIZE * READ_LEN;
    mmc.wsize_mem[0] = mmc.wsize_mem_s[0] = mmc.wsize_mem_r[0] = wm;
    mmc.matchArray[0]   = (SMEM *)_mm_malloc(wm * sizeof(SMEM), 64);
    mmc.min_intv_ar[0]  = (int32_t *)malloc(wm * sizeof(int32_t));
    mmc.query_pos_ar[0] = (int16_t *)malloc(wm * sizeof(int16_t));
    mmc.enc_qdb[0]      = (uint8_t *)malloc(wm * sizeof(uint8_t));
    mmc.rid[0]          = (int32_t *)malloc(wm * sizeof(int32_t));
    mmc.lim[0]          = (int32_t *)_mm_malloc((BATCH_SIZE + 32) * sizeof(int32_t), 64);
}

static void grow_for(mem_cache &mmc, int64_t tot_len) {  // as mem_kernel1_core :1004-1023
    if (tot_len >= mmc.wsize_mem[0]) {
        int64_t tmp = mmc.wsize_mem[0];
        mmc.wsize_mem[0] = mmc.wsize_mem_s[0] = mmc.wsize_mem_r[0] = tot_len;
        mmc.matchArray[0]   = (SMEM *)_mm_realloc(mmc.matchArray[0], tmp, tot_len, sizeof(SMEM));
        mmc.min_intv_ar[0]  = (int32_t *)realloc(mmc.min_intv_ar[0], tot_len * sizeof(int32_t));
        mmc.query_pos_ar[0] = (int16_t *)realloc(mmc.query_pos_ar[0], tot_len * sizeof(int16_t));
        mmc.enc_qdb[0]      = (uint8_t *)realloc(mmc.enc_qdb[0], tot_len * sizeof(uint8_t));
        mmc.rid[0]          = (int32_t *)realloc(mmc.rid[0], tot_len * sizeof(int32_t));
    }
}

static void dump_chains(const char *ctag, const char *stag, int base, int nseq, mem_chain_v *chain_ar) {
    Section *sc = sec(ctag), *ss = sec(stag);
    for (int l = 0; l < nseq; l++)
        for (size_t j = 0; j < chain_ar[l].n; j++) {
            mem_chain_t *c = &chain_ar[l].a[j];
            DChain d = { base + l, c->n, c->rid, (int)c->is_alt, c->pos, c->frac_rep, (int)c->w, (int)c->kept, c->first };
            put(sc, d);
            for (int i = 0; i < c->n; i++) {
                DSeed s = { c->seeds[i].rbeg, c->seeds[i].qbeg, c->seeds[i].len, c->seeds[i].score, 0 };
                put(ss, s);
            }
        }
}

static void dump_regs(const char *tag, int base, int nseq, mem_alnreg_v *regs) {
    Section *s = sec(tag);
    for (int l = 0; l < nseq; l++)
        for (size_t i = 0; i < regs[l].n; i++) {
            mem_alnreg_t *p = &regs[l].a[i];
            DReg d; memset(&d, 0, sizeof(d));
            d.read = base + l; d.rb = p->rb; d.re = p->re; d.qb = p->qb; d.qe = p->qe; d.rid = p->rid;
            d.score = p->score; d.truesc = p->truesc; d.sub = p->sub; d.alt_sc = p->alt_sc; d.csub = p->csub;
            d.sub_n = p->sub_n; d.w = p->w; d.seedcov = p->seedcov; d.secondary = p->secondary;
            d.secondary_all = p->secondary_all; d.seedlen0 = p->seedlen0; d.n_comp = p->n_comp;
            d.is_alt = p->is_alt; d.frac_rep = p->frac_rep;
            put(s, d);
        }
}

static bseq1_t *copy_reads(const std::vector<std::string> &rs) {
    bseq1_t *s = (bseq1_t *)calloc(rs.size(), sizeof(bseq1_t));
    for (size_t i = 0; i < rs.size(); i++) {
        s[i].l_seq = (int)rs[i].size();
        s[i].seq = (char *)malloc(rs[i].size() + 1);
        memcpy(s[i].seq, rs[i].c_str(), rs[i].size() + 1);
        s[i].id = (int)i;
    }
    return s;
}

int main(int argc, char **argv) {
    mem_opt_t *opt = mem_opt_init();
    const char *mode = 0;
    int c, set_a = 0, set_b = 0, set_od = 0, set_ed = 0, set_oi = 0, set_ei = 0, set_z = 0, set_l5 = 0, set_l3 = 0;
    int set_k = 0, set_W = 0, set_r = 0;
    while ((c = getopt(argc, argv, "k:w:A:B:O:E:L:d:r:y:c:D:W:m:x:G:")) >= 0) {
        if (c == 'k') opt->min_seed_len = atoi(optarg), set_k = 1;
        else if (c == 'w') opt->w = atoi(optarg);
        else if (c == 'A') opt->a = atoi(optarg), set_a = 1;
        else if (c == 'B') opt->b = atoi(optarg), set_b = 1;
        else if (c == 'O' || c == 'E' || c == 'L') {           // INT[,INT] as fastmap.cpp:681-700 parses them
            char *p;
            const int v1 = (int)strtol(optarg, &p, 10);
            const int v2 = (*p != 0 && ispunct(*p) && isdigit(p[1])) ? (int)strtol(p + 1, &p, 10) : v1;
            if (c == 'O') opt->o_del = v1, opt->o_ins = v2, set_od = set_oi = 1;
            else if (c == 'E') opt->e_del = v1, opt->e_ins = v2, set_ed = set_ei = 1;
            else opt->pen_clip5 = v1, opt->pen_clip3 = v2, set_l5 = set_l3 = 1;
        }
        else if (c == 'd') opt->zdrop = atoi(optarg), set_z = 1;
        else if (c == 'r') opt->split_factor = atof(optarg), set_r = 1;
        else if (c == 'y') opt->max_mem_intv = atol(optarg);
        else if (c == 'c') opt->max_occ = atoi(optarg);
        else if (c == 'D') opt->drop_ratio = atof(optarg);
        else if (c == 'W') opt->min_chain_weight = atoi(optarg), set_W = 1;
        else if (c == 'm') opt->max_matesw = atoi(optarg);
        else if (c == 'G') opt->max_chain_gap = atoi(optarg);
        else if (c == 'x') mode = optarg;
    }
    if (mode) {   // presets as at fastmap.cpp:801-843
        if (!strcmp(mode, "intractg")) {
            if (!set_od) opt->o_del = 16; if (!set_oi) opt->o_ins = 16; if (!set_b) opt->b = 9;
            if (!set_l5) opt->pen_clip5 = 5; if (!set_l3) opt->pen_clip3 = 5;
        } else {
            if (!set_od) opt->o_del = 1; if (!set_ed) opt->e_del = 1; if (!set_oi) opt->o_ins = 1;
            if (!set_ei) opt->e_ins = 1; if (!set_b) opt->b = 1; if (!set_r) opt->split_factor = 10.;
            int ont = !strcmp(mode, "ont2d");
            if (!set_W) opt->min_chain_weight = ont ? 20 : 40;
            if (!set_k) opt->min_seed_len = ont ? 14 : 17;
            if (!set_l5) opt->pen_clip5 = 0; if (!set_l3) opt->pen_clip3 = 0;
        }
    } else if (set_a) {   // update_a, fastmap.cpp:547-561
        if (!set_b) opt->b *= opt->a;
        opt->T *= opt->a;
        if (!set_od) opt->o_del *= opt->a; if (!set_ed) opt->e_del *= opt->a;
        if (!set_oi) opt->o_ins *= opt->a; if (!set_ei) opt->e_ins *= opt->a;
        if (!set_z) opt->zdrop *= opt->a;
        if (!set_l5) opt->pen_clip5 *= opt->a; if (!set_l3) opt->pen_clip3 *= opt->a;
        opt->pen_unpaired *= opt->a;
    }
    bwa_fill_scmat(opt->a, opt->b, opt->mat);
    if (argc - optind == 3 && !strcmp(argv[optind], "ksw")) {
        // known answers for the mate-rescue SW: every line of <pairs.txt> is "<xtra> <query> <target>" (ACGTN text);
        // out = 7 int32 per line (score, te, qe, score2, te2, tb, qb) from the reference's ksw_align2 (ksw.cpp:340-381)
        FILE *fi = fopen(argv[optind + 1], "r"), *fo = fopen(argv[optind + 2], "wb");
        if (!fi || !fo) { fprintf(stderr, "cannot open the --ksw files\n"); return 1; }
        char *line = 0; size_t cap = 0; ssize_t len;
        while ((len = getline(&line, &cap, fi)) > 0) {
            int xtra = 0, pos = 0;
            if (sscanf(line, "%d %n", &xtra, &pos) < 1) continue;
            std::vector<uint8_t> q, t; std::vector<uint8_t> *cur = &q;
            for (char *c = line + pos; *c && *c != '\n'; ++c) {
                if (*c == ' ') { cur = &t; continue; }
                cur->push_back(*c == 'A' ? 0 : *c == 'C' ? 1 : *c == 'G' ? 2 : *c == 'T' ? 3 : 4);
            }
            kswr_t r = ksw_align2((int)q.size(), q.data(), (int)t.size(), t.data(), 5, opt->mat, opt->o_del, opt->e_del, opt->o_ins, opt->e_ins, xtra, 0);
            int32_t o[7] = { r.score, r.te, r.qe, r.score2, r.te2, r.tb, r.qb };
            fwrite(o, 4, 7, fo);
        }
        fclose(fi); fclose(fo);
        return 0;
    }
    if (argc - optind == 5 && !strcmp(argv[optind], "bsw")) {
        // known answers for the banded extension (seam S1): every line of <pairs.txt> is "<h0> <query> <target>" (ACGTN text); the pairs are
        // filed under the reference's three kernels exactly as sortPairsLenExt files them (bwamem.cpp:1924-1950) and run through
        // BandedPairWiseSW::getScores8 / getScores16 / scalarBandedSWAWrapper of THIS build's ISA with band <w> and end bonus <end_bonus>.
        // out = 8 int32 per line: score, qle, tle, gtle, gscore, max_off, class (8 / 16 / 32), and 1 if the pair's result in INPUT order differs
        // from its result in the order the reference runs a class in -- sorted by target length, sortPairsLen (bwamem.cpp:2025-2064, called
        // before every getScores8 / getScores16): the vector kernels work on SIMD-wide groups whose row / column counts are the group's
        // maxima, and a pair grouped with much longer ones is not always left alone.  The sorted run is the answer.
        const int w = atoi(argv[optind + 1]), end_bonus = atoi(argv[optind + 2]);
        FILE *fi = fopen(argv[optind + 3], "r"), *fo = fopen(argv[optind + 4], "wb");
        if (!fi || !fo) { fprintf(stderr, "cannot open the bsw files\n"); return 1; }
        std::vector<SeqPair> all; std::vector<uint8_t> ref, qer;
        char *line = 0; size_t cap = 0; ssize_t len;
        while ((len = getline(&line, &cap, fi)) > 0) {
            int h0 = 0, pos = 0;
            if (sscanf(line, "%d %n", &h0, &pos) < 1) continue;
            SeqPair sp; memset(&sp, 0, sizeof sp);
            sp.idq = (int32_t)qer.size(); sp.idr = (int32_t)ref.size(); sp.h0 = h0; sp.id = (int32_t)all.size();
            std::vector<uint8_t> *cur = &qer;
            for (char *c = line + pos; *c && *c != '\n'; ++c) {
                if (*c == ' ') { cur = &ref; continue; }
                cur->push_back(*c == 'A' ? 0 : *c == 'C' ? 1 : *c == 'G' ? 2 : *c == 'T' ? 3 : 4);
            }
            sp.len2 = (int32_t)qer.size() - sp.idq; sp.len1 = (int32_t)ref.size() - sp.idr;
            all.push_back(sp);
        }
        ref.resize(ref.size() + 65536, 0); qer.resize(qer.size() + 65536, 0);        // (the vector kernels gather whole SIMD groups)
        std::vector<int32_t> cls(all.size());
        std::vector<SeqPair> by[3];
        for (size_t i = 0; i < all.size(); i++) {
            const SeqPair &sp = all[i];
            const int minval = sp.h0 + (sp.len1 < sp.len2 ? sp.len1 : sp.len2) * opt->a;
            const int k = (sp.len1 < MAX_SEQ_LEN8 && sp.len2 < MAX_SEQ_LEN8 && minval < MAX_SEQ_LEN8) ? 0 : (sp.len1 < MAX_SEQ_LEN16 && sp.len2 < MAX_SEQ_LEN16 && minval < MAX_SEQ_LEN16) ? 1 : 2;
            cls[i] = k == 0 ? 8 : k == 1 ? 16 : 32;
            by[k].push_back(sp);
        }
        BandedPairWiseSW bsw(opt->o_del, opt->e_del, opt->o_ins, opt->e_ins, opt->zdrop, end_bonus, opt->mat, opt->a, opt->b, 1);
        auto run = [&](int k, std::vector<SeqPair> &v) {
            if (v.empty()) return;
            const int n = (int)v.size();
            v.resize(v.size() + 2 * SIMD_WIDTH8);               // (the vector wrappers pad the array to a whole SIMD group in place, bandedSWA.cpp:447-455)
            if (k == 0) bsw.getScores8(v.data(), ref.data(), qer.data(), n, 1, w);
            else if (k == 1) bsw.getScores16(v.data(), ref.data(), qer.data(), n, 1, w);
            else bsw.scalarBandedSWAWrapper(v.data(), ref.data(), qer.data(), n, 1, w);
            v.resize((size_t)n);
        };
        std::vector<int32_t> out(all.size() * 8, 0);
        long order_dependent = 0;
        for (int k = 0; k < 3; k++) {
            std::vector<SeqPair> a = by[k], b = by[k];
            std::stable_sort(b.begin(), b.end(), [](const SeqPair &x, const SeqPair &y) { return x.len1 < y.len1; });
            run(k, a); run(k, b);
            std::vector<SeqPair> unsorted(all.size());
            for (const SeqPair &sp : a) unsorted[(size_t)sp.id] = sp;
            for (const SeqPair &sp : b) {
                const SeqPair &o = unsorted[(size_t)sp.id];
                const bool dep = sp.score != o.score || sp.qle != o.qle || sp.tle != o.tle || sp.gtle != o.gtle || sp.gscore != o.gscore || sp.max_off != o.max_off;
                order_dependent += dep;
                int32_t *r = &out[(size_t)sp.id * 8];
                r[0] = sp.score; r[1] = sp.qle; r[2] = sp.tle; r[3] = sp.gtle; r[4] = sp.gscore; r[5] = sp.max_off; r[6] = cls[(size_t)sp.id]; r[7] = dep;
            }
        }
        if (order_dependent) fprintf(stderr, "bsw: %ld of %zu pairs give other numbers in input order than in the reference's (target-length) order\n", order_dependent, all.size());
        fwrite(out.data(), 4, out.size(), fo);
        fclose(fi); fclose(fo);
        return 0;
    }
    if (argc - optind == 6 && !strcmp(argv[optind], "bswtime")) {
        // The reference's banded-extension kernels TIMED on <threads> host threads over a batch in the binary layout bench.py writes
        // (int32 n, then len2[n], len1[n], h0[n], then the query bases and the target bases back to back): config 2's CPU baseline and,
        // since the results are written out (8 int32 per pair as in `bsw`), its parity gate.  As the reference does it: the pairs are filed
        // under the three kernels (sortPairsLenExt), every class runs sorted by target length (sortPairsLen), every thread has its own
        // BandedPairWiseSW object and a contiguous slice (a multiple of 64 pairs) -- kt_for hands each worker its own SeqPair arrays.
        const int w = atoi(argv[optind + 1]), end_bonus = atoi(argv[optind + 2]);
        int T = atoi(argv[optind + 3]); if (T < 1) T = 1;
        FILE *fi = fopen(argv[optind + 4], "rb"), *fo = fopen(argv[optind + 5], "wb");
        if (!fi || !fo) { fprintf(stderr, "cannot open the bswtime files\n"); return 1; }
        int32_t n = 0;
        if (fread(&n, 4, 1, fi) != 1 || n < 0) { fprintf(stderr, "bswtime: bad header\n"); return 1; }
        std::vector<int32_t> l2((size_t)n), l1((size_t)n), h0((size_t)n);
        if (fread(l2.data(), 4, (size_t)n, fi) != (size_t)n || fread(l1.data(), 4, (size_t)n, fi) != (size_t)n || fread(h0.data(), 4, (size_t)n, fi) != (size_t)n) return 1;
        int64_t nq = 0, nr = 0;
        for (int32_t i = 0; i < n; i++) { nq += l2[(size_t)i]; nr += l1[(size_t)i]; }
        if (nq + 65536 >= (1LL << 31) || nr + 65536 >= (1LL << 31)) { fprintf(stderr, "bswtime: batch too large for SeqPair's int32 offsets\n"); return 1; }
        std::vector<uint8_t> qer((size_t)nq + 65536, 0), ref((size_t)nr + 65536, 0);
        if (fread(qer.data(), 1, (size_t)nq, fi) != (size_t)nq || fread(ref.data(), 1, (size_t)nr, fi) != (size_t)nr) return 1;
        fclose(fi);
        std::vector<SeqPair> by[3];
        std::vector<int32_t> cls((size_t)n);
        int64_t oq = 0, orf = 0;
        for (int32_t i = 0; i < n; i++) {
            SeqPair sp; memset(&sp, 0, sizeof sp);
            sp.idq = (int32_t)oq; sp.idr = (int32_t)orf; sp.len2 = l2[(size_t)i]; sp.len1 = l1[(size_t)i]; sp.h0 = h0[(size_t)i]; sp.id = i;
            oq += sp.len2; orf += sp.len1;
            const int minval = sp.h0 + (sp.len1 < sp.len2 ? sp.len1 : sp.len2) * opt->a;
            const int k = (sp.len1 < MAX_SEQ_LEN8 && sp.len2 < MAX_SEQ_LEN8 && minval < MAX_SEQ_LEN8) ? 0 : (sp.len1 < MAX_SEQ_LEN16 && sp.len2 < MAX_SEQ_LEN16 && minval < MAX_SEQ_LEN16) ? 1 : 2;
            cls[(size_t)i] = k == 0 ? 8 : k == 1 ? 16 : 32;
            by[k].push_back(sp);
        }
        for (int k = 0; k < 3; k++) std::stable_sort(by[k].begin(), by[k].end(), [](const SeqPair &x, const SeqPair &y) { return x.len1 < y.len1; });
        std::vector<int32_t> out((size_t)n * 8, 0);
        std::vector<double> busy((size_t)T, 0.0);
        auto now = []() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + ts.tv_nsec * 1e-9; };
        const double t_begin = now();
        std::vector<std::thread> th;
        for (int t = 0; t < T; t++) th.emplace_back([&, t]() {
            BandedPairWiseSW bsw(opt->o_del, opt->e_del, opt->o_ins, opt->e_ins, opt->zdrop, end_bonus, opt->mat, opt->a, opt->b, 1);
            const double t0 = now();
            for (int k = 0; k < 3; k++) {
                const size_t tot = by[k].size(), per = ((tot + (size_t)T - 1) / (size_t)T + 63) / 64 * 64;
                const size_t lo = std::min(tot, per * (size_t)t), hi = std::min(tot, lo + per);
                if (hi <= lo) continue;
                std::vector<SeqPair> v(by[k].begin() + (long)lo, by[k].begin() + (long)hi);
                const int m = (int)v.size();
                v.resize(v.size() + 2 * SIMD_WIDTH8);
                if (k == 0) bsw.getScores8(v.data(), ref.data(), qer.data(), m, 1, w);
                else if (k == 1) bsw.getScores16(v.data(), ref.data(), qer.data(), m, 1, w);
                else bsw.scalarBandedSWAWrapper(v.data(), ref.data(), qer.data(), m, 1, w);
                for (int i = 0; i < m; i++) {
                    const SeqPair &sp = v[(size_t)i];
                    int32_t *r = &out[(size_t)sp.id * 8];
                    r[0] = sp.score; r[1] = sp.qle; r[2] = sp.tle; r[3] = sp.gtle; r[4] = sp.gscore; r[5] = sp.max_off; r[6] = cls[(size_t)sp.id]; r[7] = 0;
                }
            }
            busy[(size_t)t] = now() - t0;
        });
        for (auto &x : th) x.join();
        const double wall = now() - t_begin;
        double mx = 0; for (double b : busy) mx = b > mx ? b : mx;
        fwrite(out.data(), 4, out.size(), fo);
        fclose(fo);
        printf("{\"pairs\": %d, \"threads\": %d, \"seconds\": %.6f, \"slowest_thread_s\": %.6f, \"class8\": %zu, \"class16\": %zu, \"class32\": %zu}\n",
               n, T, wall, mx, by[0].size(), by[1].size(), by[2].size());
        return 0;
    }
    if (argc - optind == 4 && !strcmp(argv[optind], "cigar")) {
        // known answers for CIGAR generation: every line of <tasks.txt> is "<w> <rb> <re> <query>" (query = ACGTN text of the
        // aligned part of the read); out per line: int32 score, n_cigar, NM, then n_cigar uint32 ops, then the MD string + NUL
        // padded to 4 bytes -- from the reference's bwa_gen_cigar2 (bwa.cpp:260-347); a NULL return is written as n_cigar = -1
        bntseq_t *bns = bns_restore(argv[optind + 1]);
        uint8_t *pac = (uint8_t *)calloc(bns->l_pac / 4 + 1, 1);
        err_fread_noeof(pac, 1, bns->l_pac / 4 + 1, bns->fp_pac);
        FILE *fi = fopen(argv[optind + 2], "r"), *fo = fopen(argv[optind + 3], "wb");
        if (!fi || !fo) { fprintf(stderr, "cannot open the cigar files\n"); return 1; }
        char *line = 0; size_t cap = 0; ssize_t len;
        while ((len = getline(&line, &cap, fi)) > 0) {
            int w = 0, pos = 0; long long rb = 0, re = 0;
            if (sscanf(line, "%d %lld %lld %n", &w, &rb, &re, &pos) < 3) continue;
            std::vector<uint8_t> q;
            for (char *c = line + pos; *c && *c != '\n'; ++c) q.push_back(*c == 'A' ? 0 : *c == 'C' ? 1 : *c == 'G' ? 2 : *c == 'T' ? 3 : 4);
            int score = 0, n_cigar = 0, NM = 0;
            uint32_t *cg = bwa_gen_cigar2(opt->mat, opt->o_del, opt->e_del, opt->o_ins, opt->e_ins, w, bns->l_pac, pac, (int)q.size(), q.data(), rb, re, &score, &n_cigar, &NM);
            int32_t hd[3] = { score, cg ? n_cigar : -1, NM };
            fwrite(hd, 4, 3, fo);
            if (cg) {
                fwrite(cg, 4, n_cigar, fo);
                const char *md = (const char *)(cg + n_cigar);
                size_t l = strlen(md) + 1, padded = (l + 3) & ~(size_t)3;
                std::vector<char> buf(padded, 0); memcpy(buf.data(), md, l);
                fwrite(buf.data(), 1, padded, fo);
                free(cg);
            }
        }
        fclose(fi); fclose(fo);
        return 0;
    }
    if (argc - optind < 3) {
        fprintf(stderr, "usage: refdump [mem options] <idx_prefix> <reads.fq|reads.txt> <out.bin>\n"
                        "       refdump [scoring options] cigar <idx_prefix> <tasks.txt> <out.bin>\n"
                        "       refdump [scoring options] ksw <pairs.txt> <out.bin>\n"
                        "       refdump [scoring options] bsw <w> <end_bonus> <pairs.txt> <out.bin>\n"
                        "       refdump [scoring options] bswtime <w> <end_bonus> <threads> <pairs.bin> <out.bin>\n");
        return 1;
    }
    const char *prefix = argv[optind], *reads_fn = argv[optind + 1], *out_fn = argv[optind + 2];

    FMI_search *fmi = new FMI_search(prefix);
    fmi->load_index();
    int64_t l_pac = fmi->idx->bns->l_pac, rlen = l_pac * 2;
    uint8_t *ref_string = (uint8_t *)_mm_malloc(rlen + 64, 64);
    {
        std::string fn = std::string(prefix) + ".0123";
        FILE *f = fopen(fn.c_str(), "rb");
        if (!f || (int64_t)fread(ref_string, 1, rlen, f) != rlen) { fprintf(stderr, "cannot read %s\n", fn.c_str()); return 1; }
        fclose(f);
    }
    // reads: FASTQ (4-line records) or one sequence per line
    std::vector<std::string> reads;
    {
        FILE *f = fopen(reads_fn, "r");
        if (!f) { fprintf(stderr, "cannot open %s\n", reads_fn); return 1; }
        char *line = 0; size_t cap = 0; ssize_t n; long ln = 0; int fq = -1;
        while ((n = getline(&line, &cap, f)) > 0) {
            while (n > 0 && (line[n - 1] == '\n' || line[n - 1] == '\r')) line[--n] = 0;
            if (fq < 0) fq = (line[0] == '@');
            if (fq) { if (ln % 4 == 1) reads.push_back(line); }
            else if (n > 0) reads.push_back(line);
            ln++;
        }
        fclose(f);
    }
    int n = (int)reads.size();
    fprintf(stderr, "[refdump] %d reads, l_pac=%ld\n", n, (long)l_pac);

    bseq1_t *seqsA = copy_reads(reads), *seqsB = copy_reads(reads);
    mem_cache mmcA, mmcB; memset(&mmcA, 0, sizeof(mmcA)); memset(&mmcB, 0, sizeof(mmcB));
    alloc_cache(mmcA); alloc_cache(mmcB);
    mem_chain_v *chainA = (mem_chain_v *)calloc(n, sizeof(mem_chain_v)), *chainB = (mem_chain_v *)calloc(n, sizeof(mem_chain_v));
    mem_alnreg_v *regsA = (mem_alnreg_v *)calloc(n, sizeof(mem_alnreg_v)), *regsB = (mem_alnreg_v *)calloc(n, sizeof(mem_alnreg_v));
    mem_seed_t *seedBufA = (mem_seed_t *)calloc((size_t)n * AVG_SEEDS_PER_READ + 64, sizeof(mem_seed_t));
    mem_seed_t *seedBufB = (mem_seed_t *)calloc((size_t)n * AVG_SEEDS_PER_READ + 64, sizeof(mem_seed_t));
    const bntseq_t *bns = fmi->idx->bns; const uint8_t *pac = fmi->idx->pac;

    Section *s_cnt = sec("COUNTS");
    int64_t n_smem_tot = 0, n_sa_tot = 0;
    for (int st = 0; st < n; st += BATCH_SIZE) {
        int nseq = (st + BATCH_SIZE < n) ? BATCH_SIZE : n - st;
        int64_t seedBufSz = (nseq < BATCH_SIZE) ? (int64_t)(n - st) * AVG_SEEDS_PER_READ : (int64_t)BATCH_SIZE * AVG_SEEDS_PER_READ;
        // ---------------- A: step-by-step glue with dumps ----------------
        bseq1_t *seq_ = seqsA + st;
        int64_t tot_len = 0;
        for (int l = 0; l < nseq; l++) {
            char *seq = seq_[l].seq; int len = seq_[l].l_seq; tot_len += len;
            for (int i = 0; i < len; ++i) seq[i] = seq[i] < 4 ? seq[i] : nst_nt4_table[(int)seq[i]];
        }
        grow_for(mmcA, tot_len);
        int64_t num_smem = 0;
        SMEM *matchArray = mem_collect_smem(fmi, opt, seq_, nseq, mmcA.matchArray[0], mmcA.min_intv_ar[0],
                                            mmcA.query_pos_ar[0], mmcA.enc_qdb[0], mmcA.rid[0], &mmcA, num_smem, 0);
        {
            Section *s = sec("SMEM");
            for (int64_t i = 0; i < num_smem; i++) {
                DSmem d = { st + (int)matchArray[i].rid, (int)matchArray[i].m, (int)matchArray[i].n, 0,
                            matchArray[i].k, matchArray[i].l, matchArray[i].s };
                put(s, d);
            }
            n_smem_tot += num_smem;
            // SA coordinates, per read, in the order mem_chain_seeds consumes them (:876-905)
            Section *sa = sec("SACOORD"), *sn = sec("SACNT");
            std::vector<int32_t> cnt(nseq, 0);
            int64_t i = 0;
            while (i < num_smem) {
                int64_t j = i; while (j < num_smem && matchArray[j].rid == matchArray[i].rid) j++;
                int64_t tot = 0;
                for (int64_t t = i; t < j; t++) {
                    int64_t s_ = matchArray[t].s; tot += s_ < opt->max_occ ? s_ : opt->max_occ;
                }
                std::vector<int64_t> coord(tot + 8);
                int64_t cnt_ = 0, id = 0;
                fmi->get_sa_entries_prefetch(&matchArray[i], coord.data(), &cnt_, j - i, opt->max_occ, 0, id);
                for (int64_t t = 0; t < cnt_; t++) put(sa, coord[t]);
                cnt[matchArray[i].rid] = (int32_t)cnt_;
                n_sa_tot += cnt_;
                i = j;
            }
            for (int l = 0; l < nseq; l++) put(sn, cnt[l]);
        }
        mem_chain_seeds(fmi, opt, bns, seq_, nseq, 0, chainA + st, seedBufA + (int64_t)st * AVG_SEEDS_PER_READ,
                        seedBufSz, matchArray, num_smem);
        dump_chains("CHN0", "SEED0", st, nseq, chainA + st);
        for (int l = 0; l < nseq; l++) chainA[st + l].n = mem_chain_flt(opt, chainA[st + l].n, chainA[st + l].a, 0);
        for (int l = 0; l < nseq; l++) mem_flt_chained_seeds(opt, bns, pac, seq_, chainA[st + l].n, chainA[st + l].a);
        dump_chains("CHN1", "SEED1", st, nseq, chainA + st);
        mem_alnreg_v *regs = regsA + st;
        for (int l = 0; l < nseq; l++) kv_init(regs[l]);
        mem_chain2aln_across_reads_V2(opt, bns, pac, seq_, nseq, chainA + st, regs, &mmcA, ref_string, 0);
        dump_regs("REGRAW", st, nseq, regs);
        for (int l = 0; l < nseq; l++) {             // mem_kernel2_core :1126-1169
            mem_alnreg_t *a = regs[l].a; int nn = regs[l].n, m = 0;
            for (int i = 0; i < nn; ++i) if (a[i].qe > a[i].qb) { if (m != i) a[m++] = a[i]; else ++m; }
            regs[l].n = m;
        }
        dump_regs("REGPRG", st, nseq, regs);
        for (int l = 0; l < nseq; l++)
            regs[l].n = mem_sort_dedup_patch(opt, bns, pac, (uint8_t *)seq_[l].seq, regs[l].n, regs[l].a);
        for (int l = 0; l < nseq; l++)
            for (size_t i = 0; i < regs[l].n; ++i) {
                mem_alnreg_t *p = &regs[l].a[i];
                if (p->rid >= 0 && bns->anns[p->rid].is_alt) p->is_alt = 1;
            }
        dump_regs("REGFIN", st, nseq, regs);
        // ---------------- B: the real kernels, as a check on the glue ----------------
        mem_kernel1_core(fmi, opt, seqsB + st, nseq, chainB + st, seedBufB + (int64_t)st * AVG_SEEDS_PER_READ,
                         seedBufSz, &mmcB, 0);
        mem_kernel2_core(fmi, opt, seqsB + st, regsB + st, nseq, chainB + st, &mmcB, ref_string, 0);
        for (int l = 0; l < nseq; l++) {
            mem_alnreg_v *x = &regsA[st + l], *y = &regsB[st + l];
            int bad = x->n != y->n;
            for (size_t i = 0; !bad && i < x->n; i++) {
                mem_alnreg_t *p = &x->a[i], *q = &y->a[i];
                bad = p->rb != q->rb || p->re != q->re || p->qb != q->qb || p->qe != q->qe || p->score != q->score ||
                      p->truesc != q->truesc || p->w != q->w || p->seedcov != q->seedcov || p->rid != q->rid ||
                      p->seedlen0 != q->seedlen0 || p->n_comp != q->n_comp || p->is_alt != q->is_alt;
            }
            if (bad) { fprintf(stderr, "[refdump] glue != mem_kernel{1,2}_core at read %d\n", st + l); return 2; }
        }
    }
    put(s_cnt, (int64_t)n); put(s_cnt, n_smem_tot); put(s_cnt, n_sa_tot);
    FILE *fo = fopen(out_fn, "wb");
    if (!fo) { fprintf(stderr, "cannot write %s\n", out_fn); return 1; }
    for (auto s : g_sections) {
        char tag[8]; memset(tag, 0, 8); strncpy(tag, s->tag.c_str(), 8);
        int64_t nb = (int64_t)s->data.size();
        fwrite(tag, 1, 8, fo); fwrite(&nb, 8, 1, fo); if (nb) fwrite(s->data.data(), 1, nb, fo);
    }
    fclose(fo);
    fprintf(stderr, "[refdump] wrote %s: %ld smems, %ld sa coords\n", out_fn, (long)n_smem_tot, (long)n_sa_tot);
    return 0;
}
