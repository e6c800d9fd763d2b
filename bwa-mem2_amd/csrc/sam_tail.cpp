// sam_tail.cpp -- host side, rows 1-3 of SURVEY.md 8(f): worker_sam (bwamem.cpp:1216-1337) for single-end and paired-end
// chunks.  Input: the mem_alnreg_v contents of every read (bm2_finish_regs).  Output: the SAM alignment lines, byte for
// byte what `bwa-mem2 mem` prints.  Plain C++ on the host: per read a sort, a few banded global alignments with backtrack
// (one per output record and per XA alternative), for pairs the insert-size statistics of the chunk (mem_pestat), mate
// rescue (mem_matesw: a local Smith-Waterman in the shape of the reference's SSE2 kernel, whose quirks are observable) and
// pairing (mem_pair) -- branchy and order-sensitive.  The two heavy parts run as device batches through the hooks of host_tail.h (the
// mate-rescue SW: matesw.hip; CIGAR / NM / MD: cigar.hip); their host twins here are the kernels' oracles and the no-GPU entry points.
#include <limits.h>
#include <memory>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include "../../include/bm2.h"
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <unordered_map>
#include "ksort_host.h"
#include "host_tail.h"
#include "host_pool.h"
#include <malloc.h>

void bm2_set_error(const char *fmt, ...);

// A chunk's tail runs on a few hundred threads that allocate and release small blocks all the time.  glibc gives every thread an arena and
// by default hands the top of an arena back to the kernel (madvise / munmap) whenever 128 KB of it are free, and takes blocks above 128 KB
// straight from mmap: on a 256-thread host every such call is a TLB shootdown across the process and a turn on its address-space lock, and
// the phases of a chunk were seen to take 5 ms or 800 ms depending on what the other threads were doing.  So: never trim, big blocks from
// the arenas too (up to glibc's cap of 32 MB).  That is a PROCESS-WIDE setting -- the host's own allocations stop returning memory to
// the kernel too -- so it is the host's decision: it happens (once) only when the host asks for it, with BM2_MALLOC_TUNE=1 in the environment
// (bench.py and the compiled binding do; include/bm2.h).  Since round 3 the tail allocates from per-thread arenas and append buffers and
// needs the setting far less than it did.
void bm2_tune_malloc_once() {
    static std::once_flag once;
    std::call_once(once, []() {
        const char *e = getenv("BM2_MALLOC_TUNE");
        if (!e || e[0] != '1') return;
        mallopt(M_TRIM_THRESHOLD, 1 << 30);
        mallopt(M_MMAP_THRESHOLD, 32 << 20);
        mallopt(M_TOP_PAD, 64 << 20);
    });
}

namespace {
template <class F> void run_threads(int n_threads, F f) { bm2_run_threads(n_threads, std::function<void()>(f)); }

enum { F_NOPAIRING = 0x4, F_ALL = 0x8, F_NO_MULTI = 0x10, F_NO_RESCUE = 0x20, F_REF_HDR = 0x100, F_SOFTCLIP = 0x200, F_PRIMARY5 = 0x800, F_KEEP_SUPP_MAPQ = 0x1000 };
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

template <class S> void put_int(S &s, long long v) {           // kputw / kputl (kstring.h:92-141): plain decimal
    char b[24]; int l = 24;
    unsigned long long x = v < 0 ? 0ULL - (unsigned long long)v : (unsigned long long)v;
    do { b[--l] = (char)('0' + x % 10); x /= 10; } while (x);
    if (v < 0) b[--l] = '-';
    s.append(b + l, (size_t)(24 - l));
}

// ---- memory of the text pass.  A pair's records are built from a dozen small arrays (CIGARs with their clips, XA strings, the list of
// printed alignments); from the general-purpose allocator they were the largest single cost of the pass on a many-core host.  They
// live in a per-thread bump arena that is reset per pair (nothing is freed one by one), and the text itself goes into a plain
// growing buffer whose appends do not maintain a terminator.
struct Arena {
    std::vector<std::pair<char *, size_t>> blocks; size_t cur = 0, at = 0;
    Arena() {}
    Arena(const Arena &) = delete;
    ~Arena() { for (auto &b : blocks) free(b.first); }
    void *get(size_t bytes) {
        bytes = (bytes + 15) & ~(size_t)15;
        while (cur < blocks.size() && at + bytes > blocks[cur].second) { ++cur; at = 0; }
        if (cur == blocks.size()) {
            size_t sz = blocks.empty() ? (size_t)1 << 16 : blocks.back().second * 2;
            if (sz < bytes) sz = bytes;
            char *m = (char *)malloc(sz);
            if (!m) throw std::bad_alloc();
            blocks.emplace_back(m, sz); at = 0;
        }
        void *r = blocks[cur].first + at; at += bytes;
        return r;
    }
    void reset() { cur = 0; at = 0; }
};
thread_local Arena t_scratch;

template <class T> struct AVec {    // a vector of trivially copyable T in the thread's arena (valid until the arena is reset)
    T *p = nullptr; int n = 0, cap = 0;
    void reserve(int c) {
        if (c <= cap) return;
        int nc = cap * 2 > c ? cap * 2 : c; if (nc < 4) nc = 4;
        T *q = (T *)t_scratch.get(sizeof(T) * (size_t)nc);
        if (n) memcpy((void *)q, (const void *)p, sizeof(T) * (size_t)n);
        p = q; cap = nc;
    }
    void push_back(const T &v) { reserve(n + 1); p[n++] = v; }
    void append(const T *b, size_t k) { reserve(n + (int)k); if (k) memcpy((void *)(p + n), (const void *)b, sizeof(T) * k); n += (int)k; }
    void assign(const T *b, const T *e) { n = 0; append(b, (size_t)(e - b)); }
    void fill(int count, const T &v) { n = 0; reserve(count); for (int i = 0; i < count; ++i) p[i] = v; n = count; }
    void clear() { n = 0; }
    void erase_front() { if (n > 1) memmove((void *)p, (const void *)(p + 1), sizeof(T) * (size_t)(n - 1)); --n; }
    void insert_front(const T &v) { reserve(n + 1); if (n) memmove((void *)(p + 1), (const void *)p, sizeof(T) * (size_t)n); p[0] = v; ++n; }
    void pop_back() { --n; }
    T &back() { return p[n - 1]; }
    const T &back() const { return p[n - 1]; }
    bool empty() const { return n == 0; }
    int size() const { return n; }
    T &operator[](int i) { return p[i]; }
    const T &operator[](int i) const { return p[i]; }
    const T *begin() const { return p; }
    const T *end() const { return p + n; }
    void operator+=(const char *z) { append((const T *)z, strlen(z)); }       // (T = char)
};

struct Text {                       // the output of a thread: appends only, no terminator
    char *b = nullptr; size_t n = 0, cap = 0;
    Text() {}
    Text(const Text &) = delete;
    ~Text() { free(b); }
    void grow(size_t k) {
        size_t nc = cap * 2 > n + k ? cap * 2 : n + k; if (nc < 4096) nc = 4096;
        char *q = (char *)realloc(b, nc);
        if (!q) throw std::bad_alloc();
        b = q; cap = nc;
    }
    void need(size_t k) { if (n + k > cap) grow(k); }
    void push_back(char c) { need(1); b[n++] = c; }
    void append(const char *p, size_t k) { need(k); memcpy(b + n, p, k); n += k; }
    void operator+=(const char *z) { append(z, strlen(z)); }
    char *extend(size_t k) { need(k); char *r = b + n; n += k; return r; }    // k bytes to be written in place
    void clear() { n = 0; }
    size_t size() const { return n; }
    size_t capacity() const { return cap; }
    void release() { free(b); b = nullptr; n = cap = 0; }
    bool empty() const { return n == 0; }
    const char *data() const { return b; }
};
struct MdOut {                      // gen_cigar's MD sink: appends like Text, on a std::string (the C ABI's bm2_gen_cigar hands the string out)
    std::string &s;
    void push_back(char c) { s.push_back(c); }
    void append(const char *p, size_t k) { s.append(p, k); }
};

// ---- banded global alignment with traceback ---------------------------------------------------------------------------
// Banded global alignment with traceback; must reproduce ksw_global2 (ksw.cpp:558-668) decision for decision, because the CIGAR of a
// rescued hit is printed from it.  Only the rescued hits of a chunk come here (everything numbered goes through the device batch), so this
// is written for clarity: the row state lives in two plain arrays (H of the row above, shifted to the column it is the diagonal
// predecessor of; E of the row being entered) instead of an array of structs, a cell's traceback record is three named fields (where H
// came from; whether E / F were extended rather than opened) instead of ksw's shifted byte, and the walk back is an explicit
// three-state machine.
// The reference's preferences at ties are what make the result unique: H takes M over E over F; E and F prefer opening over extending.
int global_align(int qlen, const uint8_t *query, int tlen, const uint8_t *target, const int8_t *mat, int o_del, int e_del, int o_ins,
                 int e_ins, int w, std::vector<uint32_t> &cigar) {
    enum : uint8_t { FROM_M = 0, FROM_E = 1, FROM_F = 2, E_EXTENDED = 4, F_EXTENDED = 8 };
    const int open_del = o_del + e_del, open_ins = o_ins + e_ins;
    const int width = std::min(qlen, 2 * w + 1);                // cells of a row inside the band
    static thread_local std::vector<uint8_t> trace_tl;         // [tlen][width], column index relative to the row's first band cell
    static thread_local std::vector<int32_t> rows_tl;          // two rows of qlen + 2: H(i-1, j-1) shifted to column j; E(i, j)
    if (trace_tl.size() < (size_t)width * (size_t)tlen) trace_tl.resize((size_t)width * (size_t)tlen);
    if (rows_tl.size() < 2 * ((size_t)qlen + 2)) rows_tl.resize(2 * ((size_t)qlen + 2));
    uint8_t *const trace = trace_tl.data();
    int32_t *const __restrict Hup = rows_tl.data(), *const __restrict Eup = Hup + qlen + 2;
    // row -1: only leading insertions, and only inside the band
    for (int j = 0; j <= qlen; ++j) { Hup[j] = j == 0 ? 0 : (j <= w ? -(o_ins + e_ins * j) : MINUS_INF); Eup[j] = MINUS_INF; }
    auto band_begin = [&](int i) { return i > w ? i - w : 0; };
    auto band_end = [&](int i) { return std::min(i + w + 1, qlen); };
    for (int i = 0; i < tlen; ++i) {
        const int lo = band_begin(i), hi = band_end(i);
        const int8_t *score_of = mat + target[i] * 5;
        uint8_t *const __restrict rec = trace + (size_t)i * width;
        int f = MINUS_INF;                                       // F(i, j): the only state that travels along the row
        int carry = lo == 0 ? -(o_del + e_del * (i + 1)) : MINUS_INF;     // H(i, lo - 1): becomes column lo's diagonal predecessor in the next row
        for (int j = lo; j < hi; ++j) {
            const int m = Hup[j] + score_of[query[j]], e = Eup[j];
            uint8_t r = FROM_M;
            int h = m;
            if (e > h) { h = e; r = FROM_E; }                    // H prefers M, then E, then F
            if (f > h) { h = f; r = FROM_F; }
            Hup[j] = carry; carry = h;                           // (the row shifts by one: next row's diagonal predecessor of column j + 1)
            const int e_open = m - open_del, e_ext = e - e_del;  // E(i + 1, j): opening wins a tie
            if (e_ext > e_open) { Eup[j] = e_ext; r |= E_EXTENDED; } else Eup[j] = e_open;
            const int f_open = m - open_ins, f_ext = f - e_ins;  // F(i, j + 1)
            if (f_ext > f_open) { f = f_ext; r |= F_EXTENDED; } else f = f_open;
            rec[j - lo] = r;
        }
        Hup[hi] = carry; Eup[hi] = MINUS_INF;
    }
    const int score = Hup[qlen];
    cigar.clear();
    auto emit = [&](uint32_t op, uint32_t len) {                 // runs are merged as they are found (back to front)
        if (!cigar.empty() && (cigar.back() & 0xf) == op) cigar.back() += len << 4;
        else cigar.push_back(len << 4 | op);
    };
    int i = tlen - 1, j = band_end(tlen - 1) - 1;
    uint8_t state = FROM_M;                                      // in which of the three matrices the path currently is
    while (i >= 0 && j >= 0) {
        const uint8_t r = trace[(size_t)i * width + (size_t)(j - band_begin(i))];
        if (state == FROM_M) state = r & 3;                      // leave H through the matrix it was taken from
        else if (state == FROM_E) state = (r & E_EXTENDED) ? FROM_E : FROM_M;
        else state = (r & F_EXTENDED) ? FROM_F : FROM_M;
        if (state == FROM_M) { emit(0, 1); --i; --j; }
        else if (state == FROM_E) { emit(2, 1); --i; }
        else { emit(1, 1); --j; }
    }
    if (i >= 0) emit(2, (uint32_t)i + 1);
    if (j >= 0) emit(1, (uint32_t)j + 1);
    std::reverse(cigar.begin(), cigar.end());
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
    static thread_local std::vector<uint8_t> rseq, q;
    rseq.assign(R.ref_string + rb, R.ref_string + re); q.assign(query, query + l_query);
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
        MdOut MDo{MD};
        int x = 0, y = 0, u = 0, n_mm = 0, n_gap = 0;
        const char *int2base = rb < l_pac ? "ACGTN" : "TGCAN";
        const int n = (int)cigar.size();
        for (int k = 0; k < n; ++k) {
            const int op = cigar[k] & 0xf, len = (int)(cigar[k] >> 4);
            if (op == 0) {
                for (int i = 0; i < len; ++i) {
                    if (q[x + i] != rseq[y + i]) { put_int(MDo, u); MD.push_back(int2base[rseq[y + i]]); ++n_mm; u = 0; }
                    else ++u;
                }
                x += len; y += len;
            } else if (op == 2) {
                if (k > 0 && k < n - 1) {
                    put_int(MDo, u); MD.push_back('^');
                    for (int i = 0; i < len; ++i) MD.push_back(int2base[rseq[y + i]]);
                    u = 0; n_gap += len;
                }
                y += len;
            } else if (op == 1) { x += len; n_gap += len; }
        }
        put_int(MDo, u);
        *NM = n_mm + n_gap;
    }
    return true;
}

struct Aln {                        // mem_aln_t (bwamem.h:168-178); CIGAR, MD and XA live in the thread's arena or in the batch's results
    int64_t pos = -1; int rid = -1, flag = 0, is_rev = 0, is_alt = 0, mapq = 0, NM = 0;
    AVec<uint32_t> cigar; const char *MD = ""; const AVec<char> *XA = nullptr;
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

// ---- CIGAR generation as a batch.  A hit's alignment depends on the hit alone (qb, qe, rb, re, truesc, w and the read), and nothing in
// the flow looks at a CIGAR before the text is written.  So the tail numbers the chunk's hits (bm2_alnreg_t::pad = index + 1; hits
// that mate rescue creates later carry 0), runs the flow once DRY -- reg2aln only notes which hits it is asked for -- sends those
// through one hook call (bm2h_cigar_batch_fn: the device kernel's shape), and runs the flow again with the results at hand.  Hits
// without a number (rescued ones) are aligned in place.
struct CgMemo {                      // results of the batch; task_of[hit number] = position in the batch or -1
    std::vector<int32_t> task_of;
    std::vector<bm2h_cg_hit> hits;
    bm2h_cg_out out;
};
struct CgStats { std::atomic<long long> planned{0}, used{0}, missed{0}; };
// (the per-lookup counts are tallied per thread and added to the shared counters when the thread is through with its blocks: a
//  shared atomic per lookup, hit by a few hundred threads, took longer than the rest of the pass)
thread_local long long t_cg_used = 0, t_cg_missed = 0, t_rs_used = 0, t_rs_missed = 0;
struct CgSession { int mode = 0; std::vector<int32_t> *rec = nullptr; const CgMemo *memo = nullptr; CgStats *st = nullptr; };   // mode 1 = record, 2 = replay
thread_local CgSession t_cg;

// the band of the first try (bwamem.cpp:1743-1747) and the retry loop (:1748-1766) of mem_reg2aln around bwa_gen_cigar2
int reg2aln_band(const bm2_opt *opt, int qb, int qe, int64_t rb, int64_t re, int truesc, int w_hit) {
    const int tmp = infer_bw(qe - qb, (int)(re - rb), truesc, opt->a, opt->o_del, opt->e_del);
    int w2 = infer_bw(qe - qb, (int)(re - rb), truesc, opt->a, opt->o_ins, opt->e_ins);
    w2 = w2 > tmp ? w2 : tmp;
    if (w2 > opt->w) w2 = w2 < w_hit ? w2 : w_hit;
    return w2;
}
bool cigar_with_retries(const bm2_opt *opt, const Ref &R, const uint8_t *query, int qb, int qe, int64_t rb, int64_t re, int truesc, int w_hit,
                        int *score, std::vector<uint32_t> &cigar, int *NM, std::string &MD) {
    int w2 = reg2aln_band(opt, qb, qe, rb, re, truesc, w_hit), i = 0, last_sc = -(1 << 30);
    do {
        w2 = w2 < opt->w << 2 ? w2 : opt->w << 2;
        if (!gen_cigar(opt->mat, opt->o_del, opt->e_del, opt->o_ins, opt->e_ins, w2, R, qe - qb, query + qb, rb, re, score, cigar, NM, MD)) return false;
        if (*score == last_sc || w2 == opt->w << 2) break;
        last_sc = *score;
        w2 <<= 1;
    } while (++i < 3 && *score < truesc - opt->a);
    return true;
}

// mem_reg2aln, bwamem.cpp:1732-1805; ar == NULL -> the unmapped record
bool reg2aln(const bm2_opt *opt, const bm2_sam_opt *so, const Ref &R, int l_query, const uint8_t *query, const bm2_alnreg_t *ar, Aln &a) {
    a = Aln();
    if (ar == 0 || ar->rb < 0 || ar->re < 0) { a.rid = -1; a.pos = -1; a.flag |= 0x4; return true; }
    const int qb = ar->qb, qe = ar->qe;
    const int64_t rb = ar->rb, re = ar->re;
    a.mapq = ar->secondary < 0 ? approx_mapq_se(opt, so, ar) : 0;
    if (ar->secondary >= 0) a.flag |= 0x100;
    int score = 0, NM = 0;
    if (t_cg.mode == 1) {                                        // dry pass: note that this hit is asked for; no alignment
        if (ar->pad > 0) t_cg.rec->push_back(ar->pad - 1);
        int is_rev;
        const int64_t pos = R.depos(rb < R.l_pac ? rb : re - 1, &is_rev);
        a.is_rev = is_rev; a.rid = R.pos2rid(pos); if (a.rid < 0) a.rid = 0;
        a.pos = pos - R.off[a.rid]; a.cigar.fill(1, (uint32_t)(qe - qb) << 4); a.score = ar->score; a.is_alt = ar->is_alt;
        return true;
    }
    int at = -1;
    if (t_cg.mode == 2 && ar->pad > 0) {
        const CgMemo &M = *t_cg.memo;
        if ((size_t)(ar->pad - 1) < M.task_of.size()) at = M.task_of[(size_t)(ar->pad - 1)];
        if (at >= 0) {                                           // (the hit must still be the one the batch aligned)
            const bm2h_cg_hit &h = M.hits[(size_t)at];
            if (h.qb != qb || h.qe != qe || h.rb != rb || h.re != re || h.truesc != ar->truesc || h.w != ar->w) at = -1;
        }
        if (at >= 0) ++t_cg_used; else ++t_cg_missed;
    }
    bool ok;
    if (at >= 0) {                                               // the batch has it
        const bm2h_cg_out &O = t_cg.memo->out;
        ok = O.n_cigar[(size_t)at] >= 0;
        score = O.score[(size_t)at]; NM = O.nm[(size_t)at];
        a.cigar.clear(); a.MD = "";
        if (ok) {                                                // (the ops are copied: clips are added below; the MD string is read where it lies)
            a.cigar.assign(O.cigar.data() + O.cigar_off[(size_t)at], O.cigar.data() + O.cigar_off[(size_t)at] + O.n_cigar[(size_t)at]);
            a.MD = O.md.data() + O.md_off[(size_t)at];
        }
    } else {                                                     // a rescued hit (or a hit the batch does not know): aligned here
        static thread_local std::vector<uint32_t> cg_tl; static thread_local std::string md_tl;
        ok = cigar_with_retries(opt, R, query, qb, qe, rb, re, ar->truesc, ar->w, &score, cg_tl, &NM, md_tl);
        if (ok) {
            a.cigar.assign(cg_tl.data(), cg_tl.data() + cg_tl.size());
            char *m = (char *)t_scratch.get(md_tl.size() + 1);
            memcpy(m, md_tl.c_str(), md_tl.size() + 1);
            a.MD = m;
        }
    }
    if (!ok) return false;                                      // the reference asserts a.cigar != NULL here
    a.NM = NM;
    int is_rev;
    int64_t pos = R.depos(rb < R.l_pac ? rb : re - 1, &is_rev);
    a.is_rev = is_rev;
    if (!a.cigar.empty()) {                                     // squeeze out a leading or a trailing deletion
        if ((a.cigar[0] & 0xf) == 2) { pos += a.cigar[0] >> 4; a.cigar.erase_front(); }
        else if ((a.cigar.back() & 0xf) == 2) a.cigar.pop_back();
    }
    if (qb != 0 || qe != l_query) {                             // clipping
        const int clip5 = is_rev ? l_query - qe : qb, clip3 = is_rev ? qb : l_query - qe;
        if (clip5) a.cigar.insert_front((uint32_t)clip5 << 4 | 3);
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

// ---- which hits of a read are primary (mem_mark_primary_se, bwamem.cpp:1392-1465).  The rule, in this file's words: hits are RANKED
// (score, then non-ALT first, then a hash of the read's number: ties must not depend on the input order); going down the ranking, a hit
// whose stretch of the read is mostly covered (mask_level of the shorter stretch) by a higher-ranked LEADER becomes that leader's
// follower (`secondary` = the leader's rank); the first follower's score is the leader's `sub`, followers scoring within one
// mismatch / gap of the leader count in `sub_n`.  With ALT contigs the ranking is done twice: over all hits (which gives `secondary_all`
// and the ALT score of a primary hit shadowed by an ALT hit), then with the primary-assembly hits moved to the front and judged among
// themselves.
struct QSpan { int b, e; int len() const { return e - b; } };
inline QSpan span_of(const bm2_alnreg_t &h) { return QSpan{ h.qb, h.qe }; }
inline bool mostly_covered(const QSpan &x, const QSpan &y, float mask_level) {         // the shared stretch against mask_level of the shorter one
    const int from = std::max(x.b, y.b), to = std::min(x.e, y.e);
    return to > from && to - from >= std::min(x.len(), y.len()) * mask_level;
}
void follow_leaders(const bm2_opt *opt, int n, bm2_alnreg_t *a, AVec<int> &leaders) {  // mem_mark_primary_se_core, bwamem.cpp:1392-1418
    const int close = std::max(opt->a + opt->b, std::max(opt->o_del + opt->e_del, opt->o_ins + opt->e_ins));
    leaders.clear();
    for (int i = 0; i < n; ++i) {
        const QSpan mine = span_of(a[i]);
        const int *lead = std::find_if(leaders.begin(), leaders.end(), [&](int j) { return mostly_covered(span_of(a[j]), mine, opt->mask_level); });
        if (lead == leaders.end()) { leaders.push_back(i); continue; }                  // (rank 0 always leads: the list is empty then)
        bm2_alnreg_t &L = a[*lead];
        if (L.sub == 0) L.sub = a[i].score;
        if (L.score - a[i].score <= close && (L.is_alt || !a[i].is_alt)) ++L.sub_n;
        a[i].secondary = *lead;
    }
}

int mark_primary_se(const bm2_opt *opt, int n, bm2_alnreg_t *a, int64_t id) {
    if (n == 0) return 0;
    AVec<int> scratch;                                           // (the thread's scratch arena: reset per pair / read by the caller)
    int n_assembly = 0;                                          // hits on the primary assembly (not on an ALT contig)
    for (int i = 0; i < n; ++i) {
        bm2_alnreg_t &h = a[i];
        h.sub = h.alt_sc = 0; h.secondary = h.secondary_all = -1; h.hash = hash_64((uint64_t)(id + i));
        n_assembly += !h.is_alt;
    }
    auto by_score = [](const bm2_alnreg_t &x, const bm2_alnreg_t &y) {                  // alnreg_hlt, bwamem.cpp:155
        if (x.score != y.score) return x.score > y.score;
        if (x.is_alt != y.is_alt) return x.is_alt < y.is_alt;
        return x.hash < y.hash;
    };
    auto assembly_first = [](const bm2_alnreg_t &x, const bm2_alnreg_t &y) {            // alnreg_hlt2, bwamem.cpp:158
        if (x.is_alt != y.is_alt) return x.is_alt < y.is_alt;
        if (x.score != y.score) return x.score > y.score;
        return x.hash < y.hash;
    };
    k_introsort((size_t)n, a, by_score);
    follow_leaders(opt, n, a, scratch);
    for (int i = 0; i < n; ++i) {
        a[i].secondary_all = i;                                  // (for now: the hit's rank in the all-hits ranking)
        const int lead = a[i].secondary;
        if (!a[i].is_alt && lead >= 0 && a[lead].is_alt) a[i].alt_sc = a[lead].score;
    }
    if (n_assembly == n) {                                       // no ALT hit: one ranking serves both purposes
        for (int i = 0; i < n; ++i) a[i].secondary_all = a[i].secondary;
        return n_assembly;
    }
    if (n_assembly > 0) k_introsort((size_t)n, a, assembly_first);
    AVec<int> &now_at = scratch;                                 // rank in the all-hits ranking -> position after the second sort
    now_at.fill(n, 0);
    for (int i = 0; i < n; ++i) now_at[a[i].secondary_all] = i;
    for (int i = 0; i < n; ++i) {
        bm2_alnreg_t &h = a[i];
        if (h.secondary < 0) { h.secondary_all = -1; continue; }
        h.secondary_all = now_at[h.secondary];
        if (h.is_alt) h.secondary = INT_MAX;
    }
    if (n_assembly > 0) {                                        // the assembly hits among themselves
        for (int i = 0; i < n_assembly; ++i) { a[i].sub = 0; a[i].secondary = -1; }
        follow_leaders(opt, n_assembly, a, scratch);
    }
    return n_assembly;
}

// `-5` (mem_reorder_primary5, bwamem.cpp:1496-1519): of the reportable primary hits (leading, on the assembly, score >= T) the one that
// starts leftmost on the read goes first; every reference to the two positions that trade places is renamed.
void reorder_primary5(int T, int n, bm2_alnreg_t *a) {
    auto reportable = [&](const bm2_alnreg_t &h) { return h.secondary < 0 && !h.is_alt && h.score >= T; };
    int count = 0, leftmost = -1;
    for (int k = 0; k < n; ++k) {
        if (!reportable(a[k])) continue;
        ++count;
        if (leftmost < 0 || a[k].qb < a[leftmost].qb) leftmost = k;                    // (the first of equally left hits stays the choice)
    }
    if (count <= 1 || leftmost == 0) return;
    std::swap(a[0], a[leftmost]);
    auto renamed = [&](int ref) { return ref == 0 ? leftmost : ref == leftmost ? 0 : ref; };
    for (int k = 1; k < n; ++k) { a[k].secondary = renamed(a[k].secondary); a[k].secondary_all = renamed(a[k].secondary_all); }
}

template <class S> void put_cigar(S &s, const AVec<uint32_t> &cg, const char *ops) {
    for (uint32_t c : cg) { put_int(s, c >> 4); s.push_back(ops[c & 0xf]); }
}

// mem_gen_alt, bwamem_extra.cpp:118-183: the XA:Z value of every primary hit ("" = none)
bool gen_alt(const bm2_opt *opt, const bm2_sam_opt *so, const Ref &R, int n, const bm2_alnreg_t *a, int l_query, const uint8_t *query,
             AVec<AVec<char>> &XA, bool &any) {
    auto pri_idx = [&](int i) {
        const int k = a[i].secondary_all;
        return (k >= 0 && a[i].score >= a[k].score * (double)so->XA_drop_ratio) ? k : -1;      // (a double parameter in the reference)
    };
    AVec<int> cnt; cnt.fill(n, 0);
    AVec<char> has_alt; has_alt.fill(n, 0);
    int tot = 0;
    any = false;
    for (int i = 0; i < n; ++i) {
        const int r = pri_idx(i);
        if (r >= 0) { ++cnt[r]; ++tot; if (a[i].is_alt) has_alt[r] = 1; }
    }
    if (tot == 0) return true;
    any = true;
    XA.fill(n, AVec<char>());
    for (int i = 0; i < n; ++i) {
        const int r = pri_idx(i);
        if (r < 0) continue;
        if (cnt[r] > so->max_XA_hits_alt || (!has_alt[r] && cnt[r] > so->max_XA_hits)) continue;
        Aln t;
        if (!reg2aln(opt, so, R, l_query, query, &a[i], t)) return false;
        AVec<char> &s = XA[r];
        s += R.name[t.rid]; s.push_back(','); s.push_back("+-"[t.is_rev]); put_int(s, t.pos + 1); s.push_back(',');
        put_cigar(s, t.cigar, "MIDSHN");
        s.push_back(','); put_int(s, t.NM); s.push_back(';');
    }
    return true;
}

int get_rlen(const AVec<uint32_t> &cg) {                // bwamem.cpp:1820-1829
    int l = 0;
    for (uint32_t c : cg) { const int op = c & 0xf; if (op == 0 || op == 2) l += (int)(c >> 4); }
    return l;
}

// mem_aln2sam (bwamem.cpp:1592-1730); m_ = the mate's alignment or NULL.  The reference edits copies of the two records (an unmapped
// read borrows its mate's position and loses its CIGAR, and the other way round); here the few fields that can change are locals and
// the records themselves are only read -- no copies of CIGAR vectors and MD strings per printed line.
void aln2sam(const bm2_sam_opt *so, const Ref &R, Text &s, const char *name, const char *comment, const char *qual, int l_seq,
             const uint8_t *seq, const AVec<Aln> &list, int which, const Aln *m_) {
    if (t_cg.mode == 1) return;                                  // dry pass of a CIGAR session: decisions only
    static const AVec<uint32_t> no_cigar = AVec<uint32_t>();
    const Aln &P = list[which];
    const int n = (int)list.size();
    const bool has_mate = m_ != nullptr;
    int p_rid = P.rid, p_rev = P.is_rev, p_flag = P.flag;
    int64_t p_pos = P.pos;
    const AVec<uint32_t> *p_cg = &P.cigar;
    int m_rid = has_mate ? m_->rid : -1, m_rev = has_mate ? m_->is_rev : 0;
    int64_t m_pos = has_mate ? m_->pos : -1;
    const AVec<uint32_t> *m_cg = has_mate ? &m_->cigar : &no_cigar;
    p_flag |= has_mate ? 0x1 : 0;
    p_flag |= p_rid < 0 ? 0x4 : 0;
    p_flag |= has_mate && m_rid < 0 ? 0x8 : 0;
    if (p_rid < 0 && has_mate && m_rid >= 0) { p_rid = m_rid; p_pos = m_pos; p_rev = m_rev; p_cg = &no_cigar; }         // copy mate to alignment
    if (has_mate && m_rid < 0 && p_rid >= 0) { m_rid = p_rid; m_pos = p_pos; m_rev = p_rev; m_cg = &no_cigar; }         // copy alignment to mate
    p_flag |= p_rev ? 0x10 : 0;
    p_flag |= has_mate && m_rev ? 0x20 : 0;
    auto put_cigar_of = [&](const AVec<uint32_t> &cg, int is_alt) {     // add_cigar, bwamem.cpp:1579-1590
        if (cg.empty()) { s.push_back('*'); return; }
        for (uint32_t c0 : cg) {
            int c = c0 & 0xf;
            if (!(so->flag & F_SOFTCLIP) && !is_alt && (c == 3 || c == 4)) c = which ? 4 : 3;
            put_int(s, c0 >> 4); s.push_back("MIDSH"[c]);
        }
    };
    s += name; s.push_back('\t');
    put_int(s, (p_flag & 0xffff) | (p_flag & 0x10000 ? 0x100 : 0)); s.push_back('\t');
    if (p_rid >= 0) {
        s += R.name[p_rid]; s.push_back('\t');
        put_int(s, p_pos + 1); s.push_back('\t');
        put_int(s, P.mapq); s.push_back('\t');
        put_cigar_of(*p_cg, P.is_alt);
    } else s += "*\t0\t0\t*";
    s.push_back('\t');
    if (has_mate && m_rid >= 0) {                                // mate position and template length
        if (p_rid == m_rid) s.push_back('='); else s += R.name[m_rid];
        s.push_back('\t');
        put_int(s, m_pos + 1); s.push_back('\t');
        if (p_rid == m_rid) {
            const int64_t p0 = p_pos + (p_rev ? get_rlen(*p_cg) - 1 : 0), p1 = m_pos + (m_rev ? get_rlen(*m_cg) - 1 : 0);
            if (m_cg->empty() || p_cg->empty()) s.push_back('0');
            else put_int(s, -(p0 - p1 + (p0 > p1 ? 1 : p0 < p1 ? -1 : 0)));
        } else s.push_back('0');
    } else s += "*\t0\t0";
    s.push_back('\t');
    if (p_flag & 0x100) s += "*\t*";
    else {
        int qb = 0, qe = l_seq;
        if (!p_cg->empty() && which && !(so->flag & F_SOFTCLIP) && !P.is_alt) {
            const uint32_t c0 = (*p_cg)[0], c1 = p_cg->back();
            if (!p_rev) {
                if ((c0 & 0xf) == 4 || (c0 & 0xf) == 3) qb += c0 >> 4;
                if ((c1 & 0xf) == 4 || (c1 & 0xf) == 3) qe -= c1 >> 4;
            } else {
                if ((c0 & 0xf) == 4 || (c0 & 0xf) == 3) qe -= c0 >> 4;
                if ((c1 & 0xf) == 4 || (c1 & 0xf) == 3) qb += c1 >> 4;
            }
        }
        const size_t len = qe > qb ? (size_t)(qe - qb) : 0;
        char *d = s.extend(len + 1 + (qual ? len : 1));          // bases, tab, qualities (or '*') written in place
        if (!p_rev) {
            for (size_t i = 0; i < len; ++i) d[i] = "ACGTN"[seq[qb + (int)i]];
            d[len] = '\t';
            if (qual) memcpy(d + len + 1, qual + qb, len); else d[len + 1] = '*';
        } else {
            for (size_t i = 0; i < len; ++i) d[i] = "TGCAN"[seq[qe - 1 - (int)i]];
            d[len] = '\t';
            if (qual) { for (size_t i = 0; i < len; ++i) d[len + 1 + i] = qual[qe - 1 - (int)i]; } else d[len + 1] = '*';
        }
    }
    if (!p_cg->empty()) { s += "\tNM:i:"; put_int(s, P.NM); s += "\tMD:Z:"; s += P.MD; }
    if (has_mate && !m_cg->empty()) { s += "\tMC:Z:"; put_cigar_of(*m_cg, m_->is_alt); }
    if (P.score >= 0) { s += "\tAS:i:"; put_int(s, P.score); }
    if (P.sub >= 0) { s += "\tXS:i:"; put_int(s, P.sub); }
    if (so->rg_id && so->rg_id[0]) { s += "\tRG:Z:"; s += so->rg_id; }
    if (!(p_flag & 0x100)) {
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
        if (P.alt_sc > 0) { char b[64]; snprintf(b, sizeof b, "\tpa:f:%.3f", (double)P.score / P.alt_sc); s += b; }
    }
    if (P.XA) { s += "\tXA:Z:"; s.append(P.XA->p, (size_t)P.XA->n); }
    if (comment) { s.push_back('\t'); s += comment; }
    if ((so->flag & F_REF_HDR) && p_rid >= 0 && R.anno && R.anno[p_rid] && R.anno[p_rid][0]) {
        s += "\tXR:Z:";
        for (const char *c = R.anno[p_rid]; *c; ++c) s.push_back(*c == '\t' ? ' ' : *c);
    }
    s.push_back('\n');
}

// mem_reg2sam (bwamem.cpp:1521-1577)
bool reg2sam(const bm2_opt *opt, const bm2_sam_opt *so, const Ref &R, Text &out, const char *name, const char *comment,
             const char *qual, int l_seq, const uint8_t *seq, int n, const bm2_alnreg_t *a, int extra_flag, const Aln *m) {
    AVec<AVec<char>> XA; bool any_xa = false;
    if (!(so->flag & F_ALL)) { if (!gen_alt(opt, so, R, n, a, l_seq, seq, XA, any_xa)) return false; }
    AVec<Aln> aa;
    int l = 0;
    for (int k = 0; k < n; ++k) {
        const bm2_alnreg_t *p = &a[k];
        if (p->score < so->T) continue;
        if (p->secondary >= 0 && (p->is_alt || !(so->flag & F_ALL))) continue;
        if (p->secondary >= 0 && p->secondary < INT_MAX && p->score < a[p->secondary].score * opt->drop_ratio) continue;
        Aln q;                                                   // (filled, then stored: the list may move when it grows)
        if (!reg2aln(opt, so, R, l_seq, seq, p, q)) return false;
        q.XA = (any_xa && !XA[k].empty()) ? &XA[k] : nullptr;   // XA[k] is a NULL pointer in the reference when nothing was appended
        q.flag |= extra_flag;
        if (p->secondary >= 0) q.sub = -1;
        if (l && p->secondary < 0) q.flag |= (so->flag & F_NO_MULTI) ? 0x10000 : 0x800;
        if (!(so->flag & F_KEEP_SUPP_MAPQ) && l && !p->is_alt && q.mapq > aa[0].mapq) q.mapq = aa[0].mapq;
        aa.push_back(q);
        ++l;
    }
    if (aa.empty()) {
        aa.push_back(Aln());
        reg2aln(opt, so, R, l_seq, seq, 0, aa[0]);
        aa[0].flag |= extra_flag;
        aln2sam(so, R, out, name, comment, qual, l_seq, seq, aa, 0, m);
    } else {
        for (int k = 0; k < (int)aa.size(); ++k) aln2sam(so, R, out, name, comment, qual, l_seq, seq, aa, k, m);
    }
    return true;
}

// ================================================================================================== paired-end reads
// ---- the local SW of ksw.cpp:111-381 (ksw_u8 / ksw_i16 / ksw_align2), restated lane by lane.  The SSE2 kernel is striped
// (Farrar): vector j holds query positions j, j+slen, j+2*slen, ...; E(i+1,j) is taken from H before the lazy-F pass, so an
// insertion followed by a deletion is possible inside a stripe segment but not across one -- the result depends on the
// striping, hence the vectors are kept (P = 16 byte lanes or 8 word lanes) instead of a textbook recurrence.
struct KswResult { int score = 0, te = -1, qe = -1, score2 = -1, te2 = -1, tb = -1, qb = -1; };
enum { KSW_XBYTE = 0x10000, KSW_XSTOP = 0x20000, KSW_XSUBO = 0x40000, KSW_XSTART = 0x80000 };

// Lane types: the byte kernel works on 16 unsigned bytes with saturating +/- (paddusb / psubusb), the word kernel on 8 signed
// words; both are one 16-byte compiler vector, so the code below maps one to one onto SSE2 (or NEON) registers.
template <int P> struct KswLane;
template <> struct KswLane<16> { typedef uint8_t T; };
template <> struct KswLane<8> { typedef int16_t T; };

template <int P> struct KswProfile {                            // ksw_qinit, ksw.cpp:62-109
    typedef typename KswLane<P>::T T;
    typedef T V __attribute__((vector_size(16)));
    int qlen, slen, shift, max;
    V *qp;                                                      // [m = 5][slen] vectors, lane k of segment i = query position i + k*slen
    KswProfile(int qlen_, const uint8_t *query, const int8_t *mat, std::vector<V> &store) : qlen(qlen_) {
        slen = (qlen + P - 1) / P;
        int lo = 127, hi = 0;
        for (int a = 0; a < 25; ++a) { if (mat[a] < lo) lo = mat[a]; if (mat[a] > hi) hi = mat[a]; }
        max = hi;
        shift = (256 - (lo & 0xff)) & 0xff;                     // uint8_t arithmetic of the reference
        store.resize((size_t)5 * slen);
        qp = store.data();
        const int add = P == 16 ? shift : 0;
        for (int a = 0; a < 5; ++a) {
            const int8_t *ma = mat + a * 5;
            T row[5];
            for (int c = 0; c < 5; ++c) row[c] = (T)(ma[c] + add);
            for (int i = 0; i < slen; ++i) {
                V v;
                for (int k = 0; k < P; ++k) { const int pos = i + k * slen; v[k] = pos >= qlen ? (T)add : row[query[pos]]; }
                qp[(size_t)a * slen + i] = v;
            }
        }
    }
};

template <int P> struct KswScratch {                            // per-thread buffers, reused from call to call
    typedef typename KswProfile<P>::V V;
    std::vector<V> prof, rows;
    std::vector<uint64_t> b;
};

template <int P>
KswResult ksw_striped(const KswProfile<P> &q, KswScratch<P> &ws, int tlen, const uint8_t *target, int o_del, int e_del, int o_ins, int e_ins, int xtra) {
    constexpr bool U8 = P == 16;
    typedef typename KswProfile<P>::T T;
    typedef typename KswProfile<P>::V V;
    const int slen = q.slen;
    auto splat = [](int x) { V v; for (int k = 0; k < P; ++k) v[k] = (T)x; return v; };
    const V zero = splat(0), oe_del = splat(o_del + e_del), oe_ins = splat(o_ins + e_ins), ed = splat(e_del), ei = splat(e_ins),
            shift = splat(q.shift);
    auto vmax = [](V a, V b) { return __builtin_elementwise_max(a, b); };
    auto ssub = [&](V a, V b) {                                                  // subs_epu8 / subs_epu16 (word lanes are never negative here)
        if constexpr (U8) return __builtin_elementwise_sub_sat(a, b); else return __builtin_elementwise_max(a - b, zero);
    };
    auto up1 = [&](V v) {                                                        // _mm_slli_si128 by one lane
        if constexpr (U8) return __builtin_shufflevector(zero, v, 0, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30);
        else return __builtin_shufflevector(zero, v, 0, 8, 9, 10, 11, 12, 13, 14);
    };
    const int minsc = (xtra & KSW_XSUBO) ? xtra & 0xffff : 0x10000, endsc = (xtra & KSW_XSTOP) ? xtra & 0xffff : 0x10000;
    ws.rows.assign((size_t)4 * slen, zero);
    V *H0 = ws.rows.data(), *H1 = H0 + slen, *E = H1 + slen, *Hmax = E + slen;
    std::vector<uint64_t> &b = ws.b;
    b.clear();
    int te = -1, gmax = 0;
    KswResult r;
    for (int i = 0; i < tlen; ++i) {
        const V *S = q.qp + (size_t)target[i] * slen;
        V h = up1(H0[slen - 1]), f = zero, mx = zero;
        for (int j = 0; j < slen; ++j) {
            if constexpr (U8) h = __builtin_elementwise_sub_sat(__builtin_elementwise_add_sat(h, S[j]), shift);   // adds_epu8, subs_epu8
            else h = __builtin_elementwise_add_sat(h, S[j]);                     // adds_epi16
            V e = E[j];
            h = vmax(vmax(h, e), f);
            mx = vmax(mx, h);
            H1[j] = h;
            E[j] = vmax(ssub(e, ed), ssub(h, oe_del));
            f = vmax(ssub(f, ei), ssub(h, oe_ins));
            h = H0[j];
        }
        bool done = false;                                      // the lazy-F pass (16 rounds at most, as in both kernels)
        for (int k16 = 0; k16 < 16 && !done; ++k16) {
            f = up1(f);
            for (int j = 0; j < slen; ++j) {
                const V hv = vmax(H1[j], f);
                H1[j] = hv;
                f = ssub(f, ei);
                // stop once no lane of f beats what H would start: f <= H - oe_ins everywhere  <=>  subs(f, H - oe_ins) == 0
                const V over = ssub(f, ssub(hv, oe_ins));
                if (!__builtin_reduce_or(over)) { done = true; break; }
            }
        }
        const int imax = __builtin_reduce_max(mx);
        if (imax >= minsc) {
            if (b.empty() || (int32_t)b.back() + 1 != i) b.push_back((uint64_t)imax << 32 | (uint32_t)i);
            else if ((int)(b.back() >> 32) < imax) b.back() = (uint64_t)imax << 32 | (uint32_t)i;
        }
        if (imax > gmax) {
            gmax = imax; te = i;
            memcpy(Hmax, H1, (size_t)slen * sizeof(V));
            if (U8 ? (gmax + q.shift >= 255 || gmax >= endsc) : (gmax >= endsc)) break;
        }
        V *t = H1; H1 = H0; H0 = t;
    }
    r.score = U8 ? (gmax + q.shift < 255 ? gmax : 255) : gmax;
    r.te = te;
    if (!U8 || r.score != 255) {
        int mxv = -1;
        for (int k = 0; k < P; ++k)                             // ascending query position within a lane, so ties keep the smallest
            for (int i = 0; i < slen; ++i) {
                const int t = Hmax[i][k], pos = i + k * slen;
                if (t > mxv) { mxv = t; r.qe = pos; }
            }
        if (!b.empty()) {
            const int d = (r.score + q.max - 1) / q.max, low = te - d, high = te + d;
            for (uint64_t x : b) {
                const int e = (int32_t)x;
                if ((e < low || e > high) && (int)(x >> 32) > r.score2) { r.score2 = (int)(x >> 32); r.te2 = e; }
            }
        }
    }
    return r;
}

template <int P>
KswResult ksw_pass(int qlen, const uint8_t *query, int tlen, const uint8_t *target, const int8_t *mat, int o_del, int e_del, int o_ins,
                   int e_ins, int xtra) {
    static thread_local KswScratch<P> ws;
    const KswProfile<P> prof(qlen, query, mat, ws.prof);
    return ksw_striped<P>(prof, ws, tlen, target, o_del, e_del, o_ins, e_ins, xtra);
}

// ksw_align2, ksw.cpp:340-381 (query and target are private copies here; the reference reverses them in place and back)
KswResult ksw_align2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, const int8_t *mat, int o_del, int e_del, int o_ins,
                     int e_ins, int xtra) {
    const bool byte = (xtra & KSW_XBYTE) != 0;
    KswResult r = byte ? ksw_pass<16>(qlen, query, tlen, target, mat, o_del, e_del, o_ins, e_ins, xtra)
                       : ksw_pass<8>(qlen, query, tlen, target, mat, o_del, e_del, o_ins, e_ins, xtra);
    if ((xtra & KSW_XSTART) == 0 || ((xtra & KSW_XSUBO) && r.score < (xtra & 0xffff))) return r;
    std::vector<uint8_t> rq(query, query + r.qe + 1), rt(target, target + tlen);
    std::reverse(rq.begin(), rq.end());
    std::reverse(rt.begin(), rt.begin() + r.te + 1);            // only the first te+1 bases are reversed; the rest of the target stays
    const int x2 = KSW_XSTOP | r.score;
    const KswResult rr = byte ? ksw_pass<16>(r.qe + 1, rq.data(), tlen, rt.data(), mat, o_del, e_del, o_ins, e_ins, x2)
                              : ksw_pass<8>(r.qe + 1, rq.data(), tlen, rt.data(), mat, o_del, e_del, o_ins, e_ins, x2);
    if (r.score == rr.score) { r.tb = r.te - rr.te; r.qb = r.qe - rr.qe; }
    return r;
}

struct PeStat { int low = 0, high = 0, failed = 0; double avg = 0, std = 0; };   // mem_pestat_t

// A read's hits (mem_alnreg_v): a slice of the chunk's flat store with room for what mate rescue may add (one hit per planned alignment);
// a list that outgrows its slice all the same moves to the spill arena of the thread that grows it.  No allocation per read, nothing to
// release one by one.
thread_local Arena t_spill;
thread_local uint64_t t_spill_epoch = 0;
std::atomic<uint64_t> g_call_epoch{0};
inline void spill_sync(uint64_t epoch) { if (t_spill_epoch != epoch) { t_spill.reset(); t_spill_epoch = epoch; } }   // (spilled lists of the call before are gone)
struct HitList {
    bm2_alnreg_t *p = nullptr; int n = 0, cap = 0;
    size_t size() const { return (size_t)n; }
    bool empty() const { return n == 0; }
    bm2_alnreg_t &operator[](size_t i) { return p[i]; }
    const bm2_alnreg_t &operator[](size_t i) const { return p[i]; }
    bm2_alnreg_t *data() { return p; }
    const bm2_alnreg_t *data() const { return p; }
    void room(int want) {
        if (want <= cap) return;
        int nc = cap * 2 > want ? cap * 2 : want; if (nc < 8) nc = 8;
        bm2_alnreg_t *q = (bm2_alnreg_t *)t_spill.get(sizeof(bm2_alnreg_t) * (size_t)nc);
        if (n) memcpy(q, p, sizeof(bm2_alnreg_t) * (size_t)n);
        p = q; cap = nc;
    }
    void insert_at(size_t i, const bm2_alnreg_t &b) {
        room(n + 1);
        if ((size_t)n > i) memmove(p + i + 1, p + i, sizeof(bm2_alnreg_t) * ((size_t)n - i));
        p[i] = b; ++n;
    }
    void assign(const bm2_alnreg_t *first, const bm2_alnreg_t *last) { const int k = (int)(last - first); room(k); if (k) memmove(p, first, sizeof(bm2_alnreg_t) * (size_t)k); n = k; }
};
inline HitList view_of(const bm2_alnreg_t *alnregs, const int64_t *reg_off, int i) {       // the input's list of read i, for reading only
    HitList v; v.p = const_cast<bm2_alnreg_t *>(alnregs) + reg_off[i]; v.n = v.cap = (int)(reg_off[i + 1] - reg_off[i]); return v;
}

int infer_dir(int64_t l_pac, int64_t b1, int64_t b2, int64_t *dist) {            // bwamem_pair.cpp:58-65
    const int r1 = (b1 >= l_pac), r2 = (b2 >= l_pac);
    const int64_t p2 = r1 == r2 ? b2 : (l_pac << 1) - 1 - b2;
    *dist = p2 > b1 ? p2 - b1 : b1 - p2;
    return (r1 == r2 ? 0 : 1) ^ (p2 > b1 ? 0 : 3);
}

int cal_sub(const bm2_opt *opt, const HitList &r) {            // bwamem_pair.cpp:67-79
    size_t j;
    for (j = 1; j < r.size(); ++j) {
        const int b_max = r[j].qb > r[0].qb ? r[j].qb : r[0].qb, e_min = r[j].qe < r[0].qe ? r[j].qe : r[0].qe;
        if (e_min > b_max) {
            const int min_l = r[j].qe - r[j].qb < r[0].qe - r[0].qb ? r[j].qe - r[j].qb : r[0].qe - r[0].qb;
            if (e_min - b_max >= min_l * opt->mask_level) break;
        }
    }
    return j < r.size() ? r[j].score : opt->min_seed_len * opt->a;
}

// mem_pestat, bwamem_pair.cpp:81-148 (without the log lines)
// The insert-size model of a chunk (mem_pestat, bwamem_pair.cpp:81-148).  The reference collects the insert sizes of the pairs whose ends
// are both unique into four arrays, sorts them and reads quartiles, a trimmed mean and a trimmed deviation off the sorted arrays.  An
// insert size is an integer in [1, max_ins], so here the chunk's pairs are counted into four HISTOGRAMS on all threads (integer adds:
// any order), and everything else is read off the counts: a quartile is a rank in the cumulative counts; the mean sums integers (exact
// in a double whatever the order); the deviation adds (v - mean)^2 once per pair in ascending order of v, which is the order -- and
// therefore the rounding -- of the reference's loop over its sorted array.
void pestat(const bm2_opt *opt, const bm2_sam_opt *so, int64_t l_pac, int n_reads, const bm2_alnreg_t *alnregs, const int64_t *reg_off, PeStat pes[4]) {
    const int n_pairs = n_reads >> 1;
    const int64_t top = so->max_ins > 0 ? so->max_ins : 0;       // bins 1 .. max_ins
    for (int d = 0; d < 4; ++d) pes[d] = PeStat();
    int nt = so->n_threads > 0 ? so->n_threads : bm2_effective_cpus();
    if (nt > n_pairs / 8192 + 1) nt = n_pairs / 8192 + 1;
    if (nt < 1) nt = 1;
    const size_t bins = (size_t)top + 1;
    std::vector<uint32_t> hist((size_t)nt * 4 * bins, 0);
    std::atomic<int> nx(0), tid(0);
    run_threads(nt, [&]() {
        uint32_t *mine = hist.data() + (size_t)tid.fetch_add(1) * 4 * bins;
        for (int lo; (lo = nx.fetch_add(8192)) < n_pairs;)
            for (int i = lo; i < n_pairs && i < lo + 8192; ++i) {
                const HitList r0 = view_of(alnregs, reg_off, i << 1 | 0), r1 = view_of(alnregs, reg_off, i << 1 | 1);
                if (r0.empty() || r1.empty()) continue;
                if (cal_sub(opt, r0) > 0.8 * r0[0].score) continue;
                if (cal_sub(opt, r1) > 0.8 * r1[0].score) continue;
                if (r0[0].rid != r1[0].rid) continue;
                int64_t is;
                const int dir = infer_dir(l_pac, r0[0].rb, r1[0].rb, &is);
                if (is && is <= top) ++mine[(size_t)dir * bins + (size_t)is];
            }
    });
    uint64_t count[4] = { 0, 0, 0, 0 };
    for (int d = 0; d < 4; ++d) {
        uint32_t *h = hist.data() + (size_t)d * bins;            // thread 0's histogram becomes the total
        for (int t = 1; t < nt; ++t) { const uint32_t *o = hist.data() + ((size_t)t * 4 + d) * bins; for (size_t v = 0; v < bins; ++v) h[v] += o[v]; }
        uint64_t tot = 0;
        for (size_t v = 0; v < bins; ++v) tot += h[v];
        count[d] = tot;
        PeStat *r = &pes[d];
        if (tot < 10) { r->failed = 1; continue; }
        auto at_rank = [&](uint64_t k) {                         // the k-th smallest insert size (k counted from 0)
            uint64_t c = 0;
            for (size_t v = 0; v < bins; ++v) { c += h[v]; if (c > k) return (int)v; }
            return (int)top;
        };
        const int p25 = at_rank((uint64_t)(int)(.25 * tot + .499)), p75 = at_rank((uint64_t)(int)(.75 * tot + .499));
        r->low = (int)(p25 - 2.0 * (p75 - p25) + .499);
        if (r->low < 1) r->low = 1;
        r->high = (int)(p75 + 2.0 * (p75 - p25) + .499);
        const size_t v_lo = (size_t)r->low, v_hi = r->high < 0 ? 0 : std::min((size_t)r->high, bins - 1);
        uint64_t x = 0; double sum = 0;
        for (size_t v = v_lo; v <= v_hi && r->high >= r->low; ++v) { x += h[v]; sum += (double)v * h[v]; }
        r->avg = sum / x;
        r->std = 0;
        for (size_t v = v_lo; v <= v_hi && r->high >= r->low; ++v) { const double dv = ((double)v - r->avg) * ((double)v - r->avg); for (uint32_t c = 0; c < h[v]; ++c) r->std += dv; }
        r->std = sqrt(r->std / x);
        r->low = (int)(p25 - 3.0 * (p75 - p25) + .499);
        r->high = (int)(p75 + 3.0 * (p75 - p25) + .499);
        if (r->low > r->avg - 4.0 * r->std) r->low = (int)(r->avg - 4.0 * r->std + .499);
        if (r->high < r->avg + 4.0 * r->std) r->high = (int)(r->avg + 4.0 * r->std + .499);
        if (r->low < 1) r->low = 1;
    }
    uint64_t mx = 0;
    for (int d = 0; d < 4; ++d) mx = mx > count[d] ? mx : count[d];
    for (int d = 0; d < 4; ++d) if (pes[d].failed == 0 && count[d] < mx * 0.05) pes[d].failed = 1;
}

// bns_fetch_seq (bntseq.cpp:453-482) on the unpacked reference: [*beg, *end) clamped to the contig (strand-aware) that holds mid
bool fetch_range(const Ref &R, const int32_t *ann_len, int64_t *beg, int64_t mid, int64_t *end, int *rid) {
    if (*end < *beg) { const int64_t t = *beg; *beg = *end; *end = t; }
    int is_rev;
    *rid = R.pos2rid(R.depos(mid, &is_rev));
    if (*rid < 0) return false;
    int64_t far_beg = R.off[*rid], far_end = far_beg + ann_len[*rid];
    if (is_rev) { const int64_t t = far_beg; far_beg = (R.l_pac << 1) - far_end; far_end = (R.l_pac << 1) - t; }
    *beg = *beg > far_beg ? *beg : far_beg;
    *end = *end < far_end ? *end : far_end;
    return true;                                                 // the bases are R.ref_string[*beg, *end): .0123 = what bns_get_seq unpacks
}

// ---- mate rescue (mem_matesw, bwamem_pair.cpp:150-283, MATE_SORT == 0), split so that the alignments of a whole chunk can run as
// one batch: rescue_window = where direction r of anchor `a` would look for the mate; rescue_align = the local SW there;
// rescue_apply = what the result does to the mate's hit list.  matesw() strings them together per anchor as the reference does.
struct RescueTask {                 // one (anchor, direction) alignment of a pair, enumerated before the pair is processed
    int32_t pair, j;                // j = rank of the anchor in b[end] (mem_sam_pe's candidate list)
    uint8_t end, r;                 // end = which read the ANCHOR belongs to (the mate is !end); r = direction 0..3
    int64_t rb, re;                 // the window, already clamped to the contig
    KswResult res;
};
struct RescueStats { std::atomic<long long> planned{0}, used{0}, missed{0}; };

bool rescue_window(const bm2_opt *opt, const Ref &R, const int32_t *ann_len, const PeStat pes[4], const bm2_alnreg_t *a, int l_ms, int r,
                   int64_t *rb_, int64_t *re_) {
    const int64_t l_pac = R.l_pac;
    const int is_rev = (r >> 1 != (r & 1)), is_larger = !(r >> 1);
    int64_t rb, re;
    if (!is_rev) {
        rb = is_larger ? a->rb + pes[r].low : a->rb - pes[r].high;
        re = (is_larger ? a->rb + pes[r].high : a->rb - pes[r].low) + l_ms;
    } else {
        rb = (is_larger ? a->rb + pes[r].low : a->rb - pes[r].high) - l_ms;
        re = is_larger ? a->rb + pes[r].high : a->rb - pes[r].low;
    }
    if (rb < 0) rb = 0;
    if (re > l_pac << 1) re = l_pac << 1;
    if (rb >= re) return false;
    int rid = -1;
    if (!fetch_range(R, ann_len, &rb, (rb + re) >> 1, &re, &rid)) return false;
    if (a->rid != rid || re - rb < opt->min_seed_len) return false;
    *rb_ = rb; *re_ = re;
    return true;
}

int rescue_xtra(const bm2_opt *opt, int l_ms) { return KSW_XSUBO | KSW_XSTART | (l_ms * opt->a < 250 ? KSW_XBYTE : 0) | (opt->min_seed_len * opt->a); }

void rescue_query(int l_ms, const uint8_t *ms, int r, std::vector<uint8_t> &q) {       // the mate as direction r reads it
    const int is_rev = (r >> 1 != (r & 1));
    q.resize((size_t)l_ms);
    if (!is_rev) memcpy(q.data(), ms, (size_t)l_ms);
    else for (int i = 0; i < l_ms; ++i) q[(size_t)(l_ms - 1 - i)] = ms[i] < 4 ? 3 - ms[i] : 4;
}

KswResult rescue_align(const bm2_opt *opt, const Ref &R, int l_ms, const uint8_t *ms, int r, int64_t rb, int64_t re) {
    static thread_local std::vector<uint8_t> q;
    rescue_query(l_ms, ms, r, q);
    return ksw_align2(l_ms, q.data(), (int)(re - rb), R.ref_string + rb, opt->mat, opt->o_del, opt->e_del, opt->o_ins, opt->e_ins, rescue_xtra(opt, l_ms));
}

// A rescued alignment becomes a hit of the mate (mem_matesw, bwamem_pair.cpp:205-222): the aligner saw the mate as direction r reads it and
// the window [rb, ...) of the doubled reference, so both spans are mirrored back when that view was the reverse strand's.
struct Span { int64_t b, e; };
static inline Span mirrored(Span s, int64_t len, bool flip) { return flip ? Span{ len - s.e, len - s.b } : s; }
void rescue_apply(const bm2_opt *opt, const Ref &R, const bm2_alnreg_t *a, int l_ms, int r, int64_t rb, const KswResult &aln,
                  HitList &ma) {
    if (aln.score < opt->min_seed_len || aln.qb < 0) return;
    const bool flip = (r >> 1) != (r & 1);
    const Span on_read = mirrored(Span{ aln.qb, (int64_t)aln.qe + 1 }, l_ms, flip);
    const Span on_ref = mirrored(Span{ rb + aln.tb, rb + aln.te + 1 }, R.l_pac << 1, flip);
    bm2_alnreg_t b; memset(&b, 0, sizeof b);
    b.rid = a->rid; b.is_alt = a->is_alt;
    b.qb = (int)on_read.b; b.qe = (int)on_read.e; b.rb = on_ref.b; b.re = on_ref.e;
    b.score = aln.score; b.csub = aln.score2; b.secondary = -1;
    b.seedcov = (int)(std::min<int64_t>(on_ref.e - on_ref.b, on_read.e - on_read.b) >> 1);
    size_t at = 0;                                               // in front of the first hit that scores less
    while (at < ma.size() && !(ma[at].score < b.score)) ++at;
    ma.insert_at(at, b);
}

// Which of the four directions of an anchor need no rescue: those whose insert-size model failed, and those in which the mate already has
// a hit at a plausible distance (bwamem_pair.cpp:165-174)
void rescue_skip(const Ref &R, const PeStat pes[4], const bm2_alnreg_t *a, const HitList &ma, int skip[4]) {
    unsigned served = 0;
    for (size_t i = 0; i < ma.size(); ++i) {
        int64_t dist;
        const int r = infer_dir(R.l_pac, a->rb, ma[i].rb, &dist);
        served |= (unsigned)(dist >= pes[r].low && dist <= pes[r].high) << r;
    }
    for (int r = 0; r < 4; ++r) skip[r] = pes[r].failed || (served >> r & 1) ? 1 : 0;
}

// pre[0, n_pre): the alignments planned for this anchor (end, j) and already computed, or none (then they are computed here)
// The hit list of the mate after a rescue round: what mem_sort_dedup_patch (bwamem.cpp:292-353) leaves when it is called without a
// query (bwamem_pair.cpp:274) -- then mem_patch_reg answers 0 at once (bwamem.cpp:181) and nothing is ever merged, so the
// procedure reduces to three sweeps over an order array: (1) by reference end, klib's tie permutation: of two hits of one contig
// that overlap by more than mask_level_redun on the reference AND on the read the lower-scoring one goes (the later one on a tie),
// looking back only while the earlier hit ends within max_chain_gap before the later one begins; (2) survivors by (score desc, rb,
// qb); (3) of hits equal in all three the first stays.  The full procedure, with merging, is the device's (finish.hip).
void dedup_rescued(const bm2_opt *opt, HitList &hits) {
    const int n = (int)hits.size();
    if (n <= 1) return;
    static thread_local std::vector<int> ord, keep;              // (scratch kept per thread: this runs after every rescued direction)
    static thread_local std::vector<char> gone;
    static thread_local std::vector<bm2_alnreg_t> out;
    ord.resize((size_t)n);
    for (int i = 0; i < n; ++i) ord[(size_t)i] = i;
    k_introsort((size_t)n, ord.data(), [&](int x, int y) { return hits[(size_t)x].re < hits[(size_t)y].re; });
    gone.assign((size_t)n, 0);
    for (int i = 0; i < n; ++i) hits[(size_t)i].n_comp = 1;
    auto span = [](int64_t b, int64_t e) { return e - b; };
    for (int i = 1; i < n; ++i) {
        const bm2_alnreg_t &p = hits[(size_t)ord[(size_t)i]];
        for (int j = i - 1; j >= 0 && !gone[(size_t)ord[(size_t)i]]; --j) {
            const bm2_alnreg_t &q = hits[(size_t)ord[(size_t)j]];
            if (q.rid != p.rid || p.rb >= q.re + opt->max_chain_gap) break;
            if (gone[(size_t)ord[(size_t)j]]) continue;
            const int64_t on_ref = q.re - p.rb, on_read = q.qb < p.qb ? q.qe - p.qb : p.qe - q.qb;
            const int64_t min_ref = std::min(span(q.rb, q.re), span(p.rb, p.re)), min_read = std::min<int64_t>(span(q.qb, q.qe), span(p.qb, p.qe));
            if (on_ref > opt->mask_level_redun * min_ref && on_read > opt->mask_level_redun * min_read)
                gone[(size_t)ord[(size_t)(p.score < q.score ? i : j)]] = 1;
        }
    }
    keep.clear();
    for (int i = 0; i < n; ++i) if (!gone[(size_t)ord[(size_t)i]]) keep.push_back(ord[(size_t)i]);
    k_introsort(keep.size(), keep.data(), [&](int x, int y) {
        const bm2_alnreg_t &a = hits[(size_t)x], &b = hits[(size_t)y];
        return a.score > b.score || (a.score == b.score && (a.rb < b.rb || (a.rb == b.rb && a.qb < b.qb)));
    });
    out.clear();
    for (size_t k = 0; k < keep.size(); ++k) {
        const bm2_alnreg_t &a = hits[(size_t)keep[k]];
        if (k > 0) { const bm2_alnreg_t &b = hits[(size_t)keep[k - 1]]; if (a.score == b.score && a.rb == b.rb && a.qb == b.qb) continue; }
        out.push_back(a);
    }
    hits.assign(out.data(), out.data() + out.size());
}

int matesw(const bm2_opt *opt, const bm2_sam_opt *so, const Ref &R, const int32_t *ann_len, const PeStat pes[4], const bm2_alnreg_t *a,
           int l_ms, const uint8_t *ms, HitList &ma, const RescueTask *pre, int n_pre, RescueStats *st) {
    int skip[4], n = 0;
    rescue_skip(R, pes, a, ma, skip);
    if (skip[0] + skip[1] + skip[2] + skip[3] == 4) return 0;
    for (int r = 0; r < 4; ++r) {
        if (skip[r]) continue;
        int64_t rb, re;
        if (rescue_window(opt, R, ann_len, pes, a, l_ms, r, &rb, &re)) {
            const RescueTask *hit = nullptr;
            for (int t = 0; t < n_pre; ++t) if (pre[t].r == r) { hit = &pre[t]; break; }
            if (hit) { if (st) ++t_rs_used; rescue_apply(opt, R, a, l_ms, r, rb, hit->res, ma); }
            else {
                if (st && pre) ++t_rs_missed;                    // planned before the mate's list changed: rare, computed in place
                rescue_apply(opt, R, a, l_ms, r, rb, rescue_align(opt, R, l_ms, ms, r, rb, re), ma);
            }
            ++n;
        }
        if (n) dedup_rescued(opt, ma);
    }
    return n;
}

// The candidate anchors of mem_sam_pe (bwamem_pair.cpp:371-376): hits within pen_unpaired of the best, at most max_matesw used
void rescue_anchors(const bm2_sam_opt *so, const HitList &a, std::vector<bm2_alnreg_t> &b) {
    b.clear();
    for (size_t j = 0; j < a.size(); ++j) if (a[j].score >= a[0].score - so->pen_unpaired) b.push_back(a[j]);
}

// All alignments mate rescue may ask for in one pair, judged on the hit lists as they are BEFORE any rescue.  While the pair is
// processed the mate's list only grows (each insertion is followed by a de-duplication that keeps the better of two overlapping
// hits), so a direction skipped here stays skipped and what is planned is, but for freak cases, a superset of what is used.
void rescue_plan(const bm2_opt *opt, const bm2_sam_opt *so, const Ref &R, const int32_t *ann_len, const PeStat pes[4], int pair,
                 const int l_seq[2], const HitList a[2], std::vector<RescueTask> &out) {
    static thread_local std::vector<bm2_alnreg_t> b;
    for (int i = 0; i < 2; ++i) {
        rescue_anchors(so, a[i], b);
        for (size_t j = 0; j < b.size() && (int)j < so->max_matesw; ++j) {
            int skip[4];
            rescue_skip(R, pes, &b[j], a[!i], skip);
            for (int r = 0; r < 4; ++r) {
                if (skip[r]) continue;
                RescueTask t;
                if (!rescue_window(opt, R, ann_len, pes, &b[j], l_seq[!i], r, &t.rb, &t.re)) continue;
                t.pair = pair; t.end = (uint8_t)i; t.r = (uint8_t)r; t.j = (int32_t)j;
                out.push_back(t);
            }
        }
    }
}

// The best pairing of the two reads' primary hits: what mem_pair (bwamem_pair.cpp:285-346) answers, computed this file's way.
// A pairing is a hit k of one read and a hit i of the other whose forward-strand coordinates on one contig lie `dist` apart with k first,
// low <= dist <= high of the orientation (strand of k, strand of i) in the chunk's insert-size model; it scores
// score_k + score_i + 0.721 ln(2 erfc(|dist - avg| / std / sqrt 2)) * a (rounded, floored at 0).  The answer is the best pairing, the score
// of the runner-up and how many others come within one mismatch / gap of that.
// Form: the ends are ranked in reference order once; each (strand, read) class keeps its ends in that order, so the partners of an end are a
// WINDOW of the class of the other read on each strand, and as the ends of a class are visited in order the window only moves forward (two
// cursors per class and partner strand) -- no back-scan over the mixed list, no second sort: the best and the runner-up are picked in one
// pass over the pairings' scores.  What the output can see of the reference's procedure is kept to the letter: the rank of an end in the
// order (position, score, index in its read's list, strand, read) -- the tie-breaking hash of a pairing is taken of the two ranks and the
// pair's number --, the floating-point expression of the score, and the order (score, hash, ranks) among pairings.
struct PairEnd { uint64_t pos; int32_t score, idx; uint8_t strand, read; };
int pair_hits(const bm2_opt *opt, const Ref &R, const PeStat pes[4], const HitList a[2], int id, int *sub, int *n_sub,
              int z[2], const int n_pri[2]) {
    AVec<PairEnd> ends;                                          // (in the thread's scratch arena: the caller resets it per pair)
    const int64_t l_pac = R.l_pac;
    ends.reserve(n_pri[0] + n_pri[1]);
    for (int rd = 0; rd < 2; ++rd)
        for (int i = 0; i < n_pri[rd]; ++i) {
            const bm2_alnreg_t &h = a[rd][i];
            PairEnd e;
            e.strand = h.rb >= l_pac; e.read = (uint8_t)rd; e.idx = i; e.score = h.score;
            const uint64_t fwd = (uint64_t)(e.strand ? (l_pac << 1) - 1 - h.rb : h.rb);
            e.pos = (uint64_t)h.rid << 32 | (fwd - (uint64_t)R.off[h.rid]);          // contig in the high word: ends of two contigs are never `high` apart
            ends.push_back(e);
        }
    std::sort(ends.p, ends.p + ends.n, [](const PairEnd &p, const PairEnd &q) {
        if (p.pos != q.pos) return p.pos < q.pos;
        if (p.score != q.score) return (uint32_t)p.score < (uint32_t)q.score;
        if (p.idx != q.idx) return p.idx < q.idx;
        if (p.strand != q.strand) return p.strand < q.strand;
        return p.read < q.read;
    });
    AVec<int> cls[4];                                            // ranks of the ends of class strand << 1 | read, ascending
    for (int k = 0; k < ends.n; ++k) cls[ends[k].strand << 1 | ends[k].read].push_back(k);
    int gap = opt->a + opt->b;                                   // "within one mismatch or gap" of the runner-up
    if (gap < opt->o_del + opt->e_del) gap = opt->o_del + opt->e_del;
    if (gap < opt->o_ins + opt->e_ins) gap = opt->o_ins + opt->e_ins;
    AVec<int> scores;                                            // one per pairing
    uint64_t best_key = 0, best_ranks = 0; int best_at = -1;
    for (int c = 0; c < 4; ++c) {                                // the later end i of a pairing, class by class
        const int strand_i = c >> 1, read_i = c & 1;
        for (int r = 0; r < 2; ++r) {                            // its partner's strand
            const PeStat &m = pes[r << 1 | strand_i];
            if (m.failed) continue;
            const AVec<int> &part = cls[r << 1 | (read_i ^ 1)];
            int lo = 0, hi = 0;                                  // partners [lo, hi): high >= pos_i - pos_k >= low
            for (int t = 0; t < cls[c].n; ++t) {
                const int i = cls[c][t];
                const uint64_t pi = ends[i].pos;
                while (hi < part.n && part[hi] < i && (int64_t)(pi - ends[part[hi]].pos) >= (int64_t)m.low) ++hi;      // (an earlier rank: its position is not beyond pi)
                while (lo < hi && (int64_t)(pi - ends[part[lo]].pos) > (int64_t)m.high) ++lo;
                for (int w = lo; w < hi; ++w) {
                    const int k = part[w];
                    const int64_t dist = (int64_t)(pi - ends[k].pos);
                    const double ns = (dist - m.avg) / m.std;
                    int q = (int)((uint64_t)(uint32_t)ends[i].score + (uint64_t)(uint32_t)ends[k].score + .721 * log(2. * erfc(fabs(ns) * M_SQRT1_2)) * opt->a + .499);
                    if (q < 0) q = 0;
                    const uint64_t ranks = (uint64_t)k << 32 | (uint64_t)i;
                    const uint64_t key = (uint64_t)q << 32 | (hash_64(ranks ^ (uint64_t)(int64_t)(id << 8)) & 0xffffffffU);
                    if (best_at < 0 || key > best_key || (key == best_key && ranks > best_ranks)) { best_key = key; best_ranks = ranks; best_at = scores.n; }
                    scores.push_back(q);
                }
            }
        }
    }
    *sub = 0; *n_sub = 0;
    if (best_at < 0) return 0;
    const PairEnd &ek = ends[(int)(best_ranks >> 32)], &ei = ends[(int)(best_ranks & 0xffffffffU)];
    z[ei.read] = ei.idx; z[ek.read] = ek.idx;
    if (scores.n > 1) {
        int second = -1;
        for (int t = 0; t < scores.n; ++t) if (t != best_at && scores[t] > second) second = scores[t];
        *sub = second;
        for (int t = 0; t < scores.n; ++t) if (t != best_at && second - scores[t] <= gap) ++*n_sub;
    }
    return (int)(best_key >> 32);
}

int raw_mapq(int diff, int a) { return (int)(6.02 * diff / a + .499); }

struct ReadIO { const char *name, *comment, *qual; int l_seq; const uint8_t *seq; };

// mem_sam_pe (bwamem_pair.cpp:353-551) in two halves.  pe_decide is everything that CHANGES the pair's hit lists: mate rescue (with the
// batch's results at hand), primary marking, pairing, the MAPQs of a proper pair.  pe_emit is what follows -- XA strings, CIGARs, text --
// and only reads the lists, so a CIGAR session can run it twice (once to note the hits it asks for, once to print) on the same state.
struct PairPlan { int z[2] = { 0, 0 }, n_pri[2] = { 0, 0 }, q_se[2] = { 0, 0 }, extra_flag = 1; bool paired = false; };

void pe_decide(const bm2_opt *opt, const bm2_sam_opt *so, const Ref &R, const int32_t *ann_len, const PeStat pes[4], uint64_t id,
               const ReadIO s[2], HitList a[2], const RescueTask *pre, int n_pre, RescueStats *st, PairPlan &P) {
    int o, subo, n_sub;
    P = PairPlan();
    int *z = P.z, *n_pri = P.n_pri, *q_se = P.q_se;
    if (!(so->flag & F_NO_RESCUE)) {
        static thread_local std::vector<bm2_alnreg_t> b[2];
        for (int i = 0; i < 2; ++i) rescue_anchors(so, a[i], b[i]);
        int t = 0;                                               // pre[] is ordered by (end, j, r), as rescue_plan emits it
        for (int i = 0; i < 2; ++i)
            for (size_t j = 0; j < b[i].size() && (int)j < so->max_matesw; ++j) {
                while (t < n_pre && (pre[t].end < i || (pre[t].end == i && pre[t].j < (int)j))) ++t;
                int t1 = t;
                while (t1 < n_pre && pre[t1].end == i && pre[t1].j == (int)j) ++t1;
                matesw(opt, so, R, ann_len, pes, &b[i][j], s[!i].l_seq, s[!i].seq, a[!i], pre ? pre + t : nullptr, t1 - t, st);
            }
    }
    n_pri[0] = mark_primary_se(opt, (int)a[0].size(), a[0].data(), (int64_t)(id << 1 | 0));
    n_pri[1] = mark_primary_se(opt, (int)a[1].size(), a[1].data(), (int64_t)(id << 1 | 1));
    if (so->flag & F_PRIMARY5) { reorder_primary5(so->T, (int)a[0].size(), a[0].data()); reorder_primary5(so->T, (int)a[1].size(), a[1].data()); }
    if ((so->flag & F_NOPAIRING) || !n_pri[0] || !n_pri[1] || (o = pair_hits(opt, R, pes, a, (int)id, &subo, &n_sub, z, n_pri)) <= 0) return;
    for (int i = 0; i < 2; ++i)                                  // a second primary hit above the threshold: the reads go out one by one
        for (int j = 1; j < n_pri[i]; ++j) if (a[i][j].secondary < 0 && a[i][j].score >= so->T) return;
    P.paired = true;
    int score_un = a[0][0].score + a[1][0].score - so->pen_unpaired;
    subo = subo > score_un ? subo : score_un;
    int q_pe = raw_mapq(o - subo, opt->a);
    if (n_sub > 0) q_pe -= (int)(4.343 * log(n_sub + 1) + .499);
    if (q_pe < 0) q_pe = 0;
    if (q_pe > 60) q_pe = 60;
    q_pe = (int)(q_pe * (1. - .5 * (a[0][0].frac_rep + a[1][0].frac_rep)) + .499);
    if (o > score_un) {
        bm2_alnreg_t *c[2] = { &a[0][z[0]], &a[1][z[1]] };
        for (int i = 0; i < 2; ++i) {
            if (c[i]->secondary >= 0) { c[i]->sub = a[i][c[i]->secondary].score; c[i]->secondary = -2; }
            q_se[i] = approx_mapq_se(opt, so, c[i]);
        }
        q_se[0] = q_se[0] > q_pe ? q_se[0] : q_pe < q_se[0] + 40 ? q_pe : q_se[0] + 40;
        q_se[1] = q_se[1] > q_pe ? q_se[1] : q_pe < q_se[1] + 40 ? q_pe : q_se[1] + 40;
        P.extra_flag |= 2;
        q_se[0] = q_se[0] < raw_mapq(c[0]->score - c[0]->csub, opt->a) ? q_se[0] : raw_mapq(c[0]->score - c[0]->csub, opt->a);
        q_se[1] = q_se[1] < raw_mapq(c[1]->score - c[1]->csub, opt->a) ? q_se[1] : raw_mapq(c[1]->score - c[1]->csub, opt->a);
    } else {
        z[0] = z[1] = 0;
        q_se[0] = approx_mapq_se(opt, so, &a[0][0]);
        q_se[1] = approx_mapq_se(opt, so, &a[1][0]);
    }
    for (int i = 0; i < 2; ++i) {
        const int k = a[i][z[i]].secondary_all;
        if (k >= 0 && k < n_pri[i]) {                            // switch secondary and primary if both are non-ALT
            for (size_t j = 0; j < a[i].size(); ++j)
                if (a[i][j].secondary_all == k || (int)j == k) a[i][j].secondary_all = z[i];
            a[i][z[i]].secondary_all = -1;
        }
    }
}

bool pe_emit(const bm2_opt *opt, const bm2_sam_opt *so, const Ref &R, const PeStat pes[4], const ReadIO s[2],
             const HitList a[2], const PairPlan &P, Text &out) {
    const int *z = P.z, *n_pri = P.n_pri;
    int extra_flag = P.extra_flag;
    Aln h[2];
    if (P.paired) {
        AVec<AVec<char>> XA[2]; bool any_xa[2] = { false, false };
        if (!(so->flag & F_ALL))
            for (int i = 0; i < 2; ++i)
                if (!gen_alt(opt, so, R, (int)a[i].size(), a[i].data(), s[i].l_seq, s[i].seq, XA[i], any_xa[i])) return false;
        AVec<Aln> aa[2];
        for (int i = 0; i < 2; ++i) {
            if (!reg2aln(opt, so, R, s[i].l_seq, s[i].seq, &a[i][z[i]], h[i])) return false;
            h[i].mapq = P.q_se[i];
            h[i].flag |= 0x40 << i | extra_flag;
            h[i].XA = (any_xa[i] && !XA[i][z[i]].empty()) ? &XA[i][z[i]] : nullptr;
            aa[i].push_back(h[i]);
            if (n_pri[i] < (int)a[i].size()) {                   // the read has ALT hits
                const bm2_alnreg_t *p = &a[i][n_pri[i]];
                if (p->score < so->T || p->secondary >= 0 || !p->is_alt) continue;
                Aln g;
                if (!reg2aln(opt, so, R, s[i].l_seq, s[i].seq, p, g)) return false;
                g.flag |= 0x800 | 0x40 << i | extra_flag;
                g.XA = (any_xa[i] && !XA[i][n_pri[i]].empty()) ? &XA[i][n_pri[i]] : nullptr;
                aa[i].push_back(g);
            }
        }
        for (int i = 0; i < (int)aa[0].size(); ++i) aln2sam(so, R, out, s[0].name, s[0].comment, s[0].qual, s[0].l_seq, s[0].seq, aa[0], i, &h[1]);
        for (int i = 0; i < (int)aa[1].size(); ++i) aln2sam(so, R, out, s[1].name, s[1].comment, s[1].qual, s[1].l_seq, s[1].seq, aa[1], i, &h[0]);
        return true;
    }
    // no_pairing:
    for (int i = 0; i < 2; ++i) {
        int which = -1;
        if (!a[i].empty()) {
            if (a[i][0].score >= so->T) which = 0;
            else if (n_pri[i] < (int)a[i].size() && a[i][n_pri[i]].score >= so->T) which = n_pri[i];
        }
        if (!reg2aln(opt, so, R, s[i].l_seq, s[i].seq, which >= 0 ? &a[i][which] : 0, h[i])) return false;
    }
    if (!(so->flag & F_NOPAIRING) && h[0].rid == h[1].rid && h[0].rid >= 0) {
        int64_t dist;
        const int d = infer_dir(R.l_pac, a[0][0].rb, a[1][0].rb, &dist);
        if (!pes[d].failed && dist >= pes[d].low && dist <= pes[d].high) extra_flag |= 2;
    }
    if (!reg2sam(opt, so, R, out, s[0].name, s[0].comment, s[0].qual, s[0].l_seq, s[0].seq, (int)a[0].size(), a[0].data(), 0x41 | extra_flag, &h[1])) return false;
    if (!reg2sam(opt, so, R, out, s[1].name, s[1].comment, s[1].qual, s[1].l_seq, s[1].seq, (int)a[1].size(), a[1].data(), 0x81 | extra_flag, &h[0])) return false;
    return true;
}


CgStats g_cigar;                    // counters of the last bm2_sam_pe / bm2_sam_se call that ran a CIGAR session (bm2_sam_cigar_stats)

// the hit numbers a dry pass recorded -> the batch (every hit once, in hit order) -> one call of the hook -> memo
int cigar_session_batch(const bm2_opt *opt, const bm2_reads *reads, int64_t enc_bytes, const bm2_alnreg_t *alnregs, const int64_t *reg_off,
                        std::vector<std::vector<int32_t>> &recs, bm2h_cigar_batch_fn cfn, void *cuser, CgMemo &M, bool by_pad) {
    TailProf prof("cigar_session");
    const int n_reads = reads->n_reads;
    const int64_t n_hits = reg_off[n_reads];
    M.task_of.assign((size_t)n_hits, -1);
    for (auto &v : recs) for (int32_t id : v) if (id >= 0 && id < n_hits) M.task_of[(size_t)id] = 0;
    int32_t n = 0;
    for (int64_t i = 0; i < n_hits; ++i) if (M.task_of[(size_t)i] == 0) M.task_of[(size_t)i] = n++;
    g_cigar.planned = (long long)n;
    if (n == 0) return BM2_OK;
    M.hits.resize((size_t)n);
    {
        int nt = bm2_host_threads();
        if (nt > n_reads / 16384 + 1) nt = n_reads / 16384 + 1;
        if (nt < 1) nt = 1;
        std::atomic<int> nx(0);
        run_threads(nt, [&]() {
            for (int lo; (lo = nx.fetch_add(16384)) < n_reads;)
                for (int r = lo; r < n_reads && r < lo + 16384; ++r)
                    for (int64_t h = reg_off[r]; h < reg_off[r + 1]; ++h) {
                        const bm2_alnreg_t &a = alnregs[h];          // (by_pad: the lists were reordered in place; a hit carries its number)
                        const int32_t t = M.task_of[(size_t)(by_pad ? a.pad - 1 : h)];
                        if (t < 0) continue;
                        bm2h_cg_hit &k = M.hits[(size_t)t];
                        k.rb = a.rb; k.re = a.re; k.read = r; k.qb = a.qb; k.qe = a.qe; k.truesc = a.truesc; k.w = a.w; k.pad = 0;
                    }
        });
    }
    prof.mark("batch list");
    const int rc = cfn(cuser, opt, reads, enc_bytes, n, M.hits.data(), &M.out);
    prof.mark("hook");
    return rc;
}

RescueStats g_rescue;               // counters of the last bm2_sam_pe call (diagnostic; bm2_sam_rescue_stats)

struct PeWork {                     // a chunk's working set, kept per calling thread from chunk to chunk: the hit lists (flat store + one
                                    // HitList per read), the rescue batch (tasks in pair order, their flat arrays, the results), the pairs' plans
    std::vector<bm2_alnreg_t> store; std::vector<HitList> lists; std::vector<int32_t> extra;
    std::vector<RescueTask> tasks; std::vector<int64_t> task_off, q_off, t_pos; std::vector<int32_t> q_len, t_len, xtra;
    std::vector<uint8_t> qbuf; std::vector<bm2_ksw_result> res;
};

void flush_tallies() {
    if (t_cg_used) g_cigar.used += t_cg_used;
    if (t_cg_missed) g_cigar.missed += t_cg_missed;
    if (t_rs_used) g_rescue.used += t_rs_used;
    if (t_rs_missed) g_rescue.missed += t_rs_missed;
    t_cg_used = t_cg_missed = t_rs_used = t_rs_missed = 0;
}

inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#endif
}

// items [0, n) in blocks of 256 over n_threads host threads; f(i, text) appends the text of item i.  Two passes without a wait in either:
// first every thread formats the blocks it draws, one after the other, into its own buffer (kept from call to call) and notes their sizes;
// the blocks' places in the output are a prefix sum over ~n / 256 numbers; then every thread copies ITS blocks to their places.  (One pass
// with an in-order hand-over of the offset from block to block was the earlier form: a third of the threads' time went into waiting for
// the block before, and on a host whose CPU time is capped a descheduled thread stalled everybody behind it.)
// *n_out = bytes needed; BM2_ECAP when cap is smaller (nothing useful was written); a failing item makes the call return BM2_EINVAL with the
// item's number in `bad`.
struct BlockNote { int block; size_t at, size; };
template <class F> int run_blocks(int n, int n_threads, char *out, int64_t cap, int64_t *n_out, int *bad, F f) {
    if (n_threads <= 0) n_threads = bm2_effective_cpus();
    if (n_threads < 1) n_threads = 1;
    const int block = 256;
    const int n_blocks = (n + block - 1) / block;
    if (n_threads > n_blocks) n_threads = n_blocks > 0 ? n_blocks : 1;
    std::vector<int64_t> start_of((size_t)n_blocks + 1, 0);     // sizes, then offsets
    std::atomic<int> next(0), failed(-1);
    TailProf prof("run_blocks");
    static thread_local Text text;                               // (a worker thread's own; the two passes below run on the same threads)
    static thread_local std::vector<BlockNote> notes;
    auto format = [&]() {
        text.clear(); notes.clear();
        for (;;) {
            const int b = next.fetch_add(1);
            if (b >= n_blocks) break;
            const size_t at = text.size();
            if (failed.load(std::memory_order_relaxed) < 0) {
                const int hi = (b + 1) * block < n ? (b + 1) * block : n;
                for (int i = b * block; i < hi; ++i)
                    if (!f(i, text)) { int e = -1; failed.compare_exchange_strong(e, i); break; }
            }
            notes.push_back({ b, at, text.size() - at });
            start_of[(size_t)b + 1] = (int64_t)(text.size() - at);
        }
        flush_tallies();
    };
    run_threads(n_threads, format);
    prof.mark("format");
    if (failed.load() >= 0) { *bad = failed.load(); *n_out = 0; return BM2_EINVAL; }
    for (int b = 0; b < n_blocks; ++b) start_of[(size_t)b + 1] += start_of[(size_t)b];
    *n_out = start_of[(size_t)n_blocks];
    if (*n_out > cap) return BM2_ECAP;
    if (!out) return *n_out ? BM2_EINVAL : BM2_OK;
    // (a thread keeps its buffer for the next chunk -- but not more than RETAIN bytes of it: a caller that formats one huge chunk and then
    //  small ones does not hold the huge chunk's text for the life of its threads)
    const size_t RETAIN = (size_t)64 << 20;
    run_threads(n_threads, [&]() {
        for (const BlockNote &k : notes) if (k.size) memcpy(out + start_of[(size_t)k.block], text.data() + k.at, k.size);
        notes.clear();
        if (text.capacity() > RETAIN) text.release();
    });
    prof.mark("copy");
    return BM2_OK;
}

}  // namespace

extern "C" void bm2_sam_opt_init(bm2_sam_opt *o) {
    if (!o) return;
    memset(o, 0, sizeof *o);
    o->T = 30; o->flag = 0; o->max_XA_hits = 5; o->max_XA_hits_alt = 200; o->XA_drop_ratio = 0.80f;
    o->mapQ_coef_len = 50; o->mapQ_coef_fac = (int32_t)log(o->mapQ_coef_len);     // an int in mem_opt_t: 3
    o->pen_unpaired = 17; o->max_ins = 10000; o->max_matesw = 50;
    o->rg_id = 0;
}

extern "C" int bm2_sam_header(const bm2_index_desc *idx, const char *hdr_line, char *out, int64_t cap, int64_t *n_out) {
    if (!idx || !n_out || !idx->ann_name || !idx->ann_len) { bm2_set_error("bm2_sam_header: the index descriptor needs contig names and lengths"); return BM2_EINVAL; }
    int n_SQ = 0;
    if (hdr_line)                                               // @SQ lines supplied by the caller replace ours (bwa.cpp:527-533)
        for (const char *p = hdr_line; (p = strstr(p, "@SQ\t")) != 0; p += 4) if (p == hdr_line || *(p - 1) == '\n') ++n_SQ;
    std::string s;
    if (n_SQ == 0)
        for (int i = 0; i < idx->n_seqs; ++i) {
            s += "@SQ\tSN:"; s += idx->ann_name[i]; s += "\tLN:"; put_int(s, idx->ann_len[i]);
            s += (idx->ann_is_alt && idx->ann_is_alt[i]) ? "\tAH:*\n" : "\n";
        }
    if (hdr_line) { s += hdr_line; s.push_back('\n'); }
    *n_out = (int64_t)s.size();
    if ((int64_t)s.size() > cap) return BM2_ECAP;
    if (out && !s.empty()) memcpy(out, s.data(), s.size());
    return BM2_OK;
}

extern "C" int bm2_gen_cigar(const bm2_index_desc *idx, const bm2_opt *opt, int32_t n, const uint8_t *seqs, const int64_t *q_off,
                             const int32_t *q_len, const int64_t *rb, const int64_t *re, const int32_t *w, int32_t *score, int32_t *nm,
                             int32_t *n_cigar, int64_t *cigar_off, uint32_t *cigar, int64_t cigar_cap, int64_t *cigar_need, int64_t *md_off,
                             char *md, int64_t md_cap, int64_t *md_need) {
    if (!idx || !opt || n < 0 || !idx->ref_string || (n > 0 && (!seqs || !q_off || !q_len || !rb || !re || !w || !score || !nm || !n_cigar ||
                                                                !cigar_off || !md_off)) || !cigar_need || !md_need) {
        bm2_set_error("bm2_gen_cigar: bad argument"); return BM2_EINVAL;
    }
    Ref R = { idx->l_pac, idx->ref_string, idx->n_seqs, idx->ann_offset, idx->ann_name, idx->ann_anno };
    int64_t co = 0, mo = 0;
    std::vector<uint32_t> cg; std::string mds;
    for (int i = 0; i < n; ++i) {
        int sc = 0, NM = -1;
        const bool ok = gen_cigar(opt->mat, opt->o_del, opt->e_del, opt->o_ins, opt->e_ins, w[i], R, q_len[i], seqs + q_off[i], rb[i], re[i], &sc, cg, &NM, mds);
        score[i] = sc; nm[i] = NM; n_cigar[i] = ok ? (int32_t)cg.size() : -1;
        cigar_off[i] = co; md_off[i] = mo;
        if (ok) {
            if (cigar && co + (int64_t)cg.size() <= cigar_cap) memcpy(cigar + co, cg.data(), cg.size() * 4);
            if (md && mo + (int64_t)mds.size() + 1 <= md_cap) memcpy(md + mo, mds.c_str(), mds.size() + 1);
            co += (int64_t)cg.size(); mo += (int64_t)mds.size() + 1;
        }
    }
    *cigar_need = co; *md_need = mo;
    if (co > cigar_cap || mo > md_cap || (co && !cigar) || (mo && !md)) return BM2_ECAP;
    return BM2_OK;
}

extern "C" int bm2_ksw_align2(int32_t n, const uint8_t *seqs, const int64_t *q_off, const int32_t *q_len, const int64_t *t_off,
                              const int32_t *t_len, const int32_t *xtra, const int8_t mat[25], int o_del, int e_del, int o_ins, int e_ins,
                              bm2_ksw_result *out) {
    if (n < 0 || (n > 0 && (!seqs || !q_off || !q_len || !t_off || !t_len || !xtra || !mat || !out)) || e_del <= 0 || e_ins <= 0) {
        bm2_set_error("bm2_ksw_align2: bad argument"); return BM2_EINVAL;
    }
    for (int i = 0; i < n; ++i) {
        if (q_len[i] <= 0 || t_len[i] < 0) { bm2_set_error("bm2_ksw_align2: pair %d has an empty query", i); return BM2_EINVAL; }
        const KswResult r = ksw_align2(q_len[i], seqs + q_off[i], t_len[i], seqs + t_off[i], mat, o_del, e_del, o_ins, e_ins, xtra[i]);
        out[i].score = r.score; out[i].te = r.te; out[i].qe = r.qe; out[i].score2 = r.score2; out[i].te2 = r.te2; out[i].tb = r.tb; out[i].qb = r.qb;
    }
    return BM2_OK;
}

extern "C" void bm2_sam_rescue_stats(int64_t *planned, int64_t *used, int64_t *missed) {
    if (planned) *planned = g_rescue.planned;
    if (used) *used = g_rescue.used;
    if (missed) *missed = g_rescue.missed;
}

// the CIGAR batch hook on the host (BM2_CIGAR_FLAT=1): the session machinery tested without a GPU
static int host_cigar_batch(void *user, const bm2_opt *opt, const bm2_reads *reads, int64_t, int32_t n, const bm2h_cg_hit *hits, bm2h_cg_out *out) {
    const bm2_index_desc *idx = (const bm2_index_desc *)user;
    Ref R = { idx->l_pac, idx->ref_string, idx->n_seqs, idx->ann_offset, idx->ann_name, idx->ann_anno };
    out->score.assign((size_t)n, 0); out->nm.assign((size_t)n, -1); out->n_cigar.assign((size_t)n, -1);
    out->cigar_off.assign((size_t)n, 0); out->md_off.assign((size_t)n, 0); out->cigar.clear(); out->md.clear();
    std::vector<uint32_t> cg; std::string md;
    for (int32_t i = 0; i < n; ++i) {
        const bm2h_cg_hit &h = hits[i];
        int score = 0, nm = -1;
        const bool ok = cigar_with_retries(opt, R, reads->enc + reads->off[h.read], h.qb, h.qe, h.rb, h.re, h.truesc, h.w, &score, cg, &nm, md);
        out->score[(size_t)i] = score; out->nm[(size_t)i] = nm;
        out->cigar_off[(size_t)i] = (int64_t)out->cigar.size(); out->md_off[(size_t)i] = (int64_t)out->md.size();
        if (ok) { out->n_cigar[(size_t)i] = (int32_t)cg.size(); out->cigar.insert(out->cigar.end(), cg.begin(), cg.end()); out->md.insert(out->md.end(), md.begin(), md.end()); }
        out->md.push_back(0);
    }
    return 0;
}

extern "C" void bm2_sam_cigar_stats(int64_t *planned, int64_t *used, int64_t *missed) {
    if (planned) *planned = g_cigar.planned;
    if (used) *used = g_cigar.used;
    if (missed) *missed = g_cigar.missed;
}

// the batch hook on the host: the same flat arrays the device kernel takes, aligned by the host kernel (BM2_RESCUE_FLAT=1 routes
// bm2_sam_pe through it, so that the flattening is tested without a GPU)
static int host_flat_batch(void *user, int32_t n, const uint8_t *qbuf, int64_t, const int64_t *q_off, const int32_t *q_len, const int64_t *t_pos,
                           const int32_t *t_len, const int32_t *xtra, const bm2_opt *opt, const uint8_t *ref_string, bm2_ksw_result *out) {
    const int n_threads = *(const int *)user;
    std::atomic<int> next(0);
    run_threads(n_threads, [&]() {
        for (int i; (i = next.fetch_add(1)) < n;) {
            const KswResult r = ksw_align2(q_len[i], qbuf + q_off[i], t_len[i], ref_string + t_pos[i], opt->mat, opt->o_del, opt->e_del, opt->o_ins,
                                           opt->e_ins, xtra[i]);
            out[i].score = r.score; out[i].te = r.te; out[i].qe = r.qe; out[i].score2 = r.score2; out[i].te2 = r.te2; out[i].tb = r.tb; out[i].qb = r.qb;
        }
    });
    return 0;
}

extern "C" int bm2_sam_pe(const bm2_index_desc *idx, const bm2_opt *opt, const bm2_sam_opt *so, const bm2_reads *reads,
                          const bm2_read_text *txt, const bm2_alnreg_t *alnregs, const int64_t *reg_off, int64_t n_processed,
                          const bm2_pestat *pes_in, bm2_pestat *pes_out, char *out, int64_t cap, int64_t *n_out) {
    int n_threads = so && so->n_threads > 0 ? so->n_threads : bm2_effective_cpus();
    if (n_threads < 1) n_threads = 1;
    const char *flat = getenv("BM2_RESCUE_FLAT");
    const char *cflat = getenv("BM2_CIGAR_FLAT");
    return bm2h_sam_pe(idx, opt, so, reads, txt, alnregs, reg_off, n_processed, pes_in, pes_out, out, cap, n_out,
                       flat && flat[0] == '1' ? host_flat_batch : nullptr, &n_threads,
                       cflat && cflat[0] == '1' ? host_cigar_batch : nullptr, (void *)idx);
}

int bm2h_sam_pe(const bm2_index_desc *idx, const bm2_opt *opt, const bm2_sam_opt *so, const bm2_reads *reads, const bm2_read_text *txt,
                const bm2_alnreg_t *alnregs, const int64_t *reg_off, int64_t n_processed, const bm2_pestat *pes_in, bm2_pestat *pes_out,
                char *out, int64_t cap, int64_t *n_out, bm2h_ksw_batch_fn fn, void *user, bm2h_cigar_batch_fn cfn, void *cuser) {
    if (!idx || !opt || !so || !reads || !txt || !txt->name || !reg_off || !n_out || (reads->n_reads & 1) || (!alnregs && reg_off[reads->n_reads] > 0)) {
        bm2_set_error("bm2_sam_pe: bad argument (reads must be interleaved pairs)"); return BM2_EINVAL;
    }
    if (!idx->ref_string || !idx->ann_offset || !idx->ann_len || !idx->ann_name) { bm2_set_error("bm2_sam_pe: the index descriptor needs ref_string, contig lengths and names"); return BM2_EINVAL; }
    if (so->max_ins > (1 << 24)) { bm2_set_error("bm2_sam_pe: max_ins above 2^24 is not supported (the insert sizes are counted in a histogram)"); return BM2_EUNSUP; }
    Ref R = { idx->l_pac, idx->ref_string, idx->n_seqs, idx->ann_offset, idx->ann_name, idx->ann_anno };
    struct Budget { int was; explicit Budget(int n) : was(bm2_host_thread_budget()) { bm2_host_thread_budget() = n; } ~Budget() { bm2_host_thread_budget() = was; } } budget(so->n_threads);
    bm2_tune_malloc_once();
    TailProf prof("sam_pe");
    const int n = reads->n_reads;
    std::atomic<int> name_clash(-1);
    {
        int nt = so->n_threads > 0 ? so->n_threads : bm2_effective_cpus();
        if (nt < 1) nt = 1;
        if (nt > n / 16384 + 1) nt = n / 16384 + 1;
        std::atomic<int> nx(0);
        run_threads(nt, [&]() {
            for (int lo; (lo = nx.fetch_add(16384)) < n;)
                for (int i = lo; i < n && i < lo + 16384; i += 2)
                    if (strcmp(txt->name[i], txt->name[i + 1]) != 0) { int e = -1; name_clash.compare_exchange_strong(e, i); }
        });
    }
    if (name_clash.load() >= 0) { const int i = name_clash.load(); bm2_set_error("paired reads have different names: \"%s\", \"%s\"", txt->name[i], txt->name[i + 1]); return BM2_EINVAL; }
    prof.mark("names");
    PeStat pes[4];
    if (pes_in) for (int d = 0; d < 4; ++d) { pes[d].low = pes_in[d].low; pes[d].high = pes_in[d].high; pes[d].failed = pes_in[d].failed; pes[d].avg = pes_in[d].avg; pes[d].std = pes_in[d].std; }
    else pestat(opt, so, idx->l_pac, n, alnregs, reg_off, pes);  // per chunk, as mem_process_seqs does (bwamem.cpp:1366-1370)
    if (pes_out) for (int d = 0; d < 4; ++d) { pes_out[d].low = pes[d].low; pes_out[d].high = pes[d].high; pes_out[d].failed = pes[d].failed; pes_out[d].pad = 0; pes_out[d].avg = pes[d].avg; pes_out[d].std = pes[d].std; }
    // Mate rescue in three steps, the shape a device kernel needs: plan every pair's alignments on the hit lists as they stand,
    // run them all as one batch (here: host threads over single tasks), then process the pairs with the results at hand.
    // so->rescue_inline = 1 aligns inside the pair loop as mem_sam_pe does; the output is the same.
    const int n_pairs = n >> 1;
    prof.mark("pestat");
    // (the flat arrays of the batch live in buffers the calling thread keeps from chunk to chunk: no fresh pages per chunk)
    static thread_local PeWork W_of_this_thread;
    PeWork &W = W_of_this_thread;                                // (a reference: the lambdas below run on the workers, whose own thread_local objects are other objects)
    const uint64_t epoch = ++g_call_epoch;                       // (a list that outgrows its slice moves to its thread's spill arena: valid until that thread's next call)
    if (W.lists.size() < (size_t)n) W.lists.resize((size_t)n);
    if (W.extra.size() < (size_t)n) W.extra.resize((size_t)n);
    HitList *const lists = W.lists.data();
    int32_t *const extra = W.extra.data();                       // hits mate rescue may add to read i = alignments planned with read i as the mate
    const int slack = 2;
    auto fill_store = [&](const std::vector<int64_t> &cap_base, int blk_pairs, int n_blk, int n_threads_) {   // lists[i] <- the input's hits of read i
        std::atomic<int> nb(0);
        run_threads(n_threads_ < n_blk ? n_threads_ : n_blk, [&]() {
            for (int b; (b = nb.fetch_add(1)) < n_blk;) {
                int64_t at = cap_base[(size_t)b];
                for (int i = 2 * b * blk_pairs; i < n && i < 2 * (b + 1) * blk_pairs; ++i) {
                    HitList &L = lists[i];
                    const int k = (int)(reg_off[i + 1] - reg_off[i]);
                    L.p = W.store.data() + at; L.n = k; L.cap = k + extra[i] + slack;
                    if (k) memcpy(L.p, alnregs + reg_off[i], sizeof(bm2_alnreg_t) * (size_t)k);
                    for (int h = 0; h < k; ++h) L.p[h].pad = (int32_t)(reg_off[i] + h + 1);      // the hit's number (CIGAR batch)
                    at += L.cap;
                }
            }
        });
    };
    std::vector<RescueTask> &tasks = W.tasks;
    std::vector<int64_t> &task_off = W.task_off;
    const bool batch = !(so->flag & F_NO_RESCUE) && !so->rescue_inline;
    g_rescue.planned = 0; g_rescue.used = 0; g_rescue.missed = 0;
    if (batch) {
        int n_threads = so->n_threads > 0 ? so->n_threads : bm2_effective_cpus();
        if (n_threads < 1) n_threads = 1;
        const int blk = 256, n_blk = (n_pairs + blk - 1) / blk;
        const int nt_blk = n_threads < n_blk ? n_threads : n_blk;
        std::vector<std::vector<RescueTask>> part((size_t)n_blk);
        std::vector<int64_t> base((size_t)n_blk + 1, 0), qbase((size_t)n_blk + 1, 0), cbase((size_t)n_blk + 1, 0);   // tasks / query bytes / hit slots before block b
        std::atomic<int> next(0);
        run_threads(nt_blk, [&]() {                              // plan: every block lists its pairs' alignments, in pair order (on the input's lists, read only)
            for (int b; (b = next.fetch_add(1)) < n_blk;) {
                std::vector<RescueTask> &v = part[(size_t)b];
                for (int pi = b * blk; pi < n_pairs && pi < (b + 1) * blk; ++pi) {
                    const int l_seq[2] = { reads->len[2 * pi], reads->len[2 * pi + 1] };
                    const HitList a[2] = { view_of(alnregs, reg_off, 2 * pi), view_of(alnregs, reg_off, 2 * pi + 1) };
                    extra[2 * pi] = extra[2 * pi + 1] = 0;
                    rescue_plan(opt, so, R, idx->ann_len, pes, pi, l_seq, a, v);
                }
                int64_t q = 0, slots = 0;
                for (const RescueTask &T : v) { q += reads->len[2 * T.pair + !T.end]; ++extra[2 * T.pair + !T.end]; }      // the mate is the read that is aligned
                for (int i = 2 * b * blk; i < n && i < 2 * (b + 1) * blk; ++i) slots += (reg_off[i + 1] - reg_off[i]) + extra[i] + slack;
                base[(size_t)b + 1] = (int64_t)v.size(); qbase[(size_t)b + 1] = q; cbase[(size_t)b + 1] = slots;
            }
        });
        for (int b = 0; b < n_blk; ++b) { base[(size_t)b + 1] += base[(size_t)b]; qbase[(size_t)b + 1] += qbase[(size_t)b]; cbase[(size_t)b + 1] += cbase[(size_t)b]; }
        if (W.store.size() < (size_t)cbase[(size_t)n_blk] + 1) W.store.resize((size_t)cbase[(size_t)n_blk] + 1);
        fill_store(cbase, blk, n_blk, n_threads);
        const long long tot = (long long)base[(size_t)n_blk];
        const int64_t qb_tot = qbase[(size_t)n_blk];
        prof.mark("rescue plan");
        if (tot > 0x7fffffff) { bm2_set_error("bm2_sam_pe: too many rescue alignments in one chunk"); return BM2_EINVAL; }
        g_rescue.planned = tot;
        const bool flat = fn && tot > 0;
        if (tasks.size() < (size_t)tot) tasks.resize((size_t)tot);
        if (task_off.size() < (size_t)n_pairs + 1) task_off.resize((size_t)n_pairs + 1);
        if (flat) {
            if (W.q_off.size() < (size_t)tot) { W.q_off.resize((size_t)tot); W.t_pos.resize((size_t)tot); W.q_len.resize((size_t)tot); W.t_len.resize((size_t)tot); W.xtra.resize((size_t)tot); W.res.resize((size_t)tot); }
            if (W.qbuf.size() < (size_t)qb_tot + 1) W.qbuf.resize((size_t)qb_tot + 1);
        }
        next = 0;
        run_threads(nt_blk, [&]() {                              // place: the blocks' lists at their offsets, the batch's flat arrays beside them
            for (int b; (b = next.fetch_add(1)) < n_blk;) {
                const std::vector<RescueTask> &v = part[(size_t)b];
                int64_t g = base[(size_t)b], qb = qbase[(size_t)b];
                size_t k = 0;
                for (int pi = b * blk; pi < n_pairs && pi < (b + 1) * blk; ++pi) {
                    task_off[(size_t)pi] = g;
                    for (; k < v.size() && v[k].pair == pi; ++k, ++g) {
                        const RescueTask &T = v[k];
                        tasks[(size_t)g] = T;
                        if (!flat) continue;
                        const int m = 2 * pi + !T.end, l_ms = reads->len[m];
                        W.q_off[(size_t)g] = qb; W.q_len[(size_t)g] = l_ms;
                        W.t_pos[(size_t)g] = T.rb; W.t_len[(size_t)g] = (int32_t)(T.re - T.rb); W.xtra[(size_t)g] = rescue_xtra(opt, l_ms);
                        const uint8_t *ms = reads->enc + reads->off[m];
                        uint8_t *q = W.qbuf.data() + qb;         // the mate as direction r reads it (rescue_query)
                        if (!(T.r >> 1 != (T.r & 1))) memcpy(q, ms, (size_t)l_ms);
                        else for (int i = 0; i < l_ms; ++i) q[l_ms - 1 - i] = ms[i] < 4 ? 3 - ms[i] : 4;
                        qb += l_ms;
                    }
                }
            }
        });
        task_off[(size_t)n_pairs] = tot;
        if (flat) {                                              // one call of the hook
            prof.mark("rescue flatten");
            const int rc = fn(user, (int32_t)tot, W.qbuf.data(), qb_tot, W.q_off.data(), W.q_len.data(), W.t_pos.data(), W.t_len.data(), W.xtra.data(), opt,
                              idx->ref_string, W.res.data());
            if (rc) return rc;
            prof.mark("rescue batch");
            std::atomic<long long> nr(0);
            run_threads((long long)n_threads < tot / 8192 + 1 ? n_threads : (int)(tot / 8192 + 1), [&]() {
                for (long long t0; (t0 = nr.fetch_add(8192)) < tot;)
                    for (long long t = t0; t < tot && t < t0 + 8192; ++t) {
                        const bm2_ksw_result &r = W.res[(size_t)t];
                        KswResult &o = tasks[(size_t)t].res;
                        o.score = r.score; o.te = r.te; o.qe = r.qe; o.score2 = r.score2; o.te2 = r.te2; o.tb = r.tb; o.qb = r.qb;
                    }
            });
        } else {
            std::atomic<long long> nt(0);
            const long long step = 32;
            auto align = [&]() {
                for (long long t0; (t0 = nt.fetch_add(step)) < tot;)
                    for (long long t = t0; t < tot && t < t0 + step; ++t) {
                        RescueTask &T = tasks[(size_t)t];
                        const int m = 2 * T.pair + !T.end;       // the mate is the read that is aligned
                        T.res = rescue_align(opt, R, reads->len[m], reads->enc + reads->off[m], T.r, T.rb, T.re);
                    }
            };
            run_threads((long long)n_threads < (tot + step - 1) / step ? n_threads : (int)((tot + step - 1) / step), align);
        }
    }
    if (!batch) {                                                // no plan: the lists with the slack only (inline rescue grows them through the spill arena)
        int n_threads = so->n_threads > 0 ? so->n_threads : bm2_effective_cpus();
        const int blk = 4096, n_blk = (n_pairs + blk - 1) / blk;
        std::vector<int64_t> cbase((size_t)n_blk + 1, 0);
        for (int b = 0; b < n_blk; ++b) {
            const int lo = 2 * b * blk, hi = 2 * (b + 1) * blk < n ? 2 * (b + 1) * blk : n;
            cbase[(size_t)b + 1] = cbase[(size_t)b] + (reg_off[hi] - reg_off[lo]) + (int64_t)slack * (hi - lo);
        }
        for (int i = 0; i < n; ++i) extra[i] = 0;
        if (W.store.size() < (size_t)cbase[(size_t)n_blk] + 1) W.store.resize((size_t)cbase[(size_t)n_blk] + 1);
        fill_store(cbase, blk, n_blk, n_threads < 1 ? 1 : n_threads);
    }
    auto io_of = [&](int pi, ReadIO io[2]) {
        const int i = pi << 1;
        for (int k = 0; k < 2; ++k) {
            io[k].name = txt->name[i + k]; io[k].comment = txt->comment ? txt->comment[i + k] : 0; io[k].qual = txt->qual ? txt->qual[i + k] : 0;
            io[k].l_seq = reads->len[i + k]; io[k].seq = reads->enc + reads->off[i + k];
        }
    };
    auto decide = [&](int pi, PairPlan &P) {
        ReadIO io[2]; io_of(pi, io);
        const RescueTask *pre = batch ? tasks.data() + task_off[(size_t)pi] : nullptr;
        const int n_pre = batch ? (int)(task_off[(size_t)pi + 1] - task_off[(size_t)pi]) : 0;
        t_scratch.reset(); spill_sync(epoch);
        pe_decide(opt, so, R, idx->ann_len, pes, (uint64_t)((n_processed >> 1) + pi), io, lists + 2 * pi, pre, n_pre, batch ? &g_rescue : nullptr, P);
    };
    auto emit = [&](int pi, const PairPlan &P, Text &part) {
        ReadIO io[2]; io_of(pi, io);
        t_scratch.reset();                                       // (the pair's CIGARs, XA strings and record lists: gone with the next pair)
        return pe_emit(opt, so, R, pes, io, lists + 2 * pi, P, part);
    };
    static thread_local CgMemo memo_of_this_thread;              // (kept from chunk to chunk, like W)
    CgMemo &memo = memo_of_this_thread;
    prof.mark("rescue results");
    g_cigar.planned = 0; g_cigar.used = 0; g_cigar.missed = 0;
    static thread_local std::vector<PairPlan> plans_of_this_thread;      // (kept from chunk to chunk)
    std::vector<PairPlan> &plans = plans_of_this_thread;
    if (cfn) {                                                   // CIGAR session: decide every pair, note the hits its text will ask for, batch; then print
        int n_threads = so->n_threads > 0 ? so->n_threads : bm2_effective_cpus();
        if (n_threads < 1) n_threads = 1;
        if (plans.size() < (size_t)n_pairs) plans.resize((size_t)n_pairs);
        const int blk = 256, n_blk = (n_pairs + blk - 1) / blk;
        std::vector<std::vector<int32_t>> recs((size_t)n_blk);
        std::atomic<int> next(0), failed(-1);
        run_threads(n_threads < n_blk ? n_threads : n_blk, [&]() {
            Text sink;
            for (int b; (b = next.fetch_add(1)) < n_blk;) {
                t_cg.mode = 1; t_cg.rec = &recs[(size_t)b];
                for (int pi = b * blk; pi < n_pairs && pi < (b + 1) * blk; ++pi) {
                    decide(pi, plans[(size_t)pi]);
                    if (!emit(pi, plans[(size_t)pi], sink)) { int e = -1; failed.compare_exchange_strong(e, pi); }
                    sink.clear();
                }
                t_cg = CgSession();
            }
            flush_tallies();
        });
        prof.mark("decide + dry emit");
        int64_t enc_bytes = 0;
        for (int i = 0; i < n; ++i) if (reads->off[i] + reads->len[i] > enc_bytes) enc_bytes = reads->off[i] + reads->len[i];
        const int rc = cigar_session_batch(opt, reads, enc_bytes, alnregs, reg_off, recs, cfn, cuser, memo, false);
        if (rc) return rc;
        prof.mark("cigar session");
    }
    int bad = -1;
    const int rc_out = run_blocks(n_pairs, so->n_threads, out, cap, n_out, &bad, [&](int pi, Text &part) {
        if (!cfn) { PairPlan P; decide(pi, P); return emit(pi, P, part); }
        t_cg.mode = 2; t_cg.memo = &memo; t_cg.st = &g_cigar;
        const bool r = emit(pi, plans[(size_t)pi], part);
        t_cg = CgSession();
        return r;
    });
    prof.mark("real pass + copy");
    if (bad >= 0) { bm2_set_error("bm2_sam_pe: pair %d has a hit whose CIGAR cannot be generated (range outside the reference)", bad); return BM2_EINVAL; }
    return rc_out;
}

extern "C" int bm2_sam_se(const bm2_index_desc *idx, const bm2_opt *opt, const bm2_sam_opt *so, const bm2_reads *reads,
                          const bm2_read_text *txt, bm2_alnreg_t *alnregs, const int64_t *reg_off, int64_t n_processed, char *out,
                          int64_t cap, int64_t *n_out) {
    const char *cflat = getenv("BM2_CIGAR_FLAT");
    return bm2h_sam_se(idx, opt, so, reads, txt, alnregs, reg_off, n_processed, out, cap, n_out,
                       cflat && cflat[0] == '1' ? host_cigar_batch : nullptr, (void *)idx);
}

int bm2h_sam_se(const bm2_index_desc *idx, const bm2_opt *opt, const bm2_sam_opt *so, const bm2_reads *reads, const bm2_read_text *txt,
                bm2_alnreg_t *alnregs, const int64_t *reg_off, int64_t n_processed, char *out, int64_t cap, int64_t *n_out,
                bm2h_cigar_batch_fn cfn, void *cuser) {
    if (!idx || !opt || !so || !reads || !txt || !txt->name || !reg_off || !n_out || (!alnregs && reg_off[reads->n_reads] > 0)) {
        bm2_set_error("bm2_sam_se: bad argument"); return BM2_EINVAL;
    }
    if (!idx->ref_string || !idx->ann_offset || !idx->ann_name) { bm2_set_error("bm2_sam_se: the index descriptor needs ref_string and contig names"); return BM2_EINVAL; }
    Ref R = { idx->l_pac, idx->ref_string, idx->n_seqs, idx->ann_offset, idx->ann_name, idx->ann_anno };
    struct Budget { int was; explicit Budget(int n) : was(bm2_host_thread_budget()) { bm2_host_thread_budget() = n; } ~Budget() { bm2_host_thread_budget() = was; } } budget(so->n_threads);
    bm2_tune_malloc_once();
    const int n_reads = reads->n_reads;
    auto decide = [&](int i) {                                   // what changes the read's hit list (mem_reg2sam's caller, bwamem.cpp:1240-1243)
        t_scratch.reset();
        bm2_alnreg_t *a = alnregs + reg_off[i];
        const int n = (int)(reg_off[i + 1] - reg_off[i]);
        mark_primary_se(opt, n, a, n_processed + i);
        if (so->flag & F_PRIMARY5) reorder_primary5(so->T, n, a);
    };
    auto emit = [&](int i, Text &part) {                         // reads the list only: a CIGAR session runs it twice
        t_scratch.reset();
        return reg2sam(opt, so, R, part, txt->name[i], txt->comment ? txt->comment[i] : 0, txt->qual ? txt->qual[i] : 0, reads->len[i],
                       reads->enc + reads->off[i], (int)(reg_off[i + 1] - reg_off[i]), alnregs + reg_off[i], 0, 0);
    };
    CgMemo memo;
    g_cigar.planned = 0; g_cigar.used = 0; g_cigar.missed = 0;
    for (int64_t k = 0; k < reg_off[n_reads]; ++k) alnregs[k].pad = (int32_t)(k + 1);     // the hit's number (CIGAR batch); the lists are reordered in place
    if (cfn) {                                                   // CIGAR session (see reg2aln): decide, note the hits the text will ask for, batch; then print
        int n_threads = so->n_threads > 0 ? so->n_threads : bm2_effective_cpus();
        if (n_threads < 1) n_threads = 1;
        const int blk = 512, n_blk = (n_reads + blk - 1) / blk;
        std::vector<std::vector<int32_t>> recs((size_t)n_blk);
        std::atomic<int> next(0);
        run_threads(n_threads < n_blk ? n_threads : n_blk, [&]() {
            Text sink;
            for (int b; (b = next.fetch_add(1)) < n_blk;) {
                t_cg.mode = 1; t_cg.rec = &recs[(size_t)b];
                for (int i = b * blk; i < n_reads && i < (b + 1) * blk; ++i) {
                    decide(i);
                    emit(i, sink);
                    sink.clear();
                }
                t_cg = CgSession();
            }
        });
        int64_t enc_bytes = 0;
        for (int i = 0; i < n_reads; ++i) if (reads->off[i] + reads->len[i] > enc_bytes) enc_bytes = reads->off[i] + reads->len[i];
        const int rc = cigar_session_batch(opt, reads, enc_bytes, alnregs, reg_off, recs, cfn, cuser, memo, true);
        if (rc) return rc;
    }
    int bad = -1;
    const int rc_out = run_blocks(n_reads, so->n_threads, out, cap, n_out, &bad, [&](int i, Text &part) {
        if (!cfn) { decide(i); return emit(i, part); }
        t_cg.mode = 2; t_cg.memo = &memo; t_cg.st = &g_cigar;
        const bool r = emit(i, part);
        t_cg = CgSession();
        return r;
    });
    if (bad >= 0) { bm2_set_error("bm2_sam_se: read %d has a hit whose CIGAR cannot be generated (range outside the reference)", bad); return BM2_EINVAL; }
    return rc_out;
}
