// fastq_io.cpp -- host I/O row of SURVEY.md 8(f): FASTA/FASTQ text -> the packed reads the library takes (bm2_reads) plus the
// names / comments / qualities the SAM writer needs.  The record grammar is kseq's (kseq.h:185-227, as bseq_read uses it,
// bwa.cpp:62-216): a record starts at '>' or '@'; the name ends at the first white space, the rest of the line is the comment;
// sequence lines run until a line that starts with '>', '+' or '@'; after '+' the quality lines run until they are as long
// as the sequence.  bseq_read then trims a trailing "/<digit>" from the name (trim_readno, bwa.cpp:62-66) and mem_kernel1_core
// converts the bases with nst_nt4_table (bwamem.cpp:992-1000): ACGT in either case -> 0..3, everything else -> 4.
#include <ctype.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <thread>
#include "host_pool.h"
#include <vector>
#include "../../include/bm2.h"

void bm2_set_error(const char *fmt, ...);
// bm2_api.hip: chunk-sized arrays come page-locked from a pool (plain DMA to the device, no staging copy); small ones are malloc'd
void *bm2_chunk_mem_get(size_t bytes);
void bm2_chunk_mem_put(void *p);

// The two per-base loops of the four-line fast path -- the bases through nst_nt4_table, and the look for a character that only another record
// grammar would put into a sequence line -- 32 bases at a time where the CPU has AVX2 (found at run time; the scalar loops remain for the tail of
// a sequence and for other CPUs).  The parser was 1.0 GB/s per thread with byte loops: a third of a CPU-second per million reads in a leg that
// has 16 CPUs for everything.
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
#include <immintrin.h>
#define BM2_FASTQ_AVX2 1
namespace {
// ACGT in either case -> 0..3 as ((c >> 1) ^ (c >> 2)) & 3, everything else -> 4
__attribute__((target("avx2"))) int encode_avx2(const char *s, int n, uint8_t *d) {
    const __m256i lc = _mm256_set1_epi8(0x20), three = _mm256_set1_epi8(3), four = _mm256_set1_epi8(4);
    const __m256i ca = _mm256_set1_epi8('a'), cc = _mm256_set1_epi8('c'), cg = _mm256_set1_epi8('g'), ct = _mm256_set1_epi8('t');
    int k = 0;
    for (; k + 32 <= n; k += 32) {
        const __m256i c = _mm256_loadu_si256((const __m256i *)(s + k));
        const __m256i l = _mm256_or_si256(c, lc);
        const __m256i ok = _mm256_or_si256(_mm256_or_si256(_mm256_cmpeq_epi8(l, ca), _mm256_cmpeq_epi8(l, cc)), _mm256_or_si256(_mm256_cmpeq_epi8(l, cg), _mm256_cmpeq_epi8(l, ct)));
        const __m256i code = _mm256_and_si256(_mm256_xor_si256(_mm256_srli_epi16(c, 1), _mm256_srli_epi16(c, 2)), three);     // (the bits a 16-bit shift drags in from the neighbour land in bits 6, 7)
        _mm256_storeu_si256((__m256i *)(d + k), _mm256_blendv_epi8(four, code, ok));
    }
    return k;
}
// the first k (a multiple of 32) bases hold none of '>' '+' '@' '\r'; returns -1 if one of them does
__attribute__((target("avx2"))) int clean_avx2(const char *s, int n) {
    const __m256i c1 = _mm256_set1_epi8('>'), c2 = _mm256_set1_epi8('+'), c3 = _mm256_set1_epi8('@'), c4 = _mm256_set1_epi8('\r');
    int k = 0;
    __m256i bad = _mm256_setzero_si256();
    for (; k + 32 <= n; k += 32) {
        const __m256i c = _mm256_loadu_si256((const __m256i *)(s + k));
        bad = _mm256_or_si256(bad, _mm256_or_si256(_mm256_or_si256(_mm256_cmpeq_epi8(c, c1), _mm256_cmpeq_epi8(c, c2)), _mm256_or_si256(_mm256_cmpeq_epi8(c, c3), _mm256_cmpeq_epi8(c, c4))));
    }
    return _mm256_movemask_epi8(bad) ? -1 : k;
}
const bool have_avx2 = __builtin_cpu_supports("avx2") != 0;
}  // namespace
#endif

namespace {
struct Cur {
    const char *p, *e;
    int get() { return p < e ? (unsigned char)*p++ : -1; }
    // append up to (not including) the end of the line; consumes the '\n'; drops one trailing '\r' (ks_getuntil2 with KS_SEP_LINE)
    bool line(std::string &s, bool append) {
        if (!append) s.clear();
        if (p >= e) return false;
        const char *nl = (const char *)memchr(p, '\n', (size_t)(e - p));
        const char *end = nl ? nl : e;
        s.append(p, (size_t)(end - p));
        p = nl ? nl + 1 : e;
        if (s.size() > 1 && s.back() == '\r') s.pop_back();
        return true;
    }
};
char *dup(const std::string &s) { char *d = (char *)malloc(s.size() + 1); if (d) { memcpy(d, s.data(), s.size()); d[s.size()] = 0; } return d; }
}  // namespace

extern "C" void bm2_fastq_free(bm2_fastq *f) {
    if (!f) return;
    if (f->arena) free(f->arena);                              // bm2_fastq_parse_mt: every string lives in one arena
    else for (int i = 0; i < f->n_reads; i++) {
        if (f->name) free(f->name[i]);
        if (f->comment) free(f->comment[i]);
        if (f->qual) free(f->qual[i]);
    }
    free(f->name); free(f->comment); free(f->qual); bm2_chunk_mem_put(f->enc); bm2_chunk_mem_put(f->off); bm2_chunk_mem_put(f->len);
    memset(f, 0, sizeof *f);
}

extern "C" int bm2_fastq_parse(const char *text, int64_t n_bytes, bm2_fastq *out) {
    if (!out || n_bytes < 0 || (n_bytes > 0 && !text)) { bm2_set_error("bm2_fastq_parse: bad argument"); return BM2_EINVAL; }
    memset(out, 0, sizeof *out);
    Cur c = { text, text + n_bytes };
    std::vector<std::string> names, comments, quals;
    std::vector<char> has_comment, has_qual;
    std::vector<uint8_t> enc;
    std::vector<int64_t> off;
    std::vector<int32_t> len;
    int last = 0;                                               // header character already consumed, or 0
    std::string name, comment, seq, qual;
    for (;;) {
        int ch;
        if (last == 0) {                                        // jump to the next header line
            while ((ch = c.get()) != -1 && ch != '>' && ch != '@') {}
            if (ch == -1) break;
            last = ch;
        }
        name.clear(); comment.clear(); seq.clear(); qual.clear();
        int delim = -1;                                         // name: up to the first white space
        while (c.p < c.e) { const int x = (unsigned char)*c.p++; if (isspace(x)) { delim = x; break; } name.push_back((char)x); }
        if (delim == -1 && name.empty()) break;                 // end of input right after a header character
        if (delim != '\n' && delim != -1) c.line(comment, false);
        while ((ch = c.get()) != -1 && ch != '>' && ch != '+' && ch != '@') {
            if (ch == '\n') continue;
            seq.push_back((char)ch);
            c.line(seq, true);
        }
        last = (ch == '>' || ch == '@') ? ch : 0;
        bool got_qual = false;
        if (ch == '+') {
            while ((ch = c.get()) != -1 && ch != '\n') {}
            if (ch == -1) { bm2_set_error("bm2_fastq_parse: record %zu (%s) has no quality string", names.size(), name.c_str()); return BM2_EINVAL; }
            while (c.line(qual, true) && qual.size() < seq.size()) {}
            last = 0;
            if (qual.size() != seq.size()) { bm2_set_error("bm2_fastq_parse: record %zu (%s): quality string of a different length", names.size(), name.c_str()); return BM2_EINVAL; }
            got_qual = !qual.empty();
        }
        if (name.size() > 2 && name[name.size() - 2] == '/' && isdigit((unsigned char)name.back())) name.resize(name.size() - 2);
        if (seq.size() > 0x7fffffff) { bm2_set_error("bm2_fastq_parse: record too long"); return BM2_EINVAL; }
        off.push_back((int64_t)enc.size());
        len.push_back((int32_t)seq.size());
        for (char b : seq) {
            uint8_t v = 4;
            switch (b) { case 'A': case 'a': v = 0; break; case 'C': case 'c': v = 1; break; case 'G': case 'g': v = 2; break; case 'T': case 't': v = 3; break; }
            enc.push_back(v);
        }
        names.push_back(name); comments.push_back(comment); quals.push_back(qual);
        has_comment.push_back(!comment.empty()); has_qual.push_back(got_qual);
    }
    const size_t n = names.size();
    out->n_reads = (int32_t)n; out->n_bases = (int64_t)enc.size();
    out->enc = (uint8_t *)bm2_chunk_mem_get(enc.size() + 64); out->off = (int64_t *)bm2_chunk_mem_get((n + 1) * 8); out->len = (int32_t *)bm2_chunk_mem_get((n + 1) * 4);
    out->name = (char **)calloc(n + 1, sizeof(char *)); out->comment = (char **)calloc(n + 1, sizeof(char *)); out->qual = (char **)calloc(n + 1, sizeof(char *));
    if (!out->enc || !out->off || !out->len || !out->name || !out->comment || !out->qual) { bm2_fastq_free(out); return BM2_ENOMEM; }
    if (!enc.empty()) memcpy(out->enc, enc.data(), enc.size());
    for (size_t i = 0; i < n; i++) {
        out->off[i] = off[i]; out->len[i] = len[i];
        out->name[i] = dup(names[i]);
        out->comment[i] = has_comment[i] ? dup(comments[i]) : 0;   // kseq2bseq1: NULL when empty (bwa.cpp:68-78)
        out->qual[i] = has_qual[i] ? dup(quals[i]) : 0;
    }
    return BM2_OK;
}

// ---- the same for the common case at full speed: strict four-line FASTQ records, one or two files, on n_threads host threads.
// Each file is cut into byte ranges at record starts (a line that begins with '@' whose second-next line begins with '+': a
// quality line may begin with '@' too, but then the line two below it is a sequence line); every range is scanned for
// (name, comment, sequence, quality) spans; two files are interleaved record by record as bseq_read_orig does (bwa.cpp:170-216;
// the shorter file ends the input).  All strings go into ONE arena (three million mallocs per million reads cost more than the
// scan).  Anything that is not a strict four-line record (FASTA, wrapped sequence lines, stray blank lines) makes the function
// fall back to the sequential parser above, whose grammar is kseq's; the result is the same either way.
namespace {
struct Span { const char *name; int name_len; const char *comment; int comment_len; const char *seq; int len; const char *qual; };

// records of [b, e) where b is a record start; false = not four-line FASTQ
bool scan_range(const char *b, const char *e, std::vector<Span> &out) {
    const char *p = b;
    while (p < e) {
        if (*p != '@') return false;
        const char *l1 = (const char *)memchr(p, '\n', (size_t)(e - p));
        if (!l1) return false;
        const char *l2 = (const char *)memchr(l1 + 1, '\n', (size_t)(e - l1 - 1));
        if (!l2 || l2 + 1 >= e || l2[1] != '+') return false;
        const char *l3 = (const char *)memchr(l2 + 1, '\n', (size_t)(e - l2 - 1));
        if (!l3) return false;
        const char *l4 = (const char *)memchr(l3 + 1, '\n', (size_t)(e - l3 - 1));
        const char *qe = l4 ? l4 : e;
        Span s;
        const char *he = l1; if (he > p + 1 && he[-1] == '\r') --he;
        const char *n0 = p + 1, *n1 = n0;
        while (n1 < he && !isspace((unsigned char)*n1)) ++n1;
        if (n1 == n0) return false;                             // an empty name: leave it to kseq's grammar
        s.name = n0; s.name_len = (int)(n1 - n0);
        s.comment = n1 < he ? n1 + 1 : he; s.comment_len = (int)(he - s.comment);
        const char *se = l2; if (se > l1 + 1 && se[-1] == '\r') --se;
        s.seq = l1 + 1;
        if (se - s.seq > 0x7fffffff || se == s.seq) return false;
        s.len = (int)(se - s.seq);
        {
            const char *c = s.seq;
#ifdef BM2_FASTQ_AVX2
            if (have_avx2) { const int done = clean_avx2(s.seq, s.len); if (done < 0) return false; c += done; }
#endif
            for (; c < se; ++c) if (*c == '>' || *c == '+' || *c == '@' || *c == '\r') return false;
        }
        const char *q1 = qe; if (q1 > l3 + 1 && q1[-1] == '\r') --q1;
        s.qual = l3 + 1;
        if (q1 - s.qual != s.len) return false;
        if (s.name_len > 2 && s.name[s.name_len - 2] == '/' && isdigit((unsigned char)s.name[s.name_len - 1])) s.name_len -= 2;   // trim_readno
        out.push_back(s);
        p = l4 ? l4 + 1 : e;
    }
    return true;
}

// first record start at or after p (p itself if it is one)
const char *next_record(const char *b, const char *p, const char *e) {
    if (p <= b) return b;
    const char *q = (const char *)memchr(p - 1, '\n', (size_t)(e - p + 1));
    while (q && q + 1 < e) {
        const char *h = q + 1;
        if (*h == '@') {
            const char *l1 = (const char *)memchr(h, '\n', (size_t)(e - h));
            const char *l2 = l1 ? (const char *)memchr(l1 + 1, '\n', (size_t)(e - l1 - 1)) : nullptr;
            if (l2 && l2 + 1 < e && l2[1] == '+') return h;
        }
        q = (const char *)memchr(h, '\n', (size_t)(e - h));
    }
    return e;
}

bool scan_file(const char *text, int64_t n, int n_threads, std::vector<std::vector<Span>> &parts) {
    const char *b = text, *e = text + n;
    while (b < e && *b != '@') { if (*b == '>' || !isspace((unsigned char)*b)) return false; ++b; }     // leading blank lines only
    int P = n_threads;
    if ((int64_t)P > n / 65536 + 1) P = (int)(n / 65536 + 1);
    std::vector<const char *> cut((size_t)P + 1);
    cut[0] = b; cut[(size_t)P] = e;
    for (int i = 1; i < P; ++i) cut[(size_t)i] = next_record(b, b + (e - b) * i / P, e);
    for (int i = 1; i <= P; ++i) if (cut[(size_t)i] < cut[(size_t)i - 1]) cut[(size_t)i] = cut[(size_t)i - 1];
    parts.assign((size_t)P, {});
    std::vector<char> ok((size_t)P, 1);
    std::atomic<int> nx(0);
    bm2_run_threads(P, [&]() {
        for (int i; (i = nx.fetch_add(1)) < P;) {
            parts[(size_t)i].reserve((size_t)((cut[(size_t)i + 1] - cut[(size_t)i]) / 200 + 16));
            ok[(size_t)i] = scan_range(cut[(size_t)i], cut[(size_t)i + 1], parts[(size_t)i]);
        }
    });
    for (char c : ok) if (!c) return false;
    return true;
}

int seq_fallback(const char *t1, int64_t n1, const char *t2, int64_t n2, bm2_fastq *out);
}  // namespace

extern "C" int bm2_fastq_parse_mt(const char *text1, int64_t n1, const char *text2, int64_t n2, int n_threads, bm2_fastq *out) {
    if (!out || n1 < 0 || (n1 > 0 && !text1) || n2 < 0 || (n2 > 0 && !text2)) { bm2_set_error("bm2_fastq_parse_mt: bad argument"); return BM2_EINVAL; }
    if (n_threads <= 0) n_threads = bm2_effective_cpus();
    if (n_threads < 1) n_threads = 1;
    const bool paired = text2 != nullptr;
    std::vector<std::vector<Span>> pa, pb;
    if (!scan_file(text1, n1, n_threads, pa) || (paired && !scan_file(text2, n2, n_threads, pb))) return seq_fallback(text1, n1, text2, n2, out);
    // flat record tables per file (pointers only), then the output arrays are filled in parallel
    auto flat = [](std::vector<std::vector<Span>> &parts, std::vector<const Span *> &first, std::vector<int64_t> &base) {
        int64_t tot = 0;
        for (auto &v : parts) { first.push_back(v.data()); base.push_back(tot); tot += (int64_t)v.size(); }
        base.push_back(tot);
        return tot;
    };
    std::vector<const Span *> fa, fb; std::vector<int64_t> ba, bb;
    const int64_t ra = flat(pa, fa, ba), rb = paired ? flat(pb, fb, bb) : 0;
    const int64_t n_rec = paired ? (ra < rb ? ra : rb) : ra;    // bseq_read_orig stops at the shorter file
    const int64_t n = paired ? 2 * n_rec : n_rec;
    if (n > 0x7fffffff) { bm2_set_error("bm2_fastq_parse_mt: too many records for one call"); return BM2_EINVAL; }
    auto rec = [](const std::vector<const Span *> &first, const std::vector<int64_t> &base, int64_t i, size_t &part) -> const Span & {
        while (i >= base[part + 1]) ++part;
        return first[part][i - base[part]];
    };
    memset(out, 0, sizeof *out);
    out->n_reads = (int32_t)n;
    out->off = (int64_t *)bm2_chunk_mem_get((size_t)(n + 1) * 8); out->len = (int32_t *)bm2_chunk_mem_get((size_t)(n + 1) * 4);
    out->name = (char **)calloc((size_t)n + 1, sizeof(char *)); out->comment = (char **)calloc((size_t)n + 1, sizeof(char *));
    out->qual = (char **)calloc((size_t)n + 1, sizeof(char *));
    std::vector<int64_t> soff((size_t)n + 1);                   // offsets of record i's strings in the arena
    if (!out->off || !out->len || !out->name || !out->comment || !out->qual) { bm2_fastq_free(out); return BM2_ENOMEM; }
    {   // sizes (sequential: two additions per record)
        int64_t nb = 0, sb = 0; size_t qa = 0, qb = 0;
        for (int64_t i = 0; i < n; ++i) {
            const Span &s = paired ? ((i & 1) ? rec(fb, bb, i >> 1, qb) : rec(fa, ba, i >> 1, qa)) : rec(fa, ba, i, qa);
            out->off[i] = nb; out->len[i] = s.len; nb += s.len;
            soff[(size_t)i] = sb; sb += s.name_len + 1 + (s.comment_len ? s.comment_len + 1 : 0) + s.len + 1;
        }
        soff[(size_t)n] = sb; out->n_bases = nb;
        out->enc = (uint8_t *)bm2_chunk_mem_get((size_t)nb + 64); out->arena = (char *)malloc((size_t)sb + 1);
        if (!out->enc || !out->arena) { bm2_fastq_free(out); return BM2_ENOMEM; }
    }
    static uint8_t nt4[256]; static bool nt4_ready = false;
    if (!nt4_ready) { for (int i = 0; i < 256; ++i) nt4[i] = 4; nt4[(int)'A'] = nt4[(int)'a'] = 0; nt4[(int)'C'] = nt4[(int)'c'] = 1; nt4[(int)'G'] = nt4[(int)'g'] = 2; nt4[(int)'T'] = nt4[(int)'t'] = 3; nt4_ready = true; }
    int T = n_threads; if ((int64_t)T > n / 4096 + 1) T = (int)(n / 4096 + 1);
    std::atomic<int> nx(0);
    bm2_run_threads(T, [&]() {
      for (int t; (t = nx.fetch_add(1)) < T;) {
        const int64_t lo = n * t / T, hi = n * (t + 1) / T;
        size_t qa = 0, qb = 0;
        for (int64_t i = lo; i < hi; ++i) {
            const Span &s = paired ? ((i & 1) ? rec(fb, bb, i >> 1, qb) : rec(fa, ba, i >> 1, qa)) : rec(fa, ba, i, qa);
            uint8_t *d = out->enc + out->off[i];
            int k = 0;
#ifdef BM2_FASTQ_AVX2
            if (have_avx2) k = encode_avx2(s.seq, s.len, d);
#endif
            for (; k < s.len; ++k) d[k] = nt4[(unsigned char)s.seq[k]];
            char *a = out->arena + soff[(size_t)i];
            out->name[i] = a; memcpy(a, s.name, (size_t)s.name_len); a[s.name_len] = 0; a += s.name_len + 1;
            if (s.comment_len) { out->comment[i] = a; memcpy(a, s.comment, (size_t)s.comment_len); a[s.comment_len] = 0; a += s.comment_len + 1; }
            out->qual[i] = a; memcpy(a, s.qual, (size_t)s.len); a[s.len] = 0;
        }
      }
    });
    return BM2_OK;
}

namespace {
// the sequential parser on both files, interleaved
int seq_fallback(const char *t1, int64_t n1, const char *t2, int64_t n2, bm2_fastq *out) {
    if (!t2) return bm2_fastq_parse(t1, n1, out);
    bm2_fastq a, b;
    int rc = bm2_fastq_parse(t1, n1, &a);
    if (rc) return rc;
    if ((rc = bm2_fastq_parse(t2, n2, &b))) { bm2_fastq_free(&a); return rc; }
    const int64_t np = a.n_reads < b.n_reads ? a.n_reads : b.n_reads, n = 2 * np;
    memset(out, 0, sizeof *out);
    out->n_reads = (int32_t)n;
    int64_t nb = 0;
    for (int64_t i = 0; i < np; ++i) nb += a.len[i] + b.len[i];
    out->n_bases = nb;
    out->enc = (uint8_t *)bm2_chunk_mem_get((size_t)nb + 64); out->off = (int64_t *)bm2_chunk_mem_get((size_t)(n + 1) * 8); out->len = (int32_t *)bm2_chunk_mem_get((size_t)(n + 1) * 4);
    out->name = (char **)calloc((size_t)n + 1, sizeof(char *)); out->comment = (char **)calloc((size_t)n + 1, sizeof(char *));
    out->qual = (char **)calloc((size_t)n + 1, sizeof(char *));
    if (!out->enc || !out->off || !out->len || !out->name || !out->comment || !out->qual) { bm2_fastq_free(&a); bm2_fastq_free(&b); bm2_fastq_free(out); return BM2_ENOMEM; }
    int64_t o = 0;
    for (int64_t i = 0; i < n; ++i) {
        bm2_fastq &f = (i & 1) ? b : a; const int64_t k = i >> 1;
        out->off[i] = o; out->len[i] = f.len[k];
        memcpy(out->enc + o, f.enc + f.off[k], (size_t)f.len[k]); o += f.len[k];
        out->name[i] = f.name[k]; f.name[k] = nullptr;          // strings move over
        out->comment[i] = f.comment[k]; f.comment[k] = nullptr;
        out->qual[i] = f.qual[k]; f.qual[k] = nullptr;
    }
    bm2_fastq_free(&a); bm2_fastq_free(&b);
    return BM2_OK;
}
}  // namespace

extern "C" int bm2_host_cpus(void) { return bm2_effective_cpus(); }
