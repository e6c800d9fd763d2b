// index_build.cpp -- multi-threaded builder of the reference's on-disk index (the "index loader / on-disk format" row of
// SURVEY.md section 8(f)).  Produces, for the same FASTA, byte-identical <prefix>.pac/.ann/.amb/.0123/.bwt.2bit.64 to
// `bwa-mem2 index` (bns_fasta2bntseq bntseq.cpp:249-357, bns_dump :73-105, FMI_search::build_index / build_fm_index
// FMI_search.cpp:144-382), so either program can load either index.  The FM-index of a text is unique, so only the
// construction differs: the reference runs single-threaded SA-IS (28N bytes of RAM, ~0.25-0.5 us per base: an hour for a
// human genome); here suffixes are bucketed by their first 11 bases (filed in two steps: 256 partitions, then each partition's
// buckets) and every bucket is sorted independently on all host cores -- by the 32 bases after the bucket prefix as one integer
// key per suffix, ties by word-wise comparison of the 2-bit packed text; a suffix array entry carries the base before its suffix,
// so the BWT falls out of one sequential pass -- what lets a GRCh38-size synthetic genome be indexed inside a benchmark run.
// The big arrays sit on 2 MB pages where the kernel grants them and are first touched by all threads.  Host code; no GPU involved.
#include "host_pool.h"
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <new>
#include <algorithm>
#include <atomic>
#include <string>
#include <thread>
#include <utility>
#include <vector>
#include <chrono>
#include "../../include/bm2.h"

void bm2_set_error(const char *fmt, ...);

namespace {

const int KPRE = 11;                                   // bucket = first 11 bases (4^11 = 4M buckets)

struct Contig { std::string name, anno; int64_t offset; int32_t len, n_ambs; };
struct Hole { int64_t offset; int32_t len; char amb; };

inline int nt4(int c) {                                 // nst_nt4_table, bntseq.cpp:51-68
    switch (c) { case 'A': case 'a': return 0; case 'C': case 'c': return 1; case 'G': case 'g': return 2; case 'T': case 't': return 3; case '-': return 5; default: return 4; }
}

template <class F> void parallel_for(int nthr, int64_t n, F f) {       // f(tid, begin, end) over contiguous ranges
    std::vector<std::thread> th;
    for (int t = 0; t < nthr; t++) {
        const int64_t b = n * t / nthr, e = n * (t + 1) / nthr;
        th.emplace_back([=]() { f(t, b, e); });
    }
    for (auto &x : th) x.join();
}

// The text, its packed copy and the suffix array are gigabytes that the bucket scatter, the bucket sorts and the BWT pass touch at RANDOM:
// on 4 KB pages every such access is a TLB miss on top of the cache miss.  2 MB pages where the kernel gives them out on request
// (transparent_hugepage = madvise); a plain allocation otherwise.
inline void *huge_alloc(size_t bytes) {
    void *p = nullptr;
    const size_t al = (size_t)2 << 20;
    if (posix_memalign(&p, al, (bytes + al - 1) / al * al) != 0) return nullptr;
#ifdef MADV_HUGEPAGE
    (void)madvise(p, (bytes + al - 1) / al * al, MADV_HUGEPAGE);
#endif
    return p;
}
template <class U> struct HugeAlloc {
    typedef U value_type;
    HugeAlloc() {}
    template <class V> HugeAlloc(const HugeAlloc<V> &) {}
    U *allocate(size_t n) { U *p = (U *)huge_alloc(n * sizeof(U)); if (!p) throw std::bad_alloc(); return p; }
    void deallocate(U *p, size_t) { free(p); }
    // resize() leaves new elements uninitialised (every one is written by the loop that follows it -- in parallel, which is also who should
    // take the page faults); assign(n, v) / resize(n, v) still fill
    template <class V> void construct(V *p) { ::new ((void *)p) V; }
    template <class V, class A0, class... A> void construct(V *p, A0 &&a0, A &&...a) { ::new ((void *)p) V(std::forward<A0>(a0), std::forward<A>(a)...); }
    template <class V> bool operator==(const HugeAlloc<V> &) const { return true; }
    template <class V> bool operator!=(const HugeAlloc<V> &) const { return false; }
};

// first touch of a fresh allocation by all threads at once (one write per 4 KB page): page faults -- and the zeroing of 2 MB pages -- cost
// seconds per GB on one thread
inline void prefault(void *p, size_t bytes, int n_threads) {
    volatile uint8_t *q = (volatile uint8_t *)p;
    parallel_for(n_threads, (int64_t)((bytes + 4095) / 4096), [&](int, int64_t b, int64_t e) { for (int64_t i = b; i < e; i++) q[(size_t)i * 4096] = 0; });
}

struct Text {
    int64_t N = 0;                  // 2 * l_pac
    std::vector<uint8_t, HugeAlloc<uint8_t>> T;         // one base per byte (.0123)
    std::vector<uint64_t, HugeAlloc<uint64_t>> P;       // 2-bit packed, 32 bases per word, first base in the top bits
    inline uint64_t get32(int64_t i) const {
        const int64_t w = i >> 5; const int s = (int)(i & 31) * 2;
        return s ? (P[w] << s) | (P[w + 1] >> (64 - s)) : P[w];
    }
    // suffix i < suffix j (i != j); the end of the text sorts before every base (implicit sentinel, as SA-IS does)
    inline bool less(int64_t i, int64_t j, int64_t off = 0) const {      // (off: bases already known to be equal)
        for (;;) {
            if (i + off + 32 <= N && j + off + 32 <= N) {
                const uint64_t a = get32(i + off), b = get32(j + off);
                if (a != b) return a < b;
                off += 32;
            } else {
                for (;;) {
                    if (i + off == N) return true;
                    if (j + off == N) return false;
                    const uint8_t a = T[i + off], b = T[j + off];
                    if (a != b) return a < b;
                    ++off;
                }
            }
        }
    }
};

}  // namespace

extern "C" int bm2_index_build(const char *fasta, const char *prefix, int n_threads) {
    if (!fasta || !prefix) return BM2_EINVAL;
    const bool verbose = getenv("BM2_VERBOSE") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        auto now = std::chrono::steady_clock::now();
        if (verbose) fprintf(stderr, "[bm2_index_build] %-28s %.2f s\n", what, std::chrono::duration<double>(now - t_last).count());
        t_last = now;
    };
    if (n_threads <= 0) n_threads = bm2_effective_cpus();
    if (n_threads <= 0) n_threads = 1;
    // ---- 1. FASTA -> contigs, holes, forward bases (N -> lrand48()&3 after srand48(11): bntseq.cpp:284,314-315)
    FILE *f = fopen(fasta, "rb");
    if (!f) { bm2_set_error("cannot open %s", fasta); return BM2_EIO; }
    fseek(f, 0, SEEK_END); const int64_t fsz = ftell(f); fseek(f, 0, SEEK_SET);
    std::vector<char, HugeAlloc<char>> buf((size_t)fsz + 1);      // (uninitialised: fread fills it)
    prefault(buf.data(), buf.size(), n_threads);
    buf[(size_t)fsz] = 0;
    if (fsz > 0 && (int64_t)fread(buf.data(), 1, (size_t)fsz, f) != fsz) { fclose(f); bm2_set_error("short read on %s", fasta); return BM2_EIO; }
    fclose(f);
    if (fsz >= 2 && (uint8_t)buf[0] == 0x1f && (uint8_t)buf[1] == 0x8b) { bm2_set_error("%s is gzip-compressed: give plain FASTA", fasta); return BM2_EUNSUP; }
    std::vector<Contig> ctg; std::vector<Hole> holes;
    Text tx;
    std::vector<uint8_t, HugeAlloc<uint8_t>> &T = tx.T;
    T.reserve((size_t)2 * (size_t)fsz + 64);           // (forward + reverse complement: no reallocation when the text doubles below)
    T.resize((size_t)fsz + 64);                        // bases <= file size; shrunk below
    prefault(T.data(), T.capacity(), n_threads);
    int64_t tw = 0;                                     // write cursor into T
    uint8_t tbl[256];
    for (int c = 0; c < 256; c++) tbl[c] = (uint8_t)nt4(c);
    // (kseq keeps every character of a sequence line but the '\r' of a CR-LF line end: a blank inside a line is an ambiguous base)
    srand48(11);
    {
        int64_t p = 0; int lasts = 0; Hole *q = nullptr;
        while (p < fsz) {
            if (buf[p] == '>') {
                int64_t e = p; while (e < fsz && buf[e] != '\n') e++;
                std::string hdr(buf.data() + p + 1, buf.data() + e);
                if (!hdr.empty() && hdr.back() == '\r') hdr.pop_back();
                size_t sp = hdr.find_first_of(" \t");
                Contig c; c.name = hdr.substr(0, sp);
                c.anno = "(null)";
                if (sp != std::string::npos && sp + 1 < hdr.size()) c.anno = hdr.substr(sp + 1);     // kseq: the rest of the line
                c.offset = tw; c.len = 0; c.n_ambs = 0;
                ctg.push_back(c); lasts = 0; q = nullptr;
                p = e + 1;
            } else {
                int64_t e = p; while (e < fsz && buf[e] != '\n') e++;
                if (ctg.empty()) { p = e + 1; continue; }
                Contig &c = ctg.back();
                const int64_t tw0 = tw;
                const int64_t le = (e > p && buf[e - 1] == '\r') ? e - 1 : e;
                for (int64_t k = p; k < le; k++) {
                    const int ch = (uint8_t)buf[k];
                    int v = tbl[ch];
                    if (v < 4) { T[tw++] = (uint8_t)v; lasts = ch; continue; }
                    if (v == 255) continue;
                    {                                                       // ambiguous base, bntseq.cpp:266-284
                        const int64_t pos = c.offset + c.len + (tw - tw0);
                        if (lasts == ch && q) ++q->len;
                        else { Hole h; h.offset = pos; h.len = 1; h.amb = (char)ch; holes.push_back(h); q = &holes.back(); ++c.n_ambs; }
                        v = (int)(lrand48() & 3);
                    }
                    lasts = ch;
                    T[tw++] = (uint8_t)v;
                }
                c.len += (int32_t)(tw - tw0);
                p = e + 1;
            }
        }
    }
    std::vector<char, HugeAlloc<char>>().swap(buf);
    T.resize((size_t)tw);
    lap("read + parse FASTA");
    const int64_t l_pac = tw;
    if (l_pac == 0) { bm2_set_error("%s holds no sequence", fasta); return BM2_EIO; }
    const std::string pre(prefix);
    // ---- 2. .pac (forward strand only, bntseq.cpp:331-346), .ann / .amb (bns_dump)
    {
        std::vector<uint8_t> pac((size_t)(l_pac >> 2) + 2, 0);
        for (int64_t i = 0; i < l_pac; i++) pac[i >> 2] |= T[i] << ((~i & 3) << 1);
        FILE *o = fopen((pre + ".pac").c_str(), "wb");
        if (!o) { bm2_set_error("cannot write %s.pac", prefix); return BM2_EIO; }
        bool wrote = fwrite(pac.data(), 1, (size_t)((l_pac >> 2) + ((l_pac & 3) == 0 ? 0 : 1)), o) == (size_t)((l_pac >> 2) + ((l_pac & 3) == 0 ? 0 : 1));
        if (l_pac % 4 == 0) { uint8_t z = 0; wrote = wrote && fwrite(&z, 1, 1, o) == 1; }
        uint8_t ct = (uint8_t)(l_pac % 4); wrote = wrote && fwrite(&ct, 1, 1, o) == 1;
        if (fclose(o) != 0 || !wrote) { bm2_set_error("cannot write %s.pac (disk full?)", prefix); return BM2_EIO; }
        o = fopen((pre + ".ann").c_str(), "w");
        if (!o) { bm2_set_error("cannot write %s.ann", prefix); return BM2_EIO; }
        fprintf(o, "%lld %d %u\n", (long long)l_pac, (int)ctg.size(), 11u);
        for (auto &c : ctg) {
            fprintf(o, "%d %s", 0, c.name.c_str());
            if (!c.anno.empty()) fprintf(o, " %s\n", c.anno.c_str()); else fprintf(o, "\n");
            fprintf(o, "%lld %d %d\n", (long long)c.offset, c.len, c.n_ambs);
        }
        if (fclose(o) != 0) { bm2_set_error("cannot write %s.ann", prefix); return BM2_EIO; }
        o = fopen((pre + ".amb").c_str(), "w");
        if (!o) { bm2_set_error("cannot write %s.amb", prefix); return BM2_EIO; }
        fprintf(o, "%lld %d %u\n", (long long)l_pac, (int)ctg.size(), (unsigned)holes.size());
        for (auto &h : holes) fprintf(o, "%lld %d %c\n", (long long)h.offset, h.len, h.amb);
        if (fclose(o) != 0) { bm2_set_error("cannot write %s.amb", prefix); return BM2_EIO; }
    }
    lap("pac/ann/amb");
    // ---- 3. text = forward + reverse complement (pac2nt, FMI_search.cpp:83-142); .0123
    const int64_t N = 2 * l_pac;
    tx.N = N;
    T.resize((size_t)N);
    parallel_for(n_threads, l_pac, [&](int, int64_t b, int64_t e) { for (int64_t i = b; i < e; i++) T[N - 1 - i] = 3 - T[i]; });
    // (the text is final from here on: its file -- N bytes -- is written by a thread of its own beside the suffix sort)
    struct TextWriter {
        std::thread th; bool ok = true;
        ~TextWriter() { if (th.joinable()) th.join(); }
    } tw0123;
    {
        const uint8_t *tp = T.data(); const std::string fn = pre + ".0123";
        tw0123.th = std::thread([tp, N, fn, &tw0123]() {
            FILE *o = fopen(fn.c_str(), "wb");
            if (!o || (int64_t)fwrite(tp, 1, (size_t)N, o) != N) tw0123.ok = false;
            if (o && fclose(o) != 0) tw0123.ok = false;
        });
    }
    tx.P.resize((size_t)(N >> 5) + 3);
    prefault(tx.P.data(), tx.P.size() * 8, n_threads);
    for (int64_t w = (N >> 5) + 1; w < (N >> 5) + 3; w++) tx.P[(size_t)w] = 0;      // (the words past the text that get32 may read)
    parallel_for(n_threads, (N >> 5) + 1, [&](int, int64_t b, int64_t e) {
        for (int64_t w = b; w < e; w++) {
            uint64_t v = 0;
            for (int k = 0; k < 32; k++) { const int64_t i = w * 32 + k; v = (v << 2) | (i < N ? T[i] : 0); }
            tx.P[w] = v;
        }
    });
    lap(".0123 + packed text");
    int64_t count[5] = { 0, 0, 0, 0, 0 };
    {
        std::vector<std::vector<int64_t>> c4((size_t)n_threads, std::vector<int64_t>(4, 0));
        parallel_for(n_threads, N, [&](int t, int64_t b, int64_t e) { for (int64_t i = b; i < e; i++) c4[t][T[i]]++; });
        int64_t c[4] = { 0, 0, 0, 0 };
        for (auto &v : c4) for (int k = 0; k < 4; k++) c[k] += v[k];
        count[0] = 0; count[1] = c[0]; count[2] = c[0] + c[1]; count[3] = c[0] + c[1] + c[2]; count[4] = c[0] + c[1] + c[2] + c[3];
    }
    // ---- 4. suffix array: SA[0] = N (the empty suffix), then the N suffixes in lexicographic order (FMI_search.cpp:366-368)
    const int64_t NB = 1LL << (2 * KPRE);
    auto key_of = [&](int64_t i) -> int64_t {              // first KPRE bases, zero-padded past the end (a short suffix then
        return (int64_t)(tx.get32(i) >> (64 - 2 * KPRE));  // sorts first inside the bucket it is padded into)
    };
    lap("base counts");
    // (not a std::vector: value-initialising 8 N bytes -- 50 GB for a human genome -- on one thread took longer than sorting them; every entry
    //  is written by the scatter below, whose threads also take the page faults)
    struct RawI64 { int64_t *p; explicit RawI64(size_t n) : p((int64_t *)huge_alloc(n * sizeof(int64_t))) {} ~RawI64() { free(p); } } sa_mem((size_t)N + 2);
    if (!sa_mem.p) { bm2_set_error("bm2_index_build: cannot allocate %lld bytes for the suffix array", (long long)((N + 2) * 8)); return BM2_ENOMEM; }
    int64_t *const SA = sa_mem.p;
    // An entry carries the base BEFORE its suffix in its top bits (bit 62: there is one; bits 61..60: which): it is read where the suffix is
    // filed -- in text order, for nothing -- and saves the BWT pass a cache miss per suffix into the text (positions stay below 2^40).
    const int64_t SMASK = ((int64_t)1 << 60) - 1;
    auto tagged = [&](int64_t i) -> int64_t { return i > 0 ? i | (int64_t)(4 | T[i - 1]) << 60 : i; };
    SA[0] = tagged(N); SA[N + 1] = 0;
    std::vector<int64_t> bstart((size_t)NB + 1, 0);
    const int PBASES = 4, PSHIFT = 2 * (KPRE - PBASES);           // partition = first 4 bases: 256 partitions of 4^7 buckets
    const int64_t NPART = 1LL << (2 * PBASES);
    std::vector<std::vector<int64_t>> pcur;                       // [chunk][partition]: where the chunk's next suffix of the partition goes
    {
        // text chunks: enough for all cores, few enough that the per-chunk histograms (16 MB each) stay cheap to combine
        const int n_chunks = n_threads < 64 ? n_threads : 64;
        std::vector<std::vector<uint32_t>> hist((size_t)n_chunks);
        parallel_for(n_chunks, N, [&](int t, int64_t b, int64_t e) {
            hist[t].assign((size_t)NB, 0);
            for (int64_t i = b; i < e; i++) hist[t][key_of(i)]++;
        });
        lap("bucket histogram");
        // bucket starts; and, per text chunk, how many of its suffixes fall into each PARTITION (= the buckets sharing their first PBASES bases)
        std::vector<int64_t> tot((size_t)NB);
        parallel_for(n_threads, NB, [&](int, int64_t b0, int64_t b1) {
            for (int64_t b = b0; b < b1; b++) { int64_t c = 0; for (int t = 0; t < n_chunks; t++) c += hist[t][b]; tot[b] = c; }
        });
        int64_t acc = 1;
        for (int64_t b = 0; b < NB; b++) { bstart[b] = acc; acc += tot[b]; }
        bstart[NB] = acc;
        pcur.assign((size_t)n_chunks, std::vector<int64_t>((size_t)NPART, 0));
        parallel_for(n_threads, NPART, [&](int, int64_t p0, int64_t p1) {
            for (int64_t p = p0; p < p1; p++) {
                int64_t run = bstart[p << PSHIFT];               // chunk-minor inside the partition: its suffixes arrive in text order
                for (int t = 0; t < n_chunks; t++) {
                    int64_t c = 0;
                    for (int64_t b = p << PSHIFT; b < (p + 1) << PSHIFT; b++) c += hist[t][b];
                    pcur[(size_t)t][(size_t)p] = run; run += c;
                }
            }
        });
        lap("bucket offsets");
        // Scatter in two steps.  Straight into 4^KPRE buckets, every suffix is a random write somewhere in the 8 N bytes of the array and a
        // random read-modify-write of a 16 MB counter table.  First into the NPART partitions -- NPART sequential write streams per thread --,
        // then, partition by partition (below, together with the sorting), from a copy of the partition into its buckets: counters and targets
        // of that step are a few hundred KB and a few MB.
        parallel_for(n_chunks, N, [&](int t, int64_t b, int64_t e) {
            int64_t *cur = pcur[(size_t)t].data();
            for (int64_t i = b; i < e; i++) SA[cur[key_of(i) >> PSHIFT]++] = tagged(i);
        });
    }
    lap("partition scatter");
    {
        std::atomic<int64_t> next(0);
        std::vector<std::thread> th;
        for (int t = 0; t < n_threads; t++)
            th.emplace_back([&]() {
                std::vector<std::pair<uint64_t, int64_t>> kv, kv2;
                std::vector<int64_t> part, cur((size_t)1 << PSHIFT);
                for (;;) {
                    const int64_t p = next.fetch_add(1);
                    if (p >= NPART) break;
                    const int64_t kb = p << PSHIFT, ke = (p + 1) << PSHIFT, plo = bstart[kb], phi = bstart[ke];
                    if (phi - plo > 1) {                          // the partition's suffixes (text order) into its buckets
                        part.assign(SA + plo, SA + phi);
                        for (int64_t k = kb; k < ke; k++) cur[(size_t)(k - kb)] = bstart[k];
                        for (int64_t x : part) SA[cur[(size_t)(key_of(x & SMASK) - kb)]++] = x;
                    }
                    for (int64_t b = kb; b < ke; b++) {
                        const int64_t lo = bstart[b], hi = bstart[b + 1], n = hi - lo;
                        if (n <= 1) continue;
                        if (n < 8) { std::sort(SA + lo, SA + hi, [&](int64_t x, int64_t y) { return tx.less(x & SMASK, y & SMASK, 0); }); continue; }
                        // The bucket's suffixes agree in their first KPRE bases.  Their next 32 bases as ONE integer key each, fetched once
                        // (a random read of the packed text per suffix instead of one per comparison), the (key, suffix) pairs sorted as
                        // integers, and only the runs of equal keys -- repeats longer than KPRE + 32 bases -- compared base by base.  A
                        // bucket with a suffix that ends inside the key window sorts by the full comparison (the end of the text sorts first).
                        kv.resize((size_t)n);
                        bool near_end = false;
                        for (int64_t t = 0; t < n; t++) {
                            const int64_t s = SA[lo + t] & SMASK;
                            if (s + KPRE + 32 > N) { near_end = true; break; }
                            kv[(size_t)t].first = tx.get32(s + KPRE); kv[(size_t)t].second = SA[lo + t];
                        }
                        if (near_end) { std::sort(SA + lo, SA + hi, [&](int64_t x, int64_t y) { return tx.less(x & SMASK, y & SMASK, 0); }); continue; }
                        auto kv_less = [&](const std::pair<uint64_t, int64_t> &x, const std::pair<uint64_t, int64_t> &y) {
                            return x.first != y.first ? x.first < y.first : tx.less(x.second & SMASK, y.second & SMASK, KPRE + 32);
                        };
                        // one counting pass over the key's top byte (the 4 bases after the bucket prefix) leaves runs of a handful of pairs:
                        // insertion sort for those, std::sort for what a repeat makes long
                        uint32_t cnt[257];
                        memset(cnt, 0, sizeof cnt);
                        for (int64_t t = 0; t < n; t++) cnt[(kv[(size_t)t].first >> 56) + 1]++;
                        for (int d = 0; d < 256; d++) cnt[d + 1] += cnt[d];
                        kv2.resize((size_t)n);
                        {
                            uint32_t at[256];
                            memcpy(at, cnt, sizeof at);
                            for (int64_t t = 0; t < n; t++) kv2[at[kv[(size_t)t].first >> 56]++] = kv[(size_t)t];
                        }
                        for (int d = 0; d < 256; d++) {
                            const uint32_t r0 = cnt[d], r1 = cnt[d + 1];
                            if (r1 - r0 < 2) continue;
                            if (r1 - r0 > 24) { std::sort(kv2.begin() + r0, kv2.begin() + r1, kv_less); continue; }
                            for (uint32_t a = r0 + 1; a < r1; a++) {
                                const std::pair<uint64_t, int64_t> v = kv2[a];
                                uint32_t c = a;
                                while (c > r0 && kv_less(v, kv2[c - 1])) { kv2[c] = kv2[c - 1]; --c; }
                                kv2[c] = v;
                            }
                        }
                        for (int64_t t = 0; t < n; t++) SA[lo + t] = kv2[(size_t)t].second;
                    }
                }
            });
        for (auto &x : th) x.join();
    }
    lap("bucket scatter + sort");
    // ---- 5. BWT -> CP_OCC blocks, sampled SA (build_fm_index, FMI_search.cpp:144-304)
    const int64_t ref_seq_len = N + 1;
    const int64_t n_occ = (ref_seq_len >> 6) + 1, n_sa = (ref_seq_len >> 3) + 1;
    struct CpOcc { int64_t cp_count[4]; uint64_t bwt[4]; };
    std::vector<CpOcc, HugeAlloc<CpOcc>> occ((size_t)n_occ);     // (uninitialised: every block is written below, as is every sampled entry)
    std::vector<int8_t, HugeAlloc<int8_t>> ms((size_t)n_sa); std::vector<uint32_t, HugeAlloc<uint32_t>> ls((size_t)n_sa);
    std::vector<int64_t> sent((size_t)n_threads, -1);
    parallel_for(n_threads, n_occ, [&](int t, int64_t b, int64_t e) {
        for (int64_t blk = b; blk < e; blk++) {
            CpOcc c; memset(&c, 0, sizeof c);
            for (int j = 0; j < 64; j++) {
                const int64_t i = blk * 64 + j;
                for (int k = 0; k < 4; k++) c.bwt[k] <<= 1;
                if ((j & 7) == 0 && (i >> 3) < n_sa) {           // the sampled suffix array (every 8th entry: low word + high byte) in the same pass
                    const int64_t sv = i < ref_seq_len ? SA[i] & SMASK : 0;
                    ls[(size_t)(i >> 3)] = (uint32_t)(sv & 0xffffffff); ms[(size_t)(i >> 3)] = (int8_t)((sv >> 32) & 0xff);
                }
                if (i < ref_seq_len) {
                    const int64_t v = SA[i];
                    if ((v & SMASK) == 0) sent[t] = i;
                    else { const int ch = (int)(v >> 60) & 3; c.bwt[ch] += 1; c.cp_count[ch]++; }       // (the base before the suffix rides in the entry) cp_count holds the block's own counts for now
                }
            }
            occ[blk] = c;
        }
    });
    int64_t sentinel_index = -1;
    for (int64_t v : sent) if (v >= 0) sentinel_index = v;
    {
        int64_t run[4] = { 0, 0, 0, 0 };
        for (int64_t blk = 0; blk < n_occ; blk++)
            for (int k = 0; k < 4; k++) { const int64_t c = occ[blk].cp_count[k]; occ[blk].cp_count[k] = run[k]; run[k] += c; }
    }
    lap("BWT / Occ / sampled SA");
    {
        FILE *o = fopen((pre + ".bwt.2bit.64").c_str(), "wb");
        if (!o) { bm2_set_error("cannot write %s.bwt.2bit.64", prefix); return BM2_EIO; }
        bool ok = fwrite(&ref_seq_len, 8, 1, o) == 1 && fwrite(count, 8, 5, o) == 5 &&
                  (int64_t)fwrite(occ.data(), sizeof(CpOcc), (size_t)n_occ, o) == n_occ &&
                  (int64_t)fwrite(ms.data(), 1, (size_t)n_sa, o) == n_sa && (int64_t)fwrite(ls.data(), 4, (size_t)n_sa, o) == n_sa &&
                  fwrite(&sentinel_index, 8, 1, o) == 1;
        fclose(o);
        if (!ok) { bm2_set_error("short write on %s.bwt.2bit.64", prefix); return BM2_EIO; }
    }
    tw0123.th.join();
    if (!tw0123.ok) { bm2_set_error("cannot write %s.0123", prefix); return BM2_EIO; }
    lap("write .bwt.2bit.64");
    return BM2_OK;
}
