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
#include <vector>
#include "../../include/bm2.h"

void bm2_set_error(const char *fmt, ...);

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
    for (int i = 0; i < f->n_reads; i++) {
        if (f->name) free(f->name[i]);
        if (f->comment) free(f->comment[i]);
        if (f->qual) free(f->qual[i]);
    }
    free(f->name); free(f->comment); free(f->qual); free(f->enc); free(f->off); free(f->len);
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
    out->enc = (uint8_t *)malloc(enc.size() + 64); out->off = (int64_t *)malloc((n + 1) * 8); out->len = (int32_t *)malloc((n + 1) * 4);
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
