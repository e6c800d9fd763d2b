// index_io.cpp -- host-side reader of the reference's on-disk index (stands in for FMI_search::load_index,
// FMI_search.cpp:384-494; bns_restore, bntseq.cpp:106-228; and the .0123 fread in main_mem, fastmap.cpp:860-888).
// File layout: <prefix>.bwt.2bit.64 = int64 ref_len, int64 count[5], CP_OCC[(ref_len>>6)+1],
// int8 sa_ms_byte[(ref_len>>3)+1], uint32 sa_ls_word[(ref_len>>3)+1], int64 sentinel_index.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <unordered_map>
#include <vector>
#include "../../include/bm2.h"

void bm2_set_error(const char *fmt, ...);

static bool read_exact(FILE *f, void *p, size_t n) { return n == 0 || fread(p, 1, n, f) == n; }

extern "C" void bm2_index_free(bm2_index_desc *d) {
    if (!d) return;
    free((void *)d->cp_occ); free((void *)d->sa_ms_byte); free((void *)d->sa_ls_word); free((void *)d->ref_string);
    free((void *)d->ann_offset); free((void *)d->ann_len); free((void *)d->ann_is_alt);
    if (d->ann_name) { for (int i = 0; i < d->n_seqs; i++) { free((void *)d->ann_name[i]); if (d->ann_anno) free((void *)d->ann_anno[i]); } }
    free((void *)d->ann_name); free((void *)d->ann_anno);
    memset(d, 0, sizeof *d);
}

extern "C" int bm2_index_load(const char *prefix, bm2_index_desc *d) {
    if (!prefix || !d) return BM2_EINVAL;
    memset(d, 0, sizeof *d);
    std::string pre(prefix);
    FILE *f = fopen((pre + ".bwt.2bit.64").c_str(), "rb");
    if (!f) { bm2_set_error("cannot open %s.bwt.2bit.64", prefix); return BM2_EIO; }
    bool ok = read_exact(f, &d->ref_len, 8) && read_exact(f, d->count, 40) && d->ref_len > 0;
    if (ok) {
        size_t nocc = (size_t)(d->ref_len >> 6) + 1, nsa = (size_t)(d->ref_len >> 3) + 1;
        void *occ = aligned_alloc(64, nocc * 64);
        int8_t *ms = (int8_t *)malloc(nsa);
        uint32_t *ls = (uint32_t *)malloc(nsa * 4);
        d->cp_occ = occ; d->sa_ms_byte = ms; d->sa_ls_word = ls;
        ok = occ && ms && ls && read_exact(f, occ, nocc * 64) && read_exact(f, ms, nsa) && read_exact(f, ls, nsa * 4) &&
             read_exact(f, &d->sentinel_index, 8);
    }
    fclose(f);
    if (!ok) { bm2_set_error("%s.bwt.2bit.64 is truncated or malformed", prefix); bm2_index_free(d); return BM2_EIO; }
    // .ann: "l_pac n_seqs seed" then per sequence "gi name anno..." and "offset len n_ambs" (bntseq.cpp:118-147)
    f = fopen((pre + ".ann").c_str(), "r");
    if (!f) { bm2_set_error("cannot open %s.ann", prefix); bm2_index_free(d); return BM2_EIO; }
    long long l_pac; int n_seqs; unsigned seed;
    auto bad_ann = [&](const char *what) {
        bm2_set_error("%s.ann is malformed (%s)", prefix, what);
        fclose(f); bm2_index_free(d);
        return BM2_EIO;
    };
    if (fscanf(f, "%lld%d%u", &l_pac, &n_seqs, &seed) != 3) return bad_ann("first line is not `l_pac n_seqs seed`");
    if (n_seqs <= 0 || n_seqs > 100000000 || l_pac <= 0) return bad_ann("sequence count or packed length out of range");
    if (d->ref_len != 2 * l_pac + 1) return bad_ann("its packed length does not match .bwt.2bit.64: ref_len != 2*l_pac+1");
    d->l_pac = l_pac; d->n_seqs = n_seqs;
    int64_t *off = (int64_t *)calloc((size_t)n_seqs + 1, 8);
    int32_t *len = (int32_t *)calloc((size_t)n_seqs + 1, 4), *alt = (int32_t *)calloc((size_t)n_seqs + 1, 4);
    d->ann_offset = off; d->ann_len = len; d->ann_is_alt = alt;
    std::vector<std::string> names;
    char **nm = (char **)calloc((size_t)n_seqs + 1, sizeof(char *)), **an = (char **)calloc((size_t)n_seqs + 1, sizeof(char *));
    d->ann_name = nm; d->ann_anno = an;
    if (!off || !len || !alt || !nm || !an) return bad_ann("out of memory");
    long long expect = 0;
    for (int i = 0; i < n_seqs; i++) {
        unsigned gi; char name[8193]; int c, namb; long long o;
        if (fscanf(f, "%u%8192s", &gi, name) != 2) return bad_ann("sequence name line missing");
        names.push_back(name);
        nm[i] = strdup(name);
        std::string rest;                                      // the comment up to the end of the line (bntseq.cpp:135-140)
        while ((c = fgetc(f)) != '\n' && c != EOF) if (rest.size() < 8191) rest.push_back((char)c);
        an[i] = strdup(rest.size() > 1 && rest != " (null)" ? rest.c_str() + 1 : "");
        if (fscanf(f, "%lld%d%d", &o, &len[i], &namb) != 3) return bad_ann("`offset len n_ambs` line missing");
        if (o != expect || len[i] < 0) return bad_ann("sequence offsets are not contiguous");
        off[i] = o;
        expect += len[i];
    }
    if (expect != l_pac) return bad_ann("sequence lengths do not add up to l_pac");
    fclose(f);
    if ((f = fopen((pre + ".alt").c_str(), "r")) != 0) {     // bntseq.cpp:201-226: the first field of every line that does not start with
        std::unordered_map<std::string, int> by_name;          // '@' names an ALT contig; the rest of the line (a whole SAM record in
        for (int i = 0; i < n_seqs; i++) by_name.emplace(names[i], i);     // hs38DH.fa.alt, far longer than any line buffer) is skipped
        std::string tok;
        for (int c = fgetc(f); c != EOF; c = fgetc(f)) {
            if (c == '\t' || c == '\n' || c == '\r') {
                if (!tok.empty() && tok[0] != '@') { const auto it = by_name.find(tok); if (it != by_name.end()) alt[it->second] = 1; }
                while (c != '\n' && c != EOF) c = fgetc(f);
                tok.clear();
            } else tok.push_back((char)c);
        }
        fclose(f);
    }
    f = fopen((pre + ".0123").c_str(), "rb");
    if (!f) { bm2_set_error("cannot open %s.0123", prefix); bm2_index_free(d); return BM2_EIO; }
    uint8_t *ref = (uint8_t *)malloc((size_t)(2 * d->l_pac) + 64);
    d->ref_string = ref;
    ok = ref && read_exact(f, ref, (size_t)(2 * d->l_pac));
    fclose(f);
    if (!ok) { bm2_set_error("%s.0123 is truncated", prefix); bm2_index_free(d); return BM2_EIO; }
    return BM2_OK;
}
