// host_tail.h -- seams of the host tail (sam_tail.cpp) the device batches plug into (matesw.hip, cigar.hip)
#pragma once
#include <stdint.h>
#include "../../include/bm2.h"

// The batch of mate-rescue alignments of one chunk, flat: query i = qbuf[q_off[i], +q_len[i]) (the mate, already oriented), target i
// = ref_string[t_pos[i], +t_len[i]), xtra[i] as mem_matesw builds it.  A hook of this type runs the batch (bm2_ksw_align2
// semantics, out[i] = 7 result fields); 0 = success.  The device kernel plugs in here with ref_string resident in HBM.
typedef int (*bm2h_ksw_batch_fn)(void *user, int32_t n, const uint8_t *qbuf, int64_t qbuf_bytes, const int64_t *q_off, const int32_t *q_len,
                                 const int64_t *t_pos, const int32_t *t_len, const int32_t *xtra, const bm2_opt *opt,
                                 const uint8_t *ref_string, bm2_ksw_result *out);
// The batch of CIGAR alignments of one chunk.  An alignment is a function of the HIT alone -- mem_reg2aln (bwamem.cpp:1732-1766) reads
// qb, qe, rb, re, truesc and w of the hit and the read's bases, derives its band, and retries with doubled bands while the score keeps
// improving -- so the tail numbers the chunk's hits, finds out in a dry run of the pairing flow WHICH hits get printed, and hands
// those to a hook of this type in one call; the hook runs the whole retry loop per hit (the device kernel of cigar.hip, or the host
// code for tests) and returns compact arrays: CIGAR ops of hit i at cigar[cigar_off[i] .. +n_cigar[i]) (n_cigar < 0: the reference
// returns NULL), its MD string NUL-terminated at md[md_off[i]].  0 = success.
#include <vector>
struct bm2h_cg_hit { int64_t rb, re; int32_t read, qb, qe, truesc, w, pad; };
struct bm2h_cg_out {
    std::vector<int32_t> score, nm, n_cigar;
    std::vector<int64_t> cigar_off, md_off;
    std::vector<uint32_t> cigar; std::vector<char> md;
};
typedef int (*bm2h_cigar_batch_fn)(void *user, const bm2_opt *opt, const bm2_reads *reads, int64_t enc_bytes, int32_t n, const bm2h_cg_hit *hits,
                                   bm2h_cg_out *out);
// bm2_sam_pe / bm2_sam_se with the rescue batch routed through `fn` and the CIGAR batch through `cfn` (NULL: host code in place)
int bm2h_sam_pe(const bm2_index_desc *idx, const bm2_opt *opt, const bm2_sam_opt *so, const bm2_reads *reads, const bm2_read_text *txt,
                const bm2_alnreg_t *alnregs, const int64_t *reg_off, int64_t n_processed, const bm2_pestat *pes_in, bm2_pestat *pes_out,
                char *out, int64_t cap, int64_t *n_out, bm2h_ksw_batch_fn fn, void *user, bm2h_cigar_batch_fn cfn, void *cuser);
int bm2h_sam_se(const bm2_index_desc *idx, const bm2_opt *opt, const bm2_sam_opt *so, const bm2_reads *reads, const bm2_read_text *txt,
                bm2_alnreg_t *alnregs, const int64_t *reg_off, int64_t n_processed, char *out, int64_t cap, int64_t *n_out,
                bm2h_cigar_batch_fn cfn, void *cuser);

// Phase clock of the tail (BM2_TAIL_PROF=1 prints the phases of every call to stderr): where the host time of a chunk goes.
#include <chrono>
#include <stdio.h>
#include <stdlib.h>
struct TailProf {
    bool on; const char *who; std::chrono::steady_clock::time_point t0, t;
    explicit TailProf(const char *w) : on(getenv("BM2_TAIL_PROF") != nullptr), who(w) { t0 = t = std::chrono::steady_clock::now(); }
    void mark(const char *what) {
        if (!on) return;
        const auto n = std::chrono::steady_clock::now();
        fprintf(stderr, "[tail] %-14s %-22s %8.1f ms\n", who, what, std::chrono::duration<double, std::milli>(n - t).count());
        t = n;
    }
    ~TailProf() { if (on) fprintf(stderr, "[tail] %-14s %-22s %8.1f ms\n", who, "TOTAL", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count()); }
};
