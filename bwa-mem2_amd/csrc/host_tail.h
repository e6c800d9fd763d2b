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
// The batch of CIGAR alignments of one chunk (bm2_gen_cigar semantics; queries are ranges of the chunk's read buffer `seqs`).  The
// capacities are upper bounds, so one call always fits.  0 = success.
typedef int (*bm2h_cigar_batch_fn)(void *user, const bm2_opt *opt, int32_t n, const uint8_t *seqs, int64_t seq_bytes, const int64_t *q_off,
                                   const int32_t *q_len, const int64_t *rb, const int64_t *re, const int32_t *w, int32_t *score, int32_t *nm,
                                   int32_t *n_cigar, int64_t *cigar_off, uint32_t *cigar, int64_t cigar_cap, int64_t *md_off, char *md,
                                   int64_t md_cap);
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
