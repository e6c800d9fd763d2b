/* oracle/bm2_oracle.h -- TEST INFRASTRUCTURE ONLY (see the header of bm2_oracle.c).
 *
 * Plain-C restatement of the bwa-mem2 v2.2.1 seed -> chain -> extend hot path.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
 */
#ifndef BM2_ORACLE_H
#define BM2_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* FMI_search.h:54-58 (CP_OCC): Occ checkpoint every 64 BWT symbols */
typedef struct { int64_t cp_count[4]; uint64_t bwt[4]; } ora_cpocc;

typedef struct {
    int64_t ref_len;            /* reference_seq_len = 2*l_pac + 1          FMI_search.cpp:415 */
    int64_t count[5];           /* cumulative counts, ALREADY +1            FMI_search.cpp:433-436 */
    int64_t sentinel_index;     /*                                          FMI_search.cpp:457 */
    ora_cpocc *cp_occ;          /* (ref_len>>6)+1 blocks                    FMI_search.cpp:422-431 */
    int8_t   *sa_ms_byte;       /* (ref_len>>3)+1                           FMI_search.cpp:440-446 */
    uint32_t *sa_ls_word;
    uint8_t  *ref_string;       /* <prefix>.0123, 2*l_pac bytes             fastmap.cpp:860-888 */
    uint8_t  *pac;              /* <prefix>.pac                             bntseq.cpp:188-228 */
    int64_t l_pac;
    int32_t n_seqs;
    int64_t *ann_offset;        /* bntann1_t.offset / len / is_alt          bntseq.h:42-49 */
    int32_t *ann_len;
    int32_t *ann_is_alt;
    char   **ann_name;
} ora_index;

/* the subset of mem_opt_t (bwamem.h:76-108) the hot path reads; defaults bwamem.cpp:107-143 */
typedef struct {
    int32_t a, b, o_del, e_del, o_ins, e_ins, pen_clip5, pen_clip3, w, zdrop;
    int32_t min_seed_len, split_width, max_occ, max_chain_gap, min_chain_weight, max_chain_extend;
    int64_t max_mem_intv;
    float split_factor, mask_level, drop_ratio, mask_level_redun;
    int8_t mat[25];
    int8_t pad[3];
} ora_opt;

/* record layouts shared with oracle/refdump.cpp (packed, little endian) */
#pragma pack(push, 1)
typedef struct { int32_t read, m, n, pad; int64_t k, l, s; } ora_smem;                   /* 40 B */
typedef struct { int32_t read, n, rid, is_alt; int64_t pos; float frac_rep; int32_t w, kept, first; } ora_chain_rec;
typedef struct { int64_t rbeg; int32_t qbeg, len, score, pad; } ora_seed_rec;
typedef struct { int32_t read, pad; int64_t rb, re; int32_t qb, qe, rid, score, truesc, sub, alt_sc, csub,
                 sub_n, w, seedcov, secondary, secondary_all, seedlen0, n_comp, is_alt; float frac_rep; int32_t pad2; } ora_reg_rec;
/* one banded-extension task as built by mem_chain2aln_across_reads_V2 (bwamem.cpp:2229-2418) */
typedef struct { int32_t read, reg, is_right, len1, len2, h0; int64_t ref_pos; int32_t q_pos, w_used;
                 int32_t score, qle, tle, gtle, gscore, max_off; } ora_pair_rec;
#pragma pack(pop)

typedef struct {
    int64_t n_smem;   ora_smem *smem;          /* sorted (read, m, n), as after mem_collect_smem */
    int64_t n_sa;     int64_t *sa_coord;       /* per read, per SMEM, per occurrence */
    int32_t *sa_cnt;                           /* [n_reads] */
    int64_t n_chn0;   ora_chain_rec *chn0;  int64_t n_seed0; ora_seed_rec *seed0;  /* after mem_chain_seeds */
    int64_t n_chn1;   ora_chain_rec *chn1;  int64_t n_seed1; ora_seed_rec *seed1;  /* after mem_chain_flt (+flt_chained_seeds) */
    int64_t n_regraw; ora_reg_rec *regraw;     /* after mem_chain2aln_across_reads_V2 (purged: qb=qe=-1) */
    int64_t n_regprg; ora_reg_rec *regprg;     /* after the purge at bwamem.cpp:1141-1152 */
    int64_t n_pair;   ora_pair_rec *pair;      /* every extension task with its accepted result */
    /* work counters for the roofline arithmetic */
    int64_t n_ext, n_ext_sameblk, n_lf, n_sa_lookup, n_sw_cells;
    int64_t n_regfin; ora_reg_rec *regfin;     /* after mem_sort_dedup_patch + ALT flag (bwamem.cpp:1154-1169): what worker_sam receives */
} ora_result;

ora_index *ora_index_load(const char *prefix);
void       ora_index_free(ora_index *ix);
void       ora_opt_init(ora_opt *o);
void       ora_opt_fill_scmat(ora_opt *o);

/* reads: codes 0..4 concatenated; off[i], len[i]; blocks of 512 reads as kt_for makes them */
int  ora_run(const ora_index *ix, const ora_opt *opt, int32_t n_reads, const uint8_t *enc,
             const int64_t *off, const int32_t *len, ora_result *res, int stop_after_seeding);
void ora_result_free(ora_result *r);

/* tail of mem_kernel2_core on regs grouped by read (bwamem.cpp:1154-1169); out holds n_in records, returns the count written */
int64_t ora_finish_regs(const ora_index *ix, const ora_opt *opt, int32_t n_reads, const uint8_t *enc, const int64_t *off,
                        const ora_reg_rec *in, int64_t n_in, ora_reg_rec *out);

/* ksw_extend2 == scalarBandedSWA (bandedSWA.cpp:116-237); m = 5 */
int ora_ksw_extend(int qlen, const uint8_t *query, int tlen, const uint8_t *target, const int8_t *mat,
                   int o_del, int e_del, int o_ins, int e_ins, int w, int end_bonus, int zdrop, int h0,
                   int *qle, int *tle, int *gtle, int *gscore, int *max_off, int64_t *cells);
/* band clamp as the caller's class computes it (A.3 item 15 of SURVEY.md): cls 8/16 = wrapping, 32 = signed */
int ora_ksw_extend_cls(int qlen, const uint8_t *query, int tlen, const uint8_t *target, const int8_t *mat,
                       int o_del, int e_del, int o_ins, int e_ins, int w, int end_bonus, int zdrop, int h0,
                       int *_qle, int *_tle, int *_gtle, int *_gscore, int *_max_off, int64_t *cells, int cls);
int ora_band_clamp(int w, int qlen, int max_sc, int end_bonus, int o_ins, int e_ins, int o_del, int e_del, int cls);
int ora_pair_class(int len1, int len2, int h0, int a);

#ifdef __cplusplus
}
#endif
#endif
