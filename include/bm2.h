/* include/bm2.h -- C ABI of the MI355X seed -> chain -> extend library (libbm2.so).
 *
 * bwa-mem2 has no plugin/FFI interface; the seams this library replaces are ordinary C++
 * calls inside libbwa.  Every entry point below names the reference
 * interface it stands in for (paths relative to the reference's src/).  Plain pointers and
 * sizes only; no C++ or torch types.  All functions return 0 on success or a negative
 * BM2_E* code (the reference exit()s or assert()s instead); outputs are caller-allocated
 * with a capacity and an n_out so the caller can grow and retry (the reference reallocs
 * inside the callee, e.g. bwamem.cpp:718-739).  One bm2_ctx per GPU, used from one host
 * thread at a time.  There is NO CPU fallback: without a usable HIP device every call
 * fails with BM2_ENODEV.
 */
#ifndef BM2_H
#define BM2_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define BM2_OK        0
#define BM2_ENODEV   -1   /* no HIP device / HIP runtime error (message: bm2_last_error) */
#define BM2_ENOMEM   -2   /* host or device allocation failed */
#define BM2_EINVAL   -3   /* bad argument */
#define BM2_ECAP     -4   /* caller-provided output too small; *n_out holds the needed count */
#define BM2_EUNSUP   -5   /* a path of the reference that is not implemented (stated in DESIGN.md) */
#define BM2_EIO      -6   /* index file missing / malformed */

typedef struct bm2_ctx bm2_ctx;

/* ---- index ---------------------------------------------------------------------------------
 * The arrays FMI_search::load_index (FMI_search.cpp:384-494) and main_mem (fastmap.cpp:860-888)
 * read from <prefix>.bwt.2bit.64 / .0123 / .ann / .alt, as host pointers.  `count` is the
 * on-disk cumulative count (the +1 of FMI_search.cpp:433-436 is applied inside the library). */
typedef struct {
    int64_t ref_len;                /* reference_seq_len = 2*l_pac + 1 */
    int64_t count[5];
    int64_t sentinel_index;
    const void     *cp_occ;         /* (ref_len>>6)+1 CP_OCC blocks of 64 B (FMI_search.h:54-58) */
    const int8_t   *sa_ms_byte;     /* (ref_len>>3)+1 */
    const uint32_t *sa_ls_word;     /* (ref_len>>3)+1 */
    const uint8_t  *ref_string;     /* 2*l_pac bytes, values 0..3 (.0123); the device copy holds four per byte (a value above 3 anywhere keeps bytes) */
    int64_t l_pac;
    int32_t n_seqs;
    const int64_t *ann_offset;      /* bntann1_t.offset, .len, .is_alt (bntseq.h:42-49) */
    const int32_t *ann_len;
    const int32_t *ann_is_alt;
    const char *const *ann_name;    /* bntann1_t.name / .anno: only the SAM writer reads them (may be NULL otherwise) */
    const char *const *ann_anno;
} bm2_index_desc;

/* Read the reference's index files into malloc'd host arrays (stands in for
 * FMI_search::load_index + bwa_idx_load_ele + the .0123 fread).  Free with bm2_index_free. */
int  bm2_index_load(const char *prefix, bm2_index_desc *out);
/* Build <prefix>.pac/.ann/.amb/.0123/.bwt.2bit.64 from a plain FASTA, byte-identical to `bwa-mem2 index`
 * (bns_fasta2bntseq bntseq.cpp:249-357; FMI_search::build_index FMI_search.cpp:306-382) but on n_threads host cores
 * (n_threads <= 0: all).  The <prefix>.alt list, if any, is the caller's to provide, as with the reference. */
int  bm2_index_build(const char *fasta, const char *prefix, int n_threads);
void bm2_index_free(bm2_index_desc *d);

/* ---- options: the fields of mem_opt_t (bwamem.h:76-108) the hot path reads ------------------ */
typedef struct {
    int32_t a, b, o_del, e_del, o_ins, e_ins, pen_clip5, pen_clip3, w, zdrop;
    int32_t min_seed_len, split_width, max_occ, max_chain_gap, min_chain_weight, max_chain_extend;
    int64_t max_mem_intv;
    float   split_factor, mask_level, drop_ratio, mask_level_redun;
    int8_t  mat[25];                /* bwa_fill_scmat (bwa.cpp:248-257) */
    int8_t  pad[3];
} bm2_opt;
void bm2_opt_init(bm2_opt *o);      /* mem_opt_init defaults, bwamem.cpp:107-143 */
void bm2_opt_fill_scmat(bm2_opt *o);

/* ---- records (byte-compatible with the reference where a reference struct exists) ----------- */
typedef struct {                    /* SMEM, FMI_search.h:75-83 (40 B) */
    uint32_t rid, m, n, pad;
    int64_t  k, l, s;
} bm2_smem_t;

typedef struct {                    /* SeqPair, bandedSWA.h:90-99 (56 B) */
    int32_t idr, idq, id, len1, len2, h0;
    int32_t seqid, regid;
    int32_t score, tle, gtle, qle, gscore, max_off;
} bm2_seqpair_t;

typedef struct {                    /* BandedPairWiseSW ctor arguments, bandedSWA.h:118-124 */
    int32_t o_del, e_del, o_ins, e_ins, zdrop, end_bonus, w_match, w_mismatch;
    int8_t  mat[25];
    int8_t  pad[3];
} bm2_sw_params;

typedef struct {                    /* the fields of mem_alnreg_t (bwamem.h:137-160) that mem_kernel2_core has
                                     * written when it reaches bwamem.cpp:1152 (before mem_sort_dedup_patch) */
    int64_t rb, re;
    int32_t qb, qe, rid, score, truesc, w, seedcov, seedlen0;
    float   frac_rep;
    int32_t pad;
} bm2_reg_t;

typedef struct {                    /* mem_alnreg_t (bwamem.h:137-160) without the chain pointer: what mem_kernel2_core returns */
    int64_t rb, re;
    int32_t qb, qe, rid, score, truesc, sub, alt_sc, csub, sub_n, w, seedcov, secondary, secondary_all, seedlen0;
    int32_t n_comp, is_alt;
    float   frac_rep;
    int32_t pad;
    uint64_t hash;
} bm2_alnreg_t;

typedef struct {                    /* reads of one chunk: 2-bit codes (0..3, 4 = N), as after bwamem.cpp:992-1000 */
    int32_t n_reads;
    const uint8_t *enc;             /* concatenated codes */
    const int64_t *off;             /* [n_reads] start of read i in enc */
    const int32_t *len;             /* [n_reads] */
} bm2_reads;

typedef struct {                    /* work counters measured by the kernels themselves */
    int64_t n_reads, n_bases;
    int64_t n_smem, n_sa, n_chain, n_reg_raw, n_reg;
    int64_t n_ext;                  /* backwardExt calls                -> 128 B each */
    int64_t n_lf;                   /* LF steps in the SA walk          ->  64 B each */
    int64_t n_sw_cells;             /* DP cells actually computed */
    int64_t n_sw_tasks;
} bm2_stats;

/* ---- lifetime ---------------------------------------------------------------------------- */
bm2_ctx *bm2_create(int device, const bm2_index_desc *idx);     /* uploads the index replica to HBM */
/* a second context on the same device sharing the parent's index replica (own streams / workspaces): overlap two chunks on one
 * GPU, or split one chunk over several contexts, without another upload.  Valid while the parent lives. */
bm2_ctx *bm2_create_shared(bm2_ctx *parent);
void     bm2_destroy(bm2_ctx *c);
/* the context's main stream at a hardware queue priority (level > 0: highest, < 0: lowest, 0: default): e.g. high for the contexts that
 * run the short device batches of the SAM tail beside another context's seeding .. extension */
int      bm2_set_stream_priority(bm2_ctx *c, int level);
const char *bm2_last_error(void);
int      bm2_device_count(void);
/* the CPUs this process can really use: hardware threads it may run on, capped by its cgroup CPU-time quota (a GPU slice of a shared node
 * sees every hardware thread of the host but is given the time of a few); the default of every n_threads <= 0 in this library */
int      bm2_host_cpus(void);
/* page-locked host memory for a caller's large per-chunk arrays (reads, hits, text): the library's copies from / to it are plain DMA at
 * PCIe speed instead of a staged copy through 16 MB bounce buffers; anything else the caller passes is staged, as before */
/* (hardware queues: the HIP runtime reads GPU_MAX_HW_QUEUES when it starts; libbm2 asks for 16 when it is loaded unless the variable is set.
 * A host that initialises HIP BEFORE loading libbm2 must export GPU_MAX_HW_QUEUES=16 itself, or the concurrent launches of the extension stage
 * share four queues -- about 2 ms per million-read chunk.) */
void    *bm2_host_alloc(int64_t bytes);
void     bm2_host_free(void *p);

/* ---- S1: BandedPairWiseSW::getScores8 / getScores16 / scalarBandedSWAWrapper (bandedSWA.h:126-135,
 * 199-211; call sites bwamem.cpp:2476,2544,2613,2692,2757,2828).  Fills score,tle,gtle,qle,gscore,max_off
 * of every pair; the per-pair band clamp follows the pair's class (int8/int16/scalar), as the reference's
 * three entry points do.  ref/qer are the flat seqBuf arrays indexed by idr/idq. */
int bm2_bsw(bm2_ctx *c, bm2_seqpair_t *pairs, const uint8_t *ref, int64_t ref_bytes, const uint8_t *qer,
            int64_t qer_bytes, int32_t n, int32_t w, const bm2_sw_params *p);
/* The same with the batch resident in HBM (BASELINE.json config 2: the banded-SW kernel alone, timed without the copies): upload once,
 * run any number of times (the kernel writes only the six output fields), download.  kernel_ms = duration of the kernel from HIP
 * events on its stream, cells = DP cells computed (either may be NULL; counting costs an atomic per pair: ask for it in an untimed run). */
int bm2_bsw_upload(bm2_ctx *c, const bm2_seqpair_t *pairs, const uint8_t *ref, int64_t ref_bytes, const uint8_t *qer,
                   int64_t qer_bytes, int32_t n);
int bm2_bsw_run(bm2_ctx *c, int32_t w, const bm2_sw_params *p, float *kernel_ms, int64_t *cells);
int bm2_bsw_download(bm2_ctx *c, bm2_seqpair_t *pairs, int32_t n);

/* ---- S2: mem_collect_smem (bwamem.cpp:626-803) = getSMEMsAllPosOneThread + the pass-2 selection +
 * getSMEMsOnePosOneThread + bwtSeedStrategyAllPosOneThread + sortSMEMs (FMI_search.h:106-165).
 * out is sorted by (rid, m, n); rid is the index of the read in `reads`. */
int bm2_smem(bm2_ctx *c, const bm2_reads *reads, const bm2_opt *opt, bm2_smem_t *out, int64_t cap, int64_t *n_out);

/* ---- S2: FMI_search::get_sa_entries_prefetch (FMI_search.cpp:1257-1375): coordinates of up to max_occ
 * sampled occurrences of each SMEM, in (SMEM, occurrence) order. */
int bm2_sal(bm2_ctx *c, const bm2_smem_t *smems, int64_t n, int32_t max_occ, int64_t *coords, int64_t cap,
            int64_t *n_out);

/* ---- S3: mem_kernel1_core + mem_kernel2_core up to bwamem.cpp:1152 for a whole chunk
 * (worker_bwt / worker_aln over all 512-read blocks, bwamem.cpp:1175-1214).
 * regs of read i = regs[reg_off[i] .. reg_off[i+1]); reg_off has n_reads+1 entries. */
int bm2_seed_chain_extend(bm2_ctx *c, const bm2_reads *reads, const bm2_opt *opt, bm2_reg_t *regs, int64_t cap,
                          int64_t *reg_off, int64_t *n_out, bm2_stats *stats);

/* ---- the tail of mem_kernel2_core (bwamem.cpp:1154-1169) = mem_sort_dedup_patch (bwamem.cpp:292-353, with mem_patch_reg :175-225 ->
 * bwa_gen_cigar2 -> ksw_global2) and the ALT flag, ON THE DEVICE (finish.hip).  Out: exactly the mem_alnreg_v contents the reference
 * hands to worker_sam.  bm2_batch_finish works on the regs the last bm2_batch_run left in HBM; bm2_batch_download_alnregs copies the
 * result (hits of read i = out[aln_off[i] .. aln_off[i+1])); bm2_finish_regs_dev takes the hits of any producer as host arrays
 * (regs grouped by read: reg_off has n_reads+1 entries). */
int bm2_batch_finish(bm2_ctx *c, const bm2_opt *opt);
int bm2_batch_download_alnregs(bm2_ctx *c, bm2_alnreg_t *out, int64_t cap, int64_t *aln_off, int64_t *n_out);
int bm2_finish_regs_dev(bm2_ctx *c, const bm2_opt *opt, const bm2_reads *reads, const bm2_reg_t *regs, const int64_t *reg_off,
                        bm2_alnreg_t *out, int64_t cap, int64_t *out_off, int64_t *n_out);

/* ---- one chunk over several contexts (several GPUs of a node with a replica each, or contexts sharing one replica): the chunk is
 * cut at multiples of 512 reads (the kt_for block, the only cross-read rule of the path: bwamem.cpp:834), the parts run through
 * bm2_batch_upload / run / finish on one host thread per context, and the hits come back in read order -- ready for ONE
 * bm2_sam_pe[_dev] over the whole chunk (mem_pestat is chunk-wide, bwamem.cpp:1375).  No collective.  The result does not depend on
 * n_ctx. */
int bm2_chunk_hits_sharded(bm2_ctx *const *ctxs, int n_ctx, const bm2_reads *reads, const bm2_opt *opt, bm2_alnreg_t *out, int64_t cap,
                           int64_t *aln_off, int64_t *n_out);

/* ---- host side, next row of SURVEY.md 8(f): single-end records of worker_sam (bwamem.cpp:1320-1335) =
 * mem_mark_primary_se (:1420-1465) + mem_reorder_primary5 (:1496-1519) + mem_reg2sam (:1521-1577) with mem_gen_alt
 * (bwamem_extra.cpp:130-183), mem_reg2aln (:1732-1805: mem_approx_mapq_se, bwa_gen_cigar2 -> ksw_global2 with backtrack,
 * NM / MD) and mem_aln2sam (:1592-1730).  Text is byte-identical to the alignment lines `bwa-mem2 mem` prints for
 * single-end input (header lines are the caller's).  Pure host code. */
typedef struct {                    /* the fields of mem_opt_t (bwamem.h:74-110) this tail reads beyond bm2_opt */
    int32_t T;                      /* -T: minimum score to output (30) */
    int32_t flag;                   /* MEM_F_ALL 0x8, MEM_F_NO_MULTI 0x10, MEM_F_REF_HDR 0x100, MEM_F_SOFTCLIP 0x200,
                                     * MEM_F_PRIMARY5 0x800, MEM_F_KEEP_SUPP_MAPQ 0x1000; pairs: MEM_F_NOPAIRING 0x4, MEM_F_NO_RESCUE 0x20 */
    int32_t max_XA_hits, max_XA_hits_alt;   /* 5, 200 */
    float   XA_drop_ratio;          /* 0.80 */
    float   mapQ_coef_len;          /* 50 */
    int32_t mapQ_coef_fac;          /* (int)log(50) = 3: an int in the reference */
    int32_t pen_unpaired;           /* -U, 17 */
    int32_t max_ins;                /* 10000: pairs further apart are ignored by the insert-size statistics */
    int32_t max_matesw;             /* -m, 50: mate-rescue rounds per end */
    int32_t n_threads;              /* host threads for this tail; 0 = all hardware threads (the text does not depend on it) */
    int32_t rescue_inline;          /* 0: the mate-rescue alignments of a chunk are planned up front and run as one batch (the shape the
                                     * device kernel needs); 1: aligned inside the pair loop as mem_sam_pe does.  Same output. */
    const char *rg_id;              /* bwa_rg_id: RG:Z: value, NULL or "" = none */
} bm2_sam_opt;
void bm2_sam_opt_init(bm2_sam_opt *o);                  /* the defaults of mem_opt_init, bwamem.cpp:107-143 */

typedef struct {                    /* what bseq1_t carries besides the bases (kseq) */
    const char *const *name;        /* [n_reads] */
    const char *const *comment;     /* [n_reads] or NULL; an entry may be NULL (only printed with `mem -C`) */
    const char *const *qual;        /* [n_reads] or NULL; an entry may be NULL */
} bm2_read_text;

/* alnregs: the output of bm2_batch_finish / bm2_finish_regs_dev, regs of read i = [reg_off[i], reg_off[i+1]); they are reordered and annotated in
 * place exactly as mem_mark_primary_se does.  n_processed = reads of earlier chunks (it seeds the tie-breaking hash).
 * out/cap: caller's buffer; *n_out = bytes needed.  BM2_ECAP if cap is too small: NOTHING has been written to `out` then (the text is
 * formatted into per-thread buffers first and copied once every block's place is known) -- grow and call again with FRESH alnregs.
 * Process-wide settings: none unless the host asks.  BM2_MALLOC_TUNE=1 in the environment makes the first bm2_sam_* call set glibc's
 * M_TRIM_THRESHOLD / M_MMAP_THRESHOLD / M_TOP_PAD for the whole process (the tail's threads allocate small blocks at a high rate; see
 * sam_tail.cpp) -- the host's own allocations then stop returning memory to the OS as well, which is why it is the host's choice. */
int bm2_sam_se(const bm2_index_desc *idx, const bm2_opt *opt, const bm2_sam_opt *so, const bm2_reads *reads, const bm2_read_text *txt,
               bm2_alnreg_t *alnregs, const int64_t *reg_off, int64_t n_processed, char *out, int64_t cap, int64_t *n_out);

/* Counters of the last bm2_sam_pe call on this process (diagnostic, not synchronised between concurrent calls): rescue alignments
 * planned up front, those the pairs then used, and those a pair asked for that had not been planned (computed in place). */
void bm2_sam_rescue_stats(int64_t *planned, int64_t *used, int64_t *missed);

/* The local Smith-Waterman of mate rescue for a batch of (query, target) pairs: ksw_align2 (ksw.cpp:340-381) = a forward pass
 * (score, target end, query end, second-best score / its target end outside the best's neighbourhood) and, with KSW_XSTART, a
 * reverse pass for the start.  This is the HOST implementation (the SSE2 kernel's striping is observable and kept): the oracle of the
 * device kernel below and the code that aligns the few rescues a pair asks for outside its chunk's batch.  xtra = KSW_X* flags | minimum score, as mem_matesw builds it
 * (bwamem_pair.cpp:205).  out[i] = { score, te, qe, score2, te2, tb, qb }. */
typedef struct { int32_t score, te, qe, score2, te2, tb, qb; } bm2_ksw_result;
int bm2_ksw_align2(int32_t n, const uint8_t *seqs, const int64_t *q_off, const int32_t *q_len, const int64_t *t_off, const int32_t *t_len,
                   const int32_t *xtra, const int8_t mat[25], int o_del, int e_del, int o_ins, int e_ins, bm2_ksw_result *out);

/* The same on the device (matesw.hip: one task per 16-lane row, the lanes of the reference's SSE2 register).  seq_bytes = size of
 * seqs.  Results identical to bm2_ksw_align2. */
int bm2_ksw_align2_dev(bm2_ctx *c, int32_t n, const uint8_t *seqs, int64_t seq_bytes, const int64_t *q_off, const int32_t *q_len,
                       const int64_t *t_off, const int32_t *t_len, const int32_t *xtra, const int8_t mat[25], int o_del, int e_del,
                       int o_ins, int e_ins, bm2_ksw_result *out);

/* The @SQ lines of bwa_print_sam_hdr (bwa.cpp:523-556): one per contig, "\tAH:*" for ALT contigs; followed by hdr_line (the
 * caller's @RG / extra header lines, may be NULL) and a newline, as the reference prints them.  The @PG line carries the
 * command line and stays with the caller.  *n_out = bytes needed (BM2_ECAP when cap is smaller). */
int bm2_sam_header(const bm2_index_desc *idx, const char *hdr_line, char *out, int64_t cap, int64_t *n_out);

/* CIGAR generation for a batch of hits: bwa_gen_cigar2 (bwa.cpp:260-347) = banded global alignment with backtrack of the
 * query against reference [rb, re) (ksw_global2, ksw.cpp:558-668; both reversed first for hits on the reverse strand so that
 * gaps end up leftmost on the forward strand), NM and the MD string.  This is the HOST implementation: the oracle of the device
 * kernel below and the code behind bm2_sam_se / bm2_sam_pe (no GPU) and behind the CIGARs of rescued hits.  Task i: query codes seqs[q_off[i] .. +q_len[i]), reference range [rb[i], re[i]), band w[i].
 * Out: score[i], nm[i], n_cigar[i] (-1 = the reference returns NULL: empty or strand-bridging range), its ops at
 * cigar[cigar_off[i] ..] (BAM encoding len<<4|op) and the NUL-terminated MD at md[md_off[i] ..].  cigar_cap / md_cap are the
 * callers' capacities; BM2_ECAP with the needed sizes in *cigar_need / *md_need otherwise. */
int bm2_gen_cigar(const bm2_index_desc *idx, const bm2_opt *opt, int32_t n, const uint8_t *seqs, const int64_t *q_off, const int32_t *q_len,
                  const int64_t *rb, const int64_t *re, const int32_t *w, int32_t *score, int32_t *nm, int32_t *n_cigar,
                  int64_t *cigar_off, uint32_t *cigar, int64_t cigar_cap, int64_t *cigar_need,
                  int64_t *md_off, char *md, int64_t md_cap, int64_t *md_need);

/* The same on the device (cigar.hip: one task per lane, targets from the context's resident reference).  The context must have
 * been created with the index; seq_bytes = size of seqs.  Results identical to bm2_gen_cigar. */
int bm2_gen_cigar_dev(bm2_ctx *c, const bm2_opt *opt, int32_t n, const uint8_t *seqs, int64_t seq_bytes, const int64_t *q_off,
                      const int32_t *q_len, const int64_t *rb, const int64_t *re, const int32_t *w, int32_t *score, int32_t *nm,
                      int32_t *n_cigar, int64_t *cigar_off, uint32_t *cigar, int64_t cigar_cap, int64_t *cigar_need,
                      int64_t *md_off, char *md, int64_t md_cap, int64_t *md_need);

/* FASTA / FASTQ text -> packed reads + names / comments / qualities: kseq's record grammar (kseq.h:185-227), trim_readno
 * (bwa.cpp:62-66: a trailing "/<digit>" is cut off the name) and the nst_nt4_table base codes (bwamem.cpp:992-1000).  The
 * arrays are owned by the library until bm2_fastq_free; comment[i] / qual[i] are NULL when the record has none. */
typedef struct {
    int32_t n_reads; int32_t pad; int64_t n_bases;
    uint8_t *enc; int64_t *off; int32_t *len;       /* what bm2_reads points at */
    char **name, **comment, **qual;                  /* what bm2_read_text points at */
    char *arena;                                     /* internal: non-NULL when the strings share one allocation */
} bm2_fastq;
int  bm2_fastq_parse(const char *text, int64_t n_bytes, bm2_fastq *out);
/* The reader of a whole chunk on n_threads host threads (<= 0: all): one file (text2 == NULL) or two files whose records are
 * interleaved 2i, 2i+1 as bseq_read_orig delivers paired input (bwa.cpp:170-216; the shorter file ends the input).  Same result as
 * bm2_fastq_parse on each file; strict four-line FASTQ is scanned in parallel, anything else falls back to the sequential parser. */
int  bm2_fastq_parse_mt(const char *text1, int64_t n1, const char *text2, int64_t n2, int n_threads, bm2_fastq *out);
void bm2_fastq_free(bm2_fastq *f);

/* Paired-end chunks (reads interleaved: 2i, 2i+1): mem_pestat over the chunk (bwamem_pair.cpp:81-148) unless pes_in is
 * given, then per pair mem_sam_pe (:353-551): mate rescue (mem_matesw :150-283 -> ksw_align2, ksw.cpp:340-381), mem_pair
 * (:285-346), mapping qualities, the records of both ends.  alnregs are not modified.  pes_out (optional) receives the four
 * orientation models (FF, FR, RF, RR). */
typedef struct { int32_t low, high, failed, pad; double avg, std; } bm2_pestat;      /* mem_pestat_t, bwamem.h:162-166 */
int bm2_sam_pe(const bm2_index_desc *idx, const bm2_opt *opt, const bm2_sam_opt *so, const bm2_reads *reads, const bm2_read_text *txt,
               const bm2_alnreg_t *alnregs, const int64_t *reg_off, int64_t n_processed, const bm2_pestat *pes_in, bm2_pestat *pes_out,
               char *out, int64_t cap, int64_t *n_out);

/* bm2_sam_pe / bm2_sam_se with the chunk's mate-rescue alignments (PE) and CIGAR alignments run as device batches against the
 * context's resident reference (the context must have been created with this index).  Same output.  bm2_sam_cigar_stats: CIGAR
 * alignments planned by the dry pass / looked up / computed in place by the last call. */
int bm2_sam_se_dev(bm2_ctx *c, const bm2_index_desc *idx, const bm2_opt *opt, const bm2_sam_opt *so, const bm2_reads *reads,
                   const bm2_read_text *txt, bm2_alnreg_t *alnregs, const int64_t *reg_off, int64_t n_processed, char *out, int64_t cap,
                   int64_t *n_out);
void bm2_sam_cigar_stats(int64_t *planned, int64_t *used, int64_t *missed);
int bm2_sam_pe_dev(bm2_ctx *c, const bm2_index_desc *idx, const bm2_opt *opt, const bm2_sam_opt *so, const bm2_reads *reads,
                   const bm2_read_text *txt, const bm2_alnreg_t *alnregs, const int64_t *reg_off, int64_t n_processed,
                   const bm2_pestat *pes_in, bm2_pestat *pes_out, char *out, int64_t cap, int64_t *n_out);
/* The same over SEVERAL contexts (one per GPU, or contexts sharing one replica): the chunk's rescue batch and CIGAR batch are cut into
 * contiguous parts, one context and one host thread per part, results back in task order -- same text whatever n_ctx is.  What a host that
 * shards the hot path with bm2_chunk_hits_sharded calls afterwards, so that the tail's device work shrinks with the number of GPUs too
 * (worker_sam, bwamem.cpp:1366-1381; mem_matesw, bwamem_pair.cpp:150-283). */
int bm2_sam_se_dev_multi(bm2_ctx *const *ctxs, int n_ctx, const bm2_index_desc *idx, const bm2_opt *opt, const bm2_sam_opt *so, const bm2_reads *reads,
                         const bm2_read_text *txt, bm2_alnreg_t *alnregs, const int64_t *reg_off, int64_t n_processed, char *out, int64_t cap,
                         int64_t *n_out);
int bm2_sam_pe_dev_multi(bm2_ctx *const *ctxs, int n_ctx, const bm2_index_desc *idx, const bm2_opt *opt, const bm2_sam_opt *so, const bm2_reads *reads,
                         const bm2_read_text *txt, const bm2_alnreg_t *alnregs, const int64_t *reg_off, int64_t n_processed,
                         const bm2_pestat *pes_in, bm2_pestat *pes_out, char *out, int64_t cap, int64_t *n_out);


/* ---- the same path split so that a caller can keep inputs resident in HBM and time only the device work */
int bm2_batch_upload(bm2_ctx *c, const bm2_reads *reads);                 /* H2D (pinned staging) */
int bm2_batch_run(bm2_ctx *c, const bm2_opt *opt);                         /* device only, returns after sync */
int bm2_batch_stats(bm2_ctx *c, bm2_stats *stats);
int bm2_batch_download(bm2_ctx *c, bm2_reg_t *regs, int64_t cap, int64_t *reg_off, int64_t *n_out);
/* wall time of the last bm2_batch_run measured with hipEvents on the library's stream, per kernel: SUMMED over the parts of the chunk */
int bm2_batch_kernel_ms(bm2_ctx *c, float *ms, int32_t cap, int32_t *n_out, const char **names);
/* parts the last uploaded chunk was cut into (at multiples of 512 reads, the one cross-read rule of the path: bwamem.cpp:834): each part
 * runs on streams and a workspace of its own, beside the others (launch policy BM2_N_SUB, default 1; a chunk below 64 blocks per part: 1) */
int bm2_batch_parts(const bm2_ctx *c);
/* diagnostic: copy a raw device array of the last bm2_batch_run to the host ("smem", "sa_coord", "chn", "seeds",
 * "regs_raw", ... see pipeline.hip); lets tests pin every stage against the oracle (level ii) */
int bm2_batch_fetch(bm2_ctx *c, const char *what, void *out, int64_t cap_bytes, int64_t *n_bytes);

#ifdef __cplusplus
}
#endif
#endif
