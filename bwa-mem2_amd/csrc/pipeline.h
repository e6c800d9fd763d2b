// pipeline.h -- device-side records and launch prototypes of the seed -> chain -> extend pipeline (internal).
#pragma once
#include "bm2_ctx.h"

struct __attribute__((aligned(16))) StSmem {       // staged SMEM of one read (rid implicit); 32 bytes
    int64_t k, l, s;
    int32_t m, n;
};

struct SeedParams {
    int32_t min_seed_len, split_len, split_width, max_occ;
    int64_t max_mem_intv;
};

struct ChainParams {                                // what mem_chain_seeds / mem_chain_flt / mem_chain2aln read from mem_opt_t
    int32_t a, o_del, e_del, o_ins, e_ins, w, max_chain_gap, max_occ, min_seed_len, min_chain_weight, max_chain_extend;
    int32_t pen_clip5, pen_clip3, zdrop;
    float mask_level, drop_ratio;
};

// scan.hip
int bm2_scan_i32(bm2_ctx *c, const int32_t *in, int64_t n, int64_t *out_excl /* n+1 */, DevBuf &tmp);

int bm2_perm_by_work(bm2_ctx *c, int n, const int32_t *key, int32_t *perm, uint32_t *hist32, int mode);

int bm2_partition_by_work(bm2_ctx *c, int n, const int32_t *key, int thr, int32_t *perm, DevBuf &tmp, DevBuf &scan_tmp);

// smem.hip
int bm2_launch_smem(bm2_ctx *c, const SeedParams &sp, int n_reads, const uint8_t *enc, const int64_t *off, const int32_t *len,
                    StSmem *stage, StSmem *prevbuf, int stage_cap, int prev_cap, int grid, bm2_smem_t *out, int64_t out_cap,
                    int32_t *smem_cnt, int64_t *smem_off, int32_t *occ_cnt, unsigned long long *counters);
int bm2_launch_sal_expand(bm2_ctx *c, const bm2_smem_t *smems, int64_t n_smem, const int64_t *sa_off, int32_t max_occ, int64_t *pos);
int bm2_launch_sal(bm2_ctx *c, int64_t n, int64_t *pos_coord, unsigned long long *n_lf);
int bm2_launch_smem_gather(bm2_ctx *c, int n_reads, const bm2_smem_t *in, const int64_t *in_off, const int32_t *cnt,
                           const int64_t *out_off, bm2_smem_t *out);
