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
    int32_t reg_nodes;                              // launch policy, not an option: B-tree nodes in global memory are visited through registers (chain.hip)
};

// scan.hip
int bm2_scan_i32(bm2_ctx *c, const int32_t *in, int64_t n, int64_t *out_excl /* n+1 */, DevBuf &tmp);

int bm2_perm_by_work(bm2_ctx *c, int n, const int32_t *key, int32_t *perm, uint32_t *hist32, int mode);
int bm2_partition_by_class(bm2_ctx *c, int n, const int32_t *key, int thr, int32_t *perm, DevBuf &tmp, DevBuf &scan_tmp, const int64_t **n_heavy_dev);

int bm2_partition_by_work(bm2_ctx *c, int n, const int32_t *key, int thr, int32_t *perm, DevBuf &tmp, DevBuf &scan_tmp, int heavy_first = 0,
                          const int64_t **n_heavy_dev = nullptr);
int bm2_pf_heavy_threshold();

// smem.hip
struct BHead; struct P2Task;
struct SeedBufs {                                   // workspace of the seeding task kernels (smem.hip)
    BHead *heads1, *heads2; uint4 *ents1, *ents2; int64_t slot1_cap, slot2_cap;
    uint4 *pool; int pool_cap, pool_slots;
    bm2_smem_t *recs; int64_t rec_cap;
    P2Task *tasks; int64_t task_cap;
    int32_t *heavy1, *heavy2; int64_t heavy_cap;     // slot ids of the long-list tasks of pass 1 / pass 2
    void *cont1, *cont2; int64_t cont_cap;           // the tasks k_bwd handed over at a row boundary (CTask records, BM2_BWD_EXPORT_AGE)
};
enum { BM2_SC_SLOT1 = 1, BM2_SC_REC = 3, BM2_SC_TASK = 4, BM2_SC_SLOT2 = 6, BM2_SC_NEXT = 9, BM2_SC_OVF = 10, BM2_SC_POOL = 11,
       BM2_SC_NEXT_W1 = 12 /* then W2, W3, B1, B2 */ };   // = the SC_* of smem.hip
int bm2_launch_seeding(bm2_ctx *c, const SeedParams &sp, int n_reads, const uint8_t *enc, const int64_t *off, const int32_t *len,
                       const SeedBufs &sb, int grid_walk, int grid_bwd, int32_t *smem_cnt, unsigned long long *sc,
                       void (*tick)(bm2_ctx *, const char *), int max_len);
int bm2_launch_smem_finish(bm2_ctx *c, int n_reads, const SeedBufs &sb, const unsigned long long *sc, const int32_t *smem_cnt,
                           const int64_t *smem_off, int32_t *fill, bm2_smem_t *tmp, int32_t max_occ, bm2_smem_t *out, int32_t *occ_cnt, int max_len);
int bm2_seed_sizes(size_t *head, size_t *ent, size_t *task, int *n_sc, size_t *ctask);      // returns CAPF
int bm2_launch_sal_expand(bm2_ctx *c, const bm2_smem_t *smems, int64_t n_smem, const int64_t *sa_off, int32_t max_occ, int64_t *pos);
int bm2_launch_sal(bm2_ctx *c, int64_t n, int64_t *pos_coord, unsigned long long *n_lf);
int bm2_launch_smem_gather(bm2_ctx *c, int n_reads, const bm2_smem_t *in, const int64_t *in_off, const int32_t *cnt,
                           const int64_t *out_off, bm2_smem_t *out);
