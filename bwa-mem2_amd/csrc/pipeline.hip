// pipeline.hip -- host orchestration of the device pipeline and the S2/S3 entry points of include/bm2.h.
//
// A chunk of reads stays in HBM from upload to the final regs: k_smem -> scan -> k_sal_expand -> k_sal -> k_chain ->
// k_slot_base -> k_extend -> k_postfilter -> scan -> k_reg_gather, all on one stream.  The host only reads back three
// scalars (SMEM count, SA count, reg count) to size the next stage's buffers.
#include <string.h>
#include <stdio.h>
#include <math.h>
#include <stdlib.h>
#include <vector>
#include "pipeline.h"
#include "chain_dev.h"

// launchers defined in chain.hip / extend.hip
int bm2_launch_chain(bm2_ctx *c, const ChainParams &o, int n_reads, const int32_t *len, const bm2_smem_t *smems,
                     const int32_t *smem_cnt, const int64_t *smem_off, const int64_t *sa_off, const int64_t *sa_coord,
                     WChain *wchain, WSeed *wseed, BtNode *nodes, int32_t *order, DevChain *chn, DevSeed *seeds_out,
                     int32_t *seed_owner,
                     int32_t *n_chain_out, int32_t *n_reg_out, int32_t *n_chain0_out, const int32_t *perm,
                     int heavy_thr, const int64_t *n_heavy_dev, const int32_t *n_sa_read, unsigned long long *item_cur, int max_len, int32_t *isl_cut,
                     const int32_t *isl_order, int32_t *isl_serial, const FinishOut *fuse);
int bm2_launch_chain_finish(bm2_ctx *c, const ChainParams &o, int n_reads, const int32_t *len, const int64_t *read_base,
                            const int32_t *n_chain, DevChain *chn, DevSeed *seeds_out, int32_t *srt_out, int32_t *reg_seed,
                            int32_t *reg_chain, int32_t *n_reg_out, const int32_t *perm, bool lanes_by_perm, const int32_t *n_sa_read, int done_thr, const int64_t *n_heavy_dev, int wave_thr);
int bm2_launch_seed_filter(bm2_ctx *c, const ChainParams &o, const int8_t *d_mat25, int n_reads, int64_t n_slots, const uint8_t *enc,
                           const int64_t *off, const int32_t *len, const int32_t *min_hsp, const int64_t *read_base, const int32_t *n_chain,
                           const int32_t *seed_owner, DevChain *chn, DevSeed *seeds, uint8_t *seed_keep);
int bm2_launch_extend(bm2_ctx *c, const bm2_opt &opt, const ChainParams &cp, int n_reads, int64_t n_slots, const uint8_t *enc,
                      const int64_t *off, const int32_t *len, const int64_t *read_base, const int32_t *n_chain, const int32_t *n_reg,
                      const int64_t *slot_base, const int32_t *reg_seed, const int32_t *reg_chain, const DevChain *chn,
                      const DevSeed *seeds, int32_t *srt_all, DevReg *regs, unsigned long long *counters, DevBuf &tmp, DevBuf &scan_tmp, int32_t *cursor,
                      int max_len);
int bm2_launch_slot_base(bm2_ctx *c, int n_reads, const int64_t *read_base, const int32_t *n_reg, int64_t *slot_base);
int bm2_launch_postfilter(bm2_ctx *c, const ChainParams &o, int n_reads, const int32_t *len, const int64_t *read_base,
                          const int32_t *n_chain, const int32_t *n_reg, const DevChain *chn, const DevSeed *seeds,
                          int32_t *srt_all, DevReg *regs, int32_t *n_out, const int32_t *cursor, const int32_t *perm,
                          const int32_t *heavy, const int64_t *n_heavy, unsigned long long *item_cur, int pf_heavy);
int bm2_launch_reg_gather(bm2_ctx *c, int n_reads, const int64_t *read_base, const int32_t *n_reg, const DevReg *regs,
                          const int64_t *out_off, bm2_reg_t *out, int64_t out_cap);

int bm2_run_finish(bm2_ctx *c, const bm2_opt *opt, int n_reads, const uint8_t *enc, const int64_t *off, const bm2_reg_t *regs,
                   const int64_t *reg_off, int64_t n_regs, DevBuf &work, DevBuf &ordb, DevBuf &stateb, DevBuf &nfin, DevBuf &finoff, DevBuf &reqb,
                   DevBuf &cntb, DevBuf &out, DevBuf &scan_tmp, int64_t *n_out, int *rounds);     // finish.hip

struct Batch {
    int n_reads = 0, max_len = 0;
    int64_t n_bases = 0;
    bool uploaded = false, ran = false;
    // inputs
    DevBuf enc, off, len;
    // seeding
    DevBuf stage, prevbuf, smem, occ_cnt, smem_cnt, smem_off, counters, sa_off, sa_coord, scan_tmp, read_base;
    // chaining / extension
    DevBuf wchain, wseed, nodes, order, chn, seeds, srt, reg_seed, reg_chain, regs, slot_base, n_chain, n_reg, n_chain0, n_out;
    DevBuf out_off, out_regs, smem_sorted, smem_sorted_off, ext_tmp, cursor, n_sa_read, perm, perm_hist, part_tmp;
    int64_t n_smem = 0, n_sa = 0, n_out_regs = 0;
    bm2_stats stats{};
    std::vector<int32_t> h_len;              // host copy of the read lengths (per-read filter thresholds)
    DevBuf min_hsp, seed_owner, seed_keep, mat25;
    DevBuf perm2, part_tmp2, isl_serial;
    DevBuf heads1, ents1, heads2, ents2, pool, recs, tasks, seedc, fill, smem_tmp, heavy1, heavy2, cont1, cont2;     // seeding task kernels
    DevBuf fin_work, fin_ord, fin_state, fin_n, fin_off, fin_req, fin_cnt, fin_out;                     // hit finishing (finish.hip)
    int64_t n_fin = -1; int fin_rounds = 0;     // -1: bm2_batch_finish has not run on the current regs
    int seed_attempts = 0;                      // runs of the seeding kernels the last batch needed (> 1: a workspace grew)
    int64_t seed_cap[5] = { 0, 0, 0, 0, 0 };   // learned workspace sizes: slots pass 1/2, records, pass-2 tasks, pool lists
};

void bm2_batch_destroy(bm2_ctx *c) {
    if (!c->batch) return;
    Batch *b = c->batch;
    DevBuf *all[] = { &b->enc, &b->off, &b->len, &b->stage, &b->prevbuf, &b->smem, &b->occ_cnt, &b->smem_cnt, &b->smem_off,
                      &b->counters, &b->sa_off, &b->sa_coord, &b->scan_tmp, &b->read_base, &b->wchain, &b->wseed, &b->nodes,
                      &b->order, &b->chn, &b->seeds, &b->srt, &b->reg_seed, &b->reg_chain, &b->regs, &b->slot_base, &b->n_chain,
                      &b->n_reg, &b->n_chain0, &b->n_out, &b->out_off, &b->out_regs, &b->smem_sorted, &b->smem_sorted_off, &b->ext_tmp, &b->cursor, &b->n_sa_read, &b->perm, &b->perm_hist, &b->part_tmp, &b->min_hsp, &b->seed_owner, &b->seed_keep, &b->mat25,
                      &b->heads1, &b->ents1, &b->heads2, &b->ents2, &b->pool, &b->recs, &b->tasks, &b->seedc, &b->fill, &b->smem_tmp,
                      &b->heavy1, &b->heavy2, &b->cont1, &b->cont2, &b->perm2, &b->part_tmp2, &b->isl_serial,
                      &b->fin_work, &b->fin_ord, &b->fin_state, &b->fin_n, &b->fin_off, &b->fin_req, &b->fin_cnt, &b->fin_out };
    for (DevBuf *d : all) bm2_release(*d);
    delete b;
    c->batch = nullptr;
}

static Batch *get_batch(bm2_ctx *c) {
    if (!c->batch) c->batch = new Batch();
    return c->batch;
}

static void tick(bm2_ctx *c, const char *name) {       // event after the stage `name`
    if (c->n_ev < BM2_MAX_TIMERS) {
        c->ev_name[c->n_ev] = name;
        (void)hipEventRecord(c->ev[c->n_ev + 1], c->stream);
        c->n_ev++;
    }
}

__global__ void k_read_base(int n_reads, const int32_t *__restrict__ smem_cnt, const int64_t *__restrict__ smem_off,
                            const int64_t *__restrict__ sa_off, int64_t *read_base, int32_t *n_sa_read) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    const int c = smem_cnt[r];
    read_base[r] = c > 0 ? sa_off[smem_off[r]] : 0;
    n_sa_read[r] = c > 0 ? (int32_t)(sa_off[smem_off[r] + c] - sa_off[smem_off[r]]) : 0;
}

static int batch_upload_one(bm2_ctx *c, const bm2_reads *reads) {
    if (!c || !reads || reads->n_reads < 0) { bm2_set_error("bm2_batch_upload: bad argument"); return BM2_EINVAL; }
    if (!c->has_index) { bm2_set_error("context was created without an index"); return BM2_EINVAL; }
    int rc = bm2_check(hipSetDevice(c->device), "hipSetDevice");
    if (rc) return rc;
    Batch *b = get_batch(c);
    const int n = reads->n_reads;
    int64_t nb = 0; int max_len = 0;
    for (int i = 0; i < n; i++) {
        if (reads->len[i] < 0 || reads->off[i] < 0) { bm2_set_error("read %d: negative length/offset", i); return BM2_EINVAL; }
        if (reads->len[i] >= 32768) { bm2_set_error("read %d is %d bp: the reference path handles reads < 32768 bp (bandedSWA.h:83)", i, reads->len[i]); return BM2_EUNSUP; }
        int64_t e = reads->off[i] + reads->len[i];
        if (e > nb) nb = e;
        if (reads->len[i] > max_len) max_len = reads->len[i];
    }
    b->n_reads = n; b->n_bases = nb; b->max_len = max_len; b->ran = false;
    b->h_len.assign(reads->len, reads->len + n);
    if ((rc = bm2_reserve(b->enc, (size_t)nb + 64))) return rc;
    if ((rc = bm2_reserve(b->off, (size_t)(n + 1) * 8))) return rc;
    if ((rc = bm2_reserve(b->len, (size_t)(n + 1) * 4))) return rc;
    if (n) {
        rc = bm2_copy_h2d(c, b->enc.p, reads->enc, (size_t)nb);
        if (!rc) rc = bm2_check(hipMemcpyAsync(b->off.p, reads->off, (size_t)n * 8, hipMemcpyHostToDevice, c->stream), "H2D off");
        if (!rc) rc = bm2_check(hipMemcpyAsync(b->len.p, reads->len, (size_t)n * 4, hipMemcpyHostToDevice, c->stream), "H2D len");
        if (!rc) rc = bm2_check(hipStreamSynchronize(c->stream), "upload sync");
    }
    b->uploaded = rc == 0;
    return rc;
}

static SeedParams seed_params(const bm2_opt *opt) {
    SeedParams sp;
    sp.min_seed_len = opt->min_seed_len;
    sp.split_len = (int)(opt->min_seed_len * opt->split_factor + .499);      // bwamem.cpp:639
    sp.split_width = opt->split_width; sp.max_occ = opt->max_occ; sp.max_mem_intv = opt->max_mem_intv;
    return sp;
}
static ChainParams chain_params(const bm2_opt *opt) {
    ChainParams o;
    o.a = opt->a; o.o_del = opt->o_del; o.e_del = opt->e_del; o.o_ins = opt->o_ins; o.e_ins = opt->e_ins; o.w = opt->w;
    o.max_chain_gap = opt->max_chain_gap; o.max_occ = opt->max_occ; o.min_seed_len = opt->min_seed_len;
    o.min_chain_weight = opt->min_chain_weight; o.max_chain_extend = opt->max_chain_extend;
    o.pen_clip5 = opt->pen_clip5; o.pen_clip3 = opt->pen_clip3; o.zdrop = opt->zdrop;
    o.mask_level = opt->mask_level; o.drop_ratio = opt->drop_ratio;
    o.reg_nodes = bm2_knob("BM2_CHAIN_REGNODES", 1);
    return o;
}

static int check_opt(const bm2_opt *opt) {
    if (!opt) return BM2_EINVAL;
    if (opt->e_del <= 0 || opt->e_ins <= 0 || opt->a <= 0 || opt->w <= 0 || opt->max_occ <= 0 || opt->min_seed_len <= 0) {
        bm2_set_error("bm2: option out of range (a, w, max_occ, min_seed_len, e_del, e_ins must be > 0)");
        return BM2_EINVAL;
    }
    // the kernels score with (match, mismatch, ambiguous) = (mat[0], mat[1], mat[4]), the structure bwa_fill_scmat builds
    // (bwa.cpp:248-257) and the only one the reference's SIMD kernels implement (bandedSWA.cpp:286-290)
    for (int i = 0; i < 5; i++) for (int j = 0; j < 5; j++) {
        const int want = (i == 4 || j == 4) ? opt->mat[4] : (i == j ? opt->mat[0] : opt->mat[1]);
        if (opt->mat[i * 5 + j] != want) { bm2_set_error("bm2: scoring matrix is not of the bwa_fill_scmat form"); return BM2_EUNSUP; }
    }
    return BM2_OK;
}

// seeding stages: SMEMs (bump order) + SA coordinates
static int run_seeding(bm2_ctx *c, Batch *b, const bm2_opt *opt, bool with_sal) {
    const int n = b->n_reads;
    int rc;
    hipStream_t s = c->stream;
    const SeedParams sp = seed_params(opt);
    if ((rc = bm2_reserve(b->counters, 56 * 8))) return rc;      // ([40..42]: work cursors of the chain stage's tiers beyond the fifth; [48]: the island kernel's wavefronts that have left)
    if ((rc = bm2_check(hipMemsetAsync(b->counters.p, 0, 56 * 8, s), "memset counters"))) return rc;
    if ((rc = bm2_reserve(b->smem_cnt, (size_t)(n + 1) * 4))) return rc;
    if ((rc = bm2_reserve(b->smem_off, (size_t)(n + 1) * 8))) return rc;
    // task kernels with persistent lanes (smem.hip); workspace sizes are learned: a run that overflows one of them reports
    // what it needed and is repeated
    const int bpc_w = bm2_knob("BM2_WALK_BLOCKS_PER_CU", 4);
    const int bpc_b = bm2_knob("BM2_BWD_BLOCKS_PER_CU", 3);
    const int grid_w = c->n_cu * bpc_w, grid_b = c->n_cu * bpc_b;
    const int64_t lanes = (int64_t)(grid_w > grid_b ? grid_w : grid_b) * 256;
    if (opt->split_width > 65534) { bm2_set_error("split_width %d > 65534 is not supported", opt->split_width); return BM2_EUNSUP; }
    size_t head_sz, ent_sz, task_sz, ctask_sz; int n_sc;
    const int capf = bm2_seed_sizes(&head_sz, &ent_sz, &task_sz, &n_sc, &ctask_sz);
    SeedBufs sb;
    sb.pool_cap = b->max_len + 2 > capf ? b->max_len + 2 - capf : 1;
    // (the learned sizes are kept as counts: deriving them from the buffers' byte capacities would make every buffer chase
    //  the slack of the others)
    int64_t &slot1_cap = b->seed_cap[0], &slot2_cap = b->seed_cap[1], &rec_cap = b->seed_cap[2], &task_cap = b->seed_cap[3],
            &pool_slots = b->seed_cap[4];
    const int64_t per_read = getenv("BM2_SEED_TINY") ? 0 : 1;   // test hook: start from workspaces that only hold the pool tails
    if (!per_read) { slot1_cap = slot2_cap = rec_cap = task_cap = pool_slots = 0; }
    if (slot1_cap < (int64_t)n * 6 * per_read + lanes * 4 + 4096) slot1_cap = (int64_t)n * 6 * per_read + lanes * 4 + 4096;
    if (slot2_cap < (int64_t)n * 6 * per_read + lanes * 4 + 4096) slot2_cap = (int64_t)n * 6 * per_read + lanes * 4 + 4096;
    if (rec_cap < (int64_t)n * 24 * per_read + lanes * 12 + 4096) rec_cap = (int64_t)n * 24 * per_read + lanes * 12 + 4096;      // + the pool tails: 3 kernels x 256 / wave
    if (task_cap < (int64_t)n * 6 * per_read + lanes + 4096) task_cap = (int64_t)n * 6 * per_read + lanes + 4096;
    if (pool_slots < (n / 8 + 1024) * per_read + 1) pool_slots = (n / 8 + 1024) * per_read + 1;
    if ((rc = bm2_reserve(b->seedc, (size_t)n_sc * 8))) return rc;
    if ((rc = bm2_reserve(b->fill, (size_t)(2 * (size_t)n + 8) * 4))) return rc;      // per-read fill counters | list of SMEM-rich reads | its count, cursor
    std::vector<unsigned long long> h_sc((size_t)n_sc);
    int64_t n_smem_tot = 0;
    const bool verbose = getenv("BM2_VERBOSE") != nullptr;
    auto now_ms = []() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; };
    double t0 = now_ms();
    if (verbose) (void)hipStreamSynchronize(s);
    for (int attempt = 0; ; attempt++) {
        if ((rc = bm2_reserve(b->heads1, (size_t)slot1_cap * head_sz))) return rc;
        if ((rc = bm2_reserve(b->ents1, (size_t)slot1_cap * ent_sz))) return rc;
        if ((rc = bm2_reserve(b->heads2, (size_t)slot2_cap * head_sz))) return rc;
        if ((rc = bm2_reserve(b->ents2, (size_t)slot2_cap * ent_sz))) return rc;
        if ((rc = bm2_reserve(b->recs, (size_t)rec_cap * sizeof(bm2_smem_t)))) return rc;
        if ((rc = bm2_reserve(b->tasks, (size_t)task_cap * task_sz))) return rc;
        if ((rc = bm2_reserve(b->pool, (size_t)pool_slots * sb.pool_cap * 16))) return rc;
        sb.heads1 = (BHead *)b->heads1.p; sb.ents1 = (uint4 *)b->ents1.p; sb.slot1_cap = slot1_cap;
        sb.heads2 = (BHead *)b->heads2.p; sb.ents2 = (uint4 *)b->ents2.p; sb.slot2_cap = slot2_cap;
        sb.pool = (uint4 *)b->pool.p; sb.pool_slots = (int)pool_slots;
        sb.recs = (bm2_smem_t *)b->recs.p; sb.rec_cap = rec_cap; sb.tasks = (P2Task *)b->tasks.p; sb.task_cap = task_cap;
        sb.heavy_cap = (int64_t)n / 4 + lanes + 4096;           // (a full list only means the task stays lane-per-task)
        if ((rc = bm2_reserve(b->heavy1, (size_t)sb.heavy_cap * 4))) return rc;
        if ((rc = bm2_reserve(b->heavy2, (size_t)sb.heavy_cap * 4))) return rc;
        sb.heavy1 = (int32_t *)b->heavy1.p; sb.heavy2 = (int32_t *)b->heavy2.p;
        sb.cont_cap = (int64_t)n / 2 + lanes + 4096;            // (a full list only means the task stays with its lane)
        if ((rc = bm2_reserve(b->cont1, (size_t)sb.cont_cap * ctask_sz))) return rc;
        if ((rc = bm2_reserve(b->cont2, (size_t)sb.cont_cap * ctask_sz))) return rc;
        sb.cont1 = b->cont1.p; sb.cont2 = b->cont2.p;
        if ((rc = bm2_check(hipMemsetAsync(b->seedc.p, 0, (size_t)n_sc * 8, s), "memset seed cursors"))) return rc;
        if ((rc = bm2_check(hipMemsetAsync(b->smem_cnt.p, 0, (size_t)(n + 1) * 4, s), "memset smem_cnt"))) return rc;
        if ((rc = bm2_check(hipMemsetAsync(b->fill.p, 0, (size_t)(2 * (size_t)n + 8) * 4, s), "memset fill"))) return rc;
        if (verbose) { (void)hipStreamSynchronize(s); fprintf(stderr, "[seeding] reserve+memset %.1f ms\n", now_ms() - t0); t0 = now_ms(); }
        c->n_ev = 0;                                                // (a repeated attempt restarts the stage clock)
        if ((rc = bm2_launch_seeding(c, sp, n, (const uint8_t *)b->enc.p, (const int64_t *)b->off.p, (const int32_t *)b->len.p, sb,
                                     grid_w, grid_b, (int32_t *)b->smem_cnt.p, (unsigned long long *)b->seedc.p, tick, b->max_len))) return rc;
        if ((rc = bm2_scan_i32(c, (const int32_t *)b->smem_cnt.p, n, (int64_t *)b->smem_off.p, b->scan_tmp))) return rc;
        if ((rc = bm2_check(hipMemcpyAsync(&n_smem_tot, (int64_t *)b->smem_off.p + n, 8, hipMemcpyDeviceToHost, s), "D2H n_smem"))) return rc;
        if ((rc = bm2_check(hipMemcpyAsync(h_sc.data(), b->seedc.p, (size_t)n_sc * 8, hipMemcpyDeviceToHost, s), "D2H seed cursors"))) return rc;
        if ((rc = bm2_check(hipStreamSynchronize(s), "seeding kernels"))) return rc;
        if (verbose) { fprintf(stderr, "[seeding] kernels+scan %.1f ms\n", now_ms() - t0); t0 = now_ms(); }
        b->seed_attempts = attempt + 1;
        if (!h_sc[BM2_SC_OVF]) break;
        if (getenv("BM2_VERBOSE") || getenv("BM2_WARN_RETRY"))
            fprintf(stderr, "[seeding] attempt %d overflowed (flags %llu): slots %llu/%lld + %llu/%lld, records %llu/%lld, tasks %llu/%lld, pool %llu/%lld\n",
                    attempt, h_sc[BM2_SC_OVF], h_sc[BM2_SC_SLOT1], (long long)slot1_cap, h_sc[BM2_SC_SLOT2], (long long)slot2_cap,
                    h_sc[BM2_SC_REC], (long long)rec_cap, h_sc[BM2_SC_TASK], (long long)task_cap, h_sc[BM2_SC_POOL], (long long)pool_slots);
        if (attempt == 5) { bm2_set_error("seeding workspace could not be sized (flags %llu)", h_sc[BM2_SC_OVF]); return BM2_ENOMEM; }
        auto grow = [](int64_t &cap, int64_t need) { const int64_t want = need + need / 4 + 4096; if (cap < want) cap = want; };
        grow(slot1_cap, (int64_t)h_sc[BM2_SC_SLOT1]); grow(slot2_cap, (int64_t)h_sc[BM2_SC_SLOT2]);
        grow(rec_cap, (int64_t)h_sc[BM2_SC_REC]); grow(task_cap, (int64_t)h_sc[BM2_SC_TASK]);
        grow(pool_slots, (int64_t)h_sc[BM2_SC_POOL]);
    }
    if (getenv("BM2_SMEM_PROF")) {
        static const char *nm[5] = { "walk P1", "walk P2", "walk P3", "bwd 1", "bwd 2" };
        for (int t = 0; t < 5; t++) {
            const unsigned long long rounds = h_sc[n_sc - 10 + 2 * t], act = h_sc[n_sc - 10 + 2 * t + 1];
            fprintf(stderr, "[seeding] %-8s wave rounds %10llu, extensions %11llu (%.1f%% of 64 lanes)\n", nm[t], rounds, act,
                    rounds ? 100.0 * act / (64.0 * rounds) : 0.0);
        }
        fprintf(stderr, "[seeding] slots %llu + %llu, records %llu, pass-2 tasks %llu, pool lists %llu\n", h_sc[BM2_SC_SLOT1],
                h_sc[BM2_SC_SLOT2], h_sc[BM2_SC_REC], h_sc[BM2_SC_TASK], h_sc[BM2_SC_POOL]);
    }
    unsigned long long h_cnt[3] = { (unsigned long long)n_smem_tot, h_sc[BM2_SC_NEXT], 0 };
    static_assert(BM2_SC_NEXT_W1 + 15 == 27, "bm2_batch_fetch(\"seed_counters\") exposes 27 counters ([21], [22]: tasks handed over in pass 1 / 2 (counted exactly: wave_alloc_exact), [25], [26]: rows their continuations walked)");
    if ((rc = bm2_reserve(b->smem, (size_t)(n_smem_tot + 1) * sizeof(bm2_smem_t)))) return rc;
    if ((rc = bm2_reserve(b->smem_tmp, (size_t)(n_smem_tot + 1) * sizeof(bm2_smem_t)))) return rc;
    if ((rc = bm2_reserve(b->occ_cnt, (size_t)(n_smem_tot + 2) * 4))) return rc;
    if (verbose) { fprintf(stderr, "[seeding] output reserve %.1f ms (n_smem %lld, caps %zu %zu %zu)\n", now_ms() - t0, (long long)n_smem_tot, b->smem.cap, b->smem_tmp.cap, b->occ_cnt.cap); t0 = now_ms(); }
    if ((rc = bm2_launch_smem_finish(c, n, sb, (const unsigned long long *)b->seedc.p, (const int32_t *)b->smem_cnt.p,
                                     (const int64_t *)b->smem_off.p, (int32_t *)b->fill.p, (bm2_smem_t *)b->smem_tmp.p, sp.max_occ,
                                     (bm2_smem_t *)b->smem.p, (int32_t *)b->occ_cnt.p, b->max_len))) return rc;
    tick(c, "smem.finish");
    b->n_smem = (int64_t)h_cnt[0];
    b->stats.n_smem = b->n_smem; b->stats.n_ext = (int64_t)h_cnt[1];
    if (!with_sal) return BM2_OK;
    // SA offsets per SMEM (scan of occurrence counts), then the lookups
    if ((rc = bm2_reserve(b->sa_off, (size_t)(b->n_smem + 2) * 8))) return rc;
    if ((rc = bm2_scan_i32(c, (const int32_t *)b->occ_cnt.p, b->n_smem, (int64_t *)b->sa_off.p, b->scan_tmp))) return rc;
    int64_t n_sa = 0;
    if ((rc = bm2_check(hipMemcpyAsync(&n_sa, (int64_t *)b->sa_off.p + b->n_smem, 8, hipMemcpyDeviceToHost, s), "D2H n_sa"))) return rc;
    if ((rc = bm2_check(hipStreamSynchronize(s), "scan"))) return rc;
    b->n_sa = n_sa; b->stats.n_sa = n_sa;
    if ((rc = bm2_reserve(b->sa_coord, (size_t)(n_sa + 1) * 8))) return rc;
    if ((rc = bm2_launch_sal_expand(c, (const bm2_smem_t *)b->smem.p, b->n_smem, (const int64_t *)b->sa_off.p, opt->max_occ,
                                    (int64_t *)b->sa_coord.p))) return rc;
    if ((rc = bm2_launch_sal(c, n_sa, (int64_t *)b->sa_coord.p, (unsigned long long *)b->counters.p + 4))) return rc;
    tick(c, "sal");
    return BM2_OK;
}

// (gate / part: the chunk runs as several parts and this is part `part` of them -- see StageGate; nullptr: the whole chunk, no schedule)
struct HalfPass {                   // a part's turn in one half of the path; given up when the half is through -- or, on an error path, by the destructor
    StageGate *g; int half, part, state = 0;                     // 0 = not yet in, 1 = in, 2 = through
    HalfPass(StageGate *g_, int half_, int part_) : g(g_), half(half_), part(part_) {}
    void enter() { if (g && state == 0) { g->enter(half, part); state = 1; } }
    void leave() { if (g && state == 1) { g->leave(half, part); state = 2; } }
    ~HalfPass() { enter(); leave(); }                            // (a part that never got in still takes and passes its turn: the parts behind it must not wait for ever)
};
static int batch_run_one(bm2_ctx *c, const bm2_opt *opt, StageGate *gate = nullptr, int part = 0) {
    if (!c || !c->batch || !c->batch->uploaded) { bm2_set_error("bm2_batch_run: no batch uploaded"); return BM2_EINVAL; }
    HalfPass front(gate, 0, part), back(gate, 1, part);
    int rc = check_opt(opt);
    if (rc) return rc;
    if ((rc = bm2_check(hipSetDevice(c->device), "hipSetDevice"))) return rc;
    Batch *b = c->batch;
    const int n = b->n_reads;
    hipStream_t s = c->stream;
    memset(&b->stats, 0, sizeof b->stats);
    b->stats.n_reads = n; b->stats.n_bases = b->n_bases;
    c->n_ev = 0; c->ev_ready = false;
    (void)hipEventRecord(c->ev[0], s);
    b->n_out_regs = 0; b->n_fin = -1;
    if ((rc = bm2_reserve(b->out_off, (size_t)(n + 2) * 8))) return rc;
    if (n == 0) { b->ran = true; return bm2_check(hipMemsetAsync(b->out_off.p, 0, 16, s), "memset"); }
    front.enter();
    if ((rc = run_seeding(c, b, opt, true))) return rc;
    const int64_t n_sa = b->n_sa;
    const ChainParams cp = chain_params(opt);
    size_t ns = (size_t)n_sa + 1;
    if ((rc = bm2_reserve(b->wchain, ns * sizeof(WChain)))) return rc;
    if ((rc = bm2_reserve(b->wseed, ns * sizeof(WSeed)))) return rc;
    if ((rc = bm2_reserve(b->nodes, ns * sizeof(BtNode)))) return rc;
    if ((rc = bm2_reserve(b->order, ns * 4))) return rc;
    if ((rc = bm2_reserve(b->chn, ns * sizeof(DevChain)))) return rc;
    if ((rc = bm2_reserve(b->seeds, ns * sizeof(DevSeed)))) return rc;
    if ((rc = bm2_reserve(b->srt, ns * 4))) return rc;
    if ((rc = bm2_reserve(b->reg_seed, ns * 4))) return rc;
    if ((rc = bm2_reserve(b->reg_chain, ns * 4))) return rc;
    if ((rc = bm2_reserve(b->regs, ns * sizeof(DevReg)))) return rc;
    if ((rc = bm2_reserve(b->slot_base, ns * 8))) return rc;
    if ((rc = bm2_reserve(b->read_base, (size_t)(n + 1) * 8))) return rc;
    if ((rc = bm2_reserve(b->n_chain, (size_t)(n + 1) * 4))) return rc;
    if ((rc = bm2_reserve(b->n_reg, (size_t)(n + 1) * 4))) return rc;
    if ((rc = bm2_reserve(b->n_chain0, (size_t)(n + 1) * 4))) return rc;
    if ((rc = bm2_reserve(b->n_out, (size_t)(n + 1) * 4))) return rc;
    if ((rc = bm2_reserve(b->n_sa_read, (size_t)(n + 1) * 4))) return rc;
    if ((rc = bm2_reserve(b->perm, (size_t)(n + 1) * 4))) return rc;
    if ((rc = bm2_reserve(b->perm_hist, 256))) return rc;
    if ((rc = bm2_check(hipMemsetAsync(b->reg_seed.p, 0xff, ns * 4, s), "memset reg_seed"))) return rc;
    if ((rc = bm2_reserve(b->seed_owner, ns * 4))) return rc;
    if ((rc = bm2_check(hipMemsetAsync(b->seed_owner.p, 0xff, ns * 4, s), "memset seed_owner"))) return rc;
    // mem_flt_chained_seeds thresholds (bwamem.cpp:484-490), evaluated on the host with the same libm the reference uses:
    // min_l = W ? 1.1f*W : 5.5f*log(l_query); active when !(min_l > 0.05f*l_query); min_HSP_score = (int)(a*min_l + .499)
    bool any_flt = false;
    {
        std::vector<int32_t> mh((size_t)n);
        for (int i = 0; i < n; i++) {
            const int lq = b->h_len[i];
            const double min_l = opt->min_chain_weight ? 1.1f * opt->min_chain_weight : 5.5f * log((double)lq);
            if (lq > 0 && !(min_l > 0.05f * lq)) { mh[i] = (int)(opt->a * min_l + .499); if (mh[i] < 0) mh[i] = 0; any_flt = true; }
            else mh[i] = -1;
        }
        if (any_flt) {
            if ((rc = bm2_reserve(b->min_hsp, (size_t)(n + 1) * 4))) return rc;
            if ((rc = bm2_reserve(b->seed_keep, ns))) return rc;
            if ((rc = bm2_reserve(b->mat25, 64))) return rc;
            if ((rc = bm2_check(hipMemcpyAsync(b->min_hsp.p, mh.data(), (size_t)n * 4, hipMemcpyHostToDevice, s), "H2D min_hsp"))) return rc;
            if ((rc = bm2_check(hipMemcpyAsync(b->mat25.p, opt->mat, 25, hipMemcpyHostToDevice, s), "H2D mat"))) return rc;
            if ((rc = bm2_check(hipStreamSynchronize(s), "H2D filter params"))) return rc;     // mh is a local
        }
    }
    hipLaunchKernelGGL(k_read_base, dim3((n + 255) / 256), dim3(256), 0, s, n, (const int32_t *)b->smem_cnt.p,
                       (const int64_t *)b->smem_off.p, (const int64_t *)b->sa_off.p, (int64_t *)b->read_base.p, (int32_t *)b->n_sa_read.p);
    // chaining's read order: 5 = seed-rich reads first, then the light reads in 2x classes of seed count, every class in the reads' own order
    // (profiles/r06u_, r06v_sweep_chain_classes.json: chain 10.5-10.7 -> 9.9 ms against 4 = heavy first and the rest in plain order; classes in
    //  1.4x steps measured no better, 10.0, and were not kept)
    const int perm_mode = bm2_knob("BM2_PERM_MODE", 5);
    const int perm_mode_pf = bm2_knob("BM2_PERM_MODE_PF", 0);   // post-filter: read order
    // reads with more SA coordinates go to k_chain_heavy (round 3's sweep: 40 -> 13.6 ms, 100 -> 12.1 ms; with the lane kernel's reads in classes and the chains'
    // extension tasks built by its lanes -- round 6 -- the two sides of the stage end together at 72..80: 100 -> 9.0-9.1 ms, 90 -> 8.7, 80 -> 7.9-8.6, 72 -> 8.3,
    // 64 -> 8.5, 56 -> 8.8, 48 -> 9.1, 120 -> 9.3; profiles/r06ao_*, r06ap_*)
    const int thr_sa = bm2_knob("BM2_HEAVY_SA", 80);
    const int64_t *n_heavy_chain = nullptr;                      // set when the permutation lists the seed-rich reads first: k_chain_heavy takes them
    const int chain_heavy = bm2_knob("BM2_CHAIN_HEAVY", 1);
    if (perm_mode == 5) { if ((rc = bm2_partition_by_class(c, n, (const int32_t *)b->n_sa_read.p, thr_sa, (int32_t *)b->perm.p, b->part_tmp, b->scan_tmp, chain_heavy ? &n_heavy_chain : nullptr))) return rc; }
    else if (perm_mode == 3 || perm_mode == 4) { if ((rc = bm2_partition_by_work(c, n, (const int32_t *)b->n_sa_read.p, thr_sa, (int32_t *)b->perm.p, b->part_tmp, b->scan_tmp, perm_mode == 4, perm_mode == 4 && chain_heavy ? &n_heavy_chain : nullptr))) return rc; }
    else if ((rc = bm2_perm_by_work(c, n, (const int32_t *)b->n_sa_read.p, (int32_t *)b->perm.p, (uint32_t *)b->perm_hist.p, perm_mode))) return rc;
    // long reads: the island kernel takes its reads by falling seed count (log2 classes), so that the few reads it has to chain serially --
    // hundreds of milliseconds each -- start at once instead of behind a first round of ordinary reads
    const int32_t *isl_order = nullptr;
    if (b->max_len >= 1000 && bm2_knob("BM2_CHAIN_ISL_ORDER", 1)) {
        if ((rc = bm2_reserve(b->perm2, (size_t)(n + 1) * 4))) return rc;
        if ((rc = bm2_perm_by_work(c, n, (const int32_t *)b->n_sa_read.p, (int32_t *)b->perm2.p, (uint32_t *)b->perm_hist.p, 1))) return rc;
        isl_order = (const int32_t *)b->perm2.p;
    }
    // ... and the reads it cannot chain by islands (equal chain keys) go to a launch of their own (k_chain_serial).  (Chained again inside the island kernel,
    // as before round 5, a chunk of 20 000 long reads took 551 instead of 474 ms -- profiles/r05p_config5_variants.txt; short-read chunks whose seed-richest
    // reads are sent to the island kernel, BM2_CHAIN_TIER_MAX, still use that form: no list for them.)
    int32_t *isl_serial = nullptr;
    if (b->max_len >= 1000) {
        if ((rc = bm2_reserve(b->isl_serial, (size_t)(n + 1) * 4))) return rc;
        isl_serial = (int32_t *)b->isl_serial.p;
        if ((rc = bm2_check(hipMemsetAsync(isl_serial, 0xff, (size_t)(n + 1) * 4, s), "memset isl_serial"))) return rc;      // -1: place not yet filled
    }
    // k_chain_finish's part (reference window, extension order, reg slots of every kept chain) by the lane of k_chain that has just written the chain, when
    // nothing can come between the two -- no read of the batch long enough for the seed filter (any_flt), no island kernel borrowing `srt` as scratch --;
    // k_chain_finish is then left with the reads of the wavefront-per-read launches.  BM2_CHAIN_FUSE_FINISH=0: every read by k_chain_finish.
    // BM2_CHAIN_FINISH_PERM=0: k_chain_finish takes the reads in plain order (a wavefront then waits for its seed-richest read).
    const bool fuse_finish = !any_flt && b->max_len < 1000 && bm2_knob("BM2_CHAIN_FUSE_FINISH", 1);
    const bool finish_perm = bm2_knob("BM2_CHAIN_FINISH_PERM", 1);
    const bool finish_wave = bm2_knob("BM2_CHAIN_FINISH_WAVE", 1);     // the seed-rich reads' part by k_chain_finish_wave (one read per wavefront, one chain per lane); 0: by k_chain_finish
    const FinishOut fin_out = { (const int32_t *)b->len.p, (int32_t *)b->srt.p, (int32_t *)b->reg_seed.p, (int32_t *)b->reg_chain.p };
    if ((rc = bm2_launch_chain(c, cp, n, (const int32_t *)b->len.p, (const bm2_smem_t *)b->smem.p, (const int32_t *)b->smem_cnt.p,
                               (const int64_t *)b->smem_off.p, (const int64_t *)b->sa_off.p, (const int64_t *)b->sa_coord.p,
                               (WChain *)b->wchain.p, (WSeed *)b->wseed.p, (BtNode *)b->nodes.p, (int32_t *)b->order.p,
                               (DevChain *)b->chn.p, (DevSeed *)b->seeds.p, (int32_t *)b->seed_owner.p,
                               (int32_t *)b->n_chain.p, (int32_t *)b->n_reg.p, (int32_t *)b->n_chain0.p, (const int32_t *)b->perm.p,
                               n_heavy_chain ? thr_sa : -1, n_heavy_chain, (const int32_t *)b->n_sa_read.p,
                               (unsigned long long *)b->counters.p + 10, b->max_len, (int32_t *)b->srt.p, isl_order, isl_serial, fuse_finish ? &fin_out : nullptr))) return rc;      // counters[10..15]: work cursors of the tiers and of the overflow launch, [16]: reads listed for k_chain_serial (equal chain keys; [39] of them not staged), [38]: its work cursor
    if (any_flt) {
        if ((rc = bm2_launch_seed_filter(c, cp, (const int8_t *)b->mat25.p, n, n_sa, (const uint8_t *)b->enc.p, (const int64_t *)b->off.p,
                                         (const int32_t *)b->len.p, (const int32_t *)b->min_hsp.p, (const int64_t *)b->read_base.p,
                                         (const int32_t *)b->n_chain.p, (const int32_t *)b->seed_owner.p, (DevChain *)b->chn.p,
                                         (DevSeed *)b->seeds.p, (uint8_t *)b->seed_keep.p))) return rc;
    }
    if ((rc = bm2_launch_chain_finish(c, cp, n, (const int32_t *)b->len.p, (const int64_t *)b->read_base.p, (const int32_t *)b->n_chain.p,
                                      (DevChain *)b->chn.p, (DevSeed *)b->seeds.p, (int32_t *)b->srt.p, (int32_t *)b->reg_seed.p,
                                      (int32_t *)b->reg_chain.p, (int32_t *)b->n_reg.p, (const int32_t *)b->perm.p, finish_perm,
                                      (const int32_t *)b->n_sa_read.p, fuse_finish ? (n_heavy_chain ? thr_sa : 0x7fffffff) : -1,
                                      finish_wave ? n_heavy_chain : (const int64_t *)nullptr, thr_sa))) return rc;
    tick(c, "chain");
    if (gate) {                                                  // the front half has left the GPU before the next part's seeding is let in
        if ((rc = bm2_check(hipStreamSynchronize(s), "chaining"))) return rc;
        front.leave();
        back.enter();
    }
    if ((rc = bm2_launch_slot_base(c, n, (const int64_t *)b->read_base.p, (const int32_t *)b->n_reg.p, (int64_t *)b->slot_base.p))) return rc;
    if ((rc = bm2_reserve(b->cursor, (size_t)(n + 1) * 4))) return rc;
    if ((rc = bm2_launch_extend(c, *opt, cp, n, n_sa, (const uint8_t *)b->enc.p, (const int64_t *)b->off.p, (const int32_t *)b->len.p,
                                (const int64_t *)b->read_base.p, (const int32_t *)b->n_chain.p, (const int32_t *)b->n_reg.p,
                                (const int64_t *)b->slot_base.p, (const int32_t *)b->reg_seed.p, (const int32_t *)b->reg_chain.p,
                                (const DevChain *)b->chn.p, (const DevSeed *)b->seeds.p, (int32_t *)b->srt.p, (DevReg *)b->regs.p,
                                (unsigned long long *)b->counters.p + 5, b->ext_tmp, b->scan_tmp, (int32_t *)b->cursor.p, b->max_len))) return rc;
    tick(c, "extend");
    const int thr_reg = bm2_knob("BM2_HEAVY_REG", 12);
    if (perm_mode_pf == 3 || perm_mode_pf == 4) { if ((rc = bm2_partition_by_work(c, n, (const int32_t *)b->n_reg.p, thr_reg, (int32_t *)b->perm.p, b->part_tmp, b->scan_tmp, perm_mode_pf == 4))) return rc; }
    else if ((rc = bm2_perm_by_work(c, n, (const int32_t *)b->n_reg.p, (int32_t *)b->perm.p, (uint32_t *)b->perm_hist.p, perm_mode_pf))) return rc;
    // reads with many regs: listed (heavy first) for the wave-per-read purge; counters[9] is its work cursor
    const int64_t *n_heavy_dev = nullptr;
    if ((rc = bm2_reserve(b->perm2, (size_t)(n + 1) * 4))) return rc;
    const int pf_heavy = bm2_pf_heavy_threshold();
    if ((rc = bm2_partition_by_work(c, n, (const int32_t *)b->n_reg.p, pf_heavy, (int32_t *)b->perm2.p, b->part_tmp2, b->scan_tmp, 1, &n_heavy_dev))) return rc;
    if ((rc = bm2_launch_postfilter(c, cp, n, (const int32_t *)b->len.p, (const int64_t *)b->read_base.p, (const int32_t *)b->n_chain.p,
                                    (const int32_t *)b->n_reg.p, (const DevChain *)b->chn.p, (const DevSeed *)b->seeds.p,
                                    (int32_t *)b->srt.p, (DevReg *)b->regs.p, (int32_t *)b->n_out.p, (const int32_t *)b->cursor.p, (const int32_t *)b->perm.p,
                                    (const int32_t *)b->perm2.p, n_heavy_dev, (unsigned long long *)b->counters.p + 9, pf_heavy))) return rc;
    if ((rc = bm2_scan_i32(c, (const int32_t *)b->n_out.p, n, (int64_t *)b->out_off.p, b->scan_tmp))) return rc;
    int64_t n_out = 0;
    unsigned long long h_cnt[8];
    if ((rc = bm2_check(hipMemcpyAsync(&n_out, (int64_t *)b->out_off.p + n, 8, hipMemcpyDeviceToHost, s), "D2H n_out"))) return rc;
    if ((rc = bm2_check(hipMemcpyAsync(h_cnt, b->counters.p, 64, hipMemcpyDeviceToHost, s), "D2H counters"))) return rc;
    if ((rc = bm2_check(hipStreamSynchronize(s), "postfilter"))) return rc;
    if ((rc = bm2_reserve(b->out_regs, (size_t)(n_out + 1) * sizeof(bm2_reg_t)))) return rc;
    if ((rc = bm2_launch_reg_gather(c, n, (const int64_t *)b->read_base.p, (const int32_t *)b->n_reg.p, (const DevReg *)b->regs.p,
                                    (const int64_t *)b->out_off.p, (bm2_reg_t *)b->out_regs.p, n_out))) return rc;
    tick(c, "postfilter");
    if ((rc = bm2_check(hipStreamSynchronize(s), "gather"))) return rc;
    b->n_out_regs = n_out;
    b->stats.n_reg = n_out; b->stats.n_lf = (int64_t)h_cnt[4]; b->stats.n_sw_cells = (int64_t)h_cnt[5];
    b->stats.n_sw_tasks = (int64_t)h_cnt[6];
    b->ran = true; c->ev_ready = true;
    return BM2_OK;
}

static int batch_stats_one(bm2_ctx *c, bm2_stats *st) {
    if (!c || !c->batch || !c->batch->ran || !st) return BM2_EINVAL;
    Batch *b = c->batch;
    // n_chain / n_reg_raw are summed on demand (diagnostic only)
    std::vector<int32_t> h((size_t)b->n_reads);
    int64_t nc = 0, nr = 0;
    if (b->n_reads && b->n_chain.p) {
        if (hipMemcpy(h.data(), b->n_chain.p, (size_t)b->n_reads * 4, hipMemcpyDeviceToHost) == hipSuccess) for (int32_t v : h) nc += v;
        if (hipMemcpy(h.data(), b->n_reg.p, (size_t)b->n_reads * 4, hipMemcpyDeviceToHost) == hipSuccess) for (int32_t v : h) nr += v;
    }
    b->stats.n_chain = nc; b->stats.n_reg_raw = nr;
    *st = b->stats;
    return BM2_OK;
}

static int batch_download_one(bm2_ctx *c, bm2_reg_t *regs, int64_t cap, int64_t *reg_off, int64_t *n_out) {
    if (!c || !c->batch || !c->batch->ran || !reg_off || !n_out) { bm2_set_error("bm2_batch_download: nothing to download"); return BM2_EINVAL; }
    Batch *b = c->batch;
    *n_out = b->n_out_regs;
    int rc = bm2_check(hipMemcpy(reg_off, b->out_off.p, (size_t)(b->n_reads + 1) * 8, hipMemcpyDeviceToHost), "D2H reg_off");
    if (rc) return rc;
    if (b->n_out_regs > cap) { bm2_set_error("regs capacity %ld < %ld", (long)cap, (long)b->n_out_regs); return BM2_ECAP; }
    if (b->n_out_regs && !regs) return BM2_EINVAL;
    if (b->n_out_regs) rc = bm2_copy_d2h(c, regs, b->out_regs.p, (size_t)b->n_out_regs * sizeof(bm2_reg_t));
    return rc;
}

static int batch_kernel_ms_one(bm2_ctx *c, float *ms, int32_t cap, int32_t *n_out, const char **names) {
    if (!c || !ms || !n_out) return BM2_EINVAL;
    if (!c->ev_ready) { *n_out = 0; return BM2_OK; }
    int n = c->n_ev < cap ? c->n_ev : cap;
    for (int i = 0; i < n; i++) {
        float t = 0;
        (void)hipEventElapsedTime(&t, c->ev[i], c->ev[i + 1]);
        ms[i] = t;
        if (names) names[i] = c->ev_name[i];
    }
    *n_out = n;
    return BM2_OK;
}

// ---- sub-batch pipelining ------------------------------------------------------------------------------------------------
// A chunk is cut into up to BM2_N_SUB parts at multiples of 512 reads (the kt_for block size: the one cross-read rule of
// the path, bwamem.cpp:834, is per 512-read block, so parts are independent).  Every part has its own streams and
// workspace and is driven by its own host thread, so the latency-bound stages of one part (chaining, the tails of the
// extension launches, the host round trips for buffer sizes) overlap the bandwidth- and ALU-bound kernels of the others.
#include <thread>
#include <string>

static bm2_ctx *part_ctx(bm2_ctx *c, int i) { return i == 0 ? c : c->subs[i - 1]; }

extern "C" int bm2_batch_upload(bm2_ctx *c, const bm2_reads *reads) {
    if (!c || !reads || reads->n_reads < 0) { bm2_set_error("bm2_batch_upload: bad argument"); return BM2_EINVAL; }
    const int n = reads->n_reads;
    int parts = bm2_knob("BM2_N_SUB", 1 + (int)c->subs.size());          // (the knob is read per chunk: tools/gpu/sweep.py compares in one process)
    if (parts > 1) { const int have = bm2_ensure_subs(c, parts); if (parts > have) parts = have; }
    if (parts < 1) parts = 1;
    const int blocks = (n + BM2_BLOCK_READS - 1) / BM2_BLOCK_READS;
    if (blocks < 64 * parts) parts = 1;                    // small chunk: one part
    c->n_parts = parts;
    c->part_first.assign(parts + 1, 0);
    for (int i = 0; i <= parts; i++) {
        int64_t b = (int64_t)blocks * i / parts * BM2_BLOCK_READS;
        c->part_first[i] = (int)(b < n ? b : n);
    }
    c->part_first[parts] = n;
    for (int i = 0; i < parts; i++) {
        const int lo = c->part_first[i], hi = c->part_first[i + 1];
        bm2_reads r;
        r.n_reads = hi - lo; r.enc = reads->enc; r.off = reads->off + lo; r.len = reads->len + lo;
        // offsets stay absolute: the part uploads [0, max end) of enc only if it is part 0; other parts rebase
        std::vector<int64_t> off2;
        if (i > 0 && r.n_reads > 0) {
            int64_t mn = r.off[0];
            for (int k = 0; k < r.n_reads; k++) if (r.off[k] < mn) mn = r.off[k];
            off2.resize(r.n_reads);
            for (int k = 0; k < r.n_reads; k++) off2[k] = r.off[k] - mn;
            r.enc = reads->enc + mn; r.off = off2.data();
        }
        int rc = batch_upload_one(part_ctx(c, i), &r);
        if (rc) return rc;
    }
    return BM2_OK;
}

extern "C" int bm2_batch_run(bm2_ctx *c, const bm2_opt *opt) {
    if (!c) return BM2_EINVAL;
    if (c->n_parts <= 1) return batch_run_one(c, opt);
    std::vector<int> rcs(c->n_parts, 0);
    std::vector<std::string> msgs(c->n_parts);
    std::vector<std::thread> th;
    // the parts on their own host threads, streams and workspaces; staggered by the gate (BM2_SUB_STAGGER=0: all at once, the round-2 form,
    // in which the parts sat in the same stage at the same time and gained nothing from each other)
    StageGate *gate = bm2_knob("BM2_SUB_STAGGER", 0) ? &c->gate : nullptr;   // (measured: 100.5 ms staggered, 87.9 ms together, 81.3 ms as ONE part -- profiles/r03k_staggered_parts.json)
    c->gate.reset();
    for (int i = 0; i < c->n_parts; i++)
        th.emplace_back([&, i]() { rcs[i] = batch_run_one(part_ctx(c, i), opt, gate, i); if (rcs[i]) msgs[i] = bm2_last_error(); });
    for (auto &t : th) t.join();
    for (int i = 0; i < c->n_parts; i++) if (rcs[i]) { bm2_set_error("part %d: %s", i, msgs[i].c_str()); return rcs[i]; }
    return BM2_OK;
}

extern "C" int bm2_batch_parts(const bm2_ctx *c) { return !c ? 0 : c->n_parts < 1 ? 1 : c->n_parts; }

extern "C" int bm2_batch_stats(bm2_ctx *c, bm2_stats *st) {
    if (!c || !st) return BM2_EINVAL;
    memset(st, 0, sizeof *st);
    for (int i = 0; i < (c->n_parts < 1 ? 1 : c->n_parts); i++) {
        bm2_stats s1;
        int rc = batch_stats_one(part_ctx(c, i), &s1);
        if (rc) return rc;
        int64_t *a = (int64_t *)st; const int64_t *b = (const int64_t *)&s1;
        for (size_t k = 0; k < sizeof(bm2_stats) / 8; k++) a[k] += b[k];
    }
    return BM2_OK;
}

extern "C" int bm2_batch_download(bm2_ctx *c, bm2_reg_t *regs, int64_t cap, int64_t *reg_off, int64_t *n_out) {
    if (!c || !reg_off || !n_out) return BM2_EINVAL;
    if (c->n_parts <= 1) return batch_download_one(c, regs, cap, reg_off, n_out);
    int64_t tot = 0;
    for (int i = 0; i < c->n_parts; i++) { Batch *b = part_ctx(c, i)->batch; if (!b || !b->ran) return BM2_EINVAL; tot += b->n_out_regs; }
    *n_out = tot;
    int64_t base = 0; int rc = BM2_OK;
    for (int i = 0; i < c->n_parts; i++) {
        bm2_ctx *p = part_ctx(c, i);
        const int lo = c->part_first[i], hi = c->part_first[i + 1];
        int64_t n1 = 0;
        rc = batch_download_one(p, tot <= cap ? regs + base : nullptr, tot <= cap ? cap - base : 0, reg_off + lo, &n1);
        if (rc && !(rc == BM2_ECAP && tot > cap)) return rc;
        for (int k = lo; k <= hi; k++) reg_off[k] += base;      // (entry `hi` is rewritten by the next part with the same value)
        base += n1;
    }
    if (tot > cap) { bm2_set_error("regs capacity %ld < %ld", (long)cap, (long)tot); return BM2_ECAP; }
    return BM2_OK;
}

// per-stage time of the last run: SUMMED over the parts (the time the chunk as a whole spent in a stage; where the parts' stages overlap on
// the device -- see StageGate -- the stages add up to more than the run's wall time)
extern "C" int bm2_batch_kernel_ms(bm2_ctx *c, float *ms, int32_t cap, int32_t *n_out, const char **names) {
    if (!c || !ms || !n_out) return BM2_EINVAL;
    int rc = batch_kernel_ms_one(c, ms, cap, n_out, names);
    if (rc || c->n_parts <= 1) return rc;
    std::vector<float> t((size_t)cap);
    for (int i = 1; i < c->n_parts; i++) {
        int32_t n1 = 0;
        if ((rc = batch_kernel_ms_one(part_ctx(c, i), t.data(), cap, &n1, nullptr))) return rc;
        for (int k = 0; k < *n_out && k < n1; k++) ms[k] += t[k];
    }
    return BM2_OK;
}

extern "C" int bm2_seed_chain_extend(bm2_ctx *c, const bm2_reads *reads, const bm2_opt *opt, bm2_reg_t *regs, int64_t cap,
                                     int64_t *reg_off, int64_t *n_out, bm2_stats *stats) {
    int rc = bm2_batch_upload(c, reads);
    if (!rc) rc = bm2_batch_run(c, opt);
    if (!rc && stats) rc = bm2_batch_stats(c, stats);
    if (!rc) rc = bm2_batch_download(c, regs, cap, reg_off, n_out);
    return rc;
}

// ---- the tail of mem_kernel2_core on the device (finish.hip) -------------------------------------------------------------
static int batch_finish_one(bm2_ctx *c, const bm2_opt *opt) {
    if (!c || !c->batch || !c->batch->ran) { bm2_set_error("bm2_batch_finish: no regs on the device (run a batch first)"); return BM2_EINVAL; }
    int rc = check_opt(opt);
    if (rc) return rc;
    if ((rc = bm2_check(hipSetDevice(c->device), "hipSetDevice"))) return rc;
    Batch *b = c->batch;
    if (b->n_reads == 0) {                                       // (the stream is non-blocking: the memset has landed when this returns)
        b->n_fin = 0;
        if (bm2_reserve(b->fin_off, 16)) return BM2_ENOMEM;
        if ((rc = bm2_check(hipMemsetAsync(b->fin_off.p, 0, 16, c->stream), "memset"))) return rc;
        return bm2_check(hipStreamSynchronize(c->stream), "hit finishing (empty chunk)");
    }
    int64_t n_out = 0;
    rc = bm2_run_finish(c, opt, b->n_reads, (const uint8_t *)b->enc.p, (const int64_t *)b->off.p, (const bm2_reg_t *)b->out_regs.p,
                        (const int64_t *)b->out_off.p, b->n_out_regs, b->fin_work, b->fin_ord, b->fin_state, b->fin_n, b->fin_off, b->fin_req,
                        b->fin_cnt, b->fin_out, b->scan_tmp, &n_out, &b->fin_rounds);
    if (rc) return rc;
    if ((rc = bm2_check(hipStreamSynchronize(c->stream), "hit finishing"))) return rc;
    b->n_fin = n_out;
    return BM2_OK;
}

extern "C" int bm2_batch_finish(bm2_ctx *c, const bm2_opt *opt) {
    if (!c) return BM2_EINVAL;
    for (int i = 0; i < (c->n_parts < 1 ? 1 : c->n_parts); i++) {
        const int rc = batch_finish_one(part_ctx(c, i), opt);
        if (rc) return rc;
    }
    return BM2_OK;
}

extern "C" int bm2_batch_download_alnregs(bm2_ctx *c, bm2_alnreg_t *out, int64_t cap, int64_t *aln_off, int64_t *n_out) {
    if (!c || !aln_off || !n_out) return BM2_EINVAL;
    const int parts = c->n_parts < 1 ? 1 : c->n_parts;
    int64_t tot = 0;
    for (int i = 0; i < parts; i++) {
        Batch *b = part_ctx(c, i)->batch;
        if (!b || !b->ran || b->n_fin < 0) { bm2_set_error("bm2_batch_download_alnregs: bm2_batch_finish has not run"); return BM2_EINVAL; }
        tot += b->n_fin;
    }
    *n_out = tot;
    int64_t base = 0;
    for (int i = 0; i < parts; i++) {
        bm2_ctx *p = part_ctx(c, i);
        Batch *b = p->batch;
        const int lo = parts > 1 ? c->part_first[i] : 0;
        int rc = bm2_check(hipMemcpy(aln_off + lo, b->fin_off.p, (size_t)(b->n_reads + 1) * 8, hipMemcpyDeviceToHost), "D2H aln_off");
        if (rc) return rc;
        for (int k = lo; k <= lo + b->n_reads; k++) aln_off[k] += base;
        if (tot <= cap && b->n_fin) {
            if (!out) return BM2_EINVAL;
            if ((rc = bm2_copy_d2h(p, out + base, b->fin_out.p, (size_t)b->n_fin * sizeof(bm2_alnreg_t)))) return rc;
        }
        base += b->n_fin;
    }
    if (tot > cap) { bm2_set_error("alnregs capacity %ld < %ld", (long)cap, (long)tot); return BM2_ECAP; }
    return BM2_OK;
}

// hits of any producer (host arrays) through the same kernels: uploads the reads and the regs, finishes them, downloads
extern "C" int bm2_finish_regs_dev(bm2_ctx *c, const bm2_opt *opt, const bm2_reads *reads, const bm2_reg_t *regs, const int64_t *reg_off,
                                   bm2_alnreg_t *out, int64_t cap, int64_t *out_off, int64_t *n_out) {
    if (!c || !opt || !reads || !reg_off || !out_off || !n_out) { bm2_set_error("bm2_finish_regs_dev: bad argument"); return BM2_EINVAL; }
    int rc = batch_upload_one(c, reads);
    if (rc) return rc;
    c->n_parts = 1;
    Batch *b = c->batch;
    const int n = b->n_reads;
    const int64_t n_regs = n ? reg_off[n] : 0;
    if (n_regs && !regs) return BM2_EINVAL;
    if ((rc = bm2_reserve(b->out_off, (size_t)(n + 2) * 8))) return rc;
    if ((rc = bm2_reserve(b->out_regs, (size_t)(n_regs + 1) * sizeof(bm2_reg_t)))) return rc;
    if ((rc = bm2_check(hipMemcpy(b->out_off.p, reg_off, (size_t)(n + 1) * 8, hipMemcpyHostToDevice), "H2D reg_off"))) return rc;
    if (n_regs && (rc = bm2_check(hipMemcpy(b->out_regs.p, regs, (size_t)n_regs * sizeof(bm2_reg_t), hipMemcpyHostToDevice), "H2D regs"))) return rc;
    b->n_out_regs = n_regs; b->ran = true; b->n_fin = -1;
    if ((rc = batch_finish_one(c, opt))) return rc;
    return bm2_batch_download_alnregs(c, out, cap, out_off, n_out);
}

// ---- one chunk over several contexts ----------------------------------------------------------------------
// kernel1 / kernel2 have no cross-read state except the 512-read block rule of mem_chain_seeds (bwamem.cpp:834), so a chunk may be
// cut at multiples of 512 reads (even: mates stay together) and its parts run on different contexts -- different GPUs of a node,
// each with its index replica, or contexts sharing one replica -- on one host thread each.  The hits come back in read order; the
// caller then runs the pairing ONCE over the whole chunk (mem_pestat is chunk-wide, bwamem.cpp:1375), so the SAM does not depend
// on how many contexts took part.  No collective: every part is independent until the gather.
extern "C" int bm2_chunk_hits_sharded(bm2_ctx *const *ctxs, int n_ctx, const bm2_reads *reads, const bm2_opt *opt, bm2_alnreg_t *out, int64_t cap,
                                      int64_t *aln_off, int64_t *n_out) {
    if (!ctxs || n_ctx < 1 || !reads || !opt || !aln_off || !n_out || reads->n_reads < 0) { bm2_set_error("bm2_chunk_hits_sharded: bad argument"); return BM2_EINVAL; }
    const int n = reads->n_reads;
    const int blocks = (n + BM2_BLOCK_READS - 1) / BM2_BLOCK_READS;
    std::vector<int> first((size_t)n_ctx + 1);
    for (int i = 0; i <= n_ctx; i++) { const int64_t b = (int64_t)blocks * i / n_ctx * BM2_BLOCK_READS; first[(size_t)i] = (int)(b < n ? b : n); }
    first[(size_t)n_ctx] = n;
    std::vector<int> rcs((size_t)n_ctx, 0);
    std::vector<std::string> msgs((size_t)n_ctx);
    std::vector<std::vector<int64_t>> offs((size_t)n_ctx);
    auto device_part = [&](int i) {
        const int lo = first[(size_t)i], hi = first[(size_t)i + 1];
        bm2_reads r; r.n_reads = hi - lo; r.enc = reads->enc; r.off = reads->off + lo; r.len = reads->len + lo;
        std::vector<int64_t> off2;
        if (r.n_reads > 0) {                                    // upload only the part's bases: rebase the offsets
            int64_t mn = r.off[0];
            for (int k = 0; k < r.n_reads; k++) if (r.off[k] < mn) mn = r.off[k];
            off2.resize((size_t)r.n_reads);
            for (int k = 0; k < r.n_reads; k++) off2[(size_t)k] = r.off[k] - mn;
            r.enc = reads->enc + mn; r.off = off2.data();
        }
        int rc = bm2_batch_upload(ctxs[i], &r);
        if (!rc) rc = bm2_batch_run(ctxs[i], opt);
        if (!rc) rc = bm2_batch_finish(ctxs[i], opt);
        rcs[(size_t)i] = rc;
        if (rc) msgs[(size_t)i] = bm2_last_error();
    };
    {
        std::vector<std::thread> th;
        for (int i = 1; i < n_ctx; i++) th.emplace_back(device_part, i);
        device_part(0);
        for (auto &t : th) t.join();
    }
    for (int i = 0; i < n_ctx; i++) if (rcs[(size_t)i]) { bm2_set_error("context %d: %s", i, msgs[(size_t)i].c_str()); return rcs[(size_t)i]; }
    // gather in read order
    int64_t tot = 0;
    std::vector<int64_t> cnt((size_t)n_ctx);
    for (int i = 0; i < n_ctx; i++) {
        Batch *b = ctxs[i]->batch;
        int64_t t = 0;
        for (int k = 0; k < (ctxs[i]->n_parts < 1 ? 1 : ctxs[i]->n_parts); k++) t += part_ctx(ctxs[i], k)->batch->n_fin;
        (void)b; cnt[(size_t)i] = t; tot += t;
    }
    *n_out = tot;
    if (tot > cap) { bm2_set_error("alnregs capacity %ld < %ld", (long)cap, (long)tot); return BM2_ECAP; }
    // every part comes down on its own host thread, straight into its place of the caller's arrays (the places are the prefix sums of the
    // parts' hit counts; G copies over G links instead of one after the other)
    std::vector<int64_t> base((size_t)n_ctx + 1, 0);
    for (int i = 0; i < n_ctx; i++) base[(size_t)i + 1] = base[(size_t)i] + cnt[(size_t)i];
    auto download_part = [&](int i) {
        const int lo = first[(size_t)i], hi = first[(size_t)i + 1];
        int64_t n1 = 0;
        offs[(size_t)i].assign((size_t)(hi - lo) + 1, 0);
        const int rc = bm2_batch_download_alnregs(ctxs[i], out ? out + base[(size_t)i] : nullptr, cap - base[(size_t)i], offs[(size_t)i].data(), &n1);
        rcs[(size_t)i] = rc;
        if (rc) { msgs[(size_t)i] = bm2_last_error(); return; }
        for (int k = 0; k <= hi - lo; k++) aln_off[lo + k] = offs[(size_t)i][(size_t)k] + base[(size_t)i];
    };
    {
        std::vector<std::thread> th;
        for (int i = 1; i < n_ctx; i++) th.emplace_back(download_part, i);
        download_part(0);
        for (auto &t : th) t.join();
    }
    for (int i = 0; i < n_ctx; i++) if (rcs[(size_t)i]) { bm2_set_error("context %d: %s", i, msgs[(size_t)i].c_str()); return rcs[(size_t)i]; }
    aln_off[n] = tot;
    return BM2_OK;
}

// ---- S2 --------------------------------------------------------------------------------------------------------
extern "C" int bm2_smem(bm2_ctx *c, const bm2_reads *reads, const bm2_opt *opt, bm2_smem_t *out, int64_t cap, int64_t *n_out) {
    if (!n_out) return BM2_EINVAL;
    int rc = check_opt(opt);
    if (!rc) rc = batch_upload_one(c, reads);
    if (rc) return rc;
    c->n_parts = 1;
    Batch *b = c->batch;
    const int n = b->n_reads;
    c->n_ev = 0; c->ev_ready = false;
    (void)hipEventRecord(c->ev[0], c->stream);
    *n_out = 0;
    if (n == 0) return BM2_OK;
    if ((rc = run_seeding(c, b, opt, false))) return rc;
    *n_out = b->n_smem;
    if (b->n_smem > cap) { bm2_set_error("SMEM capacity %ld < %ld", (long)cap, (long)b->n_smem); return BM2_ECAP; }
    if (b->n_smem == 0) return BM2_OK;
    // bump order -> (rid, m, n) order: offsets by a scan over the per-read counts
    if ((rc = bm2_reserve(b->smem_sorted_off, (size_t)(n + 2) * 8))) return rc;
    if ((rc = bm2_reserve(b->smem_sorted, (size_t)b->n_smem * sizeof(bm2_smem_t)))) return rc;
    if ((rc = bm2_scan_i32(c, (const int32_t *)b->smem_cnt.p, n, (int64_t *)b->smem_sorted_off.p, b->scan_tmp))) return rc;
    if ((rc = bm2_launch_smem_gather(c, n, (const bm2_smem_t *)b->smem.p, (const int64_t *)b->smem_off.p, (const int32_t *)b->smem_cnt.p,
                                     (const int64_t *)b->smem_sorted_off.p, (bm2_smem_t *)b->smem_sorted.p))) return rc;
    rc = bm2_check(hipMemcpyAsync(out, b->smem_sorted.p, (size_t)b->n_smem * sizeof(bm2_smem_t), hipMemcpyDeviceToHost, c->stream), "D2H smem");
    if (!rc) rc = bm2_check(hipStreamSynchronize(c->stream), "bm2_smem sync");
    return rc;
}

__global__ void k_occ_cnt(const bm2_smem_t *__restrict__ sm, int64_t n, int32_t max_occ, int32_t *occ) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) occ[i] = (int32_t)(sm[i].s < max_occ ? (sm[i].s < 0 ? 0 : sm[i].s) : max_occ);
}

extern "C" int bm2_sal(bm2_ctx *c, const bm2_smem_t *smems, int64_t n, int32_t max_occ, int64_t *coords, int64_t cap, int64_t *n_out) {
    if (!c || n < 0 || max_occ <= 0 || !n_out || (n > 0 && !smems)) { bm2_set_error("bm2_sal: bad argument"); return BM2_EINVAL; }
    if (!c->has_index) { bm2_set_error("context was created without an index"); return BM2_EINVAL; }
    int rc = bm2_check(hipSetDevice(c->device), "hipSetDevice");
    if (rc) return rc;
    *n_out = 0;
    if (n == 0) return BM2_OK;
    Batch *b = get_batch(c);
    hipStream_t s = c->stream;
    if ((rc = bm2_reserve(b->smem_sorted, (size_t)n * sizeof(bm2_smem_t)))) return rc;
    if ((rc = bm2_reserve(b->occ_cnt, (size_t)(n + 1) * 4))) return rc;
    if ((rc = bm2_reserve(b->sa_off, (size_t)(n + 2) * 8))) return rc;
    if ((rc = bm2_check(hipMemcpyAsync(b->smem_sorted.p, smems, (size_t)n * sizeof(bm2_smem_t), hipMemcpyHostToDevice, s), "H2D smems"))) return rc;
    hipLaunchKernelGGL(k_occ_cnt, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const bm2_smem_t *)b->smem_sorted.p, n, max_occ, (int32_t *)b->occ_cnt.p);
    if ((rc = bm2_scan_i32(c, (const int32_t *)b->occ_cnt.p, n, (int64_t *)b->sa_off.p, b->scan_tmp))) return rc;
    int64_t n_sa = 0;
    if ((rc = bm2_check(hipMemcpyAsync(&n_sa, (int64_t *)b->sa_off.p + n, 8, hipMemcpyDeviceToHost, s), "D2H n_sa"))) return rc;
    if ((rc = bm2_check(hipStreamSynchronize(s), "scan"))) return rc;
    *n_out = n_sa;
    if (n_sa > cap) { bm2_set_error("coords capacity %ld < %ld", (long)cap, (long)n_sa); return BM2_ECAP; }
    if (n_sa == 0) return BM2_OK;
    if (!coords) return BM2_EINVAL;
    if ((rc = bm2_reserve(b->sa_coord, (size_t)(n_sa + 1) * 8))) return rc;
    if ((rc = bm2_launch_sal_expand(c, (const bm2_smem_t *)b->smem_sorted.p, n, (const int64_t *)b->sa_off.p, max_occ, (int64_t *)b->sa_coord.p))) return rc;
    if ((rc = bm2_launch_sal(c, n_sa, (int64_t *)b->sa_coord.p, nullptr))) return rc;
    rc = bm2_check(hipMemcpyAsync(coords, b->sa_coord.p, (size_t)n_sa * 8, hipMemcpyDeviceToHost, s), "D2H coords");
    if (!rc) rc = bm2_check(hipStreamSynchronize(s), "bm2_sal sync");
    return rc;
}

// ---- diagnostic: raw device arrays of the last bm2_batch_run (used by the stage-level parity tests) ------------------
extern "C" int bm2_batch_fetch(bm2_ctx *c, const char *what, void *out, int64_t cap_bytes, int64_t *n_bytes) {
    if (!c || !c->batch || !what || !n_bytes) return BM2_EINVAL;
    Batch *b = c->batch;
    const int n = b->n_reads;
    const size_t ns = (size_t)b->n_sa;
    struct { const char *name; DevBuf *buf; size_t bytes; } tab[] = {
        { "smem", &b->smem, (size_t)b->n_smem * sizeof(bm2_smem_t) }, { "smem_cnt", &b->smem_cnt, (size_t)n * 4 },
        { "smem_off", &b->smem_off, (size_t)n * 8 }, { "sa_off", &b->sa_off, (size_t)(b->n_smem + 1) * 8 },
        { "sa_coord", &b->sa_coord, ns * 8 }, { "read_base", &b->read_base, (size_t)n * 8 },
        { "n_chain", &b->n_chain, (size_t)n * 4 }, { "n_chain0", &b->n_chain0, (size_t)n * 4 }, { "n_reg", &b->n_reg, (size_t)n * 4 },
        { "n_out", &b->n_out, (size_t)n * 4 }, { "chn", &b->chn, ns * sizeof(DevChain) }, { "seeds", &b->seeds, ns * sizeof(DevSeed) },
        { "seed_counters", &b->seedc, (size_t)27 * 8 }, { "counters", &b->counters, (size_t)56 * 8 }, { "regs_raw", &b->regs, ns * sizeof(DevReg) }, { "reg_seed", &b->reg_seed, ns * 4 }, { "wchain", &b->wchain, ns * sizeof(WChain) },
    };
    if (c->n_parts > 1 && (!strcmp(what, "seed_counters") || !strcmp(what, "counters"))) {       // work counters of a chunk in parts: the parts' sums
        const size_t nb = !strcmp(what, "counters") ? (size_t)56 * 8 : (size_t)27 * 8;       // (the same 27 words as the one-part table above)
        *n_bytes = (int64_t)nb;
        if ((size_t)cap_bytes < nb) return BM2_ECAP;
        if (!out) return BM2_EINVAL;
        unsigned long long acc[56] = { 0 }, one[56];
        for (int i = 0; i < c->n_parts; i++) {
            bm2_ctx *p = part_ctx(c, i);
            if (!p->batch) return BM2_EINVAL;
            DevBuf &src = !strcmp(what, "counters") ? p->batch->counters : p->batch->seedc;
            int rc = bm2_check(hipSetDevice(p->device), "hipSetDevice");
            if (!rc) rc = bm2_check(hipMemcpy(one, src.p, nb, hipMemcpyDeviceToHost), "fetch counters");
            if (rc) return rc;
            for (size_t k = 0; k < nb / 8; k++) acc[k] += one[k];
        }
        memcpy(out, acc, nb);
        return BM2_OK;
    }
    if (!strcmp(what, "seed_attempts")) {
        *n_bytes = 4;
        if (cap_bytes < 4) return BM2_ECAP;
        if (!out) return BM2_EINVAL;
        *(int32_t *)out = b->seed_attempts;
        return BM2_OK;
    }
    for (auto &t : tab) if (!strcmp(t.name, what)) {
        *n_bytes = (int64_t)t.bytes;
        if ((int64_t)t.bytes > cap_bytes) return BM2_ECAP;
        if (!t.bytes) return BM2_OK;
        if (!t.buf->p || !out) return BM2_EINVAL;
        return bm2_check(hipMemcpy(out, t.buf->p, t.bytes, hipMemcpyDeviceToHost), "bm2_batch_fetch");
    }
    bm2_set_error("bm2_batch_fetch: unknown array '%s'", what);
    return BM2_EINVAL;
}
