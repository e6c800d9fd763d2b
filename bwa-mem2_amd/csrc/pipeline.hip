// pipeline.hip -- S2/S3 entry points (placeholder until the seeding / chaining kernels land).
#include "bm2_ctx.h"
void bm2_batch_destroy(bm2_ctx *c) { (void)c; }
extern "C" int bm2_smem(bm2_ctx *, const bm2_reads *, const bm2_opt *, bm2_smem_t *, int64_t, int64_t *) { bm2_set_error("bm2_smem: not built yet"); return BM2_EUNSUP; }
extern "C" int bm2_sal(bm2_ctx *, const bm2_smem_t *, int64_t, int32_t, int64_t *, int64_t, int64_t *) { bm2_set_error("bm2_sal: not built yet"); return BM2_EUNSUP; }
extern "C" int bm2_seed_chain_extend(bm2_ctx *, const bm2_reads *, const bm2_opt *, bm2_reg_t *, int64_t, int64_t *, int64_t *, bm2_stats *) { return BM2_EUNSUP; }
extern "C" int bm2_batch_upload(bm2_ctx *, const bm2_reads *) { return BM2_EUNSUP; }
extern "C" int bm2_batch_run(bm2_ctx *, const bm2_opt *) { return BM2_EUNSUP; }
extern "C" int bm2_batch_stats(bm2_ctx *, bm2_stats *) { return BM2_EUNSUP; }
extern "C" int bm2_batch_download(bm2_ctx *, bm2_reg_t *, int64_t, int64_t *, int64_t *) { return BM2_EUNSUP; }
extern "C" int bm2_batch_kernel_ms(bm2_ctx *, float *, int32_t, int32_t *, const char **) { return BM2_EUNSUP; }
