// matesw.hip -- the local Smith-Waterman of mate rescue on the device: ksw_align2 (ksw.cpp:340-381) for a batch of
// (mate, reference window) pairs, SURVEY.md 8(f) row 1.  The reference runs Farrar's striped kernel on one SSE2 register
// (ksw_u8: 16 byte lanes, ksw_i16: 8 word lanes) and its results depend on that striping: query position p lives in lane
// p / slen, the lazy-F pass gives up after 16 rounds, an insertion cannot be followed by a deletion across a segment border,
// byte lanes saturate.  So the register IS the unit of work here: ONE TASK PER 16-LANE DPP ROW, lane k of the row = SIMD lane k
// of the reference's register, `_mm_slli_si128(x, 1 lane)` = DPP row_shr:1, `_mm_movemask` tests = a ballot masked to the row;
// four tasks per wavefront, each with its own control flow (rows diverge; the hardware serialises them).  A lane only ever reads
// LDS words it wrote itself, so there is no barrier in the DP.  Host oracle: ksw_align2 in sam_tail.cpp (bm2_ksw_align2).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#include <numeric>
#include <string>
#include <string.h>
#include <thread>
#include <vector>
#include "../../include/bm2.h"
#include "bm2_ctx.h"
#include "matesw_dev.h"
#include "host_pool.h"

__global__ void __launch_bounds__(256)
k_ksw_align2(const uint8_t *__restrict__ qbase, RefPtr tbase, const KswTask *__restrict__ tasks, const int *__restrict__ order, int n, KswPrm prm,
             int slen_max, bm2_ksw_result *__restrict__ out, unsigned long long *__restrict__ blists) {
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
    int8_t *smat = (int8_t *)lds;                                // 32 bytes, then the rows' areas
    if (threadIdx.x < 25) smat[threadIdx.x] = prm.mat[threadIdx.x];
    __syncthreads();
    const int rows = blockDim.x >> 4, row = threadIdx.x >> 4, k = threadIdx.x & 15;
    const int slot = blockIdx.x * rows + row;
    if (slot >= n) return;                                       // a whole row leaves; nothing below synchronises across rows
    const int id = order[slot];
    const KswTask T = tasks[id];
    ksw_row_task(qbase, tbase, T, prm, smat, lds + 16 + (size_t)row * 9 * slen_max * 16, slen_max, k, blists + T.b_off, out + id);
}

// the byte-kernel tasks of up to KSW_REG_SL stripe segments (every rescue alignment of a 150 bp run): rows in registers, no LDS but the score matrix
__global__ void __launch_bounds__(256)
k_ksw_align2_reg(const uint8_t *__restrict__ qbase, RefPtr tbase, const KswTask *__restrict__ tasks, const int *__restrict__ order, int n, KswPrm prm,
                 bm2_ksw_result *__restrict__ out, unsigned long long *__restrict__ blists) {
    __shared__ int8_t smat[32];
    if (threadIdx.x < 25) smat[threadIdx.x] = prm.mat[threadIdx.x];
    __syncthreads();
    const int rows = blockDim.x >> 4, row = threadIdx.x >> 4, k = threadIdx.x & 15;
    const int slot = blockIdx.x * rows + row;
    if (slot >= n) return;
    const int id = order[slot];
    const KswTask T = tasks[id];
    ksw_row_task_reg(qbase, tbase, T, prm, smat, k, blists + T.b_off, out + id);
}

#include "host_tail.h"

// Runs n tasks: queries at qbase_host[q_off[i]] (uploaded here), targets at t_off[i] either in the same uploaded buffer
// (d_tbase == NULL) or in a device-resident array (d_tbase, e.g. the context's ref_string replica).
static int ksw_batch_run(bm2_ctx *c, int32_t n, const uint8_t *qbuf, int64_t qbuf_bytes, const RefPtr *d_tbase, const int64_t *q_off,
                         const int32_t *q_len, const int64_t *t_off, const int32_t *t_len, const int32_t *xtra, const int8_t mat[25], int o_del,
                         int e_del, int o_ins, int e_ins, bm2_ksw_result *out) {
    if (n == 0) return BM2_OK;
    int rc = bm2_check(hipSetDevice(c->device), "hipSetDevice");
    if (rc) return rc;
    TailProf prof("ksw_batch");
    KswPrm prm;
    int lo = 127, hi = 0;
    for (int a = 0; a < 25; ++a) { prm.mat[a] = mat[a]; if (mat[a] < lo) lo = mat[a]; if (mat[a] > hi) hi = mat[a]; }
    prm.o_del = o_del; prm.e_del = e_del; prm.o_ins = o_ins; prm.e_ins = e_ins;
    prm.shift = (256 - (lo & 0xff)) & 0xff; prm.maxsc = hi;
    if (hi <= 0) { bm2_set_error("ksw batch: the scoring matrix has no positive entry"); return BM2_EINVAL; }
    // (host-side staging kept per calling thread from batch to batch; every loop over the tasks runs on the host's worker threads)
    static thread_local std::vector<KswTask> tasks_tl; static thread_local std::vector<int> order_tl; static thread_local std::vector<int64_t> nb_tl;
    std::vector<KswTask> &tasks = tasks_tl; std::vector<int> &order = order_tl; std::vector<int64_t> &nb_of = nb_tl;
    if (tasks.size() < (size_t)n) { tasks.resize((size_t)n); order.resize((size_t)n); }
    const int host_threads = bm2_host_threads();
    const int64_t grain = 16384, pieces = ((int64_t)n + grain - 1) / grain;
    nb_of.assign((size_t)pieces + 1, 0);
    std::atomic<int> bad(0), slen_all(1);
    bm2_parallel_ranges(n, grain, host_threads, [&](int64_t lo, int64_t hi) {
        int64_t nbp = 0; int sl = 1;
        for (int64_t i = lo; i < hi; ++i) {
            if (q_len[i] < 0 || t_len[i] < 0) { bad = 1; continue; }
            nbp += (t_len[i] + 1) / 2 + 1;
            const int P = (xtra[i] & KSW_XBYTE) ? 16 : 8;
            sl = std::max(sl, (q_len[i] + P - 1) / P);
        }
        nb_of[(size_t)(lo / grain) + 1] = nbp;
        for (int cur = slen_all.load(); sl > cur && !slen_all.compare_exchange_weak(cur, sl);) {}
    });
    if (bad.load()) { bm2_set_error("ksw batch: negative length"); return BM2_EINVAL; }
    for (int64_t p = 0; p < pieces; ++p) nb_of[(size_t)p + 1] += nb_of[(size_t)p];
    const int64_t nb = nb_of[(size_t)pieces];
    const int slen_max = slen_all.load();
    bm2_parallel_ranges(n, grain, host_threads, [&](int64_t lo, int64_t hi) {
        int64_t at = nb_of[(size_t)(lo / grain)];
        for (int64_t i = lo; i < hi; ++i) {
            KswTask &T = tasks[(size_t)i];
            T.q_off = q_off[i]; T.t_off = t_off[i]; T.qlen = q_len[i]; T.tlen = t_len[i]; T.xtra = xtra[i]; T.pad = 0;
            T.b_off = at; at += (t_len[i] + 1) / 2 + 1;
        }
    });
    // rows of a wavefront diverge: neighbours should be alike (same lane width, same segment count, similar target length): a
    // counting sort on (lane width, segments, target length / 8), largest first (which of two equal tasks comes first changes nothing)
    {
        const int NB = 128 * 512;
        bm2_counting_order(n, NB, host_threads, [&](int i) {
            const KswTask &T = tasks[(size_t)i];
            const int P = (T.xtra & KSW_XBYTE) ? 16 : 8;
            const int sl = std::min((T.qlen + P - 1) / P, 63), tl = std::min(T.tlen >> 3, 511);
            return NB - 1 - ((((P == 8) ? 64 : 0) + sl) * 512 + tl);
        }, order.data());
    }
    prof.mark("order");
    // rows (tasks) per block: as many as fit 64 KB of LDS, the per-workgroup amount every launch may ask for without further ado
    int rows = 16;
    size_t lds = 32 + (size_t)rows * 9 * slen_max * 16 * 2;
    while (lds > 64 * 1024 && rows > 4) { rows >>= 1; lds = 32 + (size_t)rows * 9 * slen_max * 16 * 2; }
    if (lds > 64 * 1024) {
        bm2_set_error("ksw batch: a query of %d stripe segments does not fit the LDS layout (mates longer than 448 bases: use bm2_sam_pe)", slen_max);
        return BM2_EUNSUP;
    }
    c->n_bsw = 0;                                               // (these scratch buffers held the resident S1 batch, if any: it is gone)
    DevBuf &b_seq = c->b_ref, &b_task = c->b_qer, &b_out = c->b_pairs, &b_misc = c->b_misc;
    const size_t task_bytes = (size_t)n * sizeof(KswTask), ord_bytes = (size_t)n * sizeof(int);
    if ((rc = bm2_reserve(b_seq, (size_t)qbuf_bytes + 64))) return rc;
    if ((rc = bm2_reserve(b_task, task_bytes + ord_bytes + 64))) return rc;
    if ((rc = bm2_reserve(b_out, (size_t)n * sizeof(bm2_ksw_result)))) return rc;
    if ((rc = bm2_reserve(b_misc, (size_t)nb * 8))) return rc;
    hipStream_t s = c->stream;
    KswTask *d_task = (KswTask *)b_task.p;
    int *d_order = (int *)((char *)b_task.p + ((task_bytes + 15) & ~(size_t)15));
    rc = bm2_copy_h2d(c, b_seq.p, qbuf, (size_t)qbuf_bytes);       // (pageable memory: through the context's pinned staging buffers)
    if (!rc) rc = bm2_copy_h2d(c, d_task, tasks.data(), task_bytes);
    if (!rc) rc = bm2_copy_h2d(c, d_order, order.data(), ord_bytes);
    if (rc) return rc;
    // The order is (word kernel first, then the byte kernel by falling segment count): the tasks whose rows fit registers are its TAIL (k_ksw_align2_reg;
    // BM2_KSW_REG=0: every task on the LDS kernel).
    int n_reg = 0;
    if (bm2_knob("BM2_KSW_REG", 1))
        while (n_reg < n && ksw_task_fits_regs(tasks[(size_t)order[(size_t)(n - 1 - n_reg)]])) n_reg++;
    const RefPtr tb_arg = d_tbase ? *d_tbase : RefPtr::bytes((const uint8_t *)b_seq.p);
    const int n_lds = n - n_reg;
    if (n_lds > 0)
        hipLaunchKernelGGL(k_ksw_align2, dim3((n_lds + rows - 1) / rows), dim3(rows * 16), lds, s, (const uint8_t *)b_seq.p, tb_arg, d_task, d_order, n_lds, prm,
                           slen_max, (bm2_ksw_result *)b_out.p, (unsigned long long *)b_misc.p);
    if (n_reg > 0)
        hipLaunchKernelGGL(k_ksw_align2_reg, dim3((n_reg + 15) / 16), dim3(256), 0, s, (const uint8_t *)b_seq.p, tb_arg, d_task, d_order + n_lds, n_reg, prm,
                           (bm2_ksw_result *)b_out.p, (unsigned long long *)b_misc.p);
    rc = bm2_check(hipGetLastError(), "k_ksw_align2 launch");
    if (prof.on) { (void)hipStreamSynchronize(s); prof.mark("H2D + kernel"); }
    if (!rc) rc = bm2_copy_d2h(c, out, b_out.p, (size_t)n * sizeof(bm2_ksw_result));     // (waits for the kernel: same stream)
    return rc;
}

// Device twin of bm2_ksw_align2 (same arguments after the context, same results).
extern "C" int bm2_ksw_align2_dev(bm2_ctx *c, int32_t n, const uint8_t *seqs, int64_t seq_bytes, const int64_t *q_off, const int32_t *q_len,
                                  const int64_t *t_off, const int32_t *t_len, const int32_t *xtra, const int8_t mat[25], int o_del, int e_del,
                                  int o_ins, int e_ins, bm2_ksw_result *out) {
    if (!c || n < 0 || (n > 0 && (!seqs || !q_off || !q_len || !t_off || !t_len || !xtra || !mat || !out))) {
        bm2_set_error("bm2_ksw_align2_dev: bad argument");
        return BM2_EINVAL;
    }
    return ksw_batch_run(c, n, seqs, seq_bytes, nullptr, q_off, q_len, t_off, t_len, xtra, mat, o_del, e_del, o_ins, e_ins, out);
}

int bm2_dev_cigar_batch(void *user, const bm2_opt *opt, const bm2_reads *reads, int64_t enc_bytes, int32_t n, const bm2h_cg_hit *hits, bm2h_cg_out *out);      // cigar.hip

// bm2_sam_pe with the mate-rescue alignments AND the CIGAR alignments of the chunk on the device: the host plans them, this hook runs them against the
// context's resident ref_string, the host replays the pairs (sam_tail.cpp: bm2h_sam_pe).
static int dev_rescue_batch(void *user, int32_t n, const uint8_t *qbuf, int64_t qbuf_bytes, const int64_t *q_off, const int32_t *q_len,
                            const int64_t *t_pos, const int32_t *t_len, const int32_t *xtra, const bm2_opt *opt, const uint8_t *,
                            bm2_ksw_result *out) {
    bm2_ctx *c = (bm2_ctx *)user;
    const RefPtr ref = c->ix.ref(0);
    return ksw_batch_run(c, n, qbuf, qbuf_bytes, &ref, q_off, q_len, t_pos, t_len, xtra, opt->mat, opt->o_del, opt->e_del,
                         opt->o_ins, opt->e_ins, out);
}

extern "C" int bm2_sam_pe_dev(bm2_ctx *c, const bm2_index_desc *idx, const bm2_opt *opt, const bm2_sam_opt *so, const bm2_reads *reads,
                              const bm2_read_text *txt, const bm2_alnreg_t *alnregs, const int64_t *reg_off, int64_t n_processed,
                              const bm2_pestat *pes_in, bm2_pestat *pes_out, char *out, int64_t cap, int64_t *n_out) {
    if (!c || !c->has_index || !c->ix.ref_string) { bm2_set_error("bm2_sam_pe_dev: the context holds no index"); return BM2_EINVAL; }
    return bm2h_sam_pe(idx, opt, so, reads, txt, alnregs, reg_off, n_processed, pes_in, pes_out, out, cap, n_out, dev_rescue_batch, c,
                       bm2_dev_cigar_batch, c);
}

// ---- the tail's two device batches over SEVERAL contexts (one per GPU, or contexts sharing a replica): once mem_pestat has run the rescue
// alignments and the CIGAR alignments of a chunk are independent per task, so each batch is cut into contiguous parts, one host thread and one
// context per part, and the results are put back in task order -- the text cannot depend on the number of contexts.  (bwamem.cpp:1366-1381 runs
// them inside worker_sam on every host thread; with G GPUs the hot path of a chunk shrinks G-fold and the tail's batches have to follow.)
namespace {
struct MultiCtx { bm2_ctx *const *ctx; int n; };
template <class F> int run_parts(int parts, int budget, F f) {          // f(part) on a thread of its own; first error wins
    std::vector<int> rcs((size_t)parts, 0);
    std::vector<std::string> msgs((size_t)parts);
    auto one = [&](int g) {
        bm2_host_thread_budget() = budget;                                  // (a fresh thread would size its host loops to the whole machine)
        rcs[(size_t)g] = f(g);
        if (rcs[(size_t)g]) msgs[(size_t)g] = bm2_last_error();
    };
    std::vector<std::thread> th;
    for (int g = 1; g < parts; g++) th.emplace_back(one, g);
    const int mine = bm2_host_thread_budget();
    one(0);
    bm2_host_thread_budget() = mine;
    for (auto &t : th) t.join();
    for (int g = 0; g < parts; g++) if (rcs[(size_t)g]) { bm2_set_error("context %d: %s", g, msgs[(size_t)g].c_str()); return rcs[(size_t)g]; }
    return BM2_OK;
}
int parts_of(const MultiCtx *m, int64_t n, int64_t grain) { int64_t p = n / grain + 1; return (int)(p < m->n ? p : m->n); }

int multi_rescue_batch(void *user, int32_t n, const uint8_t *qbuf, int64_t qbuf_bytes, const int64_t *q_off, const int32_t *q_len,
                       const int64_t *t_pos, const int32_t *t_len, const int32_t *xtra, const bm2_opt *opt, const uint8_t *unused, bm2_ksw_result *out) {
    const MultiCtx *m = (const MultiCtx *)user;
    const int G = parts_of(m, n, 8192);
    if (G <= 1) return dev_rescue_batch(m->ctx[0], n, qbuf, qbuf_bytes, q_off, q_len, t_pos, t_len, xtra, opt, unused, out);
    const int budget = bm2_host_threads() / G > 0 ? bm2_host_threads() / G : 1;
    return run_parts(G, budget, [&](int g) {
        const int64_t lo = (int64_t)n * g / G, hi = (int64_t)n * (g + 1) / G;
        return dev_rescue_batch(m->ctx[g], (int32_t)(hi - lo), qbuf, qbuf_bytes, q_off + lo, q_len + lo, t_pos + lo, t_len + lo, xtra + lo, opt, unused, out + lo);
    });
}
int multi_cigar_batch(void *user, const bm2_opt *opt, const bm2_reads *reads, int64_t enc_bytes, int32_t n, const bm2h_cg_hit *hits, bm2h_cg_out *out) {
    const MultiCtx *m = (const MultiCtx *)user;
    const int G = parts_of(m, n, 16384);
    if (G <= 1) return bm2_dev_cigar_batch(m->ctx[0], opt, reads, enc_bytes, n, hits, out);
    std::vector<bm2h_cg_out> part((size_t)G);
    const int budget = bm2_host_threads() / G > 0 ? bm2_host_threads() / G : 1;
    int rc = run_parts(G, budget, [&](int g) {
        const int64_t lo = (int64_t)n * g / G, hi = (int64_t)n * (g + 1) / G;
        return bm2_dev_cigar_batch(m->ctx[g], opt, reads, enc_bytes, (int32_t)(hi - lo), hits + lo, &part[(size_t)g]);
    });
    if (rc) return rc;
    // task order again: the per-task arrays are concatenated, the offsets into the CIGAR / MD pools move by what the parts before hold
    size_t nt = 0, nc = 0, nm = 0;
    for (const bm2h_cg_out &p : part) { nt += p.score.size(); nc += p.cigar.size(); nm += p.md.size(); }
    out->score.resize(nt); out->nm.resize(nt); out->n_cigar.resize(nt); out->cigar_off.resize(nt); out->md_off.resize(nt);
    out->cigar.resize(nc); out->md.resize(nm);
    size_t at = 0, ac = 0, am = 0;
    for (const bm2h_cg_out &p : part) {
        const size_t k = p.score.size();
        for (size_t i = 0; i < k; i++) {
            out->score[at + i] = p.score[i]; out->nm[at + i] = p.nm[i]; out->n_cigar[at + i] = p.n_cigar[i];
            out->cigar_off[at + i] = p.cigar_off[i] + (int64_t)ac; out->md_off[at + i] = p.md_off[i] + (int64_t)am;
        }
        if (!p.cigar.empty()) memcpy(out->cigar.data() + ac, p.cigar.data(), p.cigar.size() * sizeof(uint32_t));
        if (!p.md.empty()) memcpy(out->md.data() + am, p.md.data(), p.md.size());
        at += k; ac += p.cigar.size(); am += p.md.size();
    }
    return BM2_OK;
}
bool all_hold_index(bm2_ctx *const *ctxs, int n_ctx) {
    if (!ctxs || n_ctx < 1) return false;
    for (int i = 0; i < n_ctx; i++) if (!ctxs[i] || !ctxs[i]->has_index || !ctxs[i]->ix.ref_string) return false;
    return true;
}
}  // namespace

extern "C" int bm2_sam_pe_dev_multi(bm2_ctx *const *ctxs, int n_ctx, const bm2_index_desc *idx, const bm2_opt *opt, const bm2_sam_opt *so, const bm2_reads *reads,
                                    const bm2_read_text *txt, const bm2_alnreg_t *alnregs, const int64_t *reg_off, int64_t n_processed,
                                    const bm2_pestat *pes_in, bm2_pestat *pes_out, char *out, int64_t cap, int64_t *n_out) {
    if (!all_hold_index(ctxs, n_ctx)) { bm2_set_error("bm2_sam_pe_dev_multi: every context must hold the index"); return BM2_EINVAL; }
    MultiCtx m = { ctxs, n_ctx };
    return bm2h_sam_pe(idx, opt, so, reads, txt, alnregs, reg_off, n_processed, pes_in, pes_out, out, cap, n_out, multi_rescue_batch, &m, multi_cigar_batch, &m);
}
extern "C" int bm2_sam_se_dev_multi(bm2_ctx *const *ctxs, int n_ctx, const bm2_index_desc *idx, const bm2_opt *opt, const bm2_sam_opt *so, const bm2_reads *reads,
                                    const bm2_read_text *txt, bm2_alnreg_t *alnregs, const int64_t *reg_off, int64_t n_processed, char *out,
                                    int64_t cap, int64_t *n_out) {
    if (!all_hold_index(ctxs, n_ctx)) { bm2_set_error("bm2_sam_se_dev_multi: every context must hold the index"); return BM2_EINVAL; }
    MultiCtx m = { ctxs, n_ctx };
    return bm2h_sam_se(idx, opt, so, reads, txt, alnregs, reg_off, n_processed, out, cap, n_out, multi_cigar_batch, &m);
}
