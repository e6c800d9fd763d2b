// bm2_ctx.h -- host-side context of libbm2.so (internal).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <condition_variable>
#include <mutex>
#include <vector>
#include "bm2_dev.h"

// growable device buffer (never shrinks; a chunk-sized workspace is reused across batches)
struct DevBuf {
    void  *p = nullptr;
    size_t cap = 0;
};

#define BM2_MAX_TIMERS 24

// The two halves of a chunk's device path lean on different parts of the GPU: seeding .. chaining waits for random HBM lines, extension ..
// purge keeps the integer VALUs busy.  When a chunk runs as several parts (sub-batches), the gate lets exactly one part be in each half at a
// time, in part order: part i + 1 seeds while part i extends.
struct StageGate {
    std::mutex m; std::condition_variable cv;
    int turn[2] = { 0, 0 };                                       // next part to enter the front / the back half
    void reset() { std::lock_guard<std::mutex> l(m); turn[0] = turn[1] = 0; }
    void enter(int half, int part) { std::unique_lock<std::mutex> l(m); cv.wait(l, [&]() { return turn[half] == part; }); }
    void leave(int half, int part) { { std::lock_guard<std::mutex> l(m); if (turn[half] == part) turn[half] = part + 1; } cv.notify_all(); }
};

struct bm2_ctx {
    int device = -1;
    hipStream_t stream = nullptr;
    bool has_index = false;
    DevIndex ix{};
    // device copies of the index arrays (owned)
    void *d_cp_occ = nullptr, *d_sa_ms = nullptr, *d_sa_ls = nullptr, *d_ref = nullptr;
    void *d_ann_off = nullptr, *d_ann_len = nullptr, *d_ann_alt = nullptr;
    // scratch for the S1/S2 entry points
    DevBuf b_pairs, b_pairs2, b_ref, b_qer, b_misc;
    int n_bsw = 0;                 // pairs of the resident S1 batch (bm2_bsw_upload)
    // batch state of the S3 path (see pipeline.hip)
    struct Batch *batch = nullptr;
    // per-kernel timers of the last bm2_batch_run
    hipEvent_t ev[BM2_MAX_TIMERS + 1];
    const char *ev_name[BM2_MAX_TIMERS];
    int n_ev = 0;
    bool ev_ready = false;
    int n_cu = 256;
    // fork/join of the per-class extension launches (extend.hip)
    hipStream_t side_stream[12] = {};
    hipEvent_t ev_fork = nullptr, ev_join[12] = {};
    bool side_ready = false;                                      // every side stream and its events exist (bm2_side_streams)
    // Which side streams share a HARDWARE QUEUE (two launches on one queue run one after the other, whatever their streams): measured when the streams
    // are made (bm2_side_streams: a spinning one-lane kernel on every stream at once, the ones whose intervals do not overlap sit on one queue).
    // side_group[i] = the queue class of side stream i, group_rep[g] = the first stream of class g; n_side_groups = 0: not known (every stream its own).
    int side_group[12] = {}, group_rep[12] = {}, n_side_groups = 0;
    // what the extension stage of the last batch saw (page-locked; written by an asynchronous copy at the end of the stage): per phase the
    // seeds of every LDS class and the reads left pending.  The next batch sizes its launches and picks its number of lazy rounds with it.
#define BM2_EXT_PHASES 8
#define BM2_EXT_STATW 16
    uint32_t *ext_stat = nullptr;
    // kernels whose dynamic LDS limit has been raised for this context's device (hipFuncSetAttribute: once per context, result checked)
    unsigned lds_attr_done = 0;
    int ext_stat_reads = 0, ext_stat_rounds = 0;
    int bsw_stat_n = 0;                                           // pairs of the S1 batch whose class counts row BSW_STAT_ROW holds
    DevBuf b_scan;                                                // prefix-sum scratch of the S1 path
    // sub-batch pipelining (pipeline.hip): extra contexts sharing this one's index replica
    std::vector<bm2_ctx *> subs;
    StageGate gate;                                               // (of the parent: the schedule of its parts)
    bool is_child = false;     // shares another context's index replica (bm2_create_shared, sub-contexts)
    bool is_sub = false;       // a sub-context of bm2_ensure_subs: takes a part of its owner's chunk, has no sub-contexts of its own
    int n_parts = 1;
    std::vector<int> part_first;
    // two pinned staging buffers for the large host <-> device copies of a chunk (bm2_copy_h2d / bm2_copy_d2h)
    void *pin[2] = { nullptr, nullptr };
    hipEvent_t pin_ev[2] = { nullptr, nullptr };
};
// Parts a chunk is cut into, each on a context (streams, workspace, host thread) of its own: the latency-bound kernels of one part run beside
// the other's.  Launch policy, knob BM2_N_SUB, read per chunk.  ONE by default: two parts on a lone context take 63.7 ms per million-read
// chunk instead of 69.7 (profiles/r04q_sweep.json), but two parts are 24 streams on the process's 16 hardware queues, and beside the
// contexts of the other resident chunks (the bench's rotation, a pipeline's second device worker) the same chunk takes 74.0 ms instead of
// 69.0 (profiles/r04r_*.json; with 24 hardware queues 86 ms, with 32 116 ms: more queues than the hardware schedules well).
#define BM2_N_SUB 1
int bm2_ensure_subs(bm2_ctx *c, int n_sub);      // -> parts available (1 + sub-contexts)

// Launch-policy knobs (grid sizes, class routing, thresholds): none of them changes a result.  Read from the environment on every
// use so that tools/gpu/sweep.py can compare settings inside one process; the defaults are the measured best (profiles/).
#include <stdlib.h>
static inline int bm2_knob(const char *name, int dflt) { const char *v = getenv(name); return v && *v ? atoi(v) : dflt; }

int  bm2_raise_lds_limit(bm2_ctx *c, int which, const void *kernel, size_t bytes);      // which: a bit number of bm2_ctx::lds_attr_done
int  bm2_side_streams(bm2_ctx *c);                         // creates the fork / join streams of this context on first use
int  bm2_check(hipError_t e, const char *what);            // -> BM2_OK or BM2_ENODEV (+ message)
void bm2_set_error(const char *fmt, ...);
int  bm2_reserve(DevBuf &b, size_t bytes);                  // grow-only device allocation
void bm2_release(DevBuf &b);
// Large copies between PAGEABLE host memory (the caller's arrays) and the device, staged through the context's pinned buffers in
// 16 MB pieces: the DMA of one piece overlaps the host memcpy of the next (a plain hipMemcpy of pageable memory runs at a few GB/s).
// Both return after the data has arrived.
int bm2_copy_h2d(bm2_ctx *c, void *dst_dev, const void *src_host, size_t bytes);
int bm2_copy_d2h(bm2_ctx *c, void *dst_host, const void *src_dev, size_t bytes);

struct SwParams;
int bm2_launch_bsw_sorted(bm2_ctx *c, bm2_seqpair_t *d_pairs, const uint8_t *d_ref, const uint8_t *d_qer, int n, int w, const SwParams &P,
                          unsigned long long *d_cells, bool *done);      // extend.hip: S1 through the lane kernel where the pairs allow it
int bm2_launch_bsw_pairs(bm2_ctx *c, bm2_seqpair_t *d_pairs, const uint8_t *d_ref, const uint8_t *d_qer, int n, int w,
                         const SwParams &P, unsigned long long *d_cells);
