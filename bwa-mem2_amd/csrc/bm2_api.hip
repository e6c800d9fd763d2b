// bm2_api.hip -- C-ABI glue: context lifetime, error reporting, option defaults, the S1 entry point.
// (S2/S3 entry points live in pipeline.hip.)
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <new>
#include <mutex>
#include <algorithm>
#include <vector>
#include "bm2_ctx.h"

static thread_local char g_err[512] = "";

void bm2_set_error(const char *fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap);
}
extern "C" const char *bm2_last_error(void) { return g_err; }

int bm2_check(hipError_t e, const char *what) {
    if (e == hipSuccess) return BM2_OK;
    bm2_set_error("%s: %s", what, hipGetErrorString(e));
    return e == hipErrorOutOfMemory ? BM2_ENOMEM : BM2_ENODEV;
}

int bm2_reserve(DevBuf &b, size_t bytes) {
    if (bytes <= b.cap) return BM2_OK;
    if (b.p) { (void)hipFree(b.p); b.p = nullptr; b.cap = 0; }
    size_t want = bytes + bytes / 4 + 256;
    int rc = bm2_check(hipMalloc(&b.p, want), "hipMalloc");
    if (rc != BM2_OK) { b.p = nullptr; return BM2_ENOMEM; }
    b.cap = want;
    return BM2_OK;
}
void bm2_release(DevBuf &b) { if (b.p) (void)hipFree(b.p); b.p = nullptr; b.cap = 0; }

extern "C" int bm2_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// mem_opt_init, bwamem.cpp:107-143
extern "C" void bm2_opt_fill_scmat(bm2_opt *o) {     // bwa_fill_scmat, bwa.cpp:248-257
    int k = 0;
    for (int i = 0; i < 4; ++i) {
        for (int j = 0; j < 4; ++j) o->mat[k++] = (int8_t)(i == j ? o->a : -o->b);
        o->mat[k++] = -1;
    }
    for (int j = 0; j < 5; ++j) o->mat[k++] = -1;
}
extern "C" void bm2_opt_init(bm2_opt *o) {
    memset(o, 0, sizeof *o);
    o->a = 1; o->b = 4; o->o_del = o->o_ins = 6; o->e_del = o->e_ins = 1;
    o->w = 100; o->zdrop = 100; o->pen_clip5 = o->pen_clip3 = 5;
    o->max_mem_intv = 20; o->min_seed_len = 19; o->split_width = 10; o->max_occ = 500;
    o->max_chain_gap = 10000; o->mask_level = 0.50f; o->drop_ratio = 0.50f; o->split_factor = 1.5f;
    o->mask_level_redun = 0.95f; o->min_chain_weight = 0; o->max_chain_extend = 1 << 30;
    bm2_opt_fill_scmat(o);
}

template <class T>
static int upload(void **dst, const T *src, size_t n, hipStream_t s) {
    size_t bytes = n * sizeof(T);
    int rc = bm2_check(hipMalloc(dst, bytes ? bytes : 64), "hipMalloc(index)");
    if (rc) return rc;
    if (bytes) rc = bm2_check(hipMemcpyAsync(*dst, src, bytes, hipMemcpyHostToDevice, s), "hipMemcpy(index)");
    return rc;
}

void bm2_batch_destroy(bm2_ctx *c);     // pipeline.hip

#define BM2_PIN_CAP ((size_t)16 << 20)
// (BM2_PIN_PIECE / BM2_PIN_MIN: test hooks that make small copies go through the staged path in small pieces)
static size_t pin_piece() {
    const char *e = getenv("BM2_PIN_PIECE");
    long v = e && *e ? atol(e) : (long)BM2_PIN_CAP;
    return v < 4096 ? (size_t)4096 : (size_t)v > BM2_PIN_CAP ? BM2_PIN_CAP : (size_t)v;     // (0 would never advance, more than the buffers hold would overrun them)
}
static const size_t BM2_PIN_BYTES = pin_piece();
static const size_t BM2_PIN_MIN = getenv("BM2_PIN_MIN") ? (size_t)atol(getenv("BM2_PIN_MIN")) : (size_t)1 << 20;
static int pin_ready(bm2_ctx *c) {
    for (int i = 0; i < 2; i++) {
        if (!c->pin[i] && bm2_check(hipHostMalloc(&c->pin[i], BM2_PIN_CAP, 0), "hipHostMalloc")) { c->pin[i] = nullptr; return BM2_ENOMEM; }
        if (!c->pin_ev[i]) (void)hipEventCreateWithFlags(&c->pin_ev[i], hipEventDisableTiming);
    }
    return BM2_OK;
}
// Page-locked host memory for the caller's big per-chunk arrays (reads in, hits and text out): copies from / to such memory go straight
// over the DMA engines at PCIe speed and cost the host no memcpy into the staging buffers (bm2_copy_* notice it by themselves).
extern "C" void *bm2_host_alloc(int64_t bytes) {
    void *p = nullptr;
    if (bytes <= 0 || bm2_check(hipHostMalloc(&p, (size_t)bytes, hipHostMallocPortable), "hipHostMalloc")) return nullptr;
    return p;
}
extern "C" void bm2_host_free(void *p) { if (p) (void)hipHostFree(p); }
// ---- page-locked blocks for the chunk-sized arrays the LIBRARY hands out (the parser's enc / off / len): locking 160 MB of pages costs tens of
// milliseconds, so blocks go back to a pool when their chunk is freed and the next chunk's parse takes them again.  Small requests and hosts
// without a device get plain malloc.  A block is reused for a request it fits without wasting more than half of it; free blocks beyond
// BM2_PIN_POOL_MB (default 3072) are unlocked and released.
namespace {
struct PoolBlock { void *p; size_t cap; bool used, pinned; };
// (a heap singleton that is never destroyed: detached parser threads may still hand blocks back while the process's statics are being torn down)
struct ChunkPool : std::mutex { std::vector<PoolBlock> blocks; };
ChunkPool &pool() { static ChunkPool *p = new ChunkPool; return *p; }
}
void *bm2_chunk_mem_get(size_t bytes) {
    if (bytes < ((size_t)1 << 20)) return malloc(bytes ? bytes : 1);
    {
        std::lock_guard<std::mutex> l(pool());
        PoolBlock *best = nullptr;
        for (PoolBlock &b : pool().blocks)
            if (!b.used && b.cap >= bytes && b.cap <= 2 * bytes + ((size_t)8 << 20) && (!best || b.cap < best->cap)) best = &b;
        if (best) { best->used = true; return best->p; }
    }
    const size_t cap = (bytes + bytes / 8 + ((size_t)4 << 20) - 1) & ~(((size_t)4 << 20) - 1);
    void *p = nullptr; bool pinned = true;
    // (a host without a device, or out of lockable memory: plain memory of the size asked for, NOT pooled -- bm2_chunk_mem_put frees it at once, a
    //  parser-only process would otherwise sit on gigabytes of idle blocks)
    if (hipHostMalloc(&p, cap, hipHostMallocPortable) != hipSuccess) { (void)hipGetLastError(); return malloc(bytes); }
    if (!p) return nullptr;
    std::lock_guard<std::mutex> l(pool());
    pool().blocks.push_back(PoolBlock{ p, cap, true, pinned });
    return p;
}
void bm2_chunk_mem_put(void *p) {
    if (!p) return;
    std::vector<PoolBlock> drop;
    {
        std::lock_guard<std::mutex> l(pool());
        std::vector<PoolBlock> &blocks = pool().blocks;
        bool mine = false;
        for (PoolBlock &b : blocks) if (b.p == p) { b.used = false; mine = true; break; }
        if (!mine) { free(p); return; }                          // (a small request, or the plain memory of a host that cannot lock pages)
        const char *e = getenv("BM2_PIN_POOL_MB");
        long mb = e && *e ? atol(e) : 3072;
        if (mb < 0) mb = 0;                                      // (a negative or unreadable setting keeps nothing idle; it must not become a huge unsigned limit)
        if (mb > (1L << 20)) mb = 1L << 20;
        const size_t limit = (size_t)mb << 20;
        size_t idle = 0;
        for (const PoolBlock &b : blocks) if (!b.used) idle += b.cap;
        for (size_t i = 0; i < blocks.size() && idle > limit;) {
            if (!blocks[i].used) { idle -= blocks[i].cap; drop.push_back(blocks[i]); blocks.erase(blocks.begin() + (long)i); }
            else ++i;
        }
    }
    for (const PoolBlock &b : drop) { if (b.pinned) (void)hipHostFree(b.p); else free(b.p); }
}

static bool is_pinned(const void *host) {
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, host) != hipSuccess) { (void)hipGetLastError(); return false; }   // (plain malloc'd memory: "invalid value")
    return a.type == hipMemoryTypeHost;
}

int bm2_copy_h2d(bm2_ctx *c, void *dst_dev, const void *src_host, size_t bytes) {
    if (bytes < BM2_PIN_MIN || is_pinned(src_host) || pin_ready(c)) {
        int rc = bm2_check(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, c->stream), "H2D");
        return rc ? rc : bm2_check(hipStreamSynchronize(c->stream), "H2D sync");
    }
    int k = 0;
    for (size_t at = 0; at < bytes; at += BM2_PIN_BYTES, k ^= 1) {
        const size_t n = bytes - at < BM2_PIN_BYTES ? bytes - at : BM2_PIN_BYTES;
        if (at >= 2 * BM2_PIN_BYTES) { int rc = bm2_check(hipEventSynchronize(c->pin_ev[k]), "H2D staging"); if (rc) return rc; }   // buffer k is free again
        memcpy(c->pin[k], (const char *)src_host + at, n);
        int rc = bm2_check(hipMemcpyAsync((char *)dst_dev + at, c->pin[k], n, hipMemcpyHostToDevice, c->stream), "H2D");
        if (rc) return rc;
        (void)hipEventRecord(c->pin_ev[k], c->stream);
    }
    return bm2_check(hipStreamSynchronize(c->stream), "H2D sync");
}
int bm2_copy_d2h(bm2_ctx *c, void *dst_host, const void *src_dev, size_t bytes) {
    if (bytes < BM2_PIN_MIN || is_pinned(dst_host) || pin_ready(c)) {
        int rc = bm2_check(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, c->stream), "D2H");
        return rc ? rc : bm2_check(hipStreamSynchronize(c->stream), "D2H sync");
    }
    // piece i travels into buffer i & 1 while piece i - 1 is copied out of the other
    size_t prev_at = 0, prev_n = 0; int k = 0;
    for (size_t at = 0; at < bytes; at += BM2_PIN_BYTES, k ^= 1) {
        const size_t n = bytes - at < BM2_PIN_BYTES ? bytes - at : BM2_PIN_BYTES;
        int rc = bm2_check(hipMemcpyAsync(c->pin[k], (const char *)src_dev + at, n, hipMemcpyDeviceToHost, c->stream), "D2H");
        if (rc) return rc;
        (void)hipEventRecord(c->pin_ev[k], c->stream);
        if (prev_n) {
            if ((rc = bm2_check(hipEventSynchronize(c->pin_ev[k ^ 1]), "D2H staging"))) return rc;
            memcpy((char *)dst_host + prev_at, c->pin[k ^ 1], prev_n);
        }
        prev_at = at; prev_n = n;
    }
    int rc = bm2_check(hipStreamSynchronize(c->stream), "D2H sync");
    if (!rc && prev_n) memcpy((char *)dst_host + prev_at, c->pin[k ^ 1], prev_n);
    return rc;
}

static int make_streams(bm2_ctx *c) {
    if (bm2_check(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking), "hipStreamCreate")) return BM2_ENODEV;
    for (int i = 0; i <= BM2_MAX_TIMERS; i++) (void)hipEventCreate(&c->ev[i]);
    if (hipHostMalloc((void **)&c->ext_stat, sizeof(uint32_t) * BM2_EXT_PHASES * BM2_EXT_STATW, hipHostMallocPortable) != hipSuccess) c->ext_stat = nullptr;
    if (c->ext_stat) memset(c->ext_stat, 0, sizeof(uint32_t) * BM2_EXT_PHASES * BM2_EXT_STATW);      // (rows no round has written yet size grids as hints: zeros, not whatever the pool held)
    return BM2_OK;
}
// One lane that spins for `ticks` of the 100 MHz clock and leaves its start and end: launched on every side stream at once, the intervals of two streams
// overlap unless the streams share a hardware queue.
__global__ void k_queue_probe(unsigned long long *out, int slot, long long ticks) {
    const long long t0 = wall_clock64();
    long long t1 = t0;
    for (int i = 0; i < (1 << 20) && t1 - t0 < ticks; i++) { __builtin_amdgcn_s_sleep(16); t1 = wall_clock64(); }
    out[2 * slot] = (unsigned long long)t0; out[2 * slot + 1] = (unsigned long long)t1;
}
// The runtime spreads a process's streams over its hardware queues as it sees fit (least-loaded queue first; the contexts of a process, torch's own streams
// and streams long destroyed all count), so WHICH side streams share a queue differs from process to process: profiles/r05_timeline.tsv has the eight
// launches of an extension phase on eight queues, profiles/r06f_timeline.tsv has two pairs of them on two -- the 64-column class ran alone for 1.1 ms
// behind the wavefront kernel, after everything else of its phase had ended.  Nothing in HIP tells; so it is measured once per context (0.4 ms).
static void probe_side_queues(bm2_ctx *c) {
    c->n_side_groups = 0;
    if (!bm2_knob("BM2_QUEUE_PROBE", 1)) return;
#if !defined(BM2_EMU_ROW_PRIMS)                                   /* (the host emulator has no queues and no clock) */
    enum { NS = 13 };                                             // the twelve side streams and the main stream (slot 12)
    unsigned long long *d = nullptr, h[2 * NS];
    if (hipMalloc((void **)&d, sizeof h) != hipSuccess) return;
    bool ok = hipMemset(d, 0, sizeof h) == hipSuccess && hipDeviceSynchronize() == hipSuccess;
    const long long ticks = 30000;                                // 300 us: an order of magnitude beyond the time it takes to queue the launches
    auto stream_of = [&](int i) { return i < 12 ? c->side_stream[i] : c->stream; };
    for (int i = 0; ok && i < NS; i++) hipLaunchKernelGGL(k_queue_probe, dim3(1), dim3(1), 0, stream_of(i), d, i, ticks);
    for (int i = 0; ok && i < NS; i++) ok = hipStreamSynchronize(stream_of(i)) == hipSuccess;
    ok = ok && hipGetLastError() == hipSuccess && hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost) == hipSuccess;
    (void)hipFree(d);
    if (!ok) return;
    // streams in the order their kernels started; a stream joins the class whose last kernel ended just before its own began (same queue: back to back),
    // otherwise it opens a class of its own (its kernel began while every class's last kernel was still running: another queue)
    int order[NS], n_g = 0, grp_of[NS], rep[NS];
    long long last_end[NS];
    for (int i = 0; i < NS; i++) order[i] = i;
    std::sort(order, order + NS, [&](int a, int b) { return h[2 * a] < h[2 * b]; });
    for (int k = 0; k < NS; k++) {
        const int i = order[k];
        const long long st = (long long)h[2 * i], en = (long long)h[2 * i + 1];
        if (en - st < ticks / 2) return;                          // (the clock did not run as expected: leave the classes unknown)
        int g = -1;
        for (int x = 0; x < n_g; x++) if (last_end[x] <= st && (g < 0 || last_end[x] > last_end[g])) g = x;
        if (g < 0) { g = n_g++; rep[g] = -1; }
        if (rep[g] < 0 && i < 12) rep[g] = i;                     // (the class's first SIDE stream)
        grp_of[i] = g; last_end[g] = en;
    }
    if (bm2_knob("BM2_QUEUE_PROBE_LOG", 0)) {
        fprintf(stderr, "[bm2] hardware-queue classes of the side streams as made (main stream: %d; %d classes):", grp_of[12], n_g);
        for (int i = 0; i < 12; i++) fprintf(stderr, " %d", grp_of[i]);
        fprintf(stderr, "\n");
    }
    // The launchers address the side streams by fixed small numbers (seeding 0-2, chaining 1-6 and 10-11, an extension phase 0-9) and count on different
    // numbers meaning different queues, none of them the main stream's (k_bwd and k_postfilter_heavy run ON the main stream beside side launches): ONE
    // stream of every queue class first -- the main stream's class last of those --, the streams that double a queue behind them.
    hipStream_t st[12]; int grp[12], n = 0;
    bool taken[12] = {};
    for (int pass = 0; pass < 2; pass++)
        for (int g = 0; g < n_g; g++) {
            if ((g == grp_of[12]) != (pass == 1) || rep[g] < 0) continue;
            st[n] = c->side_stream[rep[g]]; grp[n] = g; taken[rep[g]] = true; c->group_rep[g] = n; n++;
        }
    for (int i = 0; i < 12; i++) if (!taken[i]) { st[n] = c->side_stream[i]; grp[n] = grp_of[i]; n++; }
    for (int i = 0; i < 12; i++) { c->side_stream[i] = st[i]; c->side_group[i] = grp[i]; }
    c->n_side_groups = n_g;
#endif
}
// The side streams of the fork / join launches (seeding, chaining, extension) exist only in contexts that run those stages: a process has
// GPU_MAX_HW_QUEUES hardware queues, its streams share them round-robin, and whatever is queued behind a long kernel on its queue waits for
// it -- a context that only runs the SAM tail's batches (one stream) must not dilute the queues of the contexts that run the hot path.
int bm2_side_streams(bm2_ctx *c) {
    if (c->side_ready) return BM2_OK;                            // (set only when every stream and event below exists)
    if (!c->ev_fork && bm2_check(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming), "hipEventCreate")) { c->ev_fork = nullptr; return BM2_ENODEV; }
    // (Side streams of the long query classes at the highest queue priority, BM2_SIDE_PRIO_MASK: the hot path 74.7 -> 105-139 ms, profiles/r04f_*: removed in round 6.)
    for (int i = 0; i < 12; i++) {
        if (!c->side_stream[i]) {
            const hipError_t e = hipStreamCreateWithFlags(&c->side_stream[i], hipStreamNonBlocking);
            if (bm2_check(e, "hipStreamCreate")) { c->side_stream[i] = nullptr; return BM2_ENODEV; }      // (a later call tries again from here: nothing half-made is ever used)
        }
        if (!c->ev_join[i] && bm2_check(hipEventCreateWithFlags(&c->ev_join[i], hipEventDisableTiming), "hipEventCreate")) { c->ev_join[i] = nullptr; return BM2_ENODEV; }
    }
    probe_side_queues(c);
    c->side_ready = true;
    return BM2_OK;
}
// The dynamic-LDS limit of a kernel is a property of the loaded code object on ONE device: raised once per context (contexts of several
// devices live in one process since the binding drives every visible GPU), and a refusal is an error, not a silent 64 KB launch.
int bm2_raise_lds_limit(bm2_ctx *c, int which, const void *kernel, size_t bytes) {
    if (c->lds_attr_done & (1u << which)) return BM2_OK;
    int rc = bm2_check(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes), "hipFuncSetAttribute(max dynamic LDS)");
    if (!rc) c->lds_attr_done |= 1u << which;
    return rc;
}
static void free_streams(bm2_ctx *c) {
    for (int i = 0; i < 2; i++) { if (c->pin[i]) (void)hipHostFree(c->pin[i]); if (c->pin_ev[i]) (void)hipEventDestroy(c->pin_ev[i]); c->pin[i] = nullptr; c->pin_ev[i] = nullptr; }
    for (int i = 0; i <= BM2_MAX_TIMERS; i++) if (c->ev[i]) (void)hipEventDestroy(c->ev[i]);
    for (int i = 0; i < 12; i++) {
        if (c->side_stream[i]) (void)hipStreamDestroy(c->side_stream[i]);
        if (c->ev_join[i]) (void)hipEventDestroy(c->ev_join[i]);
    }
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ext_stat) { (void)hipHostFree(c->ext_stat); c->ext_stat = nullptr; }
    if (c->stream) (void)hipStreamDestroy(c->stream);
}

// CP_OCC file layout -> device layout (bm2_dev.h), in place: one entry per lane
__global__ void __launch_bounds__(256) k_cp_occ_relayout(CpOcc *occ, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const CpOcc e = occ[i];
    CpOccDev d;
    for (int b = 0; b < 4; b++) { d.q[b].count = e.cp_count[b]; d.q[b].bwt = e.bwt[b]; }
    ((CpOccDev *)occ)[i] = d;
}

// The reference string on the device: four codes per byte (refseq.h).  The host's image -- one code per byte, as fastmap.cpp:873-881 reads the
// .0123 file -- goes through a staging buffer in pieces and is packed by a kernel, 64 codes per thread.  A code above 3 cannot be packed (the
// .0123 file holds none: bntseq.cpp:284 draws a random base for every ambiguous one): the kernel reports it and the caller keeps bytes.
#define REF_PIECE ((size_t)64 << 20)
#define REF_PAD 64
__global__ void __launch_bounds__(256) k_pack_ref(const uint8_t *__restrict__ src, int64_t n_codes, uint8_t *__restrict__ dst, int *bad) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, c0 = t * 64;
    if (c0 >= n_codes) return;
    uint32_t over = 0;
    if (c0 + 64 <= n_codes) {
        uint32_t o[4];
        for (int k = 0; k < 4; k++) {
            const uint4 v = ((const uint4 *)(src + c0))[k];
            const uint32_t w[4] = { v.x, v.y, v.z, v.w };
            uint32_t acc = 0;
            for (int u = 0; u < 4; u++) {
                over |= w[u] & 0xfcfcfcfcu;
                const uint32_t x = w[u] & 0x03030303u;
                acc |= ((x | x >> 6 | x >> 12 | x >> 18) & 0xffu) << (8 * u);
            }
            o[k] = acc;
        }
        *(uint4 *)(dst + (c0 >> 2)) = make_uint4(o[0], o[1], o[2], o[3]);
    } else {
        for (int64_t b = c0; b < n_codes; b += 4) {
            uint32_t acc = 0;
            for (int u = 0; u < 4 && b + u < n_codes; u++) { over |= src[b + u] & 0xfcu; acc |= (uint32_t)(src[b + u] & 3) << (2 * u); }
            dst[b >> 2] = (uint8_t)acc;
        }
    }
    if (over) *bad = 1;
}
static int upload_ref(bm2_ctx *c, const uint8_t *src, size_t n_codes) {
    int rc;
    if (bm2_knob("BM2_REF_BYTES", 0)) return upload(&c->d_ref, src, n_codes, c->stream);        // (one code per byte: for A/B measurements)
    // (REF_PAD bytes of zeros on either side: the lane kernel's 64-bit windows start up to 27 bases before a target's first base and end up to
    //  31 beyond its last, RefPtr::load4 reads one byte beyond the last code's; c->d_ref is the allocation, the codes start REF_PAD bytes in)
    const size_t packed = (n_codes + 3) / 4 + 2 * REF_PAD;
    void *stage = nullptr; int *bad = nullptr;
    if ((rc = bm2_check(hipMalloc(&c->d_ref, packed), "hipMalloc(reference, packed)"))) return rc;
    rc = bm2_check(hipMalloc(&stage, REF_PIECE + 64), "hipMalloc(reference staging)");
    if (!rc) rc = bm2_check(hipMalloc((void **)&bad, 64), "hipMalloc(flag)");
    if (!rc) rc = bm2_check(hipMemsetAsync(bad, 0, 4, c->stream), "memset");
    if (!rc) rc = bm2_check(hipMemsetAsync(c->d_ref, 0, REF_PAD, c->stream), "memset");
    if (!rc) rc = bm2_check(hipMemsetAsync((char *)c->d_ref + (packed - REF_PAD - 16), 0, REF_PAD + 16, c->stream), "memset");
    for (size_t off = 0; !rc && off < n_codes; off += REF_PIECE) {
        const size_t n = n_codes - off < REF_PIECE ? n_codes - off : REF_PIECE;
        rc = bm2_check(hipMemcpyAsync(stage, src + off, n, hipMemcpyHostToDevice, c->stream), "hipMemcpy(reference piece)");
        if (rc) break;
        const size_t thr = (n + 63) / 64;
        hipLaunchKernelGGL(k_pack_ref, dim3((unsigned)((thr + 255) / 256)), dim3(256), 0, c->stream, (const uint8_t *)stage, (int64_t)n,
                           (uint8_t *)c->d_ref + REF_PAD + off / 4, bad);
        rc = bm2_check(hipGetLastError(), "k_pack_ref");
    }
    int h_bad = 0;
    if (!rc) rc = bm2_check(hipMemcpyAsync(&h_bad, bad, 4, hipMemcpyDeviceToHost, c->stream), "D2H flag");
    if (!rc) rc = bm2_check(hipStreamSynchronize(c->stream), "reference upload");
    if (stage) (void)hipFree(stage);
    if (bad) (void)hipFree(bad);
    if (rc) return rc;
    if (h_bad) {        // not a .0123 image: keep what the caller gave
        fprintf(stderr, "[libbm2] note: the reference string holds codes above 3; kept one code per byte on the device\n");
        (void)hipFree(c->d_ref); c->d_ref = nullptr;
        return upload(&c->d_ref, src, n_codes, c->stream);
    }
    c->ix.ref_pk = 1;
    return BM2_OK;
}

// The runtime reads GPU_MAX_HW_QUEUES when it STARTS (default: four hardware queues per process).  The earliest moment this library can speak
// is when it is loaded: it asks for sixteen unless the host chose (eight launches of an extension phase, five of the chaining stage, the
// copies of the neighbouring chunk: with 16 queues the hot path takes 74.8 ms instead of 76.7 and the FASTQ -> SAM leg gains 7 %; with 24 the hot path collapses to 101 ms).  A host that has initialised HIP before loading libbm2 keeps what it had
// -- nothing in the HIP API tells (include/bm2.h says so; bm2_create notes a smaller explicit setting on stderr).
__attribute__((constructor)) static void bm2_ask_for_queues() { setenv("GPU_MAX_HW_QUEUES", "16", 0); }     // (8 / 16 / 24 measured: profiles/r04c, r04d)

extern "C" bm2_ctx *bm2_create(int device, const bm2_index_desc *idx) {
    // The extension stage forks eight concurrent launches per side, each on a stream of its own; streams share the process's hardware
    // queues round-robin and the runtime's default is four.  Eight, unless the caller chose (read when the runtime starts: a process that
    // has already used HIP keeps what it had).
    static std::once_flag queues_once;
    std::call_once(queues_once, []() {
        const char *have = getenv("GPU_MAX_HW_QUEUES");
        if (have && *have && atoi(have) < 8)
            fprintf(stderr, "[libbm2] note: GPU_MAX_HW_QUEUES=%s -- the extension stage forks eight concurrent launches; with fewer hardware queues they "
                            "run partly one after the other (about 2 ms per million-read chunk)\n", have);
    });
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        bm2_set_error("no HIP device visible: libbm2 has no CPU fallback");
        return nullptr;
    }
    if (device < 0 || device >= ndev) { bm2_set_error("device %d out of range (%d visible)", device, ndev); return nullptr; }
    if (bm2_check(hipSetDevice(device), "hipSetDevice")) return nullptr;
    // BM2_BLOCKING_SYNC (the host's decision, read once per device; default 1): a thread that waits for the device sleeps instead of spinning.  For a host
    // whose CPUs are all busy with its own work (the S1 binding: every worker thread of the reference computes while one of them waits for a batch; the
    // FASTQ -> SAM leg: parsers and tail workers beside the device workers) a spinning waiter is a CPU taken from the work -- and under a CPU-time quota it
    // is what gets the whole process throttled.  profiles/r06w_e2e_blocking_sync_ab.txt: the leg's host CPU 1.19-1.20 -> 1.03-1.13 s per chunk, its rate
    // and the hot path's step (59.4-59.7 ms both ways) unchanged.  0: HIP's default (spin).
    if (bm2_knob("BM2_BLOCKING_SYNC", 1)) {
        static std::mutex flag_mu; static unsigned long long flagged = 0;
        std::lock_guard<std::mutex> l(flag_mu);
        if (device < 64 && !(flagged >> device & 1)) { flagged |= 1ULL << device; if (hipSetDeviceFlags(hipDeviceScheduleBlockingSync) != hipSuccess) (void)hipGetLastError(); }
    }
    bm2_ctx *c = new (std::nothrow) bm2_ctx();
    if (!c) return nullptr;
    c->device = device;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) c->n_cu = prop.multiProcessorCount;
    if (make_streams(c)) { delete c; return nullptr; }
    if (idx) {
        const int64_t nocc = (idx->ref_len >> 6) + 1, nsa = (idx->ref_len >> 3) + 1;
        int rc = 0;
        rc = rc ? rc : upload(&c->d_cp_occ, (const CpOcc *)idx->cp_occ, (size_t)nocc, c->stream);
        if (!rc) {
            hipLaunchKernelGGL(k_cp_occ_relayout, dim3((unsigned)((nocc + 255) / 256)), dim3(256), 0, c->stream, (CpOcc *)c->d_cp_occ, nocc);
            rc = bm2_check(hipGetLastError(), "k_cp_occ_relayout");
        }
        rc = rc ? rc : upload(&c->d_sa_ms, idx->sa_ms_byte, (size_t)nsa, c->stream);
        rc = rc ? rc : upload(&c->d_sa_ls, idx->sa_ls_word, (size_t)nsa, c->stream);
        rc = rc ? rc : upload_ref(c, idx->ref_string, (size_t)(2 * idx->l_pac));
        rc = rc ? rc : upload(&c->d_ann_off, idx->ann_offset, (size_t)idx->n_seqs, c->stream);
        rc = rc ? rc : upload(&c->d_ann_len, idx->ann_len, (size_t)idx->n_seqs, c->stream);
        rc = rc ? rc : upload(&c->d_ann_alt, idx->ann_is_alt, (size_t)idx->n_seqs, c->stream);
        rc = rc ? rc : bm2_check(hipStreamSynchronize(c->stream), "index upload");
        if (rc) { bm2_destroy(c); return nullptr; }
        DevIndex &ix = c->ix;
        ix.cp_occ = (const CpOccDev *)c->d_cp_occ; ix.sa_ms_byte = (const int8_t *)c->d_sa_ms;
        ix.sa_ls_word = (const uint32_t *)c->d_sa_ls; ix.ref_string = (const uint8_t *)c->d_ref + (ix.ref_pk ? REF_PAD : 0);
        ix.ann_offset = (const int64_t *)c->d_ann_off; ix.ann_len = (const int32_t *)c->d_ann_len;
        ix.ann_is_alt = (const int32_t *)c->d_ann_alt;
        ix.ref_len = idx->ref_len; ix.l_pac = idx->l_pac; ix.sentinel_index = idx->sentinel_index;
        for (int i = 0; i < 5; i++) ix.count[i] = idx->count[i] + 1;      // FMI_search.cpp:433-436
        ix.n_seqs = idx->n_seqs;
        c->has_index = true;
        bm2_ensure_subs(c, bm2_knob("BM2_N_SUB", BM2_N_SUB));
    }
    return c;
}

// extra contexts for sub-batch pipelining (pipeline.hip): same index replica, own streams and workspaces; made on demand
int bm2_ensure_subs(bm2_ctx *c, int n_sub) {
    if (!c || c->is_sub || !c->has_index) return 1;
    if (n_sub > 8) n_sub = 8;
    while ((int)c->subs.size() + 1 < n_sub) {
        bm2_ctx *k = new (std::nothrow) bm2_ctx();
        if (!k) break;
        k->device = c->device; k->n_cu = c->n_cu; k->ix = c->ix; k->has_index = true; k->is_child = true; k->is_sub = true;
        if (make_streams(k)) { delete k; break; }
        c->subs.push_back(k);
    }
    return 1 + (int)c->subs.size();
}

// A second context on the same device that SHARES the parent's index replica (its own streams and workspaces): lets a caller
// overlap two chunks on one GPU -- the device stages of chunk n+1 beside the device batches of chunk n's SAM tail -- or split one
// chunk over several contexts without a second 17 GB upload.  Valid while the parent lives; destroy it first.
extern "C" bm2_ctx *bm2_create_shared(bm2_ctx *parent) {
    if (!parent || parent->is_child) { bm2_set_error("bm2_create_shared: needs a context made by bm2_create"); return nullptr; }
    if (bm2_check(hipSetDevice(parent->device), "hipSetDevice")) return nullptr;
    bm2_ctx *k = new (std::nothrow) bm2_ctx();
    if (!k) return nullptr;
    k->device = parent->device; k->n_cu = parent->n_cu; k->ix = parent->ix; k->has_index = parent->has_index; k->is_child = true;
    if (make_streams(k)) { delete k; return nullptr; }
    if (k->has_index) bm2_ensure_subs(k, bm2_knob("BM2_N_SUB", BM2_N_SUB));
    return k;
}

// The context's main stream at a hardware queue priority (level > 0: highest, < 0: lowest, 0: default).  A pipeline that runs the short
// device batches of the SAM tail (mate rescue, CIGAR) beside the seeding .. extension of the next chunk gives the tail's contexts the
// higher priority: their workgroups are dispatched first, the chunk they belong to leaves the pipeline sooner.
extern "C" int bm2_set_stream_priority(bm2_ctx *c, int level) {
    if (!c) { bm2_set_error("bm2_set_stream_priority: no context"); return BM2_EINVAL; }
    int rc = bm2_check(hipSetDevice(c->device), "hipSetDevice");
    if (rc) return rc;
    int least = 0, greatest = 0;
    if ((rc = bm2_check(hipDeviceGetStreamPriorityRange(&least, &greatest), "hipDeviceGetStreamPriorityRange"))) return rc;
    const int prio = level > 0 ? greatest : level < 0 ? least : 0;
    hipStream_t fresh = nullptr;
    if ((rc = bm2_check(hipStreamCreateWithPriority(&fresh, hipStreamNonBlocking, prio), "hipStreamCreateWithPriority"))) return rc;
    if (c->stream) { (void)hipStreamSynchronize(c->stream); (void)hipStreamDestroy(c->stream); }
    c->stream = fresh;
    return BM2_OK;
}

extern "C" void bm2_destroy(bm2_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (bm2_ctx *k : c->subs) {
        if (k->stream) (void)hipStreamSynchronize(k->stream);
        bm2_batch_destroy(k);
        free_streams(k);
        delete k;
    }
    c->subs.clear();
    bm2_batch_destroy(c);
    void *ps[] = { c->d_cp_occ, c->d_sa_ms, c->d_sa_ls, c->d_ref, c->d_ann_off, c->d_ann_len, c->d_ann_alt };
    if (!c->is_child) for (void *p : ps) if (p) (void)hipFree(p);          // a shared context does not own the replica
    bm2_release(c->b_pairs); bm2_release(c->b_pairs2); bm2_release(c->b_ref); bm2_release(c->b_qer); bm2_release(c->b_misc); bm2_release(c->b_scan);
    free_streams(c);
    delete c;
}

// ---- S1 --------------------------------------------------------------------------------------------------------
extern "C" int bm2_bsw(bm2_ctx *c, bm2_seqpair_t *pairs, const uint8_t *ref, int64_t ref_bytes, const uint8_t *qer,
                       int64_t qer_bytes, int32_t n, int32_t w, const bm2_sw_params *p) {
    if (!c || !pairs || !p || n < 0 || w < 0 || (n > 0 && (!ref || !qer))) { bm2_set_error("bm2_bsw: bad argument"); return BM2_EINVAL; }
    if (n == 0) return BM2_OK;
    if (p->e_del <= 0 || p->e_ins <= 0) { bm2_set_error("bm2_bsw: gap extension penalties must be > 0"); return BM2_EINVAL; }
    int rc = bm2_check(hipSetDevice(c->device), "hipSetDevice");
    if (rc) return rc;
    if ((rc = bm2_reserve(c->b_pairs, (size_t)n * sizeof(bm2_seqpair_t)))) return rc;
    if ((rc = bm2_reserve(c->b_ref, (size_t)ref_bytes + 64))) return rc;
    if ((rc = bm2_reserve(c->b_qer, (size_t)qer_bytes + 64))) return rc;
    SwParams P;
    P.o_del = p->o_del; P.e_del = p->e_del; P.o_ins = p->o_ins; P.e_ins = p->e_ins; P.zdrop = p->zdrop;
    P.end_bonus = p->end_bonus; P.max_sc = p->w_match;
    for (int i = 0; i < 25; i++) P.mat[i] = p->mat[i];
    c->n_bsw = 0;                                               // (the scratch buffers are shared with the resident variant)
    hipStream_t s = c->stream;
    rc = bm2_check(hipMemcpyAsync(c->b_pairs.p, pairs, (size_t)n * sizeof(bm2_seqpair_t), hipMemcpyHostToDevice, s), "H2D pairs");
    if (!rc) rc = bm2_check(hipMemcpyAsync(c->b_ref.p, ref, (size_t)ref_bytes, hipMemcpyHostToDevice, s), "H2D ref");
    if (!rc) rc = bm2_check(hipMemcpyAsync(c->b_qer.p, qer, (size_t)qer_bytes, hipMemcpyHostToDevice, s), "H2D qer");
    if (!rc) rc = bm2_launch_bsw_pairs(c, (bm2_seqpair_t *)c->b_pairs.p, (const uint8_t *)c->b_ref.p, (const uint8_t *)c->b_qer.p, n, w, P, nullptr);
    if (!rc) rc = bm2_check(hipMemcpyAsync(pairs, c->b_pairs.p, (size_t)n * sizeof(bm2_seqpair_t), hipMemcpyDeviceToHost, s), "D2H pairs");
    if (!rc) rc = bm2_check(hipStreamSynchronize(s), "bm2_bsw sync");
    return rc;
}

// S1 with the batch RESIDENT (config 2 of BASELINE.json = the banded-SW kernel alone, timed without the PCIe copies): upload once, run
// any number of times -- the kernel only writes the six output fields of a pair -- download when done.  bm2_bsw_run reports the kernel's
// duration from HIP events on the launch stream and, if asked, the DP cells it computed (its own counter: one same-address atomic per
// pair, which a timed run should not pay -- ask for the cells in a run of their own).
static SwParams sw_params_of(const bm2_sw_params *p) {
    SwParams P;
    P.o_del = p->o_del; P.e_del = p->e_del; P.o_ins = p->o_ins; P.e_ins = p->e_ins; P.zdrop = p->zdrop;
    P.end_bonus = p->end_bonus; P.max_sc = p->w_match;
    for (int i = 0; i < 25; i++) P.mat[i] = p->mat[i];
    return P;
}
extern "C" int bm2_bsw_upload(bm2_ctx *c, const bm2_seqpair_t *pairs, const uint8_t *ref, int64_t ref_bytes, const uint8_t *qer,
                              int64_t qer_bytes, int32_t n) {
    if (!c || n < 0 || (n > 0 && (!pairs || !ref || !qer)) || ref_bytes < 0 || qer_bytes < 0) { bm2_set_error("bm2_bsw_upload: bad argument"); return BM2_EINVAL; }
    int rc = bm2_check(hipSetDevice(c->device), "hipSetDevice");
    if (rc) return rc;
    c->n_bsw = 0;
    if (n == 0) return BM2_OK;
    if ((rc = bm2_reserve(c->b_pairs, (size_t)n * sizeof(bm2_seqpair_t)))) return rc;
    if ((rc = bm2_reserve(c->b_ref, (size_t)ref_bytes + 64))) return rc;
    if ((rc = bm2_reserve(c->b_qer, (size_t)qer_bytes + 64))) return rc;
    if ((rc = bm2_reserve(c->b_misc, 64))) return rc;
    if ((rc = bm2_copy_h2d(c, c->b_pairs.p, pairs, (size_t)n * sizeof(bm2_seqpair_t)))) return rc;
    if ((rc = bm2_copy_h2d(c, c->b_ref.p, ref, (size_t)ref_bytes))) return rc;
    if ((rc = bm2_copy_h2d(c, c->b_qer.p, qer, (size_t)qer_bytes))) return rc;
    c->n_bsw = n;
    return BM2_OK;
}
extern "C" int bm2_bsw_run(bm2_ctx *c, int32_t w, const bm2_sw_params *p, float *kernel_ms, int64_t *cells) {
    if (!c || !p || w < 0) { bm2_set_error("bm2_bsw_run: bad argument"); return BM2_EINVAL; }
    if (p->e_del <= 0 || p->e_ins <= 0) { bm2_set_error("bm2_bsw_run: gap extension penalties must be > 0"); return BM2_EINVAL; }
    if (kernel_ms) *kernel_ms = 0.f;
    if (cells) *cells = 0;
    if (c->n_bsw == 0) return BM2_OK;
    int rc = bm2_check(hipSetDevice(c->device), "hipSetDevice");
    if (rc) return rc;
    hipStream_t s = c->stream;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if ((rc = bm2_check(hipEventCreate(&e0), "hipEventCreate")) || (rc = bm2_check(hipEventCreate(&e1), "hipEventCreate"))) { if (e0) (void)hipEventDestroy(e0); return rc; }
    unsigned long long *d_cells = (unsigned long long *)c->b_misc.p;
    rc = bm2_check(hipMemsetAsync(d_cells, 0, sizeof(unsigned long long), s), "memset cells");
    if (!rc) rc = bm2_check(hipEventRecord(e0, s), "hipEventRecord");
    if (!rc) rc = bm2_launch_bsw_pairs(c, (bm2_seqpair_t *)c->b_pairs.p, (const uint8_t *)c->b_ref.p, (const uint8_t *)c->b_qer.p, c->n_bsw, w, sw_params_of(p), cells ? d_cells : nullptr);
    if (!rc) rc = bm2_check(hipEventRecord(e1, s), "hipEventRecord");
    unsigned long long h_cells = 0;
    if (!rc) rc = bm2_check(hipMemcpyAsync(&h_cells, d_cells, sizeof h_cells, hipMemcpyDeviceToHost, s), "D2H cells");
    if (!rc) rc = bm2_check(hipStreamSynchronize(s), "bm2_bsw_run sync");
    float ms = 0.f;
    if (!rc) rc = bm2_check(hipEventElapsedTime(&ms, e0, e1), "hipEventElapsedTime");
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (!rc) { if (kernel_ms) *kernel_ms = ms; if (cells) *cells = (int64_t)h_cells; }
    return rc;
}
extern "C" int bm2_bsw_download(bm2_ctx *c, bm2_seqpair_t *pairs, int32_t n) {
    if (!c || n < 0 || (n > 0 && !pairs)) { bm2_set_error("bm2_bsw_download: bad argument"); return BM2_EINVAL; }
    if (n != c->n_bsw) { bm2_set_error("bm2_bsw_download: %d pairs asked for, %d are resident", n, c->n_bsw); return BM2_EINVAL; }
    if (n == 0) return BM2_OK;
    int rc = bm2_check(hipSetDevice(c->device), "hipSetDevice");
    if (rc) return rc;
    return bm2_copy_d2h(c, pairs, c->b_pairs.p, (size_t)n * sizeof(bm2_seqpair_t));
}
