// Fake <hip/hip_runtime.h> -- TEST INFRASTRUCTURE (tools/emu): lets the device sources of bwa-mem2_amd/csrc compile for the host so
// that their logic can be executed without a GPU.  One OS thread per GPU thread, the blocks of a launch one after the other, the
// threads of a block at once; every wave-level primitive is a rendezvous of the 64 threads of a wavefront.
//
// Supported: wave-uniform use of the cross-lane primitives (all 64 lanes of a wavefront call the same primitive in the same order --
// how the kernels here are written: converged loops, predicates instead of divergent branches around collectives).  A primitive
// called by only some lanes of a wave deadlocks here (the harness's timeout reports it); a wavefront whose threads have all
// returned simply stops taking part.  What this cannot see: anything the compiler or the memory system does (register pressure,
// structurisation of loops, visibility without fences, timing).
#pragma once
#include <pthread.h>
#include <sched.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <functional>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ static                 /* blocks run one at a time: one static copy IS the block's LDS */

struct uint2 { unsigned x, y; }; struct uint4 { unsigned x, y, z, w; }; struct int2 { int x, y; }; struct int4 { int x, y, z, w; };
static inline uint4 make_uint4(unsigned a, unsigned b, unsigned c, unsigned d) { uint4 v = { a, b, c, d }; return v; }
static inline int2 make_int2(int a, int b) { int2 v = { a, b }; return v; }
static inline uint2 make_uint2(unsigned a, unsigned b) { uint2 v = { a, b }; return v; }
struct ulonglong2 { unsigned long long x, y; };
static inline ulonglong2 make_ulonglong2(unsigned long long a, unsigned long long b) { ulonglong2 v = { a, b }; return v; }
static inline int max(int a, int b) { return a > b ? a : b; }
static inline int min(int a, int b) { return a < b ? a : b; }
static inline long long max(long long a, long long b) { return a > b ? a : b; }
static inline long long min(long long a, long long b) { return a < b ? a : b; }
static inline long max(long a, long b) { return a > b ? a : b; }
static inline long min(long a, long b) { return a < b ? a : b; }
static inline unsigned max(unsigned a, unsigned b) { return a > b ? a : b; }
static inline unsigned min(unsigned a, unsigned b) { return a < b ? a : b; }

struct emu_dim3 { unsigned x, y, z; emu_dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };
typedef emu_dim3 dim3;
// A barrier that spins with sched_yield: rendezvous are so frequent that sleeping in a futex (pthread_barrier) dominates the run time.
struct EmuBarrier {
    std::atomic<int> count{0}, gen{0}; int n = 0;
    void init(int n_) { n = n_; count = 0; gen = 0; }
    void wait() {
        const int g = gen.load(std::memory_order_acquire);
        if (count.fetch_add(1, std::memory_order_acq_rel) + 1 == n) { count.store(0, std::memory_order_relaxed); gen.fetch_add(1, std::memory_order_acq_rel); }
        else { int spins = 0; while (gen.load(std::memory_order_acquire) == g) { if (++spins > 64) sched_yield(); } }
    }
};
struct EmuWave { EmuBarrier bar, rowbar[4]; pthread_mutex_t mu; long long slot[64]; int site[64]; long long rslot[64]; };
struct EmuThread { emu_dim3 tid, bid, bdim, gdim; EmuWave *wave; EmuBarrier *block_bar; int lane; };
extern thread_local EmuThread emu_t;
#define threadIdx (emu_t.tid)
#define blockIdx  (emu_t.bid)
#define blockDim  (emu_t.bdim)
#define gridDim   (emu_t.gdim)

// ---- rendezvous of a wavefront: publish, wait, read, wait ----
static inline void emu_wsync() { emu_t.wave->bar.wait(); }
extern thread_local int emu_site;         // source line of the primitive being executed (set by the macros below)
template <class F> static inline auto emu_exchange(long long v, F read) -> decltype(read((const long long *)0)) {
    EmuWave *w = emu_t.wave;
    w->slot[emu_t.lane] = v; w->site[emu_t.lane] = emu_site; emu_wsync();
    for (int i = 0; i < 64; ++i)
        if (w->site[i] != emu_site) {         // lanes met at different primitives: the code is not wave-uniform here
            fprintf(stderr, "emu: lanes %d and %d of a wavefront met at different primitives (source lines %d and %d): a collective is "
                            "called from divergent lanes\n", emu_t.lane, i, emu_site, w->site[i]);
            abort();
        }
    auto r = read((const long long *)w->slot); emu_wsync();
    return r;
}
static inline unsigned long long emu_ballot(bool p) {
    return emu_exchange(p ? 1 : 0, [](const long long *s) { unsigned long long m = 0; for (int i = 0; i < 64; ++i) if (s[i]) m |= 1ull << i; return m; });
}
static inline int emu_readlane(int v, int l) { return emu_exchange(v, [l](const long long *s) { return (int)s[l & 63]; }); }
static inline int emu_readfirstlane(int v) { return emu_readlane(v, 0); }              // all lanes active by contract
static inline int emu_shfl(int v, int src, int width = 64) {
    const int base = emu_t.lane & ~(width - 1);
    return emu_exchange(v, [=](const long long *s) { return (int)s[base + (src & (width - 1))]; });
}
static inline int emu_shfl_xor(int v, int m, int width = 64) {
    const int l = emu_t.lane, base = l & ~(width - 1);
    return emu_exchange(v, [=](const long long *s) { return (int)s[base + ((l ^ m) & (width - 1))]; });
}
// DPP: the controls the kernels use.  row_mask bit r enables row r (16 lanes); a disabled lane, or one whose source lies outside
// (and !bound_ctrl), keeps `old`; bound_ctrl turns an outside source into 0.
static inline int emu_dpp(int old, int v, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
    const int l = emu_t.lane;
    return emu_exchange(v, [=](const long long *s) -> int {
        if (!((row_mask >> (l >> 4)) & 1) || !((bank_mask >> ((l >> 2) & 3)) & 1)) return old;
        int src = -1;
        if (ctrl >= 0 && ctrl <= 0xff) src = (l & ~3) | ((ctrl >> (2 * (l & 3))) & 3);                       // quad_perm
        else if (ctrl >= 0x101 && ctrl <= 0x10f) { const int k = (l & 15) + (ctrl - 0x100); src = k < 16 ? (l & ~15) | k : -1; }   // row_shl
        else if (ctrl >= 0x111 && ctrl <= 0x11f) { const int k = (l & 15) - (ctrl - 0x110); src = k >= 0 ? (l & ~15) | k : -1; }   // row_shr
        else if (ctrl == 0x138) src = l - 1;                                                                   // wave_shr:1
        else if (ctrl == 0x130) src = l + 1 < 64 ? l + 1 : -1;                                                 // wave_shl:1
        else if (ctrl == 0x142) src = (l >> 4) > 0 ? ((l >> 4) - 1) * 16 + 15 : -1;                            // row_bcast:15 (lane 15 of the previous row)
        else if (ctrl == 0x143) src = l >= 32 ? 31 : -1;                                                       // row_bcast:31
        else if (ctrl == 0x140) src = (l & ~15) | (15 - (l & 15));                                             // row_mirror
        else { fprintf(stderr, "emu_dpp: control %#x not modelled\n", ctrl); abort(); }
        if (src < 0) return bound_ctrl ? 0 : old;
        return (int)s[src];
    });
}
#define EMU_AT(expr) (emu_site = __LINE__, (expr))
#define __ballot(p) EMU_AT(emu_ballot(p))
#define __shfl(...) EMU_AT(emu_shfl(__VA_ARGS__))
#define __shfl_xor(...) EMU_AT(emu_shfl_xor(__VA_ARGS__))
#define __builtin_amdgcn_readlane(v, l) EMU_AT(emu_readlane((int)(v), (int)(l)))
#define __builtin_amdgcn_readfirstlane(v) EMU_AT(emu_readfirstlane((int)(v)))
#define __builtin_amdgcn_update_dpp(old, v, ctrl, rm, bm, bc) EMU_AT(emu_dpp((int)(old), (int)(v), (ctrl), (rm), (bm), (bc)))
#define __builtin_amdgcn_mov_dpp(v, ctrl, rm, bm, bc) EMU_AT(emu_dpp(0, (int)(v), (ctrl), (rm), (bm), (bc)))
// v_perm_b32: byte i of the result = byte sel[i] of {s0 (bytes 4..7), s1 (bytes 0..3)}; 8..11 -> sign fill of 16-bit word sel - 8, 12 -> 0x00, 13.. -> 0xff
static inline unsigned emu_perm(unsigned s0, unsigned s1, unsigned sel) {
    const unsigned long long src = ((unsigned long long)s0 << 32) | s1;
    unsigned r = 0;
    for (int i = 0; i < 4; i++) {
        const unsigned v = (sel >> (8 * i)) & 0xffu;
        unsigned b;
        if (v < 8) b = (unsigned)(src >> (8 * v)) & 0xffu;
        else if (v < 12) b = ((src >> (16 * (v - 8) + 15)) & 1) ? 0xffu : 0u;       // sign fill of word v - 8
        else b = v == 12 ? 0u : 0xffu;
        r |= b << (8 * i);
    }
    return r;
}
#define __builtin_amdgcn_perm(s0, s1, sel) emu_perm((unsigned)(s0), (unsigned)(s1), (unsigned)(sel))
#define __builtin_amdgcn_s_setprio(x) ((void)0)
#define __builtin_amdgcn_wave_barrier() emu_wsync()
#define __builtin_amdgcn_fence(...) std::atomic_thread_fence(std::memory_order_seq_cst)
#define __builtin_amdgcn_mbcnt_lo(m, c) ((int)(c) + __builtin_popcount((unsigned)(m) & (emu_t.lane >= 32 ? 0xffffffffu : ((1u << emu_t.lane) - 1u))))
#define __builtin_amdgcn_mbcnt_hi(m, c) ((int)(c) + (emu_t.lane > 32 ? __builtin_popcount((unsigned)(m) & ((1u << (emu_t.lane - 32)) - 1u)) : 0))
#define __builtin_amdgcn_inverse_ballot_w64(m) ((((unsigned long long)(m)) >> emu_t.lane) & 1ull)
#define __popcll(x) __builtin_popcountll(x)
#define __popc(x) __builtin_popcount(x)
static inline int __clz(int x) { return x ? __builtin_clz((unsigned)x) : 32; }
#define __any(p) (EMU_AT(emu_ballot(p)) != 0)
#define __all(p) (EMU_AT(emu_ballot(!(p))) == 0)
#define __ffsll(x) __builtin_ffsll(x)
#define __clzll(x) __builtin_clzll(x)
#define __HIP_MEMORY_SCOPE_AGENT 0
#define __hip_atomic_load(p, order, scope) __atomic_load_n(p, __ATOMIC_SEQ_CST)
#define __hip_atomic_store(p, v, order, scope) __atomic_store_n(p, v, __ATOMIC_SEQ_CST)
#define __builtin_amdgcn_s_sleep(n) ((void)0)
#define __threadfence() std::atomic_thread_fence(std::memory_order_seq_cst)
#define __threadfence_block() std::atomic_thread_fence(std::memory_order_seq_cst)
static inline void __syncthreads() { emu_t.block_bar->wait(); }

template <class T> static inline T atomicAdd(T *p, T v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
static inline unsigned long long atomicAdd(unsigned long long *p, unsigned long long v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
static inline long long wall_clock64() { return 0; }
template <class T> static inline T atomicCAS(T *p, T expected, T desired) { __atomic_compare_exchange_n(p, &expected, desired, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST); return expected; }
template <class T> static inline T atomicMin(T *p, T v) { T o = __atomic_load_n(p, __ATOMIC_SEQ_CST); while (v < o && !__atomic_compare_exchange_n(p, &o, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {} return o; }
template <class T> static inline T atomicOr(T *p, T v) { return __atomic_fetch_or(p, v, __ATOMIC_SEQ_CST); }
template <class T> static inline T atomicMax(T *p, T v) { T o = __atomic_load_n(p, __ATOMIC_SEQ_CST); while (o < v && !__atomic_compare_exchange_n(p, &o, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {} return o; }

// A collective that the device code calls from DIVERGENT lanes (smem.hip: wave_alloc -- ballot(1) there means "whoever is here with
// me") has no fixed set of participants to wait for.  Any grouping of the callers is a legal execution on the GPU, so the emu build
// routes such functions to a one-lane-at-a-time version: every caller is its own group, serialised per wavefront.
template <int BATCH, class Pool> static inline int64_t emu_wave_alloc(Pool *wp, unsigned long long *cursor) {
    pthread_mutex_lock(&emu_t.wave->mu);
    int64_t id;
    if (wp->pos < wp->end) { id = wp->pos; wp->pos = id + 1; }
    else { id = (int64_t)__atomic_fetch_add(cursor, (unsigned long long)BATCH, __ATOMIC_SEQ_CST); wp->pos = id + 1; wp->end = id + BATCH; }
    pthread_mutex_unlock(&emu_t.wave->mu);
    return id;
}

// Primitives scoped to a 16-lane DPP row (matesw_dev.h: one task per row, the four rows of a wavefront each in their own control flow --
// divergent as a wavefront, uniform as a row).  On the GPU they are DPP row_shr / a ballot masked to the row / shuffles of width 16,
// which only ever involve the lanes of the row that is executing; here they rendezvous the 16 threads of the row.
#define BM2_EMU_ROW_PRIMS 1
static inline void emu_rsync() { emu_t.wave->rowbar[emu_t.lane >> 4].wait(); }
template <class F> static inline int emu_row_exchange(long long v, F read) {
    EmuWave *w = emu_t.wave;
    w->rslot[emu_t.lane] = v; emu_rsync();
    const int r = read((const long long *)(w->rslot + (emu_t.lane & 48)), emu_t.lane & 15); emu_rsync();
    return r;
}
static inline int row_shr1(int v) { return emu_row_exchange(v, [](const long long *s, int k) { return k ? (int)s[k - 1] : 0; }); }
static inline bool row_any(bool p) { return emu_row_exchange(p, [](const long long *s, int) { int a = 0; for (int i = 0; i < 16; ++i) a |= (int)s[i]; return a; }) != 0; }
static inline int row_xor(int v, int m) { return emu_row_exchange(v, [m](const long long *s, int k) { return (int)s[(k ^ m) & 15]; }); }
static inline int row_first(int v) { return emu_row_exchange(v, [](const long long *s, int) { return (int)s[0]; }); }

// ---- just enough of the runtime API for the launchers to compile ----
typedef int hipError_t; typedef void *hipStream_t; typedef void *hipEvent_t;
enum { hipSuccess = 0, hipErrorOutOfMemory = 2 };
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice };
static inline const char *hipGetErrorString(hipError_t) { return "emulated"; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
enum { hipDeviceScheduleBlockingSync = 4 };
static inline hipError_t hipSetDeviceFlags(unsigned) { return hipSuccess; }
static inline hipError_t hipMalloc(void **p, size_t n) { *p = calloc(1, n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
static inline hipError_t hipFree(void *p) { free(p); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t = 0) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t = 0) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipHostMalloc(void **p, size_t n, unsigned = 0) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
static inline hipError_t hipHostFree(void *p) { free(p); return hipSuccess; }
enum { hipHostMallocPortable = 1 };
enum hipMemoryType { hipMemoryTypeUnregistered = 0, hipMemoryTypeHost = 1, hipMemoryTypeDevice = 2 };
struct hipPointerAttribute_t { hipMemoryType type; };
static inline hipError_t hipPointerGetAttributes(hipPointerAttribute_t *a, const void *) { a->type = hipMemoryTypeUnregistered; return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t = 0) { return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }

static inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2 };
static inline hipError_t hipStreamCreate(hipStream_t *s) { *s = (hipStream_t)1; return hipSuccess; }
static inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = (hipStream_t)1; return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
static inline hipError_t hipDeviceGetStreamPriorityRange(int *l, int *g) { *l = 0; *g = 0; return hipSuccess; }
static inline hipError_t hipStreamCreateWithPriority(hipStream_t *s, unsigned, int) { *s = (hipStream_t)1; return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t *e) { *e = (hipEvent_t)1; return hipSuccess; }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = (hipEvent_t)1; return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }
static inline hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
static inline hipError_t hipFuncSetAttribute(const void *, hipFuncAttribute, int) { return hipSuccess; }
struct hipDeviceProp_t { int multiProcessorCount; };
static inline hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int) { p->multiProcessorCount = 2; return hipSuccess; }   // few blocks per launch
extern char emu_dyn_lds[];                /* dynamic LDS of the running block (160 KB) */

// ---- launch: blocks in sequence, the threads of a block as OS threads ----
void emu_run_block(unsigned n_threads, emu_dim3 bid, emu_dim3 bdim, emu_dim3 gdim, const std::function<void()> &body);
extern size_t emu_dyn_lds_bytes;          // what the launch asked for (the harness's definition of the extern array must be that big)
#include <mutex>
std::mutex &emu_launch_mutex();           // one launch at a time: the static __shared__ copies and the dynamic LDS buffer are per process
template <class K, class... A> static inline void emu_launch(const char *name, K kernel, emu_dim3 grid, emu_dim3 block, size_t lds, hipStream_t, A... args) {
    std::lock_guard<std::mutex> one_launch(emu_launch_mutex());       // (host threads driving different contexts launch concurrently)
    emu_dyn_lds_bytes = lds;
    if (getenv("EMU_TRACE")) fprintf(stderr, "[emu] %s <<<%u, %u, %zu>>>\n", name, grid.x * grid.y * grid.z, block.x * block.y * block.z, lds);
    const unsigned nt = block.x * block.y * block.z;
    for (unsigned bz = 0; bz < grid.z; ++bz) for (unsigned by = 0; by < grid.y; ++by) for (unsigned bx = 0; bx < grid.x; ++bx)
        emu_run_block(nt, emu_dim3(bx, by, bz), block, grid, [&]() { kernel(args...); });
}
#define hipLaunchKernelGGL(kernel, grid, block, lds, stream, ...) emu_launch(#kernel, kernel, grid, block, lds, stream, ##__VA_ARGS__)
