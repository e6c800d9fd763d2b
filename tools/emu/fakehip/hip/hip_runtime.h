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
#define EMU_EXTERN_SHARED extern          /* `extern __shared__ T name[]` (rewritten by the emu build) -> defined by the harness */

struct emu_dim3 { unsigned x, y, z; emu_dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };
typedef emu_dim3 dim3;
struct EmuWave { pthread_barrier_t bar; int alive; long long slot[64]; };
struct EmuThread { emu_dim3 tid, bid, bdim, gdim; EmuWave *wave; pthread_barrier_t *block_bar; int lane; };
extern thread_local EmuThread emu_t;
#define threadIdx (emu_t.tid)
#define blockIdx  (emu_t.bid)
#define blockDim  (emu_t.bdim)
#define gridDim   (emu_t.gdim)

// ---- rendezvous of a wavefront: publish, wait, read, wait ----
static inline void emu_wsync() { pthread_barrier_wait(&emu_t.wave->bar); }
template <class F> static inline auto emu_exchange(long long v, F read) -> decltype(read((const long long *)0)) {
    emu_t.wave->slot[emu_t.lane] = v; emu_wsync();
    auto r = read((const long long *)emu_t.wave->slot); emu_wsync();
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
#define __ballot(p) emu_ballot(p)
#define __shfl(...) emu_shfl(__VA_ARGS__)
#define __shfl_xor(...) emu_shfl_xor(__VA_ARGS__)
#define __builtin_amdgcn_readlane(v, l) emu_readlane((int)(v), (int)(l))
#define __builtin_amdgcn_readfirstlane(v) emu_readfirstlane((int)(v))
#define __builtin_amdgcn_update_dpp(old, v, ctrl, rm, bm, bc) emu_dpp((int)(old), (int)(v), (ctrl), (rm), (bm), (bc))
#define __builtin_amdgcn_mov_dpp(v, ctrl, rm, bm, bc) emu_dpp(0, (int)(v), (ctrl), (rm), (bm), (bc))
#define __builtin_amdgcn_wave_barrier() emu_wsync()
#define __builtin_amdgcn_fence(...) std::atomic_thread_fence(std::memory_order_seq_cst)
#define __builtin_amdgcn_mbcnt_lo(m, c) ((int)(c) + __builtin_popcount((unsigned)(m) & (emu_t.lane >= 32 ? 0xffffffffu : ((1u << emu_t.lane) - 1u))))
#define __builtin_amdgcn_mbcnt_hi(m, c) ((int)(c) + (emu_t.lane > 32 ? __builtin_popcount((unsigned)(m) & ((1u << (emu_t.lane - 32)) - 1u)) : 0))
#define __builtin_amdgcn_inverse_ballot_w64(m) ((((unsigned long long)(m)) >> emu_t.lane) & 1ull)
#define __popcll(x) __builtin_popcountll(x)
#define __ffsll(x) __builtin_ffsll(x)
#define __clzll(x) __builtin_clzll(x)
#define __threadfence() std::atomic_thread_fence(std::memory_order_seq_cst)
#define __threadfence_block() std::atomic_thread_fence(std::memory_order_seq_cst)
static inline void __syncthreads() { pthread_barrier_wait(emu_t.block_bar); }

template <class T> static inline T atomicAdd(T *p, T v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
static inline unsigned long long atomicAdd(unsigned long long *p, unsigned long long v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
template <class T> static inline T atomicOr(T *p, T v) { return __atomic_fetch_or(p, v, __ATOMIC_SEQ_CST); }
template <class T> static inline T atomicMax(T *p, T v) { T o = __atomic_load_n(p, __ATOMIC_SEQ_CST); while (o < v && !__atomic_compare_exchange_n(p, &o, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {} return o; }

// ---- just enough of the runtime API for the launchers to compile ----
typedef int hipError_t; typedef void *hipStream_t; typedef void *hipEvent_t;
enum { hipSuccess = 0, hipErrorOutOfMemory = 2 };
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice };
static inline const char *hipGetErrorString(hipError_t) { return "emulated"; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipMalloc(void **p, size_t n) { *p = calloc(1, n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
static inline hipError_t hipFree(void *p) { free(p); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t = 0) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t = 0) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t = 0) { return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }

// ---- launch: blocks in sequence, the threads of a block as OS threads ----
void emu_run_block(unsigned n_threads, emu_dim3 bid, emu_dim3 bdim, emu_dim3 gdim, const std::function<void()> &body);
extern size_t emu_dyn_lds_bytes;          // what the launch asked for (the harness's definition of the extern array must be that big)
template <class K, class... A> static inline void emu_launch(K kernel, emu_dim3 grid, emu_dim3 block, size_t lds, hipStream_t, A... args) {
    emu_dyn_lds_bytes = lds;
    const unsigned nt = block.x * block.y * block.z;
    for (unsigned bz = 0; bz < grid.z; ++bz) for (unsigned by = 0; by < grid.y; ++by) for (unsigned bx = 0; bx < grid.x; ++bx)
        emu_run_block(nt, emu_dim3(bx, by, bz), block, grid, [&]() { kernel(args...); });
}
#define hipLaunchKernelGGL(kernel, grid, block, lds, stream, ...) emu_launch(kernel, grid, block, lds, stream, ##__VA_ARGS__)
