// scan.hip -- exclusive prefix sum int32 -> int64 (three small kernels; plumbing between pipeline stages).
#include "pipeline.h"

#define SCAN_ITEMS 2048      // per block: 256 threads x 8

__global__ void __launch_bounds__(256) k_scan_bsum(const int32_t *__restrict__ in, int64_t n, int64_t *bsum) {
    __shared__ int64_t sh[256];
    const int64_t base = (int64_t)blockIdx.x * SCAN_ITEMS;
    int64_t s = 0;
    for (int i = 0; i < 8; i++) { int64_t j = base + threadIdx.x * 8 + i; if (j < n) s += in[j]; }
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int d = 128; d > 0; d >>= 1) { if (threadIdx.x < d) sh[threadIdx.x] += sh[threadIdx.x + d]; __syncthreads(); }
    if (threadIdx.x == 0) bsum[blockIdx.x] = sh[0];
}

__global__ void __launch_bounds__(1024) k_scan_top(int64_t *bsum, int64_t nb, int64_t *total) {
    __shared__ int64_t sh[1024];
    __shared__ int64_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int64_t base = 0; base < nb; base += 1024) {
        int64_t j = base + threadIdx.x;
        int64_t v = j < nb ? bsum[j] : 0;
        sh[threadIdx.x] = v;
        __syncthreads();
        for (int d = 1; d < 1024; d <<= 1) {
            int64_t t = threadIdx.x >= d ? sh[threadIdx.x - d] : 0;
            __syncthreads();
            sh[threadIdx.x] += t;
            __syncthreads();
        }
        if (j < nb) bsum[j] = carry + sh[threadIdx.x] - v;     // exclusive
        __syncthreads();
        if (threadIdx.x == 0) carry += sh[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry;
}

__global__ void __launch_bounds__(256) k_scan_final(const int32_t *__restrict__ in, int64_t n, const int64_t *__restrict__ boff,
                                                    const int64_t *__restrict__ total, int64_t *out) {
    __shared__ int64_t sh[256];
    const int64_t base = (int64_t)blockIdx.x * SCAN_ITEMS + threadIdx.x * 8;
    int32_t v[8]; int64_t s = 0;
    for (int i = 0; i < 8; i++) { int64_t j = base + i; v[i] = j < n ? in[j] : 0; s += v[i]; }
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
        int64_t t = threadIdx.x >= d ? sh[threadIdx.x - d] : 0;
        __syncthreads();
        sh[threadIdx.x] += t;
        __syncthreads();
    }
    int64_t run = boff[blockIdx.x] + sh[threadIdx.x] - s;
    for (int i = 0; i < 8; i++) { int64_t j = base + i; if (j < n) out[j] = run; run += v[i]; }
    if (blockIdx.x == 0 && threadIdx.x == 0) out[n] = *total;
}

int bm2_scan_i32(bm2_ctx *c, const int32_t *in, int64_t n, int64_t *out, DevBuf &tmp) {
    const int64_t nb = (n + SCAN_ITEMS - 1) / SCAN_ITEMS;
    int rc = bm2_reserve(tmp, (size_t)(nb + 2) * 8);
    if (rc) return rc;
    int64_t *bsum = (int64_t *)tmp.p, *total = bsum + nb + 1;
    if (n <= 0) return bm2_check(hipMemsetAsync(out, 0, 8, c->stream), "scan memset");
    hipLaunchKernelGGL(k_scan_bsum, dim3((unsigned)nb), dim3(256), 0, c->stream, in, n, bsum);
    hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(1024), 0, c->stream, bsum, nb, total);
    hipLaunchKernelGGL(k_scan_final, dim3((unsigned)nb), dim3(256), 0, c->stream, in, n, bsum, total, out);
    return bm2_check(hipGetLastError(), "scan launch");
}

// ---- permutation of reads by a work key (counting sort into 32 log2 bins): the one-read-per-lane kernels (chaining,
// post-filter) run as long as the heaviest lane of a wavefront, so lanes are grouped by how much sequential work their
// read carries.
#define PERM_BINS 32
static __device__ __forceinline__ int perm_bin(int v, int heavy_first) {
    int lg = v <= 0 ? 0 : 1 + (31 - __clz(v));          // 0, 1, 2, 2, 3, 3, 3, 3, 4 ...
    if (lg > PERM_BINS - 1) lg = PERM_BINS - 1;
    return heavy_first ? PERM_BINS - 1 - lg : lg;
}
__global__ void __launch_bounds__(256) k_perm_hist(int n, const int32_t *__restrict__ key, uint32_t *hist, int hf) {
    __shared__ uint32_t sh[PERM_BINS];
    if (threadIdx.x < PERM_BINS) sh[threadIdx.x] = 0;
    __syncthreads();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) atomicAdd(&sh[perm_bin(key[i], hf)], 1u);
    __syncthreads();
    if (threadIdx.x < PERM_BINS && sh[threadIdx.x]) atomicAdd(&hist[threadIdx.x], sh[threadIdx.x]);
}
__global__ void k_perm_prefix(uint32_t *hist) {
    if (threadIdx.x == 0) { uint32_t acc = 0; for (int b = 0; b < PERM_BINS; b++) { uint32_t c = hist[b]; hist[b] = acc; acc += c; } }
}
__global__ void __launch_bounds__(256) k_perm_scatter(int n, const int32_t *__restrict__ key, uint32_t *cursor, int32_t *perm, int hf) {
    __shared__ uint32_t cnt[PERM_BINS], basep[PERM_BINS];
    if (threadIdx.x < PERM_BINS) cnt[threadIdx.x] = 0;
    __syncthreads();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    int b = 0; uint32_t pos = 0;
    if (i < n) { b = perm_bin(key[i], hf); pos = atomicAdd(&cnt[b], 1u); }
    __syncthreads();
    if (threadIdx.x < PERM_BINS && cnt[threadIdx.x]) basep[threadIdx.x] = atomicAdd(&cursor[threadIdx.x], cnt[threadIdx.x]);
    __syncthreads();
    if (i < n) perm[basep[b] + pos] = i;
}
__global__ void __launch_bounds__(256) k_perm_identity(int n, int32_t *perm) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) perm[i] = i;
}

int bm2_perm_by_work(bm2_ctx *c, int n, const int32_t *key, int32_t *perm, uint32_t *hist32 /* device, PERM_BINS words */, int mode) {
    if (n <= 0) return BM2_OK;
    hipStream_t s = c->stream;
    if (mode == 0) { hipLaunchKernelGGL(k_perm_identity, dim3((n + 255) / 256), dim3(256), 0, s, n, perm); return bm2_check(hipGetLastError(), "perm"); }
    int rc = bm2_check(hipMemsetAsync(hist32, 0, PERM_BINS * 4, s), "memset perm hist");
    if (rc) return rc;
    const int hf = mode == 1;
    hipLaunchKernelGGL(k_perm_hist, dim3((n + 255) / 256), dim3(256), 0, s, n, key, hist32, hf);
    hipLaunchKernelGGL(k_perm_prefix, dim3(1), dim3(64), 0, s, hist32);
    hipLaunchKernelGGL(k_perm_scatter, dim3((n + 255) / 256), dim3(256), 0, s, n, key, hist32, perm, hf);
    return bm2_check(hipGetLastError(), "perm launch");
}

// ---- stable two-way partition of reads: light reads (key <= thr) first, in their original order (neighbouring reads own
// neighbouring memory), heavy reads after them, so that a wavefront of light reads is not held up by one heavy lane
__global__ void __launch_bounds__(256) k_part_flag(int n, const int32_t *__restrict__ key, int thr, int32_t *flag) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flag[i] = key[i] > thr ? 1 : 0;
}
__global__ void __launch_bounds__(256) k_part_scatter(int n, const int32_t *__restrict__ flag, const int64_t *__restrict__ hpos, int32_t *perm,
                                                      int heavy_first) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t n_heavy = hpos[n], h = hpos[i];
    if (heavy_first) {               // the long-running reads start first and the light ones fill in behind them
        if (flag[i]) perm[h] = i;
        else perm[n_heavy + (i - h)] = i;
    } else {
        if (flag[i]) perm[(n - n_heavy) + h] = i;
        else perm[i - h] = i;
    }
}
int bm2_partition_by_work(bm2_ctx *c, int n, const int32_t *key, int thr, int32_t *perm, DevBuf &tmp, DevBuf &scan_tmp, int heavy_first,
                          const int64_t **n_heavy_dev) {
    if (n <= 0) return BM2_OK;
    int rc = bm2_reserve(tmp, (size_t)(n + 1) * 4 + (size_t)(n + 2) * 8 + 64);
    if (rc) return rc;
    int32_t *flag = (int32_t *)tmp.p;
    int64_t *hpos = (int64_t *)((char *)tmp.p + (((size_t)(n + 1) * 4 + 15) & ~(size_t)15));
    if (n_heavy_dev) *n_heavy_dev = hpos + n;
    hipLaunchKernelGGL(k_part_flag, dim3((n + 255) / 256), dim3(256), 0, c->stream, n, key, thr, flag);
    if ((rc = bm2_scan_i32(c, flag, n, hpos, scan_tmp))) return rc;
    hipLaunchKernelGGL(k_part_scatter, dim3((n + 255) / 256), dim3(256), 0, c->stream, n, flag, hpos, perm, heavy_first);
    return bm2_check(hipGetLastError(), "partition launch");
}

// ---- stable partition of reads into work classes: heavy reads (key > thr) first, then the light reads by falling class of key (> 64, > 32, > 16,
// > 8, > 4, the rest), every class in the reads' original order.  A lane-per-read kernel is as slow as the busiest lane of each wavefront:
// with the light reads in plain order nearly every wavefront holds a read with 50+ seeds among reads with ten (k_chain: 7.2 ms); with classes a
// wavefront's reads cost alike, and because a class keeps the original order its lanes still own nearby memory (a full sort by key loses that and
// measured slower in round 1).  One flag per (class, read), ONE scan over the class-major flags, one scatter.
#define BM2_WORK_CLASSES 7
static __device__ __forceinline__ int work_class(int key, int thr) {
    return key > thr ? 0 : key > 64 ? 1 : key > 32 ? 2 : key > 16 ? 3 : key > 8 ? 4 : key > 4 ? 5 : 6;
}
__global__ void __launch_bounds__(256) k_class_flag(int n, const int32_t *__restrict__ key, int thr, int32_t *flag) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int b = work_class(key[i], thr);
    for (int k = 0; k < BM2_WORK_CLASSES; k++) flag[(int64_t)k * n + i] = k == b;
}
__global__ void __launch_bounds__(256) k_class_scatter(int n, const int32_t *__restrict__ key, int thr, const int64_t *__restrict__ pos, int32_t *perm) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    perm[pos[(int64_t)work_class(key[i], thr) * n + i]] = i;
}
int bm2_partition_by_class(bm2_ctx *c, int n, const int32_t *key, int thr, int32_t *perm, DevBuf &tmp, DevBuf &scan_tmp, const int64_t **n_heavy_dev) {
    if (n <= 0) return BM2_OK;
    const size_t m = (size_t)BM2_WORK_CLASSES * (size_t)n;
    int rc = bm2_reserve(tmp, (m + 1) * 4 + (m + 2) * 8 + 64);
    if (rc) return rc;
    int32_t *flag = (int32_t *)tmp.p;
    int64_t *pos = (int64_t *)((char *)tmp.p + (((m + 1) * 4 + 15) & ~(size_t)15));
    if (n_heavy_dev) *n_heavy_dev = pos + n;                   // where class 1 starts = how many heavy reads there are
    hipLaunchKernelGGL(k_class_flag, dim3((n + 255) / 256), dim3(256), 0, c->stream, n, key, thr, flag);
    if ((rc = bm2_scan_i32(c, flag, (int64_t)m, pos, scan_tmp))) return rc;
    hipLaunchKernelGGL(k_class_scatter, dim3((n + 255) / 256), dim3(256), 0, c->stream, n, key, thr, pos, perm);
    return bm2_check(hipGetLastError(), "class partition launch");
}
