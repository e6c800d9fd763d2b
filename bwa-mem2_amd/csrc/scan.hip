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
