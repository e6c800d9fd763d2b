// bsw.hip -- kernels around bsw_extend_wave (see bsw_dev.h for the DP itself).
#include "bsw_dev.h"
#include "bm2_ctx.h"

// LDS layout per block: [SwParams][wave 0: RH[R] RE[R]][wave 1: ...]
static __device__ __forceinline__ void lds_carve(int *lds, int R, SwParams *&sP, int *&RH, int *&RE) {
    sP = (SwParams *)lds;
    int *rings = lds + (sizeof(SwParams) + 3) / 4;
    const int wv = threadIdx.x >> 6;
    RH = rings + (size_t)wv * 2 * R;
    RE = RH + R;
}

// ---- S1: one SeqPair per wavefront, one band (the reference's getScores8/16/scalar wrappers run ONE band per
// call; the caller owns the two-try loop, bwamem.cpp:2472-2526).
__global__ void __launch_bounds__(256)
k_bsw_pairs(bm2_seqpair_t *__restrict__ pairs, const uint8_t *__restrict__ ref, const uint8_t *__restrict__ qer,
            int n, int w, SwParams P, int R, unsigned long long *cells_out) {
    extern __shared__ __attribute__((aligned(16))) int lds[];
    SwParams *sP; int *RH, *RE;
    lds_carve(lds, R, sP, RH, RE);
    if (threadIdx.x < sizeof(SwParams) / 4) ((int *)sP)[threadIdx.x] = ((const int *)&P)[threadIdx.x];
    __syncthreads();
    const int wid = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (wid >= n) return;
    const bm2_seqpair_t sp = pairs[wid];
    const int cls = pair_class(sp.len1, sp.len2, sp.h0, sP->max_sc);
    const int wc = band_clamp(w, sp.len2, *sP, cls);
    SwOut o;
    int cells = bsw_extend(qer + sp.idq, 1, sp.len2, RefPtr::bytes(ref + sp.idr), 1, sp.len1, wc, sp.h0, *sP, RH, RE, R - 1, o);
    if ((threadIdx.x & 63) == 0) {
        bm2_seqpair_t *d = &pairs[wid];
        d->score = o.score; d->tle = o.tle; d->gtle = o.gtle; d->qle = o.qle; d->gscore = o.gscore; d->max_off = o.max_off;
        if (cells_out) atomicAdd(cells_out, (unsigned long long)cells);
    }
}

static int ring_size(int w) {                // power of two >= 2*w + 4
    int R = 64;
    while (R < 2 * w + 4) R <<= 1;
    return R;
}

// the same for the pairs of a list (the part of a sorted S1 batch that the lane kernel does not take: bm2_launch_bsw_sorted, extend.hip)
__global__ void __launch_bounds__(256)
k_bsw_list(bm2_seqpair_t *__restrict__ pairs, const uint8_t *__restrict__ ref, const uint8_t *__restrict__ qer, const int32_t *__restrict__ list,
           const int64_t *__restrict__ start, int bin_lo, int bin_hi, int w, SwParams P, int R, unsigned long long *cells_out) {
    extern __shared__ __attribute__((aligned(16))) int lds[];
    SwParams *sP; int *RH, *RE;
    lds_carve(lds, R, sP, RH, RE);
    if (threadIdx.x < sizeof(SwParams) / 4) ((int *)sP)[threadIdx.x] = ((const int *)&P)[threadIdx.x];
    __syncthreads();
    const int64_t first = start[bin_lo];
    const int n = (int)(start[bin_hi] - first);
    const int wpb = blockDim.x >> 6;
    long long cells = 0;
    for (int wid = blockIdx.x * wpb + (threadIdx.x >> 6); wid < n; wid += gridDim.x * wpb) {
        const int id = uni(list[first + n - 1 - wid]);              // (the list ascends in length: longest first)
        const bm2_seqpair_t sp = pairs[id];
        const int cls = pair_class(sp.len1, sp.len2, sp.h0, sP->max_sc);
        const int wc = band_clamp(w, sp.len2, *sP, cls);
        SwOut o;
        cells += bsw_extend(qer + sp.idq, 1, sp.len2, RefPtr::bytes(ref + sp.idr), 1, sp.len1, wc, sp.h0, *sP, RH, RE, R - 1, o);
        if ((threadIdx.x & 63) == 0) {
            bm2_seqpair_t *d = &pairs[id];
            d->score = o.score; d->tle = o.tle; d->gtle = o.gtle; d->qle = o.qle; d->gscore = o.gscore; d->max_off = o.max_off;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // (the next pair of this wavefront reuses the LDS rings)
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    if (cells_out && (threadIdx.x & 63) == 0) atomicAdd(cells_out, (unsigned long long)cells);
}


int bm2_launch_bsw_list(bm2_ctx *c, bm2_seqpair_t *d_pairs, const uint8_t *d_ref, const uint8_t *d_qer, const int32_t *list, const int64_t *start,
                        int bin_lo, int bin_hi, unsigned grid, int w, const SwParams &P, unsigned long long *d_cells, hipStream_t s) {
    const int R = ring_size(w);
    const int waves = 4;
    const size_t lds = ((sizeof(SwParams) + 3) / 4) * 4 + (size_t)waves * 2 * R * 4;
    if (lds > 160 * 1024) return BM2_EUNSUP;
    hipLaunchKernelGGL(k_bsw_list, dim3(grid), dim3(waves * 64), lds, s, d_pairs, d_ref, d_qer, list, start, bin_lo, bin_hi, w, P, R, d_cells);
    return bm2_check(hipGetLastError(), "k_bsw_list launch");
}

int bm2_launch_bsw_pairs(bm2_ctx *c, bm2_seqpair_t *d_pairs, const uint8_t *d_ref, const uint8_t *d_qer, int n, int w,
                         const SwParams &P, unsigned long long *d_cells) {
    if (n <= 0) return BM2_OK;
    {   // the pairs that fit the lane kernel go one per LANE, sorted by length on the device (extend.hip); the rest one per wavefront
        bool done = false;
        const int rc = bm2_launch_bsw_sorted(c, d_pairs, d_ref, d_qer, n, w, P, d_cells, &done);
        if (rc || done) return rc;
    }
    const int R = ring_size(w);
    const int waves = 4;
    size_t lds = ((sizeof(SwParams) + 3) / 4) * 4 + (size_t)waves * 2 * R * 4;
    if (lds > 160 * 1024) return BM2_EUNSUP;
    hipLaunchKernelGGL(k_bsw_pairs, dim3((n + waves - 1) / waves), dim3(waves * 64), lds, c->stream,
                       d_pairs, d_ref, d_qer, n, w, P, R, d_cells);
    return bm2_check(hipGetLastError(), "k_bsw_pairs launch");
}
