// smem.hip -- SMEM seeding over the FM-index (the three passes of mem_collect_smem, bwamem.cpp:626-803) and the
// suffix-array lookup (get_sa_entries_prefetch, FMI_search.cpp:1257-1375).
//
// Reference shape: one thread walks 512 reads round-robin, one backwardExt at a time, prefetching the next CP_OCC
// blocks (FMI_search.cpp:693-721).  Here: ONE READ PER LANE, tens of thousands of reads in flight per GPU, so the
// dependent chain of random 64-byte CP_OCC loads (two per backwardExt, FMI_search.cpp:1025-1052) is hidden by
// occupancy instead of software prefetch.  CP_OCC blocks are 64-byte aligned = one HBM line each; every lane fetches
// its two lines with four 16-byte loads each.  Memory-bound on random 64-B HBM transactions; no LDS reuse exists
// between reads (the index is ~10^8 lines, the working set of a wave is 128 unrelated lines per step).
#include "bm2_ctx.h"
#include "pipeline.h"

// (forcing 8 waves/SIMD with __launch_bounds__(256, 8) spills 88 B/lane and measured 25 % slower: left at the natural 84 VGPRs)

struct Bi { int64_t k, l, s; };

static __device__ __forceinline__ int64_t pick4(int a, int64_t c0, int64_t c1, int64_t c2, int64_t c3) {
    return a == 0 ? c0 : a == 1 ? c1 : a == 2 ? c2 : c3;
}

struct Blk { uint64_t w[8]; };   // cp_count[0..3], bwt[0..3]
static __device__ __forceinline__ Blk load_blk(const CpOcc *p) {
    const ulonglong2 *q = (const ulonglong2 *)p;
    ulonglong2 a = q[0], b = q[1], c = q[2], d = q[3];
    Blk r; r.w[0] = a.x; r.w[1] = a.y; r.w[2] = b.x; r.w[3] = b.y; r.w[4] = c.x; r.w[5] = c.y; r.w[6] = d.x; r.w[7] = d.y;
    return r;
}

// FMI_search::backwardExt (FMI_search.cpp:1025-1052) with GET_OCC (FMI_search.h:66-73)
static __device__ __forceinline__ Bi backward_ext(const DevIndex &ix, Bi in, int a) {
    const int64_t sp = in.k, ep = in.k + in.s;
    const Blk b1 = load_blk(&ix.cp_occ[sp >> 6]);
    const Blk b2 = load_blk(&ix.cp_occ[ep >> 6]);
    const int y1 = (int)(sp & 63), y2 = (int)(ep & 63);
    const uint64_t m1 = y1 ? (~0ULL << (64 - y1)) : 0ULL, m2 = y2 ? (~0ULL << (64 - y2)) : 0ULL;   // one_hot_mask_array[y]
    int64_t o1[4], d[4];
#pragma unroll
    for (int b = 0; b < 4; b++) {
        o1[b] = (int64_t)b1.w[b] + __popcll(b1.w[4 + b] & m1);
        const int64_t o2 = (int64_t)b2.w[b] + __popcll(b2.w[4 + b] & m2);
        d[b] = o2 - o1[b];
    }
    const int64_t sent = (in.k <= ix.sentinel_index && ep > ix.sentinel_index) ? 1 : 0;
    const int64_t l3 = in.l + sent, l2 = l3 + d[3], l1 = l2 + d[2], l0 = l1 + d[1];
    Bi out;
    out.k = pick4(a, ix.count[0] + o1[0], ix.count[1] + o1[1], ix.count[2] + o1[2], ix.count[3] + o1[3]);
    out.l = pick4(a, l0, l1, l2, l3);
    out.s = pick4(a, d[0], d[1], d[2], d[3]);
    return out;
}

// forward extension = backward extension of the swapped interval by the complement (FMI_search.cpp:546-554)
static __device__ __forceinline__ Bi forward_ext(const DevIndex &ix, Bi in, int a) {
    Bi sw = { in.l, in.k, in.s };
    Bi r = backward_ext(ix, sw, 3 - a);
    Bi out = { r.l, r.k, r.s };
    return out;
}

static __device__ __forceinline__ Bi init_bi(const DevIndex &ix, int a) {     // FMI_search.cpp:531-533
    Bi b;
    b.k = pick4(a, ix.count[0], ix.count[1], ix.count[2], ix.count[3]);
    b.l = pick4(3 - a, ix.count[0], ix.count[1], ix.count[2], ix.count[3]);
    b.s = pick4(a, ix.count[1], ix.count[2], ix.count[3], ix.count[4]) - b.k;
    return b;
}

// per-lane scratch arrays live in global memory, interleaved by lane: element i of lane t at [i * stride + t]
struct LaneVec {
    StSmem *base; int64_t stride; int cap; int n;
    __device__ __forceinline__ StSmem get(int i) const { return base[(int64_t)i * stride]; }
    __device__ __forceinline__ void set(int i, const StSmem &v) { base[(int64_t)i * stride] = v; }
};

// one (read, start) step of getSMEMsOnePosOneThread (FMI_search.cpp:514-668); returns next_x
static __device__ int smem_one_pos(const DevIndex &ix, const uint8_t *q, int len, int x, int64_t min_intv, int min_seed_len,
                                   LaneVec &out, LaneVec &prev, int64_t &n_ext, int &overflow) {
    int next_x = x + 1;
    int a = q[x];
    if (a >= 4) return next_x;
    StSmem sm; sm.m = x; sm.n = x;
    { Bi b = init_bi(ix, a); sm.k = b.k; sm.l = b.l; sm.s = b.s; }
    int n_prev = 0, j;
    for (j = x + 1; j < len; j++) {                                     // forward phase :537-575
        a = q[j];
        next_x = j + 1;
        if (a >= 4) break;
        Bi cur = { sm.k, sm.l, sm.s };
        Bi nb = forward_ext(ix, cur, a); n_ext++;
        if (nb.s != sm.s) { prev.set(n_prev, sm); n_prev++; }    // (the reference stores unconditionally and bumps the count, :556-559)
        if (nb.s < min_intv) { next_x = j; break; }
        sm.k = nb.k; sm.l = nb.l; sm.s = nb.s; sm.n = j;
    }
    if (sm.s >= min_intv) { prev.set(n_prev, sm); n_prev++; }
    for (int p = 0; p < n_prev / 2; p++) {                              // longest first, :586-592
        StSmem t = prev.get(p), u = prev.get(n_prev - 1 - p);
        prev.set(p, u); prev.set(n_prev - 1 - p, t);
    }
    for (j = x - 1; j >= 0; j--) {                                      // backward phase :596-655
        int n_curr = 0, p;
        int32_t curr_s = -1;
        a = q[j];
        if (a > 3) break;
        bool first_done = false;
        for (p = 0; p < n_prev; p++) {
            StSmem s0 = prev.get(p);
            Bi cur = { s0.k, s0.l, s0.s };
            Bi nb = backward_ext(ix, cur, a); n_ext++;
            if (!first_done) {
                if (nb.s < min_intv && (s0.n - s0.m + 1) >= min_seed_len) {
                    if (out.n < out.cap) out.set(out.n, s0); else overflow = 1;
                    out.n++;
                    first_done = true;
                    continue;
                }
                if (nb.s >= min_intv && nb.s != (int64_t)curr_s) {
                    curr_s = (int32_t)nb.s;
                    StSmem ns = s0; ns.k = nb.k; ns.l = nb.l; ns.s = nb.s; ns.m = j;
                    prev.set(n_curr++, ns);
                    first_done = true;
                }
            } else if (nb.s >= min_intv && nb.s != (int64_t)curr_s) {
                curr_s = (int32_t)nb.s;
                StSmem ns = s0; ns.k = nb.k; ns.l = nb.l; ns.s = nb.s; ns.m = j;
                prev.set(n_curr++, ns);
            }
        }
        n_prev = n_curr;
        if (n_curr == 0) break;
    }
    if (n_prev != 0) {                                                  // :656-665
        StSmem s0 = prev.get(0);
        if ((s0.n - s0.m + 1) >= min_seed_len) {
            if (out.n < out.cap) out.set(out.n, s0); else overflow = 1;
            out.n++;
        }
    }
    return next_x;
}

// bwtSeedStrategyAllPosOneThread for one read (FMI_search.cpp:740-810)
static __device__ void smem_pass3(const DevIndex &ix, const uint8_t *q, int len, int64_t max_intv, int min_seed_len,
                                  LaneVec &out, int64_t &n_ext, int &overflow) {
    int x = 0;
    while (x < len) {
        int next_x = x + 1;
        int a = q[x];
        if (a < 4) {
            StSmem sm; sm.m = x; sm.n = x;
            Bi cur = init_bi(ix, a);
            for (int j = x + 1; j < len; j++) {
                next_x = j + 1;
                a = q[j];
                if (a >= 4) break;
                cur = forward_ext(ix, cur, a); n_ext++;
                sm.n = j;
                if (cur.s < max_intv && (sm.n - sm.m + 1) >= min_seed_len) {
                    if (cur.s > 0) {
                        sm.k = cur.k; sm.l = cur.l; sm.s = cur.s;
                        if (out.n < out.cap) out.set(out.n, sm); else overflow = 1;
                        out.n++;
                    }
                    break;
                }
            }
        }
        x = next_x;
    }
}

// All three passes for the reads of a chunk; one read per lane, grid-stride over reads.
__global__ void __launch_bounds__(256)
k_smem(DevIndex ix, SeedParams sp, int n_reads, const uint8_t *__restrict__ enc, const int64_t *__restrict__ off,
       const int32_t *__restrict__ len, StSmem *stage, StSmem *prevbuf, int stage_cap, int prev_cap,
       bm2_smem_t *out, int64_t out_cap, int32_t *smem_cnt, int64_t *smem_off, int32_t *occ_cnt,
       unsigned long long *counters /* [0]=n_smem [1]=n_ext [2]=overflow */) {
    const int64_t nthreads = (int64_t)gridDim.x * blockDim.x;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t n_ext = 0;
    int overflow = 0;
    for (int64_t r = tid; r < n_reads; r += nthreads) {
        const uint8_t *q = enc + off[r];
        const int L = len[r];
        LaneVec st = { stage + tid, nthreads, stage_cap, 0 };
        LaneVec pv = { prevbuf + tid, nthreads, prev_cap, 0 };
        int x = 0;
        while (x < L) x = smem_one_pos(ix, q, L, x, 1, sp.min_seed_len, st, pv, n_ext, overflow);       // pass 1
        const int n1 = st.n < st.cap ? st.n : st.cap;
        for (int i = 0; i < n1; i++) {                                                               // pass 2, bwamem.cpp:695-753
            StSmem p = st.get(i);
            const int start = p.m, end = p.n + 1;
            if (end - start < sp.split_len || p.s > sp.split_width) continue;
            smem_one_pos(ix, q, L, (end + start) >> 1, p.s + 1, sp.min_seed_len, st, pv, n_ext, overflow);
        }
        if (sp.max_mem_intv > 0) smem_pass3(ix, q, L, sp.max_mem_intv, sp.min_seed_len + 1, st, n_ext, overflow);   // pass 3
        int n = st.n < st.cap ? st.n : st.cap;
        // order (m, n) ascending within the read (sortSMEMs + ks_introsort(mem_intv1), bwamem.cpp:785-799);
        // equal (m,n) are field-identical, so any stable-or-not sort gives the same array
        for (int i = 1; i < n; i++) {
            StSmem v = st.get(i);
            int j = i - 1;
            while (j >= 0) {
                StSmem u = st.get(j);
                if (u.m < v.m || (u.m == v.m && u.n <= v.n)) break;
                st.set(j + 1, u);
                j--;
            }
            st.set(j + 1, v);
        }
        const int64_t o = (int64_t)atomicAdd(&counters[0], (unsigned long long)n);
        smem_cnt[r] = n; smem_off[r] = o;
        for (int i = 0; i < n; i++) {
            StSmem v = st.get(i);
            if (o + i < out_cap) {
                bm2_smem_t w; w.rid = (uint32_t)r; w.m = (uint32_t)v.m; w.n = (uint32_t)v.n; w.pad = 0; w.k = v.k; w.l = v.l; w.s = v.s;
                out[o + i] = w;
                occ_cnt[o + i] = (int32_t)(v.s < sp.max_occ ? v.s : sp.max_occ);       // FMI_search.cpp:1280-1290
            }
        }
    }
    atomicAdd(&counters[1], (unsigned long long)n_ext);
    if (overflow) atomicAdd(&counters[2], 1ULL);
}

// positions of the sampled occurrences of every SMEM: j = k, k+step, ... (FMI_search.cpp:1280-1290)
__global__ void __launch_bounds__(256)
k_sal_expand(const bm2_smem_t *__restrict__ smems, int64_t n_smem, const int64_t *__restrict__ sa_off, int32_t max_occ,
             int64_t *pos) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_smem) return;
    const int64_t k = smems[i].k, s = smems[i].s;
    const int64_t step = s > max_occ ? s / max_occ : 1;
    int64_t o = sa_off[i];
    int c = 0;
    for (int64_t j = k; j < k + s && c < max_occ; j += step, c++) pos[o++] = j;
}

// call_one_step iterated to completion (FMI_search.cpp:1202-1255): one SA lookup per lane, in place pos -> coord
__global__ void __launch_bounds__(256)
k_sal(DevIndex ix, int64_t n, int64_t *pos_coord, unsigned long long *n_lf_out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t n_lf = 0;
    if (t < n) {
        int64_t sp = pos_coord[t], offset = 0, res;
        if ((sp & 7) == 0) {
            res = ((int64_t)ix.sa_ms_byte[sp >> 3] << 32) + ix.sa_ls_word[sp >> 3];
        } else {
            for (;;) {
                const Blk b = load_blk(&ix.cp_occ[sp >> 6]);
                const int y = 63 - (int)(sp & 63);
                int c;
                if ((b.w[4] >> y) & 1) c = 0;
                else if ((b.w[5] >> y) & 1) c = 1;
                else if ((b.w[6] >> y) & 1) c = 2;
                else if ((b.w[7] >> y) & 1) c = 3;
                else { res = 0; break; }                          // sentinel: 0 whatever the offset (:1230-1233)
                n_lf++;
                const int yy = (int)(sp & 63);
                const uint64_t msk = yy ? (~0ULL << (64 - yy)) : 0ULL;
                const int64_t occ = (int64_t)pick4(c, b.w[0], b.w[1], b.w[2], b.w[3]) +
                                    __popcll((uint64_t)pick4(c, b.w[4], b.w[5], b.w[6], b.w[7]) & msk);
                sp = pick4(c, ix.count[0], ix.count[1], ix.count[2], ix.count[3]) + occ;
                offset++;
                if ((sp & 7) == 0) { res = ((int64_t)ix.sa_ms_byte[sp >> 3] << 32) + ix.sa_ls_word[sp >> 3] + offset; break; }
            }
        }
        pos_coord[t] = res;
    }
    if (n_lf_out) atomicAdd(n_lf_out, (unsigned long long)n_lf);
}

// gather SMEMs from the bump-allocated order into read order (for the S2 entry point only)
__global__ void __launch_bounds__(256)
k_smem_gather(int n_reads, const bm2_smem_t *__restrict__ in, const int64_t *__restrict__ in_off,
              const int32_t *__restrict__ cnt, const int64_t *__restrict__ out_off, bm2_smem_t *out) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    const int64_t a = in_off[r], b = out_off[r];
    for (int i = 0; i < cnt[r]; i++) out[b + i] = in[a + i];
}

int bm2_launch_smem(bm2_ctx *c, const SeedParams &sp, int n_reads, const uint8_t *enc, const int64_t *off, const int32_t *len,
                    StSmem *stage, StSmem *prevbuf, int stage_cap, int prev_cap, int grid, bm2_smem_t *out, int64_t out_cap,
                    int32_t *smem_cnt, int64_t *smem_off, int32_t *occ_cnt, unsigned long long *counters) {
    hipLaunchKernelGGL(k_smem, dim3(grid), dim3(256), 0, c->stream, c->ix, sp, n_reads, enc, off, len, stage, prevbuf,
                       stage_cap, prev_cap, out, out_cap, smem_cnt, smem_off, occ_cnt, counters);
    return bm2_check(hipGetLastError(), "k_smem launch");
}
int bm2_launch_sal_expand(bm2_ctx *c, const bm2_smem_t *smems, int64_t n_smem, const int64_t *sa_off, int32_t max_occ, int64_t *pos) {
    if (n_smem <= 0) return BM2_OK;
    hipLaunchKernelGGL(k_sal_expand, dim3((unsigned)((n_smem + 255) / 256)), dim3(256), 0, c->stream, smems, n_smem, sa_off, max_occ, pos);
    return bm2_check(hipGetLastError(), "k_sal_expand launch");
}
int bm2_launch_sal(bm2_ctx *c, int64_t n, int64_t *pos_coord, unsigned long long *n_lf) {
    if (n <= 0) return BM2_OK;
    hipLaunchKernelGGL(k_sal, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, c->ix, n, pos_coord, n_lf);
    return bm2_check(hipGetLastError(), "k_sal launch");
}
int bm2_launch_smem_gather(bm2_ctx *c, int n_reads, const bm2_smem_t *in, const int64_t *in_off, const int32_t *cnt,
                           const int64_t *out_off, bm2_smem_t *out) {
    if (n_reads <= 0) return BM2_OK;
    hipLaunchKernelGGL(k_smem_gather, dim3((n_reads + 255) / 256), dim3(256), 0, c->stream, n_reads, in, in_off, cnt, out_off, out);
    return bm2_check(hipGetLastError(), "k_smem_gather launch");
}
