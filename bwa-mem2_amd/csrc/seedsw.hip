// seedsw.hip -- mem_flt_chained_seeds / mem_seed_sw (bwamem.cpp:401-427, 472-504): for reads long enough that
// 5.5*ln(l) <= 0.05*l (or with -W), every seed shorter than 200 bp is re-scored by a local Smith-Waterman of the seed
// +-50 bp against the reference and dropped when the score is below min_HSP_score.  The reference calls ksw_align2
// (Farrar's striped SW, ksw.cpp:234-381) once per seed; only its score is used.  Here: one seed per lane, the DP row
// {H, E} packed 16+16 bits in LDS laid out [column][lane] (query windows are < 200 columns), plain row-by-row
// recurrence -- the maximum it finds is the one the striped evaluation finds.
#include "pipeline.h"
#include "chain_dev.h"

#define SSW_QMAX 200

static __device__ __forceinline__ int64_t depos2(const DevIndex &ix, int64_t pos, int &is_rev) {
    is_rev = pos >= ix.l_pac;
    return is_rev ? (ix.l_pac << 1) - 1 - pos : pos;
}
static __device__ int pos2rid2(const DevIndex &ix, int64_t pos_f) {
    int left = 0, mid = 0, right = ix.n_seqs;
    if (pos_f >= ix.l_pac) return -1;
    while (left < right) {
        mid = (left + right) >> 1;
        if (pos_f >= ix.ann_offset[mid]) {
            if (mid == ix.n_seqs - 1) break;
            if (pos_f < ix.ann_offset[mid + 1]) break;
            left = mid + 1;
        } else right = mid;
    }
    return mid;
}

// P8: no score of the window can exceed 255 (199 columns x a): the row as 8 + 8 bits per column and the query 4 bits per base -- 32 KB of
// LDS per wavefront instead of 64 KB (5 wavefronts per CU instead of 2).  Scores are computed from (match, mismatch, ambiguous), the form
// check_opt guarantees, instead of a global-memory lookup per cell.
template <bool P8>
__global__ void __launch_bounds__(64)
k_seed_sw(DevIndex ix, ChainParams o, const int8_t *__restrict__ mat25, int64_t n_slots, const uint8_t *__restrict__ enc,
          const int64_t *__restrict__ off, const int32_t *__restrict__ len, const int32_t *__restrict__ min_hsp /* per read, <0 = filter inactive */,
          const int32_t *__restrict__ seed_owner, DevSeed *seeds, uint8_t *seed_keep) {
    __shared__ uint32_t HE[(P8 ? SSW_QMAX / 2 : SSW_QMAX) * 64];     // P8: two columns per dword {h, e, h, e}
    __shared__ uint32_t QL[(SSW_QMAX + 7) / 8 * 64];                 // 8 bases of 4 bits per dword
    const int lane = threadIdx.x;
    const int64_t g = (int64_t)blockIdx.x * 64 + lane;
    int r = -1;
    if (g < n_slots) r = seed_owner[g];
    bool run = r >= 0 && min_hsp[r] >= 0;
    int qlen = 0, tlen = 0;
    RefPtr tp = ix.ref(0);
    DevSeed s;
    if (run) {
        s = seeds[g];
        const int l_query = len[r];
        const int64_t l_pac = ix.l_pac;
        run = false;
        if (s.len < 200) {                               // MEM_SHORT_LEN
            int qb = s.qbeg - 50, qe = s.qbeg + s.len + 50;                     // MEM_SHORT_EXT
            int64_t rb = s.rbeg - 50, re = s.rbeg + s.len + 50;
            const int64_t mid = (s.rbeg + s.rbeg + s.len) >> 1;
            qb = qb > 0 ? qb : 0; qe = qe < l_query ? qe : l_query;
            rb = rb > 0 ? rb : 0; re = re < l_pac << 1 ? re : l_pac << 1;
            if (rb < l_pac && l_pac < re) { if (mid < l_pac) re = l_pac; else rb = l_pac; }
            if (!(qe - qb >= 200 || re - rb >= 200)) {
                int is_rev;
                const int rid = pos2rid2(ix, depos2(ix, mid, is_rev));           // bns_fetch_seq: clip to the contig of mid
                int64_t far_beg = ix.ann_offset[rid], far_end = far_beg + ix.ann_len[rid];
                if (is_rev) { const int64_t tmp = far_beg; far_beg = (l_pac << 1) - far_end; far_end = (l_pac << 1) - tmp; }
                rb = rb > far_beg ? rb : far_beg;
                re = re < far_end ? re : far_end;
                qlen = qe - qb; tlen = (int)(re - rb);
                tp = ix.ref(rb);
                const uint8_t *qp = enc + off[r] + qb;
                for (int j0 = 0; j0 < qlen; j0 += 8) {
                    uint32_t wq = 0;
                    for (int u = 0; u < 8 && j0 + u < qlen; u++) wq |= (uint32_t)(qp[j0 + u] & 15) << (4 * u);
                    QL[(j0 >> 3) * 64 + lane] = wq;
                }
                for (int j = 0; j < (P8 ? (qlen + 1) / 2 : qlen); j++) HE[j * 64 + lane] = 0;
                run = true;
            }
        }
    }
    // local SW, ksw_i16 recurrence (ksw.cpp:275-291): gaps open from H, everything clamped at 0
    const int oe_del = o.o_del + o.e_del, oe_ins = o.o_ins + o.e_ins;
    const int s_match = mat25[0], s_mis = mat25[1], s_amb = mat25[4];            // (uniform loads: once per wavefront)
    int maxq = run ? qlen : 0, maxt = run ? tlen : 0;
    for (int d = 32; d > 0; d >>= 1) { maxq = max(maxq, __shfl_xor(maxq, d)); maxt = max(maxt, __shfl_xor(maxt, d)); }
    int gmax = 0;
    int t_next = run && tlen > 0 ? (int)tp[0] : 4;
    for (int i = 0; i < maxt; i++) {
        const bool rowon = run && i < tlen;
        const int tb = rowon ? t_next : 4;
        if (run && i + 1 < tlen) t_next = (int)tp[i + 1];                           // (requested a row ahead)
        int hdiag = 0, f = 0;
        if (P8) {
            for (int jp = 0; jp < maxq; jp += 2) {
                if (rowon && jp < qlen) {
                    uint32_t word = HE[(jp >> 1) * 64 + lane];
                    const uint32_t qw = QL[(jp >> 3) * 64 + lane] >> (4 * (jp & 7));
#pragma unroll
                    for (int u = 0; u < 2; u++) {
                        if (jp + u < qlen) {
                            const int qb = (int)((qw >> (4 * u)) & 15u);
                            int e = (int)((word >> (16 * u + 8)) & 0xffu);
                            int h = hdiag + ((qb == tb && tb < 4) ? s_match : ((qb > 3 || tb > 3) ? s_amb : s_mis));
                            hdiag = (int)((word >> (16 * u)) & 0xffu);
                            h = h > e ? h : e;
                            h = h > f ? h : f;
                            gmax = gmax > h ? gmax : h;
                            e = max(isub0(e, o.e_del), h - oe_del);
                            f = max(isub0(f, o.e_ins), h - oe_ins);
                            word = (word & ~(0xffffu << (16 * u))) | ((uint32_t)h | (uint32_t)e << 8) << (16 * u);
                        }
                    }
                    HE[(jp >> 1) * 64 + lane] = word;
                }
            }
        } else {
            for (int j = 0; j < maxq; j++) {
                if (rowon && j < qlen) {
                    const uint32_t p = HE[j * 64 + lane];
                    const int qb = (int)((QL[(j >> 3) * 64 + lane] >> (4 * (j & 7))) & 15u);
                    int e = (int)(p >> 16);
                    int h = hdiag + ((qb == tb && tb < 4) ? s_match : ((qb > 3 || tb > 3) ? s_amb : s_mis));
                    hdiag = (int)(p & 0xffffu);
                    h = h > e ? h : e;
                    h = h > f ? h : f;
                    gmax = gmax > h ? gmax : h;
                    e = max(isub0(e, o.e_del), h - oe_del);
                    f = max(isub0(f, o.e_ins), h - oe_ins);
                    HE[j * 64 + lane] = (uint32_t)h | ((uint32_t)e << 16);
                }
            }
        }
    }
    if (g < n_slots && r >= 0 && min_hsp[r] >= 0) {
        const int sc = run ? gmax : -1;
        const bool keep = sc < 0 || sc >= min_hsp[r];            // bwamem.cpp:494-499
        seed_keep[g] = keep ? 1 : 0;
        seeds[g].score = sc < 0 ? s.len * o.a : sc;
    }
}

// columns 10 G .. 10 G + 9 of one DP row of k_seed_sw_reg, then the next group if the wavefront's widest window reaches it (nested, so that a narrow
// wavefront leaves after one scalar test; the row `he` and the query `ql` are registers: every index is a constant)
#define SSW_GRP 5
template <int G>
static __device__ __forceinline__ void ssw_cols(uint32_t (&he)[SSW_QMAX / 2], uint32_t (&ql)[(SSW_QMAX + 7) / 8], int maxq, int qlen, bool rowon, int tb,
                                                int s_eq, int s_ne, int s_amb, int e_del, int e_ins, int oe_del, int oe_ins, int &hdiag, int &f, int &gmax) {
    if constexpr (G * SSW_GRP < SSW_QMAX / 2) {
        if (2 * G * SSW_GRP < maxq) {
#pragma unroll
            for (int jp = G * SSW_GRP; jp < (G + 1) * SSW_GRP; jp++) {
                const uint32_t word = he[jp];
                const uint32_t qw = ql[jp >> 2] >> (8 * (jp & 3));
                uint32_t nw = 0;
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    const int qb = (int)((qw >> (4 * u)) & 15u);
                    int e = (int)((word >> (16 * u + 8)) & 0xffu);
                    int h = hdiag + (qb == tb ? s_eq : (qb > 3 ? s_amb : s_ne));
                    hdiag = (int)((word >> (16 * u)) & 0xffu);
                    h = h > e ? h : e;
                    h = h > f ? h : f;
                    const bool on = rowon && 2 * jp + u < qlen;
                    gmax = on && h > gmax ? h : gmax;
                    e = max(isub0(e, e_del), h - oe_del);
                    f = max(isub0(f, e_ins), h - oe_ins);
                    nw |= ((uint32_t)h | (uint32_t)e << 8) << (16 * u);
                }
                he[jp] = nw;                             // (a column beyond the lane's window or a row beyond its target holds what nothing reads: `on` guards gmax)
            }
            ssw_cols<G + 1>(he, ql, maxq, qlen, rowon, tb, s_eq, s_ne, s_amb, e_del, e_ins, oe_del, oe_ins, hdiag, f, gmax);
        }
    }
}

// The same filter with the DP row in REGISTERS (the P8 form only: 8 + 8 bits per column).  k_seed_sw keeps {H, E} and the query in LDS, 32 KB per
// wavefront: five wavefronts per CU, every cell a dependent ds_read -> ALU -> ds_write chain with nothing to hide it behind (660 G cells/s on a chunk of
// 20 000 long reads, 217 ms of its 1.2 s step).  Here the row is 100 dwords and the query 25 dwords of the lane's own registers, the column loop is
// unrolled over them (a register has no run-time index) and leaves at the wavefront's widest window; three wavefronts per SIMD, no LDS at all.
// A column past the lane's own window is computed and thrown away (two selects per cell) instead of branched around.
#define SSW_NP (SSW_QMAX / 2)
// (registers allocated for TWO wavefronts per SIMD: for three, a tenth of the row loop's instructions were spills and the kernel slower than the LDS form --
//  chain stage of 20 000 long reads 696 ms against 594 and 723, profiles/r05p_config5_variants.txt; that instantiation is gone)
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2)))
k_seed_sw_reg(DevIndex ix, ChainParams o, const int8_t *__restrict__ mat25, int64_t n_slots, const uint8_t *__restrict__ enc,
              const int64_t *__restrict__ off, const int32_t *__restrict__ len, const int32_t *__restrict__ min_hsp /* per read, <0 = filter inactive */,
              const int32_t *__restrict__ seed_owner, DevSeed *seeds, uint8_t *seed_keep) {
    const int lane = threadIdx.x;
    const int64_t g = (int64_t)blockIdx.x * 64 + lane;
    int r = -1;
    if (g < n_slots) r = seed_owner[g];
    bool run = r >= 0 && min_hsp[r] >= 0;
    if (!__any(run)) return;                             // (most workgroups: the slots behind a read's kept seeds are empty)
    int qlen = 0, tlen = 0;
    RefPtr tp = ix.ref(0);
    DevSeed s;
    uint32_t he[SSW_NP], ql[(SSW_QMAX + 7) / 8];
#pragma unroll
    for (int j = 0; j < SSW_NP; j++) he[j] = 0;
#pragma unroll
    for (int j = 0; j < (SSW_QMAX + 7) / 8; j++) ql[j] = 0;
    const uint8_t *qp = enc;
    if (run) {
        s = seeds[g];
        const int l_query = len[r];
        const int64_t l_pac = ix.l_pac;
        run = false;
        if (s.len < 200) {                               // MEM_SHORT_LEN
            int qb = s.qbeg - 50, qe = s.qbeg + s.len + 50;                     // MEM_SHORT_EXT
            int64_t rb = s.rbeg - 50, re = s.rbeg + s.len + 50;
            const int64_t mid = (s.rbeg + s.rbeg + s.len) >> 1;
            qb = qb > 0 ? qb : 0; qe = qe < l_query ? qe : l_query;
            rb = rb > 0 ? rb : 0; re = re < l_pac << 1 ? re : l_pac << 1;
            if (rb < l_pac && l_pac < re) { if (mid < l_pac) re = l_pac; else rb = l_pac; }
            if (!(qe - qb >= 200 || re - rb >= 200)) {
                int is_rev;
                const int rid = pos2rid2(ix, depos2(ix, mid, is_rev));           // bns_fetch_seq: clip to the contig of mid
                int64_t far_beg = ix.ann_offset[rid], far_end = far_beg + ix.ann_len[rid];
                if (is_rev) { const int64_t tmp = far_beg; far_beg = (l_pac << 1) - far_end; far_end = (l_pac << 1) - tmp; }
                rb = rb > far_beg ? rb : far_beg;
                re = re < far_end ? re : far_end;
                qlen = qe - qb; tlen = (int)(re - rb);
                tp = ix.ref(rb);
                qp = enc + off[r] + qb;
                run = true;
            }
        }
    }
    int maxq = run ? qlen : 0, maxt = run ? tlen : 0;
    for (int d = 32; d > 0; d >>= 1) { maxq = max(maxq, __shfl_xor(maxq, d)); maxt = max(maxt, __shfl_xor(maxt, d)); }
    maxq = __builtin_amdgcn_readfirstlane(maxq); maxt = __builtin_amdgcn_readfirstlane(maxt);
#pragma unroll
    for (int j0 = 0; j0 < SSW_QMAX; j0 += 8) {           // the query window, 8 bases of 4 bits per register
        if (j0 < maxq) {
            uint32_t wq = 0;
#pragma unroll
            for (int u = 0; u < 8; u++) if (j0 + u < qlen) wq |= (uint32_t)(qp[j0 + u] & 15) << (4 * u);
            ql[j0 >> 3] = wq;
        }
    }
    // local SW, ksw_i16 recurrence (ksw.cpp:275-291): gaps open from H, everything clamped at 0
    const int oe_del = o.o_del + o.e_del, oe_ins = o.o_ins + o.e_ins;
    const int s_match = mat25[0], s_mis = mat25[1], s_amb = mat25[4];            // (uniform loads: once per wavefront)
    int gmax = 0;
    int t_next = run && tlen > 0 ? (int)tp[0] : 4;
    for (int i = 0; i < maxt; i++) {
        const bool rowon = run && i < tlen;
        const int tb = rowon ? t_next : 4;
        if (run && i + 1 < tlen) t_next = (int)tp[i + 1];                           // (requested a row ahead)
        const int s_eq = tb < 4 ? s_match : s_amb, s_ne = tb < 4 ? s_mis : s_amb;     // the row's scores for an equal / a different unambiguous query base
        int hdiag = 0, f = 0;
        // (the query registers pass through an empty asm every row: left alone the compiler hoists the 200 base extractions -- invariant over the rows --
        //  out of the row loop, 200 more live registers, and spills the row itself to scratch)
#pragma unroll
        for (int j = 0; j < (SSW_QMAX + 7) / 8; j++) asm volatile("" : "+v"(ql[j]));
        ssw_cols<0>(he, ql, maxq, qlen, rowon, tb, s_eq, s_ne, s_amb, o.e_del, o.e_ins, oe_del, oe_ins, hdiag, f, gmax);
    }
    if (g < n_slots && r >= 0 && min_hsp[r] >= 0) {
        const int sc = run ? gmax : -1;
        const bool keep = sc < 0 || sc >= min_hsp[r];            // bwamem.cpp:494-499
        seed_keep[g] = keep ? 1 : 0;
        seeds[g].score = sc < 0 ? s.len * o.a : sc;
    }
}

// drop the filtered seeds inside every chain of the reads the filter applies to (keeps order), bwamem.cpp:491-501
__global__ void __launch_bounds__(128)
k_seed_flt_apply(int n_reads, const int64_t *__restrict__ read_base, const int32_t *__restrict__ n_chain, const int32_t *__restrict__ min_hsp,
                 DevChain *chn, DevSeed *seeds, const uint8_t *__restrict__ seed_keep) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads || min_hsp[r] < 0) return;
    const int64_t base = read_base[r];
    for (int i = 0; i < n_chain[r]; i++) {
        DevChain c = chn[base + i];
        int k = 0;
        for (int j = 0; j < c.n; j++)
            if (seed_keep[c.seed_off + j]) { if (k != j) seeds[c.seed_off + k] = seeds[c.seed_off + j]; k++; }
        chn[base + i].n = k;
    }
}

int bm2_launch_seed_filter(bm2_ctx *c, const ChainParams &o, const int8_t *d_mat25, int n_reads, int64_t n_slots, const uint8_t *enc,
                           const int64_t *off, const int32_t *len, const int32_t *min_hsp, const int64_t *read_base, const int32_t *n_chain,
                           const int32_t *seed_owner, DevChain *chn, DevSeed *seeds, uint8_t *seed_keep) {
    if (n_slots <= 0) return BM2_OK;
    if ((SSW_QMAX - 1) * o.a <= 255)       // (a window has < 200 columns: no score above 199 a; the row in LDS -- k_seed_sw<true>, 218 instead of 93 ms per 20 000 long reads -- left the tree in round 6)
        hipLaunchKernelGGL(k_seed_sw_reg, dim3((unsigned)((n_slots + 63) / 64)), dim3(64), 0, c->stream, c->ix, o, d_mat25, n_slots, enc, off, len, min_hsp,
                           seed_owner, seeds, seed_keep);
    else
        hipLaunchKernelGGL(k_seed_sw<false>, dim3((unsigned)((n_slots + 63) / 64)), dim3(64), 0, c->stream, c->ix, o, d_mat25, n_slots, enc, off, len, min_hsp,
                           seed_owner, seeds, seed_keep);
    hipLaunchKernelGGL(k_seed_flt_apply, dim3((n_reads + 127) / 128), dim3(128), 0, c->stream, n_reads, read_base, n_chain, min_hsp, chn, seeds, seed_keep);
    return bm2_check(hipGetLastError(), "seed filter launch");
}
