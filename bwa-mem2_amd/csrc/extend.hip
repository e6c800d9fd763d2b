// extend.hip -- the extension half of mem_chain2aln_across_reads_V2 (bwamem.cpp:2440-2994) on the device.
//
// The reference gathers every left task of a 512-read block into SeqPair arrays, sorts them by length, runs the
// inter-task SIMD kernels with band w, compacts the tasks that must be retried and runs them again with 2w, and then
// does the same for the right tasks (whose h0 is the left score) -- six sort/run/compact rounds per side
// (bwamem.cpp:2472-2880).  None of that batching is semantic: each seed's outcome depends only on its own two tasks.
// Here ONE WAVEFRONT OWNS ONE SEED (= one mem_alnreg_t): it runs the left extension (retrying with 2w in place when
// the reference would), feeds the score into the right extension, applies the clip/extend decision, and computes
// seedcov -- no SeqPair arrays, no reversed sequence copies, no host round trip between the two sides.
#include "bsw_dev.h"
#include "pipeline.h"
#include "chain_dev.h"

#define MAX_BAND_TRY 2            // bwamem.cpp:51

struct ExtParams {
    int32_t a, w, pen_clip5, pen_clip3;
    SwParams left, right;         // end_bonus = pen_clip5 / pen_clip3 (bwamem.cpp:2457-2463)
};

// one side, with the accept/retry rule of bwamem.cpp:2495-2496: stop when the score did not change, or the best cell
// stayed within 3/4 of the band, or this was the last try.
static __device__ __forceinline__ int extend_side(const uint8_t *q, int qs, int len2, const uint8_t *t, int ts, int len1, int h0,
                                                  int prev, int w0, const SwParams &P, int *RH, int *RE, int RM, SwOut &o,
                                                  int &w_used, long long &cells) {
    const int cls = pair_class(len1, len2, h0, P.max_sc);
    for (int i = 0; i < MAX_BAND_TRY; i++) {
        const int w = w0 << i;
        const int wc = band_clamp(w, len2, P, cls);
        cells += bsw_extend(q, qs, len2, t, ts, len1, wc, h0, P, RH, RE, RM, o);
        w_used = w;
        if (o.score == prev || o.max_off < (w >> 1) + (w >> 2) || i + 1 == MAX_BAND_TRY) break;
        prev = o.score;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    return o.score;
}

__global__ void __launch_bounds__(256)
k_extend(DevIndex ix, ExtParams xp, int64_t n_slots, const uint8_t *__restrict__ enc, const int64_t *__restrict__ off,
         const int32_t *__restrict__ len, const int64_t *__restrict__ slot_base /* per slot: base of its read */,
         const int32_t *__restrict__ reg_seed, const int32_t *__restrict__ reg_chain, const DevChain *__restrict__ chn,
         const DevSeed *__restrict__ seeds, DevReg *regs, int R, unsigned long long *counters /* [0]=cells [1]=tasks */) {
    extern __shared__ __attribute__((aligned(16))) int lds[];
    ExtParams *sP = (ExtParams *)lds;
    int *rings = lds + (sizeof(ExtParams) + 3) / 4;
    if (threadIdx.x < sizeof(ExtParams) / 4) ((int *)sP)[threadIdx.x] = ((const int *)&xp)[threadIdx.x];
    __syncthreads();
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int *RH = rings + (size_t)wv * 2 * R, *RE = RH + R;
    const int64_t g = (int64_t)blockIdx.x * (blockDim.x >> 6) + wv;
    if (g >= n_slots) return;
    const int sidx = reg_seed[g];
    if (sidx < 0) return;                                   // slot not used by any seed
    const int64_t base = slot_base[g];
    const DevChain c = chn[base + reg_chain[g]];
    const DevSeed s = seeds[base + sidx];
    const int r = c.read;
    const uint8_t *query = enc + off[r];
    const int l_query = len[r];
    const uint8_t *ref = ix.ref_string;
    // mem_alnreg_t initialisation, bwamem.cpp:2212-2223
    int64_t rb, re; int qb, qe, score = -1, truesc = -1, w = sP->w;
    long long cells = 0; int tasks = 0;
    SwOut o;
    if (s.qbeg) {                                           // left extension, bwamem.cpp:2229-2317 + :2472-2526
        const int len2 = s.qbeg, len1 = (int)(s.rbeg - c.rmax0), h0 = s.len * sP->a;
        int w_used;
        score = extend_side(query + s.qbeg - 1, -1, len2, ref + s.rbeg - 1, -1, len1, h0, -1, sP->w, sP->left, RH, RE, R - 1, o, w_used, cells);
        tasks++;
        if (o.gscore <= 0 || o.gscore <= score - sP->pen_clip5) { qb = s.qbeg - o.qle; rb = s.rbeg - o.tle; truesc = score; }
        else { qb = 0; rb = s.rbeg - o.gtle; truesc = o.gscore; }
        w = imax(w, w_used);
    } else {
        score = truesc = s.len * sP->a; qb = 0; rb = s.rbeg;
    }
    if (s.qbeg + s.len != l_query) {                        // right extension, bwamem.cpp:2324-2418 + :2672-2740
        const int qe0 = s.qbeg + s.len;
        const int64_t re0 = s.rbeg + s.len - c.rmax0;
        const int len2 = l_query - qe0, len1 = (int)(c.rmax1 - c.rmax0 - re0), h0 = score;
        int w_used;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int sc = extend_side(query + qe0, 1, len2, ref + c.rmax0 + re0, 1, len1, h0, score, sP->w, sP->right, RH, RE, R - 1, o, w_used, cells);
        tasks++;
        score = sc;
        if (o.gscore <= 0 || o.gscore <= score - sP->pen_clip3) { qe = qe0 + o.qle; re = c.rmax0 + re0 + o.tle; truesc += score - h0; }
        else { qe = l_query; re = c.rmax0 + re0 + o.gtle; truesc += o.gscore - h0; }
        w = imax(w, w_used);
    } else {
        qe = l_query; re = s.rbeg + s.len;
    }
    // seedcov over the chain's seeds, bwamem.cpp:2507-2516 (the H0_ guard is always true for real coordinates)
    int cov = 0;
    for (int i = lane; i < c.n; i += 64) {
        const DevSeed t = seeds[c.seed_off + i];
        if (t.qbeg >= qb && t.qbeg + t.len <= qe && t.rbeg >= rb && t.rbeg + t.len <= re) cov += t.len;
    }
    for (int d = 32; d > 0; d >>= 1) cov += __shfl_xor(cov, d);
    if (lane == 0) {
        DevReg a;
        a.rb = rb; a.re = re; a.qb = qb; a.qe = qe; a.rid = c.rid; a.score = score; a.truesc = truesc; a.w = w;
        a.seedcov = cov; a.seedlen0 = s.len; a.frac_rep = c.frac_rep; a.chain = reg_chain[g];
        regs[g] = a;
        atomicAdd(&counters[0], (unsigned long long)cells);
        atomicAdd(&counters[1], (unsigned long long)tasks);
    }
}

// cal_max_gap, bwamem.cpp:66-76
static __device__ __forceinline__ int cal_max_gap2(const ChainParams &o, int qlen) {
    const int l_del = (int)((double)(qlen * o.a - o.o_del) / o.e_del + 1.);
    const int l_ins = (int)((double)(qlen * o.a - o.o_ins) / o.e_ins + 1.);
    int l = l_del > l_ins ? l_del : l_ins;
    l = l > 1 ? l : 1;
    return l < o.w << 1 ? l : o.w << 1;
}

// Redundant-seed post-filter, bwamem.cpp:2895-2989 (one read per lane): replays the original bwa-mem rule "skip a seed
// already contained in an earlier alignment unless an overlapping seed lies on another diagonal" and purges those regs.
__global__ void __launch_bounds__(128)
k_postfilter(ChainParams o, int n_reads, const int32_t *__restrict__ len, const int64_t *__restrict__ read_base,
             const int32_t *__restrict__ n_chain, const int32_t *__restrict__ n_reg, const DevChain *__restrict__ chn,
             const DevSeed *__restrict__ seeds, int32_t *srt_all, DevReg *regs, int32_t *n_out) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    const int nc = n_chain[r], nr = n_reg[r];
    if (nc == 0) { n_out[r] = 0; return; }
    const int64_t base = read_base[r];
    const int l_query = len[r];
    DevReg *av = regs + base;
    int lim = 0;
    for (int j = 0; j < nc; j++) {
        const DevChain c = chn[base + j];
        const DevSeed *cs = seeds + c.seed_off;
        int32_t *srt2 = srt_all + c.seed_off;
        for (int k = c.n - 1; k >= 0; k--) {
            const DevSeed s = cs[srt2[k]];
            int i, v = 0;
            for (i = 0; i < nr && v < lim; ++i) {
                const DevReg p = av[i];
                int64_t rd; int qd, w, max_gap;
                if (p.qb == -1 && p.qe == -1) continue;
                if (s.rbeg < p.rb || s.rbeg + s.len > p.re || s.qbeg < p.qb || s.qbeg + s.len > p.qe) { v++; continue; }
                if (s.len - p.seedlen0 > .1 * l_query) { v++; continue; }
                qd = s.qbeg - p.qb; rd = s.rbeg - p.rb;
                max_gap = cal_max_gap2(o, qd < rd ? qd : (int)rd);
                w = max_gap < p.w ? max_gap : p.w;
                if (qd - rd < w && rd - qd < w) break;
                qd = p.qe - (s.qbeg + s.len); rd = p.re - (s.rbeg + s.len);
                max_gap = cal_max_gap2(o, qd < rd ? qd : (int)rd);
                w = max_gap < p.w ? max_gap : p.w;
                if (qd - rd < w && rd - qd < w) break;
                v++;
            }
            if (v < lim) {
                for (v = k + 1; v < c.n; ++v) {
                    if (srt2[v] < 0) continue;              // UINT_MAX marker of the reference
                    const DevSeed t = cs[srt2[v]];
                    if (t.len < s.len * .95) continue;
                    if (s.qbeg <= t.qbeg && s.qbeg + s.len - t.qbeg >= s.len >> 2 && t.qbeg - s.qbeg != t.rbeg - s.rbeg) break;
                    if (t.qbeg <= s.qbeg && t.qbeg + t.len - s.qbeg >= s.len >> 2 && s.qbeg - t.qbeg != s.rbeg - t.rbeg) break;
                }
                if (v == c.n) {
                    av[s.aln].qb = -1; av[s.aln].qe = -1;
                    srt2[k] = -1;
                    continue;
                }
            }
            lim++;
        }
    }
    int m = 0;
    for (int i = 0; i < nr; i++) if (av[i].qe > av[i].qb) m++;       // bwamem.cpp:1141-1152
    n_out[r] = m;
}

// compact the surviving regs into read order
__global__ void __launch_bounds__(256)
k_reg_gather(int n_reads, const int64_t *__restrict__ read_base, const int32_t *__restrict__ n_reg, const DevReg *__restrict__ regs,
             const int64_t *__restrict__ out_off, bm2_reg_t *out, int64_t out_cap) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    const int nr = n_reg[r];
    const DevReg *av = regs + read_base[r];
    int64_t o = out_off[r];
    for (int i = 0; i < nr; i++) {
        const DevReg a = av[i];
        if (a.qe > a.qb) {
            if (o < out_cap) {
                bm2_reg_t w;
                w.rb = a.rb; w.re = a.re; w.qb = a.qb; w.qe = a.qe; w.rid = a.rid; w.score = a.score; w.truesc = a.truesc;
                w.w = a.w; w.seedcov = a.seedcov; w.seedlen0 = a.seedlen0; w.frac_rep = a.frac_rep; w.pad = 0;
                out[o] = w;
            }
            o++;
        }
    }
}

// per-slot base of the owning read (slots of read r are [base, base + n_reg[r]))
__global__ void __launch_bounds__(256)
k_slot_base(int n_reads, const int64_t *__restrict__ read_base, const int32_t *__restrict__ n_reg, int64_t *slot_base) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    const int64_t b = read_base[r];
    for (int i = 0; i < n_reg[r]; i++) slot_base[b + i] = b;
}

static int ring_size2(int w) { int R = 64; while (R < 2 * w + 4) R <<= 1; return R; }

int bm2_launch_extend(bm2_ctx *c, const bm2_opt &opt, int64_t n_slots, const uint8_t *enc, const int64_t *off, const int32_t *len,
                      const int64_t *slot_base, const int32_t *reg_seed, const int32_t *reg_chain, const DevChain *chn,
                      const DevSeed *seeds, DevReg *regs, unsigned long long *counters) {
    if (n_slots <= 0) return BM2_OK;
    ExtParams xp;
    xp.a = opt.a; xp.w = opt.w; xp.pen_clip5 = opt.pen_clip5; xp.pen_clip3 = opt.pen_clip3;
    SwParams P;
    P.o_del = opt.o_del; P.e_del = opt.e_del; P.o_ins = opt.o_ins; P.e_ins = opt.e_ins; P.zdrop = opt.zdrop; P.max_sc = opt.a;
    for (int i = 0; i < 25; i++) P.mat[i] = opt.mat[i];
    P.end_bonus = opt.pen_clip5; xp.left = P;
    P.end_bonus = opt.pen_clip3; xp.right = P;
    const int R = ring_size2(opt.w << (MAX_BAND_TRY - 1));
    const int waves = 4;
    const size_t lds = ((sizeof(ExtParams) + 3) / 4) * 4 + (size_t)waves * 2 * R * 4;
    if (lds > 160 * 1024) { bm2_set_error("band width %d needs more LDS than a CU has", opt.w); return BM2_EUNSUP; }
    hipLaunchKernelGGL(k_extend, dim3((unsigned)((n_slots + waves - 1) / waves)), dim3(waves * 64), lds, c->stream, c->ix, xp,
                       n_slots, enc, off, len, slot_base, reg_seed, reg_chain, chn, seeds, regs, R, counters);
    return bm2_check(hipGetLastError(), "k_extend launch");
}

int bm2_launch_slot_base(bm2_ctx *c, int n_reads, const int64_t *read_base, const int32_t *n_reg, int64_t *slot_base) {
    if (n_reads <= 0) return BM2_OK;
    hipLaunchKernelGGL(k_slot_base, dim3((n_reads + 255) / 256), dim3(256), 0, c->stream, n_reads, read_base, n_reg, slot_base);
    return bm2_check(hipGetLastError(), "k_slot_base launch");
}

int bm2_launch_postfilter(bm2_ctx *c, const ChainParams &o, int n_reads, const int32_t *len, const int64_t *read_base,
                          const int32_t *n_chain, const int32_t *n_reg, const DevChain *chn, const DevSeed *seeds,
                          int32_t *srt_all, DevReg *regs, int32_t *n_out) {
    if (n_reads <= 0) return BM2_OK;
    hipLaunchKernelGGL(k_postfilter, dim3((n_reads + 127) / 128), dim3(128), 0, c->stream, o, n_reads, len, read_base, n_chain,
                       n_reg, chn, seeds, srt_all, regs, n_out);
    return bm2_check(hipGetLastError(), "k_postfilter launch");
}

int bm2_launch_reg_gather(bm2_ctx *c, int n_reads, const int64_t *read_base, const int32_t *n_reg, const DevReg *regs,
                          const int64_t *out_off, bm2_reg_t *out, int64_t out_cap) {
    if (n_reads <= 0) return BM2_OK;
    hipLaunchKernelGGL(k_reg_gather, dim3((n_reads + 255) / 256), dim3(256), 0, c->stream, n_reads, read_base, n_reg, regs,
                       out_off, out, out_cap);
    return bm2_check(hipGetLastError(), "k_reg_gather launch");
}
