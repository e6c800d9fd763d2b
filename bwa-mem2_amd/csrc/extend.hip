// extend.hip -- the extension half of mem_chain2aln_across_reads_V2 (bwamem.cpp:2440-2994) on the device.
//
// The reference gathers every left task of a 512-read block into SeqPair arrays, sorts them by length, runs the
// inter-task SIMD kernels with band w, compacts the tasks that must be retried and runs them again with 2w, and then
// does the same for the right tasks (whose h0 is the left score) -- six sort/run/compact rounds per side
// (bwamem.cpp:2472-2880).  None of that batching is semantic: each seed's outcome depends only on its own two tasks.
// Here ONE LANE OWNS ONE SEED (= one mem_alnreg_t): it runs the left extension (retrying with 2w in place when the
// reference would), feeds the score into the right extension, applies the clip/extend decision -- no SeqPair arrays, no
// reversed sequence copies, no launch boundary between the two sides.  The seeds of a round are counting-sorted by
// (class of max(left, right) query length, left length, right length): the 64 seeds of a wavefront have the same
// geometry on BOTH sides (with reads of one length also the same seed length, i.e. the same h0), so the lanes walk the
// same rows and nearly the same bands (tools/ext_sim.py: 0.88 of the lane slots of the column loop hold a cell).
// The sort, its prefix sums and the launch sizes all stay on the device: a round makes no trip to the host.
#include "bsw_dev.h"
#include "pipeline.h"
#include "chain_dev.h"
#include <string.h>

#define MAX_BAND_TRY 2            // bwamem.cpp:51

struct ExtParams {
    int32_t a, w, pen_clip5, pen_clip3;
    SwParams left, right;         // end_bonus = pen_clip5 / pen_clip3 (bwamem.cpp:2457-2463)
};

#define LANE_QMAX 160             // longest query the lane-per-seed kernel takes; longer ones go one seed per wavefront
#define N_CLS 10                  // LDS classes of the lane kernel: max(left, right) query length in steps of 16
#define EB_L (LANE_QMAX + 1)
#define EB_2D (EB_L * EB_L)       // bins of one class: (left length, right length)
#define EBIN_FALLBACK (N_CLS * EB_2D)          // seeds of the wavefront kernel, ordered by (left + right) >> 3
#define N_EBINS ((N_CLS + 1) * EB_2D)
#define EBIN_NONE 0xffffffffu     // nothing to extend

// geometry of the two extension tasks of a seed (what a SeqPair + its seqBuf slices describe, bwamem.cpp:2229-2418)
struct TaskGeom { const uint8_t *q; RefPtr t; int qs, ts, len2, len1; };
static __device__ __forceinline__ TaskGeom task_geom(int side, const DevSeed &s, const DevChain &c, const uint8_t *query,
                                                     int l_query, RefPtr ref) {
    TaskGeom g;
    if (side == 0) {            // left: query prefix and reference prefix, both walked backwards
        g.len2 = s.qbeg; g.len1 = (int)(s.rbeg - c.rmax0);
        g.q = query + s.qbeg - 1; g.qs = -1; g.t = ref + (s.rbeg - 1); g.ts = -1;
    } else {
        const int qe0 = s.qbeg + s.len;
        const int64_t re0 = s.rbeg + s.len - c.rmax0;
        g.len2 = l_query - qe0; g.len1 = (int)(c.rmax1 - c.rmax0 - re0);
        g.q = query + qe0; g.qs = 1; g.t = ref + (c.rmax0 + re0); g.ts = 1;
    }
    return g;
}

// accept/clip decision of one finished side, bwamem.cpp:2497-2505 (left) and :2715-2722 (right)
static __device__ __forceinline__ void apply_side(int side, DevReg &a, const DevSeed &s, int l_query, const SwOut &o, int h0,
                                                  int w_used, int pen_clip) {
    a.score = o.score;
    if (side == 0) {
        if (o.gscore <= 0 || o.gscore <= a.score - pen_clip) { a.qb = s.qbeg - o.qle; a.rb = s.rbeg - o.tle; a.truesc = a.score; }
        else { a.qb = 0; a.rb = s.rbeg - o.gtle; a.truesc = o.gscore; }
    } else {
        if (o.gscore <= 0 || o.gscore <= a.score - pen_clip) { a.qe += o.qle; a.re += o.tle; a.truesc += a.score - h0; }
        else { a.qe = l_query; a.re += o.gtle; a.truesc += o.gscore - h0; }
    }
    a.w = imax(a.w, w_used);
}

// seedcov of a finished reg: the bases of its chain's seeds that lie inside it, bwamem.cpp:2507-2516 (the H0_ guard there is always true for real
// coordinates).  Computed where the reg becomes final -- by the lane / wavefront that extended its seed, or where it is set up when there is nothing to
// extend -- since round 6 (k_seedcov / k_seedcov_round were launches of their own behind every phase: 0.35 ms and three launch gaps per chunk).
static __device__ __forceinline__ int seed_cover(const DevChain &c, const DevSeed *__restrict__ seeds, const DevReg &a) {
    int cov = 0;
    for (int i = 0; i < c.n; i++) {
        const DevSeed t = seeds[c.seed_off + i];
        if (t.qbeg >= a.qb && t.qbeg + t.len <= a.qe && t.rbeg >= a.rb && t.rbeg + t.len <= a.re) cov += t.len;
    }
    return cov;
}

// Sort key of a seed: which kernel / LDS class extends it and where it sits among that class's seeds.
static __device__ __forceinline__ uint32_t seed_bin(const ExtParams &xp, const DevSeed &s, const DevChain &c, int l_query) {
    const bool hl = s.qbeg != 0, hr = s.qbeg + s.len != l_query;
    if (!hl && !hr) return EBIN_NONE;
    int ll = 0, lr = 0; bool ok = true;
    for (int side = 0; side < 2; side++) {
        if (!(side == 0 ? hl : hr)) continue;
        const TaskGeom tg = task_geom(side, s, c, nullptr, l_query, RefPtr::bytes(nullptr));
        ok = ok && tg.len2 <= LANE_QMAX && tg.len1 < 32768 && (l_query + tg.len1 + 1) * xp.a < 32768;
        if (side == 0) ll = tg.len2; else lr = tg.len2;
    }
    if (!ok) return (uint32_t)(EBIN_FALLBACK + imin((ll + lr) >> 3, EB_2D - 1));
    return (uint32_t)(((imax(ll, lr) - 1) >> 4) * EB_2D + ll * EB_L + lr);
}

// ---- per-seed initialisation (mem_alnreg_t set-up of bwamem.cpp:2212-2223, 2318-2322, 2419-2437) and sort key, eager phase:
// every seed at or beyond its read's cursor
__global__ void __launch_bounds__(256)
k_reg_init(ExtParams xp, int64_t n_slots, const int32_t *__restrict__ len, const int64_t *__restrict__ slot_base,
           const int32_t *__restrict__ reg_seed, const int32_t *__restrict__ reg_chain, const DevChain *__restrict__ chn,
           const DevSeed *__restrict__ seeds, DevReg *regs, uint32_t *ebin /* [n_slots] */, int32_t *hist /* [N_EBINS] */,
           const int32_t *__restrict__ cursor /* per read: regs below this index were handled by the lazy rounds */) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_slots) return;
    const int sidx = reg_seed[g];
    uint32_t b = EBIN_NONE;
    if (sidx >= 0 && g - slot_base[g] >= cursor[chn[slot_base[g] + reg_chain[g]].read]) {
        const int64_t base = slot_base[g];
        const DevChain c = chn[base + reg_chain[g]];
        const DevSeed s = seeds[base + sidx];
        const int l_query = len[c.read];
        DevReg a;
        a.w = xp.w; a.rid = c.rid; a.frac_rep = c.frac_rep; a.seedlen0 = s.len; a.chain = reg_chain[g]; a.seedcov = 0;
        a.rb = s.rbeg; a.re = s.rbeg + s.len;
        if (s.qbeg) { a.score = a.truesc = -1; a.qb = s.qbeg; }
        else { a.score = a.truesc = s.len * xp.a; a.qb = 0; }
        a.qe = (s.qbeg + s.len != l_query) ? s.qbeg + s.len : l_query;
        b = seed_bin(xp, s, c, l_query);
        if (b != EBIN_NONE) atomicAdd(&hist[b], 1);
        else a.seedcov = seed_cover(c, seeds, a);                // nothing to extend: the reg is final
        regs[g] = a;
    }
    ebin[g] = b;
}

// Counting sort of the round's seeds by their key: `start` = exclusive prefix sums of the histogram (bm2_scan_i32), and the histogram
// itself counts back down to zero as the slots are handed out -- it is clean for the next round without a memset.
__global__ void __launch_bounds__(256)
k_task_scatter(int64_t n_items, const uint32_t *__restrict__ ebin, int32_t *hist, const int64_t *__restrict__ start, int32_t *tasks,
               const int64_t *__restrict__ read_base /* round mode: item = read */, const int32_t *__restrict__ cur_slot) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_items) return;
    const uint32_t b = ebin[g];
    if (b == EBIN_NONE) return;
    const int k = atomicAdd(&hist[b], -1) - 1;
    tasks[start[b] + k] = read_base ? (int32_t)(read_base[g] + cur_slot[g]) : (int32_t)g;
}

// what the host wants to know about a phase AFTER the batch (it sizes the next batch's launches with it): seeds per class, reads still pending
__global__ void k_phase_stats(const int64_t *__restrict__ start, const uint32_t *__restrict__ pend, uint32_t *out /* [N_CLS + 2] */) {
    const int t = threadIdx.x;
    if (t <= N_CLS) out[t] = (uint32_t)(start[(int64_t)(t + 1) * EB_2D] - start[(int64_t)t * EB_2D]);
    if (t == N_CLS + 1) out[t] = pend ? *pend : 0u;
}

// ---- lane-per-task extension: the inter-task SIMD shape of the reference (one pair per lane, bandedSWA.cpp:436-1113)
// on 64-wide wavefronts.  Each lane runs the scalar recurrence of ksw_extend2 on its own task; the row state eh[j] =
// {H(i-1,j-1), E(i,j)} is packed 16+16 bits in LDS, laid out [column][lane] so a wavefront touches 64 consecutive
// dwords (conflict-free); the query bases sit next to it as [column][lane] bytes.  Tasks arrive sorted by query length,
// so the lanes of a wavefront walk almost the same loop bounds.  All lanes step the same (i, j); a lane outside its own
// band or past its own exit is masked off.
struct LaneOut { int score, qle, tle, gtle, gscore, max_off; };

static __device__ void lane_dp(bool run, int qlen, int tlen, int w, int h0, RefPtr tp, int ts, const SwParams &P,
                               uint32_t *EH, const uint8_t *QL, int lane, LaneOut &out, long long &cells, long long &iters) {
    const int o_del = P.o_del, e_del = P.e_del, o_ins = P.o_ins, e_ins = P.e_ins, oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
    const int sc_match = P.mat[0], sc_mis = P.mat[1], sc_amb = P.mat[4];
    // first row, bandedSWA.cpp:143-145
    const int e1 = h0 > oe_ins ? h0 - oe_ins : 0;
    const int cls = pair_class(tlen, qlen, h0, P.max_sc);        // which of the reference's kernels runs this pair: its Z-drop rule
    const int maxq = __builtin_amdgcn_readlane(wave_scan_max(run ? qlen : 0, 0), 63);
    for (int j = 0; j <= maxq; j++)
        if (run && j <= qlen) EH[j * 64 + lane] = (uint32_t)(j == 0 ? h0 : imax(e1 - (j - 1) * e_ins, 0));
    int beg = 0, end = qlen, maxv = h0, max_i = -1, max_j = -1, max_ie = -1, gscore = -1, max_off = 0;
    bool alive = run && tlen > 0;
    const int maxt = __builtin_amdgcn_readlane(wave_scan_max(alive ? tlen : 0, 0), 63);
    int t_next = alive ? (int)tp[0] : 4;
    for (int i = 0; i < maxt; ++i) {
        if (!__ballot(alive)) break;
        const int tb = t_next;
        if (alive && i + 1 < tlen) t_next = (int)tp[(int64_t)(i + 1) * ts];
        int h1 = 0, f = 0, m = 0, mj = -1, fnz = -1, lnz = -1;
        if (alive) {
            if (beg < i - w) beg = i - w;
            if (end > i + w + 1) end = i + w + 1;
            if (end > qlen) end = qlen;
            if (beg == 0) { h1 = h0 - (o_del + e_del * (i + 1)); if (h1 < 0) h1 = 0; }
            cells += imax(end - beg, 0);
        }
        const int s_eq = tb > 3 ? sc_amb : sc_match;
        const int jlo = (1 << 20) - __builtin_amdgcn_readlane(wave_scan_max(alive ? (1 << 20) - beg : 0, 0), 63);
        const int jhi = __builtin_amdgcn_readlane(wave_scan_max(alive ? end : 0, 0), 63);
        iters += jhi > jlo ? (jhi - jlo + 1) >> 1 : 0;           // (wave-uniform: trips of the column loop, in column pairs)
#pragma unroll 2
        for (int j = jlo; j < jhi; ++j) {
            if (alive && j >= beg && j < end) {
                const uint32_t p = EH[j * 64 + lane];
                const int qb = QL[j * 64 + lane];
                const int e = (int)(p >> 16);
                int M = (int)(p & 0xffffu);
                const int sc = (qb == tb && tb < 4) ? s_eq : ((qb > 3 || tb > 3) ? sc_amb : sc_mis);
                M = M ? M + sc : 0;
                int h = M > e ? M : e;
                h = h > f ? h : f;
                mj = m > h ? mj : j;
                m = m > h ? m : h;
                const int en = imax(isub0(e, e_del), M - oe_del);
                f = imax(isub0(f, e_ins), M - oe_ins);
                const uint32_t nw = (uint32_t)h1 | ((uint32_t)en << 16);
                EH[j * 64 + lane] = nw;
                if (nw) { lnz = j; if (fnz < 0) fnz = j; }
                h1 = h;
            }
        }
        if (alive) {
            EH[end * 64 + lane] = (uint32_t)h1;                       // eh[end] = {h1, 0}, bandedSWA.cpp:201
            if (h1) lnz = end;
            const int jfin = beg < end ? end : beg;
            if (jfin == qlen) { max_ie = gscore > h1 ? max_ie : i; gscore = gscore > h1 ? gscore : h1; }
            if (m == 0) alive = false;
            else {
                const bool new_max = m > maxv;
                if (new_max) {
                    maxv = m; max_i = i; max_j = mj;
                    const int d = mj - i;
                    max_off = imax(max_off, d < 0 ? -d : d);
                }
                if (zdrop_stop(cls, new_max, maxv, m, i - max_i, mj - max_j, e_del, e_ins, P.zdrop)) alive = false;
                const int nb = fnz >= 0 ? fnz : end;
                const int jl = imax(lnz, nb - 1);
                beg = nb;
                end = jl + 2 < qlen ? jl + 2 : qlen;
                if (i + 1 >= tlen) alive = false;
            }
        }
    }
    if (run) { out.score = maxv; out.qle = max_j + 1; out.tle = max_i + 1; out.gtle = max_ie + 1; out.gscore = gscore; out.max_off = max_off; }
}

// The same recurrence with the row packed 8+8 bits: usable when no score of the batch can exceed 255 (l_query * a <= 255,
// i.e. every short-read run with the default scoring).  Two columns share one LDS dword -- {H[j], E[j], H[j+1], E[j+1]}
// -- and the query sits 8 bases to a dword, so a cell costs half an LDS read and half an LDS write instead of two reads
// and a write, and a wave's rows take 2.6x less LDS: 6 instead of 3 wavefronts per CU for 150-base queries.
// PF: the row word of the NEXT column pair is requested (unconditionally: one row past the band still lies inside the block's
// LDS) before the current pair is computed, so the LDS round trip overlaps the ~80 VALU instructions of a pair instead of
// preceding them; what it buys depends on how many wavefronts share the SIMD (few, for the long classes).
// PT: the query sits one base per BYTE (4 to a dword) and the cell's score comes from a byte permute: the row's five scores (against
// A, C, G, T, N) are five bytes of a register pair, ONE v_perm_b32 with the query word as its selector turns four query bases into
// their four scores -- the two compares and two selects per cell of the 4-bit layout are gone (27 -> 22 VALU per cell).
static __device__ __forceinline__ uint32_t rep4(int x) { return ((uint32_t)x & 0xffu) * 0x01010101u; }
template <bool PF, bool PT = false>
static __device__ void lane_dp8(bool run, int qlen, int tlen, int w, int h0, RefPtr tp, int ts, const SwParams &P,
                                uint32_t *EH, const uint32_t *QL, int lane, LaneOut &out, long long &cells, long long &iters) {
    const int o_del = P.o_del, e_del = P.e_del, o_ins = P.o_ins, e_ins = P.e_ins, oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
    int sc_match = P.mat[0], sc_mis = P.mat[1], sc_amb = P.mat[4];
    // (the three scores are operands of v_cndmask in every cell, which takes both sources from VGPRs: kept there, or the compiler
    //  re-materialises them from SGPRs with a v_mov per use -- 3 of a cell's 33 VALU instructions)
    asm volatile("" : "+v"(sc_match), "+v"(sc_mis), "+v"(sc_amb));
    const uint32_t rep_mis = rep4(sc_mis), rep_amb = rep4(sc_amb);
    const int e1 = h0 > oe_ins ? h0 - oe_ins : 0;                // first row, bandedSWA.cpp:143-145
    const int cls = pair_class(tlen, qlen, h0, P.max_sc);
    const int maxq = __builtin_amdgcn_readlane(wave_scan_max(run ? qlen : 0, 0), 63);
    for (int jp = 0; jp <= maxq; jp += 2) {
        if (run && jp <= qlen) {
            const uint32_t v0 = (uint32_t)(jp == 0 ? h0 : imax(e1 - (jp - 1) * e_ins, 0));
            const uint32_t v1 = jp + 1 <= qlen ? (uint32_t)imax(e1 - jp * e_ins, 0) : 0u;
            EH[(jp >> 1) * 64 + lane] = v0 | v1 << 16;
        }
    }
    int beg = 0, end = qlen, maxv = h0, max_i = -1, max_j = -1, max_ie = -1, gscore = -1, max_off = 0;
    bool alive = run && tlen > 0;
    const int maxt = __builtin_amdgcn_readlane(wave_scan_max(alive ? tlen : 0, 0), 63);
    int t_next = alive ? (int)tp[0] : 4;
    for (int i = 0; i < maxt; ++i) {
        if (!__ballot(alive)) break;
        const int tb = t_next;
        if (alive && i + 1 < tlen) t_next = (int)tp[(int64_t)(i + 1) * ts];
        // The row's maximum and the LAST column holding it (bandedSWA.cpp:188-189) as one running maximum over h << 8 | j (h and j are
        // below 256 in this class); the first / last column with a non-zero stored cell as an unsigned minimum / a signed maximum over
        // (cell != 0 ? j : -1): -1 is "none" in both.
        int h1 = 0, f = 0, lnz = -1;
        unsigned key = 0, fnz_u = 0xffffffffu;
        if (alive) {
            if (beg < i - w) beg = i - w;
            if (end > i + w + 1) end = i + w + 1;
            if (end > qlen) end = qlen;
            if (beg == 0) { h1 = h0 - (o_del + e_del * (i + 1)); if (h1 < 0) h1 = 0; }
            cells += imax(end - beg, 0);
        }
        const int s_eq = tb > 3 ? sc_amb : sc_match;         // against an equal query code (both N: ambiguous)
        const int s_ne = tb > 3 ? sc_amb : sc_mis;           // against a different base
        const int jlo = (1 << 20) - __builtin_amdgcn_readlane(wave_scan_max(alive ? (1 << 20) - beg : 0, 0), 63);
        const int jhi = __builtin_amdgcn_readlane(wave_scan_max(alive ? end : 0, 0), 63);
        const int jp0 = jlo & ~1;
        iters += jhi > jp0 ? (jhi - jp0 + 1) >> 1 : 0;           // (wave-uniform: trips of the column-pair loop)
        // PT: scores of this row against the query codes 0..4: bytes 0..3 of t_lo (the target's own base: match; target N: ambiguous), byte 0 of t_hi
        const uint32_t t_lo = tb > 3 ? rep_amb : rep_mis ^ ((uint32_t)((sc_mis ^ sc_match) & 0xff) << (8 * tb)), t_hi = rep_amb;
        uint32_t qw, qnext = 0u;
        if (PT) {                                                                          // 4 score bytes per word; `qw` holds the current pair's in its low half
            qw = jp0 < jhi ? __builtin_amdgcn_perm(t_hi, t_lo, QL[(jp0 >> 2) * 64 + lane]) >> (8 * (jp0 & 3)) : 0u;
            qnext = ((jp0 | 3) + 1) < jhi ? QL[((jp0 >> 2) + 1) * 64 + lane] : 0u;
        } else qw = jp0 < jhi ? QL[(jp0 >> 3) * 64 + lane] >> (4 * (jp0 & 7)) : 0u;       // 8 query bases per word; the next word when a pair starts one
        uint32_t wnext = PF ? EH[(jp0 >> 1) * 64 + lane] : 0u;
        for (int jp = jp0; jp < jhi; jp += 2) {
            const uint32_t wcur = wnext;
            if (PF) wnext = EH[((jp >> 1) + 1) * 64 + lane];
            if (alive && jp + 1 >= beg && jp < end) {
                uint32_t word = PF ? wcur : EH[(jp >> 1) * 64 + lane];
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    const int j = jp + u;
                    if (j >= beg && j < end) {
                        const int qb = (int)((qw >> (4 * u)) & 15u);
                        const int e = (int)((word >> (16 * u + 8)) & 0xffu);
                        int M = (int)((word >> (16 * u)) & 0xffu);
                        const int sc = PT ? (int)(int8_t)(qw >> (8 * u)) : qb == tb ? s_eq : (qb > 3 ? sc_amb : s_ne);
                        M = M ? M + sc : 0;
                        int h = M > e ? M : e;
                        h = h > f ? h : f;
                        const unsigned kj = (unsigned)h << 8 | (unsigned)j;
                        key = key > kj ? key : kj;
                        const int en = imax(isub0(e, e_del), M - oe_del);     // = max(e - e_del, M - oe_del, 0): e, f >= 0; a negative M loses anyway
                        f = imax(isub0(f, e_ins), M - oe_ins);
                        const uint32_t nw = (uint32_t)h1 | ((uint32_t)en << 8);
                        word = (word & ~(0xffffu << (16 * u))) | nw << (16 * u);
                        const int jj = nw ? j : -1;
                        lnz = lnz > jj ? lnz : jj;
                        fnz_u = fnz_u < (unsigned)jj ? fnz_u : (unsigned)jj;
                        h1 = h;
                    }
                }
                EH[(jp >> 1) * 64 + lane] = word;
            }
            if (PT) {
                qw >>= 16;
                if (((jp + 2) & 3) == 0) { qw = __builtin_amdgcn_perm(t_hi, t_lo, qnext); if (jp + 6 < jhi) qnext = QL[((jp + 2) >> 2) * 64 + 64 + lane]; }
            } else {
                qw >>= 8;
                if (((jp + 2) & 7) == 0 && jp + 2 < jhi) qw = QL[((jp + 2) >> 3) * 64 + lane];
            }
        }
        const int m = (int)(key >> 8), mj = (int)(key & 255u), fnz = (int)fnz_u;
        if (alive) {
            uint32_t word = EH[(end >> 1) * 64 + lane];                // eh[end] = {h1, 0}, bandedSWA.cpp:201
            word = (word & ~(0xffffu << (16 * (end & 1)))) | (uint32_t)h1 << (16 * (end & 1));
            EH[(end >> 1) * 64 + lane] = word;
            if (h1) lnz = end;
            const int jfin = beg < end ? end : beg;
            if (jfin == qlen) { max_ie = gscore > h1 ? max_ie : i; gscore = gscore > h1 ? gscore : h1; }
            if (m == 0) alive = false;
            else {
                const bool new_max = m > maxv;
                if (new_max) {
                    maxv = m; max_i = i; max_j = mj;
                    const int d = mj - i;
                    max_off = imax(max_off, d < 0 ? -d : d);
                }
                if (zdrop_stop(cls, new_max, maxv, m, i - max_i, mj - max_j, e_del, e_ins, P.zdrop)) alive = false;
                const int nb = fnz >= 0 ? fnz : end;
                const int jl = imax(lnz, nb - 1);
                beg = nb;
                end = jl + 2 < qlen ? jl + 2 : qlen;
                if (i + 1 >= tlen) alive = false;
            }
        }
    }
    if (run) { out.score = maxv; out.qle = max_j + 1; out.tle = max_i + 1; out.gtle = max_ie + 1; out.gscore = gscore; out.max_off = max_off; }
}

// (Measured in round 6 and removed, profiles/r06r_sweep_fast_groups.json: a ballot per group and UNTESTED cells for the groups that lie inside the band of every live lane of the
//  wavefront -- bit-exact on the emulator, extension 14.46 -> 14.38 ms with the 128-column class on register rows, 15.21 -> 15.58 with every class on LDS rows, seam S1 805 -> 808
//  G cells/s: the lane kernel does not wait for its band tests.)
// The same rows again, the column loop in GROUPS OF FOUR columns -- one query word (four bases, one v_perm_b32 for their four scores) and two
// row words per trip -- with the next group's three LDS words requested into a SECOND set of registers before the current group is computed
// (two copies of the group body that hand the sets to each other: no register copy, and so no wait, between a request and its use one trip
// later).  What that removes (profiles/r04c_pmc_sq1_lanes_only.md: a wavefront of the pair loop above issued VALU in 14 % of its cycles and
// waited in 72 %): the compiler ended every trip of the pair loop with `s_waitcnt lgkmcnt(0)` + `v_mov` for the carried row word -- a wait for
// the store it had just issued -- and every second trip with the same for the query word requested one instruction earlier.
// LDS: rows [2 NG + 2][64] words, query [NG + 1][64] words, NG = (qmax + 3) / 4 (one group of slack for the request beyond the last group).
struct Dp8Row { int h1, f, lnz; unsigned key, fnz_u; };
static __device__ __forceinline__ void dp8_group(int g, uint32_t w0, uint32_t w1, uint32_t q, bool alive, int beg, int end, uint32_t t_lo, uint32_t t_hi,
                                                 int e_del, int e_ins, int oe_del, int oe_ins, uint32_t *EH, int lane, Dp8Row &r) {
    const int j0 = g << 2;
    if (alive && j0 + 3 >= beg && j0 < end) {
        const uint32_t sc4 = __builtin_amdgcn_perm(t_hi, t_lo, q);
        uint32_t word[2] = { w0, w1 };
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int j = j0 + u;
            if (j >= beg && j < end) {
                uint32_t &wd = word[u >> 1];
                const int sh = 16 * (u & 1);
                const int e = (int)((wd >> (sh + 8)) & 0xffu);
                int M = (int)((wd >> sh) & 0xffu);
                const int sc = (int)(int8_t)(sc4 >> (8 * u));
                M = M ? M + sc : 0;
                int h = M > e ? M : e;
                h = h > r.f ? h : r.f;
                const unsigned kj = (unsigned)h << 8 | (unsigned)j;
                r.key = r.key > kj ? r.key : kj;
                const int en = imax(isub0(e, e_del), M - oe_del);
                r.f = imax(isub0(r.f, e_ins), M - oe_ins);
                const uint32_t nw = (uint32_t)r.h1 | ((uint32_t)en << 8);
                wd = (wd & ~(0xffffu << sh)) | nw << sh;
                const int jj = nw ? j : -1;
                r.lnz = r.lnz > jj ? r.lnz : jj;
                r.fnz_u = r.fnz_u < (unsigned)jj ? r.fnz_u : (unsigned)jj;
                r.h1 = h;
            }
        }
        EH[(2 * g) * 64 + lane] = word[0];                         // (a word none of whose columns lay in the band goes back unchanged)
        EH[(2 * g + 1) * 64 + lane] = word[1];
    }
}

static __device__ void lane_dp8g(bool run, int qlen, int tlen, int w, int h0, RefPtr tp, int ts, const SwParams &P,
                                 uint32_t *EH, const uint32_t *QL, int lane, LaneOut &out, long long &cells, long long &iters) {
    const int o_del = P.o_del, e_del = P.e_del, o_ins = P.o_ins, e_ins = P.e_ins, oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
    const int sc_match = P.mat[0], sc_mis = P.mat[1], sc_amb = P.mat[4];
    const uint32_t rep_mis = rep4(sc_mis), rep_amb = rep4(sc_amb);
    const int e1 = h0 > oe_ins ? h0 - oe_ins : 0;                // first row, bandedSWA.cpp:143-145
    const int cls = pair_class(tlen, qlen, h0, P.max_sc);
    const int maxq = __builtin_amdgcn_readlane(wave_scan_max(run ? qlen : 0, 0), 63);
    for (int jp = 0; jp <= maxq; jp += 2) {
        if (run && jp <= qlen) {
            const uint32_t v0 = (uint32_t)(jp == 0 ? h0 : imax(e1 - (jp - 1) * e_ins, 0));
            const uint32_t v1 = jp + 1 <= qlen ? (uint32_t)imax(e1 - jp * e_ins, 0) : 0u;
            EH[(jp >> 1) * 64 + lane] = v0 | v1 << 16;
        }
    }
    int beg = 0, end = qlen, maxv = h0, max_i = -1, max_j = -1, max_ie = -1, gscore = -1, max_off = 0;
    bool alive = run && tlen > 0;
    const int maxt = __builtin_amdgcn_readlane(wave_scan_max(alive ? tlen : 0, 0), 63);
    // The target bases.  From a byte buffer (seam S1, BM2_REF_BYTES): FOUR ROWS AT A TIME, one (unaligned) 32-bit load per four rows requested
    // four rows before its first use.  From the packed reference: TWENTY-EIGHT rows per (unaligned) 64-bit load -- 56 bits of bases and up to
    // 6 bits of misalignment -- requested a whole window (28 rows) before its first use.  A lane's target is a stream of its own somewhere in
    // the genome: every load instruction is 64 different lines, most of them TLB misses, microseconds; a wavefront of a short class passes
    // four rows in less than that and there is hardly a second wavefront on the SIMD to cover for it (rows in LDS: 1-2 wavefronts per SIMD).
    const bool pk = tp.pk != 0;
    constexpr int TW = 28;
    typedef uint64_t __attribute__((aligned(1))) u64_unaligned;
    const int64_t pa0 = ts > 0 ? tp.at : tp.at - (TW - 1);      // lowest position of window 0 (before the reference's start by < 28 at most: padded)
    const int psh = (int)(pa0 & 3) << 1, pstep = ts > 0 ? TW / 4 : -(TW / 4);
    const uint8_t *pbyte = tp.p + (pa0 >> 2);
    uint64_t pw_cur = 0, pw_next = 0;
    int pr = 0;
    if (pk && run && tlen > 0) {
        pw_cur = *(const u64_unaligned *)pbyte;
        if (TW < tlen) pw_next = *(const u64_unaligned *)(pbyte + pstep);
    }
    auto bases4 = [&](int i0) -> uint32_t {                      // byte k = base of row i0 + k (4 beyond the target's end)
        uint32_t wv = 0x04040404u;
        if (run && i0 < tlen) {
            if (i0 + 3 < tlen) wv = tp.load4(i0, ts);
            else {
                wv = 0;
                for (int k = 0; k < 4; k++) wv |= (uint32_t)(i0 + k < tlen ? tp[(int64_t)(i0 + k) * ts] : 4) << (8 * k);
            }
        }
        return wv;
    };
    uint32_t tw_cur = 0, tw_next = 0;
    if (!pk) { tw_cur = bases4(0); tw_next = bases4(4); }
    for (int i = 0; i < maxt; ++i) {
        if (!__ballot(alive)) break;
        int tb;
        if (pk) {
            tb = (int)(pw_cur >> (psh + 2 * (ts > 0 ? pr : TW - 1 - pr))) & 3;
            if (++pr == TW) {
                pr = 0; pw_cur = pw_next;
                if (run && i + 1 + TW < tlen) pw_next = *(const u64_unaligned *)(pbyte + (int64_t)((i + 1) / TW + 1) * pstep);
            }
        } else {
            tb = (int)((tw_cur >> (8 * (i & 3))) & 0xffu);
            if ((i & 3) == 3) { tw_cur = tw_next; tw_next = bases4(i + 5); }
        }
        Dp8Row r; r.h1 = 0; r.f = 0; r.lnz = -1; r.key = 0; r.fnz_u = 0xffffffffu;
        if (alive) {
            if (beg < i - w) beg = i - w;
            if (end > i + w + 1) end = i + w + 1;
            if (end > qlen) end = qlen;
            if (beg == 0) { r.h1 = h0 - (o_del + e_del * (i + 1)); if (r.h1 < 0) r.h1 = 0; }
            cells += imax(end - beg, 0);
        }
        const int jlo = (1 << 20) - __builtin_amdgcn_readlane(wave_scan_max(alive ? (1 << 20) - beg : 0, 0), 63);
        const int jhi = __builtin_amdgcn_readlane(wave_scan_max(alive ? end : 0, 0), 63);
        const int g0 = jlo >> 2, g1 = (jhi + 3) >> 2;                // groups [g0, g1)
        iters += g1 > g0 ? 2 * (g1 - g0) : 0;                        // (wave-uniform; in column pairs, like the pair loop's count)
        // scores of this row against the query codes 0..4: bytes 0..3 of t_lo (the target's own base: match; target N: ambiguous), byte 0 of t_hi
        const uint32_t t_lo = tb > 3 ? rep_amb : rep_mis ^ ((uint32_t)((sc_mis ^ sc_match) & 0xff) << (8 * tb)), t_hi = rep_amb;
        if (g0 < g1) {
            uint32_t a0 = EH[(2 * g0) * 64 + lane], a1 = EH[(2 * g0 + 1) * 64 + lane], aq = QL[g0 * 64 + lane], b0, b1, bq;
            for (int g = g0;; g += 2) {
                b0 = EH[(2 * g + 2) * 64 + lane]; b1 = EH[(2 * g + 3) * 64 + lane]; bq = QL[(g + 1) * 64 + lane];
                dp8_group(g, a0, a1, aq, alive, beg, end, t_lo, t_hi, e_del, e_ins, oe_del, oe_ins, EH, lane, r);
                if (g + 1 >= g1) break;
                a0 = EH[(2 * g + 4) * 64 + lane]; a1 = EH[(2 * g + 5) * 64 + lane]; aq = QL[(g + 2) * 64 + lane];
                dp8_group(g + 1, b0, b1, bq, alive, beg, end, t_lo, t_hi, e_del, e_ins, oe_del, oe_ins, EH, lane, r);
                if (g + 2 >= g1) break;
            }
        }
        const int m = (int)(r.key >> 8), mj = (int)(r.key & 255u), fnz = (int)r.fnz_u, h1 = r.h1;
        int lnz = r.lnz;
        if (alive) {
            uint32_t word = EH[(end >> 1) * 64 + lane];                // eh[end] = {h1, 0}, bandedSWA.cpp:201
            word = (word & ~(0xffffu << (16 * (end & 1)))) | (uint32_t)h1 << (16 * (end & 1));
            EH[(end >> 1) * 64 + lane] = word;
            if (h1) lnz = end;
            const int jfin = beg < end ? end : beg;
            if (jfin == qlen) { max_ie = gscore > h1 ? max_ie : i; gscore = gscore > h1 ? gscore : h1; }
            if (m == 0) alive = false;
            else {
                const bool new_max = m > maxv;
                if (new_max) {
                    maxv = m; max_i = i; max_j = mj;
                    const int d = mj - i;
                    max_off = imax(max_off, d < 0 ? -d : d);
                }
                if (zdrop_stop(cls, new_max, maxv, m, i - max_i, mj - max_j, e_del, e_ins, P.zdrop)) alive = false;
                const int nb = fnz >= 0 ? fnz : end;
                const int jl = imax(lnz, nb - 1);
                beg = nb;
                end = jl + 2 < qlen ? jl + 2 : qlen;
                if (i + 1 >= tlen) alive = false;
            }
        }
    }
    if (run) { out.score = maxv; out.qle = max_j + 1; out.tle = max_i + 1; out.gtle = max_ie + 1; out.gscore = gscore; out.max_off = max_off; }
}

// The same rows once more, IN REGISTERS (round 6): the classes of 80..128-base queries are each phase's longest launches -- a tile of 64 seeds is a
// dependent chain of ~10^5 instructions, and with 16-25 KB of LDS rows per wavefront only 6-10 wavefronts fit a CU (1.5-2.5 per SIMD), each waiting for
// its own LDS round trips.  Here a lane's row lives in 2 NG + 2 registers (two columns of {H, E} bytes per half word, as in LDS) and its query in NG + 1
// (one base per byte: the selector of the score permute): no LDS at all, so the register file alone bounds the wavefronts per SIMD (3 at 128 columns,
// 4 at 80) and a cell is its ~22 VALU instructions with nothing to wait for.  Registers have no dynamic index: the column loop is UNROLLED over the
// class's groups, each group behind a scalar test of the wavefront's band (groups outside [g0, g1) cost two scalar instructions), and the closing store
// of a row -- eh[end] = {h1, 0}, bandedSWA.cpp:201, a per-lane index -- is folded into the walk: the column after a lane's last cell takes it
// (the walk covers [min(beg, end), end] of every live lane).  Same recurrence, same masks, same order of the cells as lane_dp8g: bit-exact by construction,
// checked on the host emulator against the oracle before it met a GPU (tests/test_device_sources_on_host.py, tests/test_bsw_gpu.py).
struct TStream {                     // the target bases of one lane: 28 rows per 64-bit load from the 2-bit reference, 4 per 32-bit load from bytes (lane_dp8g's scheme)
    static constexpr int TW = 28;
    typedef uint64_t __attribute__((aligned(1))) u64_unaligned;
    RefPtr tp; int ts, tlen; bool run, pk;
    int psh, pstep, pr; const uint8_t *pbyte; uint64_t pw_cur, pw_next; uint32_t tw_cur, tw_next;
    __device__ __forceinline__ uint32_t bases4(int i0) const {   // byte k = base of row i0 + k (4 beyond the target's end)
        uint32_t wv = 0x04040404u;
        if (run && i0 < tlen) {
            if (i0 + 3 < tlen) wv = tp.load4(i0, ts);
            else {
                wv = 0;
                for (int k = 0; k < 4; k++) wv |= (uint32_t)(i0 + k < tlen ? tp[(int64_t)(i0 + k) * ts] : 4) << (8 * k);
            }
        }
        return wv;
    }
    __device__ __forceinline__ void init(bool run_, RefPtr tp_, int ts_, int tlen_) {
        run = run_; tp = tp_; ts = ts_; tlen = tlen_; pk = tp.pk != 0;
        const int64_t pa0 = ts > 0 ? tp.at : tp.at - (TW - 1);
        psh = (int)(pa0 & 3) << 1; pstep = ts > 0 ? TW / 4 : -(TW / 4);
        pbyte = tp.p + (pa0 >> 2);
        pw_cur = pw_next = 0; pr = 0; tw_cur = tw_next = 0;
        if (pk && run && tlen > 0) {
            pw_cur = *(const u64_unaligned *)pbyte;
            if (TW < tlen) pw_next = *(const u64_unaligned *)(pbyte + pstep);
        }
        if (!pk) { tw_cur = bases4(0); tw_next = bases4(4); }
    }
    __device__ __forceinline__ int next(int i) {                 // the base of row i (rows are asked for in order)
        int tb;
        if (pk) {
            tb = (int)(pw_cur >> (psh + 2 * (ts > 0 ? pr : TW - 1 - pr))) & 3;
            if (++pr == TW) {
                pr = 0; pw_cur = pw_next;
                if (run && i + 1 + TW < tlen) pw_next = *(const u64_unaligned *)(pbyte + (int64_t)((i + 1) / TW + 1) * pstep);
            }
        } else {
            tb = (int)((tw_cur >> (8 * (i & 3))) & 0xffu);
            if ((i & 3) == 3) { tw_cur = tw_next; tw_next = bases4(i + 5); }
        }
        return tb;
    }
};

// One group of four columns on registers: the cells inside the lane's band, and -- at column `end` -- the row's closing store.  NO per-column constant may
// appear in it: a literal cannot be an operand of the three-operand VALU encodings, so the compiler put every column number beyond 64 into a register of
// its own, hoisted out of the row loop -- 68 registers at 128 columns (the first build: 358 / 328 registers, one wavefront per SIMD).  Hence: the group's
// first column j0 arrives as a run-time SCALAR (an opaque zero plus 4 g), the band is taken relative to it so that a cell compares with 0..3, the row
// maximum's column and the non-zero cells are kept per group (key | u, one mask bit per cell) and folded into the row's state once per group.
struct Dp8RowR { int h1, f; unsigned key; };
template <int NZW>
static __device__ __forceinline__ void dp8_group_reg(int g, int j0, uint32_t &w0, uint32_t &w1, uint32_t q, bool alive, int lo, int beg, int end, uint32_t t_lo, uint32_t t_hi,
                                                     int e_del, int e_ins, int oe_del, int oe_ins, Dp8RowR &r, uint32_t (&NZ)[NZW]) {
    if (alive && j0 + 3 >= lo && j0 <= end) {
        const uint32_t sc4 = __builtin_amdgcn_perm(t_hi, t_lo, q);
        const int rb = beg - j0, re = end - j0;                     // the lane's band relative to this group
        unsigned kg = 0, mg = 0;
#pragma unroll
        for (int u = 0; u < 4; u++) {
            uint32_t &wd = u < 2 ? w0 : w1;
            const int sh = 16 * (u & 1);
            if (u >= rb && u < re) {
                const int e = (int)((wd >> (sh + 8)) & 0xffu);
                int M = (int)((wd >> sh) & 0xffu);
                const int sc = (int)(int8_t)(sc4 >> (8 * u));
                M = M ? M + sc : 0;
                int h = M > e ? M : e;
                h = h > r.f ? h : r.f;
                const unsigned kj = (unsigned)h << 8 | (unsigned)u;
                kg = kg > kj ? kg : kj;
                const int en = imax(isub0(e, e_del), M - oe_del);
                r.f = imax(isub0(r.f, e_ins), M - oe_ins);
                const uint32_t nw = (uint32_t)r.h1 | ((uint32_t)en << 8);
                wd = (wd & ~(0xffffu << sh)) | nw << sh;
                mg |= (nw < 1u ? nw : 1u) << u;                      // a stored cell that is not zero
                r.h1 = h;
            } else if (u == re) {                                   // eh[end] = {h1, 0}: every cell of the lane's band lies before this column
                wd = (wd & ~(0xffffu << sh)) | (uint32_t)r.h1 << sh;
                mg |= ((uint32_t)r.h1 < 1u ? (uint32_t)r.h1 : 1u) << u;
            }
        }
        // (a group none of whose cells scored leaves j0 in the key's low byte: harmless, the column of a zero maximum is never read)
        const unsigned kgj = kg + (unsigned)j0;
        r.key = r.key > kgj ? r.key : kgj;
        NZ[g >> 3] |= mg << (4 * (g & 7));
    }
}

template <int NG>
static __device__ void lane_dp8r(bool run, int qlen, int tlen, int w, int h0, RefPtr tp, int ts, const SwParams &P,
                                 const uint32_t (&Q)[NG + 1], LaneOut &out, long long &cells, long long &iters) {
    constexpr int NZW = NG / 8 + 1;                              // one bit per column 0 .. 4 NG + 3: the stored cells that are not zero
    const int o_del = P.o_del, e_del = P.e_del, o_ins = P.o_ins, e_ins = P.e_ins, oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
    const int sc_match = P.mat[0], sc_mis = P.mat[1], sc_amb = P.mat[4];
    const uint32_t rep_mis = rep4(sc_mis), rep_amb = rep4(sc_amb);
    const int e1 = h0 > oe_ins ? h0 - oe_ins : 0;                // first row, bandedSWA.cpp:143-145
    const int cls = pair_class(tlen, qlen, h0, P.max_sc);
    uint32_t W[2 * NG + 2];
    {
        int v = e1 + e_ins;                                      // column 1 holds e1, column j > 0: max(e1 - (j - 1) e_ins, 0) -- a running value, no per-column constant
#pragma unroll
        for (int k = 0; k < 2 * NG + 2; k++) {                   // (columns beyond a lane's query are never read)
            uint32_t v0, v1;
            if (k == 0) v0 = (uint32_t)h0; else { v -= e_ins; v0 = (uint32_t)imax(v, 0); }
            v -= e_ins; v1 = (uint32_t)imax(v, 0);
            W[k] = v0 | v1 << 16;
        }
    }
    int beg = 0, end = qlen, maxv = h0, max_i = -1, max_j = -1, max_ie = -1, gscore = -1, max_off = 0;
    bool alive = run && tlen > 0;
    const int maxt = __builtin_amdgcn_readlane(wave_scan_max(alive ? tlen : 0, 0), 63);
    TStream T; T.init(run, tp, ts, tlen);
    int zero_s = 0;
    asm volatile("" : "+s"(zero_s));                             // (a scalar the compiler cannot fold: the groups' first columns are zero_s + 4 g)
    for (int i = 0; i < maxt; ++i) {
        if (!__ballot(alive)) break;
        const int tb = T.next(i);
        Dp8RowR r; r.h1 = 0; r.f = 0; r.key = 0;
        uint32_t NZ[NZW];
#pragma unroll
        for (int d = 0; d < NZW; d++) NZ[d] = 0;
        if (alive) {
            if (beg < i - w) beg = i - w;
            if (end > i + w + 1) end = i + w + 1;
            if (end > qlen) end = qlen;
            if (beg == 0) { r.h1 = h0 - (o_del + e_del * (i + 1)); if (r.h1 < 0) r.h1 = 0; }
            cells += imax(end - beg, 0);
        }
        const int lo = beg < end ? beg : end;                    // the walk of a lane: its cells [beg, end) and the closing column `end`
        const int jlo = (1 << 20) - __builtin_amdgcn_readlane(wave_scan_max(alive ? (1 << 20) - lo : 0, 0), 63);
        const int jhi = __builtin_amdgcn_readlane(wave_scan_max(alive ? end + 1 : 0, 0), 63);
        const int g0 = jlo >> 2, g1 = (jhi + 3) >> 2;                // groups [g0, g1)
        iters += g1 > g0 ? 2 * (g1 - g0) : 0;
        const uint32_t t_lo = tb > 3 ? rep_amb : rep_mis ^ ((uint32_t)((sc_mis ^ sc_match) & 0xff) << (8 * tb)), t_hi = rep_amb;
#pragma unroll
        for (int g = 0; g <= NG; g++)
            if (g >= g0 && g < g1)                                   // (wave-uniform: a scalar branch)
                dp8_group_reg<NZW>(g, zero_s + 4 * g, W[2 * g], W[2 * g + 1], Q[g], alive, lo, beg, end, t_lo, t_hi, e_del, e_ins, oe_del, oe_ins, r, NZ);
        int fnz = -1, lnz = -1;                                      // first / last column whose stored cell is not zero
#pragma unroll
        for (int d = 0; d < NZW; d++) {
            if (NZ[d]) {
                if (fnz < 0) fnz = 32 * d + __builtin_ctz(NZ[d]);
                lnz = 32 * d + 31 - __builtin_clz(NZ[d]);
            }
        }
        const int m = (int)(r.key >> 8), mj = (int)(r.key & 255u), h1 = r.h1;
        if (alive) {
            const int jfin = beg < end ? end : beg;
            if (jfin == qlen) { max_ie = gscore > h1 ? max_ie : i; gscore = gscore > h1 ? gscore : h1; }
            if (m == 0) alive = false;
            else {
                const bool new_max = m > maxv;
                if (new_max) {
                    maxv = m; max_i = i; max_j = mj;
                    const int d = mj - i;
                    max_off = imax(max_off, d < 0 ? -d : d);
                }
                if (zdrop_stop(cls, new_max, maxv, m, i - max_i, mj - max_j, e_del, e_ins, P.zdrop)) alive = false;
                const int nb = fnz >= 0 ? fnz : end;
                const int jl = imax(lnz, nb - 1);
                beg = nb;
                end = jl + 2 < qlen ? jl + 2 : qlen;
                if (i + 1 >= tlen) alive = false;
            }
        }
    }
    if (run) { out.score = maxv; out.qle = max_j + 1; out.tle = max_i + 1; out.gtle = max_ie + 1; out.gscore = gscore; out.max_off = max_off; }
}

// the query of a lane as the selector words of the score permute, four bases per register (a base > 3 is 4); words beyond the query are never read
template <int NG>
static __device__ __forceinline__ void load_query_regs(bool has, const uint8_t *q, int qs, int len2, uint32_t (&Q)[NG + 1]) {
    typedef uint32_t __attribute__((aligned(1))) u32_unaligned;
#pragma unroll
    for (int g = 0; g <= NG; g++) {
        const int j0 = 4 * g;
        uint32_t wq = 0xffffffffu;                                  // (the tail of a query, or a word with an N in it -- rare --, goes the byte way)
        if (has && j0 + 3 < len2) wq = qs > 0 ? *(const u32_unaligned *)(q + j0) : __builtin_bswap32(*(const u32_unaligned *)(q - j0 - 3));
        Q[g] = wq;
    }
#pragma unroll
    for (int g = 0; g <= NG; g++) {
        const int j0 = 4 * g;
        if (has && j0 < len2 && (Q[g] & 0xfcfcfcfcu) != 0u) {
            uint32_t wq = 0;
            for (int u = 0; u < 4 && j0 + u < len2; u++) { const uint32_t qv = q[(int64_t)(j0 + u) * qs]; wq |= (qv > 3 ? 4u : qv) << (8 * u); }
            Q[g] = wq;
        }
    }
}

// (Measured in round 6 and removed, profiles/r06b_sweep_ext_prio_regrows.json: s_setprio 1..3 for the wavefronts of the long classes and / or of the wavefront
//  kernel -- a phase is as long as one tile of its long classes, so they were let issue first: extension 15.7 -> 15.4-15.5 ms at best, 16.4 with the
//  wavefront kernel raised: within the sweep's noise.)
// One seed per lane: left side, then right side (h0 = the score after the left side, bwamem.cpp:2672-2677), each with the two-try
// band rule.  A wavefront takes tiles of 64 consecutive seeds of its class's sorted list (grid-stride: the host sizes the grid from the
// previous batch's counts, the kernel reads the real range from the device).
template <bool P8, bool PF, bool PT = false, bool G4 = false>      // PT: scores by byte permute (lane_dp8); G4: columns in groups of four (lane_dp8g)
__global__ void __launch_bounds__(64)
k_ext_seeds(DevIndex ix, ExtParams xp, const int32_t *__restrict__ tasks_all, const int64_t *__restrict__ start, int bin_lo, int bin_hi, int qmax,
            const uint8_t *__restrict__ enc, const int64_t *__restrict__ off, const int32_t *__restrict__ len,
            const int64_t *__restrict__ slot_base, const int32_t *__restrict__ reg_seed, const int32_t *__restrict__ reg_chain,
            const DevChain *__restrict__ chn, const DevSeed *__restrict__ seeds, DevReg *regs, unsigned long long *counters, int rev) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_l[];
    uint32_t *EH = lds_l;                                       // [(qmax+1)][64]           (P8: [(qmax+2)/2][64])
    uint8_t *QL = (uint8_t *)(lds_l + (size_t)(qmax + 1) * 64); // [qmax][64] bytes
    uint32_t *QL8 = lds_l + (G4 ? (size_t)(2 * ((qmax + 3) / 4) + 2) : (size_t)((qmax + 2) / 2)) * 64;      // P8: [(qmax+7)/8][64] dwords, 8 bases of 4 bits each (PT: [(qmax+3)/4 + 1][64], one per byte)
    const int lane = threadIdx.x;
    const int64_t first = start[bin_lo];
    const int n_tasks = (int)(start[bin_hi] - first);
    const int32_t *tasks = tasks_all + first;
    const int n_tiles = (n_tasks + 63) >> 6;
    long long cells = 0, iters = 0;
    unsigned long long n_done = 0;
    // (rev: the list ascends in query length; the tiles with the longest queries -- the slowest wavefronts -- are taken first)
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int idx = (rev ? n_tiles - 1 - tile : tile) * 64 + lane;
        const bool valid = idx < n_tasks;
        int g = 0, l_query = 0;
        DevSeed s; DevReg a; DevChain c;
        s.qbeg = 0; s.len = 0; s.rbeg = 0; c.rmax0 = c.rmax1 = 0; c.read = 0; a.score = 0;
        if (valid) {
            g = tasks[idx];
            const int64_t base = slot_base[g];
            c = chn[base + reg_chain[g]];
            s = seeds[base + reg_seed[g]];
            a = regs[g];
            l_query = len[c.read];
        }
#pragma nounroll
        for (int side = 0; side < 2; side++) {
            const bool has = valid && (side == 0 ? s.qbeg != 0 : s.qbeg + s.len != l_query);
            if (!__ballot(has)) continue;
            const SwParams &P = side == 0 ? xp.left : xp.right;
            TaskGeom tg;
            tg.len1 = tg.len2 = 0; tg.q = enc; tg.t = ix.ref(0); tg.qs = tg.ts = 1;
            int h0 = 0, prev = -1;
            if (has) {
                tg = task_geom(side, s, c, enc + off[c.read], l_query, ix.ref(0));
                if (side == 0) h0 = s.len * xp.a;
                else { h0 = a.score; prev = a.score; }
            }
            // stage the query bases
            const int maxq = __builtin_amdgcn_readlane(wave_scan_max(has ? tg.len2 : 0, 0), 63);
            if (PT) {                                               // one base per byte, 4 to a dword: the selector words of the byte permute
                typedef uint32_t __attribute__((aligned(1))) u32_unaligned;
                for (int jb = 0; jb < maxq; jb += 16) {             // four words requested before the first is looked at: one wait per 16 bases
                    uint32_t wq4[4];
#pragma unroll
                    for (int v = 0; v < 4; v++) {
                        const int j0 = jb + 4 * v;
                        wq4[v] = 0xffffffffu;                       // (the tail of a query, or a word with an N in it -- rare --, goes the byte way)
                        if (has && j0 + 3 < tg.len2)
                            wq4[v] = tg.qs > 0 ? *(const u32_unaligned *)(tg.q + j0) : __builtin_bswap32(*(const u32_unaligned *)(tg.q - j0 - 3));
                    }
#pragma unroll
                    for (int v = 0; v < 4; v++) {
                        const int j0 = jb + 4 * v;
                        if (has && j0 < tg.len2) {
                            uint32_t wq = wq4[v];
                            if ((wq & 0xfcfcfcfcu) != 0u) {
                                wq = 0;
                                for (int u = 0; u < 4 && j0 + u < tg.len2; u++) { const uint32_t qv = tg.q[(int64_t)(j0 + u) * tg.qs]; wq |= (qv > 3 ? 4u : qv) << (8 * u); }
                            }
                            QL8[(j0 >> 2) * 64 + lane] = wq;
                        }
                    }
                }
            } else if (!P8) {
                for (int j = 0; j < maxq; j++) if (has && j < tg.len2) QL[j * 64 + lane] = tg.q[(int64_t)j * tg.qs];
            } else {
                for (int j0 = 0; j0 < maxq; j0 += 8) {
                    if (has && j0 < tg.len2) {
                        uint32_t wq = 0;
                        for (int u = 0; u < 8 && j0 + u < tg.len2; u++) wq |= (uint32_t)(tg.q[(int64_t)(j0 + u) * tg.qs] & 15) << (4 * u);
                        QL8[(j0 >> 3) * 64 + lane] = wq;
                    }
                }
            }
            const int cls = pair_class(tg.len1, tg.len2, h0, P.max_sc);
            LaneOut o; o.score = h0; o.qle = o.tle = o.gtle = 0; o.gscore = -1; o.max_off = 0;
            bool run = has;
            int w_used = xp.w;
            for (int t = 0; t < MAX_BAND_TRY; t++) {                // two-try band rule, bwamem.cpp:2495-2496
                if (!__ballot(run)) break;
                const int w = xp.w << t;
                const int wc = band_clamp(w, tg.len2, P, cls);
                if (G4) lane_dp8g(run, tg.len2, tg.len1, wc, h0, tg.t, tg.ts, P, EH, QL8, lane, o, cells, iters);
                else if (P8) lane_dp8<PF, PT>(run, tg.len2, tg.len1, wc, h0, tg.t, tg.ts, P, EH, QL8, lane, o, cells, iters);
                else lane_dp(run, tg.len2, tg.len1, wc, h0, tg.t, tg.ts, P, EH, QL, lane, o, cells, iters);
                if (run) {
                    w_used = w;
                    if (o.score == prev || o.max_off < (w >> 1) + (w >> 2) || t + 1 == MAX_BAND_TRY) run = false;
                    else prev = o.score;
                }
            }
            if (has) {
                SwOut so; so.score = o.score; so.qle = o.qle; so.tle = o.tle; so.gtle = o.gtle; so.gscore = o.gscore; so.max_off = o.max_off;
                apply_side(side, a, s, l_query, so, h0, w_used, side == 0 ? xp.pen_clip5 : xp.pen_clip3);
                n_done++;
            }
        }
        if (valid) { a.seedcov = seed_cover(c, seeds, a); regs[g] = a; }
    }
    // (a wavefront that found no tile -- the first batch of a process sizes its grids without the previous batch's statistics: ten times the
    //  wavefronts -- leaves without queueing three atomics on one cache line)
    if (!__ballot(n_done != 0 || cells != 0)) return;
    atomicAdd(&counters[0], (unsigned long long)cells);
    atomicAdd(&counters[1], n_done);
    if (lane == 0) atomicAdd(&counters[2], (unsigned long long)iters);      // column-pair trips of this wavefront: 128 lane slots each (lane use = cells / that)
}

// One seed per lane with the rows in REGISTERS (lane_dp8r).  The row and the query of a 128-column class are 99 registers and a wavefront's share of the
// register file is what bounds the wavefronts per SIMD here, so NOTHING else stays in registers across the DP: the seed, its chain and its reg are read where
// a side needs them and read again (L1 / L2 hits) when the side is applied -- through an index the compiler cannot see through, or it would keep the first
// copies alive -- and the reg goes back to memory after every side.  With that state held across the DP (the LDS kernel's shape) the 128-column
// instantiation took 358 registers, one wavefront per SIMD, and the stage was slower than with LDS rows (profiles/r06b_sweep_ext_prio_regrows.json).
static __device__ __forceinline__ int opaque(int v) { asm volatile("" : "+v"(v)); return v; }
template <int RG>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3)))
k_ext_seeds_reg(DevIndex ix, ExtParams xp, const int32_t *__restrict__ tasks_all, const int64_t *__restrict__ start, int bin_lo, int bin_hi,
                const uint8_t *__restrict__ enc, const int64_t *__restrict__ off, const int32_t *__restrict__ len,
                const int64_t *__restrict__ slot_base, const int32_t *__restrict__ reg_seed, const int32_t *__restrict__ reg_chain,
                const DevChain *__restrict__ chn, const DevSeed *__restrict__ seeds, DevReg *regs, unsigned long long *counters, int rev) {
    const int lane = threadIdx.x;
    const int64_t first = start[bin_lo];
    const int n_tasks = (int)(start[bin_hi] - first);
    const int32_t *tasks = tasks_all + first;
    const int n_tiles = (n_tasks + 63) >> 6;
    long long cells = 0, iters = 0;
    unsigned long long n_done = 0;
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int idx = (rev ? n_tiles - 1 - tile : tile) * 64 + lane;
        const bool valid = idx < n_tasks;
        const int g = valid ? tasks[idx] : 0;
#pragma nounroll
        for (int side = 0; side < 2; side++) {
            bool has = false;
            TaskGeom tg;
            tg.len1 = tg.len2 = 0; tg.q = enc; tg.t = ix.ref(0); tg.qs = tg.ts = 1;
            int h0 = 0, prev = -1;
            if (valid) {
                const int ga = opaque(g);
                const int64_t base = slot_base[ga];
                const DevChain c = chn[base + reg_chain[ga]];
                const DevSeed s = seeds[base + reg_seed[ga]];
                const int l_query = len[c.read];
                has = side == 0 ? s.qbeg != 0 : s.qbeg + s.len != l_query;
                if (has) {
                    tg = task_geom(side, s, c, enc + off[c.read], l_query, ix.ref(0));
                    if (side == 0) h0 = s.len * xp.a;
                    else { h0 = regs[ga].score; prev = h0; }
                }
            }
            if (!__ballot(has)) continue;
            const SwParams &P = side == 0 ? xp.left : xp.right;
            uint32_t QR[RG + 1];
            load_query_regs<RG>(has, tg.q, tg.qs, tg.len2, QR);
            const int cls = pair_class(tg.len1, tg.len2, h0, P.max_sc);
            LaneOut o; o.score = h0; o.qle = o.tle = o.gtle = 0; o.gscore = -1; o.max_off = 0;
            bool run = has;
            int w_used = xp.w;
#pragma nounroll
            for (int t = 0; t < MAX_BAND_TRY; t++) {                // two-try band rule, bwamem.cpp:2495-2496
                if (!__ballot(run)) break;
                const int w = xp.w << t;
                const int wc = band_clamp(w, tg.len2, P, cls);
                lane_dp8r<RG>(run, tg.len2, tg.len1, wc, opaque(h0), tg.t, tg.ts, P, QR, o, cells, iters);      // (opaque: or the first row's 2 RG + 2 words are computed once and KEPT for the second try)
                if (run) {
                    w_used = w;
                    if (o.score == prev || o.max_off < (w >> 1) + (w >> 2) || t + 1 == MAX_BAND_TRY) run = false;
                    else prev = o.score;
                }
            }
            if (has) {
                const int gb = opaque(g);
                const int64_t base = slot_base[gb];
                const DevChain c = chn[base + reg_chain[gb]];
                const DevSeed s = seeds[base + reg_seed[gb]];
                DevReg a = regs[gb];
                SwOut so; so.score = o.score; so.qle = o.qle; so.tle = o.tle; so.gtle = o.gtle; so.gscore = o.gscore; so.max_off = o.max_off;
                apply_side(side, a, s, len[c.read], so, h0, w_used, side == 0 ? xp.pen_clip5 : xp.pen_clip3);
                regs[gb] = a;
                n_done++;
            }
        }
        if (valid) {
            const int gc = opaque(g);
            const DevChain c = chn[slot_base[gc] + reg_chain[gc]];
            const DevReg a = regs[gc];
            regs[gc].seedcov = seed_cover(c, seeds, a);
        }
    }
    if (!__ballot(n_done != 0 || cells != 0)) return;
    atomicAdd(&counters[0], (unsigned long long)cells);
    atomicAdd(&counters[1], n_done);
    if (lane == 0) atomicAdd(&counters[2], (unsigned long long)iters);
}

// one side on one wavefront, with the accept/retry rule of bwamem.cpp:2495-2496: stop when the score did not change, or
// the best cell stayed within 3/4 of the band, or this was the last try.
static __device__ __forceinline__ int extend_side(const uint8_t *q, int qs, int len2, RefPtr t, int ts, int len1, int h0,
                                                  int prev, int w0, const SwParams &P, int *RH, int *RE, int RM, SwOut &o,
                                                  int &w_used, long long &cells) {
    const int cls = pair_class(len1, len2, h0, P.max_sc);
    for (int i = 0; i < MAX_BAND_TRY; i++) {
        const int w = w0 << i;
        const int wc = band_clamp(w, len2, P, cls);
        cells += bsw_extend(q, qs, len2, t, ts, len1, wc, h0, P, RH, RE, RM, o);
        w_used = w;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // (the next try, or the other side, reuses the LDS rings)
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (o.score == prev || o.max_off < (w >> 1) + (w >> 2) || i + 1 == MAX_BAND_TRY) break;
        prev = o.score;
    }
    return o.score;
}

// ---- one seed per wavefront: long queries, int32-class scores (the scalar fallback of the reference, bwamem.cpp:2472), and the query-length
// classes the launch policy sends here; left side, then right side
__global__ void __launch_bounds__(256)
k_ext_wave(DevIndex ix, ExtParams xp, const int32_t *__restrict__ tasks_all, const int64_t *__restrict__ start, int bin_lo, int bin_hi,
           const uint8_t *__restrict__ enc, const int64_t *__restrict__ off, const int32_t *__restrict__ len,
           const int64_t *__restrict__ slot_base, const int32_t *__restrict__ reg_seed, const int32_t *__restrict__ reg_chain,
           const DevChain *__restrict__ chn, const DevSeed *__restrict__ seeds, DevReg *regs, int R, unsigned long long *counters, int rev) {
    extern __shared__ __attribute__((aligned(16))) int lds[];
    ExtParams *sP = (ExtParams *)lds;
    int *rings = lds + (sizeof(ExtParams) + 3) / 4;
    if (threadIdx.x < sizeof(ExtParams) / 4) ((int *)sP)[threadIdx.x] = ((const int *)&xp)[threadIdx.x];
    __syncthreads();
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
    int *RH = rings + (size_t)wv * 2 * R, *RE = RH + R;
    const int64_t first = start[bin_lo];
    const int n_tasks = (int)(start[bin_hi] - first);
    const int32_t *tasks = tasks_all + first;
    long long cells = 0;
    unsigned long long n_done = 0;
    for (int idx = blockIdx.x * wpb + wv; idx < n_tasks; idx += gridDim.x * wpb) {
        const int g = uni(tasks[rev ? n_tasks - 1 - idx : idx]);
        const int64_t base = slot_base[g];
        const DevChain c = chn[base + reg_chain[g]];
        const DevSeed s = seeds[base + reg_seed[g]];
        DevReg a = regs[g];
        const int l_query = len[c.read];
#pragma nounroll
        for (int side = 0; side < 2; side++) {
            if (!(side == 0 ? s.qbeg != 0 : s.qbeg + s.len != l_query)) continue;
            const TaskGeom tg = task_geom(side, s, c, enc + off[c.read], l_query, ix.ref(0));
            const int h0 = side == 0 ? s.len * sP->a : a.score;
            const int prev = side == 0 ? -1 : a.score;
            SwOut o; int w_used;
            extend_side(tg.q, tg.qs, tg.len2, tg.t, tg.ts, tg.len1, h0, prev, sP->w, side == 0 ? sP->left : sP->right, RH, RE, R - 1, o, w_used, cells);
            apply_side(side, a, s, l_query, o, h0, w_used, side == 0 ? sP->pen_clip5 : sP->pen_clip3);     // (wave-uniform: every lane holds the same reg)
            n_done++;
        }
        a.seedcov = seed_cover(c, seeds, a);                     // (wave-uniform, like the reg itself)
        if (lane == 0) regs[g] = a;
    }
    if (lane == 0 && (n_done != 0 || cells != 0)) {
        atomicAdd(&counters[0], (unsigned long long)cells);
        atomicAdd(&counters[1], n_done);
        atomicAdd(&counters[3], (unsigned long long)cells);      // the wavefront kernel's share of the cells
    }
}

// cal_max_gap, bwamem.cpp:66-76
static __device__ __forceinline__ int cal_max_gap2(const ChainParams &o, int qlen) {
    // (int)((double)(qlen*a - o)/e + 1.) == (qlen*a - o + e) / e in C integer division (both truncate toward zero, e > 0);
    // the double division of the reference would cost ~100 instructions per call on the GPU
    const int nd = qlen * o.a - o.o_del + o.e_del, ni = qlen * o.a - o.o_ins + o.e_ins;
    const int l_del = o.e_del == 1 ? nd : nd / o.e_del;
    const int l_ins = o.e_ins == 1 ? ni : ni / o.e_ins;
    int l = l_del > l_ins ? l_del : l_ins;
    l = l > 1 ? l : 1;
    return l < o.w << 1 ? l : o.w << 1;
}

// The per-seed test of the redundant-seed filter, bwamem.cpp:2922-2983: true = the seed's alignment is purged (the
// original bwa-mem would not have extended it).  `lim` = number of kept regs of the read before this seed.
static __device__ bool seed_redundant(const ChainParams &o, int l_query, const DevReg *av, int nr, int lim, const DevChain &c,
                                      const DevSeed *cs, const int32_t *srt2, int k) {
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
            if (srt2[v] < 0) continue;                      // UINT_MAX marker of the reference
            const DevSeed t = cs[srt2[v]];
            if (t.len < s.len * .95) continue;
            if (s.qbeg <= t.qbeg && s.qbeg + s.len - t.qbeg >= s.len >> 2 && t.qbeg - s.qbeg != t.rbeg - s.rbeg) break;
            if (t.qbeg <= s.qbeg && t.qbeg + t.len - s.qbeg >= s.len >> 2 && s.qbeg - t.qbeg != s.rbeg - t.rbeg) break;
        }
        if (v == c.n) return true;
    }
    return false;
}

// The same test for the lazy rounds, over the LIST of the read's kept regs instead of all its regs: in a lazy round every reg before the cursor is
// either purged (never extended) or was picked and extended -- the kept ones are exactly the picks, in order -- so `lim` is the list's length and
// the walk visits av[kept[0]], av[kept[1]], ... where the walk above skips over the purged regs between them one by one (quadratic in the regs of a
// long read's chain: k_advance took 38 ms per 10 000 reads of 10 kb).
static __device__ bool seed_redundant_kept(const ChainParams &o, int l_query, const DevReg *av, const int32_t *kept, int lim, const DevChain &c,
                                           const DevSeed *cs, const int32_t *srt2, int k) {
    const DevSeed s = cs[srt2[k]];
    int v = 0;
    for (; v < lim; ++v) {
        const DevReg p = av[kept[v]];
        int64_t rd; int qd, w, max_gap;
        if (s.rbeg < p.rb || s.rbeg + s.len > p.re || s.qbeg < p.qb || s.qbeg + s.len > p.qe) continue;
        if (s.len - p.seedlen0 > .1 * l_query) continue;
        qd = s.qbeg - p.qb; rd = s.rbeg - p.rb;
        max_gap = cal_max_gap2(o, qd < rd ? qd : (int)rd);
        w = max_gap < p.w ? max_gap : p.w;
        if (qd - rd < w && rd - qd < w) break;
        qd = p.qe - (s.qbeg + s.len); rd = p.re - (s.rbeg + s.len);
        max_gap = cal_max_gap2(o, qd < rd ? qd : (int)rd);
        w = max_gap < p.w ? max_gap : p.w;
        if (qd - rd < w && rd - qd < w) break;
    }
    if (v < lim) {
        for (v = k + 1; v < c.n; ++v) {
            if (srt2[v] < 0) continue;                      // UINT_MAX marker of the reference
            const DevSeed t = cs[srt2[v]];
            if (t.len < s.len * .95) continue;
            if (s.qbeg <= t.qbeg && s.qbeg + s.len - t.qbeg >= s.len >> 2 && t.qbeg - s.qbeg != t.rbeg - s.rbeg) break;
            if (t.qbeg <= s.qbeg && t.qbeg + t.len - s.qbeg >= s.len >> 2 && s.qbeg - t.qbeg != s.rbeg - t.rbeg) break;
        }
        if (v == c.n) return true;
    }
    return false;
}

#define PF_HEAVY 24                 // reads with more regs than this are purged by a whole wavefront (k_postfilter_heavy); knob BM2_PF_HEAVY
#define PF_RCACHE 8                 // k_postfilter_heavy: regs per lane held in registers (the first 512 regs of a read)

// Redundant-seed post-filter, bwamem.cpp:2895-2989 (one read per lane): replays the original bwa-mem rule "skip a seed
// already contained in an earlier alignment unless an overlapping seed lies on another diagonal" and purges those regs.
// Seeds already decided by the lazy rounds (k_decide_round) get the same verdict again, so replaying them is harmless.
__global__ void __launch_bounds__(128)
k_postfilter(ChainParams o, int n_reads, const int32_t *__restrict__ len, const int64_t *__restrict__ read_base,
             const int32_t *__restrict__ n_chain, const int32_t *__restrict__ n_reg, const DevChain *__restrict__ chn,
             const DevSeed *__restrict__ seeds, int32_t *srt_all, DevReg *regs, int32_t *n_out, const int32_t *__restrict__ cursor,
             const int32_t *__restrict__ perm, int pf_heavy) {
    const int tix = blockIdx.x * blockDim.x + threadIdx.x;
    if (tix >= n_reads) return;
    const int r = perm[tix];
    const int nc = n_chain[r], nr = n_reg[r];
    if (nc == 0) { n_out[r] = 0; return; }
    if (nr > pf_heavy) return;                               // k_postfilter_heavy takes these: one read per wavefront
    const int first_idx = cursor[r];
    const int64_t base = read_base[r];
    const int l_query = len[r];
    DevReg *av = regs + base;
    int lim = 0;
    for (int i = 0; i < first_idx && i < nr; i++) if (!(av[i].qb == -1 && av[i].qe == -1)) lim++;
    if (nr > first_idx) {
        for (int j = 0; j < nc; j++) {
            const DevChain c = chn[base + j];
            if (c.reg0 + c.n <= first_idx) continue;        // every seed of this chain was decided lazily
            const DevSeed *cs = seeds + c.seed_off;
            int32_t *srt2 = srt_all + c.seed_off;
            for (int k = c.n - 1; k >= 0; k--) {
                const int idx = c.reg0 + (c.n - 1 - k);     // reg index of this seed inside the read
                if (idx < first_idx) continue;
                if (seed_redundant(o, l_query, av, nr, lim, c, cs, srt2, k)) {
                    av[idx].qb = -1; av[idx].qe = -1;
                    srt2[k] = -1;
                    continue;
                }
                lim++;
            }
        }
    }
    int m = 0;
    for (int i = 0; i < nr; i++) if (av[i].qe > av[i].qb) m++;       // bwamem.cpp:1141-1152
    n_out[r] = m;
}

// The same filter for reads with many regs, one read per wavefront.  The test of one seed walks the read's regs in order
// (O(regs) per seed, O(regs^2) per read: a repeat read with 400 regs kept one lane busy for the whole 14 ms of the
// lane-per-read kernel while the other million reads took 2 ms).  The walk only counts and looks for the first reg that
// stops it, so 64 regs are judged at once and the order is restored with ballots: `v` = non-purged regs seen so far; a reg
// is looked at only while v < lim; the first looked-at reg that satisfies one of the two band tests ends the walk.
__global__ void __launch_bounds__(256)
k_postfilter_heavy(ChainParams o, const int32_t *__restrict__ heavy /* read ids, heavy ones first */, const int64_t *__restrict__ n_heavy_p,
                   const int32_t *__restrict__ len, const int64_t *__restrict__ read_base, const int32_t *__restrict__ n_chain,
                   const int32_t *__restrict__ n_reg, const DevChain *__restrict__ chn, const DevSeed *__restrict__ seeds,
                   int32_t *srt_all, DevReg *regs, int32_t *n_out, const int32_t *__restrict__ cursor, unsigned long long *item_cur,
                   int pf_heavy) {
    const int lane = threadIdx.x & 63;
    const unsigned long long lt_mask = lane ? (~0ULL >> (64 - lane)) : 0ULL;
    const int64_t n_heavy = *n_heavy_p;
    for (;;) {
        // one item per wavefront: every lane takes part in the atomic (lane 0 adds 1, the others 0), so there is no divergent
        // branch around it and the compiler cannot split lane 0 from the rest (see notes/NEXT.md)
        const unsigned long long it = atomicAdd(item_cur, lane == 0 ? 1ULL : 0ULL);
        const int64_t hid = (int64_t)(((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(it >> 32)) << 32) |
                                      (unsigned)__builtin_amdgcn_readfirstlane((unsigned)it));
        if (hid >= n_heavy) break;
        const int r = heavy[hid];
        const int nc = n_chain[r], nr = n_reg[r];
        if (nc != 0 && nr > pf_heavy) {                          // (no `continue` in this loop: see notes/NEXT.md)
        const int first_idx = cursor[r];
        const int64_t base = read_base[r];
        const int l_query = len[r];
        DevReg *av = regs + base;
        int lim = 0;
        for (int i0 = 0; i0 < first_idx && i0 < nr; i0 += 64) {
            const int i = i0 + lane;
            const bool kept = i < first_idx && i < nr && !(av[i].qb == -1 && av[i].qe == -1);
            lim += __popcll(__ballot(kept));
        }
        // The walk of every seed reads the same six fields of the read's regs: the first PF_RCACHE * 64 regs live in REGISTERS (lane l holds regs
        // l, l + 64, ...; a purge updates the owner's copy), so a walk is compares and ballots -- the heaviest read of a chunk (400+ regs, its seeds
        // judged one after the other) set the length of this kernel with 7 dependent loads per seed (3.3 ms of a 67 ms step).
        int64_t c_rb[PF_RCACHE], c_re[PF_RCACHE]; int32_t c_qb[PF_RCACHE], c_qe[PF_RCACHE], c_w[PF_RCACHE], c_sl[PF_RCACHE];
        _Pragma("unroll") for (int cc = 0; cc < PF_RCACHE; cc++) {
            const int i = cc * 64 + lane;
            c_rb[cc] = c_re[cc] = 0; c_qb[cc] = c_qe[cc] = -1; c_w[cc] = c_sl[cc] = 0;
            if (i < nr) { const DevReg p = av[i]; c_rb[cc] = p.rb; c_re[cc] = p.re; c_qb[cc] = p.qb; c_qe[cc] = p.qe; c_w[cc] = p.w; c_sl[cc] = p.seedlen0; }
        }
        if (nr > first_idx) {
            for (int j = 0; j < nc; j++) {
                const DevChain c = chn[base + j];
                if (c.reg0 + c.n <= first_idx) continue;
                const DevSeed *cs = seeds + c.seed_off;
                int32_t *srt2 = srt_all + c.seed_off;
                for (int k = c.n - 1; k >= 0; k--) {
                    const int idx = c.reg0 + (c.n - 1 - k);
                    if (idx < first_idx) continue;
                    // ---- seed_redundant, bwamem.cpp:2922-2983, 64 regs at a time
                    const DevSeed s = cs[srt2[k]];
                    int v = 0; bool stopped = false;
                    // one reg against the seed: is it live, and does it end the walk (one of the two band tests)
                    auto judge = [&](bool in_range, int64_t p_rb, int64_t p_re, int p_qb, int p_qe, int p_w, int p_sl, bool &live, bool &brk) {
                        live = in_range && !(p_qb == -1 && p_qe == -1); brk = false;
                        if (live && !(s.rbeg < p_rb || s.rbeg + s.len > p_re || s.qbeg < p_qb || s.qbeg + s.len > p_qe) &&
                            !(s.len - p_sl > .1 * l_query)) {
                            int qd = s.qbeg - p_qb; int64_t rd = s.rbeg - p_rb;
                            int max_gap = cal_max_gap2(o, qd < rd ? qd : (int)rd);
                            int w = max_gap < p_w ? max_gap : p_w;
                            if (qd - rd < w && rd - qd < w) brk = true;
                            else {
                                qd = p_qe - (s.qbeg + s.len); rd = p_re - (s.rbeg + s.len);
                                max_gap = cal_max_gap2(o, qd < rd ? qd : (int)rd);
                                w = max_gap < p_w ? max_gap : p_w;
                                if (qd - rd < w && rd - qd < w) brk = true;
                            }
                        }
                    };
                    auto tally = [&](bool live, bool brk) {     // the order of the 64 regs restored with ballots
                        const unsigned long long lm = __ballot(live);
                        const bool looked = live && v + __popcll(lm & lt_mask) < lim;     // the walk reaches this reg with v < lim
                        const unsigned long long bm = __ballot(looked && brk);
                        if (bm) {
                            const int ib = __ffsll((long long)bm) - 1;
                            v += __popcll(lm & (ib ? (~0ULL >> (64 - ib)) : 0ULL));          // regs counted before the stopping one
                            stopped = true;
                        } else v += __popcll(lm);                                            // (>= lim ends the walk: v is only compared with lim)
                    };
                    _Pragma("unroll") for (int cc = 0; cc < PF_RCACHE; cc++) {              // the regs held in registers
                        if (cc * 64 < nr && v < lim && !stopped) {
                            bool live, brk;
                            judge(cc * 64 + lane < nr, c_rb[cc], c_re[cc], c_qb[cc], c_qe[cc], c_w[cc], c_sl[cc], live, brk);
                            tally(live, brk);
                        }
                    }
                    for (int i0 = PF_RCACHE * 64; i0 < nr && v < lim && !stopped; i0 += 64) {   // a read with more regs than that: the rest from memory
                        const int i = i0 + lane;
                        bool live = false, brk = false;
                        if (i < nr) {
                            const DevReg p = av[i];
                            judge(true, p.rb, p.re, p.qb, p.qe, p.w, p.seedlen0, live, brk);
                        }
                        const unsigned long long lm = __ballot(live);
                        const bool looked = live && v + __popcll(lm & lt_mask) < lim;     // the walk reaches this reg with v < lim
                        const unsigned long long bm = __ballot(looked && brk);
                        if (bm) {
                            const int ib = __ffsll((long long)bm) - 1;
                            v += __popcll(lm & (ib ? (~0ULL >> (64 - ib)) : 0ULL));          // regs counted before the stopping one
                            stopped = true;
                        } else v += __popcll(lm);                                            // (>= lim ends the walk: v is only compared with lim)
                    }
                    bool red = false;
                    if (stopped || v < lim) {                 // the walk ended early: purge unless an overlapping seed lies on another diagonal
                        bool other = false;
                        for (int v0 = k + 1; v0 < c.n && !other; v0 += 64) {
                            const int vv = v0 + lane;
                            bool b2 = false;
                            if (vv < c.n && srt2[vv] >= 0) {
                                const DevSeed t = cs[srt2[vv]];
                                if (!(t.len < s.len * .95)) {
                                    if (s.qbeg <= t.qbeg && s.qbeg + s.len - t.qbeg >= s.len >> 2 && t.qbeg - s.qbeg != t.rbeg - s.rbeg) b2 = true;
                                    if (t.qbeg <= s.qbeg && t.qbeg + t.len - s.qbeg >= s.len >> 2 && s.qbeg - t.qbeg != s.rbeg - t.rbeg) b2 = true;
                                }
                            }
                            other = __ballot(b2) != 0;
                        }
                        red = !other;
                    }
                    if (red) {
                        if (lane == 0) { av[idx].qb = -1; av[idx].qe = -1; srt2[k] = -1; }
                        _Pragma("unroll") for (int cc = 0; cc < PF_RCACHE; cc++) if (idx == cc * 64 + lane) { c_qb[cc] = -1; c_qe[cc] = -1; }
                        // the next seed's walk re-reads av[] / srt2[] through other lanes OF THIS WAVEFRONT: wavefront scope (the lanes share the CU's L1;
                        // at agent scope every purged reg cost an L2 write-back and an L1 invalidate, ~3 us of a read whose walk is sequential anyway)
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    } else lim++;
                }
            }
        }
        int m = 0;
        for (int i0 = 0; i0 < nr; i0 += 64) {                // bwamem.cpp:1141-1152
            const int i = i0 + lane;
            m += __popcll(__ballot(i < nr && av[i].qe > av[i].qb));
        }
        if (lane == 0) n_out[r] = m;
        }
    }
}

// Lazy rounds: BEFORE extending, decide whether the next seeds of each read are redundant given the regs kept so far (the
// order the original bwa-mem extends in).  Each round advances every read over its consecutive redundant seeds (purged,
// never extended) to the next seed that must be extended; that seed is initialised and binned.  Equivalent to
// extend-everything-then-purge (bwamem.cpp:2895-2989 exists precisely to reproduce this order), but the purged seeds --
// about two thirds of all seeds on 150 bp reads -- cost nothing.  cursor[r] = index of the next undecided reg of read r.
__global__ void __launch_bounds__(128)
k_advance(ChainParams o, ExtParams xp, int n_reads, const int32_t *__restrict__ len,
          const int64_t *__restrict__ read_base, const int32_t *__restrict__ n_reg,
          const int32_t *__restrict__ reg_chain, const DevChain *__restrict__ chn,
          const DevSeed *__restrict__ seeds, int32_t *srt_all, DevReg *regs, int32_t *cursor, int32_t *cur_slot /* [n_reads] reg idx to extend or -1 */,
          uint32_t *ebin /* [n_reads] sort key of the picked seed */, int32_t *hist /* [N_EBINS] */, uint32_t *pend /* reads with undecided seeds left */,
          int32_t *kept /* [n_slots]: per read the reg indices picked so far, in order */, int32_t *n_kept /* [n_reads] */) {
    __shared__ uint32_t sh_pend;
    if (threadIdx.x == 0) sh_pend = 0;
    __syncthreads();
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n_reads) {
        int pick = -1;
        uint32_t b = EBIN_NONE;
        const int nr = n_reg[r];
        int cur = cursor[r];
        if (cur < nr) {
            const int64_t base = read_base[r];
            const int l_query = len[r];
            DevReg *av = regs + base;
            const int lim = n_kept[r];                       // (= the regs before the cursor that are not purged: the picks of the rounds before)
            while (cur < nr) {
                const int ci = reg_chain[base + cur];
                const DevChain c = chn[base + ci];
                const DevSeed *cs = seeds + c.seed_off;
                int32_t *srt2 = srt_all + c.seed_off;
                const int k = c.n - 1 - (cur - c.reg0);
                const DevSeed s = cs[srt2[k]];
                DevReg a;
                a.w = xp.w; a.rid = c.rid; a.frac_rep = c.frac_rep; a.seedlen0 = s.len; a.chain = ci; a.seedcov = 0;
                a.rb = s.rbeg; a.re = s.rbeg + s.len; a.score = a.truesc = -1; a.qb = a.qe = -1;
                if (lim > 0 && seed_redundant_kept(o, l_query, av, kept + base, lim, c, cs, srt2, k)) {
                    srt2[k] = -1;                            // purged: never extended
                    av[cur] = a;
                    cur++;
                    continue;
                }
                if (s.qbeg) { a.qb = s.qbeg; }
                else { a.score = a.truesc = s.len * xp.a; a.qb = 0; }
                a.qe = (s.qbeg + s.len != l_query) ? s.qbeg + s.len : l_query;
                b = seed_bin(xp, s, c, l_query);
                if (b != EBIN_NONE) atomicAdd(&hist[b], 1);
                else a.seedcov = seed_cover(c, seeds, a);        // nothing to extend: the reg is final
                av[cur] = a;
                pick = cur;
                kept[base + lim] = cur; n_kept[r] = lim + 1;
                cur++;
                break;
            }
            cursor[r] = cur;
            if (cur < nr) atomicAdd(&sh_pend, 1u);           // this read still has undecided seeds
        }
        cur_slot[r] = pick;
        ebin[r] = b;
    }
    __syncthreads();
    if (threadIdx.x == 0 && sh_pend) atomicAdd(pend, sh_pend);
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

#define LAZY_ROUNDS 6
#define EXT_EAGER_PHASE (BM2_EXT_PHASES - 1)      // the eager remainder's row of the per-phase statistics
#define BSW_STAT_ROW (BM2_EXT_PHASES - 2)         // the row of bm2_ctx::ext_stat the S1 batches keep their class counts in (no lazy round ever has it: n_lazy <= BSW_STAT_ROW)

struct ExtLaunch {
    bm2_ctx *c; hipStream_t s; ExtParams xp; const uint8_t *enc; const int64_t *off; const int32_t *len; const int64_t *slot_base;
    const int32_t *reg_seed, *reg_chain; const DevChain *chn; const DevSeed *seeds; DevReg *regs; unsigned long long *counters;
    int R; size_t lds_w; bool pack8;
    // launch policy (bm2_knob): which kernel takes a query-length class.  A lane-per-seed wavefront of long queries walks tens of thousands
    // of cells one after the other (milliseconds) and a phase lasts as long as its slowest wavefront, so the classes from wave_qmin up can go
    // one seed per WAVEFRONT (k_ext_wave) beside the lane kernels.
    int wave_qmin, rev, perm_scores, qmap;
    int reg_qmin;                            // classes of queries up to at least this many bases keep their rows in registers (0: none)
    // the sorted seed list of the phase and where it lives
    const int32_t *tasks; const int64_t *start;
};

// The launches of one phase: a lane-per-seed launch per LDS class (each class needs a different LDS footprint) plus the wavefront-per-seed
// kernel; they run concurrently on the context's side streams, joined by events.  Every kernel reads its range of the sorted list from the
// device (`start`); the host only chooses grid sizes, from `hint` = the class counts of this phase in the previous batch (nullptr: unknown,
// `ub` seeds at most) -- a grid that is too small costs time (the kernels stride), never correctness.
static int run_phase(const ExtLaunch &L, const uint32_t *hint, int64_t ub) {
    static const int cls_hi[N_CLS] = { 16, 32, 48, 64, 80, 96, 112, 128, 144, 160 };
    bm2_ctx *c = L.c;
    (void)hipEventRecord(c->ev_fork, L.s);
    // the classes from k_wave up (long queries) and the fallback bins are adjacent in the list: one wavefront-per-seed launch
    int k_wave = N_CLS;
    while (k_wave > 0 && (k_wave >= 2 ? cls_hi[k_wave - 2] : 0) + 1 >= L.wave_qmin) k_wave--;
    // (BM2_EXT_WAVE_BUDGET -- the bound moved per phase by the previous batch's class counts -- was measured in round 5 and bought nothing:
    //  16.3-16.4 ms against 16.1 at 30 000 / 60 000 seeds, 21-24 ms beyond, profiles/r05a_sweep.json; removed)
    auto grid_for = [&](int k_lo, int k_hi, int per_block) -> unsigned {
        int64_t n = 0;
        if (hint) { for (int k = k_lo; k < k_hi; k++) n += hint[k]; n += n / 4 + 64; }
        else n = ub;
        if (n > ub) n = ub;
        int64_t g = (n + per_block - 1) / per_block;
        // a floor: the hint is the PREVIOUS batch's count, and a class that was empty then (another read length after a file boundary, trimmed
        // reads) would get one workgroup to stride through whatever the class holds now -- the phase's critical path.  A workgroup that finds
        // no tile leaves after two loads.
        const int64_t floor_g = per_block == 64 ? c->n_cu : c->n_cu / 4, ub_g = (ub + per_block - 1) / per_block;
        if (g < floor_g) g = floor_g < ub_g ? floor_g : ub_g;
        if (g < 1) g = 1;
        if (g > (1 << 16)) g = 1 << 16;                 // (the kernels stride)
        return (unsigned)g;
    };
    // Streams sit on HARDWARE QUEUES round-robin (GPU_MAX_HW_QUEUES = 8: the context's main stream and its first seven side streams are
    // eight different queues, side stream 10 shares the queue of side stream 3), and two launches on one queue run one after the other
    // (profiles/r03x_timeline.tsv).  The eight launches of a phase -- seven lane classes and the wavefront kernel for 150 bp reads -- get
    // seven queues: the wavefront kernel takes the shortest class's side stream, the shortest class queues behind the second shortest.
    // BM2_EXT_QUEUE_MAP=0: one side stream per class index.
    int joined[N_CLS + 1], n_joined = 0;
    for (int kk = 0; kk <= N_CLS; kk++) {                  // longest queries first: their tails overlap the short classes
        const int k = kk == 0 ? N_CLS : N_CLS - kk;
        hipStream_t sk = !L.qmap ? c->side_stream[k] : k == N_CLS ? c->side_stream[0] : c->side_stream[k ? k : 1];
        if (k < N_CLS && k >= k_wave) continue;            // part of the launch of k == N_CLS
        (void)hipStreamWaitEvent(sk, c->ev_fork, 0);
        if (k < N_CLS) {
            const int hi = cls_hi[k];
            const size_t lds = L.pack8 ? (size_t)((hi + 2) / 2) * 64 * 4 + (size_t)((hi + 7) / 8) * 64 * 4
                                       : (size_t)(hi + 1) * 64 * 4 + (size_t)hi * 64;
            auto kern = L.pack8 ? k_ext_seeds<true, true> : k_ext_seeds<false, false>;
            size_t lds_k = lds;
            if (L.pack8 && L.perm_scores) {                      // the byte-permute score table (query one base per byte), the column loop in groups of four with two register sets for the LDS words
                kern = k_ext_seeds<true, true, true, true>;      // (without the next pair's LDS word requested ahead, BM2_EXT_PREFETCH=0, and with the column loop in pairs, BM2_EXT_GROUP4=0:
                lds_k = (size_t)(2 * ((hi + 3) / 4) + 2) * 64 * 4 + (size_t)((hi + 3) / 4 + 1) * 64 * 4;      //  both slower in every sweep since round 4, profiles/r04d_sweep.json, r05k_sweep.json -- those instantiations left the tree in round 6)
            }
            if (L.pack8 && L.perm_scores && L.reg_qmin > 0 && hi >= L.reg_qmin && hi >= 80 && hi <= 128) {       // rows in registers (lane_dp8r): no LDS
                // (the 144- and 160-column instantiations are not in the tree: the compiler stops unrolling their 37 / 41 groups and the row lands in scratch)
                auto kr = hi == 80 ? k_ext_seeds_reg<20> : hi == 96 ? k_ext_seeds_reg<24> : hi == 112 ? k_ext_seeds_reg<28> : k_ext_seeds_reg<32>;
                hipLaunchKernelGGL(kr, dim3(grid_for(k, k + 1, 64)), dim3(64), 0, sk, c->ix, L.xp, L.tasks, L.start, k * EB_2D, (k + 1) * EB_2D,
                                   L.enc, L.off, L.len, L.slot_base, L.reg_seed, L.reg_chain, L.chn, L.seeds, L.regs, L.counters, L.rev);
            } else
            hipLaunchKernelGGL(kern, dim3(grid_for(k, k + 1, 64)), dim3(64), lds_k, sk, c->ix, L.xp, L.tasks, L.start, k * EB_2D, (k + 1) * EB_2D, hi,
                               L.enc, L.off, L.len, L.slot_base, L.reg_seed, L.reg_chain, L.chn, L.seeds, L.regs, L.counters, L.rev);
        } else {
            hipLaunchKernelGGL(k_ext_wave, dim3(grid_for(k_wave, N_CLS + 1, 4)), dim3(256), L.lds_w, sk, c->ix, L.xp, L.tasks, L.start, k_wave * EB_2D,
                               (int)N_EBINS, L.enc, L.off, L.len, L.slot_base, L.reg_seed, L.reg_chain, L.chn, L.seeds, L.regs, L.R, L.counters, L.rev);
        }
        (void)hipEventRecord(c->ev_join[k], sk);
        joined[n_joined++] = k;
    }
    // The main stream waits for the launches only AFTER the last of them is queued.  A wait is a barrier packet in the main stream's hardware
    // queue, streams share the process's hardware queues, and nothing tells which share one: a side stream that sits on the main stream's
    // queue had its launch queued BEHIND the waits for the launches before it -- profiles/r04_timeline.tsv: the class of 65..80-base
    // queries started when every other launch of its phase had ended and ran 1.4-1.8 ms alone, three times per batch.
    for (int i = 0; i < n_joined; i++) (void)hipStreamWaitEvent(L.s, c->ev_join[joined[i]], 0);
    return bm2_check(hipGetLastError(), "extension launches");
}

// Extension stage: lazy rounds (k_advance + extension of the picked seeds), then -- for reads that still have undecided
// seeds after the lazy rounds (repeat-rich reads) -- eager extension of the rest, purged by k_postfilter exactly
// as the reference does for every seed.  cursor[r] tells the post-filter where its replay starts.
// Nothing here waits for the device: the number of lazy rounds and the grid sizes come from the statistics the PREVIOUS batch left in
// the context's page-locked `ext_stat` (they arrive before bm2_batch_run returns); neither can change a result -- a lazy round that finds
// nothing to do, or an eager phase with nothing left, launches kernels that exit.
int bm2_launch_extend(bm2_ctx *c, const bm2_opt &opt, const ChainParams &cp, int n_reads, int64_t n_slots, const uint8_t *enc,
                      const int64_t *off, const int32_t *len, const int64_t *read_base, const int32_t *n_chain, const int32_t *n_reg,
                      const int64_t *slot_base, const int32_t *reg_seed, const int32_t *reg_chain, const DevChain *chn,
                      const DevSeed *seeds, int32_t *srt_all, DevReg *regs, unsigned long long *counters, DevBuf &tmp, DevBuf &scan_tmp,
                      int32_t *cursor, int max_len) {
    hipStream_t s = c->stream;
    int rc;
    if ((rc = bm2_check(hipMemsetAsync(cursor, 0, (size_t)(n_reads + 1) * 4, s), "memset cursor"))) return rc;
    if (n_slots <= 0) return BM2_OK;
    if ((rc = bm2_side_streams(c))) return rc;
    ExtLaunch L;
    L.c = c; L.s = s; L.enc = enc; L.off = off; L.len = len; L.slot_base = slot_base; L.reg_seed = reg_seed; L.reg_chain = reg_chain;
    L.chn = chn; L.seeds = seeds; L.regs = regs; L.counters = counters;
    ExtParams &xp = L.xp;
    xp.a = opt.a; xp.w = opt.w; xp.pen_clip5 = opt.pen_clip5; xp.pen_clip3 = opt.pen_clip3;
    SwParams P;
    P.o_del = opt.o_del; P.e_del = opt.e_del; P.o_ins = opt.o_ins; P.e_ins = opt.e_ins; P.zdrop = opt.zdrop; P.max_sc = opt.a;
    for (int i = 0; i < 25; i++) P.mat[i] = opt.mat[i];
    P.end_bonus = opt.pen_clip5; xp.left = P;
    P.end_bonus = opt.pen_clip3; xp.right = P;
    // no H / E of the batch can exceed l_query * a (a full-length perfect match): 8-bit rows when that fits
    L.pack8 = !bm2_knob("BM2_NO_PACK8", 0) && (int64_t)max_len * opt.a <= 255 && opt.a > 0;
    L.wave_qmin = bm2_knob("BM2_EXT_WAVE_QMIN", 129);               // classes of queries at least this long: one seed per wavefront
    // (113 until the launches of a phase really ran beside each other, see run_phase: the wavefront kernel then took 200 000 seeds of the second
    //  round -- short seeds at a read's end, the whole rest of the read to extend -- and was the last launch of its phase to end by 2 ms;
    //  profiles/r04q_sweep.json: extension 20.1 ms at 113, 16.0 ms at 129 / 145 / 161, 25.0 ms at 97)
    L.rev = bm2_knob("BM2_EXT_REVERSE", 1);
    L.qmap = bm2_knob("BM2_EXT_QUEUE_MAP", 1);
    L.perm_scores = bm2_knob("BM2_EXT_PERM_SCORES", 1);
    // Rows in registers for the classes from this query length up (k_ext_seeds_reg; 0: none).  Measured on the 3100 Mbp chunk (profiles/r06c_*, r06d_*: two
    // sweeps): the 113..128-base class alone -- each phase's longest launch, 6 wavefronts per CU on LDS rows -- extension 15.7-16.2 -> 14.5-14.8 ms; with
    // the 97..112 class as well 14.8-15.0, from 81 bases 15.4-15.6, from 49 bases 16.0-16.1 (the shorter classes' LDS rows already fit 8-12 wavefronts per
    // CU and the register kernel pays a scalar test per group of its class, not of its band).  On seam S1 (one phase, a leaner kernel around the same
    // rows): 809 -> 1055 G cells/s from 96 bases up (r06c_bench_bsw_reg_qmin*.json).
    L.reg_qmin = bm2_knob("BM2_EXT_REG_QMIN", 128);
    for (int k : { 0, 1, 4 }) if (opt.mat[k] < -128 || opt.mat[k] > 127) L.perm_scores = 0;     // (the score table holds signed bytes)
    const int lazy_max = bm2_knob("BM2_EXT_ROUNDS", LAZY_ROUNDS), pend_div = bm2_knob("BM2_EXT_PEND_DIV", 12);
    L.R = ring_size2(opt.w << (MAX_BAND_TRY - 1));
    L.lds_w = ((sizeof(ExtParams) + 3) / 4) * 4 + (size_t)4 * 2 * L.R * 4;
    if (L.lds_w > 160 * 1024) { bm2_set_error("band width %d needs more LDS than a CU has", opt.w); return BM2_EUNSUP; }
    // what the previous batch of this context saw: seeds per class and phase (grid sizes), reads pending after each lazy round (how many
    // lazy rounds pay: a round for a twelfth of the reads costs more in launches than it saves in cells)
    uint32_t hint[BM2_EXT_PHASES][BM2_EXT_STATW];
    const bool have_hint = c->ext_stat && c->ext_stat_reads > 0;
    int n_lazy = 2;
    if (have_hint) {
        memcpy(hint, c->ext_stat, sizeof hint);
        if (c->ext_stat_reads != n_reads)                           // another batch size: the class counts scale with it
            for (int p = 0; p < BM2_EXT_PHASES; p++)
                for (int k = 0; k <= N_CLS; k++) hint[p][k] = (uint32_t)((uint64_t)hint[p][k] * (uint32_t)n_reads / (uint32_t)c->ext_stat_reads) + 64;
        n_lazy = 1;
        for (int r = 0; r < c->ext_stat_rounds && r + 1 < EXT_EAGER_PHASE; r++) {
            const uint32_t pend = hint[r][N_CLS + 1];
            if (pend && (uint64_t)pend * (uint32_t)pend_div >= (uint32_t)c->ext_stat_reads) n_lazy = r + 2; else break;
        }
    }
    if (n_lazy > lazy_max) n_lazy = lazy_max;
    if (n_lazy > BSW_STAT_ROW) n_lazy = BSW_STAT_ROW;              // (rows [0, BSW_STAT_ROW) are the lazy rounds')
    // scratch: ebin[n_items] | hist[N_EBINS] | pend[PHASES] | stat[PHASES][STATW] | start[N_EBINS + 1] | cur_slot[n_reads] | tasks[n_items] | kept[n_slots] | n_kept[n_reads]
    const size_t n_items = (size_t)(n_slots > n_reads ? n_slots : n_reads);
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t o_hist = up(n_items * 4), o_pend = up(o_hist + (size_t)N_EBINS * 4), o_stat = o_pend + 256;
    const size_t o_start = up(o_stat + sizeof hint), o_cs = up(o_start + (size_t)(N_EBINS + 2) * 8), o_task = up(o_cs + (size_t)n_reads * 4);
    const size_t o_kept = up(o_task + n_items * 4), o_nk = up(o_kept + (size_t)n_slots * 4);
    if ((rc = bm2_reserve(tmp, o_nk + (size_t)n_reads * 4 + 256))) return rc;
    char *base = (char *)tmp.p;
    int32_t *kept = (int32_t *)(base + o_kept), *n_kept = (int32_t *)(base + o_nk);
    if ((rc = bm2_check(hipMemsetAsync(n_kept, 0, (size_t)n_reads * 4, s), "memset n_kept"))) return rc;
    uint32_t *ebin = (uint32_t *)base, *pend = (uint32_t *)(base + o_pend), *stat = (uint32_t *)(base + o_stat);
    int32_t *hist = (int32_t *)(base + o_hist), *cur_slot = (int32_t *)(base + o_cs), *tasks = (int32_t *)(base + o_task);
    int64_t *start = (int64_t *)(base + o_start);
    L.tasks = tasks; L.start = start;
    if ((rc = bm2_check(hipMemsetAsync(hist, 0, o_start - o_hist, s), "memset hist"))) return rc;      // histogram, pending counters, statistics
    auto sort_and_run = [&](int phase, int64_t n_sort, bool by_read) -> int {
        int rc2;
        if ((rc2 = bm2_scan_i32(c, hist, N_EBINS, start, scan_tmp))) return rc2;
        hipLaunchKernelGGL(k_task_scatter, dim3((unsigned)((n_sort + 255) / 256)), dim3(256), 0, s, n_sort, ebin, hist, start, tasks,
                           by_read ? read_base : (const int64_t *)nullptr, by_read ? cur_slot : (const int32_t *)nullptr);
        hipLaunchKernelGGL(k_phase_stats, dim3(1), dim3(64), 0, s, start, by_read ? pend + phase : (const uint32_t *)nullptr, stat + (size_t)phase * BM2_EXT_STATW);
        return run_phase(L, have_hint ? hint[phase] : nullptr, n_sort);
    };
    const unsigned nbr = (unsigned)((n_reads + 127) / 128);
    for (int round = 0; round < n_lazy; round++) {
        hipLaunchKernelGGL(k_advance, dim3(nbr), dim3(128), 0, s, cp, xp, n_reads, len, read_base, n_reg, reg_chain, chn, seeds, srt_all,
                           regs, cursor, cur_slot, ebin, hist, pend + round, kept, n_kept);
        if ((rc = sort_and_run(round, n_reads, true))) return rc;
    }
    {       // eager remainder: every seed at or beyond its read's cursor
        const unsigned nb = (unsigned)((n_slots + 255) / 256);
        hipLaunchKernelGGL(k_reg_init, dim3(nb), dim3(256), 0, s, xp, n_slots, len, slot_base, reg_seed, reg_chain, chn, seeds, regs, ebin, hist, cursor);
        if ((rc = sort_and_run(EXT_EAGER_PHASE, n_slots, false))) return rc;
    }
    if (c->ext_stat) {      // (arrives before the caller's end-of-batch synchronisation; read by the next batch)
        // only the rows this stage wrote -- the lazy rounds and the eager phase: the row between them belongs to the S1 batches (BSW_STAT_ROW),
        // whose class counts a copy of the whole table would overwrite with zeros
        const size_t row = sizeof(uint32_t) * BM2_EXT_STATW;
        if ((rc = bm2_check(hipMemcpyAsync(c->ext_stat, stat, row * (size_t)n_lazy, hipMemcpyDeviceToHost, s), "D2H extension statistics"))) return rc;
        if ((rc = bm2_check(hipMemcpyAsync((char *)c->ext_stat + row * EXT_EAGER_PHASE, (const char *)stat + row * EXT_EAGER_PHASE, row, hipMemcpyDeviceToHost, s),
                            "D2H extension statistics (eager phase)"))) return rc;
        c->ext_stat_reads = n_reads; c->ext_stat_rounds = n_lazy;
    }
    return bm2_check(hipGetLastError(), "extension launches");
}

// ---- seam S1 on the lane kernel: BandedPairWiseSW::getScores8 / getScores16 (bandedSWA.h:126-135) are inter-task SIMD kernels -- one pair per
// SIMD lane -- and so is this: the pairs of a batch whose query fits the lane kernel's LDS rows and whose scores fit its 8-bit cells are
// counting-sorted on the device by (query length, target length / 4) and run one per lane through the same row code as the pipeline's
// extension stage (lane_dp8g); the others -- long queries, int16 / int32-class scores -- stay on the pair-per-wavefront kernel (bsw.hip).
__global__ void __launch_bounds__(256)
k_bsw_bin(const bm2_seqpair_t *__restrict__ pairs, int n, int a_match, int lanes_ok, uint32_t *ebin, int32_t *hist) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int len1 = pairs[i].len1, len2 = pairs[i].len2, h0 = pairs[i].h0;
    uint32_t b = (uint32_t)EBIN_FALLBACK;
    if (lanes_ok && len2 >= 1 && len2 <= LANE_QMAX && len1 >= 1 && len1 < 32768 && h0 >= 0 && h0 + len2 * a_match <= 255)
        b = (uint32_t)(((len2 - 1) >> 4) * EB_2D + len2 * EB_L + imin(len1 >> 2, LANE_QMAX));
    ebin[i] = b;
    atomicAdd(&hist[b], 1);
}

template <int RG>                                            // RG > 0: the rows of RG groups in registers (lane_dp8r), no LDS
__global__ void __launch_bounds__(64)
k_bsw_lanes(bm2_seqpair_t *pairs, const uint8_t *__restrict__ ref, const uint8_t *__restrict__ qer, const int32_t *__restrict__ tasks_all,
            const int64_t *__restrict__ start, int bin_lo, int bin_hi, int qmax, int w, SwParams P, unsigned long long *cells_out, int rev) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_l[];
    uint32_t *EH = lds_l;
    uint32_t *QL8 = lds_l + (size_t)(2 * ((qmax + 3) / 4) + 2) * 64;
    const int lane = threadIdx.x;
    const int64_t first = start[bin_lo];
    const int n_tasks = (int)(start[bin_hi] - first);
    const int32_t *tasks = tasks_all + first;
    const int n_tiles = (n_tasks + 63) >> 6;
    long long cells = 0, iters = 0;
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int idx = (rev ? n_tiles - 1 - tile : tile) * 64 + lane;
        const bool valid = idx < n_tasks;
        int id = 0, len1 = 0, len2 = 0, h0 = 0;
        const uint8_t *q = qer, *t = ref;
        if (valid) { id = tasks[idx]; len1 = pairs[id].len1; len2 = pairs[id].len2; h0 = pairs[id].h0; q = qer + pairs[id].idq; t = ref + pairs[id].idr; }
        const int maxq = __builtin_amdgcn_readlane(wave_scan_max(valid ? len2 : 0, 0), 63);
        uint32_t QR[RG + 1];
        if constexpr (RG > 0) load_query_regs<RG>(valid, q, 1, len2, QR);
        else
        for (int j0 = 0; j0 < maxq; j0 += 4) {
            if (valid && j0 < len2) {
                uint32_t wq = 0;
                for (int u = 0; u < 4 && j0 + u < len2; u++) { const uint32_t qv = q[j0 + u]; wq |= (qv > 3 ? 4u : qv) << (8 * u); }
                QL8[(j0 >> 2) * 64 + lane] = wq;
            }
        }
        const int cls = pair_class(len1, len2, h0, P.max_sc);
        const int wc = band_clamp(w, len2, P, cls);
        LaneOut o; o.score = h0; o.qle = o.tle = o.gtle = 0; o.gscore = -1; o.max_off = 0;
        if constexpr (RG > 0) lane_dp8r<RG>(valid, len2, len1, wc, h0, RefPtr::bytes(t), 1, P, QR, o, cells, iters);
        else lane_dp8g(valid, len2, len1, wc, h0, RefPtr::bytes(t), 1, P, EH, QL8, lane, o, cells, iters);
        if (valid) {
            bm2_seqpair_t *d = &pairs[id];
            d->score = o.score; d->tle = o.tle; d->gtle = o.gtle; d->qle = o.qle; d->gscore = o.gscore; d->max_off = o.max_off;
        }
    }
    if (cells_out) atomicAdd(cells_out, (unsigned long long)cells);
}

int bm2_launch_bsw_list(bm2_ctx *c, bm2_seqpair_t *d_pairs, const uint8_t *d_ref, const uint8_t *d_qer, const int32_t *list, const int64_t *start,
                        int bin_lo, int bin_hi, unsigned grid, int w, const SwParams &P, unsigned long long *d_cells, hipStream_t s);      // bsw.hip

// -> BM2_OK and *done = true when the batch went through the sorted path; *done = false: the caller launches the plain pair-per-wavefront kernel
int bm2_launch_bsw_sorted(bm2_ctx *c, bm2_seqpair_t *d_pairs, const uint8_t *d_ref, const uint8_t *d_qer, int n, int w, const SwParams &P,
                          unsigned long long *d_cells, bool *done) {
    *done = false;
    if (!bm2_knob("BM2_BSW_LANES", 1) || n < bm2_knob("BM2_BSW_LANES_MIN", 4096)) return BM2_OK;      // (small batches: the sort's fixed cost is not worth it)
    for (int i = 0; i < 5; i++)
        for (int j = 0; j < 5; j++) {                                   // the lane kernel scores by (match, mismatch, ambiguous) bytes: bwa_fill_scmat's form only
            const int want = (i == 4 || j == 4) ? P.mat[4] : i == j ? P.mat[0] : P.mat[1];
            if (P.mat[i * 5 + j] != want || want < -128 || want > 127) return BM2_OK;
        }
    if (P.max_sc <= 0) return BM2_OK;
    int rc;
    if ((rc = bm2_side_streams(c))) return rc;
    hipStream_t s = c->stream;
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t o_hist = up((size_t)n * 4), o_stat = up(o_hist + (size_t)N_EBINS * 4), o_start = up(o_stat + BM2_EXT_STATW * 4);
    const size_t o_task = up(o_start + (size_t)(N_EBINS + 2) * 8);
    if ((rc = bm2_reserve(c->b_pairs2, o_task + (size_t)n * 4 + 256))) return rc;
    char *base = (char *)c->b_pairs2.p;
    uint32_t *ebin = (uint32_t *)base, *stat = (uint32_t *)(base + o_stat);
    int32_t *hist = (int32_t *)(base + o_hist), *tasks = (int32_t *)(base + o_task);
    int64_t *start = (int64_t *)(base + o_start);
    if ((rc = bm2_check(hipMemsetAsync(hist, 0, o_start - o_hist, s), "memset hist"))) return rc;
    hipLaunchKernelGGL(k_bsw_bin, dim3((n + 255) / 256), dim3(256), 0, s, (const bm2_seqpair_t *)d_pairs, n, (int)P.max_sc, 1, ebin, hist);
    if ((rc = bm2_scan_i32(c, hist, N_EBINS, start, c->b_scan))) return rc;
    hipLaunchKernelGGL(k_task_scatter, dim3((n + 255) / 256), dim3(256), 0, s, (int64_t)n, ebin, hist, start, tasks, (const int64_t *)nullptr, (const int32_t *)nullptr);
    hipLaunchKernelGGL(k_phase_stats, dim3(1), dim3(64), 0, s, start, (const uint32_t *)nullptr, stat);
    static const int cls_hi[N_CLS] = { 16, 32, 48, 64, 80, 96, 112, 128, 144, 160 };
    const uint32_t *hint = c->ext_stat && c->bsw_stat_n == n ? c->ext_stat + (size_t)BSW_STAT_ROW * BM2_EXT_STATW : nullptr;
    uint32_t hcopy[BM2_EXT_STATW];
    if (hint) { memcpy(hcopy, hint, sizeof hcopy); hint = hcopy; }
    (void)hipEventRecord(c->ev_fork, s);
    for (int k = N_CLS; k >= 0; k--) {                                   // long queries first
        hipStream_t sk = c->side_stream[k];
        int64_t cnt = hint ? (int64_t)hint[k] + hint[k] / 4 + 64 : n;
        if (cnt > n) cnt = n;
        (void)hipStreamWaitEvent(sk, c->ev_fork, 0);
        if (k < N_CLS) {
            const int hi = cls_hi[k];
            const size_t lds = (size_t)(2 * ((hi + 3) / 4) + 2) * 64 * 4 + (size_t)((hi + 3) / 4 + 1) * 64 * 4;
            int64_t g = (cnt + 63) / 64; if (g < 1) g = 1; if (g > (1 << 16)) g = 1 << 16;
            const int reg_qmin = bm2_knob("BM2_BSW_REG_QMIN", 96);     // (S1's own knob: its kernel around the register rows is leaner than the pipeline's, profiles/r06c_bench_bsw_reg_qmin*.json)
            auto kb = k_bsw_lanes<0>;
            size_t lds_k = lds;
            if (reg_qmin > 0 && hi >= reg_qmin && hi >= 64 && hi <= 128) {
                lds_k = 0;
                kb = hi == 64 ? k_bsw_lanes<16> : hi == 80 ? k_bsw_lanes<20> : hi == 96 ? k_bsw_lanes<24> : hi == 112 ? k_bsw_lanes<28> : k_bsw_lanes<32>;
            }
            hipLaunchKernelGGL(kb, dim3((unsigned)g), dim3(64), lds_k, sk, d_pairs, d_ref, d_qer, tasks, start, k * EB_2D, (k + 1) * EB_2D, hi, w, P, d_cells, 1);
        } else {
            int64_t g = (cnt + 3) / 4; if (g < 1) g = 1; if (g > (1 << 20)) g = 1 << 20;
            if ((rc = bm2_launch_bsw_list(c, d_pairs, d_ref, d_qer, tasks, start, N_CLS * EB_2D, (int)N_EBINS, (unsigned)g, w, P, d_cells, sk))) return rc;
        }
        (void)hipEventRecord(c->ev_join[k], sk);
    }
    for (int k = N_CLS; k >= 0; k--) (void)hipStreamWaitEvent(s, c->ev_join[k], 0);        // (after the last launch is queued: see run_phase)
    if (c->ext_stat) {
        if ((rc = bm2_check(hipMemcpyAsync(c->ext_stat + (size_t)BSW_STAT_ROW * BM2_EXT_STATW, stat, BM2_EXT_STATW * 4, hipMemcpyDeviceToHost, s), "D2H S1 class counts"))) return rc;
        c->bsw_stat_n = n;
    }
    *done = true;
    return bm2_check(hipGetLastError(), "S1 lane launches");
}

int bm2_launch_slot_base(bm2_ctx *c, int n_reads, const int64_t *read_base, const int32_t *n_reg, int64_t *slot_base) {
    if (n_reads <= 0) return BM2_OK;
    hipLaunchKernelGGL(k_slot_base, dim3((n_reads + 255) / 256), dim3(256), 0, c->stream, n_reads, read_base, n_reg, slot_base);
    return bm2_check(hipGetLastError(), "k_slot_base launch");
}

int bm2_pf_heavy_threshold() { const int t = bm2_knob("BM2_PF_HEAVY", PF_HEAVY); return t > 0 ? t : PF_HEAVY; }

int bm2_launch_postfilter(bm2_ctx *c, const ChainParams &o, int n_reads, const int32_t *len, const int64_t *read_base,
                          const int32_t *n_chain, const int32_t *n_reg, const DevChain *chn, const DevSeed *seeds,
                          int32_t *srt_all, DevReg *regs, int32_t *n_out, const int32_t *cursor, const int32_t *perm,
                          const int32_t *heavy, const int64_t *n_heavy, unsigned long long *item_cur, int pf_heavy) {
    if (n_reads <= 0) return BM2_OK;
    // (different reads: the two kernels run beside each other -- the lane-per-read one on a side stream whose hardware queue is not the main stream's)
    hipStream_t sl = c->stream;
    const bool beside = heavy && bm2_side_streams(c) == BM2_OK;
    if (beside) {
        sl = c->side_stream[2];
        (void)hipEventRecord(c->ev_fork, c->stream);
        (void)hipStreamWaitEvent(sl, c->ev_fork, 0);
    }
    hipLaunchKernelGGL(k_postfilter, dim3((n_reads + 127) / 128), dim3(128), 0, sl, o, n_reads, len, read_base, n_chain,
                       n_reg, chn, seeds, srt_all, regs, n_out, cursor, perm, heavy ? pf_heavy : 0x7fffffff);
    if (beside) (void)hipEventRecord(c->ev_join[2], sl);
    if (heavy) {
        hipLaunchKernelGGL(k_postfilter_heavy, dim3(c->n_cu * 4), dim3(256), 0, c->stream, o, heavy, n_heavy, len, read_base, n_chain,
                           n_reg, chn, seeds, srt_all, regs, n_out, cursor, item_cur, pf_heavy);
    }
    if (beside) (void)hipStreamWaitEvent(c->stream, c->ev_join[2], 0);
    return bm2_check(hipGetLastError(), "k_postfilter launch");
}

int bm2_launch_reg_gather(bm2_ctx *c, int n_reads, const int64_t *read_base, const int32_t *n_reg, const DevReg *regs,
                          const int64_t *out_off, bm2_reg_t *out, int64_t out_cap) {
    if (n_reads <= 0) return BM2_OK;
    hipLaunchKernelGGL(k_reg_gather, dim3((n_reads + 255) / 256), dim3(256), 0, c->stream, n_reads, read_base, n_reg, regs,
                       out_off, out, out_cap);
    return bm2_check(hipGetLastError(), "k_reg_gather launch");
}
