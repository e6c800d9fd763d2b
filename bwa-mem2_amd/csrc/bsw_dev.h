// bsw_dev.h -- banded affine-gap extension DP, one task per 64-lane wavefront.
//
// Semantics = ksw_extend2 (ksw.cpp:432-533) = BandedPairWiseSW::scalarBandedSWA (bandedSWA.cpp:116-237),
// which the reference's int8/int16 SIMD kernels reproduce lane by lane.  The reference vectorises ACROSS
// tasks (one pair per SIMD lane); here one wavefront owns one task and vectorises ALONG A ROW:
//
//   * lane l of a 64-column chunk owns column j = jb + l of target row i;
//   * H(i-1,j-1) and E(i,j) live in a per-wave LDS ring indexed by (j mod R): the band [beg,end) of
//     row i is always inside [i-w, i+w+1], so R >= 2w+4 slots never alias live columns (this is what
//     keeps 10 kb queries inside a few KB of LDS);
//   * the row recurrence  F(i,j+1) = max(F(i,j) - e_ins, max(M(i,j) - o_ins - e_ins, 0))  takes M, not H
//     (bandedSWA.cpp:181-199: "separating H and M"), so T(j) = max(M - oe_ins, 0) does not depend on F and
//     F is a max-plus prefix scan over the chunk: F(jb+l) = max(f0 - l*e, max_{l'<l}(T(l') + l'*e) - (l-1)*e),
//     done with 6 DPP row_shr/row_bcast steps instead of a serial loop;
//   * everything the reference decides once per row -- band clamp to [i-w, i+w+1], first-column h1, gscore at
//     the query end, the m==0 and z-drop exits, max/max_off, the beg/end shrink from the zero pattern of the
//     stored row -- is wave-uniform scalar code driven by ballots.
//
// Integer DP: no MFMA anywhere (nothing here is a dense contraction).
#pragma once
#include "bm2_dev.h"

#define DPP_ROW_SHR(n)   (0x110 + (n))
#define DPP_WAVE_SHR1    0x138
#define DPP_ROW_BCAST15  0x142
#define DPP_ROW_BCAST31  0x143
#define NEG_BIG          (-(1 << 29))

// lane l <- value of lane l-1; lane 0 <- fill
static __device__ __forceinline__ int wave_shr1(int v, int fill) {
    return __builtin_amdgcn_update_dpp(fill, v, DPP_WAVE_SHR1, 0xf, 0xf, false);
}

// inclusive max-scan over the 64 lanes (identity = ident); all lanes must be active
static __device__ __forceinline__ int wave_scan_max(int v, int ident) {
    v = imax(v, __builtin_amdgcn_update_dpp(ident, v, DPP_ROW_SHR(1), 0xf, 0xf, false));
    v = imax(v, __builtin_amdgcn_update_dpp(ident, v, DPP_ROW_SHR(2), 0xf, 0xf, false));
    v = imax(v, __builtin_amdgcn_update_dpp(ident, v, DPP_ROW_SHR(4), 0xf, 0xf, false));
    v = imax(v, __builtin_amdgcn_update_dpp(ident, v, DPP_ROW_SHR(8), 0xf, 0xf, false));
    v = imax(v, __builtin_amdgcn_update_dpp(ident, v, DPP_ROW_BCAST15, 0xa, 0xf, false));
    v = imax(v, __builtin_amdgcn_update_dpp(ident, v, DPP_ROW_BCAST31, 0xc, 0xf, false));
    return v;
}

static __device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

struct SwOut { int score, qle, tle, gtle, gscore, max_off; };

// Pair class of the reference's three BSW entry points (bwamem.cpp:1947-1953, 2304-2313)
static __device__ __forceinline__ int pair_class(int len1, int len2, int h0, int a) {
    int minval = h0 + imin(len1, len2) * a;
    if (len1 < 128 && len2 < 128 && minval < 128) return 8;
    if (len1 < 32768 && len2 < 32768 && minval < 32768) return 16;
    return 32;
}

// Band clamp.  Scalar: bandedSWA.cpp:148-156 (signed, double division).  The int8/int16 wrappers compute the
// same bound in wrapping unsigned lane arithmetic with an integer division (bandedSWA.cpp:635-653, 1333-1353);
// identical for the default and ont2d scoring, different when len2*a + end_bonus - o < 0.
static __device__ __forceinline__ int band_clamp(int w, int qlen, const SwParams &P, int cls) {
    int max_ins, max_del;
    if (cls == 32) {
        max_ins = (int)((double)(qlen * P.max_sc + P.end_bonus - P.o_ins) / P.e_ins + 1.);
        max_del = (int)((double)(qlen * P.max_sc + P.end_bonus - P.o_del) / P.e_del + 1.);
    } else {
        unsigned mask = cls == 8 ? 0xffu : 0xffffu;
        unsigned q = (unsigned)(qlen * P.max_sc) & mask;
        unsigned ti = (q + ((unsigned)(P.end_bonus - P.o_ins) & mask)) & mask;
        unsigned td = (q + ((unsigned)(P.end_bonus - P.o_del) & mask)) & mask;
        max_ins = (int)(ti / (unsigned)P.e_ins) + 1;
        max_del = (int)(td / (unsigned)P.e_del) + 1;
    }
    max_ins = imax(max_ins, 1);
    w = imin(w, max_ins);
    max_del = imax(max_del, 1);
    w = imin(w, max_del);
    return w;
}

// Z-drop test of one row, after its maximum m (column mj) was folded into (maxv, max_i, max_j); di = i - max_i, dj = mj - max_j.
// Class 32 is ksw_extend2's test (bandedSWA.cpp:210-216).  The int8 / int16 SIMD kernels have their own (ZSCORE8 / ZSCORE16,
// bandedSWA.cpp:268-281, 309-322): evaluated on EVERY row, also the one that raised the maximum and also for zdrop <= 0, in
// arithmetic that wraps at the lane width (zdrop itself is truncated: 200 is -56 in an int8 lane), and the diagonal offset is
// not multiplied by the gap extension penalty.  oracle/bm2_oracle.c: ora_ksw_extend_cls.
static __device__ __forceinline__ bool zdrop_stop(int cls, bool new_max, int maxv, int m, int di, int dj, int e_del, int e_ins, int zdrop) {
    if (cls == 32) {
        if (new_max || zdrop <= 0) return false;
        return (di > dj ? maxv - m - (di - dj) * e_del : maxv - m - (dj - di) * e_ins) > zdrop;
    }
    const int sh = 32 - cls;
    auto wr = [sh](int x) { return (int)((unsigned)x << sh) >> sh; };              // truncate to a signed cls-bit lane
    const int ti = wr(di), tj = wr(dj);
    const int diff = ti > tj ? wr(ti - tj) : wr(tj - ti);
    return wr(wr(maxv - m) - diff) > wr(zdrop);
}

// One extension on one wavefront.  All arguments wave-uniform.  RH/RE: this wave's LDS rings (RM = R-1, R a
// power of two >= 2*w+4).  q/t are read with strides qs/ts (-1 walks a left extension backwards through the
// read and through ref_string, so no reversed copies are ever materialised, cf. bwamem.cpp:2268-2290).
// `w` must already be clamped.  Returns the number of DP cells computed.
static __device__ int bsw_extend_wave(const uint8_t *__restrict__ qp, int qs, int qlen,
                                      RefPtr tp, int ts, int tlen,
                                      int w, int h0, const SwParams &P, int *RH, int *RE, int RM, SwOut &out) {
    const int lane = threadIdx.x & 63;
    const int oe_del = P.o_del + P.e_del, oe_ins = P.o_ins + P.e_ins, e_del = P.e_del, e_ins = P.e_ins;
    const int e1 = h0 > oe_ins ? h0 - oe_ins : 0;                  // first row, bandedSWA.cpp:143-145
    const int cls = pair_class(tlen, qlen, h0, P.max_sc);
    int beg = 0, end = qlen, maxv = h0, max_i = -1, max_j = -1, max_ie = -1, gscore = -1, max_off = 0;
    int maxEnd = -1;                                               // columns <= maxEnd have been stored in the ring
    int cells = 0;
    int tchunk = 4;
    for (int i = 0; i < tlen; ++i) {
        if ((i & 63) == 0) { int ti = i + lane; tchunk = ti < tlen ? (int)tp[(int64_t)ti * ts] : 4; }
        const int tb = __builtin_amdgcn_readlane(tchunk, i & 63);
        const int s0 = P.mat[tb * 5 + 0], s1 = P.mat[tb * 5 + 1], s2 = P.mat[tb * 5 + 2], s3 = P.mat[tb * 5 + 3],
                  s4 = P.mat[tb * 5 + 4];
        if (beg < i - w) beg = i - w;                               // bandedSWA.cpp:166-168
        if (end > i + w + 1) end = i + w + 1;
        if (end > qlen) end = qlen;
        int h1 = 0;                                                 // H(i, beg-1), :170-173
        if (beg == 0) { h1 = h0 - (P.o_del + e_del * (i + 1)); if (h1 < 0) h1 = 0; }
        int fc = 0, m = 0, mj = -1, firstnz = -1, lastnz = -1;
        cells += imax(end - beg, 0);
        for (int jb = beg; jb <= end; jb += 64) {
            const int j = jb + lane;
            const bool act = j < end, st = j <= end;
            int Hd = 0, E = 0;
            if (st) {
                if (j <= maxEnd) { Hd = RH[j & RM]; E = RE[j & RM]; }
                else { Hd = j == 0 ? h0 : imax(e1 - (j - 1) * e_ins, 0); }
            }
            const int qb = act ? (int)qp[(int64_t)j * qs] : 4;
            const int sc = qb == 0 ? s0 : qb == 1 ? s1 : qb == 2 ? s2 : qb == 3 ? s3 : s4;
            const int M = (act && Hd) ? Hd + sc : 0;                // :181-186
            const int T = imax(M - oe_ins, 0);
            const int U = T + lane * e_ins;
            const int Pm = wave_scan_max(U, 0);
            const int Pprev = wave_shr1(Pm, NEG_BIG);
            const int F = imax(fc - lane * e_ins, Pprev - (lane - 1) * e_ins);
            int h = imax(imax(M, E), F);
            if (!act) h = 0;
            const int hs = wave_shr1(h, h1);                        // H(i, j-1): what eh[j].h holds for the next row
            const int en = act ? imax(E - e_del, imax(M - oe_del, 0)) : 0;
            if (st) { RH[j & RM] = hs; RE[j & RM] = en; }
            const int nact = imin(64, end - jb);
            if (nact > 0) {
                const int cm = __builtin_amdgcn_readlane(wave_scan_max(h, 0), 63);
                const unsigned long long eq = __ballot(act && h == cm);
                if (cm >= m) { m = cm; mj = jb + 63 - __builtin_clzll(eq); }   // last column among equal maxima, :188-189
                h1 = __builtin_amdgcn_readlane(h, nact - 1);
                fc = imax(fc - 64 * e_ins, __builtin_amdgcn_readlane(Pm, 63) - 63 * e_ins);
            }
            const unsigned long long nzb = __ballot(act && (hs | en) != 0);
            const unsigned long long nze = __ballot(st && (hs | en) != 0);
            if (firstnz < 0 && nzb) firstnz = jb + __builtin_ctzll(nzb);
            if (nze) lastnz = jb + 63 - __builtin_clzll(nze);
        }
        maxEnd = imax(maxEnd, end);
        const int jfin = beg < end ? end : beg;
        if (jfin == qlen) {                                         // :202-205
            max_ie = gscore > h1 ? max_ie : i;
            gscore = gscore > h1 ? gscore : h1;
        }
        if (m == 0) break;                                          // :206
        const bool new_max = m > maxv;
        if (new_max) {
            maxv = m; max_i = i; max_j = mj;
            const int d = mj - i;
            max_off = imax(max_off, d < 0 ? -d : d);
        }
        if (zdrop_stop(cls, new_max, maxv, m, i - max_i, mj - max_j, e_del, e_ins, P.zdrop)) break;     // :210-216 / ZSCORE8/16
        const int nb = firstnz >= 0 ? firstnz : end;                // :218-221
        const int jl = imax(lastnz, nb - 1);
        beg = nb;
        end = jl + 2 < qlen ? jl + 2 : qlen;
        beg = uni(beg); end = uni(end);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    out.score = maxv; out.qle = max_j + 1; out.tle = max_i + 1; out.gtle = max_ie + 1;
    out.gscore = gscore; out.max_off = max_off;
    return cells;
}

// ---------------------------------------------------------------------------------------------------------------
// Register-resident variant for queries of up to 64*NCH - 1 bases (150 bp reads: NCH <= 3).  Lane l owns the fixed
// columns l, l+64, ... so the H/E row state and the query bases never leave VGPRs: no LDS, no per-row global loads.
// Same row-uniform control flow as bsw_extend_wave; columns outside [beg, end] simply keep their registers, which
// reproduces the reference's "stale eh[] entries are seen again when the band regrows" behaviour for free.
// Every per-row decision is kept in SGPRs (readfirstlane / ballots), so the row loop is straight-line VALU with
// scalar branches: no exec-mask juggling.
// Scoring uses the (match, mismatch, ambiguous) structure of bwa_fill_scmat (bwa.cpp:248-257), which is also all the
// reference's SIMD kernels implement (bandedSWA.cpp:286-290): P.mat[0] / P.mat[1] / P.mat[4].
template <int NCH>
static __device__ int bsw_extend_reg(const uint8_t *__restrict__ qp, int qs, int qlen_,
                                     RefPtr tp, int ts, int tlen_,
                                     int w_, int h0_, const SwParams &P, SwOut &out) {
    const int lane = threadIdx.x & 63;
    const int qlen = uni(qlen_), tlen = uni(tlen_), w = uni(w_), h0 = uni(h0_);
    const int o_del = uni(P.o_del), e_del = uni(P.e_del), e_ins = uni(P.e_ins), zdrop = uni(P.zdrop);
    const int oe_del = o_del + e_del, oe_ins = uni(P.o_ins) + e_ins;
    const int sc_match = uni(P.mat[0]), sc_mis = uni(P.mat[1]), sc_amb = uni(P.mat[4]);
    const int e1 = h0 > oe_ins ? h0 - oe_ins : 0;
    const int cls = pair_class(tlen, qlen, h0, uni(P.max_sc));
    const int le = lane * e_ins, le1 = le - e_ins;
    int H[NCH], E[NCH], Q[NCH], X[NCH];
#pragma unroll
    for (int c = 0; c < NCH; c++) {
        const int j = c * 64 + lane;
        const int qv = j < qlen ? (int)qp[(int64_t)j * qs] : 4;
        Q[c] = qv > 3 ? 5 : qv;                                                     // 5 never equals a target code
        X[c] = qv > 3 ? sc_amb : sc_mis;                                            // score against a non-matching base
        H[c] = j == 0 ? h0 : (j <= qlen ? imax(e1 - (j - 1) * e_ins, 0) : 0);      // first row, bandedSWA.cpp:143-145
        E[c] = 0;
    }
    int beg = 0, end = qlen, maxv = h0, max_i = -1, max_j = -1, max_ie = -1, gscore = -1, max_off = 0;
    int cells = 0;
    int tchunk = 4;
    for (int i = 0; i < tlen; ++i) {
        if ((i & 63) == 0) { const int ti = i + lane; tchunk = ti < tlen ? (int)tp[(int64_t)ti * ts] : 4; }
        const int tb = __builtin_amdgcn_readlane(tchunk, i & 63);
        const bool t_amb = tb > 3;
        const int s_eq = t_amb ? sc_amb : sc_match;
        beg = beg < i - w ? i - w : beg;
        end = end > i + w + 1 ? i + w + 1 : end;
        end = end > qlen ? qlen : end;
        int h1 = 0;
        if (beg == 0) { h1 = h0 - (o_del + e_del * (i + 1)); h1 = h1 < 0 ? 0 : h1; }
        int fc = 0, m = 0, mj = -1, firstnz = -1, lastnz = -1;
        cells += end > beg ? end - beg : 0;
        const int cb = beg >> 6, ce = end >> 6;
        int hcarry = h1;
        if (beg <= end) {
#pragma unroll
            for (int c = 0; c < NCH; c++) {
                if (c >= cb && c <= ce) {                                           // wave-uniform
                    const int jb = c * 64;
                    // lane masks of the band inside this chunk, built on the scalar unit
                    const int lo = beg > jb ? beg - jb : 0;
                    const int hi = end - jb < 64 ? end - jb : 64;                   // active lanes [lo, hi)
                    const int hi2 = end - jb + 1 < 64 ? end - jb + 1 : 64;          // stored lanes  [lo, hi2)
                    const unsigned long long mlo = (1ULL << lo) - 1ULL;
                    const unsigned long long mhi = hi >= 64 ? ~0ULL : ((1ULL << hi) - 1ULL);
                    const unsigned long long mhi2 = hi2 >= 64 ? ~0ULL : ((1ULL << hi2) - 1ULL);
                    const unsigned long long actm = mhi & ~mlo, stm = mhi2 & ~mlo;
                    const bool act = __builtin_amdgcn_inverse_ballot_w64(actm);
                    const bool st = __builtin_amdgcn_inverse_ballot_w64(stm);
                    const int Hd = H[c], Ec = E[c];
                    int sc = Q[c] == tb ? s_eq : X[c];
                    if (t_amb) sc = sc_amb;
                    const int M = (act && Hd != 0) ? Hd + sc : 0;
                    const int Pm = wave_scan_max(imax(M - oe_ins, 0) + le, 0);
                    const int Pprev = wave_shr1(Pm, NEG_BIG);
                    const int F = imax(fc - le, Pprev - le1);
                    const int h = act ? imax(imax(M, Ec), F) : 0;
                    const int hs = wave_shr1(h, hcarry);
                    const int en = act ? imax(imax(Ec - e_del, M - oe_del), 0) : 0;
                    H[c] = st ? hs : Hd;
                    E[c] = st ? en : Ec;
                    if (actm) {
                        // row maximum and the LAST column holding it (bandedSWA.cpp:188-189): max over (h << 6 | lane) + 1
                        const int key = act ? (((h << 6) | lane) + 1) : 0;
                        const int kmax = __builtin_amdgcn_readlane(wave_scan_max(key, 0), 63) - 1;
                        const int cm = kmax >> 6;
                        if (cm >= m) { m = cm; mj = jb + (kmax & 63); }
                        h1 = __builtin_amdgcn_readlane(h, hi - 1);
                    }
                    hcarry = __builtin_amdgcn_readlane(h, 63);
                    fc = imax(fc - 64 * e_ins, __builtin_amdgcn_readlane(Pm, 63) - 63 * e_ins);
                    const unsigned long long nz = __ballot((hs | en) != 0);
                    const unsigned long long nzb = nz & actm, nze = nz & stm;
                    if (firstnz < 0 && nzb) firstnz = jb + __builtin_ctzll(nzb);
                    if (nze) lastnz = jb + 63 - __builtin_clzll(nze);
                }
            }
        }
        const int jfin = beg < end ? end : beg;
        if (jfin == qlen) {
            max_ie = gscore > h1 ? max_ie : i;
            gscore = gscore > h1 ? gscore : h1;
        }
        if (m == 0) break;
        const bool new_max = m > maxv;
        if (new_max) {
            maxv = m; max_i = i; max_j = mj;
            const int d = mj - i;
            max_off = imax(max_off, d < 0 ? -d : d);
        }
        if (zdrop_stop(cls, new_max, maxv, m, i - max_i, mj - max_j, e_del, e_ins, zdrop)) break;
        const int nb = firstnz >= 0 ? firstnz : end;
        const int jl = imax(lastnz, nb - 1);
        beg = nb;
        end = jl + 2 < qlen ? jl + 2 : qlen;
    }
    out.score = maxv; out.qle = max_j + 1; out.tle = max_i + 1; out.gtle = max_ie + 1;
    out.gscore = gscore; out.max_off = max_off;
    return cells;
}

// ---------------------------------------------------------------------------------------------------------------
// Register-resident variant for LONG queries (10 kb reads): a window of NCH 64-column chunks that SLIDES along the band.
// The live columns of row i lie in [i - w, i + w + 1] (the band clamp, and every column stored by an earlier row was inside that
// row's band), and `beg` never decreases, so chunk c0 = beg >> 6 and the NCH - 1 = ((2w + 1) >> 6) + 1 chunks after it hold every
// column that can still be read; lane l of slot k owns column 64 (c0 + k) + l.  When c0 moves on the slots shift down by one
// (3 x NCH v_mov, once per ~64 rows) and the chunk that enters is initialised with the first-row values -- it was never stored:
// a chunk outside the previous window cannot have been.  Its query bases were requested one shift earlier.  No LDS, no
// per-row global loads: the ring variant below pays a dependent global load for the query and four LDS round trips per chunk.
template <int NCH>
static __device__ int bsw_extend_slide(const uint8_t *__restrict__ qp, int qs, int qlen_,
                                       RefPtr tp, int ts, int tlen_,
                                       int w_, int h0_, const SwParams &P, SwOut &out) {
    const int lane = threadIdx.x & 63;
    const int qlen = uni(qlen_), tlen = uni(tlen_), w = uni(w_), h0 = uni(h0_);
    const int o_del = uni(P.o_del), e_del = uni(P.e_del), e_ins = uni(P.e_ins), zdrop = uni(P.zdrop);
    const int oe_del = o_del + e_del, oe_ins = uni(P.o_ins) + e_ins;
    const int sc_match = uni(P.mat[0]), sc_mis = uni(P.mat[1]), sc_amb = uni(P.mat[4]);
    const int e1 = h0 > oe_ins ? h0 - oe_ins : 0;
    const int cls = pair_class(tlen, qlen, h0, uni(P.max_sc));
    const int le = lane * e_ins, le1 = le - e_ins;
    int H[NCH], E[NCH], Q[NCH], X[NCH];
    auto fresh_h = [&](int j) { return j == 0 ? h0 : (j <= qlen ? imax(e1 - (j - 1) * e_ins, 0) : 0); };      // first row, bandedSWA.cpp:143-145
    auto load_q = [&](int c) { const int j = c * 64 + lane; return j < qlen ? (int)qp[(int64_t)j * qs] : 4; };
    int c0 = 0;                                                     // chunk of slot 0
#pragma unroll
    for (int k = 0; k < NCH; k++) {
        const int qv = load_q(k);
        Q[k] = qv > 3 ? 5 : qv; X[k] = qv > 3 ? sc_amb : sc_mis;    // 5 never equals a target code
        H[k] = fresh_h(k * 64 + lane); E[k] = 0;
    }
    int qnext = load_q(NCH);                                        // bases of the chunk that enters next
    int beg = 0, end = qlen, maxv = h0, max_i = -1, max_j = -1, max_ie = -1, gscore = -1, max_off = 0;
    int cells = 0;
    int tchunk = 4;
    for (int i = 0; i < tlen; ++i) {
        if ((i & 63) == 0) { const int ti = i + lane; tchunk = ti < tlen ? (int)tp[(int64_t)ti * ts] : 4; }
        const int tb = __builtin_amdgcn_readlane(tchunk, i & 63);
        const bool t_amb = tb > 3;
        const int s_eq = t_amb ? sc_amb : sc_match;
        beg = beg < i - w ? i - w : beg;
        end = end > i + w + 1 ? i + w + 1 : end;
        end = end > qlen ? qlen : end;
        while ((beg >> 6) > c0) {                                   // slide: slot k <- slot k + 1, a fresh chunk enters at the top
#pragma unroll
            for (int k = 0; k + 1 < NCH; k++) { H[k] = H[k + 1]; E[k] = E[k + 1]; Q[k] = Q[k + 1]; X[k] = X[k + 1]; }
            ++c0;
            Q[NCH - 1] = qnext > 3 ? 5 : qnext; X[NCH - 1] = qnext > 3 ? sc_amb : sc_mis;
            H[NCH - 1] = fresh_h((c0 + NCH - 1) * 64 + lane); E[NCH - 1] = 0;
            qnext = load_q(c0 + NCH);
        }
        int h1 = 0;
        if (beg == 0) { h1 = h0 - (o_del + e_del * (i + 1)); h1 = h1 < 0 ? 0 : h1; }
        int fc = 0, m = 0, mj = -1, firstnz = -1, lastnz = -1;
        cells += end > beg ? end - beg : 0;
        const int cb = beg >> 6, ce = end >> 6;
        int hcarry = h1;
        if (beg <= end) {
#pragma unroll
            for (int k = 0; k < NCH; k++) {
                const int c = c0 + k;
                if (c >= cb && c <= ce) {                                           // wave-uniform
                    const int jb = c * 64;
                    const int lo = beg > jb ? beg - jb : 0;
                    const int hi = end - jb < 64 ? end - jb : 64;                   // active lanes [lo, hi)
                    const int hi2 = end - jb + 1 < 64 ? end - jb + 1 : 64;          // stored lanes  [lo, hi2)
                    const unsigned long long mlo = (1ULL << lo) - 1ULL;
                    const unsigned long long mhi = hi >= 64 ? ~0ULL : ((1ULL << hi) - 1ULL);
                    const unsigned long long mhi2 = hi2 >= 64 ? ~0ULL : ((1ULL << hi2) - 1ULL);
                    const unsigned long long actm = mhi & ~mlo, stm = mhi2 & ~mlo;
                    const bool act = __builtin_amdgcn_inverse_ballot_w64(actm);
                    const bool st = __builtin_amdgcn_inverse_ballot_w64(stm);
                    const int Hd = H[k], Ec = E[k];
                    int sc = Q[k] == tb ? s_eq : X[k];
                    if (t_amb) sc = sc_amb;
                    const int M = (act && Hd != 0) ? Hd + sc : 0;
                    const int Pm = wave_scan_max(imax(M - oe_ins, 0) + le, 0);
                    const int Pprev = wave_shr1(Pm, NEG_BIG);
                    const int F = imax(fc - le, Pprev - le1);
                    const int h = act ? imax(imax(M, Ec), F) : 0;
                    const int hs = wave_shr1(h, hcarry);
                    const int en = act ? imax(imax(Ec - e_del, M - oe_del), 0) : 0;
                    H[k] = st ? hs : Hd;
                    E[k] = st ? en : Ec;
                    if (actm) {
                        // row maximum and the LAST column holding it (bandedSWA.cpp:188-189): max over (h << 6 | lane) + 1
                        const int key = act ? (((h << 6) | lane) + 1) : 0;
                        const int kmax = __builtin_amdgcn_readlane(wave_scan_max(key, 0), 63) - 1;
                        const int cm = kmax >> 6;
                        if (cm >= m) { m = cm; mj = jb + (kmax & 63); }
                        h1 = __builtin_amdgcn_readlane(h, hi - 1);
                    }
                    hcarry = __builtin_amdgcn_readlane(h, 63);
                    fc = imax(fc - 64 * e_ins, __builtin_amdgcn_readlane(Pm, 63) - 63 * e_ins);
                    const unsigned long long nz = __ballot((hs | en) != 0);
                    const unsigned long long nzb = nz & actm, nze = nz & stm;
                    if (firstnz < 0 && nzb) firstnz = jb + __builtin_ctzll(nzb);
                    if (nze) lastnz = jb + 63 - __builtin_clzll(nze);
                }
            }
        }
        const int jfin = beg < end ? end : beg;
        if (jfin == qlen) {
            max_ie = gscore > h1 ? max_ie : i;
            gscore = gscore > h1 ? gscore : h1;
        }
        if (m == 0) break;
        const bool new_max = m > maxv;
        if (new_max) {
            maxv = m; max_i = i; max_j = mj;
            const int d = mj - i;
            max_off = imax(max_off, d < 0 ? -d : d);
        }
        if (zdrop_stop(cls, new_max, maxv, m, i - max_i, mj - max_j, e_del, e_ins, zdrop)) break;
        const int nb = firstnz >= 0 ? firstnz : end;
        const int jl = imax(lastnz, nb - 1);
        beg = nb;
        end = jl + 2 < qlen ? jl + 2 : qlen;
    }
    out.score = maxv; out.qle = max_j + 1; out.tle = max_i + 1; out.gtle = max_ie + 1;
    out.gscore = gscore; out.max_off = max_off;
    return cells;
}

// dispatch: registers when the query fits 4 chunks, a sliding register window when the band fits 8, the LDS ring otherwise
static __device__ __forceinline__ int bsw_extend(const uint8_t *__restrict__ qp, int qs, int qlen,
                                                 RefPtr tp, int ts, int tlen,
                                                 int w, int h0, const SwParams &P, int *RH, int *RE, int RM, SwOut &out) {
    const int nch = (qlen >> 6) + 1;
    if (nch == 1) return bsw_extend_reg<1>(qp, qs, qlen, tp, ts, tlen, w, h0, P, out);
    if (nch == 2) return bsw_extend_reg<2>(qp, qs, qlen, tp, ts, tlen, w, h0, P, out);
    if (nch == 3) return bsw_extend_reg<3>(qp, qs, qlen, tp, ts, tlen, w, h0, P, out);
    if (nch == 4) return bsw_extend_reg<4>(qp, qs, qlen, tp, ts, tlen, w, h0, P, out);
    const int nsl = ((2 * w + 1) >> 6) + 2;                          // chunks a sliding register window needs for this band
    if (nsl <= 5) return bsw_extend_slide<5>(qp, qs, qlen, tp, ts, tlen, w, h0, P, out);
    if (nsl <= 8) return bsw_extend_slide<8>(qp, qs, qlen, tp, ts, tlen, w, h0, P, out);
    return bsw_extend_wave(qp, qs, qlen, tp, ts, tlen, w, h0, P, RH, RE, RM, out);
}
