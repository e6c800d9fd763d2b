// matesw_dev.h -- device code of the mate-rescue SW (see matesw.hip), kept free of anything but per-lane C++ and four
// cross-lane primitives (row_shr1, row_any, row_xor, row_first), so that tools/emu can run the very same source on the host, one
// thread per lane with a rendezvous of the row at every cross-lane step (matesw_emu.cpp: one task at a time; the fake hip_runtime.h:
// the real launcher and bm2_sam_pe_dev), against the host oracle.
#pragma once
#include <stdint.h>
#include "../../include/bm2.h"
#include "refseq.h"

#if defined(BM2_EMU) || defined(BM2_EMU_ROW_PRIMS)      /* the includer supplies the four row primitives (tools/emu) */
#ifndef BM2_DEV
#define BM2_DEV inline
#endif
#else
#define BM2_DEV __device__ __forceinline__
static BM2_DEV int row_shr1(int v) {                            // lane k <- lane k-1 of the same 16-lane row, lane 0 <- 0
    return __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);
}
static BM2_DEV bool row_any(bool p) {                           // p on any lane of this row (the row executes this together)
    const unsigned long long b = __ballot(p);
    return ((b >> (threadIdx.x & 48)) & 0xffffull) != 0;
}
static BM2_DEV int row_xor(int v, int m) { return __shfl_xor(v, m, 16); }
static BM2_DEV int row_first(int v) { return __shfl(v, 0, 16); }
// the maximum over the row, in every lane of it: four rotations of the 16-lane DPP row (row_ror:8, 4, 2, 1) -- one VALU instruction each,
// where a butterfly of __shfl_xor costs a trip through the LDS crossbar per step (this runs once per target row of every task)
#define BM2_HAVE_ROW_MAX 1
template <int N> static BM2_DEV int row_ror(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x120 + N, 0xf, 0xf, false); }
static BM2_DEV int row_max(int v) {
    int o = row_ror<8>(v); v = v > o ? v : o;
    o = row_ror<4>(v); v = v > o ? v : o;
    o = row_ror<2>(v); v = v > o ? v : o;
    o = row_ror<1>(v); v = v > o ? v : o;
    return v;
}
#endif
#ifndef BM2_HAVE_ROW_MAX                                          /* (the emulator: from the butterfly it already has) */
static BM2_DEV int row_max(int v) { for (int m = 8; m >= 1; m >>= 1) { const int o = row_xor(v, m); v = v > o ? v : o; } return v; }
#endif

enum { KSW_XBYTE = 0x10000, KSW_XSTOP = 0x20000, KSW_XSUBO = 0x40000, KSW_XSTART = 0x80000 };

struct KswTask { int64_t q_off, t_off, b_off; int32_t qlen, tlen, xtra, pad; };
struct KswPrm { int8_t mat[25]; int8_t pad[3]; int32_t o_del, e_del, o_ins, e_ins, shift, maxsc; };
struct KswRes { int score, te, qe, score2, te2; };

static BM2_DEV int imx(int a, int b) { return a > b ? a : b; }
static BM2_DEV int imn(int a, int b) { return a < b ? a : b; }
// max(a - b, 0) for a, b >= 0: the saturating subtractions of the SSE2 kernels (psubusb / psubusw).  On the device ONE instruction
// (v_sub_u32 ... clamp) instead of a subtraction and a maximum: the kernel is bound by VALU issue (profiles/r03h_tail_kernels_pmc_sq.md)
#if defined(__HIP_DEVICE_COMPILE__)
static BM2_DEV int sub0(int a, int b) { return (int)__builtin_elementwise_sub_sat((unsigned)a, (unsigned)b); }
#else
static BM2_DEV int sub0(int a, int b) { return a > b ? a - b : 0; }
#endif

// One pass of the striped kernel over one task, executed by the 16 lanes of a row (lanes k >= P idle along).
//   rev = false: query[0, qlen) against target[0, tlen)
//   rev = true : the reversed query prefix [0, qe0] against the target whose first te0+1 bases are reversed (ksw.cpp:366-371)
template <int P>
static BM2_DEV KswRes ksw_pass(bool rev, const uint8_t *__restrict__ q, int qlen, int qe0, RefPtr t, int tlen, int te0,
                           const KswPrm &prm, const int8_t *smat, int minsc, int endsc, uint16_t *L, int slen_max, int k,
                           unsigned long long *blist) {
    constexpr bool U8 = P == 16;
    const bool on = k < P;
    const int slen = (qlen + P - 1) / P;
    const int shift = prm.shift, ed = prm.e_del, ei = prm.e_ins, oe_del = prm.o_del + prm.e_del, oe_ins = prm.o_ins + prm.e_ins;
    int16_t *prof = (int16_t *)L;                                // [5][slen][16]
    const int H0 = 5 * slen_max * 16, SZ = slen_max * 16;       // then H0, H1, E, Hmax: [slen][16] each
    int h0o = H0, h1o = H0 + SZ;
    const int eo = H0 + 2 * SZ, hmo = H0 + 3 * SZ;
    for (int j = 0; j < slen; ++j) {                             // ksw_qinit, ksw.cpp:62-109
        const int pos = j + k * slen;
        int qc = -1;
        if (on && pos < qlen) qc = rev ? q[qe0 - pos] : q[pos];
        for (int a = 0; a < 5; ++a) prof[(a * slen + j) * 16 + k] = (int16_t)((qc < 0 ? 0 : smat[a * 5 + qc]) + (U8 ? shift : 0));
        L[h0o + j * 16 + k] = 0; L[h1o + j * 16 + k] = 0; L[eo + j * 16 + k] = 0; L[hmo + j * 16 + k] = 0;
    }
    int te = -1, gmax = 0, nb = 0, last_sc = 0, last_pos = -2;
    auto target = [&](int i) -> int { return rev ? (i <= te0 ? t[te0 - i] : t[i]) : t[i]; };
    int tb_next = tlen > 0 ? target(0) : 0;
    for (int i = 0; i < tlen; ++i) {
        const int tb = tb_next;                                  // (requested a row ahead: the row's profile address hangs on it)
        if (i + 1 < tlen) tb_next = target(i + 1);
        const int16_t *S = prof + tb * slen * 16 + k;
        int h = row_shr1((int)L[h0o + (slen - 1) * 16 + k]), f = 0, mx = 0;
        for (int j = 0; j < slen; ++j) {
            const int s = S[j * 16];
            if (U8) h = sub0(imn(h + s, 255), shift);                                      // adds_epu8, subs_epu8
            else h = imx(imn(h + s, 32767), -32768);                                       // adds_epi16
            const int e = L[eo + j * 16 + k];
            h = imx(imx(h, e), f);
            mx = imx(mx, h);
            L[h1o + j * 16 + k] = (uint16_t)h;
            L[eo + j * 16 + k] = (uint16_t)imx(sub0(e, ed), sub0(h, oe_del));
            f = imx(sub0(f, ei), sub0(h, oe_ins));
            h = L[h0o + j * 16 + k];
        }
        bool done = false;                                       // lazy F: 16 rounds at most, as in both kernels
        for (int r = 0; r < 16 && !done; ++r) {
            f = row_shr1(f);
            for (int j = 0; j < slen; ++j) {
                const int hv = imx((int)L[h1o + j * 16 + k], f);
                L[h1o + j * 16 + k] = (uint16_t)hv;
                f = sub0(f, ei);
                if (!row_any(on && f > sub0(hv, oe_ins))) { done = true; break; }
            }
        }
        if (!on) mx = 0;
        const int imax = row_max(mx);
        if (imax >= minsc) {                                     // the list of local maxima for the second-best score, ksw.cpp:179-188
            if (nb == 0 || last_pos + 1 != i) { ++nb; last_sc = imax; last_pos = i; if (k == 0) blist[nb - 1] = (unsigned long long)imax << 32 | (unsigned)i; }
            else if (last_sc < imax) { last_sc = imax; last_pos = i; if (k == 0) blist[nb - 1] = (unsigned long long)imax << 32 | (unsigned)i; }
        }
        if (imax > gmax) {
            gmax = imax; te = i;
            for (int j = 0; j < slen; ++j) L[hmo + j * 16 + k] = L[h1o + j * 16 + k];
            if (U8 ? (gmax + shift >= 255 || gmax >= endsc) : (gmax >= endsc)) break;
        }
        const int sw = h0o; h0o = h1o; h1o = sw;
    }
    KswRes r;
    r.score = U8 ? (gmax + shift < 255 ? gmax : 255) : gmax;
    r.te = te; r.qe = -1; r.score2 = -1; r.te2 = -1;
    if (!U8 || r.score != 255) {
        int bv = -2, bp = 0x7fffffff;
        if (on) {
            bv = -1;
            for (int j = 0; j < slen; ++j) { const int v = L[hmo + j * 16 + k]; if (v > bv) { bv = v; bp = j + k * slen; } }
        }
        for (int m = 8; m >= 1; m >>= 1) {                       // largest value, smallest query position among equals
            const int ov = row_xor(bv, m), op = row_xor(bp, m);
            if (ov > bv || (ov == bv && op < bp)) { bv = ov; bp = op; }
        }
        r.qe = bv < 0 ? -1 : bp;                                 // an empty query has no end
        if (nb > 0) {
            int s2 = -1, t2 = -1;
            if (k == 0) {
                const int d = (r.score + prm.maxsc - 1) / prm.maxsc, low = te - d, high = te + d;
                for (int x = 0; x < nb; ++x) {
                    const unsigned long long v = blist[x];
                    const int e = (int)(unsigned)v, sc = (int)(v >> 32);
                    if ((e < low || e > high) && sc > s2) { s2 = sc; t2 = e; }
                }
            }
            r.score2 = row_first(s2); r.te2 = row_first(t2);
        }
    }
    return r;
}

// ---- the byte kernel with its rows in REGISTERS (round 6) ---------------------------------------------------------------------------------------
// A 150-base mate is ten stripe segments: H, E and the H row of the best score are thirty registers of a lane, the query ten selector bytes.  The pass above
// spends five LDS accesses, their addresses and a loop test on every cell beside its twelve VALU instructions (the kernel is bound by VALU ISSUE at three
// wavefronts per SIMD: profiles/r03h_tail_kernels_pmc_sq.md); here the segment loop is unrolled over SL segments (a lane's segments beyond its task's slen
// are masked), H is updated in place (the old value is the next segment's diagonal), the row's five scores are bytes of a register pair and ONE byte
// permute turns four query codes into their four scores.  Same operations on the same values in the same order -- the lazy-F rounds, their early exit and
// the saturations are what the striped SSE2 kernel's results depend on (matesw.hip) -- so the results are the LDS pass's, checked by tools/emu
// (tests/test_matesw_emu.py) and on the GPU against the reference's ksw_align2 (tests/test_zz_tail_kernels_gpu.py).
#if defined(__HIP_DEVICE_COMPILE__)
static BM2_DEV uint32_t byte_perm(uint32_t hi, uint32_t lo, uint32_t sel) { return __builtin_amdgcn_perm(hi, lo, sel); }
#else
static BM2_DEV uint32_t byte_perm(uint32_t hi, uint32_t lo, uint32_t sel) {      // v_perm_b32 for selectors 0..7
    const uint64_t v = (uint64_t)hi << 32 | lo;
    uint32_t r = 0;
    for (int b = 0; b < 4; ++b) r |= (uint32_t)((v >> (8 * ((sel >> (8 * b)) & 7))) & 0xff) << (8 * b);
    return r;
}
#endif
// the row's scores against the query codes 0..3 (lo) and 4 / "beyond the query" (hi bytes 0 / 1), each plus the bias: ksw_qinit's profile entries
static BM2_DEV void ksw_score_pair(const KswPrm &prm, const int8_t *smat, int tb, uint32_t &lo, uint32_t &hi) {
    const int sh = prm.shift;
    lo = (uint32_t)((smat[tb * 5 + 0] + sh) & 0xff) | (uint32_t)((smat[tb * 5 + 1] + sh) & 0xff) << 8 | (uint32_t)((smat[tb * 5 + 2] + sh) & 0xff) << 16 |
         (uint32_t)((smat[tb * 5 + 3] + sh) & 0xff) << 24;
    hi = (uint32_t)((smat[tb * 5 + 4] + sh) & 0xff) | (uint32_t)(sh & 0xff) << 8;
}
template <int SL>
static BM2_DEV KswRes ksw_pass_u8r(bool rev, const uint8_t *__restrict__ q, int qlen, int qe0, RefPtr t, int tlen, int te0,
                                   const KswPrm &prm, const int8_t *smat, int minsc, int endsc, int k, unsigned long long *blist) {
    constexpr int NQ = (SL + 3) / 4;
    const int slen = (qlen + 15) / 16;                           // <= SL (the caller's test)
    const int shift = prm.shift, ed = prm.e_del, ei = prm.e_ins, oe_del = prm.o_del + prm.e_del, oe_ins = prm.o_ins + prm.e_ins;
    uint32_t QS[NQ];
    int H[SL], E[SL], HM[SL];
#pragma unroll
    for (int w = 0; w < NQ; ++w) QS[w] = 0x05050505u;            // 5: no query base here (profile entry 0 + bias)
#pragma unroll
    for (int j = 0; j < SL; ++j) {
        H[j] = 0; E[j] = 0; HM[j] = 0;
        const int pos = j + k * slen;
        if (j < slen && pos < qlen) {
            const uint32_t qc = rev ? q[qe0 - pos] : q[pos];
            QS[j >> 2] = (QS[j >> 2] & ~(0xffu << (8 * (j & 3)))) | (qc > 4u ? 4u : qc) << (8 * (j & 3));
        }
    }
    int te = -1, gmax = 0, nb = 0, last_sc = 0, last_pos = -2;
    auto target = [&](int i) -> int { return rev ? (i <= te0 ? t[te0 - i] : t[i]) : t[i]; };
    int tb_next = tlen > 0 ? target(0) : 0;
    for (int i = 0; i < tlen; ++i) {
        const int tb = tb_next;
        if (i + 1 < tlen) tb_next = target(i + 1);
        uint32_t t_lo, t_hi;
        ksw_score_pair(prm, smat, tb, t_lo, t_hi);
        int hl = H[0];                                           // H of the last segment of the previous row
#pragma unroll
        for (int j = 1; j < SL; ++j) hl = j == slen - 1 ? H[j] : hl;
        int h = row_shr1(hl), f = 0, mx = 0;
        uint32_t sc4 = 0;
#pragma unroll
        for (int j = 0; j < SL; ++j) {
            if ((j & 3) == 0) sc4 = byte_perm(t_hi, t_lo, QS[j >> 2]);
            if (j < slen) {
                const int s = (int)((sc4 >> (8 * (j & 3))) & 0xffu);
                h = sub0(imn(h + s, 255), shift);                // adds_epu8, subs_epu8
                const int e = E[j];
                h = imx(imx(h, e), f);
                mx = imx(mx, h);
                const int hd = H[j];                             // (the next segment's diagonal)
                H[j] = h;
                E[j] = imx(sub0(e, ed), sub0(h, oe_del));
                f = imx(sub0(f, ei), sub0(h, oe_ins));
                h = hd;
            }
        }
        bool done = false;                                       // lazy F: 16 rounds at most, as in both kernels
        for (int r = 0; r < 16 && !done; ++r) {
            f = row_shr1(f);
#pragma unroll
            for (int j = 0; j < SL; ++j) {
                if (!done && j < slen) {
                    const int hv = imx(H[j], f);
                    H[j] = hv;
                    f = sub0(f, ei);
                    if (!row_any(f > sub0(hv, oe_ins))) done = true;
                }
            }
        }
        const int imax = row_max(mx);
        if (imax >= minsc) {                                     // the list of local maxima for the second-best score, ksw.cpp:179-188
            if (nb == 0 || last_pos + 1 != i) { ++nb; last_sc = imax; last_pos = i; if (k == 0) blist[nb - 1] = (unsigned long long)imax << 32 | (unsigned)i; }
            else if (last_sc < imax) { last_sc = imax; last_pos = i; if (k == 0) blist[nb - 1] = (unsigned long long)imax << 32 | (unsigned)i; }
        }
        if (imax > gmax) {
            gmax = imax; te = i;
#pragma unroll
            for (int j = 0; j < SL; ++j) HM[j] = H[j];
            if (gmax + shift >= 255 || gmax >= endsc) break;
        }
    }
    KswRes r;
    r.score = gmax + shift < 255 ? gmax : 255;
    r.te = te; r.qe = -1; r.score2 = -1; r.te2 = -1;
    if (r.score != 255) {
        int bv = -1, bp = 0x7fffffff;
#pragma unroll
        for (int j = 0; j < SL; ++j) if (j < slen && HM[j] > bv) { bv = HM[j]; bp = j + k * slen; }
        for (int m = 8; m >= 1; m >>= 1) {                       // largest value, smallest query position among equals
            const int ov = row_xor(bv, m), op = row_xor(bp, m);
            if (ov > bv || (ov == bv && op < bp)) { bv = ov; bp = op; }
        }
        r.qe = bv < 0 ? -1 : bp;                                 // an empty query has no end
        if (nb > 0) {
            int s2 = -1, t2 = -1;
            if (k == 0) {
                const int d = (r.score + prm.maxsc - 1) / prm.maxsc, low = te - d, high = te + d;
                for (int x = 0; x < nb; ++x) {
                    const unsigned long long v = blist[x];
                    const int e = (int)(unsigned)v, sc = (int)(v >> 32);
                    if ((e < low || e > high) && sc > s2) { s2 = sc; t2 = e; }
                }
            }
            r.score2 = row_first(s2); r.te2 = row_first(t2);
        }
    }
    return r;
}
#define KSW_REG_SL 10                 // segments the register pass holds: byte-kernel tasks of up to 160 query bases
#if defined(__HIPCC__) && !defined(BM2_EMU) && !defined(BM2_EMU_ROW_PRIMS)
__host__
#endif
static BM2_DEV bool ksw_task_fits_regs(const KswTask &T) { return (T.xtra & KSW_XBYTE) != 0 && (T.qlen + 15) / 16 <= KSW_REG_SL; }
// ksw_align2 of a task that fits (ksw_task_fits_regs), both passes on registers; no LDS beyond the score matrix
static BM2_DEV void ksw_row_task_reg(const uint8_t *__restrict__ qbase, RefPtr tbase, const KswTask &T, const KswPrm &prm, const int8_t *smat, int k,
                                     unsigned long long *bl, bm2_ksw_result *out) {
    const uint8_t *q = qbase + T.q_off;
    const RefPtr t = tbase + T.t_off;
    const int minsc = (T.xtra & KSW_XSUBO) ? T.xtra & 0xffff : 0x10000, endsc = (T.xtra & KSW_XSTOP) ? T.xtra & 0xffff : 0x10000;
    KswRes r = ksw_pass_u8r<KSW_REG_SL>(false, q, T.qlen, 0, t, T.tlen, 0, prm, smat, minsc, endsc, k, bl);
    int tb = -1, qb = -1;
    if ((T.xtra & KSW_XSTART) && !((T.xtra & KSW_XSUBO) && r.score < (T.xtra & 0xffff)) && r.qe >= 0) {
        const KswRes rr = ksw_pass_u8r<KSW_REG_SL>(true, q, r.qe + 1, r.qe, t, T.tlen, r.te, prm, smat, 0x10000, r.score, k, bl);
        if (r.score == rr.score) { tb = r.te - rr.te; qb = r.qe - rr.qe; }
    }
    if (k == 0) {
        bm2_ksw_result o;
        o.score = r.score; o.te = r.te; o.qe = r.qe; o.score2 = r.score2; o.te2 = r.te2; o.tb = tb; o.qb = qb;
        *out = o;
    }
}

// One task on the 16 lanes of a row: ksw_align2, ksw.cpp:340-381.  L = this row's LDS area (9 * slen_max * 16 halfwords).
static BM2_DEV void ksw_row_task(const uint8_t *__restrict__ qbase, RefPtr tbase, const KswTask &T, const KswPrm &prm, const int8_t *smat, uint16_t *L,
                                 int slen_max, int k, unsigned long long *bl, bm2_ksw_result *out) {
    const uint8_t *q = qbase + T.q_off;
    const RefPtr t = tbase + T.t_off;
    const bool byte = (T.xtra & KSW_XBYTE) != 0;
    const int minsc = (T.xtra & KSW_XSUBO) ? T.xtra & 0xffff : 0x10000, endsc = (T.xtra & KSW_XSTOP) ? T.xtra & 0xffff : 0x10000;
    KswRes r = byte ? ksw_pass<16>(false, q, T.qlen, 0, t, T.tlen, 0, prm, smat, minsc, endsc, L, slen_max, k, bl)
                    : ksw_pass<8>(false, q, T.qlen, 0, t, T.tlen, 0, prm, smat, minsc, endsc, L, slen_max, k, bl);
    int tb = -1, qb = -1;
    if ((T.xtra & KSW_XSTART) && !((T.xtra & KSW_XSUBO) && r.score < (T.xtra & 0xffff)) && r.qe >= 0) {
        const KswRes rr = byte ? ksw_pass<16>(true, q, r.qe + 1, r.qe, t, T.tlen, r.te, prm, smat, 0x10000, r.score, L, slen_max, k, bl)
                               : ksw_pass<8>(true, q, r.qe + 1, r.qe, t, T.tlen, r.te, prm, smat, 0x10000, r.score, L, slen_max, k, bl);
        if (r.score == rr.score) { tb = r.te - rr.te; qb = r.qe - rr.qe; }
    }
    if (k == 0) {
        bm2_ksw_result o;
        o.score = r.score; o.te = r.te; o.qe = r.qe; o.score2 = r.score2; o.te2 = r.te2; o.tb = tb; o.qb = qb;
        *out = o;
    }
}
