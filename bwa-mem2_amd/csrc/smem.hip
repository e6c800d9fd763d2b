// smem.hip -- SMEM seeding over the FM-index (the three passes of mem_collect_smem, bwamem.cpp:626-803) and the
// suffix-array lookup (get_sa_entries_prefetch, FMI_search.cpp:1257-1375).
//
// Reference shape: one thread walks 512 reads round-robin, one backwardExt at a time, prefetching the next CP_OCC
// blocks (FMI_search.cpp:693-721).  Here the passes are cut into TASK KERNELS with persistent lanes (see below: forward
// walks per read, backward phases per start position), every backwardExt is QUAD-COOPERATIVE (the four lanes of a quad
// fetch one 64-byte CP_OCC entry with one coalesced request, 16 bytes each, on the per-base layout CpOccDev), and the
// dependent chain of random 64-byte loads is hidden by tens of thousands of lanes in flight instead of software
// prefetch.  Memory-bound on random 64-B HBM transactions; no LDS reuse exists between reads (the index is ~10^8 lines).
#include "bm2_ctx.h"
#include "pipeline.h"
#include "ksort_dev.h"

// (forcing 8 waves/SIMD with __launch_bounds__(256, 8) spills 88 B/lane and measured 25 % slower: left at the natural 84 VGPRs)

struct Bi { int64_t k, l, s; };

static __device__ __forceinline__ int64_t pick4(int a, int64_t c0, int64_t c1, int64_t c2, int64_t c3) {
    return a == 0 ? c0 : a == 1 ? c1 : a == 2 ? c2 : c3;
}

struct Blk { uint64_t w[8]; };   // cp_count[0..3], bwt[0..3]
static __device__ __forceinline__ Blk load_blk(const CpOccDev *p) {              // (whole entry, de-interleaved: k_sal)
    const ulonglong2 *q = (const ulonglong2 *)p;
    ulonglong2 a = q[0], b = q[1], c = q[2], d = q[3];
    Blk r; r.w[0] = a.x; r.w[4] = a.y; r.w[1] = b.x; r.w[5] = b.y; r.w[2] = c.x; r.w[6] = c.y; r.w[3] = d.x; r.w[7] = d.y;
    return r;
}

// quad helpers (DPP quad_perm; every lane of the quad must be active)
template <int CTRL> static __device__ __forceinline__ int32_t qperm32(int32_t v) {
    return __builtin_amdgcn_mov_dpp(v, CTRL, 0xf, 0xf, false);
}
typedef int32_t bm2_i32x2 __attribute__((ext_vector_type(2)));
template <int CTRL> static __device__ __forceinline__ int64_t qperm64(int64_t v) {
    const bm2_i32x2 r = { qperm32<CTRL>((int32_t)(uint32_t)v), qperm32<CTRL>((int32_t)(v >> 32)) };
    return __builtin_bit_cast(int64_t, r);           // (a register pair: shifting and or-ing the halves together costs two 64-bit adds)
}
template <int T> static __device__ __forceinline__ int64_t qbcast64(int64_t v) { return qperm64<T | T << 2 | T << 4 | T << 6>(v); }
template <int T> static __device__ __forceinline__ int32_t qbcast32(int32_t v) { return qperm32<T | T << 2 | T << 4 | T << 6>(v); }
static __device__ __forceinline__ int64_t qsum64(int64_t v) {                     // sum over the quad, in every lane
    v += qperm64<0xB1>(v);           // [1,0,3,2]
    v += qperm64<0x4E>(v);           // [2,3,0,1]
    return v;
}
static __device__ __forceinline__ uint32_t qsum32(uint32_t v) {                  // sum over the quad, in every lane
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int32_t)v, 0xB1, 0xf, 0xf, true);        // [1,0,3,2]  (bound_ctrl: lets the exchange fold into the add)
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int32_t)v, 0x4E, 0xf, 0xf, true);        // [2,3,0,1]
    return v;
}

// FMI_search::backwardExt (FMI_search.cpp:1025-1052) with GET_OCC (FMI_search.h:66-73), QUAD-COOPERATIVE: the wave
// must be converged; lanes with want == false take part in the exchange and get garbage back.
// For each of the four lanes t of a quad in turn, the quad fetches the two CP_OCC entries lane t needs -- lane b loads
// quarter b -- so that one load instruction touches 16 lines (and 16 pages) per wave, not 64: with one entry per lane
// the address-translation rate caps the kernel at ~22 G lines/s once the index exceeds ~3 GB; fetched by quads the same
// hardware delivers ~50 G lines/s (tools/ubench/randline.hip).  Lane b then ranks base b at both ends of the interval,
// and four 32-bit quad sums (coop_rank) hand lane t what it needs: occ(a, k), the size d[a], and the sizes of the bases above a.
template <int T>
static __device__ __forceinline__ void coop_issue(const DevIndex &ix, int64_t k, int64_t s, int want, int sub, ulonglong2 &e1, ulonglong2 &e2) {
    const int64_t sp = qbcast64<T>(k), ep = sp + qbcast64<T>(s);
#if defined(__HIP_DEVICE_COMPILE__)
    // (a quad that loads nothing ranks whatever the registers hold and nobody reads the result: "defined" without 32 v_mov per call)
    asm volatile("" : "=v"(e1.x), "=v"(e1.y), "=v"(e2.x), "=v"(e2.y));
#else
    e1 = make_ulonglong2(0, 0); e2 = e1;
#endif
    if (qbcast32<T>(want)) {         // (a quad whose lane T has nothing pending loads nothing: idle lanes must not all hit one line)
        e1 = ((const ulonglong2 *)&ix.cp_occ[sp >> 6])[sub];
        e2 = ((const ulonglong2 *)&ix.cp_occ[ep >> 6])[sub];
        // (the second request also when both ends lie in one block, 4 of 10 calls: skipping it under a per-quad branch measured 18 % SLOWER,
        //  profiles/r03i_bench.json vs bench_sb; with non-temporal loads -- lines nobody touches again, +8 % in the random-line micro-benchmark --
        //  the kernel is 17 % SLOWER: the first steps of every walk hit the same few thousand lines, profiles/r03v_sweep_lane_variants.json)
    }
}
// What lane T gets back travels as four 32-bit quad sums (a 64-bit sum is three instructions per step, a 32-bit one can fold its
// exchange into the add): occ and size of base a -- one lane contributes, so the halves add without carries -- and the sizes above a
// as a 31-bit part (three of them stay below 2^32) and a high part, the three high parts sharing one word.  All counts are below 2^40.
struct QuadOut { uint32_t xl, yl, zl, hw; };
template <int T>
static __device__ __forceinline__ void coop_rank(int64_t k, int64_t s, int a, int sub, const ulonglong2 e1, const ulonglong2 e2, QuadOut &out) {
    const int32_t spl = qbcast32<T>((int32_t)k), epl = spl + qbcast32<T>((int32_t)s);
    const int at = qbcast32<T>(a);
    // the top y bits of the word (one_hot_mask_array[y], y = position in the block): (w >> 1) >> (63 - y) is 0 for y = 0 without a select
    const int y1 = spl & 63, y2 = epl & 63;
    const int64_t o1 = (int64_t)e1.x + __popcll((e1.y >> 1) >> (y1 ^ 63));
    const int64_t d = (int64_t)e2.x + __popcll((e2.y >> 1) >> (y2 ^ 63)) - o1;
    const bool is_at = sub == at, above = sub > at;
    const uint32_t xl = qsum32(is_at ? (uint32_t)o1 : 0u), yl = qsum32(is_at ? (uint32_t)d : 0u);
    const uint32_t zl = qsum32(above ? (uint32_t)d & 0x7fffffffu : 0u);
    const uint32_t hw = qsum32(is_at ? (uint32_t)(o1 >> 32) | (uint32_t)(d >> 32) << 8 : above ? (uint32_t)(d >> 31) << 16 : 0u);
    if (sub == T) { out.xl = xl; out.yl = yl; out.zl = zl; out.hw = hw; }
}
static __device__ __forceinline__ Bi ext_result(const DevIndex &ix, const Bi &in, int64_t k, int64_t s, int a, const QuadOut &q) {
    const int64_t X = (int64_t)q.xl | (int64_t)(q.hw & 0xffu) << 32, Y = (int64_t)q.yl | (int64_t)((q.hw >> 8) & 0xffu) << 32;
    const int64_t Z = (int64_t)q.zl + ((int64_t)(q.hw >> 16) << 31);
    const int64_t sent = (k <= ix.sentinel_index && k + s > ix.sentinel_index) ? 1 : 0;
    Bi out;
    out.k = pick4(a, ix.count[0], ix.count[1], ix.count[2], ix.count[3]) + X;
    out.l = in.l + sent + Z;
    out.s = Y;
    return out;
}
static __device__ __forceinline__ Bi backward_ext(const DevIndex &ix, Bi in, int a, bool want) {
    const int sub = (int)(threadIdx.x & 3);
    const int64_t k = want ? in.k : 0, s = want ? in.s : 0;
    ulonglong2 e1[4], e2[4];
    const int w = want ? 1 : 0;
    coop_issue<0>(ix, k, s, w, sub, e1[0], e2[0]); coop_issue<1>(ix, k, s, w, sub, e1[1], e2[1]);
    coop_issue<2>(ix, k, s, w, sub, e1[2], e2[2]); coop_issue<3>(ix, k, s, w, sub, e1[3], e2[3]);
    QuadOut q = { 0, 0, 0, 0 };
    coop_rank<0>(k, s, a, sub, e1[0], e2[0], q); coop_rank<1>(k, s, a, sub, e1[1], e2[1], q);
    coop_rank<2>(k, s, a, sub, e1[2], e2[2], q); coop_rank<3>(k, s, a, sub, e1[3], e2[3], q);
    return ext_result(ix, in, k, s, a, q);
}
static __device__ __forceinline__ Bi init_bi(const DevIndex &ix, int a) {     // FMI_search.cpp:531-533
    Bi b;
    b.k = pick4(a, ix.count[0], ix.count[1], ix.count[2], ix.count[3]);
    b.l = pick4(3 - a, ix.count[0], ix.count[1], ix.count[2], ix.count[3]);
    b.s = pick4(a, ix.count[1], ix.count[2], ix.count[3], ix.count[4]) - b.k;
    return b;
}

// ------------------------------------------------------------------------------------------------------------------
// Seeding as TASK KERNELS.  The order in which mem_collect_smem finds the SMEMs of a read does not matter -- the array is
// sorted by (rid, m, n) afterwards (bwamem.cpp:785-799) and equal (m, n) are field-identical -- so the work is cut where
// the data dependences are, not where the reference's loops are:
//   * the forward walk of getSMEMsOnePosOneThread (FMI_search.cpp:537-575) alone decides the next start position, so
//     pass 1 is a chain of forward walks per read (k_walk<P1>); each walk leaves its candidate list (prev[]) as a TASK;
//   * the backward phase (:596-665) of a start position depends only on that list (k_bwd, one task per lane); every
//     SMEM it emits that is long and rare enough spawns a pass-2 forward walk (bwamem.cpp:695-753: k_walk<P2>, then
//     k_bwd again);
//   * pass 3 (bwtSeedStrategyAllPosOneThread, FMI_search.cpp:740-810) is a forward walk with a different stop rule
//     (k_walk<P3>) and independent of the other two.
// Every kernel is one CONVERGED LOOP over homogeneous lanes: each trip, every lane issues the two CP_OCC line loads of
// its pending backwardExt at the same program point, so a wave keeps up to 128 independent HBM lines in flight and the
// trip counts of different reads/walks never serialise.  Lanes pull work items from a cursor; anything a lane needs
// from global memory between two extensions is requested one step ahead (next query window, next candidate, next work
// item, next slot/chunk id -- the atomics return into registers nobody reads for many trips) or the lane YIELDS: it sits
// out one extension round while the load returns behind the other lanes' CP_OCC loads.  No lane ever stalls the wave.
#define CAPF 32                   // candidate-list entries stored in a task slot (longer lists continue in the pool)

enum { W_P1 = 1, W_P2 = 2, W_P3 = 3 };
enum { SC_P1_ITEM = 0, SC_SLOT1, SC_B1_ITEM, SC_REC, SC_TASK, SC_P2_ITEM, SC_SLOT2, SC_B2_ITEM, SC_P3_ITEM, SC_NEXT, SC_OVF_FLAG,
       SC_POOL, SC_NEXT_W1, SC_NEXT_W2, SC_NEXT_W3, SC_NEXT_B1, SC_NEXT_B2, SC_HEAVY1, SC_HEAVY2, SC_H1_ITEM, SC_H2_ITEM,
       SC_CONT1, SC_CONT2, SC_C1_ITEM, SC_C2_ITEM, SC_CROWS1, SC_CROWS2, SC_N };   // SC_CONT*: backward tasks k_bwd handed over at a row boundary (SeedBufs::cont1 / cont2), SC_CROWS*: rows k_bwd_cont walked   // SC_NEXT_*: backwardExt calls per kernel           // cursors / counters of the seeding kernels (unsigned long long each)
enum { OVF_SLOT1 = 1, OVF_SLOT2 = 2, OVF_REC = 4, OVF_TASK = 8, OVF_POOL = 16 };

struct __attribute__((aligned(16))) BHead {       // header of a backward-phase task (32 bytes)
    int64_t rd_off;               // offset of the read in enc
    int32_t r, L;
    uint32_t x_np;                // start position | list length << 16
    uint32_t mi_pass;             // min_intv (<= 65535) | pass << 16
    int32_t pool_id;              // continuation of the list beyond CAPF entries, -1 = none
    int32_t pad;
};
struct __attribute__((aligned(16))) P2Task {      // pass-2 forward walk (32 bytes)
    int64_t rd_off;
    int32_t r, L, x, s;
    int64_t pad;
};
struct __attribute__((aligned(16))) CTask {       // a backward task k_bwd handed over at a row boundary (48 bytes, read by k_bwd_cont as three 16-byte words): its header as the continuation reads it + its slot
    BHead h;                      // x_np: the row just finished | the survivors' count << 16; pad: index of the list's first entry + 1
    int32_t slot, pad0, pad1, pad2;
};
#define CCAP 64                   // survivors a handed-over task may have (k_bwd_cont keeps a task's row in 64 LDS entries)
// Ids (work items, task slots, record / task indices) are handed out from per-wave pools: the lanes that need one at
// the same point take consecutive ids from the wave's pool with a ballot; a pool is refilled with ONE atomic on the
// global cursor per BATCH ids (a same-address atomic per lane and id would serialise in L2: ~10 ns each, measured).
// Pool state lives in LDS (one [pos, end) pair per wave and allocator), because the lanes calling are a divergent subset.
struct WavePool { volatile int64_t pos, end; };
typedef __attribute__((address_space(3))) WavePool LdsPool;       // (explicitly LDS: a generic pointer would make these flat accesses)
template <int BATCH>
static __device__ __forceinline__ int64_t wave_alloc(LdsPool *wp, unsigned long long *cursor) {
    const unsigned long long mask = __ballot(1);
    const int lt = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
    const int cnt = __popcll(mask);
    const int64_t pos = wp->pos, rem = wp->end - pos;
    if (cnt <= rem) {
        if (lt == 0) wp->pos = pos + cnt;
        return pos + lt;
    }
    unsigned long long nb = 0;
    if (lt == 0) nb = atomicAdd(cursor, (unsigned long long)BATCH);
    const int leader = __ffsll((long long)mask) - 1;
    const unsigned lo = __builtin_amdgcn_readlane((unsigned)nb, leader), hi = __builtin_amdgcn_readlane((unsigned)(nb >> 32), leader);
    const int64_t nbase = (int64_t)(((unsigned long long)hi << 32) | lo);
    if (lt == 0) { wp->pos = nbase + (cnt - rem); wp->end = nbase + BATCH; }
    return lt < rem ? pos + lt : nbase + (lt - rem);
}
// The same for ids that are drawn RARELY (a handed-over task: a few per wavefront and launch): exactly as many as lanes ask, one atomic per call
// that the wavefront waits for -- a pool would leave most of every batch unused, and every unused id is an item the consumer has to skip.
static __device__ __forceinline__ int64_t wave_alloc_exact(LdsPool *, unsigned long long *cursor) {
    const unsigned long long mask = __ballot(1);
    const int lt = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
    unsigned long long nb = 0;
    if (lt == 0) nb = atomicAdd(cursor, (unsigned long long)__popcll(mask));
    const int leader = __ffsll((long long)mask) - 1;
    const unsigned lo = __builtin_amdgcn_readlane((unsigned)nb, leader), hi = __builtin_amdgcn_readlane((unsigned)(nb >> 32), leader);
    return (int64_t)(((unsigned long long)hi << 32) | lo) + lt;
}
#define LCAP 12                   // survivors of a backward row kept in LDS per lane (48 KB per 256-thread block)
#define HEAVY_T 40                // backward tasks with longer candidate lists go to the wave-per-task kernel ...
#define HCAP 192                  // ... if the list fits its LDS row (else they stay lane-per-task); 192: a workgroup of k_bwd_heavy (13 KB) fits beside three of k_bwd (3 x 48.5 KB of 160)
#define HEAVY_BATCH 64
#define BWD_EXPORT_AGE 256             // (default of BM2_BWD_EXPORT_AGE)
#define ITEM_BATCH 64
#define SLOT_BATCH 256
#define REC_BATCH 256
#define TASK_BATCH 64

static __device__ __forceinline__ uint64_t load8(const uint8_t *p) { uint64_t w; __builtin_memcpy(&w, p, 8); return w; }

static __device__ __forceinline__ uint4 pv_pack(int64_t k, int64_t l, int64_t s, int n) {
    uint4 v;
    v.x = (uint32_t)k; v.y = (uint32_t)l; v.z = (uint32_t)s;
    v.w = (uint32_t)((k >> 32) & 31) | (uint32_t)((l >> 32) & 31) << 5 | (uint32_t)((s >> 32) & 31) << 10 | (uint32_t)n << 16;
    return v;
}
static __device__ __forceinline__ void pv_unpack(const uint4 v, int64_t &k, int64_t &l, int64_t &s, int &n) {
    k = (int64_t)v.x | (int64_t)(v.w & 31) << 32;
    l = (int64_t)v.y | (int64_t)((v.w >> 5) & 31) << 32;
    s = (int64_t)v.z | (int64_t)((v.w >> 10) & 31) << 32;
    n = (int)(v.w >> 16);
}

// 8-base register window on the query with the next window (in walking direction) prefetched
struct QWin {
    uint64_t cur, nxt; int curb, nxtb;
    __device__ __forceinline__ void start(const uint8_t *q, int pos, int dir) {
        curb = dir > 0 ? pos : (pos > 7 ? pos - 7 : 0);
        cur = load8(q + curb);
        nxtb = dir > 0 ? curb + 8 : (curb > 8 ? curb - 8 : 0);
        nxt = load8(q + nxtb);
    }
    // base `pos`; false = the window was (re)loaded, look again after yielding
    __device__ __forceinline__ bool get(const uint8_t *q, int pos, int dir, int &base) {
        unsigned d = (unsigned)(pos - curb);
        if (d < 8u) { base = (int)((cur >> (8 * d)) & 0xff); return true; }
        d = (unsigned)(pos - nxtb);
        if (d < 8u) {
            cur = nxt; curb = nxtb;
            nxtb = dir > 0 ? curb + 8 : (curb > 8 ? curb - 8 : 0);
            nxt = load8(q + nxtb);
            base = (int)((cur >> (8 * d)) & 0xff);
            return true;
        }
        start(q, pos, dir);
        return false;
    }
};

// ---- forward walks ----------------------------------------------------------------------------------------------
// MODE W_P1: item = read; chain of start positions, every walk leaves a backward task.  W_P2: item = P2Task, one walk.
// W_P3: item = read; forward-only seeding, SMEM records written directly.
enum { F_EXT = 0, F_NEWITEM, F_START, F_NEWPOS, F_CHK, F_END, F_DONE };

template <int MODE>
__global__ void __launch_bounds__(256)
k_walk(DevIndex ix, SeedParams sp, int n_reads, const uint8_t *__restrict__ enc, const int64_t *__restrict__ off,
       const int32_t *__restrict__ len, const P2Task *__restrict__ tasks, int64_t task_cap,
       BHead *__restrict__ heads, uint4 *__restrict__ ents, int64_t slot_cap, uint4 *__restrict__ pool, int pool_cap, int pool_slots,
       bm2_smem_t *__restrict__ recs, int64_t rec_cap, int32_t *__restrict__ smem_cnt, unsigned long long *sc,
       int32_t *__restrict__ heavy_ids, int64_t heavy_cap) {
    int64_t n_ext = 0;
    unsigned ovf = 0;
    int64_t n_items;
    if (MODE == W_P2) { n_items = (int64_t)sc[SC_TASK]; if (n_items > task_cap) n_items = task_cap; }
    else n_items = n_reads;
    __shared__ WavePool pools[4][3];                           // per wave: [0] work items, [1] task slots or records, [2] heavy-task ids
    LdsPool *ip = (LdsPool *)&pools[threadIdx.x >> 6][0], *op = (LdsPool *)&pools[threadIdx.x >> 6][1], *hp = (LdsPool *)&pools[threadIdx.x >> 6][2];
    if ((threadIdx.x & 63) == 0) { ip->pos = ip->end = 0; op->pos = op->end = 0; hp->pos = hp->end = 0; }
    unsigned long long *item_cur = sc + (MODE == W_P1 ? SC_P1_ITEM : MODE == W_P2 ? SC_P2_ITEM : SC_P3_ITEM);
    unsigned long long *out_cur = sc + (MODE == W_P1 ? SC_SLOT1 : MODE == W_P2 ? SC_SLOT2 : SC_REC);
    // work items: it_a = the next item of this lane; its payload is already loaded
    int64_t it_a = wave_alloc<ITEM_BATCH>(ip, item_cur);
    int64_t pl_off = 0; int pl_len = 0; P2Task pl_t = {};
    if (it_a < n_items) { if (MODE == W_P2) pl_t = tasks[it_a]; else { pl_off = off[it_a]; pl_len = len[it_a]; } }

#ifdef BM2_SMEM_PROF
    unsigned long long prof_rounds = 0, prof_active = 0;
#endif
    int state = F_NEWITEM;
    int32_t r = 0; int64_t rd_off = 0; const uint8_t *q = enc; int L = 0;
    int x = 0, next_x = 0, j = 0, a = 0, n_prev = 0, pool_id = -1;
    int64_t min_intv = 1, slot = 0;
    int64_t smk = 0, sml = 0, sms = 0; int smn = 0;
    int64_t eik = 0, eil = 0, eis = 0; int ea = 0;
    QWin w; w.cur = w.nxt = 0; w.curb = w.nxtb = -64;

    auto push = [&](const uint4 v) {                          // prev[n_prev++] = sm
        if (slot < slot_cap) {
            if (n_prev < CAPF) ents[slot * CAPF + n_prev] = v;
            else {
                if (pool_id < 0) pool_id = (int)atomicAdd(&sc[SC_POOL], 1ULL);      // (a wait; lists this long are rare)
                if (pool_id < pool_slots && n_prev - CAPF < pool_cap) pool[(int64_t)pool_id * pool_cap + (n_prev - CAPF)] = v;
                else ovf |= OVF_POOL;
            }
        }
        n_prev++;
    };

    for (;;) {
        while (state != F_EXT && state != F_DONE) {            // `break` = yield: sit out one extension round
            if (state == F_NEWITEM) {
                if (it_a >= n_items) { state = F_DONE; break; }
                if (MODE == W_P2) { r = pl_t.r; rd_off = pl_t.rd_off; L = pl_t.L; x = pl_t.x; min_intv = (int64_t)pl_t.s + 1; }
                else { r = (int32_t)it_a; rd_off = pl_off; L = pl_len; x = 0; min_intv = 1; }
                it_a = wave_alloc<ITEM_BATCH>(ip, item_cur);
                if (it_a < n_items) { if (MODE == W_P2) pl_t = tasks[it_a]; else { pl_off = off[it_a]; pl_len = len[it_a]; } }
                if (MODE == W_P2 && r < 0) break;              // padding of a task chunk: take the next item
                q = enc + rd_off;
                if (L <= 0) break;
                w.start(q, x, 1);
                state = F_NEWPOS; break;
            }
            if (state == F_NEWPOS) {                            // FMI_search.cpp:514-535 / :746-755
                if (MODE != W_P2 && x >= L) { state = F_NEWITEM; continue; }
                if (!w.get(q, x, 1, a)) break;
                next_x = x + 1;
                if (a >= 4) {
                    if (MODE == W_P2) state = F_NEWITEM; else x = next_x;
                    continue;
                }
                const Bi b = init_bi(ix, a); smk = b.k; sml = b.l; sms = b.s; smn = x;
                j = x + 1;
                if (MODE != W_P3) { n_prev = 0; pool_id = -1; slot = wave_alloc<SLOT_BATCH>(op, out_cur); }
                state = F_CHK;
            }
            if (state == F_CHK) {                               // :537-545 / :761-770
                if (j >= L) state = F_END;
                else if (!w.get(q, j, 1, a)) break;
                else {
                    next_x = j + 1;
                    if (a < 4) { eik = sml; eil = smk; eis = sms; ea = 3 - a; state = F_EXT; }
                    else state = F_END;
                }
            }
            if (state == F_END) {
                if (MODE == W_P3) { x = next_x; state = F_NEWPOS; continue; }
                if (sms >= min_intv) push(pv_pack(smk, sml, sms, smn));                      // :576-580
                if (slot < slot_cap) {
                    bool heavy = n_prev > HEAVY_T && n_prev <= HCAP;              // long list: a whole wave will take this task
                    if (heavy) {
                        const int64_t hid = wave_alloc<HEAVY_BATCH>(hp, sc + (MODE == W_P1 ? SC_HEAVY1 : SC_HEAVY2));
                        if (hid < heavy_cap) heavy_ids[hid] = (int32_t)slot; else heavy = false;
                    }
                    BHead h; h.rd_off = rd_off; h.r = r; h.L = L; h.x_np = (uint32_t)x | (uint32_t)n_prev << 16;
                    h.mi_pass = (uint32_t)min_intv | (uint32_t)(MODE == W_P1 ? 1 : 2) << 16 | (heavy ? 1u << 24 : 0u); h.pool_id = pool_id; h.pad = 0;
                    heads[slot] = h;
                } else ovf |= (MODE == W_P1 ? OVF_SLOT1 : OVF_SLOT2);
                if (MODE == W_P1) { x = next_x; state = F_NEWPOS; } else state = F_NEWITEM;
            }
        }
        if (!__any(state != F_DONE)) break;
#ifdef BM2_SMEM_PROF
        if ((threadIdx.x & 63) == 0) prof_rounds++;
        if (state == F_EXT) prof_active++;
#endif
        const Bi ein = { eik, eil, eis };
        const Bi o = backward_ext(ix, ein, ea, state == F_EXT);   // (all lanes: quad-cooperative)
        if (state == F_EXT) {                                   // forward = swapped backward, :546-570
            n_ext++;
            if (MODE == W_P3) {                                 // :771-808
                smk = o.l; sml = o.k; sms = o.s; smn = j;
                if (sms < sp.max_mem_intv && (smn - x + 1) >= sp.min_seed_len + 1) {
                    if (sms > 0) {
                        const int64_t at = wave_alloc<REC_BATCH>(op, out_cur);
                        if (at < rec_cap) {
                            bm2_smem_t v; v.rid = (uint32_t)r; v.m = (uint32_t)x; v.n = (uint32_t)smn; v.pad = 0; v.k = smk; v.l = sml; v.s = sms;
                            recs[at] = v;
                            atomicAdd(&smem_cnt[r], 1);
                        } else ovf |= OVF_REC;
                    }
                    x = next_x; state = F_NEWPOS;
                } else { j++; state = F_CHK; }
            } else {
                if (o.s != sms) push(pv_pack(smk, sml, sms, smn));
                if (o.s < min_intv) { next_x = j; state = F_END; }
                else { smk = o.l; sml = o.k; sms = o.s; smn = j; j++; state = F_CHK; }
            }
        }
    }
    // close: the ids left in the wave's pool must not look like work
    for (int64_t at = op->pos + (threadIdx.x & 63); at < op->end; at += 64) {
        if (MODE != W_P3) { if (at < slot_cap) { BHead h = {}; h.r = -1; h.pool_id = -1; heads[at] = h; } }
        else if (at < rec_cap) recs[at].rid = 0xffffffffu;
    }
    if (MODE != W_P3) for (int64_t at = hp->pos + (threadIdx.x & 63); at < hp->end; at += 64) if (at < heavy_cap) heavy_ids[at] = -1;
    atomicAdd(&sc[SC_NEXT], (unsigned long long)n_ext);
    atomicAdd(&sc[SC_NEXT_W1 + (MODE - 1)], (unsigned long long)n_ext);
    if (ovf) atomicOr(&sc[SC_OVF_FLAG], (unsigned long long)ovf);
#ifdef BM2_SMEM_PROF
    atomicAdd(&sc[SC_N + 2 * (MODE - 1)], prof_rounds); atomicAdd(&sc[SC_N + 2 * (MODE - 1) + 1], prof_active);
#endif
}

// ---- backward phases --------------------------------------------------------------------------------------------
// One task = the candidate list of one start position (FMI_search.cpp:586-665).  The list stays where the walk wrote it
// and is compacted in place: it is read top-down (longest candidate first = the reversal of :586-592), the survivors
// of a row are written top-down behind the reader, the next candidate is requested while the current one is extended,
// and the first survivor of a row -- the first candidate of the next row -- never leaves the registers.
enum { B_EXT = 0, B_NEWITEM, B_FIRST, B_ROWEND, B_ROW, B_FIN, B_DONE };

// LC: survivors of a row kept in LDS per lane (16 B x LC x 256 lanes per block decides how many blocks share a CU's 160 KB)
template <int LC>
static __device__ __forceinline__ void
bwd_body(const DevIndex &ix, const SeedParams &sp, int pass, const uint8_t *__restrict__ enc, const BHead *__restrict__ heads,
         uint4 *__restrict__ ents, int64_t slot_cap, uint4 *__restrict__ pool, int pool_cap, int pool_slots,
         bm2_smem_t *__restrict__ recs, int64_t rec_cap, P2Task *__restrict__ tasks, int64_t task_cap,
         int32_t *__restrict__ smem_cnt, unsigned long long *sc, CTask *__restrict__ ctasks, int64_t cont_cap, int export_age) {
    int64_t n_ext = 0;
    unsigned ovf = 0;
    int64_t n_items = (int64_t)sc[pass == 1 ? SC_SLOT1 : SC_SLOT2];
    if (n_items > slot_cap) n_items = slot_cap;
    __shared__ WavePool pools[4][4];                           // per wave: [0] work items, [1] records, [2] pass-2 tasks, [3] ids of handed-over tasks
    LdsPool *ip = (LdsPool *)&pools[threadIdx.x >> 6][0], *rp = (LdsPool *)&pools[threadIdx.x >> 6][1], *tp = (LdsPool *)&pools[threadIdx.x >> 6][2];
    LdsPool *cp = (LdsPool *)&pools[threadIdx.x >> 6][3];
    if ((threadIdx.x & 63) == 0) { ip->pos = ip->end = 0; rp->pos = rp->end = 0; tp->pos = tp->end = 0; cp->pos = cp->end = 0; }
    unsigned long long *item_cur = sc + (pass == 1 ? SC_B1_ITEM : SC_B2_ITEM);
    int64_t it_a = wave_alloc<ITEM_BATCH>(ip, item_cur);
    BHead pl = {};
    if (it_a < n_items) pl = heads[it_a];

#ifdef BM2_SMEM_PROF
    unsigned long long prof_rounds = 0, prof_active = 0;
#endif
    int state = B_NEWITEM;
    int32_t r = 0; const uint8_t *q = enc; int L = 0;
    int x = 0, j = 0, a = 0, n_prev = 0, top = 0, n_curr = 0, p = 0, m_row = 0; int32_t curr_s = -1; bool first_done = false;
    int32_t min_intv = 1;                                      // <= 65535 (pass 2: s + 1 with s <= split_width)
    int age = 0;                                               // extensions this task has had so far (hand-over: see B_ROWEND)
    uint4 *lst = ents; uint4 *lpool = pool;                   // this task's list: entries [0, CAPF) and [CAPF, ...)
    int64_t ck = 0, cl = 0, cs = 0; int cn = 0;                // the candidate being extended
    int64_t fk = 0, fl = 0, fs = 0; int fn = 0;                // first survivor of the current row
    uint4 nxt_raw4 = {};                                       // the candidate after the current one, requested one round ahead
    int em = 0;                                                // an SMEM to write out: 1 = the candidate, 2 = the first survivor
    QWin w; w.cur = w.nxt = 0; w.curb = w.nxtb = -64;
    auto entry = [&](int idx) -> uint4 * { return idx < CAPF ? lst + idx : lpool + (idx - CAPF); };
    // survivors at depth 1..LC below the top of the list live in LDS ([depth][lane]); deeper ones go back to the slot
    __shared__ uint4 surv[LC * 256];
    bool row0 = true;                                          // the row being read is the list the walk wrote (global)
    auto cand_load = [&](int depth) -> uint4 {
        if (row0) return *entry(top - depth);
        uint4 v = surv[(depth <= LC ? depth - 1 : 0) * 256 + threadIdx.x];
        if (depth > LC) {
            v = *entry(top - depth);
            asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));      // (keeps the LDS and the global load apart)
        }
        return v;
    };

    for (;;) {
        while (state != B_EXT && state != B_DONE) {            // `break` = yield
            if (state == B_NEWITEM) {
                if (it_a >= n_items) { state = B_DONE; break; }
                const BHead h = pl;
                const int64_t slot = it_a;
                it_a = wave_alloc<ITEM_BATCH>(ip, item_cur);
                if (it_a < n_items) pl = heads[it_a];
                n_prev = (int)(h.x_np >> 16);
                if (h.r < 0 || n_prev == 0 || ((h.mi_pass >> 24) & 1)) break;      // unused slot / empty list / a heavy task (k_bwd_heavy): next item
                r = h.r; L = h.L; x = (int)(h.x_np & 0xffff); min_intv = (int32_t)(h.mi_pass & 0xffff);
                q = enc + h.rd_off;
                lst = ents + slot * CAPF;
                lpool = pool + (int64_t)(h.pool_id >= 0 && h.pool_id < pool_slots ? h.pool_id : 0) * pool_cap;
                top = n_prev - 1; j = x - 1; m_row = x; row0 = true; age = 0;
                nxt_raw4 = *entry(top);
                if (j >= 0) w.start(q, j, -1);
                state = B_FIRST; break;
            }
            if (state == B_FIRST) {
                pv_unpack(nxt_raw4, fk, fl, fs, fn);
                state = B_ROW;
            }
            if (state == B_ROWEND) {                            // :650-655
                n_prev = n_curr;
                if (n_curr == 0) state = B_FIN;
                else if (export_age > 0 && age >= export_age && j > 0 && n_curr <= CCAP) {
                    // HAND-OVER.  A lane-per-task kernel cannot end before its oldest task does, and the task sizes have a long tail (mean ~100
                    // extensions, one in a thousand beyond 1000: tests/seed_sim): the last third of this kernel used to be a few lanes finishing
                    // repeat-rich positions on an otherwise empty GPU.  A task that has had `export_age` extensions stops at the end of its row:
                    // the row's survivors go back into the task's slot where the reader of a walk's list expects them (entry(top - depth)), a
                    // CTask says row j is done, and k_bwd_cont -- sixteen lanes per task, the candidates of a row side by side -- finishes it in a
                    // launch of its own after this one (a kernel boundary: no hand-off between running workgroups).
                    const int64_t cid = wave_alloc_exact(cp, sc + (pass == 1 ? SC_CONT1 : SC_CONT2));
                    if (cid < cont_cap) {
                        const int64_t slot = (int64_t)(lst - ents) / CAPF;
                        *entry(top) = pv_pack(fk, fl, fs, fn);
                        const int nl = n_curr - 1 < LC ? n_curr - 1 : LC;
                        for (int d = 1; d <= nl; d++) *entry(top - d) = surv[(d - 1) * 256 + threadIdx.x];
                        CTask t;
                        t.h.rd_off = (int64_t)(q - enc); t.h.r = r; t.h.L = L;
                        t.h.x_np = (uint32_t)j | (uint32_t)n_curr << 16;        // "start position" j: the continuation sets m_row = j and goes on with row j - 1
                        t.h.mi_pass = (uint32_t)min_intv | (uint32_t)pass << 16;
                        t.h.pool_id = (int32_t)((lpool - pool) / pool_cap);     // (0 for a list without a pool part: then nothing beyond CAPF is ever read)
                        t.h.pad = top + 1;                                      // where the list's first entry is
                        t.slot = (int32_t)slot; t.pad0 = t.pad1 = t.pad2 = 0;
                        ctasks[cid] = t;
                        state = B_NEWITEM;
                    } else { m_row = j; j--; row0 = false; state = B_ROW; }
                }
                else { m_row = j; j--; row0 = false; state = B_ROW; }
            }
            if (state == B_ROW) {                               // :596-606
                if (j < 0) state = B_FIN;
                else if (!w.get(q, j, -1, a)) break;
                else if (a > 3) state = B_FIN;
                else {
                    n_curr = 0; curr_s = -1; p = 0; first_done = false;
                    ck = fk; cl = fl; cs = fs; cn = fn;
                    if (n_prev > 1) nxt_raw4 = cand_load(1);
                    state = B_EXT;
                }
            }
            if (state == B_FIN) {                               // :656-665
                if (n_prev != 0 && (fn - m_row + 1) >= sp.min_seed_len) em = 2;
                state = B_NEWITEM;
                if (em) break;                                  // write it out below, then look for work
            }
        }
        if (!__any(state != B_DONE || em)) break;
#ifdef BM2_SMEM_PROF
        if ((threadIdx.x & 63) == 0) prof_rounds++;
        if (state == B_EXT) prof_active++;
#endif
        const Bi ein = { ck, cl, cs };
        const Bi o = backward_ext(ix, ein, a, state == B_EXT);    // (all lanes: quad-cooperative)
        if (state == B_EXT) {                                   // :607-649
            n_ext++; age++;
            if (!first_done && o.s < (int64_t)min_intv && (cn - m_row + 1) >= sp.min_seed_len) {
                em = 1;
                first_done = true;
            } else if (o.s >= (int64_t)min_intv && o.s != (int64_t)curr_s) {
                curr_s = (int32_t)o.s;
                if (n_curr == 0) { fk = o.k; fl = o.l; fs = o.s; fn = cn; }
                else if (n_curr <= LC) surv[(n_curr - 1) * 256 + threadIdx.x] = pv_pack(o.k, o.l, o.s, cn);
                else *entry(top - n_curr) = pv_pack(o.k, o.l, o.s, cn);
                n_curr++;
                first_done = true;
            }
        }
        if (em) {                                               // one SMEM: record, per-read count, pass-2 task
            const int64_t ek = em == 1 ? ck : fk, el = em == 1 ? cl : fl, es = em == 1 ? cs : fs;
            const int en = em == 1 ? cn : fn;
            em = 0;
            const int64_t at = wave_alloc<REC_BATCH>(rp, sc + SC_REC);
            if (at < rec_cap) {
                bm2_smem_t v; v.rid = (uint32_t)r; v.m = (uint32_t)m_row; v.n = (uint32_t)en; v.pad = 0; v.k = ek; v.l = el; v.s = es;
                recs[at] = v;
                atomicAdd(&smem_cnt[r], 1);
            } else ovf |= OVF_REC;
            if (pass == 1 && (en + 1 - m_row) >= sp.split_len && es <= (int64_t)sp.split_width) {       // bwamem.cpp:701-703
                const int64_t ta = wave_alloc<TASK_BATCH>(tp, sc + SC_TASK);
                if (ta < task_cap) { P2Task t; t.rd_off = (int64_t)(q - enc); t.r = r; t.L = L; t.x = (en + 1 + m_row) >> 1; t.s = (int32_t)es; t.pad = 0; tasks[ta] = t; }
                else ovf |= OVF_TASK;
            }
        }
        if (state == B_EXT) {                                   // on to the next candidate of the row
            p++;
            if (p < n_prev) {
                pv_unpack(nxt_raw4, ck, cl, cs, cn);
                if (p + 1 < n_prev) nxt_raw4 = cand_load(p + 1);
            } else state = B_ROWEND;
        }
    }
    for (int64_t at = rp->pos + (threadIdx.x & 63); at < rp->end; at += 64) if (at < rec_cap) recs[at].rid = 0xffffffffu;
    for (int64_t at = tp->pos + (threadIdx.x & 63); at < tp->end; at += 64) if (at < task_cap) tasks[at].r = -1;
    atomicAdd(&sc[SC_NEXT], (unsigned long long)n_ext);
    atomicAdd(&sc[SC_NEXT_B1 + (pass - 1)], (unsigned long long)n_ext);
    if (ovf) atomicOr(&sc[SC_OVF_FLAG], (unsigned long long)ovf);
#ifdef BM2_SMEM_PROF
    atomicAdd(&sc[SC_N + 6 + 2 * (pass - 1)], prof_rounds); atomicAdd(&sc[SC_N + 6 + 2 * (pass - 1) + 1], prof_active);
#endif
}

#define BWD_ARGS DevIndex ix, SeedParams sp, int pass, const uint8_t *__restrict__ enc, const BHead *__restrict__ heads, uint4 *__restrict__ ents, \
                 int64_t slot_cap, uint4 *__restrict__ pool, int pool_cap, int pool_slots, bm2_smem_t *__restrict__ recs, int64_t rec_cap, \
                 P2Task *__restrict__ tasks, int64_t task_cap, int32_t *__restrict__ smem_cnt, unsigned long long *sc, \
                 CTask *__restrict__ ctasks, int64_t cont_cap, int export_age
#define BWD_PASS ix, sp, pass, enc, heads, ents, slot_cap, pool, pool_cap, pool_slots, recs, rec_cap, tasks, task_cap, smem_cnt, sc, ctasks, cont_cap, export_age
template <int LC> __global__ void __launch_bounds__(256) k_bwd(BWD_ARGS) { bwd_body<LC>(BWD_PASS); }
// the same with the register allocation told to leave room for 5 waves per SIMD (96 VGPRs, 100 bytes per lane spilled)
template <int LC> __global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5))) k_bwd5(BWD_ARGS) { bwd_body<LC>(BWD_PASS); }
// Measured in round 5 and removed (profiles/r05b_sweep.json, r05c_sweep_slow_rounds.json, r05h_sweep_kmer_table_pass3.json): (0) a k-mer table (the
// bi-interval of every string of up to 12 bases, 537 MB, built on the device at index upload -- the interval of a string does not depend on the order it was
// extended in, checked with the oracle on 2 M strings) from which pass 3's walks took their first 9..12 bases in one lookup: bit-exact, 40 % fewer rounds in
// k_walk<3> (beside k_walk<1>: 8.3 -> 7.1 ms), and the stage no faster: the reworked kernel beside k_bwd of pass 1 made that interval 17.0 ms with or without
// the table (14.1 with the kernel as it is), beside k_walk<1> the stage took 32.5 ms against 31.5.  (1) two candidates of a row per round (sixteen requests of a quad
// in flight, 161 VGPRs): bwd1 + bwd2 23.5 ms with it, 23.9 without -- lines in flight are not what binds the kernel; (2) the rare steps of a lane (new
// task, SMEM record) only in every 2nd / 4th / 8th round, so that a wavefront pays for those blocks of the state machine less often: -0.8 ms at every
// 4th round, and the restructured loops cost more than that in default form (phi copies of prefetched registers at the loop header: a
// `s_waitcnt vmcnt(0)` per round; k_walk<3> 3.3 -> 17 ms)

// lanes of one wavefront handing data to each other through LDS: order the accesses (the hardware runs them in lockstep)
static __device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- backward phases of LONG candidate lists: one task per WAVEFRONT ----------------------------------------------------
// A lane-per-task kernel cannot end before its longest task does, and a list of n candidates costs ~n * rows extensions in
// a row: repeat-rich positions would set the length of the whole stage.  Within a row the candidates are independent
// (FMI_search.cpp:607-649 only filters them in order), so here the 64 lanes extend 64 candidates of the row at once and the
// in-order rules become ballots: the first candidate that either dies long enough or survives decides `first_done`
// (-> at most one SMEM per row, and only if that first one died), and a live candidate is kept iff its interval size
// differs from the previous LIVE candidate's (equal to the reference's compare with the last kept size, truncated to
// int32 as there: a dropped candidate has the size of the last kept one).  The list sits in LDS, compacted in place.
__global__ void __launch_bounds__(256)
k_bwd_heavy(DevIndex ix, SeedParams sp, int pass, const uint8_t *__restrict__ enc, const BHead *__restrict__ heads,
            const uint4 *__restrict__ ents, int64_t slot_cap, const uint4 *__restrict__ pool, int pool_cap, int pool_slots,
            const int32_t *__restrict__ heavy_ids, int64_t heavy_cap,
            bm2_smem_t *__restrict__ recs, int64_t rec_cap, P2Task *__restrict__ tasks, int64_t task_cap,
            int32_t *__restrict__ smem_cnt, unsigned long long *sc) {
    __shared__ uint4 lists[4][HCAP];
    __shared__ int32_t alive_s[4][64];
    __shared__ WavePool pools[4][2];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    LdsPool *rp = (LdsPool *)&pools[wv][0], *tp = (LdsPool *)&pools[wv][1];
    if (lane == 0) { rp->pos = rp->end = 0; tp->pos = tp->end = 0; }
    uint4 *lst = lists[wv];
    int32_t *as = alive_s[wv];
    int64_t n_items = (int64_t)sc[pass == 1 ? SC_HEAVY1 : SC_HEAVY2];
    if (n_items > heavy_cap) n_items = heavy_cap;
    unsigned long long *item_cur = sc + (pass == 1 ? SC_H1_ITEM : SC_H2_ITEM);
    int64_t n_ext = 0;
    unsigned ovf = 0;
    const unsigned long long lt_mask = lane ? (~0ULL >> (64 - lane)) : 0ULL;
    for (;;) {
        // one item per wavefront.  EVERY lane executes the atomic (lane 0 adds 1, the others 0) and the body sits in an `if`, not
        // behind a `continue`: with `if (lane == 0) it = atomicAdd(...)` + `continue` the compiler may structurise the loop so
        // that lanes 1..63 go round again without the atomic and read item 0 for ever (seen in the purge kernel, notes/NEXT.md)
        const unsigned long long it = atomicAdd(item_cur, lane == 0 ? 1ULL : 0ULL);
        const int64_t hid = (int64_t)(((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(it >> 32)) << 32) |
                                      (unsigned)__builtin_amdgcn_readfirstlane((unsigned)it));
        if (hid >= n_items) break;
        const int slot = heavy_ids[hid];
        if (slot >= 0 && slot < slot_cap) {
        const BHead h = heads[slot];
        int n_prev = (int)(h.x_np >> 16);
        const int x = (int)(h.x_np & 0xffff), r = h.r, L = h.L;
        const int64_t min_intv = (int64_t)(h.mi_pass & 0xffff);
        const uint8_t *q = enc + h.rd_off;
        const uint4 *src = ents + (int64_t)slot * CAPF;
        const uint4 *psrc = pool + (int64_t)(h.pool_id >= 0 && h.pool_id < pool_slots ? h.pool_id : 0) * pool_cap;
        const int top = n_prev - 1;
        for (int d = lane; d < n_prev; d += 64) {                // longest first (:586-592) = the walk's list read top-down
            const int idx = top - d;
            lst[d] = idx < CAPF ? src[idx] : psrc[idx - CAPF];
        }
        wave_sync();
        int m_row = x;
        for (int j = x - 1; j >= 0 && n_prev > 0; j--) {          // :596-655
            const int a = q[j];
            if (a > 3) break;
            int n_curr = 0; bool first_done = false; int32_t curr_s = -1;
            for (int c0 = 0; c0 < n_prev; c0 += 64) {
                const int d = c0 + lane;
                const bool valid = d < n_prev;
                int64_t ck = 0, cl = 0, cs = 0; int cn = 0;
                if (valid) pv_unpack(lst[d], ck, cl, cs, cn);
                const Bi in = { ck, cl, cs };
                const Bi o = backward_ext(ix, in, a, valid);
                if (valid) n_ext++;
                const bool alive = valid && o.s >= min_intv;
                const bool deadlen = valid && o.s < min_intv && (cn - m_row + 1) >= sp.min_seed_len;
                const unsigned long long am = __ballot(alive), dm = __ballot(deadlen);
                if (!first_done && (am | dm)) {
                    const int f = __ffsll((long long)(am | dm)) - 1;
                    if (((dm >> f) & 1) && lane == f) {          // the first to trigger died long enough: one SMEM, FMI_search.cpp:611-621
                        const int64_t at = wave_alloc<REC_BATCH>(rp, sc + SC_REC);
                        if (at < rec_cap) {
                            bm2_smem_t v; v.rid = (uint32_t)r; v.m = (uint32_t)m_row; v.n = (uint32_t)cn; v.pad = 0; v.k = ck; v.l = cl; v.s = cs;
                            recs[at] = v;
                            atomicAdd(&smem_cnt[r], 1);
                        } else ovf |= OVF_REC;
                        if (pass == 1 && (cn + 1 - m_row) >= sp.split_len && cs <= (int64_t)sp.split_width) {
                            const int64_t ta = wave_alloc<TASK_BATCH>(tp, sc + SC_TASK);
                            if (ta < task_cap) { P2Task t; t.rd_off = h.rd_off; t.r = r; t.L = L; t.x = (cn + 1 + m_row) >> 1; t.s = (int32_t)cs; t.pad = 0; tasks[ta] = t; }
                            else ovf |= OVF_TASK;
                        }
                    }
                    first_done = true;
                }
                // keep a live candidate iff its size differs from the previous live one's (int32, :625 / :640)
                const int arank = __popcll(am & lt_mask);
                if (alive) as[arank] = (int32_t)o.s;
                wave_sync();
                const int32_t prev_s = alive ? (arank ? as[arank - 1] : curr_s) : 0;
                const bool keep = alive && o.s != (int64_t)prev_s;
                const unsigned long long km = __ballot(keep);
                if (keep) lst[n_curr + __popcll(km & lt_mask)] = pv_pack(o.k, o.l, o.s, cn);
                n_curr += __popcll(km);
                if (am) curr_s = as[__popcll(am) - 1];
                wave_sync();
            }
            n_prev = n_curr;
            m_row = j;
        }
        wave_sync();
        if (n_prev > 0 && lane == 0) {                            // :656-665
            int64_t ck, cl, cs; int cn;
            pv_unpack(lst[0], ck, cl, cs, cn);
            if ((cn - m_row + 1) >= sp.min_seed_len) {
                const int64_t at = wave_alloc<REC_BATCH>(rp, sc + SC_REC);
                if (at < rec_cap) {
                    bm2_smem_t v; v.rid = (uint32_t)r; v.m = (uint32_t)m_row; v.n = (uint32_t)cn; v.pad = 0; v.k = ck; v.l = cl; v.s = cs;
                    recs[at] = v;
                    atomicAdd(&smem_cnt[r], 1);
                } else ovf |= OVF_REC;
                if (pass == 1 && (cn + 1 - m_row) >= sp.split_len && cs <= (int64_t)sp.split_width) {
                    const int64_t ta = wave_alloc<TASK_BATCH>(tp, sc + SC_TASK);
                    if (ta < task_cap) { P2Task t; t.rd_off = h.rd_off; t.r = r; t.L = L; t.x = (cn + 1 + m_row) >> 1; t.s = (int32_t)cs; t.pad = 0; tasks[ta] = t; }
                    else ovf |= OVF_TASK;
                }
            }
        }
        }
    }
    for (int64_t at = rp->pos + lane; at < rp->end; at += 64) if (at < rec_cap) recs[at].rid = 0xffffffffu;
    for (int64_t at = tp->pos + lane; at < tp->end; at += 64) if (at < task_cap) tasks[at].r = -1;
    atomicAdd(&sc[SC_NEXT], (unsigned long long)n_ext);
    atomicAdd(&sc[SC_NEXT_B1 + (pass - 1)], (unsigned long long)n_ext);
    if (ovf) atomicOr(&sc[SC_OVF_FLAG], (unsigned long long)ovf);
}

// ---- the tasks k_bwd handed over: SIXTEEN LANES per task, four tasks per wavefront ---------------------------------------------
// A handed-over task has had its share of a lane (BM2_BWD_EXPORT_AGE extensions) and still has rows to go -- a few candidates over tens of rows,
// or tens of candidates over a few.  Its rows are sequential, its candidates are not: a group of sixteen lanes extends sixteen candidates of
// the row at once (the in-order rules as ballots over the group, as in k_bwd_heavy), so a task lasts (rows x chunks) rounds instead of (rows x
// candidates).  Four groups share a wavefront's converged backwardExt; what a group needs between two rounds -- its next task's record, the
// query bases eight rows ahead -- is requested a task / a window ahead.
enum { C_EXT = 0, C_NEWITEM, C_LIST, C_ROW, C_FIN, C_DONE };
#if !defined(BM2_EMU_ROW_PRIMS)                                /* (tools/emu supplies the row primitives: a rendezvous of the sixteen threads of a row) */
static __device__ __forceinline__ int row_first(int v) { return __shfl(v, 0, 16); }     // lane 0 of this 16-lane row; the row executes this together
#endif
__global__ void __launch_bounds__(256)
k_bwd_cont(DevIndex ix, SeedParams sp, int pass, const uint8_t *__restrict__ enc, const uint4 *__restrict__ ents, int64_t slot_cap,
           const uint4 *__restrict__ pool, int pool_cap, int pool_slots, const CTask *__restrict__ ctasks, int64_t cont_cap,
           bm2_smem_t *__restrict__ recs, int64_t rec_cap, P2Task *__restrict__ tasks, int64_t task_cap,
           int32_t *__restrict__ smem_cnt, unsigned long long *sc) {
    __shared__ uint4 lists[4][4][CCAP];
    __shared__ int32_t alive_s[4][4][16];
    __shared__ WavePool pools[4][3];                           // per wave: [0] work items, [1] records, [2] pass-2 tasks
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, g = lane >> 4, gl = lane & 15;
    LdsPool *ip = (LdsPool *)&pools[wv][0], *rp = (LdsPool *)&pools[wv][1], *tp = (LdsPool *)&pools[wv][2];
    if (lane == 0) { ip->pos = ip->end = 0; rp->pos = rp->end = 0; tp->pos = tp->end = 0; }
    uint4 *lst = lists[wv][g];
    int32_t *as = alive_s[wv][g];
    int64_t n_items = (int64_t)sc[pass == 1 ? SC_CONT1 : SC_CONT2];
    if (n_items > cont_cap) n_items = cont_cap;
    unsigned long long *item_cur = sc + (pass == 1 ? SC_C1_ITEM : SC_C2_ITEM);
    const unsigned glt = (1u << gl) - 1u;                        // the group's lanes below this one
    auto gballot = [&](bool v) -> unsigned { return (unsigned)(__ballot(v) >> (16 * g)) & 0xffffu; };
    // the group's next work item (its first lane draws, the others copy) with the item's record already requested
    auto draw = [&]() -> int64_t {
        int64_t id = 0;
        if (gl == 0) id = wave_alloc<4>(ip, item_cur);
        const unsigned lo = (unsigned)row_first((int)(unsigned)id), hi = (unsigned)row_first((int)(unsigned)(id >> 32));
        return (int64_t)(((unsigned long long)hi << 32) | lo);
    };
    int64_t it_a = draw();
    // (the record as three 16-byte words picked apart by hand: a struct copied out of memory stays a scratch alloca with this compiler)
    uint4 pl0 = {}, pl1 = {}, pl2 = {};                          // {rd_off, r, L} | {x_np, mi_pass, pool_id, pad} | {slot, -, -, -}
    auto fetch = [&](int64_t id) { const uint4 *p = (const uint4 *)(ctasks + id); pl0 = p[0]; pl1 = p[1]; pl2 = p[2]; };
    if (it_a < n_items) fetch(it_a);
    int64_t n_ext = 0, n_rows = 0;
    unsigned ovf = 0;
    int state = C_NEWITEM;
    int32_t r = 0; const uint8_t *q = enc; int L = 0; int64_t rd_off = 0;
    int j = 0, a = 0, n_prev = 0, n_curr = 0, c0 = 0, m_row = 0, top = 0; int32_t curr_s = -1; bool first_done = false;
    int64_t min_intv = 1;
    const uint4 *src = ents, *psrc = pool;
    uint4 e4a = {}, e4b = {}, e4c = {}, e4d = {};             // (four named registers: an array indexed in an unrolled loop went to scratch)
    static_assert(CCAP == 64, "k_bwd_cont stages a task's list as four entries per lane");
    QWin w; w.cur = w.nxt = 0; w.curb = w.nxtb = -64;
    auto emit = [&](int64_t ck, int64_t cl, int64_t cs, int cn) {   // one SMEM: record, per-read count, pass-2 task (FMI_search.cpp:611-621 / :656-665)
        const int64_t at = wave_alloc<REC_BATCH>(rp, sc + SC_REC);
        if (at < rec_cap) {
            bm2_smem_t v; v.rid = (uint32_t)r; v.m = (uint32_t)m_row; v.n = (uint32_t)cn; v.pad = 0; v.k = ck; v.l = cl; v.s = cs;
            recs[at] = v;
            atomicAdd(&smem_cnt[r], 1);
        } else ovf |= OVF_REC;
        if (pass == 1 && (cn + 1 - m_row) >= sp.split_len && cs <= (int64_t)sp.split_width) {       // bwamem.cpp:701-703
            const int64_t ta = wave_alloc<TASK_BATCH>(tp, sc + SC_TASK);
            if (ta < task_cap) { P2Task t; t.rd_off = rd_off; t.r = r; t.L = L; t.x = (cn + 1 + m_row) >> 1; t.s = (int32_t)cs; t.pad = 0; tasks[ta] = t; }
            else ovf |= OVF_TASK;
        }
    };
    for (;;) {
        while (state != C_EXT && state != C_DONE) {            // `break` = yield: the group sits out one extension round
            if (state == C_NEWITEM) {
                if (it_a >= n_items) { state = C_DONE; break; }
                const uint4 t0 = pl0, t1 = pl1, t2 = pl2;
                it_a = draw();
                if (it_a < n_items) fetch(it_a);
                const int32_t t_r = (int32_t)t0.z, t_slot = (int32_t)t2.x, t_pool = (int32_t)t1.z;
                n_prev = (int)(t1.x >> 16);
                if (t_r < 0 || n_prev <= 0 || n_prev > CCAP || t_slot < 0 || t_slot >= slot_cap) break;      // padding of a wave's id pool: next item
                r = t_r; L = (int)t0.w; rd_off = (int64_t)((uint64_t)t0.x | (uint64_t)t0.y << 32); q = enc + rd_off; min_intv = (int64_t)(t1.y & 0xffff);
                m_row = (int)(t1.x & 0xffff); j = m_row - 1; top = (int)t1.w - 1;
                src = ents + (int64_t)t_slot * CAPF;
                psrc = pool + (int64_t)(t_pool >= 0 && t_pool < pool_slots ? t_pool : 0) * pool_cap;
                auto list_entry = [&](int d) -> uint4 { const int idx = top - d; return idx < CAPF ? src[idx] : psrc[idx - CAPF]; };      // longest first = the list read top-down
                if (gl < n_prev) e4a = list_entry(gl);
                if (gl + 16 < n_prev) e4b = list_entry(gl + 16);
                if (gl + 32 < n_prev) e4c = list_entry(gl + 32);
                if (gl + 48 < n_prev) e4d = list_entry(gl + 48);
                if (j >= 0) w.start(q, j, -1);
                state = C_LIST; break;
            }
            if (state == C_LIST) {
                if (gl < n_prev) lst[gl] = e4a;                 // (a lane reads back what it wrote itself until the first row is compacted)
                if (gl + 16 < n_prev) lst[gl + 16] = e4b;
                if (gl + 32 < n_prev) lst[gl + 32] = e4c;
                if (gl + 48 < n_prev) lst[gl + 48] = e4d;
                state = C_ROW;
            }
            if (state == C_ROW) {                               // :596-606
                if (j < 0 || n_prev == 0) state = C_FIN;
                else if (!w.get(q, j, -1, a)) break;
                else if (a > 3) state = C_FIN;
                else { n_curr = 0; curr_s = -1; first_done = false; c0 = 0; n_rows += gl == 0; state = C_EXT; }
            }
            if (state == C_FIN) {                               // :656-665
                if (n_prev > 0 && gl == 0) {
                    int64_t ck, cl, cs; int cn;
                    pv_unpack(lst[0], ck, cl, cs, cn);
                    if ((cn - m_row + 1) >= sp.min_seed_len) emit(ck, cl, cs, cn);
                }
                state = C_NEWITEM;
            }
        }
        if (!__any(state != C_DONE)) break;
        const int d = c0 + gl;
        const bool valid = state == C_EXT && d < n_prev;
        int64_t ck = 0, cl = 0, cs = 0; int cn = 0;
        if (valid) pv_unpack(lst[d], ck, cl, cs, cn);
        const Bi in = { ck, cl, cs };
        const Bi o = backward_ext(ix, in, a, valid);            // (all lanes: quad-cooperative)
        if (valid) n_ext++;
        const bool alive = valid && o.s >= min_intv;
        const bool deadlen = valid && o.s < min_intv && (cn - m_row + 1) >= sp.min_seed_len;
        const unsigned am = gballot(alive), dm = gballot(deadlen);
        if (state == C_EXT && !first_done && (am | dm)) {
            const int f = __ffsll((long long)(am | dm)) - 1;
            if (((dm >> f) & 1u) && gl == f) emit(ck, cl, cs, cn);      // the first to trigger died long enough: one SMEM
            first_done = true;
        }
        // keep a live candidate iff its size differs from the previous live one's (int32, :625 / :640)
        const int arank = __popc(am & glt);
        if (alive) as[arank] = (int32_t)o.s;
        wave_sync();
        const int32_t prev_s = alive ? (arank ? as[arank - 1] : curr_s) : 0;
        const bool keep = alive && o.s != (int64_t)prev_s;
        const unsigned km = gballot(keep);
        if (keep) lst[n_curr + __popc(km & glt)] = pv_pack(o.k, o.l, o.s, cn);
        if (state == C_EXT) {
            n_curr += __popc(km);
            if (am) curr_s = as[__popc(am) - 1];
        }
        wave_sync();
        if (state == C_EXT) {
            c0 += 16;
            if (c0 >= n_prev) { n_prev = n_curr; m_row = j; j--; state = C_ROW; }
        }
    }
    for (int64_t at = rp->pos + lane; at < rp->end; at += 64) if (at < rec_cap) recs[at].rid = 0xffffffffu;
    for (int64_t at = tp->pos + lane; at < tp->end; at += 64) if (at < task_cap) tasks[at].r = -1;
    atomicAdd(&sc[SC_NEXT], (unsigned long long)n_ext);
    atomicAdd(&sc[SC_NEXT_B1 + (pass - 1)], (unsigned long long)n_ext);
    atomicAdd(&sc[SC_CROWS1 + (pass - 1)], (unsigned long long)n_rows);
    if (ovf) atomicOr(&sc[SC_OVF_FLAG], (unsigned long long)ovf);
}

// records (any order, with padding) -> the reads' segments of `tmp`
__global__ void __launch_bounds__(256)
k_rec_scatter(const bm2_smem_t *__restrict__ recs, int64_t rec_cap, const unsigned long long *__restrict__ sc,
              const int64_t *__restrict__ smem_off, int32_t *__restrict__ fill, bm2_smem_t *__restrict__ tmp) {
    int64_t n = (int64_t)sc[SC_REC];
    if (n > rec_cap) n = rec_cap;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const bm2_smem_t v = recs[i];
        if (v.rid == 0xffffffffu) continue;
        tmp[smem_off[v.rid] + atomicAdd(&fill[v.rid], 1)] = v;
    }
}

// Reads with many SMEMs (a 10 kb read owns thousands): ONE WORKGROUP PER READ.  One lane sorting 3000 records in global memory took 0.4 s --
// four fifths of the seeding stage of a long-read chunk.  Here the keys (m << 40 | n << 20 | index: m, n < 2^15) are sorted by a bitonic
// network in LDS (up to 16 k keys = 128 KB), then the records are gathered by all lanes.  Equal (m, n) are field-identical, so the index in
// the key only makes the order total.  A read with more SMEMs than the LDS holds is sorted by lane 0 as before.
#define SMEM_FINISH_BIG 48
#define SMEM_FINISH_LDS_KEYS 16384
__global__ void __launch_bounds__(256)
k_smem_finish_big(const bm2_smem_t *__restrict__ tmp, const int32_t *__restrict__ smem_cnt, const int64_t *__restrict__ smem_off, int32_t max_occ,
                  bm2_smem_t *out, int32_t *occ_cnt, const int32_t *__restrict__ big_list, const int32_t *__restrict__ big_cnt, int32_t *cursor, int lds_keys) {
    extern __shared__ __attribute__((aligned(16))) uint64_t fin_keys[];
    __shared__ int item_s;
    const int n_big = *big_cnt;
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) item_s = atomicAdd(cursor, 1);
        __syncthreads();
        const int item = item_s;
        if (item >= n_big) break;
        const int r = big_list[item];
        const int n = smem_cnt[r];
        const int64_t o = smem_off[r];
        const bm2_smem_t *row = tmp + o;
        if (n > lds_keys) {
            // More SMEMs than the LDS holds keys (a 30 kb read): m is the major key, so the read's SMEMs are cut into runs of m-classes (m >> 6:
            // 512 classes) that fit, each run collected into LDS, sorted there and written behind the run before it.
            __shared__ int cls_cnt[513];
            __shared__ int run_lo, run_hi, run_base, run_n, fits;
            for (int i = threadIdx.x; i < 513; i += blockDim.x) cls_cnt[i] = 0;
            __syncthreads();
            for (int i = threadIdx.x; i < n; i += blockDim.x) atomicAdd(&cls_cnt[row[i].m >> 6 < 512 ? row[i].m >> 6 : 511], 1);
            __syncthreads();
            if (threadIdx.x == 0) { fits = 1; for (int k = 0; k < 512; k++) if (cls_cnt[k] > lds_keys) fits = 0; run_lo = 0; run_base = 0; }
            __syncthreads();
            if (!fits) {                                        // (one m-class alone overflows the LDS: one lane, index array in the read's slice of occ_cnt)
                if (threadIdx.x == 0) {
                    int32_t *idx = occ_cnt + o;
                    for (int i = 0; i < n; i++) idx[i] = i;
                    k_introsort_flat(n, idx, [&](int32_t x, int32_t y) { return row[x].m < row[y].m || (row[x].m == row[y].m && row[x].n < row[y].n); });
                    for (int i = 0; i < n; i++) out[o + i] = row[idx[i]];
                    for (int i = 0; i < n; i++) { const int64_t sv = out[o + i].s; occ_cnt[o + i] = (int32_t)(sv < max_occ ? sv : max_occ); }
                }
                continue;
            }
            for (;;) {
                __syncthreads();
                if (threadIdx.x == 0) {                          // the next run: classes [run_lo, run_hi) with at most LDS_KEYS records
                    int k = run_lo, tot = 0;
                    while (k < 512 && tot + cls_cnt[k] <= lds_keys) tot += cls_cnt[k++];
                    run_hi = k; run_n = 0;
                }
                __syncthreads();
                const int lo = run_lo, hi = run_hi, base = run_base;
                if (lo >= 512) break;
                for (int i = threadIdx.x; i < n; i += blockDim.x) {
                    const int k = row[i].m >> 6 < 512 ? (int)(row[i].m >> 6) : 511;
                    if (k >= lo && k < hi) fin_keys[atomicAdd(&run_n, 1)] = (uint64_t)row[i].m << 40 | (uint64_t)row[i].n << 20 | (uint64_t)i;
                }
                __syncthreads();
                const int cnt = run_n;
                int N2 = 64; while (N2 < cnt) N2 <<= 1;
                for (int i = cnt + threadIdx.x; i < N2; i += blockDim.x) fin_keys[i] = ~(uint64_t)0;
                __syncthreads();
                for (int k = 2; k <= N2; k <<= 1)
                    for (int j = k >> 1; j > 0; j >>= 1) {
                        for (int t = threadIdx.x; t < N2; t += blockDim.x) {
                            const int p = t ^ j;
                            if (p > t) {
                                const uint64_t a = fin_keys[t], b = fin_keys[p];
                                const bool up = (t & k) == 0;
                                if ((a > b) == up) { fin_keys[t] = b; fin_keys[p] = a; }
                            }
                        }
                        __syncthreads();
                    }
                for (int i = threadIdx.x; i < cnt; i += blockDim.x) {
                    const bm2_smem_t v = row[(int)(fin_keys[i] & 0xfffffULL)];
                    out[o + base + i] = v;
                    occ_cnt[o + base + i] = (int32_t)(v.s < max_occ ? v.s : max_occ);
                }
                __syncthreads();
                if (threadIdx.x == 0) { run_lo = hi; run_base = base + cnt; }
            }
            continue;
        }
        int N = 64; while (N < n) N <<= 1;
        for (int i = threadIdx.x; i < N; i += blockDim.x)
            fin_keys[i] = i < n ? ((uint64_t)row[i].m << 40 | (uint64_t)row[i].n << 20 | (uint64_t)i) : ~(uint64_t)0;
        __syncthreads();
        for (int k = 2; k <= N; k <<= 1)
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int t = threadIdx.x; t < N; t += blockDim.x) {
                    const int p = t ^ j;
                    if (p > t) {
                        const uint64_t a = fin_keys[t], b = fin_keys[p];
                        const bool up = (t & k) == 0;
                        if ((a > b) == up) { fin_keys[t] = b; fin_keys[p] = a; }
                    }
                }
                __syncthreads();
            }
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            const bm2_smem_t v = row[(int)(fin_keys[i] & 0xfffffULL)];
            out[o + i] = v;
            occ_cnt[o + i] = (int32_t)(v.s < max_occ ? v.s : max_occ);                     // FMI_search.cpp:1280-1290
        }
    }
}

// Order one read's SMEMs by (m, n) (sortSMEMs + ks_introsort(mem_intv1), bwamem.cpp:785-799; equal (m, n) are
// field-identical, so any order among them gives the same array); `out` is dense and in read order.
__global__ void __launch_bounds__(256)
k_smem_finish(int n_reads, const bm2_smem_t *__restrict__ tmp, const int32_t *__restrict__ smem_cnt,
              const int64_t *__restrict__ smem_off, int32_t max_occ, bm2_smem_t *out, int32_t *occ_cnt, int32_t *big_list, int32_t *big_cnt) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    const int n = smem_cnt[r];
    const int64_t o = smem_off[r];
    const bm2_smem_t *row = tmp + o;
    if (n > SMEM_FINISH_BIG) { big_list[atomicAdd(big_cnt, 1)] = (int32_t)r; return; }      // a whole workgroup sorts this read's SMEMs (k_smem_finish_big)
    if (n <= 16) {
        // The usual read (~12 SMEMs): the (m, n) keys in sixteen REGISTERS (m, n < 2^15: one word each; an unused place holds the largest word), every
        // record's rank by sixteen compares on them -- the loop below reads every key again for every record (n^2 loads: 1.3 ms per million reads)
        uint32_t key[16];
        _Pragma("unroll") for (int i = 0; i < 16; i++) key[i] = i < n ? (row[i].m << 16 | row[i].n) : 0xffffffffu;
        _Pragma("unroll") for (int i = 0; i < 16; i++) {
            if (i < n) {
                int rank = 0;
                _Pragma("unroll") for (int t = 0; t < 16; t++) rank += (key[t] < key[i] || (key[t] == key[i] && t < i)) ? 1 : 0;
                const bm2_smem_t v = row[i];
                out[o + rank] = v;
                occ_cnt[o + rank] = (int32_t)(v.s < max_occ ? v.s : max_occ);           // FMI_search.cpp:1280-1290
            }
        }
        return;
    }
    for (int i = 0; i < n; i++) {
        const bm2_smem_t v = row[i];
        int rank = 0;
        for (int t = 0; t < n; t++) {
            const uint32_t tm = row[t].m, tn = row[t].n;
            rank += (tm < v.m || (tm == v.m && (tn < v.n || (tn == v.n && t < i)))) ? 1 : 0;
        }
        out[o + rank] = v;
        occ_cnt[o + rank] = (int32_t)(v.s < max_occ ? v.s : max_occ);           // FMI_search.cpp:1280-1290
    }
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

// (Measured in round 6 and not kept, profiles/r06ah_*: the lanes REFILLED from their wavefront's range of lookups -- a lookup is seven LF steps on average and
//  forty for the unluckiest of a workgroup, so most of a wavefront's life a handful of its lanes walk -- 3.2-3.4 ms at 4..16 workgroups per CU against this
//  kernel's 2.8: with one line per LANE the address translations bound the fetch, and 64 walking lanes per wavefront ask for more of them at once.)
// The same walk, QUAD-COOPERATIVE like backward_ext: each lane still owns one lookup, but the CP_OCC entry a lane needs is fetched
// by its quad -- lane b loads quarter b = {count[b], bwt[b]} -- so one load instruction touches 16 lines per wave instead of 64
// (the address-translation rate, not HBM, caps one-line-per-lane access on an index of this size: tools/ubench/randline.hip).
// The lane whose quarter has the position's bit set knows the base and ranks it; a quad sum hands the result to the owner.
// All 64 lanes stay in the loop until the whole wavefront is done (the exchange needs every lane of a quad).
template <int T>
static __device__ __forceinline__ void sal_turn(const DevIndex &ix, int sub, int64_t my_sp, bool my_walk, int64_t &new_sp, int &hit) {
    const int64_t sp = qbcast64<T>(my_sp);
    const int want = qbcast32<T>(my_walk ? 1 : 0);
    ulonglong2 e = make_ulonglong2(0, 0);
    if (want) e = ((const ulonglong2 *)&ix.cp_occ[sp >> 6])[sub];             // {count[sub], bwt[sub]}
    const int y = 63 - (int)(sp & 63);
    const bool mine = want && ((e.y >> y) & 1);                                // the symbol at sp is base `sub`
    const int yy = (int)(sp & 63);
    const uint64_t msk = yy ? (~0ULL << (64 - yy)) : 0ULL;
    const int64_t nxt = mine ? pick4(sub, ix.count[0], ix.count[1], ix.count[2], ix.count[3]) + (int64_t)e.x + __popcll(e.y & msk) : 0;
    const int64_t sum = qsum64(nxt);
    const int any = (int)qsum64(mine ? 1 : 0);
    if (sub == T) { new_sp = sum; hit = any; }
}
__global__ void __launch_bounds__(256)
k_sal_quad(DevIndex ix, int64_t n, int64_t *pos_coord, unsigned long long *n_lf_out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int sub = (int)(threadIdx.x & 3);
    int64_t n_lf = 0, sp = 0, offset = 0, res = 0;
    bool walk = false;
    if (t < n) {
        sp = pos_coord[t];
        if ((sp & 7) == 0) res = ((int64_t)ix.sa_ms_byte[sp >> 3] << 32) + ix.sa_ls_word[sp >> 3];
        else walk = true;
    }
    while (__any(walk)) {
        int64_t nsp = 0; int hit = 0;
        sal_turn<0>(ix, sub, sp, walk, nsp, hit); sal_turn<1>(ix, sub, sp, walk, nsp, hit);
        sal_turn<2>(ix, sub, sp, walk, nsp, hit); sal_turn<3>(ix, sub, sp, walk, nsp, hit);
        if (walk) {
            if (!hit) { res = 0; walk = false; }                                // sentinel: 0 whatever the offset (:1230-1233)
            else {
                n_lf++; sp = nsp; offset++;
                if ((sp & 7) == 0) { res = ((int64_t)ix.sa_ms_byte[sp >> 3] << 32) + ix.sa_ls_word[sp >> 3] + offset; walk = false; }
            }
        }
    }
    if (t < n) pos_coord[t] = res;
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

int bm2_launch_seeding(bm2_ctx *c, const SeedParams &sp, int n_reads, const uint8_t *enc, const int64_t *off, const int32_t *len,
                       const SeedBufs &sb, int grid_walk, int grid_bwd, int32_t *smem_cnt, unsigned long long *sc,
                       void (*tick)(bm2_ctx *, const char *), int max_len) {
    if (bm2_side_streams(c)) return BM2_ENODEV;
    hipStream_t s = c->stream, s3 = c->side_stream[0], sh = c->side_stream[1];
    // workgroups per CU of the wavefront-per-task kernel beside k_bwd: 12 (pass 1's backward phase 14.9 -> 14.2 ms, profiles/r06e_sweep_bwd_heavy_stream_grid.json;
    // 1 / 2 / 4 / 8 per CU and a side stream on another hardware queue than the main stream's: all within 0.3 ms of each other)
    const int grid_heavy = c->n_cu * bm2_knob("BM2_BWD_HEAVY_WG", 12);
    // k_bwd hands a task that has had this many extensions over to the wavefront-per-task kernel at its next row boundary (0 = never)
    const int export_age = bm2_knob("BM2_BWD_EXPORT_AGE", BWD_EXPORT_AGE);
    // pass 3 is independent of passes 1 and 2: it runs beside them.  WHERE is launch policy (BM2_P3_AT): 0 = beside the forward walks of
    // pass 1 (both are forward-only kernels without LDS lists: they compete for the same wave slots), 1 / 2 = beside the backward kernel of
    // pass 1 / 2, whose blocks hold 48 KB of LDS survivors and leave wave slots empty that a kernel without LDS can use
    const int p3_at = bm2_knob("BM2_P3_AT", max_len >= 1000 ? 0 : 1);      // (150 bp reads, profiles/r04c: beside bwd1 34.9 instead of 36.1 ms; 10 kb reads, whose walks are
                                                                               //  ten times as long as their backward phases, r04d: 180 instead of 138 ms there)
    const int p3_bpc = bm2_knob("BM2_P3_BPC", 0);                            // workgroups per CU of pass 3 (0: as many as the other walks)
    const int grid_p3 = p3_bpc > 0 ? c->n_cu * p3_bpc : grid_walk;
    auto launch_p3 = [&]() {
        (void)hipEventRecord(c->ev_fork, s);
        (void)hipStreamWaitEvent(s3, c->ev_fork, 0);
        hipLaunchKernelGGL(k_walk<W_P3>, dim3(grid_p3), dim3(256), 0, s3, c->ix, sp, n_reads, enc, off, len, (const P2Task *)nullptr, (int64_t)0,
                           (BHead *)nullptr, (uint4 *)nullptr, (int64_t)0, (uint4 *)nullptr, 0, 0, sb.recs, sb.rec_cap, smem_cnt, sc,
                           (int32_t *)nullptr, (int64_t)0);
        (void)hipEventRecord(c->ev_join[0], s3);
    };
    if (p3_at <= 0 || p3_at > 2) launch_p3();
    for (int pass = 1; pass <= 2; pass++) {
        BHead *heads = pass == 1 ? sb.heads1 : sb.heads2;
        uint4 *ents = pass == 1 ? sb.ents1 : sb.ents2;
        const int64_t slot_cap = pass == 1 ? sb.slot1_cap : sb.slot2_cap;
        int32_t *heavy = pass == 1 ? sb.heavy1 : sb.heavy2;
        if (pass == 1)
            hipLaunchKernelGGL(k_walk<W_P1>, dim3(grid_walk), dim3(256), 0, s, c->ix, sp, n_reads, enc, off, len, (const P2Task *)nullptr, (int64_t)0,
                               heads, ents, slot_cap, sb.pool, sb.pool_cap, sb.pool_slots, sb.recs, sb.rec_cap, smem_cnt, sc, heavy, sb.heavy_cap);
        else
            hipLaunchKernelGGL(k_walk<W_P2>, dim3(grid_walk), dim3(256), 0, s, c->ix, sp, n_reads, enc, off, len, sb.tasks, sb.task_cap,
                               heads, ents, slot_cap, sb.pool, sb.pool_cap, sb.pool_slots, sb.recs, sb.rec_cap, smem_cnt, sc, heavy, sb.heavy_cap);
        tick(c, pass == 1 ? "smem.walk1" : "smem.walk2");
        if (p3_at == pass) launch_p3();
        // the long lists go to one wavefront each, beside the lane-per-task kernel
        {       // (after k_bwd instead of beside it: +1 ms per pass, profiles/r05b_sweep.json)
            (void)hipEventRecord(c->ev_join[2], s);
            (void)hipStreamWaitEvent(sh, c->ev_join[2], 0);
            hipLaunchKernelGGL(k_bwd_heavy, dim3(grid_heavy), dim3(256), 0, sh, c->ix, sp, pass, enc, heads, ents, slot_cap, sb.pool, sb.pool_cap,
                               sb.pool_slots, heavy, sb.heavy_cap, sb.recs, sb.rec_cap, sb.tasks, sb.task_cap, smem_cnt, sc);
            (void)hipEventRecord(c->ev_join[1], sh);
        }
        const int lc = bm2_knob("BM2_BWD_LCAP", LCAP), wpe = bm2_knob("BM2_BWD_WAVES", 4);
        auto kb = wpe >= 5 ? (lc <= 4 ? k_bwd5<4> : k_bwd5<6>)
                           : (lc <= 4 ? k_bwd<4> : lc <= 6 ? k_bwd<6> : lc <= 8 ? k_bwd<8> : k_bwd<LCAP>);
        CTask *cont = (CTask *)(pass == 1 ? sb.cont1 : sb.cont2);
        hipLaunchKernelGGL(kb, dim3(grid_bwd), dim3(256), 0, s, c->ix, sp, pass, enc, heads, ents, slot_cap, sb.pool, sb.pool_cap,
                           sb.pool_slots, sb.recs, sb.rec_cap, sb.tasks, sb.task_cap, smem_cnt, sc, cont, sb.cont_cap, export_age);
        (void)hipStreamWaitEvent(s, c->ev_join[1], 0);
        tick(c, pass == 1 ? "smem.bwd1" : "smem.bwd2");
        if (export_age > 0) {                                   // the tasks k_bwd handed over, sixteen lanes each (they may file pass-2 tasks: before k_walk<P2>)
            hipLaunchKernelGGL(k_bwd_cont, dim3(c->n_cu * bm2_knob("BM2_BWD_CONT_BPC", 6)), dim3(256), 0, s, c->ix, sp, pass, enc, ents, slot_cap, sb.pool, sb.pool_cap,
                               sb.pool_slots, cont, sb.cont_cap, sb.recs, sb.rec_cap, sb.tasks, sb.task_cap, smem_cnt, sc);
            tick(c, pass == 1 ? "smem.cont1" : "smem.cont2");
        }
    }
    (void)hipStreamWaitEvent(s, c->ev_join[0], 0);
    return bm2_check(hipGetLastError(), "seeding launch");
}
int bm2_launch_smem_finish(bm2_ctx *c, int n_reads, const SeedBufs &sb, const unsigned long long *sc, const int32_t *smem_cnt,
                           const int64_t *smem_off, int32_t *fill, bm2_smem_t *tmp, int32_t max_occ, bm2_smem_t *out, int32_t *occ_cnt, int max_len) {
    if (n_reads <= 0) return BM2_OK;
    hipLaunchKernelGGL(k_rec_scatter, dim3(c->n_cu * 8), dim3(256), 0, c->stream, sb.recs, sb.rec_cap, sc, smem_off, fill, tmp);
    // (`fill` is through after the scatter: its second half lists the reads that go to the workgroup-per-read sort; [2 n + 2], [2 n + 3] = count, cursor)
    int32_t *big_list = fill + n_reads + 1, *big_cnt = fill + 2 * (int64_t)n_reads + 2, *big_cur = big_cnt + 1;
    hipLaunchKernelGGL(k_smem_finish, dim3((n_reads + 255) / 256), dim3(256), 0, c->stream, n_reads, tmp, smem_cnt, smem_off, max_occ, out, occ_cnt, big_list, big_cnt);
    int lds_keys = bm2_knob("BM2_SMEM_SORT_KEYS", SMEM_FINISH_LDS_KEYS);     // (a test hook: small values force the m-class runs on ordinary reads)
    if (lds_keys < 64) lds_keys = 64;
    if (lds_keys > SMEM_FINISH_LDS_KEYS) lds_keys = SMEM_FINISH_LDS_KEYS;
    // (the sort key packs m, n and the record's index into 20 bits each: reads of 2^18 bases and more -- up to 3 SMEMs per base -- take the
    //  kernel's one-lane path, which compares the fields themselves)
    const int keys_arg = max_len < (1 << 18) ? lds_keys : 0;
    const size_t lds = (size_t)SMEM_FINISH_LDS_KEYS * 8;
    { const int rc_a = bm2_raise_lds_limit(c, 1, (const void *)k_smem_finish_big, lds); if (rc_a) return rc_a; }
    hipLaunchKernelGGL(k_smem_finish_big, dim3(c->n_cu), dim3(256), lds, c->stream, tmp, smem_cnt, smem_off, max_occ, out, occ_cnt, big_list, big_cnt, big_cur, keys_arg);
    return bm2_check(hipGetLastError(), "k_smem_finish launch");
}
int bm2_seed_sizes(size_t *head, size_t *ent, size_t *task, int *n_sc, size_t *ctask) {
    *head = sizeof(BHead); *ent = (size_t)CAPF * 16; *task = sizeof(P2Task); *n_sc = SC_N + 10; *ctask = sizeof(CTask);
    return CAPF;
}
int bm2_launch_sal_expand(bm2_ctx *c, const bm2_smem_t *smems, int64_t n_smem, const int64_t *sa_off, int32_t max_occ, int64_t *pos) {
    if (n_smem <= 0) return BM2_OK;
    hipLaunchKernelGGL(k_sal_expand, dim3((unsigned)((n_smem + 255) / 256)), dim3(256), 0, c->stream, smems, n_smem, sa_off, max_occ, pos);
    return bm2_check(hipGetLastError(), "k_sal_expand launch");
}
int bm2_launch_sal(bm2_ctx *c, int64_t n, int64_t *pos_coord, unsigned long long *n_lf) {
    if (n <= 0) return BM2_OK;
    if (bm2_knob("BM2_SAL_QUAD", 0)) hipLaunchKernelGGL(k_sal_quad, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, c->ix, n, pos_coord, n_lf);
    else hipLaunchKernelGGL(k_sal, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, c->ix, n, pos_coord, n_lf);
    return bm2_check(hipGetLastError(), "k_sal launch");
}
int bm2_launch_smem_gather(bm2_ctx *c, int n_reads, const bm2_smem_t *in, const int64_t *in_off, const int32_t *cnt,
                           const int64_t *out_off, bm2_smem_t *out) {
    if (n_reads <= 0) return BM2_OK;
    hipLaunchKernelGGL(k_smem_gather, dim3((n_reads + 255) / 256), dim3(256), 0, c->stream, n_reads, in, in_off, cnt, out_off, out);
    return bm2_check(hipGetLastError(), "k_smem_gather launch");
}
