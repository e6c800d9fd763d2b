// chain.hip -- per-read sequential stages between seeding and extension, one read per lane:
//   mem_chain_seeds  (bwamem.cpp:806-974)   seeds -> chains through the klib B-tree (exact node-level emulation)
//   mem_chain_flt    (bwamem.cpp:506-624)   weight, klib introsort by weight, overlap filter
//   task building of mem_chain2aln_across_reads_V2 (bwamem.cpp:2127-2438): reference window, seed order, regs
// and, after the extension kernel, the redundant-seed post-filter (bwamem.cpp:2895-2989).
//
// These stages are branchy pointer-chasing over a handful of records per read (measured: 1.2 chains, 3 seeds per
// 150 bp read); they are kept on the device so a chunk never leaves HBM between seeding and extension.  All
// per-read storage is carved from arrays indexed by the read's SA range [sa_beg, sa_beg+n_sa): a read can never
// have more seeds, chains or regs than SA coordinates.
#include "pipeline.h"
#include "chain_dev.h"
#include "ksort_dev.h"

#define BM2_CHAIN_TIERS 8          // stream / event slots kept for the tier launches (five tiers are launched)
#define CHAIN_CUR_SLOTS 5           // work cursors item_cur[0..4]: the first five tiers; [5], [6]: the overflow / island launch; further tiers: item_cur[CHAIN_CUR_EXTRA ..]
#define CHAIN_CUR_EXTRA 30          // (= counters[40..42] of the batch: pipeline.hip)

// ---------------------------------------------------------------- bntseq helpers (bntseq.cpp:378-402, bntseq.h:87-90)
static __device__ __forceinline__ int64_t depos(const DevIndex &ix, int64_t pos, int &is_rev) {
    is_rev = pos >= ix.l_pac;
    return is_rev ? (ix.l_pac << 1) - 1 - pos : pos;
}
static __device__ int pos2rid(const DevIndex &ix, int64_t pos_f) {
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
// bns_intv2rid with a one-entry cache: consecutive seeds of a read almost always fall in the same contig and strand, so
// the two binary searches over the contig table are skipped when [rb, re) lies inside the cached contig's span
struct RidCache { int64_t lo, hi; int rid; };
static __device__ int intv2rid(const DevIndex &ix, int64_t rb, int64_t re);
static __device__ __forceinline__ int intv2rid_cached(const DevIndex &ix, int64_t rb, int64_t re, RidCache &rc) {
    if (rb >= rc.lo && re <= rc.hi && rb < re) return rc.rid;
    const int rid = intv2rid(ix, rb, re);
    if (rid >= 0) {
        const int64_t o = ix.ann_offset[rid], e = o + ix.ann_len[rid];
        if (rb >= ix.l_pac) { rc.lo = (ix.l_pac << 1) - e; rc.hi = (ix.l_pac << 1) - o; }
        else { rc.lo = o; rc.hi = e; }
        rc.rid = rid;
    }
    return rid;
}
static __device__ int intv2rid(const DevIndex &ix, int64_t rb, int64_t re) {
    int is_rev;
    if (rb < ix.l_pac && re > ix.l_pac) return -2;
    const int rid_b = pos2rid(ix, depos(ix, rb, is_rev));
    const int rid_e = rb < re ? pos2rid(ix, depos(ix, re - 1, is_rev)) : rid_b;
    return rid_b == rid_e ? rid_b : -1;
}

// cal_max_gap, bwamem.cpp:66-76
static __device__ __forceinline__ int cal_max_gap(const ChainParams &o, int qlen) {
    // (int)((double)(qlen*a - o)/e + 1.) == (qlen*a - o + e) / e in C integer division (both truncate toward zero, e > 0);
    // the double division of the reference would cost ~100 instructions per call on the GPU
    const int nd = qlen * o.a - o.o_del + o.e_del, ni = qlen * o.a - o.o_ins + o.e_ins;
    const int l_del = o.e_del == 1 ? nd : nd / o.e_del;
    const int l_ins = o.e_ins == 1 ? ni : ni / o.e_ins;
    int l = l_del > l_ins ? l_del : l_ins;
    l = l > 1 ? l : 1;
    return l < o.w << 1 ? l : o.w << 1;
}

// ---------------------------------------------------------------- klib B-tree, t = 5 (kbtree.h; kb_init(chn, 512+8), 48-byte keys)
#define BT_T 5
// reg: nodes in GLOBAL memory, visited through registers (below).  lnodes (the SP = "split pool" instantiations, k_chain_serial): the INTERNAL nodes live in a
// pool of their own in LDS while it has room -- a node's number then carries BT_LDS_FLAG -- and only the leaves in global memory: a descent of five levels is
// one global round trip instead of five.  (Node numbers are this file's own; kbtree's results depend on the tree's shape only.)
#define BT_LDS_FLAG 0x40000000
struct BTree { BtNode *nodes; int n_nodes, root, n_keys; const WChain *ch; bool reg = false; BtNode *lnodes = nullptr; int l_cap = 0, n_l = 0; };
template <bool SP> static __device__ __forceinline__ BtNode *bt_at(const BTree &b, int x) {
    if constexpr (SP) { if (x & BT_LDS_FLAG) return b.lnodes + (x & ~BT_LDS_FLAG); }
    return b.nodes + x;
}

template <bool SP = false> static __device__ __forceinline__ int bt_new(BTree &b, int internal) {
    if constexpr (SP) {
        if (internal && b.n_l < b.l_cap) { BtNode &z = b.lnodes[b.n_l]; z.n = 0; z.is_internal = 1; return BT_LDS_FLAG | b.n_l++; }
    }
    BtNode &z = b.nodes[b.n_nodes];
    z.n = 0; z.is_internal = internal;
    return b.n_nodes++;
}
// __kb_getp_aux, kbtree.h:124-138
static __device__ int bt_getp_aux(const BTree &b, const BtNode &x, int64_t k, int &r) {
    int begin = 0, end = x.n;
    if (x.n == 0) return -1;
    while (begin < end) {
        const int mid = (begin + end) >> 1;
        if (x.kpos[mid] < k) begin = mid + 1; else end = mid;
    }
    if (begin == x.n) { r = 1; return x.n - 1; }
    const int64_t kp = x.kpos[begin];
    r = (kp < k) - (k < kp);
    if (r < 0) --begin;
    return begin;
}
// ---- a node visited through REGISTERS.  In global memory every probe of a node is a dependent load: the binary search of __kb_getp_aux makes
// three or four per level, kb_putp two more for the child's fill and pointer -- sixty-odd dependent loads per seed over five levels, a third of
// them misses.  Here a node comes in with TEN 16-byte loads issued back to back (one latency), the search is nine compares on registers, the
// fields it then needs are picked by select chains; an insertion into a leaf shifts the keys in registers and stores the node whole.
// (The ten quads are pinned by an empty asm over every component right after the loads: without it the compiler keeps the 160-byte copy in
// SCRATCH and turns every constant-index read back into a memory access -- round 3's attempt at this, profiles/r03u_*.)
struct RNode { uint4 q[10]; };
static_assert(sizeof(BtNode) == 160, "RNode mirrors BtNode: kpos[9] at dwords 0..17, key[9] at 18..26, ptr[10] at 27..36, is_internal 37, n 38");
static __device__ __forceinline__ void rn_load(RNode &nd, const BtNode *p) {
    const uint4 *s = (const uint4 *)p;
#pragma unroll
    for (int i = 0; i < 10; i++) nd.q[i] = s[i];
#pragma unroll
    for (int i = 0; i < 10; i++) asm volatile("" : "+v"(nd.q[i].x), "+v"(nd.q[i].y), "+v"(nd.q[i].z), "+v"(nd.q[i].w));
}
static __device__ __forceinline__ void rn_store(const RNode &nd, BtNode *p) {
    uint4 *d = (uint4 *)p;
#pragma unroll
    for (int i = 0; i < 10; i++) d[i] = nd.q[i];
}
template <int D> static __device__ __forceinline__ uint32_t &rn_dw(RNode &nd) {
    return (D & 3) == 0 ? nd.q[D >> 2].x : (D & 3) == 1 ? nd.q[D >> 2].y : (D & 3) == 2 ? nd.q[D >> 2].z : nd.q[D >> 2].w;
}
template <int T> static __device__ __forceinline__ int64_t rn_kpos(RNode &nd) { return (int64_t)((uint64_t)rn_dw<2 * T + 1>(nd) << 32 | rn_dw<2 * T>(nd)); }
template <int T> static __device__ __forceinline__ void rn_set_kpos(RNode &nd, int64_t v) { rn_dw<2 * T>(nd) = (uint32_t)v; rn_dw<2 * T + 1>(nd) = (uint32_t)((uint64_t)v >> 32); }
#define RN_FOR9(F) F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7) F(8)
#define RN_FOR10(F) RN_FOR9(F) F(9)
static __device__ __forceinline__ int rn_n(RNode &nd) { return (int)rn_dw<38>(nd); }
static __device__ __forceinline__ int rn_internal(RNode &nd) { return (int)rn_dw<37>(nd); }
static __device__ __forceinline__ int rn_key(RNode &nd, int i) {                 // key[i], 0 <= i < 9
    int v = 0;
#define RN_PICK(T) v = i == T ? (int)rn_dw<18 + T>(nd) : v;
    RN_FOR9(RN_PICK)
#undef RN_PICK
    return v;
}
static __device__ __forceinline__ int rn_ptr(RNode &nd, int i) {                 // ptr[i], 0 <= i < 10
    int v = 0;
#define RN_PICK(T) v = i == T ? (int)rn_dw<27 + T>(nd) : v;
    RN_FOR10(RN_PICK)
#undef RN_PICK
    return v;
}
static __device__ __forceinline__ int64_t rn_kpos_at(RNode &nd, int i) {
    int64_t v = 0;
#define RN_PICK(T) v = i == T ? rn_kpos<T>(nd) : v;
    RN_FOR9(RN_PICK)
#undef RN_PICK
    return v;
}
// __kb_getp_aux on a register node: the first key >= k is at index `begin` = the number of keys below k (the keys of a node ascend)
static __device__ __forceinline__ int rn_getp(RNode &nd, int64_t k, int &r) {
    const int n = rn_n(nd);
    if (n == 0) return -1;
    int begin = 0;
#define RN_CNT(T) begin += (T < n && rn_kpos<T>(nd) < k) ? 1 : 0;
    RN_FOR9(RN_CNT)
#undef RN_CNT
    if (begin == n) { r = 1; return n - 1; }
    const int64_t kp = rn_kpos_at(nd, begin);
    r = (kp < k) - (k < kp);
    if (r < 0) --begin;
    return begin;
}
template <bool SP = false> static __device__ int bt_lower_reg(const BTree &b, int64_t k) {
    int lower = -1, x = b.root, r = 0;
    while (x >= 0) {
        RNode nd; rn_load(nd, bt_at<SP>(b, x));
        const int i = rn_getp(nd, k, r);
        if (i >= 0 && r == 0) return rn_key(nd, i);
        if (i >= 0) lower = rn_key(nd, i);
        if (!rn_internal(nd)) return lower;
        x = rn_ptr(nd, i + 1);
    }
    return lower;
}
template <bool SP> static __device__ void bt_split(BTree &b, int xi, int i, int yi);
// the descent from node xi, already in registers (x), to the leaf the key goes into
template <bool SP> static __device__ void bt_put_reg_tail(BTree &b, int key, int64_t k, RNode &x, int xi) {
    int r;
    for (;;) {
        if (!rn_internal(x)) {                               // leaf: keys above i move up by one, the new key goes to i + 1
            const int i = rn_getp(x, k, r), at = i + 1;
#define RN_SHIFT(T) if (8 - T > at) { rn_set_kpos<8 - T>(x, rn_kpos<(8 - T > 0 ? 8 - T - 1 : 0)>(x)); rn_dw<18 + 8 - T>(x) = rn_dw<18 + (8 - T > 0 ? 8 - T - 1 : 0)>(x); }
            RN_FOR9(RN_SHIFT)                                // t = 8 .. 0, descending: every entry takes its lower neighbour's OLD value
#undef RN_SHIFT
#define RN_PUT(T) if (at == T) { rn_set_kpos<T>(x, k); rn_dw<18 + T>(x) = (uint32_t)key; }
            RN_FOR9(RN_PUT)
#undef RN_PUT
            rn_dw<38>(x) = (uint32_t)(rn_n(x) + 1);
            rn_store(x, bt_at<SP>(b, xi));
            return;
        }
        int i = rn_getp(x, k, r) + 1;
        int child = rn_ptr(x, i);
        RNode y; rn_load(y, bt_at<SP>(b, child));
        if (rn_n(y) == 2 * BT_T - 1) {                       // (once per ~5 insertions: the split works on memory, parent and child are read again)
            bt_split<SP>(b, xi, i, child);
            if (k > bt_at<SP>(b, xi)->kpos[i]) ++i;
            child = bt_at<SP>(b, xi)->ptr[i];
            rn_load(y, bt_at<SP>(b, child));
        }
        x = y; xi = child;
    }
}
template <bool SP = false> static __device__ void bt_put_reg(BTree &b, int key, int64_t k) {
    ++b.n_keys;
    RNode x; rn_load(x, bt_at<SP>(b, b.root));                // (the node's own fill comes with it: no separate load for the "is it full" test)
    if (rn_n(x) == 2 * BT_T - 1) {
        const int s = bt_new<SP>(b, 1), r = b.root;
        b.root = s; bt_at<SP>(b, s)->ptr[0] = r;
        bt_split<SP>(b, s, 0, r);
        rn_load(x, bt_at<SP>(b, b.root));
    }
    bt_put_reg_tail<SP>(b, key, k, x, b.root);
}

// kb_intervalp, lower bound only (kbtree.h:158-175)
template <bool SP = false> static __device__ int bt_lower(const BTree &b, int64_t k) {
    if (b.reg) return bt_lower_reg<SP>(b, k);
    int lower = -1, x = b.root, r = 0;
    while (x >= 0) {
        const BtNode &nd = *bt_at<SP>(b, x);
        const int i = bt_getp_aux(b, nd, k, r);
        if (i >= 0 && r == 0) return nd.key[i];
        if (i >= 0) lower = nd.key[i];
        if (!nd.is_internal) return lower;
        x = nd.ptr[i + 1];
    }
    return lower;
}
// __kb_split, kbtree.h:179-196
template <bool SP = false> static __device__ void bt_split(BTree &b, int xi, int i, int yi) {
    const int zi = bt_new<SP>(b, bt_at<SP>(b, yi)->is_internal);
    BtNode &x = *bt_at<SP>(b, xi), &y = *bt_at<SP>(b, yi), &z = *bt_at<SP>(b, zi);
    z.n = BT_T - 1;
    for (int t = 0; t < BT_T - 1; t++) { z.key[t] = y.key[BT_T + t]; z.kpos[t] = y.kpos[BT_T + t]; }
    if (y.is_internal) for (int t = 0; t < BT_T; t++) z.ptr[t] = y.ptr[BT_T + t];
    y.n = BT_T - 1;
    for (int t = x.n; t > i; t--) x.ptr[t + 1] = x.ptr[t];
    x.ptr[i + 1] = zi;
    for (int t = x.n - 1; t >= i; t--) { x.key[t + 1] = x.key[t]; x.kpos[t + 1] = x.kpos[t]; }
    x.key[i] = y.key[BT_T - 1]; x.kpos[i] = y.kpos[BT_T - 1];
    ++x.n;
}
// kb_putp + __kb_putp_aux, kbtree.h:197-231 (the recursion is a plain descent)
// (k = the key's sort field, ch[key].pos: the caller has it in a register -- it has just written the chain)
template <bool SP = false> static __device__ void bt_put(BTree &b, int key, int64_t k) {
    if (b.reg) { bt_put_reg<SP>(b, key, k); return; }
    ++b.n_keys;
    if (bt_at<SP>(b, b.root)->n == 2 * BT_T - 1) {
        const int s = bt_new<SP>(b, 1), r = b.root;
        b.root = s; bt_at<SP>(b, s)->ptr[0] = r;
        bt_split<SP>(b, s, 0, r);
    }
    int xi = b.root, r;
    for (;;) {
        BtNode &x = *bt_at<SP>(b, xi);
        if (!x.is_internal) {
            const int i = bt_getp_aux(b, x, k, r);
            for (int t = x.n - 1; t > i; t--) { x.key[t + 1] = x.key[t]; x.kpos[t + 1] = x.kpos[t]; }
            x.key[i + 1] = key; x.kpos[i + 1] = k;
            ++x.n;
            return;
        }
        int i = bt_getp_aux(b, x, k, r) + 1;
        if (bt_at<SP>(b, x.ptr[i])->n == 2 * BT_T - 1) {
            bt_split<SP>(b, xi, i, x.ptr[i]);
            if (k > x.kpos[i]) ++i;
        }
        xi = x.ptr[i];
    }
}
// The split pool's own descent (k_chain_serial): a node in LDS is probed IN PLACE -- the binary search of __kb_getp_aux is three or four LDS reads, where
// the register form pays ten 16-byte loads and ~130 instructions of compares and select chains per level -- and the first node in global memory (the leaf,
// as a rule) comes into registers once and the register form takes over.
static __device__ void bt_put_hyb(BTree &b, int key, int64_t k) {
    ++b.n_keys;
    if (bt_at<true>(b, b.root)->n == 2 * BT_T - 1) {
        const int s = bt_new<true>(b, 1), r = b.root;
        b.root = s; bt_at<true>(b, s)->ptr[0] = r;
        bt_split<true>(b, s, 0, r);
    }
    int xi = b.root, r;
    while (xi & BT_LDS_FLAG) {                               // (a node of the LDS pool is internal)
        BtNode &x = b.lnodes[xi & ~BT_LDS_FLAG];
        int i = bt_getp_aux(b, x, k, r) + 1;
        int child = x.ptr[i];
        if (child & BT_LDS_FLAG) {
            if (b.lnodes[child & ~BT_LDS_FLAG].n == 2 * BT_T - 1) {
                bt_split<true>(b, xi, i, child);
                if (k > x.kpos[i]) ++i;
                child = x.ptr[i];
            }
            xi = child;
        } else {
            RNode y; rn_load(y, b.nodes + child);
            if (rn_n(y) == 2 * BT_T - 1) {
                bt_split<true>(b, xi, i, child);
                if (k > x.kpos[i]) ++i;
                child = x.ptr[i];
                rn_load(y, bt_at<true>(b, child));
            }
            bt_put_reg_tail<true>(b, key, k, y, child);
            return;
        }
    }
    RNode x; rn_load(x, b.nodes + xi);                       // (the root is a leaf still, or the pool was full when it was made)
    bt_put_reg_tail<true>(b, key, k, x, xi);
}
static __device__ int bt_lower_hyb(const BTree &b, int64_t k) {
    int lower = -1, x = b.root, r = 0;
    while (x >= 0) {
        if (x & BT_LDS_FLAG) {
            const BtNode &nd = b.lnodes[x & ~BT_LDS_FLAG];
            const int i = bt_getp_aux(b, nd, k, r);
            if (i >= 0 && r == 0) return nd.key[i];
            if (i >= 0) lower = nd.key[i];
            x = nd.ptr[i + 1];
        } else {
            RNode nd; rn_load(nd, b.nodes + x);
            const int i = rn_getp(nd, k, r);
            if (i >= 0 && r == 0) return rn_key(nd, i);
            if (i >= 0) lower = rn_key(nd, i);
            if (!rn_internal(nd)) return lower;
            x = rn_ptr(nd, i + 1);
        }
    }
    return lower;
}
// __kb_traverse (in-order), kbtree.h:343-366
template <bool SP = false> static __device__ int bt_traverse(const BTree &b, int32_t *out) {
    int n = 0, sp = 0;
    int stk_node[24], stk_i[24];
    stk_node[0] = b.root; stk_i[0] = 0;
    while (sp >= 0) {
        const BtNode &nd = *bt_at<SP>(b, stk_node[sp]);
        const int i = stk_i[sp];
        if (nd.is_internal) {
            if (i <= nd.n) {
                if (i > 0 && i <= nd.n) { /* key i-1 is emitted when we come back from child i-1 */ }
                // descend into child i, after emitting key i-1
                if (i > 0) out[n++] = nd.key[i - 1];
                stk_i[sp] = i + 1;
                ++sp; stk_node[sp] = nd.ptr[i]; stk_i[sp] = 0;
            } else --sp;
        } else {
            for (int t = 0; t < nd.n; t++) out[n++] = nd.key[t];
            --sp;
        }
    }
    return n;
}

// ---------------------------------------------------------------- chaining of one read
// test_and_merge, bwamem.cpp:357-399
static __device__ int test_and_merge(const ChainParams &o, int64_t l_pac, WChain &c, const WSeed &p, int seed_rid,
                                     WSeed *seeds, int si) {
    const int64_t qend = c.last_qbeg + c.last_len, rend = c.last_rbeg + c.last_len;
    if (seed_rid != c.rid) return 0;
    if (p.qbeg >= c.first_qbeg && p.qbeg + p.len <= qend && p.rbeg >= c.pos && p.rbeg + p.len <= rend) return 1;
    if ((c.last_rbeg < l_pac || c.pos < l_pac) && p.rbeg >= l_pac) return 0;
    const int64_t x = p.qbeg - c.last_qbeg, y = p.rbeg - c.last_rbeg;
    if (y >= 0 && x - y <= o.w && y - x <= o.w && x - c.last_len < o.max_chain_gap && y - c.last_len < o.max_chain_gap) {
        seeds[si] = p; seeds[si].next = -1;
        seeds[c.tail].next = si;
        c.tail = si; c.n++;
        c.last_rbeg = p.rbeg; c.last_qbeg = p.qbeg; c.last_len = p.len;
        return 2;      // merged and consumed the seed slot
    }
    return 0;
}

// mem_chain_weight, bwamem.cpp:429-448
static __device__ int chain_weight(const WChain &c, const WSeed *seeds) {
    int64_t end = 0; int w = 0, tmp;
    for (int si = c.head; si >= 0; si = seeds[si].next) {
        const WSeed &s = seeds[si];
        if (s.qbeg >= end) w += s.len;
        else if (s.qbeg + s.len > end) w += (int)(s.qbeg + s.len - end);
        end = end > s.qbeg + s.len ? end : s.qbeg + s.len;
    }
    tmp = w; w = 0; end = 0;
    for (int si = c.head; si >= 0; si = seeds[si].next) {
        const WSeed &s = seeds[si];
        if (s.rbeg >= end) w += s.len;
        else if (s.rbeg + s.len > end) w += (int)(s.rbeg + s.len - end);
        end = end > s.rbeg + s.len ? end : s.rbeg + s.len;
    }
    w = w < tmp ? w : tmp;
    return w < 1 << 30 ? w : (1 << 30) - 1;
}

// The task-building part of mem_chain2aln_across_reads_V2 (bwamem.cpp:2127-2223) for ONE kept chain whose seeds [s0, s0 + d.n) of the read's slice `os`
// are final: reference window, extension order (srt keeps the ascending one), reg slots from n_reg on.  Used by k_chain_finish and -- for reads no seed
// filter can touch -- by the lane that has just emitted the chain (chain_finish_read: the chain and its seeds are still in its registers / L1).
static __device__ __forceinline__ void chain_finish_one(const DevIndex &ix, const ChainParams &o, int l_query, int64_t base, int i, DevChain &d, DevSeed *os,
                                                        int s0, int32_t *srt_out, int32_t *reg_seed, int32_t *reg_chain, int &n_reg) {
    const int n_seed = s0 + d.n;
    d.reg0 = n_reg;
    if (d.n == 0) return;                               // bwamem.cpp:2140 (a chain emptied by the seed filter)
    // reference window of the chain, bwamem.cpp:2145-2172 (rmax, strand clip, bns_fetch_seq_v2 contig clip)
    int64_t rmax0 = ix.l_pac << 1, rmax1 = 0;
    for (int t = s0; t < n_seed; t++) {
        const DevSeed &sx = os[t];
        const int64_t bb = sx.rbeg - (sx.qbeg + cal_max_gap(o, sx.qbeg));
        const int64_t ee = sx.rbeg + sx.len + ((l_query - sx.qbeg - sx.len) + cal_max_gap(o, l_query - sx.qbeg - sx.len));
        rmax0 = rmax0 < bb ? rmax0 : bb;
        rmax1 = rmax1 > ee ? rmax1 : ee;
    }
    rmax0 = rmax0 > 0 ? rmax0 : 0;
    rmax1 = rmax1 < ix.l_pac << 1 ? rmax1 : ix.l_pac << 1;
    if (rmax0 < ix.l_pac && ix.l_pac < rmax1) {
        if (os[s0].rbeg < ix.l_pac) rmax1 = ix.l_pac; else rmax0 = ix.l_pac;
    }
    {
        int is_rev;
        const int rid = pos2rid(ix, depos(ix, os[s0].rbeg, is_rev));
        int64_t far_beg = ix.ann_offset[rid], far_end = far_beg + ix.ann_len[rid];
        if (is_rev) { const int64_t tmp = far_beg; far_beg = (ix.l_pac << 1) - far_end; far_end = (ix.l_pac << 1) - tmp; }
        rmax0 = rmax0 > far_beg ? rmax0 : far_beg;
        rmax1 = rmax1 < far_end ? rmax1 : far_end;
    }
    d.rmax0 = rmax0; d.rmax1 = rmax1;
    // seeds are extended in descending (score<<32 | index) order, bwamem.cpp:2188-2206; srt keeps the ascending order
    int32_t *srt = srt_out + base + s0;
    for (int t = 0; t < d.n; t++) srt[t] = t;
    if (d.n > 1) {
        const DevSeed *cs = os + s0;
        k_introsort_flat(d.n, srt, [&](int32_t x, int32_t y) { return cs[x].score < cs[y].score || (cs[x].score == cs[y].score && x < y); });
    }
    for (int kk = d.n - 1; kk >= 0; kk--) {
        const int reg = n_reg++;
        os[s0 + srt[kk]].aln = reg;
        reg_seed[base + reg] = (int32_t)(s0 + srt[kk]);       // seed index relative to the read's base
        reg_chain[base + reg] = i;                             // chain index relative to the read's base
    }
}

// The rest of mem_chain_flt (bwamem.cpp:548-624) and the hand-over to the extension stage, for the n chains ord[0..n) -- the chains that
// passed the weight test, in key order -- of read r: introsort by weight, overlap filter, kept chains with their seeds made contiguous.
// kept_list: scratch for n ints (the B-tree's nodes are no longer needed when this runs).
static __device__ void chain_finish_read(const ChainParams &o, int r, WChain *ch, WSeed *sd, int32_t *ord, int32_t *kept_list, int n, int64_t base,
                                         float frac_rep, DevChain *chn, DevSeed *seeds_out, int32_t *seed_owner, int32_t *n_chain_out, int32_t *n_reg_out,
                                         const DevIndex *fix = nullptr, const FinishOut *fo = nullptr /* both set: this lane does k_chain_finish's part too */) {
    int k;
    if (n > 0) {
        k_introsort_flat(n, ord, [&](int32_t x, int32_t y) { return ch[x].w > ch[y].w; });     // flt_lt, bwamem.cpp:61
        int n_kept = 0;
        ch[ord[0]].kept = 3;
        kept_list[n_kept++] = 0;
        for (int i = 1; i < n; ++i) {
            WChain &ci = ch[ord[i]];
            const int beg_i = ci.first_qbeg, end_i = ci.last_qbeg + ci.last_len;
            int large_ovlp = 0, kk;
            for (kk = 0; kk < n_kept; ++kk) {
                const int j = kept_list[kk];
                WChain &cj = ch[ord[j]];
                const int beg_j = cj.first_qbeg, end_j = cj.last_qbeg + cj.last_len;
                const int b_max = beg_j > beg_i ? beg_j : beg_i;
                const int e_min = end_j < end_i ? end_j : end_i;
                if (e_min > b_max && (!cj.is_alt || ci.is_alt)) {
                    const int li = end_i - beg_i, lj = end_j - beg_j;
                    const int min_l = li < lj ? li : lj;
                    if (e_min - b_max >= min_l * o.mask_level && min_l < o.max_chain_gap) {
                        large_ovlp = 1;
                        if (cj.first < 0) cj.first = i;
                        if (ci.w < cj.w * o.drop_ratio && cj.w - ci.w >= o.min_seed_len << 1) break;
                    }
                }
            }
            if (kk == n_kept) { kept_list[n_kept++] = i; ci.kept = large_ovlp ? 2 : 3; }
        }
        for (int i = 0; i < n_kept; ++i) {
            const WChain &c = ch[ord[kept_list[i]]];
            if (c.first >= 0) ch[ord[c.first]].kept = 1;
        }
        int i2;
        for (i2 = k = 0; i2 < n; ++i2) {
            const int kp = ch[ord[i2]].kept;
            if (kp == 0 || kp == 3) continue;
            if (++k >= o.max_chain_extend) break;
        }
        for (; i2 < n; ++i2) if (ch[ord[i2]].kept < 3) ch[ord[i2]].kept = 0;
        for (i2 = k = 0; i2 < n; ++i2) if (ch[ord[i2]].kept != 0) ord[k++] = ord[i2];
        n = k;
    }
    // ---- emit kept chains with contiguous seeds
    DevChain *oc = chn + base;
    DevSeed *os = seeds_out + base;
    int n_seed = 0, n_reg = 0;
    for (int i = 0; i < n; i++) {
        const WChain &c = ch[ord[i]];
        DevChain d;
        d.pos = c.pos; d.seed_off = base + n_seed; d.n = c.n; d.rid = c.rid; d.w = c.w; d.kept = c.kept; d.first = c.first;
        d.is_alt = c.is_alt; d.read = r; d.frac_rep = frac_rep; d.rmax0 = 0; d.rmax1 = 0;
        const int s0 = n_seed;
        for (int si = c.head; si >= 0; si = sd[si].next) {
            DevSeed s; s.rbeg = sd[si].rbeg; s.qbeg = sd[si].qbeg; s.len = sd[si].len; s.score = sd[si].len; s.aln = -1;
            os[n_seed++] = s;
        }
        d.reg0 = 0; d.pad = 0;
        for (int t = s0; t < n_seed; t++) seed_owner[base + t] = r;
        if (fo) chain_finish_one(*fix, o, fo->len[r], base, i, d, os, s0, fo->srt_out, fo->reg_seed, fo->reg_chain, n_reg);
        oc[i] = d;
    }
    n_chain_out[r] = n;
    n_reg_out[r] = n_reg;          // (without `fo`: 0, set by k_chain_finish)
}

// ---- mem_chain_flt again, for a WHOLE WAVEFRONT (BM2_CHAIN_COOP_FLT; chain_finish_read above is what one lane runs and stays as it is): the walk over
// the kept chains is quadratic in the chains of a repeat-rich read -- 64 kept chains at a time, one per lane (flt_kept_coop); the weight sort before it
// and the rest + the read's output after it (flt_rest_emit: the same statements as in chain_finish_read) are lane 0's.
static __device__ __forceinline__ void flt_sort(const WChain *ch, int32_t *ord, int n) {
    k_introsort_flat(n, ord, [&](int32_t x, int32_t y) { return ch[x].w > ch[y].w; });     // flt_lt, bwamem.cpp:61
}
// the overlap test of chain i (span [beg_i, end_i), weight w_i) against the kept chain cj: bit 0 = large overlap, bit 1 = ... and i is dropped (bwamem.cpp:570-584)
static __device__ __forceinline__ int flt_ovlp(const ChainParams &o, const WChain &cj, int beg_i, int end_i, int w_i, int alt_i) {
    const int beg_j = cj.first_qbeg, end_j = cj.last_qbeg + cj.last_len;
    const int b_max = beg_j > beg_i ? beg_j : beg_i;
    const int e_min = end_j < end_i ? end_j : end_i;
    if (e_min > b_max && (!cj.is_alt || alt_i)) {
        const int li = end_i - beg_i, lj = end_j - beg_j;
        const int min_l = li < lj ? li : lj;
        if (e_min - b_max >= min_l * o.mask_level && min_l < o.max_chain_gap)
            return (w_i < cj.w * o.drop_ratio && cj.w - w_i >= o.min_seed_len << 1) ? 3 : 1;
    }
    return 0;
}
static __device__ __forceinline__ void flt_sync() {              // lanes of ONE wavefront handing LDS or global data to each other: wavefront scope (the lanes share
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");       //  the CU's L1; agent scope -- an L2 write-back and an L1 invalidate per call -- made the heavy tiers take
    __builtin_amdgcn_wave_barrier();                             //  26 ms instead of 11, profiles/r05a_sweep.json)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// The same walk by the 64 lanes of a (converged) wavefront: chain i meets 64 kept chains at a time, one per lane.  The serial walk stops at
// the FIRST kept chain that drops i, and every kept chain up to and including that one which overlaps i largely gets `first` (if it has none):
// a ballot of the drop test gives the stopping lane, the lanes up to it apply their own chain's side effect -- no two lanes touch one chain.
static __device__ int flt_kept_coop(const ChainParams &o, WChain *ch, const int32_t *ord, int32_t *kept_list, int n, int lane) {
    if (lane == 0) { ch[ord[0]].kept = 3; kept_list[0] = 0; }
    int n_kept = 1;
    flt_sync();
    for (int i = 1; i < n; ++i) {
        const int ci_idx = ord[i];
        const int beg_i = ch[ci_idx].first_qbeg, end_i = ch[ci_idx].last_qbeg + ch[ci_idx].last_len, w_i = ch[ci_idx].w, alt_i = ch[ci_idx].is_alt;
        int large_ovlp = 0;
        bool stopped = false;
        for (int k0 = 0; k0 < n_kept && !stopped; k0 += 64) {
            const int kk = k0 + lane;
            int v = 0, j = -1;
            if (kk < n_kept) { j = ord[kept_list[kk]]; v = flt_ovlp(o, ch[j], beg_i, end_i, w_i, alt_i); }
            const unsigned long long bm = __ballot((v & 2) != 0);
            unsigned long long am = __ballot(v != 0);
            if (bm) {
                const int fb = __ffsll((long long)bm) - 1;
                am &= fb >= 63 ? ~0ULL : ((1ULL << (fb + 1)) - 1ULL);
                stopped = true;
            }
            if ((am >> lane) & 1ULL) { if (ch[j].first < 0) ch[j].first = i; }
            if (am) large_ovlp = 1;
        }
        if (!stopped) {
            if (lane == 0) { kept_list[n_kept] = i; ch[ci_idx].kept = large_ovlp ? 2 : 3; }
            n_kept++;
        }
        flt_sync();
    }
    return n_kept;
}
// the rest of mem_chain_flt and the read's chains with contiguous seeds
static __device__ void flt_rest_emit(const ChainParams &o, int r, WChain *ch, WSeed *sd, int32_t *ord, const int32_t *kept_list, int n, int n_kept, int64_t base,
                                     float frac_rep, DevChain *chn, DevSeed *seeds_out, int32_t *seed_owner, int32_t *n_chain_out, int32_t *n_reg_out) {
    int k;
    if (n > 0) {
        for (int i = 0; i < n_kept; ++i) {
            const WChain &c = ch[ord[kept_list[i]]];
            if (c.first >= 0) ch[ord[c.first]].kept = 1;
        }
        int i2;
        for (i2 = k = 0; i2 < n; ++i2) {
            const int kp = ch[ord[i2]].kept;
            if (kp == 0 || kp == 3) continue;
            if (++k >= o.max_chain_extend) break;
        }
        for (; i2 < n; ++i2) if (ch[ord[i2]].kept < 3) ch[ord[i2]].kept = 0;
        for (i2 = k = 0; i2 < n; ++i2) if (ch[ord[i2]].kept != 0) ord[k++] = ord[i2];
        n = k;
    }
    // ---- emit kept chains with contiguous seeds
    DevChain *oc = chn + base;
    DevSeed *os = seeds_out + base;
    int n_seed = 0;
    for (int i = 0; i < n; i++) {
        const WChain &c = ch[ord[i]];
        DevChain d;
        d.pos = c.pos; d.seed_off = base + n_seed; d.n = c.n; d.rid = c.rid; d.w = c.w; d.kept = c.kept; d.first = c.first;
        d.is_alt = c.is_alt; d.read = r; d.frac_rep = frac_rep; d.rmax0 = 0; d.rmax1 = 0;
        const int s0 = n_seed;
        for (int si = c.head; si >= 0; si = sd[si].next) {
            DevSeed s; s.rbeg = sd[si].rbeg; s.qbeg = sd[si].qbeg; s.len = sd[si].len; s.score = sd[si].len; s.aln = -1;
            os[n_seed++] = s;
        }
        d.reg0 = 0; d.pad = 0;
        for (int t = s0; t < n_seed; t++) seed_owner[base + t] = r;
        oc[i] = d;
    }
    n_chain_out[r] = n;
    n_reg_out[r] = 0;              // set by k_chain_finish
}
// A read whose finish the caller runs itself (with all its lanes: chain_finish_coop): what chain_one_read would have passed to chain_finish_read
struct DeferFinish { int valid, r, n; int64_t base; float frac_rep; WChain *ch; WSeed *sd; int32_t *ord; int32_t *kept; };
static __device__ void chain_finish_coop(const ChainParams &o, const DeferFinish &d, int lane, DevChain *chn, DevSeed *seeds_out, int32_t *seed_owner,
                                         int32_t *n_chain_out, int32_t *n_reg_out) {      // (every lane of the wavefront, converged; `d` is the same in all of them)
    int n_kept = 0;
    if (d.n > 0) {
        if (lane == 0) flt_sort(d.ch, d.ord, d.n);
        flt_sync();
        n_kept = flt_kept_coop(o, d.ch, d.ord, d.kept, d.n, lane);
    }
    if (lane == 0) flt_rest_emit(o, d.r, d.ch, d.sd, d.ord, d.kept, d.n, n_kept, d.base, d.frac_rep, chn, seeds_out, seed_owner, n_chain_out, n_reg_out);
    flt_sync();
}

struct IslSeed { int64_t rbeg; uint32_t ql; int32_t rid; };           // a seed staged in global memory (k_chain_islands): ql = qbeg | len << 15 | is_alt << 31
struct IslHash { unsigned long long key; int32_t cnt, start; };        // key = bucket + 1 (0: free); seeds in the island that STARTS at this bucket; its place in `perm`

// The working set of one read while it is chained: chains, seeds, B-tree nodes, an order array.  The lane-per-read kernel keeps
// them in the read's slices of global arrays; the wave-per-read kernel of seed-rich reads keeps them in LDS (k_chain_heavy).
struct ChainWork {
    WChain *ch; WSeed *sd; BtNode *nodes; int32_t *ord;
    // inputs of the read staged by the whole wavefront before lane 0 walks them (k_chain_heavy, `staged`): per seed the reference
    // position, (qbeg | len << 15 | is_alt << 31) and the contig id bns_intv2rid gives; per SMEM (m | (n + 1) << 15 | repetitive << 31)
    int64_t *st_rbeg; uint32_t *st_ql; int32_t *st_rid; uint32_t *st_sm; bool staged;
    // BM2_CHAIN_CLOCK=1: 100 MHz ticks of lane 0's walk, [1] mem_chain_seeds, [2] traversal + mem_chain_flt + the read's output (k_chain_heavy adds [0] staging, [3] reads, [4] seeds)
    unsigned long long *clk = nullptr;
};

// mem_chain_seeds + mem_chain_flt for read r.  LIGHT: the lane-per-read kernel (global slices); otherwise the caller offers LDS
// for up to lds_cap seeds in `lw`.
template <bool LIGHT>
static __device__ void chain_one_read(const DevIndex &ix, const ChainParams &o, int r, int n_reads, const int32_t *__restrict__ len,
                                      const bm2_smem_t *__restrict__ smems, const int32_t *__restrict__ smem_cnt,
                                      const int64_t *__restrict__ smem_off, const int64_t *__restrict__ sa_off,
                                      const int64_t *__restrict__ sa_coord, WChain *wchain, WSeed *wseed, BtNode *nodes_g, int32_t *order,
                                      DevChain *chn, DevSeed *seeds_out, int32_t *seed_owner, int32_t *n_chain_out, int32_t *n_reg_out,
                                      int32_t *n_chain0_out, int heavy_thr, const ChainWork *lw, int lds_cap, const IslSeed *ist = nullptr,
                                      const IslHash *isl_hash = nullptr, const int32_t *isl_slot = nullptr,
                                      DeferFinish *defer = nullptr /* !LIGHT: leave mem_chain_flt's walk and the output to the caller's wavefront */,
                                      const FinishOut *fo = nullptr /* LIGHT: the lane also builds the read's extension tasks (k_chain_finish's part) */) {
    const int n_sm = smem_cnt[r];
    const long long t_enter = (!LIGHT && lw && lw->clk) ? wall_clock64() : 0;
    n_chain_out[r] = 0; n_reg_out[r] = 0;          // (k_chain never gets here with a read it leaves to k_chain_heavy: one writer per read)
    if (n_chain0_out) n_chain0_out[r] = 0;
    if (n_sm == 0 || len[r] < o.min_seed_len) return;
    if (n_sm <= 1) {                                    // the `pos < num_smem - 1` loop bound of mem_chain_seeds (bwamem.cpp:834):
        const int b0 = (r / BM2_BLOCK_READS) * BM2_BLOCK_READS;      // a 512-read block whose SMEM total is <= 1 yields no chains
        const int b1 = b0 + BM2_BLOCK_READS < n_reads ? b0 + BM2_BLOCK_READS : n_reads;
        int tot = 0;
        for (int t = b0; t < b1 && tot <= 1; t++) tot += smem_cnt[t];
        if (tot <= 1) return;
    }
    const int64_t so = smem_off[r];
    const int64_t base = sa_off[so];
    const int n_sa = (int)(sa_off[so + n_sm] - base);
    if (n_sa == 0) return;
    const bool in_lds = !LIGHT && n_sa <= lds_cap;
    WChain *ch = in_lds ? lw->ch : wchain + base;
    WSeed *sd = in_lds ? lw->sd : wseed + base;
    int32_t *ord = in_lds ? lw->ord : order + base;
    BtNode *nodes = in_lds ? lw->nodes : nodes_g + base;
    BTree bt; bt.nodes = nodes; bt.n_nodes = 0; bt.n_keys = 0; bt.ch = ch;     // <= n_sa/4 + 1 nodes are ever needed
    bt.reg = !in_lds && o.reg_nodes;
    bt.root = bt_new(bt, 0);
    int n_ch = 0, n_sd = 0;
    RidCache ridc; ridc.lo = 1; ridc.hi = 0; ridc.rid = -1;
    int b = 0, e = 0, l_rep = 0;
    const bool staged = !LIGHT && in_lds && lw->staged;
    for (int i = 0; i < n_sm; i++) {                     // l_rep, bwamem.cpp:849-861
        int sb, se; bool rep;
        if (staged) { const uint32_t v = lw->st_sm[i]; sb = (int)(v & 0x7fffu); se = (int)((v >> 15) & 0xffffu); rep = (v >> 31) != 0; }
        else { sb = (int)smems[so + i].m; se = (int)smems[so + i].n + 1; rep = smems[so + i].s > o.max_occ; }
        if (!rep) continue;
        if (sb > e) { l_rep += e - b; b = sb; e = se; }
        else e = e > se ? e : se;
    }
    l_rep += e - b;
    // one flat loop over the read's seeds (SMEM-major, occurrence-minor -- the order of bwamem.cpp:879-905): the lanes of
    // a wavefront then run max(seeds per read) iterations instead of the sum over SMEM ranks of the per-rank maxima
    {
        int si = -1, left = 0, qbeg = 0, slen = 0;
        for (int t = 0; t < n_sa; t++) {
            WSeed s; int rid, alt_staged = 0;
            if (ist) {                                   // every seed staged in global memory by the island kernel's wavefront
                const IslSeed q = ist[t];
                s.rbeg = q.rbeg; s.qbeg = (int)(q.ql & 0x7fffu); s.len = (int)((q.ql >> 15) & 0xffffu); s.next = -1;
                rid = q.rid; alt_staged = (int)(q.ql >> 31);
            } else if (staged) {
                const uint32_t ql = lw->st_ql[t];
                s.rbeg = lw->st_rbeg[t]; s.qbeg = (int)(ql & 0x7fffu); s.len = (int)((ql >> 15) & 0xffffu); s.next = -1;
                rid = lw->st_rid[t]; alt_staged = (int)(ql >> 31);
            } else {
                while (left == 0) {                      // next SMEM with at least one sampled occurrence
                    ++si;
                    left = (int)(sa_off[so + si + 1] - sa_off[so + si]);
                    qbeg = (int)smems[so + si].m; slen = (int)(smems[so + si].n + 1 - smems[so + si].m);
                }
                --left;
                s.rbeg = sa_coord[base + t]; s.qbeg = qbeg; s.len = slen; s.next = -1;
                rid = intv2rid_cached(ix, s.rbeg, s.rbeg + s.len, ridc);
            }
            if (rid < 0) continue;                       // bwamem.cpp:915-919
            int to_add = 0;
            // (a seed that is alone in its island of reference buckets can meet no chain: whatever kb_intervalp returned, test_and_merge
            //  would fail -- see k_chain_islands -- so the look-up is skipped and only the tree's shape is kept exact)
            if (isl_hash && isl_hash[isl_slot[t]].cnt == 1) to_add = 1;
            else if (bt.n_keys) {
                const int lower = bt_lower(bt, s.rbeg);
                if (lower < 0) to_add = 1;
                else {
                    const int m = test_and_merge(o, ix.l_pac, ch[lower], s, rid, sd, n_sd);
                    if (m == 2) n_sd++;
                    else if (m == 0) to_add = 1;
                }
            } else to_add = 1;
            if (to_add) {                                // bwamem.cpp:930-951
                WChain c2;
                c2.pos = s.rbeg; c2.last_rbeg = s.rbeg; c2.first_qbeg = s.qbeg; c2.last_qbeg = s.qbeg; c2.last_len = s.len;
                c2.n = 1; c2.rid = rid; c2.is_alt = (staged || ist) ? alt_staged : (ix.ann_is_alt[rid] ? 1 : 0); c2.head = c2.tail = n_sd;
                c2.w = 0; c2.kept = 0; c2.first = -1;
                sd[n_sd] = s; n_sd++;
                ch[n_ch] = c2;
                bt_put(bt, n_ch, c2.pos);
                n_ch++;
            }
        }
    }
    long long t_walk = 0;
    if (!LIGHT && lw && lw->clk) { t_walk = wall_clock64(); atomicAdd(lw->clk + 1, (unsigned long long)(t_walk - t_enter)); }
    int n = bt_traverse(bt, ord);                        // chains in key order (bwamem.cpp:958-962)
    if (n_chain0_out) n_chain0_out[r] = n;
    const float frac_rep = (float)l_rep / len[r];        // bwamem.cpp:965-966
    // ---- mem_chain_flt, bwamem.cpp:506-624
    int k = 0;
    for (int i = 0; i < n; i++) {
        WChain &c = ch[ord[i]];
        c.first = -1; c.kept = 0;
        c.w = chain_weight(c, sd);
        if (c.w >= o.min_chain_weight) ord[k++] = ord[i];
    }
    if (k == 0 && n > 0) k = 1;      // quirk: an empty survivor list still processes the untouched a_[0] (bwamem.cpp:529-546)
    if (!LIGHT && defer) {
        defer->r = r; defer->n = k; defer->base = base; defer->frac_rep = frac_rep; defer->ch = ch; defer->sd = sd; defer->ord = ord; defer->kept = (int32_t *)nodes;
        defer->valid = 1;
        return;
    }
    chain_finish_read(o, r, ch, sd, ord, (int32_t *)nodes, k, base, frac_rep, chn, seeds_out, seed_owner, n_chain_out, n_reg_out, &ix, LIGHT ? fo : nullptr);
    if (!LIGHT && lw && lw->clk) atomicAdd(lw->clk + 2, (unsigned long long)(wall_clock64() - t_walk));
}

__global__ void __launch_bounds__(128, 6)
k_chain(DevIndex ix, ChainParams o, int n_reads, const int32_t *__restrict__ len, const bm2_smem_t *__restrict__ smems,
        const int32_t *__restrict__ smem_cnt, const int64_t *__restrict__ smem_off, const int64_t *__restrict__ sa_off,
        const int64_t *__restrict__ sa_coord, WChain *wchain, WSeed *wseed, BtNode *nodes, int32_t *order,
        DevChain *chn, DevSeed *seeds_out, int32_t *seed_owner,
        int32_t *n_chain_out, int32_t *n_reg_out, int32_t *n_chain0_out, const int32_t *__restrict__ perm, int heavy_thr,
        const int32_t *__restrict__ n_sa_read, FinishOut fo /* srt_out == nullptr: k_chain_finish does every read */) {
    const int tix = blockIdx.x * blockDim.x + threadIdx.x;
    if (tix >= n_reads) return;
    if (heavy_thr >= 0 && n_sa_read[perm[tix]] > heavy_thr) return;          // a whole wavefront takes this read (k_chain_heavy)
    chain_one_read<true>(ix, o, perm[tix], n_reads, len, smems, smem_cnt, smem_off, sa_off, sa_coord, wchain, wseed, nodes, order, chn, seeds_out,
                         seed_owner, n_chain_out, n_reg_out, n_chain0_out, heavy_thr, nullptr, 0, nullptr, nullptr, nullptr, nullptr, fo.srt_out ? &fo : nullptr);
}

static __device__ __forceinline__ void chain_wave_sync() {      // lanes of one wavefront handing data to each other through LDS
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Seed-rich reads, ONE READ PER WAVEFRONT.  Chaining is sequential per read by definition (every seed meets the B-tree the earlier
// ones built) and costs ~25 dependent memory accesses per seed: in the lane-per-read kernel a read with 600 seeds kept one lane busy
// for 12 ms at global-memory latency while the other million reads took 2 ms.  Here lane 0 runs the same code with the read's
// chains, seeds, B-tree nodes and order array in LDS (~130 bytes per seed), where a dependent access costs ~60 ns; the launch is
// tiered by seed count so that a block claims only the LDS its reads need (tier capacity `cap`: reads with lo < seeds <= cap; the
// last tier also takes the reads beyond its capacity and works on their global slices).  Items come from the heavy-first list of
// the partition; every tier scans it and skips what is not its own.
// COOP (BM2_CHAIN_COOP_FLT): mem_chain_flt's walk over the kept chains by all 64 lanes; a launch of its own so that the default one keeps its registers.
// WPE: wavefronts per SIMD the register allocation leaves room for (BM2_CHAIN_HEAVY_WPE).  Left alone the cooperative instantiation takes 171 registers --
// 176 allocated, TWO wavefronts per SIMD, eight per CU, where the tiers of the seed-poorest heavy reads could hold sixteen by their LDS; 3: 139 registers,
// nothing spilled; 4: 128 registers and 32-64 more bytes of scratch
template <bool COOP, int WPE>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WPE, WPE)))
k_chain_heavy(DevIndex ix, ChainParams o, int n_reads, const int32_t *__restrict__ len, const bm2_smem_t *__restrict__ smems,
              const int32_t *__restrict__ smem_cnt, const int64_t *__restrict__ smem_off, const int64_t *__restrict__ sa_off,
              const int64_t *__restrict__ sa_coord, WChain *wchain, WSeed *wseed, BtNode *nodes, int32_t *order,
              DevChain *chn, DevSeed *seeds_out, int32_t *seed_owner, int32_t *n_chain_out, int32_t *n_reg_out, int32_t *n_chain0_out,
              const int32_t *__restrict__ heavy /* read ids, heavy ones first */, const int64_t *__restrict__ n_heavy_p,
              const int32_t *__restrict__ n_sa_read, int lo, int cap, int last_tier, unsigned long long *item_cur, int stage,
              unsigned long long *clk /* or NULL: BM2_CHAIN_CLOCK */) {
    extern __shared__ __attribute__((aligned(16))) uint8_t chain_lds[];
    const int lane = threadIdx.x;
    ChainWork lw;
    lw.clk = clk;
    int32_t *st_cut = nullptr;
    {
        size_t at = 0;
        lw.ch = (WChain *)(chain_lds + at); at += (size_t)cap * sizeof(WChain);
        lw.nodes = (BtNode *)(chain_lds + at); at += (size_t)(cap / 4 + 2) * sizeof(BtNode);
        lw.sd = (WSeed *)(chain_lds + at); at += (size_t)cap * sizeof(WSeed);
        lw.ord = (int32_t *)(chain_lds + at); at += (size_t)cap * 4;
        at = (at + 7) & ~(size_t)7;                               // (the staged arrays exist only when the launch asked for them)
        lw.st_rbeg = (int64_t *)(chain_lds + at); at += (size_t)cap * 8;
        lw.st_ql = (uint32_t *)(chain_lds + at); at += (size_t)cap * 4;
        lw.st_rid = (int32_t *)(chain_lds + at); at += (size_t)cap * 4;
        lw.st_sm = (uint32_t *)(chain_lds + at); at += (size_t)cap * 4;
        st_cut = (int32_t *)(chain_lds + at);
        lw.staged = false;
    }
    const int64_t n_heavy = *n_heavy_p;
    for (;;) {
        // (every lane takes part in the atomic and the body sits in an `if`: see the note on work loops in smem.hip)
        const unsigned long long it = atomicAdd(item_cur, lane == 0 ? 1ULL : 0ULL);
        const int64_t hid = (int64_t)(((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(it >> 32)) << 32) |
                                      (unsigned)__builtin_amdgcn_readfirstlane((unsigned)it));
        if (hid >= n_heavy) break;
        const int r = heavy[hid];
        const int ns = n_sa_read[r];
        const bool mine = ns > lo && (ns <= cap || last_tier);
        // The serial walk of lane 0 pays a global round trip (~1 us) for every seed's position and, for a repeat read whose seeds lie
        // in many contigs, ~10 more for the two binary searches of bns_intv2rid -- the bulk of this kernel's time.  Those inputs do
        // not depend on the walk: the 64 lanes fetch them for 64 seeds at a time into LDS first.
        lw.staged = false;
        const long long t_stage = clk && mine ? wall_clock64() : 0;
        if (mine && stage && ns <= cap) {
            const int n_sm = smem_cnt[r];
            if (n_sm > 0) {
                const int64_t so = smem_off[r];
                const int64_t base = sa_off[so];
                const int n_sa = (int)(sa_off[so + n_sm] - base);
                if (n_sa > 0 && n_sa <= cap) {
                    for (int i = lane; i < n_sm; i += 64) {
                        const uint32_t m = smems[so + i].m, n1 = smems[so + i].n + 1;
                        lw.st_sm[i] = (m & 0x7fffu) | (n1 & 0xffffu) << 15 | (smems[so + i].s > o.max_occ ? 1u << 31 : 0u);
                        st_cut[i] = (int32_t)(sa_off[so + i + 1] - base);          // seeds of SMEM i end here
                    }
                    chain_wave_sync();
                    for (int t = lane; t < n_sa; t += 64) {
                        int a = 0, b = n_sm - 1;                                  // the seed's SMEM: first i with st_cut[i] > t
                        while (a < b) { const int mid = (a + b) >> 1; if (st_cut[mid] > t) b = mid; else a = mid + 1; }
                        const uint32_t sm = lw.st_sm[a];
                        const int qbeg = (int)(sm & 0x7fffu), slen = (int)((sm >> 15) & 0xffffu) - qbeg;
                        const int64_t rbeg = sa_coord[base + t];
                        const int rid = intv2rid(ix, rbeg, rbeg + slen);
                        const uint32_t alt = rid >= 0 && ix.ann_is_alt[rid] ? 1u : 0u;
                        lw.st_rbeg[t] = rbeg; lw.st_ql[t] = (uint32_t)qbeg | (uint32_t)slen << 15 | alt << 31; lw.st_rid[t] = rid;
                    }
                    chain_wave_sync();
                    lw.staged = true;
                }
            }
        }
        if (clk && mine && lane == 0) { atomicAdd(clk, (unsigned long long)(wall_clock64() - t_stage)); atomicAdd(clk + 3, 1ULL); atomicAdd(clk + 4, (unsigned long long)ns); }
        if constexpr (COOP) {
          if (mine) {                                              // (`mine` is the same in every lane)
            __shared__ DeferFinish df;
            if (lane == 0) {
                df.valid = 0;
                chain_one_read<false>(ix, o, r, n_reads, len, smems, smem_cnt, smem_off, sa_off, sa_coord, wchain, wseed, nodes, order, chn, seeds_out,
                                      seed_owner, n_chain_out, n_reg_out, n_chain0_out, -1, &lw, cap, nullptr, nullptr, nullptr, &df);
            }
            flt_sync();
            if (df.valid) {
                const DeferFinish d = df;
                chain_finish_coop(o, d, lane, chn, seeds_out, seed_owner, n_chain_out, n_reg_out);
            }
            flt_sync();
          }
        } else if (mine && lane == 0)
            chain_one_read<false>(ix, o, r, n_reads, len, smems, smem_cnt, smem_off, sa_off, sa_coord, wchain, wseed, nodes, order, chn, seeds_out,
                                  seed_owner, n_chain_out, n_reg_out, n_chain0_out, -1, &lw, cap);
    }
}

// ---------------------------------------------------------------- long reads: chaining by ISLANDS
// A 10 kb read brings ~12 000 seeds (a 30 kb read 35 000), and mem_chain_seeds is sequential by definition: every seed meets the B-tree the
// earlier ones built.  Walked by one lane through the read's global slices that is ~31 us per seed -- one second for the chunk's longest
// read, which is what the chaining of a long-read chunk took (profiles/r04b_kernel_trace_ont2d.md: 1.07 s of a 1.68 s step).
// But the seeds of a read interact only NEAR each other: test_and_merge (bwamem.cpp:357-399) lets a seed p join or be swallowed by the chain c
// below it only if p.rbeg - (c's last seed's end) < max_chain_gap, so once the seeds are filed in buckets of 2^S >= max_chain_gap + (longest
// seed) reference bases, seeds of buckets that are not adjacent can never meet: for a seed whose `lower` chain (kb_intervalp) lies in another
// run of occupied buckets the test fails exactly as it does when there is no lower chain at all -- a new chain either way.  So every maximal
// run of occupied buckets -- an island -- is chained on its own, in the read's seed order, with a B-tree of its own: the ~10 000 stray hits of
// a read are islands of one or two seeds (no tree at all), the true locus is one island of a few thousand seeds whose tree stays a handful of
// nodes.  What the rest of the path reads of mem_chain_seeds' result is the chains that pass the weight test IN KEY ORDER (plus their number
// and, when none passes, the chain with the smallest key: the a_[0] quirk): the islands' survivors are sorted by position afterwards.
// One thing a private tree cannot reproduce: chains with EQUAL keys -- where the later one lands and which of them kb_intervalp returns depend on
// the shape of the whole tree (isl_build).  An island that is about to create one raises a flag and the read is chained again by the serial code
// -- on the inputs the wavefront has already staged, so that even that walk makes no look-up of its own besides the tree's.
// One wavefront per read: all lanes stage the seeds (position, query span, contig: the two binary searches of bns_intv2rid leave the serial
// part), file them, number the islands, put the seeds of every island together in seed order (a stable counting sort); then every lane
// chains islands of its own; then lane 0 finishes the read.  Scratch: the read's slices of the OUTPUT arrays, which nothing has written yet.
static_assert(sizeof(IslHash) == 16 && sizeof(IslSeed) == 16 && sizeof(DevChain) >= 68 && sizeof(DevSeed) >= 24, "island scratch is carved from the output slices");

static __device__ __forceinline__ unsigned isl_slot0(unsigned long long key, int log_h) { return (unsigned)((key * 0x9E3779B97F4A7C15ULL) >> (64 - log_h)); }
static __device__ int isl_insert(IslHash *h, int log_h, unsigned long long key) {
    const unsigned mask = (1u << log_h) - 1u;
    for (unsigned i = isl_slot0(key, log_h);; i = (i + 1) & mask) {
        const unsigned long long prev = atomicCAS(&h[i].key, 0ULL, key);
        if (prev == 0ULL || prev == key) return (int)i;
    }
}
static __device__ int isl_find(const IslHash *h, int log_h, unsigned long long key) {
    const unsigned mask = (1u << log_h) - 1u;
    for (unsigned i = isl_slot0(key, log_h);; i = (i + 1) & mask) {
        const unsigned long long k = h[i].key;
        if (k == key) return (int)i;
        if (k == 0ULL) return -1;
    }
}
static __device__ __forceinline__ void isl_sync() {              // lanes of one wavefront handing GLOBAL data to each other
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

// one island: the seeds perm[start .. start + tot) in seed order; chains / seeds get the ids start, start + 1, ... of the read's slices
static __device__ void isl_build(const DevIndex &ix, const ChainParams &o, const IslSeed *st, const int32_t *perm, int start, int tot,
                                 WChain *ch, WSeed *sd, BtNode *nd, int32_t *ord, int *s_nsurv, int *s_ntot, int *s_dup, unsigned long long *s_min,
                                 bool quit_on_dup /* the read goes to k_chain_serial when equal keys turn up: nothing of this pass is kept */) {
    int n_ch = start, n_sd = start;
    bool dup = false;
    if (tot == 1) {                                              // a stray hit: a chain of one seed, no tree
        const IslSeed q = st[perm[start]];
        WSeed s; s.rbeg = q.rbeg; s.qbeg = (int)(q.ql & 0x7fffu); s.len = (int)((q.ql >> 15) & 0xffffu); s.next = -1; s.pad = 0;
        WChain c2;
        c2.pos = s.rbeg; c2.last_rbeg = s.rbeg; c2.first_qbeg = s.qbeg; c2.last_qbeg = s.qbeg; c2.last_len = s.len;
        c2.n = 1; c2.rid = q.rid; c2.is_alt = (int)(q.ql >> 31); c2.head = c2.tail = start; c2.w = 0; c2.kept = 0; c2.first = -1; c2.pad = 0;
        sd[start] = s; ch[start] = c2;
        n_ch = start + 1;
    } else {
        BTree bt; bt.nodes = nd + start; bt.n_nodes = 0; bt.n_keys = 0; bt.ch = ch; bt.reg = o.reg_nodes != 0;
        bt.root = bt_new(bt, 0);
        // Equal keys: where a chain lands whose key another chain already has -- after it, or BEFORE it when that chain happens to be the median of
        // a full node that kb_putp splits on its way down (`if (k > median) ++i`) -- and which of the two a later look-up finds depend on the
        // shape of the WHOLE tree.  A private tree cannot know: the island raises `dup`, the read is chained again by the serial code.
        for (int idx = start; idx < start + tot; idx++) {
            // (another island of the read has met equal keys: the read will be chained again whatever this island yields -- its long islands are what
            //  stands between the read and k_chain_serial)
            if (quit_on_dup && *(volatile int *)s_dup) return;
            const IslSeed q = st[perm[idx]];
            WSeed s; s.rbeg = q.rbeg; s.qbeg = (int)(q.ql & 0x7fffu); s.len = (int)((q.ql >> 15) & 0xffffu); s.next = -1; s.pad = 0;
            int to_add = 0;
            if (bt.n_keys) {
                const int lower = bt_lower(bt, s.rbeg);
                if (lower < 0) to_add = 1;
                else {
                    const int m = test_and_merge(o, ix.l_pac, ch[lower], s, q.rid, sd, n_sd);
                    if (m == 2) n_sd++;
                    else if (m == 0) { to_add = 1; if (ch[lower].pos == s.rbeg) { dup = true; if (quit_on_dup) { atomicAdd(s_dup, 1); return; } } }
                }
            } else to_add = 1;
            if (to_add) {
                WChain c2;
                c2.pos = s.rbeg; c2.last_rbeg = s.rbeg; c2.first_qbeg = s.qbeg; c2.last_qbeg = s.qbeg; c2.last_len = s.len;
                c2.n = 1; c2.rid = q.rid; c2.is_alt = (int)(q.ql >> 31); c2.head = c2.tail = n_sd; c2.w = 0; c2.kept = 0; c2.first = -1; c2.pad = 0;
                sd[n_sd] = s; n_sd++;
                ch[n_ch] = c2;
                bt_put(bt, n_ch, c2.pos);
                n_ch++;
            }
        }
    }
    unsigned long long mn = ~0ULL;
    for (int i = start; i < n_ch; i++) {                         // the weight test of mem_chain_flt (bwamem.cpp:516-528), island by island
        WChain &c = ch[i];
        c.first = -1; c.kept = 0;
        c.w = chain_weight(c, sd);
        if (c.w >= o.min_chain_weight) ord[atomicAdd(s_nsurv, 1)] = i;
        const unsigned long long key = (unsigned long long)c.pos << 24 | (unsigned)i;
        mn = mn < key ? mn : key;
    }
    atomicAdd(s_ntot, n_ch - start);
    atomicMin(s_min, mn);
    if (dup) atomicAdd(s_dup, 1);
}

// l_rep of mem_chain_seeds (bwamem.cpp:849-861: the query bases covered by repetitive SMEMs, a union of intervals in SMEM order) by a converged wavefront:
// 64 SMEMs are looked at per round, the repetitive ones -- a handful in a read, if any -- are merged in order from a ballot; every lane ends with the same sum.
// (One lane walking the list paid a dependent load per SMEM: a long read has a thousand of them.  Written at the end of round 5 without a GPU, run in round 6:
//  bit-exact -- the 1024-read gate and the long-read GPU tests -- and config 5's chaining stage 467 -> 463-477 ms, nothing either way: profiles/r06q_*.)
static __device__ int lrep_coop(const bm2_smem_t *__restrict__ sm, int n_sm, int max_occ, int lane) {
    int b = 0, e = 0, l_rep = 0;
    for (int i0 = 0; i0 < n_sm; i0 += 64) {
        const int i = i0 + lane;
        int sb = 0, se = 0; bool rep = false;
        if (i < n_sm) { rep = sm[i].s > max_occ; sb = (int)sm[i].m; se = (int)sm[i].n + 1; }
        unsigned long long m = __ballot(rep);
        while (m) {
            const int src = __ffsll((long long)m) - 1;
            m &= m - 1ULL;
            const int xb = __shfl(sb, src), xe = __shfl(se, src);
            if (xb > e) { l_rep += e - b; b = xb; e = xe; }
            else e = e > xe ? e : xe;
        }
    }
    return l_rep + (e - b);
}

// The list k_chain_islands (producer, any of its wavefronts' lane 0) hands to k_chain_serial (consumer, running BESIDE it on a stream of its own): a place is
// drawn with an atomic, the read's number stored with release semantics behind everything the wavefront staged for it (the caller's isl_sync); the list is
// memset to -1 before the launch, a consumer that drew place i waits for list[i] to turn up or for the last producer to leave (n_fallback[SER_DONE]).
#define SER_PLAIN 0x40000000         // a read the island kernel did not stage (plain chain_one_read)
#define SER_DONE 32                  // n_fallback[32] = counters[48] of the batch: producers that have left
static __device__ __forceinline__ void ser_publish(int32_t *list, unsigned long long *n_fallback, int v) {
    const unsigned long long idx = atomicAdd(n_fallback, 1ULL);
    __hip_atomic_store(list + idx, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
// COOP (BM2_CHAIN_COOP_FLT): see k_chain_heavy.  OWN: the reads this kernel cannot chain by islands are LISTED (serial_list) for k_chain_serial and the serial
// code is not part of this kernel at all -- with it inlined the kernel needed 264 registers, ONE wavefront per SIMD, and a chunk of 20 000 long reads was
// throughput-bound at four wavefronts per CU
// (OWN = 3 or 4: the wavefronts per SIMD the registers are allocated for, BM2_CHAIN_ISL_WPE -- 145 registers, or 128 and four more dwords of scratch)
template <bool COOP, int OWN>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(OWN ? OWN : 1, OWN ? OWN : 8)))
k_chain_islands(DevIndex ix, ChainParams o, int n_reads, const int32_t *__restrict__ len, const bm2_smem_t *__restrict__ smems,
                const int32_t *__restrict__ smem_cnt, const int64_t *__restrict__ smem_off, const int64_t *__restrict__ sa_off,
                const int64_t *__restrict__ sa_coord, WChain *wchain, WSeed *wseed, BtNode *nodes, int32_t *order,
                DevChain *chn, DevSeed *seeds_out, int32_t *seed_owner, int32_t *cut_all, int32_t *n_chain_out, int32_t *n_reg_out, int32_t *n_chain0_out,
                const int32_t *__restrict__ heavy /* read ids, heavy ones first */, const int64_t *__restrict__ n_heavy_p,
                const int32_t *__restrict__ n_sa_read, int lo, unsigned long long *item_cur, unsigned long long *n_fallback, int n_items,
                int32_t *serial_list /* or NULL: reads with equal chain keys are listed for k_chain_serial instead of being chained again here */) {
    __shared__ int s_nsurv, s_ntot, s_dup;
    __shared__ unsigned long long s_min;
    __shared__ DeferFinish df;                                     // (only the COOP launch touches it)
    const int lane = threadIdx.x;
    const unsigned long long lt_mask = lane ? (~0ULL >> (64 - lane)) : 0ULL;
    const int64_t n_heavy = n_items >= 0 ? (int64_t)n_items : *n_heavy_p;     // (n_items: `heavy` lists EVERY read, the seed-richest first)
    for (;;) {
        // (every lane takes part in the atomic and the body sits in an `if`: see the note on work loops in smem.hip)
        const unsigned long long it = atomicAdd(item_cur, lane == 0 ? 1ULL : 0ULL);
        const int64_t hid = (int64_t)(((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(it >> 32)) << 32) |
                                      (unsigned)__builtin_amdgcn_readfirstlane((unsigned)it));
        if (hid >= n_heavy) break;
        const int r = __builtin_amdgcn_readfirstlane(heavy[hid]);
        const int ns = __builtin_amdgcn_readfirstlane(n_sa_read[r]);
        if (ns > lo) {
        const int n_sm = __builtin_amdgcn_readfirstlane(smem_cnt[r]);
        if (lane == 0) { n_chain_out[r] = 0; n_reg_out[r] = 0; if (n_chain0_out) n_chain0_out[r] = 0; }
        const int64_t so = smem_off[r];
        const int64_t base = sa_off[so];
        const int n_sa = n_sm > 0 ? (int)(sa_off[so + n_sm] - base) : 0;
        if (n_sm > 1 && n_sa >= 64 && len[r] >= o.min_seed_len) {     // (a read that comes here has more than `lo` seeds: the block rule of one-SMEM reads cannot apply)
        WChain *ch = wchain + base; WSeed *sd = wseed + base; BtNode *nd = nodes + base; int32_t *ord = order + base;
        IslHash *hash = (IslHash *)(chn + base);
        IslSeed *st = (IslSeed *)(seeds_out + base);
        // (chn slice, 72 B per seed: the table, < 64 B per seed, then the island list; seeds_out slice, 24 B per seed: staged seed, its island, the permutation)
        int32_t *cslot = (int32_t *)(st + n_sa), *perm = cslot + n_sa, *clist = (int32_t *)((char *)(chn + base) + (size_t)64 * n_sa), *cut = cut_all + base;
        long long tck = wall_clock64();                              // phase clock (lane 0 adds every phase's ticks to the counters behind n_fallback: bm2_batch_fetch("counters"))
#define ISL_TICK(ph) do { const long long t_now = wall_clock64(); if (lane == 0) { atomicAdd(n_fallback + 3 + (ph), (unsigned long long)(t_now - tck)); atomicMax(n_fallback + 15 + (ph), (unsigned long long)(t_now - tck)); } tck = t_now; } while (0)
        int log_h = 7;
        while ((1 << log_h) < 2 * n_sa) log_h++;
        const int H = 1 << log_h;                                    // < 4 n_sa entries of 16 bytes: inside the read's slice of chn
        // ---- stage: SMEM cuts, the bucket width, the table
        for (int i = lane; i < H; i += 64) { IslHash e; e.key = 0ULL; e.cnt = 0; e.start = 0; hash[i] = e; }
        int mx = 0;
        for (int i = lane; i < n_sm; i += 64) {
            cut[i] = (int32_t)(sa_off[so + i + 1] - base);           // seeds of SMEM i end here
            const int l = (int)(smems[so + i].n + 1 - smems[so + i].m);
            mx = mx > l ? mx : l;
        }
        for (int d = 32; d > 0; d >>= 1) { const int y = __shfl_xor(mx, d); mx = mx > y ? mx : y; }
        int S = 1;
        while ((1LL << S) < (long long)o.max_chain_gap + mx + 1) S++;
        isl_sync();
        ISL_TICK(0);
        // ---- every seed: position, query span, contig; its bucket goes into the table
        for (int t = lane; t < n_sa; t += 64) {
            int a = 0, b = n_sm - 1;                                  // the seed's SMEM: first i with cut[i] > t
            while (a < b) { const int mid = (a + b) >> 1; if (cut[mid] > t) b = mid; else a = mid + 1; }
            const int qbeg = (int)smems[so + a].m, slen = (int)(smems[so + a].n + 1 - smems[so + a].m);
            const int64_t rbeg = sa_coord[base + t];
            const int rid = intv2rid(ix, rbeg, rbeg + slen);
            const uint32_t alt = rid >= 0 && ix.ann_is_alt[rid] ? 1u : 0u;
            IslSeed q; q.rbeg = rbeg; q.ql = (uint32_t)qbeg | (uint32_t)slen << 15 | alt << 31; q.rid = rid;
            st[t] = q;
            if (rid >= 0) (void)isl_insert(hash, log_h, (unsigned long long)(rbeg >> S) + 1ULL);      // (rid < 0: the seed is skipped, bwamem.cpp:915-919)
        }
        isl_sync();
        ISL_TICK(1);
        // ---- the island of every seed = the first bucket of its run of occupied buckets
        for (int t = lane; t < n_sa; t += 64) {
            int slot = -1;
            if (st[t].rid >= 0) {
                long long b = st[t].rbeg >> S;
                while (b > 0 && isl_find(hash, log_h, (unsigned long long)b) >= 0) b--;      // key of bucket b - 1 is b
                slot = isl_find(hash, log_h, (unsigned long long)b + 1ULL);
                atomicAdd(&hash[slot].cnt, 1);
            }
            cslot[t] = slot;
        }
        isl_sync();
        ISL_TICK(2);
        // ---- places: prefix sums of the island sizes over the table, the list of islands
        int run = 0, n_comp = 0;
        for (int h0 = 0; h0 < H; h0 += 64) {
            const int c = hash[h0 + lane].cnt;
            int x = c;
            for (int d = 1; d < 64; d <<= 1) { const int y = __shfl(x, lane >= d ? lane - d : lane); if (lane >= d) x += y; }
            hash[h0 + lane].start = run + x - c;
            const unsigned long long m = __ballot(c > 0);
            if (c > 0) clist[n_comp + __popcll(m & lt_mask)] = h0 + lane;
            n_comp += __popcll(m);
            run += __shfl(x, 63);
        }
        isl_sync();
        ISL_TICK(3);
        // ---- the seeds of an island together, in seed order: blocks of 64 seeds in order, inside a block the lanes of one island ranked by lane
        // (`start` of an island with several seeds moves on as its seeds are placed: it ends at start + cnt)
        for (int t0 = 0; t0 < n_sa; t0 += 64) {
            const int t = t0 + lane;
            const int slot = t < n_sa ? cslot[t] : -1;
            int tot = 0, stt = 0;
            if (slot >= 0) { tot = hash[slot].cnt; stt = hash[slot].start; }
            const bool multi = slot >= 0 && tot > 1;
            if (slot >= 0 && tot == 1) perm[stt] = t;
            unsigned long long rem = __ballot(multi);
            const bool any_multi = rem != 0ULL;
            while (rem) {
                const int leader = __ffsll((long long)rem) - 1;
                const int key = __shfl(slot, leader);
                const bool mine = multi && slot == key;
                const unsigned long long m = __ballot(mine);
                if (mine) perm[stt + __popcll(m & lt_mask)] = t;
                if (lane == leader) hash[key].start = stt + __popcll(m);
                rem &= ~m;
            }
            if (any_multi) isl_sync();
        }
        if (lane == 0) { s_nsurv = 0; s_ntot = 0; s_dup = 0; s_min = ~0ULL; }
        isl_sync();
        ISL_TICK(4);
        // ---- the islands, one per lane at a time (lanes diverge from here to the next rendezvous: no wavefront primitive inside)
        for (int k = lane; k < n_comp; k += 64) {
            const int slot = clist[k];
            const int tot = hash[slot].cnt;
            const int start = hash[slot].start - (tot > 1 ? tot : 0);
            if (OWN != 0 && *(volatile int *)&s_dup) break;
            isl_build(ix, o, st, perm, start, tot, ch, sd, nd, ord, &s_nsurv, &s_ntot, &s_dup, &s_min, OWN != 0);
        }
        isl_sync();
        ISL_TICK(5);
        // ---- the read
        const int l_rep_all = lrep_coop(smems + so, n_sm, o.max_occ, lane);
        if (lane == 0) {
            if constexpr (COOP) df.valid = 0;
            if (s_dup) {                                             // chains with equal keys: the serial code on the read's slices
                if constexpr (OWN) ser_publish(serial_list, n_fallback, r);      // ... in k_chain_serial (the read's staged seeds, table and islands stay where they are)
                else {
                    atomicAdd(n_fallback, 1ULL);
                    const long long t_fb = wall_clock64();
                    chain_one_read<false>(ix, o, r, n_reads, len, smems, smem_cnt, smem_off, sa_off, sa_coord, wchain, wseed, nodes, order, chn, seeds_out,
                                          seed_owner, n_chain_out, n_reg_out, n_chain0_out, -1, nullptr, 0, st, hash, cslot, COOP ? &df : (DeferFinish *)nullptr);
                    const unsigned long long dt_fb = (unsigned long long)(wall_clock64() - t_fb);
                    atomicAdd(n_fallback + 12, dt_fb); atomicAdd(n_fallback + 13, (unsigned long long)n_sa); atomicMax(n_fallback + 14, dt_fb);
                }
            } else {
                const int n_all = s_ntot;
                if (n_chain0_out) n_chain0_out[r] = n_all;
                const int l_rep = l_rep_all;                         // l_rep, bwamem.cpp:849-861
                int k = s_nsurv;
                if (k == 0 && n_all > 0) { ord[0] = (int32_t)(s_min & 0xffffffULL); k = 1; }       // the a_[0] quirk: the chain with the smallest key
                else if (k > 1) k_introsort_flat(k, ord, [&](int32_t x, int32_t y) { return ch[x].pos < ch[y].pos; });     // key order (the keys are distinct here)
                if constexpr (COOP) {
                    df.r = r; df.n = k; df.base = base; df.frac_rep = (float)l_rep / len[r]; df.ch = ch; df.sd = sd; df.ord = ord; df.kept = (int32_t *)nd;
                    df.valid = 1;
                } else chain_finish_read(o, r, ch, sd, ord, (int32_t *)nd, k, base, (float)l_rep / len[r], chn, seeds_out, seed_owner, n_chain_out, n_reg_out);
                atomicAdd(n_fallback + 1, 1ULL);                     // reads chained by islands
                atomicAdd(n_fallback + 2, (unsigned long long)n_comp);      // islands
                atomicAdd(n_fallback + 10, (unsigned long long)k);           // chains that passed the weight test
                atomicAdd(n_fallback + 11, (unsigned long long)n_all);       // chains
            }
        }
        isl_sync();
        if constexpr (COOP) {                                        // mem_chain_flt's walk over the kept chains and the read's output, by the whole wavefront
            if (df.valid) {
                const DeferFinish d = df;
                chain_finish_coop(o, d, lane, chn, seeds_out, seed_owner, n_chain_out, n_reg_out);
            }
            isl_sync();
        }
        ISL_TICK(6);
        } else if (lane == 0) {                                      // (too few seeds for the table's place in the slices -- cannot happen above `lo` >= 64 -- or nothing to chain)
            if constexpr (OWN) { atomicAdd(n_fallback + 23, 1ULL); ser_publish(serial_list, n_fallback, r | SER_PLAIN); }       // (flagged: plain chain_one_read there)
            else chain_one_read<false>(ix, o, r, n_reads, len, smems, smem_cnt, smem_off, sa_off, sa_coord, wchain, wseed, nodes, order, chn, seeds_out,
                                       seed_owner, n_chain_out, n_reg_out, n_chain0_out, -1, nullptr, 0);
        }
        }
    }
    if constexpr (OWN != 0) {                                        // this wavefront lists nothing any more: k_chain_serial stops waiting when every one has said so
        if (lane == 0) { __threadfence(); atomicAdd(n_fallback + SER_DONE, 1ULL); }
    }
}

// The reads k_chain_islands could not chain by islands (two chains with EQUAL keys: their order depends on the shape of the whole tree), chained again by the
// serial code -- the kbtree walk of mem_chain_seeds, bwamem.cpp:906-951, seed by seed on lane 0 -- in a launch of their own.  Inside the island kernel one such
// read cost 12 us per seed (five dependent node loads from global memory per insertion; 200 ms per read, 390 ms the slowest of a chunk of 20 000, and the
// kernel could not end before it did).  Here a workgroup is ONE wavefront with most of a CU's LDS: the tree's INTERNAL nodes live there (BTree::lnodes),
// so that a descent ends with its only global round trip at the leaf; the other 63 lanes stage the next 64 seeds (staged record, "alone in its island")
// into LDS while lane 0 walks, and all lanes share the weight test of mem_chain_flt (bwamem.cpp:516-528) at the end.
struct SerStage { int64_t rbeg; uint32_t ql; int32_t rid, single, pad; };
template <bool COOP>
__global__ void __launch_bounds__(64)
k_chain_serial(DevIndex ix, ChainParams o, int n_reads, const int32_t *__restrict__ len, const bm2_smem_t *__restrict__ smems, const int32_t *__restrict__ smem_cnt,
               const int64_t *__restrict__ smem_off, const int64_t *__restrict__ sa_off, const int64_t *__restrict__ sa_coord, WChain *wchain, WSeed *wseed, BtNode *nodes, int32_t *order,
               DevChain *chn, DevSeed *seeds_out, int32_t *seed_owner, int32_t *n_chain_out, int32_t *n_reg_out, int32_t *n_chain0_out,
               const int32_t *serial_list /* n_reads places, -1 = not yet listed: reads with equal chain keys, staged by k_chain_islands; | SER_PLAIN: reads it did not touch */,
               unsigned long long *n_fallback /* [0]: reads listed so far, [22]: this launch's work cursor, [SER_DONE]: producers that have left */, int l_cap,
               int n_producers, int hyb /* nodes in LDS probed in place (BM2_CHAIN_SERIAL_HYB) */) {
    extern __shared__ __attribute__((aligned(16))) uint8_t ser_lds[];
    BtNode *lnodes = (BtNode *)ser_lds;
    SerStage *stg = (SerStage *)(ser_lds + (size_t)l_cap * sizeof(BtNode));
    __shared__ DeferFinish df;
    __shared__ int s_n;
    const int lane = threadIdx.x;
    const unsigned long long lt_mask = lane ? (~0ULL >> (64 - lane)) : 0ULL;
    for (;;) {
        const unsigned long long it = atomicAdd(n_fallback + 22, lane == 0 ? 1ULL : 0ULL);
        const int item = __builtin_amdgcn_readfirstlane((int)it);
        if (item >= n_reads) break;
        int listed = -1;
        if (lane == 0) {                                             // wait for place `item` to be filled, or for the producers to have left without filling it
            long long t_w = wall_clock64();
            unsigned long long seen = ~0ULL;                         // (producers that have left + reads listed when the clock was last reset)
            for (;;) {
                listed = __hip_atomic_load(serial_list + item, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
                if (listed != -1) break;
                const unsigned long long gone = __hip_atomic_load(n_fallback + SER_DONE, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
                if (gone >= (unsigned long long)n_producers) {
                    listed = __hip_atomic_load(serial_list + item, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
                // the guard is against producers that do not MOVE (s_memrealtime ticks at 100 MHz: 20 s): any progress of theirs -- one more has left, one
                // more read is listed -- starts the clock again, however long the whole island launch takes
                const unsigned long long now = gone + __hip_atomic_load(n_fallback + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (now != seen) { seen = now; t_w = wall_clock64(); }
                else if (wall_clock64() - t_w > 2000000000LL) __builtin_trap();      // fail loudly, do not hang the device
                __builtin_amdgcn_s_sleep(64);
            }
        }
        listed = __builtin_amdgcn_readfirstlane(listed);
        if (listed == -1) break;
        isl_sync();                                                  // (every lane: what the producer staged is behind the list entry)
        if (listed & SER_PLAIN) {                                    // a read the island kernel had no table for (or nothing to chain)
            const int r = listed & ~SER_PLAIN;
            if (lane == 0) chain_one_read<false>(ix, o, r, n_reads, len, smems, smem_cnt, smem_off, sa_off, sa_coord, wchain, wseed, nodes, order, chn, seeds_out,
                                                 seed_owner, n_chain_out, n_reg_out, n_chain0_out, -1, nullptr, 0);
            continue;
        }
        const long long t_fb = wall_clock64();
        const int r = listed;
        const int n_sm = __builtin_amdgcn_readfirstlane(smem_cnt[r]);
        const int64_t so = smem_off[r];
        const int64_t base = sa_off[so];
        const int n_sa = (int)(sa_off[so + n_sm] - base);
        WChain *ch = wchain + base; WSeed *sd = wseed + base; int32_t *ord = order + base;
        const IslHash *hash = (const IslHash *)(chn + base);
        const IslSeed *st = (const IslSeed *)(seeds_out + base);
        const int32_t *cslot = (const int32_t *)(st + n_sa);
        BTree bt; bt.nodes = nodes + base; bt.n_nodes = 0; bt.n_keys = 0; bt.ch = ch; bt.reg = o.reg_nodes != 0;
        bt.lnodes = lnodes; bt.l_cap = l_cap; bt.n_l = 0;
        bt.root = bt_new<true>(bt, 0);
        int n_ch = 0, n_sd = 0;
        for (int t0 = 0; t0 < n_sa; t0 += 64) {
            if (t0 + lane < n_sa) {
                const IslSeed q = st[t0 + lane];
                SerStage e; e.rbeg = q.rbeg; e.ql = q.ql; e.rid = q.rid; e.pad = 0;
                e.single = q.rid >= 0 && hash[cslot[t0 + lane]].cnt == 1 ? 1 : 0;
                stg[lane] = e;
            }
            // (Tried and removed, profiles/r05p_config5_variants.txt: every lane walking the tree as it is down the LDS levels for its own seed and touching
            //  the leaf it reaches, so that lane 0's sixty-four descents find their leaves in the cache: the stage 476-481 ms with it, 463 without.)
            chain_wave_sync();
            if (lane == 0) {
                const int nb = n_sa - t0 < 64 ? n_sa - t0 : 64;
                for (int j = 0; j < nb; j++) {
                    const SerStage q = stg[j];
                    if (q.rid < 0) continue;                         // bwamem.cpp:915-919
                    WSeed sdd; sdd.rbeg = q.rbeg; sdd.qbeg = (int)(q.ql & 0x7fffu); sdd.len = (int)((q.ql >> 15) & 0xffffu); sdd.next = -1; sdd.pad = 0;
                    int to_add = 0;
                    // (a seed alone in its island can meet no chain -- see chain_one_read -- only the tree's shape is kept exact)
                    if (q.single) to_add = 1;
                    else if (bt.n_keys) {
                        const int lower = hyb ? bt_lower_hyb(bt, sdd.rbeg) : bt_lower<true>(bt, sdd.rbeg);
                        if (lower < 0) to_add = 1;
                        else {
                            const int m = test_and_merge(o, ix.l_pac, ch[lower], sdd, q.rid, sd, n_sd);
                            if (m == 2) n_sd++;
                            else if (m == 0) to_add = 1;
                        }
                    } else to_add = 1;
                    if (to_add) {                                    // bwamem.cpp:930-951
                        WChain c2;
                        c2.pos = sdd.rbeg; c2.last_rbeg = sdd.rbeg; c2.first_qbeg = sdd.qbeg; c2.last_qbeg = sdd.qbeg; c2.last_len = sdd.len;
                        c2.n = 1; c2.rid = q.rid; c2.is_alt = (int)(q.ql >> 31); c2.head = c2.tail = n_sd; c2.w = 0; c2.kept = 0; c2.first = -1; c2.pad = 0;
                        sd[n_sd] = sdd; n_sd++;
                        ch[n_ch] = c2;
                        if (hyb) bt_put_hyb(bt, n_ch, c2.pos); else bt_put<true>(bt, n_ch, c2.pos);
                        n_ch++;
                    }
                }
            }
            chain_wave_sync();
        }
        if (lane == 0) s_n = bt_traverse<true>(bt, ord);             // chains in key order (bwamem.cpp:958-962)
        isl_sync();
        const int n = s_n;
        // the weight test, 64 chains at a time; the survivors keep their order (the serial loop's `ord[k++] = ord[i]`: a block writes below what it has read)
        int k = 0;
        for (int i0 = 0; i0 < n; i0 += 64) {
            const int i = i0 + lane;
            int id = 0; bool keep = false;
            if (i < n) {
                id = ord[i];
                WChain &c = ch[id];
                c.first = -1; c.kept = 0;
                c.w = chain_weight(c, sd);
                keep = c.w >= o.min_chain_weight;
            }
            const unsigned long long m = __ballot(keep);
            if (keep) ord[k + __popcll(m & lt_mask)] = id;
            k += __popcll(m);
        }
        if (k == 0 && n > 0) k = 1;      // quirk: an empty survivor list still processes the untouched a_[0] (bwamem.cpp:529-546)
        const int l_rep = lrep_coop(smems + so, n_sm, o.max_occ, lane);      // l_rep, bwamem.cpp:849-861
        if (lane == 0) {
            if (n_chain0_out) n_chain0_out[r] = n;
            df.r = r; df.n = k; df.base = base; df.frac_rep = (float)l_rep / len[r]; df.ch = ch; df.sd = sd; df.ord = ord; df.kept = (int32_t *)(nodes + base);
            df.valid = 1;
        }
        isl_sync();
        {
            const DeferFinish d = df;
            if constexpr (COOP) chain_finish_coop(o, d, lane, chn, seeds_out, seed_owner, n_chain_out, n_reg_out);
            else if (lane == 0) chain_finish_read(o, d.r, d.ch, d.sd, d.ord, d.kept, d.n, d.base, d.frac_rep, chn, seeds_out, seed_owner, n_chain_out, n_reg_out);
        }
        isl_sync();
        if (lane == 0) {
            const unsigned long long dt_fb = (unsigned long long)(wall_clock64() - t_fb);
            atomicAdd(n_fallback + 12, dt_fb); atomicAdd(n_fallback + 13, (unsigned long long)n_sa); atomicMax(n_fallback + 14, dt_fb);
        }
    }
}

// After the (optional) short-seed filter: reference window, extension order and reg slots of every kept chain
// (the task-building part of mem_chain2aln_across_reads_V2, bwamem.cpp:2127-2223: chain_finish_one).  One read per lane, in the order of the chaining's
// permutation (reads with alike seed counts share a wavefront); with `done_thr` >= 0 the reads with at most that many SA coordinates have had
// this done by k_chain's lane (chain_finish_read) and only the seed-rich ones -- the wavefront-per-read launches' -- are left.
__global__ void __launch_bounds__(128)
k_chain_finish(DevIndex ix, ChainParams o, int n_reads, const int32_t *__restrict__ len, const int64_t *__restrict__ read_base,
               const int32_t *__restrict__ n_chain, DevChain *chn, DevSeed *seeds_out, int32_t *srt_out, int32_t *reg_seed,
               int32_t *reg_chain, int32_t *n_reg_out, const int32_t *__restrict__ perm, const int32_t *__restrict__ n_sa_read, int done_thr, int wave_thr) {
    const int tix = blockIdx.x * blockDim.x + threadIdx.x;
    if (tix >= n_reads) return;
    const int r = perm ? perm[tix] : tix;
    if (done_thr >= 0 && n_sa_read[r] <= done_thr) return;
    if (wave_thr >= 0 && n_sa_read[r] > wave_thr) return;       // k_chain_finish_wave's
    const int n = n_chain[r];
    if (n == 0) { n_reg_out[r] = 0; return; }
    const int64_t base = read_base[r];
    DevChain *oc = chn + base;
    DevSeed *os = seeds_out + base;
    const int l_query = len[r];
    int n_reg = 0;
    for (int i = 0; i < n; i++) {
        DevChain d = oc[i];
        chain_finish_one(ix, o, l_query, base, i, d, os, (int)(d.seed_off - base), srt_out, reg_seed, reg_chain, n_reg);
        oc[i] = d;
    }
    n_reg_out[r] = n_reg;
}

// The same for the seed-rich reads -- the first *n_heavy entries of the chaining's permutation, the reads of the wavefront-per-read launches --, one read
// per WAVEFRONT and one chain per lane: a read with hundreds of chains kept one lane of k_chain_finish busy for a millisecond (the kernel with only these
// reads left: 1.04 of its 1.51 ms per million-read chunk, profiles/r06ab_*).  A chain's first reg slot is the number of seeds in the chains before it: a
// prefix sum over the wavefront per 64 chains.
__global__ void __launch_bounds__(256)
k_chain_finish_wave(DevIndex ix, ChainParams o, const int32_t *__restrict__ perm, const int64_t *__restrict__ n_heavy_dev, const int32_t *__restrict__ len,
                    const int64_t *__restrict__ read_base, const int32_t *__restrict__ n_chain, DevChain *chn, DevSeed *seeds_out, int32_t *srt_out,
                    int32_t *reg_seed, int32_t *reg_chain, int32_t *n_reg_out) {
    const int lane = (int)(threadIdx.x & 63);
    const int64_t n_heavy = *n_heavy_dev, n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t wv = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; wv < n_heavy; wv += n_waves) {
        const int r = perm[wv];
        const int n = n_chain[r];
        const int64_t base = read_base[r];
        DevChain *oc = chn + base;
        DevSeed *os = seeds_out + base;
        const int l_query = len[r];
        int before = 0;                                         // seeds in the chains before this group of 64
        for (int i0 = 0; i0 < n; i0 += 64) {
            const int i = i0 + lane;
            DevChain d = {};
            if (i < n) d = oc[i];
            int inc = i < n ? d.n : 0;
            const int own = inc;
            for (int dd = 1; dd < 64; dd <<= 1) { const int v = __shfl(inc, (lane - dd) & 63); if (lane >= dd) inc += v; }
            int n_reg = before + inc - own;
            if (i < n) {
                chain_finish_one(ix, o, l_query, base, i, d, os, (int)(d.seed_off - base), srt_out, reg_seed, reg_chain, n_reg);
                oc[i] = d;
            }
            before += __shfl(inc, 63);
        }
        if (lane == 0) n_reg_out[r] = before;
    }
}

int bm2_launch_chain_finish(bm2_ctx *c, const ChainParams &o, int n_reads, const int32_t *len, const int64_t *read_base,
                            const int32_t *n_chain, DevChain *chn, DevSeed *seeds_out, int32_t *srt_out, int32_t *reg_seed,
                            int32_t *reg_chain, int32_t *n_reg_out, const int32_t *perm, bool lanes_by_perm, const int32_t *n_sa_read, int done_thr,
                            const int64_t *n_heavy_dev /* or NULL.  Set: the permutation's first *n_heavy_dev reads (more than wave_thr SA coordinates) by k_chain_finish_wave */, int wave_thr) {
    if (n_reads <= 0) return BM2_OK;
    const bool lanes = !(n_heavy_dev && done_thr >= wave_thr);      // (nothing left for the lane kernel when k_chain has done the reads below the bound)
    if (lanes) hipLaunchKernelGGL(k_chain_finish, dim3((n_reads + 127) / 128), dim3(128), 0, c->stream, c->ix, o, n_reads, len, read_base, n_chain,
                                  chn, seeds_out, srt_out, reg_seed, reg_chain, n_reg_out, lanes_by_perm ? perm : (const int32_t *)nullptr, n_sa_read, done_thr,
                                  n_heavy_dev ? wave_thr : -1);
    if (n_heavy_dev) {
        const int64_t waves = (int64_t)(n_reads < 64 * 1024 ? n_reads : 64 * 1024);
        hipLaunchKernelGGL(k_chain_finish_wave, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, c->stream, c->ix, o, perm, n_heavy_dev, len, read_base, n_chain,
                           chn, seeds_out, srt_out, reg_seed, reg_chain, n_reg_out);
    }
    return bm2_check(hipGetLastError(), "k_chain_finish launch");
}

size_t bm2_chain_lds_bytes(int cap, int stage) {      // working set (chains, seeds, order, B-tree nodes) + staged inputs (24 bytes per seed)
    return (size_t)cap * (sizeof(WChain) + sizeof(WSeed) + 4) + (size_t)(cap / 4 + 2) * sizeof(BtNode) + 16 + (stage ? (size_t)cap * 24 + 8 : 0);
}

int bm2_launch_chain(bm2_ctx *c, const ChainParams &o, int n_reads, const int32_t *len, const bm2_smem_t *smems,
                     const int32_t *smem_cnt, const int64_t *smem_off, const int64_t *sa_off, const int64_t *sa_coord,
                     WChain *wchain, WSeed *wseed, BtNode *nodes, int32_t *order, DevChain *chn, DevSeed *seeds_out,
                     int32_t *seed_owner,
                     int32_t *n_chain_out, int32_t *n_reg_out, int32_t *n_chain0_out, const int32_t *perm,
                     int heavy_thr, const int64_t *n_heavy_dev, const int32_t *n_sa_read, unsigned long long *item_cur /* one per tier + 2 */, int max_len,
                     int32_t *isl_cut /* scratch of the island kernel: one int per SA coordinate */,
                     const int32_t *isl_order /* or NULL: every read, the seed-richest first -- the island kernel's longest reads start first */,
                     int32_t *isl_serial /* or NULL: one int per read (all -1), the list of reads k_chain_islands leaves to k_chain_serial */,
                     const FinishOut *fuse /* or NULL.  Set (no read of the batch can meet the seed filter): k_chain's lanes do k_chain_finish's part for their reads */) {
    if (n_reads <= 0) return BM2_OK;
    hipStream_t s = c->stream;
    const bool heavy = heavy_thr >= 0 && n_heavy_dev != nullptr;
    if (heavy && bm2_side_streams(c)) return BM2_ENODEV;
    if (heavy) (void)hipEventRecord(c->ev_fork, s);
    // With the wavefront-per-read launches beside it, the lane-per-read kernel goes to a side stream of its own and the main stream only
    // waits, and waits for all of them AFTER the last is queued: streams share hardware queues, nothing tells which, and a launch whose
    // stream sits on the main stream's queue starts behind whatever the main stream queued before it -- the fourth tier started when
    // k_chain had ended, 7.6 ms late (profiles/r04_timeline.tsv).  BM2_CHAIN_MAIN_SIDE=0: k_chain on the main stream, as before.
    const bool main_side = heavy && bm2_knob("BM2_CHAIN_MAIN_SIDE", 1);
    hipStream_t s_main = main_side ? c->side_stream[1] : s;
    if (main_side) (void)hipStreamWaitEvent(s_main, c->ev_fork, 0);
    hipLaunchKernelGGL(k_chain, dim3((n_reads + 127) / 128), dim3(128), 0, s_main, c->ix, o, n_reads, len, smems, smem_cnt,
                       smem_off, sa_off, sa_coord, wchain, wseed, nodes, order, chn, seeds_out, seed_owner,
                       n_chain_out, n_reg_out, n_chain0_out, perm, heavy ? heavy_thr : -1, n_sa_read, fuse ? *fuse : FinishOut{ nullptr, nullptr, nullptr, nullptr });
    int joined[BM2_CHAIN_TIERS + 4], n_joined = 0;
    if (main_side) { (void)hipEventRecord(c->ev_join[1], s_main); joined[n_joined++] = 1; }
    if (heavy) {
        // tiers by seed count (LDS per block follows the tier): the launches run beside the lane-per-read kernel and each other
        const int stage = bm2_knob("BM2_CHAIN_STAGE", 1);     // (sweep of round 3, profiles/r03a_sweep.json: chaining 12.4 -> 10.3 ms)
        // The tiers' launches share the CUs' LDS (together they ask for four times what there is) and a block reserves its tier's cap whatever its
        // read holds: with caps a factor 2 apart a read uses 60-70 % of its block's LDS on average.  (Eight tiers with caps where the number of blocks per
        // CU changes bought 0.2 ms -- profiles/r04w_* -- and left the tree in round 6: the 257..512-seed reads are the pole whatever tier they sit in.)
        // BM2_CHAIN_CLOCK=1: the wavefront-per-read launches clock their reads (counters[43..47] of the batch: staging, mem_chain_seeds, the rest, reads, seeds)
        unsigned long long *clk = bm2_knob("BM2_CHAIN_CLOCK", 0) ? item_cur + CHAIN_CUR_EXTRA + 3 : (unsigned long long *)nullptr;
        // BM2_CHAIN_COOP_FLT: the wavefront-per-read launches run mem_chain_flt's walk over the kept chains with all 64 lanes.  Measured in round 5 (with
        // wavefront-scope fences: profiles/r05b / r05d / r05e_sweep.json): chaining of the 150 bp workload 11.5-12.4 -> 10.9-11.1 ms in three sweeps (ON for
        // short reads); a chunk of 10 kb reads -- the island kernel, a handful of kept chains per read -- 454 -> 463 ms (OFF there)
        const int coop = bm2_knob("BM2_CHAIN_COOP_FLT", max_len < 1000 ? 1 : 0);
        const int last_cap = stage ? 1000 : 1184;                 // (the last tier fills a CU's 160 KB of LDS)
        const int caps[5] = { 64, 128, 256, 512, last_cap };
        const int n_tiers = 5;
        const int wpe = bm2_knob("BM2_CHAIN_HEAVY_WPE", 3);
        auto k_heavy = coop ? (wpe >= 4 ? k_chain_heavy<true, 4> : wpe == 3 ? k_chain_heavy<true, 3> : k_chain_heavy<true, 2>)
                            : (wpe >= 4 ? k_chain_heavy<false, 4> : wpe == 3 ? k_chain_heavy<false, 3> : k_chain_heavy<false, 2>);
        { const int which = (coop ? 2 : 0) + 8 * (wpe >= 4 ? 2 : wpe == 3 ? 1 : 0);        // (one flag per instantiation: the limit is a property of the kernel)
          const int rc_a = bm2_raise_lds_limit(c, which, (const void *)k_heavy, coop ? 160 * 1024 - 256 : 160 * 1024);       // (the cooperative one's static record rides on top of the dynamic LDS)
          if (rc_a) return rc_a; }
        // reads with more seeds than the largest tier holds: a launch of their own where they are the norm (long reads), otherwise the last tier's
        const bool own_overflow = max_len >= bm2_knob("BM2_CHAIN_OVF_MIN_LEN", 1000);
        int lo = heavy_thr;
        // BM2_CHAIN_TIER_MAX: tiers beyond it are left out and their reads -- the seed-richest -- go to the island kernel with the long reads.  Short-read chunks
        // leave out the last tier (513..1000 seeds: a workgroup of it reserves a CU's whole LDS for one read) since the end of round 6: chaining 8.2 -> 7.8 ms in
        // four adjacent pairs of one process and in three earlier sweeps (profiles/r06au_*, r06aq_*, r06ar_*); without the 257..512 tier as well: 8.4-8.6 (and
        // 12.7-15.9 ms in round 4, r04j_sweep.json).  Long-read chunks keep all five (their reads are the island kernel's anyway, by `own_overflow`).
        const int tier_max = bm2_knob("BM2_CHAIN_TIER_MAX", max_len < 1000 ? 512 : 1 << 30);
        const bool use_islands = own_overflow || caps[n_tiers - 1] > tier_max;
        for (int t = 0; t < n_tiers; t++) {
            if (caps[t] <= lo || caps[t] > tier_max) continue;
            hipStream_t sk = c->side_stream[2 + t];
            const size_t lds = bm2_chain_lds_bytes(caps[t], stage);
            const int per_cu_max = bm2_knob("BM2_CHAIN_WAVES_PER_CU", 16);
            int per_cu = (int)(160 * 1024 / lds); if (per_cu < 1) per_cu = 1; if (per_cu > per_cu_max) per_cu = per_cu_max;
            (void)hipStreamWaitEvent(sk, c->ev_fork, 0);
            hipLaunchKernelGGL(k_heavy, dim3(c->n_cu * per_cu), dim3(64), lds, sk, c->ix, o, n_reads, len, smems, smem_cnt, smem_off, sa_off,
                               sa_coord, wchain, wseed, nodes, order, chn, seeds_out, seed_owner, n_chain_out, n_reg_out, n_chain0_out, perm,
                               n_heavy_dev, n_sa_read, lo, caps[t], (t == n_tiers - 1 && !use_islands) ? 1 : 0,
                               item_cur + (t < CHAIN_CUR_SLOTS ? t : CHAIN_CUR_EXTRA + t - CHAIN_CUR_SLOTS), stage, clk);
            (void)hipEventRecord(c->ev_join[2 + t], sk);
            joined[n_joined++] = 2 + t;
            lo = caps[t];
        }
        // Reads with more seeds than the largest tier holds in LDS (long reads: ~12 k seeds per 10 kb read) walk their GLOBAL slices.  They
        // used to ride in the last tier -- whose blocks reserve a CU's whole LDS, so only one such walk ran per CU: 256 at a time, 5.1 s for
        // a chunk of 10 000 ONT-like reads.  A launch of their own with (almost) no LDS: as many walks in flight as the CUs hold wavefronts.
        // (Short-read chunks keep the old routing: a read of theirs beyond 1000 seeds is a rarity, and one more launch scanning the heavy list is not free.)
        if (use_islands) {
            hipStream_t sk = c->side_stream[2 + BM2_CHAIN_TIERS];
            (void)hipStreamWaitEvent(sk, c->ev_fork, 0);
            if (bm2_knob("BM2_CHAIN_ISLANDS", 1)) {                 // chaining by islands (k_chain_islands): one wavefront per read, every lane at work
                // (wavefronts per CU: 32 asked for, 16 resident at 128 registers -- the kernel itself is shortest there, 298 ms for 20 000 long reads, but every read
                //  is three times slower than at 4 per CU, and the stage ends when the seed-richest read with equal keys has gone through this kernel AND
                //  k_chain_serial: 6 per CU gave the shortest stage -- 4: 551 ms, 6: 474, 8: 542-557, 12: 574-589, 16: 594 -- profiles/r05p_config5_variants.txt)
                const int per_cu = bm2_knob("BM2_CHAIN_ISL_WAVES_PER_CU", isl_serial ? 6 : 32);
                const int isl_wpe = bm2_knob("BM2_CHAIN_ISL_WPE", 4);
                auto k_isl = !isl_serial ? (coop ? k_chain_islands<true, 0> : k_chain_islands<false, 0>)
                           : isl_wpe >= 4 ? (coop ? k_chain_islands<true, 4> : k_chain_islands<false, 4>) : (coop ? k_chain_islands<true, 3> : k_chain_islands<false, 3>);
                hipLaunchKernelGGL(k_isl, dim3(c->n_cu * per_cu), dim3(64), 0, sk, c->ix, o, n_reads, len, smems, smem_cnt, smem_off, sa_off,
                                   sa_coord, wchain, wseed, nodes, order, chn, seeds_out, seed_owner, isl_cut, n_chain_out, n_reg_out, n_chain0_out,
                                   isl_order ? isl_order : perm, n_heavy_dev, n_sa_read, lo, item_cur + CHAIN_CUR_SLOTS, item_cur + CHAIN_CUR_SLOTS + 1,
                                   isl_order ? n_reads : -1, isl_serial);
                if (isl_serial) {
                    // the reads with equal chain keys, one wavefront per CU each with the internal nodes of its tree in LDS (BM2_CHAIN_SERIAL_LNODES of 160 bytes),
                    // on a stream of its own BEHIND the island kernel's launch (never before it: a consumer must not hold a queue its producer waits in): it takes a
                    // read as soon as the island kernel lists it (BM2_CHAIN_SERIAL_BESIDE=0: after the island kernel, same stream)
                    const bool beside = bm2_knob("BM2_CHAIN_SERIAL_BESIDE", 1) != 0;
                    hipStream_t sc2 = beside ? c->side_stream[11] : sk;
                    if (beside) (void)hipStreamWaitEvent(sc2, c->ev_fork, 0);
                    int l_cap = bm2_knob("BM2_CHAIN_SERIAL_LNODES", 960);
                    if (l_cap < 1) l_cap = 1; if (l_cap > 1000) l_cap = 1000;
                    const size_t lds_s = (size_t)l_cap * sizeof(BtNode) + 64 * sizeof(SerStage);
                    auto k_ser = coop ? k_chain_serial<true> : k_chain_serial<false>;
                    { const int rc_a = bm2_raise_lds_limit(c, coop ? 3 : 4, (const void *)k_ser, 160 * 1024 - 512); if (rc_a) return rc_a; }
                    const int per_cu_s = (int)((160 * 1024 - 512) / lds_s) < 1 ? 1 : (int)((160 * 1024 - 512) / lds_s);
                    // A resident consumer must never be what keeps its producer from starting: its workgroups pin most of a CU's LDS each while they wait, and
                    // the island launch may sit in a hardware queue behind a tier launch whose workgroups (10-156 KB of LDS) have not all been placed.  Beside
                    // the producer the consumer's grid therefore covers THREE QUARTERS of the CUs at most: on the others anything queued ahead of the island
                    // kernel keeps running, so the island kernel starts and the list fills (a chunk of 20 000 long reads lists ~150: no shorter for it).
                    int grid_s = c->n_cu * (per_cu_s > 8 ? 8 : per_cu_s);
                    if (beside) { const int cus = c->n_cu - (c->n_cu + 3) / 4; grid_s = (cus < 1 ? 1 : cus) * (per_cu_s > 8 ? 8 : per_cu_s); }
                    hipLaunchKernelGGL(k_ser, dim3(grid_s), dim3(64), lds_s, sc2, c->ix, o, n_reads, len, smems, smem_cnt, smem_off, sa_off, sa_coord,
                                       wchain, wseed, nodes, order, chn, seeds_out, seed_owner, n_chain_out, n_reg_out, n_chain0_out,
                                       (const int32_t *)isl_serial, item_cur + CHAIN_CUR_SLOTS + 1, l_cap, c->n_cu * per_cu,
                                       o.reg_nodes ? bm2_knob("BM2_CHAIN_SERIAL_HYB", 1) : 0);
                    if (beside) { (void)hipEventRecord(c->ev_join[11], sc2); joined[n_joined++] = 11; }
                }
            } else {
                const int per_cu = bm2_knob("BM2_CHAIN_OVF_WAVES_PER_CU", 32);
                hipLaunchKernelGGL(k_heavy, dim3(c->n_cu * per_cu), dim3(64), bm2_chain_lds_bytes(0, 0), sk, c->ix, o, n_reads, len, smems, smem_cnt, smem_off, sa_off,
                                   sa_coord, wchain, wseed, nodes, order, chn, seeds_out, seed_owner, n_chain_out, n_reg_out, n_chain0_out, perm,
                                   n_heavy_dev, n_sa_read, lo, 0, 1, item_cur + CHAIN_CUR_SLOTS, 0, clk);
            }
            (void)hipEventRecord(c->ev_join[2 + BM2_CHAIN_TIERS], sk);
            joined[n_joined++] = 2 + BM2_CHAIN_TIERS;
        }
    }
    for (int i = 0; i < n_joined; i++) (void)hipStreamWaitEvent(s, c->ev_join[joined[i]], 0);
    return bm2_check(hipGetLastError(), "k_chain launch");
}
