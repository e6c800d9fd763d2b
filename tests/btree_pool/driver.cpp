// tests/btree_pool/driver.cpp -- TEST INFRASTRUCTURE (tests/test_btree_pool.py): the kbtree restatement of chain.hip compiled for the host (the emulator's
// rewrite of the source is #included below), all its forms driven by ONE thread over the same insertion sequences:
//   plain   nodes in one pool, probed in memory            (bt_put / bt_lower with reg = false: what k_chain_heavy runs on LDS nodes)
//   reg     nodes in one pool, visited through registers   (reg = true: k_chain, k_chain_islands)
//   split   internal nodes in a second ("LDS") pool of a given capacity, leaves and the overflow in the first (bt_put<true>, k_chain_serial with BM2_CHAIN_SERIAL_HYB=0)
//   hyb     the same pools, second-pool nodes probed in place (bt_put_hyb / bt_lower_hyb: k_chain_serial)
// Node numbers differ between the forms; what must be equal is everything the chaining code sees: the in-order sequence of keys (equal keys included:
// their order depends on the shape of the whole tree, kbtree.h:197-231) and the lower neighbour kb_intervalp gives for any position at any time.
#include CHAIN_EMU_CPP
#include <random>
#include <vector>
#include <cstdio>

namespace {
struct Form { int kind; int l_cap; BTree bt; std::vector<BtNode> nodes, lnodes; std::vector<int32_t> ord; };

void form_init(Form &f, int kind, int l_cap, int n_max, const WChain *ch) {
    f.kind = kind; f.l_cap = l_cap;
    f.nodes.assign((size_t)n_max + 8, BtNode()); f.lnodes.assign((size_t)(l_cap > 0 ? l_cap : 1), BtNode()); f.ord.assign((size_t)n_max + 8, 0);
    f.bt = BTree(); f.bt.nodes = f.nodes.data(); f.bt.n_nodes = 0; f.bt.n_keys = 0; f.bt.ch = ch; f.bt.reg = kind != 0;
    f.bt.lnodes = f.lnodes.data(); f.bt.l_cap = kind >= 2 ? l_cap : 0; f.bt.n_l = 0;
    f.bt.root = kind >= 2 ? bt_new<true>(f.bt, 0) : bt_new(f.bt, 0);
}
void form_put(Form &f, int key, int64_t k) {
    if (f.kind == 3) bt_put_hyb(f.bt, key, k);
    else if (f.kind == 2) bt_put<true>(f.bt, key, k);
    else bt_put(f.bt, key, k);
}
int form_lower(Form &f, int64_t k) {
    return f.kind == 3 ? bt_lower_hyb(f.bt, k) : f.kind == 2 ? bt_lower<true>(f.bt, k) : bt_lower(f.bt, k);
}
int form_traverse(Form &f) { return f.kind >= 2 ? bt_traverse<true>(f.bt, f.ord.data()) : bt_traverse(f.bt, f.ord.data()); }
}  // namespace

// n keys from [0, span) (a small span: many equal keys), `queries` lower-bound look-ups between insertions; returns 0 or the number of the first check that failed
extern "C" int bt_pool_check(unsigned seed, int n, long long span, int l_cap, int queries, long long *detail) {
    std::mt19937_64 rng(seed);
    std::vector<WChain> ch((size_t)n);
    Form f[4];
    for (int v = 0; v < 4; v++) form_init(f[v], v, l_cap, n, ch.data());
    for (int i = 0; i < n; i++) {
        const int64_t k = (int64_t)(rng() % (unsigned long long)span);
        ch[(size_t)i] = WChain(); ch[(size_t)i].pos = k;
        if (i) for (int q = 0; q < queries; q++) {
            const int64_t x = (int64_t)(rng() % (unsigned long long)(span + 2)) - 1;
            const int l0 = form_lower(f[0], x);
            for (int v = 1; v < 4; v++) if (form_lower(f[v], x) != l0) { detail[0] = i; detail[1] = v; detail[2] = x; return 1; }
        }
        for (int v = 0; v < 4; v++) form_put(f[v], i, k);
        if (i % 97 == 0 || i == n - 1) {
            const int m = form_traverse(f[0]);
            if (m != i + 1) { detail[0] = i; detail[1] = 0; detail[2] = m; return 2; }
            for (int t = 1; t < m; t++) if (ch[(size_t)f[0].ord[t - 1]].pos > ch[(size_t)f[0].ord[t]].pos) { detail[0] = i; detail[1] = t; return 3; }
            for (int v = 1; v < 4; v++) {
                if (form_traverse(f[v]) != m) { detail[0] = i; detail[1] = v; return 4; }
                for (int t = 0; t < m; t++) if (f[v].ord[t] != f[0].ord[t]) { detail[0] = i; detail[1] = v; detail[2] = t; return 5; }
            }
        }
    }
    detail[0] = f[3].bt.n_l; detail[1] = f[3].bt.n_nodes; detail[2] = f[0].bt.n_nodes;
    if (f[3].bt.n_l + f[3].bt.n_nodes != f[0].bt.n_nodes || f[2].bt.n_l != f[3].bt.n_l) return 6;       // the same tree: the same number of nodes, however they are housed
    return 0;
}
