// chain_dev.h -- working records of the chaining kernel (internal).
#pragma once
#include "bm2_dev.h"

struct WSeed {                       // a seed while chains are being built: singly linked in arrival order
    int64_t rbeg;
    int32_t qbeg, len, next, pad;
};

struct WChain {                      // mem_chain_t while it is in the B-tree; first/last seed fields are cached so that
    int64_t pos;                     // test_and_merge never walks the list.  pos = rbeg of the first seed (the key)
    int64_t last_rbeg;
    int32_t first_qbeg, last_qbeg, last_len;
    int32_t n, rid, is_alt, head, tail, w, kept, first, pad;
};

struct BtNode {                      // kbnode_t with t = 5: up to 9 keys (chain indices) and 10 children (node indices).
    int64_t kpos[9];                 // the key's sort field (chain.pos) is kept next to the index so that a search touches the
    int32_t key[9];                  // node only (one memory round trip per level instead of one per probe)
    int32_t ptr[10];
    int32_t is_internal, n, pad;
};

// where k_chain_finish's outputs go when the chaining kernel's lane writes them itself (chain.hip: chain_finish_one)
struct FinishOut { const int32_t *len; int32_t *srt_out, *reg_seed, *reg_chain; };
