"""Shared test helpers: random extension-like task generators."""
import numpy as np


def random_pairs(seed, n, max_len=150, h0_max=150, n_frac=0.005, div=(0.02, 0.15), long_tail=True):
    """Extension-like (query, target, h0) triples: target = mutated copy of query + flank, like the pairs
    mem_chain2aln builds (bwamem.cpp:2229-2418)."""
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        ql = int(rng.integers(1, max_len + 1))
        q = rng.integers(0, 4, size=ql, dtype=np.uint8)
        d = rng.uniform(*div)
        t = []
        i = 0
        while i < ql:
            u = rng.random()
            if u < d * 0.7:
                t.append((q[i] + rng.integers(1, 4)) % 4); i += 1
            elif u < d * 0.85:
                i += int(rng.integers(1, 4))                       # deletion from target
            elif u < d:
                t.extend(rng.integers(0, 4, size=int(rng.integers(1, 4))).tolist())   # insertion in target
            else:
                t.append(q[i]); i += 1
        if rng.random() < 0.15:                                     # big gap
            p = int(rng.integers(0, len(t) + 1))
            t[p:p] = rng.integers(0, 4, size=int(rng.integers(5, 40))).tolist()
        if rng.random() < 0.2 and len(t) > 10:                      # divergence to trigger z-drop
            p = int(rng.integers(len(t) // 2, len(t)))
            t[p:] = rng.integers(0, 4, size=len(t) - p).tolist()
        if long_tail:
            t.extend(rng.integers(0, 4, size=int(rng.integers(0, max(2, ql)))).tolist())
        t = np.array(t if t else [0], dtype=np.uint8) % 4
        if rng.random() < n_frac * 20:
            q[int(rng.integers(0, ql))] = 4
        h0 = int(rng.integers(1, h0_max + 1))
        out.append((q, t, h0))
    return out


def pack_pairs(bm2, triples):
    """-> (pairs SEQPAIR_DT array, ref bytes, qer bytes) in the S1 layout (flat seqBuf arrays, bandedSWA.h:90-99)."""
    pairs = np.zeros(len(triples), bm2.SEQPAIR_DT)
    refs, qers = [], []
    ro = qo = 0
    for i, (q, t, h0) in enumerate(triples):
        pairs[i]["idr"], pairs[i]["idq"], pairs[i]["id"] = ro, qo, i
        pairs[i]["len1"], pairs[i]["len2"], pairs[i]["h0"] = len(t), len(q), h0
        refs.append(t); qers.append(q)
        ro += len(t); qo += len(q)
    return pairs, np.concatenate(refs), np.concatenate(qers)
