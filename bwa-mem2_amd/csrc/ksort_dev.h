// ksort_dev.h -- klib's introsort on the device (internal).  The permutation klib gives to EQUAL keys is observable downstream
// (chain filter ties, hit order), so the algorithm is reproduced step by step; it sorts an index array through a strict order on
// the referenced records (the comparisons, and with them the permutation, are those of sorting the records themselves).
#pragma once
#include <stdint.h>

// klib introsort (ksort.h:185-236) on an index array
// lt(a, b) is a strict order on the referenced records; the permutation of equal keys must match klib exactly
// because mem_flt ties are observable.
template <class LT>
static __device__ void k_insertsort(int32_t *s, int32_t *t, LT lt) {
    for (int32_t *i = s + 1; i < t; ++i)
        for (int32_t *j = i; j > s && lt(*j, *(j - 1)); --j) { int32_t tmp = *j; *j = *(j - 1); *(j - 1) = tmp; }
}
template <class LT>
static __device__ void k_combsort(int n, int32_t *a, LT lt) {
    const double shrink = 1.2473309501039786540366528676643;
    int do_swap, gap = n;
    do {
        if (gap > 2) { gap = (int)(gap / shrink); if (gap == 9 || gap == 10) gap = 11; }
        do_swap = 0;
        for (int32_t *i = a; i < a + n - gap; ++i) {
            int32_t *j = i + gap;
            if (lt(*j, *i)) { int32_t tmp = *i; *i = *j; *j = tmp; do_swap = 1; }
        }
    } while (do_swap || gap > 2);
    if (gap != 1) k_insertsort(a, a + n, lt);
}
template <class LT>
static __device__ void k_introsort(int n, int32_t *a, LT lt) {
    if (n < 1) return;
    if (n == 2) { if (lt(a[1], a[0])) { int32_t t = a[0]; a[0] = a[1]; a[1] = t; } return; }
    int d;
    for (d = 2; (1 << d) < n; ++d) {}
    int32_t *stk_l[72], *stk_r[72]; int stk_d[72]; int top = 0;
    int32_t *s = a, *t = a + (n - 1);
    d <<= 1;
    for (;;) {
        if (s < t) {
            if (--d == 0) { k_combsort((int)(t - s) + 1, s, lt); t = s; continue; }
            int32_t *i = s, *j = t, *k = i + ((j - i) >> 1) + 1;
            if (lt(*k, *i)) { if (lt(*k, *j)) k = j; }
            else k = lt(*j, *i) ? i : j;
            const int32_t rp = *k;
            if (k != t) { int32_t tmp = *k; *k = *t; *t = tmp; }
            for (;;) {
                do ++i; while (lt(*i, rp));
                do --j; while (i <= j && lt(rp, *j));
                if (j <= i) break;
                int32_t tmp = *i; *i = *j; *j = tmp;
            }
            { int32_t tmp = *i; *i = *t; *t = tmp; }
            if (i - s > t - i) {
                if (i - s > 16) { stk_l[top] = s; stk_r[top] = i - 1; stk_d[top] = d; ++top; }
                s = t - i > 16 ? i + 1 : t;
            } else {
                if (t - i > 16) { stk_l[top] = i + 1; stk_r[top] = t; stk_d[top] = d; ++top; }
                t = i - s > 16 ? i - 1 : s;
            }
        } else {
            if (top == 0) { k_insertsort(a, a + n, lt); return; }
            --top; s = stk_l[top]; t = stk_r[top]; d = stk_d[top];
        }
    }
}


// The same algorithm as ONE loop over an explicit state (no loops nested in loops; indices, not pointers, so the array keeps its
// address space): the form the hit-finishing kernels use.  Same comparisons in the same order as k_introsort above, hence the same
// permutation.
template <class LT>
static __device__ void k_introsort_flat(int n, int32_t *a, LT lt) {
    if (n < 1) return;
    if (n == 2) { if (lt(a[1], a[0])) { const int32_t t = a[0]; a[0] = a[1]; a[1] = t; } return; }
    int d = 2;
    while ((1 << d) < n) ++d;
    d <<= 1;
    int stk_s[72], stk_t[72], stk_d[72], top = 0;
    int s = 0, t = n - 1, i = 0, j = 0;
    int32_t rp = 0;
    enum { ST_TOP, ST_I, ST_J, ST_DONE };
    int st = ST_TOP;
    while (st != ST_DONE) {
        if (st == ST_TOP) {
            if (s < t) {
                if (--d == 0) { k_combsort(t - s + 1, a + s, lt); t = s; }
                else {
                    i = s; j = t;
                    int k = i + ((j - i) >> 1) + 1;
                    if (lt(a[k], a[i])) { if (lt(a[k], a[j])) k = j; }
                    else k = lt(a[j], a[i]) ? i : j;
                    rp = a[k];
                    if (k != t) { const int32_t tmp = a[k]; a[k] = a[t]; a[t] = tmp; }
                    st = ST_I;
                }
            } else if (top == 0) st = ST_DONE;
            else { --top; s = stk_s[top]; t = stk_t[top]; d = stk_d[top]; }
        } else if (st == ST_I) {                                  // do ++i; while (lt(*i, rp));
            ++i;
            if (!lt(a[i], rp)) st = ST_J;
        } else {                                                  // do --j; while (i <= j && lt(rp, *j));
            --j;
            if (!(i <= j && lt(rp, a[j]))) {
                if (j <= i) {                                     // the partition is through
                    { const int32_t tmp = a[i]; a[i] = a[t]; a[t] = tmp; }
                    if (i - s > t - i) {
                        if (i - s > 16) { stk_s[top] = s; stk_t[top] = i - 1; stk_d[top] = d; ++top; }
                        s = t - i > 16 ? i + 1 : t;
                    } else {
                        if (t - i > 16) { stk_s[top] = i + 1; stk_t[top] = t; stk_d[top] = d; ++top; }
                        t = i - s > 16 ? i - 1 : s;
                    }
                    st = ST_TOP;
                } else { const int32_t tmp = a[i]; a[i] = a[j]; a[j] = tmp; st = ST_I; }
            }
        }
    }
    k_insertsort(a, a + n, lt);
}
