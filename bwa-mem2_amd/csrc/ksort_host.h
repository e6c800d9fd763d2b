// ksort_host.h -- klib introsort (ksort.h:185-236) restated as templates over a strict order: equal keys must end up in
// klib's permutation, because the reference's tie order is observable downstream (shared by finish_regs.cpp and sam_se.cpp)
#pragma once
#include <stddef.h>
#include <vector>

template <class T, class LT> void k_insertsort(T *s, T *t, LT lt) {
    for (T *i = s + 1; i < t; ++i)
        for (T *j = i; j > s && lt(*j, *(j - 1)); --j) { T tmp = *j; *j = *(j - 1); *(j - 1) = tmp; }
}
template <class T, class LT> void k_combsort(size_t n, T *a, LT lt) {
    const double shrink = 1.2473309501039786540366528676643;
    int do_swap; size_t gap = n;
    do {
        if (gap > 2) { gap = (size_t)(gap / shrink); if (gap == 9 || gap == 10) gap = 11; }
        do_swap = 0;
        for (T *i = a; i < a + n - gap; ++i) {
            T *j = i + gap;
            if (lt(*j, *i)) { T tmp = *i; *i = *j; *j = tmp; do_swap = 1; }
        }
    } while (do_swap || gap > 2);
    if (gap != 1) k_insertsort(a, a + n, lt);
}
template <class T, class LT> void k_introsort(size_t n, T *a, LT lt) {
    if (n < 1) return;
    if (n == 2) { if (lt(a[1], a[0])) { T t = a[0]; a[0] = a[1]; a[1] = t; } return; }
    int d;
    for (d = 2; (1ul << d) < n; ++d) {}
    struct Fr { T *l, *r; int d; };
    std::vector<Fr> stack((size_t)(sizeof(size_t) * d) + 2);
    size_t top = 0;
    T *s = a, *t = a + (n - 1);
    d <<= 1;
    for (;;) {
        if (s < t) {
            if (--d == 0) { k_combsort((size_t)(t - s) + 1, s, lt); t = s; continue; }
            T *i = s, *j = t, *k = i + ((j - i) >> 1) + 1;
            if (lt(*k, *i)) { if (lt(*k, *j)) k = j; }
            else k = lt(*j, *i) ? i : j;
            T rp = *k;
            if (k != t) { T tmp = *k; *k = *t; *t = tmp; }
            for (;;) {
                do ++i; while (lt(*i, rp));
                do --j; while (i <= j && lt(rp, *j));
                if (j <= i) break;
                T tmp = *i; *i = *j; *j = tmp;
            }
            { T tmp = *i; *i = *t; *t = tmp; }
            if (i - s > t - i) {
                if (i - s > 16) { stack[top].l = s; stack[top].r = i - 1; stack[top].d = d; ++top; }
                s = t - i > 16 ? i + 1 : t;
            } else {
                if (t - i > 16) { stack[top].l = i + 1; stack[top].r = t; stack[top].d = d; ++top; }
                t = i - s > 16 ? i - 1 : s;
            }
        } else {
            if (top == 0) { k_insertsort(a, a + n, lt); return; }
            --top; s = stack[top].l; t = stack[top].r; d = stack[top].d;
        }
    }
}

