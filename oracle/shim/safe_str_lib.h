/* Minimal stand-in for intel/safestringlib (an un-vendored git submodule of the
 * reference, see /root/reference/.gitmodules).  Only used when compiling the
 * reference sources into oracle/_ref/ as test infrastructure.  The reference
 * uses five helpers (strcpy_s, strcat_s, strncpy_s, strncat_s, memcpy_s); no
 * alignment arithmetic lives there.  Written from the C11 Annex K contract. */
#ifndef BM2_ORACLE_SAFE_STR_SHIM_H
#define BM2_ORACLE_SAFE_STR_SHIM_H
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif
#ifndef BM2_SHIM_TYPES
#define BM2_SHIM_TYPES
typedef int errno_t;
typedef size_t rsize_t;
#endif
#ifndef RSIZE_MAX_MEM
#define RSIZE_MAX_MEM (256UL << 20)
#endif
#ifndef RSIZE_MAX_STR
#define RSIZE_MAX_STR (4UL << 10)
#endif
static inline errno_t strcpy_s(char *d, rsize_t dmax, const char *s) {
    size_t n;
    if (!d || !s || dmax == 0) return 1;
    n = strlen(s);
    if (n + 1 > dmax) { d[0] = 0; return 2; }
    memcpy(d, s, n + 1);
    return 0;
}
static inline errno_t strcat_s(char *d, rsize_t dmax, const char *s) {
    size_t a, b;
    if (!d || !s || dmax == 0) return 1;
    a = strlen(d); b = strlen(s);
    if (a + b + 1 > dmax) return 2;
    memcpy(d + a, s, b + 1);
    return 0;
}
static inline errno_t strncpy_s(char *d, rsize_t dmax, const char *s, rsize_t n) {
    size_t l;
    if (!d || !s || dmax == 0) return 1;
    l = strnlen(s, n);
    if (l + 1 > dmax) { d[0] = 0; return 2; }
    memcpy(d, s, l); d[l] = 0;
    return 0;
}
static inline errno_t strncat_s(char *d, rsize_t dmax, const char *s, rsize_t n) {
    size_t a, l;
    if (!d || !s || dmax == 0) return 1;
    a = strlen(d); l = strnlen(s, n);
    if (a + l + 1 > dmax) return 2;
    memcpy(d + a, s, l); d[a + l] = 0;
    return 0;
}
#ifdef __cplusplus
}
#endif
#endif
