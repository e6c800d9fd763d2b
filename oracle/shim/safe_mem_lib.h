/* See safe_str_lib.h in this directory: stand-in for the un-vendored
 * intel/safestringlib submodule, test infrastructure only. */
#ifndef BM2_ORACLE_SAFE_MEM_SHIM_H
#define BM2_ORACLE_SAFE_MEM_SHIM_H
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stddef.h>
#include "safe_str_lib.h"   /* the reference reaches the string helpers through this header too */
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
static inline errno_t memcpy_s(void *d, rsize_t dmax, const void *s, rsize_t n) {
    if (!d || !s) return 1;
    if (n > dmax) return 2;
    memcpy(d, s, n);
    return 0;
}
#ifdef __cplusplus
}
#endif
#endif
