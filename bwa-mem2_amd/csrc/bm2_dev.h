// bm2_dev.h -- internal types shared by the HIP kernels and the C-ABI glue (not installed).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/bm2.h"
#include "refseq.h"

#define BM2_WAVE 64
#define BM2_BLOCK_READS 512          // BATCH_SIZE of the reference (macro.h:48): the kt_for block (kthread.cpp:53-78)
#define BM2_H0 (-99)                 // H0_ (macro.h:44)

// CP_OCC (FMI_search.h:54-58): Occ checkpoint per 64 BWT symbols; exactly one 64-byte HBM line.  This is the layout of
// the index FILE and of bm2_index_desc; bm2_create re-lays every entry in HBM as CpOccDev.
struct __attribute__((aligned(64))) CpOcc {
    int64_t  cp_count[4];
    uint64_t bwt[4];                 // bit 63 = first symbol of the block (FMI_search.cpp:234-246)
};
// Device layout: the same 64 bytes, interleaved per base -- quarter b = { cp_count[b], bwt[b] } -- so that the four lanes
// of a quad fetch one entry with ONE coalesced 64-byte request, lane b getting exactly what the rank of base b needs.
struct __attribute__((aligned(64))) CpOccDev {
    struct { int64_t count; uint64_t bwt; } q[4];
};

struct DevIndex {
    const CpOccDev *cp_occ;
    const int8_t   *sa_ms_byte;
    const uint32_t *sa_ls_word;
    const uint8_t  *ref_string;      // .0123: forward then reverse complement; four bases per byte when ref_pk (refseq.h), else one
    const int64_t  *ann_offset;
    const int32_t  *ann_len;
    const int32_t  *ann_is_alt;
    int64_t ref_len, l_pac, sentinel_index;
    int64_t count[5];                // already +1 (FMI_search.cpp:433-436)
    int32_t n_seqs;
    int32_t ref_pk;
    __host__ __device__ RefPtr ref(int64_t pos) const { return RefPtr{ref_string, pos, ref_pk}; }
};

struct SwParams {                    // scoring for one extension side
    int32_t o_del, e_del, o_ins, e_ins, zdrop, end_bonus, max_sc;
    int32_t mat[25];
};

// One seed's alignment region while it is being built (fields of mem_alnreg_t, bwamem.h:137-160)
struct DevReg {
    int64_t rb, re;
    int32_t qb, qe, rid, score, truesc, w, seedcov, seedlen0;
    float   frac_rep;
    int32_t chain;                   // global index of the owning chain (the reference's a->c pointer)
};

struct DevSeed {                     // mem_seed_t (bwamem.h:113-124) without the unused fields
    int64_t rbeg;
    int32_t qbeg, len, score, aln;
};

struct DevChain {                    // mem_chain_t (bwamem.h:126-133); seeds are [seed_off, seed_off+n) in the seed array
    int64_t pos;
    int64_t seed_off;
    int32_t n, rid, w, kept, first, is_alt, read;
    float   frac_rep;
    int64_t rmax0, rmax1;            // reference window of the chain (bwamem.cpp:2145-2172)
    int32_t reg0, pad;               // index (within the read) of the chain's first reg; regs follow in extension order
};

// one banded-extension task (what a SeqPair + its seqBuf slices describe, bwamem.cpp:2229-2418)
struct DevTask {
    int64_t ref_pos;                 // first reference base; step -1 for a left extension, +1 for a right one
    int32_t q_pos;                   // first query base (offset inside the read); same step
    int32_t len1, len2;              // reference / query length
    int32_t h0;                      // left: seed_len * a; right: filled from the reg's score
};

static __device__ __forceinline__ int imax(int a, int b) { return a > b ? a : b; }
// max(a - b, 0) for a, b >= 0 in ONE VALU instruction (v_sub_u32 ... clamp) instead of a subtraction and a maximum
#if defined(__HIP_DEVICE_COMPILE__)
static __device__ __forceinline__ int isub0(int a, int b) { return (int)__builtin_elementwise_sub_sat((unsigned)a, (unsigned)b); }
#else
static __device__ __forceinline__ int isub0(int a, int b) { return a > b ? a - b : 0; }
#endif
static __device__ __forceinline__ int imin(int a, int b) { return a < b ? a : b; }
static __device__ __forceinline__ int64_t lmax(int64_t a, int64_t b) { return a > b ? a : b; }
static __device__ __forceinline__ int64_t lmin(int64_t a, int64_t b) { return a < b ? a : b; }
