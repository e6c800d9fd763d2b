// refseq.h -- a position in a sequence of base codes that is held either one code per byte (the caller's buffers at seams S1 / S4, the reads)
// or four codes per byte (the index's reference on the device: bm2_create packs the .0123 image of fastmap.cpp:873-881 -- forward strand then
// reverse complement, codes 0..3 only, bntseq.cpp:284 writes a random base for every ambiguous one -- from 2 * l_pac bytes to l_pac / 2:
// 6.2 -> 1.55 GB for a human genome, a quarter of the pages the extension / rescue / CIGAR kernels' scattered target reads touch).
// Plain C++: the device kernels, the host launchers and tools/emu all include it.  Kernels index it like the byte pointer it replaces.
#pragma once
#include <stdint.h>
#if defined(__HIPCC__)
#define BM2_REFSEQ_FN __host__ __device__ __forceinline__
#else
#define BM2_REFSEQ_FN inline
#endif

struct RefPtr {
    const uint8_t *p;       // first byte of the sequence's storage
    int64_t at;             // position of element 0 of this view, in codes
    int32_t pk;             // 1 = four codes per byte (code i in bits 2 * (i & 3) of byte i >> 2)
    BM2_REFSEQ_FN int operator[](int64_t k) const {
        const int64_t a = at + k;
        return pk ? (p[a >> 2] >> ((int)(a & 3) << 1)) & 3 : p[a];
    }
    BM2_REFSEQ_FN RefPtr operator+(int64_t k) const { return RefPtr{p, at + k, pk}; }
    BM2_REFSEQ_FN RefPtr operator-(int64_t k) const { return RefPtr{p, at - k, pk}; }
    // elements k, k + s, k + 2s, k + 3s (s = +1 or -1) as the four bytes of a word, the first in the lowest byte: one unaligned load
    // (the storage is padded by 8 bytes: the packed form reads one byte beyond the last code's)
    BM2_REFSEQ_FN uint32_t load4(int64_t k, int s) const {
        typedef uint32_t __attribute__((aligned(1))) u32u;
        typedef uint16_t __attribute__((aligned(1))) u16u;
        const int64_t a = at + (s > 0 ? k : -k - 3);            // the lowest of the four positions
        uint32_t w;
        if (pk) {
            const uint32_t v = ((uint32_t)*(const u16u *)(p + (a >> 2)) >> ((int)(a & 3) << 1)) & 0xffu;
            w = (v | v << 6 | v << 12 | v << 18) & 0x03030303u;
        } else w = *(const u32u *)(p + a);
        return s > 0 ? w : __builtin_bswap32(w);
    }
    // elements k, k + s, ..., k + 7s (s = +1 or -1) as the eight 4-bit fields of a word, the first in the lowest: ONE unaligned load instead of eight
    // byte loads (a lane that walks its own sequence byte by byte fetches every 64-byte line 64 times -- 16 times when packed -- and the lanes of a CU
    // together hold far more lines than its L1 does).  All eight elements must exist: the caller takes the tail of a sequence one by one.
    BM2_REFSEQ_FN uint32_t nib8(int64_t k, int s) const {
        typedef uint32_t __attribute__((aligned(1))) u32u;
        typedef uint64_t __attribute__((aligned(1))) u64u;
        const int64_t a = at + (s > 0 ? k : k - 7);             // the lowest of the eight positions
        uint32_t x;
        if (pk) {
            x = (*(const u32u *)(p + (a >> 2)) >> ((int)(a & 3) << 1)) & 0xffffu;      // (22 bits of the word at most)
            x = (x | x << 8) & 0x00ff00ffu; x = (x | x << 4) & 0x0f0f0f0fu; x = (x | x << 2) & 0x33333333u;
        } else x = nib8_of_bytes(*(const u64u *)(p + a));
        return s > 0 ? x : nib8_reverse(x);
    }
    static BM2_REFSEQ_FN uint32_t nib8_of_bytes(uint64_t w) {   // the low 4 bits of byte i -> field i
        w &= 0x0f0f0f0f0f0f0f0full;
        w = (w | w >> 4) & 0x00ff00ff00ff00ffull; w = (w | w >> 8) & 0x0000ffff0000ffffull;
        return (uint32_t)(w | w >> 16);
    }
    static BM2_REFSEQ_FN uint32_t nib8_reverse(uint32_t x) {     // field i <-> field 7 - i
        x = __builtin_bswap32(x);
        return (x & 0x0f0f0f0fu) << 4 | ((x >> 4) & 0x0f0f0f0fu);
    }
    static BM2_REFSEQ_FN RefPtr bytes(const uint8_t *q) { return RefPtr{q, 0, 0}; }
};
