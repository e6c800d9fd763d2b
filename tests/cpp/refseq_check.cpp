// refseq.h on the host: a packed sequence (four codes per byte) read through RefPtr -- element access, views, load4 in both directions, the
// lane kernel's 28-row 64-bit windows -- against the same codes as bytes.  Prints "ok".
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "refseq.h"

int main(int argc, char **argv) {
    const int64_t n = argc > 1 ? atoll(argv[1]) : 100003;
    const int PAD = 64;
    std::vector<uint8_t> bytes(n), store((n + 3) / 4 + 2 * PAD, 0);
    uint64_t x = 88172645463325252ull;
    for (int64_t i = 0; i < n; i++) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; bytes[i] = (uint8_t)(x & 3); }
    uint8_t *pk = store.data() + PAD;
    for (int64_t i = 0; i < n; i++) pk[i >> 2] |= (uint8_t)(bytes[i] << ((i & 3) * 2));
    const RefPtr B = RefPtr::bytes(bytes.data()), P{pk, 0, 1};
    for (int64_t i = 0; i < n; i++) if (B[i] != P[i]) { printf("element %ld\n", (long)i); return 1; }
    for (int64_t at = 0; at < n; at += 977) {
        const RefPtr b = B + at, p = P + at;
        for (int64_t k = 0; k < 40 && at + k < n; k++) if (b[k] != p[k] || (b - 3 + 3)[k] != (p + 5 - 5)[k]) { printf("view %ld+%ld\n", (long)at, (long)k); return 1; }
        for (int64_t k = 0; at + k + 3 < n && k < 64; k++) if (b.load4(k, 1) != p.load4(k, 1)) { printf("load4 fwd %ld+%ld\n", (long)at, (long)k); return 1; }
        for (int64_t k = 0; at - k - 3 >= 0 && k < 64; k++) {          // elements at, at-1, ...: walked backwards like a left extension
            if (b.load4(k, -1) != p.load4(k, -1)) { printf("load4 back %ld-%ld\n", (long)at, (long)k); return 1; }
            const uint32_t w = p.load4(k, -1);
            for (int u = 0; u < 4; u++) if ((int)((w >> (8 * u)) & 0xff) != B[at - k - u]) { printf("load4 order %ld\n", (long)at); return 1; }
        }
        // the windows of lane_dp8g (extend.hip): row i of a target that starts at `at` and is walked with stride ts
        for (int ts = -1; ts <= 1; ts += 2) {
            const int TW = 28;
            typedef uint64_t __attribute__((aligned(1))) u64u;
            const int64_t pa0 = ts > 0 ? p.at : p.at - (TW - 1);
            const int psh = (int)(pa0 & 3) << 1, pstep = ts > 0 ? TW / 4 : -(TW / 4);
            const uint8_t *pbyte = p.p + (pa0 >> 2);
            const int tlen = (int)(ts > 0 ? (n - at < 200 ? n - at : 200) : (at + 1 < 200 ? at + 1 : 200));
            for (int i = 0; i < tlen; i++) {
                const uint64_t wv = *(const u64u *)(pbyte + (int64_t)(i / TW) * pstep);
                const int pr = i % TW;
                const int tb = (int)(wv >> (psh + 2 * (ts > 0 ? pr : TW - 1 - pr))) & 3;
                if (tb != B[at + (int64_t)i * ts]) { printf("window ts=%d at=%ld row %d\n", ts, (long)at, i); return 1; }
            }
        }
    }
    printf("ok\n");
    return 0;
}
