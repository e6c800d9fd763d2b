// RefPtr::nib8 (bwa-mem2_amd/csrc/refseq.h) against element-by-element access: every start, both directions, both storages (tests/test_refseq.py).
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "refseq.h"
int main() {
    const int N = 1000;
    std::vector<uint8_t> codes(N), packed(N / 4 + 16, 0), bytes(N + 16);
    srand(5);
    for (int i = 0; i < N; i++) { codes[i] = (uint8_t)(rand() & 3); packed[i >> 2] |= (uint8_t)(codes[i] << ((i & 3) * 2)); }
    for (int i = 0; i < N; i++) bytes[i] = (uint8_t)(rand() % 5);                     // one code per byte: 0..4 (4 = ambiguous)
    long bad = 0, n = 0;
    for (int at = 0; at < 9; at++)
        for (int k = 0; k + 8 + at <= N; k++)
            for (int s = -1; s <= 1; s += 2) {
                if (s < 0 && k < 7) continue;
                const RefPtr P{packed.data(), at, 1}, B{bytes.data(), at, 0};
                const uint32_t x = P.nib8(k, s), y = B.nib8(k, s);
                for (int i = 0; i < 8; i++, n += 2) {
                    if (((x >> (4 * i)) & 15) != (uint32_t)P[k + s * i]) bad++;
                    if (((y >> (4 * i)) & 15) != (uint32_t)B[k + s * i]) bad++;
                }
            }
    // load4 on the same data (the extension kernels' loader: elements (k + i) * s of the view, i = 0..3 -- a view that walks down starts at its last position)
    for (int k = 0; k + 4 <= N - 8; k++)
        for (int s = -1; s <= 1; s += 2) {
            const int at = s > 0 ? 0 : N - 9;
            const RefPtr P{packed.data(), at, 1}, B{bytes.data(), at, 0};
            const uint32_t x = P.load4(k, s), y = B.load4(k, s);
            for (int i = 0; i < 4; i++, n += 2) {
                if (((x >> (8 * i)) & 255) != (uint32_t)P[(int64_t)(k + i) * s]) bad++;
                if (((y >> (8 * i)) & 255) != (uint32_t)B[(int64_t)(k + i) * s]) bad++;
            }
        }
    printf("checked %ld fields, %ld differ\n", n, bad);
    return bad != 0;
}
