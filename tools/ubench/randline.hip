// randline.hip -- micro-benchmark: how many independent random 64-byte lines per second can one MI355X fetch?
// Each lane walks a dependent chain: the next line index is derived from the data just loaded (so nothing can be
// prefetched), `LPL` lines are requested per lane and step, each line is fetched as four 16-byte loads exactly as the
// CP_OCC blocks are in smem.hip.  Footprint and occupancy are swept from the command line:
//   randline <footprint_MB> <blocks_per_cu> <steps> <lines_per_lane(1|2)>
// Prints lines/s and GB/s.  This is the practical roofline of the FM-index seeding kernels.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

template <int LPL>
__global__ void __launch_bounds__(256) k_rand(const ulonglong2 *__restrict__ buf, uint64_t n_lines, int steps, uint64_t *out) {
    uint64_t x = (uint64_t)(blockIdx.x * blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ULL + 12345;
    uint64_t acc = 0;
    for (int s = 0; s < steps; s++) {
        uint64_t sum = 0;
#pragma unroll
        for (int t = 0; t < LPL; t++) {
            const uint64_t line = (x + (uint64_t)t * 0xD1B54A32D192ED03ULL) % n_lines;
            const ulonglong2 *p = buf + line * 4;
            const ulonglong2 a = p[0], b = p[1], c = p[2], d = p[3];
            sum += a.x ^ a.y ^ b.x ^ b.y ^ c.x ^ c.y ^ d.x ^ d.y;
        }
        x = x * 6364136223846793005ULL + sum + 1442695040888963407ULL;      // depends on the loaded data
        x ^= x >> 29;
        acc += sum;
    }
    if (acc == 0x1234567) out[0] = x;
}

// quad-cooperative variant: the four lanes of a quad fetch the four 16-byte quarters of ONE line with one instruction
// (one coalesced 64-byte request); each lane still consumes LPL lines of its own per step (4 * LPL instructions)
template <int LPL, bool NT = false>       // NT: the same with non-temporal loads (`global_load ... nt`: a line nobody will touch again)
__global__ void __launch_bounds__(256) k_rand_quad(const ulonglong2 *__restrict__ buf, uint64_t n_lines, int steps, uint64_t *out) {
    uint64_t x = (uint64_t)(blockIdx.x * blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ULL + 12345;
    uint64_t acc = 0;
    const int lane = threadIdx.x & 63, qb = lane & ~3, sub = lane & 3;
    for (int s = 0; s < steps; s++) {
        uint64_t sum = 0;
#pragma unroll
        for (int t = 0; t < LPL; t++) {
            const uint64_t mine = (x + (uint64_t)t * 0xD1B54A32D192ED03ULL) % n_lines;
            uint64_t part[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint64_t line = __shfl(mine, qb + u);
                ulonglong2 a;
                if (NT) { const uint64_t *q = (const uint64_t *)&buf[line * 4 + sub]; a.x = __builtin_nontemporal_load(q); a.y = __builtin_nontemporal_load(q + 1); }
                else a = buf[line * 4 + sub];
                part[u] = a.x ^ a.y;
            }
            // every lane needs the xor over the four quarters of ITS line: quad all-to-all
#pragma unroll
            for (int u = 0; u < 4; u++) {
                uint64_t v = part[u];
                v ^= __shfl_xor(v, 1); v ^= __shfl_xor(v, 2);
                if (u == sub) sum += v;
            }
        }
        x = x * 6364136223846793005ULL + sum + 1442695040888963407ULL;
        x ^= x >> 29;
        acc += sum;
    }
    if (acc == 0x1234567) out[0] = x;
}

int main(int argc, char **argv) {
    const size_t mb = argc > 1 ? atol(argv[1]) : 1024;
    const int bpc = argc > 2 ? atoi(argv[2]) : 8;
    const int steps = argc > 3 ? atoi(argv[3]) : 2000;
    const int lpl = argc > 4 ? atoi(argv[4]) : 2;
    const int quad = argc > 5 ? atoi(argv[5]) : 0;
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    const size_t bytes = mb << 20; const uint64_t n_lines = bytes / 64;
    void *buf; uint64_t *out;
    if (hipMalloc(&buf, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMalloc(&out, 8);
    std::vector<uint64_t> h(1 << 20);
    for (size_t i = 0; i < h.size(); i++) h[i] = i * 0x9E3779B97F4A7C15ULL;
    for (size_t o = 0; o < bytes; o += h.size() * 8) hipMemcpy((char *)buf + o, h.data(), (bytes - o < h.size() * 8) ? bytes - o : h.size() * 8, hipMemcpyHostToDevice);
    const int grid = pr.multiProcessorCount * bpc;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        if (quad == 2) hipLaunchKernelGGL((k_rand_quad<2, true>), dim3(grid), dim3(256), 0, 0, (const ulonglong2 *)buf, n_lines, steps, out);
        else if (quad) hipLaunchKernelGGL(k_rand_quad<2>, dim3(grid), dim3(256), 0, 0, (const ulonglong2 *)buf, n_lines, steps, out);
        else if (lpl == 1) hipLaunchKernelGGL(k_rand<1>, dim3(grid), dim3(256), 0, 0, (const ulonglong2 *)buf, n_lines, steps, out);
        else hipLaunchKernelGGL(k_rand<2>, dim3(grid), dim3(256), 0, 0, (const ulonglong2 *)buf, n_lines, steps, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double lines = (double)grid * 256 * steps * lpl;
        if (rep) printf("%s footprint %6zu MB  blocks/CU %2d  lines/lane %d : %7.2f ms  %6.1f G lines/s  %7.1f GB/s  (%.2f us per step)\n", quad == 2 ? "quad-nt" : quad ? "quad" : "lane", mb, bpc, lpl, ms,
                        lines / ms / 1e6, lines * 64 / ms / 1e6, ms * 1e3 / steps);
    }
    return 0;
}
