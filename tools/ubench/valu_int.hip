// valu_int.hip -- micro-benchmark: the integer VALU issue rate of one MI355X for the instruction mix of the banded-SW kernels
// ("int16 VALU op/s ... measured on the build node").  Every lane runs ILP independent chains of one instruction
// kind, N instructions per chain and loop trip; the kinds are the ones a DP cell is made of: v_max_i32, v_add_u32 + v_max_i32 (a
// recurrence step), v_cndmask_b32, v_pk_max_i16 (two 16-bit values per lane) and v_max3_i32.  Occupancy (waves per SIMD) and ILP are swept.
// Prints wave-instructions/s over the whole GPU and the implied cycles per wave64 instruction per SIMD: the denominator of `extend_kernel.valu_frac`.
//   valu_int [trips]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

enum { K_MAX = 0, K_ADDMAX = 1, K_CNDMASK = 2, K_PKMAX = 3, K_MAX3 = 4 };
static const char *kind_name[] = { "v_max_i32", "v_add_u32+v_max_i32", "v_cndmask_b32", "v_pk_max_i16", "v_max3_i32" };

template <int KIND, int ILP>
__global__ void __launch_bounds__(256) k_valu(int trips, int seed, int *out) {
    int v[ILP];
#pragma unroll
    for (int c = 0; c < ILP; c++) v[c] = threadIdx.x * 7 + c * 13 + seed;
    int a = seed | 1, b = (seed >> 1) | 3;
    asm volatile("" : "+v"(a), "+v"(b));
    for (int t = 0; t < trips; t++) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
#pragma unroll
            for (int c = 0; c < ILP; c++) {
                if (KIND == K_MAX) asm volatile("v_max_i32 %0, %0, %1" : "+v"(v[c]) : "v"(a));
                else if (KIND == K_ADDMAX) asm volatile("v_add_u32 %0, %0, %1\n\tv_max_i32 %0, %0, %2" : "+v"(v[c]) : "v"(a), "v"(b));
                else if (KIND == K_CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[c]) : "v"(a));
                else if (KIND == K_PKMAX) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(v[c]) : "v"(a));
                else asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(v[c]) : "v"(a), "v"(b));
            }
        }
    }
    int s = 0;
#pragma unroll
    for (int c = 0; c < ILP; c++) s ^= v[c];
    if (s == 0x7fffabcd) out[0] = s;
}

template <int KIND, int ILP>
static void run(int n_cu, int waves_per_simd, int trips, int *d_out, double clock_ghz) {
    const int blocks = n_cu * waves_per_simd;                    // 256 threads = 4 waves = one per SIMD of a CU
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k_valu<KIND, ILP>), dim3(blocks), dim3(256), 0, 0, 4, 1, d_out);      // warm-up
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_valu<KIND, ILP>), dim3(blocks), dim3(256), 0, 0, trips, 1, d_out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double per_inst = KIND == K_ADDMAX ? 2.0 : 1.0;
    const double winst = (double)blocks * 4 * trips * 16.0 * ILP * per_inst;     // wave-instructions
    const double rate = winst / (ms * 1e-3);
    printf("%-22s ilp %d  waves/SIMD %d : %8.3f ms  %8.1f G wave-instr/s  = %.2f cycles per wave64 instruction per SIMD at %.2f GHz\n", kind_name[KIND], ILP,
           waves_per_simd, ms, rate / 1e9, (double)n_cu * 4 * clock_ghz * 1e9 / rate, clock_ghz);
    hipEventDestroy(e0); hipEventDestroy(e1);
}

int main(int argc, char **argv) {
    const int trips = argc > 1 ? atoi(argv[1]) : 20000;
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    const int n_cu = pr.multiProcessorCount;
    const double ghz = pr.clockRate / 1e6;
    printf("# %s: %d CUs, %.2f GHz\n", pr.name, n_cu, ghz);
    int *d_out; hipMalloc(&d_out, 64);
    for (int w : { 1, 2, 4, 8 }) {
        run<K_MAX, 1>(n_cu, w, trips, d_out, ghz); run<K_MAX, 4>(n_cu, w, trips, d_out, ghz);
    }
    for (int w : { 2, 4 }) {
        run<K_ADDMAX, 1>(n_cu, w, trips, d_out, ghz); run<K_ADDMAX, 4>(n_cu, w, trips, d_out, ghz);
        run<K_CNDMASK, 4>(n_cu, w, trips, d_out, ghz);
        run<K_PKMAX, 1>(n_cu, w, trips, d_out, ghz); run<K_PKMAX, 4>(n_cu, w, trips, d_out, ghz);
        run<K_MAX3, 4>(n_cu, w, trips, d_out, ghz);
    }
    hipFree(d_out);
    return 0;
}
