// harness.cpp -- the host tail (bm2_sam_pe with the flat batch hooks = the code path bm2_sam_pe_dev drives, host alignments standing in
// for the device batches) on inputs dumped by dump_inputs.py: phase clock (BM2_TAIL_PROF=1) and gprof without Python or a GPU.
//   g++ -O2 -g -pg -std=c++17 -I../../include harness.cpp ../../bwa-mem2_amd/csrc/{sam_tail,index_io,fastq_io}.cpp -lpthread -o /tmp/tail_harness
//   BM2_RESCUE_FLAT=1 BM2_CIGAR_FLAT=1 /tmp/tail_harness in.bin <threads> <repeats>;  gprof /tmp/tail_harness gmon.out
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
#include <string>
#include <vector>
#include "bm2.h"

// (the parser's chunk arrays come from the library's page-locked pool, bm2_api.hip: plain memory here)
void *bm2_chunk_mem_get(size_t bytes) { return malloc(bytes); }
void bm2_chunk_mem_put(void *p) { free(p); }
void bm2_set_error(const char *fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
extern "C" void bm2_opt_fill_scmat(bm2_opt *o) {
    int k = 0;
    for (int i = 0; i < 4; ++i) { for (int j = 0; j < 4; ++j) o->mat[k++] = (int8_t)(i == j ? o->a : -o->b); o->mat[k++] = -1; }
    for (int j = 0; j < 5; ++j) o->mat[k++] = -1;
}

int main(int argc, char **argv) {
    if (argc < 2) { fprintf(stderr, "usage: harness in.bin [threads] [repeats]\n"); return 2; }
    const int threads = argc > 2 ? atoi(argv[2]) : 1, reps = argc > 3 ? atoi(argv[3]) : 1;
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 1; }
    int64_t hd[3];
    if (fread(hd, 8, 3, f) != 3) return 1;
    const int64_t n = hd[0], nb = hd[1], nh = hd[2];
    std::vector<uint8_t> enc((size_t)nb); std::vector<int64_t> off((size_t)n), aoff((size_t)n + 1); std::vector<int32_t> len((size_t)n);
    std::vector<bm2_alnreg_t> aln((size_t)nh);
    if (fread(enc.data(), 1, (size_t)nb, f) != (size_t)nb || fread(off.data(), 8, (size_t)n, f) != (size_t)n || fread(len.data(), 4, (size_t)n, f) != (size_t)n ||
        fread(aln.data(), sizeof(bm2_alnreg_t), (size_t)nh, f) != (size_t)nh || fread(aoff.data(), 8, (size_t)n + 1, f) != (size_t)n + 1) { fprintf(stderr, "short file\n"); return 1; }
    fclose(f);
    char prefix[4096];
    { FILE *p = fopen((std::string(argv[1]) + ".prefix").c_str(), "r"); if (!p || !fgets(prefix, sizeof prefix, p)) return 1; fclose(p); prefix[strcspn(prefix, "\n")] = 0; }
    bm2_index_desc idx;
    if (bm2_index_load(prefix, &idx)) return 1;
    std::vector<std::string> names((size_t)n), quals((size_t)n);
    std::vector<const char *> np((size_t)n), qp((size_t)n);
    for (int64_t i = 0; i < n; ++i) { names[(size_t)i] = "p" + std::to_string(i / 2); quals[(size_t)i].assign((size_t)len[(size_t)i], 'F'); np[(size_t)i] = names[(size_t)i].c_str(); qp[(size_t)i] = quals[(size_t)i].c_str(); }
    bm2_reads R = { (int32_t)n, enc.data(), off.data(), len.data() };
    bm2_read_text T; memset(&T, 0, sizeof T); T.name = np.data(); T.qual = qp.data();
    bm2_opt opt; memset(&opt, 0, sizeof opt);                    // mem_opt_init defaults (bm2_opt_init lives in the HIP part of the library)
    opt.a = 1; opt.b = 4; opt.o_del = opt.o_ins = 6; opt.e_del = opt.e_ins = 1; opt.w = 100; opt.zdrop = 100; opt.pen_clip5 = opt.pen_clip3 = 5;
    opt.max_mem_intv = 20; opt.min_seed_len = 19; opt.split_width = 10; opt.max_occ = 500; opt.max_chain_gap = 10000; opt.mask_level = 0.50f;
    opt.drop_ratio = 0.50f; opt.split_factor = 1.5f; opt.mask_level_redun = 0.95f; opt.min_chain_weight = 0; opt.max_chain_extend = 1 << 30;
    bm2_opt_fill_scmat(&opt);
    bm2_sam_opt so; bm2_sam_opt_init(&so); so.n_threads = threads;
    std::vector<char> out((size_t)(3 * (nb + 200 * n)));
    for (int r = 0; r < reps; ++r) {
        int64_t n_out = 0;
        const auto t0 = std::chrono::steady_clock::now();
        const int rc = bm2_sam_pe(&idx, &opt, &so, &R, &T, aln.data(), aoff.data(), 0, nullptr, nullptr, out.data(), (int64_t)out.size(), &n_out);
        const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        unsigned long long h = 1469598103934665603ULL;
        for (int64_t i = 0; i < n_out; ++i) { h ^= (unsigned char)out[(size_t)i]; h *= 1099511628211ULL; }
        fprintf(stderr, "rc=%d  %lld reads in %.3f s = %.0f reads/s on %d threads; %lld SAM bytes, fnv %016llx\n", rc, (long long)n, s, n / s, threads, (long long)n_out, h);
    }
    bm2_index_free(&idx);
    return 0;
}
