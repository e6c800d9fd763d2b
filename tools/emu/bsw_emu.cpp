// bsw_emu.cpp -- bwa-mem2_amd/csrc/bsw.hip (k_bsw_pairs + its launcher, with bsw_dev.h's wave-per-task DP) on the host emulator.
//   bsw_emu <pairs.txt> <out.bin>   pairs.txt: "h0 QUERY TARGET" per line; env: W (band), A B O_DEL E_DEL O_INS E_INS ZDROP END_BONUS
// out.bin: 6 int32 per pair (score, qle, tle, gtle, gscore, max_off).  Build: see tests/test_device_sources_on_host.py.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <string>
#include BSW_SRC                      /* bsw.hip as rewritten by build_emu.rewrite() */

void bm2_set_error(const char *fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
int bm2_check(hipError_t e, const char *) { return e == hipSuccess ? BM2_OK : BM2_ENODEV; }
// (the sorted S1 path lives in extend.hip, which this single-source harness does not link: every pair goes to the pair-per-wavefront kernel)
int bm2_launch_bsw_sorted(bm2_ctx *, bm2_seqpair_t *, const uint8_t *, const uint8_t *, int, int, const SwParams &, unsigned long long *, bool *done) { *done = false; return BM2_OK; }

static int env_int(const char *n, int d) { const char *v = getenv(n); return v ? atoi(v) : d; }

int main(int argc, char **argv) {
    if (argc != 3) { fprintf(stderr, "usage: bsw_emu <pairs.txt> <out.bin>\n"); return 2; }
    const int a = env_int("A", 1), b = env_int("B", 4);
    SwParams P; memset(&P, 0, sizeof P);
    for (int i = 0, k = 0; i < 5; ++i) for (int j = 0; j < 5; ++j, ++k) P.mat[k] = (i == 4 || j == 4) ? -1 : (i == j ? a : -b);
    P.o_del = env_int("O_DEL", 6); P.e_del = env_int("E_DEL", 1); P.o_ins = env_int("O_INS", 6); P.e_ins = env_int("E_INS", 1);
    P.zdrop = env_int("ZDROP", 100); P.end_bonus = env_int("END_BONUS", 5); P.max_sc = a;
    const int w = env_int("W", 100);
    FILE *f = fopen(argv[1], "r");
    if (!f) { perror(argv[1]); return 1; }
    std::vector<uint8_t> ref, qer; std::vector<bm2_seqpair_t> pairs;
    static char q[1 << 20], t[1 << 20], line[1 << 21];
    auto code = [](char c) { return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : 4; };
    int h0;
    while (fgets(line, sizeof line, f)) {
        q[0] = t[0] = 0;
        if (sscanf(line, "%d %s %s", &h0, q, t) < 1) continue;
        bm2_seqpair_t p; memset(&p, 0, sizeof p);
        p.idq = (int)qer.size(); p.len2 = (int)strlen(q); for (const char *c = q; *c; ++c) qer.push_back((uint8_t)code(*c));
        p.idr = (int)ref.size(); p.len1 = (int)strlen(t); for (const char *c = t; *c; ++c) ref.push_back((uint8_t)code(*c));
        p.h0 = h0;
        pairs.push_back(p);
    }
    fclose(f);
    ref.resize(ref.size() + 64); qer.resize(qer.size() + 64);
    bm2_ctx c;
    const int rc = bm2_launch_bsw_pairs(&c, pairs.data(), ref.data(), qer.data(), (int)pairs.size(), w, P, nullptr);
    if (rc) { fprintf(stderr, "launch failed: %d\n", rc); return 1; }
    f = fopen(argv[2], "wb");
    for (const bm2_seqpair_t &p : pairs) { const int32_t o[6] = { p.score, p.qle, p.tle, p.gtle, p.gscore, p.max_off }; fwrite(o, 4, 6, f); }
    fclose(f);
    return 0;
}
