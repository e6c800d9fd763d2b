// matesw_emu.cpp -- runs bwa-mem2_amd/csrc/matesw_dev.h (the device code of the mate-rescue SW) on the host through lane_emu.h.
//   matesw_emu <pairs.txt> <out.bin>      pairs.txt: "xtra QUERY TARGET" per line (ACGTN), as `refdump ksw` reads
//   scoring: env A B O_DEL E_DEL O_INS E_INS (defaults 1 4 6 1 6 1); EMU_KSW_REG=1: byte-kernel tasks of up to 160 bases go through ksw_row_task_reg
// out.bin: 7 int32 per task (score, te, qe, score2, te2, tb, qb).  Tasks run one after the other, each on 16 lane threads.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include "lane_emu.h"
#include "../../bwa-mem2_amd/csrc/matesw_dev.h"

struct Job {
    RowEmu row;
    const uint8_t *seqs; const KswTask *T; const KswPrm *prm; const int8_t *smat; uint16_t *L; int slen_max; unsigned long long *bl;
    bm2_ksw_result *out;
};
struct LaneArg { Job *job; int lane; };

static void *lane_main(void *p) {
    LaneArg *a = (LaneArg *)p;
    emu_row = &a->job->row; emu_lane = a->lane;
    Job *j = a->job;
    if (getenv("EMU_KSW_REG") && ksw_task_fits_regs(*j->T))       // the register pass (k_ksw_align2_reg) for the tasks it takes
        ksw_row_task_reg(j->seqs, RefPtr::bytes(j->seqs), *j->T, *j->prm, j->smat, a->lane, j->bl, j->out);
    else
    ksw_row_task(j->seqs, RefPtr::bytes(j->seqs), *j->T, *j->prm, j->smat, j->L, j->slen_max, a->lane, j->bl, j->out);
    return 0;
}

static int env_int(const char *n, int d) { const char *v = getenv(n); return v ? atoi(v) : d; }

int main(int argc, char **argv) {
    if (argc != 3) { fprintf(stderr, "usage: matesw_emu <pairs.txt> <out.bin>\n"); return 2; }
    const int a = env_int("A", 1), b = env_int("B", 4);
    KswPrm prm; memset(&prm, 0, sizeof prm);
    int lo = 127, hi = 0;
    for (int i = 0, k = 0; i < 5; ++i) for (int j = 0; j < 5; ++j, ++k) {                    // bwa_fill_scmat, bwa.cpp:248-257
        prm.mat[k] = (int8_t)((i == 4 || j == 4) ? -1 : (i == j ? a : -b));
        if (prm.mat[k] < lo) lo = prm.mat[k];
        if (prm.mat[k] > hi) hi = prm.mat[k];
    }
    prm.o_del = env_int("O_DEL", 6); prm.e_del = env_int("E_DEL", 1); prm.o_ins = env_int("O_INS", 6); prm.e_ins = env_int("E_INS", 1);
    prm.shift = (256 - (lo & 0xff)) & 0xff; prm.maxsc = hi;
    FILE *f = fopen(argv[1], "r");
    if (!f) { perror(argv[1]); return 1; }
    std::vector<uint8_t> seqs; std::vector<KswTask> tasks;
    static char q[1 << 20], t[1 << 20];
    int xtra;
    auto code = [](char c) { return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : 4; };
    char line[1 << 21];
    while (fgets(line, sizeof line, f)) {
        q[0] = t[0] = 0;
        const int got = sscanf(line, "%d %s %s", &xtra, q, t);          // an empty query / target leaves its field out
        if (got < 1) continue;
        KswTask T; memset(&T, 0, sizeof T);
        const char *qs = got >= 2 ? q : "", *ts = got >= 3 ? t : "";
        if (got == 2 && getenv("EMU_EMPTY_QUERY")) { ts = q; qs = ""; }
        T.q_off = (int64_t)seqs.size(); T.qlen = (int)strlen(qs); for (const char *p = qs; *p; ++p) seqs.push_back((uint8_t)code(*p));
        T.t_off = (int64_t)seqs.size(); T.tlen = (int)strlen(ts); for (const char *p = ts; *p; ++p) seqs.push_back((uint8_t)code(*p));
        T.xtra = xtra;
        tasks.push_back(T);
    }
    fclose(f);
    std::vector<bm2_ksw_result> out(tasks.size());
    for (size_t i = 0; i < tasks.size(); ++i) {
        const KswTask &T = tasks[i];
        const int P = (T.xtra & KSW_XBYTE) ? 16 : 8, slen_max = (T.qlen + P - 1) / P + 1;
        std::vector<uint16_t> L((size_t)9 * slen_max * 16 + 64, 0xdead);                      // poisoned: nothing may rely on zeroed LDS
        std::vector<unsigned long long> bl((size_t)(T.tlen + 1) / 2 + 1, ~0ull);
        Job job; pthread_barrier_init(&job.row.bar, 0, 16);
        job.seqs = seqs.data(); job.T = &T; job.prm = &prm; job.smat = prm.mat; job.L = L.data() + 32; job.slen_max = slen_max; job.bl = bl.data();
        job.out = &out[i];
        pthread_t th[16]; LaneArg la[16];
        for (int k = 0; k < 16; ++k) { la[k].job = &job; la[k].lane = k; pthread_create(&th[k], 0, lane_main, &la[k]); }
        for (int k = 0; k < 16; ++k) pthread_join(th[k], 0);
        pthread_barrier_destroy(&job.row.bar);
    }
    f = fopen(argv[2], "wb");
    if (!f) { perror(argv[2]); return 1; }
    fwrite(out.data(), sizeof(bm2_ksw_result), out.size(), f);
    fclose(f);
    return 0;
}
