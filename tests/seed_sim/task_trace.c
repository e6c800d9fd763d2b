/* tests/seed_sim/task_trace.c -- analysis tool (TEST INFRASTRUCTURE like everything that touches oracle/, not product): the work items of the seeding kernels as the
 * device cuts them (smem.hip: one forward walk per start position, one backward task per candidate list), with their sizes, from the
 * oracle's restatement of getSMEMsOnePosOneThread.  Output: one line per task "pass rid x n_prev fwd_ext bwd_ext rows first_blk [candidates per row ...]".
 *   gcc -O2 -o /tmp/task_trace tests/seed_sim/task_trace.c -lm && /tmp/task_trace <index prefix> <reads.bin> <n_reads> <read_len> > tasks.txt */
#include "../../oracle/bm2_oracle.c"

typedef struct { int fwd, n_prev, bwd, rows; int64_t blk; int rowc[512]; } trace_t;   /* rowc[i] = candidates extended in the i-th backward row */

static int traced_one_pos(const ora_index *ix, const uint8_t *q, int len, uint32_t rid, int x, int64_t min_intv,
                          int min_seed_len, smem_v *out, smem_t *prev, trace_t *tr) {
    fm_stat st; memset(&st, 0, sizeof st);
    int next_x = x + 1;
    int a = q[x];
    tr->fwd = tr->n_prev = tr->bwd = tr->rows = 0; tr->blk = -1;
    if (a >= 4) return next_x;
    smem_t sm; sm.rid = rid; sm.m = (uint32_t)x; sm.n = (uint32_t)x;
    sm.iv.k = ix->count[a]; sm.iv.l = ix->count[3 - a]; sm.iv.s = ix->count[a + 1] - ix->count[a];
    int n_prev = 0, j;
    for (j = x + 1; j < len; j++) {
        a = q[j];
        next_x = j + 1;
        if (a >= 4) break;
        smem_t ns = sm;
        ns.iv = forward_ext(ix, sm.iv, a, &st);
        ns.n = (uint32_t)j;
        prev[n_prev] = sm;
        n_prev += (ns.iv.s != sm.iv.s);
        if (ns.iv.s < min_intv) { next_x = j; break; }
        sm = ns;
    }
    tr->fwd = (int)st.n_ext; st.n_ext = 0;
    if (sm.iv.s >= min_intv) prev[n_prev++] = sm;
    tr->n_prev = n_prev;
    for (int p = 0; p < n_prev / 2; p++) { smem_t t = prev[p]; prev[p] = prev[n_prev - 1 - p]; prev[n_prev - 1 - p] = t; }
    if (n_prev) tr->blk = prev[0].iv.k >> 6;
    for (j = x - 1; j >= 0; j--) {
        int n_curr = 0, p;
        int32_t curr_s = -1;
        a = q[j];
        if (a > 3) break;
        if (tr->rows < 512) tr->rowc[tr->rows] = n_prev;
        tr->rows++;
        for (p = 0; p < n_prev; p++) {
            smem_t s0 = prev[p], ns = s0;
            ns.iv = backward_ext(ix, s0.iv, a, &st);
            ns.m = (uint32_t)j;
            if (ns.iv.s < min_intv && (int)(s0.n - s0.m + 1) >= min_seed_len) { vec_push(smem_t, *out, s0); break; }
            if (ns.iv.s >= min_intv && ns.iv.s != (int64_t)curr_s) { curr_s = (int32_t)ns.iv.s; prev[n_curr++] = ns; break; }
        }
        p++;
        for (; p < n_prev; p++) {
            smem_t s0 = prev[p], ns = s0;
            ns.iv = backward_ext(ix, s0.iv, a, &st);
            ns.m = (uint32_t)j;
            if (ns.iv.s >= min_intv && ns.iv.s != (int64_t)curr_s) { curr_s = (int32_t)ns.iv.s; prev[n_curr++] = ns; }
        }
        n_prev = n_curr;
        if (n_curr == 0) break;
    }
    if (n_prev != 0) { smem_t s0 = prev[0]; if ((int)(s0.n - s0.m + 1) >= min_seed_len) vec_push(smem_t, *out, s0); }
    tr->bwd = (int)st.n_ext;
    return next_x;
}

int main(int argc, char **argv) {
    if (argc < 5) { fprintf(stderr, "usage: task_trace <prefix> <reads.bin> <n_reads> <read_len>\n"); return 2; }
    ora_index *ix = ora_index_load(argv[1]);
    if (!ix) { fprintf(stderr, "no index\n"); return 1; }
    const int n = atoi(argv[3]), L = atoi(argv[4]);
    const int rows_too = argc > 5;        /* a fifth argument: the per-row candidate counts of the long tasks (> 200 extensions) behind the fixed fields */
    uint8_t *enc = (uint8_t *)malloc((size_t)n * L);
    FILE *f = fopen(argv[2], "rb");
    if (!f || fread(enc, 1, (size_t)n * L, f) != (size_t)n * L) { fprintf(stderr, "reads?\n"); return 1; }
    fclose(f);
    ora_opt opt; ora_opt_init(&opt);
    const int split_len = (int)(opt.min_seed_len * opt.split_factor + .499);
    smem_t *prev = (smem_t *)malloc((size_t)(L + 1) * sizeof(smem_t));
    smem_v out; memset(&out, 0, sizeof out);
    trace_t tr;
    for (int r = 0; r < n; r++) {
        int x = 0;
        const int64_t n0 = out.n;
        while (x < L) {
            const int x0 = x;
            x = traced_one_pos(ix, enc + (size_t)r * L, L, (uint32_t)r, x, 1, opt.min_seed_len, &out, prev, &tr);
            printf("1 %d %d %d %d %d %d %lld", r, x0, tr.n_prev, tr.fwd, tr.bwd, tr.rows, (long long)tr.blk);
            if (rows_too && tr.bwd > 200) for (int i = 0; i < tr.rows && i < 512; i++) printf(" %d", tr.rowc[i]);
            printf("\n");
        }
        const int64_t n1 = out.n;
        for (int64_t i = n0; i < n1; i++) {
            smem_t p = out.a[i];
            int start = (int)p.m, end = (int)p.n + 1;
            if (end - start < split_len || p.iv.s > opt.split_width) continue;
            const int x0 = (end + start) >> 1;
            traced_one_pos(ix, enc + (size_t)r * L, L, (uint32_t)r, x0, p.iv.s + 1, opt.min_seed_len, &out, prev, &tr);
            printf("2 %d %d %d %d %d %d %lld", r, x0, tr.n_prev, tr.fwd, tr.bwd, tr.rows, (long long)tr.blk);
            if (rows_too && tr.bwd > 200) for (int i = 0; i < tr.rows && i < 512; i++) printf(" %d", tr.rowc[i]);
            printf("\n");
        }
        out.n = 0;
    }
    return 0;
}
