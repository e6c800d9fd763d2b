#!/usr/bin/env python3
"""How a task-to-lane mapping of the banded extension uses the lanes of a wavefront -- a CPU model, no GPU.

The oracle (ORA_TRACE_ROWS) writes the band [beg, end) of every row of every ksw_extend call of a pe150-like run; the
extension tasks of the seeds whose regs survive the purge (= what the device's lazy rounds extend) are then packed into
wavefronts the way a kernel would pack them and the lane-slots of the DP inner loop are counted:

  lockstep   -- tasks of one side sorted by query length, 64 consecutive ones per wavefront, all lanes at the same row,
                columns [min beg, max end) in pairs (k_ext_lanes of rounds 1-3)
  seedstep   -- left then right of one seed in the same lane, seeds sorted by max(len) (and the other length), lockstep inside a side
  dynamic    -- persistent lanes: a lane takes the next seed when it is done (left, retry, right back to back, its own rows and
                its own band; an iteration lasts as long as the widest lane's row), new seeds fetched when >= K lanes wait

Usage: ext_sim.py <workdir> [n_pairs] [genome_mbp]"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import synth, refio, oracle          # noqa: E402
from tests import helpers                         # noqa: E402


def main():
    wd = sys.argv[1]
    n_pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
    mbp = int(sys.argv[3]) if len(sys.argv) > 3 else 16
    os.makedirs(wd, exist_ok=True)
    pre = os.path.join(wd, "g%d.fa" % mbp)
    seed = 20260924
    w = np.array([248, 242, 198, 190, 181, 171, 159, 145], dtype=np.float64)
    lens = [int(x) for x in (w / w.sum() * mbp * 1e6)]
    names, ctg, alts = synth.make_genome(seed, lens, n_repeat_families=max(8, min(mbp, 512)), repeat_len=(300, 6000), copies=(5, 200),
                                         divergence=(0.01, 0.15), n_gaps=8, gap_len=(100, 5000), alt_contigs=3, alt_len=50000)
    if not os.path.exists(pre + ".bwt.2bit.64"):
        synth.write_fasta(pre, names, ctg); synth.write_alt(pre + ".alt", alts)
        assert helpers.build_index(pre)
    r1, r2 = synth.make_reads_pe(seed + 1, ctg, n_pairs, L=150)
    seqs = np.empty((2 * len(r1), 150), np.uint8); seqs[0::2] = r1; seqs[1::2] = r2
    enc, off, ln = refio.pack_reads(list(seqs))
    trace = os.path.join(wd, "rows.bin")
    os.environ["ORA_TRACE_ROWS"] = trace
    ix = oracle.Index(pre)
    res = ix.run(enc, off, ln)
    ix.close()
    pairs, raw = res["PAIR"], res["REGRAW"]
    print("reads %d, tasks %d, regs raw %d kept %d, cells %d" % (len(ln), len(pairs), len(raw), len(res["REGPRG"]), res["counters"]["n_sw_cells"]))
    # --- the trace: calls in task order, two calls for a task whose w_used is 2w
    buf = np.fromfile(trace, np.int32)
    calls = []
    p = 0
    while p < len(buf):
        qlen, tlen, h0, wv, n = buf[p:p + 5]
        rows = buf[p + 5:p + 5 + n].view(np.int16).reshape(n, 2).astype(np.int32)
        calls.append((int(qlen), int(tlen), int(h0), int(wv), rows))
        p += 5 + n
    # regs of a read are numbered in REGRAW order: kept <=> qe > qb
    kept = {}
    cnt = {}
    for a in raw:
        r = int(a["read"]); i = cnt.get(r, 0); cnt[r] = i + 1
        kept[(r, i)] = a["qe"] > a["qb"]
    tasks = []                                   # (read, reg, side, [calls])
    ci = 0
    for pr in pairs:
        n_calls = 2 if pr["w_used"] > 100 else 1
        tasks.append((int(pr["read"]), int(pr["reg"]), int(pr["is_right"]), calls[ci:ci + n_calls]))
        ci += n_calls
    assert ci == len(calls), (ci, len(calls))
    tasks = [t for t in tasks if kept[(t[0], t[1])]]
    cells = sum(int(np.maximum(c[4][:, 1] - c[4][:, 0], 0).sum()) for t in tasks for c in t[3])
    print("tasks of kept regs: %d, their cells %d (%.0f per read), calls with a retry: %d" % (len(tasks), cells, cells / len(ln), sum(len(t[3]) > 1 for t in tasks)))

    def pairs_of(beg, end):                      # pair-iterations of one lane for a row
        return np.maximum((end + 1 >> 1) - (beg >> 1), 0) if False else max(((end + 1) >> 1) - (beg >> 1), 0) if end > beg else 0

    # ---- lockstep per side
    def lockstep(task_list):
        """task_list: tasks of one launch in lane order -> pair-iterations x 64 (lane slots / 2 cells)"""
        slots = 0
        for w0 in range(0, len(task_list), 64):
            wave = task_list[w0:w0 + 64]
            ntry = max(len(t[3]) for t in wave)
            for tr in range(ntry):
                rows = [t[3][tr][4] for t in wave if len(t[3]) > tr]
                nmax = max(len(r) for r in rows)
                lo = np.full(nmax, 1 << 20); hi = np.zeros(nmax, np.int64)
                for r in rows:
                    n = len(r)
                    lo[:n] = np.minimum(lo[:n], r[:, 0]); hi[:n] = np.maximum(hi[:n], r[:, 1])
                it = np.maximum(((hi + 1) >> 1) - (lo >> 1), 0)
                it[hi <= lo] = 0
                slots += int(it.sum()) * 64
        return slots

    # ---- instruction model of the lockstep lane kernel (counts from the gfx950 ISA of k_ext_lanes<side, P8, PF, PT>: ~180 VALU per row outside
    # the column loop, ~62 VALU per column pair, ~400 per task set-up), per query-length class
    print("class   tasks      cells   wave-rows  pair-iters  cells/row  model VALU per cell (x64 lanes)")
    for lo_q, hi_q in ((1, 16), (17, 32), (33, 48), (49, 64), (65, 80), (81, 96), (97, 112), (113, 160)):
        rows_w = its_w = cells_c = nt = 0
        for side in (0, 1):
            tl = sorted([t for t in tasks if t[2] == side and lo_q <= t[3][0][0] <= hi_q], key=lambda t: t[3][0][0])
            nt += len(tl)
            for w0 in range(0, len(tl), 64):
                wave = tl[w0:w0 + 64]
                rows = [t[3][0][4] for t in wave]
                nmax = max(len(r) for r in rows)
                lo = np.full(nmax, 1 << 20); hi = np.zeros(nmax, np.int64)
                for r in rows:
                    n = len(r)
                    lo[:n] = np.minimum(lo[:n], r[:, 0]); hi[:n] = np.maximum(hi[:n], r[:, 1])
                    cells_c += int(np.maximum(r[:, 1] - r[:, 0], 0).sum())
                it = np.maximum(((hi + 1) >> 1) - (lo >> 1), 0); it[hi <= lo] = 0
                rows_w += nmax; its_w += int(it.sum())
        if cells_c:
            print("%3d-%3d %7d %10d %10d %10d %9.1f  %.1f" % (lo_q, hi_q, nt, cells_c, rows_w, its_w, cells_c / max(rows_w, 1) / 64,
                                                        (rows_w * 180 + its_w * 62 + (nt / 64) * 400) * 64 / cells_c))
    for qmax_lane in (112, 160):
        tot = 0
        for side in (0, 1):
            tl = sorted([t for t in tasks if t[2] == side and t[3][0][0] <= qmax_lane], key=lambda t: t[3][0][0])
            tot += lockstep(tl)
        c_l = sum(int(np.maximum(c[4][:, 1] - c[4][:, 0], 0).sum()) for t in tasks if t[3][0][0] <= qmax_lane for c in t[3])
        print("lockstep by side (queries <= %d): %d pair-slots for %d cells -> lane utilisation %.3f" % (qmax_lane, tot, c_l, c_l / 2 / tot))

    # ---- seeds
    seeds = {}
    for t in tasks:
        seeds.setdefault((t[0], t[1]), [None, None])[t[2]] = t
    sl = list(seeds.values())

    def qlen_of(t):
        return t[3][0][0] if t else 0
    for key_name, key in (("max", lambda s: max(qlen_of(s[0]), qlen_of(s[1]))),
                          ("max,min/16", lambda s: (max(qlen_of(s[0]), qlen_of(s[1])), min(qlen_of(s[0]), qlen_of(s[1])) // 16)),
                          ("max/16,left", lambda s: ((max(qlen_of(s[0]), qlen_of(s[1])) + 15) // 16, qlen_of(s[0])))):
        order = sorted(sl, key=key)
        tot = 0
        for w0 in range(0, len(order), 64):
            wave = order[w0:w0 + 64]
            for side in (0, 1):
                tl = [s[side] for s in wave if s[side]]
                if tl:
                    # lanes keep their place: lockstep() only needs the set of tasks of the wave
                    tot += lockstep(tl) if len(tl) <= 64 else 0
        print("seedstep sorted by %-12s: %d pair-slots -> lane utilisation %.3f" % (key_name, tot, cells / 2 / tot))

    # ---- dynamic lanes
    def lane_rows(s):
        out = []
        for side in (0, 1):
            if s[side]:
                for c in s[side][3]:
                    r = c[4]
                    it = np.maximum(((r[:, 1] + 1) >> 1) - (r[:, 0] >> 1), 0); it[r[:, 1] <= r[:, 0]] = 0
                    out.append(it)
        return np.concatenate(out) if out else np.zeros(0, np.int64)
    for key_name, key, rev in (("max desc", lambda s: max(qlen_of(s[0]), qlen_of(s[1])), True),):
        order = sorted(sl, key=key, reverse=rev)
        work = [lane_rows(s) for s in order]
        for n_cls, K in ((1, 1), (1, 8), (1, 16), (8, 8)):
            # n_cls classes by position in the sorted list, each class its own waves; waves of a class take seeds from the class's cursor
            tot = 0; idle_wait = 0
            per = (len(work) + n_cls - 1) // n_cls
            for c0 in range(0, len(work), per):
                cw = work[c0:c0 + per]
                n_waves = max(1, len(cw) // (64 * 3))            # ~3 seeds per lane
                cur = 0
                # waves run independently; simulate each wave taking from the shared cursor round-robin by time: approximate with a
                # static interleave (wave k takes every n_waves-th block of 64 at start), then dynamic refills from the cursor
                lanes = [[None, 0] for _ in range(64 * n_waves)]  # (rows, pos)
                import heapq
                # event-free model: waves advance one iteration at a time in lock step with each other (same speed per iteration is a
                # simplification; what matters is the per-wave slot count)
                wave_slots = [0] * n_waves
                active = True
                while active:
                    active = False
                    for wv in range(n_waves):
                        L = lanes[wv * 64:(wv + 1) * 64]
                        waiting = [l for l in L if l[0] is None or l[1] >= len(l[0])]
                        running = 64 - len(waiting)
                        if cur < len(cw) and (len(waiting) >= K or running == 0):
                            for l in waiting:
                                if cur < len(cw):
                                    l[0] = cw[cur]; l[1] = 0; cur += 1
                        widths = [int(l[0][l[1]]) for l in L if l[0] is not None and l[1] < len(l[0])]
                        if not widths:
                            continue
                        active = True
                        wave_slots[wv] += max(max(widths), 1) * 64
                        for l in L:
                            if l[0] is not None and l[1] < len(l[0]):
                                l[1] += 1
                tot += sum(wave_slots)
            print("dynamic lanes (%d classes, refill at %2d waiting): %d pair-slots -> lane utilisation %.3f" % (n_cls, K, tot, cells / 2 / tot))


if __name__ == "__main__":
    main()
