#!/usr/bin/env python3
"""tests/seed_sim/sim_export.py -- analysis tool (not product): a lane-level model of k_bwd's persistent lanes (smem.hip) on the task
sizes tests/seed_sim/task_trace.c writes, with and without the hand-over of old tasks to the wavefront-per-task kernel at a row
boundary (BM2_BWD_EXPORT_AGE).  One "round" = one converged backwardExt of a wavefront; all rounds cost the same here, which the GPU's
do not (a wavefront alone on its SIMD goes round faster), so the model overstates the tail in time and is right about its cause.

    python tests/seed_sim/sim_export.py tasks_rows.txt [scale]      (scale = 1 M reads / reads traced: the lanes are cut by it)
"""
import heapq
import sys

import numpy as np


def load(fn):
    fixed, rowc = [], []
    for ln in open(fn):
        t = ln.split()
        fixed.append([int(v) for v in t[:8]])
        rowc.append(np.array(t[8:], np.int32) if len(t) > 8 else None)
    return np.array(fixed, np.int64), rowc


def sim(L, n_waves, batch=64):
    n = len(L)
    lanes = n_waves * 64
    h = [(0.0, i) for i in range(lanes)]
    pool_pos = np.zeros(n_waves, np.int64); pool_end = np.zeros(n_waves, np.int64)
    cur = 0; busy = 0.0
    end_lane = np.zeros(lanes)
    while h:
        t, l = heapq.heappop(h)
        w = l // 64
        if pool_pos[w] >= pool_end[w]:
            if cur >= n:
                end_lane[l] = t; continue
            pool_pos[w] = cur; pool_end[w] = min(cur + batch, n); cur += batch
        i = pool_pos[w]; pool_pos[w] += 1
        busy += L[i]
        heapq.heappush(h, (t + L[i] + 2, l))
    wave_end = end_lane.reshape(n_waves, 64).max(1)
    T = wave_end.max()
    return T, busy / lanes, busy / (wave_end.sum() * 64), wave_end.sum() / (T * n_waves)


def main():
    fixed, rowc = load(sys.argv[1])
    scale = float(sys.argv[2]) if len(sys.argv) > 2 else 5.0
    p, rid, x, npv, fwd, bwd, rows, blk = fixed.T
    nw = int(3072 / scale) // 4 * 4
    for ps in (1, 2):
        idx = np.nonzero((p == ps) & (npv > 0) & (npv <= 40))[0]
        print("pass %d: %d lane tasks, %d wavefronts of 64 lanes" % (ps, len(idx), nw))
        for age in (0, 1024, 512, 384, 256, 192, 128):
            L = bwd[idx].copy()
            n_exp = 0; exp_rows = []; exp_ext = 0
            if age:
                for k, i in enumerate(idx):
                    rc = rowc[i]
                    if rc is None or bwd[i] <= age:
                        continue
                    cs = np.cumsum(rc)
                    r = int(np.searchsorted(cs, age, side="right"))      # rows whose end is reached with at most `age` extensions ... the first row boundary past the age
                    r = min(r + 1, len(rc))
                    if r >= len(rc):
                        continue                                          # the boundary is the task's end
                    L[k] = cs[r - 1]
                    n_exp += 1; exp_rows.append(len(rc) - r); exp_ext += int(cs[-1] - cs[r - 1])
            T, ideal, util, occ = sim(L, nw)
            er = np.array(exp_rows) if exp_rows else np.zeros(1)
            # the continuation kernel: one wavefront per task, one round per row; 8192 wavefronts resident at full scale
            cont_waves = int(8192 / scale)
            cont_T = max(er.max(), er.sum() / cont_waves) if n_exp else 0
            print("  age %4d: lane kernel %5.0f rounds (ideal %5.0f, lane use %.2f, wave occupancy %.2f) | handed over %6d tasks (%.2f%%), %4.1f%% of the extensions, rows: mean %.0f max %d -> continuation kernel ~%4.0f rounds | together %5.0f"
                  % (age, T, ideal, util, occ, n_exp, 100.0 * n_exp / len(idx), 100.0 * exp_ext / bwd[idx].sum(), er.mean(), er.max(), cont_T, T + cont_T))


if __name__ == "__main__":
    main()
