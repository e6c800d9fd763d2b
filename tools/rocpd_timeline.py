#!/usr/bin/env python3
"""Per-dispatch timeline of a rocprofv3 (ROCm 7.2) rocpd sqlite result: one line per kernel dispatch in start order --
start and end in ms since the first dispatch, duration, queue / stream when the view has them, kernel name.
Usage: rocpd_timeline.py results.db out.tsv [first_row [n_rows]]"""
import sqlite3
import sys


def main():
    con = sqlite3.connect(sys.argv[1])
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    extra = [c for c in ("queue_id", "stream_id", "tid", "grid_x", "workgroup_x", "lds_size", "vgpr_count") if c in cols]
    rows = cur.execute("select name, start, end%s from kernels order by start" % "".join(", " + c for c in extra)).fetchall()
    first = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    n = int(sys.argv[4]) if len(sys.argv) > 4 else len(rows)
    t0 = rows[0][1] if rows else 0
    with open(sys.argv[2], "w") as f:
        f.write("\t".join(["start_ms", "end_ms", "dur_ms"] + extra + ["kernel"]) + "\n")
        for r in rows[first:first + n]:
            f.write("\t".join(["%.4f" % ((r[1] - t0) / 1e6), "%.4f" % ((r[2] - t0) / 1e6), "%.4f" % ((r[2] - r[1]) / 1e6)] +
                              [str(x) for x in r[3:]] + [r[0].split("(")[0][:70]]) + "\n")
    print("%d dispatches, columns of the view: %s" % (len(rows), ", ".join(cols)))


if __name__ == "__main__":
    main()
