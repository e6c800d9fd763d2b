#!/usr/bin/env python3
"""Summarise a rocprofv3 (ROCm 7.2) rocpd sqlite result: per-kernel count / total / avg duration, and PMC counter
sums per kernel when counters were collected.  Usage: rocpd_summary.py results.db [out.md]"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    con = sqlite3.connect(db)
    cur = con.cursor()
    lines = []
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    lines.append("# kernel summary of %s\n" % db)
    rows = cur.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                       "from kernels group by name order by sum(end-start) desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    lines.append("| kernel | calls | total ms | avg ms | min ms | max ms | % |")
    lines.append("|---|---|---|---|---|---|---|")
    for n, c, s, a, mi, ma in rows:
        lines.append("| %s | %d | %.3f | %.3f | %.3f | %.3f | %.1f |" % (n.split("(")[0][:60], c, s / 1e6, a / 1e6, mi / 1e6, ma / 1e6, 100.0 * s / tot))
    try:
        ccols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
        if ccols:
            prow = cur.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection "
                               "group by kernel_name, counter_name order by kernel_name").fetchall()
            if prow:
                lines.append("\n## PMC counters (sum over dispatches)\n")
                lines.append("| kernel | counter | sum | dispatches*dims |")
                lines.append("|---|---|---|---|")
                for k, cn, v, c in prow:
                    lines.append("| %s | %s | %.6g | %d |" % (k.split("(")[0][:50], cn, v, c))
            # The SECOND HALF of each kernel's dispatches on their own: a bench run of 2n steps whose first steps size their grids without the
            # previous batch's statistics (the extension stage's hints) or learn workspace sizes is not the steady state the bench line times.
            key = next((k for k in ("dispatch_id", "id", "start") if k in ccols), None)
            if prow and key:
                lines.append("\n## PMC counters, second half of each kernel's dispatches (by %s; columns: %s)\n" % (key, ", ".join(ccols)))
                lines.append("| kernel | counter | sum | dispatches*dims |")
                lines.append("|---|---|---|---|")
                for (k,) in cur.execute("select distinct kernel_name from counters_collection order by kernel_name").fetchall():
                    ids = sorted(r[0] for r in cur.execute("select distinct %s from counters_collection where kernel_name = ?" % key, (k,)))
                    if len(ids) < 2:
                        continue
                    cut = ids[len(ids) // 2]
                    for cn, v, c in cur.execute("select counter_name, sum(value), count(*) from counters_collection where kernel_name = ? and %s >= ? "
                                                "group by counter_name order by counter_name" % key, (k, cut)).fetchall():
                        lines.append("| %s | %s | %.6g | %d |" % (k.split("(")[0][:50], cn, v, c))
    except sqlite3.Error as e:
        lines.append("(no counters: %s)" % e)
    out = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out)
    print(out)


if __name__ == "__main__":
    main()
