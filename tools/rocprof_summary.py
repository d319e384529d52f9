#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (--kernel-trace) as a per-kernel table: calls, total / avg / min / max us, share.
usage: rocprof_summary.py results.db [> profiles/rNN_kernel_stats.txt]"""
import sqlite3
import sys


def main(path):
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3, max(vgpr_count), max(lds_size), "
                       "max(grid_x), max(grid_y), max(workgroup_x) from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows) or 1.0
    print("# rocprofv3 --kernel-trace summary of %s" % path)
    print("%-44s %6s %12s %10s %10s %10s %6s %5s %6s %s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "%", "vgpr", "lds", "grid x wg"))
    for r in rows:
        name = r[0].split("(")[0].replace("void ", "")
        print("%-44s %6d %12.1f %10.2f %10.2f %10.2f %6.2f %5d %6d %dx%d x %d" % (name[:44], r[1], r[2], r[3], r[4], r[5], 100 * r[2] / tot, r[6] or 0, r[7] or 0, r[8] or 0, r[9] or 0, r[10] or 0))


if __name__ == "__main__":
    main(sys.argv[1])
