#!/usr/bin/env python3
"""Start / end of every kernel of the LAST encode step and the last decode of a rocprofv3 --kernel-trace database, by stream: which kernels ran side by side and
what that did to their durations.  usage: timeline.py <results.db> [min_us=20]"""
import sqlite3
import sys


def main():
    cur = sqlite3.connect(sys.argv[1]).cursor()
    min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
    rows = list(cur.execute("select name, start, end, stream_id from kernels order by start"))
    big = [i for i, r in enumerate(rows) if "k_gather2" in r[0] and r[2] - r[1] > 1e6] or [i for i, r in enumerate(rows) if "k_gather" in r[0]]
    emit = [i for i, r in enumerate(rows) if "k_dec_emit" in r[0]]
    for title, i0, n in (("encode: from the gather on", big[-1] if big else 0, 45), ("decode: up to the emitter", max(0, (emit[-1] if emit else 0) - 36), 38)):
        t0 = rows[i0][1]
        print("# " + title)
        print("%-40s %10s %10s %10s  stream" % ("kernel", "start us", "end us", "dur us"))
        for r in rows[i0:i0 + n]:
            if (r[2] - r[1]) / 1e3 >= min_us or "coder" in r[0]:
                print("%-40s %10.1f %10.1f %10.1f  %s" % (r[0].replace("void ", "")[:40], (r[1] - t0) / 1e3, (r[2] - t0) / 1e3, (r[2] - r[1]) / 1e3, r[3]))
        print()


if __name__ == "__main__":
    main()
