#!/usr/bin/env python3
"""HBM traffic per kernel from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected in SEPARATE runs, --kernel-trace only).
Units / gfx950 correction per /opt/skills/guides/MI355X_MICROARCH.md §HBM: counters are in KB; FETCH_SIZE under-counts wide coalesced
streams by exactly 2x (calibrated here on k_nl_bitmap, which reads the whole FASTQ once: 2 x FETCH_SIZE x 1024 == input bytes).
usage: pmc_summary.py fetch.db write.db [out.json]"""
import json
import sqlite3
import sys


def per_kernel(path, counter, last_step=False):
    """{kernel: (dispatches, sum of the counter)}.  last_step: only the dispatches of the LAST bench step (from its first line-index kernel on) - a warm-up step may hold
    work a timed step never does (round 6: the repeat of a batch whose arenas a fresh context sized too small)."""
    cur = sqlite3.connect(path).cursor()
    if last_step:
        allr = cur.execute("select name, start, counter_value from pmc_events where counter_name=? order by start", (counter,)).fetchall()
        idx = [i for i, r in enumerate(allr) if "k_line_index" in r[0] or "k_nl_bitmap" in r[0]]
        starts = [i for k, i in enumerate(idx) if k == 0 or i - idx[k - 1] > 6]
        i0 = starts[-1] if starts else 0
        agg = {}
        for name, _, v in allr[i0:]:
            a = agg.get(name, (0, 0.0)); agg[name] = (a[0] + 1, a[1] + (v or 0.0))
        rows = [(k, n, v) for k, (n, v) in agg.items()]
    else:
        rows = cur.execute("select name, count(*), sum(counter_value) from pmc_events where counter_name=? group by name", (counter,)).fetchall()
    # (kernels templated on switches are reported under their plain name - k_gather2<false, 23552u> -> k_gather2 -, launches of several
    # instantiations of one kernel are added up; the scans keep their element type: k_scan_apply<U4>)
    import re
    out = {}
    for name, n, v in rows:
        k = name.split("(")[0].replace("void ", "")
        if not k.startswith("k_scan_"):
            k = re.sub(r"<.*>$", "", k)
        a = out.get(k, (0, 0.0)); out[k] = (a[0] + n, a[1] + (v or 0.0))
    return out


def main():
    f = per_kernel(sys.argv[1], "FETCH_SIZE"); w = per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {}
    print("%-28s %6s %14s %14s %14s" % ("kernel", "calls", "fetch_MB/launch", "write_MB/launch", "total_MB/launch"))
    for k in sorted(set(f) | set(w), key=lambda k: -(2 * f.get(k, (1, 0))[1] / max(1, f.get(k, (1, 0))[0]) + w.get(k, (1, 0))[1] / max(1, w.get(k, (1, 0))[0]))):
        nf, vf = f.get(k, (1, 0.0)); nw, vw = w.get(k, (1, 0.0))
        fb = 2.0 * vf * 1024 / max(1, nf); wb = vw * 1024 / max(1, nw)
        out[k] = {"fetch_bytes": fb, "write_bytes": wb, "calls": nf}
        print("%-28s %6d %14.1f %14.1f %14.1f" % (k[:28], nf, fb / 1e6, wb / 1e6, (fb + wb) / 1e6))
    if len(sys.argv) > 3:
        json.dump(out, open(sys.argv[3], "w"), indent=0, sort_keys=True)


if __name__ == "__main__":
    main()
