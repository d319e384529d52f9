#!/usr/bin/env python3
"""Per-kernel SQ counters from rocprofv3 PMC passes (each pass its own run, --kernel-trace only): prints counter / launch.
usage: pmc_sq.py pass1.db [pass2.db ...]"""
import sqlite3
import sys


def main():
    tab = {}
    names = []
    for path in sys.argv[1:]:
        cur = sqlite3.connect(path).cursor()
        for name, cname, n, v in cur.execute("select name, counter_name, count(*), sum(counter_value) from pmc_events group by name, counter_name"):
            k = name.split("(")[0].replace("void ", "")
            tab.setdefault(k, {})[cname] = v / max(1, n)
            if cname not in names:
                names.append(cname)
    print("%-26s" % "kernel" + "".join("%18s" % c[:17] for c in names))
    for k in sorted(tab, key=lambda k: -tab[k].get(names[0], 0)):
        print("%-26s" % k[:26] + "".join("%18.4g" % tab[k].get(c, float("nan")) for c in names))


if __name__ == "__main__":
    main()
