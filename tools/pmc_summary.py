#!/usr/bin/env python
"""Per-kernel averages of the PMC counters in rocprofv3 rocpd databases (one db per --pmc pass)."""
import sqlite3, sys, collections


def main(paths, out):
    rows = collections.defaultdict(dict)
    for p in paths:
        con = sqlite3.connect(p)
        for name, cname, val, n in con.execute(
                "select name, counter_name, avg(counter_value), count(*) from pmc_events group by name, counter_name"):
            rows[name][cname] = (val, n)
    names = sorted({c for r in rows.values() for c in r})
    lines = ["# per-kernel AVERAGE per dispatch of rocprofv3 --pmc counters (separate passes per counter group)",
             "# FETCH_SIZE/WRITE_SIZE are KiB as reported; on gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x",
             "kernel\t" + "\t".join(names)]
    for k in sorted(rows, key=lambda k: -sum(v[0] for v in rows[k].values())):
        lines.append(k[:90] + "\t" + "\t".join("%.4g" % rows[k][c][0] if c in rows[k] else "-" for c in names))
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:14]))


if __name__ == "__main__":
    main(sys.argv[2:], sys.argv[1])
