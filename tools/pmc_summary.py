#!/usr/bin/env python
"""Per-kernel averages (over dispatches) of the PMC counters in rocprofv3 rocpd databases (one db per --pmc pass).

rocprofv3 stores one row per counter INSTANCE and dispatch: SQ_* counters come as 32 shader-engine instances (summed
here: checked against a known MFMA count -- sum SQ_VALU_MFMA_BUSY_CYCLES = 16 cycles x #v_mfma_16x16x32 wave-instructions),
GRBM_GUI_ACTIVE as 8 XCD instances (averaged: every XCD counts the same wall clock), FETCH_SIZE / WRITE_SIZE / TCC_*_sum
as one.  Derived:
    MFMA_UTIL_PCT = 100 * sum(SQ_VALU_MFMA_BUSY_CYCLES) / (GRBM_GUI_ACTIVE * 256 CUs * 4 SIMDs)
    HBM_BYTES     = (2 * FETCH_SIZE + WRITE_SIZE) * 1024   (gfx950 FETCH_SIZE counts 64 B per 128-B request,
                                                            MI355X_MICROARCH.md "HBM")
The header records a hash of csrc/*.hip at collection time so that bench.py can tell a stale profile from a current one.
"""
import collections
import glob
import hashlib
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def csrc_sha() -> str:
    h = hashlib.sha1()
    for f in sorted(glob.glob(os.path.join(ROOT, "generativeimage2text_amd", "csrc", "*.hip")) +
                    glob.glob(os.path.join(ROOT, "generativeimage2text_amd", "csrc", "*.h"))):
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def main(out, paths):
    per = collections.defaultdict(lambda: collections.defaultdict(list))     # kernel -> counter -> per-dispatch values
    for p in paths:
        con = sqlite3.connect(p)
        q = ("select name, counter_name, dispatch_id, sum(counter_value), avg(counter_value), count(*) "
             "from pmc_events group by name, counter_name, dispatch_id")
        for name, cname, _, vsum, vavg, n in con.execute(q):
            per[name][cname].append(vavg if cname.startswith("GRBM") else vsum)
    rows = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in per.items()}
    for k, r in rows.items():
        if "SQ_VALU_MFMA_BUSY_CYCLES" in r and r.get("GRBM_GUI_ACTIVE", 0) > 0:
            r["MFMA_UTIL_PCT"] = 100.0 * r["SQ_VALU_MFMA_BUSY_CYCLES"] / (r["GRBM_GUI_ACTIVE"] * 1024.0)
        if "FETCH_SIZE" in r and "WRITE_SIZE" in r:
            r["HBM_BYTES"] = (2.0 * r["FETCH_SIZE"] + r["WRITE_SIZE"]) * 1024.0
        if "TCC_HIT_sum" in r and "TCC_MISS_sum" in r and r["TCC_HIT_sum"] + r["TCC_MISS_sum"] > 0:
            r["L2_HIT_PCT"] = 100.0 * r["TCC_HIT_sum"] / (r["TCC_HIT_sum"] + r["TCC_MISS_sum"])
        r["DISPATCHES"] = max(len(v) for v in per[k].values())
    names = sorted({c for r in rows.values() for c in r})
    lines = ["# per-kernel AVERAGE per dispatch of rocprofv3 --pmc counters (separate passes per counter group; SQ_* summed over "
             "the 32 shader-engine instances, GRBM_GUI_ACTIVE averaged over XCDs)",
             "# FETCH_SIZE/WRITE_SIZE are KiB as reported; HBM_BYTES = (2*FETCH_SIZE + WRITE_SIZE)*1024; "
             "MFMA_UTIL_PCT = 100*SQ_VALU_MFMA_BUSY_CYCLES/(GRBM_GUI_ACTIVE*1024 SIMDs)",
             "# csrc_sha=" + csrc_sha(),
             "kernel\t" + "\t".join(names)]
    for k in sorted(rows, key=lambda k: -rows[k].get("GRBM_GUI_ACTIVE", 0) * rows[k].get("DISPATCHES", 1)):
        lines.append(k[:110] + "\t" + "\t".join("%.5g" % rows[k][c] if c in rows[k] else "-" for c in names))
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:16]))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2:])
