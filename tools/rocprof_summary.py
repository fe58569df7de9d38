#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace run (rocpd sqlite .db) into a per-kernel stats table
(the same numbers `--stats` reports): calls, total/avg/min/max duration, % of GPU kernel time."""
import sqlite3
import sys


def csrc_sha():
    """hash of the kernel sources the trace was taken for (the same one bench.py / tools/pmc_summary.py record)"""
    import glob, hashlib, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha1()
    for f in sorted(glob.glob(os.path.join(root, "generativeimage2text_amd", "csrc", "*.hip")) +
                    glob.glob(os.path.join(root, "generativeimage2text_amd", "csrc", "*.h"))):
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def main(db, out=None, calls_per_step=None):
    con = sqlite3.connect(db)
    cols = [r[1] for r in con.execute("pragma table_info(kernels)").fetchall()]
    grid = next((c for c in ("grid_size_x", "grid_x", "grid_size") if c in cols), None)
    wg = next((c for c in ("workgroup_size_x", "workgroup_x", "workgroup_size") if c in cols), None)
    # one row per (kernel, workgroups): the same GEMM instantiation serves several shapes (QKV / prefill ...)
    key = "name" if not (grid and wg) else "name || '  <<<' || (%s / max(%s, 1)) || ' WGs>>>'" % (grid, wg)
    rows = con.execute(
        "select %s, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start), "
        "max(vgpr_count), max(accum_vgpr_count), max(lds_size) from kernels group by 1 order by 3 desc" % key).fetchall()
    total = sum(r[2] for r in rows) or 1
    lines = ["# rocprofv3 --kernel-trace summary of %s" % db,
             "# csrc_sha=%s" % csrc_sha(),
             "# total GPU kernel time %.3f ms over %d dispatches" % (total / 1e6, sum(r[1] for r in rows)),
             "%-9s %7s %12s %10s %10s %10s %5s %5s %7s  %s" % ("pct", "calls", "total_ms", "avg_us", "min_us", "max_us",
                                                              "vgpr", "agpr", "lds", "kernel")]
    for r in rows:
        lines.append("%8.2f%% %7d %12.3f %10.2f %10.2f %10.2f %5s %5s %7s  %s" % (
            100.0 * r[2] / total, r[1], r[2] / 1e6, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3, r[6], r[7], r[8], r[0]))
    text = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(text)
    else:
        sys.stdout.write(text)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
