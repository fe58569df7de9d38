#!/usr/bin/env python
"""Static resource table of every kernel in csrc/ (no GPU needed): registers, scratch (spills), LDS, occupancy as reported
by `hipcc -Rpass-analysis=kernel-resource-usage` for gfx950.    python tools/kernel_resources.py [out.txt] [--f16]"""
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "generativeimage2text_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-Wno-unused-result",
         "-Rpass-analysis=kernel-resource-usage", "-c"]
KEYS = [("TotalSGPRs", "sgpr"), ("VGPRs", "vgpr"), ("AGPRs", "agpr"), ("ScratchSize [bytes/lane]", "scratch"),
        ("Occupancy [waves/SIMD]", "occ"), ("LDS Size [bytes/block]", "lds")]


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout
        return out.splitlines()
    except Exception:
        return names


def main(out_path=None):
    import bench
    if "--f16" in sys.argv:               # the fp16-operand build (the folded-LayerNorm GEMM forms exist only there)
        FLAGS.insert(-1, "-DGITMI_OPS_F16")
    rows = []
    for f in sorted(glob.glob(os.path.join(CSRC, "*.hip"))):
        r = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + [f, "-o", "/tmp/_kres.o"], capture_output=True, text=True)
        cur = None
        for line in r.stderr.splitlines():
            m = re.search(r"remark: (?:\S+: )?\s*Function Name: (\S+)", line)
            if m:
                cur = {"file": os.path.basename(f), "name": m.group(1)}
                rows.append(cur)
                continue
            for key, short in KEYS:
                m = re.search(r"remark:\s+" + re.escape(key) + r": (\d+)", line)
                if m and cur is not None:
                    cur[short] = int(m.group(1))
    for row, nm in zip(rows, demangle([r["name"] for r in rows])):
        row["pretty"] = nm
    lines = ["# static kernel resources (hipcc -Rpass-analysis=kernel-resource-usage, gfx950); csrc sha " + bench.csrc_sha(),
             "%-22s %5s %5s %5s %8s %4s %7s  %s" % ("file", "sgpr", "vgpr", "agpr", "scratch", "occ", "lds", "kernel")]
    for r in rows:
        lines.append("%-22s %5d %5d %5d %8d %4d %7d  %s" % (r["file"], r.get("sgpr", 0), r.get("vgpr", 0), r.get("agpr", 0),
                                                            r.get("scratch", 0), r.get("occ", 0), r.get("lds", 0), r["pretty"][:150]))
    spill = [r for r in rows if r.get("scratch", 0)]
    lines.append("# kernels with scratch (spills): " + (", ".join("%s (%d B/lane)" % (r["pretty"][:60], r["scratch"]) for r in spill) or "none"))
    text = "\n".join(lines) + "\n"
    if out_path:
        open(out_path, "w").write(text)
    else:
        sys.stdout.write(text)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else None)
