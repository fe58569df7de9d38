#!/usr/bin/env python
"""One summary line per bench.py JSON file (A/B runs produce dozens): python tools/bench_lines.py gpurun_out/r03_s_bench_*.json"""
import json
import os
import sys


def main():
    for f in sys.argv[1:]:
        try:
            d = json.loads(open(f).read().strip().splitlines()[-1])
        except Exception as e:                                   # a run that died leaves an empty file
            print("%-44s  (no line: %s)" % (os.path.basename(f), type(e).__name__))
            continue
        p, r, rd, c = d.get("parity") or {}, d.get("roofline") or {}, d.get("roofline_decode") or {}, d.get("config") or {}
        print("%-44s %8.1f captions/s  %6.3f ms/pass  latency %6.2f  gemm %5.1f us frac %.4f  enc+prefill %s  decode step %.4f ms"
              "  identical %s/%s  ids==solo %s  | %s, %s contexts, chains %s" % (
                  os.path.basename(f), d["value"], d["ms_per_step"], (d.get("batch_latency_ms") or {}).get("median", 0),
                  1e3 * r.get("avg_launch_ms", 0), r.get("frac", 0), (d.get("phases_ms") or {}).get("graph_encode_prefill_ms"),
                  rd.get("avg_step_ms", 0), p.get("identical"), p.get("rows"), d.get("timed_ids_equal_solo"),
                  c.get("schedule", "?")[:60], c.get("contexts_in_flight"), c.get("encoder_chains")))


if __name__ == "__main__":
    main()
