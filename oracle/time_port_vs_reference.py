#!/usr/bin/env python
"""How representative is the CPU port (oracle/git_oracle.py) of the reference's own CPU speed?

bench.py's `cpu_baseline` times the port on the GPU box's host cores because /root/reference does not exist there
(kind = "port").  This script -- run HERE, where the reference is importable -- times the unmodified reference modules
(CaptioningModel.forward: generativeimage2text/layers/decoder.py:838-1011) and the port on the same weights, images,
thread count and batch size, and writes oracle/port_vs_reference.json; bench.py copies the ratio into its line
(`cpu_baseline.port_vs_reference_ratio`).  Test infrastructure: nothing in the product imports it."""
import json
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
from oracle import git_oracle as O                                   # noqa: E402
import make_golden as MG                                             # noqa: E402


def main(batch=8, threads=8, repeats=3, max_steps=20):
    torch.set_num_threads(threads)
    cfg = O.CONFIGS["GIT_BASE"]
    w = O.make_weights(cfg, seed=1234)
    frames = O.make_images(cfg, batch, 1, seed=0)
    search = O.SearchConfig("greedy", max_steps, 1, 1)
    ref = MG.build_reference(cfg, w, search, tie=True)

    def t_ref():
        t0 = time.time()
        with torch.no_grad():
            out = ref({"image": frames[0]})
        return time.time() - t0, out["predictions"]

    def t_port():
        t0 = time.time()
        with torch.no_grad():
            out = O.caption(cfg, w, frames, search, cached=False)
        return time.time() - t0, out["predictions"]
    t_ref(), t_port()                                                 # warm-up
    rs, ps = [], []
    for _ in range(repeats):
        a, pa = t_ref()
        b, pb = t_port()
        assert torch.equal(pa, pb)
        rs.append(a); ps.append(b)
    r, p = sorted(rs)[len(rs) // 2], sorted(ps)[len(ps) // 2]
    out = {"batch": batch, "threads": threads, "host_cpus": os.cpu_count(), "max_steps": max_steps,
           "reference_s": round(r, 2), "port_s": round(p, 2),
           "reference_captions_per_s": round(batch / r, 4), "port_captions_per_s": round(batch / p, 4),
           "port_vs_reference_ratio": round(r / p, 3),
           "note": "ratio = port captions/s / reference captions/s on the same host; median of %d alternating passes after one "
                   "warm-up; ids equal" % repeats}
    with open(os.path.join(HERE, "port_vs_reference.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
