#!/usr/bin/env python
"""Where does a request's time go in the mixed schedule?  (GIT_BASE, bs = 64, greedy, bf16)

Per-request phase durations, measured with events around the two halves of every call while the other contexts keep the
device busy -- rocprofv3 cannot show this (its kernel trace serialises the kernels, profiles/r03_h_mix_timeline.txt):

    default    4 contexts, whole calls split into gitmi_generate_encode + gitmi_generate_decode on the context's stream
    enc-only   the same ring without the decode halves (what the image encoders alone deliver)
    group G    decode groups of G requests (gitmi_set_decode_group / gitmi_group_decode)

    python tools/group_probe.py [--steps 48] [--contexts 4] [--chains 2]
"""
from __future__ import annotations

import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def med(x):
    x = sorted(x)
    return x[len(x) // 2] if x else float("nan")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=48)
    ap.add_argument("--contexts", type=int, default=4)
    ap.add_argument("--chains", type=int, default=2)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--groups", type=str, default="2,4")
    args = ap.parse_args()
    from generativeimage2text_amd.engine import use_experiment_build
    use_experiment_build(True)          # these hooks / schedules are exported by libgitmi_exp.so only
    from generativeimage2text_amd.configs import config_for_model
    from generativeimage2text_amd.engine import Engine
    from generativeimage2text_amd.synthetic import random_frames, random_state_dict
    cfg = config_for_model("GIT_BASE")
    eng = Engine(cfg, precision="bf16", max_batch=args.batch, max_beams=1, max_frames=1, max_text_len=20)
    eng.load_state_dict(random_state_dict(cfg, seed=1234))
    frames = random_frames(cfg, args.batch, 1, seed=0)
    search = Engine.make_search("greedy", 20, 1, 1)
    E = lambda: torch.cuda.Event(enable_timing=True)

    def chain(ctxs, c):
        if len(ctxs) > c:
            for i, x in enumerate(ctxs):
                x.set_encode_after(ctxs[i - c])

    def report(tag, wall, n, phases):
        print("%-22s %7.3f ms per request  (%6.0f captions/s)  " % (tag, wall / n * 1e3, n * args.batch / wall)
              + "  ".join("%s %.2f" % (k, med(v)) for k, v in phases.items()), flush=True)

    def run_split(decode=True):
        ctxs = [eng.clone() for _ in range(args.contexts)]
        chain(ctxs, args.chains)
        streams = [torch.cuda.Stream() for _ in ctxs]
        rec = []

        def one(i, keep):
            with torch.cuda.stream(streams[i]):
                a, b, c = E(), E(), E()
                a.record()
                ctxs[i].generate_encode(frames, search)
                b.record()
                if decode:
                    ctxs[i].generate_decode(search, sync=False)
                c.record()
                if keep:
                    rec.append((a, b, c))
        for k in range(2 * len(ctxs)):
            one(k % len(ctxs), False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(args.steps):
            one(k % len(ctxs), True)
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        report("default (split calls)" if decode else "encoders only", wall, args.steps,
               {"enc+prefill ms": [a.elapsed_time(b) for a, b, c in rec], "decode ms": [b.elapsed_time(c) for a, b, c in rec],
                "request ms": [a.elapsed_time(c) for a, b, c in rec]})
        for x in ctxs:
            x.close()

    def run_group(G, n_members):
        members = [eng.clone() for _ in range(n_members)]
        groups = [eng.clone(max_batch=G * args.batch) for _ in range(n_members // G)]
        for i, m in enumerate(members):
            m.set_decode_group(groups[i // G], (i % G) * args.batch)
        chain(members, args.chains)
        streams = [torch.cuda.Stream() for _ in members]
        gstreams = [torch.cuda.Stream() for _ in groups]
        rec, drec = [], []
        pend = [[] for _ in groups]

        def one(i, keep):
            g, slot = divmod(i, G)
            with torch.cuda.stream(streams[i]):
                a, b = E(), E()
                a.record()
                members[i].generate_encode(frames, search)
                b.record()
                pend[g].append((a, b))
            if slot == G - 1:
                with torch.cuda.stream(gstreams[g]):
                    c, d = E(), E()
                    c.record()                   # = the group's previous decode has finished
                    groups[g].group_decode(1, G * args.batch, search, sync=False)
                    d.record()
                    if keep:
                        for a, b in pend[g]:
                            rec.append((a, b, d))
                        drec.append((pend[g][-1][1], c, d))
                    pend[g] = []
        for k in range(2 * n_members):
            one(k % n_members, False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(args.steps):
            one(k % n_members, True)
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        report("group %d, %d members" % (G, n_members), wall, args.steps,
               {"enc+prefill+publish ms": [a.elapsed_time(b) for a, b, d in rec],
                "last publish -> decode end ms": [b.elapsed_time(d) for b, c, d in drec],
                "request ms": [a.elapsed_time(d) for a, b, d in rec]})
        for x in members + groups:
            x.close()

    run_split(True)
    run_split(False)
    for G in [int(x) for x in args.groups.split(",") if x]:
        for n in sorted({max(2, args.contexts // G) * G, 2 * G}):
            run_group(G, n)
    run_split(True)
    eng.close()


if __name__ == "__main__":
    main()
