#!/usr/bin/env python
"""Which stage owns the bf16 logit error?  ("Next round" 1d)

Teacher-forced logits of the benchmark workload (GIT_BASE, the bench's own weights and images: tests/golden/
full_bench_b64_greedy.npz carries the REFERENCE's fp32 logits for 4 rows x every third token) with the stages of the hot
path switched between the engine's two precisions one at a time.  Two contexts of the same model (bf16 and f32) hand
each other the products of a stage through gitmi_debug_import_stage / gitmi_debug_head_from:

    stages:  V = image encoder (ViT)   P = decoder prefill over the image tokens   C = decode chain (6 layers, text rows)
             H = vocabulary head
    a row "V:f32 P:bf16 C:bf16 H:bf16" = the fp32 context's visual features imported into the bf16 context, which runs
    the rest.

    python tools/error_attribution.py [--model GIT_BASE] [--out profiles/r03_error_attribution.txt]

Uses only golden files and the synthetic weight generator (no oracle import).
"""
from __future__ import annotations

import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CASES = {"GIT_BASE": ("full_bench_b64_greedy", 1234, 64, 1), "GIT_LARGE": ("full_large_b32_greedy", 1242, 32, 1),
         "GIT_BASE_VATEX": ("full_vatex_b16_greedy", 1243, 16, 6)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="GIT_BASE", choices=sorted(CASES))
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    from generativeimage2text_amd.engine import use_experiment_build
    use_experiment_build(True)          # these hooks / schedules are exported by libgitmi_exp.so only
    from generativeimage2text_amd.configs import config_for_model
    from generativeimage2text_amd.engine import Engine
    from generativeimage2text_amd.synthetic import random_frames, random_state_dict

    golden, seed, B, F = CASES[args.model]
    g = np.load(os.path.join(ROOT, "tests", "golden", golden + ".npz"))
    cfg = config_for_model(args.model)
    w = random_state_dict(cfg, seed=seed)
    frames = random_frames(cfg, B, F, seed=0)
    tf = torch.from_numpy(g["tf_tokens"])
    ref = g["tf_logits"]                                     # reference fp32 logits, rows 0..3, every third token
    span = float(ref.max() - ref.min())

    def err(logits):
        d = logits[:4, ::3].float().cpu().numpy() - ref
        return float(np.abs(d).max()), float(np.sqrt((d * d).mean()))

    eng = {}
    for prec in ("bf16", "f32"):
        e = Engine(cfg, precision=prec, max_batch=B, max_beams=1, max_frames=F, max_text_len=20)
        e.load_state_dict(w)
        eng[prec] = e
    rows = []

    def run(tag, fn):
        mx, rms = err(fn())
        rows.append((tag, mx, rms))
        print("%-44s max %.5f  rms %.5f  (%.2e x span)" % (tag, mx, rms, mx / span), flush=True)

    def own(prec):
        e = eng[prec]
        e.encode(frames, return_features=False)
        return e.step_logits(tf)

    def imported(dst, src, stage):
        s, d = eng[src], eng[dst]
        s.encode(frames, return_features=False)
        s.step_logits(tf)                                    # runs the source's prefill (stage 2 needs it)
        d.debug_import_stage(s, stage)
        return d.step_logits(tf)

    def head_only():
        eng["f32"].encode(frames, return_features=False)
        eng["f32"].step_logits(tf)
        return eng["bf16"].debug_head_from(eng["f32"], tf.shape[0])

    print("model %s, golden %s, logit span %.3f" % (args.model, golden, span))
    run("V:f32  P:f32  C:f32  H:f32   (f32 engine)", lambda: own("f32"))
    run("V:bf16 P:bf16 C:bf16 H:bf16  (bf16 engine)", lambda: own("bf16"))
    run("V:f32  P:bf16 C:bf16 H:bf16", lambda: imported("bf16", "f32", 1))
    run("V:f32  P:f32  C:bf16 H:bf16", lambda: imported("bf16", "f32", 2))
    run("V:f32  P:f32  C:f32  H:bf16  (head alone)", head_only)
    run("V:bf16 P:f32  C:f32  H:f32   (ViT alone)", lambda: imported("f32", "bf16", 1))
    run("V:bf16 P:bf16 C:f32  H:f32   (ViT + prefill)", lambda: imported("f32", "bf16", 2))
    for e in eng.values():
        e.close()
    if args.out:
        with open(args.out, "w") as f:
            f.write("# teacher-forced logit error vs the reference's fp32 logits (%s, %d rows x %d tokens sampled), logit span %.3f\n"
                    % (golden, ref.shape[0], ref.shape[1], span))
            f.write("# stages: V image encoder, P decoder prefill (image rows), C decode chain (text rows), H vocabulary head\n")
            f.write("%-46s %10s %10s %12s\n" % ("precision per stage", "max|err|", "rms err", "max / span"))
            for tag, mx, rms in rows:
                f.write("%-46s %10.5f %10.5f %12.2e\n" % (tag, mx, rms, mx / span))


if __name__ == "__main__":
    main()
