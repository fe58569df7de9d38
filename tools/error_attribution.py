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

    python tools/error_attribution.py [--model GIT_BASE] [--weights bench|trained|oracle] [--out profiles/rNN_error_attribution.txt]
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


def case_inputs(args):
    """-> (golden name, cfg, weights, frames).  --weights bench: the benchmark's generator; trained: the same with
    synthetic.apply_trained_statistics (tests/golden/full_trained_b64_greedy); oracle: the oracle's generator with perturbed
    LayerNorms, width^-0.5 decoder matrices and an untied output (tests/golden/full_base_b64_greedy -- a study tool may use the
    oracle's weight generator, as tools/residual_precision_study.py does; nothing in the product does)."""
    from generativeimage2text_amd.configs import config_for_model
    from generativeimage2text_amd.synthetic import random_frames, random_state_dict, seeded_images
    cfg = config_for_model(args.model)
    if args.weights == "bench":
        golden, seed, B, F = CASES[args.model]
        return golden, cfg, random_state_dict(cfg, seed=seed), random_frames(cfg, B, F, seed=0)
    assert args.model == "GIT_BASE", "--weights trained / oracle: GIT_BASE only"
    if args.weights == "trained":
        import ast
        golden = "full_trained_b64_greedy"
        g = np.load(os.path.join(ROOT, "tests", "golden", golden + ".npz"))
        wsrc = ast.literal_eval(str(g["weights"]))
        return (golden, cfg, random_state_dict(cfg, seed=wsrc[1], eos_bias=wsrc[2], successor=wsrc[3], stats="trained"),
                seeded_images(cfg, g["image_seeds"].tolist()))
    from oracle import git_oracle as O
    golden = "full_base_b64_greedy"
    w = O.make_weights(O.CONFIGS["GIT_BASE"], seed=1240, tie_output=False, successor=1.0)
    frames = [f.cuda() for f in O.make_images(O.CONFIGS["GIT_BASE"], 64, 1, seed=sum(map(ord, golden)))]
    return golden, cfg, w, frames


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="GIT_BASE", choices=sorted(CASES))
    ap.add_argument("--weights", default="bench", choices=["bench", "trained", "oracle"])
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    from generativeimage2text_amd.engine import use_experiment_build
    use_experiment_build(True)          # these hooks / schedules are exported by libgitmi_exp.so only
    from generativeimage2text_amd.engine import Engine

    golden, cfg, w, frames = case_inputs(args)
    B, F = int(frames[0].shape[0]), len(frames)
    g = np.load(os.path.join(ROOT, "tests", "golden", golden + ".npz"))
    tf = torch.from_numpy(g["tf_tokens"])
    ref = g["tf_logits"]                                     # reference fp32 logits, rows 0..3, every third token
    span = float(ref.max() - ref.min())
    full_ref = {}

    def err(logits):
        """vs the REFERENCE's frozen logits on the sampled entries (max, rms) and, once the f32 engine has run, vs the f32 engine
        mode over ALL B x V entries (max, rms)"""
        l = logits.float().cpu().numpy()
        d = l[:4, ::3] - ref
        if "f32" not in full_ref:
            full_ref["f32"] = l
        da = l - full_ref["f32"]
        return float(np.abs(d).max()), float(np.sqrt((d * d).mean())), float(np.abs(da).max()), float(np.sqrt((da * da).mean()))

    eng = {}
    for prec in ("bf16", "f32"):
        e = Engine(cfg, precision=prec, max_batch=B, max_beams=1, max_frames=F, max_text_len=20)
        e.load_state_dict(w)
        eng[prec] = e
    rows = []

    def run(tag, fn):
        mx, rms, amx, arms = err(fn())
        rows.append((tag, mx, rms, amx, arms))
        print("%-44s max %.5f  rms %.5f  (%.2e x span) | all entries vs f32 mode: max %.5f rms %.5f (%.2e x span)"
              % (tag, mx, rms, mx / span, amx, arms, amx / span), flush=True)

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

    print("model %s, weights %s, golden %s, logit span %.3f" % (args.model, args.weights, golden, span))
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
            f.write("# teacher-forced logit error (%s weights, %s): vs the reference's fp32 logits on %d rows x %d sampled tokens, and vs the\n"
                    "# f32 engine mode over all %d x %d logits; logit span %.3f; north_star's tolerance = 1e-3 x span = %.4f\n"
                    % (args.weights, golden, ref.shape[0], ref.shape[1], B, cfg.vocab, span, 1e-3 * span))
            f.write("# stages: V image encoder, P decoder prefill (image rows), C decode chain (text rows), H vocabulary head\n")
            f.write("%-46s %10s %10s %12s | %10s %10s %12s\n" % ("precision per stage", "max|err|", "rms err", "max / span",
                                                                 "all: max", "all: rms", "max / span"))
            for tag, mx, rms, amx, arms in rows:
                f.write("%-46s %10.5f %10.5f %12.2e | %10.5f %10.5f %12.2e\n" % (tag, mx, rms, mx / span, amx, arms, amx / span))


if __name__ == "__main__":
    main()
