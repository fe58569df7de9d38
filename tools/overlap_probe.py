#!/usr/bin/env python
"""How much of the image encoder and of the decode chains overlaps when several contexts are in flight:
ms per batch for (full call | encoder + prefill only) x contexts x CU reserve."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from generativeimage2text_amd.configs import config_for_model
from generativeimage2text_amd.engine import Engine
from generativeimage2text_amd.synthetic import random_state_dict, random_frames

cfg = config_for_model("GIT_BASE")
B, T = 64, 20
eng = Engine(cfg, precision="bf16", max_batch=B, max_beams=1, max_frames=1, max_text_len=T)
eng.load_state_dict(random_state_dict(cfg, seed=1234))
ctxs = [eng] + [eng.clone() for _ in range(7)]
streams = [torch.cuda.Stream() for _ in ctxs]
frames = random_frames(cfg, B, 1, seed=0)

def run(n_ctx, steps_T, n=48, ring=True, stride=1):
    search = Engine.make_search("greedy", steps_T, 1, 1)
    use = ctxs[:n_ctx]
    for i, c in enumerate(use):
        c.set_encode_after(None)
    if ring and n_ctx > stride:
        for i, c in enumerate(use):
            c.set_encode_after(use[i - stride])        # stride 1: one encoder at a time; stride s: s interleaved chains
    def go(k):
        for j in range(k):
            i = j % n_ctx
            with torch.cuda.stream(streams[i]):
                use[i].generate(frames, search, sync=False)
    go(2 * n_ctx); torch.cuda.synchronize()
    t0 = time.perf_counter(); go(n); t_sub = time.perf_counter() - t0; torch.cuda.synchronize()
    run.submit_ms = t_sub / n * 1e3
    return (time.perf_counter() - t0) / n * 1e3

for n_ctx, stride in ((4, 1), (4, 2), (6, 2), (6, 3), (8, 2), (8, 4), (3, 1), (6, 1)):
    ms = run(n_ctx, 20, ring=True, stride=stride)
    print("contexts=%d encoder chains=%d: %.3f ms/batch" % (n_ctx, stride, ms), flush=True)
for n_ctx in (2, 3, 4, 6, 8):
    print("contexts=%d free-running: %.3f ms/batch" % (n_ctx, run(n_ctx, 20, ring=False)), flush=True)
