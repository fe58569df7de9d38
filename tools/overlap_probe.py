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

def run(n_ctx, steps_T, n=40, ring=True):
    search = Engine.make_search("greedy", steps_T, 1, 1)
    use = ctxs[:n_ctx]
    for i, c in enumerate(use):
        c.set_encode_after(None)
    if ring and n_ctx > 1:
        for i, c in enumerate(use):
            c.set_encode_after(use[i - 1])
    def go(k):
        for j in range(k):
            i = j % n_ctx
            with torch.cuda.stream(streams[i]):
                use[i].generate(frames, search, sync=False)
    go(2 * n_ctx); torch.cuda.synchronize()
    t0 = time.perf_counter(); go(n); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

for T_ in (20, 1):
    for n_ctx in (1, 2, 4, 8):
        print("%s contexts=%d: %.3f ms/batch" % ("full call   " if T_ == 20 else "encoder only", n_ctx, run(n_ctx, T_)), flush=True)
