#!/usr/bin/env python
"""Does a chain of EMPTY-ish dependent launches on another stream slow the image encoder?  (kernel-boundary effects:
cache write-back / invalidate, dispatcher) -- encoder-only ring throughput with and without a background chain.
The footprint variants (a spin kernel of W workgroups x T microseconds) need the `gitmi_debug_spin` hook of commit
514a2f0; on the current tree only the one-workgroup chain runs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from generativeimage2text_amd.configs import config_for_model
from generativeimage2text_amd.engine import Engine
from generativeimage2text_amd.synthetic import random_state_dict, random_frames

cfg = config_for_model("GIT_BASE")
B = 64
eng = Engine(cfg, precision="bf16", max_batch=B, max_beams=1, max_frames=1, max_text_len=20)
eng.load_state_dict(random_state_dict(cfg, seed=1234))
ctxs = [eng, eng.clone()]
streams = [torch.cuda.Stream() for _ in ctxs]
frames = random_frames(cfg, B, 1, seed=0)
search = Engine.make_search("greedy", 1, 1, 1)

bg_stream = torch.cuda.Stream()
x = torch.zeros(64, device="cuda")
import ctypes
from generativeimage2text_amd import engine as E
lib = E.load_library()
HAVE_SPIN = hasattr(lib, "gitmi_debug_spin")
if HAVE_SPIN:
    lib.gitmi_debug_spin.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p]

def make_bg(n_nodes, blocks=0, threads=0, ticks=0):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(bg_stream):
        x.add_(1.0); torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=bg_stream):
            for _ in range(n_nodes):
                if blocks:
                    assert lib.gitmi_debug_spin(blocks, threads, ticks, torch.cuda.current_stream().cuda_stream) == 0
                else:
                    x.add_(1.0)
    torch.cuda.synchronize()
    return g

def run(bg, n=40):
    def go(k):
        for j in range(k):
            i = j % 2
            with torch.cuda.stream(streams[i]):
                ctxs[i].generate(frames, search, sync=False)
            if bg is not None:
                with torch.cuda.stream(bg_stream):
                    bg.replay()
    go(4); torch.cuda.synchronize()
    t0 = time.perf_counter(); go(n); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

print("encoder only, no background chain: %.3f ms/batch" % run(None), flush=True)
for nodes, blocks, threads, ticks in ((600, 0, 0, 0), (600, 768, 128, 600), (600, 144, 256, 400), (600, 256, 256, 400),
                                      (600, 48, 256, 400), (600, 24, 256, 400), (600, 48, 256, 1200), (600, 768, 128, 100)):
    if blocks and not HAVE_SPIN:
        continue
    g = make_bg(nodes, blocks, threads, ticks)
    print("  background kernel: %d workgroups x %d threads x %.1f us" % (blocks, threads, ticks / 100.0))
    with torch.cuda.stream(bg_stream):
        torch.cuda.synchronize(); t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize()
        solo = (time.perf_counter() - t0) * 1e3
    print("encoder only + %4d-launch chain per batch (chain alone %.2f ms): %.3f ms/batch" % (nodes, solo, run(g)), flush=True)
