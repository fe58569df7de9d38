#!/usr/bin/env python
"""Small fixed workload for rocprofv3 --pmc passes: whole-path generate() calls (GIT_BASE bs=64 greedy, one context, eager
launches, the serving kernel shapes) in the HEADLINE precision (fp16 operands since round 6; `bf16` as argument for the other
build, which also runs the four encoder GEMM shapes as unit launches)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from generativeimage2text_amd import engine as E
from generativeimage2text_amd.configs import config_for_model
from generativeimage2text_amd.synthetic import random_state_dict, random_frames

prec = sys.argv[1] if len(sys.argv) > 1 else "f16"
g = torch.Generator().manual_seed(0)
# N = hidden GEMMs as the engine runs them since round 3: fp16 residual-stream rows in and out
for (M, N, K, odt, act, res) in [(12608, 2304, 768, torch.bfloat16, 0, False), (12608, 768, 768, torch.float16, 0, True),
                                 (12608, 3072, 768, torch.bfloat16, 1, False), (12608, 768, 3072, torch.float16, 0, True)] if prec == "bf16" else []:
    A = torch.randn(M, K, generator=g).bfloat16().cuda()
    W = (torch.randn(N, K, generator=g) * K ** -0.5).bfloat16().cuda()
    bias = torch.randn(N, generator=g).cuda()
    r = torch.randn(M, N, generator=g).half().cuda() if res else None
    for _ in range(5):
        E.op_gemm(A, W, bias, r, act, odt)
torch.cuda.synchronize()
cfg = config_for_model("GIT_BASE")
eng = E.Engine(cfg, precision=prec, max_batch=64, max_beams=1, max_frames=1, max_text_len=20)
eng.load_state_dict(random_state_dict(cfg, seed=1234))
eng.set_graph(False)
eng.set_shared_device(True)          # the kernel shapes of the benchmarked (multi-context) schedule
frames = random_frames(cfg, 64, 1, seed=0)
s = E.Engine.make_search("greedy", 20, 1, 1)
for _ in range(3):
    eng.generate(frames, s)
torch.cuda.synchronize()
