#!/usr/bin/env python
"""One round of 256x256 tiles of the K = 768 encoder GEMM at 63 / 126 / 189 / 252 workgroups (M = 256 k rows, N = 2304):
is a tile's time a property of the tile (latency-bound) or of how many run at once (feed-bound)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from generativeimage2text_amd import engine as E
E.use_experiment_build(True)          # gitmi_debug_set_gemm_impl lives in libgitmi_exp.so
from tools.gemm_bench import bench
g = torch.Generator().manual_seed(0)
E.set_gemm_impl(9 | (128 << 8))          # p8, 256-row tiles forced
for N, K in ((2304, 768), (768, 768), (768, 3072), (3072, 768)):
    line = "N=%d K=%d:" % (N, K)
    for k in (1, 7, 14, 21, 28, 56):
        M = 256 * k
        A = torch.randn(M, K, generator=g).bfloat16().cuda()
        W = (torch.randn(N, K, generator=g) * K ** -0.5).bfloat16().cuda()
        bias = torch.randn(N, generator=g).cuda()
        ms = bench(lambda: E.op_gemm(A, W, bias, None, 0, torch.bfloat16), reps=50)
        line += "  %d tiles %.1f us" % (k * (N // 256), ms * 1e3)
    print(line, flush=True)
E.set_gemm_impl(-1)
