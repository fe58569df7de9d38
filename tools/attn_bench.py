"""A/B of the full-attention kernels on the GIT_BASE bs=64 shape: python tools/attn_bench.py"""
import sys, os
sys.path.insert(0, os.getcwd())
import torch
from generativeimage2text_amd import engine as E
from tools.gemm_bench import bench
for (B, N, H) in [(64, 197, 12), (32, 257, 16)]:
    qkv = (torch.randn(B * N, 3 * H * 64) * 1.5).bfloat16().cuda()
    line = f"B={B} N={N} H={H}:"
    outs = {}
    for impl in (1, 2):
        outs[impl] = E.op_attention(qkv, B, N, H, impl=impl)
        ms = bench(lambda: E.op_attention(qkv, B, N, H, impl=impl), reps=30)
        line += f"  impl{impl}={ms*1e3:.1f}us"
    line += f"  maxdiff={(outs[1].float()-outs[2].float()).abs().max().item():.3g}"
    print(line, flush=True)
