import sys, os
sys.path.insert(0, os.getcwd())
import torch
from generativeimage2text_amd import engine as E
from tools.gemm_bench import bench
g = torch.Generator().manual_seed(0)
for (name, M, N, K, odt, act) in [("c_fc", 12608, 3072, 768, torch.bfloat16, 1), ("qkv", 12608, 2304, 768, torch.bfloat16, 0),
                                  ("c_proj", 12608, 768, 3072, torch.float32, 0), ("big4k", 4096, 4096, 4096, torch.bfloat16, 0)]:
    A = torch.randn(M, K, generator=g).bfloat16().cuda()
    W = (torch.randn(N, K, generator=g) * K ** -0.5).bfloat16().cuda()
    bias = torch.randn(N, generator=g).cuda()
    line = f"{name:7s}"
    for dbg, label in [(0, "full"), (1, "noStore"), (2, "noEpi"), (16, "noLdsWrite"), (32, "noReadback"), (48, "syncOnly")]:
        E.set_gemm_impl(2 | (dbg << 8))
        ms = bench(lambda: E.op_gemm(A, W, bias, None, act, odt))
        line += f"  {label}={ms*1e3:.1f}us"
    print(line, flush=True)
    E.set_gemm_impl(-1)
