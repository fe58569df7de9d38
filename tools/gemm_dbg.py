"""Ablation timings of a GEMM variant on the encoder shapes: python tools/gemm_dbg.py <impl> (dbg bits: see the kernel)."""
import sys, os
sys.path.insert(0, os.getcwd())
import torch
from generativeimage2text_amd import engine as E
E.use_experiment_build(True)          # gitmi_debug_set_gemm_impl lives in libgitmi_exp.so
from tools.gemm_bench import bench
impl = int(sys.argv[1]) if len(sys.argv) > 1 else 9
variants = [(0, "full"), (1, "noStore"), (2, "noEpi"), (6, "loadsOnly"), (10, "mfmaOnly")] if impl == 9 else \
    [(0, "full"), (1, "noStore"), (2, "noEpi"), (16, "noLdsWrite"), (32, "noReadback"), (48, "syncOnly")]
g = torch.Generator().manual_seed(0)
for (name, M, N, K, odt, act, use_res) in [("qkv", 12608, 2304, 768, torch.bfloat16, 0, False), ("out", 12608, 768, 768, torch.float32, 0, True),
                                           ("c_fc", 12608, 3072, 768, torch.bfloat16, 0, False), ("c_proj", 12608, 768, 3072, torch.float32, 0, True),
                                           ("big4k", 4096, 4096, 4096, torch.bfloat16, 0, False)]:
    A = torch.randn(M, K, generator=g).bfloat16().cuda()
    W = (torch.randn(N, K, generator=g) * K ** -0.5).bfloat16().cuda()
    bias = torch.randn(N, generator=g).cuda()
    res = torch.randn(M, N, generator=g).cuda() if use_res else None
    line = f"{name:7s}"
    for dbg, label in variants:
        E.set_gemm_impl(impl | (dbg << 8))
        ms = bench(lambda: E.op_gemm(A, W, bias, res, act, odt))
        line += f"  {label}={ms*1e3:.1f}us"
    print(line, flush=True)
    E.set_gemm_impl(-1)
