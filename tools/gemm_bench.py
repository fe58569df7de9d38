#!/usr/bin/env python
"""A/B micro-benchmark of the large-M bf16 GEMM variants on the GIT_BASE bs=64 shapes (random data).
impl codes (gitmi_debug_set_gemm_impl): 0 tile kernel, 9 LDS-DMA kernel; 9 | (bits << 8) with bits 64 / 128 = 192- / 256-row tile,
256 staged fp32 epilogue, 512 plain stores, 1024 / 2048 / 4096 / 8192 = XCD partition ng 1 / 2 / 4 / 8."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from generativeimage2text_amd import engine as E
E.use_experiment_build(True)          # gitmi_debug_set_gemm_impl lives in libgitmi_exp.so

SHAPES = [  # (name, M, N, K, out dtype, act, residual)
    ("vit.qkv", 12608, 2304, 768, torch.bfloat16, 0, False),
    ("vit.out", 12608, 768, 768, torch.float32, 0, True),
    ("vit.c_fc", 12608, 3072, 768, torch.bfloat16, 1, False),
    ("vit.c_proj", 12608, 768, 3072, torch.float32, 0, True),
    ("patch", 12544, 768, 768, torch.float32, 0, False),
    ("big8k", 8192, 8192, 8192, torch.bfloat16, 0, False),
]


# GEMM_BENCH_EXTRA="name:M:N:K[:f32res]" (comma-separated): extra shapes, e.g. the two K halves of a split-K launch stacked as rows
for _spec in filter(None, os.environ.get("GEMM_BENCH_EXTRA", "").split(",")):
    _f = _spec.split(":")
    _res = len(_f) > 4 and _f[4] == "f32res"
    SHAPES.append((_f[0], int(_f[1]), int(_f[2]), int(_f[3]), torch.float32 if _res else torch.bfloat16, 0, _res))


def bench(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    impls = [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else [str(9 | (128 << 8)), str(9 | (64 << 8))])]
    g = torch.Generator().manual_seed(0)
    only = os.environ.get("GEMM_BENCH_SHAPES")         # comma-separated shape names
    for name, M, N, K, odt, act, use_res in SHAPES:
        if only and name not in only.split(","):
            continue
        A = (torch.randn(M, K, generator=g)).bfloat16().cuda()
        W = (torch.randn(N, K, generator=g) * K ** -0.5).bfloat16().cuda()
        bias = torch.randn(N, generator=g).cuda()
        res = torch.randn(M, N, generator=g).cuda() if use_res else None
        ref = A[:512].float() @ W.float().t() + bias
        if act == 1:
            ref = ref * torch.sigmoid(1.702 * ref)
        if use_res:
            ref = ref + res[:512]
        line = f"{name:10s} M={M} N={N} K={K} out={'f32' if odt == torch.float32 else 'bf16'}:"
        base = None
        for impl in impls:
            E.set_gemm_impl(impl)
            out = E.op_gemm(A, W, bias, res, act, odt)
            if base is None:
                base = out.clone()
            err = (out[:512].float() - ref).abs().max().item()
            tail = (out[-64:].float() - (lambda r: r)(
                (A[-64:].float() @ W.float().t() + bias))).abs().max().item() if (act == 0 and not use_res) else 0.0
            ms = bench(lambda: E.op_gemm(A, W, bias, res, act, odt))
            tf = 2.0 * M * N * K / ms / 1e9
            # same K order in every variant -> bitwise equal to the first one; a rerun after the timing loop screens races
            again = E.op_gemm(A, W, bias, res, act, odt)
            dv = (out.float() - base.float()).abs().max().item()
            dr = (again.float() - out.float()).abs().max().item()
            line += f"  impl{impl}: {ms*1e3:8.1f}us {tf:7.1f}TF err={err:.3g}/{tail:.3g} dbase={dv:.3g} drerun={dr:.3g}"
        print(line, flush=True)
    E.set_gemm_impl(-1)


if __name__ == "__main__":
    main()
