#!/usr/bin/env python
"""A/B of the persistent GEMM (kernels_gemm11.hip, impl 11) against the round-2 kernel (impl 9) on the shapes of the
image encoder / prefill at the benchmark batch (random data), interleaved rounds in one process.

For the N = hidden GEMMs the comparison is between the two SCHEDULES: impl 9 adds the residual stream in its epilogue
(fp16 stream rows in and out), impl 11 writes the fp16 branch output and the LayerNorm kernel adds it (the extra cost of
that is the `add_ln` vs `ln` line at the end)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from generativeimage2text_amd import engine as E

SHAPES = [  # (name, M, N, K, act, hidden-GEMM?)
    ("vit.qkv", 12608, 2304, 768, 0, False),
    ("vit.c_fc", 12608, 3072, 768, 1, False),
    ("vit.out", 12608, 768, 768, 0, True),
    ("vit.c_proj", 12608, 768, 3072, 0, True),
    ("dec.ffn1", 12608, 3072, 768, 2, False),
    ("large.qkv", 8224, 3072, 1024, 0, False),
    ("large.c_fc", 8224, 4096, 1024, 1, False),
    ("large.c_proj", 8224, 1024, 4096, 0, True),
    ("big8k", 8192, 8192, 8192, 0, False),
]
VARIANTS = [("p8", 9), ("p8/ring", 9 | (1024 << 8)), ("p8/ring+2ph", 9 | (2048 << 8)), ("p9", 11)]


def timed(fn, reps):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def main():
    g = torch.Generator().manual_seed(0)
    only = os.environ.get("GEMM_BENCH_SHAPES")
    rounds, reps = 3, 10
    for name, M, N, K, act, hidden in SHAPES:
        if only and name not in only.split(","):
            continue
        A = torch.randn(M, K, generator=g).bfloat16().cuda()
        W = (torch.randn(N, K, generator=g) * K ** -0.5).bfloat16().cuda()
        bias = torch.randn(N, generator=g).cuda()
        res16 = torch.randn(M, N, generator=g).half().cuda() if hidden else None

        def run(impl):
            E.set_gemm_impl(impl)
            if hidden and (impl & 0xff) == 9:
                return E.op_gemm(A, W, bias, res16, act, torch.float16)      # residual stream RMW in the epilogue
            return E.op_gemm(A, W, bias, None, act, torch.float16 if hidden else torch.bfloat16)
        outs = {v: run(i) for v, i in VARIANTS}
        base = outs["p9"]
        eq = {v: bool(torch.equal(o, outs["p8"])) for v, o in outs.items() if v.startswith("p8/")}
        if not hidden:
            eq["p9"] = bool(torch.equal(outs["p8"], base))
        times = {v: [] for v, _ in VARIANTS}
        for _ in range(rounds):
            for v, i in VARIANTS:
                run(i)
                times[v].append(timed(lambda: run(i), reps))
        line = f"{name:12s} M={M} N={N} K={K} act={act}:"
        for v, _ in VARIANTS:
            t = sorted(times[v])[len(times[v]) // 2]
            line += f"  {v} {t:7.1f}us {2.0 * M * N * K / t / 1e6:6.0f}TF"
        print(line + f"  bitwise== {eq}", flush=True)
    E.set_gemm_impl(-1)
    # the LayerNorm that follows a hidden GEMM: plain (stream in -> bf16 out) vs residual add fused (stream + fp16 branch
    # in -> stream + bf16 out)
    rows, D = 12608, 768
    x = torch.randn(rows, D, generator=g).half().cuda()
    y = torch.randn(rows, D, generator=g).half().cuda()
    gm, bt = torch.ones(D).cuda(), torch.zeros(D).cuda()
    for _ in range(3):
        E.op_add_layernorm(x, y, gm, bt, 1e-5)
    t_add = timed(lambda: E.op_add_layernorm(x, y, gm, bt, 1e-5), 20)
    t_ln = timed(lambda: E.op_add_layernorm(None, y, gm, bt, 1e-5), 20)
    print(f"layernorm {rows}x{D}: add_ln (stream + branch in, stream + bf16 + stream copy out) {t_add:.1f}us, "
          f"ln (fp16 in, bf16 + copy out) {t_ln:.1f}us  (both include the torch.empty of the wrapper)")


if __name__ == "__main__":
    main()
