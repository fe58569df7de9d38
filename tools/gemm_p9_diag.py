#!/usr/bin/env python
"""Where does the persistent GEMM (kernels_gemm11.hip) lose against the round-2 kernel?  Exact-round shapes (N = 2048:
8 column tiles; M = 8192 k: 256 k tiles = k per CU) isolate the cost of the 2nd / 3rd tile of a workgroup from load
imbalance; measurement builds remove the output stores / the whole quadrant output; fewer workgroups per XCD show what
chip-wide contention costs a tile."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from generativeimage2text_amd import engine as E

P9 = 11
VARIANTS = [("p8", 9 | (128 << 8)), ("p9", P9 | (128 << 8)), ("p9-nostore", P9 | ((128 | 1) << 8)), ("p9-noflush", P9 | ((128 | 2) << 8)),
            ("p9-ring8", P9 | ((128 | 512) << 8)), ("p9-wpx16", P9 | ((128 | (16 << 12)) << 8)), ("p9-wpx24", P9 | ((128 | (24 << 12)) << 8))]
SHAPES = [("1 tile/CU", 8192, 2048, 768), ("2 tiles/CU", 16384, 2048, 768), ("3 tiles/CU", 24576, 2048, 768),
          ("4 tiles/CU", 32768, 2048, 768), ("2 tiles/CU K=3072", 16384, 2048, 3072), ("vit.qkv", 12608, 2304, 768)]


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
    for name, M, N, K in SHAPES:
        A = torch.randn(M, K, generator=g).bfloat16().cuda()
        W = (torch.randn(N, K, generator=g) * K ** -0.5).bfloat16().cuda()
        bias = torch.randn(N, generator=g).cuda()
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        lib = E.load_library()

        def run(impl):
            E.set_gemm_impl(impl)
            E._ck(lib.gitmi_op_gemm(A.data_ptr(), W.data_ptr(), bias.data_ptr(), None, out.data_ptr(), M, N, K, K, N,
                                    E.DTYPE_BF16, E.DTYPE_BF16, 0, E._stream()))
        times = {v: [] for v, _ in VARIANTS}
        for _ in range(3):
            for v, i in VARIANTS:
                run(i)
                times[v].append(timed(lambda: run(i), 10))
        line = f"{name:18s} M={M} N={N} K={K}:"
        for v, _ in VARIANTS:
            t = sorted(times[v])[1]
            line += f"  {v} {t:6.1f}us"
        print(line, flush=True)
    E.set_gemm_impl(-1)


if __name__ == "__main__":
    main()
