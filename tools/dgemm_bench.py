#!/usr/bin/env python
"""Timing of the decode-chain kernels in isolation (back-to-back launches on one stream, HIP events around N launches)
with compile-time-free ablations (gitmi_debug_set_dgemm: 1 no activation loads, 2 no weight loads, 4 no MFMA,
8 no epilogue loads).  Prints microseconds per launch incl. the launch boundary."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from generativeimage2text_amd import engine as E

E.use_experiment_build(True)        # gitmi_debug_set_dgemm is exported by libgitmi_exp.so only
lib = E.load_library()
lib.gitmi_debug_set_dgemm.argtypes = [ctypes.c_int]


def timeit(fn, n=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def graph_timeit(fn, n=100):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n):
                fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        g.replay(); torch.cuda.synchronize()
        a.record()
        for _ in range(5):
            g.replay()
        b.record()
        torch.cuda.synchronize()
    return a.elapsed_time(b) / (5 * n) * 1e3


gen = torch.Generator().manual_seed(0)
R = int(sys.argv[1]) if len(sys.argv) > 1 else 64
for name, (N, K, res) in {"qkv": (2304, 768, False), "ffn1": (3072, 768, False), "out": (768, 768, True), "ffn2": (768, 3072, True)}.items():
    A = E.to_frag(torch.randn(R, K, generator=gen).bfloat16().cuda(), 64)
    W = E.to_frag((torch.randn(N, K, generator=gen) * K ** -0.5).bfloat16().cuda())
    bias = torch.randn(N, generator=gen).cuda()
    x = torch.randn(R, K if not res else N, generator=gen).cuda()
    stats = E.strip_stats(x)
    cs = torch.randn(N, generator=gen).cuda()
    gm, bt = torch.ones(N).cuda(), torch.zeros(N).cuda()
    line = [name]
    for dbg in (0, 1, 2, 3, 4, 8, 15):
        lib.gitmi_debug_set_dgemm(dbg)
        if res:
            fn = lambda: E.op_dgemm_res(A, W, bias, x, stats, gm, bt, packed=True)
        else:
            fn = lambda: E.op_dgemm(A, W, bias, cs, stats, 1e-12, 0, packed=True)
        try:
            line.append("dbg%d %.2f" % (dbg, graph_timeit(fn)))
        except Exception as exc:
            line.append("dbg%d ERR %s" % (dbg, type(exc).__name__))
    lib.gitmi_debug_set_dgemm(0)
    print("  ".join(line), flush=True)
# empty-ish kernel floor: layernorm of 64 rows
xx = torch.randn(64, 768).cuda(); g1 = torch.ones(768).cuda(); b1 = torch.zeros(768).cuda()
print("layernorm64 %.2f" % graph_timeit(lambda: E.op_layernorm(xx, g1, b1, 1e-5)))
# vocabulary head
V, K = 30522, 768
A = E.to_frag(torch.randn(R, K, generator=gen).bfloat16().cuda(), 64)
W = E.to_frag((torch.randn(V, K, generator=gen) * K ** -0.5).bfloat16().cuda(), 128)
Vp = W.shape[0]
bias = torch.zeros(Vp).cuda(); bias[:V] = torch.randn(V, generator=gen).cuda()
x = torch.randn(R, K, generator=gen).cuda()
stats = E.strip_stats(x); cs = torch.zeros(Vp).cuda(); cs[:V] = torch.randn(V, generator=gen).cuda()
for cols in (64, 128):
    for mtop in (1, 8):
        print("vocab cols=%d mtop=%d  %.2f us" % (cols, mtop, graph_timeit(
            lambda: E.op_vocab_topm(A, W, bias, mtop, cols, cs, stats, packed=True, rows=R, V=V), n=30)), flush=True)
# decode attention, 64 images x 12 heads x 197 keys
B, H, N_img, T = R, 12, 197, 20
d = H * 64
qkv = torch.randn(B, 3 * d, generator=gen).bfloat16().cuda()
ik = torch.randn(B, H, N_img, 64, generator=gen).bfloat16().cuda(); iv = torch.randn(B, H, N_img, 64, generator=gen).bfloat16().cuda()
tk = torch.randn(B, T, d, generator=gen).bfloat16().cuda(); tv = torch.randn(B, T, d, generator=gen).bfloat16().cuda()
src = torch.arange(B, dtype=torch.int32)[:, None].repeat(1, T).cuda()
kf, vt = E.kv_repack(ik, iv)
for dbg in (0, 0, 1, 2, 3, 4, 8, 12):
    print("attn_decode (matrix cores, bf16) dbg%d %.2f us" % (dbg, graph_timeit(lambda: E.op_attn_decode(qkv, kf, vt, tk, tv, src, B, H, N_img, T, 9, 1, dbg), n=50)), flush=True)
q32, ik32, iv32, tk32, tv32 = qkv.float(), ik.float(), iv.float(), tk.float(), tv.float()
print("attn_decode (scalar, fp32)       %.2f us" % graph_timeit(lambda: E.op_attn_decode(q32, ik32, iv32, tk32, tv32, src, B, H, N_img, T, 9, 1), n=50), flush=True)
