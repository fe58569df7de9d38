#!/usr/bin/env python
"""Would touching a decode GEMM's weights one launch ahead (from arbitrary CUs: lines land in the memory-side cache)
shorten the GEMM?  Chain of [touch(X), dgemm(W_i)] over S rotating weight sets (S x |W| beyond the Infinity Cache, so
W_i is HBM-cold when its turn comes): X = W_(i+1) (prefetch) vs X = an unrelated rotating buffer (no prefetch).
Same launches, same bytes; the difference is what prefetch buys per GEMM."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from generativeimage2text_amd import engine as E

gen = torch.Generator().manual_seed(0)
R = 64
def chain_time(build, n):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        build(0); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for i in range(n):
                build(i)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        g.replay(); torch.cuda.synchronize()
        a.record()
        for _ in range(3):
            g.replay()
        b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / (3 * n) * 1e3

for name, (N, K) in {"qkv": (2304, 768), "ffn1": (3072, 768)}.items():
    S = 96
    A = E.to_frag(torch.randn(R, K, generator=gen).bfloat16().cuda(), 64)
    Ws = [E.to_frag((torch.randn(N, K, generator=gen) * K ** -0.5).bfloat16().cuda()) for _ in range(S)]
    Us = [torch.randn_like(Ws[0].float()).bfloat16() for _ in range(S)]
    bias = torch.randn(N, generator=gen).cuda()
    x = torch.randn(R, K, generator=gen).cuda()
    stats = E.strip_stats(x)
    cs = torch.randn(N, generator=gen).cuda()
    sink = torch.zeros(1, device="cuda")
    def touch(t):
        # one 4-byte element per 128-byte line
        sink.add_(t.view(-1).view(torch.int32)[::32].sum())
    def gemm(i):
        E.op_dgemm(A, Ws[i % S], bias, cs, stats, 1e-12, 0, packed=True)
    t_warm = chain_time(lambda i: gemm(0), 192)
    t_cold = chain_time(lambda i: gemm(i), 192)
    t_pref = chain_time(lambda i: (touch(Ws[(i + 1) % S]), gemm(i)), 192)
    t_nopf = chain_time(lambda i: (touch(Us[i % S]), gemm(i)), 192)
    print("%s: dgemm warm %.2f us, cold (rotating %d sets) %.2f us; [touch next W + dgemm] %.2f us vs [touch unrelated + dgemm] %.2f us"
          % (name, t_warm, S, t_cold, t_pref, t_nopf), flush=True)
