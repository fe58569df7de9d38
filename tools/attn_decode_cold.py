import os, sys
sys.path.insert(0, "/root/repo")
import torch
from generativeimage2text_amd import engine as E
gen = torch.Generator().manual_seed(0)
B, H, N_img, T = 64, 12, 197, 20
d = H * 64
qkv = torch.randn(B, 3 * d, generator=gen).bfloat16().cuda()
tk = torch.randn(B, T, d, generator=gen).bfloat16().cuda(); tv = torch.randn(B, T, d, generator=gen).bfloat16().cuda()
src = torch.arange(B, dtype=torch.int32)[:, None].repeat(1, T).cuda()
sets = []
for i in range(8):
    ik = torch.randn(B, H, N_img, 64, generator=gen).bfloat16().cuda(); iv = torch.randn(B, H, N_img, 64, generator=gen).bfloat16().cuda()
    sets.append(E.kv_repack(ik, iv))
def timeit(nsets, n=96):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        g = torch.cuda.CUDAGraph()
        for i in range(3):
            E.op_attn_decode(qkv, sets[0][0], sets[0][1], tk, tv, src, B, H, N_img, T, 9, 1)
        with torch.cuda.graph(g, stream=s):
            for i in range(n):
                kf, vt = sets[i % nsets]
                E.op_attn_decode(qkv, kf, vt, tk, tv, src, B, H, N_img, T, 9, 1)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        g.replay(); torch.cuda.synchronize()
        a.record()
        for _ in range(5): g.replay()
        b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / (5 * n) * 1e3
for ns in (1, 2, 4, 6, 8):
    print("attn_decode bf16, %d rotating KV sets (%.0f MB): %.2f us" % (ns, ns * 43.0 * 224 / 197, timeit(ns)), flush=True)
