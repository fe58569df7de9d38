import sys, os
sys.path.insert(0, os.getcwd())
import torch
from generativeimage2text_amd import engine as E
from tools.gemm_bench import bench
B, H, N, T, pos, k = 64, 12, 197, 20, 10, 1
d, R = H * 64, B * k
g = torch.Generator().manual_seed(0)
mk = lambda *s: torch.randn(*s, generator=g).bfloat16().cuda()
qkv, tk, tv = mk(R, 3 * d), mk(R, T, d), mk(R, T, d)
# 6 "layers" of image KV so that the working set exceeds nothing in particular but rotates like the real decode step
iks, ivs = [mk(B, H, N, 64) for _ in range(6)], [mk(B, H, N, 64) for _ in range(6)]
src = torch.arange(R).int()[:, None].repeat(1, T).contiguous().cuda()
for dbg, label in [(0, "full"), (1, "noImgLoads"), (2, "noScores"), (4, "noPV"), (6, "noScores+noPV"), (7, "skeleton")]:
    def run():
        for l in range(6):
            E.op_attn_decode(qkv, iks[l], ivs[l], tk, tv, src, B, H, N, T, pos, k, dbg)
    ms = bench(run, reps=20)
    print(f"{label:16s} {ms*1e3/6:.2f} us per launch", flush=True)
