"""GPU: every HIP kernel that has a single-op C-ABI entry point against a plain PyTorch fp32/fp64
reference of the same op (asymmetric random data, ragged sizes)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def _act(x, act):
    if act == 1:
        return x * torch.sigmoid(1.702 * x)
    if act == 2:
        return x * 0.5 * (1.0 + torch.erf(x / 2 ** 0.5))
    return x


GEMM_SHAPES = [(1000, 768, 768), (197, 2304, 768), (64, 768, 3072), (300, 1002, 128), (1, 128, 64),
               (130, 70, 640), (12608, 768, 768)]


@pytest.mark.parametrize("M,N,K", GEMM_SHAPES)
@pytest.mark.parametrize("act", [0, 1, 2])
def test_gemm_bf16(M, N, K, act):
    from generativeimage2text_amd import engine as E
    A = _rand(M, K, seed=1).bfloat16()
    W = _rand(N, K, seed=2, scale=K ** -0.5).bfloat16()
    bias = _rand(N, seed=3)
    res = _rand(M, N, seed=4)
    ref = _act(A.double() @ W.double().t() + bias.double(), act) + res.double()
    out = E.op_gemm(A.cuda(), W.cuda(), bias.cuda(), res.cuda(), act, torch.float32).cpu().double()
    # fp32 accumulation of exact bf16 products: only summation-order error
    assert (out - ref).abs().max().item() < 2e-4 * max(1.0, ref.abs().max().item())
    out_b = E.op_gemm(A.cuda(), W.cuda(), bias.cuda(), None, act, torch.bfloat16).cpu().double()
    ref_b = _act(A.double() @ W.double().t() + bias.double(), act)
    assert (out_b - ref_b).abs().max().item() < 1e-2 * max(1.0, ref_b.abs().max().item())


@pytest.mark.parametrize("M,N,K", [(1000, 512, 128), (777, 256, 192), (2100, 768, 768), (515, 1024, 3072)])
@pytest.mark.parametrize("act", [0, 1, 2])
@pytest.mark.parametrize("tile", [9 | (128 << 8), 9 | (64 << 8)])          # forced 256x256 / 192x256 tile
def test_gemm_bf16_p8_variant(M, N, K, act, tile):
    """The 256x256 half-tile pipeline kernel (kernels_gemm10.hip), forced: shortest K (2 and 3 K tiles), ragged M,
    every epilogue; it must also equal the 256x128 ring kernel bit for bit (same K order)."""
    from generativeimage2text_amd import engine as E
    A = _rand(M, K, seed=11).bfloat16()
    W = _rand(N, K, seed=12, scale=K ** -0.5).bfloat16()
    bias = _rand(N, seed=13)
    res = _rand(M, N, seed=14)
    ref = _act(A.double() @ W.double().t() + bias.double(), act)
    try:
        E.set_gemm_impl(tile)
        out = E.op_gemm(A.cuda(), W.cuda(), bias.cuda(), res.cuda(), act, torch.float32).cpu()
        out_b = E.op_gemm(A.cuda(), W.cuda(), bias.cuda(), None, act, torch.bfloat16).cpu()
        out_br = E.op_gemm(A.cuda(), W.cuda(), None, res.cuda(), act, torch.bfloat16).cpu()
        E.set_gemm_impl(2)
        ring = E.op_gemm(A.cuda(), W.cuda(), bias.cuda(), res.cuda(), act, torch.float32).cpu()
        ring_b = E.op_gemm(A.cuda(), W.cuda(), bias.cuda(), None, act, torch.bfloat16).cpu()
    finally:
        E.set_gemm_impl(-1)
    assert (out.double() - (ref + res.double())).abs().max().item() < 2e-4 * max(1.0, ref.abs().max().item())
    assert (out_b.double() - ref).abs().max().item() < 1e-2 * max(1.0, ref.abs().max().item())
    ref_br = _act(A.double() @ W.double().t(), act) + res.double()
    assert (out_br.double() - ref_br).abs().max().item() < 2e-2 * max(1.0, ref_br.abs().max().item())
    assert torch.equal(out, ring) and torch.equal(out_b, ring_b)


@pytest.mark.parametrize("M,N,K", [(1000, 768, 768), (64, 768, 3072), (300, 1002, 128), (1, 128, 64), (130, 70, 592)])
def test_gemm_f32_exact_class(M, N, K):
    from generativeimage2text_amd import engine as E
    A = _rand(M, K, seed=5)
    W = _rand(N, K, seed=6, scale=K ** -0.5)
    bias = _rand(N, seed=7)
    ref = (A.double() @ W.double().t() + bias.double())
    out = E.op_gemm(A.cuda(), W.cuda(), bias.cuda(), None, 0, torch.float32).cpu().double()
    assert (out - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item())


def test_gemm_detects_transpose():
    # A = I with asymmetric W: a row/col swap in the accumulator write would transpose the output
    from generativeimage2text_amd import engine as E
    K = 128
    A = torch.eye(K).bfloat16()
    W = (torch.arange(96 * K).reshape(96, K).float() % 251 - 125).bfloat16()
    out = E.op_gemm(A.cuda(), W.cuda(), None, None, 0, torch.float32).cpu()
    assert torch.equal(out, W.float().t())


@pytest.mark.parametrize("rows,D", [(5, 768), (1000, 1024), (33, 128), (7, 192)])
@pytest.mark.parametrize("eps", [1e-5, 1e-12])
def test_layernorm(rows, D, eps):
    from generativeimage2text_amd import engine as E
    x = _rand(rows, D, seed=8, scale=3.0) + 0.5
    g, b = 1 + _rand(D, seed=9, scale=0.1), _rand(D, seed=10, scale=0.1)
    ref = torch.nn.functional.layer_norm(x.double(), (D,), g.double(), b.double(), eps)
    out = E.op_layernorm(x.cuda(), g.cuda(), b.cuda(), eps, torch.float32).cpu().double()
    assert (out - ref).abs().max().item() < 2e-5
    out_b = E.op_layernorm(x.cuda(), g.cuda(), b.cuda(), eps, torch.bfloat16).cpu().double()
    assert (out_b - ref).abs().max().item() < 3e-2


def _attn_ref(qkv, B, N, H):
    D = H * 64
    q, k, v = qkv.double().reshape(B, N, 3, H, 64).permute(2, 0, 3, 1, 4)
    s = (q @ k.transpose(-1, -2)) / 8.0
    return (torch.softmax(s, -1) @ v).permute(0, 2, 1, 3).reshape(B * N, D)


@pytest.mark.parametrize("B,N,H", [(2, 17, 2), (3, 197, 12), (1, 257, 16), (2, 300, 3), (1, 1182, 2), (1, 64, 1),
                                   (2, 193, 3), (2, 208, 2), (2, 272, 2), (1, 209, 1)])
def test_attention_full(B, N, H):
    from generativeimage2text_amd import engine as E
    qkv = _rand(B * N, 3 * H * 64, seed=11, scale=1.5)
    ref = _attn_ref(qkv, B, N, H)
    out = E.op_attention(qkv.cuda(), B, N, H, impl=0).cpu().double()
    assert (out - ref).abs().max().item() < 2e-5
    qb = qkv.bfloat16()
    ref_b = _attn_ref(qb.float(), B, N, H)
    for impl in (0, 1, 2):      # fp32-math VALU twin | auto (single-pass kernel for 193..208 / 257..272 keys) | 64-key flash
        out_b = E.op_attention(qb.cuda(), B, N, H, impl=impl).cpu().double()
        err = (out_b - ref_b).abs().max().item()
        assert err < 3e-2, (impl, err)


def test_attention_softmax_rescale_branch():
    # a spiked key in a LATER tile forces the online-softmax running max to jump (rescale path)
    from generativeimage2text_amd import engine as E
    B, N, H = 1, 200, 1
    qkv = _rand(N, 192, seed=12, scale=0.5)
    qkv[150, 64:128] = qkv[3, 0:64] * 40.0
    ref = _attn_ref(qkv, B, N, H)
    out = E.op_attention(qkv.cuda(), B, N, H, impl=0).cpu().double()
    assert (out - ref).abs().max().item() < 5e-5
    qb = qkv.bfloat16()
    ref_b = _attn_ref(qb.float(), B, N, H)
    for impl in (1, 2):         # 200 keys: single-pass kernel (no rescale at all) and the online-softmax flash kernel
        out_b = E.op_attention(qb.cuda(), B, N, H, impl=impl).cpu().double()
        assert (out_b - ref_b).abs().max().item() < 5e-2, impl


@pytest.mark.parametrize("M,N,K", [(64, 768, 768), (64, 2304, 768), (64, 30522, 768), (256, 3072, 768),
                                   (5, 130, 128), (100, 1002, 96), (64, 768, 3072)])
@pytest.mark.parametrize("NT", [1, 2])
@pytest.mark.parametrize("act", [0, 2])
def test_gemm_skinny(M, N, K, NT, act):
    from generativeimage2text_amd import engine as E
    A = _rand(M, K, seed=21).bfloat16()
    W = _rand(N, K, seed=22, scale=K ** -0.5).bfloat16()
    bias = _rand(N, seed=23)
    res = _rand(M, N, seed=24)
    ref = _act(A.double() @ W.double().t() + bias.double(), act) + res.double()
    out = E.op_gemm_skinny(A.cuda(), W.cuda(), bias.cuda(), res.cuda(), act, torch.float32, NT).cpu().double()
    assert (out - ref).abs().max().item() < 2e-4 * max(1.0, ref.abs().max().item())
    out_b = E.op_gemm_skinny(A.cuda(), W.cuda(), bias.cuda(), None, act, torch.bfloat16, NT).cpu().double()
    ref_b = _act(A.double() @ W.double().t() + bias.double(), act)
    assert (out_b - ref_b).abs().max().item() < 1e-2 * max(1.0, ref_b.abs().max().item())


@pytest.mark.parametrize("M,N,K,S", [(64, 768, 768, 3), (64, 768, 3072, 4), (256, 768, 3072, 2), (7, 128, 512, 4),
                                     (33, 128, 128, 2), (64, 768, 768, 8)])
def test_gemm_splitk_layernorm(M, N, K, S):
    from generativeimage2text_amd import engine as E
    A = _rand(M, K, seed=31).bfloat16()
    W = _rand(N, K, seed=32, scale=K ** -0.5).bfloat16()
    bias, res = _rand(N, seed=33), _rand(M, N, seed=34)
    g, b = 1 + _rand(N, seed=35, scale=0.1), _rand(N, seed=36, scale=0.1)
    pre = A.double() @ W.double().t() + bias.double() + res.double()
    ref = torch.nn.functional.layer_norm(pre, (N,), g.double(), b.double(), 1e-12)
    y_f, y_t = E.op_gemm_splitk_ln(A.cuda(), W.cuda(), bias.cuda(), res.cuda(), g.cuda(), b.cuda(), 1e-12, S)
    assert (y_f.cpu().double() - ref).abs().max().item() < 3e-4
    assert (y_t.cpu().double() - ref).abs().max().item() < 3e-2
    y_f2, _ = E.op_gemm_splitk_ln(A.cuda(), W.cuda(), bias.cuda(), res.cuda(), g.cuda(), b.cuda(), 1e-12, S)
    assert torch.equal(y_f, y_f2)            # fixed summation order: bitwise reproducible


@pytest.mark.parametrize("B,H,N_img,pos,beams", [(2, 2, 17, 0, 1), (3, 12, 197, 5, 1), (2, 12, 197, 7, 4), (1, 2, 300, 3, 3),
                                                  (1, 1, 1182, 11, 2), (2, 2, 40, 60, 4)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_attention_decode(B, H, N_img, pos, beams, dtype):
    """one new text position per row against [shared image K/V | per-beam text K/V through kv_src]"""
    from generativeimage2text_amd import engine as E
    d, R, T = H * 64, B * beams, max(pos + 1, 8) + 3
    g = torch.Generator().manual_seed(100 + N_img + pos)
    qkv = (torch.randn(R, 3 * d, generator=g) * 1.2).to(dtype)
    ik = torch.randn(B, H, N_img, 64, generator=g).to(dtype)
    iv = torch.randn(B, H, N_img, 64, generator=g).to(dtype)
    tk = torch.randn(R, T, d, generator=g).to(dtype)
    tv = torch.randn(R, T, d, generator=g).to(dtype)
    src = torch.stack([torch.randint(b * beams, (b + 1) * beams, (T,), generator=g) for b in range(B) for _ in range(beams)]).int()
    out = E.op_attn_decode(qkv.cuda(), ik.cuda(), iv.cuda(), tk.cuda().clone(), tv.cuda().clone(), src.cuda(), B, H, N_img, T, pos,
                           beams).cpu().double()
    ref = torch.zeros(R, d, dtype=torch.float64)
    for r in range(R):
        b = r // beams
        for h in range(H):
            q = qkv[r, h * 64:(h + 1) * 64].double() * 0.125
            ks = [ik[b, h].double()] + [tk[src[r, s], s, h * 64:(h + 1) * 64].double()[None] for s in range(pos)] + \
                 [qkv[r, d + h * 64: d + (h + 1) * 64].double()[None]]
            vs = [iv[b, h].double()] + [tv[src[r, s], s, h * 64:(h + 1) * 64].double()[None] for s in range(pos)] + \
                 [qkv[r, 2 * d + h * 64: 2 * d + (h + 1) * 64].double()[None]]
            Kc, Vc = torch.cat(ks), torch.cat(vs)
            p = torch.softmax(Kc @ q, 0)
            ref[r, h * 64:(h + 1) * 64] = p @ Vc
    tol = 3e-5 if dtype == torch.float32 else 3e-2
    assert (out - ref).abs().max().item() < tol
