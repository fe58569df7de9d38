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


@pytest.mark.usefixtures("experiment_build")
@pytest.mark.parametrize("M,N,K", [(1000, 512, 128), (777, 256, 192), (2100, 768, 768), (515, 1024, 3072)])
@pytest.mark.parametrize("act", [0, 1, 2])
def test_gemm_bf16_p8_variant(M, N, K, act):
    """The 256x256 half-tile pipeline kernel (kernels_gemm10.hip) with its tile height forced (measurement build): shortest K
    (2 and 3 K tiles), ragged M, every epilogue; the 224- / 192- / 160- / 128-row tiles (round 5: second half tile shorter than
    the first) and the 256-row tile must agree bit for bit (same K order), and all stay within summation-order error of the
    register-staged tile kernel (impl 0)."""
    from generativeimage2text_amd import engine as E
    A = _rand(M, K, seed=11).bfloat16()
    W = _rand(N, K, seed=12, scale=K ** -0.5).bfloat16()
    bias = _rand(N, seed=13)
    res = _rand(M, N, seed=14)
    ref = _act(A.double() @ W.double().t() + bias.double(), act)
    outs = {}
    try:
        for tile in (9 | (128 << 8), 9 | (64 << 8), 9 | (32768 << 8), 9 | (16384 << 8), 9 | (65536 << 8), 0):
            E.set_gemm_impl(tile)
            outs[tile] = (E.op_gemm(A.cuda(), W.cuda(), bias.cuda(), res.cuda(), act, torch.float32).cpu(),
                          E.op_gemm(A.cuda(), W.cuda(), bias.cuda(), None, act, torch.bfloat16).cpu(),
                          E.op_gemm(A.cuda(), W.cuda(), None, res.cuda(), act, torch.bfloat16).cpu())
    finally:
        E.set_gemm_impl(-1)
    ref_br = _act(A.double() @ W.double().t(), act) + res.double()
    for tile, (out, out_b, out_br) in outs.items():
        assert (out.double() - (ref + res.double())).abs().max().item() < 2e-4 * max(1.0, ref.abs().max().item()), tile
        assert (out_b.double() - ref).abs().max().item() < 1e-2 * max(1.0, ref.abs().max().item()), tile
        assert (out_br.double() - ref_br).abs().max().item() < 2e-2 * max(1.0, ref_br.abs().max().item()), tile
    hi = outs[9 | (128 << 8)]
    for bits in (64, 32768, 16384, 65536):
        assert all(torch.equal(a, b) for a, b in zip(hi, outs[9 | (bits << 8)])), bits


@pytest.mark.usefixtures("experiment_build")
@pytest.mark.parametrize("M,N,K", [(12608, 768, 768), (12608, 3072, 768), (8224, 1024, 4096), (6001, 2304, 768), (24272, 768, 3072)])
def test_gemm_p8_automatic_tile_height_equals_the_256_row_tile(M, N, K):
    """the tile height a context alone gets is picked by a time model (kernels_gemm10.hip: gemm_p8_cost) among
    256 / 224 / 192 / 160 / 128 rows.  Whatever it picks at the encoder's real (ragged) row counts -- 64 x 197, 32 x 257,
    16 x 6 x 197 ..., where the mixed-height instantiations (224 = 128 + 96, 160 = 96 + 64 rows) are what runs -- the result is
    the 256-row tile's, bit for bit, for the 16-bit and the residual-stream epilogues."""
    from generativeimage2text_amd import engine as E
    A = _rand(M, K, seed=21).bfloat16().cuda()
    W = _rand(N, K, seed=22, scale=K ** -0.5).bfloat16().cuda()
    bias = _rand(N, seed=23).cuda()
    res = _rand(M, N, seed=24).cuda()
    outs = {}
    try:
        for tile in (-1, 9 | (128 << 8)):
            E.set_gemm_impl(tile)
            outs[tile] = (E.op_gemm(A, W, bias, None, 1, torch.bfloat16).cpu(), E.op_gemm(A, W, bias, res, 0, torch.float32).cpu())
    finally:
        E.set_gemm_impl(-1)
    assert all(torch.equal(a, b) for a, b in zip(outs[-1], outs[9 | (128 << 8)]))
    ref = (A[:64].double() @ W.double().t() + bias.double()).cpu()
    ref = ref * torch.sigmoid(1.702 * ref)
    assert (outs[-1][0][:64].double() - ref).abs().max().item() < 1e-2 * max(1.0, ref.abs().max().item())


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


@pytest.mark.parametrize("rows,D", [(5, 768), (1000, 1024), (33, 128), (7, 192), (12608, 768), (8224, 1024)])
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


# ---- decode-step GEMM chain (kernels_dgemm.hip) ---------------------------------------------------------------
def _bf16_round(x):
    return x.bfloat16().float()


def _fold(W, bias, gamma, beta):
    """What gitmi_finalize_weights prepares for a GEMM behind a LayerNorm: W' = bf16(W . gamma), beta W^T + b, colsum(W')."""
    Wf = _bf16_round(W * gamma[None, :])
    return Wf.bfloat16(), (bias.double() + W.double() @ beta.double()).float(), Wf.double().sum(1).float()


@pytest.mark.parametrize("M,N,K", [(64, 2304, 768), (64, 3072, 768), (256, 3072, 768), (5, 130, 128), (33, 1002, 96),
                                   (17, 512, 128), (100, 2304, 768)])
@pytest.mark.parametrize("act", [0, 2])
@pytest.mark.parametrize("fold", [False, True])
def test_dgemm_qkv_ffn1_form(M, N, K, act, fold):
    """QKV / FFN1 form: plain, and with the LayerNorm in front folded into weights + epilogue (stats from strip partials)."""
    from generativeimage2text_amd import engine as E
    W = _rand(N, K, seed=22, scale=K ** -0.5)
    bias = _rand(N, seed=23)
    if not fold:
        A = _rand(M, K, seed=21).bfloat16()
        ref = _act(A.double() @ _bf16_round(W).double().t() + bias.double(), act)
        out = E.op_dgemm(A.cuda(), W.bfloat16().cuda(), bias.cuda(), act=act)
        assert (out.cpu().double() - ref).abs().max().item() < 1e-2 * max(1.0, ref.abs().max().item())
        if N % 32 == 0:      # fragment-major output (operand of the next chain GEMM) holds the same values
            assert torch.equal(E.op_dgemm(A.cuda(), W.bfloat16().cuda(), bias.cuda(), act=act, frag_out=True), out)
        if 32 < M <= 64 and N >= 1536:      # 2 / 4 / 6 strips per workgroup (serving policy) == one strip per workgroup, bit for bit
            for nst in (2, 4, 6):
                assert torch.equal(E.op_dgemm(A.cuda(), W.bfloat16().cuda(), bias.cuda(), act=act, strips_per_wg=nst), out)
        if M > 64:           # the row-walking kernel (> 64 rows) == the one-block kernel on the same rows, bit for bit
            hi = min(M, 128)
            sub = E.op_dgemm(A[64:hi].contiguous().cuda(), W.bfloat16().cuda(), bias.cuda(), act=act)
            assert torch.equal(sub, out[64:hi])
        return
    if K % 16:
        pytest.skip("strip partials need K % 16 == 0")
    x = _rand(M, K, seed=21, scale=1.3) + 0.2                      # raw rows: non-zero mean, non-unit variance
    gamma, beta = 1 + _rand(K, seed=24, scale=0.1), _rand(K, seed=25, scale=0.1)
    ref = _act(torch.nn.functional.layer_norm(x.double(), (K,), gamma.double(), beta.double(), 1e-12) @ W.double().t()
               + bias.double(), act)
    Wf, bf, cs = _fold(W, bias, gamma, beta)
    stats = E.strip_stats(x.cuda())
    out_dev = E.op_dgemm(x.bfloat16().cuda(), Wf.cuda(), bf.cuda(), cs.cuda(), stats, 1e-12, act)
    if 32 < M <= 64 and N >= 1536:
        for nst in (2, 4, 6):
            assert torch.equal(E.op_dgemm(x.bfloat16().cuda(), Wf.cuda(), bf.cuda(), cs.cuda(), stats, 1e-12, act,
                                          strips_per_wg=nst), out_dev)
    if M > 64:               # row-walking kernel vs one-block kernel, folded LayerNorm included
        hi = min(M, 128)
        sub = E.op_dgemm(x[64:hi].bfloat16().contiguous().cuda(), Wf.cuda(), bf.cuda(), cs.cuda(),
                         stats[:, 64:hi].contiguous(), 1e-12, act)
        assert torch.equal(sub, out_dev[64:hi])
    out = out_dev.cpu().double()
    # bf16 operands on both sides: error ~ 2^-8 relative to the output scale
    assert (out - ref).abs().max().item() < 2.5e-2 * max(1.0, ref.abs().max().item())
    # against the same arithmetic in fp64 (bf16 operands as the kernel sees them): only fp32 summation-order error + bf16 output rounding
    mean, var = x.double().mean(1, keepdim=True), x.double().var(1, unbiased=False, keepdim=True)
    exact = _act((x.bfloat16().double() @ Wf.double().t() - mean * cs.double()) / torch.sqrt(var + 1e-12) + bf.double(), act)
    assert (out - exact).abs().max().item() < 6e-3 * max(1.0, exact.abs().max().item())


@pytest.mark.parametrize("M,N,K", [(64, 768, 768), (64, 768, 3072), (256, 768, 3072), (7, 128, 512), (33, 128, 128), (100, 768, 768)])
@pytest.mark.parametrize("ln_res", [False, True])
def test_dgemm_residual_stats_form(M, N, K, ln_res):
    """N = hidden form: x = A W^T + bias + residual (the hidden state, or LayerNorm(previous raw x) rebuilt from its
    strip partials), plus the strip partials of x; fixed summation order => bitwise reproducible."""
    from generativeimage2text_amd import engine as E
    A = _rand(M, K, seed=31).bfloat16()
    W = _rand(N, K, seed=32, scale=K ** -0.5).bfloat16()
    bias, xprev = _rand(N, seed=33), _rand(M, N, seed=34, scale=1.2) + 0.1
    g, b = 1 + _rand(N, seed=35, scale=0.1), _rand(N, seed=36, scale=0.1)
    res = torch.nn.functional.layer_norm(xprev.double(), (N,), g.double(), b.double(), 1e-12) if ln_res else xprev.double()
    ref = A.double() @ W.double().t() + bias.double() + res
    args = (A.cuda(), W.cuda(), bias.cuda(), xprev.cuda())
    kw = dict(res_stats=E.strip_stats(xprev.cuda()), res_gamma=g.cuda(), res_beta=b.cuda()) if ln_res else {}
    x, xb, st = E.op_dgemm_res(*args, **kw)
    assert (x.cpu().double() - ref).abs().max().item() < 3e-4 * max(1.0, ref.abs().max().item())
    assert (xb.cpu().double() - ref).abs().max().item() < 1.2e-2 * max(1.0, ref.abs().max().item())
    want = E.strip_stats(x).cpu()
    assert (st.cpu() - want).abs().max().item() < 1e-3 * max(1.0, want.abs().max().item())
    x2, _, st2 = E.op_dgemm_res(*args, **kw)
    assert torch.equal(x, x2) and torch.equal(st, st2)
    # a row's result does not depend on its batch-mates
    x3, _, _ = E.op_dgemm_res(A[:3].contiguous().cuda(), W.cuda(), bias.cuda(), xprev[:3].contiguous().cuda(),
                              **({k: (v[:, :3].contiguous() if k == "res_stats" else v) for k, v in kw.items()}))
    assert torch.equal(x3, x[:3])


@pytest.mark.parametrize("M,V,K,mtop", [(64, 30522, 768, 1), (64, 30522, 768, 8), (256, 30522, 768, 8), (130, 30522, 768, 4),
                                        (5, 1000, 128, 4), (33, 1000, 128, 2), (16, 5003, 256, 16), (40, 999, 96, 8),
                                        (200, 1000, 128, 4), (7, 1000, 768, 16), (100, 2000, 768, 2)])
@pytest.mark.parametrize("fold", [False, True])
def test_vocab_head_fused_topm(M, V, K, mtop, fold):
    """Vocabulary head with running top-M / log-sum-exp: merging the per-column-block lists must give exactly the top-M
    and the log-softmax of the logits the same kernel materialises on request; those logits against fp64.  Then the same
    lists, bit for bit, from the kernels that run in the decode loop (no logits output: the one-row-block kernel with the
    lists parked in LDS, the row-block-walking kernel) for several grid sizes: one workgroup per column block, and fewer
    workgroups that WALK their column blocks with the rolling weight refill (an odd number of blocks per workgroup, an
    uneven split, 8 blocks per workgroup; V = 1000 / 999 end in a 7-strip tail block)."""
    from generativeimage2text_amd import engine as E
    cols = 128
    W = _rand(V, K, seed=41, scale=K ** -0.5 * 2.0)
    bias = _rand(V, seed=42, scale=0.5)
    x = _rand(M, K, seed=43, scale=1.1) + 0.15
    sup = torch.randint(0, V, (M,), generator=torch.Generator().manual_seed(44), dtype=torch.int32)
    if fold:
        gamma, beta = 1 + _rand(K, seed=45, scale=0.1), _rand(K, seed=46, scale=0.1)
        ref = torch.nn.functional.layer_norm(x.double(), (K,), gamma.double(), beta.double(), 1e-12) @ W.double().t() + bias.double()
        Wf, bf, cs = _fold(W, bias, gamma, beta)
        if K % 16:
            pytest.skip("strip partials need K % 16 == 0")
        pv, pi, pl, lg = E.op_vocab_topm(x.bfloat16().cuda(), Wf.cuda(), bf.cuda(), mtop, cols, cs.cuda(), E.strip_stats(x.cuda()),
                                         1e-12, sup.cuda(), True)
    else:
        ref = x.bfloat16().double() @ _bf16_round(W).double().t() + bias.double()
        pv, pi, pl, lg = E.op_vocab_topm(x.bfloat16().cuda(), W.bfloat16().cuda(), bias.cuda(), mtop, cols, suppress_tok=sup.cuda(),
                                         want_logits=True)
    lg = lg.cpu()
    tol = (2.5e-2 if fold else 2e-4) * max(1.0, ref.abs().max().item())
    assert (lg.double() - ref).abs().max().item() < tol
    # the lists: built from the suppressed logits (decoder.py:330)
    sl = lg.clone()
    sl[torch.arange(M), sup.long()] = -10000.0
    pv, pi, pl = pv.cpu(), pi.cpu().long(), pl.cpu()
    nparts = pv.shape[1]
    for p in range(nparts):
        blk = sl[:, p * cols:(p + 1) * cols]
        k = min(mtop, blk.shape[1])
        tv, ti = blk.topk(k, dim=1)
        assert torch.equal(pv[:, p, :k], tv), p
        # ties may be broken either way only if values are equal: compare through the values the indices point at
        assert torch.equal(torch.gather(sl, 1, pi[:, p, :k]), tv), p
        assert torch.allclose(pl[:, p, 0], blk.max(1).values)
        assert torch.allclose(pl[:, p, 1], torch.exp(blk - blk.max(1, keepdim=True).values).sum(1), rtol=2e-5)
    lse = torch.logsumexp(sl.double(), 1)
    got = torch.log((pl[:, :, 1].double() * torch.exp(pl[:, :, 0].double() - pl[:, :, 0].double().max(1, keepdim=True).values)).sum(1)) \
        + pl[:, :, 0].double().max(1).values
    assert (got - lse).abs().max().item() < 1e-4
    # the decode-loop kernels (no logits output), every grid size
    for max_wgs in (0, 1000, (nparts + 2) // 3, max(1, nparts - 1), (nparts + 7) // 8, 60):
        if fold:
            qv, qi, ql, _ = E.op_vocab_topm(x.bfloat16().cuda(), Wf.cuda(), bf.cuda(), mtop, cols, cs.cuda(), E.strip_stats(x.cuda()),
                                            1e-12, sup.cuda(), False, max_wgs=max_wgs)
        else:
            qv, qi, ql, _ = E.op_vocab_topm(x.bfloat16().cuda(), W.bfloat16().cuda(), bias.cuda(), mtop, cols, suppress_tok=sup.cuda(),
                                            want_logits=False, max_wgs=max_wgs)
        k = min(mtop, cols)
        assert torch.equal(qv.cpu()[:, :, :k], pv[:, :, :k]), max_wgs
        assert torch.equal(qi.cpu().long()[:, :, :k], pi[:, :, :k]), max_wgs
        assert torch.equal(ql.cpu(), pl), max_wgs


@pytest.mark.parametrize("M", [8, 70, 256])
@pytest.mark.parametrize("mtop", [2, 8])
def test_vocab_head_topm_tie_order(M, mtop):
    """Equal logits: every list of the fused head is ordered by (value descending, column ascending) -- the order the
    positional insertion (kernels_dgemm.hip: topm_insert; round 5) must share with the bubble insertion it replaced and with
    the 4-lane merge -- in all three kernel forms (logits-materialising, one-row-block, row-block-walking) and for walked
    column blocks.  Integer-valued operands make the logits exact integers with hundreds of ties per 128-column block."""
    from generativeimage2text_amd import engine as E
    V, K, cols = 1000, 768, 128
    g = torch.Generator().manual_seed(7)
    x = torch.randint(-2, 3, (M, K), generator=g).float()
    W = torch.randint(-1, 2, (V, K), generator=g).float()
    bias = torch.zeros(V)
    ref = x.double() @ W.double().t()                                  # exact integers, |.| << 2^24
    assert max(r.unique().numel() for r in ref) < V // 4               # every row: plenty of ties
    order = torch.sort(ref, dim=1, descending=True, stable=True)       # stable: the lower column first among equals
    for want_logits, max_wgs in ((True, 0), (False, 0), (False, 3), (False, 60)):
        pv, pi, pl, lg = E.op_vocab_topm(x.bfloat16().cuda(), W.bfloat16().cuda(), bias.cuda(), mtop, cols, want_logits=want_logits,
                                         max_wgs=max_wgs)
        if want_logits:
            assert torch.equal(lg.cpu().double(), ref)
        pv, pi = pv.cpu(), pi.cpu().long()
        for p in range(pv.shape[1]):
            blk = ref[:, p * cols:(p + 1) * cols]
            o = torch.sort(blk, dim=1, descending=True, stable=True)
            k = min(mtop, blk.shape[1])
            assert torch.equal(pv[:, p, :k].double(), o.values[:, :k]), (want_logits, max_wgs, p)
            assert torch.equal(pi[:, p, :k], o.indices[:, :k] + p * cols), (want_logits, max_wgs, p)
    del order


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
    if dtype == torch.bfloat16:
        # the kernel forms of the bf16 path (gitmi_op_attn_decode `dbg` bits 16..): one wave per pair with the K/V chunk in
        # registers, and the streaming kernel (K/V through an LDS ring; each wave walks several pairs) on 1, 2 and 96
        # workgroups -- the streaming kernel repeats the one-wave arithmetic operation for operation: bitwise equal
        def run(form):
            return E.op_attn_decode(qkv.cuda(), ik.cuda(), iv.cuda(), tk.cuda().clone(), tv.cuda().clone(), src.cuda(), B, H, N_img,
                                    T, pos, beams, dbg=form).cpu()
        one = run(1 << 16)
        assert (one.double() - ref).abs().max().item() < tol
        for wgs in (1, 2, 96):
            st = run(wgs << 18)
            assert torch.equal(st, one), (wgs, (st.double() - one.double()).abs().max().item())


# ---- sampling branch (decoder.py:1146-1166, 1343-1375) -------------------------------------------------------------
@pytest.mark.parametrize("top_k,top_p,temp", [(0, 0.9, 1.0), (50, 1.0, 1.0), (20, 0.7, 1.3), (5, 0.3, 0.7), (0, 0.05, 1.0),
                                              (3, 0.999, 1.0), (1, 0.5, 2.0), (0, 1.0, 1.0)])
def test_sampling_filter_equals_reference_filter(top_k, top_p, temp):
    """The device filter (thresholds found by bisection, no sort) keeps exactly the tokens top_k_top_p_filtering keeps
    (the oracle restatement, pinned against the reference's function on the same seeded logits), except where a token
    sits within fp32 rounding of the nucleus boundary; the draws come from the kept set with the filtered log-probs."""
    from generativeimage2text_amd import engine as E
    from oracle import git_oracle as O
    logits = _rand(6, 3000, seed=3, scale=3.0)
    want = O.sampling_distribution(logits, temp, top_k, top_p)
    filt, tok, lp = E.op_sample_rows(logits.cuda(), temp, top_k, top_p, ndraw=2, seed=5, step=1)
    filt, tok, lp = filt.cpu(), tok.cpu().long(), lp.cpu()
    kept_ref, kept = torch.isfinite(want), torch.isfinite(filt)
    diff = kept_ref != kept
    if diff.any():
        # only tokens whose "mass before me" is within rounding of top_p may differ
        srt, idx = torch.sort(logits / temp, descending=True)
        cum = torch.cumsum(torch.softmax(srt.double(), -1), -1)
        before = torch.zeros_like(cum)
        before[:, 1:] = cum[:, :-1]
        pos = torch.argsort(idx, dim=1)
        assert (torch.gather(before, 1, pos)[diff] - top_p).abs().max().item() < 1e-5
        assert diff.sum().item() <= 2
    same = kept & kept_ref
    got_lp = torch.log_softmax(filt, -1)
    assert (got_lp[same] - want[same]).abs().max().item() < (1e-4 if not diff.any() else 1e-2)
    assert torch.equal(filt[kept], (logits / temp)[kept]) or torch.allclose(filt[kept], (logits / temp)[kept], rtol=1e-6)
    # draws: distinct tokens of the kept set, log-probabilities of the filtered softmax
    assert (tok[:, 0] != tok[:, 1]).all()
    assert kept.gather(1, tok).all()
    assert (got_lp.gather(1, tok) - lp).abs().max().item() < 1e-4


def test_sampling_draws_follow_the_filtered_distribution():
    """First draws over many independent steps: chi-square against the filtered softmax; second draw != first
    (without replacement); same (seed, step) -> same draws."""
    from generativeimage2text_amd import engine as E
    from oracle import git_oracle as O
    row = torch.tensor([[2.0, 1.5, 1.0, 0.5, 0.0, -0.5, -1.0, -3.0, 0.2, 0.7]])
    R = 4096
    logits = row.repeat(R, 1).cuda()
    counts = torch.zeros(10)
    for step in range(1, 6):
        _, tok, _ = E.op_sample_rows(logits, 1.0, 8, 1.0, ndraw=2, seed=123, step=step, want_filtered=False)
        assert (tok[:, 0] != tok[:, 1]).all()
        counts += torch.bincount(tok[:, 0].cpu().long(), minlength=10).float()
    p = torch.exp(O.sampling_distribution(row, 1.0, 8, 1.0))[0]
    n = counts.sum().item()
    exp = p * n
    on = exp > 0
    chi2 = (((counts - exp) ** 2)[on] / exp[on]).sum().item()
    assert counts[~on].sum().item() == 0                     # filtered tokens are never drawn
    assert chi2 < 30.0, (chi2, counts.tolist(), exp.tolist())     # 7 degrees of freedom: P(chi2 > 30) ~ 1e-4
    _, a, _ = E.op_sample_rows(logits, 1.0, 8, 1.0, ndraw=2, seed=123, step=3, want_filtered=False)
    _, b, _ = E.op_sample_rows(logits, 1.0, 8, 1.0, ndraw=2, seed=123, step=3, want_filtered=False)
    _, c, _ = E.op_sample_rows(logits, 1.0, 8, 1.0, ndraw=2, seed=124, step=3, want_filtered=False)
    assert torch.equal(a, b) and not torch.equal(a, c)
