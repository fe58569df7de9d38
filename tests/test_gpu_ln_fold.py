"""GPU: LayerNorm folding of the image encoder and the prefill (gitmi_set_ln_fold, fp16-operand library; round 6).

The folded form replaces 35 of the 36 LayerNorm launches of a GIT_BASE request (CLIP/model.py:161-168, 189-202 ln_1 / ln_2;
modeling_bert.py:171-178, 243-250 and decoder.py:35 over the image rows) by epilogue arithmetic of the GEMMs either side of
them.  It must (a) stay inside the tolerances the unfolded form is held to, against the f32 engine mode (itself pinned to the
reference's frozen tensors by tests/test_gpu_parity.py), on random-init AND trained-statistics weights, (b) not depend on tile
heights, (c) leave small batches (<= 512 rows: LayerNorm launches, plain matrices) untouched, (d) fail loudly where the
library cannot fold."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _weights(cfg, trained):
    from generativeimage2text_amd.synthetic import apply_trained_statistics, random_state_dict
    sd = random_state_dict(cfg, seed=1234)
    if trained:
        apply_trained_statistics(cfg, sd, seed=77)
    return sd


@pytest.mark.parametrize("model,B,frames", [("GIT_BASE", 16, 1), ("GIT_LARGE", 8, 1), ("GIT_BASE_VATEX", 4, 6)])
@pytest.mark.parametrize("trained", [False, True])
def test_folded_and_unfolded_forms_agree_with_the_f32_engine(model, B, frames, trained):
    """Features and teacher-forced logits of the f16 engine with the fold on / off against the f32 engine on the same weights
    and images: both inside the same bound (features 4e-3 of the largest feature magnitude; logits 1e-3 of the f32 logit span
    -- north_star's constant), and the folded form not worse than 1.5 x the unfolded one (measured: equal within +-7 %)."""
    from generativeimage2text_amd.configs import config_for_model
    from generativeimage2text_amd.engine import Engine
    from generativeimage2text_amd.synthetic import random_frames
    cfg = config_for_model(model)
    sd = _weights(cfg, trained)
    fr = random_frames(cfg, B, frames, seed=3)
    toks = torch.randint(1000, 20000, (B, 6), device="cuda")
    toks[:, 0] = 101

    def run(precision, fold):
        eng = Engine(cfg, precision=precision, max_batch=B, max_beams=1, max_frames=frames, max_text_len=20)
        eng.load_state_dict(sd)
        if fold is not None:
            eng.set_ln_fold(fold)
        feats = eng.encode(fr, return_features=True).clone()
        eng.prefill()
        logits = eng.step_logits(toks).clone()
        eng.close()
        return feats, logits

    f_ref, l_ref = run("f32", None)
    span = (l_ref.max() - l_ref.min()).item()
    fmax = f_ref.abs().max().item()
    err = {}
    for fold in (False, True):
        f, l = run("f16", fold)
        assert torch.isfinite(f).all() and torch.isfinite(l).all()
        err[fold] = ((f - f_ref).abs().max().item(), (l - l_ref).abs().max().item())
    # ViT features do not depend on the fold only if the ln_post input agrees: both forms within the feature bound
    for fold in (False, True):
        assert err[fold][0] < 4e-3 * fmax, (model, trained, fold, err, fmax)        # measured 0.8e-3 ... 2.4e-3 of max |feature|
        assert err[fold][1] <= 1e-3 * span, (model, trained, fold, err, span)
    assert err[True][1] <= 1.5 * err[False][1] + 1e-4 * span, (model, trained, err, span)
    from test_gpu_parity import record_measurement
    record_measurement(case=f"ln_fold_{model}_b{B}", trained=trained, feat_err_unfolded=err[False][0], feat_err_folded=err[True][0],
                       logit_err_unfolded=err[False][1], logit_err_folded=err[True][1], logit_span=span, feature_max=fmax)


def test_folded_ids_equal_unfolded_ids_on_the_wide_margin_fixture():
    """full_wide_b64_greedy (every fp32 decision margin >= 0.2): both forms return the reference's ids on 64 of 64 rows."""
    from conftest import load_golden
    from test_gpu_parity import MG, format_like_reference, make_engine, search_struct
    name = "full_wide_b64_greedy"
    g = load_golden(name)
    cfg, w, frames, search, _ = MG.full_case_inputs(name)
    B = frames[0].shape[0]
    eng = make_engine(cfg, w, "f16", B, search, frames=len(frames))
    dev = [f.cuda() for f in frames]
    for fold in (True, False):
        eng.set_ln_fold(fold)
        tokens, logprobs, info = eng.generate(dev, search_struct(search))
        preds, _ = format_like_reference(search, tokens, logprobs, info, None)
        assert (preds.numpy() == g["predictions"]).all(), fold
    eng.close()


def test_small_batches_keep_the_layernorm_launches_and_large_ones_do_not_depend_on_batch_size():
    """M = B x 197 <= 512 rows runs the plain matrices + LayerNorm launches whatever the switch says (bit-identical with the
    switch on and off); above that the folded pass gives every image the same features whether it is encoded in a batch of 3
    or of 16 (row statistics are per row: no cross-row coupling, tile height irrelevant)."""
    from generativeimage2text_amd.configs import config_for_model
    from generativeimage2text_amd.engine import Engine
    from generativeimage2text_amd.synthetic import random_frames
    cfg = config_for_model("GIT_BASE")
    eng = Engine(cfg, precision="f16", max_batch=16, max_beams=1, max_frames=1, max_text_len=20)
    eng.load_state_dict(_weights(cfg, True))
    fr = random_frames(cfg, 16, 1, seed=5)
    small = [f[:2] for f in fr]
    eng.set_ln_fold(True)
    a = eng.encode(small, return_features=True).clone()
    eng.set_ln_fold(False)
    b = eng.encode(small, return_features=True).clone()
    assert torch.equal(a, b)
    eng.set_ln_fold(True)
    big = eng.encode(fr, return_features=True).clone()
    three = eng.encode([f[:3] for f in fr], return_features=True).clone()
    assert torch.equal(big[:3], three)
    eng.set_shared_device(True)              # 256-row tiles everywhere instead of the modelled heights
    big_s = eng.encode(fr, return_features=True).clone()
    assert torch.equal(big, big_s)
    eng.close()


@pytest.mark.parametrize("precision", ["bf16", "f32"])
def test_engines_that_cannot_fold_say_so(precision):
    from generativeimage2text_amd.configs import config_for_model
    from generativeimage2text_amd.engine import Engine, GitmiError
    cfg = config_for_model("GIT_BASE")
    eng = Engine(cfg, precision=precision, max_batch=2, max_beams=1, max_frames=1, max_text_len=8)
    eng.set_ln_fold(False)                   # always allowed
    with pytest.raises(GitmiError, match="gitmi_set_ln_fold"):
        eng.set_ln_fold(True)
    eng.close()


# ---- the two epilogue forms as single launches (gitmi_op_gemm_ln) against fp64 torch ----------------------------------------
def _tile_partials(rows_f16):
    """(sum, sumsq) of fp16 rows per 256-column tile, laid out [M, 4, 2] like the kernels exchange them"""
    x = rows_f16.double()
    M, N = x.shape
    out = torch.zeros(M, 4, 2, dtype=torch.float64)
    for t in range(N // 256):
        blk = x[:, t * 256:(t + 1) * 256]
        out[:, t, 0] = blk.sum(1)
        out[:, t, 1] = (blk * blk).sum(1)
    return out


def _stream_rows(M, N, seed, outliers=True, offset=0.0):
    """raw residual-stream rows with what makes folding delicate: a mean comparable to the spread (offset: a common offset many
    times the spread), a different scale per row, a few channels two to three orders above the rest"""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(M, N, generator=g) * (0.5 + 2.0 * torch.rand(M, 1, generator=g)) + torch.randn(M, 1, generator=g) * 1.5
    if offset:
        x = x + offset * torch.sign(torch.randn(M, 1, generator=g))
    if outliers:
        x[:, 17] += 300.0
        x[:, 401] -= 100.0
        x[:, N - 3] += 1000.0
    return x.half()


@pytest.mark.parametrize("M,N,K,act", [(1000, 2304, 768, 0), (2100, 3072, 768, 1), (777, 3072, 768, 2), (1500, 4096, 1024, 1)])
@pytest.mark.parametrize("outliers", [False, True])
def test_op_consumer_gemm_with_folded_layernorm_matches_fp64(M, N, K, act, outliers):
    """C = act(LayerNorm_K(X) W^T + b) from the raw fp16 rows, the folded weight set and the row partials, against fp64 torch on
    the same fp16 inputs: error bounded by the fp16 rounding of the OUTPUT plus the weight rounding of W . gamma (the operand
    rows are exact: they ARE the stored stream)."""
    from generativeimage2text_amd import engine as E
    from test_gpu_ops import _act, _rand
    X = _stream_rows(M, K, seed=31, outliers=outliers)
    W0 = _rand(N, K, seed=32, scale=K ** -0.5)
    b0 = _rand(N, seed=33, scale=0.1)
    gamma = torch.exp(_rand(K, seed=34) * 0.6)
    if outliers:
        gamma[[17, 401, K - 3]] = 0.2                # a trained model scales its outlier channels down
    beta = _rand(K, seed=35)
    Wf = (W0 * gamma).half()
    colsum = Wf.float().sum(1)
    bias_f = (b0.double() + W0.double() @ beta.double()).float()
    part = _tile_partials(X).float()
    out = E.op_gemm_ln(X.cuda(), Wf.cuda(), bias_f.cuda(), colsum=colsum.cuda(), ln_part=part.cuda(), ln_eps=1e-5, act=act).cpu().double()
    xd = X.double()
    mean = xd.mean(1, keepdim=True)
    var = ((xd - mean) ** 2).mean(1, keepdim=True)
    ln = (xd - mean) / torch.sqrt(var + 1e-5)
    ref_folded = _act(ln @ Wf.double().t() + bias_f.double(), act)           # same rounded matrix: kernel arithmetic only
    ref_exact = _act((ln * gamma.double() + beta.double()) @ W0.double().t() + b0.double(), act)
    scale = max(1.0, ref_exact.abs().max().item())
    assert torch.isfinite(out).all()
    assert (out - ref_folded).abs().max().item() < 1.5e-3 * scale          # fp16 output rounding + fp32 accumulation
    assert (out - ref_exact).abs().max().item() < 4e-3 * scale             # + the fp16 rounding of W . gamma


@pytest.mark.parametrize("M,N,K", [(1000, 768, 768), (2100, 768, 3072), (777, 1024, 4096)])
@pytest.mark.parametrize("mode", ["plain", "residual", "post_norm"])
def test_op_producer_gemm_stream_rows_and_partials_match_fp64(M, N, K, mode):
    """Stream rows C = A W^T + b (+ residual rows | + LayerNorm_N(raw residual rows) rebuilt from their partials) and the
    (sum, sumsq) of the rows AS STORED per 256-column tile."""
    from generativeimage2text_amd import engine as E
    from test_gpu_ops import _rand
    A = _rand(M, K, seed=41).half()
    W = _rand(N, K, seed=42, scale=K ** -0.5).half()
    b = _rand(N, seed=43, scale=0.1)
    R = _stream_rows(M, N, seed=44)
    kw = {}
    ref = A.double() @ W.double().t() + b.double()
    if mode == "residual":
        kw = dict(residual=R.cuda())
        ref = ref + R.double()
    elif mode == "post_norm":
        g2 = torch.exp(_rand(N, seed=45) * 0.5)
        b2 = _rand(N, seed=46, scale=0.5)
        kw = dict(residual=R.cuda(), res_part=_tile_partials(R).float().cuda(), res_gamma=g2.cuda(), res_beta=b2.cuda(), res_eps=1e-12)
        rd = R.double()
        mu = rd.mean(1, keepdim=True)
        ref = ref + (rd - mu) / torch.sqrt(((rd - mu) ** 2).mean(1, keepdim=True) + 1e-12) * g2.double() + b2.double()
    out, part = E.op_gemm_ln(A.cuda(), W.cuda(), b.cuda(), want_part=True, **kw)
    out, part = out.cpu(), part.cpu().double()
    scale = max(1.0, ref.abs().max().item())
    assert (out.double() - ref).abs().max().item() < 1.5e-3 * scale
    want = _tile_partials(out)                                               # statistics of the values as STORED
    assert torch.allclose(part[:, :, 0], want[:, :, 0], rtol=1e-5, atol=1e-3 * scale)
    assert torch.allclose(part[:, :, 1], want[:, :, 1], rtol=1e-5, atol=1e-3 * scale * scale)
    assert (part[:, N // 256:] == 0).all()


def test_op_consumer_gemm_rows_with_a_mean_many_times_their_spread():
    """The cancellation rstd (x W'^T - mean colsum) at |mean| / sigma ~ 10 ... 60 (a common offset of 30 on rows of spread
    0.5 ... 2.5): the fp32 accumulator carries products ~30x larger than the result; the error against fp64 on the same fp16
    inputs stays at the output-rounding level."""
    from generativeimage2text_amd import engine as E
    from test_gpu_ops import _rand
    M, N, K = 1000, 2304, 768
    X = _stream_rows(M, K, seed=51, outliers=False, offset=30.0)
    W0 = _rand(N, K, seed=52, scale=K ** -0.5)
    b0 = _rand(N, seed=53, scale=0.1)
    gamma = torch.exp(_rand(K, seed=54) * 0.6)
    beta = _rand(K, seed=55)
    Wf = (W0 * gamma).half()
    bias_f = (b0.double() + W0.double() @ beta.double()).float()
    out = E.op_gemm_ln(X.cuda(), Wf.cuda(), bias_f.cuda(), colsum=Wf.float().sum(1).cuda(), ln_part=_tile_partials(X).float().cuda(),
                       ln_eps=1e-5).cpu().double()
    xd = X.double()
    mean = xd.mean(1, keepdim=True)
    ln = (xd - mean) / torch.sqrt(((xd - mean) ** 2).mean(1, keepdim=True) + 1e-5)
    ref = ln @ Wf.double().t() + bias_f.double()
    assert (mean.abs() / xd.std(1, keepdim=True)).median().item() > 10
    assert (out - ref).abs().max().item() < 2e-3 * max(1.0, ref.abs().max().item())


def test_op_gemm_ln_refuses_what_it_cannot_run():
    from generativeimage2text_amd import engine as E
    A = torch.zeros(300, 768, dtype=torch.float16, device="cuda")           # <= 512 rows: not gemm_p8_kernel
    W = torch.zeros(768, 768, dtype=torch.float16, device="cuda")
    with pytest.raises(E.GitmiError, match="gemm_p8_kernel only"):
        E.op_gemm_ln(A, W, None, want_part=True)
