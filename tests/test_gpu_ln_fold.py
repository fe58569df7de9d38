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
    and images: both inside the same bound (features 0.02 absolute; logits 1e-3 of the f32 logit span -- north_star's
    constant), and the folded form not worse than 1.5 x the unfolded one."""
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
    err = {}
    for fold in (False, True):
        f, l = run("f16", fold)
        assert torch.isfinite(f).all() and torch.isfinite(l).all()
        err[fold] = ((f - f_ref).abs().max().item(), (l - l_ref).abs().max().item())
    # ViT features do not depend on the fold only if the ln_post input agrees: both forms within the feature bound
    for fold in (False, True):
        assert err[fold][0] < 0.02, (model, trained, fold, err)
        assert err[fold][1] <= 1e-3 * span, (model, trained, fold, err, span)
    assert err[True][1] <= 1.5 * err[False][1] + 1e-4 * span, (model, trained, err, span)
    from test_gpu_parity import record_measurement
    record_measurement(case=f"ln_fold_{model}_b{B}", trained=trained, feat_err_unfolded=err[False][0], feat_err_folded=err[True][0],
                       logit_err_unfolded=err[False][1], logit_err_folded=err[True][1], logit_span=span)


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
