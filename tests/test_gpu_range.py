"""GPU: the 16-bit builds fail LOUDLY when a checkpoint does not fit their operand format (the fp16 build is the default
16-bit mode; f2bf / pack2bf are plain casts).  Three layers of defence, each tested here through the C ABI:
  * gitmi_load_tensor rejects a tensor that holds inf / NaN, naming it;
  * gitmi_finalize_weights rejects, in the fp16 build, a matrix (or a decoder matrix with its LayerNorm gain folded in) whose
    max |w| exceeds 65504 -- the bf16 build and the f32 mode accept it;
  * an ACTIVATION that overflows at run time turns its sentence's log-prob into NaN (inf -> LayerNorm statistics -> softmax ->
    log-sum-exp); gitmi_generate counts such sequences in info[3] and the binding raises instead of returning garbage ids.
The trained-statistics goldens (tests/test_gpu_parity.py, full_trained_*) show the other side: LayerNorm gains up to 5, biases of
order 1 and residual channels 1000x above the rest stay finite and inside the specification in the fp16 build."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _tiny(seed=12):
    from oracle import git_oracle as O
    cfg = O.CONFIGS["TINY"]
    return O, cfg, O.make_weights(cfg, seed=seed, tie_output=False, eos_bias=1.0)


def _engine(cfg, prec):
    from generativeimage2text_amd.engine import Engine
    return Engine(cfg, precision=prec, max_batch=4, max_beams=1, max_frames=1, max_text_len=12)


@pytest.mark.parametrize("prec", ["f32", "bf16", "f16"])
def test_non_finite_weight_is_rejected_by_name(prec):
    from generativeimage2text_amd.engine import GitmiError
    O, cfg, w = _tiny()
    key = "textual.transformer.encoder.layer.1.intermediate.dense.weight"
    w = dict(w)
    w[key] = w[key].clone()
    w[key][3, 5] = float("nan")
    eng = _engine(cfg, prec)
    with pytest.raises(GitmiError, match="intermediate.dense.weight.*non-finite"):
        eng.load_state_dict(w)
    eng.close()


def test_weight_outside_fp16_range_is_rejected_by_the_fp16_build_only():
    from generativeimage2text_amd.engine import Engine, GitmiError
    O, cfg, w = _tiny()
    frames = [f.cuda() for f in O.make_images(cfg, 2, 1, seed=3)]
    w = dict(w)
    key = "image_encoder.transformer.resblocks.0.mlp.c_fc.weight"
    w[key] = w[key].clone()
    w[key][0, 0] = 1.0e5
    eng = _engine(cfg, "f16")
    with pytest.raises(GitmiError, match="c_fc.weight.*fp16 operand range"):
        eng.load_state_dict(w)
    eng.close()
    for prec in ("bf16", "f32"):                       # same exponent range as fp32: loads and runs
        eng = _engine(cfg, prec)
        eng.load_state_dict(w)
        eng.close()
    # a LayerNorm gain folded into a decoder matrix (W . gamma) can leave the range although W and gamma are both inside it
    O, cfg, w = _tiny()
    w = dict(w)
    w["textual.transformer.encoder.layer.0.attention.output.LayerNorm.weight"] = torch.full((cfg.dec_hidden,), 3.0e4)
    k2 = "textual.transformer.encoder.layer.0.intermediate.dense.weight"
    w[k2] = w[k2].clone()
    w[k2][1, 1] = 10.0
    eng = _engine(cfg, "f16")
    with pytest.raises(GitmiError, match="fp16 operand range"):
        eng.load_state_dict(w)
    eng.close()


def test_activation_overflow_raises_instead_of_returning_garbage():
    from generativeimage2text_amd.engine import Engine, GitmiError
    O, cfg, w = _tiny()
    frames = [f.cuda() for f in O.make_images(cfg, 3, 1, seed=3)]
    w = dict(w)
    key = "image_encoder.transformer.resblocks.0.mlp.c_fc.weight"
    w[key] = w[key] * 1.0e5                            # weights ~ 1e5 * width^-0.5 ~ 9e3: inside fp16; the MLP's hidden units (~1e5) are not
    search = Engine.make_search("greedy", 12, 1, 1)
    eng = _engine(cfg, "f32")                          # the exact mode computes the same model without trouble
    eng.load_state_dict(w)
    tokens, lps, info = eng.generate(frames, search)
    assert torch.isfinite(lps).all() and info.tolist()[3] == 0
    eng.close()
    eng = _engine(cfg, "f16")
    eng.load_state_dict(w)
    with pytest.raises(GitmiError, match="non-finite log-probability"):
        eng.generate(frames, search)
    # the asynchronous form hands the flag back for the caller to check
    tokens, lps, info = eng.generate(frames, search, sync=False)
    torch.cuda.synchronize()
    assert info.tolist()[3] == 3
    with pytest.raises(GitmiError):
        eng.check_finite(info)
    eng.close()
