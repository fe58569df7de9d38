"""GPU: the kernel-shape policy of gitmi_set_shared_device never changes results (include/gitmi.h)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("precision", ["bf16", "f16", "f32"])
def test_shared_device_policy_is_bitwise_neutral(precision):
    """gitmi_set_shared_device changes kernel SHAPES (256-row GEMM tiles everywhere, two (sentence, head) pairs per
    attention workgroup), never results: features, ids and log-probs of the benchmark geometry are bit-identical, for
    a context and for a clone that inherits the setting.  This is the guard of include/gitmi.h's promise for
    gitmi_set_shared_device -- in particular that the streaming decode attention (solo policy) and the register form (serving
    policy) agree bit for bit (`fp contract(off)` + explicit fmaf in both, kernels_attn_decode.hip) -- in every operand build."""
    from generativeimage2text_amd.configs import config_for_model
    from generativeimage2text_amd.engine import Engine
    from generativeimage2text_amd.synthetic import random_frames, random_state_dict
    cfg = config_for_model("GIT_BASE")
    B = 8 if precision == "f32" else 64
    eng = Engine(cfg, precision=precision, max_batch=B, max_beams=4, max_frames=1, max_text_len=20)
    eng.load_state_dict(random_state_dict(cfg, seed=1234))
    frames = random_frames(cfg, B, 1, seed=0)
    out = {}
    for search in (Engine.make_search("greedy", 20, 1, 1), Engine.make_search("beam", 20, 4, 2, 0.6)):
        for on in (False, True):
            eng.set_shared_device(on)
            feats = eng.encode(frames, return_features=True)
            tok, lp, _ = eng.generate(frames, search)
            out[on] = (feats.clone(), tok.clone(), lp.clone())
        clone = eng.clone()                                  # inherits "on"
        tok_c, lp_c, _ = clone.generate(frames, search)
        clone.close()
        assert torch.equal(out[False][0], out[True][0])
        assert torch.equal(out[False][1], out[True][1]) and torch.equal(out[False][2], out[True][2])
        assert torch.equal(tok_c, out[True][1]) and torch.equal(lp_c, out[True][2])
    eng.close()
