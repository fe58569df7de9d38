"""GPU: the two storage types of the residual streams in bf16 engine mode.  fp16 is the default (the ViT / prefill
residual streams stored in fp16 instead of fp32 -- half the bytes of their read-modify-writes; +3.6 % captions/s, same
ids, profiles/r03_a_bench_f16_*.json); GITMI_STREAM_F16=0 keeps them in fp32 -- in the MEASUREMENT build (libgitmi_exp.so:
the product libraries read no environment, their streams are always fp16).  Both must meet the same fixed bounds
(tests/test_gpu_parity.py::check_bf16): this file runs the parity cases with the NON-default fp32 stream and compares the
two directly."""
import os

import numpy as np
import pytest
import torch

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("experiment_build")]


@pytest.fixture()
def stream32(monkeypatch):
    monkeypatch.setenv("GITMI_STREAM_F16", "0")
    yield
    monkeypatch.delenv("GITMI_STREAM_F16", raising=False)


CASES = ["tiny_greedy_long", "tiny_beam4", "tiny_video_beam4", "tinyl_greedy", "tiny_varres_up", "base_greedy",
         "base_prefix_beam4", "large_greedy", "vatex_greedy"]


@pytest.mark.parametrize("name", CASES)
def test_bf16_with_fp32_stream_within_tolerance(name, stream32):
    from test_gpu_parity import check_bf16
    check_bf16(name)


def test_fp16_stream_features_close_to_fp32_stream(monkeypatch):
    """Same weights and images through both stream precisions: the features differ by far less than the bf16 budget
    (prediction: ~0.003 max on unit-variance outputs), and the f32 engine mode ignores the switch."""
    from oracle import git_oracle as O
    from generativeimage2text_amd.engine import Engine
    cfg = O.CONFIGS["GIT_BASE"]
    w = O.make_weights(cfg, seed=1234)
    frames = [f.cuda() for f in O.make_images(cfg, 4, 1, seed=0)]

    def feats(prec):
        eng = Engine(cfg, precision=prec, max_batch=4, max_beams=1, max_frames=1, max_text_len=8)
        eng.load_state_dict(w)
        out = eng.encode(frames).cpu()
        eng.close()
        return out
    monkeypatch.setenv("GITMI_STREAM_F16", "1")
    f16s, f32mode_on = feats("bf16"), feats("f32")
    monkeypatch.setenv("GITMI_STREAM_F16", "0")
    f32s, f32mode_off = feats("bf16"), feats("f32")
    monkeypatch.delenv("GITMI_STREAM_F16")
    assert torch.equal(f32mode_on, f32mode_off)
    d = (f16s - f32s).abs().max().item()
    assert 0 < d < 0.02, d
    e16, e32 = (f16s - f32mode_off).abs().max().item(), (f32s - f32mode_off).abs().max().item()
    print("max feature error vs fp32 mode: fp16 stream %.4f, fp32 stream %.4f, between them %.4f" % (e16, e32, d))
    assert e16 < 0.05 and e16 < 1.5 * e32 + 0.005
