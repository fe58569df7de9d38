"""GPU: the `addln` schedule of the bf16 engine mode (GITMI_ADDLN=1): the N = hidden GEMMs of the image encoder and the
prefill write fp16 branch outputs and the LayerNorm kernel behind each adds the residual stream, so that every large GEMM
of the pass runs on the persistent kernel (kernels_gemm11.hip, GITMI_GEMM_IMPL=11).  Same fixed bounds as the default
schedule (tests/test_gpu_parity.py::check_bf16); the f32 engine mode ignores both switches."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["addln+p9", "addln", "p9"])
def schedule(request, monkeypatch):
    from generativeimage2text_amd import engine as E
    if "addln" in request.param:
        monkeypatch.setenv("GITMI_ADDLN", "1")
    if "p9" in request.param:
        monkeypatch.setenv("GITMI_GEMM_IMPL", "11")
    yield request.param
    monkeypatch.delenv("GITMI_ADDLN", raising=False)
    monkeypatch.delenv("GITMI_GEMM_IMPL", raising=False)
    E.set_gemm_impl(-1)                       # GITMI_GEMM_IMPL sets a process-wide switch at gitmi_create


CASES = ["tiny_greedy_long", "tiny_beam4", "tiny_video_beam4", "tinyl_greedy", "tiny_varres_up", "tiny_prefix_beam4",
         "base_greedy", "base_prefix_beam4", "large_greedy", "vatex_greedy", "vqa_base_480x640"]


@pytest.mark.parametrize("name", CASES)
def test_bf16_addln_schedule_within_tolerance(name, schedule):
    from test_gpu_parity import check_bf16
    check_bf16(name)


@pytest.mark.parametrize("name", ["full_bench_b64_greedy", "full_large_b32_greedy", "full_vatex_b16_greedy"])
def test_full_batch_ids_addln_schedule(name, monkeypatch):
    """The BASELINE configs at full batch under the addln + persistent-GEMM schedule: same floors on identical rows."""
    from generativeimage2text_amd import engine as E
    from generativeimage2text_amd.parity import IDENTICAL_FLOORS, bf16_bounds, ids_parity, lerr_frac_bound
    from test_gpu_parity import MG, format_like_reference, make_engine, record_measurement, search_struct
    from conftest import load_golden
    monkeypatch.setenv("GITMI_ADDLN", "1")
    monkeypatch.setenv("GITMI_GEMM_IMPL", "11")
    try:
        g = load_golden(name)
        cfg, w, frames, search, _ = MG.full_case_inputs(name)
        B, F = frames[0].shape[0], len(frames)
        eng = make_engine(cfg, w, "bf16", B, search, frames=F)
        tokens, logprobs, info = eng.generate([f.cuda() for f in frames], search_struct(search))
        preds, _ = format_like_reference(search, tokens, logprobs, info, None)
        logits = eng.step_logits(torch.from_numpy(g["tf_tokens"]))[:4, ::3].cpu().numpy()
        eng.close()
        lerr = float(np.abs(logits - g["tf_logits"]).max())
        span = float(g["tf_logits"].max() - g["tf_logits"].min())
        assert lerr < lerr_frac_bound(name, cfg.name) * span, (lerr, span)
        stats = ids_parity(preds.numpy(), g["predictions"], g["step_margin"], bf16_bounds(cfg.name)["thr"], False,
                           min_identical=IDENTICAL_FLOORS[name])
        record_measurement(case=name + "+addln+p9", config=cfg.name, lerr=round(lerr, 5), span=round(span, 3), **stats)
    finally:
        E.set_gemm_impl(-1)


def test_f32_mode_ignores_the_schedule_switches(monkeypatch):
    from oracle import git_oracle as O
    from generativeimage2text_amd import engine as E
    cfg = O.CONFIGS["TINY"]
    w = O.make_weights(cfg, seed=12, tie_output=False)
    frames = [f.cuda() for f in O.make_images(cfg, 3, 1, seed=3)]

    def feats():
        eng = E.Engine(cfg, precision="f32", max_batch=3, max_beams=1, max_frames=1, max_text_len=8)
        eng.load_state_dict(w)
        out = eng.encode(frames).cpu()
        eng.close()
        return out
    a = feats()
    monkeypatch.setenv("GITMI_ADDLN", "1")
    b = feats()
    assert torch.equal(a, b)
