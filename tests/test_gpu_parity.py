"""GPU: the HIP engine, called through the C ABI, against the CPU oracle and the golden vectors that
were generated from the real reference (oracle/make_golden.py).

Tolerances (stated per north_star):
  f32 engine mode  : tokens bit-identical to the reference; features / logits / log-probs |err| <= 1e-4
                     (measured ~3e-6: fp32 summation order only)
  f16 engine mode  : the HEADLINE build (fp16 operands).  ONE tolerance, from the specification (tools/parity.py):
                     |logit error| <= SPEC_LOGIT_FRAC = 1e-3 of the reference's logit span, on every case; a decision may
                     flip only where the fp32 margin is below 2 x that bound; END-TO-END ids compared on EVERY row: a row may
                     leave the reference's ids only at such a near-tie; fixtures whose every margin is wider must be identical.
  bf16 engine mode : the alternative build: 3 fewer mantissa bits, bound 2^3 x the constant (it does NOT meet the
                     specification on general weights: profiles/r06_*error_attribution*).
                     Regression guards beside the tolerance (constants, never derived from the run under test): visual
                     features <= FEATURE_ERR, floors on identical rows of the full-batch goldens (IDENTICAL_FLOORS).  Every case
                     appends its measured figures to gpurun_out/parity_measured.jsonl (copied to profiles/ per round).
"""
import numpy as np
import pytest
import torch

from conftest import golden_case, load_golden

pytestmark = pytest.mark.gpu

TINY_CASES = ["tiny_greedy_early_return", "tiny_greedy_untied", "tiny_greedy_long", "tiny_beam4", "tiny_beam4_noeos",
              "tiny_beam4_early_done", "tiny_beam3_pn3", "tiny_ar_beam3", "tiny_prefix_greedy", "tiny_prefix_beam4",
              "tiny_video_greedy", "tiny_video_beam4", "tiny_image_two_frames", "tinyl_greedy",
              "tiny_varres_up", "tiny_varres_down_beam4", "tiny_varres_prefix", "tinyl_varres"]
BIG_CASES = ["base_greedy", "base_greedy_eos", "base_beam4", "base_prefix_beam4", "large_greedy", "vatex_greedy",
             "vqa_base_480x640"]
FULL_CASES = ["full_bench_b64_greedy", "full_base_b64_greedy", "full_base_b64_beam4", "full_large_b32_greedy",
              "full_vatex_b16_greedy"]


def record_measurement(**kw):
    """One JSON line per 16-bit comparison (the measured side of every bound in tools/parity.py)."""
    import json, os
    from conftest import ROOT
    d = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "parity_measured.jsonl"), "a") as f:
            f.write(json.dumps(kw) + "\n")
    except OSError:
        pass


def make_engine(cfg, w, precision, B, search, frames=1, T=None, max_image_hw=None, serving=False):
    """serving: the kernel shapes of gitmi_set_shared_device -- the ones the benchmark's mixed schedule runs"""
    from generativeimage2text_amd.engine import Engine
    eng = Engine(cfg, precision=precision, max_batch=B, max_beams=max(1, search.beam_size), max_frames=frames,
                 max_text_len=T or search.max_steps, max_image_hw=max_image_hw)
    eng.load_state_dict(w)
    if serving:
        eng.set_shared_device(True)
    return eng


def search_struct(search):
    from generativeimage2text_amd.engine import Engine
    return Engine.make_search(search.kind, search.max_steps, search.beam_size, search.per_node_beam_size,
                              search.length_penalty)


def format_like_reference(search, tokens, logprobs, info, prefix):
    """What CaptioningModel.infer returns (decoder.py:1001-1011) from the raw engine outputs."""
    seq_len, early, _, _ = info.tolist()
    P = 1 if prefix is None else prefix.numel()
    if search.kind == "greedy":
        preds = tokens[:, P:P + 1] if early else tokens[:, :seq_len]
        lps = logprobs[:, None] if early else logprobs
    else:
        preds, lps = tokens, logprobs[:, None]
    if prefix is not None:
        preds = preds[:, P:]
    return preds.cpu(), lps.cpu()


def run_case(name, precision, serving=False):
    g, cfg, w, frames, search, prefix = golden_case(name)
    B, F = frames[0].shape[0], len(frames)
    hw = tuple(frames[0].shape[2:])
    eng = make_engine(cfg, w, precision, B, search, frames=F,
                      max_image_hw=hw if hw != (cfg.image_size, cfg.image_size) else None, serving=serving)
    dev_frames = [f.cuda() for f in frames]
    feats = eng.encode(dev_frames).cpu()
    logits = eng.step_logits(torch.from_numpy(g["tf_tokens"])).cpu()
    tokens, logprobs, info = eng.generate(dev_frames, search_struct(search), prefix=prefix)
    preds, lps = format_like_reference(search, tokens, logprobs, info, prefix)
    eng.close()
    return g, cfg, feats, logits, preds, lps


def check_f32(name):
    g, cfg, feats, logits, preds, lps = run_case(name, "f32")
    big = cfg.vocab > 5000
    fs = feats[:, ::7, ::5] if big else feats
    assert np.abs(fs.numpy() - g["feat_sample"]).max() < 1e-4
    ls = logits[:, ::3] if big else logits
    assert np.abs(ls.numpy() - g["tf_logits"]).max() < 1e-4
    assert np.array_equal(logits.argmax(-1).numpy(), g["tf_argmax"])
    assert preds.shape == g["predictions"].shape, (preds.shape, g["predictions"].shape)
    assert np.array_equal(preds.numpy(), g["predictions"]), (preds, g["predictions"])
    assert np.allclose(lps.numpy(), g["logprobs"], atol=1e-4), (lps, g["logprobs"])


def check_bf16(name, precision="bf16", serving=False):
    """precision "bf16" (benchmarked build) or "f16" (the fp16-operand build of the same kernels: scaled bounds)"""
    from tools.parity import FEATURE_ERR, ids_parity, logit_bound, margin_threshold
    g, cfg, feats, logits, preds, lps = run_case(name, precision, serving=serving)
    big = cfg.vocab > 5000
    fs = feats[:, ::7, ::5] if big else feats
    ferr = float(np.abs(fs.numpy() - g["feat_sample"]).max())
    ls = logits[:, ::3] if big else logits
    ref = g["tf_logits"]
    lerr = float(np.abs(ls.numpy() - ref).max())
    span = float(ref.max() - ref.min())
    rec = {"case": (name if precision == "bf16" else name + "@" + precision) + ("@serving" if serving else ""),
           "config": cfg.name, "ferr": round(ferr, 5),
           "lerr": round(lerr, 5), "span": round(span, 3), "lerr_frac": round(lerr / span, 6)}
    try:
        assert ferr < FEATURE_ERR[precision], ferr
        lbound = logit_bound(precision, span)               # the specification's constant x span (x 2^3 for bf16)
        assert lerr < lbound, (lerr, span)
        # token identity wherever the reference's own margin is resolvable within that bound
        am = logits.argmax(-1).numpy()
        for r in range(am.shape[0]):
            if g["tf_top2_margin"][r] > margin_threshold(precision, lbound, False):
                assert am[r] == g["tf_argmax"][r]
        # end-to-end ids, every row
        ref_p = g["predictions"]
        import ast
        kind, _, k, _, _ = ast.literal_eval(str(g["search"]))
        if ref_p.shape[1] <= 1 and kind == "greedy":                        # first-step early return (decoder.py:279-291)
            if preds.shape == ref_p.shape:
                assert np.array_equal(preds.numpy(), ref_p)
            return
        chained = not (kind == "greedy" and k == 1)
        # golden predictions of prefixed cases have the prefix stripped (decoder.py:1004-1006): decision s wrote position s
        stats = ids_parity(preds.numpy(), ref_p, g["step_margin"], margin_threshold(precision, lbound, chained),
                           chained, first_decision_pos=0 if g["prefix"].size else 1)
        rec.update(stats)
        print(name, stats)
    finally:
        record_measurement(**rec)


@pytest.mark.parametrize("name", TINY_CASES)
def test_tiny_f32_bit_identical_tokens(name):
    check_f32(name)


@pytest.mark.parametrize("name", TINY_CASES)
def test_tiny_bf16_within_tolerance(name):
    check_bf16(name)


@pytest.mark.parametrize("name", BIG_CASES)
def test_full_size_f32_bit_identical_tokens(name):
    check_f32(name)


@pytest.mark.parametrize("name", BIG_CASES)
def test_full_size_bf16_within_tolerance(name):
    check_bf16(name)


@pytest.mark.parametrize("prec", ["f32", "bf16"])
@pytest.mark.parametrize("name", ["tiny_greedy_long", "tiny_beam4", "tiny_video_beam4", "tiny_prefix_beam4", "base_greedy",
                                  "base_beam4", "base_prefix_beam4", "large_greedy", "vatex_greedy"])
def test_serving_policy_shapes_against_reference_goldens(name, prec):
    """The kernel shapes of gitmi_set_shared_device (what bench.py's mixed schedule runs: 256-row encoder tiles for N = 768,
    64-row / two-strip chain GEMMs, 8-pair one-wave decode attention, the walking vocabulary head on ~60 workgroups) straight
    against the reference's frozen outputs, not only through their bitwise equality with the solo shapes: f32 ids bit for
    bit, bf16 within the fixed bounds."""
    if prec == "f32":
        g, cfg, feats, logits, preds, lps = run_case(name, "f32", serving=True)
        assert preds.shape == g["predictions"].shape and np.array_equal(preds.numpy(), g["predictions"])
        assert np.allclose(lps.numpy(), g["logprobs"], atol=1e-4)
    else:
        check_bf16(name, serving=True)


@pytest.mark.parametrize("name", TINY_CASES + BIG_CASES)
def test_f16_operand_build_within_tolerance(name):
    """libgitmi_f16.so, the headline build: the same kernels built for fp16 operands (Engine(precision="f16")), held to the
    specification's constant itself (tools/parity.SPEC_LOGIT_FRAC)."""
    check_bf16(name, "f16")


def _scripted_module():
    import importlib.util, os
    from conftest import ROOT
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(ROOT, "oracle", "make_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


MG = _scripted_module()


@pytest.mark.parametrize("name", FULL_CASES)
def test_full_batch_ids_against_reference(name):
    """BASELINE.json configs at their full batch sizes (cfg2 B=64 greedy as benchmarked and with perturbed affines, cfg3
    B=64 beam 4, cfg4 GIT_LARGE B=32, cfg5 VATEX 6 frames B=16): reference ids from tests/golden/full_*.npz.
    f32 mode: bit-identical ids on every row.  bf16 mode (the benchmarked one): every row compared, divergence only at
    a near-tie of the fp32 reference (tools/parity.py)."""
    from tools.parity import ids_parity
    g = load_golden(name)
    cfg, w, frames, search, _ = MG.full_case_inputs(name)
    B, F = frames[0].shape[0], len(frames)
    dev = [f.cuda() for f in frames]
    ref_p, ref_l = g["predictions"], g["logprobs"]
    chained = search.kind != "greedy"
    tf = torch.from_numpy(g["tf_tokens"])
    # (precision, serving kernel shapes): the serving shapes -- what the benchmark's mixed schedule runs -- in the benchmarked
    # precision, on the SAME engine (gitmi_set_shared_device re-selects the kernels, the weights stay)
    eng = None
    for prec, serving in (("f32", False), ("bf16", False), ("bf16", True), ("f16", False)):
        if not serving:
            if eng is not None:
                eng.close()
            eng = make_engine(cfg, w, prec, B, search, frames=F)
        else:
            eng.set_shared_device(True)
        tokens, logprobs, info = eng.generate(dev, search_struct(search))
        preds, lps = format_like_reference(search, tokens, logprobs, info, None)
        logits = eng.step_logits(tf)[:4, ::3].cpu().numpy()
        lerr = float(np.abs(logits - g["tf_logits"]).max())
        if prec == "f32":
            assert lerr < 1e-4, lerr
            assert preds.shape == ref_p.shape and np.array_equal(preds.numpy(), ref_p)
            assert np.allclose(lps.numpy(), ref_l, atol=1e-4)
        else:
            from tools.parity import IDENTICAL_FLOORS, IDENTICAL_FLOORS_F16, logit_bound, margin_threshold
            floors = IDENTICAL_FLOORS_F16 if prec == "f16" else IDENTICAL_FLOORS
            span = float(g["tf_logits"].max() - g["tf_logits"].min())
            rec = {"case": (name if prec == "bf16" else name + "@" + prec) + ("@serving" if serving else ""),
                   "config": cfg.name, "lerr": round(lerr, 5), "span": round(span, 3), "lerr_frac": round(lerr / span, 6)}
            try:
                lbound = logit_bound(prec, span)
                assert lerr < lbound, (lerr, span)
                stats = ids_parity(preds.numpy(), ref_p, g["step_margin"], margin_threshold(prec, lbound, chained),
                                   chained, min_identical=floors[name])
                rec.update(stats)
                print(name, prec, "logit err %.4f of span %.2f" % (lerr, span), stats)
            finally:
                record_measurement(**rec)
    eng.close()


TF_CASES = ["full_bench_b64_greedy", "full_base_b64_greedy", "full_large_b32_greedy", "full_vatex_b16_greedy",
            "full_wide_b64_greedy", "full_wide_large_b32_greedy", "full_wide_vatex_b16_greedy",
            "full_trained_b8_greedy", "full_trained_b64_greedy"]


@pytest.mark.parametrize("name", TF_CASES)
def test_teacher_forced_decisions_against_reference(name):
    """EVERY decision of every row of the full-batch greedy goldens, on the workload's own weights: the engine is fed the
    reference's ids[:, :t] for t = 1 .. 19 (gitmi_step_logits == the reference's `step` callable, decoder.py:1013-1054), the
    no-repeat rule is applied as the search applies it (decoder.py:330) and the argmax must be the id the reference chose
    wherever the fp32 margin is >= 2 x the logit-error bound -- 900+ of the 1 216 decisions of the benchmark fixture instead of
    the 152 a free-running row reaches before its first near-tie.  The logit error is measured on all rows at all decisions:
    against the frozen reference values (top-8 + 128 sampled columns per decision, <case>_tf.npz, written by
    oracle/make_golden.py from ONE teacher-forced pass of the unmodified reference) and over ALL 30 522 columns against the
    f32 engine mode, which is itself held to 1e-4 of the frozen values.  f32 mode: every live decision must agree.
    bf16 and f16 builds, solo and serving kernel shapes."""
    from tools.parity import teacher_forced_parity, tf_bounds
    g, gt = load_golden(name), load_golden(name + "_tf")
    cfg, w, frames, search, _ = MG.full_case_inputs(name)
    B, F = frames[0].shape[0], len(frames)
    dev = [f.cuda() for f in frames]
    ref = g["predictions"]
    span = float(gt["logit_max"]) - float(gt["logit_min"])
    e32 = make_engine(cfg, w, "f32", B, search, frames=F)
    e32.encode(dev)
    cache = {}

    def f32_logits(tokens):
        t = tokens.shape[1]
        if t not in cache:
            cache[t] = e32.step_logits(tokens).clone()
        return cache[t]

    b32 = tf_bounds("f32", span)
    st = teacher_forced_parity(f32_logits, ref, gt, cfg.eos, b32["thr"], b32["lerr"])
    record_measurement(case=name + "@tf@f32", config=cfg.name, **st)
    assert st["ok"], st
    assert st["agree"] == st["decisions"], st                 # the reference-identical mode: all of them
    e32.close()
    for prec in ("bf16", "f16"):
        eng = make_engine(cfg, w, prec, B, search, frames=F)
        b = tf_bounds(prec, span)                              # the specification's constant x span (x 2^3 for bf16)
        for serving in (False, True):
            eng.set_shared_device(serving)
            eng.encode(dev)
            st = teacher_forced_parity(eng.step_logits, ref, gt, cfg.eos, b["thr"], b["lerr"], f32_step_logits=f32_logits)
            record_measurement(case=name + "@tf@" + prec + ("@serving" if serving else ""), config=cfg.name, **st)
            print(name, prec, "serving" if serving else "solo", st)
            assert st["ok"], st
            assert np.isfinite(st["max_logit_err"]), st
        eng.close()


@pytest.mark.parametrize("prec", ["f32", "bf16", "f16"])
@pytest.mark.parametrize("name", ["full_wide_b64_greedy", "full_wide_large_b32_greedy", "full_wide_vatex_b16_greedy",
                                  "full_trained_b8_greedy", "full_trained_b64_greedy"])
def test_wide_margin_batch_ids_identical_to_reference(name, prec):
    """north_star: "greedy outputs bit-identical to reference token IDs".  With plain random-init weights that clause is
    undecidable for ANY 16-bit pipeline: Gaussian logits over 30522 tokens put a top-1 / top-2 gap below the pipeline's own
    logit error somewhere in almost every 19-step row (on the benchmark's golden every one of the 64 rows has such a
    step).  tests/golden/full_wide_b64_greedy.npz is BASELINE's cfg2 (GIT_BASE, B = 64, greedy, 19 steps) in the regime
    where it IS decidable -- the benchmark's weight family with a successor structure on the output matrix
    (synthetic.random_state_dict(successor=1.0)) and 64 images on which EVERY decision of the fp32 reference has a margin
    >= 0.2 (oracle/make_golden.py: select_wide_images; first tokens differ with the image, 20 distinct ids per row) -- and
    there every precision of the engine must return the reference's ids on 64 of 64 rows.  The same construction for the
    other two greedy BASELINE configurations: cfg4 GIT_LARGE B = 32 (32 of 32) and cfg5 VATEX 6 frames B = 16 (16 of 16).
    full_trained_b8 / b64 (round 6): cfg2 with the STATISTICS of a trained checkpoint (synthetic.apply_trained_statistics:
    LayerNorm gains over [0.2, 5], biases of order 1, three ViT residual channels 100x / 300x / 1000x above the rest): every
    margin >= 0.03 = 2 x the specification's logit tolerance, so the f32 mode and the headline fp16 build must return every
    row; the bf16 build (bound 2^3 x wider than those margins) is held to finite outputs and its own logit bound."""
    from tools.parity import identity_required, ids_parity, logit_bound
    g = load_golden(name)
    cfg, w, frames, search, _ = MG.full_case_inputs(name)
    B, F = frames[0].shape[0], len(frames)
    trained = "trained" in name
    min_margin = float(g["step_margin"].min())
    assert min_margin >= (0.03 if trained else 0.1) and B == MG.FULL_CASES[name][2]
    span_ref = float(g["tf_logits"].max() - g["tf_logits"].min())
    must = identity_required(prec, span_ref, min_margin)
    assert must or (trained and prec == "bf16"), (name, prec, min_margin, span_ref)
    eng = make_engine(cfg, w, prec, B, search, frames=F)
    dev = [f.cuda() for f in frames]
    for serving in (False, True):               # solo kernel shapes, then the serving policy's, on the same engine
        eng.set_shared_device(serving)
        tokens, logprobs, info = eng.generate(dev, search_struct(search))
        preds, lps = format_like_reference(search, tokens, logprobs, info, None)
        logits = eng.step_logits(torch.from_numpy(g["tf_tokens"]))[:4, ::3].cpu().numpy()
        lerr = float(np.abs(logits - g["tf_logits"]).max())
        span = float(g["tf_logits"].max() - g["tf_logits"].min())
        assert np.isfinite(logits).all() and torch.isfinite(lps).all(), (name, prec, "non-finite outputs")
        assert lerr < logit_bound(prec, span), (lerr, span)
        thr = min(min_margin, 0.1) if must else 2.0 * logit_bound(prec, span)
        stats = ids_parity(preds.numpy(), g["predictions"], g["step_margin"], thr, chained=False, min_identical=B if must else None)
        record_measurement(case=name + "@" + prec + ("@serving" if serving else ""), config=cfg.name, lerr=round(lerr, 5),
                           span=round(span, 3), lerr_frac=round(lerr / span, 6),
                           min_margin=round(min_margin, 4), required=bool(must), **stats)
        if must:
            assert stats["identical"] == B and stats["safe_rows"] == B, (serving, stats)
            assert np.allclose(lps.numpy(), g["logprobs"], atol=1e-4 if prec == "f32" else 0.05), np.abs(lps.numpy() - g["logprobs"]).max()
        # the rows are not copies of each other (the VATEX model's first token is all but image-independent: one caption)
        assert len({tuple(r) for r in preds.numpy().tolist()}) >= ({64: 8, 32: 4}.get(B, 1) if not trained else 2)
    eng.close()


@pytest.mark.parametrize("prec", ["f32", "bf16", "f16"])
def test_wide_margin_batch_beam4_ids_identical_to_reference(prec):
    """The same weights and images under the SHIPPED search class (BASELINE cfg3: GeneratorWithBeamSearch, beam 4,
    length_penalty 0.6): tests/golden/full_wide_b64_beam4.npz, frozen from the unmodified reference.  No margin certificate
    exists for a beam search (the 2k candidates a step keeps include Gaussian-close runner-ups for any weights: median
    adjacent gap 0.01): f32 mode must return 64 of 64 rows, the 16-bit modes at least parity.WIDE_BEAM_FLOOR (measured 62 / 63)."""
    from tools.parity import WIDE_BEAM_FLOOR
    name = "full_wide_b64_beam4"
    g = load_golden(name)
    cfg, w, frames, search, _ = MG.full_case_inputs(name)
    B = frames[0].shape[0]
    assert B == 64 and search.beam_size == 4
    eng = make_engine(cfg, w, prec, B, search)
    dev = [f.cuda() for f in frames]
    ref = g["predictions"]
    for serving in (False, True):
        eng.set_shared_device(serving)
        tokens, logprobs, info = eng.generate(dev, search_struct(search))
        preds, lps = format_like_reference(search, tokens, logprobs, info, None)
        same = int((preds.numpy() == ref).all(axis=1).sum()) if preds.shape == ref.shape else 0
        record_measurement(case=name + "@" + prec + ("@serving" if serving else ""), config=cfg.name, rows=B, identical=same)
        assert same >= (B if prec == "f32" else WIDE_BEAM_FLOOR), (serving, same, B)
        eq = (preds.numpy() == ref).all(axis=1)
        assert np.allclose(lps.numpy()[eq], g["logprobs"][eq], atol=1e-4 if prec == "f32" else 0.05)
    eng.close()


# ---- the search seam with scripted logits (no model): device search == reference search ----------
@pytest.mark.parametrize("name", sorted(MG.SCRIPTED))
def test_device_search_scripted(name):
    from generativeimage2text_amd.engine import Engine
    from oracle import git_oracle as O
    kind, B, P, V, eos, T, k, pn, lpn, seed, at, boost = MG.SCRIPTED[name]
    gold = load_golden("scripted_search")
    start = torch.from_numpy(gold[name + ".start"])
    step = MG.scripted_step_factory(seed, V, eos, at, boost)
    import dataclasses
    cfg = dataclasses.replace(O.CONFIGS["TINY"], eos=eos, vocab=1000)
    nkeep, nret = MG.SCRIPTED_KEEP.get(name, (1, 1))          # GeneratorWithBeamSearch.search(num_keep_best, num_return_sequences)
    start = start.repeat_interleave(nret, dim=0)              # decoder.py:1093-1097: the seam sees every sentence nret times
    eng = Engine(cfg, precision="f32", max_batch=start.shape[0], max_beams=k, max_frames=1, max_text_len=T)
    eng.load_state_dict(O.make_weights(cfg, seed=1))
    s = Engine.make_search("greedy" if kind == "greedy" else "beam", T, k, pn, lpn if lpn > 0 else 1.0,
                           repetition_penalty=MG.SCRIPTED_RP.get(name, 1.0),       # decoder.py:1135-1144
                           num_keep_best=nkeep)
    eng.search_begin(s, start, V)
    for _ in range(T - P):
        rows = eng.search_rows().cpu()
        eng.search_advance(step(rows).cuda())
    tokens, logprobs, info = eng.search_finish()
    seq_len, early, _, _ = info.tolist()
    exp_p, exp_l = gold[name + ".pred"], gold[name + ".logprob"]
    if kind == "greedy":
        if early:
            got_p, got_l = tokens[:, P:P + 1], logprobs[:, None]
        else:
            got_p, got_l = tokens[:, :seq_len], logprobs
    else:
        got_p, got_l = tokens, (logprobs[:, None] if nkeep == 1 else logprobs)
    assert got_p.shape == exp_p.shape, (got_p.shape, exp_p.shape)
    assert np.array_equal(got_p.cpu().numpy(), exp_p), (got_p, exp_p)
    assert np.allclose(got_l.cpu().numpy(), exp_l, atol=1e-4), (got_l, exp_l)

    eng.close()


def test_step_logits_with_beams_matches_oracle():
    from oracle import git_oracle as O
    cfg = O.CONFIGS["TINY"]
    w = O.make_weights(cfg, seed=31)
    frames = O.make_images(cfg, 3, 1, seed=5)
    feats = O.visual_features(cfg, w, frames)
    g = torch.Generator().manual_seed(1)
    toks = torch.randint(0, cfg.vocab, (3 * 4, 6), generator=g)
    with torch.no_grad():
        ref = O.textual_logits_full(cfg, w, feats.repeat_interleave(4, 0), toks)[:, -1]
    eng = make_engine(cfg, w, "f32", 3, O.BEAM4)
    eng.encode([f.cuda() for f in frames], return_features=False)
    out = eng.step_logits(toks).cpu()
    assert (out - ref).abs().max().item() < 2e-3
    eng.close()


# ---- BASELINE.json full size (B=64, GIT_BASE, bf16): size-independent properties -------------------
@pytest.fixture(scope="module")
def base_engine():
    from generativeimage2text_amd.configs import config_for_model
    from generativeimage2text_amd.engine import Engine
    from generativeimage2text_amd.synthetic import random_state_dict
    cfg = config_for_model("GIT_BASE")
    eng = Engine(cfg, precision="bf16", max_batch=64, max_beams=4, max_frames=1, max_text_len=20)
    eng.load_state_dict(random_state_dict(cfg, seed=7, eos_bias=0.0))
    yield cfg, eng
    eng.close()


def test_full_size_batch_invariance_and_determinism(base_engine):
    from generativeimage2text_amd.engine import Engine
    from generativeimage2text_amd.synthetic import random_frames
    cfg, eng = base_engine
    frames = random_frames(cfg, 64, 1, seed=3)
    s = Engine.make_search("greedy", 20, 1, 1)
    t1, l1, i1 = eng.generate(frames, s)
    t2, l2, i2 = eng.generate(frames, s)
    assert torch.equal(t1, t2) and torch.equal(l1, l2)               # deterministic, graph replay included
    eng.set_graph(False)
    t3, l3, _ = eng.generate(frames, s)
    eng.set_graph(True)
    assert torch.equal(t1, t3) and torch.allclose(l1, l3)              # hipGraph replay == eager launches
    sub = [frames[0][8:12].contiguous()]
    t4, l4, _ = eng.generate(sub, s)
    assert torch.equal(t4, t1[8:12])                                   # captions do not depend on batch-mates
    assert (t1[:, 0] == cfg.sos).all()
    assert i1.tolist()[2] == 19                                        # 19 decode steps per caption


def test_full_size_beam_search_properties(base_engine):
    from generativeimage2text_amd.engine import Engine
    from generativeimage2text_amd.synthetic import random_frames
    cfg, eng = base_engine
    frames = random_frames(cfg, 64, 1, seed=4)
    tb, lb, ib = eng.generate(frames, Engine.make_search("beam", 20, 4, 2, 0.6))
    assert tb.shape == (64, 20) and (tb[:, 0] == cfg.sos).all()
    # every returned hypothesis ends with EOS padding and has a finite normalised score
    assert (tb[:, -1] == cfg.eos).all() and torch.isfinite(lb).all() and (lb > -1e4).all()
    sub = [frames[0][:2].contiguous()]
    tb2, lb2, _ = eng.generate(sub, Engine.make_search("beam", 20, 4, 2, 0.6))
    assert torch.equal(tb2, tb[:2])


def test_full_size_b16_matches_oracle_tokens_f32():
    # GIT_BASE, B=16, greedy: engine (exact fp32 mode) vs the cached oracle, token for token
    from oracle import git_oracle as O
    cfg = O.CONFIGS["GIT_BASE"]
    w = O.make_weights(cfg, seed=1235, tie_output=False, eos_bias=0.25)
    frames = O.make_images(cfg, 16, 1, seed=9)
    with torch.no_grad():
        ref = O.caption(cfg, w, frames, O.GREEDY, cached=True)
    eng = make_engine(cfg, w, "f32", 16, O.GREEDY)
    tokens, logprobs, info = eng.generate([f.cuda() for f in frames], search_struct(O.GREEDY))
    L = info.tolist()[0]
    assert L == ref["predictions"].shape[1]
    assert torch.equal(tokens[:, :L].cpu(), ref["predictions"])
    assert torch.allclose(logprobs.cpu(), ref["logprobs"], atol=2e-3)
    eng.close()


def test_cloned_context_shares_weights_and_overlaps(base_engine):
    # a clone borrows the packed weights: same captions, and two contexts may run on two streams at once
    from generativeimage2text_amd.engine import Engine
    from generativeimage2text_amd.synthetic import random_frames
    cfg, eng = base_engine
    other = eng.clone()
    fa, fb = random_frames(cfg, 16, 1, seed=11), random_frames(cfg, 16, 1, seed=12)
    s = Engine.make_search("greedy", 20, 1, 1)
    ta, _, _ = eng.generate(fa, s)
    tb, _, _ = eng.generate(fb, s)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(s1):
        ta2, _, _ = eng.generate(fa, s, sync=False)
    with torch.cuda.stream(s2):
        tb2, _, _ = other.generate(fb, s, sync=False)
    torch.cuda.synchronize()
    assert torch.equal(ta, ta2) and torch.equal(tb, tb2)
    other.close()


# ---- the reference-named task functions, end to end through a checkpoint file ---------------------------
def test_task_function_single_image_end_to_end(tmp_path, monkeypatch, caplog):
    """test_git_inference_single_image(image_path, model_name, prefix): JPEG on disk -> PIL transform ->
    checkpoint file ({'model': state_dict}, module.-prefixed keys) -> engine -> logged ids, vs the oracle."""
    import logging
    from PIL import Image
    from oracle import git_oracle as O
    from generativeimage2text_amd import inference as I
    cfg = O.CONFIGS["GIT_BASE"]
    w = O.make_weights(cfg, seed=1239, tie_output=False, eos_bias=0.3)
    ckpt = tmp_path / "model.pt"
    torch.save({"model": {"module." + k: v for k, v in w.items()}}, str(ckpt))
    rng = np.random.RandomState(3)
    img_path = tmp_path / "img.png"
    Image.fromarray(rng.randint(0, 255, (260, 340, 3), dtype=np.uint8)).save(str(img_path))
    monkeypatch.setenv("GIT_VOCAB", str(tmp_path / "no_vocab.txt"))
    monkeypatch.setattr(I, "get_tokenizer", lambda: I.IdTokenizer())
    # the shipped decoder (beam 4, length_penalty 0.6) with a bounded step budget for the oracle's sake
    from generativeimage2text_amd.model import GeneratorWithBeamSearch
    real_build = I.build_model
    monkeypatch.setattr(I, "build_model", lambda name, tok, c, **kw: real_build(
        name, tok, c, decoder=GeneratorWithBeamSearch(eos_index=102, max_steps=12, beam_size=4, length_penalty=0.6),
        **kw))
    with caplog.at_level(logging.INFO):
        I.test_git_inference_single_image(str(img_path), "GIT_BASE", "2054 2003", checkpoint=str(ckpt), precision="f32")
    got = I.test_git_inference_single_image.last_output
    x = I.image_transform(I.load_image_by_pil(str(img_path)), 224)[None]
    with torch.no_grad():
        ref = O.caption(cfg, w, [x], O.SearchConfig("beam", 12, 4, 2, 0.6), prefix=torch.tensor([[101, 2054, 2003]]),
                        cached=True)
    want = I.IdTokenizer().decode(ref["predictions"][0].tolist())
    assert got == want, (got, want)
    assert any("output:" in r.message for r in caplog.records)


def test_task_function_single_tsv(tmp_path, monkeypatch):
    """test_git_inference_single_tsv: base64 JPEG rows in, `key \\t [{"caption": ...}]` rows out, batched."""
    import base64, io, json
    from PIL import Image
    from oracle import git_oracle as O
    from generativeimage2text_amd import inference as I, tsv_io
    from generativeimage2text_amd.model import AutoRegressiveBeamSearch
    cfg = O.CONFIGS["GIT_BASE"]
    w = O.make_weights(cfg, seed=1240, tie_output=False, eos_bias=0.3)
    rng = np.random.RandomState(5)
    rows, imgs = [], []
    for i in range(5):
        im = Image.fromarray(rng.randint(0, 255, (230 + 7 * i, 300, 3), dtype=np.uint8))
        buf = io.BytesIO()
        im.save(buf, format="PNG")
        rows.append(["img%d" % i, base64.b64encode(buf.getvalue()).decode()])
        imgs.append(I.image_transform(im.convert("RGB"), 224))
    tsv_io.tsv_writer(rows, str(tmp_path / "in.tsv"))
    monkeypatch.setattr(I, "get_tokenizer", lambda: I.IdTokenizer())
    real_build = I.build_model
    monkeypatch.setattr(I, "build_model", lambda name, tok, c, **kw: real_build(
        name, tok, c, decoder=AutoRegressiveBeamSearch(eos_index=102, max_steps=10, beam_size=1, per_node_beam_size=1,
                                                       fix_missing_prefix=True), **kw))
    out = str(tmp_path / "out.tsv")
    I.test_git_inference_single_tsv(str(tmp_path / "in.tsv"), "GIT_BASE", None, out, checkpoint=w, batch_size=4,
                                    precision="f32")
    got = list(tsv_io.tsv_reader(out))
    assert [r[0] for r in got] == ["img%d" % i for i in range(5)]
    with torch.no_grad():
        ref = O.caption(cfg, w, [torch.stack(imgs)], O.SearchConfig("greedy", 10, 1, 1), cached=True)
    for r, pred in zip(got, ref["predictions"].tolist()):
        assert json.loads(r[1])[0]["caption"] == I.IdTokenizer().decode(pred)
    # pipelined (default: 4 contexts in flight, thread-pool decoding) == the serial loop, in the default 16-bit precision too
    outs = {}
    for name, kw in (("pipe", dict(contexts=4, batch_size=2)), ("serial", dict(contexts=1, batch_size=1))):
        st = {}
        monkeypatch.setenv("GIT_DECODE_THREADS", "4" if name == "pipe" else "0")
        I.test_git_inference_single_tsv(str(tmp_path / "in.tsv"), "GIT_BASE", None, str(tmp_path / (name + ".tsv")), checkpoint=w,
                                        stats=st, **kw)
        assert st["precision"] == "f16" and st["images"] == 5
        outs[name] = open(str(tmp_path / (name + ".tsv"))).read()
    assert outs["pipe"] == outs["serial"]


def test_task_function_vqa_tsv_variable_resolution(tmp_path, monkeypatch):
    """test_git_inference_single_tsv with a question TSV on a test_respect_ratio_max model (the VQAv2/TextVQA
    geometry, scaled down to crop 160 / max 224 so that the CPU oracle stays cheap): every image keeps its aspect ratio
    (MinMaxResizeForTest on the GPU), the engine follows the resolution image by image (positional grid resized at run
    time), one row `key \t {"answer", "question_id"}` per question."""
    import base64, dataclasses, io, json
    from PIL import Image
    from oracle import git_oracle as O
    from generativeimage2text_amd import configs, inference as I, tsv_io
    from generativeimage2text_amd.model import GeneratorWithBeamSearch
    monkeypatch.setitem(configs.MODEL_PARAMS, "GIT_BASE_VQAv2", {"test_crop_size": 160, "test_respect_ratio_max": 224})
    monkeypatch.setitem(I.MODEL_PARAMS, "GIT_BASE_VQAv2", {"test_crop_size": 160, "test_respect_ratio_max": 224})
    cfg = dataclasses.replace(O.CONFIGS["GIT_BASE"], name="vqa_small", image_size=160)
    w = O.make_weights(cfg, seed=1241, tie_output=False, eos_bias=0.3)
    rng = np.random.RandomState(6)
    # (h, w): landscape, portrait beyond the ratio cap, square, and two more of the FIRST shape: images that share a resized
    # shape are answered in ONE engine call (ragged questions about several images), the others in calls of their own; the
    # rows still come out in input order
    sizes = [(300, 400), (500, 260), (333, 333), (300, 400), (300, 400)]
    questions = [[("2054 2003 2023", 11)], [("2129 2116", 12), ("2054 3609", 13)], [("2003 2009 1037 4937", 14)],
                 [("2054 2003 2023", 15), ("2129 2116 1996", 16)], [("2003 2009", 17)]]
    img_rows, q_rows, want = [], [], []
    for i, ((h, wd), qs) in enumerate(zip(sizes, questions)):
        im = Image.fromarray(rng.randint(0, 255, (h, wd, 3), dtype=np.uint8))
        buf = io.BytesIO()
        im.save(buf, format="PNG")
        img_rows.append(["img%d" % i, base64.b64encode(buf.getvalue()).decode()])
        q_rows.append(["img%d" % i, json.dumps([{"question": q, "question_id": qid} for q, qid in qs])])
        x = I.minmax_image_transform(im.convert("RGB"), 160, 224)[None]
        assert x.shape[2] != x.shape[3] or (h, wd) == (333, 333)
        for q, qid in qs:
            ids = [101] + [int(t) for t in q.split()]
            with torch.no_grad():
                ref = O.caption(cfg, w, [x], O.SearchConfig("beam", 14, 4, 2, 0.6), prefix=torch.tensor([ids]), cached=True)
            want.append(("img%d" % i, qid, I.IdTokenizer().decode(ref["predictions"][0].tolist())))
    tsv_io.tsv_writer(img_rows, str(tmp_path / "img.tsv"))
    tsv_io.tsv_writer(q_rows, str(tmp_path / "q.tsv"))
    monkeypatch.setattr(I, "get_tokenizer", lambda: I.IdTokenizer())
    real_build = I.build_model
    monkeypatch.setattr(I, "build_model", lambda name, tok, c, **kw: real_build(
        name, tok, c, decoder=GeneratorWithBeamSearch(eos_index=102, max_steps=14, beam_size=4, length_penalty=0.6), **kw))
    out = str(tmp_path / "out.tsv")
    I.test_git_inference_single_tsv(str(tmp_path / "img.tsv"), "GIT_BASE_VQAv2", str(tmp_path / "q.tsv"), out, checkpoint=w,
                                    precision="f32")
    # one-column rows json_dump({"answer", "question_id"}) (inference.py:199), questions of an image answered in ONE engine call
    got = [(json.loads(s)["question_id"], json.loads(s)["answer"]) for s, in tsv_io.tsv_reader(out)]
    assert got == [(qid, ans) for _, qid, ans in want], (got, want)
    # the serial loop of the reference (one image per call, nothing in flight) writes the same file
    st = {}
    out1 = str(tmp_path / "out_serial.tsv")
    I.test_git_inference_single_tsv(str(tmp_path / "img.tsv"), "GIT_BASE_VQAv2", str(tmp_path / "q.tsv"), out1, checkpoint=w,
                                    precision="f32", contexts=1, batch_size=1, stats=st)
    assert open(out1).read() == open(out).read() and st["batches"] == 5 and st["questions"] == 7


@pytest.mark.parametrize("kind", ["greedy", "beam"])
def test_long_step_budget_polling_path(kind):
    """The shipped step budget is max_steps=1024 (model.py:37): far beyond 32 steps the engine launches eagerly
    and polls the device-side 'every sentence finished' flag every 8 steps.  Results must not depend on when it
    notices (decoder.py:319 / :1251 semantics), tokens bit-identical to the oracle in fp32 mode."""
    from oracle import git_oracle as O
    cfg = O.CONFIGS["TINY"]
    w = O.make_weights(cfg, seed=77, tie_output=False, eos_bias=2.2)
    frames = O.make_images(cfg, 5, 1, seed=8)
    search = O.SearchConfig("greedy", 60, 1, 1) if kind == "greedy" else O.SearchConfig("beam", 60, 4, 2, 0.6)
    trace = []
    with torch.no_grad():
        ref = O.caption(cfg, w, frames, search, cached=True, trace=trace)
    margins = torch.stack(trace, dim=1).numpy()                       # [B, decisions] fp32 decision margins of the oracle
    for prec in ("f32", "bf16"):
        eng = make_engine(cfg, w, prec, 5, search)
        tokens, logprobs, info = eng.generate([f.cuda() for f in frames], search_struct(search))
        preds, lps = format_like_reference(search, tokens, logprobs, info, None)
        assert info.tolist()[2] < 59              # stopped early: fewer decode steps than the budget
        if prec == "f32":
            assert preds.shape == ref["predictions"].shape, (preds.shape, ref["predictions"].shape)
            assert torch.equal(preds, ref["predictions"])
            assert torch.allclose(lps, ref["logprobs"], atol=2e-3)
        else:
            # bf16 (fused head, folded LayerNorms, eager polling path): a row may differ from the reference only if one of
            # its decisions is a near-tie (fixed threshold, tools/parity.BEAM_MARGIN_THR); compared on EOS-padded rows
            from tools.parity import BEAM_MARGIN_THR, ids_parity
            L = max(preds.shape[1], ref["predictions"].shape[1])
            pad = lambda x: torch.cat([x, torch.full((x.shape[0], L - x.shape[1]), cfg.eos, dtype=x.dtype)], 1).numpy()
            stats = ids_parity(pad(preds), pad(ref["predictions"]), margins, BEAM_MARGIN_THR["bf16"], chained=True)
            record_measurement(case="long_budget_" + kind, config=cfg.name, **stats)
            assert torch.isfinite(lps).all()
        eng.close()


# ---- batched VQA: ragged per-sentence prefixes in one call == one reference call per question --------------------
@pytest.mark.parametrize("kind", ["greedy", "beam", "ar_beam"])
def test_ragged_prefixes_equal_per_question_reference_calls(kind):
    """The reference answers one question per model call (decoder.py:984-989; inference.py:172-199).  The engine runs
    questions of different lengths about several images in one call (image K/V shared by the questions of an image);
    in f32 mode every sentence must get bit for bit what its own batch-1 call returns, incl. early ends."""
    from oracle import git_oracle as O
    cfg = O.CONFIGS["TINY"]
    w = O.make_weights(cfg, seed=51, tie_output=False, successor=2.0, eos_bias=1.0)
    frames = O.make_images(cfg, 3, 1, seed=9)
    search = {"greedy": O.SearchConfig("greedy", 18, 1, 1), "beam": O.SearchConfig("beam", 18, 4, 2, 0.6),
              "ar_beam": O.SearchConfig("greedy", 18, 3, 2)}[kind]
    prefixes = [[101, 7, 44], [101], [101, 300, 2, 9, 512, 77], [101, 5], [101, 7, 44, 13, 8], [101, 640, 3, 3, 21, 90, 14, 2]]
    image_of = [0, 0, 1, 2, 2, 1]
    want, min_margin = [], []
    with torch.no_grad():
        for p, im in zip(prefixes, image_of):
            tr = []
            ref = O.caption(cfg, w, [frames[0][im:im + 1]], search, prefix=torch.tensor([p]), cached=True, trace=tr)
            want.append((ref["predictions"][0].tolist(), float(ref["logprobs"].flatten()[0])))
            min_margin.append(float(torch.stack(tr, 1).min()) if tr else float("inf"))
    eng = make_engine(cfg, w, "f32", 8, search)
    dev = [f.cuda() for f in frames]
    for graph in (True, False):
        eng.set_graph(graph)
        tokens, logprobs, sent, info = eng.generate_prefixed(dev, search_struct(search), prefixes, image_of)
        tokens, logprobs, sent = tokens.cpu(), logprobs.cpu(), sent.cpu()
        for q, p in enumerate(prefixes):
            P, (L, early) = len(p), sent[q].tolist()
            if search.kind == "greedy":
                got = (tokens[q, P:P + 1] if early else tokens[q, :L])[P:].tolist()
            else:
                got = tokens[q, P:].tolist()
            assert got == want[q][0], (kind, q, got, want[q][0])
            assert abs(float(logprobs[q]) - want[q][1]) < 1e-4, (kind, q)
    # bf16 mode runs the same call (fused vocabulary head, folded LayerNorms): a sentence may differ from its own
    # reference call only if one of that call's decisions is a near-tie (fixed thresholds of tools/parity.py)
    from tools.parity import BEAM_MARGIN_THR, GREEDY_MARGIN_CAP
    thr = GREEDY_MARGIN_CAP["bf16"] if (search.kind == "greedy" and search.beam_size == 1) else BEAM_MARGIN_THR["bf16"]
    eb = make_engine(cfg, w, "bf16", 8, search)
    tb, lb, sb, _ = eb.generate_prefixed(dev, search_struct(search), prefixes, image_of)
    tb, sb = tb.cpu(), sb.cpu()
    same = 0
    for q, p in enumerate(prefixes):
        assert tb[q, :len(p)].tolist() == p
        P, (L, early) = len(p), sb[q].tolist()
        if search.kind == "greedy":
            got = (tb[q, P:P + 1] if early else tb[q, :L])[P:].tolist()
        else:
            got = tb[q, P:].tolist()
        same += got == want[q][0]
        if min_margin[q] >= thr:
            assert got == want[q][0], (kind, q, got, want[q][0], min_margin[q])
    record_measurement(case="ragged_prefixes_" + kind, config=cfg.name, identical=same, rows=len(prefixes),
                       min_margins=[round(m, 4) for m in min_margin])
    assert torch.isfinite(lb).all()
    eng.close()
    eb.close()


def test_model_answer_matches_single_prefix_calls():
    """CaptioningModel.answer(image, [prefixes]) == [model({'image', 'prefix'}) per prefix] (the reference loop)."""
    from oracle import git_oracle as O
    from generativeimage2text_amd.model import CaptioningModel, GeneratorWithBeamSearch
    cfg = O.CONFIGS["TINY"]
    w = O.make_weights(cfg, seed=52, tie_output=False, successor=2.0, eos_bias=1.5)
    img = O.make_images(cfg, 1, 1, seed=4)[0].cuda()
    dec = GeneratorWithBeamSearch(eos_index=cfg.eos, max_steps=16, beam_size=4, length_penalty=0.6)
    model = CaptioningModel(cfg, dec, precision="f32", max_batch=4)
    model.load_state_dict(w)
    prefixes = [[101, 9, 8, 7], [101, 400], [101, 3, 3, 3, 3, 3, 3]]
    batched = model.answer(img, prefixes)
    for p, got in zip(prefixes, batched):
        single = model({"image": img, "prefix": torch.tensor([p]).cuda()})["predictions"][0].tolist()
        assert got == single


def test_video_model_with_bare_tensor_skips_temporal_embedding():
    """decoder.py:845-857: img_temperal_embedding is added only when batch['image'] is a LIST of frames."""
    from oracle import git_oracle as O
    from generativeimage2text_amd.model import CaptioningModel, AutoRegressiveBeamSearch
    cfg = O.CONFIGS["TINY_VIDEO"]
    w = O.make_weights(cfg, seed=53, tie_output=False, successor=2.0)
    frames = O.make_images(cfg, 2, 1, seed=6)
    dec = AutoRegressiveBeamSearch(eos_index=cfg.eos, max_steps=12, beam_size=1, per_node_beam_size=1, fix_missing_prefix=True)
    model = CaptioningModel(cfg, dec, precision="f32", max_batch=2)
    model.load_state_dict(w)
    search = O.SearchConfig("greedy", 12, 1, 1)
    with torch.no_grad():
        ref_tensor = O.caption(cfg, w, frames, search, cached=True, feats=O.visual_features(cfg, w, frames, as_list=False))
        ref_list = O.caption(cfg, w, frames, search, cached=True)
    got_tensor = model({"image": frames[0].cuda()})["predictions"].cpu()
    got_list = model({"image": [frames[0].cuda()]})["predictions"].cpu()
    assert torch.equal(got_tensor, ref_tensor["predictions"])
    assert torch.equal(got_list, ref_list["predictions"])
    assert not torch.equal(ref_tensor["predictions"], ref_list["predictions"])      # the embedding matters for this seed


def test_single_rank_rccl_gather_path(tmp_path):
    """The RCCL ("nccl") result gather, executed for real on one rank: bench.gather_results with an initialised
    1-rank group is a no-op by design, so the collective itself is driven directly with the same packing."""
    import subprocess, sys, os
    from conftest import ROOT
    code = (
        "import os, torch, torch.distributed as dist\n"
        "os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29543', RANK='0', WORLD_SIZE='1')\n"
        "torch.cuda.set_device(0)\n"
        "dist.init_process_group('nccl', rank=0, world_size=1)\n"
        "toks = torch.arange(64 * 20, device='cuda').reshape(64, 20)\n"
        "lps = torch.linspace(-3, 0, 64, device='cuda')\n"
        "packed = torch.cat([toks, lps.view(torch.int32).to(torch.int64)[:, None]], 1).contiguous()\n"
        "out = [torch.empty_like(packed)]\n"
        "dist.gather(packed, out, dst=0)\n"
        "dist.barrier(); torch.cuda.synchronize()\n"
        "assert torch.equal(out[0], packed)\n"
        "rows = [['k%d' % i, 'c%d' % i] for i in range(5)]\n"
        "got = [None]\n"
        "dist.gather_object(rows, got, dst=0)\n"
        "assert got[0] == rows\n"
        "dist.destroy_process_group()\n"
        "print('RCCL-1-RANK-OK')\n")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert "RCCL-1-RANK-OK" in r.stdout, r.stdout + r.stderr


def test_task_function_single_image_with_wordpiece_text(tmp_path, monkeypatch, caplog):
    """test_git_inference_single_image end to end with REAL text in and out: BertTokenizer on a WordPiece vocabulary
    (tests/data/vocab.txt, bert-base-uncased special-token ids), question text -> prefix ids (inference.py:93-101),
    engine, ids -> decoded answer string (inference.py:108) -- against the oracle's ids decoded by the same tokenizer."""
    import logging, os
    from PIL import Image
    from conftest import ROOT
    from oracle import git_oracle as O
    from generativeimage2text_amd import inference as I
    from generativeimage2text_amd.model import GeneratorWithBeamSearch
    monkeypatch.setenv("GIT_VOCAB", os.path.join(ROOT, "tests", "data", "vocab.txt"))
    cfg = O.CONFIGS["TINY"]                                       # vocab 1000 = the synthetic vocabulary's size
    w = O.make_weights(cfg, seed=61, tie_output=False, successor=2.0, eos_bias=1.5)
    ckpt = tmp_path / "model.pt"
    torch.save({"model": w}, str(ckpt))
    rng = np.random.RandomState(9)
    img_path = tmp_path / "img.png"
    Image.fromarray(rng.randint(0, 255, (90, 120, 3), dtype=np.uint8)).save(str(img_path))
    monkeypatch.setattr(I, "config_for_model", lambda name: cfg)
    monkeypatch.setitem(I.MODEL_PARAMS, "TINY_TEXT", {"test_crop_size": cfg.image_size})
    real_build = I.build_model
    monkeypatch.setattr(I, "build_model", lambda name, tok, c, **kw: real_build(
        name, tok, c, decoder=GeneratorWithBeamSearch(eos_index=tok.sep_token_id, max_steps=24, beam_size=4, length_penalty=0.6),
        **kw))
    question = "what color is the cat sitting on the table?"
    with caplog.at_level(logging.INFO):
        I.test_git_inference_single_image(str(img_path), "TINY_TEXT", question, checkpoint=str(ckpt))
    got = I.test_git_inference_single_image.last_output
    tok = I.get_tokenizer()
    ids = I._prefix_ids(tok, question)
    assert len(ids) > 8
    x = I.image_transform(I.load_image_by_pil(str(img_path)), cfg.image_size)[None]
    with torch.no_grad():
        ref = O.caption(cfg, w, [x], O.SearchConfig("beam", 24, 4, 2, 0.6), prefix=torch.tensor([ids]), cached=True)
    want = tok.decode(ref["predictions"][0].tolist(), skip_special_tokens=True)
    assert got == want and len(want) > 0, (got, want)


def test_generate_with_sampling_search():
    """model(batch, search_param={'do_sample': True, ...}) (decoder.py:895-905 passes these to decoder.search): runs the
    sampling branch of GeneratorWithBeamSearch end to end, is reproducible per seed, differs across seeds, and with
    top_k = 1 (a single kept token... plus the min-keep rule) stays inside the two most likely tokens at every step."""
    from oracle import git_oracle as O
    from generativeimage2text_amd.model import CaptioningModel, GeneratorWithBeamSearch
    cfg = O.CONFIGS["TINY"]
    w = O.make_weights(cfg, seed=71, tie_output=False, successor=2.0, eos_bias=1.0)
    frames = O.make_images(cfg, 3, 1, seed=2)
    dec = GeneratorWithBeamSearch(eos_index=cfg.eos, max_steps=14, beam_size=2, length_penalty=0.6, temperature=1.2)
    for prec in ("f32", "bf16"):
        model = CaptioningModel(cfg, dec, precision=prec, max_batch=3)
        model.load_state_dict(w)
        img = frames[0].cuda()
        a = model({"image": img}, {"do_sample": True, "top_k": 20, "top_p": 0.9, "seed": 7})
        b = model({"image": img}, {"do_sample": True, "top_k": 20, "top_p": 0.9, "seed": 7})
        c = model({"image": img}, {"do_sample": True, "top_k": 20, "top_p": 0.9, "seed": 8})
        g = model({"image": img})
        assert torch.equal(a["predictions"], b["predictions"]) and torch.equal(a["logprobs"], b["logprobs"])
        assert not torch.equal(a["predictions"], c["predictions"])
        assert a["predictions"].shape == g["predictions"].shape == (3, 14)
        assert (a["predictions"][:, 0] == cfg.sos).all() and torch.isfinite(a["logprobs"]).all()


@pytest.mark.parametrize("name", sorted(MG.SCRIPTED))
def test_search_method_scripted(name):
    """decoder.search(start_predictions, step) of the two mirror classes (the reference's search seam as a method, over
    the engine's device search) against the reference's own outputs for the scripted `step` functions."""
    from generativeimage2text_amd.model import AutoRegressiveBeamSearch, GeneratorWithBeamSearch
    kind, B, P, V, eos, T, k, pn, lpn, seed, at, boost = MG.SCRIPTED[name]
    gold = load_golden("scripted_search")
    start = torch.from_numpy(gold[name + ".start"])
    step = MG.scripted_step_factory(seed, V, eos, at, boost)
    if kind == "greedy":
        dec = AutoRegressiveBeamSearch(eos_index=eos, max_steps=T, beam_size=k, per_node_beam_size=pn, fix_missing_prefix=True)
    else:
        dec = GeneratorWithBeamSearch(eos_index=eos, max_steps=T, beam_size=k, per_node_beam_size=pn, length_penalty=lpn,
                                      repetition_penalty=MG.SCRIPTED_RP.get(name, 1.0))
    calls = []

    def counted_step(rows):
        calls.append(int(rows.shape[1]))
        return step(rows.cpu()).cuda()
    nkeep, nret = MG.SCRIPTED_KEEP.get(name, (1, 1))
    keep_kw = {} if (nkeep, nret) == (1, 1) else {"num_keep_best": nkeep, "num_return_sequences": nret}
    got_p, got_l = dec.search(start.cuda(), counted_step, **keep_kw)
    exp_p, exp_l = gold[name + ".pred"], gold[name + ".logprob"]
    assert got_p.shape == exp_p.shape, (got_p.shape, exp_p.shape)
    assert np.array_equal(got_p.cpu().numpy(), exp_p), (got_p, exp_p)
    assert np.allclose(got_l.cpu().numpy(), exp_l, atol=1e-4), (got_l, exp_l)
    # ... and `step` was called on exactly the row lengths the reference called it on (golden: counted on the reference)
    assert calls == gold[name + ".step_calls"].tolist(), (calls, gold[name + ".step_calls"].tolist())
    # the reference calls `step` exactly as often (counted on the reference class itself)
    D = MG.import_reference()[1] if hasattr(MG, "import_reference") and __import__("os").path.isdir("/root/reference") else None
    if D is not None:
        ref_calls = []

        def ref_step(rows):
            ref_calls.append(int(rows.shape[1]))
            return step(rows)
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            if kind == "greedy":
                D.AutoRegressiveBeamSearch(eos_index=eos, max_steps=T, beam_size=k, per_node_beam_size=pn,
                                           fix_missing_prefix=True).search(start, ref_step)
            else:
                D.GeneratorWithBeamSearch(eos_index=eos, max_steps=T, beam_size=k, per_node_beam_size=pn, length_penalty=lpn,
                                          repetition_penalty=MG.SCRIPTED_RP.get(name, 1.0)).search(start, ref_step, **keep_kw)
        assert calls == ref_calls, (calls, ref_calls)


# ---- trie-constrained greedy decoding (trie_decoder.py:27-257) -----------------------------------------------------
def test_model_num_keep_best_and_num_return_sequences_match_oracle():
    """model(batch, search_param={'num_keep_best': n, 'num_return_sequences': r}) with the shipped search class
    (GeneratorWithBeamSearch.search keywords, decoder.py:1087-1097, 1262-1290; pinned against the reference itself by the
    scripted goldens s2_keep* / s2_ret*): the n best hypotheses of every sentence, best first, ids bit for bit in f32
    mode; r sentences per image (identical copies without sampling); the reference's prefix strip acts on dim 1."""
    from oracle import git_oracle as O
    from generativeimage2text_amd.model import CaptioningModel, GeneratorWithBeamSearch
    cfg = O.CONFIGS["TINY"]
    w = O.make_weights(cfg, seed=47, tie_output=False, successor=2.0, eos_bias=4.0)          # mixed: full-length and early-EOS hypotheses
    frames = O.make_images(cfg, 3, 1, seed=12)
    T, k, pn, lpn, n = 12, 4, 2, 0.6, 3
    dec = GeneratorWithBeamSearch(eos_index=cfg.eos, max_steps=T, beam_size=k, per_node_beam_size=pn, length_penalty=lpn)
    model = CaptioningModel(cfg, dec, precision="f32", max_batch=6)
    model.load_state_dict(w)
    with torch.no_grad():
        feats = O.visual_features(cfg, w, frames)
        start = torch.full((3, 1), cfg.sos, dtype=torch.long)
        want_p, want_l = O.search_generator(start, O.make_step(cfg, w, feats, cached=True), cfg.eos, T, k, pn, lpn,
                                            num_keep_best=n)
    assert want_p.shape == (3, n, T) and (want_l > -1e4).sum() >= 6         # the lists hold real hypotheses
    out = model({"image": frames[0].cuda()}, search_param={"num_keep_best": n})
    assert torch.equal(out["predictions"].cpu(), want_p), (out["predictions"].cpu(), want_p)
    assert torch.allclose(out["logprobs"].cpu(), want_l, atol=1e-3)
    one = model({"image": frames[0].cuda()})                                # the default call is the first hypothesis
    assert torch.equal(one["predictions"].cpu(), want_p[:, 0]) and one["logprobs"].shape == (3, 1)
    # num_return_sequences = 2: rows b * 2 + j, every copy equal to the image's own result (no sampling)
    rep = model({"image": frames[0].cuda()}, search_param={"num_keep_best": n, "num_return_sequences": 2})
    assert rep["predictions"].shape == (6, n, T)
    assert torch.equal(rep["predictions"].cpu(), want_p.repeat_interleave(2, dim=0))
    with pytest.raises(ValueError):
        model({"image": frames[0].cuda()}, search_param={"num_return_sequences": 3})       # 9 sentences > max_batch 6
    from generativeimage2text_amd.engine import Engine, GitmiError
    with pytest.raises(GitmiError, match="num_keep_best"):
        model.engine.generate([frames[0].cuda()], Engine.make_search("greedy", T, 1, 1, num_keep_best=2))
    model.engine.close()


@pytest.mark.parametrize("name", sorted(MG.SCRIPTED_TRIE))
def test_device_trie_search_scripted(name):
    """The device trie search (gitmi_set_trie + GITMI_SEARCH_TRIE behind decoder.search(start, step)) against the
    reference's TrieAutoRegressiveBeamSearch.search on scripted `step` functions (goldens frozen from the reference)."""
    from generativeimage2text_amd.model import TokenTrie, TrieAutoRegressiveBeamSearch
    P, V, eos, T, seed, n_seqs, len_range = MG.SCRIPTED_TRIE[name]
    gold = load_golden("scripted_trie")
    start = torch.from_numpy(gold[name + ".start"])
    step = MG.scripted_step_factory(seed, V, eos)
    seqs = MG.scripted_trie_sequences(seed, V, eos, n_seqs, len_range)
    dec = TrieAutoRegressiveBeamSearch(eos_index=eos, max_steps=T, beam_size=1, trie=TokenTrie.construct(seqs))
    got_p, got_l = dec.search(start.cuda(), lambda rows: step(rows.cpu()).cuda())
    exp_p, exp_l = gold[name + ".pred"], gold[name + ".logprob"]
    assert got_p.shape == exp_p.shape, (got_p.shape, exp_p.shape)
    assert np.array_equal(got_p.cpu().numpy(), exp_p), (got_p, exp_p)
    assert np.allclose(got_l.cpu().numpy().reshape(exp_l.shape), exp_l, rtol=2e-6, atol=1e-4), (got_l, exp_l)


@pytest.mark.parametrize("precision", ["f32", "bf16"])
def test_model_with_trie_decoder_matches_oracle(precision):
    """model(batch) with decoder = TrieAutoRegressiveBeamSearch (how the reference classifies with a caption model:
    the answer is forced onto the token sequences of a vocabulary trie): every image of a batch gets what its own batch-1
    reference call returns (f32: ids bit for bit; bf16: an id may differ only where the constrained choice was a near-tie,
    which these seeded cases do not contain), and every answer is a path of the trie."""
    from oracle import git_oracle as O
    from generativeimage2text_amd.model import CaptioningModel, TokenTrie, TrieAutoRegressiveBeamSearch
    cfg = O.CONFIGS["TINY"]
    w = O.make_weights(cfg, seed=91, tie_output=False, successor=2.0)
    frames = O.make_images(cfg, 3, 1, seed=7)
    g = torch.Generator().manual_seed(4)
    seqs = []
    for _ in range(120):
        L = int(torch.randint(3, 9, (1,), generator=g))
        seqs.append(torch.randint(3, cfg.vocab, (L - 1,), generator=g).tolist() + [cfg.eos])
    T = 12
    want = []
    with torch.no_grad():
        for i in range(3):
            ref = O.caption(cfg, w, [frames[0][i:i + 1]], O.SearchConfig("trie", T, 1, 1), cached=True,
                            trie=O.TokenTrie.construct(seqs))
            want.append((ref["predictions"][0].tolist(), float(ref["logprobs"].flatten()[0])))
    dec = TrieAutoRegressiveBeamSearch(eos_index=cfg.eos, max_steps=T, beam_size=1, trie=TokenTrie.construct(seqs))
    model = CaptioningModel(cfg, dec, precision=precision, max_batch=3)
    model.load_state_dict(w)
    out = model({"image": frames[0].cuda()})
    preds, lps = out["predictions"].cpu(), out["logprobs"].cpu()
    for i in range(3):
        row = preds[i].tolist()
        Lw = len(want[i][0])
        assert row[:Lw] == want[i][0] and all(t == cfg.eos for t in row[Lw:]), (precision, i, row, want[i][0])
        gen = want[i][0][1:]
        assert any(gen == sq[:len(gen)] for sq in seqs)
        if precision == "f32":
            assert abs(float(lps[i]) - want[i][1]) < 2e-6 * abs(want[i][1]) + 1e-3, (float(lps[i]), want[i][1])
    # a second call (graph replay) resets the cursors
    out2 = model({"image": frames[0].cuda()})
    assert torch.equal(out2["predictions"].cpu(), preds)
