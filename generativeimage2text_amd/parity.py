"""Row-by-row comparison of generated token ids with reference ids, given the reference's decision margins.

Used by the GPU parity tests and by bench.py's `parity` field.  A bf16 pipeline cannot reproduce fp32 ids at a step
whose top candidates are closer than its own logit error; what CAN be demanded -- and is asserted here -- is that a row
leaves the reference only at such a near-tie, and that rows without one are identical token for token.

`step_margin[b, s]` (tests/golden/*.npz, written by oracle/make_golden.py) is the fp32 decision margin of search
step s for image b: the smallest gap between neighbours among the candidates the step's top-k keeps (greedy: top-1 vs
top-2 log-prob after the no-repeat rule, decoder.py:330-366; beam: the top 2k+1 of the flattened scores, :1175).
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np

# ---- fixed acceptance constants of the bf16 engine mode ------------------------------------------------------------
# None of these is derived from the engine under test (round-2 review: a threshold of "4 x the measured error" widens its
# own acceptance band when a kernel regresses).  They were set ONCE from the measurements of profiles/r03_a_parity_measured.jsonl
# (MI355X, both residual-stream storage types):
#   thr        cap on the margin below which a row may leave the reference's ids (x 2 for beam search: summed log-probs); for
#              one-beam searches the threshold actually applied is min(thr, 2 x the logit-error bound): margin_threshold().
#              Largest margin at which a row actually diverged: 0.0078 (greedy), 0.027 (beam)
#   ferr       bound on the visual-feature error (unit-variance LayerNorm outputs; measured 0.015 - 0.023)
#   lerr_frac  bound on the teacher-forced logit error as a fraction of the logit span: 1.5 x the measured value of every
#              golden case (LERR_FRAC; the benchmark's identity-LayerNorm weights measure 0.9e-3, the oracle's weights with
#              perturbed LayerNorm gains / successor structure 2e-3 ... 4.9e-3); the per-geometry value is the fallback for
#              cases without an entry
BF16_BOUNDS = {
    "TINY":            {"thr": 0.06, "lerr_frac": 4.0e-3, "ferr": 0.04},
    "TINY_VIDEO":      {"thr": 0.06, "lerr_frac": 4.0e-3, "ferr": 0.04},
    "TINY_L":          {"thr": 0.06, "lerr_frac": 4.0e-3, "ferr": 0.04},
    "GIT_BASE":        {"thr": 0.05, "lerr_frac": 7.0e-3, "ferr": 0.04},
    "GIT_BASE_VATEX":  {"thr": 0.05, "lerr_frac": 7.0e-3, "ferr": 0.04},
    "GIT_BASE_VQAv2":  {"thr": 0.05, "lerr_frac": 7.0e-3, "ferr": 0.04},
    "GIT_LARGE":       {"thr": 0.05, "lerr_frac": 7.0e-3, "ferr": 0.04},
}
LERR_FRAC = {
    "tiny_greedy_early_return": 0.0039, "tiny_greedy_untied": 0.004, "tiny_greedy_long": 0.0034, "tiny_beam4": 0.0038,
    "tiny_beam4_noeos": 0.0037, "tiny_beam4_early_done": 0.0027, "tiny_beam3_pn3": 0.0035, "tiny_ar_beam3": 0.0031,
    "tiny_prefix_greedy": 0.0041, "tiny_prefix_beam4": 0.0063, "tiny_video_greedy": 0.0031, "tiny_video_beam4": 0.0025,
    "tiny_image_two_frames": 0.0036, "tinyl_greedy": 0.0038, "tiny_varres_up": 0.004, "tiny_varres_down_beam4": 0.0035,
    "tiny_varres_prefix": 0.003, "tinyl_varres": 0.0041, "base_greedy": 0.0052, "base_greedy_eos": 0.0057,
    "base_beam4": 0.0053, "base_prefix_beam4": 0.0074, "large_greedy": 0.0071, "vatex_greedy": 0.007,
    "vqa_base_480x640": 0.0063, "full_bench_b64_greedy": 0.0015, "full_bench_b64_beam4": 0.0015,
    "full_base_b64_greedy": 0.0072, "full_base_b64_beam4": 0.0053, "full_large_b32_greedy": 0.0016,
    "full_vatex_b16_greedy": 0.0017,
}
# floors on rows whose ids equal the reference's token for token, per full-batch golden: a regression guard, 1-2 rows below
# the LOWEST count measured over the kernel variants of rounds 3-4 (any re-ordering of fp32 partial sums moves a near-tie row
# or two: the wide LayerNorm kernel took VATEX from 13 to 12 of 16 and the benchmark rows from 49 to 50 of 64;
# full_base_b64_greedy uses the oracle's perturbed-LayerNorm weights: 5x the logit error of the benchmark's weights, hence
# fewer identical rows).  A kernel regression that loses rows fails here even when every lost row has a near-tie somewhere in
# its 19 steps.  With random-init weights every row of these cases has such a near-tie (median row-minimum margin 0.006), so
# the floor is the only bound with teeth there; the case on which identity IS decidable -- reference margins >= 0.1 on every
# decision of every row -- is full_wide_b64_greedy, where the bf16 engine must return 64 of 64 rows (IDENTICAL_REQUIRED).
IDENTICAL_FLOORS = {       # measured over the kernel variants of rounds 3-4: 49-52, 42-46, 56-60, 59-61, 26-28, 12-14
    "full_bench_b64_greedy": 47, "full_base_b64_greedy": 40, "full_base_b64_beam4": 52, "full_bench_b64_beam4": 58,
    "full_large_b32_greedy": 24, "full_vatex_b16_greedy": 11,
}
# goldens whose reference margins are wide on every decision (oracle/make_golden.py asserts >= 0.2 when it freezes them): every
# row must equal the reference's ids in EVERY precision -- north_star's identity clause in the regime where it is decidable
IDENTICAL_REQUIRED = ("full_wide_b64_greedy", "full_wide_large_b32_greedy", "full_wide_vatex_b16_greedy")     # cfg2, cfg4, cfg5
# the same weights and images under beam 4 (full_wide_b64_beam4): a beam step keeps 2k = 8 candidates whose runner-ups are
# Gaussian-close for any weights (median adjacent gap 0.01), so no margin certificate exists; measured 62 of 64 rows in
# bf16 and 63 in f16, solo and serving shapes alike (profiles/r04_f_parity_measured.jsonl), f32 mode 64 of 64
WIDE_BEAM_FLOOR = 60


# fp16-operand build (libgitmi_f16.so, precision "f16"): the same kernels with 3 more mantissa bits per operand.  Bounds =
# the bf16 bounds scaled (measured: logit error 4-8x smaller, profiles/r03_*_parity_measured.jsonl), thresholds and floors
# of their own.
F16_SCALE = {"lerr": 0.2, "ferr": 0.3, "thr": 0.4}       # measured ratios to the bf16 build: logit error 0.11-0.15, features 0.2-0.4
IDENTICAL_FLOORS_F16 = {                                   # measured (profiles/r03_i_parity_measured.jsonl): 60, 59, 61, 64, 31, 16
    "full_bench_b64_greedy": 58, "full_base_b64_greedy": 56, "full_base_b64_beam4": 58, "full_bench_b64_beam4": 61,
    "full_large_b32_greedy": 29, "full_vatex_b16_greedy": 14,
}


def margin_threshold(config_name: str, logit_err_bound: float, chained: bool, precision: str = "bf16") -> float:
    """Margin below which a row may leave the reference's ids.
    One beam (not chained): until its first divergence a row is fed exactly the reference's tokens, so its logits are the
    teacher-forced logits, asserted elsewhere to lie within `logit_err_bound` (absolute) of the reference's; log-softmax shifts
    all logits of a row alike, so a decision can flip only if its fp32 margin is below 2 x logit_err_bound.  The threshold is
    that bound, capped by the fixed per-geometry constant (which is tighter on the wide-span oracle weights) -- it follows
    from the logit bound instead of being calibrated on observed divergences.
    Beam search (chained): candidates are SUMS of up to T log-probs, for which 2 x err x T is far too loose to be a test; the
    fixed constant x 2 stays (largest margin at which a beam row was ever observed to diverge: 0.027)."""
    thr = bf16_bounds(config_name)["thr"] * (F16_SCALE["thr"] if precision == "f16" else 1.0)
    if chained:
        return 2.0 * thr
    return min(thr, 2.0 * float(logit_err_bound))


def lerr_frac_bound(case: str, config_name: str, precision: str = "bf16") -> float:
    """Logit-error bound (fraction of the logit span) of a golden case."""
    b = LERR_FRAC.get(case, bf16_bounds(config_name)["lerr_frac"])
    return b * F16_SCALE["lerr"] if precision == "f16" else b


def bf16_bounds(config_name: str) -> Dict[str, float]:
    """Constants for a model geometry; names that extend a known one (GIT_LARGE_COCO, vqa_small ...) fall back to
    their family."""
    if config_name in BF16_BOUNDS:
        return BF16_BOUNDS[config_name]
    for key in ("GIT_LARGE", "GIT_BASE_VATEX", "GIT_BASE_VQAv2", "GIT_BASE", "TINY_VIDEO", "TINY_L", "TINY"):
        if config_name.startswith(key):
            return BF16_BOUNDS[key]
    return BF16_BOUNDS["GIT_BASE"]


def ids_parity(got: np.ndarray, ref: np.ndarray, step_margin: np.ndarray, thr: float, chained: bool,
               first_decision_pos: int = 1, min_identical: Optional[int] = None) -> Dict[str, float]:
    """got / ref: int [B, L*] id matrices in the reference's return convention.
    chained=False (greedy, one beam): decision s wrote position first_decision_pos + s; a row must equal the
        reference up to the first decision whose margin is below thr.
    chained=True (beam search): the decisions of an image are coupled through beam re-ordering, so a row may differ
        only if SOME decision margin of its image is below thr.
    min_identical: floor on the rows that must equal the reference token for token (IDENTICAL_FLOORS).
    Raises AssertionError on a violation; returns the counts."""
    B = ref.shape[0]
    identical, safe, worst = 0, 0, 0.0
    for r in range(B):
        m = step_margin[r]
        row_safe = bool((m >= thr).all())
        safe += row_safe
        L = min(got.shape[1], ref.shape[1])
        diff = [t for t in range(L) if got[r, t] != ref[r, t]]
        if got.shape[1] != ref.shape[1] and not diff:
            diff = [L]
        if not diff:
            identical += 1
            continue
        assert not row_safe, f"row {r}: ids differ although every decision margin >= {thr:.4f}"
        if chained:
            worst = max(worst, float(m[m < thr].max()))
        else:
            s_idx = diff[0] - first_decision_pos
            assert 0 <= s_idx < m.shape[0], (r, diff[0], m.shape)
            assert m[s_idx] < thr, (f"row {r}: first divergence at position {diff[0]} where the fp32 margin is "
                                    f"{m[s_idx]:.4f} >= {thr:.4f}")
            worst = max(worst, float(m[s_idx]))
    assert identical >= safe
    if min_identical is not None:
        assert identical >= min_identical, f"only {identical} of {B} rows equal the reference ids (floor {min_identical})"
    return {"rows": int(B), "identical": int(identical), "safe_rows": int(safe),
            "first_divergence_margin_max": round(worst, 5), "threshold": round(float(thr), 5)}


# ---- teacher-forced decisions (round 5) ---------------------------------------------------------------------------------
# ids_parity() can follow a free-running 16-bit row only to its first near-tie (mean: decision 2.4 of 19 on the benchmark
# fixture, 152 of 1 216 decisions).  gitmi_step_logits is the reference's `step` callable: fed the REFERENCE's ids[:, :t]
# for t = 1 .. L-1 it makes EVERY decision of every row comparable, on the benchmark's own weights.
#   * every live decision whose fp32 margin is >= thr must pick the reference's id                        (`decidable`)
#   * the logit error is measured on every row at every decision: against the frozen reference values (top-8 logits +
#     128 sampled columns per decision, tests/golden/<case>_tf.npz) and, when an f32-mode engine is supplied, over ALL
#     vocabulary columns against that engine's logits -- itself asserted to lie within 1e-4 of the frozen reference values
#     on the same entries (so the all-column figure is anchored to the reference, not to the engine family)
# TF_LERR_BOUND: fixed absolute bounds on that all-entry maximum, set once from profiles/r05_a_parity_measured.jsonl
# (largest value over solo / serving shapes x 1.25); the decision threshold is 2 x the bound, exactly the argument of
# margin_threshold(): within the bound, log-softmax shifts a row alike, so only a decision with margin < 2 x bound can flip.
TF_LERR_BOUND: Dict[str, Dict[str, float]] = {
    # measured all-entry maxima (rows x 30 522 columns x decisions; solo == serving shapes): bf16 0.01326 / 0.09214 / 0.01512 /
    # 0.01405 / 0.02387 / 0.02849 / 0.0239, f16 0.00182 / 0.01086 / 0.00193 / 0.00175 / 0.00334 / 0.00362 / 0.00257
    "bf16": {"full_bench_b64_greedy": 0.0166, "full_base_b64_greedy": 0.115, "full_large_b32_greedy": 0.0189,
             "full_vatex_b16_greedy": 0.0176, "full_wide_b64_greedy": 0.030, "full_wide_large_b32_greedy": 0.0356,
             "full_wide_vatex_b16_greedy": 0.030},
    "f16": {"full_bench_b64_greedy": 0.0023, "full_base_b64_greedy": 0.0136, "full_large_b32_greedy": 0.0024,
            "full_vatex_b16_greedy": 0.0022, "full_wide_b64_greedy": 0.0042, "full_wide_large_b32_greedy": 0.0045,
            "full_wide_vatex_b16_greedy": 0.0032},
}


def tf_bounds(case: str, config_name: str, precision: str, span: float) -> Dict[str, float]:
    """(logit-error bound, decision threshold) of a teacher-forced case.  f32: 1e-4 / 1e-3 (every decision of the frozen
    fixtures has a margin far above; ids must agree wherever the fp32 margin exceeds the engine's own rounding)."""
    if precision == "f32":
        return {"lerr": 1e-4, "thr": 1e-3}
    b = TF_LERR_BOUND.get(precision, {}).get(case)
    if b is None:                                   # no pinned figure: the sampled-row bound of the free-running test
        b = lerr_frac_bound(case, config_name, precision) * span
    return {"lerr": float(b), "thr": 2.0 * float(b)}


def teacher_forced_parity(step_logits, ref_ids: np.ndarray, tf_gold, eos: int, thr: float, lerr_bound: float,
                          f32_step_logits=None) -> Dict[str, float]:
    """step_logits(tokens int64 [B, t]) -> fp32 [B, V] torch tensor (Engine.step_logits).  ref_ids: the reference's greedy
    ids [B, L] incl. the start token; tf_gold: the arrays of <case>_tf.npz.  Decision s (0-based) reads ids[:, :s+1] and
    chooses ids[:, s+1]; the no-repeat rule (-10000 on the last token, decoder.py:330) applies from the second decision on.
    Returns counts and errors; `violation` names the first broken rule (nothing is raised: callers assert on it)."""
    import torch
    B, L = ref_ids.shape
    ids = torch.from_numpy(np.ascontiguousarray(ref_ids)).long()
    live = np.asarray(tf_gold["live"]).astype(bool)
    margin = np.asarray(tf_gold["margin"], dtype=np.float64)
    top_ids, top_vals = np.asarray(tf_gold["top_ids"]), np.asarray(tf_gold["top_vals"])
    cols, col_vals = np.asarray(tf_gold["cols"]), np.asarray(tf_gold["col_vals"])
    agree = np.zeros((B, L - 1), dtype=bool)
    err_frozen = err_all = err_f32_frozen = 0.0
    err_where = None
    for s in range(L - 1):
        lg = step_logits(ids[:, :s + 1]).float()
        dev = lg.device
        idx = torch.cat([torch.from_numpy(top_ids[:, s].astype(np.int64)),
                         torch.from_numpy(cols[s].astype(np.int64))[None].expand(B, -1)], dim=1).to(dev)
        frozen = torch.cat([torch.from_numpy(top_vals[:, s]), torch.from_numpy(col_vals[:, s])], dim=1).to(dev)
        e = (lg.gather(1, idx) - frozen).abs().max().item()
        err_frozen = max(err_frozen, e)
        if f32_step_logits is not None:
            l32 = f32_step_logits(ids[:, :s + 1]).float()
            err_f32_frozen = max(err_f32_frozen, (l32.gather(1, idx) - frozen).abs().max().item())
            d = (lg - l32).abs()
            e_all = d.max().item()
            if e_all > err_all:
                flat = int(d.argmax().item())
                err_all, err_where = e_all, (flat // d.shape[1], s, flat % d.shape[1])
        dec = lg.clone()
        if s >= 1:
            dec.scatter_(1, ids[:, s:s + 1].to(dev), -10000.0)
        agree[:, s] = (dec.argmax(dim=1).cpu().numpy() == ref_ids[:, s + 1])
    decidable = live & (margin >= thr)
    flipped = live & ~agree
    out = {"decisions": int(live.sum()), "decidable": int(decidable.sum()), "agree_decidable": int((decidable & agree).sum()),
           "agree": int((live & agree).sum()), "threshold": round(float(thr), 5),
           "max_flipped_margin": round(float(margin[flipped].max()), 5) if flipped.any() else 0.0,
           "rows_all_agree": int((agree | ~live).all(axis=1).sum()), "rows": int(B),
           "max_logit_err_frozen": round(err_frozen, 5), "logit_err_bound": round(float(lerr_bound), 5),
           "logit_span": round(float(tf_gold["logit_max"]) - float(tf_gold["logit_min"]), 3)}
    if f32_step_logits is not None:
        out["max_logit_err"] = round(err_all, 5)                         # every row x every column x every decision
        out["max_logit_err_at"] = list(err_where) if err_where else None
        out["f32_mode_vs_reference"] = round(err_f32_frozen, 7)
    worst = max(err_all, err_frozen)
    out["max_logit_err_frac_of_span"] = round(worst / out["logit_span"], 6)
    viol = None
    if int((decidable & ~agree).sum()):
        r, s = [int(v[0]) for v in np.nonzero(decidable & ~agree)]
        viol = f"row {r} decision {s}: engine leaves the reference id at an fp32 margin of {margin[r, s]:.4f} >= {thr:.4f}"
    elif worst > lerr_bound:
        viol = f"teacher-forced logit error {worst:.5f} above the bound {lerr_bound:.5f}"
    elif f32_step_logits is not None and err_f32_frozen > 1e-4:
        viol = f"f32 engine mode is {err_f32_frozen:.2e} from the frozen reference logits (> 1e-4)"
    out["ok"] = viol is None
    if viol:
        out["violation"] = viol
    return out
