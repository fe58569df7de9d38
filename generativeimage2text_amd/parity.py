"""Row-by-row comparison of generated token ids with reference ids, given the reference's decision margins.

Used by the GPU parity tests and by bench.py's `parity` field.  A bf16 pipeline cannot reproduce fp32 ids at a step
whose top candidates are closer than its own logit error; what CAN be demanded -- and is asserted here -- is that a row
leaves the reference only at such a near-tie, and that rows without one are identical token for token.

`step_margin[b, s]` (tests/golden/*.npz, written by oracle/make_golden.py) is the fp32 decision margin of search
step s for image b: the smallest gap between neighbours among the candidates the step's top-k keeps (greedy: top-1 vs
top-2 log-prob after the no-repeat rule, decoder.py:330-366; beam: the top 2k+1 of the flattened scores, :1175).
"""
from __future__ import annotations

from typing import Dict

import numpy as np


def ids_parity(got: np.ndarray, ref: np.ndarray, step_margin: np.ndarray, thr: float, chained: bool,
               first_decision_pos: int = 1) -> Dict[str, float]:
    """got / ref: int [B, L*] id matrices in the reference's return convention.
    chained=False (greedy, one beam): decision s wrote position first_decision_pos + s; a row must equal the
        reference up to the first decision whose margin is below thr.
    chained=True (beam search): the decisions of an image are coupled through beam re-ordering, so a row may differ
        only if SOME decision margin of its image is below thr.
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
    return {"rows": int(B), "identical": int(identical), "safe_rows": int(safe),
            "first_divergence_margin_max": round(worst, 5), "threshold": round(float(thr), 5)}
