#!/usr/bin/env python
"""Where does the wall time of the mixed schedule go?  Timeline analysis of a rocprofv3 --kernel-trace database
(rocpd sqlite, view `kernels`: name, start, end, queue_id, stream_id; ns) of `bench.py` with several contexts in flight.

    rocprofv3 --kernel-trace -d out -o bench -- python bench.py --no-cpu-baseline --steps 40
    python tools/mix_timeline.py out/bench_results.db [--lo 0.35 --hi 0.75] [--solo solo_results.db]

The kernels of a pass fall into two classes: ENCODER (image encoder + decoder prefill: the MFMA GEMMs, LayerNorms, full
attention, patchify, K/V repack) and DECODE (the per-step chain).  Reported for the window [lo, hi] of the trace (default:
the middle, i.e. the timed steady state, not the warm-up or the roofline passes at the end):

  * concurrency profile: share of the wall time with 0 / 1 / 2 / 3+ kernels running, with an encoder GEMM running,
    with only decode kernels running, with nothing running;
  * per class: busy time (union), summed kernel time, and the average number of its kernels in flight;
  * start delays: for every kernel, the time between the end of the previous kernel ON ITS OWN QUEUE and its start
    (what a dependent launch waited), split by class and by what was running on the OTHER queues in that interval;
  * duration inflation of the encoder GEMMs against a solo trace (--solo: one context, nothing beside it);
  * per batch (a batch = one patchify launch on a queue up to the next): encoder span, decode span, whole latency.
"""
from __future__ import annotations

import argparse
import bisect
import collections
import sqlite3
import statistics

ENCODER_MARKS = ("gemm_p8", "gemm_ring", "gemm_dlds", "gemm_kernel", "layernorm_kernel", "layernorm_wide", "attn_full", "im2col", "vit_assemble",
                 "kv_repack", "convert_pad", "pos_bicubic")
DECODE_MARKS = ("dgemm_kernel", "attn_decode", "vocab_topm", "search_step", "search_init", "search_finish", "embed_ln",
                "row_topm", "sample_rows", "fill_start", "fill_i32", "load_ids")


def classify(name: str) -> str:
    if any(m in name for m in DECODE_MARKS):
        return "decode"
    if any(m in name for m in ENCODER_MARKS):
        return "encoder"
    return "other"


def load(db: str):
    con = sqlite3.connect(db)
    rows = con.execute("select name, start, end, queue_id, stream_id from kernels order by start").fetchall()
    return [dict(name=r[0], start=int(r[1]), end=int(r[2]), queue=r[3], stream=r[4], cls=classify(r[0])) for r in rows]


def union_length(intervals):
    total, cur_s, cur_e = 0, None, None
    for s, e in sorted(intervals):
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                total += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        total += cur_e - cur_s
    return total


def concurrency_profile(ks, t0, t1):
    """Sweep over kernel start / end events inside [t0, t1]."""
    ev = []
    for k in ks:
        s, e = max(k["start"], t0), min(k["end"], t1)
        if s < e:
            gemm = 1 if ("gemm_p8" in k["name"] or "gemm_ring" in k["name"]) else 0
            ev.append((s, 1, k["cls"], gemm))
            ev.append((e, -1, k["cls"], gemm))
    ev.sort(key=lambda x: (x[0], x[1]))
    n = collections.Counter()
    by_count = collections.Counter()
    with_gemm = only_decode = 0
    last = t0
    for t, d, cls, gemm in ev:
        dt = t - last
        if dt > 0:
            tot = n["encoder"] + n["decode"] + n["other"]
            by_count[min(tot, 3)] += dt
            if n["gemm"] > 0:
                with_gemm += dt
            if tot > 0 and n["encoder"] == 0 and n["other"] == 0:
                only_decode += dt
        n[cls] += d
        n["gemm"] += d * gemm
        last = t
    by_count[0] += max(0, t1 - last)
    return by_count, with_gemm, only_decode


def start_delays(ks, t0, t1):
    """delay = start - end of the previous kernel on the same queue; what ran on the other queues meanwhile."""
    per_queue = collections.defaultdict(list)
    for k in ks:
        per_queue[k["queue"]].append(k)
    others_by_queue = {}
    for q in per_queue:
        iv = sorted((o["start"], o["end"], o["cls"]) for o in ks if o["queue"] != q)
        others_by_queue[q] = (iv, [x[0] for x in iv])
    out = collections.defaultdict(list)      # (cls, what) -> delays in us
    for q, lst in per_queue.items():
        lst.sort(key=lambda k: k["start"])
        iv, starts = others_by_queue[q]
        for prev, k in zip(lst, lst[1:]):
            if k["start"] < t0 or k["start"] > t1:
                continue
            gap0, gap1 = prev["end"], k["start"]
            delay = (gap1 - gap0) / 1e3
            if gap1 <= gap0:
                out[(k["cls"], "none (back to back)")].append(0.0)
                continue
            # what overlapped [gap0, gap1] on other queues
            hi = bisect.bisect_left(starts, gap1)
            cover = collections.Counter()
            for s, e, c in iv[max(0, hi - 64):hi]:
                ov = min(e, gap1) - max(s, gap0)
                if ov > 0:
                    cover[c] += ov
            what = "idle" if not cover else max(cover, key=cover.get)
            out[(k["cls"], "other queues: " + what)].append(delay)
    return out


def batches(ks, t0, t1):
    """per queue: a batch starts at a patchify (im2col) launch."""
    per_queue = collections.defaultdict(list)
    for k in ks:
        per_queue[k["queue"]].append(k)
    res = []
    for q, lst in per_queue.items():
        lst.sort(key=lambda k: k["start"])
        idx = [i for i, k in enumerate(lst) if "im2col" in k["name"]]
        for a, b in zip(idx, idx[1:] + [len(lst)]):
            seg = lst[a:b]
            if seg[0]["start"] < t0 or seg[-1]["end"] > t1:
                continue
            enc = [k for k in seg if k["cls"] == "encoder"]
            dec = [k for k in seg if k["cls"] == "decode"]
            if not enc or not dec:
                continue
            res.append(dict(queue=q, enc_ms=(enc[-1]["end"] - enc[0]["start"]) / 1e6,
                            dec_ms=(dec[-1]["end"] - dec[0]["start"]) / 1e6,
                            lat_ms=(seg[-1]["end"] - seg[0]["start"]) / 1e6,
                            enc_busy_ms=sum(k["end"] - k["start"] for k in enc) / 1e6,
                            dec_busy_ms=sum(k["end"] - k["start"] for k in dec) / 1e6))
    return res


def pct(x, tot):
    return "%5.1f%%" % (100.0 * x / tot if tot else 0.0)


def describe(vals):
    if not vals:
        return "n=0"
    vs = sorted(vals)
    return "n=%d mean %.1f us  median %.1f  p90 %.1f  max %.1f  sum %.2f ms" % (
        len(vs), statistics.fmean(vs), vs[len(vs) // 2], vs[int(0.9 * (len(vs) - 1))], vs[-1], sum(vs) / 1e3)


def report(db, lo=0.35, hi=0.75, solo=None):
    ks = load(db)
    if not ks:
        return "no kernels in " + db
    T0, T1 = ks[0]["start"], max(k["end"] for k in ks)
    t0, t1 = T0 + int(lo * (T1 - T0)), T0 + int(hi * (T1 - T0))
    wall = t1 - t0
    win = [k for k in ks if k["end"] > t0 and k["start"] < t1]
    lines = ["# mix timeline of %s" % db,
             "window %.1f .. %.1f ms of a %.1f ms trace (%d of %d kernels, %d queues)" % (
                 (t0 - T0) / 1e6, (t1 - T0) / 1e6, (T1 - T0) / 1e6, len(win), len(ks), len({k['queue'] for k in win}))]
    by_count, with_gemm, only_decode = concurrency_profile(win, t0, t1)
    lines.append("kernels in flight:  0: %s   1: %s   2: %s   3+: %s" % tuple(pct(by_count[i], wall) for i in range(4)))
    lines.append("an encoder GEMM is running %s of the time; only decode kernels run %s; nothing runs %s" % (
        pct(with_gemm, wall), pct(only_decode, wall), pct(by_count[0], wall)))
    for cls in ("encoder", "decode", "other"):
        iv = [(max(k["start"], t0), min(k["end"], t1)) for k in win if k["cls"] == cls]
        if not iv:
            continue
        busy, summed = union_length(iv), sum(e - s for s, e in iv)
        lines.append("%-8s busy (union) %s of the wall, summed kernel time %.2fx the wall, %.2f in flight while busy" % (
            cls, pct(busy, wall), summed / wall, summed / busy if busy else 0.0))
    nb = batches(ks, t0, t1)
    if nb:
        lines.append("batches completed in the window: %d -> %.3f ms of wall per batch" % (len(nb), wall / 1e6 / len(nb)))
        for key, label in (("enc_ms", "encoder span"), ("enc_busy_ms", "encoder kernel time"), ("dec_ms", "decode span"),
                           ("dec_busy_ms", "decode kernel time"), ("lat_ms", "batch latency")):
            v = sorted(b[key] for b in nb)
            lines.append("  %-20s median %.3f ms  (min %.3f, max %.3f)" % (label, v[len(v) // 2], v[0], v[-1]))
    lines.append("start delay of a kernel behind the previous kernel of its own queue, by what the other queues ran meanwhile:")
    sd = start_delays(ks, t0, t1)
    for (cls, what), vals in sorted(sd.items()):
        lines.append("  %-8s | %-24s %s" % (cls, what, describe(vals)))
    if solo:
        base = collections.defaultdict(list)
        for k in load(solo):
            base[k["name"]].append((k["end"] - k["start"]) / 1e3)
        lines.append("duration of the encoder GEMMs against the solo trace %s:" % solo)
        cur = collections.defaultdict(list)
        for k in win:
            if "gemm_p8" in k["name"]:
                cur[k["name"]].append((k["end"] - k["start"]) / 1e3)
        for name, vals in sorted(cur.items(), key=lambda kv: -sum(kv[1])):
            if name in base:
                a, b = statistics.fmean(vals), statistics.fmean(base[name])
                lines.append("  %7.1f us vs %7.1f us solo (x%.3f, n=%d)  %s" % (a, b, a / b, len(vals), name[:90]))
    return "\n".join(lines) + "\n"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--lo", type=float, default=0.35)
    ap.add_argument("--hi", type=float, default=0.75)
    ap.add_argument("--solo", default=None)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    text = report(a.db, a.lo, a.hi, a.solo)
    if a.out:
        open(a.out, "w").write(text)
    print(text, end="")


if __name__ == "__main__":
    main()
