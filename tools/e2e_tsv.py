#!/usr/bin/env python
"""End-to-end throughput of the reference's own entry point (VERDICT r05 "missing" 2 / "next" 3):

    TSV row -> base64 -> JPEG decode (PIL, host threads) -> upload -> GPU resize / crop / normalise -> batch of 64 ->
    ViT + prefill + 19 decode steps (several batches in flight) -> token ids -> caption strings -> TSV row

i.e. `test_git_inference_single_tsv` (reference inference.py:134-225) as shipped in generativeimage2text_amd/inference.py, on N
synthetic 640x480 JPEG rows, GIT_BASE, random-init weights, the id-passthrough tokenizer (no vocabulary offline).  One JSON line:

    e2e captions/s per host-thread count, the GPU-only rate of the same pipeline (batches already resident: what the host has
    to keep up with), GPU-busy fraction = e2e / GPU-only, what ONE host thread decodes per second, and from those the host
    threads one GPU needs; plus the check that the pipelined task writes exactly the rows the serial one-image-per-call path
    (contexts = 1, batch_size = 1, no decode threads: the reference's loop) writes.

    python tools/e2e_tsv.py [--rows 2048] [--procs 8,16,32,48] [--threads 16] [--precision f16] [--out profiles/rNN_e2e_tsv.json]
    python bench.py --e2e-tsv 2048
"""
from __future__ import annotations

import argparse
import base64
import io
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def make_jpeg_rows(n_rows: int, distinct: int = 128, seed: int = 0):
    """`distinct` different 640x480 photographs-like JPEGs (smooth structure + texture: ~60-90 KB at quality 90), cycled to n_rows."""
    from PIL import Image
    rng = np.random.RandomState(seed)
    raws = []
    for _ in range(distinct):
        low = rng.rand(15, 20, 3)
        img = Image.fromarray((low * 255).astype(np.uint8)).resize((640, 480), Image.BICUBIC)
        arr = np.asarray(img).astype(np.float32) + rng.randn(480, 640, 3) * 12.0
        buf = io.BytesIO()
        Image.fromarray(np.clip(arr, 0, 255).astype(np.uint8)).save(buf, format="JPEG", quality=90)
        raws.append(base64.b64encode(buf.getvalue()).decode())
    return [["img%06d" % i, raws[i % distinct]] for i in range(n_rows)], sum(len(r) for r in raws) * 3 // 4 // distinct


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=2048)
    ap.add_argument("--procs", default="8,16,32,48", help="worker-process counts of the pooled path (GIT_DECODE_PROCS)")
    ap.add_argument("--threads", default="16", help="thread counts of the per-image thread-pool path (GIT_DECODE_PROCS=0)")
    ap.add_argument("--precision", default="f16")
    ap.add_argument("--contexts", type=int, default=4)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--check-rows", type=int, default=64, help="rows compared with the serial one-image-per-call path")
    ap.add_argument("--out", default=None)
    args = ap.parse_args(argv)
    os.environ["GIT_VOCAB"] = "ids"
    from generativeimage2text_amd import inference as I, tsv_io
    from generativeimage2text_amd.configs import config_for_model
    from generativeimage2text_amd.model import AutoRegressiveBeamSearch
    from generativeimage2text_amd.synthetic import random_state_dict

    cfg = config_for_model("GIT_BASE")
    # the wide-margin weight family (tests/golden/full_wide_*): decisions are decidable for a 16-bit pipeline, so "the pipelined
    # task writes the rows the serial path writes" is a meaningful check; throughput does not depend on the weights' values
    weights = random_state_dict(cfg, seed=1250, successor=1.0)
    real_build = I.build_model
    # BASELINE's workload: greedy, max_len 20 (the task's shipped default is beam 4 / 1024 steps: model.py:34-40)
    I.build_model = lambda name, tok, c, **kw: real_build(
        name, tok, c, decoder=AutoRegressiveBeamSearch(eos_index=102, max_steps=20, beam_size=1, per_node_beam_size=1,
                                                       fix_missing_prefix=True), **kw)
    tmp = tempfile.mkdtemp(prefix="e2e_tsv_")
    rows, jpeg_bytes = make_jpeg_rows(args.rows)
    in_tsv = os.path.join(tmp, "in.tsv")
    tsv_io.tsv_writer(rows, in_tsv)

    # what ONE host thread does per second: base64 -> JPEG decode -> RGB array (the part that runs on the thread pool)
    t0 = time.perf_counter()
    for _, b64 in rows[:100]:
        np.asarray(I.load_image_by_pil(base64.b64decode(b64)))
    one_thread = 100 / (time.perf_counter() - t0)

    runs = []
    out_first = None
    settings = [("procs", int(t)) for t in args.procs.split(",") if t] + [("threads", int(t)) for t in args.threads.split(",") if t]
    for kind, n in settings:
        os.environ["GIT_DECODE_PROCS"] = str(n) if kind == "procs" else "0"
        os.environ["GIT_DECODE_THREADS"] = str(n) if kind == "threads" else "16"
        st = {}
        out = os.path.join(tmp, "out_%s_%d.tsv" % (kind, n))
        I.test_git_inference_single_tsv(in_tsv, "GIT_BASE", None, out, checkpoint=weights, batch_size=args.batch,
                                        precision=args.precision, contexts=args.contexts, stats=st)
        runs.append({"host_path": "worker processes + batched GPU transform" if kind == "procs" else "thread pool + per-image GPU transform",
                     kind: n, "captions_per_s": round(st["images"] / st["run_s"], 1), "run_s": round(st["run_s"], 3),
                     "build_s": round(st["build_s"], 2), "batches": st["batches"], "staging_pinned": st.get("staging_pinned"),
                     "steady_captions_per_s": round(st["steady_captions_per_s"], 1) if "steady_captions_per_s" in st else None,
                     "parent_s": {k: round(st[k], 3) for k in ("pool_start_s", "first_batch_ready_s", "wait_decode_s", "upload_s",
                                                                "transform_s", "submit_s", "result_s", "device_wait_s") if k in st}})
        print("%s %3d: %.1f captions/s end to end (%.2f s for %d rows; steady state %s)" % (
            kind, n, runs[-1]["captions_per_s"], st["run_s"], st["images"], runs[-1]["steady_captions_per_s"]), file=sys.stderr, flush=True)
        got = [r for r in tsv_io.tsv_reader(out)]
        assert [r[0] for r in got] == [r[0] for r in rows], "row order"
        if out_first is None:
            out_first = got
        else:
            assert got == out_first, "the task's output must not depend on how the host side decodes"

    # the serial path of the reference: one image per model call, no thread pool, one context
    n_chk = min(args.check_rows, args.rows)
    chk_tsv = os.path.join(tmp, "chk.tsv")
    tsv_io.tsv_writer(rows[:n_chk], chk_tsv)
    os.environ["GIT_DECODE_THREADS"] = "0"
    os.environ["GIT_DECODE_PROCS"] = "0"
    st1 = {}
    I.test_git_inference_single_tsv(chk_tsv, "GIT_BASE", None, os.path.join(tmp, "chk_out.tsv"), checkpoint=weights, batch_size=1,
                                    precision=args.precision, contexts=1, stats=st1)
    serial = [r for r in tsv_io.tsv_reader(os.path.join(tmp, "chk_out.tsv"))]
    same = sum(a == b for a, b in zip(serial, out_first[:n_chk]))

    # GPU-only: the same pipeline (contexts, streams, batch size) on batches that are already resident in HBM
    tok = I.IdTokenizer()
    model = I.build_model("GIT_BASE", tok, weights, max_batch=args.batch, precision=args.precision)
    model.set_pipeline(args.contexts)
    g = torch.Generator().manual_seed(0)
    batch = torch.randn(args.batch, 3, 224, 224, generator=g).cuda()
    n_b = max(8, args.rows // args.batch)
    import collections
    pend = collections.deque()
    for it in range(2):                                   # first pass: graph capture per context
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n_b):
            pend.append(model.submit({"image": batch}))
            while len(pend) > args.contexts:
                pend.popleft().result()
        while pend:
            pend.popleft().result()
        gpu_only = n_b * args.batch / (time.perf_counter() - t0)
    model.close()
    best = max(runs, key=lambda r: r["captions_per_s"])
    steady = max((r["steady_captions_per_s"] or 0.0) for r in runs)
    res = {"what": "test_git_inference_single_tsv end to end: TSV -> base64 -> PIL JPEG decode (host threads) -> GPU resize/crop/normalise -> "
                   "GIT_BASE greedy max_len 20 (ViT + prefill + 19 decode steps, %d batches of %d in flight) -> caption rows -> TSV"
                   % (args.contexts, args.batch),
           "rows": args.rows, "image": "640x480 JPEG q90, %d KB mean" % (jpeg_bytes // 1024), "precision": args.precision,
           "host_cpus": os.cpu_count(), "host_cpus_usable": I.effective_cpus(),
           "host_cpus_note": "usable = affinity mask capped by the cgroup CPU quota (cpu.max): what the workers AND the parent share",
           "runs": runs,
           "e2e_captions_per_s": best["captions_per_s"], "e2e_host_setting": {k: best[k] for k in ("procs", "threads") if k in best},
           "e2e_steady_state_captions_per_s": steady,
           "steady_state_note": "rate after the first batch is ready, i.e. without the start-up of the worker interpreters (0.3-0.5 s: it "
                                "dominates a 2-second run and vanishes in a 40k-image evaluation set)",
           "gpu_only_captions_per_s": round(gpu_only, 1),
           "gpu_busy_fraction": round(best["captions_per_s"] / gpu_only, 3),
           "gpu_busy_fraction_steady_state": round(steady / gpu_only, 3),
           "one_host_thread_images_per_s": round(one_thread, 1),
           "host_cores_to_saturate_one_gpu": int(np.ceil(gpu_only / (steady / max(1, I.effective_cpus())))) if steady else None,
           "host_cores_note": "GPU-only rate / (measured steady-state rate per usable core: workers, parent and the runtime's threads "
                              "together); decode alone (GPU-only rate / one core's base64 + JPEG -> RGB rate) would be %d"
                              % int(np.ceil(gpu_only / one_thread)),
           "serial_path": {"what": "contexts=1, batch_size=1, no decode threads: one image per model call, as the reference's loop",
                           "rows": n_chk, "captions_per_s": round(st1["images"] / st1["run_s"], 1),
                           "rows_identical_to_pipelined_output": same}}
    line = json.dumps(res)
    print(line, flush=True)
    if args.out:
        with open(args.out, "w") as f:
            f.write(line + "\n")
    return res


if __name__ == "__main__":
    main()
