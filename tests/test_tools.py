"""CPU: the profile-analysis tools on synthetic rocpd databases (same `kernels` view columns as rocprofv3 writes)."""
import os
import sqlite3
import sys

import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))


def _db(path, rows):
    con = sqlite3.connect(path)
    con.execute("create table kernels (name text, start integer, end integer, queue_id integer, stream_id integer, "
                "vgpr_count integer, accum_vgpr_count integer, lds_size integer)")
    con.executemany("insert into kernels values (?,?,?,?,?,0,0,0)", rows)
    con.commit()
    con.close()


def test_mix_timeline_on_a_known_schedule(tmp_path):
    import mix_timeline as M
    us = 1000
    rows = []
    # queue 1: two batches of [patchify 10us, GEMM 50us, LayerNorm 10us] back to back, then its decode kernels
    # queue 2: a decode chain of 5-us kernels with 5-us gaps that runs beside queue 1's encoder
    t = 0
    for b in range(2):
        base = b * 200 * us
        rows += [("gitmi::im2col_kernel<u16>", base, base + 10 * us, 1, 1),
                 ("gitmi::gemm_p8_kernel<float>", base + 10 * us, base + 60 * us, 1, 1),
                 ("gitmi::layernorm_kernel<u16>", base + 60 * us, base + 70 * us, 1, 1)]
        for i in range(4):                       # decode of this batch: 4 kernels of 5 us, 10 us apart
            s = base + 80 * us + i * 10 * us
            rows.append(("gitmi::dgemm_kernel<4,4,0>", s, s + 5 * us, 1, 1))
    for i in range(40):                          # queue 2: 5 us on, 5 us off over the whole 400 us
        rows.append(("gitmi::attn_decode_mfma_kernel<1,3>", i * 10 * us, i * 10 * us + 5 * us, 2, 2))
    db = str(tmp_path / "t.db")
    _db(db, rows)
    ks = M.load(db)
    assert {k["cls"] for k in ks} == {"encoder", "decode"}
    # whole trace as the window
    T0, T1 = 0, 395 * us
    by_count, with_gemm, only_decode = M.concurrency_profile(ks, T0, T1)
    assert with_gemm == 2 * 50 * us
    # encoder busy = 2 * 70 us; queue 2 is busy half of the time everywhere
    enc = [(k["start"], k["end"]) for k in ks if k["cls"] == "encoder"]
    assert M.union_length(enc) == 140 * us
    dec = [(k["start"], k["end"]) for k in ks if k["cls"] == "decode"]
    # queue 1's decode kernels [80..85, 90..95, ...] coincide with queue 2's [80..85, ...]: the union is queue 2's busy time
    assert M.union_length(dec) == 40 * 5 * us
    assert by_count[0] + by_count[1] + by_count[2] + by_count[3] == T1 - T0
    assert by_count[2] == (2 * 35 + 2 * 4 * 5) * us          # encoder beside queue 2 (half of 70 us) + coinciding decodes
    nb = M.batches(ks, T0, T1 + 1)
    assert len(nb) == 2 and abs(nb[0]["enc_ms"] - 0.070) < 1e-9 and abs(nb[0]["dec_ms"] - 0.035) < 1e-9
    sd = M.start_delays(ks, T0, T1)
    # queue 1's first decode kernel starts 10 us after the LayerNorm ended, the later ones 5 us after their predecessor
    d1 = sorted(v for (cls, what), vals in sd.items() if cls == "decode" for v in vals)
    assert d1.count(10.0) == 2 and d1.count(5.0) >= 6
    text = M.report(db, 0.0, 1.0)
    assert "batches completed in the window: 2" in text and "an encoder GEMM is running" in text


def test_rocprof_summary_on_synthetic_db(tmp_path):
    import rocprof_summary as R
    db = str(tmp_path / "s.db")
    _db(db, [("k_a", 0, 1000, 1, 1), ("k_a", 2000, 5000, 1, 1), ("k_b", 0, 500, 2, 2)])
    out = str(tmp_path / "s.txt")
    R.main(db, out)
    txt = open(out).read()
    assert "total GPU kernel time 0.004 ms over 3 dispatches" in txt.replace("0.0045", "0.004") or "3 dispatches" in txt
    assert "k_a" in txt and "k_b" in txt
    assert "# csrc_sha=" in txt                                   # the trace says which kernel sources it was taken for
    # a rocpd database with grid / workgroup columns: one row per (kernel, workgroup count) -- the same GEMM instantiation
    # serves several shapes
    db2 = str(tmp_path / "g.db")
    con = sqlite3.connect(db2)
    con.execute("create table kernels (name text, start integer, end integer, vgpr_count integer, accum_vgpr_count integer, "
                "lds_size integer, grid_size_x integer, workgroup_size_x integer)")
    con.executemany("insert into kernels values (?,?,?,0,0,0,?,?)",
                    [("gemm", 0, 1000, 512 * 450, 512), ("gemm", 0, 3000, 512 * 450, 512), ("gemm", 0, 700, 512 * 150, 512)])
    con.commit()
    con.close()
    out2 = str(tmp_path / "g.txt")
    R.main(db2, out2)
    txt2 = open(out2).read()
    assert "gemm  <<<450 WGs>>>" in txt2 and "gemm  <<<150 WGs>>>" in txt2 and txt2.count("WGs>>>") == 2


def _run_bench_with_fakes(monkeypatch, capsys, argv):
    """bench.main() with torch.cuda and the engine replaced by stand-ins: checks the script's control flow (every
    schedule flag) and the shape of its JSON line without a GPU.  Nothing here measures anything."""
    import contextlib, json, time, types
    import torch
    import bench
    from generativeimage2text_amd import engine as E, synthetic

    class FakeEvent:
        def __init__(self, enable_timing=False):
            self.t = None

        def synchronize(self):
            pass

        def record(self):
            self.t = time.perf_counter()

        def elapsed_time(self, other):
            return (other.t - self.t) * 1e3

    class FakeStream:
        def wait_event(self, ev):
            assert ev.t is not None, "waiting on an event that was never recorded"

        def synchronize(self):
            pass

    log = []

    class FakeEngine:
        make_search = staticmethod(E.Engine.make_search)
        generate_coalesced = E.Engine.generate_coalesced

        def __init__(self, cfg, precision="bf16", max_batch=64, max_beams=1, max_frames=1, max_text_len=20, **kw):
            self.c = types.SimpleNamespace(max_batch=max_batch, vocab=cfg.vocab)
            self.half = None

        def clone(self):
            other = FakeEngine.__new__(FakeEngine)
            other.c, other.half = types.SimpleNamespace(max_batch=self.c.max_batch, vocab=self.c.vocab), None
            other.shared = getattr(self, "shared", False)
            return other

        def set_shared_device(self, on=True):
            self.shared = on

        def load_state_dict(self, sd): pass
        def close(self): pass
        def set_graph(self, on): pass
        def set_encode_after(self, other): pass
        def profile_enable(self, on): pass

        def profile_read(self):
            return dict(vit_ms=4.0, prefill_ms=1.0, decode_ms=5.0, total_ms=10.0, gemm_ms=4.0, gemm_launches=66, gemm_flops=3e12,
                        vit_gemm_ms=3.0, vit_gemm_launches=49, vit_gemm_flops=2.2e12, decode_step_ms=0.25, decode_steps=19,
                        decode_step_bytes=3.77e8)

        def _out(self, B, search):
            toks = torch.full((B, search.max_steps), 7, dtype=torch.int64)
            return toks, torch.zeros(B), torch.tensor([search.max_steps, 0, search.max_steps - 1, 0], dtype=torch.int32)

        def generate(self, frames, search, prefix=None, sync=True):
            log.append(("generate", int(frames[0].shape[0])))
            return self._out(int(frames[0].shape[0]), search)

        def step_logits(self, tokens):
            return torch.zeros(tokens.shape[0], self.c.vocab)

        def encode(self, frames, return_features=True):
            return None

    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.cuda, "set_device", lambda d: None)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    monkeypatch.setattr(torch.cuda, "Stream", FakeStream)
    monkeypatch.setattr(torch.cuda, "Event", FakeEvent)
    monkeypatch.setattr(torch.cuda, "stream", lambda s: contextlib.nullcontext())
    monkeypatch.setattr(torch.cuda, "synchronize", lambda: None)
    monkeypatch.setattr(E, "Engine", FakeEngine)
    monkeypatch.setattr(synthetic, "random_state_dict", lambda cfg, seed=0, **kw: {})
    monkeypatch.setattr(synthetic, "random_frames", lambda cfg, B, F, seed=0: [torch.zeros(B, 3, 8, 8) for _ in range(F)])
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "BENCH_GEMM_IMPL"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--no-cpu-baseline"] + argv)
    monkeypatch.setattr(bench, "alt_precision_line", lambda child: {"precision": child[1], "child_argv": child})
    children = []

    def fake_child(argv_child, timeout_s=150):
        children.append(list(argv_child))
        return {"value": 1000.0, "ms_per_step": 1.0, "steps": 8, "dtype": "f16", "config": {"workload": " ".join(argv_child[:4])},
                "roofline_decode": {"avg_step_ms": 0.3}, "parity": {"ok": True, "rows": 64, "identical": 60, "wide_margin": None}}
    monkeypatch.setattr(bench, "child_line", fake_child)
    bench._test_children = children
    try:
        bench.main()
    finally:
        E.use_experiment_build(False)           # --experiment flips a module switch
    line = capsys.readouterr().out.strip().splitlines()[-1]
    return json.loads(line), log


def test_bench_control_flow_all_schedules(monkeypatch, capsys):
    # the contract's fields, default (mixed) schedule
    d, log = _run_bench_with_fakes(monkeypatch, capsys, ["--steps", "8", "--warmup", "3"])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "roofline_decode", "parity"):
        assert k in d, k
    assert d["steps"] == 8 and d["warmup"] == 3 and d["n_gpus"] == 1 and d["unit"] == "captions/s" and d["vs_baseline"] is None
    assert d["config"]["schedule"] == "mixed" and d["config"]["contexts_in_flight"] == 4 and "model" not in d["config"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in d["roofline"] and k in d["roofline_decode"], k
    timed = [e for e in log if e[0] == "generate"]
    assert len(timed) >= 4 + 3 + 8                      # priming pass over the contexts + warm-up + timed steps (+ roofline passes)
    # round 5: every decision teacher-forced against the reference; round 6: the headline is the fp16-operand build (the one
    # that meets the specification's logit tolerance), the bf16 build runs as a child on the same schedule
    assert d["dtype"] == "f16" and d["config"]["library"] == "libgitmi_f16.so"
    tf = d["parity"]["teacher_forced"]
    assert tf["decisions"] == 64 * 19 and 0 < tf["decidable"] <= tf["decisions"] and "max_logit_err" in tf
    assert tf["spec_logit_frac"] == pytest.approx(1e-3) and tf["logit_err_bound"] == pytest.approx(1e-3 * tf["logit_span"], rel=1e-3)
    child = d["alt_precision"]["child_argv"]
    assert child[:3] == ["--precision", "bf16", "--brief"] and child[child.index("--steps") + 1] == "8"
    # the second parity leg: trained-checkpoint statistics, same geometry, every row required of the headline build
    tr = d["parity"]["trained_statistics"]
    assert tr["reference"].endswith("full_trained_b64_greedy.npz") and tr["required"] == 64 and tr["teacher_forced"]["decisions"] == 64 * 19
    assert tr["ok"] is False and d["parity"]["ok"] is False           # the stand-in engine returns constant ids: the leg must notice
    # the other BASELINE configurations ride in the same line, one short child each, in this run's precision
    assert set(d["other_configs"]) == {"cfg3_base_b64_beam4", "cfg4_large_b32_greedy", "cfg5_vatex_b16_6frames"}
    import bench
    assert all(c[c.index("--precision") + 1] == "f16" and "--brief" in c for c in bench._test_children)
    assert d["other_configs"]["cfg3_base_b64_beam4"]["parity"]["identical"] == 60
    # --brief (what that child runs): timed loop + parity, no roofline passes
    d, log = _run_bench_with_fakes(monkeypatch, capsys, ["--steps", "8", "--warmup", "3", "--brief", "--precision", "f16"])
    assert "roofline" not in d and "alt_precision" not in d and "other_configs" not in d and "parity" in d and d["dtype"] == "f16"
    assert d["roofline_decode"]["avg_step_ms"] == 0.25
    # coalesced: two requests per engine pass
    d, log = _run_bench_with_fakes(monkeypatch, capsys, ["--steps", "8", "--warmup", "3", "--coalesce", "2"])
    assert "2 requests of 64 images coalesced" in d["config"]["schedule"] and d["warmup"] == 4
    passes = [e for e in log if e == ("generate", 128)]
    assert len(passes) == 4 + 2 + 4                     # priming (one pass per context), warm-up 4 steps, 8 timed steps
    with pytest.raises(SystemExit):
        _run_bench_with_fakes(monkeypatch, capsys, ["--steps", "7", "--coalesce", "2"])
    # the schedules that lost were removed in round 5: their flags no longer exist
    for flags in (["--decode-group", "2"], ["--phased", "4"]):
        with pytest.raises(SystemExit):
            _run_bench_with_fakes(monkeypatch, capsys, ["--steps", "8", "--warmup", "4"] + flags)


def test_bench_reads_the_pmc_summary_taken_for_this_csrc(tmp_path, monkeypatch):
    """bench.pmc_profile: of several committed summaries the one whose `csrc_sha` header equals the tree's hash is read
    (whatever its name sorts like) and reported fresh; with none matching, the last by name is read and reported stale."""
    import bench
    prof = tmp_path / "profiles"
    prof.mkdir()
    hdr = "kernel\tDISPATCHES\tHBM_BYTES\tMFMA_UTIL_PCT\tL2_HIT_PCT\n"

    def write(name, sha, hbm):
        (prof / name).write_text("# comment\n# csrc_sha=%s\n%svoid gitmi::gemm_p8_kernel<x>\t10\t%d\t30.5\t70\n"
                                 "void gitmi::gemm_p8_kernel<y>\t30\t%d\t32.5\t74\nother_kernel\t5\t1\t-\t-\n" % (sha, hdr, hbm, 2 * hbm))
    write("r03_final_pmc_summary.tsv", "aaaa", 100)
    write("r03_zz_pmc_summary.tsv", "bbbb", 1000)              # sorts last by name
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "csrc_sha", lambda: "aaaa")
    p = bench.pmc_profile({"gemm": "gemm_p8", "none": "no_such_kernel"})
    assert p["source"].endswith("r03_final_pmc_summary.tsv") and p["stale"] is False and "none" not in p
    assert p["gemm"]["hbm_bytes"] == round((10 * 100 + 30 * 200) / 40, 2) and p["gemm"]["mfma_util_pct"] == 32.0
    monkeypatch.setattr(bench, "csrc_sha", lambda: "cccc")
    p = bench.pmc_profile({"gemm": "gemm_p8"})
    assert p["source"].endswith("r03_zz_pmc_summary.tsv") and p["stale"] is True and p["gemm"]["hbm_bytes"] == 1750.0
