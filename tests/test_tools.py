"""CPU: the profile-analysis tools on synthetic rocpd databases (same `kernels` view columns as rocprofv3 writes)."""
import os
import sqlite3
import sys

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
