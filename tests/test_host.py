"""CPU: the C-ABI library loads and exports every symbol include/gitmi.h declares; host-side logic."""
import base64
import io
import json
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT
from generativeimage2text_amd import configs, engine, inference, model, tsv_io


def _declared_functions(header="gitmi.h"):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(gitmi_[a-z_0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = engine.load_library()
    declared = _declared_functions()
    assert len(declared) >= 18
    for name in declared:
        assert hasattr(lib, name), name
    assert sorted(engine.EXPORTED_SYMBOLS) == declared
    assert lib.gitmi_abi_version() == 10
    assert len(declared) <= 40                  # the product ABI stays small: schedules that lost and debug hooks live elsewhere


def test_product_libraries_export_the_c_abi_and_nothing_else():
    """nm -D: exactly the entry points of include/gitmi.h (no C++ launcher symbols, no experiment entry points), no getenv
    in the product (it reads no environment); the measurement build adds exactly include/gitmi_experiment.h (+ one timing
    switch of the decode-chain unit entry point)."""
    import shutil
    import subprocess
    if shutil.which("nm") is None:
        pytest.skip("no nm")
    declared = set(_declared_functions())
    extra = set(_declared_functions("gitmi_experiment.h")) - declared
    assert extra == set(engine.EXPERIMENT_SYMBOLS)

    def dyn(path):
        out = subprocess.run(["nm", "-D", path], capture_output=True, text=True, check=True).stdout.splitlines()
        defined = {l.split()[-1] for l in out if " T " in l}
        undefined = {l.split()[-1].split("@")[0] for l in out if " U " in l}
        return defined, undefined
    for path in (engine.LIB_PATH, engine.LIB_PATH_F16):
        if not os.path.exists(path):
            pytest.skip("library not built")
        defined, undefined = dyn(path)
        assert defined == declared, (path, sorted(defined ^ declared))
        assert "getenv" not in undefined and "secure_getenv" not in undefined, path
        lib = engine.load_library("bf16" if path == engine.LIB_PATH else "f16")
        for name in extra:
            assert not hasattr(lib, name), name
    if os.path.exists(engine.LIB_PATH_EXP):
        defined, undefined = dyn(engine.LIB_PATH_EXP)
        assert defined == declared | extra
        assert "getenv" in undefined


def test_struct_layouts_match_header():
    assert engine.C.sizeof(engine.GitmiConfig) == 21 * 4
    assert engine.C.sizeof(engine.GitmiSearch) == 72          # 4 x int32, double, 2 x int32, 2 x double, uint64, double, 2 x int32
    assert engine.GitmiSearch.length_penalty.offset == 16 and engine.GitmiSearch.top_p.offset == 32
    assert engine.GitmiSearch.seed.offset == 48 and engine.GitmiSearch.num_keep_best.offset == 64
    import re, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "gitmi.h")).read()
    body = hdr[hdr.index("typedef struct gitmi_search {"):hdr.index("} gitmi_search;")]
    fields = re.findall(r"^\s*(?:int32_t|double|uint64_t)\s+(\w+);", body, re.M)
    assert fields == [n for n, _ in engine.GitmiSearch._fields_]


@pytest.mark.skipif(torch.cuda.is_available(), reason="CPU-only check")
def test_engine_fails_loudly_without_gpu():
    with pytest.raises(engine.GitmiError):
        engine.Engine(configs.config_for_model("GIT_BASE"))


def test_model_param_table():
    c = configs.config_for_model("GIT_LARGE_COCO")
    assert (c.patch, c.vit_width, c.vit_layers, c.vit_heads, c.n_tok) == (14, 1024, 24, 16, 257)
    c = configs.config_for_model("GIT_BASE_VATEX")
    assert c.num_frames == 6 and c.n_tok == 197
    c = configs.config_for_model("GIT_BASE_VQAv2")
    assert c.image_size == 480 and c.test_respect_ratio_max == 640 and c.max_image_hw == (480, 640) and c.n_tok == 901
    assert configs.config_for_model("GIT_LARGE_TEXTVQA").max_image_hw == (420, 560)
    assert configs.config_for_model("GIT_BASE").max_image_hw is None
    with pytest.raises(KeyError):
        configs.config_for_model("nope")


def test_state_dict_suffix_alignment():
    cfg = configs.config_for_model("GIT_BASE")
    keys = model.expected_state_dict_keys(cfg)
    assert len(keys) == len(set(keys))
    loaded = {"module." + k: torch.zeros(1) for k in keys}
    loaded["module.image_encoder.proj"] = torch.zeros(1)
    out = model.load_state_dict_by_suffix(keys, loaded)
    assert set(out) == set(keys)
    # longest suffix wins
    got = model.load_state_dict_by_suffix(["a.b.weight"], {"weight": torch.ones(1), "b.weight": torch.zeros(1)})
    assert got["a.b.weight"].item() == 0


def test_search_config_holders():
    d = model.GeneratorWithBeamSearch(eos_index=102, max_steps=1024, beam_size=4, length_penalty=0.6)
    assert (d.per_node_beam_size, d.kind) == (2, "generator")
    a = model.AutoRegressiveBeamSearch(eos_index=102, max_steps=20, beam_size=1, per_node_beam_size=1,
                                       fix_missing_prefix=True)
    assert a.kind == "autoregressive"
    with pytest.raises(AssertionError):
        model.AutoRegressiveBeamSearch(eos_index=102)


def test_shard_range_matches_reference_rule():
    # inference.py:165-169: ceil(N/W) rows per rank, contiguous
    for n, w in [(10, 3), (8, 8), (5, 8), (1000, 7)]:
        covered = []
        for r in range(w):
            s, e = inference.shard_range(n, r, w)
            covered += list(range(s, max(s, e)))
        assert covered == list(range(n))


def test_tsv_roundtrip_and_concat(tmp_path):
    a, b, out = str(tmp_path / "a.tsv"), str(tmp_path / "b.tsv"), str(tmp_path / "all.tsv")
    tsv_io.tsv_writer([["k%d" % i, "v%d" % i] for i in range(5)], a)
    tsv_io.tsv_writer([["q%d" % i, "wé%d" % i] for i in range(3)], b)
    assert os.path.getsize(str(tmp_path / "a.lineidx.8b")) == 5 * 8
    t = tsv_io.TSVFile(a)
    assert len(t) == 5 and t[3] == ["k3", "v3"]
    tsv_io.concat_tsv_files([a, b], out)
    t = tsv_io.TSVFile(out)
    assert len(t) == 8 and t[6] == ["q1", "wé1"] and t[0] == ["k0", "v0"]
    assert list(tsv_io.tsv_reader(out))[7] == ["q2", "wé2"]


def test_state_dict_key_alignment_equals_reference():
    """model key -> checkpoint key exactly as the reference's load_state_dict pairs them (torch_common.py:93-145),
    on the synthetic key sets frozen in tests/golden/state_dict_align.json."""
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "state_dict_align.json")))
    for name, case in g["cases"].items():
        loaded = {k: k for k in case["loaded_keys"]}                 # the "tensor" is the loaded key's own name
        got = model.load_state_dict_by_suffix(g["model_keys"], loaded)
        # the fixture names loaded keys as given (with their 'module.' prefixes)
        assert got == case["mapping"], name


def test_model_params_equal_reference_yaml(tmp_path):
    """configs.MODEL_PARAMS against every aux_data/models/*/parameter.yaml of the reference as its own loader reads
    them (tests/golden/model_params.json); the `_base_` include chain; the run-time lookup order of the tasks."""
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "model_params.json")))
    for name, param in g["models"].items():
        assert configs.MODEL_PARAMS[name] == param, name
    for name in set(configs.MODEL_PARAMS) - set(g["models"]):
        assert configs.MODEL_PARAMS[name] == {}, name          # GIT_BASE, GIT_BASE_COCO, GIT_BASE_TEXTCAPS: no yaml
    d = tmp_path / "models" / "MY_MODEL"
    d.mkdir(parents=True)
    for n, txt in g["base_chain"]["files"].items():
        (d / n).write_text(txt)
    assert configs.load_from_yaml_file(str(d / "leaf.yaml")) == g["base_chain"]["leaf"]
    (d / "parameter.yaml").write_text("_base_: leaf.yaml\nnum_image_with_embedding: 6\n")
    p = configs.load_model_param("MY_MODEL", str(tmp_path / "models"))
    assert p["num_image_with_embedding"] == 6 and p["test_crop_size"] == 420
    cfg = configs.config_from_param(p, name="MY_MODEL")
    assert (cfg.patch, cfg.vit_width, cfg.num_frames, cfg.image_size, cfg.max_image_hw) == (14, 1024, 6, 420, (420, 560))
    assert configs.load_model_param("GIT_BASE_VATEX", str(tmp_path / "models")) == {"num_image_with_embedding": 6}
    with pytest.raises(KeyError):
        configs.load_model_param("NOT_A_MODEL", str(tmp_path / "models"))
    p, from_file = inference._task_param("GIT_LARGE_VQAv2", str(tmp_path / "models"))
    assert p["test_respect_ratio_max"] == 560 and not from_file
    p, from_file = inference._task_param("MY_MODEL", str(tmp_path / "models"))
    assert p["visual_feature_size"] == 1024 and from_file
    with pytest.raises(KeyError):
        inference._task_param("NOT_A_MODEL", str(tmp_path / "models"))


def test_generate_coalesced_splits_results_per_request():
    """Engine.generate_coalesced: requests concatenated into one pass, results handed back per request (the pass itself
    is Engine.generate, stood in for here: no GPU)."""
    import types

    class Fake:
        c = types.SimpleNamespace(max_batch=8)
        generate_coalesced = engine.Engine.generate_coalesced

        def generate(self, frames, search, sync=True):
            self.seen = [tuple(f.shape) for f in frames]
            return frames[0][:, 0, 0, :2].long(), frames[-1][:, 0, 0, 0].float(), torch.zeros(4, dtype=torch.int32)

    f = Fake()
    a = [torch.arange(6.).reshape(3, 1, 1, 2), 10 + torch.arange(6.).reshape(3, 1, 1, 2)]        # 3 images, 2 frames
    b = [100 + torch.arange(4.).reshape(2, 1, 1, 2), 200 + torch.arange(4.).reshape(2, 1, 1, 2)]
    outs, info = f.generate_coalesced([a, b], None)
    assert f.seen == [(5, 1, 1, 2), (5, 1, 1, 2)]
    assert outs[0][0].tolist() == [[0, 1], [2, 3], [4, 5]] and outs[1][0].tolist() == [[100, 101], [102, 103]]
    assert outs[0][1].tolist() == [10.0, 12.0, 14.0] and outs[1][1].tolist() == [200.0, 202.0]
    outs, _ = f.generate_coalesced([a], None)
    assert f.seen == [(3, 1, 1, 2), (3, 1, 1, 2)] and len(outs) == 1
    with pytest.raises(engine.GitmiError):
        f.generate_coalesced([a, a, a], None)


def _gold_tsv():
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "tsv_wire.npz"))


def _rows_fixture():
    # the rows oracle/make_host_golden.py fed to the REFERENCE's tsv_writer
    jpeg_like = base64.b64encode(bytes(range(256)) * 3).decode()
    return [["img_0", jpeg_like], ["img 1 with spaces", "café 中文"], ["k2", ""],
            ["k3", "  padded field  ", "third"], [4, 5.5, "mixed"], ["only_one_column"]]


def test_tsv_writer_bytes_equal_reference(tmp_path):
    """.tsv, .lineidx and .lineidx.8b byte for byte what the reference's tsv_writer wrote for the same rows
    (tests/golden/tsv_wire.npz, frozen by oracle/make_host_golden.py from /root/reference)."""
    g = _gold_tsv()
    a = str(tmp_path / "a.tsv")
    tsv_io.tsv_writer(_rows_fixture(), a)
    for ext in (".tsv", ".lineidx", ".lineidx.8b"):
        got = open(os.path.splitext(a)[0] + ext, "rb").read()
        assert got == g["a" + ext].tobytes(), ext


def test_tsv_reads_equal_reference(tmp_path):
    """Files written by the reference, read back here: TSVFile[i], iteration, tsv_reader, get_key."""
    g = _gold_tsv()
    a = str(tmp_path / "a.tsv")
    for ext in (".tsv", ".lineidx", ".lineidx.8b"):
        open(os.path.splitext(a)[0] + ext, "wb").write(g["a" + ext].tobytes())
    t = tsv_io.TSVFile(a)
    assert len(t) == int(g["a_len"]) == t.num_rows()
    assert [t[i] for i in range(len(t))] == json.loads(str(g["a_rows_getitem"]))
    assert [r for r in tsv_io.TSVFile(a)] == json.loads(str(g["a_rows_iter"]))
    assert [r for r in tsv_io.tsv_reader(a)] == json.loads(str(g["a_rows_reader"]))
    assert [t.get_key(i) for i in (0, 1, 3)] == json.loads(str(g["a_keys"]))
    # the reference requires the .8b index; here the text index or the data alone are enough as well
    os.remove(os.path.splitext(a)[0] + ".lineidx.8b")
    assert [r for r in tsv_io.TSVFile(a)] == json.loads(str(g["a_rows_iter"]))
    os.remove(os.path.splitext(a)[0] + ".lineidx")
    assert [r for r in tsv_io.TSVFile(a)] == json.loads(str(g["a_rows_iter"]))


def test_tsv_concat_and_task_rows_equal_reference(tmp_path):
    """Caption rows / one-column VQA rows as inference.py:199, 212 writes them, the two-shard concat of the
    multi-rank hand-off, and the two converters of the outputs."""
    g = _gold_tsv()
    caps = ["a dog on a couch", 'café "quoted"', ""]
    cap_rows = [["img_%d" % i, inference.json_dump([{"caption": c}])] for i, c in enumerate(caps)]
    b, c, allp = str(tmp_path / "caps.0.2.tsv"), str(tmp_path / "caps.1.2.tsv"), str(tmp_path / "caps.tsv")
    tsv_io.tsv_writer(cap_rows[:2], b)
    tsv_io.tsv_writer(cap_rows[2:], c)
    tsv_io.concat_tsv_files([b, c], allp)
    for name, p in (("caps0", b), ("caps1", c), ("caps_all", allp)):
        assert open(p, "rb").read() == g[name + ".tsv"].tobytes(), name
        assert open(os.path.splitext(p)[0] + ".lineidx.8b", "rb").read() == g[name + ".lineidx.8b"].tobytes(), name
    assert [r for r in tsv_io.TSVFile(allp)] == json.loads(str(g["caps_all_rows"]))
    tsv_io.concat_tsv_files([allp], allp)                 # the reference's no-op case (tsv_io.py:23-24)
    assert open(allp, "rb").read() == g["caps_all.tsv"].tobytes()
    v = str(tmp_path / "vqa.tsv")
    tsv_io.tsv_writer([[inference.json_dump({"answer": a_, "question_id": q})] for a_, q in
                       [("yes", 17), ("two", 4), ("café", 900001)]], v)
    assert open(v, "rb").read() == g["vqa.tsv"].tobytes()
    out_json = str(tmp_path / "sub" / "vqa.json")
    inference.convert_tsv_to_vqa_json(v, out_json)
    assert open(out_json).read() == str(g["vqa_json"])
    assert [inference.json_dump({"b": 1, "a": [1, 2, {"z": None, "y": True}], "c": "café 中"}),
            inference.json_dump([{"caption": "x"}]),
            inference.json_dump({"answer": "no", "question_id": 3})] == json.loads(str(g["json_dump_cases"]))
    coco = str(tmp_path / "coco.json")
    inference.convert_tsv_to_coco_format(allp, coco)
    # (the empty caption is kept as a real, empty string: the row's JSON column is `[{"caption":""}]`)
    assert json.load(open(coco)) == [{"image_id": "img_%d" % i, "caption": c_} for i, c_ in enumerate(caps)]


def test_image_transform_shape_and_normalisation():
    from PIL import Image
    rng = np.random.RandomState(0)
    img = Image.fromarray(rng.randint(0, 255, (300, 400, 3), dtype=np.uint8))
    x = inference.image_transform(img, 224)
    assert x.shape == (3, 224, 224) and x.dtype == torch.float32
    assert -2.5 < x.mean().item() < 2.5
    buf = io.BytesIO()
    img.save(buf, format="JPEG")
    im2 = inference.load_image_by_pil(base64.b64decode(base64.b64encode(buf.getvalue())))
    assert im2.size == (400, 300)


def test_prefix_ids_and_id_tokenizer():
    tok = inference.IdTokenizer()
    assert inference._prefix_ids(tok, "2054 2003") == [101, 2054, 2003]
    long = " ".join(str(i) for i in range(1000, 1100))
    ids = inference._prefix_ids(tok, long)
    assert len(ids) == 39 and ids[0] == 101 and ids[-1] == 1099          # keeps the LAST 38 (inference.py:99-100)
    assert tok.decode([101, 7, 8, 102, 102]) == "7 8"


def test_wordpiece_tokenizer_text_io(monkeypatch):
    """inference.py:72, 93-101, 108 with a real BertTokenizer on a (synthetic) WordPiece vocabulary that carries
    bert-base-uncased's special-token ids: question -> [CLS] + ids (keep the LAST 38 of the truncated 40), ids -> text."""
    import os
    from conftest import ROOT
    monkeypatch.setenv("GIT_VOCAB", os.path.join(ROOT, "tests", "data", "vocab.txt"))
    tok = inference.get_tokenizer()
    assert type(tok).__name__ == "BertTokenizer" and (tok.cls_token_id, tok.sep_token_id) == (101, 102)
    ids = inference._prefix_ids(tok, "What color is the cat?")
    assert ids[0] == 101 and tok.decode(ids, skip_special_tokens=True) == "what color is the cat?"
    assert tok.decode([101] + ids[1:] + [102, 102], skip_special_tokens=True) == "what color is the cat?"
    words = ("what is this " * 30).split()
    long_ids = inference._prefix_ids(tok, " ".join(words))
    full = tok(" ".join(words), add_special_tokens=False)["input_ids"]
    assert len(long_ids) == 39 and long_ids[0] == 101 and long_ids[1:] == full[:40][-38:]
    # unknown words fall back to word pieces / [UNK], never crash
    assert len(inference._prefix_ids(tok, "zebras skateboarding")) > 1


def test_missing_vocabulary_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setenv("GIT_VOCAB", str(tmp_path / "nope.txt"))
    monkeypatch.setenv("HF_HOME", str(tmp_path / "hf"))
    monkeypatch.setenv("HF_HUB_OFFLINE", "1")
    with pytest.raises(FileNotFoundError):
        inference.get_tokenizer()
    monkeypatch.setenv("GIT_VOCAB", "ids")
    assert isinstance(inference.get_tokenizer(), inference.IdTokenizer)
    with pytest.raises(ValueError):
        inference.IdTokenizer()("what is this")


def test_minmax_resize_sizes_match_reference():
    """MinMaxResizeForTest.get_size against outputs of the reference class (tests/golden/minmax_sizes.npz, written by
    oracle/make_golden.py from inference.py:29-64) -- incl. the early-return and the truncation/rounding cases."""
    from conftest import load_golden
    g = load_golden("minmax_sizes")
    for (mn, mx), wh, ref in ((tuple(g["cfg_a"]), g["wh"], g["out_a"]), (tuple(g["cfg_b"]), g["wh"], g["out_b"])):
        t = inference.MinMaxResizeForTest(int(mn), int(mx))
        for (w, h), (oh, ow) in zip(wh.tolist(), ref.tolist()):
            assert t.get_size((w, h)) == (oh, ow), (mn, mx, w, h)


def test_minmax_transform_cpu_shape():
    from PIL import Image
    rng = np.random.RandomState(1)
    img = Image.fromarray(rng.randint(0, 255, (300, 500, 3), dtype=np.uint8))
    x = inference.get_image_transform({"test_crop_size": 480, "test_respect_ratio_max": 640})(img)
    assert x.shape == (3, 384, 640)          # long side capped at 640, short = round(640 * 300 / 500)


def test_every_entry_point_has_ctypes_prototypes():
    # a missing argtypes list makes ctypes pass 64-bit pointers as C ints (truncated): every bound symbol must declare them
    lib = engine.load_library()
    for name in engine.EXPORTED_SYMBOLS:
        if name in ("gitmi_abi_version", "gitmi_last_error", "gitmi_operand_dtype"):      # no arguments
            continue
        assert getattr(lib, name).argtypes is not None, name


def test_both_operand_builds_load_and_identify_themselves():
    """libgitmi.so (bf16 operands, the benchmarked build) and libgitmi_f16.so (the same sources with -DGITMI_OPS_F16):
    same ABI, every declared symbol, and each says which 16-bit operand type it was built for."""
    a, b = engine.load_library("bf16"), engine.load_library("f16")
    assert a.gitmi_operand_dtype() == engine.DTYPE_BF16 and b.gitmi_operand_dtype() == engine.DTYPE_F16
    assert a.gitmi_abi_version() == b.gitmi_abi_version() == 10
    for name in engine.EXPORTED_SYMBOLS:
        getattr(b, name)


def test_config_struct_fields_agree_across_header_binding_and_integration_doc():
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "gitmi.h")).read()
    body = hdr[hdr.index("typedef struct gitmi_config {"):hdr.index("} gitmi_config;")]
    header_fields = re.findall(r"int32_t\s+(\w+);", body)
    assert header_fields == [n for n, _ in engine.GitmiConfig._fields_]
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    stub = doc[doc.index("class Cfg(C.Structure):"):doc.index("class Search(C.Structure):")]
    doc_fields = " ".join(re.findall(r'"([^"]+)"', stub)).split()
    assert doc_fields == header_fields


def test_prefetch_ordered_keeps_order_and_bounds_the_window():
    import threading, time
    started, lock = [], threading.Lock()

    def load(i):
        with lock:
            started.append(i)
        time.sleep(0.002 * ((7 * i) % 5))            # uneven durations: completion order != submission order
        return i * i

    got = []
    for v in inference.prefetch_ordered(40, load, threads=4, window=6):
        with lock:
            ahead = len(started) - len(got)
        assert ahead <= 6 + 1                       # never more than the window (+ the one being consumed) in flight
        got.append(v)
    assert got == [i * i for i in range(40)]
    assert list(inference.prefetch_ordered(5, lambda i: -i, threads=0, window=3)) == [0, -1, -2, -3, -4]
    assert list(inference.prefetch_ordered(0, load, threads=4, window=3)) == []

    def boom(i):
        if i == 3:
            raise ValueError("bad row 3")
        return i
    with pytest.raises(ValueError):
        list(inference.prefetch_ordered(8, boom, threads=3, window=4))


def test_run_tsv_inference_with_decode_threads_matches_serial(tmp_path):
    """The TSV task's loop with the decode stage on host threads: same rows, same order as the serial loop, captions and VQA."""
    rows = [["key%d" % i, base64.b64encode(b"image-bytes-%03d" % i).decode()] for i in range(23)]
    tsv_io.tsv_writer(rows, str(tmp_path / "img.tsv"))
    q = [["key%d" % i, json.dumps([{"question": "q%d-%d" % (i, j), "question_id": 100 * i + j} for j in range(1 + i % 3)])]
         for i in range(23)]
    tsv_io.tsv_writer(q, str(tmp_path / "q.tsv"))
    outs = {}
    for threads in (0, 5):
        for with_q in (False, True):
            out = str(tmp_path / ("out_%d_%d.tsv" % (threads, with_q)))
            seen = []
            inference.run_tsv_inference(
                str(tmp_path / "img.tsv"), str(tmp_path / "q.tsv") if with_q else None, out,
                decode=lambda b: ("decoded", b.decode()), decode_threads=threads,
                transform=lambda d: (seen.append(d[1]), d[1])[1],
                caption_batch=lambda imgs: ["cap<%s>" % im for im in imgs],
                answer_questions=lambda img, qs: ["ans<%s|%s>" % (img, qq) for qq in qs],
                batch_size=4, rank=0, world=1)
            assert seen == ["image-bytes-%03d" % i for i in range(23)]          # transform runs in row order, on this thread
            outs[(threads, with_q)] = open(out, "rb").read()
    assert outs[(0, False)] == outs[(5, False)] and outs[(0, True)] == outs[(5, True)]
    assert outs[(0, False)].count(b"\n") == 23 and outs[(0, True)].count(b"\n") == sum(1 + i % 3 for i in range(23))


def test_tsv_task_function_plumbing_without_a_gpu(tmp_path, monkeypatch):
    """test_git_inference_single_tsv end to end on the CPU with the engine replaced by a stand-in model: real PNG rows,
    base64, PIL decoding on the host thread pool, the (CPU) image transform, batching, caption / VQA row formats, the
    parameter lookup.  The stand-in "captions" an image by the rounded mean of its pixels, which also checks that every
    key is paired with its own image."""
    from PIL import Image
    rng = np.random.RandomState(11)
    img_rows, q_rows, means = [], [], []
    for i in range(9):
        arr = rng.randint(0, 255, (60 + 5 * i, 80, 3), dtype=np.uint8)
        buf = io.BytesIO()
        Image.fromarray(arr).save(buf, format="PNG")
        img_rows.append(["img%d" % i, base64.b64encode(buf.getvalue()).decode()])
        q_rows.append(["img%d" % i, json.dumps([{"question": "%d %d" % (2000 + i, 3000 + j), "question_id": 10 * i + j}
                                                for j in range(1 + i % 2)])])
        means.append(int(round(float(inference.image_transform(Image.fromarray(arr), 224).mean()) * 1000)))
    tsv_io.tsv_writer(img_rows, str(tmp_path / "img.tsv"))
    tsv_io.tsv_writer(q_rows, str(tmp_path / "q.tsv"))

    class FakeModel:
        def __call__(self, batch):
            x = batch["image"]
            return {"predictions": torch.tensor([[101, int(round(float(im.mean()) * 1000)) % 30000 + 1000, 102] for im in x])}

        def answer(self, image, prefixes):
            m = int(round(float(image.mean()) * 1000)) % 30000 + 1000
            return [[m] + list(p[1:]) for p in prefixes]

    monkeypatch.setattr(inference, "get_tokenizer", lambda: inference.IdTokenizer())
    monkeypatch.setattr(inference, "build_model", lambda *a, **k: FakeModel())
    monkeypatch.setattr(inference, "get_image_transform", lambda param, gpu=False: (lambda im: inference.image_transform(im, 224)))
    monkeypatch.setattr(torch.cuda, "set_device", lambda d: None)
    monkeypatch.setattr(torch.Tensor, "cuda", lambda self, *a, **k: self)
    for k in ("RANK", "WORLD_SIZE", "OMPI_COMM_WORLD_SIZE", "OMPI_COMM_WORLD_RANK"):
        monkeypatch.delenv(k, raising=False)
    want = [m % 30000 + 1000 for m in means]
    for threads in ("0", "4"):
        monkeypatch.setenv("GIT_DECODE_THREADS", threads)
        out = str(tmp_path / ("cap%s.tsv" % threads))
        inference.test_git_inference_single_tsv(str(tmp_path / "img.tsv"), "GIT_BASE", None, out, batch_size=4)
        rows = list(tsv_io.tsv_reader(out))
        assert [r[0] for r in rows] == ["img%d" % i for i in range(9)]
        assert [json.loads(r[1])[0]["caption"] for r in rows] == [str(w) for w in want]
        out = str(tmp_path / ("vqa%s.tsv" % threads))
        inference.test_git_inference_single_tsv(str(tmp_path / "img.tsv"), "GIT_BASE", str(tmp_path / "q.tsv"), out)
        got = [json.loads(s) for s, in tsv_io.tsv_reader(out)]
        exp = [{"answer": "%d %d %d" % (want[i], 2000 + i, 3000 + j), "question_id": 10 * i + j}
               for i in range(9) for j in range(1 + i % 2)]
        assert got == exp


def test_single_image_task_plumbing_without_a_gpu(tmp_path, monkeypatch, caplog):
    """test_git_inference_single_image on the CPU with a stand-in model: image file(s) -> transform -> the batch the model
    sees (list of [1,3,H,W] frames, prefix [1,P] starting with [CLS], the keep-the-last-38 rule), the parameter.yaml found
    where the reference looks for it, 'output: ...' logged."""
    import logging
    from PIL import Image
    rng = np.random.RandomState(12)
    paths = []
    for i in range(2):
        p = tmp_path / ("f%d.png" % i)
        Image.fromarray(rng.randint(0, 255, (50, 70, 3), dtype=np.uint8)).save(str(p))
        paths.append(str(p))
    seen = {}

    class FakeModel:
        def cuda(self): return self
        def eval(self): return self

        def __call__(self, batch):
            seen["image"], seen["prefix"] = batch["image"], batch["prefix"]
            return {"predictions": torch.tensor([[2001, 2002, 102]])}

    def fake_build(name, tok, ckpt, **kw):
        seen["build"] = (name, kw)
        return FakeModel()

    monkeypatch.setattr(inference, "get_tokenizer", lambda: inference.IdTokenizer())
    monkeypatch.setattr(inference, "build_model", fake_build)
    monkeypatch.setattr(inference, "get_image_transform",
                        lambda param, gpu=False: (lambda im: inference.image_transform(im, param.get("test_crop_size", 224))))
    monkeypatch.setattr(torch.Tensor, "cuda", lambda self, *a, **k: self)
    monkeypatch.chdir(tmp_path)
    with caplog.at_level(logging.INFO):
        inference.test_git_inference_single_image(paths, "GIT_BASE_VATEX", " ".join(str(3000 + i) for i in range(50)))
    assert inference.test_git_inference_single_image.last_output == "2001 2002"
    assert any(r.message == "output: 2001 2002" for r in caplog.records)
    assert isinstance(seen["image"], list) and [tuple(t.shape) for t in seen["image"]] == [(1, 3, 224, 224)] * 2
    assert seen["prefix"].shape == (1, 39) and seen["prefix"][0, 0].item() == 101
    assert seen["prefix"][0, 1:].tolist() == [3000 + i for i in range(12, 50)]          # the LAST 38 tokens are kept
    assert seen["build"][0] == "GIT_BASE_VATEX" and "param" not in seen["build"][1]       # table entry, no yaml on disk
    # a parameter.yaml where the reference reads it (aux_data/models/<name>/) takes over
    d = tmp_path / "aux_data" / "models" / "MY_FINETUNE"
    d.mkdir(parents=True)
    (d / "parameter.yaml").write_text("test_crop_size: 160\nnum_image_with_embedding: 6\n")
    inference.test_git_inference_single_image(paths[0], "MY_FINETUNE", "")
    assert seen["build"][1]["param"] == {"test_crop_size": 160, "num_image_with_embedding": 6}
    assert [tuple(t.shape) for t in seen["image"]] == [(1, 3, 160, 160)] and seen["prefix"].tolist() == [[101]]
    with pytest.raises(KeyError):
        inference.test_git_inference_single_image(paths[0], "NO_SUCH_MODEL", "")


class _FakeSearchContext:
    """Stand-in for the engine's search seam (begin / rows / advance / finish): greedy over whatever logits it is given,
    first beam only -- enough to check the host loop of decoder.search(): which rows `step` sees, when the loop stops,
    how the results are shaped.  (The search itself is tested on the GPU against the scripted reference goldens.)"""

    def __init__(self, eos, B, beams, T):
        self.eos, self.B, self.k, self.T = eos, B, beams, T
        self.calls = []

    def search_begin(self, search, start, vocab):
        self.vocab = vocab
        self.seq = start.clone()                                   # [B, t]
        self.lp = torch.zeros(self.B)
        self.P = start.shape[1]
        self.early = 0

    def search_rows(self):
        return self.seq.repeat_interleave(self.k, dim=0)

    def search_advance(self, logits):
        assert logits.shape == (self.B * self.k, self.vocab), logits.shape
        lg = logits.float()[::self.k]
        ended = self.seq[:, -1] == self.eos if self.seq.shape[1] > self.P else torch.zeros(self.B, dtype=torch.bool)
        nxt = lg.log_softmax(-1).argmax(-1)
        nxt[ended] = self.eos
        self.lp += torch.where(ended, torch.zeros(self.B), lg.log_softmax(-1).max(-1).values)
        self.seq = torch.cat([self.seq, nxt[:, None]], 1)
        if self.seq.shape[1] == self.P + 1 and self.k == 1 and bool((nxt == self.eos).all()):
            self.early = 1

    def search_done_count(self):
        # stand-in rule: a sentence is done once its row ends with EOS
        if self.seq.shape[1] <= self.P:
            return 0
        return int((self.seq[:, -1] == self.eos).sum())

    def search_finish(self):
        t = self.seq.shape[1]
        tokens = torch.full((self.B, self.T), self.eos, dtype=torch.int64)
        tokens[:, :t] = self.seq
        return tokens, self.lp.clone(), torch.tensor([t, self.early, t - self.P, 0], dtype=torch.int32)

    def close(self):
        pass


def test_search_methods_host_loop(monkeypatch):
    """AutoRegressiveBeamSearch.search / GeneratorWithBeamSearch.search (decoder.py:224-231, 1083-1092): the rows the
    caller's `step` receives, the early exits, the shapes handed back -- on a stand-in search context."""
    made = []

    def factory(eos, B, beams, T):
        made.append(_FakeSearchContext(eos, B, beams, T))
        return made[-1]

    V, eos = 30, 2
    seen = []

    def step_never_eos(rows):
        seen.append(tuple(rows.shape))
        lg = torch.zeros(rows.shape[0], V)
        lg[torch.arange(rows.shape[0]), (rows[:, -1] + 3) % (V - 3) + 3] = 5.0       # successor token, never EOS (= 2)
        return lg

    # AutoRegressiveBeamSearch, beam 3: first call sees ONE row per sentence, later calls B*k rows; runs to max_steps
    model._SEARCH_ENGINES.clear()
    dec = model.AutoRegressiveBeamSearch(eos_index=eos, max_steps=7, beam_size=3, per_node_beam_size=2, fix_missing_prefix=True)
    start = torch.tensor([[5, 9], [5, 11]])
    preds, lps = dec.search(start, step_never_eos, _engine_factory=factory)
    assert seen == [(2, 2)] + [(6, t) for t in range(3, 7)]
    assert preds.shape == (2, 7) and lps.shape == (2,) and preds[:, :2].tolist() == start.tolist()
    assert (preds != eos).all()

    # beam 1, every sentence ends at its first step: the early return ([B,1], [B,1]) and only ONE step call
    model._SEARCH_ENGINES.clear()
    seen.clear()

    def step_eos(rows):
        seen.append(tuple(rows.shape))
        lg = torch.zeros(rows.shape[0], V)
        lg[:, eos] = 9.0
        return lg
    dec1 = model.AutoRegressiveBeamSearch(eos_index=eos, max_steps=7, beam_size=1, per_node_beam_size=1, fix_missing_prefix=True)
    preds, lps = dec1.search(torch.tensor([[5], [6], [7]]), step_eos, _engine_factory=factory)
    assert seen == [(3, 1)] and preds.tolist() == [[eos]] * 3 and lps.shape == (3, 1)

    # EOS in the middle: the loop stops calling `step` once every row's last token is EOS (decoder.py:319-320)
    model._SEARCH_ENGINES.clear()
    seen.clear()

    def step_eos_at_4(rows):
        seen.append(tuple(rows.shape))
        return step_eos(rows) if rows.shape[1] >= 3 else step_never_eos(rows)
    preds, lps = dec1.search(torch.tensor([[5], [6]]), step_eos_at_4, _engine_factory=factory)
    seen_shapes = [s for s in seen if len(s) == 2]
    assert preds.shape == (2, 4) and preds[:, -1].tolist() == [eos, eos]
    assert max(t for _, t in seen_shapes) == 3                      # no call on the 4-token rows

    # GeneratorWithBeamSearch: B*k rows from the first call on, one call per position up to max_steps, [B, T] + [B, 1]
    model._SEARCH_ENGINES.clear()
    seen.clear()
    gen = model.GeneratorWithBeamSearch(eos_index=eos, max_steps=6, beam_size=4, per_node_beam_size=2, length_penalty=0.6)
    decoded, lps = gen.search(torch.tensor([[5, 9, 4]]), step_never_eos, _engine_factory=factory)
    assert seen == [(4, t) for t in range(3, 6)]
    assert decoded.shape == (1, 6) and lps.shape == (1, 1)
    # ... and no further `step` calls once the search reports every sentence done (decoder.py:1251)
    model._SEARCH_ENGINES.clear()
    seen.clear()
    gen9 = model.GeneratorWithBeamSearch(eos_index=eos, max_steps=9, beam_size=4, per_node_beam_size=2, length_penalty=0.6)
    decoded, lps = gen9.search(torch.tensor([[5], [6]]), step_eos_at_4, _engine_factory=factory)
    assert max(t for _, t in seen) == 3 and decoded.shape == (2, 9)
    # num_return_sequences (decoder.py:1093-1097): the seam sees every sentence r times, r * beam_size rows per start row
    model._SEARCH_ENGINES.clear()
    seen.clear()
    decoded, lps = gen.search(torch.tensor([[5], [7]]), step_never_eos, num_return_sequences=3, _engine_factory=factory)
    assert made[-1].B == 6 and seen[0] == (24, 1) and decoded.shape == (6, 6) and lps.shape == (6, 1)
    assert decoded[:, 0].tolist() == [5, 5, 5, 7, 7, 7]
    with pytest.raises(NotImplementedError):
        dec.search(start, step_never_eos, do_sample=True, _engine_factory=factory)
    model._SEARCH_ENGINES.clear()


def test_token_trie_mirror_matches_the_oracle_restatement():
    """model.TokenTrie (the reference's interface, trie_decoder.py:224-257) and its CSR export for gitmi_set_trie."""
    from oracle import git_oracle as O
    seqs = [[5, 6, 2], [5, 7, 9, 2], [8, 2], [5, 6, 4, 2], [5, 6, 2]]
    a, b = model.TokenTrie.construct(seqs), O.TokenTrie.construct(seqs)
    assert a.get_curr_valid() == b.get_curr_valid() == [5, 8]
    assert a.get_valid([5]) == b.get_valid([5]) == [6, 7] and a.get_valid([5, 6]) == [2, 4] and a.get_valid([9]) == []
    a.move(5); a.move(6)
    assert a.get_curr_valid() == [2, 4]
    a.reset()
    assert a.get_curr_valid() == [5, 8]
    with pytest.raises(AssertionError):
        a.move(77)
    off, tok, node = a.csr()
    bo, bt, bn = b.csr()
    assert off == bo.tolist() and tok == bt.tolist() and node == bn.tolist()
    # CSR walk == dictionary walk
    cur = 0
    for t in [5, 7, 9, 2]:
        edges = range(off[cur], off[cur + 1])
        cur = next(node[e] for e in edges if tok[e] == t)
    assert off[cur + 1] == off[cur]                                       # [SEP] is a leaf

    class Tok:
        sep_token_id = 2

        def __call__(self, text, padding=None, add_special_tokens=False):
            return {"input_ids": [10 + len(w) for w in text.split()]}
    trie = model.get_trie(Tok(), texts=["a bb", "a ccc", "dddd"])
    assert trie.get_curr_valid() == [11, 14] and trie.get_valid([11]) == [12, 13] and trie.get_valid([14]) == [2]


def test_header_is_plain_c_and_links_against_the_library(tmp_path):
    """include/gitmi.h is a C header (C99, no C++ or HIP types in any signature): a C translation unit that includes it
    and takes the address of every declared entry point compiles with gcc and links against libgitmi.so."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    names = _declared_functions()
    src = tmp_path / "abi.c"
    src.write_text('#include "gitmi.h"\n#include <stdio.h>\ntypedef void (*fn_t)(void);\nint main(void) {\n  fn_t f[] = {\n'
                   + "".join("    (fn_t)%s,\n" % n for n in names)
                   + '  };\n  printf("%d %d\\n", (int)(sizeof f / sizeof f[0]), gitmi_abi_version());\n  return 0;\n}\n')
    inc = os.path.join(ROOT, "include")
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-fsyntax-only", "-I", inc, str(src)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    lib = engine.LIB_PATH
    if not os.path.exists(lib):
        pytest.skip("libgitmi.so not built")
    exe = tmp_path / "abi"
    r = subprocess.run(["gcc", "-std=c99", "-I", inc, str(src), "-o", str(exe), lib, "-Wl,-rpath," + os.path.dirname(lib),
                        "-Wl,--allow-shlib-undefined"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr          # every symbol the header declares resolves in the library


def test_margin_threshold_follows_the_logit_bound():
    """tools/parity.py: ONE tolerance, the specification's 1e-3 of the reference's logit span, for the headline (fp16) build on
    every case; 2^3 x for bf16 (three fewer mantissa bits); f32 mode 1e-4 absolute.  The margin below which a one-beam row may
    leave the reference's ids is 2 x that bound (capped by a regression constant); beam search keeps a fixed constant.
    ids_parity with it: a divergence at a wide margin fails, at a narrow one passes, the floor on identical rows is enforced."""
    from tools import parity as P
    assert P.SPEC_LOGIT_FRAC == 1e-3 and P.FORMAT_FACTOR == {"f16": 1.0, "bf16": 8.0}
    assert P.logit_bound("f16", 14.5) == pytest.approx(0.0145) and P.logit_bound("bf16", 14.5) == pytest.approx(0.116)
    assert P.logit_bound("f32", 14.5) == 1e-4
    assert P.tf_bounds("f16", 18.5) == {"lerr": pytest.approx(0.0185), "thr": pytest.approx(0.037)}
    assert P.margin_threshold("f16", 0.0145, False) == pytest.approx(0.029)                       # follows the bound
    assert P.margin_threshold("bf16", 0.35, False) == P.GREEDY_MARGIN_CAP["bf16"]                  # wide-span weights: the cap
    assert P.margin_threshold("bf16", 0.0179, True) == P.BEAM_MARGIN_THR["bf16"]                   # beam: fixed
    assert P.identity_required("f32", 14.0, 1e-3) and P.identity_required("f16", 14.0, 0.03)
    assert not P.identity_required("bf16", 14.0, 0.03) and P.identity_required("bf16", 14.5, 0.25)
    import generativeimage2text_amd
    assert not os.path.exists(os.path.join(os.path.dirname(generativeimage2text_amd.__file__), "parity.py")), \
        "acceptance policy is test infrastructure: it must not ship inside the product package"
    ref = np.array([[101, 5, 6, 7], [101, 8, 9, 10]])
    margin = np.array([[0.5, 0.01, 0.5], [0.5, 0.5, 0.5]], dtype=np.float32)
    got = ref.copy()
    got[0, 2:] = [60, 70]                                    # row 0 diverges at decision 1 (margin 0.01): allowed
    st = P.ids_parity(got, ref, margin, 0.0358, chained=False)
    assert st["identical"] == 1 and st["safe_rows"] == 1 and st["first_divergence_margin_max"] == pytest.approx(0.01)
    with pytest.raises(AssertionError, match="floor"):
        P.ids_parity(got, ref, margin, 0.0358, chained=False, min_identical=2)
    bad = ref.copy()
    bad[1, 1] = 99                                           # row 1 has no narrow decision: any divergence is a failure
    with pytest.raises(AssertionError):
        P.ids_parity(bad, ref, margin, 0.0358, chained=False)
    assert set(P.IDENTICAL_REQUIRED) == {"full_wide_b64_greedy", "full_wide_large_b32_greedy", "full_wide_vatex_b16_greedy",
                                         "full_trained_b8_greedy", "full_trained_b64_greedy"}


def test_teacher_forced_parity_counts_and_violations():
    """parity.teacher_forced_parity on a scripted `step`: decisions are read with the no-repeat rule, a flip at a narrow
    margin is counted but allowed, a flip at a wide margin and a logit error above the bound are violations, and the frozen
    fixtures it runs on in the GPU suite are self-consistent (argmax of the frozen top-8 after the no-repeat rule == the
    reference's ids; margins == the step margins of the free-running golden)."""
    import torch
    from tools import parity as P
    V, B, L = 50, 3, 5
    rng = np.random.RandomState(0)
    table = rng.randn(L - 1, B, V).astype(np.float32)                       # logits of decision s for row r
    ref = np.zeros((B, L), dtype=np.int64)
    ref[:, 0] = 1
    margin = np.zeros((B, L - 1), dtype=np.float32)
    for s in range(L - 1):
        d = table[s].copy()
        if s >= 1:
            d[np.arange(B), ref[:, s]] = -10000.0
        order = np.argsort(-d, axis=1)
        ref[:, s + 1] = order[:, 0]
        margin[:, s] = d[np.arange(B), order[:, 0]] - d[np.arange(B), order[:, 1]]
    top = np.argsort(-table, axis=2)[:, :, :8]                              # [L-1, B, 8]
    cols = np.stack([np.sort(rng.permutation(V)[:16]) for _ in range(L - 1)])
    gold = {"live": np.ones((B, L - 1), bool), "margin": margin,
            "top_ids": np.transpose(top, (1, 0, 2)).astype(np.int32),
            "top_vals": np.transpose(np.take_along_axis(table, top, axis=2), (1, 0, 2)),
            "cols": cols.astype(np.int32), "col_vals": np.stack([table[s][:, cols[s]] for s in range(L - 1)], axis=1),
            "logit_min": np.float32(table.min()), "logit_max": np.float32(table.max())}

    def step_of(tab):
        return lambda tokens: torch.from_numpy(tab[tokens.shape[1] - 1])

    st = P.teacher_forced_parity(step_of(table), ref, gold, eos=2, thr=1e-3, lerr_bound=1e-4, f32_step_logits=step_of(table))
    assert st["ok"] and st["agree"] == st["decisions"] == B * (L - 1) and st["max_logit_err"] == 0.0, st
    # a flip at a narrow margin: lift the runner-up of (row 0, decision 2) just above the winner
    narrow = table.copy()
    d = narrow[2][0].copy(); d[ref[0, 2]] = -10000.0
    second = int(np.argsort(-d)[1])
    narrow[2][0][second] += margin[0, 2] + 1e-3
    thr = float(margin[0, 2]) + 0.5
    st = P.teacher_forced_parity(step_of(narrow), ref, gold, eos=2, thr=thr, lerr_bound=10.0)
    assert st["ok"] and st["agree"] == st["decisions"] - 1 and st["max_flipped_margin"] == pytest.approx(margin[0, 2], abs=1e-4), st
    assert st["rows_all_agree"] == B - 1
    st = P.teacher_forced_parity(step_of(narrow), ref, gold, eos=2, thr=float(margin[0, 2]) * 0.5, lerr_bound=10.0)
    assert not st["ok"] and "row 0 decision 2" in st["violation"], st
    st = P.teacher_forced_parity(step_of(table + 0.01), ref, gold, eos=2, thr=1e-3, lerr_bound=5e-3)
    assert not st["ok"] and "above the bound" in st["violation"], st
    # the frozen fixtures: internally consistent with the free-running goldens
    gdir = os.path.join(ROOT, "tests", "golden")
    names = sorted(f[:-7] for f in os.listdir(gdir) if f.endswith("_tf.npz"))
    assert "full_bench_b64_greedy" in names
    for name in names:
        g, t = np.load(os.path.join(gdir, name + ".npz")), np.load(os.path.join(gdir, name + "_tf.npz"))
        ids, live = g["predictions"], t["live"]
        Bn, Ln = ids.shape
        assert t["top_ids"].shape == (Bn, Ln - 1, 8) and t["col_vals"].shape[:2] == (Bn, Ln - 1)
        vals = t["top_vals"].astype(np.float64).copy()
        for s in range(1, Ln - 1):
            vals[:, s][t["top_ids"][:, s] == ids[:, s][:, None]] = -10000.0
        order = np.argsort(-vals, axis=2)
        choice = np.take_along_axis(t["top_ids"], order[:, :, :1], axis=2)[:, :, 0]
        assert (choice == ids[:, 1:])[live].all(), name
        sv = np.take_along_axis(vals, order[:, :, :2], axis=2)
        m = sv[:, :, 0] - sv[:, :, 1]
        fin = live & np.isfinite(g["step_margin"])
        assert np.abs(m - t["margin"])[live].max() < 1e-5 and np.abs(t["margin"] - g["step_margin"])[fin].max() < 2e-4, name
        span = float(t["logit_max"]) - float(t["logit_min"])
        assert P.tf_bounds("f16", span)["thr"] == pytest.approx(2e-3 * span)      # ONE constant for every fixture
        if name.startswith("full_trained"):          # every margin >= 2 x the headline build's bound: identity is REQUIRED
            assert P.identity_required("f16", span, float(g["step_margin"].min())), name


def test_decode_pool_writes_the_pixels_pil_decodes(tmp_path):
    """decode_pool.DecodePool (the host side of the TSV task at the engine's rate): spawned worker processes read their rows from
    the TSV themselves, decode (PNG and JPEG) and write the RGB pixels into the shared staging buffer slot the parent named;
    the parent sees exactly the array load_image_by_pil gives.  An image larger than a slot comes back flagged (negative
    size) for the parent to decode itself; a row that cannot be decoded raises in the parent instead of hanging it."""
    from PIL import Image
    from generativeimage2text_amd.decode_pool import DecodePool
    rng = np.random.RandomState(4)
    rows, arrays = [], []
    for i in range(11):
        h, w = 40 + 7 * i, 90 - 3 * i
        arr = rng.randint(0, 255, (h, w, 3), dtype=np.uint8)
        buf = io.BytesIO()
        Image.fromarray(arr).save(buf, format="PNG" if i % 2 else "JPEG", quality=90)
        rows.append(["k%d" % i, base64.b64encode(buf.getvalue()).decode()])
        arrays.append(np.asarray(inference.load_image_by_pil(buf.getvalue())))
    rows.append(["broken", base64.b64encode(b"not an image").decode()])
    tsv_io.tsv_writer(rows, str(tmp_path / "img.tsv"))
    slot_bytes = 6000 * 3                                       # the four largest images (> 6000 pixels) do not fit
    pool = DecodePool(str(tmp_path / "img.tsv"), workers=3, slots=11, slot_bytes=slot_bytes)
    try:
        for i in range(11):
            pool.submit(slot=(i * 4) % 11, row=i)               # any slot order
        seen = {}
        for _ in range(11):
            slot, row, key, h, w = pool.next_result(timeout=60)
            assert key == "k%d" % row and slot == (row * 4) % 11
            seen[row] = (h, w)
            if h > 0:
                got = pool.buffer[slot * slot_bytes: slot * slot_bytes + h * w * 3].reshape(h, w, 3)
                assert np.array_equal(got, arrays[row]), row
        assert sorted(seen) == list(range(11))
        too_big = sorted(r for r, (h, w) in seen.items() if h < 0)
        assert too_big == [r for r in range(11) if arrays[r].size > slot_bytes] and len(too_big) >= 1
        assert all((-seen[r][0], -seen[r][1]) == arrays[r].shape[:2] for r in too_big)
        pool.submit(slot=0, row=11)
        with pytest.raises(RuntimeError, match="row 11"):
            pool.next_result(timeout=60)
    finally:
        pool.close()
