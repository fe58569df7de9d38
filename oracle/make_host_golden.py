"""Pin the host-side formats (SURVEY.md 8f-2 TSV wire format, 8f-4 checkpoint key alignment) against the REAL
reference and freeze them as fixtures.

Runs only in the build container (needs /root/reference).  The reference's tsv_io.py imports `azfuse.File`
(absent here, tsv_io.py:8) and, inside concate_lineidx_8b, `pathos` (tsv_io.py:40); both only wrap local file
operations on this path, so they are replaced by local-filesystem stand-ins:
  * azfuse.File -> open / isfile / getsize / no-op prepare,
  * tsv_io.parallel_map -> a plain loop.
Everything that defines the FORMAT (tsv_writer, TSVFile, tsv_reader, concat_tsv_files with its offset rebasing,
json_dump of common.py) is the reference's own code.  What is frozen into tests/golden/tsv_wire.npz:
  * the bytes of <name>.tsv / .lineidx / .lineidx.8b the reference's tsv_writer produces for a set of rows that
    exercises unicode, empty fields, inner spaces, numbers, one-column rows, base64 payloads,
  * the rows its TSVFile[i], iteration and tsv_reader hand back,
  * the bytes of concat_tsv_files over two shards,
  * json_dump strings of nested objects (key order, separators, non-ASCII escaping),
  * the caption / VQA rows of inference.py:199, 212 written by the reference's writer and what
    convert_tsv_to_vqa_json / convert_tsv_to_coco_format-style readers see.

Second fixture, tests/golden/state_dict_align.json: torch_common.load_state_dict's key handling
(torch_common.py:45-54, 93-145: strip every leading 'module.', give each model key the loaded key that is its longest
suffix, drop model keys nothing matches) run on synthetic key sets: plain, DataParallel-wrapped once and twice,
checkpoints saved from a sub-module (shorter keys), ambiguous suffixes, unmatched keys on both sides.

Third fixture, tests/golden/model_params.json: every aux_data/models/*/parameter.yaml of the reference read with the
reference's own load_from_yaml_file (tsv_io.py:92-107) -- the table generativeimage2text_amd/configs.py restates --
plus a `_base_` chain resolved by that loader (the include mechanism of tsv_io.py:97-106).

Usage:  python oracle/make_host_golden.py
"""
from __future__ import annotations

import base64
import json
import os
import shutil
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference"
GOLD = os.path.join(ROOT, "tests", "golden", "tsv_wire.npz")


class _LocalFile:
    """Local-filesystem stand-in for azfuse.File (only the calls tsv_io.py makes)."""

    @staticmethod
    def open(name, mode="r"):
        d = os.path.dirname(name)
        if d and ("w" in mode or "a" in mode):
            os.makedirs(d, exist_ok=True)
        return open(name, mode)

    @staticmethod
    def isfile(name):
        return os.path.isfile(name)

    @staticmethod
    def prepare(names):
        return None

    @staticmethod
    def get_file_size(name):
        return os.path.getsize(name)


def import_reference_tsv():
    az = types.ModuleType("azfuse")
    az.File = _LocalFile
    sys.modules["azfuse"] = az
    sys.path.insert(0, REF)
    from generativeimage2text import tsv_io as R
    from generativeimage2text import common as C
    R.parallel_map = lambda func, tasks, num_worker=0: [func(t) for t in tasks]
    return R, C


def rows_fixture():
    jpeg_like = base64.b64encode(bytes(range(256)) * 3).decode()
    return [
        ["img_0", jpeg_like],
        ["img 1 with spaces", "café 中文"],
        ["k2", ""],
        ["k3", "  padded field  ", "third"],
        [4, 5.5, "mixed"],
        ["only_one_column"],
    ]


def read_bytes(path):
    with open(path, "rb") as f:
        return np.frombuffer(f.read(), dtype=np.uint8).copy()


def align_cases():
    model_keys = ["image_encoder.conv1.weight", "image_encoder.ln_pre.weight", "image_encoder.ln_pre.bias",
                  "image_encoder.transformer.resblocks.0.ln_1.weight", "image_encoder.transformer.resblocks.0.ln_2.weight",
                  "image_encoder.transformer.resblocks.10.ln_1.weight", "image_encoder.transformer.resblocks.1.ln_1.weight",
                  "textual.embedding.words.weight", "textual.output.weight", "textual.output.bias",
                  "textual.transformer.encoder.layer.0.output.dense.weight",
                  "textual.transformer.encoder.layer.0.attention.output.dense.weight",
                  "textual.transformer.encoder.layer.0.output.LayerNorm.weight",
                  "textual.transformer.encoder.layer.0.attention.output.LayerNorm.weight",
                  "img_temperal_embedding.0", "textual.visual_projection.0.weight"]
    cases = {
        "plain": list(model_keys) + ["image_encoder.proj", "extra.unused.key"],
        "module_once": ["module." + k for k in model_keys],
        "module_twice": ["module.module." + k for k in model_keys[:6]] + model_keys[6:],
        # saved from sub-modules: shorter keys that are suffixes of the model's
        "submodule": ["conv1.weight", "ln_pre.weight", "ln_pre.bias", "resblocks.0.ln_1.weight", "0.ln_2.weight",
                      "resblocks.10.ln_1.weight", "1.ln_1.weight", "words.weight", "output.weight", "output.bias",
                      "layer.0.output.dense.weight", "attention.output.dense.weight", "output.LayerNorm.weight",
                      "attention.output.LayerNorm.weight", "img_temperal_embedding.0", "visual_projection.0.weight"],
        # several loaded keys are suffixes of one model key: the longest wins
        "ambiguous": ["weight", "dense.weight", "output.dense.weight", "attention.output.dense.weight", "bias",
                      "output.bias", "ln_1.weight", "0.ln_1.weight", "resblocks.0.ln_1.weight"],
        "nothing_matches": ["foo.bar", "baz"],
    }
    return model_keys, cases


def make_align_fixture():
    import torch
    az = types.ModuleType("azfuse")
    az.File = _LocalFile
    sys.modules.setdefault("azfuse", az)
    if REF not in sys.path:
        sys.path.insert(0, REF)
    from generativeimage2text import torch_common as T
    model_keys, cases = align_cases()
    out = {"model_keys": model_keys, "cases": {}}
    for name, loaded_keys in cases.items():
        # tensors carry an id so that the mapping model key -> loaded key can be read back
        loaded = {k: torch.tensor([float(i)]) for i, k in enumerate(loaded_keys)}
        model_sd = {k: torch.tensor([-1.0]) for k in model_keys}
        stripped = T.strip_prefix_if_present(loaded, prefix="module.")            # torch_common.py:94
        T.align_and_update_state_dicts(model_sd, stripped)                        # torch_common.py:95
        out["cases"][name] = {"loaded_keys": loaded_keys,
                              "mapping": {k: loaded_keys[int(v.item())] for k, v in model_sd.items()}}
    path = os.path.join(ROOT, "tests", "golden", "state_dict_align.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote", path, {n: len(c["mapping"]) for n, c in out["cases"].items()})


def make_param_fixture(R):
    import glob
    out = {"models": {}, "base_chain": {}}
    for p in sorted(glob.glob(os.path.join(REF, "aux_data", "models", "*", "parameter.yaml"))):
        out["models"][os.path.basename(os.path.dirname(p))] = R.load_from_yaml_file(p)
    tmp = tempfile.mkdtemp(prefix="yamlgold_")
    try:
        files = {
            "root.yaml": "image_encoder_type: CLIPViT_L_14\nvisual_feature_size: 1024\nnested:\n  a: 1\n  b: 2\n",
            "mid.yaml": "_base_: root.yaml\ntest_crop_size: 420\nnested:\n  b: 3\n",
            "leaf.yaml": "_base_: mid.yaml\ntest_respect_ratio_max: 560\nnested:\n  c: 4\n",
        }
        for n, txt in files.items():
            with open(os.path.join(tmp, n), "w") as f:
                f.write(txt)
        out["base_chain"] = {"files": files, "leaf": R.load_from_yaml_file(os.path.join(tmp, "leaf.yaml"))}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    path = os.path.join(ROOT, "tests", "golden", "model_params.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote", path, sorted(out["models"]), out["base_chain"]["leaf"])


def main():
    make_align_fixture()
    make_param_fixture(import_reference_tsv()[0])
    R, C = import_reference_tsv()
    tmp = tempfile.mkdtemp(prefix="tsvgold_")
    old_tmp = os.environ.get("GIT_TMP_FOLDER")
    os.environ["GIT_TMP_FOLDER"] = tmp
    out = {}
    try:
        rows = rows_fixture()
        a = os.path.join(tmp, "a.tsv")
        R.tsv_writer(rows, a)
        for ext in (".tsv", ".lineidx", ".lineidx.8b"):
            out["a" + ext] = read_bytes(os.path.splitext(a)[0] + ext)
        t = R.TSVFile(a)
        out["a_len"] = np.int64(len(t))
        out["a_rows_getitem"] = np.array(json.dumps([t[i] for i in range(len(t))]))
        out["a_rows_iter"] = np.array(json.dumps([r for r in R.TSVFile(a)]))
        out["a_rows_reader"] = np.array(json.dumps([r for r in R.tsv_reader(a)]))
        out["a_keys"] = np.array(json.dumps([R.TSVFile(a).get_key(i) for i in (0, 1, 3)]))

        # caption / VQA rows exactly as inference.py:199, 212 yield them
        cap_rows = [("img_%d" % i, C.json_dump([{"caption": c}])) for i, c in
                    enumerate(["a dog on a couch", "café \"quoted\"", ""])]
        vqa_rows = [(C.json_dump({"answer": a_, "question_id": q}),) for a_, q in
                    [("yes", 17), ("two", 4), ("café", 900001)]]
        b = os.path.join(tmp, "caps.0.2.tsv")
        c = os.path.join(tmp, "caps.1.2.tsv")
        R.tsv_writer(cap_rows[:2], b)
        R.tsv_writer(cap_rows[2:], c)
        allp = os.path.join(tmp, "caps.tsv")
        R.concat_tsv_files([b, c], allp)
        for name, p in (("caps0", b), ("caps1", c), ("caps_all", allp)):
            out[name + ".tsv"] = read_bytes(p)
            out[name + ".lineidx.8b"] = read_bytes(os.path.splitext(p)[0] + ".lineidx.8b")
        out["caps_all_rows"] = np.array(json.dumps([r for r in R.TSVFile(allp)]))
        v = os.path.join(tmp, "vqa.tsv")
        R.tsv_writer(vqa_rows, v)
        out["vqa.tsv"] = read_bytes(v)
        # convert_tsv_to_vqa_json (inference.py:227-229) minus the file write: what it hands to json_dump
        out["vqa_json"] = np.array(C.json_dump([json.loads(s) for s, in R.tsv_reader(v)]))
        out["json_dump_cases"] = np.array(json.dumps([
            C.json_dump({"b": 1, "a": [1, 2, {"z": None, "y": True}], "c": "café 中"}),
            C.json_dump([{"caption": "x"}]),
            C.json_dump({"answer": "no", "question_id": 3}),
        ]))
    finally:
        if old_tmp is None:
            os.environ.pop("GIT_TMP_FOLDER", None)
        else:
            os.environ["GIT_TMP_FOLDER"] = old_tmp
        shutil.rmtree(tmp, ignore_errors=True)
    os.makedirs(os.path.dirname(GOLD), exist_ok=True)
    np.savez_compressed(GOLD, **out)
    print("wrote", GOLD, {k: (v.shape if v.ndim else str(v)[:60]) for k, v in out.items()})


if __name__ == "__main__":
    main()
