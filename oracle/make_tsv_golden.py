"""Pin the TSV wire format against the REAL reference and freeze it as a fixture.

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

Usage:  python oracle/make_tsv_golden.py
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


def main():
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
