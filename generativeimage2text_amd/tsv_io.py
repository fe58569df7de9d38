"""TSV row files with an 8-byte little-endian offset index -- wire-compatible with the reference's
`TSVFile` / `tsv_writer` / `concat_tsv_files` (generativeimage2text/tsv_io.py:22-31, 121-374):

  <name>.tsv           rows of tab separated fields, '\\n' terminated
  <name>.lineidx       the byte offset of every row as decimal text, one per line
  <name>.lineidx.8b    one little-endian uint64 byte offset per row

Only what `test_git_inference_single_tsv` and the converters of its outputs need (random row access, streaming
write, concat).  The format is pinned against the reference's own writer / reader / concat by
oracle/make_host_golden.py -> tests/golden/tsv_wire.npz (tests/test_host.py).
"""
from __future__ import annotations

import os
import struct
from typing import Iterable, Iterator, List, Sequence


def _idx8b(tsv_path: str) -> str:
    return os.path.splitext(tsv_path)[0] + ".lineidx.8b"


def _idx_txt(tsv_path: str) -> str:
    return os.path.splitext(tsv_path)[0] + ".lineidx"


def build_lineidx(tsv_path: str) -> List[int]:
    offsets = []
    with open(tsv_path, "rb") as f:
        pos = 0
        for line in f:
            offsets.append(pos)
            pos += len(line)
    with open(_idx8b(tsv_path), "wb") as f:
        for o in offsets:
            f.write(struct.pack("<Q", o))
    return offsets


class TSVFile:
    """Random access to rows: ``tsv[i] -> list[str]`` (reference tsv_io.py:121-354)."""

    def __init__(self, tsv_file: str):
        self.tsv_file = tsv_file
        self._fp = None
        self._offsets: List[int] = []
        if os.path.isfile(_idx8b(tsv_file)):
            with open(_idx8b(tsv_file), "rb") as f:
                raw = f.read()
            self._offsets = list(struct.unpack("<%dQ" % (len(raw) // 8), raw))
        elif os.path.isfile(_idx_txt(tsv_file)):
            with open(_idx_txt(tsv_file)) as f:
                self._offsets = [int(x) for x in f.read().split()]
        else:
            self._offsets = build_lineidx(tsv_file)

    def __len__(self) -> int:
        return len(self._offsets)

    def num_rows(self) -> int:
        return len(self)

    def __getitem__(self, idx: int) -> List[str]:
        if self._fp is None:
            self._fp = open(self.tsv_file, "rb")
        self._fp.seek(self._offsets[idx])
        # fields are whitespace-stripped like the reference's TSVFile.seek / __iter__ (tsv_io.py:199-212, 233-238)
        return [x.strip() for x in self._fp.readline().decode("utf-8").split("\t")]

    def get_key(self, idx: int) -> str:
        """First column of row idx (reference tsv_io.py:223-224, 264-268: read up to the first tab, not stripped)."""
        if self._fp is None:
            self._fp = open(self.tsv_file, "rb")
        self._fp.seek(self._offsets[idx])
        line = self._fp.readline()
        assert b"\t" in line, "get_key needs a second column (the reference reads up to the first tab)"
        return line[:line.index(b"\t")].decode("utf-8")

    def __iter__(self) -> Iterator[List[str]]:
        for i in range(len(self)):
            yield self[i]

    def close(self) -> None:
        if self._fp is not None:
            self._fp.close()
            self._fp = None


def tsv_reader(tsv_file: str, sep: str = "\t") -> Iterator[List[str]]:
    """Sequential rows, fields stripped (reference tsv_io.py:87-90)."""
    with open(tsv_file, "r", encoding="utf-8") as f:
        for line in f:
            yield [x.strip() for x in line.split(sep)]


def tsv_writer(rows: Iterable[Sequence], tsv_file: str) -> None:
    """Stream rows to <tsv_file>, its .lineidx and its .lineidx.8b -- byte for byte what the reference's
    tsv_writer produces (tsv_io.py:356-374: fields are str()-ed unless bytes).  Written under temporary names and
    renamed into place, the data file last: a reader polling for the .tsv (the multi-rank hand-off of
    inference.py:214-225) never sees a partial shard."""
    d = os.path.dirname(tsv_file)
    if d:
        os.makedirs(d, exist_ok=True)
    tmp, tmp_txt, tmp_idx = tsv_file + ".tmp", _idx_txt(tsv_file) + ".tmp", _idx8b(tsv_file) + ".tmp"
    pos = 0
    with open(tmp, "wb") as f, open(tmp_txt, "w") as ft, open(tmp_idx, "wb") as fi:
        for row in rows:
            assert row is not None
            line = b"\t".join(x if isinstance(x, bytes) else str(x).encode("utf-8") for x in row) + b"\n"
            f.write(line)
            ft.write(str(pos) + "\n")
            fi.write(struct.pack("<Q", pos))
            pos += len(line)
    os.replace(tmp_txt, _idx_txt(tsv_file))
    os.replace(tmp_idx, _idx8b(tsv_file))
    os.replace(tmp, tsv_file)


def concat_tsv_files(tsvs: Sequence[str], out_tsv: str) -> None:
    """Byte-concatenate shard files and rebase their .lineidx.8b offsets (reference tsv_io.py:22-31, 57-85; like
    the reference, no text .lineidx is produced for the concatenation)."""
    if len(tsvs) == 1 and tsvs[0] == out_tsv:
        return
    base = 0
    with open(out_tsv + ".tmp", "wb") as fo, open(_idx8b(out_tsv) + ".tmp", "wb") as fi:
        for t in tsvs:
            offs = TSVFile(t)._offsets
            with open(t, "rb") as f:
                data = f.read()
            fo.write(data)
            for o in offs:
                fi.write(struct.pack("<Q", base + o))
            base += len(data)
    os.replace(out_tsv + ".tmp", out_tsv)
    os.replace(_idx8b(out_tsv) + ".tmp", _idx8b(out_tsv))


# ---- consumers of the task outputs (inference.py:227-252) ------------------------------------------------
def json_dump(obj) -> str:
    """common.py:223-226: sorted keys, compact separators -- what the reference writes into every TSV row."""
    import json
    return json.dumps(obj, sort_keys=True, separators=(",", ":"))


def convert_tsv_to_vqa_json(predict_file: str, out_json: str) -> None:
    """One-column VQA rows `{"answer":..,"question_id":..}` -> one JSON list (inference.py:227-229)."""
    import json
    result = [json.loads(s) for s, in tsv_reader(predict_file)]
    d = os.path.dirname(out_json)
    if d:
        os.makedirs(d, exist_ok=True)
    with open(out_json, "wb") as f:
        f.write(json_dump(result).encode())


def convert_tsv_to_coco_format(res_tsv: str, outfile: str, sep: str = "\t", key_col: int = 0, cap_col: int = 1) -> None:
    """Caption rows `key \\t [{"caption": ..}]` -> COCO result JSON `[{"image_id", "caption"}]` (inference.py:231-252;
    a row without a caption column, or with an empty list, counts as the empty caption)."""
    import json
    results = []
    with open(res_tsv) as fp:
        for line in fp:
            parts = line.strip().split(sep)
            key = parts[key_col]
            if cap_col < len(parts):
                caps = json.loads(parts[cap_col])
                if len(caps) == 0:
                    caps = [{"caption": ""}]
                assert len(caps) == 1, "cannot evaluate multiple captions per image"
                cap = caps[0]["caption"]
            else:
                cap = ""
            results.append({"image_id": key, "caption": cap})
    with open(outfile, "w") as fp:
        json.dump(results, fp)
