"""TSV row files with an 8-byte little-endian offset index -- wire-compatible with the reference's
`TSVFile` / `tsv_writer` / `concat_tsv_files` (generativeimage2text/tsv_io.py:22-31, 121-374):

  <name>.tsv           rows of tab separated fields, '\\n' terminated
  <name>.lineidx.8b    one little-endian uint64 byte offset per row

Only what `test_git_inference_single_tsv` needs (random row access, streaming write, concat).
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
        return self._fp.readline().decode("utf-8").rstrip("\n").split("\t")

    def __iter__(self) -> Iterator[List[str]]:
        for i in range(len(self)):
            yield self[i]

    def close(self) -> None:
        if self._fp is not None:
            self._fp.close()
            self._fp = None


def tsv_reader(tsv_file: str) -> Iterator[List[str]]:
    with open(tsv_file, "r", encoding="utf-8") as f:
        for line in f:
            yield line.rstrip("\n").split("\t")


def tsv_writer(rows: Iterable[Sequence], tsv_file: str) -> None:
    """Stream rows to <tsv_file> and its .lineidx.8b, via temporary names then rename
    (reference tsv_io.py:356-374)."""
    d = os.path.dirname(tsv_file)
    if d:
        os.makedirs(d, exist_ok=True)
    tmp, tmp_idx = tsv_file + ".tmp", _idx8b(tsv_file) + ".tmp"
    pos = 0
    with open(tmp, "wb") as f, open(tmp_idx, "wb") as fi:
        for row in rows:
            assert row is not None
            line = ("\t".join(x if isinstance(x, str) else str(x) for x in row) + "\n").encode("utf-8")
            f.write(line)
            fi.write(struct.pack("<Q", pos))
            pos += len(line)
    os.replace(tmp, tsv_file)
    os.replace(tmp_idx, _idx8b(tsv_file))


def concat_tsv_files(tsvs: Sequence[str], out_tsv: str) -> None:
    """Byte-concatenate shard files and rebase their offsets (reference tsv_io.py:22-31, 57-85)."""
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
