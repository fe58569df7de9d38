"""JPEG decoding of the TSV task on worker PROCESSES, straight into a shared staging buffer.

`test_git_inference_single_tsv` (reference inference.py:171-212) decodes, transforms and runs the model strictly one image after
the other.  At the engine's rate (~10k captions/s per GPU) the host side has to deliver an image every 100 us; a Python thread pool
cannot (PIL hands the interpreter lock back only inside the entropy decoder: measured 1.4k images/s for 8 ... 64 threads on a
256-thread host, profiles/r06_b_e2e_tsv.json).  So:

  * N worker processes (forked; they run PIL + numpy code only) each read their rows from the TSV file themselves
    (seek by the .lineidx.8b offsets), base64-decode, JPEG-decode to RGB and write the uint8 [H, W, 3] pixels into the slot the
    parent named -- a region of ONE shared staging buffer (a file in /dev/shm mapped by every process; the parent page-locks it
    for DMA when it can);
  * the parent receives (slot, key, H, W) messages, and once the slots of a batch are filled uploads the batch with ONE copy and
    runs the resize / crop / normalise for all of it in one launch pair (gitmi_preprocess_batch).

Nothing here touches the GPU; `tests/test_host.py` runs it on the CPU.
"""
from __future__ import annotations

import base64
import io
import mmap
import multiprocessing as mp
import os
import struct
import tempfile
from typing import List, Optional, Tuple

import numpy as np


def _offsets_of(tsv_path: str) -> List[int]:
    idx = os.path.splitext(tsv_path)[0] + ".lineidx.8b"
    with open(idx, "rb") as f:
        raw = f.read()
    return list(struct.unpack("<%dQ" % (len(raw) // 8), raw))


def _map_shared(path: str, size: int):
    fd = os.open(path, os.O_RDWR)
    try:
        return mmap.mmap(fd, size)
    finally:
        os.close(fd)


def _worker(tsv_path: str, shm_path: str, shm_size: int, slot_bytes: int, tasks, results) -> None:
    from PIL import Image
    mem = _map_shared(shm_path, shm_size)
    offsets = _offsets_of(tsv_path)
    fp = open(tsv_path, "rb")
    while True:
        task = tasks.get()
        if task is None:
            break
        slot, row = task
        try:
            fp.seek(offsets[row])
            line = fp.readline()
            key, b64 = line.rstrip(b"\n").split(b"\t")[:2]
            img = Image.open(io.BytesIO(base64.b64decode(b64))).convert("RGB")       # load_image_by_pil
            w, h = img.size
            n = h * w * 3
            if n > slot_bytes:                                   # the parent decodes this one itself
                results.put((slot, row, key.decode(), -h, -w, None))
                continue
            mem[slot * slot_bytes: slot * slot_bytes + n] = img.tobytes()       # raw RGB, row-major [H, W, 3]
            results.put((slot, row, key.decode(), h, w, None))
        except Exception as exc:                                # a broken row must not hang the parent
            results.put((slot, row, "", 0, 0, "%s: %s" % (type(exc).__name__, exc)))
    fp.close()
    results.close()
    results.join_thread()                                   # flush what this worker still has to say
    os._exit(0)                                              # no interpreter teardown in a forked copy of the parent


class DecodePool:
    """slots: number of image slots of `slot_bytes` bytes each in the shared staging buffer."""

    def __init__(self, tsv_path: str, workers: int, slots: int, slot_bytes: int = 1 << 20):
        if not os.path.isfile(os.path.splitext(tsv_path)[0] + ".lineidx.8b"):
            from .tsv_io import build_lineidx
            build_lineidx(tsv_path)
        self.slots, self.slot_bytes = int(slots), int(slot_bytes)
        size = self.slots * self.slot_bytes
        shm_dir = "/dev/shm" if os.path.isdir("/dev/shm") else None          # memory-backed; any directory works
        fd, self.path = tempfile.mkstemp(prefix="gitmi_decode_", dir=shm_dir)
        os.ftruncate(fd, size)
        os.close(fd)
        self._mem = _map_shared(self.path, size)
        self.buffer = np.frombuffer(self._mem, dtype=np.uint8)                # the staging buffer, [slots * slot_bytes]
        # fork, as torch's DataLoader workers: a child runs _worker only (PIL + numpy, its own mapping of the staging file) and never
        # touches the GPU runtime it inherited; it leaves through os._exit.  (spawn would re-import the parent's __main__ -- torch
        # and all -- in every worker: measured 48 workers = 5 s of start-up and a slower pool than 8.)
        ctx = mp.get_context("fork")
        self.tasks, self.results = ctx.Queue(), ctx.Queue()
        self.procs = [ctx.Process(target=_worker, args=(tsv_path, self.path, size, self.slot_bytes, self.tasks, self.results),
                                  daemon=True) for _ in range(max(1, int(workers)))]
        for p in self.procs:
            p.start()
        self._closed = False

    def submit(self, slot: int, row: int) -> None:
        self.tasks.put((int(slot), int(row)))

    def next_result(self, timeout: Optional[float] = 120.0) -> Tuple[int, int, str, int, int]:
        """-> (slot, row, key, H, W) of the next finished image (any order); H, W negative: the image did not fit its slot (the
        caller decodes that row itself); raises on a row that could not be decoded."""
        slot, row, key, h, w, err = self.results.get(timeout=timeout)
        if err is not None:
            raise RuntimeError("row %d: %s" % (row, err))
        return slot, row, key, h, w

    def close(self) -> None:
        """Stop the workers and remove the backing file (the mapping itself goes when its last view does)."""
        if self._closed:
            return
        self._closed = True
        for _ in self.procs:
            self.tasks.put(None)
        for p in self.procs:
            p.join(timeout=10)
            if p.is_alive():
                p.terminate()
        self.buffer = None
        try:
            os.unlink(self.path)
        except OSError:
            pass

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
