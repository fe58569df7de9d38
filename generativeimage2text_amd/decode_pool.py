"""JPEG decoding of the TSV task on worker PROCESSES, straight into a shared staging buffer.

`test_git_inference_single_tsv` (reference inference.py:171-212) decodes, transforms and runs the model strictly one image after
the other.  At the engine's rate (~10k captions/s per GPU) the host side has to deliver an image every 100 us; a Python thread pool
cannot (PIL hands the interpreter lock back only inside the entropy decoder: measured 1.4k images/s for 8 ... 64 threads on a
256-thread host, profiles/r06_b_e2e_tsv.json).  So:

  * N worker processes (fresh interpreters that import numpy + PIL only) each read their rows from the TSV file themselves
    (seek by the .lineidx.8b offsets), base64-decode, JPEG-decode to RGB and write the uint8 [H, W, 3] pixels into the slot the
    parent named -- a region of ONE shared staging buffer (a file in /dev/shm mapped by every process; the parent page-locks it
    for DMA when it can);
  * the parent receives (slot, row, key, H, W) records over one pipe, and once the slots of a batch are filled uploads the batch
    and runs the resize / crop / normalise for all of it in one launch pair per 24 images (gitmi_preprocess_batch).

Nothing here touches the GPU; `tests/test_host.py` runs it on the CPU.
"""
from __future__ import annotations

import base64
import io
import mmap
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


TASK = struct.Struct("<qq")                       # (slot, row); slot < 0: stop
RESULT_BYTES = 256
RESULT_HEAD = struct.Struct("<qqiii")             # slot, row, H, W, length of the key (or of an error text when H == W == 0)
KEY_MAX = RESULT_BYTES - RESULT_HEAD.size


def _read_exact(fd: int, n: int) -> bytes:
    buf = b""
    while len(buf) < n:
        chunk = os.read(fd, n - len(buf))
        if not chunk:
            return b""
        buf += chunk
    return buf


def worker_main(tsv_path: str, shm_path: str, shm_size: int, slot_bytes: int, task_fd: int, result_fd: int) -> None:
    """One worker: fixed-size task records in (shared pipe, one record per read), pixels into the shared staging buffer, one
    fixed-size result record out (a write below PIPE_BUF is atomic: all workers share one result pipe)."""
    from PIL import Image
    mem = _map_shared(shm_path, shm_size)
    offsets = _offsets_of(tsv_path)
    fp = open(tsv_path, "rb")

    def reply(slot, row, h, w, text: bytes):
        text = text[:KEY_MAX]
        os.write(result_fd, (RESULT_HEAD.pack(slot, row, h, w, len(text)) + text).ljust(RESULT_BYTES, b"\0"))

    while True:
        rec = _read_exact(task_fd, TASK.size)
        if not rec:
            break                                               # the parent closed the task pipe (or died)
        slot, row = TASK.unpack(rec)
        if slot < 0:
            break
        try:
            fp.seek(offsets[row])
            line = fp.readline()
            key, b64 = line.rstrip(b"\n").split(b"\t")[:2]
            img = Image.open(io.BytesIO(base64.b64decode(b64))).convert("RGB")       # load_image_by_pil
            w, h = img.size
            n = h * w * 3
            if len(key) > KEY_MAX:
                key = b""                                       # the parent reads long keys itself
            if n > slot_bytes:                                   # the parent decodes this one itself
                reply(slot, row, -h, -w, key)
                continue
            mem[slot * slot_bytes: slot * slot_bytes + n] = img.tobytes()       # raw RGB, row-major [H, W, 3]
            reply(slot, row, h, w, key)
        except Exception as exc:                                # a broken row must not hang the parent
            reply(slot, row, 0, 0, ("%s: %s" % (type(exc).__name__, exc)).encode("utf-8", "replace"))
    fp.close()


class DecodePool:
    """slots: number of image slots of `slot_bytes` bytes each in the shared staging buffer.

    Workers are forks of ONE small launcher interpreter (`python -m generativeimage2text_amd.decode_pool`, numpy + PIL only),
    NOT forks of this process and not multiprocessing children: forking a process that drives a GPU write-protects its whole
    address space (copy-on-write), the driver's MMU notifiers answer by evicting and restoring the process's GPU queues, and the
    engine stalls for seconds (measured: 24 forked workers = 0.5k captions/s end to end, the parent blocked in kernel launches);
    multiprocessing's spawn re-imports the parent's __main__ -- torch and all -- in every worker."""

    def __init__(self, tsv_path: str, workers: int, slots: int, slot_bytes: int = 1 << 20):
        import subprocess
        import sys
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
        n_workers = max(1, int(workers))
        # one task pipe PER worker (tasks go round robin): dozens of readers blocked on one shared pipe scale negatively (measured on
        # the 256-thread host: 7.8k images/s with 16 workers, 4.1k with 128); one result pipe for all (atomic 256-byte writes)
        task_pipes = [os.pipe() for _ in range(n_workers)]
        self._task_w = [w for _, w in task_pipes]
        self._next = 0
        self._result_r, result_w = os.pipe()
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
        # ONE child of this (large) process: a launcher that imports numpy + PIL once and FORKS the N workers -- starting a child
        # costs this process tens of ms each (its page tables are copied for the fork in front of the exec); the small launcher
        # forks a worker in about a millisecond
        task_r = [r for r, _ in task_pipes]
        cmd = [sys.executable, "-m", "generativeimage2text_amd.decode_pool", tsv_path, self.path, str(size), str(self.slot_bytes),
               str(result_w)] + [str(fd) for fd in task_r]
        self.procs = [subprocess.Popen(cmd, env=env, pass_fds=tuple(task_r) + (result_w,), stdin=subprocess.DEVNULL)]
        for fd in task_r:
            os.close(fd)
        os.close(result_w)
        self._closed = False

    def submit(self, slot: int, row: int) -> None:
        os.write(self._task_w[self._next], TASK.pack(int(slot), int(row)))
        self._next = (self._next + 1) % len(self._task_w)

    def next_result(self, timeout: Optional[float] = 120.0) -> Tuple[int, int, str, int, int]:
        """-> (slot, row, key, H, W) of the next finished image (any order); H, W negative: the image did not fit its slot (the
        caller decodes that row itself); key "" : longer than a result record holds (the caller reads it from the TSV); raises on
        a row that could not be decoded, on a dead pool and on a timeout."""
        import select
        if timeout is not None and not select.select([self._result_r], [], [], timeout)[0]:
            raise TimeoutError("decode pool: no result within %.0f s" % timeout)
        rec = _read_exact(self._result_r, RESULT_BYTES)
        if not rec:
            raise RuntimeError("decode pool: every worker has exited")
        slot, row, h, w, n = RESULT_HEAD.unpack_from(rec)
        text = rec[RESULT_HEAD.size: RESULT_HEAD.size + n].decode("utf-8", "replace")
        if h == 0 and w == 0:
            raise RuntimeError("row %d: %s" % (row, text))
        return slot, row, text, h, w

    def close(self) -> None:
        """Stop the workers and remove the backing file (the mapping itself goes when its last view does)."""
        if self._closed:
            return
        self._closed = True
        for fd in self._task_w:                              # EOF on its task pipe: a worker leaves its loop
            try:
                os.close(fd)
            except OSError:
                pass
        for p in self.procs:
            try:
                p.wait(timeout=10)
            except Exception:
                p.kill()
        try:
            os.close(self._result_r)
        except OSError:
            pass
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


if __name__ == "__main__":
    # the launcher: argv = tsv, staging file, its size, slot bytes, result fd, one task fd per worker
    import sys
    from PIL import Image  # noqa: F401  (imported once here; the forked workers share it)
    _a = sys.argv[1:]
    _result_fd, _task_fds = int(_a[4]), [int(v) for v in _a[5:]]
    _kids = []
    for _i, _fd in enumerate(_task_fds):
        _pid = os.fork()
        if _pid == 0:
            for _other in _task_fds:
                if _other != _fd:
                    os.close(_other)
            try:
                worker_main(_a[0], _a[1], int(_a[2]), int(_a[3]), _fd, _result_fd)
            finally:
                os._exit(0)
        _kids.append(_pid)
    for _fd in _task_fds + [_result_fd]:
        os.close(_fd)
    for _pid in _kids:
        os.waitpid(_pid, 0)
