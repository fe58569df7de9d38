"""The two inference tasks of the reference, same names and positional signatures
(generativeimage2text/inference.py:67-109 and :134-225), running on the HIP engine.

Differences forced by the environment, not by design:
  * torchvision / azfuse are not imported: the image transform is restated on PIL (torchvision's
    Resize/CenterCrop on PIL images call PIL themselves), checkpoints are read with torch.load.
  * the WordPiece vocabulary is looked up offline ($GIT_VOCAB or aux_data/vocab.txt, then the HF cache);
    without one the tasks fail loudly unless GIT_VOCAB=ids asks for raw token ids.
  * images are decoded on a few host threads ahead of the GPU (order kept; GIT_DECODE_THREADS=0 restores the serial loop)
    and batched (the reference runs batch 1, one host sync per image; models with test_respect_ratio_max batch the images
    that share a resized shape) with several batches in flight on the device, and ranks return
    their results through one RCCL gather (the task forms the process group itself from the launcher's
    RANK/WORLD_SIZE/MASTER_* variables) and fall back to the shared-filesystem poll + concat of
    inference.py:214-225 when no group can be formed; shard files `{out}.{rank}.{world}.tsv` are always written.
  * the task functions default to precision="f16" (round 6): fp16 operands, the 16-bit build that meets the specification's
    logit tolerance (1e-3 of the logit span) on every weight family incl. trained-checkpoint statistics, and the one bench.py's
    headline measures.  "f32" = token ids bit-identical to the reference's fp32 run; "bf16" = BASELINE.json's named operand
    format (+2 % captions/s, 5e-3 of the span on trained-like weights).  Environment read by this module (and by nothing in the
    shared libraries): GIT_VOCAB (vocabulary file or "ids"), GIT_DECODE_THREADS / GIT_DECODE_PROCS / GIT_DECODE_SLOT_MB (host JPEG decoding of the TSV task: threads of the
    per-image path, worker processes and slot size of the pooled captioning path), the
    launcher's RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (and OMPI_COMM_WORLD_*).
"""
from __future__ import annotations

import base64
import io
import json
import logging
import os
import os.path as op
from typing import List, Optional, Sequence

import numpy as np
import torch

from .configs import MODEL_PARAMS, config_for_model, config_from_param, load_model_param      # noqa: F401
from .model import GeneratorWithBeamSearch, CaptioningModel
from .tsv_io import (TSVFile, tsv_writer, concat_tsv_files, json_dump,                 # noqa: F401  (re-exported:
                     convert_tsv_to_vqa_json, convert_tsv_to_coco_format)                    #  the reference has them here)

MAX_VQA_QUESTIONS = 16      # questions of one image answered in one engine call (batched ragged prefixes)

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)      # inference.py:126-129
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


# ---- rank helpers (common.py:106-119) ---------------------------------------------------------
def get_mpi_rank() -> int:
    return int(os.environ.get("RANK", os.environ.get("OMPI_COMM_WORLD_RANK", "0")))


def get_mpi_local_rank() -> int:
    return int(os.environ.get("LOCAL_RANK", os.environ.get("OMPI_COMM_WORLD_LOCAL_RANK", "0")))


def get_mpi_size() -> int:
    return int(os.environ.get("WORLD_SIZE", os.environ.get("OMPI_COMM_WORLD_SIZE", "1")))


def shard_range(num_rows: int, rank: int, world: int):
    """Contiguous rows of this rank, inference.py:165-169."""
    per = (num_rows + world - 1) // world
    start = per * rank
    return start, min(start + per, num_rows)


def effective_cpus() -> int:
    """CPUs this process may actually use: the affinity mask, capped by the cgroup's CPU quota (a container on a 256-thread host
    is routinely limited to a few cores' worth of time: cpu.max = "1600000 100000" is 16)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", ):
        try:
            quota, period = open(path).read().split()[:2]
            if quota != "max":
                n = min(n, max(1, int(int(quota) / int(period))))
        except (OSError, ValueError):
            pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p_ = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0:
            n = min(n, max(1, q // p_))
    except (OSError, ValueError):
        pass
    return n


# ---- tokenizer --------------------------------------------------------------------------------
class IdTokenizer:
    """Explicit opt-in (GIT_VOCAB=ids) for machines without a WordPiece vocabulary: ids in, ids out."""
    cls_token_id, sep_token_id = 101, 102

    def __call__(self, text, **kw):
        words = text.split()
        if not all(w.isdigit() for w in words):
            raise ValueError("IdTokenizer (GIT_VOCAB=ids) only accepts space-separated token ids, got %r; "
                             "point GIT_VOCAB at a bert-base-uncased vocab.txt to use text" % text)
        return {"input_ids": [int(x) for x in words]}

    def decode(self, ids, skip_special_tokens=True):
        if skip_special_tokens:
            ids = [i for i in ids if i not in (0, 100, 101, 102, 103)]
        return " ".join(str(i) for i in ids)


def get_tokenizer():
    """BertTokenizer.from_pretrained('bert-base-uncased', do_lower_case=True) (inference.py:72), offline:
    $GIT_VOCAB (a vocab.txt path; default aux_data/vocab.txt), else the local HF cache.  Fails loudly when no
    vocabulary is found; GIT_VOCAB=ids selects the id-passthrough stand-in explicitly."""
    vocab = os.environ.get("GIT_VOCAB", "aux_data/vocab.txt")
    if vocab == "ids":
        return IdTokenizer()
    from transformers import BertTokenizer
    if op.isfile(vocab):
        return BertTokenizer(vocab, do_lower_case=True)
    os.environ.setdefault("HF_HUB_OFFLINE", "1")
    why = "empty tokenizer"
    try:
        tok = BertTokenizer.from_pretrained("bert-base-uncased", do_lower_case=True)
        if len(tok) >= 30522:          # (an offline from_pretrained may hand back a tokenizer with only the special tokens)
            return tok
    except Exception as exc:
        why = type(exc).__name__
    raise FileNotFoundError(
        f"no bert-base-uncased vocabulary: {vocab} does not exist and the HF cache has none ({why}). "
        f"Set GIT_VOCAB=/path/to/vocab.txt, or GIT_VOCAB=ids to work with raw token ids.")


# ---- image transform (inference.py:111-132) ------------------------------------------------------
def load_image_by_pil(path_or_bytes):
    from PIL import Image
    if isinstance(path_or_bytes, (bytes, bytearray)):
        return Image.open(io.BytesIO(path_or_bytes)).convert("RGB")
    return Image.open(path_or_bytes).convert("RGB")


def image_transform(img, crop_size: int = 224) -> torch.Tensor:
    """Resize(crop, BICUBIC) -> CenterCrop(crop) -> RGB -> ToTensor -> Normalize(CLIP mean/std)."""
    from PIL import Image
    w, h = img.size
    if w <= h:
        nw, nh = crop_size, int(crop_size * h / w)
    else:
        nw, nh = int(crop_size * w / h), crop_size
    if (w, h) != (nw, nh):
        img = img.resize((nw, nh), Image.BICUBIC)
    left, top = int(round((nw - crop_size) / 2.0)), int(round((nh - crop_size) / 2.0))
    img = img.crop((left, top, left + crop_size, top + crop_size)).convert("RGB")
    x = torch.from_numpy(np.asarray(img, dtype=np.uint8).copy()).permute(2, 0, 1).float().div_(255.0)
    mean = torch.tensor(CLIP_MEAN).view(3, 1, 1)
    std = torch.tensor(CLIP_STD).view(3, 1, 1)
    return (x - mean) / std


def decode_to_array(path_or_bytes) -> np.ndarray:
    """load_image_by_pil + the RGB uint8 [H,W,3] array the GPU transforms upload: everything of an image that can run on a host
    thread ahead of the GPU (PIL releases the GIL while it decodes)."""
    return np.array(load_image_by_pil(path_or_bytes), dtype=np.uint8)          # a writable copy: torch.from_numpy takes it as is


def _rgb_array(img) -> np.ndarray:
    """PIL image or an already decoded uint8 [H,W,3] array -> the array (no copy for arrays)"""
    if isinstance(img, np.ndarray):
        return img
    return np.asarray(img.convert("RGB"), dtype=np.uint8)


def gpu_image_transform(img, crop_size: int = 224) -> torch.Tensor:
    """Same transform on the GPU (csrc/kernels_preproc.hip): the decoded uint8 RGB image is uploaded as is and
    Pillow's bicubic resampler, the centre crop and the normalisation run in two HIP kernels -- bit-exact with
    image_transform() (tests/test_preprocess.py).  JPEG decoding stays on the host (PIL).  img: PIL image or uint8 [H,W,3]."""
    from .engine import preprocess_image
    arr = _rgb_array(img)
    if not arr.flags.writeable:
        arr = arr.copy()
    return preprocess_image(torch.from_numpy(arr).cuda(non_blocking=True), crop_size)


class MinMaxResizeForTest(object):
    """inference.py:29-64: keep the aspect ratio, short side -> min_size unless that would push the long side over
    max_size (then the long side -> max_size).  get_size() returns (height, width)."""

    def __init__(self, min_size, max_size):
        self.min_size = min_size
        self.max_size = max_size

    def get_size(self, image_size):
        w, h = image_size
        size = self.min_size
        lo, hi = float(min(w, h)), float(max(w, h))
        if hi / lo * size > self.max_size:
            size = int(round(self.max_size * lo / hi))
        if (w <= h and w == size) or (h <= w and h == size):
            return (h, w)
        if w < h:
            return (int(size * h / w), size)
        return (size, int(size * w / h))

    def __repr__(self):
        return "MinMaxResizeForTest({}, {})".format(self.min_size, self.max_size)

    def __call__(self, img):
        from PIL import Image
        oh, ow = self.get_size(img.size)
        return img.resize((ow, oh), Image.BICUBIC)          # torchvision F.resize(img, (h, w)) on a PIL image


def minmax_image_transform(img, min_size: int, max_size: int) -> torch.Tensor:
    """MinMaxResizeForTest -> ToTensor -> Normalize (inference.py:113-116, 123-131); no crop, no RGB conversion step
    of its own (load_image_by_pil already returns RGB)."""
    img = MinMaxResizeForTest(min_size, max_size)(img)
    x = torch.from_numpy(np.asarray(img.convert("RGB"), dtype=np.uint8).copy()).permute(2, 0, 1).float().div_(255.0)
    mean = torch.tensor(CLIP_MEAN).view(3, 1, 1)
    std = torch.tensor(CLIP_STD).view(3, 1, 1)
    return (x - mean) / std


def gpu_minmax_image_transform(img, min_size: int, max_size: int) -> torch.Tensor:
    """The same on the GPU: Pillow-exact resize to get_size() + normalise (gitmi_preprocess_image_to).  img: PIL image or
    uint8 [H,W,3]."""
    from .engine import preprocess_image_to
    arr = _rgb_array(img)
    oh, ow = MinMaxResizeForTest(min_size, max_size).get_size((arr.shape[1], arr.shape[0]))
    if not arr.flags.writeable:
        arr = arr.copy()
    return preprocess_image_to(torch.from_numpy(arr).cuda(non_blocking=True), oh, ow)


def get_image_transform(param: dict, gpu: bool = False):
    """inference.py:111-132."""
    crop = param.get("test_crop_size", 224)
    if "test_respect_ratio_max" in param:
        mx = param["test_respect_ratio_max"]
        if gpu:
            return lambda im: gpu_minmax_image_transform(im, crop, mx)
        return lambda im: minmax_image_transform(im, crop, mx)
    if gpu:
        return lambda im: gpu_image_transform(im, crop)
    return lambda im: image_transform(im, crop)


# ---- model construction -----------------------------------------------------------------------
def load_checkpoint(model_name: str, checkpoint: Optional[str] = None):
    path = checkpoint or f"output/{model_name}/snapshot/model.pt"       # inference.py:84
    if not op.isfile(path):
        raise FileNotFoundError(
            f"checkpoint {path} not found (the reference downloads it through azfuse; this build is offline). "
            f"Place the file there or pass checkpoint=...")
    ckpt = torch.load(path, map_location="cpu")
    return ckpt["model"] if "model" in ckpt else ckpt


def build_model(model_name: str, tokenizer, checkpoint=None, max_batch: int = 64, precision: str = "f16",
                decoder=None, param: Optional[dict] = None) -> CaptioningModel:
    """param: the model's parameter dict when the caller already read it (a parameter.yaml); default: the built-in
    table entry of `model_name`."""
    cfg = config_for_model(model_name) if param is None else config_from_param(param, name=model_name)
    if decoder is None:
        decoder = GeneratorWithBeamSearch(eos_index=tokenizer.sep_token_id, max_steps=1024, beam_size=4,
                                          length_penalty=0.6)               # model.py:34-40
    model = CaptioningModel(cfg, decoder, precision=precision, max_batch=max_batch)
    state = checkpoint if isinstance(checkpoint, dict) else load_checkpoint(model_name, checkpoint)
    model.load_state_dict(state)
    return model


def _prefix_ids(tokenizer, prefix: str, max_text_len: int = 40) -> List[int]:
    """inference.py:92-101."""
    enc = tokenizer(prefix, padding="do_not_pad", truncation=True, add_special_tokens=False, max_length=max_text_len)
    payload = enc["input_ids"]
    if len(payload) > max_text_len - 2:
        payload = payload[-(max_text_len - 2):]
    return [tokenizer.cls_token_id] + payload


# ---- tasks --------------------------------------------------------------------------------------
def _task_param(model_name: str, yaml_dir: str):
    """-> (param, from_file): `<yaml_dir>/<model_name>/parameter.yaml` if present (the file the reference task
    reads), else the built-in table entry; a name in neither -> KeyError."""
    if op.isfile(op.join(yaml_dir, model_name, "parameter.yaml")):
        return load_model_param(model_name, yaml_dir), True
    if model_name not in MODEL_PARAMS:
        raise KeyError(f"unknown GIT model '{model_name}': no {yaml_dir}/{model_name}/parameter.yaml and not one of "
                       f"{sorted(MODEL_PARAMS)}")
    return dict(MODEL_PARAMS[model_name]), False


def test_git_inference_single_image(image_path, model_name, prefix, *, checkpoint=None, precision="f16"):
    """inference.py:67-109.  image_path: str or list of str (video frames); logs 'output: <caption>'.
    precision "f16" (default) = the headline 16-bit build; "f32" = the reference's arithmetic (ids bit-identical)."""
    param, from_file = _task_param(model_name, "aux_data/models")           # inference.py:68-70
    tokenizer = get_tokenizer()
    if isinstance(image_path, str):
        image_path = [image_path]
    transforms = get_image_transform(param, gpu=True)
    img = [transforms(load_image_by_pil(p)) for p in image_path]
    model = build_model(model_name, tokenizer, checkpoint, max_batch=1, precision=precision,
                        **({"param": param} if from_file else {}))
    model.cuda()
    model.eval()
    img = [i.unsqueeze(0).cuda() for i in img]
    input_ids = _prefix_ids(tokenizer, prefix)
    with torch.no_grad():
        result = model({"image": img, "prefix": torch.tensor(input_ids).unsqueeze(0).cuda()})
    cap = tokenizer.decode(result["predictions"][0].tolist(), skip_special_tokens=True)
    logging.info("output: {}".format(cap))
    test_git_inference_single_image.last_output = cap


def ensure_process_group(rank: int, world: int) -> bool:
    """Form the result-gather group of a multi-rank run when the launcher gave us a rendezvous
    (torchrun / torch.distributed.run export MASTER_ADDR+MASTER_PORT; so can an mpirun wrapper).
    RCCL ("nccl") when this rank has a GPU, gloo otherwise (CPU tests).  Returns False when no group
    can be formed (plain `mpirun -n 8` as in the reference README): the caller then falls back to the
    reference's shared-filesystem hand-off (inference.py:214-225)."""
    import torch.distributed as dist
    if world <= 1 or not dist.is_available():
        return False
    if dist.is_initialized():
        return True
    if "MASTER_ADDR" not in os.environ or "MASTER_PORT" not in os.environ:
        return False
    backend = "nccl" if torch.cuda.is_available() else "gloo"
    dist.init_process_group(backend, rank=rank, world_size=world)
    return True


def _gather_rows(rows: List[list]) -> Optional[List[list]]:
    """Result gather over RCCL (replaces the file poll of inference.py:214-225). Rank 0 gets all rows in
    rank order.  Only called with an initialised process group."""
    import torch.distributed as dist
    out = [None] * dist.get_world_size() if dist.get_rank() == 0 else None
    dist.gather_object(rows, out, dst=0)
    if dist.get_rank() != 0:
        return None
    return [r for part in out for r in part]


def _wait_and_concat_shards(out_tsv: str, world: int, poll_s: float = 0.2, timeout_s: float = 3600.0) -> None:
    """The reference's hand-off when ranks share nothing but a filesystem (inference.py:214-225): rank 0 waits
    for every `{out}.{rank}.{world}.tsv` (tsv_writer renames them into place when complete) and concatenates."""
    import time
    shards = [f"{out_tsv}.{r}.{world}.tsv" for r in range(world)]
    t0 = time.time()
    while True:
        not_ready = [t for t in shards if not op.isfile(t)]
        if not not_ready:
            break
        if time.time() - t0 > timeout_s:
            raise TimeoutError("shard files never appeared: " + ",".join(not_ready))
        logging.info("waiting {}".format(",".join(not_ready)))
        time.sleep(poll_s)
    concat_tsv_files(shards, out_tsv)


def prefetch_ordered(n: int, load, threads: int, window: int):
    """load(i) for i in range(n), computed by `threads` host threads up to `window` items ahead of the consumer and
    yielded IN ORDER.  threads <= 0: plain serial calls.  Used to keep JPEG decoding (PIL releases the GIL while it
    decodes) off the critical path of the GPU: the reference decodes, transforms and runs the model strictly one image
    after the other (inference.py:171-212)."""
    if threads <= 0 or n <= 1:
        for i in range(n):
            yield load(i)
        return
    import collections
    from concurrent.futures import ThreadPoolExecutor
    pending = collections.deque()
    with ThreadPoolExecutor(max_workers=threads) as pool:
        nxt = 0
        while nxt < n or pending:
            while nxt < n and len(pending) < max(1, window):
                pending.append(pool.submit(load, nxt))
                nxt += 1
            yield pending.popleft().result()


def run_tsv_inference(image_tsv: str, question_tsv: Optional[str], out_tsv: str, *, transform, caption_batch,
                      answer_questions, batch_size: int, rank: Optional[int] = None, world: Optional[int] = None,
                      poll_s: float = 0.2, decode=None, decode_threads: int = 0, submit_captions=None, submit_answers=None,
                      in_flight: int = 1, stats: Optional[dict] = None, batch_source=None,
                      max_questions: Optional[int] = None) -> None:
    """Everything of test_git_inference_single_tsv (inference.py:134-225) except the model: shard the rows by
    rank (:165-169), write this rank's rows, and deliver the complete, ordered `out_tsv` on rank 0 --
    through ONE RCCL gather when a process group exists or can be formed, else through the reference's
    shard-file poll + concat.  Rank 0 never writes `out_tsv` from its own rows alone.

      transform(bytes) -> image tensor;  caption_batch(list of images) -> list of caption strings;
      answer_questions(image, list of question strings) -> list of answer strings.
      decode (optional): bytes -> decoded image, run on `decode_threads` host threads ahead of the loop (order kept);
      `transform` then receives the decoded image instead of the bytes and stays on the calling thread (it may launch
      GPU work on the caller's device and stream).
      submit_captions (optional): list of images -> handle whose .result() is the list of caption strings; up to `in_flight`
      batches are kept enqueued on the device while the host decodes the next ones (FIFO: rows stay in input order).
      submit_answers (optional): (list of same-shape images, list of question lists) -> handle whose .result() is one answer
      list per image; images are bucketed BY SHAPE (aspect-preserving resize: a COCO-style set has two or three shapes) up to
      `batch_size` images or questions per call instead of one call per image; rows are written in input order.
      batch_source (optional, captioning only): callable(start, end) -> iterator of (keys, device batch) in input order,
      replacing the decode / transform loop of this function (pooled_caption_batches: worker processes + batched GPU transform).
    Row formats are the reference's: `key \t json_dump([{"caption": ...}])` (:212) and the ONE-column
    `json_dump({"answer": ..., "question_id": ...})` (:199) that convert_tsv_to_vqa_json (:227-229) reads."""
    import collections
    rank = get_mpi_rank() if rank is None else rank
    world = get_mpi_size() if world is None else world
    tsv = TSVFile(image_tsv)
    start, end = shard_range(len(tsv), rank, world)
    shard_file = out_tsv if world == 1 else f"{out_tsv}.{rank}.{world}.tsv"    # inference.py:159-164
    questions = TSVFile(question_tsv) if question_tsv else None
    rows: List[list] = []
    pending = collections.deque()                 # (keys, handle) in submission order
    if stats is not None:
        stats.update(images=0, batches=0, questions=0)

    import time

    def drain(limit):
        while len(pending) > limit:
            keys, handle = pending.popleft()
            t0 = time.perf_counter()
            caps = handle.result()
            for key, cap in zip(keys, caps):
                rows.append([key, json_dump([{"caption": cap}])])
            if stats is not None:
                stats["result_s"] = stats.get("result_s", 0.0) + (time.perf_counter() - t0)

    def flush(keys, imgs):
        if stats is not None:
            stats["batches"] += 1
        if submit_captions is None:
            for key, cap in zip(keys, caption_batch(imgs)):
                rows.append([key, json_dump([{"caption": cap}])])
            return
        t0 = time.perf_counter()
        pending.append((keys, submit_captions(imgs)))
        if stats is not None:
            stats["submit_s"] = stats.get("submit_s", 0.0) + (time.perf_counter() - t0)
        drain(in_flight if in_flight > 1 else 0)

    import threading
    lock = threading.Lock()

    def load(j):
        with lock:                      # TSVFile keeps ONE file handle: row reads are serialised, the decoding is not
            row = tsv[start + j]
        raw = base64.b64decode(row[1])
        return row[0], (decode(raw) if decode is not None else raw)

    # VQA: answers by image index (buckets complete out of order; rows are written in input order at the end)
    answers: dict = {}
    q_infos: dict = {}
    buckets: dict = {}                   # image shape -> [image indices, images, question lists]
    vqa_pending = collections.deque()

    def vqa_drain(limit):
        while len(vqa_pending) > limit:
            idxs, handle = vqa_pending.popleft()
            for i_, ans in zip(idxs, handle.result()):
                answers[i_] = ans

    def vqa_flush(shape):
        idxs, imgs, qs = buckets.pop(shape)
        if stats is not None:
            stats["batches"] += 1
        vqa_pending.append((idxs, submit_answers(imgs, qs)))
        vqa_drain(max(0, in_flight - 1))

    if batch_source is not None and questions is None:
        for bkeys, batch in batch_source(start, end):
            if stats is not None:
                stats["images"] += len(bkeys)
            flush(bkeys, batch)
        start = end                                 # nothing left for the per-image loop below
    keys, imgs = [], []
    for j, (key, item) in enumerate(prefetch_ordered(end - start, load, decode_threads if decode is not None else 0,
                                                     window=max(2 * batch_size, 8))):
        i = start + j
        img = transform(item)
        if stats is not None:
            stats["images"] += 1
        if questions is None:
            keys.append(key)
            imgs.append(img)
            if len(keys) == batch_size:
                flush(keys, imgs)
                keys, imgs = [], []
            continue
        qkey, qjson = questions[i][0], questions[i][1]
        assert qkey == key
        q_info = json.loads(qjson)                                            # inference.py:172-199
        q_infos[i] = q_info
        qs = [q["question"] for q in q_info]
        if stats is not None:
            stats["questions"] += len(qs)
        if submit_answers is None:
            answers[i] = answer_questions(img, qs)
            continue
        shape = tuple(img.shape)
        b = buckets.get(shape)
        if b is not None and (len(b[0]) + 1 > batch_size or sum(len(q) for q in b[2]) + len(qs) > (max_questions or batch_size)):
            vqa_flush(shape)
            b = None
        if b is None:
            b = buckets[shape] = [[], [], []]
        b[0].append(i)
        b[1].append(img)
        b[2].append(qs)
    if keys:
        flush(keys, imgs)
    drain(0)
    for shape in list(buckets):
        vqa_flush(shape)
    vqa_drain(0)
    for i in sorted(q_infos):
        for q, ans in zip(q_infos[i], answers[i]):
            rows.append([json_dump({"answer": ans, "question_id": q["question_id"]})])
    tsv_writer(rows, shard_file)
    if world == 1:
        return
    if ensure_process_group(rank, world):
        all_rows = _gather_rows(rows)
        if rank == 0:
            tsv_writer(all_rows, out_tsv)
    elif rank == 0:
        _wait_and_concat_shards(out_tsv, world, poll_s=poll_s)


def pooled_caption_batches(image_tsv: str, start: int, end: int, batch_size: int, procs: int, crop: int, in_flight: int,
                           slot_bytes: int = 2 << 20, stats: Optional[dict] = None):
    """The host side of the captioning task at the engine's rate: rows [start, end) of `image_tsv` as (keys, fp32 batch
    [n, 3, crop, crop] on the device) in input order.  `procs` worker processes decode straight into a shared staging buffer
    (decode_pool.DecodePool), `in_flight + 2` batches of slots deep; a finished batch is uploaded image by image into ONE
    device buffer (only the bytes used) and transformed by ONE launch pair per 24 images (gitmi_preprocess_batch) --
    bit-identical to load_image_by_pil + gpu_image_transform per image."""
    import time
    from .decode_pool import DecodePool
    from .engine import preprocess_batch
    ring = max(2, in_flight) + 2
    n_rows = end - start
    n_batches = (n_rows + batch_size - 1) // batch_size
    t_pool = time.perf_counter()
    pool = DecodePool(image_tsv, procs, slots=ring * batch_size, slot_bytes=slot_bytes)
    host = torch.from_numpy(pool.buffer)
    pinned = False
    try:                                            # page-lock the staging buffer: uploads become asynchronous DMA
        pinned = int(torch.cuda.cudart().cudaHostRegister(host.data_ptr(), host.numel(), 0)) == 0
    except Exception:
        pinned = False
    if stats is not None:
        stats.update(decode_procs=procs, staging_pinned=pinned, staging_mb=host.numel() >> 20,
                     pool_start_s=time.perf_counter() - t_pool, t_pool_started=time.perf_counter())
    tsv = None
    try:
        meta = {}                                   # batch -> {position: (key, H, W)}
        uploaded = {}                               # ring buffer -> event after its last upload
        dispatched = 0

        def rows_of(b):
            return range(b * batch_size, min((b + 1) * batch_size, n_rows))

        for b in range(n_batches):
            while dispatched < n_batches and dispatched < b + ring - 1:
                ev = uploaded.pop(dispatched % ring, None)
                if ev is not None:
                    ev.synchronize()                # the slots are free once the previous occupant has been uploaded
                for r in rows_of(dispatched):
                    pool.submit((dispatched % ring) * batch_size + r % batch_size, start + r)
                meta[dispatched] = {}
                dispatched += 1
            want = len(rows_of(b))
            t_a = time.perf_counter()
            while len(meta[b]) < want:
                slot, row, key, h, w = pool.next_result()
                r = row - start
                if not key:                         # longer than a result record holds
                    if tsv is None:
                        tsv = TSVFile(image_tsv)
                    key = tsv[row][0]
                meta[r // batch_size][r % batch_size] = (key, h, w, slot)
            t_b = time.perf_counter()
            if stats is not None and b == 0:
                stats["first_batch_ready_s"] = t_b - stats["t_pool_started"]      # interpreter start-up of the workers + one batch
                stats["t_first_batch"] = t_b
            items = [meta[b][j] for j in range(want)]
            del meta[b]
            sizes = [abs(h) * abs(w) * 3 for _, h, w, _ in items]
            offs = [0]
            for n in sizes:
                offs.append(offs[-1] + (n + 63) // 64 * 64)
            dev = torch.empty(offs[-1], dtype=torch.uint8, device="cuda")
            desc = []
            for j, (key, h, w, slot) in enumerate(items):
                if h < 0:                           # did not fit its slot: decoded here (rare; GIT_DECODE_SLOT_MB raises the size)
                    if tsv is None:
                        tsv = TSVFile(image_tsv)
                    arr = decode_to_array(base64.b64decode(tsv[start + b * batch_size + j][1]))
                    h, w = arr.shape[0], arr.shape[1]
                    dev[offs[j]: offs[j] + sizes[j]].copy_(torch.from_numpy(arr).reshape(-1), non_blocking=False)
                else:
                    dev[offs[j]: offs[j] + sizes[j]].copy_(host[slot * slot_bytes: slot * slot_bytes + sizes[j]], non_blocking=True)
                desc.append((offs[j], h, w))
            ev = torch.cuda.Event()
            ev.record()
            uploaded[b % ring] = ev
            t_c = time.perf_counter()
            out = preprocess_batch(dev, desc, crop)
            if stats is not None:           # where the parent's time goes: waiting for the workers / uploads / the transform launches
                stats["wait_decode_s"] = stats.get("wait_decode_s", 0.0) + (t_b - t_a)
                stats["upload_s"] = stats.get("upload_s", 0.0) + (t_c - t_b)
                stats["transform_s"] = stats.get("transform_s", 0.0) + (time.perf_counter() - t_c)
            yield [it[0] for it in items], out
    finally:
        if pinned:
            try:
                torch.cuda.synchronize()
                torch.cuda.cudart().cudaHostUnregister(host.data_ptr())
            except Exception:
                pass
        del host
        pool.close()


class _Mapped:
    """handle.result() post-processed (token ids -> strings)"""

    def __init__(self, handle, fn):
        self._h, self._fn = handle, fn

    def result(self):
        return self._fn(self._h.result())


def test_git_inference_single_tsv(image_tsv, model_name, question_tsv, out_tsv, *, checkpoint=None,
                                  batch_size=64, precision="f16", contexts=4, stats=None):
    """inference.py:134-225.  image_tsv rows: key \\t base64(jpeg).  question_tsv (optional) rows:
    key \\t json list of {'question', 'question_id'}.  Writes out_tsv rows
    key \\t [{"caption": ...}]   or the one-column   {"answer": ..., "question_id": ...}.

    precision: "f16" (default since round 6) is the 16-bit build that meets the specification's logit tolerance on every
    weight family and the one bench.py's headline measures; "f32" reproduces the reference's fp32 token ids
    bit for bit; "bf16" is the other 16-bit operand format.
    contexts: requests kept in flight on the device (CaptioningModel.set_pipeline: clones on their own HIP streams, two image
    encoders at a time) while host threads decode the next batches; 1 = the serial loop of the reference.
    Models with test_respect_ratio_max (VQAv2 / TextVQA) batch images of EQUAL resized shape into one engine call
    (gitmi_generate_prefixed takes ragged questions about several images) instead of one call per image."""
    import time
    t_build = time.perf_counter()
    param, from_file = _task_param(model_name, "output")                    # inference.py:135-137
    tokenizer = get_tokenizer()
    torch.cuda.set_device(get_mpi_local_rank())                             # inference.py:152
    is_vqa = bool(question_tsv)
    # sentences per engine call: images when captioning; questions (of up to batch_size images) when answering
    max_batch = max(batch_size, MAX_VQA_QUESTIONS) if is_vqa else batch_size
    model = build_model(model_name, tokenizer, checkpoint, max_batch=max_batch, precision=precision,
                        **({"param": param} if from_file else {}))
    transforms = get_image_transform(param, gpu=True)
    pipelined = contexts > 1 and hasattr(model, "set_pipeline")
    if pipelined:
        model.set_pipeline(contexts)

    def decode_ids(preds) -> List[str]:
        return [tokenizer.decode(pred, skip_special_tokens=True) for pred in preds]

    def as_batch(imgs):
        return imgs.cuda() if isinstance(imgs, torch.Tensor) else torch.stack(list(imgs)).cuda()

    def caption_batch(imgs: Sequence[torch.Tensor]) -> List[str]:
        with torch.no_grad():
            res = model({"image": as_batch(imgs)})
        return decode_ids(res["predictions"].tolist())

    def submit_captions(imgs: Sequence[torch.Tensor]):
        with torch.no_grad():
            h = model.submit({"image": as_batch(imgs)})
        return _Mapped(h, lambda res: decode_ids(res["predictions"].tolist()))

    def answer_questions(img: torch.Tensor, qs: Sequence[str]) -> List[str]:
        out: List[str] = []
        for lo in range(0, len(qs), MAX_VQA_QUESTIONS):
            chunk = [_prefix_ids(tokenizer, q) for q in qs[lo:lo + MAX_VQA_QUESTIONS]]
            with torch.no_grad():
                preds = model.answer(img.unsqueeze(0).cuda(), chunk)
            out += decode_ids(preds)
        return out

    def submit_answers(imgs: Sequence[torch.Tensor], qss: Sequence[Sequence[str]]):
        if sum(len(qs) for qs in qss) > max_batch:          # one image with more questions than an engine call takes
            assert len(qss) == 1
            img0, texts = imgs[0], answer_questions(imgs[0], qss[0])
            return _Mapped(type("Done", (), {"result": lambda self: None})(), lambda _: [texts])
        prefixes, image_of = [], []
        for b, qs in enumerate(qss):
            for q in qs:
                prefixes.append(_prefix_ids(tokenizer, q))
                image_of.append(b)
        counts = [len(qs) for qs in qss]
        with torch.no_grad():
            h = model.submit_answers(torch.stack(list(imgs)).cuda(), prefixes, image_of)

        def split(preds):
            texts, out, lo = decode_ids(preds), [], 0
            for n in counts:
                out.append(texts[lo:lo + n])
                lo += n
            return out
        return _Mapped(h, split)

    # JPEG decoding on a few host threads ahead of the GPU (GIT_DECODE_THREADS, default min(16, cores); 0 = serial as in
    # the reference); the transform itself (upload + resize kernels) stays on this thread and this device
    threads = int(os.environ.get("GIT_DECODE_THREADS", str(min(16, effective_cpus()))))
    can_batch_vqa = hasattr(model, "submit_answers")
    # captioning on the engine: worker PROCESSES decode into a shared staging buffer, one upload + one launch pair per batch
    # (GIT_DECODE_PROCS, default min(32, usable cores) with the cgroup's CPU quota counted; 0 = the thread pool above).  Aspect-preserving models and VQA keep the
    # per-image path (every image has its own output shape).
    procs = int(os.environ.get("GIT_DECODE_PROCS", str(min(32, effective_cpus()))))
    batch_source = None
    if procs > 0 and not is_vqa and "test_respect_ratio_max" not in param and hasattr(model, "engine"):
        crop = int(param.get("test_crop_size", 224))
        slot = int(os.environ.get("GIT_DECODE_SLOT_MB", "2")) << 20
        batch_source = lambda s_, e_: pooled_caption_batches(image_tsv, s_, e_, batch_size, procs, crop,
                                                             contexts if pipelined else 1, slot_bytes=slot, stats=stats)
    t_run = time.perf_counter()
    # decode on the pool all the way to the uint8 array the GPU transform uploads (stand-in models of the CPU tests keep PIL images)
    decode = decode_to_array if hasattr(model, "engine") else load_image_by_pil
    # the host side (uploads, transform kernels, torch ops) gets a stream of its own when requests are kept in flight: work on the
    # DEFAULT stream synchronises with every blocking stream of the process (legacy null-stream semantics), streams a running
    # hipGraph executes on included -- measured: uploads on the default stream serialise with the requests in flight
    import contextlib
    side = torch.cuda.stream(torch.cuda.Stream()) if pipelined else contextlib.nullcontext()
    with side:
        run_tsv_inference(image_tsv, question_tsv, out_tsv, decode=decode, decode_threads=threads,
                          transform=transforms, caption_batch=caption_batch, answer_questions=answer_questions,
                          batch_size=batch_size, submit_captions=submit_captions if pipelined else None,
                          submit_answers=submit_answers if (can_batch_vqa and is_vqa) else None,
                          in_flight=contexts if pipelined else 1, stats=stats, batch_source=batch_source, max_questions=max_batch)
    if stats is not None:
        if "t_first_batch" in stats and stats.get("images", 0) > batch_size:
            # rate once the workers are up: everything after the first batch became ready
            stats["steady_captions_per_s"] = (stats["images"] - batch_size) / (time.perf_counter() - stats["t_first_batch"])
        from .model import Pending
        stats["device_wait_s"] = Pending.wait_s
        Pending.wait_s = 0.0
        stats.update(build_s=t_run - t_build, run_s=time.perf_counter() - t_run, decode_threads=threads,
                     contexts=contexts if pipelined else 1, precision=precision, batch_size=batch_size)
    if hasattr(model, "close"):
        model.close()
