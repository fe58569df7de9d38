"""ctypes binding of libgitmi.so (include/gitmi.h) -- the only way Python reaches the HIP kernels.

PyTorch is used for device memory, streams and host<->device copies only; every FLOP of the
hot path runs inside the shared library.  There is no CPU fallback: if the library is missing
or no gfx950 device is present this module raises, it never silently computes on the host.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Mapping, Optional, Sequence, Tuple

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgitmi.so")                 # bf16 operands (BASELINE's named precision; bench.py: alt_precision)
LIB_PATH_F16 = os.path.join(_HERE, "libgitmi_f16.so")         # the same sources built for fp16 operands (-DGITMI_OPS_F16)
LIB_PATH_EXP = os.path.join(_HERE, "libgitmi_exp.so")         # measurement build of libgitmi.so (-DGITMI_EXPERIMENT): see use_experiment_build

PREC_BF16, PREC_F32 = 0, 1
DTYPE_F32, DTYPE_BF16, DTYPE_F16 = 0, 1, 2
SEARCH_AUTOREGRESSIVE, SEARCH_GENERATOR, SEARCH_TRIE = 0, 1, 2
ACT_NONE, ACT_QUICKGELU, ACT_GELU_ERF = 0, 1, 2

EXPORTED_SYMBOLS = [
    "gitmi_abi_version", "gitmi_last_error", "gitmi_create", "gitmi_destroy", "gitmi_load_tensor",
    "gitmi_finalize_weights", "gitmi_encode_frames", "gitmi_prefill", "gitmi_step_logits",
    "gitmi_generate", "gitmi_search_begin", "gitmi_search_rows", "gitmi_search_advance",
    "gitmi_search_finish", "gitmi_profile_enable", "gitmi_profile_read", "gitmi_set_graph",
    "gitmi_op_gemm", "gitmi_op_layernorm", "gitmi_op_attention", "gitmi_op_dgemm", "gitmi_op_dgemm_res",
    "gitmi_op_vocab_topm", "gitmi_clone", "gitmi_op_attn_decode", "gitmi_preprocess_image",
    "gitmi_set_image_shape", "gitmi_preprocess_image_to", "gitmi_generate_prefixed", "gitmi_set_temporal_embedding", "gitmi_op_kv_repack", "gitmi_set_encode_after", "gitmi_op_sample_rows",
    "gitmi_search_done_count", "gitmi_set_trie", "gitmi_operand_dtype", "gitmi_set_shared_device", "gitmi_preprocess_batch",
    "gitmi_set_ln_fold", "gitmi_op_gemm_ln",
]
# libgitmi_exp.so only (include/gitmi_experiment.h): schedules that measured slower than the default, debug hooks
EXPERIMENT_SYMBOLS = [
    "gitmi_debug_import_stage", "gitmi_debug_head_from", "gitmi_debug_set_gemm_impl", "gitmi_debug_set_dgemm",
]


class GitmiConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "image_size", "patch", "vit_width", "vit_layers", "vit_heads", "dec_hidden", "dec_layers",
        "dec_heads", "dec_ffn", "vocab", "max_pos", "num_frames", "sos", "eos", "precision",
        "max_batch", "max_beams", "max_frames", "max_text_len", "max_image_pixels", "max_image_tokens")]


class GitmiSearch(C.Structure):
    _fields_ = [("kind", C.c_int32), ("beam_size", C.c_int32), ("per_node_beam_size", C.c_int32),
                ("max_steps", C.c_int32), ("length_penalty", C.c_double),
                ("do_sample", C.c_int32), ("top_k", C.c_int32), ("top_p", C.c_double), ("temperature", C.c_double),
                ("seed", C.c_uint64), ("repetition_penalty", C.c_double),
                ("num_keep_best", C.c_int32), ("reserved_", C.c_int32)]


class GitmiProfile(C.Structure):
    _fields_ = [("vit_ms", C.c_float), ("prefill_ms", C.c_float), ("decode_ms", C.c_float),
                ("total_ms", C.c_float), ("gemm_ms", C.c_float), ("gemm_launches", C.c_int32),
                ("gemm_flops", C.c_double), ("vit_gemm_ms", C.c_float), ("vit_gemm_launches", C.c_int32),
                ("vit_gemm_flops", C.c_double), ("decode_step_ms", C.c_float), ("decode_steps", C.c_int32),
                ("decode_step_bytes", C.c_double)]

    def as_dict(self) -> Dict[str, float]:
        return {n: getattr(self, n) for n, _ in self._fields_}


class GitmiError(RuntimeError):
    pass


_libs: Dict[str, C.CDLL] = {}
_experiment = False


def use_experiment_build(on: bool = True) -> None:
    """Measurement harnesses only (bench.py --experiment, tools/): serve "bf16" from libgitmi_exp.so, the same kernels built
    with -DGITMI_EXPERIMENT -- kernel-shape overrides and work-skipping switches read from GITMI_* environment variables
    at gitmi_create.  The product libraries read no environment; nothing in the package turns this on."""
    global _experiment
    _experiment = bool(on)


def load_library(operands: str = "bf16") -> C.CDLL:
    """dlopen libgitmi.so (operands="bf16") or libgitmi_f16.so (operands="f16"); raises (never falls back) when it has
    not been built."""
    if operands == "bf16" and _experiment:
        operands = "exp"
    if operands in _libs:
        return _libs[operands]
    path = {"bf16": LIB_PATH, "f16": LIB_PATH_F16, "exp": LIB_PATH_EXP}[operands]
    if not os.path.exists(path):
        raise GitmiError(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"or `make -C generativeimage2text_amd/csrc`.  There is no CPU fallback.")
    lib = C.CDLL(path)
    vp, i32, i64p, fp = C.c_void_p, C.c_int, C.POINTER(C.c_int64), C.c_void_p
    lib.gitmi_abi_version.restype = C.c_int
    lib.gitmi_last_error.restype = C.c_char_p
    lib.gitmi_create.argtypes = [C.POINTER(GitmiConfig), i32, C.POINTER(vp)]
    lib.gitmi_destroy.argtypes = [vp]
    lib.gitmi_destroy.restype = None
    lib.gitmi_load_tensor.argtypes = [vp, C.c_char_p, vp, i64p, i32, i32]
    lib.gitmi_finalize_weights.argtypes = [vp]
    lib.gitmi_encode_frames.argtypes = [vp, C.POINTER(vp), i32, i32, vp, vp]
    lib.gitmi_prefill.argtypes = [vp, vp]
    lib.gitmi_step_logits.argtypes = [vp, vp, i32, i32, vp, vp]
    lib.gitmi_generate.argtypes = [vp, C.POINTER(vp), i32, i32, vp, i32, C.POINTER(GitmiSearch), vp, vp, vp, vp]
    lib.gitmi_search_begin.argtypes = [vp, C.POINTER(GitmiSearch), i32, vp, i32, i32, vp]
    lib.gitmi_search_rows.argtypes = [vp, vp, C.POINTER(C.c_int), C.POINTER(C.c_int), vp]
    lib.gitmi_search_advance.argtypes = [vp, vp, vp]
    lib.gitmi_search_finish.argtypes = [vp, vp, vp, vp, vp]
    lib.gitmi_search_done_count.argtypes = [vp, C.POINTER(C.c_int), vp]
    lib.gitmi_set_trie.argtypes = [vp, i32, vp, vp, vp]
    lib.gitmi_profile_enable.argtypes = [vp, i32]
    lib.gitmi_profile_read.argtypes = [vp, C.POINTER(GitmiProfile)]
    lib.gitmi_set_graph.argtypes = [vp, i32]
    lib.gitmi_set_temporal_embedding.argtypes = [vp, i32]
    lib.gitmi_set_encode_after.argtypes = [vp, vp]
    lib.gitmi_set_shared_device.argtypes = [vp, i32]
    lib.gitmi_set_ln_fold.argtypes = [vp, i32]
    lib.gitmi_op_gemm.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp]
    lib.gitmi_op_gemm_ln.argtypes = [vp, vp, vp, vp, vp, C.c_float, vp, vp, vp, vp, C.c_float, vp, vp, i32, i32, i32, i32, vp]
    lib.gitmi_op_layernorm.argtypes = [vp, vp, vp, C.c_float, vp, vp, i32, i32, i32, vp]
    lib.gitmi_op_attention.argtypes = [vp, vp, i32, i32, i32, i32, i32, vp]
    lib.gitmi_op_dgemm.argtypes = [vp, vp, vp, vp, vp, i32, C.c_float, vp, i32, i32, i32, i32, i32, i32, vp]
    lib.gitmi_op_dgemm_res.argtypes = [vp, vp, vp, vp, vp, i32, vp, vp, C.c_float, vp, vp, vp, i32, i32, i32, vp]
    lib.gitmi_op_vocab_topm.argtypes = [vp, vp, vp, vp, vp, i32, C.c_float, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp, i32, vp]
    lib.gitmi_generate_prefixed.argtypes = [vp, C.POINTER(vp), i32, i32, vp, i32, C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                            i32, C.POINTER(GitmiSearch), vp, vp, vp, vp, vp]
    lib.gitmi_clone.argtypes = [vp, C.POINTER(vp)]
    lib.gitmi_preprocess_image.argtypes = [vp, i32, i32, i32, vp, C.c_size_t, vp, vp]
    lib.gitmi_preprocess_image_to.argtypes = [vp, i32, i32, i32, i32, vp, C.c_size_t, vp, vp]
    lib.gitmi_preprocess_batch.argtypes = [vp, C.c_size_t, i64p, i32, i32, vp, C.c_size_t, vp, vp]
    lib.gitmi_set_image_shape.argtypes = [vp, i32, i32, vp]
    lib.gitmi_op_attn_decode.argtypes = [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp]
    lib.gitmi_op_kv_repack.argtypes = [vp, vp, vp, i32, i32, i32, vp]
    lib.gitmi_op_sample_rows.argtypes = [vp, i32, i32, C.c_float, i32, C.c_float, i32, C.c_uint64, i32, vp, vp, vp, vp]
    if operands == "exp":
        lib.gitmi_debug_import_stage.argtypes = [vp, vp, i32, vp]
        lib.gitmi_debug_head_from.argtypes = [vp, vp, i32, vp, vp]
        lib.gitmi_debug_set_gemm_impl.argtypes = [i32]
        lib.gitmi_debug_set_dgemm.argtypes = [i32]
    for name in EXPORTED_SYMBOLS + (EXPERIMENT_SYMBOLS if operands == "exp" else []):
        if name not in ("gitmi_last_error", "gitmi_destroy"):
            getattr(lib, name).restype = C.c_int
    if lib.gitmi_abi_version() != 10:
        raise GitmiError("libgitmi.so ABI version mismatch")
    lib.gitmi_operand_dtype.restype = C.c_int
    if lib.gitmi_operand_dtype() != {"bf16": DTYPE_BF16, "f16": DTYPE_F16, "exp": DTYPE_BF16}[operands]:
        raise GitmiError(f"{path} was not built for {operands} operands")
    _libs[operands] = lib
    return lib


def _experiment_only(lib, name: str):
    """An entry point of include/gitmi_experiment.h: present in libgitmi_exp.so only."""
    try:
        return getattr(lib, name)
    except AttributeError:
        raise GitmiError(f"{name} is exported by the measurement build only (libgitmi_exp.so): call "
                         f"generativeimage2text_amd.engine.use_experiment_build() before creating the engine") from None


def _ck(rc: int, lib=None) -> None:
    """Raise GitmiError with the failing library's own message (the message is thread-local PER LIBRARY and is not cleared
    by later successful calls: with both libgitmi.so and libgitmi_f16.so loaded, asking every library would report the other's
    stale text as well)."""
    if rc != 0:
        libs = [lib] if lib is not None else list(_libs.values())
        msgs = [m for m in ((l.gitmi_last_error() or b"").decode("utf-8", "replace") for l in libs) if m]
        raise GitmiError(" | ".join(msgs) or "gitmi call failed")


def _stream() -> int:
    return int(torch.cuda.current_stream().cuda_stream)


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else int(t.data_ptr())


def _torch_dtype_code(t: torch.Tensor) -> int:
    return {torch.float32: DTYPE_F32, torch.bfloat16: DTYPE_BF16, torch.float16: DTYPE_F16}[t.dtype]


class Engine:
    """One GIT engine on one GPU.  Mirrors what `get_git_model(...).cuda()` holds in the reference."""

    def __init__(self, model_cfg, precision: str = "bf16", max_batch: int = 64, max_beams: int = 4,
                 max_frames: int = 1, max_text_len: int = 40, device: Optional[int] = None,
                 max_image_hw: Optional[Tuple[int, int]] = None):
        """max_image_hw: largest (H, W) input the engine must accept when images are not all image_size x image_size
        (MinMaxResizeForTest models); default: the model config's max_image_hw, else the native square."""
        if not torch.cuda.is_available():
            raise GitmiError("no GPU visible: the GIT engine runs on MI355X (gfx950) only, there is no CPU fallback")
        # precision: "f16" (the benchmarked build: fp16 operands) / "bf16" (the same kernels on bf16 operands) / "f32" (exact parity mode)
        self.lib = load_library("f16" if precision in ("f16", "fp16") else "bf16")
        self.device = torch.cuda.current_device() if device is None else int(device)
        self.cfg = model_cfg
        self.precision = precision
        c = GitmiConfig()
        for name in ("image_size", "patch", "vit_width", "vit_layers", "vit_heads", "dec_hidden", "dec_layers",
                     "dec_heads", "dec_ffn", "vocab", "max_pos", "num_frames", "sos", "eos"):
            setattr(c, name, int(getattr(model_cfg, name)))
        c.precision = {"bf16": PREC_BF16, "f16": PREC_BF16, "fp16": PREC_BF16, "f32": PREC_F32, "fp32": PREC_F32}[precision]
        c.max_batch, c.max_beams = int(max_batch), int(max_beams)
        c.max_frames, c.max_text_len = int(max_frames), int(max_text_len)
        if max_image_hw is None:
            max_image_hw = getattr(model_cfg, "max_image_hw", None)
        if max_image_hw is not None:
            mh, mw = int(max_image_hw[0]), int(max_image_hw[1])
            c.max_image_pixels = mh * mw
            c.max_image_tokens = (mh // c.patch) * (mw // c.patch) + 1
        self.c = c
        self.n_tok = (c.image_size // c.patch) ** 2 + 1          # tokens per frame at the CURRENT input resolution
        self._hw = (int(c.image_size), int(c.image_size))
        self._h = C.c_void_p()
        self._ck(self.lib.gitmi_create(C.byref(c), self.device, C.byref(self._h)))
        self._finalized = False
        self._cur_B = 0
        self._cur_F = 0

    # -- lifecycle ---------------------------------------------------------------------------
    def _ck(self, rc: int) -> None:
        _ck(rc, self.lib)

    def clone(self) -> "Engine":
        """A second context sharing this engine's packed weights (own workspaces / KV caches / graph),
        for keeping several batches in flight on different streams.  Keep `self` alive while it is used."""
        other = object.__new__(Engine)
        other.lib, other.device, other.cfg, other.precision = self.lib, self.device, self.cfg, self.precision
        other.c = GitmiConfig.from_buffer_copy(self.c)
        other.n_tok = (self.c.image_size // self.c.patch) ** 2 + 1
        other._hw = (int(self.c.image_size), int(self.c.image_size))
        other._h = C.c_void_p()
        self._ck(self.lib.gitmi_clone(self._h, C.byref(other._h)))
        other._finalized, other._cur_B, other._cur_F = True, 0, 0
        other._parent = self
        return other

    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h.value:
            self.lib.gitmi_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- weights -------------------------------------------------------------------------------
    def load_state_dict(self, state_dict: Mapping[str, torch.Tensor]) -> None:
        """Reference keys; tensors may be fp32/bf16/fp16, any device."""
        for key, t in state_dict.items():
            if key == "image_encoder.proj" or key.endswith("attn_mask"):
                continue
            t = t.detach().to("cpu").contiguous()
            if t.dtype not in (torch.float32, torch.bfloat16, torch.float16):
                t = t.float()
            shape = (C.c_int64 * max(t.dim(), 1))(*t.shape)
            self._ck(self.lib.gitmi_load_tensor(self._h, key.encode(), t.data_ptr(), shape, t.dim(), _torch_dtype_code(t)))
        self._ck(self.lib.gitmi_finalize_weights(self._h))
        self._finalized = True

    # -- phases --------------------------------------------------------------------------------
    def set_image_shape(self, H: int, W: int) -> None:
        """Input resolution of the following calls (CLIP/model.py:243-251: token grid (H//patch) x (W//patch),
        positional table resized on the device).  Called automatically from the frames' shape."""
        if (H, W) != self._hw:
            self._ck(self.lib.gitmi_set_image_shape(self._h, int(H), int(W), _stream()))
            self._hw = (int(H), int(W))
            self.n_tok = (H // self.c.patch) * (W // self.c.patch) + 1

    def _frames_arg(self, frames: Sequence[torch.Tensor]) -> Tuple[C.Array, List[torch.Tensor], int]:
        keep = [f.to(device=f"cuda:{self.device}", dtype=torch.float32).contiguous() for f in frames]
        B, _, H, W = keep[0].shape
        for f in keep:
            assert f.shape == (B, 3, H, W), f"frame shape {tuple(f.shape)}: all frames of a call share one resolution"
        self.set_image_shape(H, W)
        arr = (C.c_void_p * len(keep))(*[f.data_ptr() for f in keep])
        return arr, keep, B

    def encode(self, frames: Sequence[torch.Tensor], return_features: bool = True) -> Optional[torch.Tensor]:
        arr, keep, B = self._frames_arg(frames)
        F = len(keep)
        F_eff = min(F, self.c.num_frames) if self.c.num_frames > 0 else F
        out = None
        if return_features:
            out = torch.empty(B, F_eff * self.n_tok, self.c.vit_width, device=keep[0].device, dtype=torch.float32)
        self._ck(self.lib.gitmi_encode_frames(self._h, arr, F, B, _ptr(out), _stream()))
        self._cur_B, self._cur_F = B, F_eff
        return out

    def prefill(self) -> None:
        self._ck(self.lib.gitmi_prefill(self._h, _stream()))

    def step_logits(self, tokens: torch.Tensor) -> torch.Tensor:
        tokens = tokens.to(device=f"cuda:{self.device}", dtype=torch.int64).contiguous()
        R, t = tokens.shape
        out = torch.empty(R, self.c.vocab, device=tokens.device, dtype=torch.float32)
        self._ck(self.lib.gitmi_step_logits(self._h, tokens.data_ptr(), R, t, out.data_ptr(), _stream()))
        return out

    @staticmethod
    def make_search(kind: str, max_steps: int, beam_size: int, per_node_beam_size: int,
                    length_penalty: float = 1.0, do_sample: bool = False, top_k: int = 0, top_p: float = 1.0,
                    temperature: float = 1.0, seed: int = 0, repetition_penalty: float = 1.0,
                    num_keep_best: int = 1) -> GitmiSearch:
        s = GitmiSearch()
        s.kind = (SEARCH_AUTOREGRESSIVE if kind in ("greedy", "autoregressive") else SEARCH_TRIE if kind == "trie"
                  else SEARCH_GENERATOR)
        s.beam_size, s.per_node_beam_size, s.max_steps = int(beam_size), int(per_node_beam_size), int(max_steps)
        s.length_penalty = float(length_penalty)
        s.do_sample, s.top_k, s.top_p = int(bool(do_sample)), int(top_k or 0), float(1.0 if top_p is None else top_p)
        s.temperature, s.seed = float(temperature), int(seed)
        s.repetition_penalty = float(repetition_penalty)
        s.num_keep_best = int(num_keep_best)
        return s

    @staticmethod
    def _out(n_sent: int, search: GitmiSearch, dev, host: bool = False):
        """Output buffers of a search over n_sent sentences: [n, T] / [n], or [n, num_keep_best, T] / [n, num_keep_best]
        when GeneratorWithBeamSearch keeps more than one hypothesis (decoder.py:1283-1290).  host: page-locked host tensors."""
        nh = max(1, int(search.num_keep_best))
        shape = (n_sent,) if nh == 1 else (n_sent, nh)
        kw = dict(pin_memory=True) if host else dict(device=dev)
        return (torch.empty(*shape, search.max_steps, dtype=torch.int64, **kw),
                torch.empty(*shape, dtype=torch.float32, **kw))

    def generate(self, frames: Sequence[torch.Tensor], search: GitmiSearch,
                 prefix: Optional[torch.Tensor] = None, sync: bool = True, host_out: bool = False):
        """-> (tokens int64 [B, max_steps] incl. start tokens / EOS padded, logprobs fp32 [B], info int32 [4]).
        host_out: the three come back as PAGE-LOCKED HOST tensors, filled by the request itself (valid once the stream has reached
        the end of the call): a server with other requests in flight reads them without enqueueing a read-back."""
        arr, keep, B = self._frames_arg(frames)
        dev = keep[0].device
        tokens, logprobs = self._out(B, search, dev, host_out)
        info = torch.empty(4, dtype=torch.int32, pin_memory=True) if host_out else torch.empty(4, device=dev, dtype=torch.int32)
        P, pfx = 1, None
        if prefix is not None:
            pfx = prefix.to(device=dev, dtype=torch.int64).reshape(-1).contiguous()
            P = int(pfx.numel())
        self._ck(self.lib.gitmi_generate(self._h, arr, len(keep), B, _ptr(pfx), P, C.byref(search), tokens.data_ptr(),
                                    logprobs.data_ptr(), info.data_ptr(), _stream()))
        self._cur_B = B
        if sync:
            torch.cuda.current_stream().synchronize()
            self.check_finite(info)
        return tokens, logprobs, info

    def check_finite(self, info) -> None:
        """info[3] of a finished call (include/gitmi.h): sequences with a non-finite log-prob -- an activation left the range
        of the 16-bit operand format (fp16 tops out at 65504).  Raises instead of handing back garbage ids; callers of
        generate(sync=False) call this once the stream has been synchronised."""
        bad = int(info[3]) if isinstance(info, (list, tuple)) else int(info[3].item())
        if bad:
            raise GitmiError(f"{bad} sequence(s) came back with a non-finite log-probability: an activation overflowed the "
                             f"{self.precision} operand range of this build -- run this checkpoint with precision='bf16' or 'f32'")

    def generate_coalesced(self, requests: Sequence[Sequence[torch.Tensor]], search: GitmiSearch, sync: bool = True):
        """Several requests (each a list of F frame tensors [B_i,3,H,W], same F and resolution) served by ONE engine pass
        over sum(B_i) rows: the decode chain's launches and weight reads are shared by all of them (measured on MI355X:
        two 64-image requests per pass +4.6 % captions/s at twice the batch latency).  Captions are independent
        of their batch neighbours, so every request gets what its own generate() call returns.
        -> ([(tokens [B_i, max_steps], logprobs [B_i]) per request], info)"""
        sizes = [int(r[0].shape[0]) for r in requests]
        F = len(requests[0])
        assert all(len(r) == F for r in requests), "coalesced requests must have the same number of frames"
        if sum(sizes) > self.c.max_batch:
            raise GitmiError(f"{sum(sizes)} coalesced rows exceed max_batch={self.c.max_batch}")
        frames = [torch.cat([r[f] for r in requests], dim=0) for f in range(F)] if len(requests) > 1 else list(requests[0])
        tokens, logprobs, info = self.generate(frames, search, sync=sync)
        out, lo = [], 0
        for n in sizes:
            out.append((tokens[lo:lo + n], logprobs[lo:lo + n]))
            lo += n
        return out, info

    def generate_prefixed(self, frames: Sequence[torch.Tensor], search: GitmiSearch,
                          prefixes: Sequence[Sequence[int]], image_of: Optional[Sequence[int]] = None, sync: bool = True,
                          host_out: bool = False):
        """Batched VQA: sentence q starts from its own prefix `prefixes[q]` (token ids incl. [CLS], any lengths) and
        attends to image `image_of[q]` of the encoded batch (default: sentence q <-> image q).  Every sentence gets
        exactly what a batch-1 reference call with that image and prefix returns (decoder.py:984-1006).
        -> (tokens int64 [Q, max_steps], logprobs fp32 [Q], sent int32 [Q, 2] = (returned length, early), info int32 [4])"""
        arr, keep, B = self._frames_arg(frames)
        dev = keep[0].device
        Q = len(prefixes)
        lens = [len(p) for p in prefixes]
        ld = max(lens)
        table = torch.zeros(Q, ld, dtype=torch.int64)
        for q, p in enumerate(prefixes):
            table[q, :len(p)] = torch.as_tensor(list(p), dtype=torch.int64)
        table = table.to(dev)
        tokens, logprobs = self._out(Q, search, dev, host_out)
        hk = dict(pin_memory=True) if host_out else dict(device=dev)
        sent = torch.empty(Q, 2, dtype=torch.int32, **hk)
        info = torch.empty(4, dtype=torch.int32, **hk)
        lens_c = (C.c_int32 * Q)(*lens)
        img_c = None if image_of is None else (C.c_int32 * Q)(*[int(i) for i in image_of])
        self._ck(self.lib.gitmi_generate_prefixed(self._h, arr, len(keep), B, table.data_ptr(), ld, lens_c, img_c, Q,
                                             C.byref(search), tokens.data_ptr(), logprobs.data_ptr(), sent.data_ptr(),
                                             info.data_ptr(), _stream()))
        self._cur_B = B
        if sync:
            torch.cuda.current_stream().synchronize()
            self.check_finite(info)
        return tokens, logprobs, sent, info

    # -- search seam ---------------------------------------------------------------------------
    def search_begin(self, search: GitmiSearch, start: torch.Tensor, vocab: int) -> None:
        start = start.to("cpu", torch.int64).contiguous()
        B, P = start.shape
        self._ck(self.lib.gitmi_search_begin(self._h, C.byref(search), B, start.data_ptr(), P, vocab, _stream()))
        self._search_k = search.beam_size
        self._search_B = B
        self._search_T = search.max_steps
        self._search_cfg = search

    def search_rows(self) -> torch.Tensor:
        R, t = C.c_int(), C.c_int()
        self._ck(self.lib.gitmi_search_rows(self._h, None, C.byref(R), C.byref(t), _stream()))
        out = torch.empty(R.value, t.value, device=f"cuda:{self.device}", dtype=torch.int64)
        self._ck(self.lib.gitmi_search_rows(self._h, out.data_ptr(), C.byref(R), C.byref(t), _stream()))
        return out

    def search_advance(self, logits: torch.Tensor) -> None:
        logits = logits.to(device=f"cuda:{self.device}", dtype=torch.float32).contiguous()
        self._keep_logits = logits
        self._ck(self.lib.gitmi_search_advance(self._h, logits.data_ptr(), _stream()))

    def set_trie(self, child_off, child_tok, child_node) -> None:
        """Token trie of the "trie" search kind as CSR int32 arrays (include/gitmi.h gitmi_set_trie); None removes it."""
        if child_off is None:
            self._ck(self.lib.gitmi_set_trie(self._h, 0, None, None, None))
            return
        off = torch.as_tensor(child_off, dtype=torch.int32).cpu().contiguous()
        tok = torch.as_tensor(child_tok, dtype=torch.int32).cpu().contiguous()
        node = torch.as_tensor(child_node, dtype=torch.int32).cpu().contiguous()
        assert off.numel() >= 2 and tok.numel() == node.numel() == int(off[-1])
        self._ck(self.lib.gitmi_set_trie(self._h, int(off.numel()) - 1, off.data_ptr(), tok.data_ptr() if tok.numel() else None,
                                    node.data_ptr() if node.numel() else None))

    # -- error attribution hooks (tools/error_attribution.py) ---------------------------------------
    def debug_import_stage(self, src: "Engine", stage: int) -> None:
        """Take the image features (stage 1) or features + image K/V of every decoder layer (stage 2) from `src`, a
        context of the same model in the other precision; step_logits() then continues from there."""
        self._ck(_experiment_only(self.lib, "gitmi_debug_import_stage")(self._h, src._h, int(stage), _stream()))
        self._cur_B = src._cur_B

    def debug_head_from(self, src: "Engine", R: int) -> torch.Tensor:
        """This (bf16) context's fused vocabulary head on the last hidden state of src's (fp32) latest step_logits."""
        out = torch.empty(R, self.c.vocab, device=f"cuda:{self.device}", dtype=torch.float32)
        self._ck(_experiment_only(self.lib, "gitmi_debug_head_from")(self._h, src._h, int(R), out.data_ptr(), _stream()))
        torch.cuda.current_stream().synchronize()
        return out

    def search_done_count(self) -> int:
        """Sentences of the running search that need no further step (synchronises the stream)."""
        n = C.c_int()
        self._ck(self.lib.gitmi_search_done_count(self._h, C.byref(n), _stream()))
        return int(n.value)

    def search_finish(self):
        dev = f"cuda:{self.device}"
        tokens, logprobs = self._out(self._search_B, self._search_cfg, dev)
        info = torch.empty(4, device=dev, dtype=torch.int32)
        self._ck(self.lib.gitmi_search_finish(self._h, tokens.data_ptr(), logprobs.data_ptr(), info.data_ptr(), _stream()))
        torch.cuda.current_stream().synchronize()
        return tokens, logprobs, info

    # -- profiling -----------------------------------------------------------------------------
    def profile_enable(self, on) -> None:
        """False/0 off; True/1 eager launches with per-launch HIP events; 2 graph replays split into an
        (encode + prefill) graph and a decode graph with events between them."""
        self._ck(self.lib.gitmi_profile_enable(self._h, int(on)))

    def profile_read(self) -> Dict[str, float]:
        p = GitmiProfile()
        self._ck(self.lib.gitmi_profile_read(self._h, C.byref(p)))
        return p.as_dict()

    def set_shared_device(self, on: bool = True) -> None:
        """Serving policy: other contexts run beside this one (kernel shapes by whole-device cost; bit-identical results).
        Clones made afterwards inherit it."""
        self._ck(self.lib.gitmi_set_shared_device(self._h, 1 if on else 0))

    def set_encode_after(self, other: Optional["Engine"]) -> None:
        """Serving schedule: this context's image encoder starts only after `other`'s (most recently submitted) has
        finished; chain contexts in a ring in submission order (one encoder in flight, decode chains fill in)."""
        self._ck(self.lib.gitmi_set_encode_after(self._h, other._h if other is not None else None))

    def set_temporal_embedding(self, on: bool) -> None:
        """on (default): frames come as a list -> frame i gets img_temperal_embedding[i]; off: a bare image tensor
        (decoder.py:845-857 adds the embedding only in the list branch)."""
        self._ck(self.lib.gitmi_set_temporal_embedding(self._h, 1 if on else 0))

    def set_ln_fold(self, on: bool) -> None:
        """fp16-operand library only: fold the encoder's / prefill's LayerNorms into the GEMMs either side of them (default
        there) or run one LayerNorm launch per module.  Raises if `on` is asked of an engine that cannot fold."""
        self._ck(self.lib.gitmi_set_ln_fold(self._h, 1 if on else 0))

    def set_graph(self, on: bool) -> None:
        self._ck(self.lib.gitmi_set_graph(self._h, 1 if on else 0))


# ---- single-kernel entry points (unit parity tests) -----------------------------------------------
def op_gemm(A: torch.Tensor, W: torch.Tensor, bias: Optional[torch.Tensor] = None,
            residual: Optional[torch.Tensor] = None, act: int = ACT_NONE,
            out_dtype: torch.dtype = torch.float32) -> torch.Tensor:
    lib = load_library()
    assert A.is_cuda and W.is_cuda and A.dtype == W.dtype and A.is_contiguous() and W.is_contiguous()
    M, K = A.shape
    N = W.shape[0]
    out = torch.empty(M, N, device=A.device, dtype=out_dtype)
    _ck(lib.gitmi_op_gemm(A.data_ptr(), W.data_ptr(), _ptr(bias), _ptr(residual), out.data_ptr(), M, N, K, K, N,
                          _torch_dtype_code(A), _torch_dtype_code(out), act, _stream()), lib)
    return out


def op_gemm_ln(A: torch.Tensor, W: torch.Tensor, bias: Optional[torch.Tensor], *, colsum: Optional[torch.Tensor] = None,
               ln_part: Optional[torch.Tensor] = None, ln_eps: float = 1e-5, residual: Optional[torch.Tensor] = None,
               res_part: Optional[torch.Tensor] = None, res_gamma: Optional[torch.Tensor] = None,
               res_beta: Optional[torch.Tensor] = None, res_eps: float = 1e-12, want_part: bool = False, act: int = ACT_NONE):
    """The folded-LayerNorm forms of the large-M GEMM (fp16-operand library): consumer (ln_part given) or producer; returns C
    (fp16 [M, N]) and, for a producer with want_part, the row partials float [M, 4, 2]."""
    lib = load_library("f16")
    assert A.dtype == torch.float16 and W.dtype == torch.float16 and A.is_contiguous() and W.is_contiguous()
    M, K = A.shape
    N = W.shape[0]
    out = torch.empty(M, N, device=A.device, dtype=torch.float16)
    part = torch.zeros(M, 4, 2, device=A.device, dtype=torch.float32) if want_part else None
    _ck(lib.gitmi_op_gemm_ln(A.data_ptr(), W.data_ptr(), _ptr(bias), _ptr(colsum), _ptr(ln_part), ln_eps, _ptr(residual),
                             _ptr(res_part), _ptr(res_gamma), _ptr(res_beta), res_eps, out.data_ptr(), _ptr(part), M, N, K, act,
                             _stream()), lib)
    return (out, part) if want_part else out


def op_layernorm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float,
                 out_dtype: torch.dtype = torch.float32) -> torch.Tensor:
    lib = load_library()
    rows, D = x.shape
    out = torch.empty(rows, D, device=x.device, dtype=out_dtype)
    _ck(lib.gitmi_op_layernorm(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), eps, out.data_ptr(), None, rows, D,
                               _torch_dtype_code(out), _stream()), lib)
    return out


def op_attention(qkv: torch.Tensor, B: int, N: int, H: int, impl: int) -> torch.Tensor:
    lib = load_library()
    assert qkv.is_cuda and qkv.is_contiguous() and qkv.shape == (B * N, 3 * H * 64)
    out = torch.empty(B * N, H * 64, device=qkv.device, dtype=qkv.dtype)
    _ck(lib.gitmi_op_attention(qkv.data_ptr(), out.data_ptr(), B, N, H, _torch_dtype_code(qkv), impl, _stream()), lib)
    return out


def strip_stats(x: torch.Tensor) -> torch.Tensor:
    """Row partials of a [M, N] fp32 tensor per 16-column strip, in the layout the decode-chain kernels exchange:
    [N/16][M][2] = (sum, sum of squares)."""
    M, N = x.shape
    xs = x.float().reshape(M, N // 16, 16)
    return torch.stack([xs.sum(-1), (xs * xs).sum(-1)], dim=-1).permute(1, 0, 2).contiguous()


def to_frag(x: torch.Tensor, row_multiple: int = 16) -> torch.Tensor:
    """Row-major bf16 [R, K] -> the fragment-major operand layout of the decode chain (include/gitmi.h): 16-row x
    32-k tiles in MFMA operand order, rows zero-padded to `row_multiple`.  ACTIVATION operands of the decode-chain kernels
    must be padded to 64 rows: the wide GEMMs and the vocabulary head load four 16-row tiles at a time whatever M is
    (rows past M are loaded, never stored)."""
    R, K = x.shape
    Rp = (R + row_multiple - 1) // row_multiple * row_multiple
    xp = torch.zeros(Rp, K, dtype=x.dtype, device=x.device)
    xp[:R] = x
    return xp.reshape(Rp // 16, 16, K // 32, 4, 8).permute(0, 2, 3, 1, 4).contiguous().reshape(Rp, K)


def from_frag(xf: torch.Tensor, rows: int) -> torch.Tensor:
    Rp, K = xf.shape
    return xf.reshape(Rp // 16, K // 32, 4, 16, 8).permute(0, 3, 1, 2, 4).reshape(Rp, K)[:rows].contiguous()


def _pad_vec(v: Optional[torch.Tensor], n: int) -> Optional[torch.Tensor]:
    if v is None:
        return None
    out = torch.zeros(n, dtype=v.dtype, device=v.device)
    out[:v.numel()] = v
    return out


def op_dgemm(A: torch.Tensor, W: torch.Tensor, bias: torch.Tensor, colsum: Optional[torch.Tensor] = None,
             stats: Optional[torch.Tensor] = None, eps: float = 1e-12, act: int = ACT_NONE,
             frag_out: bool = False, packed: bool = False, strips_per_wg: int = 0) -> torch.Tensor:
    """Decode-chain GEMM, QKV / FFN1 form (kernels_dgemm.hip): bf16 A [M,K], W [N,K] -> bf16 [M,N].
    With `stats` ([K/16][M][2] strip partials of the raw rows behind A) the LayerNorm in front of the GEMM is folded:
    out = rstd * (A W^T - mean * colsum) + bias.  Operands are given row-major and packed here (packed=True: A, W are
    already fragment-major and M, N are taken from bias / stats)."""
    lib = load_library()
    if packed:
        Af, Wf, M, N, K = A, W, int(A.shape[0]), int(bias.numel()), int(A.shape[1])
        if stats is not None:
            M = int(stats.shape[1])
    else:
        M, K = A.shape
        N = W.shape[0]
        Af, Wf = to_frag(A, 64), to_frag(W)
    out = torch.empty((M + 15) // 16 * 16 if frag_out else M, N, device=A.device, dtype=torch.bfloat16)
    strips = 0 if stats is None else int(stats.shape[0])
    _ck(lib.gitmi_op_dgemm(Af.data_ptr(), Wf.data_ptr(), bias.data_ptr(), _ptr(colsum), _ptr(stats), strips, eps,
                           out.data_ptr(), 1 if frag_out else 0, M, N, K, act, int(strips_per_wg), _stream()), lib)
    return from_frag(out, M) if (frag_out and not packed) else out


def op_dgemm_res(A: torch.Tensor, W: torch.Tensor, bias: torch.Tensor, res_x: torch.Tensor,
                 res_stats: Optional[torch.Tensor] = None, res_gamma: Optional[torch.Tensor] = None,
                 res_beta: Optional[torch.Tensor] = None, res_eps: float = 1e-12, packed: bool = False):
    """Decode-chain GEMM, N = hidden form: x = A W^T + bias + residual, where the residual is `res_x` itself or
    LayerNorm(res_x) rebuilt from its strip partials.  -> (x fp32 [M,N], bf16 copy, strip partials of x)."""
    lib = load_library()
    M, N = res_x.shape
    K = A.shape[1]
    Af, Wf = (A, W) if packed else (to_frag(A, 64), to_frag(W))
    x = torch.empty(M, N, device=A.device, dtype=torch.float32)
    xb = torch.empty((M + 15) // 16 * 16, N, device=A.device, dtype=torch.bfloat16)
    st = torch.empty(N // 16, M, 2, device=A.device, dtype=torch.float32)
    strips = 0 if res_stats is None else int(res_stats.shape[0])
    _ck(lib.gitmi_op_dgemm_res(Af.data_ptr(), Wf.data_ptr(), bias.data_ptr(), res_x.data_ptr(), _ptr(res_stats), strips,
                               _ptr(res_gamma), _ptr(res_beta), res_eps, x.data_ptr(), xb.data_ptr(), st.data_ptr(),
                               M, N, K, _stream()), lib)
    return x, (xb if packed else from_frag(xb, M)), st


def op_vocab_topm(A: torch.Tensor, W: torch.Tensor, bias: torch.Tensor, mtop: int, cols_per_wg: int = 128,
                  colsum: Optional[torch.Tensor] = None, stats: Optional[torch.Tensor] = None, eps: float = 1e-12,
                  suppress_tok: Optional[torch.Tensor] = None, want_logits: bool = False, packed: bool = False,
                  rows: Optional[int] = None, V: Optional[int] = None, max_wgs: int = 0):
    """Vocabulary head with the fused running top-M / log-sum-exp: -> (part_val [M, nparts, slots], part_idx,
    part_lse [M, nparts, 2] = (max, sum exp), logits [M, V] or None).  packed=True: A / W fragment-major (W rows and
    bias / colsum padded to a multiple of cols_per_wg), `rows` = M, `V` = vocabulary size."""
    lib = load_library()
    K = A.shape[1]
    V = int(bias.numel()) if V is None else int(V)
    if packed:
        Af, Wf, M, bp, cp = A, W, int(rows), bias, colsum
    else:
        M = A.shape[0]
        Vp = (V + cols_per_wg - 1) // cols_per_wg * cols_per_wg
        Af, Wf, bp, cp = to_frag(A, 64), to_frag(W, cols_per_wg), _pad_vec(bias, Vp), _pad_vec(colsum, Vp)
    nparts = (V + cols_per_wg - 1) // cols_per_wg
    slots = 1 if mtop <= 1 else 2 if mtop <= 2 else 4 if mtop <= 4 else 8 if mtop <= 8 else 16
    pv = torch.empty(M, nparts, slots, device=A.device, dtype=torch.float32)
    pi = torch.empty(M, nparts, slots, device=A.device, dtype=torch.int32)
    pl = torch.empty(M, nparts, 2, device=A.device, dtype=torch.float32)
    lg = torch.empty(M, V, device=A.device, dtype=torch.float32) if want_logits else None
    strips = 0 if stats is None else int(stats.shape[0])
    _ck(lib.gitmi_op_vocab_topm(Af.data_ptr(), Wf.data_ptr(), bp.data_ptr(), _ptr(cp), _ptr(stats), strips, eps,
                                M, V, K, cols_per_wg, mtop, _ptr(suppress_tok), pv.data_ptr(), pi.data_ptr(),
                                pl.data_ptr(), _ptr(lg), int(max_wgs), _stream()), lib)
    return pv, pi, pl, lg


def set_gemm_impl(impl: int) -> None:
    """Measurement build only (use_experiment_build()): -1 auto, 0 register-staged tile kernel only, 9 the LDS-DMA kernel
    wherever it can run; 9 | (bits << 8): 64 / 128 force its 192- / 256-row tile (tools/gemm_bench.py lists the others)."""
    lib = load_library()
    _ck(_experiment_only(lib, "gitmi_debug_set_gemm_impl")(int(impl)), lib)


def op_attn_decode(qkv, img_k, img_v, txt_k, txt_v, kv_src, B, H, N_img, T_max, pos, beams, dbg=0):
    """qkv [R,3d]; img_k/img_v [B,H,N_img,64]; txt_k/txt_v [R,T_max,d] (position pos gets appended); kv_src int32 [R,T_max]."""
    lib = load_library()
    R, d = B * beams, H * 64
    out = torch.empty(R, d, device=qkv.device, dtype=qkv.dtype)
    if qkv.dtype == torch.bfloat16 and img_k.dim() == 4:
        # head-major [B,H,N,64] given: build the prefill-layout rows and repack into the matrix-core layouts
        img_k, img_v = kv_repack(img_k, img_v)
    _ck(lib.gitmi_op_attn_decode(qkv.data_ptr(), img_k.data_ptr(), img_v.data_ptr(), txt_k.data_ptr(), txt_v.data_ptr(),
                                 kv_src.data_ptr(), out.data_ptr(), B, H, N_img, T_max, pos, beams, _torch_dtype_code(qkv),
                                 dbg, _stream()), lib)
    return out


def op_sample_rows(logits: torch.Tensor, temperature: float = 1.0, top_k: int = 0, top_p: float = 1.0, ndraw: int = 2,
                   seed: int = 0, step: int = 1, want_filtered: bool = True):
    """One step of the sampling branch (decoder.py:1146-1166) on fp32 logits [R, V]:
    -> (filtered logits [R, V] or None, draw tokens int32 [R, ndraw], their log-probs fp32 [R, ndraw])."""
    lib = load_library()
    logits = logits.float().contiguous()
    R, V = logits.shape
    lp = torch.empty(R, ndraw, device=logits.device, dtype=torch.float32)
    tok = torch.empty(R, ndraw, device=logits.device, dtype=torch.int32)
    filt = torch.empty(R, V, device=logits.device, dtype=torch.float32) if want_filtered else None
    _ck(lib.gitmi_op_sample_rows(logits.data_ptr(), R, V, float(temperature), int(top_k), float(top_p), int(ndraw), int(seed),
                                 int(step), lp.data_ptr(), tok.data_ptr(), _ptr(filt), _stream()), lib)
    return filt, tok, lp


def kv_repack(img_k: torch.Tensor, img_v: torch.Tensor):
    """Head-major bf16 image K/V [B,H,N,64] -> the decode layouts of kernels_attn_decode.hip (flat [B*H*Npad*64] each)."""
    lib = load_library()
    B, H, N, _ = img_k.shape
    d = H * 64
    rows = torch.zeros(B * N, 3 * d, device=img_k.device, dtype=torch.bfloat16)
    rows[:, d:2 * d] = img_k.permute(0, 2, 1, 3).reshape(B * N, d)
    rows[:, 2 * d:] = img_v.permute(0, 2, 1, 3).reshape(B * N, d)
    Np = (N + 31) // 32 * 32
    kf = torch.empty(B * H * Np * 64, device=img_k.device, dtype=torch.bfloat16)
    vt = torch.empty(B * H * Np * 64, device=img_k.device, dtype=torch.bfloat16)
    _ck(lib.gitmi_op_kv_repack(rows.data_ptr(), kf.data_ptr(), vt.data_ptr(), B, N, H, _stream()), lib)
    return kf, vt


def preprocess_image(rgb_hwc: torch.Tensor, crop: int = 224) -> torch.Tensor:
    """uint8 [H,W,3] device tensor (a decoded RGB image) -> fp32 [3,crop,crop], bit-exact with the reference's
    PIL/torchvision transform (Resize(crop, BICUBIC) -> CenterCrop -> ToTensor -> Normalize)."""
    lib = load_library()
    assert rgb_hwc.is_cuda and rgb_hwc.dtype == torch.uint8 and rgb_hwc.dim() == 3 and rgb_hwc.shape[2] == 3
    rgb_hwc = rgb_hwc.contiguous()
    H, W = int(rgb_hwc.shape[0]), int(rgb_hwc.shape[1])
    nw = crop if W <= H else int(crop * W / H)
    tmp = torch.empty(H * nw * 3, dtype=torch.uint8, device=rgb_hwc.device)
    out = torch.empty(3, crop, crop, dtype=torch.float32, device=rgb_hwc.device)
    _ck(lib.gitmi_preprocess_image(rgb_hwc.data_ptr(), H, W, crop, tmp.data_ptr(), tmp.numel(), out.data_ptr(), _stream()), lib)
    return out


def preprocess_image_to(rgb_hwc: torch.Tensor, out_h: int, out_w: int) -> torch.Tensor:
    """uint8 [H,W,3] device tensor -> fp32 [3,out_h,out_w]: Pillow-exact bicubic resize to (out_h, out_w), ToTensor,
    Normalize -- the MinMaxResizeForTest branch of the reference transform (inference.py:113-116)."""
    lib = load_library()
    assert rgb_hwc.is_cuda and rgb_hwc.dtype == torch.uint8 and rgb_hwc.dim() == 3 and rgb_hwc.shape[2] == 3
    rgb_hwc = rgb_hwc.contiguous()
    H, W = int(rgb_hwc.shape[0]), int(rgb_hwc.shape[1])
    tmp = torch.empty(H * out_w * 3, dtype=torch.uint8, device=rgb_hwc.device)
    out = torch.empty(3, out_h, out_w, dtype=torch.float32, device=rgb_hwc.device)
    _ck(lib.gitmi_preprocess_image_to(rgb_hwc.data_ptr(), H, W, int(out_h), int(out_w), tmp.data_ptr(), tmp.numel(),
                                      out.data_ptr(), _stream()), lib)
    return out


def preprocess_batch(staging: torch.Tensor, desc: Sequence[Tuple[int, int, int]], crop: int = 224) -> torch.Tensor:
    """A batch of decoded images in ONE uint8 device buffer -> fp32 [n, 3, crop, crop] (gitmi_preprocess_batch: the reference's
    Resize(BICUBIC) -> CenterCrop -> ToTensor -> Normalize, bit-exact with Pillow, one launch pair per 24 images).
    desc: (byte offset, H, W) of every image in `staging`."""
    lib = load_library()
    assert staging.is_cuda and staging.dtype == torch.uint8 and staging.is_contiguous()
    n = len(desc)
    table = (C.c_int64 * (3 * n))(*[int(v) for d in desc for v in d])
    need = 0
    for _, H, W in desc:
        nw = crop if W <= H else int(crop * W / H)
        if nw != W:
            need += (H * nw * 3 + 63) // 64 * 64
    tmp = torch.empty(max(need, 64), dtype=torch.uint8, device=staging.device)
    out = torch.empty(n, 3, crop, crop, dtype=torch.float32, device=staging.device)
    _ck(lib.gitmi_preprocess_batch(staging.data_ptr(), staging.numel(), table, n, int(crop), tmp.data_ptr(), tmp.numel(),
                                   out.data_ptr(), _stream()), lib)
    return out
