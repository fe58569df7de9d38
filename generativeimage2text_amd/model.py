"""Host-side mirror of the reference's model interface for the hot path.

Same names, argument meaning and return conventions as
  generativeimage2text/model.py:9-61              get_git_model
  generativeimage2text/layers/decoder.py:774-1054 CaptioningModel (forward / infer)
  generativeimage2text/layers/decoder.py:208-222  AutoRegressiveBeamSearch  (constructor arguments)
  generativeimage2text/layers/decoder.py:1056-1081 GeneratorWithBeamSearch (constructor arguments)
but all arithmetic happens in libgitmi.so (HIP, gfx950).  The two search classes hold the reference's constructor
arguments; the search itself runs on the device (csrc/kernels_search.hip), inside gitmi_generate for model(batch) and
behind `decoder.search(start_predictions, step)` for callers that bring their own `step` (decoder.py:224-231, 1083-1092).
"""
from __future__ import annotations

import logging
from typing import Dict, Mapping, Optional, Sequence, Union

import torch

from .configs import GitModelConfig, config_from_param
from .engine import Engine


# ---- decoder.search(start_predictions, step): the reference's search seam as a method ----------------------------
# The search itself runs on the device (csrc/kernels_search.hip) behind gitmi_search_begin / rows / advance / finish.
# Those entry points live on an engine context, so a search with a caller-supplied `step` gets a SMALL context of its own
# (a minimal model geometry with placeholder weights that are never used: only the search state and kernels are).
_SEARCH_ENGINES: Dict[tuple, Engine] = {}


def _search_engine(eos: int, sos: int, B: int, beams: int, T: int, factory=None) -> Engine:
    """A context that only hosts searches: capacity (B sentences, `beams` beams, T positions).  `factory` (tests) builds
    the context instead of Engine."""
    key = (int(eos), int(B), int(beams), int(T))
    eng = _SEARCH_ENGINES.get(key)
    if eng is None:
        if factory is not None:
            eng = factory(eos, B, beams, T)
        else:
            from .synthetic import random_state_dict
            cfg = GitModelConfig(name="search-only", image_size=64, patch=16, vit_width=128, vit_layers=2, vit_heads=2,
                                 dec_hidden=128, dec_layers=2, dec_heads=2, dec_ffn=512, vocab=1000, max_pos=max(64, int(T)),
                                 sos=int(sos), eos=int(eos))
            eng = Engine(cfg, precision="f32", max_batch=int(B), max_beams=int(beams), max_frames=1, max_text_len=int(T))
            eng.load_state_dict(random_state_dict(cfg, seed=0))
        if len(_SEARCH_ENGINES) >= 4:                        # a few capacities at most stay alive
            _SEARCH_ENGINES.pop(next(iter(_SEARCH_ENGINES))).close()
        _SEARCH_ENGINES[key] = eng
    return eng


def _begin_and_run(decoder, start_predictions, step, search_struct, first_step_rows_per_sentence, stop_when_all_eos, fmt,
                   engine_factory=None):
    start = start_predictions.detach().to("cpu", torch.int64)
    assert start.dim() == 2, "start_predictions is [batch, prefix_length]"
    B, P = start.shape
    T, k = int(decoder.max_steps), int(decoder.beam_size)
    if P > T:
        raise ValueError(f"prefix of {P} tokens exceeds max_steps={T}")
    dev = start_predictions.device
    eng = _search_engine(decoder._eos_index, int(start[0, 0]), B, k, T, engine_factory)
    if getattr(decoder, "kind", "") == "trie":
        eng.set_trie(*decoder.trie.csr())
    # the vocabulary size is only known from the first logits: the first `step` call is made on the start rows as the
    # reference makes it (one row per sentence for AutoRegressiveBeamSearch, decoder.py:259; B*k rows for the generator,
    # decoder.py:1101-1102), then the device search begins and receives those logits as its first advance
    if P < T:
        first_rows = start if (first_step_rows_per_sentence or k == 1) else start.repeat_interleave(k, dim=0)
        logits = step(first_rows.to(dev))
        if logits.shape[0] != B * k:
            logits = logits.repeat_interleave(k, dim=0)
        eng.search_begin(search_struct, start, int(logits.shape[-1]))
        eng.search_advance(logits)
    else:
        eng.search_begin(search_struct, start, 2)
    while True:
        rows = eng.search_rows()                                        # int64 [B*k, t], rows of a sentence contiguous
        t = int(rows.shape[1])
        if t >= T:
            break
        if stop_when_all_eos and bool((rows[:, -1] == decoder._eos_index).all()):
            break                                                       # decoder.py:319-320
        eng.search_advance(step(rows.to(dev)))
        if not stop_when_all_eos and eng.search_done_count() >= B:
            break                                                       # decoder.py:1251: every sentence is done
    tokens, logprobs, info = eng.search_finish()
    seq_len, early = int(info[0]), int(info[1])
    tokens, logprobs = tokens.to(dev), logprobs.to(dev)
    if fmt == "autoregressive":
        if early:                                                       # decoder.py:279-291
            return tokens[:, P:P + 1], logprobs[:, None]
        return tokens[:, :seq_len], logprobs
    return tokens, (logprobs[:, None] if logprobs.dim() == 1 else logprobs)      # [B, num_keep_best]


class AutoRegressiveBeamSearch:
    """Constructor-compatible with the reference class (decoder.py:209-222)."""

    def __init__(self, eos_index: int, max_steps: int = 50, beam_size: int = 5,
                 per_node_beam_size: int = 2, fix_missing_prefix: bool = False) -> None:
        assert fix_missing_prefix, "should always true"          # decoder.py:222
        self._eos_index = eos_index
        self.max_steps = max_steps
        self.beam_size = beam_size
        self.per_node_beam_size = per_node_beam_size or beam_size
        self.kind = "autoregressive"
        self.length_penalty = 1.0

    def search(self, start_predictions: torch.Tensor, step, only_return_best: bool = True, do_sample: bool = False,
               top_k: int = 0, top_p=None, num_return_sequences: int = 1, temperature: float = 1, _engine_factory=None):
        """decoder.py:224-440 with a caller-supplied `step(rows int64 [R, t]) -> logits fp32 [R, V]`:
        -> (predictions int64 [B, length <= max_steps] incl. the start tokens, logprobs fp32 [B]); when every sentence
        ends at its first step with beam_size == 1: ([B, 1], [B, 1]) (decoder.py:279-291).  The sampling branch and
        only_return_best=False are used by SCST training only and are not implemented."""
        if do_sample or not only_return_best or num_return_sequences != 1 or temperature != 1:
            raise NotImplementedError("AutoRegressiveBeamSearch.search: only the inference form "
                                      "(only_return_best=True, do_sample=False) is implemented")
        s = Engine.make_search("autoregressive", self.max_steps, self.beam_size, self.per_node_beam_size)
        return _begin_and_run(self, start_predictions, step, s, first_step_rows_per_sentence=True, stop_when_all_eos=True,
                              fmt="autoregressive", engine_factory=_engine_factory)


class TokenTrie:
    """Same interface as the reference's TokenTrie (trie_decoder.py:224-257): construct / insert / get_valid / reset /
    get_curr_valid / move.  The search itself walks a CSR copy of it on the device (`csr()` -> gitmi_set_trie)."""

    def __init__(self):
        self._children = [{}]                  # node -> {token: child node}; node 0 is the root
        self.curr = 0

    @classmethod
    def construct(cls, all_tokens):
        ret = cls()
        for ts in all_tokens:
            ret.insert(ts)
        return ret

    def insert(self, tokens):
        cur = 0
        for t in tokens:
            nxt = self._children[cur].get(int(t))
            if nxt is None:
                nxt = len(self._children)
                self._children.append({})
                self._children[cur][int(t)] = nxt
            cur = nxt

    def get_valid(self, tokens):
        cur = 0
        for t in tokens:
            cur = self._children[cur].get(int(t))
            if cur is None:
                return []
        return list(self._children[cur].keys())

    def reset(self):
        self.curr = 0

    def get_curr_valid(self):
        return list(self._children[self.curr].keys())

    def move(self, t):
        assert int(t) in self._children[self.curr]
        self.curr = self._children[self.curr][int(t)]

    def csr(self):
        off, tok, node = [0], [], []
        for ch in self._children:
            tok.extend(ch.keys())
            node.extend(ch.values())
            off.append(len(tok))
        return off, tok, node


def get_output_vocab_tokens(tokenizer, texts):
    """trie_decoder.py:18-25: every allowed answer as its token ids + [SEP]."""
    return [tokenizer(a, padding="do_not_pad", add_special_tokens=False)["input_ids"] + [tokenizer.sep_token_id] for a in texts]


def get_trie(tokenizer, texts=None, fname="./aux_data/imagenet/imagenet_unique_readable_names.txt"):
    """trie_decoder.py:6-16: the trie over the allowed output texts (default: the ImageNet names file of the reference)."""
    if texts is None:
        with open(fname, "r") as fp:
            texts = list(fp)
    return TokenTrie.construct(get_output_vocab_tokens(tokenizer, texts))


class TrieAutoRegressiveBeamSearch:
    """Constructor-compatible with the reference class (trie_decoder.py:27-39): greedy decoding (beam_size == 1) restricted
    to the token sequences of `trie`.  Every sentence of a batch walks its own trie cursor, i.e. gets what its own batch-1
    reference call returns (the reference's single cursor follows row 0)."""

    def __init__(self, eos_index: int, max_steps: int = 50, beam_size: int = 5, trie=None) -> None:
        self._eos_index = eos_index
        self.max_steps = max_steps
        assert beam_size == 1                                       # trie_decoder.py:37
        self.beam_size = beam_size
        self.per_node_beam_size = 1
        self.trie = trie
        self.kind = "trie"
        self.length_penalty = 1.0

    def search(self, start_predictions: torch.Tensor, step, only_return_best: bool = True, do_sample: bool = False,
               top_k: int = 0, top_p=None, num_return_sequences: int = 1, temperature: float = 1, _engine_factory=None):
        """trie_decoder.py:41-218 with a caller-supplied `step`: -> (predictions [B, length <= max_steps] incl. the start
        tokens, logprobs [B]); ([B, 1], [B, 1]) when every first prediction is EOS."""
        if do_sample or not only_return_best or num_return_sequences != 1 or temperature != 1:
            raise NotImplementedError("TrieAutoRegressiveBeamSearch.search: only the inference form is implemented")
        s = Engine.make_search("trie", self.max_steps, 1, 1)
        return _begin_and_run(self, start_predictions, step, s, first_step_rows_per_sentence=True, stop_when_all_eos=True,
                              fmt="autoregressive", engine_factory=_engine_factory)


class GeneratorWithBeamSearch:
    """Constructor-compatible with the reference class (decoder.py:1057-1081)."""

    def __init__(self, eos_index: int, max_steps: int, beam_size: int, per_node_beam_size: int = 2,
                 length_penalty: float = 1, repetition_penalty: float = 1, temperature: float = 1) -> None:
        self._eos_index = eos_index
        self.max_steps = max_steps
        self.beam_size = beam_size
        self.per_node_beam_size = per_node_beam_size or beam_size
        self.length_penalty = length_penalty
        assert self.per_node_beam_size > 1
        assert self.length_penalty > 0, "`length_penalty` should be strictely positive."
        assert repetition_penalty >= 1.0, "`repetition_penalty` should be >= 1."        # decoder.py:1080
        self.repetition_penalty = repetition_penalty
        assert temperature > 0, "`temperature` should be strictely positive."        # decoder.py:1081
        self.temperature = temperature
        self.kind = "generator"

    def search(self, input_ids: torch.Tensor, step, num_keep_best: int = 1, do_sample: bool = False, top_k=None,
               top_p=None, num_return_sequences: int = 1, seed: int = 0, _engine_factory=None):
        """decoder.py:1083-1290 with a caller-supplied `step`: -> (decoded int64 [B, max_steps] = best hypothesis + EOS,
        right-padded with EOS, logprobs fp32 [B, 1]); with num_keep_best = n > 1 the n best hypotheses of every sentence,
        best first: ([B, n, max_steps], [B, n]), missing ones all EOS at -1e5 (decoder.py:1262-1290); B counts every
        sentence num_return_sequences times (decoder.py:1093-1097).  Like the reference, the loop stops calling `step` once
        every sentence is done (decoder.py:1251)."""
        if num_return_sequences != 1:                   # decoder.py:1093-1097: every sentence num_return_sequences times
            input_ids = input_ids[:, None, :].expand(input_ids.shape[0], num_return_sequences, input_ids.shape[1])
            input_ids = input_ids.reshape(-1, input_ids.shape[-1])
        s = Engine.make_search("generator", self.max_steps, self.beam_size, self.per_node_beam_size, self.length_penalty,
                               do_sample=do_sample, top_k=top_k or 0, top_p=top_p, temperature=self.temperature, seed=seed,
                               repetition_penalty=self.repetition_penalty, num_keep_best=num_keep_best)
        return _begin_and_run(self, input_ids, step, s, first_step_rows_per_sentence=False, stop_when_all_eos=False,
                              fmt="generator", engine_factory=_engine_factory)


def load_state_dict_by_suffix(model_keys: Sequence[str], loaded: Mapping[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Key alignment of torch_common.py:45-54, 93-145: strip EVERY leading 'module.' (DataParallel wrapped any number
    of times), then give every model key the loaded key that is its LONGEST (purely textual) suffix; model keys that
    nothing matches are left out.  Pinned against the reference's align_and_update_state_dicts by
    tests/golden/state_dict_align.json (oracle/make_host_golden.py)."""
    stripped = {}
    for k, v in loaded.items():
        while k.startswith("module."):
            k = k[len("module."):]
        stripped[k] = v
    out: Dict[str, torch.Tensor] = {}
    for key in model_keys:
        best = None
        for cand in stripped:
            if key.endswith(cand) and (best is None or len(cand) > len(best)):
                best = cand
        if best is not None:
            out[key] = stripped[best]
    return out


def expected_state_dict_keys(cfg: GitModelConfig, tied_output: bool = False) -> Sequence[str]:
    """Keys the engine ingests."""
    keys = ["image_encoder.class_embedding", "image_encoder.positional_embedding", "image_encoder.conv1.weight",
            "image_encoder.ln_pre.weight", "image_encoder.ln_pre.bias", "image_encoder.ln_post.weight",
            "image_encoder.ln_post.bias"]
    for i in range(cfg.vit_layers):
        p = f"image_encoder.transformer.resblocks.{i}."
        keys += [p + s for s in ("attn.in_proj_weight", "attn.in_proj_bias", "attn.out_proj.weight",
                                 "attn.out_proj.bias", "ln_1.weight", "ln_1.bias", "mlp.c_fc.weight",
                                 "mlp.c_fc.bias", "mlp.c_proj.weight", "mlp.c_proj.bias", "ln_2.weight", "ln_2.bias")]
    keys += ["textual.visual_projection.0.weight", "textual.visual_projection.0.bias",
             "textual.visual_projection.1.weight", "textual.visual_projection.1.bias",
             "textual.embedding.words.weight", "textual.embedding.positions.weight",
             "textual.embedding.layer_norm.weight", "textual.embedding.layer_norm.bias"]
    for i in range(cfg.dec_layers):
        p = f"textual.transformer.encoder.layer.{i}."
        for nm in ("query", "key", "value"):
            keys += [p + f"attention.self.{nm}.weight", p + f"attention.self.{nm}.bias"]
        keys += [p + s for s in ("attention.output.dense.weight", "attention.output.dense.bias",
                                 "attention.output.LayerNorm.weight", "attention.output.LayerNorm.bias",
                                 "intermediate.dense.weight", "intermediate.dense.bias", "output.dense.weight",
                                 "output.dense.bias", "output.LayerNorm.weight", "output.LayerNorm.bias")]
    if not tied_output:
        keys.append("textual.output.weight")
    keys.append("textual.output.bias")
    keys += [f"img_temperal_embedding.{i}" for i in range(cfg.num_frames)]
    return keys


class CaptioningModel:
    """Callable like the reference model: ``model(batch) -> {'predictions', 'logprobs'}``.

    batch['image']  : FloatTensor [B,3,H,W] or a list of such (video frames)   (decoder.py:845-857)
    batch['prefix'] : LongTensor [1,P] starting with [CLS] (VQA question)       (decoder.py:984-989)
    """

    # precision: "f32" (reference-identical ids), "f16" (default 16-bit mode: fp16 operands, same MFMA rate as bf16 on gfx950,
    # 8 x smaller logit error: the build that meets the specification and the benchmarked one since round 6) or "bf16" (BASELINE.json's named precision)
    def __init__(self, cfg: GitModelConfig, decoder, precision: str = "f16", max_batch: int = 64,
                 max_frames: Optional[int] = None, max_text_len: Optional[int] = None,
                 device: Optional[int] = None):
        self.cfg = cfg
        self.decoder = decoder
        self.sos_index = cfg.sos
        self.eos_index = cfg.eos
        if max_frames is None:
            max_frames = max(1, cfg.num_frames)
        if max_text_len is None:
            max_text_len = min(cfg.max_pos, max(int(decoder.max_steps), 2))
        self.engine = Engine(cfg, precision=precision, max_batch=max_batch,
                             max_beams=max(1, int(decoder.beam_size)), max_frames=max_frames,
                             max_text_len=max_text_len, device=device)
        self._loaded = False

    # nn.Module look-alikes so that reference call sites (`model.cuda(); model.eval()`) keep working
    def cuda(self, *a, **k):
        return self

    def eval(self):
        return self

    def load_state_dict(self, state_dict: Mapping[str, torch.Tensor], strict: bool = False):
        tied = not any(k.endswith("textual.output.weight") for k in state_dict)   # keys may carry 'module.' prefixes
        keys = expected_state_dict_keys(self.cfg, tied_output=tied)
        aligned = load_state_dict_by_suffix(keys, state_dict)
        missing = [k for k in keys if k not in aligned]
        if missing:
            raise KeyError(f"checkpoint is missing {len(missing)} tensors, e.g. {missing[:4]}")
        self.engine.load_state_dict(aligned)
        self._loaded = True
        return self

    def _search_struct(self, search_param: Optional[dict] = None):
        """search_param: the keyword arguments CaptioningModel.infer forwards to decoder.search (decoder.py:1001-1003):
        do_sample / top_k / top_p for GeneratorWithBeamSearch (decoder.py:1088-1090); `seed` selects the random stream."""
        d = self.decoder
        sp = dict(search_param or {})
        unknown = set(sp) - {"do_sample", "top_k", "top_p", "seed", "num_keep_best", "num_return_sequences"}
        if unknown:
            raise NotImplementedError(f"search parameters {sorted(unknown)} are not implemented")
        if d.kind != "generator" and (int(sp.get("num_keep_best", 1)) != 1 or int(sp.get("num_return_sequences", 1)) != 1):
            raise NotImplementedError("num_keep_best / num_return_sequences: GeneratorWithBeamSearch.search only "
                                      "(the other classes use them in SCST training only)")
        return Engine.make_search(d.kind, d.max_steps, d.beam_size, d.per_node_beam_size, d.length_penalty,
                                  do_sample=bool(sp.get("do_sample", False)), top_k=sp.get("top_k") or 0,
                                  top_p=sp.get("top_p"), temperature=getattr(d, "temperature", 1.0),
                                  seed=int(sp.get("seed", 0)),
                                  repetition_penalty=getattr(d, "repetition_penalty", 1.0),
                                  num_keep_best=int(sp.get("num_keep_best", 1)))

    def close(self) -> None:
        """Free the engine contexts (workspaces, KV caches, packed weights) now rather than at garbage collection."""
        for c in (getattr(self, "_ctxs", None) or [])[1:]:
            c.close()
        self._ctxs = None
        self.engine.close()
        self._loaded = False

    # ---- several requests in flight (serving / the TSV task) -----------------------------------------------------------------
    def set_pipeline(self, contexts: int = 4, encoder_chains: int = 2) -> None:
        """Keep up to `contexts` requests in flight: context i (a clone that borrows the packed weights, gitmi_clone) runs on a
        HIP stream of its own, so the latency-bound decode steps of one request overlap the MFMA-bound image encoder of the
        next; at most `encoder_chains` encoders run at a time (gitmi_set_encode_after) -- the schedule bench.py measures.
        submit() then rotates over the contexts; model(batch) keeps using context 0 synchronously."""
        if not self._loaded:
            raise RuntimeError("weights not loaded (call load_state_dict first)")
        contexts = max(1, int(contexts))
        if getattr(self, "_ctxs", None) and len(self._ctxs) == contexts:
            return
        for c in getattr(self, "_ctxs", [])[1:]:
            c.close()
        if contexts > 1:
            self.engine.set_shared_device(True)          # kernel shapes by whole-device cost (bit-identical results)
        self._ctxs = [self.engine] + [self.engine.clone() for _ in range(contexts - 1)]
        self._streams = [torch.cuda.Stream() for _ in self._ctxs]
        chains = max(1, int(encoder_chains))
        if len(self._ctxs) > chains:
            for i, c in enumerate(self._ctxs):
                c.set_encode_after(self._ctxs[i - chains])
        self._next = 0

    def _context(self):
        """-> (engine context, its stream or None) of the next submission"""
        if not getattr(self, "_ctxs", None):
            return self.engine, None
        i = self._next % len(self._ctxs)
        self._next += 1
        return self._ctxs[i], self._streams[i]

    def _prepare(self, eng, frames_is_list: bool):
        eng.set_temporal_embedding(frames_is_list)                          # decoder.py:845-857: list branch only
        if self.decoder.kind == "trie" and getattr(eng, "_trie_loaded", None) is not self.decoder.trie:
            eng.set_trie(*self.decoder.trie.csr())
            eng._trie_loaded = self.decoder.trie

    def submit(self, batch: Mapping[str, Union[torch.Tensor, Sequence[torch.Tensor]]],
               search_param: Optional[dict] = None) -> "Pending":
        """Asynchronous model(batch): enqueue the request on the next context's stream and return at once; `.result()` waits
        for it and returns what model(batch) returns.  The batch's tensors must have been produced on the CURRENT stream (the
        context's stream waits for it)."""
        if not self._loaded:
            raise RuntimeError("weights not loaded (call load_state_dict first)")
        image = batch["image"]
        is_list = isinstance(image, (list, tuple))
        frames = list(image) if is_list else [image]
        eng, stream = self._context()
        if self.cfg.num_frames == 0 and len(frames) > eng.c.max_frames:
            # the reference concatenates the features of every frame of a list, also on an image model
            raise ValueError(f"{len(frames)} frames on an image model: construct CaptioningModel(..., max_frames="
                             f"{len(frames)}) (workspaces are sized at construction; now max_frames={eng.c.max_frames})")
        self._prepare(eng, is_list)
        prefix = batch.get("prefix")
        if prefix is not None:
            assert len(prefix) == 1, "not supported"                       # decoder.py:988
        search = self._search_struct(search_param)
        P = 1 if prefix is None else int(prefix.numel())
        nret = int((search_param or {}).get("num_return_sequences", 1))
        kind = self.decoder.kind

        def launch():
            if nret != 1:
                # decoder.py:1093-1097: every image's start tokens num_return_sequences times -- r sentences per image, each
                # with its own beams (they differ only when sampling); rows b * r + j, as the reference returns them
                B = int(frames[0].shape[0])
                start = [int(self.cfg.sos)] if prefix is None else [int(t) for t in prefix.reshape(-1).tolist()]
                if B * nret > eng.c.max_batch:
                    raise ValueError(f"{B} images x num_return_sequences={nret} exceed max_batch={eng.c.max_batch}")
                tokens, logprobs, _, info = eng.generate_prefixed(
                    frames, search, [start] * (B * nret), image_of=[b for b in range(B) for _ in range(nret)], sync=False,
                    host_out=stream is not None)
            else:
                # requests in flight on other streams: results straight into page-locked host memory (no read-back to enqueue)
                tokens, logprobs, info = eng.generate(frames, search, prefix=prefix, sync=False, host_out=stream is not None)
            return tokens, logprobs, info

        def finish(out):
            tokens, logprobs, info = out
            info_h = info.tolist()                                              # ONE small read-back for the four fields
            eng.check_finite(info_h)
            seq_len, early, _, _ = info_h
            if kind in ("autoregressive", "trie"):
                if early:                                                       # decoder.py:279-291 / trie_decoder.py:76-83
                    predictions = tokens[:, P:P + 1]
                    logprobs = logprobs[:, None]
                else:
                    predictions = tokens[:, :seq_len]
            else:
                predictions = tokens                                            # [B, T], or [B, num_keep_best, T]
                if logprobs.dim() == 1:
                    logprobs = logprobs[:, None]                                # [B, num_keep_best]
            if prefix is not None:
                predictions = predictions[:, P:]                                # decoder.py:1004-1006 (dim 1, whatever it is)
            return {"predictions": predictions, "logprobs": logprobs}

        return Pending(stream, launch, finish, keep=(frames, prefix))

    def forward(self, batch: Mapping[str, Union[torch.Tensor, Sequence[torch.Tensor]]],
                search_param: Optional[dict] = None) -> Dict[str, torch.Tensor]:
        saved, self._ctxs = getattr(self, "_ctxs", None), None              # model(batch): context 0, the caller's stream
        try:
            return self.submit(batch, search_param).result()
        finally:
            self._ctxs = saved

    __call__ = forward

    def submit_answers(self, images: Union[torch.Tensor, Sequence[torch.Tensor]], prefixes: Sequence[Sequence[int]],
                       image_of: Optional[Sequence[int]] = None) -> "Pending":
        """Questions about SEVERAL images of one resolution in one engine call (batched ragged prefixes): `images` [B,3,H,W]
        (or a list of frames of that shape), question q = token ids starting with [CLS], about image image_of[q] (default: all
        about image 0).  `.result()` returns, per question, the list of predicted token ids exactly as
        ``model({'image': image, 'prefix': [prefix]})['predictions'][0]`` gives them for that image alone -- the reference loop
        of inference.py:172-199 without re-encoding an image per question and without one call per image."""
        if not self._loaded:
            raise RuntimeError("weights not loaded (call load_state_dict first)")
        is_list = isinstance(images, (list, tuple))
        frames = list(images) if is_list else [images]
        Q = len(prefixes)
        image_of = [0] * Q if image_of is None else [int(i) for i in image_of]
        eng, stream = self._context()
        if Q > eng.c.max_batch or int(frames[0].shape[0]) > eng.c.max_batch:
            raise ValueError(f"{Q} questions / {int(frames[0].shape[0])} images exceed max_batch={eng.c.max_batch}")
        self._prepare(eng, is_list)
        search = self._search_struct()
        kind = self.decoder.kind

        def launch():
            return eng.generate_prefixed(frames, search, prefixes, image_of=image_of, sync=False, host_out=stream is not None)

        def finish(out):
            tokens, logprobs, sent, info = out
            eng.check_finite(info.tolist())
            tokens, sent = tokens.cpu(), sent.cpu()
            res = []
            for q, p in enumerate(prefixes):
                P = len(p)
                L, early = int(sent[q, 0]), int(sent[q, 1])
                if kind in ("autoregressive", "trie"):
                    row = (tokens[q, P:P + 1] if early else tokens[q, :L])[P:]     # decoder.py:279-291, then :1004-1006
                else:
                    row = tokens[q, P:]
                res.append(row.tolist())
            return res

        return Pending(stream, launch, finish, keep=(frames,))

    def answer(self, image: Union[torch.Tensor, Sequence[torch.Tensor]], prefixes: Sequence[Sequence[int]]):
        """Several questions about ONE image in one engine call: submit_answers(...).result() on context 0."""
        frames = list(image) if isinstance(image, (list, tuple)) else [image]
        assert frames[0].shape[0] == 1, "answer() takes one image (or one clip)"
        saved, self._ctxs = getattr(self, "_ctxs", None), None
        try:
            return self.submit_answers(image, prefixes).result()
        finally:
            self._ctxs = saved


class Pending:
    """A request enqueued on a context's stream (CaptioningModel.submit / submit_answers)."""

    def __init__(self, stream, launch, finish, keep=()):
        self._finish, self._keep, self._done, self._value, self._stream = finish, keep, False, None, stream
        if stream is None:
            self._out = launch()
            self._event = None
        else:
            stream.wait_stream(torch.cuda.current_stream())          # the inputs were produced on the caller's stream
            with torch.cuda.stream(stream):
                self._out = launch()
                self._event = torch.cuda.Event()
                self._event.record()

    wait_s = 0.0            # class-wide: seconds spent waiting for the device in result() (diagnostics of the TSV task)

    def result(self):
        if not self._done:
            import time
            t0 = time.perf_counter()
            if self._event is not None:
                self._event.synchronize()
            else:
                torch.cuda.current_stream().synchronize()
            Pending.wait_s += time.perf_counter() - t0
            self._value = self._finish(self._out)
            self._done, self._out, self._keep = True, None, ()
        return self._value


def get_git_model(tokenizer, param: Optional[dict], precision: str = "f16", max_batch: int = 64,
                  decoder=None, device: Optional[int] = None) -> CaptioningModel:
    """Same role as the reference's get_git_model (model.py:9-61): GIT decoder hyper-parameters are
    fixed, the encoder follows param['image_encoder_type'].  The default search is the shipped one:
    GeneratorWithBeamSearch(beam_size=4, length_penalty=0.6, max_steps=1024) (model.py:34-40)."""
    cfg = config_from_param(param)
    sos = getattr(tokenizer, "cls_token_id", None)
    eos = getattr(tokenizer, "sep_token_id", None)
    if sos is not None and eos is not None and (sos != cfg.sos or eos != cfg.eos):
        import dataclasses
        cfg = dataclasses.replace(cfg, sos=int(sos), eos=int(eos))
    if decoder is None:
        decoder = GeneratorWithBeamSearch(eos_index=cfg.eos, max_steps=1024, beam_size=4, length_penalty=0.6)
    logging.info("building GIT engine: %s, search %s", cfg, type(decoder).__name__)
    return CaptioningModel(cfg, decoder, precision=precision, max_batch=max_batch, device=device)
