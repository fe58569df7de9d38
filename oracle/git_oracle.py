"""CPU oracle for the GIT captioning hot path  --  TEST INFRASTRUCTURE ONLY.

This file is a plain fp32 restatement (torch CPU tensor algebra, no nn.Module,
no reference imports) of the algorithm that microsoft/GenerativeImage2Text runs
for  image -> caption token IDs.  It exists so that the HIP engine in
``generativeimage2text_amd/`` can be checked on a machine where the reference
itself is not present (the GPU box).  Only ``tests/``, ``__graft_entry__.smoke``
and the ``cpu_baseline`` leg of ``bench.py`` may import it; the product path never
does (it fails loudly when the HIP library is missing).

Parity pin: the reference ships no tests/golden vectors for this path
(SURVEY.md section 4), so the oracle is pinned the other way round:
``oracle/make_golden.py`` imports the *real* reference modules from
/root/reference in the build container, loads the same seeded weights into them,
and asserts this restatement reproduces their features / logits / token IDs;
the resulting vectors are frozen under ``tests/golden/``.

Each function cites the reference lines it follows (paths relative to
/root/reference/generativeimage2text/).
"""
from __future__ import annotations

import dataclasses
import math
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch

Tensor = torch.Tensor
Weights = Dict[str, Tensor]


# ----------------------------------------------------------------------------
# configuration (model.py:9-61 hard-codes the decoder; parameter.yaml picks the
# encoder / visual_feature_size / num_image_with_embedding)
# ----------------------------------------------------------------------------
@dataclasses.dataclass(frozen=True)
class GitConfig:
    name: str = "GIT_BASE"
    image_size: int = 224
    patch: int = 16
    vit_width: int = 768
    vit_layers: int = 12
    vit_heads: int = 12
    dec_hidden: int = 768
    dec_layers: int = 6
    dec_heads: int = 12
    dec_ffn: int = 3072
    vocab: int = 30522
    max_pos: int = 1024
    num_frames: int = 0          # num_image_with_embedding (temporal embeddings)
    sos: int = 101               # tokenizer.cls_token_id
    eos: int = 102               # tokenizer.sep_token_id

    @property
    def grid(self) -> int:
        return self.image_size // self.patch

    @property
    def n_tok(self) -> int:      # tokens per frame, class token included
        return self.grid * self.grid + 1

    @property
    def vit_mlp(self) -> int:
        return 4 * self.vit_width

    @property
    def vfs(self) -> int:        # visual_feature_size
        return self.vit_width


CONFIGS: Dict[str, GitConfig] = {
    # model.py:64-67 + CLIP ViT-B/16: VisualTransformer(224,16,768,12,12,512)
    "GIT_BASE": GitConfig(),
    # aux_data/models/GIT_LARGE*/parameter.yaml: ViT-L/14 = (224,14,1024,24,16,768)
    "GIT_LARGE": GitConfig(name="GIT_LARGE", patch=14, vit_width=1024, vit_layers=24, vit_heads=16),
    # aux_data/models/GIT_BASE_VATEX/parameter.yaml: num_image_with_embedding: 6
    "GIT_BASE_VATEX": GitConfig(name="GIT_BASE_VATEX", num_frames=6),
    # aux_data/models/GIT_BASE_VQAv2/parameter.yaml: test_crop_size 480 (native grid 30x30), inputs up to 640 long
    "GIT_BASE_VQAv2": GitConfig(name="GIT_BASE_VQAv2", image_size=480),
    # reduced shapes for fast CPU tests (same structure, head_dim stays 64)
    "TINY": GitConfig(name="TINY", image_size=64, patch=16, vit_width=128, vit_layers=2, vit_heads=2,
                      dec_hidden=128, dec_layers=2, dec_heads=2, dec_ffn=512, vocab=1000, max_pos=64),
    "TINY_VIDEO": GitConfig(name="TINY_VIDEO", image_size=64, patch=16, vit_width=128, vit_layers=2,
                            vit_heads=2, dec_hidden=128, dec_layers=2, dec_heads=2, dec_ffn=512,
                            vocab=1000, max_pos=64, num_frames=3),
    # odd patch (K = 3*14*14 = 588, like ViT-L/14) and wider encoder than decoder
    "TINY_L": GitConfig(name="TINY_L", image_size=56, patch=14, vit_width=192, vit_layers=2, vit_heads=3,
                        dec_hidden=128, dec_layers=2, dec_heads=2, dec_ffn=512, vocab=1000, max_pos=64),
}


# ----------------------------------------------------------------------------
# seeded synthetic weights, keyed exactly like the reference state dict
# (SURVEY.md 8a-D).  No checkpoint is available offline, so parity is on
# token IDs / logits with these weights.
# ----------------------------------------------------------------------------
def make_weights(cfg: GitConfig, seed: int = 1234, tie_output: bool = True,
                 eos_bias: float = 0.0, out_scale: float = 1.0, successor: float = 0.0) -> Weights:
    """Deterministic fp32 weights.

    Shapes/initial scales follow CLIP/model.py:224-235 (ViT: scale*randn
    embeddings, default Linear init) and decoder.py:507-517 (decoder: N(0, .02)).
    LayerNorm affines and temporal embeddings are perturbed (they are 1/0/0 at
    reference init, which would hide a missing affine or a missing add).
    ``eos_bias`` lifts the EOS logit so that end-of-sentence paths get exercised.
    ``successor`` (untied output only) adds s * E[perm] to the output matrix, E = word embedding,
    perm a seeded permutation: the logit of token v then grows with the similarity of the hidden
    state to the embedding of perm[v], i.e. tokens tend to be followed by "their successor".  Plain
    random decoders fall into A-B-A-B loops under the no-immediate-repeat rule (decoder.py:330);
    this keeps 19-step decodes long and diverse without changing any shape or code path.
    """
    g = torch.Generator().manual_seed(seed)

    def rn(*shape, std=1.0):
        return torch.randn(*shape, generator=g, dtype=torch.float32) * std

    def ln(prefix, d, w):
        w[prefix + ".weight"] = 1.0 + rn(d, std=0.1)
        w[prefix + ".bias"] = rn(d, std=0.05)

    w: Weights = {}
    D, F = cfg.vit_width, cfg.vit_mlp
    sc = D ** -0.5
    w["image_encoder.class_embedding"] = rn(D, std=sc)
    w["image_encoder.positional_embedding"] = rn(cfg.n_tok, D, std=sc)
    w["image_encoder.conv1.weight"] = rn(D, 3, cfg.patch, cfg.patch, std=(3 * cfg.patch ** 2) ** -0.5)
    ln("image_encoder.ln_pre", D, w)
    for i in range(cfg.vit_layers):
        p = f"image_encoder.transformer.resblocks.{i}."
        w[p + "attn.in_proj_weight"] = rn(3 * D, D, std=D ** -0.5)
        w[p + "attn.in_proj_bias"] = rn(3 * D, std=0.02)
        w[p + "attn.out_proj.weight"] = rn(D, D, std=D ** -0.5)
        w[p + "attn.out_proj.bias"] = rn(D, std=0.02)
        ln(p + "ln_1", D, w)
        w[p + "mlp.c_fc.weight"] = rn(F, D, std=D ** -0.5)
        w[p + "mlp.c_fc.bias"] = rn(F, std=0.02)
        w[p + "mlp.c_proj.weight"] = rn(D, F, std=F ** -0.5)
        w[p + "mlp.c_proj.bias"] = rn(D, std=0.02)
        ln(p + "ln_2", D, w)
    ln("image_encoder.ln_post", D, w)

    d, f, V = cfg.dec_hidden, cfg.dec_ffn, cfg.vocab
    w["textual.visual_projection.0.weight"] = rn(d, cfg.vfs, std=cfg.vfs ** -0.5)
    w["textual.visual_projection.0.bias"] = rn(d, std=0.02)
    ln("textual.visual_projection.1", d, w)
    w["textual.embedding.words.weight"] = rn(V, d, std=0.05)
    w["textual.embedding.positions.weight"] = rn(cfg.max_pos, d, std=0.05)
    ln("textual.embedding.layer_norm", d, w)
    for i in range(cfg.dec_layers):
        p = f"textual.transformer.encoder.layer.{i}."
        for nm in ("query", "key", "value"):
            w[p + f"attention.self.{nm}.weight"] = rn(d, d, std=d ** -0.5)
            w[p + f"attention.self.{nm}.bias"] = rn(d, std=0.02)
        w[p + "attention.output.dense.weight"] = rn(d, d, std=d ** -0.5)
        w[p + "attention.output.dense.bias"] = rn(d, std=0.02)
        ln(p + "attention.output.LayerNorm", d, w)
        w[p + "intermediate.dense.weight"] = rn(f, d, std=d ** -0.5)
        w[p + "intermediate.dense.bias"] = rn(f, std=0.02)
        w[p + "output.dense.weight"] = rn(d, f, std=f ** -0.5)
        w[p + "output.dense.bias"] = rn(d, std=0.02)
        ln(p + "output.LayerNorm", d, w)
    if tie_output:   # decoder.py:503-505
        w["textual.output.weight"] = w["textual.embedding.words.weight"]
    else:
        w["textual.output.weight"] = rn(V, d, std=0.05)
        if successor != 0.0:
            perm = torch.randperm(V, generator=torch.Generator().manual_seed(seed + 7919))
            w["textual.output.weight"] = w["textual.output.weight"] + successor * w["textual.embedding.words.weight"][perm]
    if out_scale != 1.0:
        w["textual.output.weight"] = w["textual.output.weight"] * out_scale
        if tie_output:
            w["textual.embedding.words.weight"] = w["textual.output.weight"]
    ob = rn(V, std=0.02)
    ob[cfg.eos] += eos_bias
    w["textual.output.bias"] = ob
    for i in range(cfg.num_frames):   # decoder.py:831-836 (zeros at init; perturbed here)
        w[f"img_temperal_embedding.{i}"] = rn(1, 1, cfg.vfs, std=0.2)
    return w


def make_images(cfg: GitConfig, batch: int, frames: int = 1, seed: int = 0,
                hw: Optional[Tuple[int, int]] = None) -> List[Tensor]:
    """Synthetic post-Normalize images, one [B,3,H,W] tensor per frame.  hw: non-native
    resolution (the MinMaxResizeForTest models)."""
    g = torch.Generator().manual_seed(seed)
    H, W = hw if hw is not None else (cfg.image_size, cfg.image_size)
    return [torch.randn(batch, 3, H, W, generator=g) for _ in range(frames)]


# ----------------------------------------------------------------------------
# small algebra helpers
# ----------------------------------------------------------------------------
def _affine(x: Tensor, weight: Tensor, bias: Optional[Tensor]) -> Tensor:
    y = x @ weight.t()
    return y if bias is None else y + bias


def _layer_norm(x: Tensor, gamma: Tensor, beta: Tensor, eps: float) -> Tensor:
    mu = x.mean(dim=-1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=-1, keepdim=True)        # biased, like nn.LayerNorm
    return (x - mu) * torch.rsqrt(var + eps) * gamma + beta


def _split_heads(x: Tensor, heads: int) -> Tensor:          # [B,N,D] -> [B,H,N,hd]
    b, n, d = x.shape
    return x.reshape(b, n, heads, d // heads).permute(0, 2, 1, 3)


def _merge_heads(x: Tensor) -> Tensor:                      # [B,H,N,hd] -> [B,N,D]
    b, h, n, hd = x.shape
    return x.permute(0, 2, 1, 3).reshape(b, n, h * hd)


# ----------------------------------------------------------------------------
# image encoder  (layers/CLIP/model.py)
# ----------------------------------------------------------------------------
def _cubic_weights(t: Tensor, A: float = -0.75) -> Tensor:
    """Cubic-convolution coefficients for taps at offsets -1, 0, +1, +2 (ATen UpSample.h
    get_cubic_upsample_coefficients; Keys kernel with A = -0.75).  t in [0,1) -> [..., 4]."""
    def near(x):   # |x| <= 1
        return ((A + 2.0) * x - (A + 3.0)) * x * x + 1.0

    def far(x):    # 1 < |x| < 2
        return ((A * x - 5.0 * A) * x + 8.0 * A) * x - 4.0 * A
    return torch.stack([far(t + 1.0), near(t), near(1.0 - t), far(2.0 - t)], dim=-1)


def bicubic_resize_grid(grid: Tensor, out_h: int, out_w: int) -> Tensor:
    """torch.nn.functional.interpolate(mode='bicubic', align_corners=False, size=...) restated for a
    [gh, gw, D] grid (what CLIP/model.py:243-251 applies to the positional embedding when the input is
    not the native resolution): source coordinate (o + 0.5) * in/out - 0.5 (NOT clamped for cubic),
    4x4 taps with border-clamped indices, rows first then columns.  -> [out_h, out_w, D]"""
    gh, gw, _ = grid.shape

    def axis(n_in, n_out):
        src = (torch.arange(n_out, dtype=torch.float32) + 0.5) * (float(n_in) / float(n_out)) - 0.5
        i0 = torch.floor(src)
        wts = _cubic_weights(src - i0)                                           # [n_out, 4]
        idx = (i0.long()[:, None] + torch.arange(-1, 3)[None, :]).clamp(0, n_in - 1)   # [n_out, 4]
        return idx, wts
    iy, wy = axis(gh, out_h)
    ix, wx = axis(gw, out_w)
    rows = grid[iy]                                              # [out_h, 4, gw, D]
    cols = rows[:, :, ix]                                        # [out_h, 4, out_w, 4, D]
    horiz = (cols * wx[None, None, :, :, None]).sum(dim=3)       # interpolate along x for each tap row
    return (horiz * wy[:, :, None, None]).sum(dim=1)             # then along y


def vit_positional(cfg: GitConfig, w: Weights, gh: int, gw: int) -> Tensor:
    """Positional embedding for a gh x gw token grid: the stored table at the native grid, otherwise its
    bicubic resize with the class row kept (CLIP/model.py:243-251)."""
    pos = w["image_encoder.positional_embedding"]
    g = cfg.grid
    if (gh, gw) == (g, g):
        return pos
    body = bicubic_resize_grid(pos[1:].reshape(g, g, -1), gh, gw).reshape(gh * gw, -1)
    return torch.cat([pos[:1], body], dim=0)


def vit_stem(cfg: GitConfig, w: Weights, images: Tensor) -> Tensor:
    """CLIP/model.py:241-257: patchify-conv (no bias), class token, positional add, ln_pre.  Any H, W >= patch:
    the stride-p convolution drops the H % p / W % p remainder rows and columns."""
    b = images.shape[0]
    p, D = cfg.patch, cfg.vit_width
    gh, gw = images.shape[2] // p, images.shape[3] // p
    images = images[:, :, : gh * p, : gw * p]
    # conv k=s=p == per-patch dot product; K index = c*p*p + ky*p + kx (row-major patches)
    patches = images.reshape(b, 3, gh, p, gw, p).permute(0, 2, 4, 1, 3, 5).reshape(b, gh * gw, 3 * p * p)
    x = patches @ w["image_encoder.conv1.weight"].reshape(D, 3 * p * p).t()
    cls = w["image_encoder.class_embedding"].expand(b, 1, D)
    x = torch.cat([cls, x], dim=1) + vit_positional(cfg, w, gh, gw)
    return _layer_norm(x, w["image_encoder.ln_pre.weight"], w["image_encoder.ln_pre.bias"], 1e-5)


def vit_block(cfg: GitConfig, w: Weights, i: int, x: Tensor) -> Tensor:
    """CLIP/model.py:189-202: pre-LN block, nn.MultiheadAttention + QuickGELU MLP."""
    p = f"image_encoder.transformer.resblocks.{i}."
    H = cfg.vit_heads
    hd = cfg.vit_width // H
    h = _layer_norm(x, w[p + "ln_1.weight"], w[p + "ln_1.bias"], 1e-5)
    qkv = _affine(h, w[p + "attn.in_proj_weight"], w[p + "attn.in_proj_bias"])
    q, k, v = (_split_heads(t, H) for t in qkv.chunk(3, dim=-1))
    att = torch.softmax((q * hd ** -0.5) @ k.transpose(-1, -2), dim=-1) @ v
    x = x + _affine(_merge_heads(att), w[p + "attn.out_proj.weight"], w[p + "attn.out_proj.bias"])
    h = _layer_norm(x, w[p + "ln_2.weight"], w[p + "ln_2.bias"], 1e-5)
    u = _affine(h, w[p + "mlp.c_fc.weight"], w[p + "mlp.c_fc.bias"])
    u = u * torch.sigmoid(1.702 * u)                         # QuickGELU, model.py:171-173
    return x + _affine(u, w[p + "mlp.c_proj.weight"], w[p + "mlp.c_proj.bias"])


def vit_forward(cfg: GitConfig, w: Weights, images: Tensor) -> Tensor:
    """VisualTransformer.forward with output_grid=grid_after_ln=True (model.py:73-74):
    ln_post over ALL tokens, no projection.  -> [B, N, vit_width]"""
    x = vit_stem(cfg, w, images)
    for i in range(cfg.vit_layers):
        x = vit_block(cfg, w, i, x)
    return _layer_norm(x, w["image_encoder.ln_post.weight"], w["image_encoder.ln_post.bias"], 1e-5)


def visual_features(cfg: GitConfig, w: Weights, frames: Sequence[Tensor], as_list: bool = True) -> Tensor:
    """CaptioningModel.forward_one image branch (decoder.py:845-857): per-frame encode,
    + temporal embedding i, concat on the token axis (zip truncates to #embeddings).
    as_list=False: batch['image'] was a bare tensor -- image_encoder only, NO temporal embedding even on a
    video model (the else branch, decoder.py:856-857)."""
    feats = [vit_forward(cfg, w, f) for f in frames]
    if not as_list:
        assert len(feats) == 1
        return feats[0]
    if cfg.num_frames:
        feats = [f + w[f"img_temperal_embedding.{i}"] for i, f in enumerate(feats[: cfg.num_frames])]
    return feats[0] if len(feats) == 1 else torch.cat(feats, dim=1)


# ----------------------------------------------------------------------------
# text decoder (layers/decoder.py + layers/bert/modeling_bert.py)
# ----------------------------------------------------------------------------
def project_visual(cfg: GitConfig, w: Weights, feats: Tensor) -> Tensor:
    """'linearLn' projection, decoder.py:22-39 applied at :535 (LayerNorm default eps 1e-5)."""
    y = _affine(feats, w["textual.visual_projection.0.weight"], w["textual.visual_projection.0.bias"])
    return _layer_norm(y, w["textual.visual_projection.1.weight"], w["textual.visual_projection.1.bias"], 1e-5)


def embed_tokens(cfg: GitConfig, w: Weights, tokens: Tensor, first_pos: int = 0) -> Tensor:
    """WordAndPositionalEmbedding.forward, decoder.py:65-78 (LayerNorm eps 1e-8)."""
    t = tokens.shape[1]
    x = w["textual.embedding.words.weight"][tokens] + \
        w["textual.embedding.positions.weight"][first_pos: first_pos + t]
    return _layer_norm(x, w["textual.embedding.layer_norm.weight"], w["textual.embedding.layer_norm.bias"], 1e-8)


def _gelu_erf(u: Tensor) -> Tensor:                           # activations.py:15-22
    return u * 0.5 * (1.0 + torch.erf(u / math.sqrt(2.0)))


def bert_layer(cfg: GitConfig, w: Weights, i: int, x_q: Tensor, x_kv: Tensor, mask: Optional[Tensor]) -> Tensor:
    """One post-norm BertLayer (modeling_bert.py:122-159, 171-178, 228-250, 283-297).

    ``x_q`` are the rows whose outputs are wanted, ``x_kv`` the rows they may attend to
    (the reference always passes the same tensor; splitting them is what lets the
    cached variant below restate the same math).  ``mask`` is additive [.., Nq, Nk]."""
    p = f"textual.transformer.encoder.layer.{i}."
    H = cfg.dec_heads
    hd = cfg.dec_hidden // H
    q = _split_heads(_affine(x_q, w[p + "attention.self.query.weight"], w[p + "attention.self.query.bias"]), H)
    k = _split_heads(_affine(x_kv, w[p + "attention.self.key.weight"], w[p + "attention.self.key.bias"]), H)
    v = _split_heads(_affine(x_kv, w[p + "attention.self.value.weight"], w[p + "attention.self.value.bias"]), H)
    s = (q / math.sqrt(hd)) @ k.transpose(-1, -2)            # qk2attn, modeling_bert.py:41-47
    if mask is not None:
        s = s + mask
    ctx = _merge_heads(torch.softmax(s, dim=-1) @ v)
    a = _affine(ctx, w[p + "attention.output.dense.weight"], w[p + "attention.output.dense.bias"])
    a = _layer_norm(a + x_q, w[p + "attention.output.LayerNorm.weight"], w[p + "attention.output.LayerNorm.bias"], 1e-12)
    u = _gelu_erf(_affine(a, w[p + "intermediate.dense.weight"], w[p + "intermediate.dense.bias"]))
    o = _affine(u, w[p + "output.dense.weight"], w[p + "output.dense.bias"])
    return _layer_norm(o + a, w[p + "output.LayerNorm.weight"], w[p + "output.LayerNorm.bias"], 1e-12)


def joint_mask(n_img: int, t: int) -> Tensor:
    """decoder.py:111-149 + 602-610: image->image 0, image->text -inf, text->image 0, text->text causal."""
    n = n_img + t
    m = torch.zeros(n, n)
    m[:n_img, n_img:] = float("-inf")
    m[n_img:, n_img:] = torch.triu(torch.full((t, t), float("-inf")), diagonal=1)
    return m


def textual_logits_full(cfg: GitConfig, w: Weights, feats: Tensor, tokens: Tensor) -> Tensor:
    """TransformerDecoderTextualHead.forward as the reference runs it at inference:
    FULL recompute of projection + all layers over [image | text] every call
    (decoder.py:521-600, 97-163).  feats [R,N_img,vfs], tokens [R,t] -> logits [R,t,V]."""
    n_img = feats.shape[1]
    t = tokens.shape[1]
    x = torch.cat([project_visual(cfg, w, feats), embed_tokens(cfg, w, tokens)], dim=1)
    mask = joint_mask(n_img, t)
    for i in range(cfg.dec_layers):
        x = bert_layer(cfg, w, i, x, x, mask)
    return _affine(x[:, n_img:], w["textual.output.weight"], w["textual.output.bias"])


class CachedDecoder:
    """Mathematically identical restatement with the image rows computed once.

    Image rows never see text (mask top-right = -inf) and text is causal, so the hidden
    state of every image row at every layer is independent of the caption, and the text
    rows' K/V at positions < t do not change when tokens are appended (SURVEY.md headline
    fact 3).  Used where the full-recompute oracle would take minutes (B=64), and pinned
    against ``textual_logits_full`` in tests/test_oracle.py."""

    def __init__(self, cfg: GitConfig, w: Weights, feats: Tensor):
        self.cfg, self.w = cfg, w
        x = project_visual(cfg, w, feats)
        self.img_hidden: List[Tensor] = []            # input of layer i, image rows [B,N_img,d]
        for i in range(cfg.dec_layers):
            self.img_hidden.append(x)
            if i + 1 < cfg.dec_layers:                # layer-L outputs of image rows are never used
                x = bert_layer(cfg, w, i, x, x, None)

    def logits_last(self, tokens: Tensor, beams: int = 1) -> Tensor:
        """tokens [B*beams, t] (row = b*beams + j, decoder.py:1019-1025) -> next-token logits [R,V]."""
        cfg, w = self.cfg, self.w
        t = tokens.shape[1]
        x = embed_tokens(cfg, w, tokens)
        causal = torch.triu(torch.full((t, t), float("-inf")), diagonal=1)
        n_img = self.img_hidden[0].shape[1]
        mask = torch.cat([torch.zeros(t, n_img), causal], dim=1)
        for i in range(cfg.dec_layers):
            img = self.img_hidden[i]
            if beams > 1:
                img = img.repeat_interleave(beams, dim=0)
            x = bert_layer(cfg, w, i, x, torch.cat([img, x], dim=1), mask)
        return _affine(x[:, -1], w["textual.output.weight"], w["textual.output.bias"])


def make_step(cfg: GitConfig, w: Weights, feats: Tensor, cached: bool = False) -> Callable[[Tensor], Tensor]:
    """CaptioningModel.decoding_step (decoder.py:1013-1054): repeat image features per beam
    (rows of one image contiguous), run the textual head, return last-position fp32 logits."""
    b = feats.shape[0]
    cache = CachedDecoder(cfg, w, feats) if cached else None

    def step(tokens: Tensor) -> Tensor:
        beams = tokens.shape[0] // b
        if cache is not None:
            return cache.logits_last(tokens, beams)
        f = feats if beams == 1 else feats.repeat_interleave(beams, dim=0)
        return textual_logits_full(cfg, w, f, tokens)[:, -1, :].float()

    return step


# ----------------------------------------------------------------------------
# search strategies
# ----------------------------------------------------------------------------
def _adjacent_gap(sorted_scores: Tensor) -> Tensor:
    """Smallest difference between neighbours of a descending score list [B, n] -> [B]: the amount of
    logit noise that could change which candidates a top-k keeps, or their order."""
    return (sorted_scores[:, :-1] - sorted_scores[:, 1:]).min(dim=1).values


def search_autoregressive(start: Tensor, step: Callable[[Tensor], Tensor], eos: int, max_steps: int,
                          beam_size: int = 1, per_node_beam_size: int = 1,
                          trace: Optional[List[Tensor]] = None) -> Tuple[Tensor, Tensor]:
    """AutoRegressiveBeamSearch.search with fix_missing_prefix=True, only_return_best=True,
    no sampling (decoder.py:224-440).  Greedy oracle = beam_size=per_node=1.

    Returns (predictions int64 [B, len<=max_steps] INCLUDING the start tokens,
             logprobs fp32 [B] = summed logprob / num_valid)."""
    B, P = start.shape
    k, pn = beam_size, per_node_beam_size
    preds = start[:, None, :].expand(B, k, P)                                   # :249
    lp0 = torch.log_softmax(step(start), dim=1)                                 # :257-265
    V = lp0.shape[1]
    last_lp, cls0 = lp0.topk(k)                                                 # :271
    if trace is not None:       # test diagnostics only: decision margin of this step per image
        trace.append(_adjacent_gap(lp0.topk(k + 1).values))
    if k == 1 and bool((cls0 == eos).all()):                                    # :279-289
        return cls0, last_lp
    preds = torch.cat([preds, cls0[:, :, None]], dim=-1)                        # :298
    while preds.shape[-1] < max_steps:                                          # :313
        last = preds[:, :, -1].reshape(B * k)
        if bool((last == eos).all()):                                           # :319
            break
        flat = preds.reshape(B * k, -1)
        logits = step(flat).clone()
        logits.scatter_(1, last[:, None], -10000.0)                             # :330 no immediate repeat
        ended = last == eos
        if bool(ended.any()):                                                   # :347-351 force EOS after EOS
            forced = torch.full((V,), float("-inf"))
            forced[eos] = 0.0
            logits[ended] = forced
        lp = torch.log_softmax(logits, dim=1)                                   # :358
        top_lp, top_cls = lp.topk(pn)                                           # :366
        summed = (top_lp + last_lp.reshape(B * k, 1)).reshape(B, k * pn)        # :382-393
        cand_cls = top_cls.reshape(B, k * pn)
        cand_seq = torch.cat([flat[:, None, :].expand(B * k, pn, flat.shape[1]).reshape(B, k * pn, -1),
                              cand_cls[:, :, None]], dim=-1)                    # :399-405
        last_lp, keep = summed.topk(k)                                          # :409
        if trace is not None:
            if k == 1 and pn == 1:
                trace.append(_adjacent_gap(lp.topk(2).values))
            else:
                trace.append(_adjacent_gap(summed.topk(min(k + 1, k * pn)).values))
        preds = cand_seq.gather(1, keep[:, :, None].expand(B, k, cand_seq.shape[-1]))
    best = preds[:, 0, :]                                                       # :431
    best_lp = last_lp[:, 0]
    n_valid = (best != eos).sum(dim=-1) + ((best == eos).sum(dim=-1) > 0).long() - P   # :433-436
    return best, best_lp / n_valid.clamp(min=1)


class _Hypotheses:
    """BeamHypotheses (decoder.py:1292-1341): n-best list with OpenNMT length norm."""

    def __init__(self, n_keep: int, max_length: int, length_penalty: float):
        self.cap = max_length - 1
        self.alpha = length_penalty
        self.n_keep = n_keep
        self.items: List[Tuple[float, Tensor]] = []
        self.worst = 1e9

    def norm(self, length: int) -> float:
        return (5 + length) ** self.alpha / (5 + 1) ** self.alpha

    def add(self, seq: Tensor, sum_lp: float) -> None:
        score = sum_lp / self.norm(len(seq))
        if len(self.items) < self.n_keep or score > self.worst:
            self.items.append((score, seq))
            if len(self.items) > self.n_keep:
                order = sorted((s, j) for j, (s, _) in enumerate(self.items))
                del self.items[order[0][1]]
                self.worst = order[1][0]
            else:
                self.worst = min(score, self.worst)

    def is_done(self, best_sum_lp: float) -> bool:
        if len(self.items) < self.n_keep:
            return False
        return self.worst >= best_sum_lp / self.norm(self.cap)


def search_generator(start: Tensor, step: Callable[[Tensor], Tensor], eos: int, max_steps: int,
                     beam_size: int = 4, per_node_beam_size: int = 2,
                     length_penalty: float = 0.6, trace: Optional[List[Tensor]] = None,
                     repetition_penalty: float = 1.0, num_keep_best: int = 1,
                     num_return_sequences: int = 1) -> Tuple[Tensor, Tensor]:
    """GeneratorWithBeamSearch.search, greedy-beam branch (decoder.py:1083-1290).  Shipped default: beam 4, per_node 2,
    length_penalty 0.6.
    repetition_penalty != 1 (decoder.py:1135-1144): the raw score of every token already in a row's history is
    multiplied (score < 0) or divided (score >= 0) by it before the log-softmax.
    num_return_sequences r (decoder.py:1093-1097): every start row r times -> B*r sentences.
    num_keep_best n (decoder.py:1113-1115, 1262-1290): every sentence returns its n best finished hypotheses, best first.

    Returns (decoded int64 [B, max_steps] EOS-padded incl. start tokens, logprobs fp32 [B,1]); n > 1: ([B, n, max_steps],
    [B, n]), hypotheses the list cannot fill all EOS at -1e5."""
    if num_return_sequences != 1:
        start = start[:, None, :].expand(start.shape[0], num_return_sequences, start.shape[1]).reshape(-1, start.shape[1])
    B, cur = start.shape
    k = beam_size
    ids = start[:, None, :].expand(B, k, cur).reshape(B * k, cur)
    hyps = [_Hypotheses(num_keep_best, max_steps, length_penalty) for _ in range(B)]
    beam_scores = torch.zeros(B, k)
    beam_scores[:, 1:] = -1e9                                                   # :1118-1120
    beam_scores = beam_scores.reshape(-1)
    done = [False] * B
    while cur < max_steps:                                                      # :1128
        scores = step(ids)
        if repetition_penalty != 1.0:                                           # :1136-1144
            scores = scores.clone()
            for i in range(B * k):
                for tok in set(ids[i].tolist()):
                    if scores[i, tok] < 0:
                        scores[i, tok] *= repetition_penalty
                    else:
                        scores[i, tok] /= repetition_penalty
        lp = torch.log_softmax(scores, dim=-1)                                  # :1169
        V = lp.shape[-1]
        tot = (lp + beam_scores[:, None]).reshape(B, k * V)
        nxt_s, nxt_i = torch.topk(tot, per_node_beam_size * k, dim=1, largest=True, sorted=True)   # :1175
        if trace is not None:       # test diagnostics only (live images; finished ones report +inf)
            gap = _adjacent_gap(torch.topk(tot, per_node_beam_size * k + 1, dim=1).values)
            trace.append(torch.where(torch.tensor(done), torch.full_like(gap, float("inf")), gap))
        rows: List[Tuple[float, int, int]] = []
        for b in range(B):                                                      # :1184-1222
            done[b] = done[b] or hyps[b].is_done(float(nxt_s[b].max()))
            if done[b]:
                rows.extend([(0.0, eos, 0)] * k)
                continue
            sent: List[Tuple[float, int, int]] = []
            for idx, sc in zip(nxt_i[b].tolist(), nxt_s[b].tolist()):
                beam_id, word = idx // V, idx % V
                if word == eos or cur + 1 == max_steps:
                    hyps[b].add(ids[b * k + beam_id, :cur].clone(), sc)
                else:
                    sent.append((sc, word, b * k + beam_id))
                if len(sent) == k:
                    break
            assert len(sent) == (0 if cur + 1 == max_steps else k)
            if not sent:
                sent = [(0.0, eos, 0)] * k
            rows.extend(sent)
        beam_scores = torch.tensor([r[0] for r in rows], dtype=torch.float32)   # :1226-1232
        words = torch.tensor([r[1] for r in rows], dtype=ids.dtype)
        src = torch.tensor([r[2] for r in rows], dtype=torch.long)
        ids = torch.cat([ids[src], words[:, None]], dim=-1)
        cur += 1
        if all(done):
            break
    out = torch.full((B, num_keep_best, max_steps), eos, dtype=ids.dtype)       # :1283-1289 (pad id = eos)
    logprobs = torch.full((B, num_keep_best), -1e5)
    for b, h in enumerate(hyps):
        # :1270-1280: torch.topk over the list's scores (equal scores: list order)
        order = sorted(range(len(h.items)), key=lambda j: (-h.items[j][0], j))[:num_keep_best]
        for i, j in enumerate(order):
            score, seq = h.items[j]
            out[b, i, : len(seq)] = seq
            logprobs[b, i] = score
    if num_keep_best == 1:
        out = out[:, 0]                                                         # :1288-1289
    return out, logprobs


# ----------------------------------------------------------------------------
# trie-constrained greedy decoding (trie_decoder.py:27-257)
class TokenTrie:
    """trie_decoder.py:224-257: children keyed by token id; a cursor `curr` that search() resets and moves."""

    def __init__(self):
        self.children: List[Dict[int, int]] = [{}]          # node -> {token: child node}; node 0 = root
        self.curr = 0

    @classmethod
    def construct(cls, all_tokens: Sequence[Sequence[int]]) -> "TokenTrie":
        t = cls()
        for ts in all_tokens:
            t.insert(ts)
        return t

    def insert(self, tokens: Sequence[int]) -> None:
        cur = 0
        for tok in tokens:
            nxt = self.children[cur].get(int(tok))
            if nxt is None:
                nxt = len(self.children)
                self.children.append({})
                self.children[cur][int(tok)] = nxt
            cur = nxt

    def get_valid(self, tokens: Sequence[int]) -> List[int]:
        cur = 0
        for tok in tokens:
            cur = self.children[cur].get(int(tok))
            if cur is None:
                return []
        return list(self.children[cur].keys())

    def reset(self) -> None:
        self.curr = 0

    def get_curr_valid(self) -> List[int]:
        return list(self.children[self.curr].keys())

    def move(self, tok: int) -> None:
        assert int(tok) in self.children[self.curr]
        self.curr = self.children[self.curr][int(tok)]

    def csr(self):
        """(child_off int32 [nodes + 1], child_tok int32 [edges], child_node int32 [edges]) -- the device layout of
        gitmi_set_trie; the children of a node in insertion order."""
        off, tok, node = [0], [], []
        for ch in self.children:
            for t, n in ch.items():
                tok.append(t)
                node.append(n)
            off.append(len(tok))
        return (torch.tensor(off, dtype=torch.int32), torch.tensor(tok, dtype=torch.int32),
                torch.tensor(node, dtype=torch.int32))


def search_trie(start: Tensor, step: Callable[[Tensor], Tensor], eos: int, max_steps: int,
                trie: TokenTrie) -> Tuple[Tensor, Tensor]:
    """TrieAutoRegressiveBeamSearch.search (trie_decoder.py:41-218), beam_size = 1, only_return_best=True.  At every
    step the log-probabilities of ROW 0's trie-valid tokens are raised by (max - min + 1) of the step's (modified) logits
    and the trie cursor follows row 0's choice -- so the class is meaningful for batch 1 (with more rows, `trie.move`
    asserts as soon as row 0 has ended while another row has not).

    Returns (predictions int64 [B, len <= max_steps] incl. the start tokens, logprobs fp32 [B] = summed (bonus
    included) log-prob / num_valid); when every first prediction is EOS: ([B, 1], [B, 1]) (:76-83)."""
    trie.reset()
    B, P = start.shape
    preds = start[:, None, :]                                                    # [B, 1, P]
    logits0 = step(start)
    lp0 = torch.log_softmax(logits0, dim=1)
    idx = trie.get_curr_valid()
    lp0[0, idx] += logits0.max() - logits0.min() + 1                             # :64
    last_lp, cls0 = lp0.topk(1)                                                  # :69-71
    trie.move(int(cls0[0, 0]))
    if bool((cls0 == eos).all()):
        return cls0, last_lp
    preds = torch.cat([preds, cls0[:, :, None]], dim=-1)
    V = lp0.shape[1]
    after_end = torch.full((V,), float("-inf"))
    after_end[eos] = 0.0
    while preds.shape[-1] < max_steps:                                           # :106
        last = preds[:, :, -1].reshape(B)
        if bool((last == eos).all()):
            break
        flat = preds.reshape(B, -1)
        logits = step(flat)
        logits = logits.scatter(1, last[:, None], -10000.0)                      # :121
        logits = torch.where((last == eos)[:, None], after_end[None, :], logits)  # :138-142
        lp = torch.log_softmax(logits, dim=1)
        idx = trie.get_curr_valid()
        lp[0, idx] += logits.max() - logits.min() + 1                            # :151
        top_lp, top_cls = lp.topk(1)
        trie.move(int(top_cls[0, 0]))
        last_lp = top_lp + last_lp                                               # :170-176 (beam 1)
        preds = torch.cat([preds, top_cls[:, :, None]], dim=-1)
    best = preds[:, 0, :]
    best_lp = last_lp[:, 0]
    n_valid = (best != eos).sum(dim=-1) + ((best == eos).sum(dim=-1) > 0).long() - P   # :210-213
    return best, best_lp / n_valid.clamp(min=1)


# ----------------------------------------------------------------------------
# sampling branch helpers (decoder.py:1146-1166, 1343-1375)
# ----------------------------------------------------------------------------
def top_k_top_p_filtering(logits: Tensor, top_k: int = 0, top_p: Optional[float] = 1.0,
                          min_tokens_to_keep: int = 1) -> Tensor:
    """decoder.py:1343-1375 restated (returns a new tensor; removed entries are -inf).  Note the order of operations in
    the nucleus part: the removal flags of the first `min_tokens_to_keep` sorted positions are cleared BEFORE the flags
    are shifted right by one, so min_tokens_to_keep + 1 tokens always survive."""
    logits = logits.clone()
    if top_k and top_k > 0:
        k = min(max(top_k, min_tokens_to_keep), logits.shape[-1])
        kth = torch.topk(logits, k)[0][..., -1, None]
        logits[logits < kth] = float("-inf")
    if top_p and top_p < 1.0:
        sorted_logits, sorted_idx = torch.sort(logits, descending=True)
        cum = torch.cumsum(torch.softmax(sorted_logits, dim=-1), dim=-1)
        remove = cum > top_p
        if min_tokens_to_keep > 1:
            remove[..., :min_tokens_to_keep] = False
        remove[..., 1:] = remove[..., :-1].clone()
        remove[..., 0] = False
        logits[remove.scatter(1, sorted_idx, remove)] = float("-inf")
    return logits


def sampling_distribution(logits: Tensor, temperature: float = 1.0, top_k: int = 0,
                          top_p: Optional[float] = 1.0) -> Tensor:
    """What GeneratorWithBeamSearch.search samples from at one step (decoder.py:1146-1158):
    softmax(top_k_top_p_filtering(scores / temperature, min_tokens_to_keep=2)); log-probabilities, -inf = removed."""
    scores = logits / temperature if temperature != 1.0 else logits
    return torch.log_softmax(top_k_top_p_filtering(scores, top_k=top_k, top_p=top_p, min_tokens_to_keep=2), dim=-1)


# ----------------------------------------------------------------------------
# CaptioningModel.forward / infer  (decoder.py:838-877, 977-1011)
# ----------------------------------------------------------------------------
@dataclasses.dataclass(frozen=True)
class SearchConfig:
    kind: str = "greedy"            # "greedy" -> AutoRegressiveBeamSearch, "beam" -> GeneratorWithBeamSearch
    max_steps: int = 20
    beam_size: int = 1
    per_node_beam_size: int = 1
    length_penalty: float = 0.6


GREEDY = SearchConfig()                                          # model.py:27-33 alternative, max_steps=20
BEAM4 = SearchConfig("beam", 20, 4, 2, 0.6)                      # model.py:34-40 with max_steps=20


def caption(cfg: GitConfig, w: Weights, frames: Sequence[Tensor], search: SearchConfig = GREEDY,
            prefix: Optional[Tensor] = None, cached: bool = False,
            feats: Optional[Tensor] = None, trace: Optional[List[Tensor]] = None,
            trie: Optional["TokenTrie"] = None) -> Dict[str, Tensor]:
    """model({'image': ..., 'prefix': ...}) -> {'predictions','logprobs'} exactly as
    CaptioningModel.infer returns them (prefix stripped, decoder.py:1004-1006).  search.kind == "trie": the model's
    decoder is TrieAutoRegressiveBeamSearch(trie=trie) (trie_decoder.py)."""
    if feats is None:
        feats = visual_features(cfg, w, frames)
    B = feats.shape[0]
    if prefix is None:
        start = torch.full((B, 1), cfg.sos, dtype=torch.long)                   # decoder.py:981-983
    else:
        assert prefix.shape[0] == 1 and B == 1, "reference asserts len(prefix)==1 (decoder.py:988)"
        start = prefix.long()
    step = make_step(cfg, w, feats, cached=cached)
    if search.kind == "trie":
        preds, lps = search_trie(start, step, cfg.eos, search.max_steps, trie)
    elif search.kind == "greedy":
        preds, lps = search_autoregressive(start, step, cfg.eos, search.max_steps,
                                           search.beam_size, search.per_node_beam_size, trace=trace)
    else:
        preds, lps = search_generator(start, step, cfg.eos, search.max_steps, search.beam_size,
                                      search.per_node_beam_size, search.length_penalty, trace=trace)
    if prefix is not None:
        preds = preds[:, start.shape[1]:]
    return {"predictions": preds, "logprobs": lps, "visual_features": feats}
