"""Random-init GIT weights / images of the right shapes for benchmarking and smoke runs.

No checkpoint or dataset is reachable offline, so throughput is measured on weights drawn with the
reference's init scales (CLIP/model.py:226-228: width**-0.5 embeddings; decoder.py:507-517: N(0, .02))
and on N(0,1) "post-Normalize" images.  Keys follow the reference state dict.
"""
from __future__ import annotations

from typing import Dict, List

import torch

from .configs import GitModelConfig


def random_state_dict(cfg: GitModelConfig, seed: int = 1234, eos_bias: float = -5.0,
                      successor: float = 0.0, stats: str = "init") -> Dict[str, torch.Tensor]:
    """eos_bias < 0 keeps captions from ending early so that every caption costs max_steps-1 decode steps
    (the fixed-work protocol of SURVEY.md 8d).

    successor > 0: the WIDE-MARGIN variant of the same weights (tests/golden/full_wide_*: the regime in which "greedy ids
    identical to the reference" is decidable for a 16-bit pipeline).  Plain random-init weights give Gaussian logits over
    30522 tokens, whose top-1 / top-2 gap is below any 16-bit pipeline's logit error somewhere in almost every 19-step
    row; a trained captioner is confident at most steps.  Here the output matrix is untied and carries a successor
    structure, W_out[v] = N(0, .02) + successor * E[perm[v]] (E = word embedding, perm a seeded permutation): after
    token u the logit of perm^-1(u) stands out by ~2 (20 x the bf16 logit error).  The word embedding of [CLS] and
    position 0 are zero, so the hidden state of the first step is what attention reads from the IMAGE: the first token
    is image-dependent, every later one follows the chain of its predecessor -- rows differ, no row loops.

    stats="trained": the same matrices with the STATISTICS a trained checkpoint has and a fresh initialisation lacks
    (apply_trained_statistics below): LayerNorm gains spread over [0.2, 5], LayerNorm / Linear biases of order 1, a few
    residual-stream channels 100-1000x above the rest, class / positional embeddings well above their init scale."""
    if stats not in ("init", "trained"):
        raise ValueError(f"stats must be 'init' or 'trained', not {stats!r}")
    g = torch.Generator().manual_seed(seed)

    def rn(*shape, std):
        return torch.randn(*shape, generator=g) * std

    sd: Dict[str, torch.Tensor] = {}
    D, F = cfg.vit_width, 4 * cfg.vit_width
    sc = D ** -0.5
    sd["image_encoder.class_embedding"] = rn(D, std=sc)
    sd["image_encoder.positional_embedding"] = rn(cfg.n_tok, D, std=sc)
    sd["image_encoder.conv1.weight"] = rn(D, 3, cfg.patch, cfg.patch, std=(3 * cfg.patch ** 2) ** -0.5)
    for nm in ("ln_pre", "ln_post"):
        sd[f"image_encoder.{nm}.weight"] = torch.ones(D)
        sd[f"image_encoder.{nm}.bias"] = torch.zeros(D)
    for i in range(cfg.vit_layers):
        p = f"image_encoder.transformer.resblocks.{i}."
        sd[p + "attn.in_proj_weight"] = rn(3 * D, D, std=sc)
        sd[p + "attn.in_proj_bias"] = torch.zeros(3 * D)
        sd[p + "attn.out_proj.weight"] = rn(D, D, std=sc)
        sd[p + "attn.out_proj.bias"] = torch.zeros(D)
        sd[p + "mlp.c_fc.weight"] = rn(F, D, std=sc)
        sd[p + "mlp.c_fc.bias"] = torch.zeros(F)
        sd[p + "mlp.c_proj.weight"] = rn(D, F, std=F ** -0.5)
        sd[p + "mlp.c_proj.bias"] = torch.zeros(D)
        for nm in ("ln_1", "ln_2"):
            sd[p + nm + ".weight"] = torch.ones(D)
            sd[p + nm + ".bias"] = torch.zeros(D)
    d, f, V = cfg.dec_hidden, cfg.dec_ffn, cfg.vocab
    sd["textual.visual_projection.0.weight"] = rn(d, D, std=0.02)
    sd["textual.visual_projection.0.bias"] = torch.zeros(d)
    sd["textual.visual_projection.1.weight"] = torch.ones(d)
    sd["textual.visual_projection.1.bias"] = torch.zeros(d)
    sd["textual.embedding.words.weight"] = rn(V, d, std=0.02)
    sd["textual.embedding.positions.weight"] = rn(cfg.max_pos, d, std=0.02)
    sd["textual.embedding.layer_norm.weight"] = torch.ones(d)
    sd["textual.embedding.layer_norm.bias"] = torch.zeros(d)
    for i in range(cfg.dec_layers):
        p = f"textual.transformer.encoder.layer.{i}."
        for nm in ("query", "key", "value"):
            sd[p + f"attention.self.{nm}.weight"] = rn(d, d, std=0.02)
            sd[p + f"attention.self.{nm}.bias"] = torch.zeros(d)
        sd[p + "attention.output.dense.weight"] = rn(d, d, std=0.02)
        sd[p + "attention.output.dense.bias"] = torch.zeros(d)
        sd[p + "intermediate.dense.weight"] = rn(f, d, std=0.02)
        sd[p + "intermediate.dense.bias"] = torch.zeros(f)
        sd[p + "output.dense.weight"] = rn(d, f, std=0.02)
        sd[p + "output.dense.bias"] = torch.zeros(d)
        for nm in ("attention.output.LayerNorm", "output.LayerNorm"):
            sd[p + nm + ".weight"] = torch.ones(d)
            sd[p + nm + ".bias"] = torch.zeros(d)
    ob = torch.zeros(V)
    ob[cfg.eos] = eos_bias
    sd["textual.output.bias"] = ob          # textual.output.weight tied to the word embedding
    if successor != 0.0:
        g2 = torch.Generator().manual_seed(seed + 7919)
        perm = torch.randperm(V, generator=g2)
        boost = successor * sd["textual.embedding.words.weight"][perm]
        boost[perm == cfg.sos] = 0.0        # nothing is "the successor of [CLS]": the first token is read from the image
        boost[cfg.eos] = 0.0                # [SEP] is never a successor: every caption runs max_steps - 1 decode steps
        sd["textual.output.weight"] = torch.randn(V, d, generator=g2) * 0.02 + boost
        sd["textual.embedding.words.weight"][cfg.sos] = 0.0
        sd["textual.embedding.positions.weight"][0] = 0.0
    for i in range(cfg.num_frames):
        sd[f"img_temperal_embedding.{i}"] = rn(1, 1, D, std=0.02)
    if stats == "trained":
        apply_trained_statistics(cfg, sd, seed)
    return sd


# residual-stream channels of the image encoder that carry an outlier, and what puts it there: (block, bias, multiple of the
# stream's typical magnitude ~1).  Trained CLIP ViTs carry a handful of such channels from the first blocks on, two to three
# orders of magnitude above the rest; every later LayerNorm, GEMM and the 16-bit stream rows have to live with them.
OUTLIER_CHANNELS = ((1, "mlp.c_proj.bias", 300.0), (2, "attn.out_proj.bias", -100.0), (4, "mlp.c_proj.bias", 1000.0))


def apply_trained_statistics(cfg: GitModelConfig, sd: Dict[str, torch.Tensor], seed: int) -> None:
    """Rewrite, in place, the parts of a random-init state dict whose init values are degenerate (LayerNorm 1 / 0, Linear
    bias 0, width**-0.5 embeddings) with trained-checkpoint-like statistics; the weight MATRICES are left alone.
      * every LayerNorm: gain log-uniform in [0.2, 5], bias N(0, 1) (decoder post-norm LayerNorms: N(0, 0.5) -- their output IS
        the hidden state the next layer adds to);
      * every Linear bias N(0, 0.1); q / k / v biases N(0, 0.5);
      * OUTLIER_CHANNELS: three residual channels of the image encoder lifted to 100x / 300x / 1000x the stream's magnitude by a
        bias of the block that writes them (the gain of those channels in the following LayerNorms is 0.2: a trained model has
        learned to scale its outliers down, it has not learned to do without them);
      * class embedding N(0, 0.5), positional embedding N(0, 0.1) with a class-row of N(0, 0.5)."""
    import math
    g = torch.Generator().manual_seed(seed + 104729)
    D = cfg.vit_width
    picks = torch.randperm(D, generator=g)[:len(OUTLIER_CHANNELS)].tolist()

    def ln(prefix, n, bias_std=1.0, calm=()):
        gain = torch.exp(torch.rand(n, generator=g) * (math.log(5.0) - math.log(0.2)) + math.log(0.2))
        for ch in calm:
            gain[ch] = 0.2
        sd[prefix + ".weight"] = gain
        sd[prefix + ".bias"] = torch.randn(n, generator=g) * bias_std

    def lin_bias(key, std=0.1):
        sd[key] = torch.randn(sd[key].shape, generator=g) * std

    sd["image_encoder.class_embedding"] = torch.randn(D, generator=g) * 0.5
    pos = torch.randn(cfg.n_tok, D, generator=g) * 0.1
    pos[0] = torch.randn(D, generator=g) * 0.5
    sd["image_encoder.positional_embedding"] = pos
    ln("image_encoder.ln_pre", D)
    live = []                                           # outlier channels present in the stream so far
    for i in range(cfg.vit_layers):
        p = f"image_encoder.transformer.resblocks.{i}."
        ln(p + "ln_1", D, calm=live)
        lin_bias(p + "attn.in_proj_bias", 0.5)
        lin_bias(p + "attn.out_proj.bias")
        for (blk, key, mult), ch in zip(OUTLIER_CHANNELS, picks):
            if blk == i and key.startswith("attn"):
                sd[p + key][ch] = mult
                live = live + [ch]
        ln(p + "ln_2", D, calm=live)
        lin_bias(p + "mlp.c_fc.bias")
        lin_bias(p + "mlp.c_proj.bias")
        for (blk, key, mult), ch in zip(OUTLIER_CHANNELS, picks):
            if blk == i and key.startswith("mlp"):
                sd[p + key][ch] = mult
                live = live + [ch]
    ln("image_encoder.ln_post", D, calm=live)
    d = cfg.dec_hidden
    lin_bias("textual.visual_projection.0.bias")
    ln("textual.visual_projection.1", d, bias_std=0.5)
    ln("textual.embedding.layer_norm", d, bias_std=0.5)
    for i in range(cfg.dec_layers):
        p = f"textual.transformer.encoder.layer.{i}."
        for nm in ("query", "key", "value"):
            lin_bias(p + f"attention.self.{nm}.bias", 0.5)
        lin_bias(p + "attention.output.dense.bias")
        lin_bias(p + "intermediate.dense.bias")
        lin_bias(p + "output.dense.bias")
        for nm in ("attention.output.LayerNorm", "output.LayerNorm"):
            ln(p + nm, d, bias_std=0.5)


def random_frames(cfg: GitModelConfig, batch: int, frames: int = 1, seed: int = 0, device="cuda") -> List[torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(batch, 3, cfg.image_size, cfg.image_size, generator=g).to(device) for _ in range(frames)]


def seeded_images(cfg: GitModelConfig, image_seeds, device="cuda", frames: int = 1) -> List[torch.Tensor]:
    """`frames` batches whose image i is drawn from its OWN generator (seed image_seeds[i]; the frames of a clip one after
    the other from it): a fixture can name the images it kept (tests/golden/full_wide_*.npz `image_seeds`) without storing
    them."""
    per_frame = [[] for _ in range(frames)]
    for sd in image_seeds:
        g = torch.Generator().manual_seed(7_000_000 + int(sd))
        for f in range(frames):
            per_frame[f].append(torch.randn(3, cfg.image_size, cfg.image_size, generator=g))
    return [torch.stack(imgs).to(device) for imgs in per_frame]
