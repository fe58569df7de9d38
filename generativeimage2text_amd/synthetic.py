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
                      successor: float = 0.0) -> Dict[str, torch.Tensor]:
    """eos_bias < 0 keeps captions from ending early so that every caption costs max_steps-1 decode steps
    (the fixed-work protocol of SURVEY.md 8d).

    successor > 0: the WIDE-MARGIN variant of the same weights (tests/golden/full_wide_*: the regime in which "greedy ids
    identical to the reference" is decidable for a 16-bit pipeline).  Plain random-init weights give Gaussian logits over
    30522 tokens, whose top-1 / top-2 gap is below any 16-bit pipeline's logit error somewhere in almost every 19-step
    row; a trained captioner is confident at most steps.  Here the output matrix is untied and carries a successor
    structure, W_out[v] = N(0, .02) + successor * E[perm[v]] (E = word embedding, perm a seeded permutation): after
    token u the logit of perm^-1(u) stands out by ~2 (20 x the bf16 logit error).  The word embedding of [CLS] and
    position 0 are zero, so the hidden state of the first step is what attention reads from the IMAGE: the first token
    is image-dependent, every later one follows the chain of its predecessor -- rows differ, no row loops."""
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
    return sd


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
