#!/usr/bin/env python
"""CPU study (no GPU): what would a bf16 RESIDUAL STREAM cost the image encoder in feature error?

The engine keeps the ViT residual stream in fp32; its read-modify-write is ~0.4 ms of a 5.3 ms
encode + prefill (section 4).  Halving it means storing the stream in bf16.  This emulates the engine's bf16 arithmetic
on the CPU -- every GEMM operand (activations and weights) rounded to bf16, fp32 accumulation, fp32 LayerNorm /
softmax statistics, bf16 attention operands -- with the stream kept in fp32, rounded to fp16 or rounded to bf16
after every residual add, and reports the error of the ln_post features against the fp32 oracle for seeded GIT_BASE
weights.  The engine's measured error of the fp32-stream variant is ~0.02 abs (tests' bound: 0.05).

    python tools/residual_precision_study.py [--model GIT_BASE] [--batch 2]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import git_oracle as O          # a study tool, not the product: the oracle is the fp32 yardstick here


def bf(x):
    return x.to(torch.bfloat16).to(torch.float32)


def lin(x, W, b):
    return bf(x) @ bf(W).t() + (0 if b is None else b)


def f16(x):
    return x.to(torch.float16).to(torch.float32)


def vit_bf16(cfg, w, images, stream: str):
    rs = {"fp32": (lambda t: t), "bf16": bf, "fp16": f16}[stream]
    b = images.shape[0]
    p, D = cfg.patch, cfg.vit_width
    gh, gw = images.shape[2] // p, images.shape[3] // p
    patches = images.reshape(b, 3, gh, p, gw, p).permute(0, 2, 4, 1, 3, 5).reshape(b, gh * gw, 3 * p * p)
    x = lin(patches, w["image_encoder.conv1.weight"].reshape(D, 3 * p * p), None)
    cls = w["image_encoder.class_embedding"].expand(b, 1, D)
    x = torch.cat([cls, x], dim=1) + O.vit_positional(cfg, w, gh, gw)
    x = rs(O._layer_norm(x, w["image_encoder.ln_pre.weight"], w["image_encoder.ln_pre.bias"], 1e-5))
    H, hd = cfg.vit_heads, cfg.vit_width // cfg.vit_heads
    for i in range(cfg.vit_layers):
        q_ = f"image_encoder.transformer.resblocks.{i}."
        h = O._layer_norm(x, w[q_ + "ln_1.weight"], w[q_ + "ln_1.bias"], 1e-5)
        qkv = bf(lin(h, w[q_ + "attn.in_proj_weight"], w[q_ + "attn.in_proj_bias"]))          # bf16 QKV tensor
        q, k, v = (O._split_heads(t, H) for t in qkv.chunk(3, dim=-1))
        pr = torch.softmax((q @ k.transpose(-1, -2)) * hd ** -0.5, dim=-1)
        att = bf(bf(pr) @ v)                                                                   # bf16 P, bf16 context
        x = rs(x + lin(O._merge_heads(att), w[q_ + "attn.out_proj.weight"], w[q_ + "attn.out_proj.bias"]))
        h = O._layer_norm(x, w[q_ + "ln_2.weight"], w[q_ + "ln_2.bias"], 1e-5)
        u = lin(h, w[q_ + "mlp.c_fc.weight"], w[q_ + "mlp.c_fc.bias"])
        u = bf(u * torch.sigmoid(1.702 * u))
        x = rs(x + lin(u, w[q_ + "mlp.c_proj.weight"], w[q_ + "mlp.c_proj.bias"]))
    global mx
    mx = float(x.abs().max())
    return O._layer_norm(x, w["image_encoder.ln_post.weight"], w["image_encoder.ln_post.bias"], 1e-5)


mx = 0.0


def prefill_kv_bf16(cfg, w, feats, stream: str):
    """Image rows through the decoder (prefill) with the engine's bf16 arithmetic; the post-norm residual (the
    LayerNorm output) and the pre-norm sums stored as `stream`; "operand" = no separate residual copy at all: the
    residual is read back from the bf16 operand copy of the LayerNorm output.  -> per layer (K, V) as bf16-rounded
    tensors (what the decode attention reads)."""
    rs = {"fp32": (lambda t: t), "fp16": f16, "bf16": bf, "operand": bf}[stream]
    ys = rs if stream != "operand" else f16            # the pre-norm sum keeps a stream of its own (fp16 there)
    P = "textual.visual_projection."
    y = ys(lin(feats, w[P + "0.weight"], w[P + "0.bias"]))
    h = O._layer_norm(y, w[P + "1.weight"], w[P + "1.bias"], 1e-5)
    H, hd = cfg.dec_heads, cfg.dec_hidden // cfg.dec_heads
    out = []
    for i in range(cfg.dec_layers):
        p = f"textual.transformer.encoder.layer.{i}."
        h_res = rs(h)                                  # residual copy of the hidden state
        q = bf(lin(h, w[p + "attention.self.query.weight"], w[p + "attention.self.query.bias"]))
        k = bf(lin(h, w[p + "attention.self.key.weight"], w[p + "attention.self.key.bias"]))
        v = bf(lin(h, w[p + "attention.self.value.weight"], w[p + "attention.self.value.bias"]))
        out.append((k, v))
        if i + 1 == cfg.dec_layers:
            break
        qh, kh, vh = (O._split_heads(t, H) for t in (q, k, v))
        pr = torch.softmax((qh @ kh.transpose(-1, -2)) * hd ** -0.5, dim=-1)
        ctx = bf(O._merge_heads(bf(pr) @ vh))
        y = ys(lin(ctx, w[p + "attention.output.dense.weight"], w[p + "attention.output.dense.bias"]) + h_res)
        a = O._layer_norm(y, w[p + "attention.output.LayerNorm.weight"], w[p + "attention.output.LayerNorm.bias"], 1e-12)
        a_res = rs(a)
        u = bf(O._gelu_erf(lin(a, w[p + "intermediate.dense.weight"], w[p + "intermediate.dense.bias"])))
        y = ys(lin(u, w[p + "output.dense.weight"], w[p + "output.dense.bias"]) + a_res)
        h = O._layer_norm(y, w[p + "output.LayerNorm.weight"], w[p + "output.LayerNorm.bias"], 1e-12)
    return out


def prefill_kv_ref(cfg, w, feats):
    x = O.project_visual(cfg, w, feats)
    out = []
    for i in range(cfg.dec_layers):
        p = f"textual.transformer.encoder.layer.{i}."
        out.append((O._affine(x, w[p + "attention.self.key.weight"], w[p + "attention.self.key.bias"]),
                    O._affine(x, w[p + "attention.self.value.weight"], w[p + "attention.self.value.bias"])))
        if i + 1 < cfg.dec_layers:
            x = O.bert_layer(cfg, w, i, x, x, None)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="GIT_BASE")
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--seeds", type=int, default=2)
    a = ap.parse_args()
    cfg = O.CONFIGS[a.model]
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    print(f"# {a.model}: ln_post features [B={a.batch}, {cfg.n_tok if hasattr(cfg, 'n_tok') else ''} tokens, {cfg.vit_width}], error vs the fp32 oracle")
    print("seed | stream | max abs err | rms err | feature rms")
    for seed in range(a.seeds):
        w = O.make_weights(cfg, seed=1234 + seed)
        img = O.make_images(cfg, a.batch, 1, seed=seed)[0]
        with torch.no_grad():
            ref = O.vit_forward(cfg, w, img)
            for name in ("fp32", "fp16", "bf16"):
                out = vit_bf16(cfg, w, img, name)
                err = (out - ref).abs()
                print(f"{seed:4d} | {name:6s} | {err.max().item():.4f}      | {err.pow(2).mean().sqrt().item():.5f} | {ref.pow(2).mean().sqrt().item():.3f}"
                      + (f" | stream max |x| {mx:.1f}" if name == "fp32" else ""))
                if name == "fp32":
                    feats_eng = out                    # the features the engine's prefill starts from
            # prefill: image K / V of every decoder layer vs the fp32 oracle's, by how the prefill's streams are stored
            kv_ref = prefill_kv_ref(cfg, w, ref)
            print("     prefill stream | max |dK| | max |dV| | rms dK  | rms dV   (over all decoder layers; K rms %.3f)" %
                  kv_ref[0][0].pow(2).mean().sqrt().item())
            for name in ("fp32", "fp16", "bf16", "operand"):
                kv = prefill_kv_bf16(cfg, w, feats_eng, name)
                dk = torch.cat([(a[0] - b[0]).flatten() for a, b in zip(kv, kv_ref)])
                dv = torch.cat([(a[1] - b[1]).flatten() for a, b in zip(kv, kv_ref)])
                print(f"     {name:14s} | {dk.abs().max().item():.4f}   | {dv.abs().max().item():.4f}   | "
                      f"{dk.pow(2).mean().sqrt().item():.5f} | {dv.pow(2).mean().sqrt().item():.5f}")


if __name__ == "__main__":
    main()
