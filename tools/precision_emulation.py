#!/usr/bin/env python
"""CPU study (no GPU): which rounding sites of the 16-bit engine modes own the teacher-forced logit error, per weight family?

VERDICT r05 "What's weak" 1: the 16-bit modes are 6-7x worse on the oracle's weights (perturbed LayerNorms, width^-0.5
decoder matrices, untied output) than on the benchmark's (identity LayerNorms, N(0, .02) decoder) while the logit span grows
1.28x; prime suspect was the folded LayerNorm of the decode chain.  This emulates the engine's arithmetic on the CPU -- every
MFMA operand rounded to the operand format, fp32 accumulation / statistics / softmax, the residual-stream storage types of
DESIGN.md section 3, the folded-LayerNorm algebra of kernels_dgemm.hip -- with the rounding switched on ONE stage at a time:

    V  image encoder            P  visual projection + decoder prefill (image rows, K/V cache storage)
    C  decode chain (text rows) H  vocabulary head
    C sub-sites: cq QKV operands, ca attention (q, text K/V, P, context), co out-proj operand, cf FFN1 operands + GELU
                 output, c2 FFN2 operands;   fold = the chain's LayerNorms in the folded form (raw x as the operand)

    python tools/precision_emulation.py [--weights bench|oracle|trained] [--fmt bf16|f16] [--batch 4]

A study tool: the oracle is the fp32 yardstick here (as in residual_precision_study.py); nothing in the product imports it.
"""
import argparse
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import git_oracle as O


def rounder(fmt):
    if fmt is None:
        return lambda t: t
    dt = {"bf16": torch.bfloat16, "f16": torch.float16}[fmt]
    return lambda t: t.to(dt).to(torch.float32)


ident = rounder(None)
f16 = rounder("f16")


class Emu:
    """sites: dict site -> format name or None"""

    def __init__(self, cfg, w, sites, fold=True, stream16=True):
        self.cfg, self.w = cfg, w
        self.r = {k: rounder(v) for k, v in sites.items()}
        self.on = {k: v is not None for k, v in sites.items()}
        self.fold = fold
        self.stream16 = stream16

    def lin(self, site, x, W, b):
        r = self.r[site]
        return r(x) @ r(W).t() + (0 if b is None else b)

    # ---- image encoder --------------------------------------------------------------------------------------------
    def vit(self, images):
        cfg, w, r = self.cfg, self.w, self.r["V"]
        rs = f16 if (self.on["V"] and self.stream16) else ident
        b = images.shape[0]
        p, D = cfg.patch, cfg.vit_width
        gh, gw = images.shape[2] // p, images.shape[3] // p
        patches = images.reshape(b, 3, gh, p, gw, p).permute(0, 2, 4, 1, 3, 5).reshape(b, gh * gw, 3 * p * p)
        x = self.lin("V", patches, w["image_encoder.conv1.weight"].reshape(D, 3 * p * p), None)
        cls = w["image_encoder.class_embedding"].expand(b, 1, D)
        x = torch.cat([cls, x], dim=1) + O.vit_positional(cfg, w, gh, gw)
        x = rs(O._layer_norm(x, w["image_encoder.ln_pre.weight"], w["image_encoder.ln_pre.bias"], 1e-5))
        H, hd = cfg.vit_heads, cfg.vit_width // cfg.vit_heads
        for i in range(cfg.vit_layers):
            q_ = f"image_encoder.transformer.resblocks.{i}."
            h = O._layer_norm(x, w[q_ + "ln_1.weight"], w[q_ + "ln_1.bias"], 1e-5)
            qkv = r(self.lin("V", h, w[q_ + "attn.in_proj_weight"], w[q_ + "attn.in_proj_bias"]))
            q, k, v = (O._split_heads(t, H) for t in qkv.chunk(3, dim=-1))
            pr = torch.softmax((q @ k.transpose(-1, -2)) * hd ** -0.5, dim=-1)
            att = r(r(pr) @ v)
            x = rs(x + self.lin("V", O._merge_heads(att), w[q_ + "attn.out_proj.weight"], w[q_ + "attn.out_proj.bias"]))
            h = O._layer_norm(x, w[q_ + "ln_2.weight"], w[q_ + "ln_2.bias"], 1e-5)
            u = self.lin("V", h, w[q_ + "mlp.c_fc.weight"], w[q_ + "mlp.c_fc.bias"])
            u = r(u * torch.sigmoid(1.702 * u))
            x = rs(x + self.lin("V", u, w[q_ + "mlp.c_proj.weight"], w[q_ + "mlp.c_proj.bias"]))
        self.vit_stream_max = float(x.abs().max())
        return O._layer_norm(x, w["image_encoder.ln_post.weight"], w["image_encoder.ln_post.bias"], 1e-5)

    # ---- decoder prefill: image rows, K / V per layer -------------------------------------------------------------------
    def prefill(self, feats):
        cfg, w, r = self.cfg, self.w, self.r["P"]
        rs = f16 if (self.on["P"] and self.stream16) else ident
        P = "textual.visual_projection."
        y = rs(self.lin("P", feats, w[P + "0.weight"], w[P + "0.bias"]))
        h = O._layer_norm(y, w[P + "1.weight"], w[P + "1.bias"], 1e-5)
        H, hd = cfg.dec_heads, cfg.dec_hidden // cfg.dec_heads
        out = []
        for i in range(cfg.dec_layers):
            p = f"textual.transformer.encoder.layer.{i}."
            h_res = rs(h)
            q = r(self.lin("P", h, w[p + "attention.self.query.weight"], w[p + "attention.self.query.bias"]))
            k = r(self.lin("P", h, w[p + "attention.self.key.weight"], w[p + "attention.self.key.bias"]))
            v = r(self.lin("P", h, w[p + "attention.self.value.weight"], w[p + "attention.self.value.bias"]))
            out.append((k, v))
            if i + 1 == cfg.dec_layers:
                break
            qh, kh, vh = (O._split_heads(t, H) for t in (q, k, v))
            pr = torch.softmax((qh @ kh.transpose(-1, -2)) * hd ** -0.5, dim=-1)
            ctx = r(O._merge_heads(r(pr) @ vh))
            y = rs(self.lin("P", ctx, w[p + "attention.output.dense.weight"], w[p + "attention.output.dense.bias"]) + h_res)
            a = O._layer_norm(y, w[p + "attention.output.LayerNorm.weight"], w[p + "attention.output.LayerNorm.bias"], 1e-12)
            a_res = rs(a)
            u = r(O._gelu_erf(self.lin("P", a, w[p + "intermediate.dense.weight"], w[p + "intermediate.dense.bias"])))
            y = rs(self.lin("P", u, w[p + "output.dense.weight"], w[p + "output.dense.bias"]) + a_res)
            h = O._layer_norm(y, w[p + "output.LayerNorm.weight"], w[p + "output.LayerNorm.bias"], 1e-12)
        return out

    # ---- LayerNorm + Linear as the chain computes it ---------------------------------------------------------------------
    def ln_lin(self, site, x, gamma, beta, eps, W, b):
        """x: raw pre-LayerNorm sum (fp32).  fold: rstd * (r(x) r(W.gamma)^T - mean * colsum(r(W.gamma))) + (beta W^T + b), statistics
        from the fp32 x (kernels_dgemm.hip EPI 0); else: the LayerNorm applied in fp32, then rounded (centred operand)."""
        r = self.r[site]
        if not self.fold:
            return self.lin(site, O._layer_norm(x, gamma, beta, eps), W, b)
        mean = x.mean(dim=-1, keepdim=True)
        var = (x * x).mean(dim=-1, keepdim=True) - mean * mean
        rstd = torch.rsqrt(var.clamp_min(0) + eps)
        Wg = r(W * gamma[None, :])
        acc = r(x) @ Wg.t()
        return rstd * (acc - mean * Wg.sum(dim=1)[None, :]) + (beta @ W.t() + b)

    # ---- decode chain over the text rows (teacher-forced, all positions at once; causal) --------------------------------------
    def chain(self, kv_img, tokens):
        cfg, w = self.cfg, self.w
        ca = self.r["ca"]
        t = tokens.shape[1]
        e = w["textual.embedding.words.weight"][tokens] + w["textual.embedding.positions.weight"][:t]
        g_prev, b_prev, eps_prev = w["textual.embedding.layer_norm.weight"], w["textual.embedding.layer_norm.bias"], 1e-8
        x = e                                                       # raw sum; its LayerNorm is (g_prev, b_prev, eps_prev)
        H, hd = cfg.dec_heads, cfg.dec_hidden // cfg.dec_heads
        causal = torch.triu(torch.full((t, t), float("-inf")), diagonal=1)
        n_img = kv_img[0][0].shape[1]
        mask = torch.cat([torch.zeros(t, n_img), causal], dim=1)
        for i in range(cfg.dec_layers):
            p = f"textual.transformer.encoder.layer.{i}."
            Wqkv = torch.cat([w[p + f"attention.self.{n}.weight"] for n in ("query", "key", "value")], dim=0)
            bqkv = torch.cat([w[p + f"attention.self.{n}.bias"] for n in ("query", "key", "value")], dim=0)
            qkv = self.ln_lin("cq", x, g_prev, b_prev, eps_prev, Wqkv, bqkv)
            q, k, v = (ca(u) for u in qkv.chunk(3, dim=-1))
            kk = torch.cat([kv_img[i][0], k], dim=1)
            vv = torch.cat([kv_img[i][1], v], dim=1)
            qh, kh, vh = (O._split_heads(u, H) for u in (q, kk, vv))
            pr = torch.softmax((qh @ kh.transpose(-1, -2)) * hd ** -0.5 + mask, dim=-1)
            ctx = ca(O._merge_heads(ca(pr) @ vh))
            res = O._layer_norm(x, g_prev, b_prev, eps_prev)        # rebuilt from the fp32 raw sum and its statistics
            y = self.lin("co", ctx, w[p + "attention.output.dense.weight"], w[p + "attention.output.dense.bias"]) + res
            ga, ba = w[p + "attention.output.LayerNorm.weight"], w[p + "attention.output.LayerNorm.bias"]
            u = self.r["cf"](O._gelu_erf(self.ln_lin("cf", y, ga, ba, 1e-12, w[p + "intermediate.dense.weight"],
                                                    w[p + "intermediate.dense.bias"])))
            res = O._layer_norm(y, ga, ba, 1e-12)
            x = self.lin("c2", u, w[p + "output.dense.weight"], w[p + "output.dense.bias"]) + res
            g_prev, b_prev, eps_prev = w[p + "output.LayerNorm.weight"], w[p + "output.LayerNorm.bias"], 1e-12
        self.last_x = x
        return self.ln_lin("H", x[:, -1], g_prev, b_prev, eps_prev, w["textual.output.weight"], w["textual.output.bias"])

    def logits(self, images, tokens):
        return self.chain(self.prefill(self.vit(images)), tokens)


SITES = ("V", "P", "cq", "ca", "co", "cf", "c2", "H")
GROUPS = {"all": SITES, "V": ("V",), "P": ("P",), "C": ("cq", "ca", "co", "cf", "c2"), "H": ("H",),
          "C+H": ("cq", "ca", "co", "cf", "c2", "H"), "P+C+H": SITES[1:],
          "cq": ("cq",), "ca": ("ca",), "co": ("co",), "cf": ("cf",), "c2": ("c2",), "none": ()}


def weights_for(kind, cfg_name):
    cfg = O.CONFIGS[cfg_name]
    if kind == "oracle":
        return cfg, O.make_weights(cfg, seed=1240, tie_output=False, successor=1.0)
    from generativeimage2text_amd.configs import config_for_model
    from generativeimage2text_amd.synthetic import random_state_dict
    kw = {} if kind == "bench" else {"stats": kind}
    w = {k: v.float() for k, v in random_state_dict(config_for_model(cfg_name), seed=1234, eos_bias=-5.0, **kw).items()}
    w.setdefault("textual.output.weight", w["textual.embedding.words.weight"])
    return cfg, w


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--weights", default="bench")
    ap.add_argument("--model", default="GIT_BASE")
    ap.add_argument("--fmt", default="bf16")
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--tokens", type=int, default=5)
    ap.add_argument("--groups", default="all,V,P,C,H,cq,ca,co,cf,c2")
    ap.add_argument("--mixed", default=None, help="site=fmt,... one extra row with per-site formats, e.g. V=bf16,P=f16,...")
    a = ap.parse_args()
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    cfg, w = weights_for(a.weights, a.model)
    g = torch.Generator().manual_seed(0)
    images = torch.randn(a.batch, 3, cfg.image_size, cfg.image_size, generator=g)
    g = torch.Generator().manual_seed(5)
    tokens = torch.randint(0, cfg.vocab, (a.batch, a.tokens), generator=g)
    tokens[:, 0] = cfg.sos
    with torch.no_grad():
        ref = O.textual_logits_full(cfg, w, O.vit_forward(cfg, w, images), tokens)[:, -1]
        base = Emu(cfg, w, {s: None for s in SITES}, fold=False).logits(images, tokens)
        span = float(ref.max() - ref.min())
        print(f"# {a.model} weights={a.weights} fmt={a.fmt} B={a.batch} t={a.tokens}: logit span {span:.3f}, std {ref.std().item():.3f}; "
              f"emulator with no rounding vs oracle: {(base - ref).abs().max().item():.2e}")
        print("%-28s %-6s %10s %10s %12s" % ("16-bit sites", "fold", "max|err|", "rms err", "max / span"))

        def row(tag, sites, fold):
            e = Emu(cfg, w, sites, fold=fold)
            d = e.logits(images, tokens) - ref
            mean_ratio = ""
            if tag == "all" and fold:
                x = e.last_x
                mean_ratio = "   (chain rows: |mean|/std max %.3f; ViT stream max |x| %.1f)" % (
                    float((x.mean(-1).abs() / x.std(-1)).max()), e.vit_stream_max)
            print("%-28s %-6s %10.5f %10.5f %12.2e%s" % (tag, "yes" if fold else "no", d.abs().max().item(),
                                                         d.pow(2).mean().sqrt().item(), d.abs().max().item() / span, mean_ratio),
                  flush=True)

        for grp in a.groups.split(","):
            sites = {s: (a.fmt if s in GROUPS[grp] else None) for s in SITES}
            folds = (True, False) if any(s in GROUPS[grp] for s in ("cq", "cf", "H")) else (True,)
            for fold in folds:
                row(grp, sites, fold)
        if a.mixed:
            sites = {s: None for s in SITES}
            for kv in a.mixed.split(","):
                k, v = kv.split("=")
                for s in GROUPS.get(k, (k,)):
                    sites[s] = None if v in ("f32", "none") else v
            row("mixed " + a.mixed, sites, True)
            row("mixed " + a.mixed, sites, False)


if __name__ == "__main__":
    main()
