#!/usr/bin/env python
"""CPU emulation of the HIP path's rounding points (16-bit MFMA operands, fp32 accumulate, fp32 residual stream) with per-stage
toggles: which rounding stage costs how much of the 1e-3 output budget?  TEST / DESIGN tooling -- never imported by the product.

  python tools/precision_study.py [--backbone dinov2_vitl14] [--img 448] [--dtype f16] [--persons 8] [--seed 22]

Prints, per configuration, the relative L2 error of the backbone features and of every north-star output (heads evaluated by the
CPU oracle in fp32 from the emulated features, plus the heads' own 16-bit operands when `heads` is on)."""
from __future__ import annotations

import argparse
import math
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import synthetic  # noqa: E402
from oracle.multihmr_ref import OracleModel  # noqa: E402
from oracle import dinov2_ref  # noqa: E402

QSCALE = 0.125 * math.log2(math.e)
WS = ("w_qk", "w_v", "w_proj", "w_fc1", "w_fc2")
ALL = WS + ("xn", "q", "k", "v", "p", "att", "hid", "gelu", "patch")
# "lnfold" (not part of ALL): LayerNorm folded into the consuming GEMM -- the operand is the RAW residual stream rounded to 16 bits,
# the LayerNorm weight is folded into W before its rounding, and y = rstd * (x16 . W'^T - mean * rowsum(W')) + (b + W . b_ln)


def gelu_fast(x):
    """csrc/mhmr_common.h gelu_fast (round 6: the tail as exp2 of a degree-5 polynomial)"""
    ax = x.abs()
    p = torch.full_like(ax, -0.0004733092791866511)
    for c in (0.007084557320922613, -0.051827382296323776, -0.4599924385547638, -1.1507878303527832, -1.000037670135498):
        p = p * ax + c
    return torch.clamp(x, min=0) - ax * torch.exp2(p)


def ln_folded(t, norm, W, bias, parts, tdt):
    """LayerNorm folded into the consuming linear (see ALL above).  parts: ((round this row block?, rows), ...)."""
    mean = t.mean(-1, keepdim=True)
    rstd = torch.rsqrt(t.var(-1, unbiased=False, keepdim=True) + norm.eps)
    Wf = W * norm.weight[None, :]
    rows, r0 = [], 0
    for rnd, n in parts:
        blk = Wf[r0:r0 + n]
        rows.append(blk.to(tdt).float() if rnd else blk)
        r0 += n
    Wr = torch.cat(rows, 0)
    s = Wr.sum(1)
    b2 = bias + W @ norm.bias
    x16 = t.to(tdt).float()
    return rstd * (x16 @ Wr.T - mean * s) + b2


def emulate_vit(vit, x, on, tdt, blocks_on=None, exempt=None):
    """exempt = (keys, blocks): in those blocks those rounding points are switched off (e.g. the weights of the last 8 blocks carry a
    low half)."""
    r = lambda t, key: t.to(tdt).float() if key in on else t
    G = x.shape[-1] // 14
    B = x.shape[0]
    C = vit.embed_dim
    H = vit.num_heads
    # patch embed: im2col operand and weight in 16 bits
    pw = vit.patch_embed.proj.weight.reshape(C, -1)
    cols = F.unfold(x, kernel_size=14, stride=14).transpose(1, 2)                   # [B,N,588] (c,py,px)
    tok = r(cols, "patch") @ r(pw, "patch").T + vit.patch_embed.proj.bias
    t = torch.cat((vit.cls_token.expand(B, -1, -1), tok), dim=1) + dinov2_ref.interpolate_pos_embed(vit.pos_embed, G)
    T = t.shape[1]
    for bi, blk in enumerate(vit.blocks):
        act = on if (blocks_on is None or bi in blocks_on) else ()
        if exempt is not None and bi in exempt[1]:
            act = set(act) - set(exempt[0])
        rr = lambda u, key: u.to(tdt).float() if key in act else u
        wqkv = blk.attn.qkv.weight
        if "lnfold" in act:
            qkv = ln_folded(t, blk.norm1, wqkv, blk.attn.qkv.bias, (("w_qk" in act, 2 * C), ("w_v" in act, C)), tdt)
        else:
            xn = rr(blk.norm1(t), "xn")
            w_eff = torch.cat((rr(wqkv[:2 * C], "w_qk"), rr(wqkv[2 * C:], "w_v")), 0)
            qkv = xn @ w_eff.T + blk.attn.qkv.bias
        q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
        q = rr(q * QSCALE, "q").reshape(B, T, H, 64).permute(0, 2, 1, 3)
        k = rr(k, "k").reshape(B, T, H, 64).permute(0, 2, 1, 3)
        v = rr(v, "v").reshape(B, T, H, 64).permute(0, 2, 1, 3)
        s = q @ k.transpose(-1, -2)
        p = torch.exp2(s - s.amax(dim=-1, keepdim=True))
        o = (rr(p, "p") @ v) / p.sum(dim=-1, keepdim=True)
        att = rr(o.permute(0, 2, 1, 3).reshape(B, T, C), "att")
        t = t + blk.ls1.gamma * (att @ rr(blk.attn.proj.weight, "w_proj").T + blk.attn.proj.bias)
        if "lnfold" in act:
            h = ln_folded(t, blk.norm2, blk.mlp.fc1.weight, blk.mlp.fc1.bias, (("w_fc1" in act, 4 * C),), tdt)
        else:
            xn = rr(blk.norm2(t), "xn")
            h = xn @ rr(blk.mlp.fc1.weight, "w_fc1").T + blk.mlp.fc1.bias
        h = gelu_fast(h) if "gelu" in act else F.gelu(h)
        h = rr(h, "hid")
        t = t + blk.ls2.gamma * (h @ rr(blk.mlp.fc2.weight, "w_fc2").T + blk.mlp.fc2.bias)
    return vit.norm(t)[:, 1:]


def heads_16bit(tdt):
    """Context manager: the heads' two 16-bit GEMMs as the HIP path runs them -- mlp_classif.0 on the 16-bit features and the HPH
    cross-attention to_kv on the 16-bit context (features | camera embedding | cross_values) with 16-bit weights."""
    import contextlib
    from oracle import multihmr_ref as M

    @contextlib.contextmanager
    def cm():
        orig_ca, orig_mlp2 = M.cross_attention, M.mlp2

        def ca(sd, p, x, context, mask, heads):
            sd2 = dict(sd)
            sd2[p + "to_kv.weight"] = sd[p + "to_kv.weight"].to(tdt).float()
            return orig_ca(sd2, p, x, context.to(tdt).float(), mask, heads)

        def mlp2(sd, prefix, x):
            if prefix != "mlp_classif":
                return orig_mlp2(sd, prefix, x)
            h = F.relu(F.linear(x.to(tdt).float(), sd[prefix + ".0.weight"].to(tdt).float(), sd[prefix + ".0.bias"])).to(tdt).float()
            return F.linear(h, sd[prefix + ".2.weight"], sd[prefix + ".2.bias"])
        M.cross_attention, M.mlp2 = ca, mlp2
        try:
            yield
        finally:
            M.cross_attention, M.mlp2 = orig_ca, orig_mlp2
    return cm()


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backbone", default="dinov2_vitl14")
    ap.add_argument("--img", type=int, default=448)
    ap.add_argument("--dtype", default="f16")
    ap.add_argument("--persons", type=int, default=8)
    ap.add_argument("--seed", type=int, default=22)
    ap.add_argument("--golden-case", default="", help="take backbone / size / seed / inputs from tests/golden/make_golden.py CASES")
    ap.add_argument("--configs", default="all,none,w,xn,qkv,p,att,hid,gelu,first_half,second_half,all-w,all-xn,all-hid")
    a = ap.parse_args()
    if a.golden_case:
        sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
        import make_golden
        c = make_golden.CASES[a.golden_case]
        a.backbone, a.img, a.seed = c["backbone"], c["img_size"], c["seed"]
    tdt = torch.float16 if a.dtype == "f16" else torch.bfloat16
    torch.manual_seed(0)
    smplx_data, mean_params = synthetic.make_smplx_data(0), synthetic.make_mean_params(0)
    sd = synthetic.make_state_dict(a.backbone, a.img, seed=a.seed, mean_params=mean_params)
    if a.golden_case:      # (incl. the hostile weight statistics of the *_hostile_* cases)
        sd = make_golden.case_state_dict(make_golden.CASES[a.golden_case])
    ref = OracleModel(sd, smplx_data, backbone=a.backbone, img_size=a.img)
    g = torch.Generator().manual_seed(1000 + a.seed)
    x = torch.randn(1, 3, a.img, a.img, generator=g)
    K = synthetic.get_camera_K(a.img, 1)
    idx = synthetic.make_pinned_idx(1, a.img // 14, a.persons, seed=a.seed)
    if a.golden_case:      # the inputs of a tests/golden case (same seeds for weights, image and pinned cells)
        sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
        import make_golden
        cfg = make_golden.CASES[a.golden_case]
        x, K, idx = make_golden.case_inputs(cfg)
    keys = ["scores", "offset", "dist", "shape", "expression", "rotmat", "transl", "v3d"]
    with torch.no_grad():
        t0 = time.time()
        z0 = ref.backbone(x)
        out0 = ref.forward(x, idx=idx, K=K, is_training=True, z=z0)
        print(f"exact forward {time.time() - t0:.1f} s", flush=True)
        L = len(ref.vit.blocks)
        for cfg in a.configs.split(","):
            blocks_on = None
            lnfold = cfg.endswith("@lnfold")
            if lnfold:
                cfg = cfg[:-7]
            exempt, label = None, cfg
            if "@L" in cfg:           # all-w_proj+w_v@L12-23 : those weights exact in blocks 12..23 only
                cfg, rng = cfg.split("@L")
                lo, hi = (int(v) for v in rng.split("-"))
                assert cfg.startswith("all-")
                exempt = (set(cfg[4:].replace("w", "+".join(WS)).split("+")) if cfg[4:] == "w" else set(cfg[4:].split("+")), set(range(lo, hi + 1)))
                cfg = "all"
            if cfg == "all":
                on = set(ALL)
            elif cfg == "none":
                on = set()
            elif cfg == "qkv":
                on = {"q", "k", "v"}
            elif cfg == "first_half":
                on, blocks_on = set(ALL), set(range(L // 2))
            elif cfg == "second_half":
                on, blocks_on = set(ALL), set(range(L // 2, L))
            elif cfg == "w":
                on = set(WS)
            elif cfg == "all-w":
                on = set(ALL) - set(WS)
            elif cfg.startswith("all-"):
                on = set(ALL) - set(cfg[4:].split("+"))
            else:
                on = {cfg}
            heads = cfg.endswith("+heads")
            if heads:
                cfg0 = cfg[:-6]
                on = set(ALL) if cfg0 == "all" else (set() if cfg0 == "none" else on)
            if lnfold:
                on = set(on) | {"lnfold"}
            z = emulate_vit(ref.vit, x, on, tdt, blocks_on, exempt)
            cfg = label + ("@lnfold" if lnfold else "")
            if heads:
                with heads_16bit(tdt):
                    out = ref.forward(x, idx=idx, K=K, is_training=True, z=z)
            else:
                out = ref.forward(x, idx=idx, K=K, is_training=True, z=z)
            print(f"{cfg:28s} feat {rel(z, z0):.2e} | " + " ".join(f"{k}={rel(out[k], out0[k]):.1e}" for k in keys), flush=True)


if __name__ == "__main__":
    main()
