"""CPU fp32 restatement of the Anny-variant HPH (reference multi_hmr_anny/hph.py:12-151).  TEST INFRASTRUCTURE (oracle).
Functional, driven by a state_dict with the reference's key names.  Pinned against the reference file itself (imported by
path where /root/reference exists) and against tests/golden/anny_hph.npz."""
from __future__ import annotations

import torch
import torch.nn.functional as F


def _ln(x, w, b):
    return F.layer_norm(x, (x.shape[-1],), w, b, 1e-5)


def forward(sd: dict, x: torch.Tensor, context: torch.Tensor, mask: torch.Tensor, depth: int, heads: int) -> torch.Tensor:
    """HPH.forward (hph.py:148-151) -> TransformerCrossAttn.forward (:132-140): per layer
    x = SA(LN x) + x ; x = CA(LN x, context) + x ; x = FF(LN x) + x.  SA masks padded KEYS additively (-1e11, :61-62); CA's
    additive mask is per query row (no effect, :102-103); there are no mask multiplies in this variant."""
    B, n, _ = x.shape
    for l in range(depth):
        p = f"transformer.layers.{l}."
        h = _ln(x, sd[p + "0.norm.weight"], sd[p + "0.norm.bias"])
        q, k, v = [t.reshape(B, n, heads, -1).permute(0, 2, 1, 3) for t in F.linear(h, sd[p + "0.fn.to_qkv.weight"]).chunk(3, dim=-1)]
        dots = torch.matmul(q, k.transpose(-1, -2)) * (q.shape[-1] ** -0.5)
        dots = dots - (1 - mask)[:, None, None, :] * 10e10
        o = torch.matmul(dots.softmax(dim=-1), v).permute(0, 2, 1, 3).reshape(B, n, -1)
        x = F.linear(o, sd[p + "0.fn.to_out.0.weight"], sd[p + "0.fn.to_out.0.bias"]) + x
        h = _ln(x, sd[p + "1.norm.weight"], sd[p + "1.norm.bias"])
        kk, vv = F.linear(context, sd[p + "1.fn.to_kv.weight"]).chunk(2, dim=-1)
        q = F.linear(h, sd[p + "1.fn.to_q.weight"])
        q, kk, vv = [t.reshape(B, t.shape[1], heads, -1).permute(0, 2, 1, 3) for t in (q, kk, vv)]
        dots = torch.matmul(q, kk.transpose(-1, -2)) * (q.shape[-1] ** -0.5)
        dots = dots - (1 - mask).float()[:, None, :, None] * 1e6
        o = torch.matmul(dots.softmax(dim=-1), vv).permute(0, 2, 1, 3).reshape(B, n, -1)
        x = F.linear(o, sd[p + "1.fn.to_out.0.weight"], sd[p + "1.fn.to_out.0.bias"]) + x
        h = _ln(x, sd[p + "2.norm.weight"], sd[p + "2.norm.bias"])
        h = F.linear(F.gelu(F.linear(h, sd[p + "2.fn.net.0.weight"], sd[p + "2.fn.net.0.bias"])), sd[p + "2.fn.net.3.weight"], sd[p + "2.fn.net.3.bias"])
        x = h + x
    return x


def make_case(seed=0, dim=512, depth=8, heads=16, mlp=2048, counts=(3, 11, 1), N=256):
    """Seeded weights (reference key names) and padded inputs for the golden / parity tests."""
    g = torch.Generator().manual_seed(seed)
    inner = 32 * heads
    rn = lambda *s, std=1.0: std * torch.empty(*s).normal_(0, 1, generator=g)
    sd = {}
    for l in range(depth):
        p = f"transformer.layers.{l}."
        for k in range(3):
            sd[f"{p}{k}.norm.weight"], sd[f"{p}{k}.norm.bias"] = 1 + 0.1 * rn(dim), 0.05 * rn(dim)
        sd[p + "0.fn.to_qkv.weight"] = rn(3 * inner, dim, std=dim ** -0.5)
        sd[p + "0.fn.to_out.0.weight"], sd[p + "0.fn.to_out.0.bias"] = rn(dim, inner, std=inner ** -0.5), 0.02 * rn(dim)
        sd[p + "1.fn.to_kv.weight"] = rn(2 * inner, dim, std=dim ** -0.5)
        sd[p + "1.fn.to_q.weight"] = rn(inner, dim, std=dim ** -0.5)
        sd[p + "1.fn.to_out.0.weight"], sd[p + "1.fn.to_out.0.bias"] = rn(dim, inner, std=inner ** -0.5), 0.02 * rn(dim)
        sd[p + "2.fn.net.0.weight"], sd[p + "2.fn.net.0.bias"] = rn(mlp, dim, std=dim ** -0.5), 0.02 * rn(mlp)
        sd[p + "2.fn.net.3.weight"], sd[p + "2.fn.net.3.bias"] = rn(dim, mlp, std=mlp ** -0.5), 0.02 * rn(dim)
    B, nmax = len(counts), max(counts)
    x = rn(B, nmax, dim)
    mask = torch.zeros(B, nmax)
    for i, c in enumerate(counts):
        mask[i, :c] = 1
    x = x * mask[:, :, None]          # the reference pads queries with zeros (multi_hmr_anny/multi_hmr.py:133)
    context = rn(B, N, dim)
    return sd, x, context, mask
