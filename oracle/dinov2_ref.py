"""CPU restatement of facebookresearch/dinov2 ``DinoVisionTransformer`` (torch.hub ``dinov2_vit{s,b,l}14``).

TEST INFRASTRUCTURE (oracle) -- see oracle/__init__.py.  Third-party dependency of the reference, not
vendored under /root/reference and unpinned (``torch.hub.load('facebookresearch/dinov2', name)``,
reference blocks/dinov2.py:12).  Restated from the published algorithm; the
reference call site that fixes *which* method/output is used is blocks/dinov2.py:16-26
(``encoder.get_intermediate_layers(x)[0]`` -> last block, final norm applied, class token dropped).

Hub factory configuration restated: img_size=518, patch_size=14, init_values=1.0 (LayerScale),
ffn_layer='mlp', block_chunks=0, num_register_tokens=0, interpolate_antialias=False,
interpolate_offset=0.1, qkv/proj/ffn bias, mlp_ratio=4, LayerNorm(eps=1e-6), exact-erf GELU.
Parameter names equal the hub module's so a real ``state_dict`` loads unchanged.
"""
from __future__ import annotations

import math
import torch
from torch import nn
import torch.nn.functional as F

CFG = {
    "dinov2_vits14": dict(embed_dim=384, depth=12, num_heads=6),
    "dinov2_vitb14": dict(embed_dim=768, depth=12, num_heads=12),
    "dinov2_vitl14": dict(embed_dim=1024, depth=24, num_heads=16),
}


class PatchEmbed(nn.Module):
    def __init__(self, patch_size, embed_dim):
        super().__init__()
        self.proj = nn.Conv2d(3, embed_dim, kernel_size=patch_size, stride=patch_size)

    def forward(self, x):
        x = self.proj(x)                        # [B, C, G, G]
        return x.flatten(2).transpose(1, 2)     # [B, N, C], n = y*G + x


class Attention(nn.Module):
    def __init__(self, dim, num_heads):
        super().__init__()
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=True)
        self.proj = nn.Linear(dim, dim, bias=True)

    def forward(self, x):
        B, T, C = x.shape
        qkv = self.qkv(x).reshape(B, T, 3, self.num_heads, C // self.num_heads).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0] * self.scale, qkv[1], qkv[2]
        attn = (q @ k.transpose(-2, -1)).softmax(dim=-1)
        x = (attn @ v).transpose(1, 2).reshape(B, T, C)
        return self.proj(x)


class Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.act = nn.GELU()
        self.fc2 = nn.Linear(hidden, dim)

    def forward(self, x):
        return self.fc2(self.act(self.fc1(x)))


class LayerScale(nn.Module):
    def __init__(self, dim, init_values=1.0):
        super().__init__()
        self.gamma = nn.Parameter(init_values * torch.ones(dim))

    def forward(self, x):
        return x * self.gamma


class Block(nn.Module):
    def __init__(self, dim, num_heads):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = Attention(dim, num_heads)
        self.ls1 = LayerScale(dim)
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = Mlp(dim, dim * 4)
        self.ls2 = LayerScale(dim)

    def forward(self, x):
        x = x + self.ls1(self.attn(self.norm1(x)))
        x = x + self.ls2(self.mlp(self.norm2(x)))
        return x


def interpolate_pos_embed(pos_embed: torch.Tensor, G: int, offset: float = 0.1) -> torch.Tensor:
    """``DinoVisionTransformer.interpolate_pos_encoding`` for a square G x G grid -> [1, 1+G*G, C].

    Bicubic (A=-0.75, align_corners=False, antialias=False) with the *given* scale factor
    ``(G+offset)/M`` used for the source coordinates, output size floor(M*scale)=G."""
    N = pos_embed.shape[1] - 1
    M = int(math.sqrt(N))
    if G == M:
        return pos_embed
    pe = pos_embed.float()
    cls_pe, patch_pe = pe[:, 0], pe[:, 1:]
    C = pe.shape[-1]
    s = float(G + offset) / M
    patch_pe = F.interpolate(patch_pe.reshape(1, M, M, C).permute(0, 3, 1, 2), mode="bicubic",
                             antialias=False, scale_factor=(s, s))
    assert patch_pe.shape[-2:] == (G, G)
    patch_pe = patch_pe.permute(0, 2, 3, 1).reshape(1, -1, C)
    return torch.cat((cls_pe.unsqueeze(0), patch_pe), dim=1)


class DinoVisionTransformer(nn.Module):
    def __init__(self, embed_dim=1024, depth=24, num_heads=16, patch_size=14, img_size=518):
        super().__init__()
        self.embed_dim = self.num_features = embed_dim
        self.patch_size = patch_size
        self.num_heads = num_heads
        self.patch_embed = PatchEmbed(patch_size, embed_dim)
        n = (img_size // patch_size) ** 2
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, n + 1, embed_dim))
        self.mask_token = nn.Parameter(torch.zeros(1, embed_dim))
        self.blocks = nn.ModuleList([Block(embed_dim, num_heads) for _ in range(depth)])
        self.norm = nn.LayerNorm(embed_dim, eps=1e-6)
        nn.init.trunc_normal_(self.pos_embed, std=0.02)
        nn.init.normal_(self.cls_token, std=1e-6)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=0.02)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)

    def prepare_tokens(self, x):
        B, _, H, W = x.shape
        x = self.patch_embed(x)
        x = torch.cat((self.cls_token.expand(B, -1, -1), x), dim=1)
        assert H == W
        return x + interpolate_pos_embed(self.pos_embed, H // self.patch_size).to(x.dtype)

    def get_intermediate_layers(self, x, n=1, reshape=False, return_class_token=False, norm=True):
        assert n == 1 and not reshape
        x = self.prepare_tokens(x)
        for blk in self.blocks:
            x = blk(x)
        if norm:
            x = self.norm(x)
        cls, out = x[:, 0], x[:, 1:]
        if return_class_token:
            return ((out, cls),)
        return (out,)


def build(name: str, depth_override: int | None = None) -> DinoVisionTransformer:
    cfg = dict(CFG[name])
    if depth_override is not None:
        cfg["depth"] = depth_override
    return DinoVisionTransformer(**cfg)
