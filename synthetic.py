"""Seeded synthetic assets for the Multi-HMR inference path (test / benchmark infrastructure; not part of the product package).

There is no network in the build/bench environment, so neither the released
checkpoints (``models/multiHMR/*.pt``), nor ``SMPLX_NEUTRAL.npz``, nor
``smpl_mean_params.npz`` exist.  This module fabricates arrays of exactly the
shapes / dtypes / key names the reference loads, so that the real files are a
drop-in replacement when they are available:

* ``make_smplx_data``   -> the keys of ``SMPLX_NEUTRAL.npz`` consumed through
  ``smplx.create(SMPLX_DIR, 'smplx', ...)`` (reference blocks/smpl_layer.py:38).
* ``make_mean_params``  -> ``models/smpl_mean_params.npz`` (reference model.py:442-462).
* ``make_state_dict``   -> ``ckpt['model_state_dict']`` (reference demo.py:103), key names of
  SURVEY.md Appendix A.1 / C.

Nothing here is on the compute path; it only produces inputs.
"""
from __future__ import annotations

import math
import numpy as np
import torch

from multi_hmr_amd.constants import (ANNY_NUM_JOINTS, ANNY_USEFUL_ROTMAT, SMPLX_EXTRA_JOINT_VERTS, SMPLX_JOINT_NAMES,  # noqa: F401
                                     SMPLX_NUM_FACES, SMPLX_NUM_JOINTS, SMPLX_NUM_VERTS, SMPLX_PARENTS, VIT_CFG, anny_init_body_pose,
                                     anny_sincos_pos_embed, get_camera_K)

def make_smplx_data(seed: int = 0, num_verts: int = SMPLX_NUM_VERTS, num_faces: int = SMPLX_NUM_FACES,
                    max_influences: int = 4) -> dict:
    """A synthetic stand-in for ``SMPLX_NEUTRAL.npz`` with the real topology sizes.

    Geometry is a crude 'blob per bone' body so that magnitudes (metres) are realistic: joints
    form a chain with 5-25 cm bones, vertices sit within a few cm of their primary bone, skinning
    weights have ``max_influences`` non-zeros per vertex and sum to one, the joint regressor rows
    are convex combinations, pose correctives are mm-scale, shape directions cm-scale.
    """
    rng = np.random.RandomState(seed)
    V, J = num_verts, SMPLX_NUM_JOINTS
    parents = np.asarray(SMPLX_PARENTS, dtype=np.int64)

    # template skeleton
    jt = np.zeros((J, 3), dtype=np.float64)
    for j in range(1, J):
        bone = 0.04 if j >= 25 else 0.18  # fingers are short
        d = rng.randn(3)
        d /= np.linalg.norm(d)
        jt[j] = jt[parents[j]] + bone * (0.6 + 0.4 * rng.rand()) * d

    # vertices around a primary joint
    primary = rng.randint(0, J, size=V)
    primary[:J] = np.arange(J)  # every joint owns at least one vertex
    v_template = jt[primary] + 0.035 * rng.randn(V, 3)

    # skinning weights: primary, its parent, + random others
    weights = np.zeros((V, J), dtype=np.float64)
    for v in range(V):
        js = {int(primary[v])}
        p = int(parents[primary[v]])
        if p >= 0:
            js.add(p)
        while len(js) < max_influences:
            js.add(int(rng.randint(0, J)))
        js = sorted(js)
        w = rng.dirichlet(np.ones(len(js)) * 0.7)
        weights[v, js] = w

    # joint regressor: convex combination of 24 vertices owned by (or near) the joint
    J_regressor = np.zeros((J, V), dtype=np.float64)
    for j in range(J):
        owned = np.nonzero(primary == j)[0]
        pick = owned[:24] if len(owned) >= 24 else np.concatenate([owned, rng.randint(0, V, 24 - len(owned))])
        w = rng.dirichlet(np.ones(len(pick)))
        np.add.at(J_regressor[j], pick, w)

    shapedirs = np.zeros((V, 3, 400), dtype=np.float32)
    shapedirs[:, :, :16] = (0.012 * rng.randn(V, 3, 16)).astype(np.float32)        # betas (first 10/11 used)
    shapedirs[:, :, 300:310] = (0.004 * rng.randn(V, 3, 10)).astype(np.float32)    # expression
    posedirs = (0.0025 * rng.randn(V, 3, 486)).astype(np.float32)

    f = rng.randint(0, V, size=(num_faces, 3)).astype(np.int64)
    lmk_faces_idx = rng.randint(0, num_faces, size=51).astype(np.int64)
    lmk_bary = rng.dirichlet(np.ones(3), size=51)

    kintree = np.stack([np.where(parents < 0, 2 ** 32 - 1, parents), np.arange(J)]).astype(np.int64)
    return {
        "v_template": v_template.astype(np.float32),
        "f": f,
        "shapedirs": shapedirs,
        "posedirs": posedirs,
        "J_regressor": J_regressor.astype(np.float32),
        "weights": weights.astype(np.float32),
        "kintree_table": kintree,
        "lmk_faces_idx": lmk_faces_idx,
        "lmk_bary_coords": lmk_bary.astype(np.float32),
    }


def make_mean_params(seed: int = 0) -> dict:
    """Stand-in for ``smpl_mean_params.npz``: keys pose[144] (24 x 6D), shape[10], cam[3]."""
    rng = np.random.RandomState(seed + 1000)
    ident6 = np.array([1, 0, 0, 0, 1, 0], dtype=np.float32)
    pose = np.tile(ident6, 24) + 0.08 * rng.randn(144).astype(np.float32)
    shape = (0.3 * rng.randn(10)).astype(np.float32)
    cam = np.array([0.9, 0.0, 0.0], dtype=np.float32)
    return {"pose": pose.astype(np.float32), "shape": shape, "cam": cam}


def _trunc_normal(gen, shape, std=0.02):
    t = torch.empty(shape, dtype=torch.float32)
    t.normal_(0.0, std, generator=gen)
    return t.clamp_(-2 * std, 2 * std)


def make_state_dict(backbone: str = "dinov2_vitl14", img_size: int = 896, xat_depth: int = 2,
                    xat_num_heads: int = 8, num_betas: int = 10, seed: int = 0,
                    depth_override: int | None = None, mean_params: dict | None = None,
                    layerscale: float = 1.0, camera_num_bands: int = 16) -> dict:
    """Seeded random ``model_state_dict`` with the reference's key names and shapes.

    Initialisation follows the reference constructors (DINOv2: trunc-normal(0.02) linears, zero bias,
    LN 1/0, LayerScale ``layerscale``; heads/HPH: small normal) with two deliberate conditionings that a
    trained checkpoint would also satisfy:
      * ``decpose.bias`` moves init_body_pose joints 24..52 from the degenerate ``[1,0,0,1,0,0]`` to
        ``[1,0,0,0,1,0]`` so that 6D->rotmat is well conditioned (the reference NaNs otherwise);
      * read-out weights are small, so predictions are 'mean + small delta'.
    """
    cfg = dict(VIT_CFG[backbone])
    if depth_override is not None:
        cfg["depth"] = depth_override
    C, L = cfg["embed_dim"], cfg["depth"]
    g = torch.Generator().manual_seed(seed)
    sd = {}
    p = "backbone.encoder."
    sd[p + "cls_token"] = torch.empty(1, 1, C).normal_(0, 0.02, generator=g)
    sd[p + "pos_embed"] = _trunc_normal(g, (1, 1 + 37 * 37, C))
    sd[p + "mask_token"] = torch.zeros(1, C)
    sd[p + "patch_embed.proj.weight"] = _trunc_normal(g, (C, 3, 14, 14), std=0.04)
    sd[p + "patch_embed.proj.bias"] = torch.empty(C).normal_(0, 0.02, generator=g)
    for i in range(L):
        b = f"{p}blocks.{i}."
        for n in ("norm1", "norm2"):
            sd[b + n + ".weight"] = 1.0 + 0.1 * torch.empty(C).normal_(0, 1, generator=g)
            sd[b + n + ".bias"] = 0.05 * torch.empty(C).normal_(0, 1, generator=g)
        sd[b + "attn.qkv.weight"] = _trunc_normal(g, (3 * C, C), std=0.04)
        sd[b + "attn.qkv.bias"] = 0.02 * torch.empty(3 * C).normal_(0, 1, generator=g)
        sd[b + "attn.proj.weight"] = _trunc_normal(g, (C, C))
        sd[b + "attn.proj.bias"] = 0.02 * torch.empty(C).normal_(0, 1, generator=g)
        sd[b + "ls1.gamma"] = layerscale * (1.0 + 0.1 * torch.empty(C).normal_(0, 1, generator=g))
        sd[b + "ls2.gamma"] = layerscale * (1.0 + 0.1 * torch.empty(C).normal_(0, 1, generator=g))
        sd[b + "mlp.fc1.weight"] = _trunc_normal(g, (4 * C, C))
        sd[b + "mlp.fc1.bias"] = 0.02 * torch.empty(4 * C).normal_(0, 1, generator=g)
        sd[b + "mlp.fc2.weight"] = _trunc_normal(g, (C, 4 * C))
        sd[b + "mlp.fc2.bias"] = 0.02 * torch.empty(C).normal_(0, 1, generator=g)
    sd[p + "norm.weight"] = 1.0 + 0.1 * torch.empty(C).normal_(0, 1, generator=g)
    sd[p + "norm.bias"] = 0.05 * torch.empty(C).normal_(0, 1, generator=g)

    def lin(name, out_f, in_f, std=None, bias=True):
        s = std if std is not None else 1.0 / math.sqrt(in_f)
        sd[name + ".weight"] = s * torch.empty(out_f, in_f).normal_(0, 1, generator=g)
        if bias:
            sd[name + ".bias"] = 0.02 * torch.empty(out_f).normal_(0, 1, generator=g)

    lin("mlp_classif.0", C, C)
    lin("mlp_classif.2", 1, C)
    lin("mlp_offset.0", C, C)
    lin("mlp_offset.2", 2, C, std=0.2 / math.sqrt(C))

    G = img_size // 14
    Cc = C + 3 + 6 * camera_num_bands            # 99 camera channels for the released checkpoints' 16 bands
    h = "x_attention_head."
    for n in ("cross_queries_x", "cross_queries_y", "cross_values_x", "cross_values_y"):
        sd[h + n] = 0.2 * torch.empty(G, Cc).normal_(0, 1, generator=g)
    mp = mean_params if mean_params is not None else make_mean_params(seed)
    init_body_pose = torch.eye(3).reshape(1, 3, 3).repeat(53, 1, 1)[:, :, :2].flatten(1).reshape(1, -1)
    init_body_pose[:, : 24 * 6] = torch.from_numpy(np.asarray(mp["pose"], dtype=np.float32))
    sd[h + "init_body_pose"] = init_body_pose
    init_betas = torch.from_numpy(np.asarray(mp["shape"], dtype=np.float32)).unsqueeze(0)
    sd[h + "init_betas_kid"] = torch.cat([init_betas, torch.zeros_like(init_betas[:, :1])], 1)
    if num_betas == 11:
        init_betas = torch.cat([init_betas, torch.zeros_like(init_betas[:, :1])], 1)
    sd[h + "init_betas"] = init_betas
    sd[h + "init_cam"] = torch.from_numpy(np.asarray(mp["cam"], dtype=np.float32)).unsqueeze(0)
    sd[h + "init_expression"] = torch.zeros(1, 10)

    dim, inner, mlp_dim = 1024, 32 * xat_num_heads, 1024
    token_dim = 318 + num_betas + 3 + Cc
    t = h + "transformer."
    sd[t + "pos_embedding"] = torch.empty(1, 1, dim).normal_(0, 1, generator=g)
    lin(t + "to_token_embedding", dim, token_dim)
    for l in range(xat_depth):
        b = f"{t}transformer.layers.{l}."
        for k in range(3):
            sd[f"{b}{k}.norm.weight"] = 1.0 + 0.1 * torch.empty(dim).normal_(0, 1, generator=g)
            sd[f"{b}{k}.norm.bias"] = 0.05 * torch.empty(dim).normal_(0, 1, generator=g)
        lin(b + "0.fn.to_qkv", 3 * inner, dim, bias=False)
        lin(b + "0.fn.to_out.0", dim, inner)
        lin(b + "1.fn.to_kv", 2 * inner, Cc, bias=False)
        lin(b + "1.fn.to_q", inner, dim, bias=False)
        lin(b + "1.fn.to_out.0", dim, inner)
        lin(b + "2.fn.net.0", mlp_dim, dim)
        lin(b + "2.fn.net.3", dim, mlp_dim)
    small = 0.15 / math.sqrt(dim)
    lin(h + "decpose", 318, dim, std=small)
    lin(h + "decshape", num_betas, dim, std=small)
    lin(h + "deccam", 3, dim, std=small)
    lin(h + "decexpression", 10, dim, std=small)
    # Appendix B.6 conditioning: init+bias == [1,0,0,0,1,0] for joints 24..52
    bias = sd[h + "decpose.bias"]
    ident6 = torch.tensor([1.0, 0, 0, 0, 1, 0])
    for j in range(24, 53):
        bias[6 * j: 6 * j + 6] = ident6 - init_body_pose[0, 6 * j: 6 * j + 6] + 0.01 * torch.empty(6).normal_(0, 1, generator=g)
    return sd


def make_state_dict_anny(backbone: str = "dinov2_vits14", img_size: int = 224, xat_dim: int = 512, xat_depth: int = 8,
                         xat_heads: int = 16, xat_mlp_dim: int = 2048, num_betas: int = 11, seed: int = 0,
                         depth_override: int | None = None) -> dict:
    """Seeded random state_dict with the key names of ``multi_hmr_anny.multi_hmr.Multi_HMR`` (encoder.backbone.* = the hub
    DINOv2, encoder.mlp_det / mlp_fov_unique, dec_to_token, decoder.transformer.layers.*, mlp_offset / mlp_pose / mlp_shape /
    mlp_dist, buffers dec_pos_emb / init_body_pose / encoder.fov_max, parameters eye / useful_rotmat).  The body model's own
    tensors (``anny`` package) are not part of it."""
    assert len(ANNY_USEFUL_ROTMAT) == ANNY_NUM_JOINTS
    base = make_state_dict(backbone, img_size, seed=seed, depth_override=depth_override)
    sd = {"encoder.backbone." + k[len("backbone.encoder."):]: v for k, v in base.items() if k.startswith("backbone.encoder.")}
    C = VIT_CFG[backbone]["embed_dim"]
    g = torch.Generator().manual_seed(seed + 977)

    def lin(name, out_f, in_f, std=None):
        s = std if std is not None else 1.0 / math.sqrt(in_f)
        sd[name + ".weight"] = s * torch.empty(out_f, in_f).normal_(0, 1, generator=g)
        sd[name + ".bias"] = 0.02 * torch.empty(out_f).normal_(0, 1, generator=g)

    lin("encoder.mlp_det.0", C, C)
    lin("encoder.mlp_det.2", 1, C)
    lin("encoder.mlp_fov_unique.0", C, C)
    lin("encoder.mlp_fov_unique.2", 1, C, std=0.5 / math.sqrt(C))
    sd["encoder.fov_max"] = torch.tensor([math.pi])
    lin("dec_to_token", xat_dim, C)
    inner = 32 * xat_heads
    for l in range(xat_depth):
        b = f"decoder.transformer.layers.{l}."
        for k in range(3):
            sd[f"{b}{k}.norm.weight"] = 1.0 + 0.1 * torch.empty(xat_dim).normal_(0, 1, generator=g)
            sd[f"{b}{k}.norm.bias"] = 0.05 * torch.empty(xat_dim).normal_(0, 1, generator=g)
        sd[b + "0.fn.to_qkv.weight"] = xat_dim ** -0.5 * torch.empty(3 * inner, xat_dim).normal_(0, 1, generator=g)
        lin(b + "0.fn.to_out.0", xat_dim, inner)
        sd[b + "1.fn.to_kv.weight"] = xat_dim ** -0.5 * torch.empty(2 * inner, xat_dim).normal_(0, 1, generator=g)
        sd[b + "1.fn.to_q.weight"] = xat_dim ** -0.5 * torch.empty(inner, xat_dim).normal_(0, 1, generator=g)
        lin(b + "1.fn.to_out.0", xat_dim, inner)
        lin(b + "2.fn.net.0", xat_mlp_dim, xat_dim)
        lin(b + "2.fn.net.3", xat_dim, xat_mlp_dim)
    J = ANNY_NUM_JOINTS
    lin("mlp_offset.0", xat_dim, xat_dim)
    lin("mlp_offset.2", 2, xat_dim, std=0.2 / math.sqrt(xat_dim))
    lin("mlp_pose.0", xat_dim, xat_dim + 6 * J)
    lin("mlp_pose.2", 6 * J, xat_dim, std=0.15 / math.sqrt(xat_dim))
    lin("mlp_shape.0", xat_dim, xat_dim)
    lin("mlp_shape.2", num_betas, xat_dim)
    lin("mlp_dist.0", xat_dim, xat_dim)
    lin("mlp_dist.2", 1, xat_dim, std=0.3 / math.sqrt(xat_dim))
    sd["mlp_dist.2.bias"] = torch.tensor([5.0])       # exp(5) ~ 148: distances of a few metres at f ~ 200-800 px
    sd["dec_pos_emb"] = torch.from_numpy(anny_sincos_pos_embed(xat_dim, img_size // 14)).float()
    sd["init_body_pose"] = anny_init_body_pose()
    sd["eye"] = torch.eye(3).unsqueeze(0)
    sd["useful_rotmat"] = torch.tensor(ANNY_USEFUL_ROTMAT).unsqueeze(0)
    return sd


def make_pinned_idx(batch: int, grid: int, per_image: int, seed: int = 0):
    """Seeded distinct (y, x) cells per image, sorted like ``torch.where`` would return them
    (ascending b, y, x; reference model.py:146-149).  Used through the reference's own
    ``idx=...``/``is_training=True`` hook (model.py:141-151) to pin the number of persons."""
    g = torch.Generator().manual_seed(seed + 77)
    bs, ys, xs = [], [], []
    for b in range(batch):
        cells = torch.randperm(grid * grid, generator=g)[:per_image].sort().values
        bs.append(torch.full((per_image,), b, dtype=torch.long))
        ys.append(cells // grid)
        xs.append(cells % grid)
    b, y, x = torch.cat(bs), torch.cat(ys), torch.cat(xs)
    return (b, y, x, torch.zeros_like(b))


def make_hostile(sd: dict, kind: str, seed: int = 0, strength: float = 1.0) -> dict:
    """Weight statistics a TRAINED DINOv2 checkpoint can have and ``make_state_dict`` does not (round-3 review: the precision scheme --
    low-half weight passes, LayerNorm fold -- was tuned on N(1, 0.1) LayerNorm / LayerScale weights).  Applied to the backbone in place:

      "weights"  LayerScale gamma log-uniform in [1e-3, 1] per channel; LayerNorm weight log-normal (sigma 0.5) with six channels per
                 norm scaled x10 ... x30; LayerNorm bias N(0, 0.7) clipped to +-2; qkv bias N(0, 1)
      "mean"     token rows far from zero: a DC offset on the patch-embedding bias (every channel +2 sigma of a token row at block 0)
                 and on every proj / fc2 bias, so that |mean| / std of a residual row stays around 2 through the depth -- the case the
                 LayerNorm fold is sensitive to (it rounds the RAW row to 16 bits and centres afterwards)
    """
    # strength (kind "weights" only): 1 = the statistics above (what the goldens were made with: the same random draws, bit for bit);
    # < 1 interpolates towards benign weights -- hot channels x(1 + strength (f - 1)), log-normal sigma and biases scaled by strength --
    # for probing where the `precision="auto"` rule (multi_hmr_amd/vit.py logit_gain) has to switch
    g = torch.Generator().manual_seed(7000 + seed)
    p = "backbone.encoder."
    L = 1 + max(int(k.split(".")[3]) for k in sd if k.startswith(p + "blocks."))
    C = sd[p + "norm.weight"].shape[0]
    rn = lambda *s: torch.empty(*s).normal_(0, 1, generator=g)
    if kind == "weights":
        for i in range(L):
            b = f"{p}blocks.{i}."
            for n in ("ls1.gamma", "ls2.gamma"):
                sd[b + n] = torch.exp(torch.empty(C).uniform_(math.log(1e-3), 0.0, generator=g))
            for n in ("norm1", "norm2"):
                w = torch.exp(0.5 * strength * rn(C))
                hot = torch.randperm(C, generator=g)[:6]
                w[hot] *= 1.0 + strength * (torch.tensor([10.0, 10.0, 10.0, 10.0, 30.0, 30.0]) - 1.0)
                sd[b + n + ".weight"] = w
                sd[b + n + ".bias"] = (0.7 * strength * rn(C)).clamp_(-2, 2)
            sd[b + "attn.qkv.bias"] = strength * rn(3 * C)
    elif kind == "mean":
        sd[p + "patch_embed.proj.bias"] = sd[p + "patch_embed.proj.bias"] + 1.7
        for i in range(L):
            b = f"{p}blocks.{i}."
            # LayerScale is ~1 in make_state_dict: the DC of a block's two branch outputs is what its residual row gains
            sd[b + "attn.proj.bias"] = sd[b + "attn.proj.bias"] + 0.12
            sd[b + "mlp.fc2.bias"] = sd[b + "mlp.fc2.bias"] + 0.12
    else:
        raise ValueError(kind)
    return sd
