"""Portable CPU fp32 restatement of the reference ``Model.forward`` (naver/multi-hmr model.py:205-349).

TEST INFRASTRUCTURE (oracle) -- never imported by the product.  This is what the ``-m gpu`` parity tests,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg use on the GPU box, where
/root/reference does not exist.  It is a *functional* restatement driven directly by a ``state_dict``
(no nn.Module mirroring of the reference classes); every function cites the reference lines it follows.
It is pinned against the reference's own code executed verbatim (oracle/ref_shim.py) through the golden
vectors in tests/golden/ (tests/test_oracle_golden.py).  Third-party arithmetic comes from
oracle/{dinov2_ref,smplx_ref,roma_ref}.py.
"""
from __future__ import annotations

import math
import numpy as np
import torch
import torch.nn.functional as F

from oracle import dinov2_ref, smplx_ref, roma_ref

PATCH = 14


# ---------------------------------------------------------------- utils/camera.py
def perspective_projection(x, K):
    """utils/camera.py:14-27"""
    y = x / x[:, :, -1].unsqueeze(-1)
    y = torch.einsum("bij,bkj->bki", K, y)
    return y[:, :, :2]


def inverse_perspective_projection(points, K, distance):
    """utils/camera.py:30-48"""
    points = torch.cat([points, torch.ones_like(points[..., :1])], -1)
    points = torch.einsum("bij,bkj->bki", torch.inverse(K), points)
    return points * distance


def focal_from_fov(fov=60, img_size=512):
    """utils/camera.py:50-60"""
    return img_size / (2 * np.tan(np.radians(fov) / 2))


# ---------------------------------------------------------------- blocks/camera_embed.py
def fourier_features(pos, num_bands=16, max_resolution=64):
    """blocks/camera_embed.py:39-58 : [pos, sin(pi*pos*f), cos(pi*pos*f)], f=linspace(1,res/2,bands) per axis."""
    b, n = pos.shape[:2]
    freq = torch.stack([torch.linspace(1.0, max_resolution / 2, num_bands) for _ in range(3)], dim=0)  # [3,bands]
    feat = (pos[:, :, :, None] * freq[None, None, :, :]).reshape(b, n, -1)
    feat = torch.cat([torch.sin(np.pi * feat), torch.cos(np.pi * feat)], dim=-1)
    return torch.cat([pos, feat], dim=-1)


def embedd_camera(K, G, num_bands=16, max_resolution=64):
    """model.py:160-187.  NB the ROW index is fed as pixel-x and the COLUMN index as pixel-y (model.py:164-177)."""
    bs = K.shape[0]
    pts = torch.stack([torch.arange(G).reshape(-1, 1).repeat(1, G), torch.arange(G).reshape(1, -1).repeat(G, 1)], -1).float()
    pts = pts * PATCH + PATCH // 2
    pts = pts.reshape(1, -1, 2).repeat(bs, 1, 1)
    rays = inverse_perspective_projection(pts, K, torch.ones(bs, pts.shape[1], 1))
    return fourier_features(rays, num_bands, max_resolution).reshape(bs, G, G, 3 + 6 * num_bands)


# ---------------------------------------------------------------- model.py helpers
def mlp2(sd, prefix, x):
    """regression_mlp([C, C, out]) model.py:596-609: Linear-ReLU-Linear."""
    h = F.relu(F.linear(x, sd[prefix + ".0.weight"], sd[prefix + ".0.bias"]))
    return F.linear(h, sd[prefix + ".2.weight"], sd[prefix + ".2.bias"])


def nms(heat, kernel=3):
    """model.py:620-638"""
    pad = (kernel - 1) // 2 if kernel not in (2, 4) else (1 if kernel == 2 else 2)
    hmax = F.max_pool2d(heat, (kernel, kernel), stride=1, padding=pad)
    if hmax.shape[2] > heat.shape[2]:
        hmax = hmax[:, :, : heat.shape[2], : heat.shape[3]]
    return heat * (hmax == heat).float()


def detection(sd, z, G, nms_kernel_size, det_thresh, idx, is_training):
    """model.py:133-158 (+ _sigmoid 641-643, unpatch utils/image.py:39-52, apply_threshold 612-617)."""
    B = z.shape[0]
    s = torch.clamp(torch.sigmoid(mlp2(sd, "mlp_classif", z)), min=1e-4, max=1 - 1e-4)   # [B,N,1]
    scores = s.reshape(B, G, G, 1).permute(0, 3, 1, 2)                                    # [B,1,G,G]
    if not is_training:
        if nms_kernel_size > 1:
            scores = nms(scores, nms_kernel_size)
        thr = det_thresh[0] if isinstance(det_thresh, list) else det_thresh
        idx = torch.where(scores.permute(0, 2, 3, 1) >= thr)
    else:
        assert idx is not None
    scores_det = scores[idx[0], idx[3], idx[1], idx[2]]
    return scores.permute(0, 2, 3, 1), scores_det, idx


# ---------------------------------------------------------------- utils/tensor_manip.py
def rebatch_dense(idx0):
    """Semantics of utils/tensor_manip.py:7-26 as probed: counts of the non-empty
    images in ascending image order and the dense re-indexing of idx0 onto 0..B'-1."""
    values, inverse, counts = torch.unique(idx0, sorted=True, return_inverse=True, return_counts=True)
    return counts, inverse


# ---------------------------------------------------------------- blocks/cross_attn_transformer.py
def _ln(x, w, b):
    return F.layer_norm(x, (x.shape[-1],), w, b, 1e-5)


def self_attention(sd, p, x, mask, heads):
    """Attention.forward blocks/cross_attn_transformer.py:129-159 (mask multiplies AND additive -1e11)."""
    B, n, _ = x.shape
    qkv = F.linear(x, sd[p + "to_qkv.weight"]).chunk(3, dim=-1)
    q, k, v = [t.reshape(B, n, heads, -1).permute(0, 2, 1, 3) for t in qkv]
    q, k, v = [t * mask[:, None, :, None] for t in (q, k, v)]
    dots = torch.matmul(q, k.transpose(-1, -2)) * (q.shape[-1] ** -0.5)
    dots = dots - (1 - mask)[:, None, None, :] * 10e10
    attn = dots.softmax(dim=-1) * mask[:, None, None, :]
    out = torch.matmul(attn, v).permute(0, 2, 1, 3).reshape(B, n, -1)
    return F.linear(out, sd[p + "to_out.0.weight"], sd[p + "to_out.0.bias"])


def cross_attention(sd, p, x, context, mask, heads):
    """CrossAttention.forward blocks/cross_attn_transformer.py:185-205 (context is NOT normalised)."""
    B, n, _ = x.shape
    k, v = F.linear(context, sd[p + "to_kv.weight"]).chunk(2, dim=-1)
    q = F.linear(x, sd[p + "to_q.weight"])
    q, k, v = [t.reshape(B, t.shape[1], heads, -1).permute(0, 2, 1, 3) for t in (q, k, v)]
    q = q * mask[:, None, :, None]
    dots = torch.matmul(q, k.transpose(-1, -2)) * (q.shape[-1] ** -0.5)
    dots = dots - (1 - mask).float()[:, None, :, None] * 1e6
    out = torch.matmul(dots.softmax(dim=-1), v) * mask[:, None, :, None]
    out = out.permute(0, 2, 1, 3).reshape(B, n, -1)
    return F.linear(out, sd[p + "to_out.0.weight"], sd[p + "to_out.0.bias"])


def transformer_decoder(sd, p, token, context, mask, depth, heads):
    """TransformerDecoder.forward :351-359 + TransformerCrossAttn.forward :239-261 (pre-norm SA, CA, FF)."""
    x = F.linear(token, sd[p + "to_token_embedding.weight"], sd[p + "to_token_embedding.bias"])
    x = x + sd[p + "pos_embedding"][:, 0][:, None, :]
    for l in range(depth):
        b = f"{p}transformer.layers.{l}."
        x = x * mask[:, :, None]
        x = self_attention(sd, b + "0.fn.", _ln(x, sd[b + "0.norm.weight"], sd[b + "0.norm.bias"]), mask, heads) + x
        x = cross_attention(sd, b + "1.fn.", _ln(x, sd[b + "1.norm.weight"], sd[b + "1.norm.bias"]), context, mask, heads) + x
        h = _ln(x, sd[b + "2.norm.weight"], sd[b + "2.norm.bias"])
        h = F.linear(F.gelu(F.linear(h, sd[b + "2.fn.net.0.weight"], sd[b + "2.fn.net.0.bias"])),
                     sd[b + "2.fn.net.3.weight"], sd[b + "2.fn.net.3.bias"])
        x = h + x
    return x * mask[:, :, None]


# ---------------------------------------------------------------- model.py HPH
def hph_forward(sd, z_central, z_all, idx, depth, heads):
    """HPH.cross_attn_inputs model.py:479-525 + HPH.forward 527-593.

    ``z_all`` is the per-IMAGE context [B,Cc,G,G] (the reference first replicates it per person,
    model.py:278-280, then keeps one copy per image, :511 -- Appendix B.10)."""
    h = "x_attention_head."
    P = z_central.shape[0]
    counts, idx_det_0 = rebatch_dense(idx[0])
    # learned query embeddings: the *_x table is indexed by the ROW y, *_y by the COLUMN x (model.py:500-502)
    xc = z_central + sd[h + "cross_queries_x"][idx[1]] + sd[h + "cross_queries_y"][idx[2]]
    nmax = int(counts.max())
    Bp = counts.shape[0]
    xpad = xc.new_zeros(Bp, nmax, xc.shape[1])
    mask = xc.new_zeros(Bp, nmax)
    start = 0
    for i, c in enumerate(counts.tolist()):          # pad_to_max utils/tensor_manip.py:36-45
        xpad[i, :c] = xc[start:start + c]
        mask[i, :c] = 1
        start += c
    images = torch.unique(idx[0], sorted=True)
    xx = z_all[images].clone()                       # [B',Cc,G,G]  (model.py:511)
    xx[idx_det_0, :, idx[1], idx[2]] += sd[h + "cross_values_x"][idx[1]] + sd[h + "cross_values_y"][idx[2]]  # :514-517
    context = xx.flatten(2).transpose(1, 2)          # [B',N,Cc]
    expand = lambda t: t.expand(Bp, nmax, -1)
    init_pose, init_betas, init_cam, init_expr = [sd[h + n] for n in ("init_body_pose", "init_betas", "init_cam", "init_expression")]
    token = torch.cat([xpad, expand(init_pose), expand(init_betas), expand(init_cam)], dim=-1)    # :550
    out = transformer_decoder(sd, h + "transformer.", token, context, mask, depth, heads)
    out = torch.cat([out[i, :c] for i, c in enumerate(counts.tolist())], dim=0)                   # :558-561
    lin = lambda n: F.linear(out, sd[h + n + ".weight"], sd[h + n + ".bias"])
    pose6d = lin("decpose") + init_pose
    betas = lin("decshape") + init_betas
    cam = lin("deccam") + init_cam
    expr = lin("decexpression") + init_expr
    # rot6d_to_rotmat utils/humans.py:12-22: reshape(-1,2,3).permute(0,2,1) then Gram-Schmidt
    rotmat = roma_ref.special_gramschmidt(pose6d.reshape(-1, 2, 3).permute(0, 2, 1).contiguous()).view(P, 53, 3, 3)
    return rotmat, betas, expr, cam


# ---------------------------------------------------------------- blocks/smpl_layer.py
def smpl_layer_forward(bm, pose, shape, loc, dist, K, expression, person_center_idx=15):
    """SMPL_Layer.forward blocks/smpl_layer.py:47-155 (type='smplx', person_center='head' -> joint 15)."""
    bs = pose.shape[0]
    z3 = torch.zeros(bs, 3)
    out = bm(betas=shape, global_orient=z3, body_pose=pose[:, 1:22].flatten(1), left_hand_pose=pose[:, 22:37].flatten(1),
             right_hand_pose=pose[:, 37:52].flatten(1), jaw_pose=pose[:, 52:53].flatten(1), expression=expression.flatten(1),
             leye_pose=z3, reye_pose=z3)
    verts, j3d = out.vertices, out.joints
    R = roma_ref.rotvec_to_rotmat(pose[:, 0])
    pelvis = j3d[:, [0]]
    j3d = (R.unsqueeze(1) @ (j3d - pelvis).unsqueeze(-1)).squeeze(-1)
    verts = (R.unsqueeze(1) @ (verts - pelvis).unsqueeze(-1)).squeeze(-1)
    transl = inverse_perspective_projection(loc.unsqueeze(1), K, dist.unsqueeze(1))[:, 0]
    transl_up = transl.clone()
    if person_center_idx is None:                 # smpl_layer.py:128-130: the pelvis is added to the translation, nothing recentred
        transl_up = transl_up + pelvis[:, 0]
    else:                                         # smpl_layer.py:131-136
        center = j3d[:, [person_center_idx]]
        verts = verts - center
        j3d = j3d - center
    j3d_cam = j3d + transl_up.unsqueeze(1)
    verts_cam = verts + transl_up.unsqueeze(1)
    return {"v3d": verts_cam, "j3d": j3d_cam, "j2d": perspective_projection(j3d_cam, K),
            "v2d": perspective_projection(verts_cam, K), "transl": transl, "transl_pelvis": j3d_cam[:, [0]]}


# ---------------------------------------------------------------- the model
class OracleModel:
    """Functional CPU fp32 Multi-HMR.  ``forward`` == reference Model.forward (model.py:205-349)."""

    def __init__(self, state_dict: dict, smplx_data: dict, backbone="dinov2_vitl14", img_size=896, xat_depth=2,
                 xat_num_heads=8, num_betas=10, depth_override=None, nearness=True, camera_embedding_num_bands=16,
                 camera_embedding_max_resolution=64):
        self.num_bands, self.max_resolution = camera_embedding_num_bands, camera_embedding_max_resolution      # model.py:39-40
        self.sd = {k: v.detach().float().cpu() for k, v in state_dict.items()}
        self.img_size, self.depth, self.heads, self.nearness = img_size, xat_depth, xat_num_heads, nearness
        self.vit = dinov2_ref.build(backbone, depth_override)
        p = "backbone.encoder."
        missing, unexpected = self.vit.load_state_dict({k[len(p):]: v for k, v in self.sd.items() if k.startswith(p)}, strict=True)
        self.vit.eval()
        self.bm = smplx_ref.SMPLX(smplx_data, num_betas=num_betas)
        self.G = img_size // PATCH

    @torch.no_grad()
    def backbone(self, x):
        """blocks/dinov2.py:16-26"""
        return self.vit.get_intermediate_layers(x.float())[0]

    @torch.no_grad()
    def forward(self, x, idx=None, det_thresh=0.3, nms_kernel_size=3, K=None, is_training=False, z=None):
        sd, G = self.sd, self.G
        z = self.backbone(x) if z is None else z                                   # model.py:229
        B, N, C = z.shape
        scores, scores_det, idx = detection(sd, z, G, nms_kernel_size, det_thresh, idx, is_training)  # :233-240
        if len(idx[0]) == 0 and not is_training:
            return []
        zmap = z.reshape(B, G, G, C).permute(0, 3, 1, 2)                           # unpatch, :246-248
        z_central = zmap[idx[0], :, idx[1], idx[2]]                                # :255
        offset = mlp2(sd, "mlp_offset", z_central)                                 # :258
        K_det = K[idx[0]]
        z_K = embedd_camera(K, G, self.num_bands, self.max_resolution)             # :262
        z_central = torch.cat([z_central, z_K[idx[0], idx[1], idx[2]]], 1)         # :263-265
        z_all = torch.cat([zmap, z_K.permute(0, 3, 1, 2)], 1)                      # :266-268
        loc = (torch.stack([idx[2], idx[1]]).permute(1, 0) + 0.5 + offset) * PATCH  # :272-275
        rotmat, shape, expression, cam = hph_forward(sd, z_central, z_all, idx, self.depth, self.heads)  # :278-283
        rotvec = roma_ref.rotmat_to_rotvec(rotmat)                                 # :291
        dist_pp = cam[:, 0][:, None]
        focal = K_det[:, [0], [0]]
        dist = dist_pp * (focal / focal_from_fov(60, x.shape[-1]))                 # to_euclidean_dist :189-203
        if self.nearness:
            dist = torch.exp(dist) - 1e-10
        dist = torch.clamp(dist, 0, 50)
        out = {"scores": scores, "offset": offset, "dist": dist, "dist_postprocessed": dist_pp, "expression": expression,
               "rotmat": rotmat, "shape": shape, "rotvec": rotvec, "loc": loc}
        out.update(smpl_layer_forward(self.bm, rotvec, shape, loc, dist, K_det, expression))           # :319-322
        if is_training:
            return out
        persons = []
        for i in range(idx[0].shape[0]):                                           # :329-347
            persons.append({"scores": scores_det[i], "loc": out["loc"][i], "transl": out["transl"][i],
                            "transl_pelvis": out["transl_pelvis"][i], "rotvec": out["rotvec"][i],
                            "expression": out["expression"][i], "shape": out["shape"][i], "v3d": out["v3d"][i],
                            "j3d": out["j3d"][i], "j2d": out["j2d"][i]})
        return persons
