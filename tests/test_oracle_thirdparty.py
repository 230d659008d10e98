"""Cross-checks of the three third-party restatements in oracle/ against INDEPENDENT implementations that
are installed (HF transformers Dinov2, scipy Rotation) and analytic known answers (SMPL-X LBS).
SURVEY.md Appendix A.4."""
import math

import numpy as np
import pytest
import torch

from oracle import dinov2_ref, roma_ref, smplx_ref
import synthetic


# ------------------------------------------------------------------ DINOv2
def _hf_from_ref(ref, depth, C, H, S):
    from transformers import Dinov2Config, Dinov2Model
    cfg = Dinov2Config(hidden_size=C, num_hidden_layers=depth, num_attention_heads=H, mlp_ratio=4, image_size=518,
                       patch_size=14, layerscale_value=1.0, layer_norm_eps=1e-6, hidden_act="gelu", qkv_bias=True)
    hf = Dinov2Model(cfg).eval()
    sd = ref.state_dict()
    m = {}
    m["embeddings.cls_token"] = sd["cls_token"]
    m["embeddings.mask_token"] = sd["mask_token"]
    m["embeddings.patch_embeddings.projection.weight"] = sd["patch_embed.proj.weight"]
    m["embeddings.patch_embeddings.projection.bias"] = sd["patch_embed.proj.bias"]
    # HF interpolates with size= (no +0.1 offset): feed it the table the restatement interpolates to, at the
    # native resolution of a config whose image_size == S so that HF does not interpolate at all.
    m["embeddings.position_embeddings"] = dinov2_ref.interpolate_pos_embed(sd["pos_embed"], S // 14)
    for i in range(depth):
        p, q = f"blocks.{i}.", f"encoder.layer.{i}."
        w, b = sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"]
        for j, n in enumerate(("query", "key", "value")):
            m[q + f"attention.attention.{n}.weight"] = w[j * C:(j + 1) * C]
            m[q + f"attention.attention.{n}.bias"] = b[j * C:(j + 1) * C]
        m[q + "attention.output.dense.weight"] = sd[p + "attn.proj.weight"]
        m[q + "attention.output.dense.bias"] = sd[p + "attn.proj.bias"]
        for n in ("norm1", "norm2"):
            m[q + n + ".weight"], m[q + n + ".bias"] = sd[p + n + ".weight"], sd[p + n + ".bias"]
        m[q + "layer_scale1.lambda1"], m[q + "layer_scale2.lambda1"] = sd[p + "ls1.gamma"], sd[p + "ls2.gamma"]
        for n in ("fc1", "fc2"):
            m[q + f"mlp.{n}.weight"], m[q + f"mlp.{n}.bias"] = sd[p + f"mlp.{n}.weight"], sd[p + f"mlp.{n}.bias"]
    m["layernorm.weight"], m["layernorm.bias"] = sd["norm.weight"], sd["norm.bias"]
    cfg.image_size = S
    hf = Dinov2Model(cfg).eval()
    missing, unexpected = hf.load_state_dict(m, strict=False)
    assert not unexpected and all("pooler" in k for k in missing), (missing, unexpected)
    return hf


def test_dinov2_restatement_matches_hf_transformers():
    torch.manual_seed(0)
    C, H, depth, S = 384, 6, 3, 224
    ref = dinov2_ref.DinoVisionTransformer(embed_dim=C, depth=depth, num_heads=H).eval()
    sd = synthetic.make_state_dict("dinov2_vits14", S, seed=5, depth_override=depth)
    ref.load_state_dict({k[len("backbone.encoder."):]: v for k, v in sd.items() if k.startswith("backbone.encoder.")})
    hf = _hf_from_ref(ref, depth, C, H, S)
    x = torch.randn(2, 3, S, S)
    with torch.no_grad():
        a = ref.get_intermediate_layers(x)[0]
        b = hf(pixel_values=x).last_hidden_state[:, 1:]
    assert a.shape == (2, 256, C)
    assert torch.allclose(a, b, atol=2e-5, rtol=1e-5), (a - b).abs().max()


def test_dinov2_token_order_and_pos_identity():
    ref = dinov2_ref.DinoVisionTransformer(embed_dim=64, depth=0, num_heads=1).eval()
    # native 37x37 grid: no interpolation
    assert dinov2_ref.interpolate_pos_embed(ref.pos_embed, 37) is ref.pos_embed
    x = torch.randn(1, 3, 28, 28)
    tok = ref.patch_embed(x)                       # n = y*G + x
    w = ref.patch_embed.proj.weight.reshape(64, -1)
    manual = x[0, :, 0:14, 14:28].reshape(-1) @ w.T + ref.patch_embed.proj.bias    # y=0, x=1 -> n=1
    assert torch.allclose(tok[0, 1], manual, atol=1e-5)


# ------------------------------------------------------------------ roma
def test_roma_restatement_vs_scipy():
    from scipy.spatial.transform import Rotation
    rng = np.random.RandomState(0)
    rv = rng.randn(500, 3) * rng.uniform(0.01, 1.5, size=(500, 1))
    R_sc = Rotation.from_rotvec(rv).as_matrix()
    R = roma_ref.rotvec_to_rotmat(torch.from_numpy(rv))
    assert np.abs(R.numpy() - R_sc).max() < 1e-12
    back = roma_ref.rotmat_to_rotvec(torch.from_numpy(R_sc)).numpy()
    assert np.abs(back - Rotation.from_matrix(R_sc).as_rotvec()).max() < 1e-10
    # identity / small angles / near pi
    assert torch.allclose(roma_ref.rotvec_to_rotmat(torch.zeros(1, 3)), torch.eye(3)[None])
    assert torch.allclose(roma_ref.rotmat_to_rotvec(torch.eye(3)[None]), torch.zeros(1, 3))
    big = torch.tensor([[0.0, 0.0, math.pi - 1e-3]], dtype=torch.float64)
    assert torch.allclose(roma_ref.rotmat_to_rotvec(roma_ref.rotvec_to_rotmat(big)), big, atol=1e-6)


def test_gramschmidt_orthonormal_and_convention():
    torch.manual_seed(0)
    M = torch.randn(100, 3, 2, dtype=torch.float64)
    R = roma_ref.special_gramschmidt(M)
    assert torch.allclose(R.transpose(-1, -2) @ R, torch.eye(3, dtype=torch.float64).expand(100, 3, 3), atol=1e-12)
    assert torch.allclose(torch.det(R), torch.ones(100, dtype=torch.float64))
    # reference convention (utils/humans.py:20): 6-vector -> (2,3) -> transpose: first three numbers = column 0
    six = torch.tensor([[1.0, 0, 0, 0, 1, 0]])
    assert torch.allclose(roma_ref.special_gramschmidt(six.reshape(-1, 2, 3).permute(0, 2, 1)), torch.eye(3)[None])


# ------------------------------------------------------------------ smplx LBS
def _bm(smplx_data):
    return smplx_ref.SMPLX(smplx_data, num_betas=10)


def _call(bm, pose55, betas, expr):
    B = pose55.shape[0]
    return bm(betas=betas, global_orient=pose55[:, 0], body_pose=pose55[:, 1:22].reshape(B, -1), jaw_pose=pose55[:, 22],
              leye_pose=pose55[:, 23], reye_pose=pose55[:, 24], left_hand_pose=pose55[:, 25:40].reshape(B, -1),
              right_hand_pose=pose55[:, 40:55].reshape(B, -1), expression=expr)


def test_lbs_zero_pose_is_shape_blend(smplx_data):
    bm = _bm(smplx_data)
    betas, expr = torch.randn(2, 10), torch.randn(2, 10)
    out = _call(bm, torch.zeros(2, 55, 3), betas, expr)
    v_shaped = bm.v_template + torch.einsum("bl,mkl->bmk", torch.cat([betas, expr], 1), torch.cat([bm.shapedirs, bm.expr_dirs], -1))
    assert torch.allclose(out.vertices, v_shaped, atol=2e-6)
    J = torch.einsum("bik,ji->bjk", v_shaped, bm.J_regressor)
    assert torch.allclose(out.joints[:, :55], J, atol=2e-6)
    assert out.joints.shape == (2, 127, 3)
    assert torch.allclose(out.joints[:, 55:76], out.vertices[:, synthetic.SMPLX_EXTRA_JOINT_VERTS], atol=0)


def test_lbs_global_rotation_commutes(smplx_data):
    bm = _bm(smplx_data)
    torch.manual_seed(1)
    pose = 0.3 * torch.randn(1, 55, 3)
    pose[:, 0] = 0
    betas, expr = torch.randn(1, 10), torch.randn(1, 10)
    base = _call(bm, pose, betas, expr)
    g = torch.tensor([[0.4, -0.7, 0.2]])
    pose2 = pose.clone()
    pose2[:, 0] = g
    rot = _call(bm, pose2, betas, expr)
    R = roma_ref.rotvec_to_rotmat(g)[0]
    root = base.joints[:, 0:1]          # global rotation is about the (rest) pelvis joint
    assert torch.allclose(rot.vertices, (base.vertices - root) @ R.T + root, atol=5e-6)


def test_lbs_one_hot_weights_single_joint(smplx_data):
    data = dict(smplx_data)
    W = np.zeros_like(data["weights"])
    owner = np.argmax(data["weights"], axis=1)
    W[np.arange(W.shape[0]), owner] = 1
    data["weights"] = W
    data["posedirs"] = np.zeros_like(data["posedirs"])
    bm = smplx_ref.SMPLX(data, num_betas=10)
    pose = torch.zeros(1, 55, 3)
    pose[0, 18] = torch.tensor([0.0, 0.0, 0.9])      # left elbow: descendants = wrist(20) + left hand (25..39)
    out = _call(bm, pose, torch.zeros(1, 10), torch.zeros(1, 10))
    rest = _call(bm, torch.zeros(1, 55, 3), torch.zeros(1, 10), torch.zeros(1, 10))
    moved = set([18, 20] + list(range(25, 40)))
    still = torch.from_numpy(~np.isin(owner, list(moved)))
    assert torch.allclose(out.vertices[0, still], rest.vertices[0, still], atol=1e-6)
    assert (out.vertices[0, ~still] - rest.vertices[0, ~still]).abs().max() > 1e-3


def _lbs_loops(data, pose55, betas, expr):
    """SMPL-X forward written a second time, differently on purpose: numpy float64, scipy's Rodrigues, the kinematic chain as a
    recursion over 4x4 matrices, the skinning as an explicit sum over (vertex, joint) pairs with non-zero weight.  One body."""
    from scipy.spatial.transform import Rotation
    v_t = np.asarray(data["v_template"], np.float64)
    sd = np.asarray(data["shapedirs"], np.float64)
    dirs = np.concatenate([sd[:, :, :10], sd[:, :, 300:310]], -1)                       # [V,3,20]
    coef = np.concatenate([betas, expr]).astype(np.float64)
    v_shaped = v_t + dirs @ coef
    J = np.asarray(data["J_regressor"], np.float64) @ v_shaped                          # rest joints [55,3]
    R = Rotation.from_rotvec(pose55.astype(np.float64)).as_matrix()                     # [55,3,3]
    pd = np.asarray(data["posedirs"], np.float64)                                       # [V,3,486]
    feat = np.concatenate([(R[j] - np.eye(3)).reshape(-1) for j in range(1, 55)])       # joint-major, row-major 3x3
    v_posed = v_shaped + pd @ feat
    parents = np.asarray(data["kintree_table"])[0].astype(int)

    def world(j):                                                                        # 4x4 world transform of joint j (rest -> posed)
        T = np.eye(4)
        T[:3, :3] = R[j]
        if j == 0:
            T[:3, 3] = J[0]
            return T
        T[:3, 3] = J[j] - J[parents[j]]
        return world(int(parents[j])) @ T

    G = [world(j) for j in range(55)]
    A = []
    for j in range(55):                                                                  # remove the rest pose: x -> G_j (x - J_j)
        Tj = G[j].copy()
        Tj[:3, 3] = G[j][:3, 3] - G[j][:3, :3] @ J[j]
        A.append(Tj)
    W = np.asarray(data["weights"], np.float64)
    verts = np.zeros_like(v_posed)
    for v in range(0, v_posed.shape[0], 1):
        for j in np.nonzero(W[v])[0]:
            verts[v] += W[v, j] * (A[j][:3, :3] @ v_posed[v] + A[j][:3, 3])
    joints = np.stack([G[j][:3, 3] for j in range(55)])
    return verts, joints


def test_lbs_restatement_matches_a_second_independent_implementation(smplx_data):
    """oracle/smplx_ref.py (the restated smplx.lbs, batched torch fp32) against the loop implementation above (numpy fp64, scipy
    Rodrigues): vertices, the 55 posed joints, the 21 picked vertices and the 51 barycentric landmarks."""
    bm = _bm(smplx_data)
    g = torch.Generator().manual_seed(11)
    pose = 0.4 * torch.randn(1, 55, 3, generator=g)
    betas, expr = torch.randn(1, 10, generator=g), torch.randn(1, 10, generator=g)
    out = _call(bm, pose, betas, expr)
    verts, joints = _lbs_loops(smplx_data, pose[0].numpy(), betas[0].numpy(), expr[0].numpy())
    assert np.abs(out.vertices[0].double().numpy() - verts).max() < 5e-6
    assert np.abs(out.joints[0, :55].double().numpy() - joints).max() < 5e-6
    faces = np.asarray(smplx_data["f"]).astype(np.int64)
    lf, lb = np.asarray(smplx_data["lmk_faces_idx"]).astype(np.int64), np.asarray(smplx_data["lmk_bary_coords"], np.float64)
    lmk = np.einsum("lk,lkc->lc", lb, verts[faces[lf]])
    assert np.abs(out.joints[0, 76:].double().numpy() - lmk).max() < 5e-6
    assert np.abs(out.joints[0, 55:76].double().numpy() - verts[list(smplx_ref.SMPLX_EXTRA_JOINT_VERTS)]).max() < 5e-6


def test_smplx_tables_of_oracle_and_product_agree():
    """oracle/smplx_ref.py and multi_hmr_amd/constants.py each hold their own literal copy of the smplx vertex ids / joint names
    (so that a wrong id on one side is a parity failure); this is the one place where the two are compared."""
    from multi_hmr_amd import constants
    from oracle import smplx_ref
    assert list(smplx_ref.SMPLX_EXTRA_JOINT_VERTS) == list(constants.SMPLX_EXTRA_JOINT_VERTS) and len(smplx_ref.SMPLX_EXTRA_JOINT_VERTS) == 21
    assert list(smplx_ref.SMPLX_JOINT_NAMES) == list(constants.SMPLX_JOINT_NAMES)
    assert smplx_ref.SMPLX_JOINT_NAMES[15] == "head" and smplx_ref.SMPLX_JOINT_NAMES[55] == "nose" and smplx_ref.SMPLX_JOINT_NAMES[0] == "pelvis"
