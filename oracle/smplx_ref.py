"""CPU restatement of ``smplx.create(..., 'smplx', use_pca=False, flat_hand_mean=True)`` -> ``SMPLX.forward``
-> ``lbs``.  TEST INFRASTRUCTURE (oracle).

``smplx`` is an unpinned pip dependency (reference requirements.txt:7) absent from /root/reference.
Reference call sites: construction blocks/smpl_layer.py:38, keyword mapping :85-101, call :104,
``output.vertices`` / ``output.joints`` :105-106, ``bm_x.faces`` demo.py:310.  Restated from the
published algorithm: dense-matmul formulation exactly as upstream writes it
(blend shapes einsum, dense J_regressor, dense [V,55] skinning-weight matmul, 4x4 homogeneous chain).
"""
from __future__ import annotations

import os
from types import SimpleNamespace
import numpy as np
import torch
from torch import nn

# The oracle keeps its OWN literal copy of the smplx tables (smplx/vertex_ids.py['smplx'], smplx/joint_names.py): a wrong id in the
# product's multi_hmr_amd/constants.py must show up as a parity failure, not be wrong on both sides
# (tests/test_oracle_thirdparty.py::test_smplx_tables_of_oracle_and_product_agree compares the two copies).
SMPLX_EXTRA_JOINT_VERTS = [
    9120,  # nose
    9929,  # reye
    9448,  # leye
    616,   # rear
    6,     # lear
    5770,  # LBigToe
    5780,  # LSmallToe
    8846,  # LHeel
    8463,  # RBigToe
    8474,  # RSmallToe
    8635,  # RHeel
    5361,  # lthumb
    4933,  # lindex
    5058,  # lmiddle
    5169,  # lring
    5286,  # lpinky
    8079,  # rthumb
    7669,  # rindex
    7794,  # rmiddle
    7905,  # rring
    8022,  # rpinky
]
SMPLX_JOINT_NAMES = [
    "pelvis", "left_hip", "right_hip", "spine1", "left_knee", "right_knee", "spine2", "left_ankle", "right_ankle", "spine3", "left_foot",
    "right_foot", "neck", "left_collar", "right_collar", "head", "left_shoulder", "right_shoulder", "left_elbow", "right_elbow",
    "left_wrist", "right_wrist", "jaw", "left_eye_smplhf", "right_eye_smplhf",
    "left_index1", "left_index2", "left_index3", "left_middle1", "left_middle2", "left_middle3", "left_pinky1", "left_pinky2",
    "left_pinky3", "left_ring1", "left_ring2", "left_ring3", "left_thumb1", "left_thumb2", "left_thumb3",
    "right_index1", "right_index2", "right_index3", "right_middle1", "right_middle2", "right_middle3", "right_pinky1", "right_pinky2",
    "right_pinky3", "right_ring1", "right_ring2", "right_ring3", "right_thumb1", "right_thumb2", "right_thumb3",
    "nose", "right_eye", "left_eye", "right_ear", "left_ear", "left_big_toe", "left_small_toe", "left_heel", "right_big_toe",
    "right_small_toe", "right_heel", "left_thumb", "left_index", "left_middle", "left_ring", "left_pinky", "right_thumb", "right_index",
    "right_middle", "right_ring", "right_pinky",
] + [f"face_landmark_{i}" for i in range(51)]                 # the first 127 of smplx.joint_names.JOINT_NAMES (utils/humans.py:25-26)
assert len(SMPLX_JOINT_NAMES) == 127

JOINT_NAMES = list(SMPLX_JOINT_NAMES) + [f"contour_{i}" for i in range(17)]  # smplx.joint_names.JOINT_NAMES

#: set by the test / golden harness when no SMPLX_NEUTRAL.npz exists on disk
DATA_OVERRIDE: dict | None = None


def batch_rodrigues(rot_vecs: torch.Tensor) -> torch.Tensor:
    """smplx.lbs.batch_rodrigues: angle = |v + 1e-8|, R = I + sin K + (1-cos) K^2."""
    n = rot_vecs.shape[0]
    angle = torch.norm(rot_vecs + 1e-8, dim=1, keepdim=True)
    rot_dir = rot_vecs / angle
    cos = torch.unsqueeze(torch.cos(angle), dim=1)
    sin = torch.unsqueeze(torch.sin(angle), dim=1)
    rx, ry, rz = torch.split(rot_dir, 1, dim=1)
    zeros = torch.zeros((n, 1), dtype=rot_vecs.dtype)
    K = torch.cat([zeros, -rz, ry, rz, zeros, -rx, -ry, rx, zeros], dim=1).view((n, 3, 3))
    ident = torch.eye(3, dtype=rot_vecs.dtype).unsqueeze(0)
    return ident + sin * K + (1 - cos) * torch.bmm(K, K)


def batch_rigid_transform(rot_mats, joints, parents):
    """smplx.lbs.batch_rigid_transform: world transforms along the kinematic tree and the
    rest-pose-removed 'relative' transforms A'_j = A_j - [0 | A_j J_j]."""
    B, J = joints.shape[:2]
    joints = torch.unsqueeze(joints, dim=-1)
    rel = joints.clone()
    rel[:, 1:] -= joints[:, parents[1:]]
    T = torch.zeros(B, J, 4, 4, dtype=joints.dtype)
    T[:, :, :3, :3] = rot_mats
    T[:, :, :3, 3:] = rel
    T[:, :, 3, 3] = 1
    chain = [T[:, 0]]
    for i in range(1, J):
        chain.append(torch.matmul(chain[int(parents[i])], T[:, i]))
    transforms = torch.stack(chain, dim=1)
    posed_joints = transforms[:, :, :3, 3]
    joints_h = torch.nn.functional.pad(joints, [0, 0, 0, 1])
    rel_transforms = transforms - torch.nn.functional.pad(torch.matmul(transforms, joints_h), [3, 0, 0, 0, 0, 0, 0, 0])
    return posed_joints, rel_transforms


def lbs(betas, pose, v_template, shapedirs, posedirs, J_regressor, parents, lbs_weights):
    """smplx.lbs.lbs with pose2rot=True.  posedirs is [486, V*3]."""
    B = max(betas.shape[0], pose.shape[0])
    v_shaped = v_template + torch.einsum("bl,mkl->bmk", betas, shapedirs)
    J = torch.einsum("bik,ji->bjk", v_shaped, J_regressor)
    ident = torch.eye(3, dtype=betas.dtype)
    rot_mats = batch_rodrigues(pose.view(-1, 3)).view(B, -1, 3, 3)
    pose_feature = (rot_mats[:, 1:] - ident).view(B, -1)
    v_posed = torch.matmul(pose_feature, posedirs).view(B, -1, 3) + v_shaped
    J_transformed, A = batch_rigid_transform(rot_mats, J, parents)
    W = lbs_weights.unsqueeze(0).expand(B, -1, -1)
    nj = J_regressor.shape[0]
    T = torch.matmul(W, A.view(B, nj, 16)).view(B, -1, 4, 4)
    v_h = torch.cat([v_posed, torch.ones(B, v_posed.shape[1], 1, dtype=betas.dtype)], dim=2)
    verts = torch.matmul(T, v_h.unsqueeze(-1))[:, :, :3, 0]
    return verts, J_transformed


def vertices2landmarks(vertices, faces, lmk_faces_idx, lmk_bary_coords):
    B = vertices.shape[0]
    lmk_faces = faces[lmk_faces_idx]                                  # [L,3]
    lmk_vertices = vertices[:, lmk_faces.reshape(-1)].view(B, -1, 3, 3)
    return torch.einsum("blfi,lf->bli", lmk_vertices, lmk_bary_coords)


class SMPLX(nn.Module):
    NUM_JOINTS = 55

    def __init__(self, data: dict, num_betas: int = 10, num_expression_coeffs: int = 10):
        super().__init__()
        f32 = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float32))
        self.num_betas = num_betas
        self.faces = np.asarray(data["f"], dtype=np.int64)
        self.register_buffer("faces_tensor", torch.from_numpy(self.faces))
        self.register_buffer("v_template", f32(data["v_template"]))
        sd = np.asarray(data["shapedirs"])
        self.register_buffer("shapedirs", f32(sd[:, :, :num_betas]))
        self.register_buffer("expr_dirs", f32(sd[:, :, 300:300 + num_expression_coeffs]))
        pd = np.asarray(data["posedirs"])
        self.register_buffer("posedirs", f32(pd.reshape(-1, pd.shape[-1]).T))    # [486, V*3]
        self.register_buffer("J_regressor", f32(data["J_regressor"]))
        self.register_buffer("lbs_weights", f32(data["weights"]))
        parents = torch.from_numpy(np.asarray(data["kintree_table"])[0].astype(np.int64)).clone()
        parents[0] = -1
        self.register_buffer("parents", parents)
        self.register_buffer("lmk_faces_idx", torch.from_numpy(np.asarray(data["lmk_faces_idx"], dtype=np.int64)))
        self.register_buffer("lmk_bary_coords", f32(data["lmk_bary_coords"]))
        self.register_buffer("extra_joints_idxs", torch.tensor(SMPLX_EXTRA_JOINT_VERTS, dtype=torch.long))
        z = lambda n: nn.Parameter(torch.zeros(1, n), requires_grad=False)
        self.global_orient, self.jaw_pose, self.leye_pose, self.reye_pose = z(3), z(3), z(3), z(3)
        self.body_pose, self.left_hand_pose, self.right_hand_pose = z(63), z(45), z(45)
        self.betas, self.expression, self.transl = z(num_betas), z(num_expression_coeffs), z(3)

    def forward(self, betas, global_orient, body_pose, left_hand_pose, right_hand_pose, jaw_pose,
                leye_pose, reye_pose, expression, **kw):
        full_pose = torch.cat([global_orient.reshape(-1, 1, 3), body_pose.reshape(-1, 21, 3),
                               jaw_pose.reshape(-1, 1, 3), leye_pose.reshape(-1, 1, 3), reye_pose.reshape(-1, 1, 3),
                               left_hand_pose.reshape(-1, 15, 3), right_hand_pose.reshape(-1, 15, 3)], dim=1)
        B = full_pose.shape[0]
        shape_components = torch.cat([betas, expression], dim=-1)
        shapedirs = torch.cat([self.shapedirs, self.expr_dirs], dim=-1)
        vertices, joints = lbs(shape_components, full_pose.reshape(B, -1), self.v_template, shapedirs,
                               self.posedirs, self.J_regressor, self.parents, self.lbs_weights)
        landmarks = vertices2landmarks(vertices, self.faces_tensor, self.lmk_faces_idx, self.lmk_bary_coords)
        joints = torch.cat([joints, vertices[:, self.extra_joints_idxs]], dim=1)   # VertexJointSelector
        joints = torch.cat([joints, landmarks], dim=1)
        joints = joints + self.transl.unsqueeze(1)
        vertices = vertices + self.transl.unsqueeze(1)
        return SimpleNamespace(vertices=vertices, joints=joints, full_pose=full_pose)


def create(model_path, model_type="smplx", gender="neutral", use_pca=False, flat_hand_mean=True,
           num_betas=10, **kwargs):
    assert model_type == "smplx" and not use_pca and flat_hand_mean
    if DATA_OVERRIDE is not None:
        data = DATA_OVERRIDE
    else:
        data = dict(np.load(os.path.join(model_path, "smplx", f"SMPLX_{gender.upper()}.npz"), allow_pickle=True))
    return SMPLX(data, num_betas=num_betas)
