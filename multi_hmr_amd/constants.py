"""Fixed tables of the path: SMPL-X topology (smplx package), the DINOv2 backbone sizes, the Anny head's fixed buffers and the
demo's default camera.  Seeded stand-ins for the released checkpoints / body-model files live OUTSIDE the package (synthetic.py at
the repository root: test and benchmark infrastructure)."""
from __future__ import annotations

import math

import numpy as np
import torch

# ----------------------------------------------------------------------------------------------
# SMPL-X topology constants (smplx package)
# ----------------------------------------------------------------------------------------------
SMPLX_NUM_VERTS = 10475
SMPLX_NUM_FACES = 20908
SMPLX_NUM_JOINTS = 55

SMPLX_PARENTS = [
    -1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 15, 15, 15,
    20, 25, 26, 20, 28, 29, 20, 31, 32, 20, 34, 35, 20, 37, 38,
    21, 40, 41, 21, 43, 44, 21, 46, 47, 21, 49, 50, 21, 52, 53,
]

# vertex ids of the 21 "extra" joints appended by smplx's VertexJointSelector (smplx/vertex_ids.py)
SMPLX_EXTRA_JOINT_VERTS = [
    9120, 9929, 9448, 616, 6,            # nose, reye, leye, rear, lear
    5770, 5780, 8846, 8463, 8474, 8635,  # LBigToe, LSmallToe, LHeel, RBigToe, RSmallToe, RHeel
    5361, 4933, 5058, 5169, 5286,        # lthumb, lindex, lmiddle, lring, lpinky
    8079, 7669, 7794, 7905, 8022,        # rthumb, rindex, rmiddle, rring, rpinky
]

_BODY = [
    "pelvis", "left_hip", "right_hip", "spine1", "left_knee", "right_knee", "spine2", "left_ankle",
    "right_ankle", "spine3", "left_foot", "right_foot", "neck", "left_collar", "right_collar", "head",
    "left_shoulder", "right_shoulder", "left_elbow", "right_elbow", "left_wrist", "right_wrist",
    "jaw", "left_eye_smplhf", "right_eye_smplhf",
]
_FINGERS = ["index", "middle", "pinky", "ring", "thumb"]
_HANDS = [f"{s}_{f}{i}" for s in ("left", "right") for f in _FINGERS for i in (1, 2, 3)]
_EXTRA = [
    "nose", "right_eye", "left_eye", "right_ear", "left_ear", "left_big_toe", "left_small_toe", "left_heel",
    "right_big_toe", "right_small_toe", "right_heel", "left_thumb", "left_index", "left_middle", "left_ring",
    "left_pinky", "right_thumb", "right_index", "right_middle", "right_ring", "right_pinky",
]
_LMK = [f"face_landmark_{i}" for i in range(51)]
#: first 127 entries of smplx.joint_names.JOINT_NAMES (reference utils/humans.py:25-26)
SMPLX_JOINT_NAMES = _BODY + _HANDS + _EXTRA + _LMK
assert len(SMPLX_JOINT_NAMES) == 127 and SMPLX_JOINT_NAMES[15] == "head" and SMPLX_JOINT_NAMES[55] == "nose"


VIT_CFG = {
    "dinov2_vits14": dict(embed_dim=384, depth=12, num_heads=6),
    "dinov2_vitb14": dict(embed_dim=768, depth=12, num_heads=12),
    "dinov2_vitl14": dict(embed_dim=1024, depth=24, num_heads=16),
}


ANNY_NUM_JOINTS = 163
#: multi_hmr_anny/multi_hmr.py:78-88 (which of the 163 bone rotations are predicted; the others are forced to identity)
ANNY_USEFUL_ROTMAT = ([1.] * 7 + [0.] * 14 + [1.] * 6 + [0.] * 14 + [1.] * 4 + [0.] * 2 + [1.] * 58 + [0.] * 58)


def anny_sincos_pos_embed(embed_dim: int, grid_size: int) -> np.ndarray:
    """multi_hmr_anny/pos_embed.py:12-61 (2D sine-cosine embedding, w goes first, no cls token): [grid*grid, embed_dim] float64.
    First half of the channels encodes the x (column) index, second half the y (row) index; each half = [sin | cos] over
    embed_dim/4 frequencies 1 / 10000^(i / (embed_dim/4))."""
    assert embed_dim % 4 == 0
    gw, gh = np.meshgrid(np.arange(grid_size, dtype=np.float32), np.arange(grid_size, dtype=np.float32))

    def one(pos):
        half = embed_dim // 2
        omega = np.arange(half // 2, dtype=float)
        omega /= half / 2.0
        omega = 1.0 / 10000 ** omega
        out = np.einsum("m,d->md", pos.reshape(-1), omega)
        return np.concatenate([np.sin(out), np.cos(out)], axis=1)
    return np.concatenate([one(gw), one(gh)], axis=1)


def anny_init_body_pose() -> torch.Tensor:
    """multi_hmr_anny/multi_hmr.py:90-96: root = the first two columns of Rx(pi/2), the other 162 bones = those of I; [1, 978]."""
    c, s_ = math.cos(math.pi / 2), math.sin(math.pi / 2)
    Rx = torch.tensor([[1.0, 0.0, 0.0], [0.0, c, -s_], [0.0, s_, c]])
    root = Rx[:, :2].reshape(1, -1)
    body = torch.eye(3).reshape(1, 3, 3).repeat(ANNY_NUM_JOINTS - 1, 1, 1)[:, :, :2].flatten(1).reshape(1, -1)
    return torch.cat([root, body], -1)


def get_camera_K(img_size: int, batch: int = 1, fov: float = 60.0) -> torch.Tensor:
    """K of reference demo.py:53-68 (fx=fy=S/(2 tan(fov/2)), principal point S//2), repeated."""
    K = torch.eye(3)
    focal = img_size / (2 * np.tan(np.radians(fov) / 2))
    K[0, 0], K[1, 1] = focal, focal
    K[0, -1], K[1, -1] = img_size // 2, img_size // 2
    return K.unsqueeze(0).repeat(batch, 1, 1)
