"""Load-time repacking of reference-format weights / SMPL-X arrays into the layouts the HIP kernels read.

Host-side, run once per (device, precision); nothing here is on the per-image path.
"""
from __future__ import annotations

import math
import numpy as np
import torch

from . import _lib

OP_DTYPES = {"bf16": (_lib.DT_BF16, torch.bfloat16), "f16": (_lib.DT_F16, torch.float16), "fp16": (_lib.DT_F16, torch.float16)}


def roundup(x: int, m: int) -> int:
    return (x + m - 1) // m * m


# ---------------------------------------------------------------------------------------------- pos-embed
def _cubic_weights(t: np.ndarray, A: float = -0.75) -> np.ndarray:
    """Keys cubic-convolution coefficients for taps at floor-1, floor, floor+1, floor+2."""
    def inner(x):   # |x| <= 1
        return ((A + 2) * x - (A + 3)) * x * x + 1
    def outer(x):   # 1 < |x| < 2
        return ((A * x - 5 * A) * x + 8 * A) * x - 4 * A
    return np.stack([outer(t + 1), inner(t), inner(1 - t), outer(2 - t)], axis=-1)


def interpolate_pos_embed(pos_embed: np.ndarray, G: int, offset: float = 0.1) -> np.ndarray:
    """DINOv2 ``interpolate_pos_encoding`` for a G x G grid: the M x M patch table is
    resampled bicubically (A=-0.75, align_corners=False, no antialias) with the source coordinate computed from
    the *given* scale factor (G+offset)/M -- not from G/M.  pos_embed: [1, 1+M*M, C] -> [1+G*G, C] (fp32)."""
    pe = np.asarray(pos_embed, dtype=np.float64)[0]
    n = pe.shape[0] - 1
    M = int(round(math.sqrt(n)))
    assert M * M == n
    if G == M:
        return pe.astype(np.float32)
    C = pe.shape[1]
    grid = pe[1:].reshape(M, M, C)
    scale = np.float32(float(G + offset) / M)          # torch keeps the scale as a C++ double made from a python float
    inv = 1.0 / float(G + offset) * M                  # 1 / scale_factor
    o = np.arange(G, dtype=np.float64)
    src = (o + 0.5) * inv - 0.5
    fl = np.floor(src)
    t = src - fl
    w = _cubic_weights(t)                               # [G,4]
    idx = np.clip(fl[:, None].astype(np.int64) + np.arange(-1, 3)[None, :], 0, M - 1)   # [G,4]
    rows = np.einsum("ya,yaxc->yxc", w, grid[idx])      # interpolate along y: [G, M, C]
    out = np.einsum("xa,yxac->yxc", w, rows[:, idx])    # then along x: [G, G, C]
    return np.concatenate([pe[:1], out.reshape(G * G, C)], axis=0).astype(np.float32)


# ---------------------------------------------------------------------------------------------- SMPL-X
#: vertices per workgroup tile of the vertex kernel (csrc/lbs.hip LBS_TV)
LBS_TILE = 48
#: joints 55..126 of the 127 the SMPL-X layer returns: 21 vertices picked by id + 51 barycentric face landmarks
N_EXTRA_JOINTS = 72


def pack_smplx(data: dict, num_betas: int, device, person_center_idx: int = 15) -> dict:  # person_center_idx < 0: no recentring
    """SMPL-X arrays (keys of SMPLX_NEUTRAL.npz) -> mhmr_lbs_consts tensors.

    * blend basis rows = [posedirs (486) | shapedirs[:, :, :nb] | shapedirs[:, :, 300:310] | 0-pad], scaled by 2^10 into the f16
      normal range, tile-major (48-vertex tiles = three MFMA column blocks; one 16-byte line per lane is the 16x16x32 MFMA operand for
      8 consecutive k): per tile the k blocks 0..Kb/8-9 (k < 448: pose correctives, millimetres) as f16 [Kb/8-8][3][48][8] -- the HIGH
      half alone, |error| <= 2^-12 |D| --, then the last eight k blocks (the last pose columns and every shape / expression direction)
      as an f16 PAIR hi + lo [8][hi|lo][3][48][8] (fp32 accuracy); the template stays fp32 ([3][Vp]);
    * the dense joint regressor is pre-contracted with the template and the blend shapes (J = J0 + JS.coef);
    * skinning weights: the dense [64, Vp] matrix as an f16 pair hi + lo in MFMA operand order (``skin16``; the kernel blends the
      joint transforms as a GEMM), plus the K-sparse (index, weight) list, K = max non-zeros per vertex (tools / tests);
    * the 72 extra joints (21 vertices picked by id + 51 barycentric face landmarks, smplx ``vertices2landmarks``) are VIRTUAL vertices:
      five more 48-vertex tiles after the real ones (from vertex ``Vl`` on) in which extra joint e = 16 t + i owns column i of the three
      16-vertex blocks of tile t -- copies of the basis / template / skinning columns of its three corner vertices -- so that one lane of
      the vertex kernel ends up with all three posed corners and combines them with ``xbary[e]`` ((1, 0, 0) for a picked vertex).
    """
    from .constants import SMPLX_EXTRA_JOINT_VERTS
    f64 = lambda a: np.asarray(a, dtype=np.float64)
    v_t = f64(data["v_template"])
    V = v_t.shape[0]
    sd = np.asarray(data["shapedirs"])
    shp = np.concatenate([f64(sd[:, :, :num_betas]), f64(sd[:, :, 300:310])], axis=-1)       # [V,3,nb+10]
    pd = f64(data["posedirs"])                                                                # [V,3,486]
    assert pd.shape[-1] == 486
    ncoef = num_betas + 10
    Kb = roundup(486 + ncoef, 32)
    # extra joints 55..126 as virtual vertices: corner k of extra joint e sits at Vl + 48 (e // 16) + 16 k + e % 16
    faces = np.asarray(data["f"], dtype=np.int64)
    lmk_vidx = faces[np.asarray(data["lmk_faces_idx"], dtype=np.int64)].astype(np.int32)       # [51,3]
    picked = np.asarray(SMPLX_EXTRA_JOINT_VERTS, dtype=np.int32)
    xcorner = np.concatenate([np.repeat(picked[:, None], 3, axis=1), lmk_vidx], axis=0)         # [72,3] source vertex of every corner
    xbary = np.concatenate([np.tile(np.array([[1.0, 0.0, 0.0]], dtype=np.float32), (len(picked), 1)),
                            np.asarray(data["lmk_bary_coords"], dtype=np.float32)], axis=0)     # [72,3]
    NX = xcorner.shape[0]
    assert NX == N_EXTRA_JOINTS
    Vl = roundup(V, LBS_TILE)
    Vp = Vl + LBS_TILE * ((NX + 15) // 16)
    src = np.full(Vp, -1, dtype=np.int64)                                                        # source vertex of every column (-1: zero padding)
    src[:V] = np.arange(V)
    e = np.arange(NX)
    for k in range(3):
        src[Vl + 48 * (e // 16) + 16 * k + e % 16] = xcorner[:, k]
    live = src >= 0
    D = np.zeros((Kb, 3, Vp), dtype=np.float64)
    D[:486, :, live] = pd.transpose(2, 1, 0)[:, :, src[live]]
    D[486:486 + ncoef, :, live] = shp.transpose(2, 1, 0)[:, :, src[live]]
    Ds = (D * 1024.0).astype(np.float32)
    hi = Ds.astype(np.float16)
    lo = (Ds - hi.astype(np.float32)).astype(np.float16)
    # dynamic range of the pair: |D| 2^10 must stay a NORMAL f16 below 65504, and hi + lo must reproduce the fp32 value to
    # 2^-22 relative (+ the f16 subnormal step where lo underflows) -- a body model with a wildly different scale fails here, loudly
    if not (np.isfinite(hi).all() and float(np.abs(Ds).max()) < 6.0e4):
        raise ValueError("blend basis x 2^10 leaves the f16 range")
    if not bool((np.abs(hi.astype(np.float64) + lo.astype(np.float64) - Ds) <= 2.0 ** -22 * np.abs(Ds) + 2.0 ** -25).all()):
        raise ValueError("blend basis: the f16 hi + lo pair does not reproduce the fp32 value")
    # tile-major: the slice of one 48-vertex tile is ONE contiguous 162 KiB block (the kernel DMAs it into LDS an eighth of the k
    # range at a time: whole DRAM pages instead of 768-byte pieces 168 KB apart): [Kb/8 - 8][3][48][8] high halves, then the last
    # eight k blocks as [8][hi|lo][3][48][8]
    T = LBS_TILE
    lay = lambda a: a.reshape(Kb // 8, 8, 3, Vp // T, T).transpose(3, 0, 2, 4, 1)           # [Vp/48, Kb/8, 3, 48, 8]
    nt, kp = Vp // T, Kb // 8 - 8
    if 8 * kp > 486:
        raise ValueError("the pair part of the blend basis (last 64 k) must hold every shape / expression direction")
    basis16 = np.ascontiguousarray(np.concatenate(
        [lay(hi)[:, :kp].reshape(nt, -1), np.stack([lay(hi)[:, kp:], lay(lo)[:, kp:]], axis=2).reshape(nt, -1)], axis=1))   # [Vp/48, 82944]
    vtemp = np.zeros((3, Vp), dtype=np.float32)
    vtemp[:, live] = v_t.T[:, src[live]]

    Jr = f64(data["J_regressor"])
    J0 = Jr @ v_t                                                                             # [55,3]
    JS = np.einsum("jv,vkl->jkl", Jr, shp)                                                    # [55,3,ncoef]

    W = f64(data["weights"])
    nnz = (W != 0).sum(axis=1)
    Kinf = max(1, int(nnz.max()))
    order = np.argsort(-np.abs(W), axis=1, kind="stable")[:, :Kinf]
    skin_idx = order.astype(np.int32)
    skin_w = np.take_along_axis(W, order, axis=1).astype(np.float32)

    # dense weights as the B operand of the skinning GEMM: [8 joint blocks][hi | lo][Vp][8], joints 55..63 and vertices >= V zero
    Wd = np.zeros((64, Vp), dtype=np.float32)
    Wd[: W.shape[1], live] = W.T.astype(np.float32)[:, src[live]]
    whi = Wd.astype(np.float16)
    wlo = (Wd - whi.astype(np.float32)).astype(np.float16)
    if not bool((np.abs(whi.astype(np.float64) + wlo.astype(np.float64) - Wd) <= 2.0 ** -22 * np.abs(Wd) + 2.0 ** -25).all()):
        raise ValueError("skinning weights: the f16 hi + lo pair does not reproduce the fp32 value")
    wlay = lambda a: a.reshape(8, 8, Vp // T, T).transpose(2, 0, 3, 1)                        # [Vp/48, 8, 48, 8]
    skin16 = np.ascontiguousarray(np.stack([wlay(whi), wlay(wlo)], axis=2))                   # [Vp/48, 8, 2, 48, 8]

    parents = np.asarray(data["kintree_table"])[0].astype(np.int64).copy()
    parents[0] = -1
    t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a)).to(device=device, dtype=dt)
    return {
        "V": V, "Vp": Vp, "Vl": Vl, "Kb": Kb, "nb": num_betas, "Kinf": Kinf, "center_joint": person_center_idx,
        "basis16": t(basis16, torch.float16), "vtemp": t(vtemp, torch.float32), "J0": t(J0, torch.float32), "JS": t(JS.reshape(55 * 3, ncoef), torch.float32),
        "pose_tasks": (lambda tk: t(tk[0], torch.int32) if tk[0] is not None else None)(pose_level_tasks(parents)),
        "pose_levels": pose_level_tasks(parents)[1],
        "parents": t(parents.astype(np.int32), torch.int32), "skin_idx": t(skin_idx, torch.int32), "skin_w": t(skin_w, torch.float32),
        "skin16": t(skin16, torch.float16),
        "xbary": t(xbary, torch.float32), "extra_vid": t(picked, torch.int32), "lmk_vidx": t(lmk_vidx, torch.int32),
        "lmk_bary": t(np.asarray(data["lmk_bary_coords"], dtype=np.float32), torch.float32),
        "faces": faces,
    }


def lbs_consts_struct(p: dict) -> "_lib.LbsConsts":
    c = _lib.LbsConsts()
    for k in ("V", "Vp", "Vl", "Kb", "nb", "Kinf", "center_joint"):
        setattr(c, k, int(p[k]))
    for k in ("basis16", "vtemp", "J0", "JS", "parents", "skin_idx", "skin_w", "skin16", "xbary"):
        setattr(c, k, p[k].data_ptr())
    # the kinematic tree's level schedule for the pose kernel (include/mhmr.h: pose_tasks), when the tree fits its fast path
    c.pose_tasks = p["pose_tasks"].data_ptr() if p.get("pose_tasks") is not None else None
    c.pose_levels = int(p.get("pose_levels", 0))
    return c


def pose_level_tasks(parents) -> "tuple[np.ndarray | None, int]":
    """parents [J] (root: -1) -> (int32 [16, 256] level schedule, number of levels), or (None, 0) when the tree does not fit the pose
    kernel's fast path (more than 16 levels or more than 21 joints on one level).  Lane 12 s + e of level L works on element e of the
    s-th joint (in joint order) of that level: value = joint | parent << 8, parent 0xff for a root; -1 = idle."""
    parents = [int(v) for v in parents]
    depth = []
    for j in range(len(parents)):
        d, a = 0, parents[j]
        while a >= 0:
            d, a = d + 1, parents[a]
        depth.append(d)
    nlev = max(depth) + 1
    if nlev > 16 or len(parents) > 255:
        return None, 0
    tasks = np.full((16, 256), -1, dtype=np.int32)
    for L in range(nlev):
        joints = [j for j in range(len(parents)) if depth[j] == L]
        if len(joints) * 12 > 256:
            return None, 0
        for s, j in enumerate(joints):
            pa = parents[j] if parents[j] >= 0 else 0xff
            tasks[L, 12 * s:12 * s + 12] = j | (pa << 8)
    return tasks, nlev
