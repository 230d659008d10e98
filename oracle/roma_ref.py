"""CPU restatement of the four ``roma`` functions the reference path calls.  TEST INFRASTRUCTURE (oracle).

``roma`` is an unpinned pip dependency (reference requirements.txt:5) absent from /root/reference.
Call sites: ``special_gramschmidt`` utils/humans.py:21, ``rotmat_to_rotvec`` model.py:291,
``rotvec_to_rotmat`` blocks/smpl_layer.py:107.  Restated from the published algorithms
(SURVEY.md Appendix A.3); cross-checked against scipy in tests/test_oracle_roma.py.
"""
from __future__ import annotations

import torch


def special_gramschmidt(M: torch.Tensor, epsilon: float = 0.0) -> torch.Tensor:
    """[...,3,2] -> [...,3,3]: x=M[:,0]/|.|, y=(M[:,1]-(x.y)x)/|.|, z=x cross y, columns [x y z]."""
    x, y = M[..., 0], M[..., 1]
    x = x / torch.clamp_min(torch.norm(x, dim=-1, keepdim=True), epsilon)
    y = y - torch.sum(x * y, dim=-1, keepdim=True) * x
    y = y / torch.clamp_min(torch.norm(y, dim=-1, keepdim=True), epsilon)
    z = torch.cross(x, y, dim=-1)
    return torch.stack((x, y, z), dim=-1)


def rotvec_to_rotmat(rotvec: torch.Tensor, epsilon: float = 1e-6) -> torch.Tensor:
    """Rodrigues; theta=|v|, axis=v/max(theta,eps)."""
    shp = rotvec.shape[:-1]
    v = rotvec.reshape(-1, 3)
    theta = torch.norm(v, dim=-1)
    axis = v / theta.clamp_min(epsilon)[..., None]
    kx, ky, kz = axis[:, 0], axis[:, 1], axis[:, 2]
    s, c = torch.sin(theta), torch.cos(theta)
    omc = 1 - c
    xs, ys, zs = kx * s, ky * s, kz * s
    xyc, xzc, yzc = kx * ky * omc, kx * kz * omc, ky * kz * omc
    xxc, yyc, zzc = kx ** 2 * omc, ky ** 2 * omc, kz ** 2 * omc
    R = torch.stack([1 - yyc - zzc, xyc - zs, xzc + ys,
                     xyc + zs, 1 - xxc - zzc, -xs + yzc,
                     xzc - ys, xs + yzc, 1 - xxc - yyc], dim=-1).reshape(-1, 3, 3)
    return R.reshape(*shp, 3, 3)


def rotmat_to_unitquat(R: torch.Tensor) -> torch.Tensor:
    """XYZW unit quaternion; branch on the largest of (R00, R11, R22, trace) as scipy does."""
    shp = R.shape[:-2]
    m = R.reshape(-1, 3, 3)
    n = m.shape[0]
    dec = torch.empty((n, 4), dtype=m.dtype)
    dec[:, :3] = m.diagonal(dim1=1, dim2=2)
    dec[:, 3] = dec[:, :3].sum(dim=1)
    choices = dec.argmax(dim=1)
    q = torch.empty((n, 4), dtype=m.dtype)
    ind = torch.nonzero(choices != 3, as_tuple=True)[0]
    i = choices[ind]
    j = (i + 1) % 3
    k = (j + 1) % 3
    q[ind, i] = 1 - dec[ind, 3] + 2 * m[ind, i, i]
    q[ind, j] = m[ind, j, i] + m[ind, i, j]
    q[ind, k] = m[ind, k, i] + m[ind, i, k]
    q[ind, 3] = m[ind, k, j] - m[ind, j, k]
    ind = torch.nonzero(choices == 3, as_tuple=True)[0]
    q[ind, 0] = m[ind, 2, 1] - m[ind, 1, 2]
    q[ind, 1] = m[ind, 0, 2] - m[ind, 2, 0]
    q[ind, 2] = m[ind, 1, 0] - m[ind, 0, 1]
    q[ind, 3] = 1 + dec[ind, 3]
    q = q / torch.norm(q, dim=1, keepdim=True)
    return q.reshape(*shp, 4)


def unitquat_to_rotvec(quat: torch.Tensor) -> torch.Tensor:
    shp = quat.shape[:-1]
    q = quat.reshape(-1, 4).clone()
    q[q[:, 3] < 0] *= -1                           # shortest arc: w >= 0
    angle = 2 * torch.atan2(torch.norm(q[:, :3], dim=1), q[:, 3])
    small = torch.abs(angle) <= 1e-3
    scale = torch.empty_like(angle)
    a = angle[small]
    scale[small] = 2 + a ** 2 / 12 + 7 * a ** 4 / 2880
    a = angle[~small]
    scale[~small] = a / torch.sin(a / 2)
    return (scale[:, None] * q[:, :3]).reshape(*shp, 3)


def rotmat_to_rotvec(R: torch.Tensor) -> torch.Tensor:
    return unitquat_to_rotvec(rotmat_to_unitquat(R))


def special_procrustes(M: torch.Tensor, return_singular_values: bool = False):
    """roma.special_procrustes (restated from the published algorithm): R in SO(3) minimising |R - M|_F.
    M = U S V^T  ->  R = U diag(1, 1, det(U V^T)) V^T;  the returned singular values carry the same sign on the last one."""
    U, S, Vh = torch.linalg.svd(M)
    d = torch.det(U @ Vh)
    D = torch.ones_like(S)
    D[..., -1] = d
    R = (U * D.unsqueeze(-2)) @ Vh
    return (R, S * D) if return_singular_values else R


def rigid_points_registration(x: torch.Tensor, y: torch.Tensor, compute_scaling: bool = False):
    """roma.rigid_points_registration (restated; call sites train.py:384, 420): (R, t, s) minimising sum |s R x_n + t - y_n|^2
    over x, y [..., N, 3]."""
    xmean, ymean = x.mean(dim=-2, keepdim=True), y.mean(dim=-2, keepdim=True)
    xhat, yhat = x - xmean, y - ymean
    M = yhat.transpose(-1, -2) @ xhat
    if compute_scaling:
        R, DS = special_procrustes(M, return_singular_values=True)
        scale = DS.sum(-1) / (xhat ** 2).sum(dim=(-1, -2))
        t = ymean.squeeze(-2) - scale.unsqueeze(-1) * (R @ xmean.transpose(-1, -2)).squeeze(-1)
        return R, t, scale
    R = special_procrustes(M)
    t = ymean.squeeze(-2) - (R @ xmean.transpose(-1, -2)).squeeze(-1)
    return R, t, None
