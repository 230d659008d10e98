"""TEST INFRASTRUCTURE -- never imported by the product path.

CPU restatement of the per-match metric lines of the reference's evaluation loop (/root/reference/train.py:372-395 PVE and
PA-PVE; 415-423 the same two formulas on 14 regressed joints), with roma.rigid_points_registration restated in
oracle/roma_ref.py (roma is an unpinned pip dependency absent from /root/reference: requirements.txt:5).  Parity for these
formulas is unpinned by the reference (no vectors); the registration is cross-checked by known-answer tests (exact
similarity transforms are recovered, the optimum beats perturbed transforms) in tests/test_evaluate.py.  The matching /
detection-count functions (utils/training.py:9-195) ARE importable reference code: their goldens come from running the
reference's own file (tests/golden/make_golden_eval.py)."""
import torch

from . import roma_ref


def mesh_errors(v3d_hat, pelvis_hat, v3d, pelvis, dtype=torch.float64):
    """train.py:372-389 for one match -> (pve_mm, pa_pve_mm, R, t, s)."""
    v3d_ctx = (v3d - pelvis.reshape(1, 3)).to(dtype)
    v3d_hat_ctx = (v3d_hat - pelvis_hat.reshape(1, 3)).to(dtype)
    pve = (torch.sqrt(((v3d_ctx - v3d_hat_ctx) ** 2).sum(-1)) * 1000).mean()
    R, t, s = roma_ref.rigid_points_registration(v3d_hat_ctx, v3d_ctx, compute_scaling=True)
    pa = s * (R.reshape(1, 3, 3) @ v3d_hat_ctx.reshape(-1, 3, 1)).reshape(-1, 3) + t
    pa_pve = (torch.sqrt(((v3d_ctx - pa) ** 2).sum(-1)) * 1000).mean()
    return pve, pa_pve, R, t, s
