"""Run the reference's OWN model code (/root/reference/model.py, blocks/, utils/) unmodified on CPU.

TEST INFRASTRUCTURE (oracle).  Works only where /root/reference exists (the build container, never the
GPU box).  The reference imports three packages that are not installed and not vendored -- ``roma``,
``smplx`` (+ ``smplx.joint_names``) and, through utils/render.py, ``pyrender``/``trimesh`` -- and pulls
the backbone from ``torch.hub`` (network).  This shim registers the restatements in ``oracle/`` under
those names in ``sys.modules``, patches ``torch.hub.load`` to build ``oracle.dinov2_ref``, provides the
cwd-relative asset ``models/smpl_mean_params.npz`` (utils/constants.py:8) in a scratch directory, and
then imports the reference modules by path.  Nothing is copied from the reference.
"""
from __future__ import annotations

import contextlib
import os
import sys
import tempfile
import types

import numpy as np
import torch

REFERENCE_ROOT = "/root/reference"


def available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "model.py"))


def _install_stubs(smplx_data: dict):
    from oracle import roma_ref, smplx_ref, dinov2_ref

    roma = types.ModuleType("roma")
    for n in ("special_gramschmidt", "rotvec_to_rotmat", "rotmat_to_rotvec"):
        setattr(roma, n, getattr(roma_ref, n))
    sys.modules["roma"] = roma

    smplx = types.ModuleType("smplx")
    smplx.create = smplx_ref.create
    jn = types.ModuleType("smplx.joint_names")
    jn.JOINT_NAMES = smplx_ref.JOINT_NAMES
    smplx.joint_names = jn
    sys.modules["smplx"] = smplx
    sys.modules["smplx.joint_names"] = jn
    smplx_ref.DATA_OVERRIDE = smplx_data

    for name in ("pyrender", "trimesh"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)

    def hub_load(repo, name, pretrained=False, **kw):
        assert repo == "facebookresearch/dinov2", repo
        return dinov2_ref.build(name, depth_override=_DEPTH_OVERRIDE[0])

    torch.hub.load = hub_load


_DEPTH_OVERRIDE = [None]


@contextlib.contextmanager
def reference_modules(smplx_data: dict, mean_params: dict, depth_override: int | None = None):
    """Context manager yielding the imported reference ``model`` module (``model.Model`` etc.).

    The cwd is a scratch dir containing ``models/smpl_mean_params.npz`` while the context is active,
    because the reference reads that asset cwd-relative inside ``HPH.__init__`` (model.py:442)."""
    assert available(), "reference tree not present (this only runs in the build container)"
    _DEPTH_OVERRIDE[0] = depth_override
    _install_stubs(smplx_data)
    tmp = tempfile.mkdtemp(prefix="mhmr_ref_")
    os.makedirs(os.path.join(tmp, "models"), exist_ok=True)
    np.savez(os.path.join(tmp, "models", "smpl_mean_params.npz"), **mean_params)
    old_cwd = os.getcwd()
    # the reference's top-level names are generic ('model', 'utils', 'blocks'): isolate them
    saved = {k: sys.modules.pop(k) for k in list(sys.modules) if k in ("model", "utils", "blocks")
             or k.startswith(("utils.", "blocks."))}
    sys.path.insert(0, REFERENCE_ROOT)
    os.chdir(tmp)
    try:
        import model as ref_model  # noqa: the reference's model.py
        yield ref_model
    finally:
        os.chdir(old_cwd)
        sys.path.remove(REFERENCE_ROOT)
        for k in [k for k in sys.modules if k in ("model", "utils", "blocks") or k.startswith(("utils.", "blocks."))]:
            sys.modules.pop(k)
        sys.modules.update(saved)
