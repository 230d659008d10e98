"""Anny-variant model up to the body model.  Goldens come from the reference's own multi_hmr_anny/multi_hmr.py
run on CPU with a stub for the absent `anny` body model (tests/golden/make_golden_anny_model.py); everything compared here is
computed before the body model is called."""
import os

import numpy as np
import pytest
import torch

import synthetic
from multi_hmr_amd.anny_model import Multi_HMR

HERE = os.path.dirname(__file__)
import sys
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_golden_anny_model as mg  # noqa: E402

GOLD = np.load(os.path.join(HERE, "golden", "anny_model.npz"))


def _model(precision="f16", **kw):
    cfg = mg.CASE
    m = Multi_HMR(img_size=cfg["img_size"], backbone=cfg["backbone"], xat_depth=cfg["xat_depth"], simple_depth_encoding=1,
                  backbone_depth=cfg["depth_override"], precision=precision, **kw)
    m.load_state_dict(mg.case_state_dict(cfg), strict=True)
    return m


def test_state_dict_surface_and_buffers():
    m = _model()
    sd = mg.case_state_dict()
    assert set(m.state_dict().keys()) == set(sd.keys())
    fresh = Multi_HMR(img_size=224, backbone="dinov2_vits14", xat_depth=1, simple_depth_encoding=1, backbone_depth=1)
    assert torch.equal(fresh.dec_pos_emb, sd["dec_pos_emb"])                   # sin-cos embedding restated from pos_embed.py
    assert torch.allclose(fresh.init_body_pose, sd["init_body_pose"], atol=1e-7)
    assert torch.equal(fresh.useful_rotmat.data, sd["useful_rotmat"])
    with pytest.raises(AssertionError):
        Multi_HMR(simple_depth_encoding=0)
    from multi_hmr_amd import _lib
    with pytest.raises(_lib.MhmrError):
        m(torch.zeros(1, 3, 224, 224))


def _rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-12))


@pytest.mark.gpu
@pytest.mark.parametrize("precision,tol", [("f16", 1e-3), ("bf16", 2e-2)])
def test_training_mode_matches_reference_golden(precision, tol):
    m = _model(precision).to("cuda:0").eval()
    x, idx, K_user = mg.case_inputs()
    for tag, K in (("", None), ("userK_", K_user.cuda())):
        out = m(x.cuda(), K=K, idx=tuple(t.cuda() for t in idx), is_training=True)
        for k in ("scores", "scores_logits", "K", "fov_regressed", "loc", "offset", "dist", "dist_postprocessed", "shape", "rotvec", "rotmat",
                  "transl"):
            e = _rel(out[k].cpu().numpy(), GOLD[tag + k])
            assert e < tol, (tag, k, e)
        assert _rel(out["feat"][:, ::4, ::4].cpu().numpy(), GOLD[tag + "feat"]) < tol
        assert torch.all(out["rotmat"][:, torch.tensor(synthetic.ANNY_USEFUL_ROTMAT) == 0] == torch.eye(3, device="cuda:0"))


@pytest.mark.gpu
def test_inference_mode_detects_sorts_and_calls_the_body_model():
    class Body:       # a stand-in with the interface multi_hmr.py uses (the real one is the `anny` package)
        bone_labels = ["root", "head"] + [f"b{i}" for i in range(2, 163)]
        phenotype_labels = ["gender", "age", "muscle", "weight", "height", "proportions", "x"]

        def __call__(self, pose_parameters, phenotype_kwargs):
            assert set(phenotype_kwargs) == {"gender", "age", "muscle", "weight", "height", "proportions"}
            P = pose_parameters.shape[0]
            v = torch.linspace(-0.3, 0.3, 30, device=pose_parameters.device).view(1, 10, 3).repeat(P, 1, 1)
            bp = pose_parameters.clone()
            bp[:, :, :3, 3] = torch.linspace(0, 1, 163, device=bp.device).view(1, 163, 1)
            return {"vertices": v, "bone_poses": bp, "blendshape_coeffs": None}

    m = _model("f16", body_model=Body()).to("cuda:0").eval()
    x, _, _ = mg.case_inputs()
    persons = m(x.cuda(), is_training=False, det_thresh=float(GOLD["infer_thresh"]), nms_kernel_size=3)
    assert len(persons) == int(GOLD["infer_n"])
    z = [float(p["transl"][2]) for p in persons]
    assert z == sorted(z)
    for k in ("loc", "transl", "rotvec", "shape"):
        assert _rel(torch.stack([p[k] for p in persons]).cpu().numpy(), GOLD["infer_" + k]) < 1e-3, k
    p0 = persons[0]
    assert p0["v3d"].shape == (10, 3) and p0["j3d"].shape == (163, 3) and p0["j2d"].shape == (163, 2)
    assert torch.allclose(p0["j3d"][1], p0["transl"], atol=1e-5)           # the person centre ('head') sits at the predicted translation
    assert m(x.cuda(), is_training=False, det_thresh=0.999) == []
