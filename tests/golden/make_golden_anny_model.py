#!/usr/bin/env python
"""Golden vectors for the Anny-variant model up to the body model: runs the REFERENCE's own
/root/reference/multi_hmr_anny/multi_hmr.py (Multi_HMR) + encoder.py + hph.py + pos_embed.py verbatim on CPU with
  * torch.hub.load -> oracle.dinov2_ref, roma -> oracle.roma_ref (as for the Multi-HMR goldens, oracle/ref_shim.py),
  * the absent, unpinned ``anny`` package (requirements.txt:28) replaced by a stub body model that returns zeros --
    everything stored here is computed BEFORE the body model is called (multi_hmr.py:107-166).
Build container only.  Output: tests/golden/anny_model.npz."""
import os
import sys
import types

import numpy as np
import torch
from torch import nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import synthetic  # noqa: E402
from oracle import ref_shim  # noqa: E402

CASE = dict(backbone="dinov2_vits14", img_size=224, depth_override=3, xat_depth=3, batch=3, persons=[2, 0, 3], seed=7)


def case_inputs(cfg=CASE):
    S, B = cfg["img_size"], cfg["batch"]
    g = torch.Generator().manual_seed(3000 + cfg["seed"])
    x = torch.randn(B, 3, S, S, generator=g)
    G = S // 14
    bs, ys, xs = [], [], []
    for b, n in enumerate(cfg["persons"]):
        cells = torch.randperm(G * G, generator=g)[:n].sort().values
        bs.append(torch.full((n,), b, dtype=torch.long)); ys.append(cells // G); xs.append(cells % G)
    idx = (torch.cat(bs), torch.cat(ys), torch.cat(xs))
    K_user = synthetic.get_camera_K(S, B)
    K_user[:, 0, 2] += torch.arange(B) * 2.0
    return x, idx, K_user


def case_state_dict(cfg=CASE):
    return synthetic.make_state_dict_anny(cfg["backbone"], cfg["img_size"], xat_depth=cfg["xat_depth"], seed=cfg["seed"],
                                          depth_override=cfg["depth_override"])


class _StubBody(nn.Module):
    """Stands in for anny.create_fullbody_model(...): only the attributes multi_hmr.py touches."""
    def __init__(self):
        super().__init__()
        self.bone_labels = ["root", "head"] + [f"bone_{i}" for i in range(2, 163)]
        self.phenotype_labels = ["gender", "age", "muscle", "weight", "height", "proportions", "cupsize", "firmness", "african", "asian",
                                 "caucasian"]

    def set_skinning_method(self, m):
        pass

    def forward(self, pose_parameters, phenotype_kwargs):
        P = pose_parameters.shape[0]
        return {"vertices": torch.zeros(P, 8, 3), "bone_poses": pose_parameters, "blendshape_coeffs": torch.zeros(P, 1)}


def main():
    cfg = CASE
    smplx_data, mean_params = synthetic.make_smplx_data(seed=0), synthetic.make_mean_params(seed=0)
    with ref_shim.reference_modules(smplx_data, mean_params, cfg["depth_override"]):
        anny = types.ModuleType("anny")
        anny.create_fullbody_model = lambda **kw: _StubBody()
        sys.modules["anny"] = anny
        import multi_hmr_anny.multi_hmr as am
        model = am.Multi_HMR(img_size=cfg["img_size"], backbone=cfg["backbone"], xat_depth=cfg["xat_depth"], simple_depth_encoding=1)
        sd = case_state_dict(cfg)
        assert torch.equal(model.useful_rotmat.data, sd["useful_rotmat"]), "ANNY_USEFUL_ROTMAT differs from the reference"
        assert torch.equal(model.dec_pos_emb, sd["dec_pos_emb"]), float((model.dec_pos_emb - sd["dec_pos_emb"]).abs().max())
        assert torch.allclose(model.init_body_pose, sd["init_body_pose"], atol=1e-7), float((model.init_body_pose - sd["init_body_pose"]).abs().max())
        sd["init_body_pose"] = model.init_body_pose.clone()       # (roma's Rx(pi/2) carries cos(pi/2) = -4.4e-8; keep the reference's bits)
        ref_keys = {k for k in model.state_dict().keys() if not k.startswith("body_model.")}
        assert ref_keys == set(sd.keys()), (sorted(ref_keys - set(sd)), sorted(set(sd) - ref_keys))
        sd_ref = dict(sd)
        missing, unexpected = model.load_state_dict(sd_ref, strict=False)
        assert not unexpected and all(m.startswith("body_model.") for m in missing), (missing, unexpected)
        model.eval()
        x, idx, K_user = case_inputs(cfg)
        out = {}
        with torch.no_grad():
            for tag, K in (("", None), ("userK_", K_user)):
                res = model(x, K=K, idx=idx, is_training=True)
                for k in ("scores", "scores_logits", "K", "fov_regressed", "loc", "offset", "dist", "dist_postprocessed", "shape", "rotvec",
                          "rotmat", "transl"):
                    out[tag + k] = res[k].numpy()
                out[tag + "feat"] = res["feat"][:, ::4, ::4].numpy()
            # inference mode: detections by threshold + NMS, persons sorted by depth
            logits = torch.from_numpy(out["scores_logits"])
            thr = float(torch.sigmoid(logits).flatten().sort(descending=True).values[6:8].mean())
            persons = model(x, is_training=False, det_thresh=thr, nms_kernel_size=3)
            out["infer_thresh"] = np.float32(thr)
            out["infer_n"] = np.int64(len(persons))
            for k in ("loc", "transl", "rotvec", "shape"):
                out["infer_" + k] = torch.stack([p[k] for p in persons]).numpy()
        sys.modules.pop("anny", None)
        for k in [k for k in sys.modules if k.startswith("multi_hmr_anny")]:
            sys.modules.pop(k)
    path = os.path.join(HERE, "anny_model.npz")
    np.savez_compressed(path, **out)
    print({k: getattr(v, "shape", v) for k, v in out.items()}, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
