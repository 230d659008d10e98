#!/usr/bin/env python
"""Relative L2 of the HIP path vs the golden vectors produced by the reference's own code."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_golden
from multi_hmr_amd import Model
import synthetic
sm, mp = synthetic.make_smplx_data(seed=0), synthetic.make_mean_params(seed=0)
for name in ("vitl_224_train", "vitb_224_train", "vits_224_train"):
    cfg = make_golden.CASES[name]
    gold = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    x, K, idx = make_golden.case_inputs(cfg)
    for prec in ("f16", "bf16"):
        m = Model(backbone=cfg["backbone"], img_size=cfg["img_size"], smplx_data=sm, mean_params=mp, precision=prec, backbone_depth=cfg["depth_override"])
        m.load_state_dict(make_golden.case_state_dict(cfg), strict=True)
        m = m.to("cuda:0").eval()
        out = m(x.cuda(), idx=tuple(t.cuda() for t in idx), K=K.cuda(), is_training=True)
        feat = m.backbone_features(x.cuda())
        step = max(1, feat.shape[1] // 64)
        rel = lambda a, b: float(np.linalg.norm(a.astype(np.float64) - b) / np.linalg.norm(b.astype(np.float64)))
        row = {"backbone": rel(feat[:, ::step].cpu().numpy(), gold["backbone"])}
        for k in ("scores", "shape", "expression", "rotmat", "transl", "v3d", "j3d"):
            row[k] = rel(out[k].cpu().numpy(), gold[k])
        row["v3d_max_mm"] = float(np.abs(out["v3d"].cpu().numpy() - gold["v3d"]).max() * 1e3)
        print(f"{name:16s} {prec:5s} " + "  ".join(f"{k} {v:.2e}" if k != "v3d_max_mm" else f"{k} {v:.2f}" for k, v in row.items()))
