#!/usr/bin/env python
"""Where does plain f16 leave the 1e-3 contract as the attention logits get steeper?  Seeded ViT-L weights with synthetic.make_hostile(kind=
"weights", strength=s) for a ladder of strengths: vit.logit_gain (max / mean over the blocks), then the forward in precision "f16" and
"f16x3" against the CPU fp32 oracle (reference model.py semantics) -- the data behind vit.LOGIT_GAIN_LIMIT.  One line per strength."""
import copy
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import synthetic  # noqa: E402
from multi_hmr_amd import Model, vit  # noqa: E402
from oracle.multihmr_ref import OracleModel  # noqa: E402

S = int(os.environ.get("PROBE_SIZE", "448"))
KEYS = ("scores", "offset", "dist", "shape", "expression", "rotmat", "transl", "v3d", "j3d")
sm, mp = synthetic.make_smplx_data(0), synthetic.make_mean_params(0)
sd0 = synthetic.make_state_dict("dinov2_vitl14", S, seed=31, mean_params=mp)
g = torch.Generator().manual_seed(5)
x = torch.randn(1, 3, S, S, generator=g)
K = synthetic.get_camera_K(S, 1)
idx = synthetic.make_pinned_idx(1, S // 14, 8, seed=3)
rel = lambda a, b: float(np.linalg.norm(a.astype(np.float64) - b) / max(np.linalg.norm(b.astype(np.float64)), 1e-30))
for st in [float(v) for v in (sys.argv[1:] or ["0.0", "0.3", "0.45", "0.55", "0.65", "0.8", "1.0"])]:
    sd = copy.deepcopy(sd0)
    if st > 0:
        synthetic.make_hostile(sd, "weights", seed=31, strength=st)
    ref = OracleModel(sd, sm, backbone="dinov2_vitl14", img_size=S).forward(x, idx=idx, K=K, is_training=True)
    row = {"strength": st}
    for prec in ("f16", "f16x3"):
        m = Model(backbone="dinov2_vitl14", img_size=S, smplx_data=sm, mean_params=mp, precision=prec)
        m.load_state_dict(sd, strict=True)
        if prec == "f16":
            gains = vit.logit_gain(m.backbone.encoder)
            row.update(gain_max=round(max(gains), 2), gain_mean=round(sum(gains) / len(gains), 2), auto=vit.resolve_precision(m.backbone.encoder, "auto"))
        m = m.cuda().eval()
        out = m(x.cuda(), idx=tuple(t.cuda() for t in idx), K=K.cuda(), is_training=True)
        errs = {k: rel(out[k].float().cpu().numpy(), ref[k].numpy()) for k in KEYS}
        row[prec] = {"worst": float("%.2e" % max(errs.values())), "worst_key": max(errs, key=errs.get), "v3d": float("%.2e" % errs["v3d"]),
                     "rotmat": float("%.2e" % errs["rotmat"]), "offset": float("%.2e" % errs["offset"])}
        del m
        torch.cuda.empty_cache()
    print(json.dumps(row), flush=True)
