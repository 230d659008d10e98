#!/usr/bin/env python
"""The first thing to run when a RELEASED checkpoint is at hand (none is, offline): what does `precision="auto"` make of it, and what would
forcing the fast path cost?

  python tools/checkpoint_report.py models/multiHMR/multiHMR_896_L.pt            # a reference checkpoint ({args, model_state_dict})
  python tools/checkpoint_report.py --synthetic dinov2_vitl14 --hostile weights   # the seeded stand-ins (what the tests use)

Prints, per block, vit.logit_gain (the predicted spread of the pre-softmax logits over the keys of one query, natural-log units; the rule
switches to the f16x3 mode above vit.LOGIT_GAIN_LIMIT) and the precision `auto` resolves to.  With --compare (needs the MI355X) it also
runs one seeded image through precision="f16" and precision="f16x3" and prints the relative L2 of every output between the two: f16x3 sits
at fp32 accuracy (tests/test_gpu_x3.py), so that column IS the plain-f16 error on these weights -- against the 1e-3 contract."""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import synthetic  # noqa: E402
from multi_hmr_amd import Model, vit  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("checkpoint", nargs="?", help="reference checkpoint (.pt with args + model_state_dict)")
    ap.add_argument("--synthetic", help="backbone name: seeded stand-in weights instead of a checkpoint")
    ap.add_argument("--hostile", choices=["weights", "mean"], help="synthetic.make_hostile on the stand-in")
    ap.add_argument("--img-size", type=int, default=448)
    ap.add_argument("--compare", action="store_true", help="run f16 against f16x3 on one seeded image (GPU)")
    a = ap.parse_args()
    sm, mp = synthetic.make_smplx_data(0), synthetic.make_mean_params(0)
    if a.synthetic:
        backbone, S = a.synthetic, a.img_size
        sd = synthetic.make_state_dict(backbone, S, seed=0, mean_params=mp)
        if a.hostile:
            synthetic.make_hostile(sd, a.hostile, seed=0)
        kw = dict(backbone=backbone, img_size=S)
    else:
        ck = torch.load(a.checkpoint, map_location="cpu", weights_only=False)
        kw = dict(vars(ck["args"]))
        kw["img_size"] = kw["img_size"][0] if isinstance(kw["img_size"], (list, tuple)) else kw["img_size"]
        sd, S = ck["model_state_dict"], kw["img_size"]
    build = lambda prec: Model(**dict(kw, smplx_data=sm, mean_params=mp, precision=prec))
    m = build("auto")
    m.load_state_dict(sd, strict=False)
    gains = vit.logit_gain(m.backbone.encoder)
    print("block  logit spread (limit %.1f at <= 4097 tokens per image, %.2f at 1288^2)" % (vit.LOGIT_GAIN_LIMIT, vit.logit_gain_limit(8465)))
    for i, gval in enumerate(gains):
        print("%5d  %8.2f %s" % (i, gval, "  <-- steep" if gval > vit.LOGIT_GAIN_LIMIT else ""))
    print("precision='auto' packs these weights as:", vit.resolve_precision(m.backbone.encoder, "auto"))
    if not a.compare:
        return
    assert torch.cuda.is_available(), "--compare needs the MI355X"
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1, 3, S, S, generator=g).cuda()
    K = synthetic.get_camera_K(S, 1).cuda()
    idx = tuple(t.cuda() for t in synthetic.make_pinned_idx(1, S // 14, 6, seed=0))
    outs = {}
    for prec in ("f16", "f16x3"):
        mm = build(prec)
        mm.load_state_dict(sd, strict=False)
        mm = mm.cuda().eval()
        outs[prec] = {k: v.float().cpu().numpy() for k, v in mm(x, idx=idx, K=K, is_training=True).items()}
        del mm
        torch.cuda.empty_cache()
    rel = lambda p, q: float(np.linalg.norm(p.astype(np.float64) - q) / max(np.linalg.norm(q.astype(np.float64)), 1e-30))
    print("relative L2 of precision='f16' against precision='f16x3' (the 1e-3 contract applies to the f16 column of DESIGN.md section 3):")
    for k in ("scores", "offset", "dist", "shape", "expression", "rotmat", "transl", "v3d", "j3d"):
        print("  %-12s %.2e" % (k, rel(outs["f16"][k], outs["f16x3"][k])))


if __name__ == "__main__":
    main()
