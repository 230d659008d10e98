#!/usr/bin/env python
"""GPU side of the low-half weight pass experiment: relative L2 of the HIP path against the reference's golden
vectors at BASELINE.json's sizes, for several selections of projections that carry the low halves of their weights.
usage: python tools/wlo_study_gpu.py [--specs ",v+proj@0-11,v+proj"] [--cases vitl_672_full,...] > gpurun_out/wlo_study.json"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_golden  # noqa: E402
import parity  # noqa: E402
import synthetic  # noqa: E402
from multi_hmr_amd import Model  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--specs", default="none,v+proj@0-7,v+proj@0-11,v+proj@0-15,v+proj,proj")
    ap.add_argument("--cases", default="vits_672_full,vitl_672_full,vitl_896_full,vitl_1288_full")
    ap.add_argument("--precision", default="f16")
    a = ap.parse_args()
    sm, mp = synthetic.make_smplx_data(seed=0), synthetic.make_mean_params(seed=0)
    res = {}
    for name in a.cases.split(","):
        cfg = make_golden.CASES[name]
        gold = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
        vs = cfg.get("vstride", 1)
        x, K, idx = make_golden.case_inputs(cfg)
        model = Model(backbone=cfg["backbone"], img_size=cfg["img_size"], smplx_data=sm, mean_params=mp, precision=a.precision)
        model.load_state_dict(make_golden.case_state_dict(cfg), strict=True)
        model = model.to("cuda:0").eval()
        for spec in a.specs.split(","):
            model.wlo = "" if spec == "none" else spec.replace("|", ",")        # ("a|b" on the command line = the two-part spec "a,b")
            model.repack()
            z = model.backbone_features(x.cuda()).cpu()
            out = model(x.cuda(), idx=tuple(i.cuda() for i in idx), K=K.cuda(), is_training=True)
            row = {"backbone": parity.rel(z[:, :: max(1, z.shape[1] // 64)].numpy(), gold["backbone"])}
            for k in parity.CHECKED:
                g = out[k].cpu()
                if k in ("v3d", "v2d"):
                    g = g[:, ::vs]
                row[k] = parity.rel(g.numpy(), gold[k])
            res[f"{name}/{spec}"] = row
            print(f"{name:16s} {spec:14s} " + " ".join(f"{k}={v:.2e}" for k, v in row.items() if k in ("backbone", "scores", "offset", "shape", "expression", "rotmat", "transl", "v3d")),
                  file=sys.stderr, flush=True)
        del model
        torch.cuda.empty_cache()
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
