#!/usr/bin/env python
"""Image blocks of the backbone (Model(split=n): n blocks of B / n images on n streams) for the other BASELINE configurations:
   python tools/split_probe.py            # cfg2 / cfg3 / cfg5 at n = 1, 2, 4 (8), interleaved, two rounds
Prints images/s per (config, n); the automatic rule (Model._nsplit) is what `n = auto` resolves to."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import synthetic  # noqa: E402
from multi_hmr_amd import Model  # noqa: E402

dev = torch.device("cuda", 0)
sm, mp = synthetic.make_smplx_data(0), synthetic.make_mean_params(0)
CASES = [("cfg2 multiHMR_672_S", "dinov2_vits14", 672, 16, 8, (1, 2, 4, 8)), ("cfg3 multiHMR_672_L", "dinov2_vitl14", 672, 32, 8, (1, 2, 4)),
         ("cfg5 multiHMR_1288_L", "dinov2_vitl14", 1288, 8, 20, (1, 2, 4))]
only = sys.argv[1:] or None
out = []
for name, backbone, S, B, q, ns in CASES:
    if only and not any(o in name for o in only):
        continue
    sd = synthetic.make_state_dict(backbone, S, seed=0, mean_params=mp)
    x, K, idx = bench.make_inputs(B, S, q, 0, dev)
    models = {}
    for n in ns:
        m = Model(backbone=backbone, img_size=S, smplx_data=sm, mean_params=mp, precision="f16", split=n)
        m.load_state_dict(sd, strict=True)
        models[n] = m.to(dev).eval()
    auto = Model(backbone=backbone, img_size=S, smplx_data=sm, mean_params=mp, precision="f16")._nsplit(B)
    res = {n: [] for n in ns}
    for rnd in range(2):
        for n in ns:
            dt = bench.time_steps(lambda: models[n](x, idx=idx, K=K, is_training=True), 10, 3, dev)
            res[n].append(round(B * 10 / dt, 1))
    out.append({"config": name, "auto": auto, "images_per_s": res})
    print(json.dumps(out[-1]), flush=True)
    del models
    torch.cuda.empty_cache()
