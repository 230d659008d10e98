#!/usr/bin/env python
"""What the f16x3 precision mode costs: throughput of the whole forward with hostile weights (precision='auto' -> f16x3) beside the
same shapes in plain f16 (seeded default weights), ViT-L 672^2.  One JSON line."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import synthetic  # noqa: E402
from multi_hmr_amd import Model  # noqa: E402


def run(B, S, backbone, hostile, precision, steps=5):
    dev = torch.device("cuda:0")
    sm, mp = synthetic.make_smplx_data(0), synthetic.make_mean_params(0)
    sd = synthetic.make_state_dict(backbone, S, seed=0, mean_params=mp)
    if hostile:
        synthetic.make_hostile(sd, "weights", seed=0)
    m = Model(backbone=backbone, img_size=S, smplx_data=sm, mean_params=mp, precision=precision)
    m.load_state_dict(sd, strict=True)
    m = m.to(dev).eval()
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn(B, 3, S, S, generator=g, device=dev)
    K = synthetic.get_camera_K(S, B).to(dev)
    idx = tuple(t.to(dev) for t in synthetic.make_pinned_idx(B, S // 14, 8, seed=0))
    for _ in range(2):
        m(x, idx=idx, K=K, is_training=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        m(x, idx=idx, K=K, is_training=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    return {"backbone": backbone, "S": S, "B": B, "hostile_weights": hostile, "requested": precision, "packed": m.packed_precision,
            "ms_per_step": round(1e3 * dt, 2), "images_per_s": round(B / dt, 2)}


if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    if len(sys.argv) > 2 and sys.argv[2] == "x3only":          # kernel traces: the f16x3 forward alone
        out = [run(B, 672, "dinov2_vitl14", True, "auto")]
    else:
        out = [run(B, 672, "dinov2_vitl14", True, "auto"), run(B, 672, "dinov2_vitl14", False, "auto"), run(B, 672, "dinov2_vitl14", True, "f16")]
    print(json.dumps(out))
