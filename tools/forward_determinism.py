#!/usr/bin/env python
"""The headline forward (ViT-L 896^2, 32 images, 8 pinned persons per image) REPS times on the same input: every output bit-equal to the first
run's?  (Round 6 added this after a class-row launch that read its predecessor's block sums went out as an any-order launch for one
session: results differed run to run.)  usage: python tools/forward_determinism.py [reps=30] [batch=32] [img=896]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import synthetic  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
S = int(sys.argv[3]) if len(sys.argv) > 3 else 896
dev = torch.device("cuda", 0)
model = bench.build_model("dinov2_vitl14", S, "f16", synthetic.make_smplx_data(0), synthetic.make_mean_params(0), dev)
x, K, idx = bench.make_inputs(B, S, 8, 0, dev)
KEYS = ("scores", "v3d", "rotmat", "transl", "shape", "expression", "j2d")
ref, bad = None, 0
for r in range(reps):
    out = model(x, idx=tuple(t.clone() for t in idx), K=K, is_training=True)
    cur = {k: out[k].clone() for k in KEYS}
    torch.cuda.synchronize()
    if ref is None:
        ref = cur
        continue
    diff = [k for k in KEYS if not torch.equal(ref[k], cur[k])]
    if diff:
        bad += 1
        if bad <= 3:
            print(f"run {r}: differs in {diff}; max |d| " + ", ".join(f"{k} {float((ref[k] - cur[k]).abs().max()):.2e}" for k in diff))
print(f"{S}^2 x {B}: {reps - 1} repeats, {bad} with a difference -> {'BIT-REPRODUCIBLE' if bad == 0 else 'NOT REPRODUCIBLE'} "
      f"(MHMR_ANYORDER={os.environ.get('MHMR_ANYORDER', 'default')}, MHMR_CLS_STATS={os.environ.get('MHMR_CLS_STATS', 'default')})")
sys.exit(1 if bad else 0)
