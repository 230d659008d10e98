#!/usr/bin/env python
"""Inference-mode host share (GPU box): the headline batch through is_training=False against the training hook pinned to the same
detections -- person list, batched return, and the person list with the previous result kept alive by the caller."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
import synthetic
from bench import build_model, make_inputs, time_steps

dev = torch.device("cuda", 0)
S, B, q = 896, 32, 8
sm, mp = synthetic.make_smplx_data(0), synthetic.make_mean_params(0)
model = build_model("dinov2_vitl14", S, "f16", sm, mp, dev)
x, K, idx = make_inputs(B, S, q, 0, dev)
out = model(x, idx=idx, K=K, is_training=True)
s = out["scores"][..., 0]
m = F.max_pool2d(s[:, None], 3, stride=1, padding=1)[:, 0]
surv = torch.sort(s[m == s], descending=True).values
n = min(B * q, surv.numel() - 1)
thr = float(0.5 * (surv[n - 1] + surv[n]))
keep = (m == s) & (s >= thr)
idx2 = tuple(torch.where(keep)) + (torch.zeros(int(keep.sum()), dtype=torch.long, device=dev),)
steps = 15
for rnd in range(2):
    hook = time_steps(lambda: model(x, idx=idx2, K=K, is_training=True), steps, 3, dev)
    print(f"training hook, same detections: {1e3 * hook / steps:.3f} ms/step")
    dt = time_steps(lambda: model(x, K=K, det_thresh=thr, nms_kernel_size=3), steps, 3, dev)
    print(f"person list: {1e3 * dt / steps:.3f} ms/step (+{1e3 * (dt - hook) / steps:.3f})")
    dt = time_steps(lambda: model(x, K=K, det_thresh=thr, nms_kernel_size=3, return_batched=True), steps, 3, dev)
    print(f"return_batched (no per-person dicts): {1e3 * dt / steps:.3f} ms/step (+{1e3 * (dt - hook) / steps:.3f})")
    keepalive = []
    def run_keep():
        keepalive.append(model(x, K=K, det_thresh=thr, nms_kernel_size=3))
        if len(keepalive) > 1:
            keepalive.pop(0)
    dt = time_steps(run_keep, steps, 3, dev)
    print(f"person list, previous result dropped only after the next call returned: {1e3 * dt / steps:.3f} ms/step (+{1e3 * (dt - hook) / steps:.3f})")
