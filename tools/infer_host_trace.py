#!/usr/bin/env python
"""Where the inference-mode host share goes (GPU box): the headline batch through is_training=False with the per-phase host clocks of
Model._forward (MHMR_TRACE_HOST), person dicts made before vs after the count read-back (MHMR_LATE_DICTS), against the training hook
pinned to the same detections."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["MHMR_TRACE_HOST"] = "1"
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
hook = time_steps(lambda: model(x, idx=idx2, K=K, is_training=True), steps, 3, dev)
print(f"training hook, same detections: {1e3 * hook / steps:.3f} ms/step")
for late in (False, True, False, True):
    if late:
        os.environ["MHMR_LATE_DICTS"] = "1"
    else:
        os.environ.pop("MHMR_LATE_DICTS", None)
    model._host_trace = []
    run = lambda: model(x, K=K, det_thresh=thr, nms_kernel_size=3)
    dt = time_steps(run, steps, 3, dev)
    tr = model._host_trace[-steps:]
    a = lambda i: 1e3 * sum(t[i] for t in tr) / len(tr)
    gaps = [1e3 * (tr[i + 1][3] - tr[i][3]) for i in range(len(tr) - 1)]
    print(f"dicts {'after' if late else 'before'} the read-back: {1e3 * dt / steps:.3f} ms/step (+{1e3 * (dt - hook) / steps:.3f}); host: enqueue heads {a(0):.3f} ms, "
          f"dicts {a(1):.3f} ms, wait for the count {a(2):.3f} ms; step-to-step {sum(gaps) / len(gaps):.3f} ms")
# the same loop with the person list kept alive one step longer (deallocation of 2560 tensor views off the critical path?)
keepalive = []
def run_keep():
    keepalive.append(model(x, K=K, det_thresh=thr, nms_kernel_size=3))
    if len(keepalive) > 2:
        keepalive.pop(0)
os.environ.pop("MHMR_LATE_DICTS", None)
dt = time_steps(run_keep, steps, 3, dev)
print(f"dicts before, previous result kept alive during the next call: {1e3 * dt / steps:.3f} ms/step (+{1e3 * (dt - hook) / steps:.3f})")
# batched return (no person list at all)
dt = time_steps(lambda: model(x, K=K, det_thresh=thr, nms_kernel_size=3, return_batched=True), steps, 3, dev)
print(f"return_batched (no per-person dicts): {1e3 * dt / steps:.3f} ms/step (+{1e3 * (dt - hook) / steps:.3f})")
