#!/usr/bin/env python
"""Throughput of mhmr_preprocess_u8 (device-resident decoded frames -> normalised [3,S,S]) vs PIL on the host cores."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multi_hmr_amd import preprocess as pp
from oracle import preprocess_ref as ref
from PIL import Image

W, H, S, n = 1920, 1080, 896, 200
rng = np.random.default_rng(0)
img = rng.integers(0, 256, size=(H, W, 3), dtype=np.uint8)
pre = pp.Preprocessor(S, "cuda:0")
d = torch.from_numpy(img).cuda()
out = torch.empty(32, 3, S, S, device="cuda:0")
for i in range(4):
    pre(d, out=out[i % 32])
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(n):
    pre(d, out=out[i % 32])
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / n
bytes_alg = H * W * 3 + 3 * S * S * 4
print(f"GPU  {W}x{H} -> {S}: {ms*1e3:.1f} us/image, {1e3/ms:.0f} images/s, {bytes_alg/ms/1e6:.1f} GB/s algorithmic (in u8 + out f32)")
pil = Image.fromarray(img)
t0 = time.perf_counter()
for _ in range(5):
    ref.open_image_ref(pil, S)
t = (time.perf_counter() - t0) / 5
print(f"PIL (1 core) {t*1e3:.1f} ms/image, {1/t:.1f} images/s")
