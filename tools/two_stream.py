#!/usr/bin/env python
"""Experiment: ViT-L 896 backbone on 32 images as ONE batch vs TWO half batches on two HIP streams (tail filling across
kernels of independent halves)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multi_hmr_amd import Model
import synthetic

dev = torch.device("cuda:0")
sm, mp = synthetic.make_smplx_data(seed=0), synthetic.make_mean_params()
sd = synthetic.make_state_dict("dinov2_vitl14", 896, seed=0, mean_params=mp)
def mk():
    m = Model(backbone="dinov2_vitl14", img_size=896, smplx_data=sm, mean_params=mp, precision="bf16")
    m.load_state_dict(sd, strict=True)
    return m.to(dev).eval()
nparts = int(os.environ.get("PARTS", "2"))
ms = [mk() for _ in range(nparts)]
m0 = ms[0]
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(32, 3, 896, 896, generator=g, device=dev)
xs = list(x.chunk(nparts))
streams = [torch.cuda.Stream(dev) for _ in range(nparts)]

def run_one():
    m0.backbone_features(x)
def run_seq():
    for m, xi in zip(ms, xs):
        m.backbone_features(xi)
def run_par():
    cur = torch.cuda.current_stream(dev)
    for s, m, xi in zip(streams, ms, xs):
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            m.backbone_features(xi)
    for s in streams:
        cur.wait_stream(s)

def timeit(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
ref = m0.backbone_features(x).clone()
for name, fn in (("one batch of 32", run_one), (f"{nparts} parts sequential", run_seq), (f"{nparts} parts on {nparts} streams", run_par), ("one batch of 32", run_one)):
    print(f"{name:28s}: {timeit(fn):8.2f} ms  -> {32e3/timeit(fn):6.1f} img/s")
run_par(); torch.cuda.synchronize()
got = torch.cat([m.backbone_features(xi).clone() for m, xi in zip(ms, xs)])
run_par(); torch.cuda.synchronize()
got2 = torch.cat([m._workspace(m._packed, xi.shape[0])["feat32"].view(xi.shape[0], 4096, 1024).clone() for m, xi in zip(ms, xs)])
print("parts vs full batch: max rel diff", float((got - ref).abs().max() / ref.abs().max()), " par == seq bitwise:", torch.equal(got, got2))
