#!/usr/bin/env python
"""SMPL-X layer alone at P persons: wall per call of the layer + hipEvent time of its bracketed launch.  mhmr_lbs_forward (pose kernel + vertex kernel; the bracket is the
vertex kernel) or, with MHMR_LBS_FUSED=1, mhmr_lbs_forward_fused (one launch; the bracket is the whole layer)."""
import ctypes as C, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multi_hmr_amd import _lib, packing
import synthetic
if os.environ.get("MHMR_LIB"):
    _lib.LIB_PATH = os.environ["MHMR_LIB"]
P = int(sys.argv[1]) if len(sys.argv) > 1 else 160
dev = torch.device("cuda:0"); L = _lib.lib()
lb = packing.pack_smplx(synthetic.make_smplx_data(0), 10, dev); cs = packing.lbs_consts_struct(lb)
g = torch.Generator(device=dev).manual_seed(5)
f = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
pose = 0.3 * torch.randn(P, 53, 3, generator=g, device=dev)
shape, expr = torch.randn(P, 10, generator=g, device=dev), torch.randn(P, 10, generator=g, device=dev)
loc, dist = 1288 * torch.rand(P, 2, generator=g, device=dev), 2 + 6 * torch.rand(P, 1, generator=g, device=dev)
K = synthetic.get_camera_K(1288, 8).to(dev); det_b = (torch.arange(P, device=dev, dtype=torch.int32) * 8 // P).contiguous()
V = lb["V"]
bufs = [f((P + 15) // 16 * 16, lb["Kb"]), f((P + 15) // 16 * 16, 768), f(P, 24), f(P, V, 3), f(P, V, 2), f(P, 127, 3), f(P, 127, 2), f(P, 3)]
st = torch.cuda.current_stream(dev).cuda_stream
sync = torch.zeros(1 + (P + 15) // 16 * 16, dtype=torch.int32, device=dev)
_a = [C.byref(cs), pose.data_ptr(), shape.data_ptr(), expr.data_ptr(), loc.data_ptr(), dist.data_ptr(), K.data_ptr(), det_b.data_ptr(), P] + [b.data_ptr() for b in bufs]
_fused = os.environ.get("MHMR_LBS_FUSED", "0") == "1"
run = lambda: _lib.check(L.mhmr_lbs_forward_fused(*_a, sync.data_ptr(), st) if _fused else L.mhmr_lbs_forward(*_a, st), "lbs")
for _ in range(5): run()
torch.cuda.synchronize()
L.mhmr_prof_enable(2)
t0 = time.perf_counter()
for _ in range(50): run()
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / 50
n, ms, work = C.c_int(0), C.c_double(0), C.c_double(0)
L.mhmr_prof_collect(C.byref(n), C.byref(ms), C.byref(work)); L.mhmr_prof_enable(-1)
print(f"P={P} fused={int(_fused)}: layer {wall*1e3:.4f} ms ({wall*1e6/P:.3f} us/person), bracketed launch {ms.value/n.value*1e3:.1f} us")
