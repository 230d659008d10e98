#!/usr/bin/env python
"""Where the SMPL-X POSE kernel's time goes (round 6: four waves per person made it no faster, so it is not issue-bound): wall-clock
(100 MHz) stamps of a debug build of lbs.hip (-DMHMR_LBS_STAMPS -> tools/dbg/libmhmr_stamps.so), per person workgroup:
  0 start | 1-4 the four phase-A roles done (Rodrigues | joint regression | root + camera + tail | topology) | 5 first barrier passed |
  6 kinematic chain done | 7 recentring done | 8 read-outs done | 9 operand rows stored (issued) | 10 stores left the wave
and, from the vertex kernel, stamp 12 = its workgroups' start on the same clock (the gap between the two launches).
usage: bash tools/build_lbs_stamps.sh; MHMR_LIB=tools/dbg/libmhmr_stamps.so python tools/lbs_pose_timeline.py [P]"""
import ctypes as C, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multi_hmr_amd import packing
import synthetic
lib = C.CDLL(os.path.abspath(os.environ["MHMR_LIB"]))
P = int(sys.argv[1]) if len(sys.argv) > 1 else 160
dev = torch.device("cuda:0")
lb = packing.pack_smplx(synthetic.make_smplx_data(0), 10, dev); cs = packing.lbs_consts_struct(lb)
g = torch.Generator(device=dev).manual_seed(5)
f = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
pose = 0.3 * torch.randn(P, 53, 3, generator=g, device=dev)
shape, expr = torch.randn(P, 10, generator=g, device=dev), torch.randn(P, 10, generator=g, device=dev)
loc, dist = 1288 * torch.rand(P, 2, generator=g, device=dev), 2 + 6 * torch.rand(P, 1, generator=g, device=dev)
K = synthetic.get_camera_K(1288, 8).to(dev); det_b = (torch.arange(P, device=dev, dtype=torch.int32) * 8 // P).contiguous()
V = lb["V"]; Pp = (P + 15) // 16 * 16
bufs = [f(Pp, lb["Kb"]), f(Pp, 768), f(P, 24), f(P, V, 3), f(P, V, 2), f(P, 127, 3), f(P, 127, 2), f(P, 3)]
st = torch.cuda.current_stream(dev).cuda_stream
lib.mhmr_lbs_forward.argtypes = [C.c_void_p] * 8 + [C.c_int] + [C.c_void_p] * 9
run = lambda: lib.mhmr_lbs_forward(C.byref(cs), pose.data_ptr(), shape.data_ptr(), expr.data_ptr(), loc.data_ptr(), dist.data_ptr(),
                                   K.data_ptr(), det_b.data_ptr(), P, *[b.data_ptr() for b in bufs], st)
nwg = lb["Vp"] // 48
ps = torch.zeros(Pp, 16, dtype=torch.int64, device=dev)
vs = torch.zeros(nwg, 16, dtype=torch.int64, device=dev)
for _ in range(5):
    assert run() == 0
torch.cuda.synchronize()
for fn, t in (("mhmr_debug_pose_stamps", ps), ("mhmr_debug_lbs_stamps", vs)):
    getattr(lib, fn).argtypes = [C.c_void_p]
    assert getattr(lib, fn)(t.data_ptr()) == 0
for rep in range(3):
    ps.zero_(); vs.zero_()
    assert run() == 0
    torch.cuda.synchronize()
    t = ps.cpu().numpy().astype(np.float64)[:P] * 0.01          # us
    v = vs.cpu().numpy().astype(np.float64)[:, 12] * 0.01
    t0 = t[:, 0].min()
    names = ["start", "A: rodrigues", "A: joints", "A: root/cam", "A: topology", "barrier 1", "chain done", "recentred", "read-outs", "stores issued", "stores left"]
    print(f"-- run {rep}, P={P}: us since the first pose workgroup's start (min / median / max over the {P} person workgroups)")
    for i, n in enumerate(names):
        c = t[:, i] - t0
        print(f"   {n:14s} {c.min():7.2f} {np.median(c):7.2f} {c.max():7.2f}")
    print(f"   vertex kernel workgroups start: {v.min() - t0:7.2f} .. {v.max() - t0:7.2f}   (pose kernel's last stamp {t[:, 10].max() - t0:.2f})")
