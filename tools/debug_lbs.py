import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multi_hmr_amd import _lib, packing
import synthetic
from oracle import smplx_ref
from oracle.multihmr_ref import smpl_layer_forward
P = int(sys.argv[1])
center = None if len(sys.argv) > 2 and sys.argv[2] == "None" else 15
dev = torch.device("cuda:0"); L = _lib.lib()
data = synthetic.make_smplx_data(0)
lb = packing.pack_smplx(data, 10, dev, -1 if center is None else center); cs = packing.lbs_consts_struct(lb)
g = torch.Generator().manual_seed(P)
pose = 0.35 * torch.randn(P, 53, 3, generator=g); shape, expr = torch.randn(P, 10, generator=g), torch.randn(P, 10, generator=g)
det_b = torch.randint(0, 3, (P,), generator=g).sort().values
K = synthetic.get_camera_K(448, 3); loc = 448 * torch.rand(P, 2, generator=g); dist = 2 + 6 * torch.rand(P, 1, generator=g)
ref = smpl_layer_forward(smplx_ref.SMPLX(data, num_betas=10), pose, shape, loc, dist, K[det_b], expr, person_center_idx=center)
d = lambda t, dt=torch.float32: t.to(device=dev, dtype=dt).contiguous()
V = lb["V"]; f = lambda *s: torch.zeros(*s, device=dev)
v3d, v2d, j3d, j2d, tr = f(P, V, 3), f(P, V, 2), f(P, 127, 3), f(P, 127, 2), f(P, 3)
Pp = (P + 15) // 16 * 16
wsF, wsA, wsX = f(Pp, lb["Kb"]), f(Pp, 768), f(P, 24)
args = [d(pose), d(shape), d(expr), d(loc), d(dist), d(K), d(det_b, torch.int32)]
_lib.check(L.mhmr_lbs_forward(C.byref(cs), *[a.data_ptr() for a in args], P, wsF.data_ptr(), wsA.data_ptr(), wsX.data_ptr(), v3d.data_ptr(), v2d.data_ptr(),
                              j3d.data_ptr(), j2d.data_ptr(), tr.data_ptr(), torch.cuda.current_stream().cuda_stream), "lbs")
torch.cuda.synchronize()
F16 = wsF.view(torch.float16); A16 = wsA.view(torch.float16)
print("F16 nan", int(torch.isnan(F16.float()).sum()), "inf", int(torch.isinf(F16.float()).sum()), "A16 nan", int(torch.isnan(A16.float()).sum()), "inf", int(torch.isinf(A16.float()).sum()))
nanp = torch.isnan(v3d).any(dim=2)          # [P, V]
print("persons with NaN:", nanp.any(1).nonzero().flatten().tolist()[:40])
print("vertices with NaN (first 40):", nanp.any(0).nonzero().flatten().tolist()[:40], "count", int(nanp.any(0).sum()))
err = (v3d.cpu() - ref["v3d"]).abs()
err[torch.isnan(err)] = 0
print("max err (non-NaN)", float(err.max()), "at person", int(err.amax(dim=(1, 2)).argmax()))
pe = err.amax(dim=(1, 2))
print("per-person max err:", [f"{x:.1e}" for x in pe.tolist()][:80])

e2 = (v2d.cpu() - ref["v2d"]).abs()
print("v2d max err", float(e2.max()), "per-person", [f"{x:.1e}" for x in e2.amax(dim=(1, 2)).tolist()])
pw = int(e2.amax(dim=(1, 2)).argmax()); vw = int(e2[pw].amax(dim=1).argmax())
print("worst person", pw, "vertex", vw, "got", v2d[pw, vw].tolist(), "ref", ref["v2d"][pw, vw].tolist(), "v3d", v3d[pw, vw].tolist(), "ref v3d", ref["v3d"][pw, vw].tolist())
bad = (e2 > 0.01).any(dim=2)
print("bad (person, vertex) pairs:", bad.nonzero().tolist()[:40], "total", int(bad.sum()))
for pp, vv in bad.nonzero().tolist()[:10]:
    print(pp, vv, "l15", vv % 16, "tile", vv // 16, "got", v2d[pp, vv].tolist(), "ref", ref["v2d"][pp, vv].tolist(), "z", float(v3d[pp, vv, 2]))
