#!/usr/bin/env python
"""Run every heavy kernel repeatedly on identical inputs at the ViT-L 896 shapes and compare bit-for-bit."""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multi_hmr_amd import _lib
L = _lib.lib(); dev = torch.device("cuda:0"); st = torch.cuda.current_stream().cuda_stream
B = int(os.environ.get("B", "4")); C, H = 1024, 16; T = 4097; Tp = 4224; M = B * Tp   # B * Tp must be a multiple of 256
assert M % 256 == 0
dt, tdt = _lib.DT_BF16, torch.bfloat16
reps = int(os.environ.get("REPS", "6"))
def check(name, fn, out_fn):
    ref = None; bad = 0
    for i in range(reps):
        o = out_fn(); fn(o); torch.cuda.synchronize()
        if ref is None: ref = o.clone()
        elif not torch.equal(ref, o):
            d = (ref.float() - o.float()); nz = d.nonzero()
            bad += 1
            if bad == 1: print(f"   first mismatch: {nz.shape[0]} elements differ, max |d| {float(d.abs().max()):.3e}, rows {sorted(set((nz[:,0] if nz.dim()>1 else nz//1).tolist()))[:8]}")
    print(f"{name:34s} {'DETERMINISTIC' if bad == 0 else f'NON-DETERMINISTIC ({bad}/{reps-1})'}")
for name, N, K, epi in [("qk OP16", 2*C, C, _lib.EPI_OP16), ("v VT", C, C, _lib.EPI_VT), ("proj RESID", C, C, _lib.EPI_RESID), ("fc1 GELU", 4*C, C, _lib.EPI_OP16_GELU), ("fc2 RESID", C, 4*C, _lib.EPI_RESID), ("tokv F32 K1152", 512, 1152, _lib.EPI_F32), ("patch K640", C, 640, _lib.EPI_PATCH)]:
    A = (torch.randn(M, K, device=dev) * 0.5).to(tdt); W = (torch.randn(N, K, device=dev) / math.sqrt(K)).to(tdt)
    bias, gamma = torch.randn(N, device=dev), torch.randn(N, device=dev); pos = torch.randn(1 + 4096, N, device=dev)
    f32 = epi in (_lib.EPI_RESID, _lib.EPI_F32, _lib.EPI_PATCH)
    init = torch.randn(M + (B * Tp if epi == _lib.EPI_PATCH else 0), N, device=dev) if f32 else torch.zeros(M * N, dtype=tdt, device=dev)
    Mv = B * 4096 if epi == _lib.EPI_PATCH else M
    fn = lambda o: _lib.check(L.mhmr_gemm16(A.data_ptr(), K, W.data_ptr(), K, M, N, K, bias.data_ptr(), gamma.data_ptr(), o.data_ptr(), N, pos.data_ptr(), 4096, Tp, H, Mv, epi, dt, st), "g")
    check("gemm " + name, fn, lambda: init.clone())
Ma = B * Tp
qk = (torch.randn(Ma, 2 * C, device=dev) * 0.5).to(tdt); vt = (torch.randn(B * H * 64, Tp, device=dev) * 0.5).to(tdt)
check("attention", lambda o: _lib.check(L.mhmr_attention16(qk.data_ptr(), vt.data_ptr(), o.data_ptr(), B, T, Tp, C, H, dt, st), "a"), lambda: torch.zeros(Ma, C, dtype=tdt, device=dev))
x = torch.randn(Ma, C, device=dev); w_, b_ = torch.randn(C, device=dev), torch.randn(C, device=dev)
check("layernorm", lambda o: _lib.check(L.mhmr_layernorm16(x.data_ptr(), w_.data_ptr(), b_.data_ptr(), o.data_ptr(), Ma, C, 1e-6, dt, st), "l"), lambda: torch.zeros(Ma, C, dtype=tdt, device=dev))
