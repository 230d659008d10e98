import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multi_hmr_amd import _lib
L = _lib.lib(); dev = torch.device("cuda:0"); st = torch.cuda.current_stream().cuda_stream
dt, tdt = _lib.DT_BF16, torch.bfloat16
def run(B, H, T, reps=8):
    C = H * 64; Tp = (T + 127) // 128 * 128; M = B * Tp
    g = torch.Generator(device=dev).manual_seed(0)
    qk = (torch.randn(M, 2 * C, device=dev, generator=g) * 0.5).to(tdt); vt = (torch.randn(B * H * 64, Tp, device=dev, generator=g) * 0.5).to(tdt)
    ref = None; nbad = 0; info = ""
    for i in range(reps):
        o = torch.zeros(M, C, dtype=tdt, device=dev)
        _lib.check(L.mhmr_attention16(qk.data_ptr(), vt.data_ptr(), o.data_ptr(), B, T, Tp, C, H, dt, st), "a"); torch.cuda.synchronize()
        if ref is None: ref = o
        elif not torch.equal(ref, o):
            nbad += 1
            if not info:
                d = (ref.float() - o.float()).abs()
                rows = (d.amax(1) > 0).nonzero().flatten()
                cols = (d.amax(0) > 0).nonzero().flatten()
                info = f" rows differing {rows.numel()} (first {rows[:6].tolist()} mod128 {sorted(set((rows%128).tolist()))[:10]}), heads {sorted(set((cols//64).tolist()))[:8]}, max {float(d.max()):.2e}"
    print(f"B={B} H={H} T={T}: {nbad}/{reps-1} mismatching runs;{info}")
import os
if os.environ.get('BIG'):
    run(32, 16, 4097, reps=int(os.environ['BIG']))
else:
    run(1, 1, 128); run(1, 1, 4097); run(1, 16, 4097); run(4, 16, 4096); run(3, 16, 4097); run(8, 16, 2305)
