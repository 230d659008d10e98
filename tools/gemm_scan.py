#!/usr/bin/env python
"""K-scan of the 256x256 GEMM: time = a + b*K separates the per-K-tile cost from the per-tile epilogue."""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multi_hmr_amd import _lib
from kbench import timeit
L = _lib.lib(); dev = torch.device("cuda:0"); st = torch.cuda.current_stream().cuda_stream
M, Tp, H = 135168, 4224, 16
dt, tdt = _lib.DT_BF16, torch.bfloat16
for epi, name in ((_lib.EPI_OP16, "OP16"), (_lib.EPI_RESID, "RESID"), (_lib.EPI_OP16_GELU, "GELU")):
    for N in (1024, 4096):
        res = []
        for K in (512, 1024, 2048, 4096):
            A = (torch.randn(M, K, device=dev) * 0.5).to(tdt); W = (torch.randn(N, K, device=dev) / math.sqrt(K)).to(tdt)
            bias, gamma = torch.randn(N, device=dev), torch.randn(N, device=dev)
            out = torch.zeros(M * N, dtype=torch.float32 if epi == _lib.EPI_RESID else tdt, device=dev)
            fn = lambda: _lib.check(L.mhmr_gemm16(A.data_ptr(), K, W.data_ptr(), K, M, N, K, bias.data_ptr(), gamma.data_ptr(), out.data_ptr(), N, None, 0, Tp, H, M, epi, dt, st), "g")
            res.append((K, timeit(fn, 5)))
            del A, W, out
        tiles = (M // 256) * (N // 256); rounds = math.ceil(tiles / 256)
        b = (res[-1][1] - res[1][1]) / (res[-1][0] - res[1][0])          # ms per K
        a = res[1][1] - b * res[1][0]
        print(f"{name:6s} N={N}: " + " ".join(f"K={k}:{t:.3f}ms" for k, t in res) + f" | per K-tile(64) per round {b*64/rounds*1e3:.2f} us, epilogue/round {a/rounds*1e3:.1f} us, rounds={rounds}")
