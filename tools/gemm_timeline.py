#!/usr/bin/env python
"""Where a tile's time goes in the persistent 256x256 GEMM: wall-clock (100 MHz) stamps of wave 0 of every workgroup and tile from a debug
build of gemm256.hip (-DMHMR_GEMM_STAMPS, tools/build_gemm_stamps.sh -> tools/dbg/libmhmr_gemm_stamps.so; the product carries no stamps).
  stamp 0 tile start | 1 k loop done | 2 wave groups re-aligned | 3 epilogue issued (stores in flight) | 4 store / DMA queue drained
usage: bash tools/build_gemm_stamps.sh; python tools/gemm_timeline.py [f16|bf16]"""
import ctypes as C, math, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from multi_hmr_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "tools", "dbg", "libmhmr_gemm_stamps.so")
L = _lib.lib()
L.mhmr_debug_gemm_stamps.argtypes = [C.c_void_p]
dtype = sys.argv[1] if len(sys.argv) > 1 else "f16"
dt, tdt = (_lib.DT_F16, torch.float16) if dtype == "f16" else (_lib.DT_BF16, torch.bfloat16)
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
B, Cd, H, N_img, Tp = 32, 1024, 16, 4096, 4160
M, Mg = B * Tp, B * N_img
stamps = torch.zeros(256, 64, 8, dtype=torch.int64, device=dev)
shapes = [("qk", 2 * Cd, Cd, _lib.EPI_OP16_QK, 0), ("v", Cd, Cd, _lib.EPI_VT, 0), ("v+lo", Cd, Cd, _lib.EPI_VT, 1), ("proj", Cd, Cd, _lib.EPI_RESID, 0),
          ("fc1", 4 * Cd, Cd, _lib.EPI_OP16_GELU, 0), ("fc2", Cd, 4 * Cd, _lib.EPI_RESID, 0), ("plain16 K=1024", Cd, Cd, _lib.EPI_OP16, 0)]
L.mhmr_debug_gemm_sametile.argtypes = [C.c_int]
SAME = int(os.environ.get("SAMETILE", "0"))      # 1: every tile = tile 0's operands and outputs (no memory stalls), 2: every tile of a workgroup = its first
assert L.mhmr_debug_gemm_sametile(SAME) == 0
print(f"sametile = {SAME}")
for name, N, K, epi, lo in shapes:
    Kw = 2 * K if lo else K
    A = (torch.randn(M, K, device=dev) * 0.5).to(tdt)
    W = (torch.randn(N, Kw, device=dev) / math.sqrt(K)).to(tdt)
    bias, gamma = torch.randn(N, device=dev), torch.randn(N, device=dev)
    out = torch.zeros(M * N, dtype=torch.float32 if epi == _lib.EPI_RESID else tdt, device=dev)
    fn = lambda: _lib.check(L.mhmr_gemm16_ex(A.data_ptr(), K, W.data_ptr(), Kw, Mg, N, Kw, bias.data_ptr(), gamma.data_ptr(), out.data_ptr(),
                                             N, None, 0, Tp, H, Mg, epi, dt, N_img, Tp, K if lo else 0, st), "gemm")
    assert L.mhmr_debug_gemm_stamps(None) == 0
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    stamps.zero_()
    assert L.mhmr_debug_gemm_stamps(stamps.data_ptr()) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record()
    torch.cuda.synchronize()
    assert L.mhmr_debug_gemm_stamps(None) == 0
    t = stamps.cpu().numpy().astype(np.float64) * 0.01          # us
    ntile = (Mg // 256) * (N // 256) // 256
    t = t[:, :ntile]
    kl, al, ep, dr = t[:, :, 1] - t[:, :, 0], t[:, :, 2] - t[:, :, 1], t[:, :, 3] - t[:, :, 2], t[:, :, 4] - t[:, :, 3]
    tile = t[:, 1:, 0] - t[:, :-1, 0]
    md = lambda x: f"{np.median(x):6.2f} (p10 {np.percentile(x, 10):6.2f}, p90 {np.percentile(x, 90):6.2f})"
    print(f"== {name}: N={N} K={Kw} {dtype}, {ntile} tiles per workgroup, launch {e0.elapsed_time(e1) * 1e3:.0f} us; microseconds per tile, median over workgroups x tiles")
    print(f"   k loop ({Kw // 64} k tiles) {md(kl)}  = {np.median(kl) / (Kw // 64):.3f} per k tile")
    print(f"   group re-align          {md(al)}")
    print(f"   epilogue issue          {md(ep)}")
    print(f"   drain (vmcnt 0)         {md(dr)}")
    print(f"   tile period             {md(tile)}   first tile start spread {t[:, 0, 0].max() - t[:, 0, 0].min():.2f}, last drain spread {t[:, -1, 4].max() - t[:, -1, 4].min():.2f}")
    # do the workgroups hit their epilogues together?  spread of the epilogue start of tile r over workgroups
    sp = [t[:, r, 2].max() - t[:, r, 2].min() for r in range(ntile)]
    print(f"   spread over workgroups of the epilogue start, per tile round: {' '.join(f'{x:.1f}' for x in sp[:16])}")
    tot = t[:, -1, 4] - t[:, 0, 0].min()                       # when each workgroup finished, since the first start
    xcd = np.arange(256) & 7
    print(f"   workgroup finish times: mean {tot.mean():.1f}, max {tot.max():.1f} (a perfect tile queue would end near mean + half a tile = {tot.mean() + 0.5 * np.median(tile):.1f});"
          f" per XCD mean: {' '.join(f'{tot[xcd == x].mean():.0f}' for x in range(8))}")
    del A, W, out
