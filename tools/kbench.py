#!/usr/bin/env python
"""Kernel micro-benchmarks at the ViT-L 896x896 batch-32 shapes (random data): the four ViT GEMMs, attention,
LayerNorm.  Prints one line per kernel: ms, TFLOP/s (algorithmic, T not Tp) or TB/s.
usage: python tools/kbench.py [--dtype bf16|f16] [--only attn|gemm|ln] [--iters 10] [--batch 32]"""
import argparse
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multi_hmr_amd import _lib  # noqa: E402


def timeit(fn, iters):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(iters):
        fn()
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--only", default="")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--img", type=int, default=896)
    ap.add_argument("--tp", type=int, default=0, help="override the padded token count per image (GEMM rows = batch * tp)")
    ap.add_argument("--tokens", type=int, default=0, help="override the token count T per image (attention only; e.g. 4096: no lone class-token query block, no lone last key)")
    ap.add_argument("--variants", default="0", help="attention kernel forms to time (comma separated, csrc/attention.hip)")
    ap.add_argument("--thr", type=float, default=15.0, help="attention reference-level limit (log2)")
    ap.add_argument("--rows", default="all,map", help="GEMM row modes to time: all (B * Tp rows) and / or map (the B * N patch rows)")
    a = ap.parse_args()
    L = _lib.lib()
    dt, tdt = (_lib.DT_F16, torch.float16) if a.dtype == "f16" else (_lib.DT_BF16, torch.bfloat16)
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    B, C, H = a.batch, 1024, 16
    G = a.img // 14
    T = G * G + 1
    Tp = a.tp if a.tp else (T + 63) // 64 * 64
    if a.tokens:
        assert a.only == "attn" and a.tokens <= Tp
        T = a.tokens
    M = B * Tp
    rnd = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(tdt)
    if a.only in ("", "gemm"):
        N_img = T - 1
        shapes = [("qk   (EPI_OP16_QK)", 2 * C, C, _lib.EPI_OP16_QK, 0), ("v    (EPI_VT)", C, C, _lib.EPI_VT, 0), ("v+lo (EPI_VT)", C, C, _lib.EPI_VT, 1),
                  ("proj (EPI_RESID)", C, C, _lib.EPI_RESID, 0), ("proj+lo (EPI_RESID)", C, C, _lib.EPI_RESID, 1),
                  ("fc1  (EPI_GELU)", 4 * C, C, _lib.EPI_OP16_GELU, 0), ("fc1 shape, no GELU (EPI_OP16)", 4 * C, C, _lib.EPI_OP16, 0),
                  ("fc1 shape, ReLU (EPI_OP16_RELU)", 4 * C, C, _lib.EPI_OP16_RELU, 0), ("qk shape, plain (EPI_OP16)", 2 * C, C, _lib.EPI_OP16, 0),
                  ("fc2  (EPI_RESID)", C, 4 * C, _lib.EPI_RESID, 0)]
        # rows: "all" = one GEMM over all B * Tp rows (class + padding rows included), "map" = the B * N patch rows only (token-row map)
        modes = [m for m in a.rows.split(",") if m]
        for name, N, K, epi, lo in shapes:
            for mode in modes:
                if mode == "map" and N_img % 256:
                    continue
                Mg, ir, istr = (B * N_img, N_img, Tp) if mode == "map" else (M, 0, 0)
                Kw = 2 * K if lo else K
                A, W = rnd(M, K), (torch.randn(N, Kw, device=dev) / math.sqrt(K) * (1.0 if not lo else 1.0)).to(tdt)
                if lo:
                    W[:, K:] = (W[:, K:].float() * 2.0 ** -11).to(tdt)
                bias, gamma = torch.randn(N, device=dev), torch.randn(N, device=dev)
                out = torch.zeros(M * N, dtype=torch.float32 if epi == _lib.EPI_RESID else tdt, device=dev)
                fn = lambda: _lib.check(L.mhmr_gemm16_ex(A.data_ptr(), K, W.data_ptr(), Kw, Mg, N, Kw, bias.data_ptr(), gamma.data_ptr(), out.data_ptr(),
                                                         N, None, 0, Tp, H, Mg, epi, dt, ir, istr, K if lo else 0, st), "gemm")
                ms = timeit(fn, a.iters)
                print(f"gemm {name:20s} rows={mode:3s} M={Mg} N={N} K={Kw}: {ms:8.4f} ms  {2.0 * B * T * N * K / ms / 1e9:8.1f} TFLOP/s (alg, single pass)", flush=True)
                del A, W, out
    if a.only in ("", "attn"):
        qk, vt, out = rnd(M, 2 * C), rnd(B * H * 64, Tp), torch.zeros(M, C, dtype=tdt, device=dev)
        qk[:, :C] = (qk[:, :C].float() * _lib.ATTN_QSCALE).to(tdt)          # the Q half arrives pre-scaled (MHMR_EPI_OP16_QK)
        flags = torch.zeros(L.mhmr_attention_flag_count(B, Tp, H), dtype=torch.int32, device=dev)
        for rnd_ in range(2):                                                # two interleaved rounds: within-process A/B
            for var in [int(v) for v in a.variants.split(",")]:
                fn = lambda: _lib.check(L.mhmr_attention16_ex(qk.data_ptr(), vt.data_ptr(), out.data_ptr(), B, T, Tp, C, H, dt, a.thr,
                                                              var, flags.data_ptr() if var in (0, 4, 5, 6, 7, 8, 9, 10) else None, st), "attn")
                ms = timeit(fn, a.iters)
                print(f"attention variant {var} B={B} H={H} T={T} {a.dtype}: {ms:8.4f} ms  {4.0 * B * H * T * T * 64 / ms / 1e9:8.1f} TFLOP/s (alg)", flush=True)
    if a.only in ("", "ln"):
        x, w, b = torch.randn(M, C, device=dev), torch.randn(C, device=dev), torch.randn(C, device=dev)
        o = torch.zeros(M, C, dtype=tdt, device=dev)
        fn = lambda: _lib.check(L.mhmr_layernorm16(x.data_ptr(), w.data_ptr(), b.data_ptr(), o.data_ptr(), M, C, 1e-6, dt, st), "ln")
        ms = timeit(fn, a.iters)
        print(f"layernorm rows={M}: {ms:8.4f} ms  {M * C * 6 / ms / 1e9:8.2f} TB/s")


if __name__ == "__main__":
    main()
