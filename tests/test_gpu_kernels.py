"""-m gpu: every HIP kernel family, called through the C ABI, against a plain PyTorch fp32/fp64 reference of the
same op on the same seeded inputs.  Tolerances: 16-bit-operand kernels are compared on identical (already
rounded) operands, so only fp32-accumulation-order and output rounding differ; fp32 kernels to ~1e-5."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from multi_hmr_amd import _lib, packing  # noqa: E402
import synthetic  # noqa: E402

DTYPES = [("f16", _lib.DT_F16, torch.float16, 2e-3), ("bf16", _lib.DT_BF16, torch.bfloat16, 1.6e-2)]


@pytest.fixture(scope="module")
def L():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return _lib.lib()


def dev():
    return torch.device("cuda:0")


def stream():
    return torch.cuda.current_stream().cuda_stream


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def maxrel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def test_gemm256_rows_beyond_2_31_elements(L):
    """M x K = 2^31 + 2^18 operand elements: the 256x256 kernel addresses tiles through 64-bit bases (it used to hand such shapes to
    the slower 128x128 kernel silently).  The last row tile, 4 GiB into A, against torch on the same rows."""
    M, N, K = (1 << 21) + 256, 256, 1024
    tdt, dt = torch.float16, _lib.DT_F16
    A = torch.empty(M, K, dtype=tdt, device=dev())
    g = torch.Generator(device=dev()).manual_seed(3)
    A[:256].normal_(generator=g)
    A[-256:].normal_(generator=g)
    A[256:-256].zero_()
    W = (torch.randn(N, K, generator=g, device=dev()) / math.sqrt(K)).to(tdt)
    bias = torch.randn(N, generator=g, device=dev())
    out = torch.full((M, N), 7.0, dtype=tdt, device=dev())
    _lib.check(L.mhmr_gemm16(A.data_ptr(), K, W.data_ptr(), K, M, N, K, bias.data_ptr(), None, out.data_ptr(), N, None, 0, 128, 1, M,
                             _lib.EPI_OP16, dt, stream()), "gemm")
    for rows in (slice(0, 256), slice(M - 256, M)):
        ref = A[rows].float() @ W.float().T + bias
        assert maxrel(out[rows].float(), ref) < 2e-3
    assert float((out[256:512].float() - bias).abs().max()) < 2e-3          # a zero row tile: bias only
    del A, out
    torch.cuda.empty_cache()


def test_gelu_epilogue_max_abs_error(L):
    """The fc1 epilogue's GELU is max(x, 0) - |x| exp2(P(|x|)) with a fitted degree-5 P (csrc/mhmr_common.h gelu_fast), not erff:
    swept over x = a_m + b_n in [-10, 10] (step ~ 4e-5) through the GEMM itself (A[:, 0] = a, W[:, 0] = 1, bias = b), the stored
    f16 value stays within 1.5e-6 absolute + half an f16 ulp of the exact x Phi(x) (rounds 4-5, Abramowitz-Stegun 7.1.25: 2.6e-5)."""
    M, N, K = 4096, 128, 64
    a = torch.linspace(-10, 10, M).half()
    b = torch.linspace(0, 20.0 / M, N)
    A = torch.zeros(M, K, dtype=torch.float16)
    A[:, 0] = a
    W = torch.zeros(N, K, dtype=torch.float16)
    W[:, 0] = 1
    out = torch.zeros(M, N, dtype=torch.float16, device=dev())
    Ad, Wd, bd = A.to(dev()), W.to(dev()), b.to(dev())
    _lib.check(L.mhmr_gemm16(Ad.data_ptr(), K, Wd.data_ptr(), K, M, N, K, bd.data_ptr(), None, out.data_ptr(), N, None, 0, 128, 1, M,
                             _lib.EPI_OP16_GELU, _lib.DT_F16, stream()), "gemm")
    x = (a.float()[:, None] + b[None, :]).double()
    ref = 0.5 * x * (1 + torch.erf(x / math.sqrt(2)))
    err = (out.cpu().double() - ref).abs()
    bound = 1.5e-6 + 2.0 ** -11 * ref.abs() + 2.0 ** -25         # (+ the f16 subnormal step)
    assert bool((err <= bound).all()), float((err - bound).max())
    assert float(err.max()) > 1e-6                                # the sweep really reached values whose f16 step shows


def swap23(t):
    return (t & ~12) | ((t & 4) << 1) | ((t & 8) >> 1)


# ------------------------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("name,dt,tdt,tol", DTYPES)
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 192), (384, 128, 640),
                                   (256, 256, 128), (512, 256, 1024), (256, 768, 384)])   # the last three run the 256x256 8-phase kernel
def test_gemm_epilogues(L, name, dt, tdt, tol, M, N, K):
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g).to(dev()).to(tdt)
    W = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dev()).to(tdt)
    bias = torch.randn(N, generator=g).to(dev())
    gamma = torch.randn(N, generator=g).to(dev())
    ref = A.float() @ W.float().T + bias
    call = lambda out, ldo, epi, **kw: _lib.check(L.mhmr_gemm16(A.data_ptr(), K, W.data_ptr(), K, M, N, K, bias.data_ptr(),
                                                                kw.get("gamma"), out.data_ptr(), ldo, kw.get("pos"), kw.get("Np", 0),
                                                                kw.get("Tp", 128), kw.get("H", 1), kw.get("Mvalid", M), epi, dt, stream()), "gemm")
    out = torch.zeros(M, N, dtype=tdt, device=dev())
    call(out, N, _lib.EPI_OP16)
    assert maxrel(out.float(), ref) < tol, ("op16", maxrel(out.float(), ref))
    call(out, N, _lib.EPI_OP16_GELU)
    assert maxrel(out.float(), torch.nn.functional.gelu(ref)) < tol
    call(out, N, _lib.EPI_OP16_RELU)
    assert maxrel(out.float(), torch.relu(ref)) < tol
    call(out, N, _lib.EPI_OP16_QK)             # Q | K projection: the first N/2 columns carry the softmax scale in the exp2 domain
    colscale = torch.ones(N, device=dev())
    colscale[: N // 2] = _lib.ATTN_QSCALE
    assert maxrel(out.float(), ref * colscale) < tol
    o32 = torch.zeros(M, N, device=dev())
    call(o32, N, _lib.EPI_F32)
    assert maxrel(o32, ref) < 2e-5, ("f32", maxrel(o32, ref))
    res = torch.randn(M, N, generator=g).to(dev())
    o32 = res.clone()
    call(o32, N, _lib.EPI_RESID, gamma=gamma.data_ptr())
    assert maxrel(o32, res + gamma * ref) < 2e-5


@pytest.mark.parametrize("name,dt,tdt,tol", DTYPES)
def test_gemm_patch_and_vt_layouts(L, name, dt, tdt, tol):
    g = torch.Generator(device="cpu").manual_seed(7)
    # patch epilogue: rows scattered to (b*Tp + n) -- the class token is the LAST token row of an image --, + pos[1+n]; rows >= Mvalid dropped
    B, Np, Tp, N, K = 3, 100, 128, 128, 128
    M, Mvalid = 384, B * Np
    A = torch.randn(M, K, generator=g).to(dev()).to(tdt)
    W = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dev()).to(tdt)
    bias = torch.randn(N, generator=g).to(dev())
    pos = torch.randn(1 + Np, N, generator=g).to(dev())
    out = torch.full((B * Tp, N), 7.0, device=dev())
    _lib.check(L.mhmr_gemm16(A.data_ptr(), K, W.data_ptr(), K, M, N, K, bias.data_ptr(), None, out.data_ptr(), N, pos.data_ptr(), Np, Tp,
                             2, Mvalid, _lib.EPI_PATCH, dt, stream()), "gemm patch")
    ref = (A.float() @ W.float().T + bias)[:Mvalid].view(B, Np, N) + pos[1:]
    got = out.view(B, Tp, N)
    assert maxrel(got[:, :Np], ref) < 2e-5
    assert torch.all(got[:, Np:] == 7.0)                                            # untouched rows (class slot + padding)
    # V^T epilogue
    B, H, Tp = 2, 2, 256
    M, N, K = B * Tp, H * 64, 64
    A = torch.randn(M, K, generator=g).to(dev()).to(tdt)
    W = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dev()).to(tdt)
    bias = torch.randn(N, generator=g).to(dev())
    vt = torch.zeros(B, H, 64, Tp, dtype=tdt, device=dev())
    _lib.check(L.mhmr_gemm16(A.data_ptr(), K, W.data_ptr(), K, M, N, K, bias.data_ptr(), None, vt.data_ptr(), 0, None, 0, Tp, H, M,
                             _lib.EPI_VT, dt, stream()), "gemm vt")
    ref = (A.float() @ W.float().T + bias).view(B, Tp, H, 64).permute(0, 2, 3, 1)        # [B,H,64,t]
    perm = swap23(torch.arange(Tp, device=dev()))
    assert maxrel(vt.float()[..., perm], ref) < tol
    # same two layouts through the 256x256 kernel (blocks straddle image boundaries: Tp = 384, 3 x 256 rows)
    B, Np, Tp, N, K = 3, 170, 256, 256, 256
    M, Mvalid = 512, B * Np
    A = torch.randn(M, K, generator=g).to(dev()).to(tdt)
    W = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dev()).to(tdt)
    bias = torch.randn(N, generator=g).to(dev())
    pos = torch.randn(1 + Np, N, generator=g).to(dev())
    out = torch.full((B * Tp, N), 7.0, device=dev())
    _lib.check(L.mhmr_gemm16(A.data_ptr(), K, W.data_ptr(), K, M, N, K, bias.data_ptr(), None, out.data_ptr(), N, pos.data_ptr(), Np, Tp,
                             2, Mvalid, _lib.EPI_PATCH, dt, stream()), "gemm patch 256")
    ref = (A.float() @ W.float().T + bias)[:Mvalid].view(B, Np, N) + pos[1:]
    got = out.view(B, Tp, N)
    assert maxrel(got[:, :Np], ref) < 2e-5
    assert torch.all(got[:, Np:] == 7.0)
    B, H, Tp = 2, 4, 384
    M, N, K = B * Tp, H * 64, 128
    A = torch.randn(M, K, generator=g).to(dev()).to(tdt)
    W = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dev()).to(tdt)
    bias = torch.randn(N, generator=g).to(dev())
    vt = torch.zeros(B, H, 64, Tp, dtype=tdt, device=dev())
    _lib.check(L.mhmr_gemm16(A.data_ptr(), K, W.data_ptr(), K, M, N, K, bias.data_ptr(), None, vt.data_ptr(), 0, None, 0, Tp, H, M,
                             _lib.EPI_VT, dt, stream()), "gemm vt 256")
    ref = (A.float() @ W.float().T + bias).view(B, Tp, H, 64).permute(0, 2, 3, 1)
    perm = swap23(torch.arange(Tp, device=dev()))
    assert maxrel(vt.float()[..., perm], ref) < tol


@pytest.mark.parametrize("name,dt,tdt,tol", DTYPES)
@pytest.mark.parametrize("N,K,Nimg", [(256, 256, 256), (512, 128, 256), (384, 192, 256), (256, 128, 768)])   # 256x256 kernel (one / three tiles per image), 128x128 kernel
def test_gemm_token_row_map(L, name, dt, tdt, tol, N, K, Nimg):
    """mhmr_gemm16_ex with img_rows / img_stride: the GEMM covers rows b * Tp + [0, Nimg) of every image and must neither read nor
    write the class / padding rows behind them (ViT block linears over the patch rows only)."""
    B, Tp, H = 3, Nimg + 64, N // 64
    g = torch.Generator(device="cpu").manual_seed(N + K)
    A = torch.randn(B * Tp, K, generator=g).to(dev()).to(tdt)
    A.view(B, Tp, K)[:, Nimg:] = float("nan")                                    # a read of a skipped row poisons the result
    W = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dev()).to(tdt)
    bias, gamma = torch.randn(N, generator=g).to(dev()), torch.randn(N, generator=g).to(dev())
    ref = A.view(B, Tp, K)[:, :Nimg].float() @ W.float().T + bias                # [B, Nimg, N]
    call = lambda out, ldo, epi, gm=None: _lib.check(L.mhmr_gemm16_ex(A.data_ptr(), K, W.data_ptr(), K, B * Nimg, N, K, bias.data_ptr(), gm,
                                                                      out.data_ptr(), ldo, None, 0, Tp, H, B * Nimg, epi, dt, Nimg, Tp, 0,
                                                                      stream()), "gemm_ex")
    out = torch.full((B, Tp, N), 7.0, dtype=tdt, device=dev())
    call(out, N, _lib.EPI_OP16)
    assert maxrel(out[:, :Nimg].float(), ref) < tol and torch.all(out[:, Nimg:] == 7.0)
    call(out, N, _lib.EPI_OP16_GELU)
    assert maxrel(out[:, :Nimg].float(), torch.nn.functional.gelu(ref)) < tol and torch.all(out[:, Nimg:] == 7.0)
    res = torch.randn(B, Tp, N, generator=g).to(dev())
    o32 = res.clone()
    call(o32, N, _lib.EPI_RESID, gamma.data_ptr())
    assert maxrel(o32[:, :Nimg], res[:, :Nimg] + gamma * ref) < 2e-5 and torch.equal(o32[:, Nimg:], res[:, Nimg:])
    vt = torch.full((B, H, 64, Tp), 7.0, dtype=tdt, device=dev())
    call(vt, 0, _lib.EPI_VT)
    perm = swap23(torch.arange(Nimg, device=dev()))
    assert maxrel(vt.float()[..., perm], ref.view(B, Nimg, H, 64).permute(0, 2, 3, 1)) < tol
    assert torch.all(vt[..., Nimg:] == 7.0)


@pytest.mark.parametrize("name,dt,tdt,tol", DTYPES)
@pytest.mark.parametrize("M,N,K", [(256, 256, 256), (128, 128, 64), (512, 256, 1024)])
def test_gemm_low_half_weight_pass(L, name, dt, tdt, tol, M, N, K):
    """W = [W_hi | W_lo] along k with the activation's k index wrapping (a_k): one accumulator chain computes A . (W_hi + W_lo)^T.  The
    low halves of f16 weights are mostly SUBNORMAL f16 numbers -- the matrix pipe must take them at full precision -- so the result
    is compared with the fp32 weights: the residual weight error drops from 2^-11 to ~2^-19 relative."""
    from multi_hmr_amd import vit
    g = torch.Generator(device="cpu").manual_seed(M + K)
    A = torch.randn(M, K, generator=g).to(dev()).to(tdt)
    W32 = (torch.randn(N, K, generator=g) * 0.02).to(dev())
    W2 = vit.hi_lo(W32, tdt)                                                     # [N, 2K]
    if name == "f16":
        lo = W2[:, K:].float().abs()
        assert float((lo[lo > 0] < 2.0 ** -14).float().mean()) > 0.9             # the low halves really are subnormal
    bias = torch.randn(N, generator=g).to(dev())
    exact = A.double() @ W32.double().T + bias.double()
    o32 = torch.zeros(M, N, device=dev())
    _lib.check(L.mhmr_gemm16_ex(A.data_ptr(), K, W2.data_ptr(), 2 * K, M, N, 2 * K, bias.data_ptr(), None, o32.data_ptr(), N, None, 0, 128,
                                1, M, _lib.EPI_F32, dt, 0, 0, K, stream()), "gemm lo")
    one = torch.zeros(M, N, device=dev())
    _lib.check(L.mhmr_gemm16(A.data_ptr(), K, W2[:, :K].contiguous().data_ptr(), K, M, N, K, bias.data_ptr(), None, one.data_ptr(), N, None, 0,
                             128, 1, M, _lib.EPI_F32, dt, stream()), "gemm hi")
    prod = exact - bias.double()                                                 # (relative to the product alone: the bias is exact in both)
    e2, e1 = rel(o32.double() - bias.double(), prod), rel(one.double() - bias.double(), prod)
    assert e1 > (1e-4 if name == "f16" else 8e-4), e1                            # a single pass carries the weight rounding ...
    assert e2 < e1 / 30, (e1, e2)                                                # ... the low-half pass removes it
    # V^T form (the activation is the FIRST operand there)
    if M % 128 == 0 and N % 64 == 0:
        B, H, Tp = 1, N // 64, M
        vt = torch.zeros(B, H, 64, Tp, dtype=tdt, device=dev())
        _lib.check(L.mhmr_gemm16_ex(A.data_ptr(), K, W2.data_ptr(), 2 * K, M, N, 2 * K, bias.data_ptr(), None, vt.data_ptr(), 0, None, 0, Tp,
                                    H, M, _lib.EPI_VT, dt, 0, 0, K, stream()), "gemm lo vt")
        perm = swap23(torch.arange(Tp, device=dev()))
        assert maxrel(vt.float()[..., perm], exact.float().view(B, Tp, H, 64).permute(0, 2, 3, 1)) < tol


@pytest.mark.parametrize("name,dt,tdt,tol", DTYPES)
@pytest.mark.parametrize("B", [3, 32, 40])
def test_cls_linear(L, name, dt, tdt, tol, B):
    """The class-token rows of the block linears (csrc/vit_cls.hip): B rows Tp apart, the three epilogues, with and without the
    low-half weight pass, against torch on the same (rounded) operands."""
    from multi_hmr_amd import vit
    C, H, Tp, Ncls = 256, 4, 192, 130
    g = torch.Generator(device="cpu").manual_seed(B)
    xn = torch.randn(B * Tp, C, generator=g).to(dev()).to(tdt)
    rows = xn.view(B, Tp, C)[:, Ncls].float()                                   # [B, C]
    Wqkv = (torch.randn(3 * C, C, generator=g) / math.sqrt(C)).to(dev()).to(tdt)
    bqkv = torch.randn(3 * C, generator=g).to(dev())
    qk = torch.full((B * Tp, 2 * C), 7.0, dtype=tdt, device=dev())
    vt = torch.full((B, H, 64, Tp), 7.0, dtype=tdt, device=dev())
    vcol = int(swap23(torch.tensor(Ncls)))
    esz = 2
    _lib.check(L.mhmr_cls_linear16(xn.data_ptr() + Ncls * C * esz, Tp * C, Wqkv.data_ptr(), C, B, 3 * C, C, 0, bqkv.data_ptr(), None,
                                   qk.data_ptr() + Ncls * 2 * C * esz, Tp * 2 * C, 0, C, vt.data_ptr(), H, Tp, vcol, 0, dt, stream()), "cls qkv")
    ref = rows @ Wqkv.float().T + bqkv
    got = qk.view(B, Tp, 2 * C)
    assert maxrel(got[:, Ncls, :C].float(), ref[:, :C] * _lib.ATTN_QSCALE) < tol
    assert maxrel(got[:, Ncls, C:].float(), ref[:, C:2 * C]) < tol
    assert maxrel(vt[..., vcol].float(), ref[:, 2 * C:].view(B, H, 64)) < tol
    assert torch.all(got[:, :Ncls] == 7.0) and torch.all(got[:, Ncls + 1:] == 7.0)
    assert torch.all(vt[..., :vcol] == 7.0) and torch.all(vt[..., vcol + 1:] == 7.0)
    # residual epilogue with the low-half pass (K = 2 a_k)
    W32 = (torch.randn(C, C, generator=g) * 0.05).to(dev())
    W2 = vit.hi_lo(W32, tdt)
    bias, gamma = torch.randn(C, generator=g).to(dev()), torch.randn(C, generator=g).to(dev())
    res = torch.randn(B * Tp, C, generator=g).to(dev())
    o32 = res.clone()
    _lib.check(L.mhmr_cls_linear16(xn.data_ptr() + Ncls * C * esz, Tp * C, W2.data_ptr(), 2 * C, B, C, 2 * C, C, bias.data_ptr(), gamma.data_ptr(),
                                   o32.data_ptr() + Ncls * C * 4, Tp * C, 0, C, None, H, Tp, 0, 1, dt, stream()), "cls resid")
    exact = res.view(B, Tp, C)[:, Ncls] + gamma * (rows @ W32.T + bias)
    assert maxrel(o32.view(B, Tp, C)[:, Ncls], exact) < 3e-5 if name == "f16" else 3e-4
    keep = torch.ones(B, Tp, dtype=torch.bool, device=dev())
    keep[:, Ncls] = False
    assert torch.equal(o32.view(B, Tp, C)[keep], res.view(B, Tp, C)[keep])
    # GELU epilogue
    W1 = (torch.randn(2 * C, C, generator=g) / math.sqrt(C)).to(dev()).to(tdt)
    b1 = torch.randn(2 * C, generator=g).to(dev())
    hid = torch.full((B * Tp, 2 * C), 7.0, dtype=tdt, device=dev())
    _lib.check(L.mhmr_cls_linear16(xn.data_ptr() + Ncls * C * esz, Tp * C, W1.data_ptr(), C, B, 2 * C, C, 0, b1.data_ptr(), None,
                                   hid.data_ptr() + Ncls * 2 * C * esz, Tp * 2 * C, 0, C, None, H, Tp, 0, 2, dt, stream()), "cls gelu")
    assert maxrel(hid.view(B, Tp, 2 * C)[:, Ncls].float(), torch.nn.functional.gelu(rows @ W1.float().T + b1)) < tol


@pytest.mark.parametrize("name,dt,tdt,tol", DTYPES)
def test_layernorm_fold(L, name, dt, tdt, tol):
    """LayerNorm folded into the neighbouring GEMMs (csrc/gemm256.hip): the residual epilogue leaves the 16-bit copy of the RAW residual
    rows + per-block (sum, sum of squares); mhmr_ln_stats finishes (mean, rstd) per row (class rows from the fp32 row itself); the
    consuming linears (GELU / Q|K / V^T epilogues) compute rstd (x16 . W'^T - mean colsum) + b' -- against torch's layer_norm + linear."""
    B, Nimg, Tp, C = 3, 256, 320, 256
    H, M = C // 64, B * Nimg
    g = torch.Generator(device="cpu").manual_seed(17)
    patch = torch.zeros(B, Tp, dtype=torch.bool)
    patch[:, :Nimg] = True
    patch = patch.reshape(-1).to(dev())
    resid0 = (torch.randn(B * Tp, C, generator=g) * 2.0 + 0.3).to(dev())
    att = torch.randn(B * Tp, C, generator=g).to(dev()).to(tdt)
    Wp = (torch.randn(C, C, generator=g) / math.sqrt(C)).to(dev()).to(tdt)
    bp, gamma = torch.randn(C, generator=g).to(dev()), torch.randn(C, generator=g).to(dev())
    # ---- producer ----
    out = resid0.clone()
    x16 = torch.full((B * Tp, C), 7.0, dtype=tdt, device=dev())
    pstats = torch.full((B * Tp, C // 64, 2), -1.0, device=dev())
    _lib.check(L.mhmr_gemm16_ln(att.data_ptr(), C, Wp.data_ptr(), C, M, C, C, bp.data_ptr(), gamma.data_ptr(), out.data_ptr(), C, Tp, H,
                                _lib.EPI_RESID, dt, Nimg, Tp, 0, x16.data_ptr(), pstats.data_ptr(), None, None, None, stream()), "resid + fold")
    ref = resid0 + gamma * (att.float() @ Wp.float().T + bp)
    assert maxrel(out[patch], ref[patch]) < 2e-5 and torch.equal(out[~patch], resid0[~patch])
    assert torch.equal(x16[patch].float(), out[patch].to(tdt).float()) and torch.all(x16[~patch] == 7.0)
    blocks = out.view(B * Tp, C // 64, 64)
    assert maxrel(pstats[patch][..., 0], blocks.sum(-1)[patch]) < 1e-5 and maxrel(pstats[patch][..., 1], (blocks * blocks).sum(-1)[patch]) < 1e-5
    assert torch.all(pstats[~patch] == -1.0)
    # ---- row statistics (patch rows from the block sums, class rows = row Nimg of every image from the fp32 row) ----
    rowstats = torch.full((B * Tp, 2), -1.0, device=dev())
    _lib.check(L.mhmr_ln_stats(pstats.data_ptr(), out.data_ptr(), rowstats.data_ptr(), B, Nimg, Tp, C, 1e-6, stream()), "ln_stats")
    mean = out.double().mean(-1)
    rstd = 1.0 / torch.sqrt(out.double().var(-1, unbiased=False) + 1e-6)
    has = patch.clone()
    has.view(B, Tp)[:, Nimg] = True
    assert maxrel(rowstats[has][:, 0], mean[has]) < 1e-5 and maxrel(rowstats[has][:, 1], rstd[has]) < 1e-5
    assert torch.all(rowstats[~has] == -1.0)
    # ---- consumers ----
    ln_w, ln_b = (1.0 + 0.2 * torch.randn(C, generator=g)).to(dev()), (0.1 * torch.randn(C, generator=g)).to(dev())
    xn_true = torch.nn.functional.layer_norm(out, (C,), ln_w, ln_b, 1e-6)
    for epi, N in ((_lib.EPI_OP16_GELU, 512), (_lib.EPI_OP16_QK, 512), (_lib.EPI_VT, 256)):
        W = (torch.randn(N, C, generator=g) / math.sqrt(C)).to(dev())
        b = torch.randn(N, generator=g).to(dev())
        Wf = (W * ln_w).to(tdt)
        colsum = Wf.double().sum(1).float()
        fb = (b.double() + W.double() @ ln_b.double()).float()
        lin = rowstats[:, 1:2].double() * (x16.double() @ Wf.double().T - rowstats[:, 0:1].double() * colsum.double()) + fb.double()   # the kernel's formula
        true = xn_true.double() @ W.double().T + b.double()                                                                        # what it stands for
        if epi == _lib.EPI_VT:
            vt = torch.full((B, N // 64, 64, Tp), 7.0, dtype=tdt, device=dev())
            _lib.check(L.mhmr_gemm16_ln(x16.data_ptr(), C, Wf.data_ptr(), C, M, N, C, None, None, vt.data_ptr(), 0, Tp, N // 64, epi, dt, Nimg, Tp,
                                        0, None, None, rowstats.data_ptr(), colsum.data_ptr(), fb.data_ptr(), stream()), "vt + fold")
            perm = swap23(torch.arange(Nimg, device=dev()))
            got = vt.float()[..., perm].permute(0, 3, 1, 2).reshape(B, Nimg, N)
            want, wtrue = lin.view(B, Tp, N)[:, :Nimg], true.view(B, Tp, N)[:, :Nimg]
            assert torch.all(vt[..., Nimg:] == 7.0)
        else:
            o16 = torch.full((B * Tp, N), 7.0, dtype=tdt, device=dev())
            _lib.check(L.mhmr_gemm16_ln(x16.data_ptr(), C, Wf.data_ptr(), C, M, N, C, None, None, o16.data_ptr(), N, Tp, H, epi, dt, Nimg, Tp, 0,
                                        None, None, rowstats.data_ptr(), colsum.data_ptr(), fb.data_ptr(), stream()), "consumer + fold")
            assert torch.all(o16[~patch] == 7.0)
            got = o16[patch].float()
            want, wtrue = lin[patch], true[patch]
            if epi == _lib.EPI_OP16_GELU:
                want, wtrue = torch.nn.functional.gelu(want), torch.nn.functional.gelu(wtrue)
            else:
                sc = torch.ones(N, device=dev(), dtype=torch.float64)
                sc[: N // 2] = _lib.ATTN_QSCALE
                want, wtrue = want * sc, wtrue * sc
        assert maxrel(got, want) < tol, (epi, maxrel(got, want))                 # same 16-bit operands: accumulation order + output rounding
        assert rel(got, wtrue) < (2e-3 if name == "f16" else 1.6e-2), (epi, rel(got, wtrue))      # vs the real LayerNorm + linear


# ------------------------------------------------------------------------------------------------------ attention
def _attn_inputs(B, H, T, tdt, seed, qscale=1.0, pad=128):
    """q is handed over PRE-SCALED (include/mhmr.h: the Q half of qk holds q * MHMR_ATTN_QSCALE, scores are in the exp2 domain)."""
    C, Tp = H * 64, packing.roundup(T, pad)
    g = torch.Generator(device="cpu").manual_seed(seed)
    q = (torch.randn(B, Tp, H, 64, generator=g) * _lib.ATTN_QSCALE * qscale).to(dev()).to(tdt)
    k = torch.randn(B, Tp, H, 64, generator=g).to(dev()).to(tdt)
    v = torch.randn(B, Tp, H, 64, generator=g).to(dev()).to(tdt)
    return q, k, v, C, Tp


def _attn_run(L, q, k, v, B, H, T, C, Tp, dt, tdt, thr=None, variant=0):
    qk = torch.cat([q.reshape(B * Tp, C), k.reshape(B * Tp, C)], dim=1).contiguous()
    vt = torch.zeros(B, H, 64, Tp, dtype=tdt, device=dev())
    vt[..., swap23(torch.arange(Tp, device=dev()))] = v.permute(0, 2, 3, 1)
    out = torch.zeros(B * Tp, C, dtype=tdt, device=dev())
    if thr is None:
        _lib.check(L.mhmr_attention16(qk.data_ptr(), vt.data_ptr(), out.data_ptr(), B, T, Tp, C, H, dt, stream()), "attention")
    else:
        flags = torch.full((L.mhmr_attention_flag_count(B, Tp, H),), 7, dtype=torch.int32, device=dev())   # (the call writes every entry)
        _lib.check(L.mhmr_attention16_ex(qk.data_ptr(), vt.data_ptr(), out.data_ptr(), B, T, Tp, C, H, dt, thr, variant,
                                         flags.data_ptr() if variant in (0, 4, 5, 6, 7, 8, 9, 10) else None, stream()), "attention_ex")
        _attn_run.last_flags = flags
    return out.view(B, Tp, C)


def _attn_ref(q, k, v, T, rows=None):
    """fp64 softmax_2(q k^T) v on the (already rounded) operands; rows = query rows to evaluate (None = all T)."""
    qf, kf, vf = [t.double().permute(0, 2, 1, 3) for t in (q, k, v)]            # [B,H,t,64]
    qs = qf[:, :, :T] if rows is None else qf[:, :, rows]
    att = torch.softmax(qs @ kf[:, :, :T].transpose(-1, -2) * math.log(2.0), dim=-1) @ vf[:, :, :T]
    return att.permute(0, 2, 1, 3).reshape(att.shape[0], att.shape[2], -1)       # [B,rows,C]


@pytest.mark.parametrize("name,dt,tdt,tol", DTYPES)
@pytest.mark.parametrize("variant", [None, 0, 4, 5, 6, 7, 8, 9, 10])     # None = mhmr_attention16; 0 = 32 queries per wave; 4 / 5 = 64 queries per wave; 6 = 16x16x32 MFMAs; 7 / 8 / 9 = its round-6 experiment forms (early copies, three-slot ring, both); 10 = 6 with the class query of T = 128 n + 1 on workgroups of its own
@pytest.mark.parametrize("B,H,T,pad", [(2, 3, 200, 128), (1, 2, 256, 128), (1, 1, 65, 128), (2, 6, 257, 128), (2, 6, 257, 64), (3, 2, 130, 64),
                                       (1, 2, 577, 64), (2, 1, 40, 64)])
def test_attention(L, name, dt, tdt, tol, B, H, T, pad, variant):
    q, k, v, C, Tp = _attn_inputs(B, H, T, tdt, T, pad=pad)          # pad = 64: rows per image not a multiple of the 128-query workgroup
    k[0, min(T - 1, 70), 0] *= 6.0          # a spiked key moves the row maximum late in the loop
    got = _attn_run(L, q, k, v, B, H, T, C, Tp, dt, tdt, thr=None if variant is None else 15.0, variant=variant or 0)[:, :T].double()
    err = float((got - _attn_ref(q, k, v, T)).abs().max())
    assert err < (4e-3 if name == "f16" else 3e-2), err
    assert torch.isfinite(got).all()


@pytest.mark.parametrize("name,dt,tdt,tol", DTYPES)
@pytest.mark.parametrize("variant", [0, 4, 5, 6, 10])
@pytest.mark.parametrize("target", [10.0, 40.0])
def test_attention_last_key_rank_one_update(L, name, dt, tdt, tol, target, variant):
    """T = 64 n + 1: the default kernel form folds the lone key of the last tile in as a rank-1 update.  A last key that dominates
    a query's softmax (score +10 above the level: stays in range) or leaves the 16-bit range (+40: the workgroup is flagged and
    recomputed by the textbook form) must both match fp64."""
    B, H, T = 2, 2, 257
    q, k, v, C, Tp = _attn_inputs(B, H, T, tdt, 11, qscale=3.0)
    kk = k.float()
    for (bb, row, hh) in [(0, 3, 0), (1, 200, 1), (1, 256, 0)]:
        d = q.float()[bb, row, hh]
        kk[bb, T - 1, hh] = d / d.norm() ** 2 * target
    k = kk.to(tdt)
    got = _attn_run(L, q, k, v, B, H, T, C, Tp, dt, tdt, thr=15.0, variant=variant)[:, :T].double()
    flagged = int(_attn_run.last_flags.sum().item())
    assert (flagged > 0) == (target > 15.0), flagged
    err = float((got - _attn_ref(q, k, v, T)).abs().max())
    assert err < (4e-3 if name == "f16" else 3e-2), err


@pytest.mark.parametrize("name,dt,tdt,tol", DTYPES)
@pytest.mark.parametrize("pad", [128, 64, 320])          # 320: a tile of nothing but padding rows behind the class query's
def test_attention_class_query_role(L, name, dt, tdt, tol, pad):
    """Variant 10: the lone query of T = 128 n + 1 (the class token) runs on a wave of its own -- exact online softmax on the vector ALU,
    32 keys per step, a level that moves only when a key beats it by 2^64.  Keys aligned with the class query FORCE that branch (+100 in
    the exp2 domain at key 200, +30 at key 40: below the threshold, and the class token's own key at +90); every row must match fp64, the
    class row as closely as the others, and no flag entry may keep the caller's garbage."""
    B, H, T = 2, 4, 385
    q, k, v, C, Tp = _attn_inputs(B, H, T, tdt, 21, qscale=3.0, pad=pad)
    kk = k.float()
    for (bb, hh, key, target) in [(0, 0, 200, 100.0), (0, 1, 40, 30.0), (1, 3, 384, 90.0), (1, 2, 5, 80.0)]:
        d = q.float()[bb, T - 1, hh]
        kk[bb, key, hh] = d / d.norm() ** 2 * target
    k = kk.to(tdt)
    got = _attn_run(L, q, k, v, B, H, T, C, Tp, dt, tdt, thr=15.0, variant=10)[:, :T].double()
    ref = _attn_ref(q, k, v, T)
    bound = 4e-3 if name == "f16" else 3e-2
    assert float((got - ref).abs().max()) < bound
    assert float((got[:, T - 1] - ref[:, T - 1]).abs().max()) < bound / 2          # (fp32 p: no 16-bit rounding of the probabilities)
    assert torch.isfinite(got).all()
    assert int((_attn_run.last_flags == 7).sum().item()) == 0                      # every entry written (the fill value was 7)
    six = _attn_run(L, q, k, v, B, H, T, C, Tp, dt, tdt, thr=15.0, variant=6)[:, :T - 1]
    ten = _attn_run(L, q, k, v, B, H, T, C, Tp, dt, tdt, thr=15.0, variant=10)[:, :T - 1]
    assert torch.equal(six, ten)                                                   # the 128-query workgroups are the same code


@pytest.mark.parametrize("variant", [0, 4, 6, 10])
@pytest.mark.parametrize("pad", [128, 64])               # rows per image: 2432 / 4224 / 8576 or 2368 / 4160 / 8512 (vit.padded_tokens)
@pytest.mark.parametrize("T", [2305, 4097, 8465])        # 672^2, 896^2, 1288^2: 37 / 65 / 133 key tiles, the last one masked
def test_attention_full_length_against_fp64(L, T, pad, variant):
    """BASELINE sequence lengths, f16 operands, against fp64 on sampled query rows (first / last rows, tile and workgroup
    boundaries, random rows); absolute error of an output that is an average of N(0,1) values."""
    B, H = 2, 2
    q, k, v, C, Tp = _attn_inputs(B, H, T, torch.float16, T, pad=pad)
    got = _attn_run(L, q, k, v, B, H, T, C, Tp, _lib.DT_F16, torch.float16, thr=15.0, variant=variant)
    g = torch.Generator(device="cpu").manual_seed(1)
    rows = torch.cat([torch.tensor([0, 1, 31, 32, 63, 64, 127, 128, T - 130, T - 129, T - 65, T - 64, T - 2, T - 1]),
                      torch.randint(0, T, (50,), generator=g)]).unique().to(dev())
    ref = _attn_ref(q, k, v, T, rows)
    err = float((got[:, rows].double() - ref).abs().max())
    scale = float(ref.abs().max())
    assert err < 1e-3 * max(scale, 0.05) + 2e-4, (err, scale)        # f16 output rounding of values ~0.05-0.2: ~1e-4
    assert torch.isfinite(got[:, :T].float()).all()


@pytest.mark.parametrize("name,dt,tdt,tol", DTYPES)
def test_attention_reference_level_branches_are_exact(L, name, dt, tdt, tol):
    """The kernel subtracts a per-query reference level inside the matrix pipe; it is the exact row maximum of key tile 0 and
    moves again only when exp2(score - level) would leave the 16-bit range (a lane's tile sum > 2^15).  Inputs that FORCE every
    branch (cdna guide: a rare data-dependent branch needs its own test): keys whose score beats the level by +20 ... +60 late in
    the loop (rescale with score recomputation), a row whose first tiles lie far BELOW every later key, rows that never move.
    The shipped limit (2^15), "rescale nearly every tile" (2^0) and an intermediate limit must agree to one output ulp, and all
    kernel forms must match an fp64 reference of the FULL tensor."""
    B, H, T = 2, 2, 900
    q, k, v, C, Tp = _attn_inputs(B, H, T, tdt, 5, qscale=3.0)
    kk = k.float()
    qq = q.float()
    # late spikes: keys in tiles 10, 5 and 1 aligned with query rows of head 0 -> scores +50 / +25 / +60 (exp2 domain)
    for (row, key, target) in [(5, 700, 50.0), (130, 333, 25.0), (899, 64, 60.0), (64, 899, 20.0)]:
        d = qq[0, row, 0]
        kk[0, key, 0] = d / d.norm() ** 2 * target
    # a query row whose every score is very negative in the first tiles: all keys 0..191 of image 1 / head 1 point against it
    d = qq[1, 7, 1]
    kk[1, :192, 1] = -d / d.norm() ** 2 * 40.0 + 0.01 * kk[1, :192, 1]
    k = kk.to(tdt)
    ref = _attn_ref(q, k, v, T)
    outs = {lim: _attn_run(L, q, k, v, B, H, T, C, Tp, dt, tdt, thr=lim)[:, :T].double() for lim in (15.0, 0.0, 6.0)}
    bound = 4e-3 if name == "f16" else 3e-2
    for lim, got in outs.items():
        assert torch.isfinite(got).all(), lim
        err = float((got - ref).abs().max())
        assert err < bound, (lim, err)
    for lim in (15.0, 6.0):      # same function, different rounding order: at most one output ulp apart (|O| <= ~4)
        assert float((outs[lim] - outs[0.0]).abs().max()) <= (2.0 ** -8 if name == "f16" else 2.0 ** -5), lim
    for variant in (1, 2, 4, 5, 6):       # the A/B forms compute the same function
        got = _attn_run(L, q, k, v, B, H, T, C, Tp, dt, tdt, thr=15.0, variant=variant)[:, :T].double()
        assert float((got - ref).abs().max()) < bound, variant
    for var in (4, 6):                 # 64 queries per wave / 16x16x32 MFMAs: the flag path (workgroup numbering of the fallback kernel) at low limits
        for lim in (0.0, 6.0):
            got = _attn_run(L, q, k, v, B, H, T, C, Tp, dt, tdt, thr=lim, variant=var)[:, :T].double()
            assert float((got - ref).abs().max()) < bound, (var, lim)
            assert float((got - outs[0.0]).abs().max()) <= (2.0 ** -8 if name == "f16" else 2.0 ** -5), (var, lim)


# ------------------------------------------------------------------------------------------------------ norms, patchify
@pytest.mark.parametrize("name,dt,tdt,tol", DTYPES)
@pytest.mark.parametrize("C", [384, 768, 1024])
def test_layernorm16(L, name, dt, tdt, tol, C):
    x = torch.randn(301, C, device=dev()) * 3 + 1
    w, b = torch.randn(C, device=dev()), torch.randn(C, device=dev())
    out = torch.zeros(301, C, dtype=tdt, device=dev())
    _lib.check(L.mhmr_layernorm16(x.data_ptr(), w.data_ptr(), b.data_ptr(), out.data_ptr(), 301, C, 1e-6, dt, stream()), "ln")
    ref = torch.nn.functional.layer_norm(x, (C,), w, b, 1e-6)
    assert maxrel(out.float(), ref) < tol


def test_linear_f32_and_layernorm_f32(L):
    g = torch.Generator(device="cpu").manual_seed(3)
    # (K % 64 == 0, K >= 256 and few tiles: the split-k form; otherwise one wave per 16 x 32 tile over the whole k range)
    for (M, N, K, act) in [(37, 130, 48, 0), (5, 2, 384, 1), (64, 1024, 1456, 2), (257, 341, 1024, 0), (256, 1024, 1024, 2), (300, 159, 2048, 1)]:
        X = torch.randn(M, K, generator=g).to(dev())
        W = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dev())
        b = torch.randn(N, generator=g).to(dev())
        R = torch.randn(M, N, generator=g).to(dev())
        Y = R.clone()
        _lib.check(L.mhmr_linear_f32(X.data_ptr(), K, None, W.data_ptr(), K, b.data_ptr(), Y.data_ptr(), N, Y.data_ptr(), N, M, N, K, act,
                                     stream()), "linear")
        ref = X.double() @ W.double().T + b.double()
        ref = torch.relu(ref) if act == 1 else torch.nn.functional.gelu(ref) if act == 2 else ref
        assert maxrel(Y, ref + R.double()) < 1e-5, (M, N, K, act)
    # gathered rows
    X = torch.randn(50, 64, device=dev())
    W = torch.randn(20, 64, device=dev())
    ridx = torch.tensor([3, 3, 49, 0, 17], dtype=torch.int32, device=dev())
    Y = torch.zeros(5, 20, device=dev())
    _lib.check(L.mhmr_linear_f32(X.data_ptr(), 64, ridx.data_ptr(), W.data_ptr(), 64, None, None, 0, Y.data_ptr(), 20, 5, 20, 64, 0, stream()), "linear")
    assert maxrel(Y, X[ridx.long()].double() @ W.double().T) < 1e-5
    X = torch.randn(50, 256, device=dev())
    W = torch.randn(20, 256, device=dev()) / 16
    Y = torch.zeros(5, 20, device=dev())
    _lib.check(L.mhmr_linear_f32(X.data_ptr(), 256, ridx.data_ptr(), W.data_ptr(), 256, None, None, 0, Y.data_ptr(), 20, 5, 20, 256, 0, stream()), "linear")
    assert maxrel(Y, X[ridx.long()].double() @ W.double().T) < 1e-5
    x = torch.randn(33, 1024, device=dev()) * 2 - 0.5
    w, b = torch.randn(1024, device=dev()), torch.randn(1024, device=dev())
    out = torch.zeros_like(x)
    _lib.check(L.mhmr_layernorm_f32(x.data_ptr(), w.data_ptr(), b.data_ptr(), out.data_ptr(), 33, 1024, 1e-5, stream()), "ln32")
    assert maxrel(out, torch.nn.functional.layer_norm(x.double(), (1024,), w.double(), b.double(), 1e-5)) < 1e-5


# ------------------------------------------------------------------------------------------------------ detection
@pytest.mark.parametrize("k", [1, 2, 3, 4, 5])
def test_nms_threshold_compaction(L, k):
    from oracle.multihmr_ref import nms
    B, G = 3, 16
    g = torch.Generator(device="cpu").manual_seed(k)
    s = torch.rand(B, G * G, generator=g)
    s[0, 5] = s[0, 6] = 0.93          # an exact tie between neighbours: both survive (hmax == heat)
    s[2] *= 0.1                         # an image without detections
    thr = 0.8
    sc = s.to(dev()).contiguous()
    counts = torch.zeros(B, dtype=torch.int32, device=dev())
    _lib.check(L.mhmr_detect_count(sc.data_ptr(), B, G, k, thr, counts.data_ptr(), stream()), "count")
    heat = s.view(B, 1, G, G)
    heat = nms(heat, k) if k > 1 else heat
    ref_idx = torch.where(heat.permute(0, 2, 3, 1) >= thr)
    ref_counts = torch.bincount(ref_idx[0], minlength=B)
    assert counts.cpu().tolist() == ref_counts.tolist()
    P = int(ref_counts.sum())
    base = (torch.cumsum(counts, 0) - counts).to(torch.int32)
    det = torch.zeros(3, P, dtype=torch.int32, device=dev())
    dsc = torch.zeros(P, device=dev())
    _lib.check(L.mhmr_detect_write(sc.data_ptr(), B, G, k, thr, base.data_ptr(), det[0].data_ptr(), det[1].data_ptr(), det[2].data_ptr(),
                                   dsc.data_ptr(), stream()), "write")
    assert det.cpu().long().tolist() == [ref_idx[0].tolist(), ref_idx[1].tolist(), ref_idx[2].tolist()]
    assert torch.equal(dsc.cpu(), heat[ref_idx[0], ref_idx[3], ref_idx[1], ref_idx[2]])


@pytest.mark.parametrize("nbands,maxres", [(16, 64), (8, 64), (20, 32), (1, 64)])      # 16 / 64: the released checkpoints (model.py:39-40)
def test_camera_embed(L, nbands, maxres):
    from oracle.multihmr_ref import embedd_camera
    B, G, C, Kc = 2, 16, 384, 512
    E = 3 + 6 * nbands
    K = synthetic.get_camera_K(G * 14, B)
    K[1, 0, 0] *= 1.1
    K[1, 0, 2] += 5
    K[1, 0, 1] = 0.3       # a skewed camera exercises the general 3x3 inverse
    freq = torch.stack([torch.linspace(1.0, maxres / 2, nbands) for _ in range(3)]).to(dev()).contiguous()
    zK = torch.zeros(B * G * G, E, device=dev())
    ctx = torch.full((B * G * G, Kc), 5.0, dtype=torch.float16, device=dev())
    Kd = K.to(dev()).contiguous()
    _lib.check(L.mhmr_camera_embed(Kd.data_ptr(), freq.data_ptr(), B, G, 14, zK.data_ptr(), ctx.data_ptr(), Kc, C, _lib.DT_F16, nbands, stream()), "cam")
    ref = embedd_camera(K, G, nbands, maxres).reshape(B * G * G, E)
    assert float((zK.cpu() - ref).abs().max()) < 2e-4     # sin/cos of arguments up to ~60 rad in fp32
    assert torch.all(ctx[:, :C] == 5.0) and torch.all(ctx[:, C + E:] == 0.0)
    assert float((ctx[:, C:C + E].float().cpu() - ref).abs().max()) < 2e-3
    # more bands than one thread per camera column can write behind the features: refused, not truncated
    assert L.mhmr_camera_embed(Kd.data_ptr(), freq.data_ptr(), B, G, 14, zK.data_ptr(), ctx.data_ptr(), Kc, C, _lib.DT_F16, 22, stream()) == -2


# ------------------------------------------------------------------------------------------------------ LBS
@pytest.mark.parametrize("P,center", [(1, 15), (5, 15), (70, 15), (300, 15), (37, None), (37, 0)])   # 300: one full 256-person slab + a remainder
def test_lbs_against_oracle(L, smplx_data, P, center):
    """center: person_center joint (15 = 'head', the released checkpoints; 0 = 'pelvis'; None = vanilla SMPL-X placement,
    reference blocks/smpl_layer.py:128-136)."""
    import ctypes as C
    from oracle import smplx_ref
    from oracle.multihmr_ref import smpl_layer_forward
    pk = packing.pack_smplx(smplx_data, 10, dev(), -1 if center is None else center)
    cs = packing.lbs_consts_struct(pk)
    g = torch.Generator(device="cpu").manual_seed(P)
    pose = 0.35 * torch.randn(P, 53, 3, generator=g)
    pose[0, 3] = 0                                            # a zero rotation vector
    shape, expr = torch.randn(P, 10, generator=g), torch.randn(P, 10, generator=g)
    B = 3
    det_b = torch.randint(0, B, (P,), generator=g).sort().values
    K = synthetic.get_camera_K(448, B)
    K[:, 0, 2] += torch.arange(B) * 4.0
    loc = 448 * torch.rand(P, 2, generator=g)
    dist = 2 + 6 * torch.rand(P, 1, generator=g)
    ref = smpl_layer_forward(smplx_ref.SMPLX(smplx_data, num_betas=10), pose, shape, loc, dist, K[det_b], expr, person_center_idx=center)
    d = lambda t, dt=torch.float32: t.to(device=dev(), dtype=dt).contiguous()
    V = pk["V"]
    f = lambda *s: torch.zeros(*s, device=dev())
    v3d, v2d, j3d, j2d, transl = f(P, V, 3), f(P, V, 2), f(P, 127, 3), f(P, 127, 2), f(P, 3)
    wsF, wsA, wsX = f(packing.roundup(P, 16), pk["Kb"]), f(packing.roundup(P, 16), 768), f(P, 24)
    args = [d(pose), d(shape), d(expr), d(loc), d(dist), d(K), d(det_b, torch.int32)]
    _lib.check(L.mhmr_lbs_forward(C.byref(cs), *[a.data_ptr() for a in args], P, wsF.data_ptr(), wsA.data_ptr(), wsX.data_ptr(),
                                  v3d.data_ptr(), v2d.data_ptr(), j3d.data_ptr(), j2d.data_ptr(), transl.data_ptr(), stream()), "lbs")
    # metres.  Vertices (and the extra joints made of them): the pose correctives run as ONE f16 product per term (csrc/lbs.hip,
    # LBS_EBYTES_HI: rms 4e-6 m, worst 2.5e-5 m on this synthetic basis whose correctives reach 8 cm), everything else at fp32 accuracy
    for name, got, tol in (("v3d", v3d, 5e-5), ("j3d", j3d, 5e-5), ("transl", transl, 2e-5)):
        err = float((got.cpu() - ref[name]).abs().max())
        assert err < tol, (name, err)
    assert float((j3d[:, :55].cpu() - ref["j3d"][:, :55]).abs().max()) < 2e-5          # the 55 posed joints do not pass through the blend
    for name, got in (("v2d", v2d), ("j2d", j2d)):
        err = float((got.cpu() - ref[name]).abs().max())
        assert err < 1.5e-2, (name, err)                      # pixels
    assert float((j3d[:, [0]].cpu() - ref["transl_pelvis"]).abs().max()) < 2e-5
    # joints 55..75 are vertices picked by id: the extra-joint tiles of the vertex kernel (virtual vertices with corner weights (1, 0, 0))
    # must reproduce those vertices bit for bit, in 3D and in the image
    vid = pk["extra_vid"].long()
    assert torch.equal(j3d[:, 55:76], v3d[:, vid]) and torch.equal(j2d[:, 55:76], v2d[:, vid])
    # joints 76..126: barycentric face landmarks against the same combination of the kernel's own vertices (fp32 rounding order only)
    lm = (v3d[:, pk["lmk_vidx"].long()] * pk["lmk_bary"][None, :, :, None]).sum(2)
    assert float((j3d[:, 76:] - lm).abs().max()) < 2e-6


@pytest.mark.parametrize("P", [1, 5, 16, 20, 70, 160, 161, 300])
def test_lbs_fused_launch_is_bit_identical_and_leaves_its_workspace_clean(L, smplx_data, P):
    """mhmr_lbs_forward_fused: the pose role as the leading workgroups of the vertex grid, per-person ready flags between them.  Same
    arithmetic as the two launches of mhmr_lbs_forward -> every output bit-identical; the flag workspace is zero again after every call
    (the last vertex workgroup clears it), so the SAME workspace serves many calls back to back, with different inputs, without a
    memset between them; P > 160 takes the two-launch path and never touches the workspace."""
    import ctypes as C
    pk = packing.pack_smplx(smplx_data, 10, dev(), 15)
    cs = packing.lbs_consts_struct(pk)
    V, B = pk["V"], 4
    d = lambda t, dt=torch.float32: t.to(device=dev(), dtype=dt).contiguous()
    f = lambda *s: torch.full(s, float("nan"), device=dev())
    sync = torch.zeros(1 + packing.roundup(P, 16), dtype=torch.int32, device=dev())
    if P > 160:
        sync.fill_(7)                                         # the fallback path must not read or write it
    K = synthetic.get_camera_K(896, B)
    K[:, 0, 2] += torch.arange(B) * 4.0

    def inputs(seed):
        g = torch.Generator(device="cpu").manual_seed(100 * P + seed)
        pose = 0.35 * torch.randn(P, 53, 3, generator=g)
        shape, expr = torch.randn(P, 10, generator=g), torch.randn(P, 10, generator=g)
        det_b = torch.randint(0, B, (P,), generator=g).sort().values
        loc, dist = 896 * torch.rand(P, 2, generator=g), 2 + 6 * torch.rand(P, 1, generator=g)
        return [d(pose), d(shape), d(expr), d(loc), d(dist), d(K), d(det_b, torch.int32)]

    def run(fused, args):
        outs = [f(P, V, 3), f(P, V, 2), f(P, 127, 3), f(P, 127, 2), f(P, 3)]
        ws = [f(packing.roundup(P, 16), pk["Kb"]), f(packing.roundup(P, 16), 768), f(P, 24)]       # NaN-poisoned: nothing may be read unwritten
        ptrs = [a.data_ptr() for a in args] + [P] + [w.data_ptr() for w in ws] + [o.data_ptr() for o in outs]
        if fused:
            _lib.check(L.mhmr_lbs_forward_fused(C.byref(cs), *ptrs, sync.data_ptr(), stream()), "lbs fused")
        else:
            _lib.check(L.mhmr_lbs_forward(C.byref(cs), *ptrs, stream()), "lbs")
        return outs

    for rep in range(6):                                      # back to back on one stream, new inputs every time, no memset in between
        args = inputs(rep % 3)
        got = run(True, args)
        ref = run(False, args)
        for name, a, b in zip(("v3d", "v2d", "j3d", "j2d", "transl"), got, ref):
            assert bool(torch.isfinite(a).all()), (name, rep)
            assert torch.equal(a, b), (name, rep, float((a - b).abs().max()))
        torch.cuda.synchronize()
        assert int(sync.abs().sum()) == (0 if P <= 160 else 7 * sync.numel()), (rep, sync[:8].tolist())


def test_lbs_max_abs_gate_160_persons_x_20_seeds(L, smplx_data):
    """The SLP-packed build of the vertex kernel's epilogue once returned, for about one (person, vertex tile) pair in 10^4, a projection
    computed with a zero focal length (csrc/lbs.hip is built with -fno-slp-vectorize, enforced by a compile-time check in the source).
    A relative-L2 comparison barely notices one 500-pixel outlier among 10^6 values: this gate is MAX-ABS over 160 persons x 20 seeds
    (3200 persons x 10475 vertices = 7 x 10^5 (person, tile) pairs) for v3d AND v2d."""
    import ctypes as C
    from oracle import smplx_ref
    from oracle.multihmr_ref import smpl_layer_forward
    pk = packing.pack_smplx(smplx_data, 10, dev(), 15)
    cs = packing.lbs_consts_struct(pk)
    bm = smplx_ref.SMPLX(smplx_data, num_betas=10)
    P, B, V = 160, 8, pk["V"]
    d = lambda t, dt=torch.float32: t.to(device=dev(), dtype=dt).contiguous()
    f = lambda *s: torch.zeros(*s, device=dev())
    v3d, v2d, j3d, j2d, transl = f(P, V, 3), f(P, V, 2), f(P, 127, 3), f(P, 127, 2), f(P, 3)
    wsF, wsA, wsX = f(packing.roundup(P, 16), pk["Kb"]), f(packing.roundup(P, 16), 768), f(P, 24)
    worst3, worst2 = 0.0, 0.0
    for seed in range(20):
        g = torch.Generator(device="cpu").manual_seed(1000 + seed)
        pose = 0.35 * torch.randn(P, 53, 3, generator=g)
        shape, expr = torch.randn(P, 10, generator=g), torch.randn(P, 10, generator=g)
        det_b = torch.randint(0, B, (P,), generator=g).sort().values
        K = synthetic.get_camera_K(1288, B)
        K[:, 0, 2] += torch.arange(B) * 4.0
        K[:, 1, 1] *= 1.0 + 0.03 * torch.arange(B)
        loc = 1288 * torch.rand(P, 2, generator=g)
        dist = 2 + 6 * torch.rand(P, 1, generator=g)
        ref = smpl_layer_forward(bm, pose, shape, loc, dist, K[det_b], expr, person_center_idx=15)
        args = [d(pose), d(shape), d(expr), d(loc), d(dist), d(K), d(det_b, torch.int32)]
        _lib.check(L.mhmr_lbs_forward(C.byref(cs), *[a.data_ptr() for a in args], P, wsF.data_ptr(), wsA.data_ptr(), wsX.data_ptr(),
                                      v3d.data_ptr(), v2d.data_ptr(), j3d.data_ptr(), j2d.data_ptr(), transl.data_ptr(), stream()), "lbs")
        worst3 = max(worst3, float((v3d.cpu() - ref["v3d"]).abs().max()), float((j3d.cpu() - ref["j3d"]).abs().max()))
        worst2 = max(worst2, float((v2d.cpu() - ref["v2d"]).abs().max()), float((j2d.cpu() - ref["j2d"]).abs().max()))
    assert worst3 < 5e-5, worst3           # metres (one-product pose correctives: csrc/lbs.hip LBS_EBYTES_HI; 2e-5 before round 4)
    assert worst2 < 4e-2, worst2           # pixels at 1288^2 (focal ~1115 px: 5e-5 m at 2 m is 3e-2 px)


def test_attention_is_bit_reproducible(L):
    """Regression: with the compiler-inserted waits only, hipcc hoisted the LDS-DMA vmcnt wait out of the KV loop and this
    configuration (several workgroups per CU) gave different results on every run."""
    B, H, T = 8, 16, 2305
    C, Tp = H * 64, packing.roundup(T, 128)
    g = torch.Generator(device="cuda:0").manual_seed(0)
    qk = (torch.randn(B * Tp, 2 * C, device=dev(), generator=g) * 0.5).to(torch.bfloat16)
    vt = (torch.randn(B * H * 64, Tp, device=dev(), generator=g) * 0.5).to(torch.bfloat16)
    ref = None
    for _ in range(8):
        out = torch.zeros(B * Tp, C, dtype=torch.bfloat16, device=dev())
        _lib.check(L.mhmr_attention16(qk.data_ptr(), vt.data_ptr(), out.data_ptr(), B, T, Tp, C, H, _lib.DT_BF16, stream()), "attention")
        torch.cuda.synchronize()
        if ref is None:
            ref = out
        assert torch.equal(ref, out)


# ---------------------------------------------------------------------------------------------------------------- fp8 low-half range (round 5)
def _lo8_operands(M, N, K, seed, hot=True):
    from multi_hmr_amd import vit
    g = torch.Generator(device="cpu").manual_seed(seed)
    A32 = torch.randn(M, K, generator=g)
    if hot:
        A32[:, 5] *= 60.0                       # a massive-activation channel: bf8 (e5m2) has the range for it
    W32 = torch.randn(N, K, generator=g) * 0.03
    A16 = A32.half()
    A8 = A32.to(torch.float8_e5m2)
    Arow = torch.cat([A16.view(torch.uint8).reshape(M, 2 * K), A8.view(torch.uint8)], 1).contiguous()          # [M, 3K] bytes
    W16 = W32.half()
    Wrow, scale, lo_deq = vit.lo8_rows(W16, W32)                                                                 # [N, 3K] bytes
    exact = A16.double() @ W32.double().T                                              # what the low half is for: the UNROUNDED weight
    emul = A16.double() @ W16.double().T + A8.double() @ lo_deq.T                      # what the kernel computes, term by term
    return A32, W32, Arow.to(dev()), Wrow.to(dev()), scale, exact.to(dev()), emul.to(dev()), A16.to(dev()), W16.to(dev())


@pytest.mark.parametrize("M,N,K", [(256, 256, 256), (512, 256, 1024), (256, 512, 768)])
def test_gemm_fp8_low_half_range(L, M, N, K):
    """GemmArgs::lo8 (csrc/gemm256.hip): operand rows [K 16-bit values | K bytes fp8]; the k tiles behind K run on
    v_mfma_scale_f32_16x16x128_f8f6f4 (weight: e4m3 x 2^e, activation: e5m2).  Against the term-by-term emulation (fp64) the kernel is
    exact to fp32 accumulation -- layout, formats, the E8M0 scale and the k permutation of the fragments all have to be right for that --
    and against the unrounded weights the error is several times below a single f16 pass (the low half kept to three bits)."""
    A32, W32, Arow, Wrow, scale, exact, emul, A16, W16 = _lo8_operands(M, N, K, M + N + K)
    bias = torch.randn(N, generator=torch.Generator().manual_seed(1)).to(dev())
    gamma = (0.5 + torch.rand(N, generator=torch.Generator().manual_seed(2))).to(dev())
    # residual epilogue, with the producer outputs in a pitched x16 (+ its bf8 copy) and the block sums
    r0 = torch.randn(M, N, generator=torch.Generator().manual_seed(3)).to(dev())
    out = r0.clone()
    pit = N + N // 2
    x16 = torch.zeros(M, pit, dtype=torch.float16, device=dev())
    pstats = torch.zeros(M, N // 64, 2, device=dev())
    _lib.check(L.mhmr_gemm16_lo8(Arow.data_ptr(), K + K // 2, Wrow.data_ptr(), K + K // 2, M, N, K, 1, scale, bias.data_ptr(), gamma.data_ptr(),
                                 out.data_ptr(), N, 128, 1, _lib.EPI_RESID, _lib.DT_F16, 0, 0, x16.data_ptr(), pit, 2 * N, pstats.data_ptr(), None, None,
                                 None, stream()), "gemm lo8 resid")
    got = (out.double() - r0.double()) / gamma.double() - bias.double()
    e_emul, e_exact = rel(got, emul), rel(got, exact)
    one = A16.double() @ W16.double().T
    e_one = rel(one, exact)
    assert e_emul < 2e-5, e_emul                                    # (read back through out - r0: fp32 cancellation)
    assert e_one > 1e-4 and e_exact < e_one / 4, (e_exact, e_one)
    # producer outputs: the 16-bit copy of the new rows, their bf8 copy behind it, the per-64-column sums
    assert torch.equal(x16[:, :N], out.half())
    x8 = x16.view(torch.uint8).reshape(M, 2 * pit)[:, 2 * N:3 * N].contiguous().view(torch.float8_e5m2)
    want8 = out.to(torch.float8_e5m2)
    assert float((x8.view(torch.uint8) == want8.view(torch.uint8)).float().mean()) > 0.999
    assert float((x8.float() - out).abs().max() / out.abs().max()) < 0.13
    s1 = out.double().view(M, N // 64, 64).sum(-1)
    assert rel(pstats[..., 0].double(), s1) < 1e-5
    # V^T epilogue (the activation is the FIRST operand there), plain and with the folded LayerNorm
    if M % 128 == 0 and N % 64 == 0:
        B, H, Tp = 1, N // 64, M
        perm = swap23(torch.arange(Tp, device=dev()))
        vt = torch.zeros(B, H, 64, Tp, dtype=torch.float16, device=dev())
        _lib.check(L.mhmr_gemm16_lo8(Arow.data_ptr(), K + K // 2, Wrow.data_ptr(), K + K // 2, M, N, K, 1, scale, bias.data_ptr(), None, vt.data_ptr(), 0,
                                     Tp, H, _lib.EPI_VT, _lib.DT_F16, 0, 0, None, 0, 0, None, None, None, None, stream()), "gemm lo8 vt")
        ref = (emul + bias.double()).float().view(B, Tp, H, 64).permute(0, 2, 3, 1)
        assert maxrel(vt.float()[..., perm], ref) < 2e-3
        mean = A16.double().mean(1)
        rstd = 1.0 / (A16.double().var(1, unbiased=False) + 1e-6).sqrt()
        rowstats = torch.stack([mean, rstd], 1).float().contiguous()
        colsum = (W16.double().sum(1) + (emul - A16.double() @ W16.double().T).sum(0) * 0).float()       # placeholder shape; real sum below
        lo_deq = (Wrow[:, 2 * K:].contiguous().view(torch.float8_e4m3fn).double() * 2.0 ** (scale - 127))
        colsum = (W16.double().sum(1) + lo_deq.sum(1)).float().contiguous()
        vt2 = torch.zeros_like(vt)
        _lib.check(L.mhmr_gemm16_lo8(Arow.data_ptr(), K + K // 2, Wrow.data_ptr(), K + K // 2, M, N, K, 1, scale, None, None, vt2.data_ptr(), 0, Tp, H,
                                     _lib.EPI_VT, _lib.DT_F16, 0, 0, None, 0, 0, None, rowstats.data_ptr(), colsum.data_ptr(), bias.data_ptr(), stream()),
                   "gemm lo8 vt fold")
        ref2 = (rstd[:, None] * (emul - mean[:, None] * colsum.double()[None, :]) + bias.double()).float().view(B, Tp, H, 64).permute(0, 2, 3, 1)
        assert maxrel(vt2.float()[..., perm], ref2) < 3e-3


def test_attention_and_layernorm_with_a_row_pitch_and_the_bf8_copy(L):
    """The producers of the fp8 low-half range's activation bytes: attention (variant 6) and the LayerNorm kernel write rows of 3C/2
    elements -- C 16-bit values, bit-identical to the plain form, and behind them the bf8 (e5m2) copy of the same values."""
    B, H, T = 2, 4, 321
    C, Tp = 64 * H, 384
    pit = C + C // 2
    g = torch.Generator(device="cuda:0").manual_seed(5)
    qk = (torch.randn(B * Tp, 2 * C, device=dev(), generator=g) * 0.5).half()
    vt = (torch.randn(B * H * 64, Tp, device=dev(), generator=g) * 0.5).half()
    nfl = L.mhmr_attention_flag_count(B, Tp, H)
    flags = torch.zeros(nfl, dtype=torch.int32, device=dev())
    plain = torch.zeros(B * Tp, C, dtype=torch.float16, device=dev())
    _lib.check(L.mhmr_attention16_ex(qk.data_ptr(), vt.data_ptr(), plain.data_ptr(), B, T, Tp, C, H, _lib.DT_F16, 15.0, 6, flags.data_ptr(), stream()), "attn")
    wide = torch.zeros(B * Tp, pit, dtype=torch.float16, device=dev())
    _lib.check(L.mhmr_attention16_pitch(qk.data_ptr(), vt.data_ptr(), wide.data_ptr(), B, T, Tp, C, H, _lib.DT_F16, flags.data_ptr(), pit, 2 * C, stream()), "attn pitch")
    assert torch.equal(wide[:, :C], plain)
    a8 = wide.view(torch.uint8).reshape(B * Tp, 2 * pit)[:, 2 * C:3 * C].contiguous().view(torch.float8_e5m2).float()
    real = torch.arange(B * Tp, device=dev()) % Tp < T
    err = (a8[real] - plain[real].float()).abs()
    assert float((err <= 0.13 * plain[real].float().abs() + 1e-4).float().mean()) > 0.9999
    assert float(a8[real].abs().max()) > 0
    # LayerNorm
    rows, Cl = 200, 768
    x = (torch.randn(rows, Cl, device=dev(), generator=g) * 2 + 0.7)
    w, b = torch.rand(Cl, device=dev(), generator=g) + 0.5, torch.randn(Cl, device=dev(), generator=g)
    p2 = Cl + Cl // 2
    o1 = torch.zeros(rows, Cl, dtype=torch.float16, device=dev())
    o2 = torch.zeros(rows, p2, dtype=torch.float16, device=dev())
    _lib.check(L.mhmr_layernorm16(x.data_ptr(), w.data_ptr(), b.data_ptr(), o1.data_ptr(), rows, Cl, 1e-6, _lib.DT_F16, stream()), "ln")
    _lib.check(L.mhmr_layernorm16_pitch(x.data_ptr(), w.data_ptr(), b.data_ptr(), o2.data_ptr(), p2, 2 * Cl, rows, Cl, 1e-6, _lib.DT_F16, stream()), "ln pitch")
    assert torch.equal(o2[:, :Cl], o1)
    y8 = o2.view(torch.uint8).reshape(rows, 2 * p2)[:, 2 * Cl:3 * Cl].contiguous().view(torch.float8_e5m2).float()
    ref = torch.nn.functional.layer_norm(x, (Cl,), w, b, 1e-6)
    assert float(((y8 - ref).abs() <= 0.13 * ref.abs() + 1e-4).float().mean()) > 0.9999


@pytest.mark.parametrize("name,dt,tdt,tol", DTYPES)
@pytest.mark.parametrize("M,N,K,a_k", [(4352, 1024, 4096, 0),      # fc2 of multiHMR_896_L at batch 1: 68 tiles -> 3 slices of 22 / 22 / 20 k tiles
                                        (4352, 1024, 2048, 1024),   # its output projection with the low-half k range: 12 / 12 / 8, one slice straddles the wrap
                                        (2560, 1024, 1024, 0),      # 672^2: 40 tiles -> 4 slices of 4 k tiles
                                        (2560, 768, 3072, 0),       # ViT-B: 30 tiles -> 8 slices of 6
                                        (256, 256, 512, 0)])        # one tile, two slices
def test_splitk_residual_linear(L, name, dt, tdt, tol, M, N, K, a_k):
    """mhmr_gemm16_splitk_resid = split-k GEMM (fp32 partial tiles, slices summed in slice order) + the row-wise residual epilogue that also
    leaves the op16 copy and the (mean, rstd) of every updated row: against fp64 torch, against the unsplit residual linear of the same
    operands (same products, another summation order), and bit-reproducible."""
    g = torch.Generator(device=dev()).manual_seed(M + N + K + a_k)
    ka = a_k if a_k else K
    A = torch.randn(M, ka, generator=g, device=dev()).to(tdt)
    W = (torch.randn(N, K, generator=g, device=dev()) / math.sqrt(ka)).to(tdt)
    if a_k:
        W[:, a_k:] *= 2.0 ** -11                      # a low half's magnitude
    bias, gamma = torch.randn(N, generator=g, device=dev()), 0.5 + torch.rand(N, generator=g, device=dev())
    r0 = torch.randn(M, N, generator=g, device=dev())
    nbytes = L.mhmr_splitk_workspace_bytes(M, N, K)
    assert nbytes > 0 and nbytes % (M * N * 4) == 0 and 2 <= nbytes // (M * N * 4) <= 8
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=dev())
    outs = []
    for _ in range(2):
        r = r0.clone()
        x16 = torch.zeros(M, N, dtype=tdt, device=dev())
        st = torch.zeros(M, 2, device=dev())
        ws.fill_(float("nan"))
        _lib.check(L.mhmr_gemm16_splitk_resid(A.data_ptr(), ka, W.data_ptr(), K, M, N, K, a_k, bias.data_ptr(), gamma.data_ptr(), r.data_ptr(),
                                              x16.data_ptr(), 0, st.data_ptr(), 1e-6, ws.data_ptr(), nbytes, dt, stream()), "splitk")
        outs.append((r, x16, st))
    (r, x16, st), (r2, x162, st2) = outs
    assert torch.equal(r, r2) and torch.equal(x16, x162) and torch.equal(st, st2)
    Aw = torch.cat([A, A], 1) if a_k else A
    ref = r0.double() + gamma.double() * (Aw.double() @ W.double().T + bias.double())
    assert rel(r, ref) < 2e-6 and maxrel(r, ref) < 1e-5
    assert torch.equal(x16, r.to(tdt))
    mean, var = ref.mean(1), ref.var(1, unbiased=False)
    assert maxrel(st[:, 0], mean) < 1e-5 and maxrel(st[:, 1], (var + 1e-6).rsqrt()) < 1e-5
    # the unsplit residual linear: the same fp32 products in another order
    r3 = r0.clone()
    _lib.check(L.mhmr_gemm16_ex(A.data_ptr(), ka, W.data_ptr(), K, M, N, K, bias.data_ptr(), gamma.data_ptr(), r3.data_ptr(), N, None, 0, 128, 1,
                                M, _lib.EPI_RESID, dt, 0, 0, a_k, stream()), "gemm")
    assert rel(r, r3) < 1e-6


def test_splitk_plan_declines_long_launches(L):
    """A launch that already fills more than half of the CUs, or a k range too short to cut, is not split (0 bytes)."""
    assert L.mhmr_splitk_workspace_bytes(256 * 129, 1024, 4096) == 0      # 516 tiles
    assert L.mhmr_splitk_workspace_bytes(4352, 1024, 256) == 0             # four k tiles
    assert L.mhmr_splitk_workspace_bytes(4352 + 128, 1024, 4096) == 0      # M not a multiple of 256


@pytest.mark.parametrize("name,dt,tdt,tol", DTYPES)
@pytest.mark.parametrize("K,a_k", [(384, 0), (1536, 0), (768, 384)])
def test_gemm_masked_output_width(L, name, dt, tdt, tol, K, a_k):
    """mhmr_gemm16_masked: an output width of 384 (ViT-S) on the 256x256 kernel as N = 512 with zero-padded weight rows / per-column vectors
    and the last 128 columns masked -- the residual epilogue (with the LayerNorm-fold producer outputs at the real width) and the V^T
    epilogue (plain and as a fold consumer) against fp64 torch; nothing may be written behind column 384 / head 6."""
    M, Nv, Np, Tp = 512, 384, 512, 256
    B, H = M // Tp, Nv // 64
    ka = a_k if a_k else K
    g = torch.Generator(device=dev()).manual_seed(K + a_k)
    A = torch.randn(M, ka, generator=g, device=dev()).to(tdt)
    Aw = torch.cat([A, A], 1) if a_k else A

    def padded(t):
        return torch.cat([t, torch.zeros(Np - Nv, *t.shape[1:], dtype=t.dtype, device=t.device)], 0).contiguous()

    W = (torch.randn(Nv, K, generator=g, device=dev()) / math.sqrt(ka)).to(tdt)
    bias, gamma = torch.randn(Nv, generator=g, device=dev()), 0.5 + torch.rand(Nv, generator=g, device=dev())
    Wp, bp, gp = padded(W), padded(bias), padded(gamma)
    # ---- residual epilogue + fold producer ----
    r0 = torch.randn(M + 1, Nv, generator=g, device=dev())             # one guard row behind the matrix
    r = r0.clone()
    x16 = torch.full((M + 1, Nv), 7.0, dtype=tdt, device=dev())
    pst = torch.full((M + 1, Nv // 64, 2), -1.0, device=dev())
    _lib.check(L.mhmr_gemm16_masked(A.data_ptr(), ka, Wp.data_ptr(), K, M, Np, Nv, K, a_k, bp.data_ptr(), gp.data_ptr(), r.data_ptr(), Nv, Tp, H,
                                    _lib.EPI_RESID, dt, x16.data_ptr(), pst.data_ptr(), None, None, None, stream()), "masked resid")
    ref = r0[:M].double() + gamma.double() * (Aw.double() @ W.double().T + bias.double())
    assert maxrel(r[:M], ref) < 2e-5 and torch.equal(r[M], r0[M])
    assert torch.equal(x16[:M], r[:M].to(tdt)) and torch.all(x16[M] == 7.0)
    blocks = r[:M].view(M, Nv // 64, 64)
    assert maxrel(pst[:M, :, 0], blocks.sum(-1)) < 1e-5 and maxrel(pst[:M, :, 1], (blocks * blocks).sum(-1)) < 1e-5 and torch.all(pst[M] == -1.0)
    # ---- V^T epilogue: plain, and as the consumer of a folded LayerNorm ----
    perm = swap23(torch.arange(Tp, device=dev()))
    for fold in (False, True):
        vt = torch.full((B * H + 1, 64, Tp), 7.0, dtype=tdt, device=dev())          # one guard head
        if fold:
            rs = torch.stack([torch.randn(M, generator=g, device=dev()) * 0.3, 0.5 + torch.rand(M, generator=g, device=dev())], 1).contiguous()
            colsum = padded(W.double().sum(1).float())
            _lib.check(L.mhmr_gemm16_masked(A.data_ptr(), ka, Wp.data_ptr(), K, M, Np, Nv, K, a_k, None, None, vt.data_ptr(), 0, Tp, H, _lib.EPI_VT, dt,
                                            None, None, rs.data_ptr(), colsum.data_ptr(), bp.data_ptr(), stream()), "masked vt fold")
            want = rs[:, 1:2].double() * (Aw.double() @ W.double().T - rs[:, 0:1].double() * W.double().sum(1)) + bias.double()
        else:
            if K < 256:
                continue
            _lib.check(L.mhmr_gemm16_masked(A.data_ptr(), ka, Wp.data_ptr(), K, M, Np, Nv, K, a_k, bp.data_ptr(), None, vt.data_ptr(), 0, Tp, H, _lib.EPI_VT, dt,
                                            None, None, None, None, None, stream()), "masked vt")
            want = Aw.double() @ W.double().T + bias.double()
        got = vt[: B * H].float().view(B, H, 64, Tp)[..., perm].permute(0, 3, 1, 2).reshape(M, Nv)
        assert maxrel(got, want) < tol, (fold, maxrel(got, want))
        assert torch.all(vt[B * H] == 7.0)
    # the unmasked kernels refuse a width that is not a multiple of 256; the masked form refuses anything but N = n_valid + 128
    assert L.mhmr_gemm16_masked(A.data_ptr(), ka, Wp.data_ptr(), K, M, Np, 256, K, a_k, bp.data_ptr(), gp.data_ptr(), r.data_ptr(), Nv, Tp, H,
                                _lib.EPI_RESID, dt, None, None, None, None, None, stream()) != 0


@pytest.mark.parametrize("name,dt,tdt,tol", DTYPES)
@pytest.mark.parametrize("fold", [False, True])
def test_merged_qkv_launch_equals_the_two_launches(L, name, dt, tdt, tol, fold):
    """mhmr_qkv16 (one launch for Q | K | V + the transpose of the V rows; a short batch) against the Q | K launch and the V^T launch it
    replaces: the same products in the same order -> bit-identical qk and vt; and against fp64."""
    B, Tp, C = 2, 512, 256
    H, M = C // 64, B * Tp
    g = torch.Generator(device=dev()).manual_seed(11 + fold)
    A = torch.randn(M, C, generator=g, device=dev()).to(tdt)
    W = (torch.randn(3 * C, C, generator=g, device=dev()) / math.sqrt(C)).to(tdt)
    b = torch.randn(3 * C, generator=g, device=dev())
    rs = torch.stack([torch.randn(M, generator=g, device=dev()) * 0.3, 0.5 + torch.rand(M, generator=g, device=dev())], 1).contiguous()
    colsum = W.double().sum(1).float().contiguous()
    qk, v16, vt = (torch.full(s, 7.0, dtype=tdt, device=dev()) for s in ((M, 2 * C), (M, C), (B, H, 64, Tp)))
    st = (rs.data_ptr(), colsum.data_ptr(), b.data_ptr()) if fold else (None, None, None)
    _lib.check(L.mhmr_qkv16(A.data_ptr(), C, W.data_ptr(), C, B, Tp, C, H, None if fold else b.data_ptr(), qk.data_ptr(), v16.data_ptr(), vt.data_ptr(),
                            dt, *st, stream()), "qkv16")
    qk2, vt2 = torch.full((M, 2 * C), 7.0, dtype=tdt, device=dev()), torch.full((B, H, 64, Tp), 7.0, dtype=tdt, device=dev())
    Wv, bv, cv = W[2 * C:].contiguous(), b[2 * C:].contiguous(), colsum[2 * C:].contiguous()
    _lib.check(L.mhmr_gemm16_ln(A.data_ptr(), C, W.data_ptr(), C, M, 2 * C, C, None if fold else b.data_ptr(), None, qk2.data_ptr(), 2 * C, Tp, H,
                                _lib.EPI_OP16_QK, dt, 0, 0, 0, None, None, *(st if fold else (None, None, None)), stream()), "qk")
    _lib.check(L.mhmr_gemm16_ln(A.data_ptr(), C, Wv.data_ptr(), C, M, C, C, None if fold else bv.data_ptr(), None, vt2.data_ptr(), 0, Tp, H,
                                _lib.EPI_VT, dt, 0, 0, 0, None, None, *((rs.data_ptr(), cv.data_ptr(), bv.data_ptr()) if fold else (None, None, None)),
                                stream()), "vt")
    # Q | K: the same epilogue in both forms -> bit-equal.  V: the merged launch computes it in the row-major orientation (weight = first
    # MFMA operand, row statistics on the second operand's side), the V^T launch the other way round: without the fold the same fp32
    # values; with it the two epilogues associate  rstd * acc + (-mean * rstd) * colsum + b'  differently -> the last fp32 bit, i.e. an
    # occasional 16-bit ulp
    assert torch.equal(qk, qk2)
    assert torch.equal(vt, vt2) if not fold else (rel(vt, vt2) < 2e-5 and maxrel(vt, vt2) < (1e-3 if name == "f16" else 8e-3))
    lin = A.double() @ W.double().T
    want = (rs[:, 1:2].double() * (lin - rs[:, 0:1].double() * colsum.double()) + b.double()) if fold else lin + b.double()
    want[:, :C] *= _lib.ATTN_QSCALE
    assert maxrel(qk, want[:, :2 * C]) < tol
    perm = swap23(torch.arange(Tp, device=dev()))
    got_v = vt.float()[..., perm].permute(0, 3, 1, 2).reshape(M, C)
    assert maxrel(got_v, want[:, 2 * C:]) < tol
