"""-m "not gpu": host-side logic of the product -- the C-ABI library loads and exports every declared symbol, the
load-time repacking is exact, the nn.Module surface matches the reference's, and nothing computes on the CPU."""
import os
import re

import numpy as np
import pytest
import torch

from multi_hmr_amd import Model, _lib, packing
import synthetic

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_builds_loads_and_exports_every_declared_symbol():
    _lib.build()
    lib = _lib.lib()
    assert lib.mhmr_version() == _lib.VERSION == 106
    header = open(os.path.join(ROOT, "include", "mhmr.h")).read()
    declared = set(re.findall(r"\b(?:int|long long|const char\*)\s+(mhmr_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    for sym in declared:
        assert getattr(lib, sym) is not None
    # argument validation happens before any launch, so it is testable without a GPU
    assert lib.mhmr_prof_enable(99) == -1
    # the library says which sources it was built from, and build() trusts that, not file times
    assert lib.mhmr_source_hash().decode() == _lib.source_hash() == _lib.built_source_hash()


def test_pos_embed_bicubic_matches_torch_scale_factor_form():
    from oracle.dinov2_ref import interpolate_pos_embed
    g = torch.Generator().manual_seed(0)
    pe = torch.randn(1, 1 + 37 * 37, 48, generator=g) * 0.02
    for G in (16, 32, 48, 64, 92, 37):
        ref = interpolate_pos_embed(pe, G)[0].numpy()
        got = packing.interpolate_pos_embed(pe.numpy(), G)
        assert got.shape == (1 + G * G, 48)
        assert np.abs(got - ref).max() < 1e-6, G      # torch evaluates the source coordinate in fp32


def test_pack_smplx_is_an_exact_refactoring_of_smplx_lbs(smplx_data):
    """J0 + JS.coef == J_regressor.(v_template + blendshapes) and F.D == v_template + shape blend + pose blend."""
    from oracle import smplx_ref
    pk = packing.pack_smplx(smplx_data, 10, "cpu")
    bm = smplx_ref.SMPLX(smplx_data, num_betas=10)
    g = torch.Generator().manual_seed(0)
    coef = torch.randn(20, generator=g)
    v_shaped = bm.v_template + torch.einsum("l,mkl->mk", coef, torch.cat([bm.shapedirs, bm.expr_dirs], -1))
    J = bm.J_regressor @ v_shaped
    J2 = pk["J0"] + (pk["JS"] @ coef).reshape(55, 3)
    assert float((J - J2).abs().max()) < 2e-6
    pf = 0.1 * torch.randn(486, generator=g)
    F = torch.zeros(pk["Kb"], dtype=torch.float64)
    F[:486], F[486:506] = pf.double(), coef.double()
    # per tile: [Kb/8 - 8][axis][48][8] high halves of k < 448, then [8][hi|lo][axis][48][8] of the last 64 k; scaled by 2^10
    b16 = pk["basis16"].double()
    nt, kp = pk["Vp"] // 48, pk["Kb"] // 8 - 8
    assert b16.shape == (nt, kp * 3 * 48 * 8 + 8 * 2 * 3 * 48 * 8) and pk["basis16"].dtype == torch.float16
    head = b16[:, : kp * 1152].reshape(nt, kp, 3, 48, 8)
    tail = b16[:, kp * 1152:].reshape(nt, 8, 2, 3, 48, 8)
    Dt = torch.cat([head, tail[:, :, 0] + tail[:, :, 1]], dim=1)                               # [tile, k block, axis, 48, 8]
    D = Dt.permute(1, 4, 2, 0, 3).reshape(pk["Kb"], 3, pk["Vp"]) / 1024.0                       # [k, axis, v]
    v_posed = (torch.einsum("k,kav->va", F, D) + pk["vtemp"].double().T)[: pk["V"]].float()
    ref = v_shaped + (pf @ bm.posedirs).view(-1, 3)
    # the pose correctives of k < 448 carry the f16 rounding of the basis (2^-12 relative per term: micrometres) ...
    assert float((v_posed - ref).abs().max()) < 1e-5
    # ... the shape / expression directions and the last pose columns are exact to fp32 (the pair form)
    F2 = F.clone(); F2[:448] = 0
    ref2 = torch.einsum("l,mkl->mk", coef, torch.cat([bm.shapedirs, bm.expr_dirs], -1)) + (pf[448:] @ bm.posedirs[448:]).view(-1, 3)
    assert float((torch.einsum("k,kav->va", F2, D)[: pk["V"]].float() - ref2).abs().max()) < 2e-6
    assert float((tail[:, :, 1]).abs().max()) > 0                                               # hi alone would not do there
    assert pk["Kinf"] == 4 and pk["Vp"] % 48 == 0 and pk["Kb"] % 32 == 0
    W = torch.zeros(pk["V"], 55).scatter_add_(1, pk["skin_idx"].long(), pk["skin_w"])
    assert torch.allclose(W, bm.lbs_weights, atol=0)
    s16 = pk["skin16"].double()                                                   # dense [Vp/48, 8, hi|lo, 48, 8] = w[v][j], j = 8 * block + e
    Wd = (s16[:, :, 0] + s16[:, :, 1]).permute(0, 2, 1, 3).reshape(pk["Vp"], 64)
    assert float((Wd[: pk["V"], :55] - bm.lbs_weights.double()).abs().max()) < 1e-7 and float(Wd[:, 55:].abs().max()) == 0
    assert (pk["lmk_vidx"].numpy() == bm.faces[bm.lmk_faces_idx.numpy()]).all()


def test_model_surface_matches_reference(smplx_data, mean_params):
    m = Model(backbone="dinov2_vits14", img_size=448, smplx_data=smplx_data, mean_params=mean_params, backbone_depth=2,
              some_training_flag=123, train_return_type="foo")          # arbitrary extra kwargs are swallowed (model.py:48-49)
    sd = synthetic.make_state_dict("dinov2_vits14", 448, depth_override=2, mean_params=mean_params)
    assert set(m.state_dict().keys()) == set(sd.keys())
    for k, v in m.state_dict().items():
        assert tuple(v.shape) == tuple(sd[k].shape), k
    assert m.load_state_dict(sd, strict=False).missing_keys == []
    assert m.img_size == 448 and m.patch_size == 14 and m.nearness is True and m.embed_dim == 384
    assert m.smpl_layer["neutral_10"].bm_x.faces.shape == (20908, 3)            # demo.py:310
    assert m.smpl_layer["neutral_10"].person_center_idx == 15
    assert m.x_attention_head.init_body_pose.shape == (1, 318)
    # Appendix C shapes (ViT-S: Cc = 483, token_dim = 814)
    assert m.state_dict()["x_attention_head.transformer.to_token_embedding.weight"].shape == (1024, 814)
    assert m.state_dict()["x_attention_head.cross_queries_x"].shape == (32, 483)
    assert m.state_dict()["x_attention_head.transformer.transformer.layers.1.1.fn.to_kv.weight"].shape == (512, 483)


def test_camera_embedding_constructor_contract(smplx_data, mean_params):
    """Reference model.py:67-83: camera_embedding=None is accepted by the constructor (camera_embed_dim = 0, context = features only) but
    such a model cannot run forward (model.py:262 calls embedd_camera unconditionally -> AttributeError on self.camera); anything but
    'geometric' / None raises NotImplementedError; camera_embedding_max_resolution only sets the frequency table."""
    kw = dict(backbone="dinov2_vits14", img_size=224, smplx_data=smplx_data, mean_params=mean_params, backbone_depth=1)
    m = Model(camera_embedding=None, **kw)
    assert m.camera_embed_dim == 0 and m.state_dict()["x_attention_head.cross_queries_x"].shape == (16, 384)
    assert m.state_dict()["x_attention_head.transformer.to_token_embedding.weight"].shape == (1024, 318 + 10 + 3 + 384)
    with pytest.raises(AttributeError):
        m(torch.zeros(1, 3, 224, 224), K=synthetic.get_camera_K(224))
    with pytest.raises(NotImplementedError):
        Model(camera_embedding="learned", **kw)
    assert Model(camera_embedding_max_resolution=128, **kw).camera_embed_dim == 99


def test_no_cpu_fallback(smplx_data, mean_params):
    m = Model(backbone="dinov2_vits14", img_size=224, smplx_data=smplx_data, mean_params=mean_params, backbone_depth=1)
    with pytest.raises(_lib.MhmrError):
        m(torch.zeros(1, 3, 224, 224), K=synthetic.get_camera_K(224))
    # the product never imports the oracle
    for f in os.listdir(os.path.join(ROOT, "multi_hmr_amd")):
        if f.endswith(".py"):
            assert "oracle" not in open(os.path.join(ROOT, "multi_hmr_amd", f)).read().replace("the oracle", "").replace("CPU oracle", ""), f


GELU_P = (-1.000037670135498, -1.1507878303527832, -0.4599924385547638, -0.051827382296323776, 0.007084557320922613, -0.0004733092791866511)


def test_gelu_exp2_polynomial_error_bound():
    """The constants of csrc/mhmr_common.h gelu_fast -- x Phi(x) = max(x, 0) - |x| exp2(P(|x|)), P of degree 5 fitted to the log2 of
    the normal law's upper tail (tools/gelu_fit.py) -- restated: the form is within 4.5e-7 absolute of x Phi(x) in float64 and within
    8e-7 with every step rounded to float32 (the GPU sweep of the epilogue is tests/test_gpu_kernels.py), the tail term dies for
    any large |x| (negative leading coefficient), and the file holds exactly these constants."""
    import math
    from scipy.special import erfc
    x = np.linspace(-12, 12, 400001)
    ax = np.abs(x)
    ref = 0.5 * x * erfc(-x / math.sqrt(2))
    p = np.polyval(GELU_P[::-1], ax)
    assert np.abs(np.maximum(x, 0) - ax * np.exp2(p) - ref).max() < 4.5e-7
    x32 = x.astype(np.float32)
    a32 = np.abs(x32)
    q = np.full_like(a32, np.float32(GELU_P[5]))
    for c in GELU_P[4::-1]:
        q = (q.astype(np.float64) * a32 + np.float32(c)).astype(np.float32)          # one fma, rounded once
    e = np.exp2(q.astype(np.float64)).astype(np.float32)
    g = (np.maximum(x32, 0).astype(np.float64) - a32.astype(np.float64) * e).astype(np.float32)
    assert np.abs(g - 0.5 * x32.astype(np.float64) * erfc(-x32.astype(np.float64) / math.sqrt(2))).max() < 8e-7
    big = np.array([15.0, 40.0, 1e3, 1e6, 3e38])
    with np.errstate(over="ignore"):
        assert np.all(np.polyval(GELU_P[::-1], big) < -250)
    src = open(os.path.join(ROOT, "multi_hmr_amd", "csrc", "mhmr_common.h")).read()
    body = src[src.index("float gelu_fast(float x)"):]
    body = body[:body.index("}")]
    for c in GELU_P:
        assert repr(abs(c)) + "f" in body, c


def test_no_dot_product_hides_inside_inline_assembly():
    """gfx90a+: a VALU instruction that is not the same dot opcode needs three wait states before it reads a dot product's result, and
    hipcc's hazard recogniser does not look inside asm statements (round 6: a kernel with `asm("v_dot2_f32_f16 ...")` read stale sums;
    DESIGN 12.8).  Dot products go through __builtin_amdgcn_fdot2 / fdot2_f32_bf16; no csrc file may bring the asm form back."""
    csrc = os.path.join(ROOT, "multi_hmr_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".hip", ".h")):
            text = re.sub(r"//[^\n]*", "", open(os.path.join(csrc, f)).read())          # (comments may quote the old form)
            for m in re.finditer(r'asm\s*(volatile)?\s*\(\s*"([^"]*)"', text):
                assert "v_dot" not in m.group(2), (f, m.group(2))


def test_pack_smplx_rejects_a_basis_outside_the_f16_pair_range(smplx_data):
    """The blend basis travels as an f16 pair scaled by 2^10: a body model whose blend shapes are 100x larger would overflow
    silently on the GPU; pack_smplx refuses it."""
    import pytest
    bad = dict(smplx_data)
    bad["posedirs"] = np.asarray(smplx_data["posedirs"]) * 1e4
    with pytest.raises(ValueError):
        packing.pack_smplx(bad, 10, "cpu")
    packing.pack_smplx(smplx_data, 10, "cpu")


def test_token_rows_per_image_rule(monkeypatch):
    """vit.padded_tokens: 64-row padding while every encoder linear stays on the 256x256 kernel -- the block GEMMs cover the patch rows
    only (N % 256 == 0: 672^2, 896^2), or C and B * Tp are multiples of 256; otherwise the 128-row tile of the 128x128 kernel."""
    from multi_hmr_amd import vit
    L = {"C": 1024, "T": 4097}
    assert all(vit.padded_tokens(L, b) == 4160 for b in (32, 4, 1, 6)) and vit.row_map(L, 32)
    assert vit.padded_tokens({"C": 1024, "T": 2305}, 32) == 2368 and vit.padded_tokens({"C": 1024, "T": 2305}, 1) == 2368
    assert vit.padded_tokens({"C": 1024, "T": 8465}, 8) == 8512 and not vit.row_map({"C": 1024, "T": 8465}, 8)      # 1288^2: N = 8464
    assert vit.padded_tokens({"C": 1024, "T": 8465}, 1) == 8576
    # ... and with folded LayerNorms but no row map (N = 8464 is not a multiple of 256): rows per image padded to whole 256-row tiles,
    # whatever the batch size (the fold needs every block linear on the 256x256 kernel over all B * Tp rows)
    assert all(vit.padded_tokens({"C": 1024, "T": 8465, "fold": True}, b) == 8704 for b in (1, 3, 8))
    assert vit.padded_tokens({"C": 1024, "T": 4097, "fold": True}, 32) == 4160 and vit.fold_eligible(1024, 8464) and not vit.fold_eligible(384, 2304)
    # round 6: ViT-S (C = 384) CAN fold too -- its C-wide linears as N = 512 with masked columns, always over ALL rows (whole 256-row tiles,
    # no token-row map) -- but that form measured slower at config 2's batch, so it is opt-in (MHMR_VITS_256=1)
    S6 = {"C": 384, "T": 2305, "fold": True, "cpad": 512}
    assert [vit.padded_tokens(S6, b) for b in (1, 8, 16)] == [2560, 2560, 2560] and not vit.row_map(S6, 8)
    assert vit.default_wlo(384) == "" and vit.default_wlo(768) == vit.default_wlo(1024) == vit.DEFAULT_WLO == "proj@0-11"
    monkeypatch.setenv("MHMR_VITS_256", "1")
    assert vit.fold_eligible(384, 2304) and vit.fold_eligible(768, 2304)
    monkeypatch.delenv("MHMR_VITS_256")
    # tiny batches (the narrowest linear at most half a round of 256x256 tiles) with folded LayerNorms: all rows through the big GEMMs, no
    # class-row kernels -- 896^2: one image; 672^2: up to three
    F = {"C": 1024, "T": 4097, "fold": True}
    assert vit.padded_tokens(F, 1) == 4352 and not vit.row_map(F, 1) and vit.padded_tokens(F, 2) == 4160 and vit.row_map(F, 2)
    F6 = {"C": 1024, "T": 2305, "fold": True}
    assert [vit.padded_tokens(F6, b) for b in (1, 2, 3, 4, 32)] == [2560, 2560, 2560, 2368, 2368] and vit.row_map(F6, 4) and not vit.row_map(F6, 3)
    assert vit.padded_tokens({"C": 768, "T": 2305, "fold": True}, 1) == 2560 and vit.padded_tokens({"C": 384, "T": 2305}, 1) == 2432
    assert vit.padded_tokens({"C": 384, "T": 2305}, 16) == 2432           # ViT-S without the fold: proj / fc2 / V run on the 128x128 kernel
    assert vit.padded_tokens({"C": 768, "T": 2305}, 16) == 2368           # ViT-B: 256-multiples throughout
    assert vit.padded_tokens({"C": 1024, "T": 257}, 8) == 320 and vit.padded_tokens({"C": 1024, "T": 257}, 2) == 320 and vit.row_map({"C": 1024, "T": 257}, 2)
    assert not vit.row_map(L, 512)                                        # 32-bit residual offsets of the 256x256 kernel
    monkeypatch.setenv("MHMR_ROWMAP", "0")                                # the A/B switch of csrc/capi.hip
    assert vit.padded_tokens(L, 32) == 4160 and vit.padded_tokens(L, 1) == 4224 and vit.padded_tokens(L, 6) == 4224
    assert vit.padded_tokens({"C": 1024, "T": 257}, 2) == 384
    # beyond the 32-bit residual offsets of the 256x256 kernel the residual GEMMs fall back to the 128x128 kernel: its row tile decides
    monkeypatch.delenv("MHMR_ROWMAP")
    assert vit.padded_tokens(L, 512) == 4224 and vit.padded_tokens({"C": 1024, "T": 8465}, 128) == 8576
    monkeypatch.setenv("MHMR_GEMM128", "1")                               # the switch that forces the 128x128 kernel everywhere
    assert vit.padded_tokens(L, 32) == 4224 and not vit.row_map(L, 32) and not vit.fold_eligible(1024, 4096)


def test_low_half_weight_split_and_spec_parser():
    """vit.hi_lo: [W_hi | W_lo] reproduces the fp32 weight to 2^-22 relative (f16) where one rounding leaves 2^-11; vit.parse_wlo."""
    import torch
    from multi_hmr_amd import vit
    torch.manual_seed(3)
    w = torch.randn(64, 128) / 11.0
    wd = w.double()
    for tdt, bits, floor in ((torch.float16, 11, 2.0 ** -25), (torch.bfloat16, 8, 0.0)):      # (f16: the subnormal step where W_lo underflows)
        p = vit.hi_lo(w, tdt)
        assert p.shape == (64, 256) and p.dtype == tdt
        hi, lo = p[:, :128].double(), p[:, 128:].double()
        assert bool(((hi - wd).abs() <= 2.0 ** -bits * wd.abs() + floor).all()) and float((hi - wd).abs().max()) > 2.0 ** -(bits + 3) * float(wd.abs().max())
        assert bool(((hi + lo - wd).abs() <= 2.0 ** -(2 * bits) * wd.abs() + floor).all())
    assert vit.parse_wlo("v+proj@0-11", 24) == {i: {"v", "proj"} for i in range(12)}
    assert vit.parse_wlo("v+proj@0-11", 4) == {i: {"v", "proj"} for i in range(4)}          # clipped to the depth
    assert vit.parse_wlo("proj@2,v@2-3", 24) == {2: {"proj", "v"}, 3: {"v"}}
    assert vit.parse_wlo("", 24) == {} and vit.parse_wlo(None, 24) == {}
    assert vit.parse_wlo("v", 3) == {0: {"v"}, 1: {"v"}, 2: {"v"}}
    with pytest.raises(ValueError):
        vit.parse_wlo("fc1@0-3", 24)


def test_layernorm_fold_packing_is_the_same_linear_map(monkeypatch):
    """vit.pack_encoder(lnfold=True) on the CPU: for a folded linear, rstd (x . W'^T - mean colsum) + b' computed in fp64 from the PACKED
    tensors equals Linear(LayerNorm(x)) up to the 16-bit rounding of W' (reference blocks/dinov2.py -> hub Block: norm1 -> attn.qkv,
    norm2 -> mlp.fc1); block 0's norm1 is not folded; the V rows with a low half carry [W'_hi | W'_lo] and their colsum counts both."""
    import ctypes as C
    import torch
    from oracle import dinov2_ref
    from multi_hmr_amd import vit
    torch.manual_seed(5)
    monkeypatch.setenv("MHMR_LO8", "1")                              # also pack the fp8 form of the low halves (opt-in: vit.lo8_eligible)
    enc = dinov2_ref.DinoVisionTransformer(embed_dim=256, depth=3, num_heads=4).double()
    for b in enc.blocks:                                             # non-trivial LayerNorm parameters
        for n in (b.norm1, b.norm2):
            n.weight.data = 1.0 + 0.3 * torch.randn_like(n.weight)
            n.bias.data = 0.2 * torch.randn_like(n.bias)
    P = vit.pack_encoder(enc.float(), 224, "f16", torch.device("cpu"), wlo="v@1", lnfold=True)
    blocks = P["vit"]["blocks"]
    assert [blocks[i].flags for i in range(3)] == [2, 3, 3]
    torch.set_grad_enabled(False)
    by_ptr = {t.data_ptr(): t for t in P["keep"]}
    x = torch.randn(7, 256, dtype=torch.float64) * 2 + 0.3
    mean, var = x.mean(1, keepdim=True), x.var(1, unbiased=False, keepdim=True)
    rstd = (var + 1e-6).rsqrt()
    for i in (1, 2):
        b = enc.blocks[i].double()
        for w_ptr, b_ptr, cs_ptr, lin, norm in ((blocks[i].qkv_w, blocks[i].qkv_b, blocks[i].qkv_colsum, b.attn.qkv, b.norm1),
                                                (blocks[i].fc1_w, blocks[i].fc1_b, blocks[i].fc1_colsum, b.mlp.fc1, b.norm2)):
            W, bias, cs = by_ptr[w_ptr].double(), by_ptr[b_ptr].double(), by_ptr[cs_ptr].double()
            want = lin(norm(x))
            got = rstd * (x @ W.T - mean * cs[None, :]) + bias[None, :]
            rows = slice(0, 512) if lin is b.attn.qkv and i == 1 else slice(None)      # block 1's V rows live in v_w2 (low-half pass)
            err = float((got[:, rows] - want[:, rows]).abs().max() / want.abs().max())
            assert err < 2e-3, (i, err)                                               # one f16 rounding of W'
    v2, cs = by_ptr[blocks[1].v_w2].double(), by_ptr[blocks[1].qkv_colsum].double()[512:]
    b = enc.blocks[1].double()
    want = b.attn.qkv(b.norm1(x))[:, 512:]
    assert v2.shape == (256, 512) and P["lo8"] and blocks[1].v_w8 and not blocks[1].proj_w8 and not blocks[0].v_w8
    # embed_dim 256 is eligible for the fp8 low-half range: the big GEMM multiplies [W'_hi | e4m3(W'_lo 2^-e)] and the column sums count
    # exactly those values; the 16-bit pair (what the class-row kernel multiplies) sums to the same within the e4m3 rounding of W'_lo
    rows8 = by_ptr[blocks[1].v_w8]
    assert rows8.dtype == torch.uint8 and rows8.shape == (256, 3 * 256)
    hi = rows8[:, :512].contiguous().view(torch.float16).double()
    lo = rows8[:, 512:].contiguous().view(torch.float8_e4m3fn).double() * 2.0 ** (blocks[1].v_w8_scale - 127)
    assert torch.equal(hi, v2[:, :256]) and torch.allclose(cs, (hi + lo).sum(1), atol=1e-6) and torch.allclose(cs, v2.sum(1), atol=2e-5)
    assert float((lo - v2[:, 256:]).abs().max()) <= 2.0 ** -4 * float(v2[:, 256:].abs().max()) + 1e-12      # three significant bits of W'_lo
    got8 = rstd * (x @ (hi + lo).T - mean * cs[None, :]) + by_ptr[blocks[1].qkv_b].double()[None, 512:]
    assert float((got8 - want).abs().max() / want.abs().max()) < 3e-5                  # hi + e4m3 lo: ~15 bits of the weight
    got = rstd * (torch.cat([x, x], 1) @ v2.T - mean * v2.sum(1)[None, :]) + by_ptr[blocks[1].qkv_b].double()[None, 512:]
    assert float((got - want).abs().max() / want.abs().max()) < 2e-6                   # hi + lo: 22 bits
    torch.set_grad_enabled(True)


def test_extra_joint_tiles_of_the_packed_body_model(smplx_data):
    """packing.pack_smplx: the 72 extra joints as virtual vertices behind the real ones -- extra joint e = 16 t + i owns column i of the
    three 16-vertex blocks of tile t, as copies of its corner vertices' columns; picked vertices get the weights (1, 0, 0)."""
    from multi_hmr_amd import packing, constants
    pk = packing.pack_smplx(smplx_data, 10, "cpu")
    V, Vl, Vp = pk["V"], pk["Vl"], pk["Vp"]
    assert Vl == packing.roundup(V, 48) and Vp == Vl + 5 * 48 and pk["basis16"].shape[0] == Vp // 48
    b16, s16, vt = pk["basis16"].numpy(), pk["skin16"].numpy(), pk["vtemp"].numpy()
    kp = pk["Kb"] // 8 - 8

    def col(v):          # every basis value of vertex column v: the high-half part and the pair part of its tile
        t = b16[v // 48]
        return np.concatenate([t[: kp * 1152].reshape(kp, 3, 48, 8)[:, :, v % 48].ravel(), t[kp * 1152:].reshape(8, 2, 3, 48, 8)[:, :, :, v % 48].ravel()])
    faces = np.asarray(smplx_data["f"]).astype(np.int64)
    corners = np.concatenate([np.repeat(np.asarray(constants.SMPLX_EXTRA_JOINT_VERTS)[:, None], 3, 1),
                              faces[np.asarray(smplx_data["lmk_faces_idx"]).astype(np.int64)]], 0)
    assert corners.shape == (72, 3)
    for e in range(72):
        for k in range(3):
            vv, src = Vl + 48 * (e // 16) + 16 * k + e % 16, int(corners[e, k])
            assert np.array_equal(col(vv), col(src))
            assert np.array_equal(s16[vv // 48][..., vv % 48, :], s16[src // 48][..., src % 48, :]) and np.array_equal(vt[:, vv], vt[:, src])
    xb = pk["xbary"].numpy()
    assert np.array_equal(xb[:21], np.tile([[1.0, 0.0, 0.0]], (21, 1))) and np.allclose(xb[21:], np.asarray(smplx_data["lmk_bary_coords"]))
    unused = [Vl + 48 * 4 + 16 * k + i for k in range(3) for i in range(8, 16)]           # slots 72..79 of the fifth tile
    assert all(not col(v).any() for v in unused)


def test_bench_starts_its_own_ranks_and_refuses_more_gpus_than_the_node_has():
    """`python bench.py --gpus N` from a bare shell (no WORLD_SIZE): the script launches N ranks itself; more ranks than visible GPUs ->
    ONE JSON error line and a non-zero exit code (here: no GPU at all)."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 2, (r.returncode, r.stderr[-500:])
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert "error" in d and d["n_gpus"] == 2 and d["gpus_visible"] == torch.cuda.device_count()


def test_image_block_rule_of_the_backbone(smplx_data, mean_params):
    """Model._nsplit: two image blocks when the batch is even, >= 8 images and the narrowest block linear is under six
    rounds of 256 tiles -- BASELINE configs 2, 3, 5 -- one block for the headline (8 exact rounds) and for small batches; `split=` / MHMR_SPLIT
    override it, and a count that does not divide the batch falls back to the next one that does."""
    mk = lambda bb, S, **kw: Model(backbone=bb, img_size=S, smplx_data=smplx_data, mean_params=mean_params, backbone_depth=1, **kw)
    assert mk("dinov2_vitl14", 896)._nsplit(32) == 1            # config 4, the bench default
    assert mk("dinov2_vitl14", 672)._nsplit(32) == 2            # config 3
    assert mk("dinov2_vitl14", 1288)._nsplit(8) == 2            # config 5
    assert mk("dinov2_vits14", 672)._nsplit(16) == 2            # config 2
    m = mk("dinov2_vitl14", 224)
    assert m._nsplit(2) == 1 and m._nsplit(7) == 1 and m._nsplit(8) == 2
    assert mk("dinov2_vitl14", 224, split=1)._nsplit(8) == 1 and mk("dinov2_vitl14", 224, split=4)._nsplit(8) == 4
    assert mk("dinov2_vitl14", 224, split=4)._nsplit(6) == 3 and mk("dinov2_vitl14", 224, split=2)._nsplit(5) == 1


def test_logit_gain_statistic_and_the_auto_precision_rule():
    """vit.logit_gain: the spread of a block's pre-softmax logits over the keys of one query, predicted from the weights alone --
    checked here against the EMPIRICAL spread of the same block on unit-variance LayerNorm inputs; and the rule built on it."""
    from multi_hmr_amd import vit
    from multi_hmr_amd.model import Dinov2Backbone
    for hostile in (False, True):
        sd = synthetic.make_state_dict("dinov2_vits14", 224, seed=5, depth_override=2)
        if hostile:
            synthetic.make_hostile(sd, "weights", seed=5)
        bb = Dinov2Backbone("dinov2_vits14", pretrained=False, depth_override=2)
        bb.load_state_dict({k[len("backbone."):]: v for k, v in sd.items() if k.startswith("backbone.")}, strict=True)
        enc = bb.encoder
        gains = vit.logit_gain(enc)
        assert len(gains) == 2
        g = torch.Generator().manual_seed(0)
        blk, C, H = enc.blocks[0], 384, 6
        xh = torch.randn(4096, C, generator=g, dtype=torch.float64)                        # the LayerNorm's normalised rows
        y = xh * blk.norm1.weight.double() + blk.norm1.bias.double()
        qkv = y @ blk.attn.qkv.weight.double().T + blk.attn.qkv.bias.double()
        q, k = qkv[:64, :C].view(64, H, 64), qkv[:, C:2 * C].view(-1, H, 64)
        logits = torch.einsum("qhd,khd->hqk", q, k) * 0.125
        emp = float(logits.detach().var(dim=-1).mean(dim=-1).mean().sqrt())                          # rms over heads and queries of the per-row std
        assert abs(gains[0] - emp) / emp < 0.12, (hostile, gains[0], emp)
        assert (gains[0] > vit.LOGIT_GAIN_LIMIT) == hostile, gains
        assert vit.resolve_precision(enc, "auto") == ("f16x3" if hostile else "f16")
        assert vit.resolve_precision(enc, "bf16") == "bf16" and vit.resolve_precision(enc, "fp16") == "f16"
        # the limit follows the tokens per image above 896^2 (measured: plain f16 leaves the contract earlier on longer key sets)
        assert vit.logit_gain_limit(None) == vit.logit_gain_limit(1025) == vit.logit_gain_limit(4097) == vit.LOGIT_GAIN_LIMIT == 4.0
        assert 3.3 < vit.logit_gain_limit(8465) < 3.4 and vit.logit_gain_limit(10 ** 6) == 3.0
        assert vit.resolve_precision(enc, "auto", tokens=8465) == ("f16x3" if hostile else "f16")
        # 'auto' choosing the slow mode FOR the caller is said out loud, once per pack, with the statistic and the way to force f16;
        # an explicit choice and the fast resolution are silent
        import warnings
        with warnings.catch_warnings(record=True) as rec:
            warnings.simplefilter("always")
            P = vit.pack_encoder(enc, 224, "auto", "cpu")
            vit.pack_encoder(enc, 224, "f16x3" if hostile else "f16", "cpu")
        said = [str(w.message) for w in rec if issubclass(w.category, RuntimeWarning) and "f16x3" in str(w.message)]
        assert len(said) == (1 if hostile else 0), said
        assert P["precision"] == ("f16x3" if hostile else "f16")
        if hostile:
            assert "precision='f16'" in said[0] and f"{max(gains):.2f}" in said[0]
    with pytest.raises(ValueError):
        vit.resolve_precision(enc, "fp8")


def test_three_product_operands_reconstruct_the_fp32_product():
    """vit.triple against an activation pair: hi.hi + hi.lo + lo.hi (what the k ranges of GemmArgs::a_k with K = 3 a_k pair up) is the
    fp32 product to ~2^-21; and the f16x3 pack lays every backbone linear out that way."""
    from multi_hmr_amd import vit
    g = torch.Generator().manual_seed(0)
    W, A = torch.randn(64, 128, generator=g) * 0.03, torch.randn(32, 128, generator=g) * 5
    W3 = vit.triple(W, torch.float16)
    ah = A.half()
    A2 = torch.cat([ah, (A - ah.float()).half()], 1)
    K = 128
    Amap = torch.cat([A2[:, :K], A2[:, :K], A2[:, K:]], 1)                                  # the kernel's k-tile map: hi, hi, lo
    got = Amap.double() @ W3.double().T
    exact = A.double() @ W.double().T
    one = ah.double() @ W.half().double().T
    e3 = float((got - exact).norm() / exact.norm())
    e1 = float((one - exact).norm() / exact.norm())
    assert e3 < 1e-6 and e1 > 2e-4 and W3.shape == (64, 384)
    from multi_hmr_amd.model import Dinov2Backbone
    bb = Dinov2Backbone("dinov2_vits14", pretrained=False, depth_override=1)
    P = vit.pack_encoder(bb.encoder, 224, "f16x3", "cpu")
    assert P["x3"] and P["precision"] == "f16x3" and not P["fold"] and P["wlo"] == {}
    assert vit.padded_tokens(P, 3) == 384 and not vit.row_map(P, 4)                         # T = 257 -> whole 128-row tiles (C = 384)
    shapes = sorted(tuple(t.shape) for t in P["keep"] if t.dtype == torch.float16)
    assert (384, 3 * 640) in shapes and (3 * 384, 3 * 384) in shapes and (384, 12 * 384) in shapes and (4 * 384, 3 * 384) in shapes


def test_output_block_layout_and_the_graph_wrapper_without_a_gpu(smplx_data, mean_params):
    """Model._alloc_outputs carves every head output out of one allocation (256-byte aligned, contiguous views, rebuildable on a copy of
    the block: what graphed.GraphedForward hands out); GraphedForward itself refuses to exist without the GPU (no CPU path)."""
    from multi_hmr_amd import GraphedForward
    m = Model(backbone="dinov2_vits14", img_size=224, smplx_data=smplx_data, mean_params=mean_params, backbone_depth=1)
    P = {"hph": {"nb": 10}, "lbs": {"V": 10475}}
    cpu = torch.device("cpu")
    o = m._alloc_outputs(P, 5, cpu)
    flat = o["_flat"]
    names = [n for n, _ in Model.OUTPUT_SHAPES]
    assert set(o) == set(names) | {"transl_pelvis", "_flat"} and set(Model.PERSON_KEYS) <= set(o)
    assert o["v3d"].shape == (5, 10475, 3) and o["rotmat"].shape == (5, 53, 3, 3) and o["scores"].shape == (5,) and o["shape"].shape == (5, 10)
    spans = []
    for n in names:
        t = o[n]
        assert t.is_contiguous() and (t.data_ptr() - flat.data_ptr()) % 256 == 0, n
        spans.append((t.data_ptr() - flat.data_ptr(), t.numel() * 4))
    spans.sort()
    assert all(a + la <= b for (a, la), (b, _) in zip(spans, spans[1:])) and spans[-1][0] + spans[-1][1] <= flat.numel() * 4     # disjoint, inside
    assert o["transl_pelvis"].shape == (5, 1, 3) and o["transl_pelvis"].data_ptr() == o["j3d"].data_ptr()
    for i, n in enumerate(names):
        o[n].fill_(float(i))
    o2 = m._alloc_outputs(P, 5, cpu, flat=flat.clone())
    assert all(torch.equal(o2[n], o[n]) and o2[n].data_ptr() != o[n].data_ptr() for n in names)
    with pytest.raises(AssertionError):
        m._alloc_outputs(P, 6, cpu, flat=flat)           # a block of another capacity
    if not torch.cuda.is_available():
        with pytest.raises(_lib.MhmrError):
            GraphedForward(m, batch=1)


def test_pose_level_schedule_of_the_kinematic_tree(smplx_data):
    """packing.pose_level_tasks (mhmr_lbs_consts::pose_tasks): every joint appears exactly once, on the level of its depth, with its parent,
    twelve lanes per joint; a level's joints are in joint order; trees that do not fit the pose kernel's fast path return None."""
    par = np.asarray(smplx_data["kintree_table"])[0].astype(np.int64).copy()
    par[0] = -1
    tasks, nlev = packing.pose_level_tasks(par)
    assert tasks.shape == (16, 256) and tasks.dtype == np.int32 and nlev == 11 and (tasks[nlev:] == -1).all()
    seen = {}
    for L in range(nlev):
        row = tasks[L]
        live = row[row >= 0]
        assert live.size % 12 == 0 and (row[live.size:] == -1).all()
        joints = [int(v) & 0xff for v in live[::12]]
        assert joints == sorted(joints)
        for s, j in enumerate(joints):
            assert (row[12 * s:12 * s + 12] == row[12 * s]).all()
            pa = int(row[12 * s]) >> 8
            assert pa == (0xff if par[j] < 0 else par[j])
            d, a = 0, par[j]
            while a >= 0:
                d, a = d + 1, par[a]
            assert d == L
            seen[j] = seen.get(j, 0) + 1
    assert sorted(seen) == list(range(55)) and set(seen.values()) == {1}
    assert packing.pose_level_tasks(np.arange(-1, 20))[0] is None                    # a chain of 21 joints: 21 levels
    assert packing.pose_level_tasks(np.array([-1] + [0] * 30))[0] is None            # a star: 30 joints on one level
