"""Shared host glue for the DINOv2 ViT backbone kernels: weight packing and per-batch workspaces / descriptors.

Used by ``model.Model`` (reference model.py:30-349) and ``anny_model.Multi_HMR`` (reference multi_hmr_anny/multi_hmr.py); both run
``mhmr_vit_forward`` (include/mhmr.h) on the result.  Nothing here is on the per-image path except ``workspace()``'s dictionary
lookup.
"""
from __future__ import annotations

import ctypes as C
import math

import torch

from . import _lib, packing
from .packing import roundup

PATCH = 14


#: default low-half weight passes (tools/precision_study.py): which projections of which blocks also carry the low
#: half of their weights.  "v+proj@0-11" = the V and the attention output projections of blocks 0..11.  "" = none.
#: Round 6: "proj@0-11" (rounds 3-5: "v+proj@0-11").  Re-measured on the current kernels over the five ViT-B / ViT-L goldens (profiles/
#: r06_wlo_study_gpu.txt): worst key 5.9e-4 with the output projections of blocks 0-11 alone against 6.05e-4 with V and proj, the bench's own
#: parity leg 7.2e-4 against 6.9e-4 -- the V low halves bought nothing measurable any more and cost twelve doubled V launches per forward
#: (-1.3 ms per headline step).  none: 8.3e-4; proj@0-7: 6.4e-4; proj (all 24): 5.4e-4.
DEFAULT_WLO = "proj@0-11"


def default_wlo(embed_dim: int) -> str:
    """The default low-half set by backbone width.  ViT-B / ViT-L: ``DEFAULT_WLO``.  ViT-S (12 blocks of width 384): none -- its goldens sit
    at <= 4.4e-4 with or without the low halves (profiles/r03_wlo_study_gpu.txt: vits_672_full 3.9e-4 without, 4.4e-4 with), so the second
    k range of V and proj in all twelve blocks bought nothing there and cost a sixth of the model's GEMM work (round 6)."""
    return DEFAULT_WLO if embed_dim > 384 else ""


def parse_wlo(spec: str, depth: int) -> dict:
    """'v+proj@0-11,proj@12-15' -> {block index: {'v', 'proj'}}; block ranges are clipped to the encoder's depth."""
    out = {}
    for part in filter(None, (s.strip() for s in (spec or "").split(","))):
        names, _, rng = part.partition("@")
        lo, _, hi = (rng or f"0-{depth - 1}").partition("-")
        names = set(names.split("+"))
        if not names <= {"v", "proj"}:
            raise ValueError(f"low-half weight passes exist for 'v' and 'proj', not {sorted(names)}")
        for i in range(int(lo), min(int(hi or lo), depth - 1) + 1):
            out.setdefault(i, set()).update(names)
    return out


def hi_lo(w: torch.Tensor, tdt) -> torch.Tensor:
    """fp32 [N, K] -> 16-bit [N, 2K] = [W_hi | W_lo]: W_hi = round16(W), W_lo = round16(W - W_hi) (for f16 mostly subnormal: the
    matrix pipe takes them at full precision, tests/test_gpu_kernels.py::test_gemm_low_half_weight_pass)."""
    hi = w.to(tdt)
    lo = (w - hi.float()).to(tdt)
    return torch.cat([hi, lo], dim=1).contiguous()


#: MFMA operand formats of ``Model(precision=...)``; "f16x3" = f16 operand PAIRS, three products per term, fp32 attention;
#: "auto" = "f16" unless the checkpoint's attention logits are too steep for one 16-bit rounding per operand (``logit_gain``)
PRECISIONS = ("f16", "fp16", "bf16", "f16x3", "auto")
#: "auto" switches to "f16x3" when the steepest block's logit spread exceeds this (natural-log units; see ``logit_gain``)
LOGIT_GAIN_LIMIT = 4.0


def logit_gain_limit(tokens: int | None = None) -> float:
    """The spread above which 'auto' packs a checkpoint as f16x3, as a function of the tokens per image (round 6).

    Measured (tools/auto_rule_probe.py; full-depth ViT-L, one image, a ladder between the seeded weights and `hostile_w`; plain f16 against
    the fp32 CPU reference path, worst key): plain f16 leaves 1e-3 at a steepest-block spread of ~6.5 at 448^2 (1 025 tokens; profiles/
    r05_auto_rule_probe.txt), ~5.5 at 896^2 (4 097: 6.9e-4 at 4.5, 1.2e-3 at 6.2) and ~4.7 at 1288^2 (8 465: 9.1e-4 at 4.5, 1.7e-3 at
    6.2; profiles/r06_auto_rule_probe_{896,1288}.txt): longer key sets average MORE rounded terms per softmax row and the error of a
    steep softmax grows with them, roughly as tokens^0.15.  4.0 switches before the contract is lost at every BASELINE size (margin 1.6x /
    1.4x / 1.2x); above 4 097 tokens the limit follows the measured slope down -- 4 (4097 / tokens)^0.25: 3.3 at 1288^2 -- so that the
    margin does not shrink further with resolutions nobody measured.  Never above 4.0, never below 3.0."""
    if not tokens or tokens <= 4097:
        return LOGIT_GAIN_LIMIT
    return max(3.0, LOGIT_GAIN_LIMIT * (4097.0 / float(tokens)) ** 0.25)


def logit_gain(enc) -> list:
    """Per block: the standard deviation, ACROSS THE KEYS of one query, of the pre-softmax attention logits (natural-log units) that the
    block's weights produce from a LayerNorm output with unit-variance, independent channels -- rms over the heads.

    Why this number: a 16-bit rounding of q, k (and of everything upstream of them) perturbs a logit by about 2^-12 times its own size,
    i.e. the softmax weights by a RELATIVE 2^-12 x (logit spread) -- and every later block's logits see the earlier blocks' errors
    through the same gain.  With norm1 = (gamma, beta):  q = Wq (gamma . xhat + beta) + bq ~ N(mu_q, S_q),  mu_q = Wq beta + bq,
    S_q = Wq diag(gamma^2) Wq^T  (k likewise), so for one query the logit over the keys has variance
        ( tr(S_q,h S_k,h) + mu_q,h^T S_k,h mu_q,h ) / 64        per head h (64 = head_dim; the scale 1/8 squared)
    (the term that is constant over the keys shifts every logit of the row and cancels in the softmax).  Seeded DINOv2-style weights
    (synthetic.make_state_dict): 1.7 for ViT-L, 0.6 for ViT-S.  synthetic.make_hostile(kind="weights") -- LayerNorm weights with
    x10 ... x30 channels in front of q / k -- : ~10, where the fp32 network itself turns ONE 2^-12 input rounding into 2.5e-3 on an
    output (tests/golden/vitl_672_hostile_w.npz: sens_*), and a single-rounding f16 pipeline ends 2-10x outside the 1e-3 contract."""
    out = []
    for b in enc.blocks:
        Cd = b.attn.qkv.weight.shape[1]
        W = b.attn.qkv.weight.detach().double().cpu()
        bias = b.attn.qkv.bias.detach().double().cpu() if b.attn.qkv.bias is not None else torch.zeros(3 * Cd, dtype=torch.float64)
        gam, beta = b.norm1.weight.detach().double().cpu(), b.norm1.bias.detach().double().cpu()
        H = Cd // 64
        Wq, Wk = (W[:Cd] * gam[None, :]).view(H, 64, Cd), (W[Cd:2 * Cd] * gam[None, :]).view(H, 64, Cd)
        mu_q = (W[:Cd] @ beta + bias[:Cd]).view(H, 64)
        Sq, Sk = Wq @ Wq.transpose(1, 2), Wk @ Wk.transpose(1, 2)                    # [H, 64, 64]
        var = ((Sq * Sk).sum((1, 2)) + torch.einsum("hi,hij,hj->h", mu_q, Sk, mu_q)) / 64.0
        out.append(float(var.mean().sqrt()))
    return out


def resolve_precision(enc, precision: str, tokens: int | None = None) -> str:
    """'auto' -> 'f16' or 'f16x3' from the weights (``logit_gain``) and the tokens per image (``logit_gain_limit``); anything else is
    returned as given ('fp16' = 'f16')."""
    if precision not in PRECISIONS:
        raise ValueError(f"precision must be one of {list(PRECISIONS)}")
    if precision == "fp16":
        return "f16"
    if precision != "auto":
        return precision
    return "f16x3" if max(logit_gain(enc)) > logit_gain_limit(tokens) else "f16"


def triple(w: torch.Tensor, tdt) -> torch.Tensor:
    """fp32 [N, K] -> 16-bit [N, 3K] = [W_hi | W_lo | W_hi]: the weight operand of a three-product linear (GemmArgs::a_k, K = 3 a_k):
    against the activation pair [A_hi | A_lo] the k ranges pair up as A_hi W_hi + A_hi W_lo + A_lo W_hi."""
    hi = w.to(tdt)
    lo = (w - hi.float()).to(tdt)
    return torch.cat([hi, lo, hi], dim=1).contiguous()


def lo8_rows(w_hi16: torch.Tensor, w32: torch.Tensor):
    """One linear's weight with an fp8 low-half range (csrc/gemm256.hip GemmArgs::lo8): rows of 3K bytes = [W_hi: K 16-bit values |
    e4m3(W_lo * 2^-e): K bytes], W_lo = W - W_hi, and the E8M0 scale byte 127 + e.  e puts the largest |W_lo| at <= 256 (e4m3: three
    significant bits down to 2^-6, finite to 448), i.e. the low half is kept to three bits -- 2^-15 of the weight instead of 2^-12.
    Returns (uint8 [N, 3K], scale byte, the dequantised low half as fp64 [N, K] -- for the folded linears' column sums)."""
    lo = (w32.double() - w_hi16.double())
    amax = float(lo.abs().max())
    e = (math.ceil(math.log2(amax)) - 8) if amax > 0 else -127
    e = max(min(e, 127), -127)
    q = (lo * 2.0 ** (-e)).float().to(torch.float8_e4m3fn)
    rows = torch.cat([w_hi16.contiguous().view(torch.uint8).reshape(w_hi16.shape[0], -1), q.view(torch.uint8)], dim=1).contiguous()
    return rows, 127 + e, q.double() * 2.0 ** e


def lo8_eligible(C: int) -> bool:
    """fp8 low-half ranges need the 256x256 kernel's 128-deep fp8 k tiles in whole pairs: embed_dim a multiple of 256 (ViT-B / ViT-L).
    OFF unless MHMR_LO8=1: built, parity-green (tests/test_gpu_kernels.py::test_gemm_fp8_low_half_range, the full-size goldens) and
    measured SLOWER than the 16-bit low halves on this power-clocked chip -- 135.8 vs 133.6 ms per headline step, interleaved on one box
    (profiles/r05_session_d_fp8_low_half_ab.txt, r05_session_e_lo8_kernel_trace.txt): the fp8 k tiles, nominally twice the rate, took
    as long as the 16-bit tiles they replaced (V 425 vs ~443 us, proj 569 vs ~513 us), and the 3C/2 row pitch the bf8 copies need cost
    every other consumer of those rows (fc1 +28 us, QK +17 us, attention +28 us per launch)."""
    import os
    return C % 256 == 0 and os.environ.get("MHMR_LO8", "0") == "1" and "MHMR_GEMM128" not in os.environ


def fold_eligible(C: int, N: int) -> bool:
    """The LayerNorm fold (csrc/gemm256.hip, GemmArgs::pstats / rowstats) needs every block linear on the 256x256 kernel: embed_dim a
    multiple of 256 (ViT-B / ViT-L), and either the token-row map (N a multiple of 256) or, for any other N (1288^2: 8464), the rows
    of an image padded to a multiple of 256 so that the GEMMs cover whole tiles of ALL rows (padded_tokens).  MHMR_LNFOLD=0 /
    MHMR_ROWMAP=0 switch it off (A/B measurements)."""
    import os
    wide = C % 256 == 0
    # ViT-S (C = 384, round 6): its C-wide linears run as N = 512 with masked columns (mhmr_vit_desc.cpad), always over all rows
    # OFF unless MHMR_VITS_256=1: built, parity-green and measured SLOWER at BASELINE's config 2 (16 images, two blocks of 8: 2523-2763 against
    # 2755-2768 img/s on the 128x128 kernel, one box, interleaved: profiles/r06_session_b.txt): 80 row tiles x 2 column tiles = 160 tiles of
    # 256x256 are 0.6 rounds of the chip, a quarter of them masked, and K = 384 is six k tiles -- the tile is too coarse for this batch
    narrow = C % 128 == 0 and os.environ.get("MHMR_VITS_256", "0") == "1" and os.environ.get("MHMR_LNFOLD_ALLROWS", "1") != "0"
    return (os.environ.get("MHMR_LNFOLD", "1") != "0" and os.environ.get("MHMR_ROWMAP", "1") != "0" and "MHMR_GEMM128" not in os.environ and
            ((wide and (N % 256 == 0 or os.environ.get("MHMR_LNFOLD_ALLROWS", "1") != "0")) or (not wide and narrow)))


def pack_encoder(enc, img_size: int, precision: str, device, wlo: str | None = None, lnfold: bool | None = None) -> dict:
    """DINOv2 encoder parameters (key names of torch.hub dinov2_vit*14) -> device tensors in the kernels' layouts.

    16-bit [N, K] linears (K contiguous = MFMA operand order), fp32 biases / LayerNorm / LayerScale, the pos-embed bicubically
    interpolated to the G x G grid on the host (``packing.interpolate_pos_embed``), the patch-embed weight flattened (c, py, px)
    and zero-padded to Kp = 640.  The returned dict owns every tensor the block descriptors point at (``keep``)."""
    requested = precision
    precision = resolve_precision(enc, precision, tokens=(img_size // PATCH) ** 2 + 1)
    x3 = precision == "f16x3"
    dt_id, tdt = packing.OP_DTYPES["f16" if x3 else precision]
    Cd, H, L = enc.embed_dim, enc.num_heads, len(enc.blocks)
    G = img_size // PATCH
    N, T = G * G, G * G + 1
    f32 = lambda t: t.detach().to(device=device, dtype=torch.float32).contiguous()
    op = lambda t: t.detach().to(device=device, dtype=torch.float32).to(tdt).contiguous()
    keep = []

    def k(t):
        keep.append(t)
        return t.data_ptr()

    P = {"dt_id": dt_id, "tdt": tdt, "C": Cd, "H": H, "L": L, "G": G, "N": N, "T": T, "Kp": 640,
         "S": img_size, "device": device, "precision": precision, "precision_requested": requested, "x3": x3}
    if requested == "auto" or x3:
        P["logit_gain"] = logit_gain(enc)
    if requested == "auto" and x3:
        # the slow mode was chosen FOR the caller: say so once per pack (a checkpoint that trips the rule runs ~4x slower than plain f16
        # and allocates two fp32 workspaces of B x Tp x 3C / 4C; advisor finding of round 5)
        import warnings
        warnings.warn(f"multi_hmr_amd: precision='auto' resolved to 'f16x3' for this checkpoint: steepest attention-logit spread "
                      f"{max(P['logit_gain']):.2f} > {logit_gain_limit(T):.2f} (vit.logit_gain / logit_gain_limit at {T} tokens; per block: min {min(P['logit_gain']):.2f}, "
                      f"mean {sum(P['logit_gain']) / len(P['logit_gain']):.2f}).  f16 operand pairs with three products per term and an fp32 "
                      f"attention keep the 1e-3 contract on such weights at about 4x the time of plain f16; Model(precision='f16') forces "
                      f"the fast path (tools/checkpoint_report.py and tools/parity_table.py say what that costs on these weights).",
                      RuntimeWarning, stacklevel=3)
    pos = torch.from_numpy(packing.interpolate_pos_embed(enc.pos_embed.detach().float().cpu().numpy(), G)).to(device)
    cls_pos0 = f32(enc.cls_token.reshape(-1)) + pos[0]
    pw = torch.zeros(Cd, P["Kp"], dtype=torch.float32, device=device)
    pw[:, :588] = f32(enc.patch_embed.proj.weight).reshape(Cd, 588)
    blocks = (_lib.VitBlock * L)()
    if x3:
        # every linear as three products over operand pairs; fp32 LayerNorm / GELU / attention (csrc/capi.hip vit_forward_x3)
        tr = lambda t: triple(f32(t), tdt)
        P["wlo"], P["fold"] = {}, False
        for i, b in enumerate(enc.blocks):
            blk = blocks[i]
            blk.flags, blk.v_w2, blk.proj_w2, blk.qkv_colsum, blk.fc1_colsum = 0, None, None, None, None
            blk.ln1_w, blk.ln1_b = k(f32(b.norm1.weight)), k(f32(b.norm1.bias))
            blk.qkv_w, blk.qkv_b = k(tr(b.attn.qkv.weight)), k(f32(b.attn.qkv.bias))
            blk.proj_w, blk.proj_b, blk.ls1 = k(tr(b.attn.proj.weight)), k(f32(b.attn.proj.bias)), k(f32(b.ls1.gamma))
            blk.ln2_w, blk.ln2_b = k(f32(b.norm2.weight)), k(f32(b.norm2.bias))
            blk.fc1_w, blk.fc1_b = k(tr(b.mlp.fc1.weight)), k(f32(b.mlp.fc1.bias))
            blk.fc2_w, blk.fc2_b, blk.ls2 = k(tr(b.mlp.fc2.weight)), k(f32(b.mlp.fc2.bias)), k(f32(b.ls2.gamma))
        P["vit"] = dict(blocks=blocks, patch_w=k(triple(pw, tdt)), patch_b=k(f32(enc.patch_embed.proj.bias)),
                        cls_pos0=k(cls_pos0.contiguous()), pos=k(pos.contiguous()), norm_w=k(f32(enc.norm.weight)),
                        norm_b=k(f32(enc.norm.bias)))
        P["keep"] = keep
        return P
    lo_passes = parse_wlo(default_wlo(Cd) if wlo is None else wlo, L)
    P["wlo"] = {i: sorted(v) for i, v in lo_passes.items()}
    # LayerNorm fold: norm2 -> fc1 in every block, norm1 -> qkv from block 1 on (block 0's norm1 follows the patch embedding, whose
    # epilogue leaves no row statistics: it stays a LayerNorm pass).  A folded linear consumes the RAW 16-bit residual rows:
    #   y = rstd (x16 . W'^T - mean colsum) + b',   W' = W diag(w_ln) (rounded to 16 bits AFTER the fold),  b' = b + W b_ln,
    #   colsum[n] = sum_k W'[n][k] over exactly the 16-bit values the matrix pipe multiplies (hi + lo where there is a low half)
    P["fold"] = fold_eligible(Cd, N) if lnfold is None else bool(lnfold)
    if P["fold"] and Cd % 128:
        raise ValueError("lnfold needs embed_dim to be a multiple of 128")
    # C = 384 with the fold: every array indexed by the output channel of the C-wide linears (V, proj, fc2) is zero-padded to Cp = 512 rows /
    # entries -- they run as N = 512 on the 256x256 kernel with the last 128 columns masked (include/mhmr.h, mhmr_vit_desc.cpad)
    Cp = roundup(Cd, 256)
    P["cpad"] = Cp if (P["fold"] and Cd % 256) else 0

    def padn(t, n=None):
        """zero-pad dim 0 to n (default Cp) entries when this pack runs the masked form"""
        n = Cp if n is None else n
        if not P["cpad"] or t.shape[0] >= n:
            return t.contiguous()
        return torch.cat([t, t.new_zeros((n - t.shape[0],) + tuple(t.shape[1:]))], 0).contiguous()

    P["lo8"] = lo8_eligible(Cd) and bool(lo_passes)

    def folded(lin, norm, lo_rows=None):
        """-> (op16 W' [N, K], fp32 b' [N], fp32 colsum [N], hi|lo of rows lo_rows or None, their fp8 form (rows, scale) or None)"""
        W, lw, lb = f32(lin.weight).double(), f32(norm.weight).double(), f32(norm.bias).double()
        Wf = (W * lw[None, :]).float()
        bias = (f32(lin.bias).double() + W @ lb).float()
        W16 = Wf.to(tdt)
        colsum = W16.double().sum(1)
        w2 = w8 = None
        if lo_rows is not None:
            w2 = hi_lo(Wf[lo_rows], tdt)
            colsum[lo_rows] = w2.double().sum(1)
            if P["lo8"]:
                # the big GEMM multiplies the e4m3 low half: the column sums are over exactly those values (the class-row kernel keeps the
                # 16-bit low half: its sum differs by the e4m3 rounding of W_lo, ~2^-16 of a weight -- far below the row statistics' own rounding)
                rows8, sc, lo_deq = lo8_rows(W16[lo_rows], Wf[lo_rows])
                colsum[lo_rows] = W16[lo_rows].double().sum(1) + lo_deq.sum(1)
                w8 = (rows8, sc)
        return W16.contiguous(), bias.contiguous(), colsum.float().contiguous(), w2, w8

    for i, b in enumerate(enc.blocks):
        blk = blocks[i]
        f1, f2 = P["fold"] and i > 0, P["fold"]            # block 0's norm1 reads the patch embedding: no producer to fold into
        v_rows = slice(2 * Cd, 3 * Cd)
        blk.flags = (1 if f1 else 0) | (2 if f2 else 0)
        blk.proj_w2 = k(padn(hi_lo(f32(b.attn.proj.weight), tdt))) if "proj" in lo_passes.get(i, ()) else None
        blk.v_w8, blk.proj_w8, blk.v_w8_scale, blk.proj_w8_scale = None, None, 127, 127
        if P["lo8"] and "proj" in lo_passes.get(i, ()):
            pw32 = f32(b.attn.proj.weight)
            rows8, blk.proj_w8_scale, _ = lo8_rows(pw32.to(tdt), pw32)
            blk.proj_w8 = k(rows8)
        blk.ln1_w, blk.ln1_b = k(f32(b.norm1.weight)), k(f32(b.norm1.bias))
        if f1:
            w16, bias, colsum, v2, v8 = folded(b.attn.qkv, b.norm1, v_rows if "v" in lo_passes.get(i, ()) else None)
            blk.qkv_w, blk.qkv_b, blk.qkv_colsum = k(padn(w16, 2 * Cd + Cp)), k(padn(bias, 2 * Cd + Cp)), k(padn(colsum, 2 * Cd + Cp))
            blk.v_w2 = k(padn(v2)) if v2 is not None else None
            if v8 is not None:
                blk.v_w8, blk.v_w8_scale = k(v8[0]), v8[1]
        else:
            blk.qkv_w, blk.qkv_b, blk.qkv_colsum = k(padn(op(b.attn.qkv.weight), 2 * Cd + Cp)), k(padn(f32(b.attn.qkv.bias), 2 * Cd + Cp)), None
            blk.v_w2 = k(padn(hi_lo(f32(b.attn.qkv.weight)[v_rows], tdt))) if "v" in lo_passes.get(i, ()) else None
            if P["lo8"] and "v" in lo_passes.get(i, ()):
                vw32 = f32(b.attn.qkv.weight)[v_rows]
                rows8, blk.v_w8_scale, _ = lo8_rows(vw32.to(tdt), vw32)
                blk.v_w8 = k(rows8)
        blk.proj_w, blk.proj_b, blk.ls1 = k(padn(op(b.attn.proj.weight))), k(padn(f32(b.attn.proj.bias))), k(padn(f32(b.ls1.gamma)))
        blk.ln2_w, blk.ln2_b = k(f32(b.norm2.weight)), k(f32(b.norm2.bias))
        if f2:
            w16, bias, colsum, _, _ = folded(b.mlp.fc1, b.norm2)
            blk.fc1_w, blk.fc1_b, blk.fc1_colsum = k(w16), k(bias), k(colsum)
        else:
            blk.fc1_w, blk.fc1_b, blk.fc1_colsum = k(op(b.mlp.fc1.weight)), k(f32(b.mlp.fc1.bias)), None
        blk.fc2_w, blk.fc2_b, blk.ls2 = k(padn(op(b.mlp.fc2.weight))), k(padn(f32(b.mlp.fc2.bias))), k(padn(f32(b.ls2.gamma)))
    P["vit"] = dict(blocks=blocks, patch_w=k(pw.to(tdt).contiguous()), patch_b=k(f32(enc.patch_embed.proj.bias)),
                    cls_pos0=k(cls_pos0.contiguous()), pos=k(pos.contiguous()), norm_w=k(f32(enc.norm.weight)),
                    norm_b=k(f32(enc.norm.bias)))
    P["keep"] = keep
    return P


def tiny_batch(P: dict, B: int) -> bool:
    """The narrowest block linear (N = embed_dim) of the whole batch is at most HALF a round of 256x256 tiles on the chip (896^2: one
    image; 672^2: up to three).  Such a launch is latency, not throughput: exact tile rounds buy nothing, while the class-row kernels the
    token-row map needs are six more launches per block (~25 % of a batch-1 forward together with the row statistics)."""
    return P["C"] % 256 == 0 and (B * roundup(P["T"], 256) // 256) * (P["C"] // 256) <= 128


def row_map(P: dict, B: int) -> bool:
    """Mirror of the rule in csrc/capi.hip (mhmr_vit_forward): the five big GEMMs of a block run over the B * N patch rows only (the
    class rows through csrc/vit_cls.hip) when an image's patch rows are whole 256-row tiles of the 256x256 kernel -- except for tiny
    batches with folded LayerNorms, which run ALL rows (class and padding rows included, Tp a multiple of 256: that is how capi.hip tells)
    through the big GEMMs and launch no class-row kernel at all."""
    import os
    T = P["T"]
    return (not P.get("x3") and os.environ.get("MHMR_ROWMAP", "1") != "0" and "MHMR_GEMM128" not in os.environ and P["C"] % 256 == 0 and (T - 1) % 256 == 0 and
            B * roundup(T, 64) * P["C"] * 4 < 2 ** 32 and not (P.get("fold") and tiny_batch(P, B) and os.environ.get("MHMR_TINY_ALLROWS", "1") != "0"))


def padded_tokens(P: dict, B: int) -> int:
    """Rows per image in the token-major workspaces (patch tokens, the class token, zero padding).  A multiple of 64 (the attention key
    tile, the V^T row granularity) when every linear of the encoder stays on the 256x256 kernel -- the GEMMs cover the patch rows only
    (row_map), or embed_dim and B * Tp are multiples of 256; otherwise a multiple of 128, the row tile of the 128x128 kernel.
    896^2: 4160 instead of 4224 rows per image."""
    import os
    if P.get("x3"):
        return roundup(P["T"], 256 if P["C"] % 256 == 0 else 128)      # all rows through every linear: whole tiles for every batch size
    if P.get("fold") and (P.get("cpad") or (P["T"] - 1) % 256 or (not row_map(P, B) and tiny_batch(P, B) and os.environ.get("MHMR_ROWMAP", "1") != "0" and
                                                                  "MHMR_GEMM128" not in os.environ)):
        return roundup(P["T"], 256)        # folded LayerNorms without the row map: whole 256-row tiles of all rows, for every batch size
    t64, t128 = roundup(P["T"], 64), roundup(P["T"], 128)
    # the predicate of csrc/gemm256.hip mhmr_gemm256_eligible for the residual GEMMs over all B * Tp rows (32-bit residual offsets),
    # and the switch that forces the 128x128 kernel everywhere
    all_rows_256 = (P["C"] % 256 == 0 and (B * t64) % 256 == 0 and B * t64 * P["C"] * 4 < 2 ** 32 and "MHMR_GEMM128" not in os.environ)
    return t64 if (row_map(P, B) and "MHMR_GEMM128" not in os.environ) or all_rows_256 else t128


class WorkspaceCache:
    """The workspaces of the TWO most recent (pack, batch size) pairs: a run whose last batch of an epoch is smaller alternates between
    two batch sizes without re-allocating (and zero-filling) a multi-GB workspace at every switch, while a model does not pin one per
    batch size it has ever seen.  A repack (load_state_dict, .to(), repack()) drops everything, so a descriptor never outlives the
    tensors its raw pointers refer to."""
    KEEP = 2

    def __init__(self):
        self._ws = {}          # key -> workspace, oldest first

    def clear(self):
        self._ws = {}

    def get(self, P: dict, B: int, extra, nsplit: int = 1):
        """extra(P, B, z) -> dict of additional workspace tensors (z = zero-tensor factory in the operand dtype).

        nsplit > 1: the batch is cut into nsplit contiguous image blocks with a backbone workspace + descriptor each (``ws["parts"]``:
        dicts with ``desc``, ``B``, ``img0``), so that the blocks can run on different streams (model.Model, MHMR_SPLIT); ``feat32`` and
        the extras cover the whole batch.  The top-level buffer names (``hid`` ...) are those of part 0."""
        if B % nsplit:
            raise ValueError(f"batch {B} does not split into {nsplit} equal image blocks")
        key = (id(P), B, nsplit, padded_tokens(P, B // nsplit))       # (the row padding follows environment switches: A/B runs in one process)
        if key in self._ws:
            self._ws[key] = self._ws.pop(key)             # most recent last
            return self._ws[key]
        while len(self._ws) >= self.KEEP:                 # free the oldest before allocating
            self._ws.pop(next(iter(self._ws)))
        dev, tdt = P["device"], P["tdt"]
        Bh = B // nsplit
        Cd, N, Tp, H = P["C"], P["N"], padded_tokens(P, Bh), P["H"]
        z = lambda *s, dtype=tdt: torch.zeros(*s, dtype=dtype, device=dev)
        if P.get("fold") and Bh * Tp * Cd * 4 >= 2 ** 32:
            raise _lib.MhmrError(f"batch {Bh} is too large for this pack's folded LayerNorms (32-bit residual offsets of the 256x256 kernel); "
                                 "build the model with lnfold=False for such batches")
        v = P["vit"]
        parts = []
        x3 = bool(P.get("x3"))
        for i in range(nsplit):
            Mp = roundup(Bh * N, 128)
            if x3:       # operand pairs [hi | lo] and the two fp32 intermediates (include/mhmr.h, mhmr_vit_desc.x3)
                b = dict(a_patch=z(Mp, 2 * P["Kp"]), resid=z(Bh * Tp, Cd, dtype=torch.float32), xn=z(Bh * Tp, 2 * Cd), att=z(Bh * Tp, 2 * Cd),
                         hid=z(Bh * Tp, 8 * Cd), qkv32=z(Bh * Tp, 3 * Cd, dtype=torch.float32), hid32=z(Bh * Tp, 4 * Cd, dtype=torch.float32))
            else:
                pit = Cd + Cd // 2 if P.get("lo8") else Cd       # lo8: a row of xn / att carries its bf8 copy behind its C values
                b = dict(a_patch=z(Mp, P["Kp"]), resid=z(Bh * Tp, Cd, dtype=torch.float32), xn=z(Bh * Tp, pit), qk=z(Bh * Tp, 2 * Cd),
                         vt=z(Bh * H * 64, Tp), att=z(Bh * Tp, pit), hid=z(Bh * Tp, 4 * Cd),
                         attn_flags=z(_lib.lib().mhmr_attention_flag_count(Bh, Tp, H), dtype=torch.int32))
            if P.get("fold"):
                b.update(pstats=z(Bh * Tp, Cd // 64, 2, dtype=torch.float32), rowstats=z(Bh * Tp, 2, dtype=torch.float32))
            if P.get("fold") and not x3 and row_map(P, Bh):
                b["cls_pstats"] = z(Bh, Cd // 16, 2, dtype=torch.float32)       # block sums of the class rows (csrc/vit_cls.hip)
            # split-k residual linears (csrc/capi.hip): only the all-rows form of a tiny batch has launches short enough to be split
            skb = 0
            if P.get("fold") and not x3 and not P.get("lo8") and Tp % 256 == 0 and not row_map(P, Bh):
                skb = max(_lib.lib().mhmr_splitk_workspace_bytes(Bh * Tp, Cd, kk) for kk in (Cd, 2 * Cd, 4 * Cd))
            if skb:
                b["splitk"] = z(skb // 4, dtype=torch.float32)
                b["v16"] = z(Bh * Tp, Cd)            # the merged qkv launch of a short batch (csrc/capi.hip) leaves the V rows here
            d = _lib.VitDesc()
            d.dtype, d.B, d.S, d.C, d.H, d.L = P["dt_id"], Bh, P["S"], Cd, H, P["L"]
            d.G, d.N, d.T, d.Tp, d.Kp = P["G"], N, P["T"], Tp, P["Kp"]
            d.patch_w, d.patch_b, d.cls_pos0, d.pos = v["patch_w"], v["patch_b"], v["cls_pos0"], v["pos"]
            d.blocks = C.cast(v["blocks"], C.POINTER(_lib.VitBlock))
            d.norm_w, d.norm_b = v["norm_w"], v["norm_b"]
            for n in ("a_patch", "resid", "xn", "qk", "vt", "att", "hid", "attn_flags", "qkv32", "hid32"):
                setattr(d, n, b[n].data_ptr() if n in b else None)
            d.x3 = 1 if x3 else 0
            d.lo8 = 1 if (P.get("lo8") and not x3) else 0
            d.pstats = b["pstats"].data_ptr() if P.get("fold") else None
            d.rowstats = b["rowstats"].data_ptr() if P.get("fold") else None
            d.splitk, d.splitk_bytes = (b["splitk"].data_ptr(), skb) if skb else (None, 0)
            d.cpad = P.get("cpad", 0) if not x3 else 0
            d.v16 = b["v16"].data_ptr() if "v16" in b else None
            d.cls_pstats = b["cls_pstats"].data_ptr() if "cls_pstats" in b else None
            parts.append(dict(desc=d, B=Bh, img0=i * Bh, bufs=b))
        ws = dict(parts[0]["bufs"])
        ws["feat32"] = z(B * N, Cd, dtype=torch.float32)
        ws.update(extra(P, B, z))
        ws["parts"], ws["vit_desc"], ws["Tp"] = parts, parts[0]["desc"], Tp
        self._ws[key] = ws
        return ws
