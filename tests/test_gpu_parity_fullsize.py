"""-m gpu: the north-star parity contract AT BASELINE.json's sizes.  The golden vectors are the outputs of the reference's own
model.py run verbatim on CPU fp32 (tests/golden/make_golden.py, cases *_full: full depth, full resolution, one or two images):

  vits_672_full   multiHMR_672_S   672^2,  T = 2305, ViT-S/14 12 blocks, 8 + 5 persons        (config 2)
  vitl_672_full   multiHMR_672_L   672^2,  T = 2305, ViT-L/14 24 blocks, 8 persons            (config 3)
  vitl_896_full   multiHMR_896_L   896^2,  T = 4097, ViT-L/14 24 blocks, 8 persons            (config 4, the benchmark)
  vitl_1288_full  multiHMR_1288_L  1288^2, T = 8465, ViT-L/14 24 blocks, 20 persons           (config 5)
  vitb_672_full   multiHMR_672_B   672^2,  T = 2305, ViT-B/14 12 blocks, 6 + 9 persons        (released size without a BASELINE config)
  vitl_672_hostile_w / _m   config 3's shape with hostile weight statistics (synthetic.make_hostile: LayerScale over three decades,
                  LayerNorm weights with x10 ... x30 channels, large biases / token rows ~2 sigma away from zero through the depth).
                  hostile_w is what precision="auto" (the product default) exists for: its attention logits are steep (vit.logit_gain
                  ~20 against 1.5), the pack selects the f16x3 mode and the case is held to the plain 1e-3 on every key, like every
                  other case -- no sensitivity-scaled slack anywhere in this file.  The same weights with the precision forced to f16 /
                  bf16 are measured beside it (reported; 2-10x outside the contract, as the network's own sensitivity predicts).

Every tensor of the output dict must be within the tolerances of tests/parity.py (1e-3 relative L2 for f16 operands, the product
precision and what bench.py reports -- every key, no exceptions: the V / attention-output projections of blocks 0..11 carry the low
halves of their weights, vit.DEFAULT_WLO).  The measured values are also written to gpurun_out/parity_fullsize.json (pytest -q hides prints);
bf16 operands are measured beside, held to 2e-2."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import make_golden  # noqa: E402
import parity  # noqa: E402
from multi_hmr_amd import Model  # noqa: E402
from oracle import roma_ref  # noqa: E402
from parity import CHECKED, MAXTOL, TOL, maxrel, rel  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
REPORT = os.path.join(ROOT, "gpurun_out", "parity_fullsize.json")

def _report(name, precision, entry):
    os.makedirs(os.path.dirname(REPORT), exist_ok=True)
    data = {}
    if os.path.isfile(REPORT):
        with open(REPORT) as f:
            data = json.load(f)
    data[f"{name}/{precision}"] = entry
    with open(REPORT, "w") as f:
        json.dump(data, f, indent=1, sort_keys=True)


BASE_CASES = ["vits_672_full", "vitb_672_full", "vitl_672_full", "vitl_896_full", "vitl_1288_full", "vitl_672_hostile_m"]
#: (golden, precision).  "auto" is the product default: plain f16 for every case but hostile_w, whose weights select f16x3 at pack time
PARAMS = ([(n, p) for n in BASE_CASES for p in ("f16", "bf16")] +
          [("vitl_672_hostile_w", "auto"), ("vitl_672_hostile_w", "f16"), ("vitl_672_hostile_w", "bf16"), ("vitl_896_full", "auto"),
           # the goldens are one or two images: since round 5 such tiny batches run ALL rows through the big GEMMs (vit.tiny_batch).  The
           # benchmark's path -- token-row map + class-row kernels -- is what larger batches take: the same goldens through it as well
           ("vitl_672_full", "f16+rowmap"), ("vitl_896_full", "f16+rowmap"), ("vitb_672_full", "f16+rowmap")])


@pytest.mark.parametrize("name,precision", PARAMS)
def test_full_size_forward_matches_reference_golden(name, precision, smplx_data, mean_params, monkeypatch):
    if precision.endswith("+rowmap"):
        precision = precision[:-7]
        monkeypatch.setenv("MHMR_TINY_ALLROWS", "0")
    cfg = make_golden.CASES[name]
    gold = np.load(os.path.join(GOLD, name + ".npz"))
    vs = cfg.get("vstride", 1)
    model = Model(backbone=cfg["backbone"], img_size=cfg["img_size"], smplx_data=smplx_data, mean_params=mean_params, precision=precision)
    model.load_state_dict(make_golden.case_state_dict(cfg), strict=True)
    model = model.to("cuda:0").eval()
    x, K, idx = make_golden.case_inputs(cfg)
    z = model.backbone_features(x.cuda()).cpu()
    packed = model.packed_precision                      # what "auto" resolved to
    from multi_hmr_amd import vit as _vit
    rowmap = _vit.row_map(model._packed, x.shape[0])
    if os.environ.get("MHMR_TINY_ALLROWS") == "0" and cfg["backbone"] != "dinov2_vits14" and cfg["img_size"] != 1288:
        assert rowmap
    hostile_w = name == "vitl_672_hostile_w"
    if precision == "auto":
        assert packed == ("f16x3" if hostile_w else "f16"), (packed, model._packed.get("logit_gain"))
    e_bb = rel(z[:, :: max(1, z.shape[1] // 64)].numpy(), gold["backbone"])
    out = model(x.cuda(), idx=tuple(i.cuda() for i in idx), K=K.cuda(), is_training=True)
    got = {k: out[k].cpu() for k in CHECKED + ["rotvec"]}
    for k in ("v3d", "v2d"):
        got[k] = got[k][:, ::vs]
    errs = {k: rel(got[k].numpy(), gold[k]) for k in CHECKED}
    errs["rotvec"] = rel(roma_ref.rotvec_to_rotmat(got["rotvec"]).numpy(), roma_ref.rotvec_to_rotmat(torch.from_numpy(gold["rotvec"])).numpy())
    errs["smplx_params"] = rel(parity.smplx_param_vector(got["rotmat"], got["shape"], got["expression"]),
                               parity.smplx_param_vector(gold["rotmat"], gold["shape"], gold["expression"]))
    vmax_mm = 1e3 * float(np.abs(got["v3d"].numpy() - gold["v3d"]).max())
    finite = all(bool(torch.isfinite(v).all()) for v in got.values())
    # the same keys in the max norm (rotations through the matrices they encode)
    merrs = {k: maxrel(got[k].numpy(), gold[k]) for k in CHECKED}
    tolkey = "bf16" if precision == "bf16" else "f16"           # f16, auto and f16x3 answer to the 1e-3 contract
    sens = {k[5:]: float(gold[k]) for k in gold.files if k.startswith("sens_")}
    _report(name, precision + ("+rowmap" if (rowmap and x.shape[0] <= 3) else ""), {"tolerance": TOL[tolkey], "packed_precision": packed, "token_row_map": bool(rowmap), "wlo": model._packed["wlo"], "lnfold": bool(model._packed["fold"]),
                              "logit_gain_max": max(model._packed["logit_gain"]) if "logit_gain" in model._packed else None,
                              "backbone_rel_l2": e_bb, "max_vertex_error_mm": vmax_mm,
                              "worst_rel_l2": max(errs.values()), "rel_l2": errs, "finite": finite,
                              "max_norm_tolerance": {k: parity.maxtol(k, tolkey) for k in merrs}, "worst_max_norm": max(merrs.values()), "max_norm": merrs,
                              "network_sensitivity": sens,
                              "tokens": int(cfg["img_size"] // 14) ** 2 + 1, "persons": int(sum(cfg["persons"]))})
    print(f"\n[parity {name} {precision} -> {packed}] backbone {e_bb:.2e}; max vertex error {vmax_mm:.3f} mm; " + " ".join(f"{k}={v:.1e}" for k, v in errs.items()))
    assert finite
    if hostile_w and precision != "auto":
        # Hostile weight statistics with the precision FORCED to a single 16-bit rounding per operand: measured and reported, not held to
        # the contract.  The golden stores the NETWORK's own sensitivity sens_<key> (the fp32 reference's response to ONE f16 rounding
        # of its input image: `offset` 2.5e-3); a pipeline that makes ~10^2 such roundings lands at 2-10x the contract here (round 4:
        # offset 9.8e-3, rotmat 2.8e-3, v3d 1.8e-3).  These weights are what precision="auto" exists for: the case above this one
        # packs them as f16x3 and is held to the plain 1e-3 on EVERY key, max norm included.  What is asserted here is only that the
        # single-rounding result is the well-behaved one that analysis predicts (no blow-up), so that a regression still shows.
        for k, v in errs.items():
            assert v < (3e-2 if precision == "f16" else 3e-1), (name, k, v)
        return
    # the contract, per key, no slack: 1e-3 relative L2 (bf16: its own, looser bound) ...
    for k, v in errs.items():
        assert v < TOL[tolkey], (name, precision, k, v, TOL[tolkey])
    assert e_bb < 2 * TOL[tolkey], e_bb                  # not a north-star output; informational bound
    # ... and the max-norm gate, per key (parity.maxtol: 1e-3, except the rotation matrices -- 2e-3: the 6D decode amplifies -- and the two
    # pixel keys behind them, 1.5e-3)
    for k, v in merrs.items():
        assert v < parity.maxtol(k, tolkey), (name, "max-norm", k, v, parity.maxtol(k, tolkey))
    if packed == "f16x3":
        assert e_bb < 1e-4 and max(errs.values()) < 3e-4, (e_bb, errs)      # the pair mode sits at fp32 accuracy, far inside the contract


def test_four_image_batch_uses_64_row_padding_and_matches_golden(smplx_data, mean_params):
    """B = 4 images of ViT-L 672^2: rows per image = 2368 (a multiple of 64, vit.padded_tokens) instead of 2432; every image of
    the batch is the golden case's image, so every person block must match the golden vectors."""
    from multi_hmr_amd import vit
    name = "vitl_672_full"
    cfg = make_golden.CASES[name]
    gold = np.load(os.path.join(GOLD, name + ".npz"))
    model = Model(backbone=cfg["backbone"], img_size=cfg["img_size"], smplx_data=smplx_data, mean_params=mean_params, precision="f16")
    model.load_state_dict(make_golden.case_state_dict(cfg), strict=True)
    model = model.to("cuda:0").eval()
    x, K, idx = make_golden.case_inputs(cfg)
    assert x.shape[0] == 1
    B, P1 = 4, idx[0].shape[0]
    xb, Kb = x.repeat(B, 1, 1, 1), K.repeat(B, 1, 1)
    idxb = tuple(torch.cat([(i + b) if j == 0 else i for b in range(B)]) for j, i in enumerate(idx))
    out = model(xb.cuda(), idx=tuple(i.cuda() for i in idxb), K=Kb.cuda(), is_training=True)
    assert vit.padded_tokens(model._packed, B) == 2368
    vs = cfg.get("vstride", 1)
    for b in (0, B - 1):
        got = {k: out[k][b * P1:(b + 1) * P1].cpu() for k in CHECKED}
        for k in ("v3d", "v2d"):
            got[k] = got[k][:, ::vs]
        errs = {k: rel(got[k].numpy(), gold[k]) for k in CHECKED}
        parity.assert_within(errs, "f16", f"{name} image {b} of {B}")
