"""BASELINE.json config 1: `multiHMR_672_S on example_data/ via demo.py`.  Golden vectors = the reference's own demo.open_image ->
get_camera_parameters -> forward_model run verbatim on its seven example photographs (tests/golden/make_golden_demo.py).

* CPU (build container only: the photographs live under /root/reference): the drop-in's preprocessing -- host tables + the integer
  two-pass algorithm the kernel implements -- reproduces the reference's padded uint8 image of ALL SEVEN photographs bit for bit.
* GPU: for the two photographs whose decoded pixels are part of the fixture, ``Preprocessor`` (mhmr_preprocess_u8) is bit-exact
  and ``forward_model`` returns the same persons (count, order) with every tensor within 1e-3 (f16 operands)."""
import glob
import os
import zlib

import numpy as np
import pytest
import torch

import parity
from multi_hmr_amd import preprocess as pp
import synthetic
from oracle import preprocess_ref as ref

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "demo_672_s.npz"))
S = 672
EXAMPLES = "/root/reference/example_data"


@pytest.mark.skipif(not os.path.isdir(EXAMPLES), reason="the reference's example photographs exist only in the build container")
def test_preprocessing_of_the_seven_example_photographs_is_bit_exact():
    from PIL import Image
    paths = sorted(glob.glob(os.path.join(EXAMPLES, "*.jpg")))
    assert [os.path.basename(p) for p in paths] == list(GOLD["names"])
    lut = pp.norm_lut()
    for i, p in enumerate(paths):
        img = np.asarray(Image.open(p).convert("RGB"))
        H, W = img.shape[:2]
        ow, oh = pp.contain_size(W, H, S)
        px, py = pp.pad_offsets(ow, oh, S)
        kh, bh, _ = pp.resample_coeffs(W, ow)
        kv, bv, _ = pp.resample_coeffs(H, oh)
        got = np.zeros((S, S, 3), np.uint8)
        got[py:py + oh, px:px + ow] = ref.resample_u8(img, kh, bh, kv, bv)
        assert zlib.crc32(got.tobytes()) == int(GOLD[f"crc_{i}"]), p
        x = np.stack([lut[c][got[..., c]] for c in range(3)])
        assert abs(float(x.astype(np.float64).sum()) - float(GOLD[f"x_sum_{i}"])) < 1e-6 * x.size


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


@pytest.mark.gpu
def test_open_image_and_forward_model_on_real_photographs(smplx_data, mean_params):
    from multi_hmr_amd import Model, forward_model, get_camera_parameters
    from oracle import roma_ref
    sd = synthetic.make_state_dict("dinov2_vits14", S, seed=31)
    sd["mlp_classif.2.bias"] = torch.from_numpy(GOLD["classif_bias"])
    model = Model(backbone="dinov2_vits14", img_size=S, smplx_data=smplx_data, mean_params=mean_params, precision="f16")
    model.load_state_dict(sd, strict=True)
    model = model.to("cuda:0").eval()
    pre = pp.Preprocessor(S, "cuda:0")
    lut = torch.from_numpy(pp.norm_lut()).cuda()
    K = get_camera_parameters(S, device=torch.device("cuda:0"))
    stored = [i for i in range(7) if f"pixels_{i}" in GOLD.files]
    assert len(stored) == 2
    for i in stored:
        x = pre(torch.from_numpy(GOLD[f"pixels_{i}"]))
        # bit-exact preprocessing: invert the 256-entry normalisation table and compare the padded uint8 image by CRC
        u8 = torch.stack([(x[0, c][..., None] == lut[c]).float().argmax(-1) for c in range(3)], -1).to(torch.uint8)
        assert torch.equal(torch.stack([lut[c][u8[..., c].long()] for c in range(3)]), x[0])
        assert zlib.crc32(u8.cpu().numpy().tobytes()) == int(GOLD[f"crc_{i}"])
        humans = forward_model(model, x, K, det_thresh=float(GOLD["det_thresh"]), nms_kernel_size=int(GOLD["nms_kernel_size"]))
        assert len(humans) == int(GOLD[f"n_{i}"]), (i, len(humans))
        for k in humans[0].keys():
            got = torch.stack([h[k] for h in humans]).cpu()
            want = GOLD[f"h{i}_{k}"]
            if k == "v3d":
                got = got[:, ::8]
            if k == "rotvec":
                e = _rel(roma_ref.rotvec_to_rotmat(got).numpy(), roma_ref.rotvec_to_rotmat(torch.from_numpy(want)).numpy())
            else:
                e = _rel(got.numpy(), want)
            assert e < parity.tolerance(k, "f16"), (i, k, e)
