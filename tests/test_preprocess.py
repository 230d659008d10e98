"""Input preprocessing.  CPU: the host tables + the integer two-pass algorithm are bit-exact against the
installed Pillow running the reference's own open_image arithmetic (oracle/preprocess_ref.py).  GPU: mhmr_preprocess_u8
through the C ABI is bit-exact against the same."""
import numpy as np
import pytest
import torch

from multi_hmr_amd import preprocess as pp
from oracle import preprocess_ref as ref

PIL = pytest.importorskip("PIL")
from PIL import Image  # noqa: E402

SIZES = [(640, 480, 224), (480, 640, 224), (1920, 1080, 448), (333, 500, 448), (896, 896, 896), (100, 75, 224),
         (1000, 37, 224), (224, 224, 224), (500, 499, 224), (61, 4000, 448)]


def _image(W, H, seed):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:H, 0:W]
    base = np.stack([(xx * 255 // max(W - 1, 1)), (yy * 255 // max(H - 1, 1)), ((xx + yy) % 256)], -1).astype(np.int32)
    noise = rng.integers(-60, 60, size=(H, W, 3))
    img = np.clip(base + noise, 0, 255).astype(np.uint8)
    img[: H // 7, : W // 5] = 255                      # saturated block: exercises the clip after negative bicubic lobes
    img[H // 2:H // 2 + 3, :] = 0
    return img


@pytest.mark.parametrize("W,H,S", SIZES)
def test_tables_and_integer_algorithm_match_pillow(W, H, S):
    img = _image(W, H, W + H + S)
    x_ref, u8_ref = ref.open_image_ref(Image.fromarray(img), S)
    ow, oh = pp.contain_size(W, H, S)
    px, py = pp.pad_offsets(ow, oh, S)
    kh, bh, _ = pp.resample_coeffs(W, ow)
    kv, bv, _ = pp.resample_coeffs(H, oh)
    got = np.zeros((S, S, 3), np.uint8)
    got[py:py + oh, px:px + ow] = ref.resample_u8(img, kh, bh, kv, bv)
    assert np.array_equal(got, u8_ref)
    lut = pp.norm_lut()
    x = np.stack([lut[c][got[..., c]] for c in range(3)])[None]
    assert x.dtype == np.float32 and np.array_equal(x, x_ref)


def test_camera_parameters_batch():
    K = pp.get_camera_parameters(896, fov=60, device="cpu", batch=3)
    assert K.shape == (3, 3, 3)
    assert abs(float(K[0, 0, 0]) - 775.9587) < 1e-3 and float(K[2, 0, 2]) == 448 and float(K[1, 2, 2]) == 1
    K = pp.get_camera_parameters(448, p_x=[0.5, 0.25], p_y=[0.5, 0.75], device="cpu", batch=2)
    assert float(K[1, 0, 2]) == 112 and float(K[1, 1, 2]) == 336


def test_preprocessor_refuses_cpu():
    from multi_hmr_amd import _lib
    with pytest.raises(_lib.MhmrError):
        pp.Preprocessor(224, device="cpu")


@pytest.mark.gpu
@pytest.mark.parametrize("W,H,S", SIZES)
def test_gpu_preprocess_bit_exact(W, H, S):
    img = _image(W, H, W + H + S)
    x_ref, _ = ref.open_image_ref(Image.fromarray(img), S)
    pre = pp.Preprocessor(S, "cuda:0")
    x = pre(torch.from_numpy(img))
    assert x.shape == (1, 3, S, S) and x.dtype == torch.float32
    assert np.array_equal(x.cpu().numpy(), x_ref)
    x2 = pre(torch.from_numpy(img).cuda())            # device-resident input, cached tables
    assert torch.equal(x, x2)


@pytest.mark.gpu
def test_gpu_open_image_file(tmp_path):
    img = _image(801, 533, 5)
    p = tmp_path / "im.png"
    Image.fromarray(img).save(p)
    x, pil_full = pp.open_image(str(p), 448, torch.device("cuda:0"))
    x_ref, _ = ref.open_image_ref(Image.open(p), 448)
    assert np.array_equal(x.cpu().numpy(), x_ref) and pil_full.size == (801, 533)
