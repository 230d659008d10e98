"""TEST INFRASTRUCTURE -- never imported by the product path (multi_hmr_amd/).

CPU oracle for the input preprocessing row:
  * ``open_image_ref``: the reference's ``demo.open_image`` arithmetic (/root/reference/demo.py:27-51 +
    utils/image.py:12-24) run with the real Pillow (a dependency of the reference that IS installed here and on the GPU
    box), on an in-memory PIL image instead of a path.  Parity for this row is therefore pinned to Pillow itself.
  * ``resample_u8``: numpy restatement of what ``mhmr_preprocess_u8`` computes from the coefficient tables (Pillow's
    ImagingResampleHorizontal_8bpc / Vertical_8bpc fixed-point loops, Resample.c), used on CPU to check the tables and the
    integer algorithm against Pillow before the kernel ever runs."""
import numpy as np

IMG_NORM_MEAN = [0.485, 0.456, 0.406]
IMG_NORM_STD = [0.229, 0.224, 0.225]
PRECISION_BITS = 22


def normalize_rgb(img):
    """utils/image.py:12-24."""
    img = img.astype(np.float32) / 255.
    img = np.transpose(img, (2, 0, 1))
    img = (img - np.asarray(IMG_NORM_MEAN).reshape(3, 1, 1)) / np.asarray(IMG_NORM_STD).reshape(3, 1, 1)
    return img.astype(np.float32)


def open_image_ref(img_pil, img_size):
    """demo.py:31-49 on an already opened PIL image -> (x [1,3,S,S] float32 numpy, resized-and-padded uint8 [S,S,3])."""
    from PIL import ImageOps
    img_pil = img_pil.convert("RGB")
    img_pil = ImageOps.contain(img_pil, (img_size, img_size))
    img_pil = ImageOps.pad(img_pil, size=(img_size, img_size))
    u8 = np.asarray(img_pil)
    return normalize_rgb(u8)[None], u8


def _pass(src, kk, bounds):
    """One 8 bpc pass along axis 0 of src [n_in, m, 3] -> [n_out, m, 3]."""
    n_out = kk.shape[0]
    out = np.empty((n_out,) + src.shape[1:], dtype=np.uint8)
    s64 = src.astype(np.int64)
    for o in range(n_out):
        x0, n = int(bounds[o, 0]), int(bounds[o, 1])
        acc = (1 << (PRECISION_BITS - 1)) + np.tensordot(kk[o, :n].astype(np.int64), s64[x0:x0 + n], axes=(0, 0))
        out[o] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return out


def resample_u8(img, kh, bh, kv, bv):
    """img uint8 [H,W,3] -> uint8 [oh,ow,3]: horizontal pass then vertical pass, uint8 in between (Pillow's order)."""
    tmp = _pass(np.transpose(img, (1, 0, 2)), kh, bh)          # [ow, H, 3]
    return _pass(np.transpose(tmp, (1, 0, 2)), kv, bv)         # [oh, ow, 3]
