"""GPU input preprocessing: the step in front of ``Model.forward``.

Drop-in for the reference's ``demo.open_image`` (``demo.py:27-51``) with the resize / pad / normalise done by
``mhmr_preprocess_u8`` on the device, plus a batched ``get_camera_parameters`` (``demo.py:53-68``).  Image *decoding*
(JPEG/PNG -> uint8 RGB) stays with PIL on the host, as in the reference.

Host side = table construction only (a few KB per image size, cached):
  * ``contain_size`` / ``pad_offsets``: the geometry of ``ImageOps.contain`` + ``ImageOps.pad`` (Pillow, a dependency of the
    reference: ``requirements.txt``), restated from Pillow's documented behaviour;
  * ``resample_coeffs``: Pillow's ``precompute_coeffs`` + ``normalize_coeffs_8bpc`` for the bicubic filter (a = -0.5,
    support 2, widened by the down-scale factor; float64 weights normalised to 1, then quantised to 22 fractional bits);
  * ``norm_lut``: ``utils/image.py:12-24`` evaluated on the 256 possible byte values.
``tests/test_preprocess.py`` holds these bit-exact against the installed Pillow itself."""
from __future__ import annotations

import math
from functools import lru_cache

import numpy as np
import torch

from . import _lib

IMG_NORM_MEAN = [0.485, 0.456, 0.406]           # reference utils/image.py:9-10
IMG_NORM_STD = [0.229, 0.224, 0.225]
PRECISION_BITS = 32 - 8 - 2                     # Pillow's 8 bpc fixed point


def contain_size(W: int, H: int, S: int) -> tuple[int, int]:
    """Output (width, height) of ``ImageOps.contain(img, (S, S))`` (demo.py:40)."""
    im_ratio, dest_ratio = W / H, 1.0
    ow, oh = S, S
    if im_ratio != dest_ratio:
        if im_ratio > dest_ratio:
            nh = round(H / W * S)
            if nh != S:
                oh = nh
        else:
            nw = round(W / H * S)
            if nw != S:
                ow = nw
    return ow, oh


def pad_offsets(ow: int, oh: int, S: int) -> tuple[int, int]:
    """Paste position of ``ImageOps.pad(.., (S, S))`` with the default centring (demo.py:44)."""
    if (ow, oh) == (S, S):
        return 0, 0
    if ow != S:
        return round((S - ow) * 0.5), 0
    return 0, round((S - oh) * 0.5)


def _bicubic(x: float) -> float:
    a = -0.5
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


@lru_cache(maxsize=256)
def resample_coeffs(in_size: int, out_size: int):
    """Pillow's coefficient tables for resizing ``in_size`` -> ``out_size`` samples with the bicubic filter over the whole
    axis.  Returns (kk int32 [out_size, ksize], bounds int32 [out_size, 2] = (first tap, tap count), ksize)."""
    scale = filterscale = in_size / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = [_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        if ww != 0.0:
            w = [v / ww for v in w]
        for x, v in enumerate(w):
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return kk, bounds, ksize


def norm_lut() -> np.ndarray:
    """[3, 256] float32: ``normalize_rgb`` (utils/image.py:12-24) of every byte value, with the reference's own arithmetic
    (float32 divide by 255, float64 mean / std broadcast, cast back to float32)."""
    v = np.arange(256, dtype=np.uint8).reshape(1, 256, 1).repeat(3, axis=2)        # (W=1, H=256, 3)
    img = v.astype(np.float32) / 255.
    img = np.transpose(img, (2, 0, 1))
    img = (img - np.asarray(IMG_NORM_MEAN).reshape(3, 1, 1)) / np.asarray(IMG_NORM_STD).reshape(3, 1, 1)
    return np.ascontiguousarray(img.astype(np.float32).reshape(3, 256))


class Preprocessor:
    """``pre = Preprocessor(img_size, device); x = pre(img_u8)`` with ``img_u8`` a uint8 ``[H, W, 3]`` RGB tensor (host or
    device) -> ``[1, 3, S, S]`` float32 on the device, bit-identical to ``demo.open_image``'s tensor.  Asynchronous on the
    current stream; coefficient tables are cached per source size."""

    def __init__(self, img_size: int, device=torch.device("cuda")):
        self.S = int(img_size)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.MhmrError("Preprocessor needs a HIP device: there is no CPU fallback")
        self.lut = torch.from_numpy(norm_lut()).to(self.device)
        self._tables = {}

    def _plan(self, H: int, W: int):
        key = (H, W)
        if key not in self._tables:
            if H > W * 100:
                raise ValueError("images taller than 100x their width take a different pass order in PIL; not supported")
            S = self.S
            ow, oh = contain_size(W, H, S)
            px, py = pad_offsets(ow, oh, S)
            kh, bh, ksh = resample_coeffs(W, ow)
            kv, bv, ksv = resample_coeffs(H, oh)
            y0 = int(bv[0, 0])
            rows = int(bv[-1, 0] + bv[-1, 1]) - y0
            dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(self.device)
            self._tables[key] = dict(ow=ow, oh=oh, px=px, py=py, kh=dev(kh), bh=dev(bh), ksh=ksh, kv=dev(kv), bv=dev(bv), ksv=ksv,
                                     y0=y0, rows=rows, tmp=torch.empty(rows * ow * 3, dtype=torch.uint8, device=self.device))
        return self._tables[key]

    def __call__(self, img_u8: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
        if img_u8.dtype != torch.uint8 or img_u8.dim() != 3 or img_u8.shape[2] != 3:
            raise ValueError("expected a uint8 [H, W, 3] RGB image")
        img = img_u8.to(self.device, non_blocking=True).contiguous()
        H, W = int(img.shape[0]), int(img.shape[1])
        t = self._plan(H, W)
        S = self.S
        if out is None:
            out = torch.empty(1, 3, S, S, dtype=torch.float32, device=self.device)
        assert out.is_contiguous() and out.numel() == 3 * S * S and out.dtype == torch.float32
        st = torch.cuda.current_stream(self.device).cuda_stream
        _lib.check(_lib.lib().mhmr_preprocess_u8(img.data_ptr(), H, W, t["kh"].data_ptr(), t["bh"].data_ptr(), t["ksh"], t["kv"].data_ptr(),
                                                 t["bv"].data_ptr(), t["ksv"], t["ow"], t["oh"], t["y0"], t["rows"], S, t["px"], t["py"],
                                                 self.lut.data_ptr(), t["tmp"].data_ptr(), out.data_ptr(), st), "mhmr_preprocess_u8")
        return out


_PRE = {}


def open_image(img_path, img_size, device=torch.device("cuda")):
    """Same contract as the reference's ``demo.open_image`` (demo.py:27-51) -> ``(x [1,3,S,S] float32 on device, PIL image)``;
    PIL only decodes, the resize / pad / normalise run in ``mhmr_preprocess_u8``."""
    from PIL import Image
    img_pil = Image.open(img_path).convert("RGB")
    key = (int(img_size), str(device))
    if key not in _PRE:
        _PRE[key] = Preprocessor(img_size, device)
    x = _PRE[key](torch.from_numpy(np.array(img_pil)))
    return x, img_pil.copy()


def get_camera_parameters(img_size, fov=60, p_x=None, p_y=None, device=torch.device("cuda"), batch: int = 1):
    """demo.py:53-68 for a whole batch: ``K [batch, 3, 3]``; ``p_x`` / ``p_y`` may be scalars or per-image sequences."""
    focal = img_size / (2 * np.tan(np.radians(fov) / 2))
    K = torch.eye(3).repeat(batch, 1, 1)
    K[:, 0, 0] = K[:, 1, 1] = focal
    if p_x is not None and p_y is not None:
        K[:, 0, 2] = torch.as_tensor(p_x, dtype=torch.float32) * img_size
        K[:, 1, 2] = torch.as_tensor(p_y, dtype=torch.float32) * img_size
    else:
        K[:, 0, 2] = K[:, 1, 2] = img_size // 2
    return K.to(device)
