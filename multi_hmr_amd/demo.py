"""Drop-in for the inference helpers of the reference ``demo.py`` (lines 27-126): ``open_image``,
``get_camera_parameters``, ``load_model``, ``forward_model``.  Rendering / CLI (demo.py:128-386) is out of scope."""
from __future__ import annotations

import os

import numpy as np
import torch

from .model import Model

CACHE_DIR_MULTIHMR = "models/multiHMR"          # reference utils/constants.py:9
IMG_NORM_MEAN = [0.485, 0.456, 0.406]           # reference utils/image.py:9-10
IMG_NORM_STD = [0.229, 0.224, 0.225]


def normalize_rgb(img):
    """utils/image.py:12-24: uint8 HWC -> float32 CHW, ImageNet-normalised."""
    img = img.astype(np.float32) / 255.0
    img = np.transpose(img, (2, 0, 1))
    img = (img - np.asarray(IMG_NORM_MEAN).reshape(3, 1, 1)) / np.asarray(IMG_NORM_STD).reshape(3, 1, 1)
    return img.astype(np.float32)


def open_image(img_path, img_size, device=torch.device("cuda")):
    """demo.py:27-51: open, resize keeping the aspect ratio, zero-pad to a square, normalise."""
    from PIL import Image, ImageOps
    img_pil = Image.open(img_path).convert("RGB")
    img_pil_full = img_pil.copy()
    img_pil = ImageOps.contain(img_pil, (img_size, img_size))
    img_pil = ImageOps.pad(img_pil, size=(img_size, img_size))
    x = torch.from_numpy(normalize_rgb(np.asarray(img_pil))).unsqueeze(0).to(device)
    return x, img_pil_full


def get_camera_parameters(img_size, fov=60, p_x=None, p_y=None, device=torch.device("cuda")):
    """demo.py:53-68."""
    K = torch.eye(3)
    focal = img_size / (2 * np.tan(np.radians(fov) / 2))
    K[0, 0], K[1, 1] = focal, focal
    if p_x is not None and p_y is not None:
        K[0, -1], K[1, -1] = p_x * img_size, p_y * img_size
    else:
        K[0, -1], K[1, -1] = img_size // 2, img_size // 2
    return K.unsqueeze(0).to(device)


def load_model(model_name, device=torch.device("cuda"), **model_kwargs):
    """demo.py:70-106: checkpoint -> ``Model(**vars(ckpt['args']))`` -> ``load_state_dict(strict=False)``.
    No download is attempted (no network); the checkpoint must exist under models/multiHMR/."""
    ckpt_path = os.path.join(CACHE_DIR_MULTIHMR, model_name + ".pt")
    if not os.path.isfile(ckpt_path):
        raise FileNotFoundError(f"{ckpt_path} not found")
    ckpt = torch.load(ckpt_path, map_location="cpu", weights_only=False)
    kwargs = dict(vars(ckpt["args"]))
    kwargs["type"] = ckpt["args"].train_return_type
    kwargs["img_size"] = ckpt["args"].img_size[0]
    kwargs.update(model_kwargs)
    model = Model(**kwargs).to(device)
    model.load_state_dict(ckpt["model_state_dict"], strict=False)
    return model


def forward_model(model, input_image, camera_parameters, det_thresh=0.3, nms_kernel_size=1):
    """demo.py:108-126.  The reference wraps the call in fp16 autocast; ``Model.forward`` disables autocast for
    its own body, so the precision is whatever ``Model(precision=...)`` selected."""
    with torch.no_grad():
        with torch.autocast("cuda", enabled=True):
            humans = model(input_image, is_training=False, nms_kernel_size=int(nms_kernel_size), det_thresh=det_thresh,
                           K=camera_parameters)
    return humans
