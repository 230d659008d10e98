"""Drop-in for the inference helpers of the reference ``demo.py`` (lines 27-126): ``open_image``,
``get_camera_parameters``, ``load_model``, ``forward_model``.  Rendering / CLI (demo.py:128-386) is out of scope."""
from __future__ import annotations

import os

import torch

from .model import Model

from .preprocess import IMG_NORM_MEAN, IMG_NORM_STD, get_camera_parameters, open_image  # noqa: F401  (demo.py:27-68 on the GPU)

CACHE_DIR_MULTIHMR = "models/multiHMR"          # reference utils/constants.py:9


def load_model(model_name, device=torch.device("cuda"), **model_kwargs):
    """demo.py:70-106: checkpoint -> ``Model(**vars(ckpt['args']))`` -> ``load_state_dict(strict=False)``.
    No download is attempted (no network); the checkpoint must exist under models/multiHMR/."""
    ckpt_path = os.path.join(CACHE_DIR_MULTIHMR, model_name + ".pt")
    if not os.path.isfile(ckpt_path):
        raise FileNotFoundError(f"{ckpt_path} not found")
    ckpt = torch.load(ckpt_path, map_location="cpu", weights_only=False)
    kwargs = dict(vars(ckpt["args"]))
    if "anny" in ckpt_path:                      # demo.py:94-95: the Anny checkpoints build multi_hmr_anny.multi_hmr.Multi_HMR
        from .anny_model import Multi_HMR as ModelAnny
        kwargs.update(model_kwargs)
        model = ModelAnny(**kwargs).to(device)
    else:
        kwargs["type"] = ckpt["args"].train_return_type
        kwargs["img_size"] = ckpt["args"].img_size[0]
        kwargs.update(model_kwargs)
        model = Model(**kwargs).to(device)
    model.load_state_dict(ckpt["model_state_dict"], strict=False)
    return model


def forward_model(model, input_image, camera_parameters, det_thresh=0.3, nms_kernel_size=1, use_graph=False):
    """demo.py:108-126.  The reference wraps the call in fp16 autocast; ``Model.forward`` disables autocast for
    its own body, so the precision is whatever ``Model(precision=...)`` selected.
    Extension: ``use_graph=True`` replays the forward from a hipGraph recorded on first use for this batch size / threshold / NMS window
    (``graphed.GraphedForward``: same kernels, same results; what it saves is the host's per-launch work, which matters at batch 1)."""
    if use_graph:
        from .graphed import graphed
        return graphed(model, input_image.shape[0], det_thresh, nms_kernel_size)(input_image.float(), camera_parameters.float())
    with torch.no_grad():
        with torch.autocast("cuda", enabled=True):
            humans = model(input_image, is_training=False, nms_kernel_size=int(nms_kernel_size), det_thresh=det_thresh,
                           K=camera_parameters)
    return humans
