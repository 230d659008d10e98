"""Image-sharded inference over the GPUs of one node: the N > 1 form of ``demo.forward_model`` (reference demo.py:108-126).

One process per GPU (``torch.distributed``, backend "nccl" = RCCL over xGMI; "gloo" in the CPU tests).  Every rank calls
``forward_sharded`` with the SAME global batch; rank r runs the whole path on its contiguous block of images
(``collate.shard_images``) with replicated weights, and one exchange at the end (``collate.allgather_persons``) gives every rank the
reference's person list for the whole batch, in the (b, y, x) order of the unsharded run (model.py:146-149).  No other collective
is on the path.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

from . import collate


def person_fields(model) -> list:
    """Record layout of the model's person dicts (reference model.py:330-346) -- needed by a rank that detected nobody."""
    nb = int(getattr(model, "num_betas", 10))
    V = int(model.smpl_layer[f"neutral_{nb}"].bm_x.num_vertices) if hasattr(model, "smpl_layer") else 10475
    return [("scores", ()), ("loc", (2,)), ("transl", (3,)), ("transl_pelvis", (1, 3)), ("rotvec", (53, 3)), ("expression", (10,)),
            ("shape", (nb,)), ("j3d", (127, 3)), ("j2d", (127, 2)), ("v3d", (V, 3))]


@torch.no_grad()
def forward_sharded(model, x, K, det_thresh=0.3, nms_kernel_size=3, group=None, device=None, return_image_index=False, fields=None):
    """x: [B, 3, S, S], K: [B, 3, 3] -- the GLOBAL batch, identical on every rank (any device).  Returns the list of per-person
    dicts of the whole batch (empty list if nobody is detected anywhere), identical on every rank; with ``return_image_index`` also
    the image id [P] of every person.  Without an initialised process group this is ``model(x, K=K, ...)``."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    B = x.shape[0]
    imgs = collate.shard_images(B, rank, world)
    fields = person_fields(model) if fields is None else fields
    if device is None:
        device = next(model.parameters()).device if hasattr(model, "parameters") else x.device
    # the local shard as BATCHED tensors [P_local, ...] (what the model has anyway: no person list -> torch.stack round trip)
    batched = {k: torch.zeros(0, *shp, dtype=torch.float32, device=device) for k, shp in fields}
    local_ids = torch.zeros(0, dtype=torch.int32, device=device)
    if len(imgs) > 0:
        xs, Ks = x[imgs.start:imgs.stop].to(device), K[imgs.start:imgs.stop].to(device)
        b, local_ids = _forward_batched(model, xs, Ks, det_thresh, nms_kernel_size, fields, device)
        if b:
            batched = b
    out, image_index = collate.allgather_persons(batched, image_offset=imgs.start, image_index=local_ids, group=group, fields=fields)
    humans = collate.persons_from_batched(out, fields)
    return (humans, image_index) if return_image_index else humans


def _forward_batched(model, xs, Ks, det_thresh, nms_kernel_size, fields, device):
    """Inference on the local shard -> (dict of [P, ...] tensors or {} when nobody is detected, local image id [P] int32).  The
    reference's person dicts carry no image id (only their order does): ``multi_hmr_amd.Model`` reports them together with the batched
    tensors (``return_batched=True``); any other callable with the reference's signature is run one image at a time and its person
    list is stacked."""
    if getattr(model, "supports_batched", False):
        batched, ids = model(xs, K=Ks, det_thresh=det_thresh, nms_kernel_size=nms_kernel_size, return_batched=True)
        return batched, ids.to(torch.int32)
    if getattr(model, "supports_image_index", False):
        persons, ids = model(xs, K=Ks, det_thresh=det_thresh, nms_kernel_size=nms_kernel_size, return_image_index=True)
        return (collate.batched_from_persons(persons, fields, device) if persons else {}), ids.to(torch.int32)
    ids, persons = [], []
    for b in range(xs.shape[0]):
        pb = model(xs[b:b + 1], K=Ks[b:b + 1], det_thresh=det_thresh, nms_kernel_size=nms_kernel_size)
        persons += pb
        ids += [b] * len(pb)
    return (collate.batched_from_persons(persons, fields, device) if persons else {}), torch.tensor(ids, dtype=torch.int32, device=xs.device)
