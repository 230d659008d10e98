"""Image-sharded multi-GPU inference: the only exchange on the path.

Images are independent (reference Model.forward has no cross-image op, model.py:229-349), so rank r runs the
whole path on its own contiguous block of images and the detections are collated at the end with ONE
all-gather of per-rank person counts plus ONE all-gather of fixed-stride person records padded to the
maximum count (an all-gather-v by padding).  On ROCm backend "nccl" is RCCL over xGMI; the same code runs on
"gloo" for the CPU tests.  Global order = rank-major = the (b, y, x) order of the unsharded run.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

#: (key, trailing shape) of the per-person record, in the order of the reference's person dict (model.py:330-346)
RECORD = [("scores", ()), ("loc", (2,)), ("transl", (3,)), ("transl_pelvis", (1, 3)), ("rotvec", (53, 3)), ("expression", (10,)),
          ("shape", (10,)), ("j3d", (127, 3)), ("j2d", (127, 2)), ("v3d", (10475, 3))]


def record_width(fields=RECORD) -> int:
    w = 0
    for _, shp in fields:
        n = 1
        for s in shp:
            n *= s
        w += n
    return w


def pack_records(batched: dict, fields=RECORD) -> torch.Tensor:
    """dict of [P, ...] tensors -> [P, W] fp32 records."""
    P = batched[fields[0][0]].shape[0]
    return torch.cat([batched[k].reshape(P, -1).float() for k, _ in fields], dim=1).contiguous()


def unpack_records(rec: torch.Tensor, fields=RECORD) -> dict:
    out, o = {}, 0
    for k, shp in fields:
        n = 1
        for s in shp:
            n *= s
        out[k] = rec[:, o:o + n].reshape(rec.shape[0], *shp)
        o += n
    return out


def shard_images(num_images: int, rank: int, world: int) -> range:
    """Contiguous block of images owned by `rank` (earlier ranks take the remainder)."""
    q, r = divmod(num_images, world)
    lo = rank * q + min(rank, r)
    return range(lo, lo + q + (1 if rank < r else 0))


def allgather_persons(batched: dict, image_offset: int = 0, image_index: torch.Tensor | None = None, group=None, fields=RECORD):
    """Collate every rank's persons.  Returns (dict of [P_total, ...] tensors in global order, image_index [P_total]).

    ``image_index`` (local image id of each person) is shifted by ``image_offset`` so the result indexes the
    unsharded batch."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rec = pack_records(batched, fields)
    dev = rec.device
    if image_index is None:
        image_index = torch.zeros(rec.shape[0], dtype=torch.long, device=dev)
    rec = torch.cat([rec, (image_index.to(dev).float() + image_offset).unsqueeze(1)], dim=1)
    if not dist.is_initialized():          # single process without a process group; an initialised group of one still runs the collectives
        return unpack_records(rec[:, :-1], fields), rec[:, -1].long()
    counts = torch.zeros(world, dtype=torch.int64, device=dev)
    mine = torch.tensor([rec.shape[0]], dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(counts, mine, group=group)
    cl = counts.tolist()
    pmax = max(max(cl), 1)
    padded = torch.zeros(pmax, rec.shape[1], dtype=rec.dtype, device=dev)
    padded[: rec.shape[0]] = rec
    gathered = torch.empty(world * pmax, rec.shape[1], dtype=rec.dtype, device=dev)
    dist.all_gather_into_tensor(gathered, padded, group=group)
    parts = [gathered[r * pmax: r * pmax + cl[r]] for r in range(world)]
    allrec = torch.cat(parts, dim=0)
    return unpack_records(allrec[:, :-1], fields), allrec[:, -1].long()


class PendingGather:
    """Handle of ``allgather_persons_async``: ``wait()`` -> (dict of [P_total, ...] tensors in global order, image_index [P_total])."""

    def __init__(self, works, gathered, counts, capacity, world, fields):
        self._works, self._gathered, self._counts, self._cap, self._world, self._fields = works, gathered, counts, capacity, world, fields

    def wait(self):
        for w in self._works:
            w.wait()
        cl = self._counts.tolist()
        parts = [self._gathered[r * self._cap: r * self._cap + cl[r]] for r in range(self._world)]
        allrec = torch.cat(parts, dim=0)
        return unpack_records(allrec[:, :-1], self._fields), allrec[:, -1].long()


def allgather_persons_async(batched: dict, capacity: int, image_offset: int = 0, image_index: torch.Tensor | None = None, group=None,
                            fields=RECORD) -> PendingGather:
    """The same exchange without a host round trip in front of it: every rank pads its records to ``capacity`` persons (an upper
    bound the caller knows, e.g. images x max detections), and both collectives (counts, records) are enqueued with
    ``async_op=True`` so that RCCL moves this step's persons over xGMI while the next step's kernels run; the caller ``wait()``s
    later.  Needs an initialised process group."""
    world = dist.get_world_size(group)
    rec = pack_records(batched, fields)
    dev = rec.device
    if image_index is None:
        image_index = torch.zeros(rec.shape[0], dtype=torch.long, device=dev)
    assert rec.shape[0] <= capacity, (rec.shape[0], capacity)
    padded = torch.zeros(capacity, rec.shape[1] + 1, dtype=rec.dtype, device=dev)
    padded[: rec.shape[0], :-1] = rec
    padded[: rec.shape[0], -1] = image_index.to(dev).float() + image_offset
    counts = torch.zeros(world, dtype=torch.int64, device=dev)
    mine = torch.tensor([rec.shape[0]], dtype=torch.int64, device=dev)
    gathered = torch.empty(world * capacity, padded.shape[1], dtype=rec.dtype, device=dev)
    works = [dist.all_gather_into_tensor(counts, mine, group=group, async_op=True),
             dist.all_gather_into_tensor(gathered, padded, group=group, async_op=True)]
    return PendingGather(works, gathered, counts, capacity, world, fields)


def persons_from_batched(batched: dict, fields=RECORD) -> list:
    """[P, ...] tensors -> the reference's list of per-person dicts (model.py:329-347)."""
    P = batched[fields[0][0]].shape[0]
    return [{k: batched[k][i] for k, _ in fields} for i in range(P)]
