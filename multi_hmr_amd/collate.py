"""Image-sharded multi-GPU inference: the only exchange on the path.

Images are independent (reference Model.forward has no cross-image op, model.py:229-349), so rank r runs the
whole path on its own contiguous block of images and the detections are collated at the end with ONE
all-gather of per-rank person counts plus ONE all-gather of fixed-stride person records padded to the
maximum count (an all-gather-v by padding).  On ROCm backend "nccl" is RCCL over xGMI; the same code runs on
"gloo" for the CPU tests.  Global order = rank-major = the (b, y, x) order of the unsharded run.

Records are fp32 rows; the image id of every person travels in its own int32 side gather (never as a float inside the record).
The record layout is derived from the tensors handed in (``fields_of``), so other body models / ``num_betas`` work unchanged;
``RECORD`` is the layout of the reference's person dict for SMPL-X with 10 betas (model.py:330-346), 129 KB per person.

The C ABI has no collective entry point (SURVEY.md section 8b had planned an ``mhmr_allgather_persons``): the exchange is two
``torch.distributed`` all-gathers on tensors PyTorch already owns, RCCL is reached through the process group the caller
initialised, and a second, library-private RCCL communicator would have to be bootstrapped beside it for no gain.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

#: (key, trailing shape) of the per-person record, in the order of the reference's person dict (model.py:330-346)
RECORD = [("scores", ()), ("loc", (2,)), ("transl", (3,)), ("transl_pelvis", (1, 3)), ("rotvec", (53, 3)), ("expression", (10,)),
          ("shape", (10,)), ("j3d", (127, 3)), ("j2d", (127, 2)), ("v3d", (10475, 3))]
PERSON_KEYS = [k for k, _ in RECORD]


def fields_of(batched: dict, keys=PERSON_KEYS) -> list:
    """Record layout [(key, trailing shape)] of a dict of [P, ...] tensors (P may be 0)."""
    return [(k, tuple(batched[k].shape[1:])) for k in keys]


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
    for k, shp in fields:
        assert tuple(batched[k].shape[1:]) == tuple(shp), (k, tuple(batched[k].shape), shp)
    width = lambda shp: int(torch.tensor(shp).prod()) if len(shp) else 1
    return torch.cat([batched[k].reshape(P, width(shp)).float() for k, shp in fields], dim=1).contiguous()      # (P may be 0)


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


def _ids(image_index, n, offset, dev):
    if image_index is None:
        image_index = torch.zeros(n, dtype=torch.int32, device=dev)
    return (image_index.to(device=dev, dtype=torch.int32) + int(offset)).contiguous()


def allgather_persons(batched: dict, image_offset: int = 0, image_index: torch.Tensor | None = None, group=None, fields=None):
    """Collate every rank's persons.  Returns (dict of [P_total, ...] tensors in global order, image_index [P_total] int64).

    ``image_index`` (local image id of each person) is shifted by ``image_offset`` so the result indexes the
    unsharded batch.  ``fields`` defaults to the layout of ``batched`` itself; every rank must pass the same layout."""
    fields = fields_of(batched) if fields is None else fields
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rec = pack_records(batched, fields)
    dev = rec.device
    ids = _ids(image_index, rec.shape[0], image_offset, dev)
    if not dist.is_initialized():          # single process without a process group; an initialised group of one still runs the collectives
        return unpack_records(rec, fields), ids.long()
    counts = torch.zeros(world, dtype=torch.int64, device=dev)
    mine = torch.tensor([rec.shape[0]], dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(counts, mine, group=group)
    cl = counts.tolist()
    pmax = max(max(cl), 1)
    padded = torch.zeros(pmax, rec.shape[1], dtype=rec.dtype, device=dev)
    padded[: rec.shape[0]] = rec
    pids = torch.zeros(pmax, dtype=torch.int32, device=dev)
    pids[: rec.shape[0]] = ids
    gathered = torch.empty(world * pmax, rec.shape[1], dtype=rec.dtype, device=dev)
    gids = torch.empty(world * pmax, dtype=torch.int32, device=dev)
    dist.all_gather_into_tensor(gathered, padded, group=group)
    dist.all_gather_into_tensor(gids, pids, group=group)
    keep = torch.cat([torch.arange(r * pmax, r * pmax + cl[r], device=dev) for r in range(world)]) if sum(cl) else torch.zeros(0, dtype=torch.long, device=dev)
    return unpack_records(gathered[keep], fields), gids[keep].long()


class PendingGather:
    """Handle of ``allgather_persons_async``: ``wait()`` -> (dict of [P_total, ...] tensors in global order, image_index [P_total]).
    It owns every buffer of its exchange (send and receive side), so any number of exchanges may be in flight."""

    def __init__(self, works, gathered, gids, counts, capacity, world, fields, keepalive):
        self._works, self._gathered, self._gids, self._counts = works, gathered, gids, counts
        self._cap, self._world, self._fields, self._keepalive = capacity, world, fields, keepalive

    def wait(self):
        for w in self._works:
            w.wait()
        cl = self._counts.tolist()
        dev = self._gathered.device
        keep = (torch.cat([torch.arange(r * self._cap, r * self._cap + cl[r], device=dev) for r in range(self._world)]) if sum(cl)
                else torch.zeros(0, dtype=torch.long, device=dev))
        self._keepalive = None
        return unpack_records(self._gathered[keep], self._fields), self._gids[keep].long()


def allgather_persons_async(batched: dict, capacity: int, image_offset: int = 0, image_index: torch.Tensor | None = None, group=None,
                            fields=None) -> PendingGather:
    """The same exchange without a host round trip in front of it: every rank pads its records to ``capacity`` persons (an upper
    bound the caller knows, e.g. images x max detections), and the collectives (counts, records, image ids) are enqueued with
    ``async_op=True`` so that RCCL moves this step's persons over xGMI while the next step's kernels run; the caller ``wait()``s
    later.  Needs an initialised process group; ``capacity`` must be the same on every rank (the collectives are fixed-size)."""
    fields = fields_of(batched) if fields is None else fields
    world = dist.get_world_size(group)
    rec = pack_records(batched, fields)
    dev = rec.device
    n = rec.shape[0]
    assert n <= capacity, (n, capacity)
    padded = torch.zeros(capacity, rec.shape[1], dtype=rec.dtype, device=dev)
    padded[:n] = rec
    pids = torch.zeros(capacity, dtype=torch.int32, device=dev)
    pids[:n] = _ids(image_index, n, image_offset, dev)
    counts = torch.zeros(world, dtype=torch.int64, device=dev)
    mine = torch.tensor([n], dtype=torch.int64, device=dev)
    gathered = torch.empty(world * capacity, padded.shape[1], dtype=rec.dtype, device=dev)
    gids = torch.empty(world * capacity, dtype=torch.int32, device=dev)
    works = [dist.all_gather_into_tensor(counts, mine, group=group, async_op=True),
             dist.all_gather_into_tensor(gathered, padded, group=group, async_op=True),
             dist.all_gather_into_tensor(gids, pids, group=group, async_op=True)]
    return PendingGather(works, gathered, gids, counts, capacity, world, fields, (padded, pids, mine))


def persons_from_batched(batched: dict, fields=None) -> list:
    """[P, ...] tensors -> the reference's list of per-person dicts (model.py:329-347)."""
    fields = fields_of(batched) if fields is None else fields
    keys = [k for k, _ in fields]
    if batched[keys[0]].shape[0] == 0:
        return []
    # one unbind per key instead of P x len(keys) indexing calls
    return [dict(zip(keys, vals)) for vals in zip(*(batched[k].unbind(0) for k in keys))]


def batched_from_persons(persons: list, fields, device) -> dict:
    """The inverse (a rank without detections yields [0, ...] tensors of the right layout)."""
    if persons:
        return {k: torch.stack([p[k] for p in persons]) for k, _ in fields}
    return {k: torch.zeros(0, *shp, dtype=torch.float32, device=device) for k, shp in fields}
