"""The N > 1 path on CPU: two gloo ranks, image-sharded, one all-gather-v of person records -> the same persons,
in the same (b, y, x) order, as the unsharded run."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from multi_hmr_amd import collate

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIELDS = [("scores", ()), ("loc", (2,)), ("rotvec", (53, 3)), ("v3d", (97, 3))]


def _fake_persons(image_ids):
    """Deterministic per-person records keyed by (global image id, slot)."""
    P = len(image_ids)
    g = torch.tensor(image_ids, dtype=torch.float32)
    base = g * 1000 + torch.arange(P)
    return {"scores": base.clone(), "loc": base[:, None] + torch.tensor([0.25, 0.5]), "rotvec": base[:, None, None] + torch.rand(1, 53, 3) * 0,
            "v3d": base[:, None, None] * torch.ones(P, 97, 3)}


def _worker(rank, world, port, counts, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    imgs = collate.shard_images(len(counts), rank, world)
    local_ids = [b - imgs.start for b in imgs for _ in range(counts[b])]
    glob_ids = [b for b in imgs for _ in range(counts[b])]
    batched = _fake_persons(glob_ids)
    out, img = collate.allgather_persons(batched, image_offset=imgs.start, image_index=torch.tensor(local_ids, dtype=torch.long), fields=FIELDS)
    pend = collate.allgather_persons_async(batched, capacity=8, image_offset=imgs.start, image_index=torch.tensor(local_ids, dtype=torch.long),
                                           fields=FIELDS)
    out2, img2 = pend.wait()                      # the fixed-capacity asynchronous exchange gives the same collation
    assert torch.equal(img2, img) and all(torch.equal(out2[k], out[k]) for k in out)
    if rank == 0:
        q.put(({k: v.clone() for k, v in out.items()}, img.clone()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_allgather_matches_unsharded_order():
    counts = [2, 0, 3, 1, 0, 4, 1]          # persons per image; rank 0 gets 4 images (6 persons), rank 1 gets 3 (5 persons)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, counts, q)) for r in range(2)]
    for p in procs:
        p.start()
    out, img = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    glob_ids = [b for b, c in enumerate(counts) for _ in range(c)]
    assert img.tolist() == glob_ids
    assert list(collate.shard_images(7, 0, 2)) == [0, 1, 2, 3] and list(collate.shard_images(7, 1, 2)) == [4, 5, 6]
    # records: the slot index restarts per rank, the image id is global
    exp = torch.cat([_fake_persons([b for b in rng for _ in range(counts[b])])["scores"] for rng in (range(0, 4), range(4, 7))])
    assert torch.equal(out["scores"], exp)
    assert out["v3d"].shape == (len(glob_ids), 97, 3) and torch.equal(out["v3d"][:, 0, 0], exp)
    persons = collate.persons_from_batched(out, FIELDS)
    assert len(persons) == 11 and persons[3]["loc"].shape == (2,)


def test_single_process_passthrough():
    b = _fake_persons([0, 0, 1])
    out, img = collate.allgather_persons(b, image_offset=5, image_index=torch.tensor([0, 0, 1]), fields=FIELDS)
    assert img.tolist() == [5, 5, 6] and torch.equal(out["rotvec"], b["rotvec"])
    assert collate.record_width() == 1 + 2 + 3 + 3 + 159 + 10 + 10 + 381 + 254 + 31425
