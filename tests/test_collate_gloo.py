"""The N > 1 path on CPU: gloo ranks, image-sharded, one all-gather-v of person records -> the same persons, in the same
(b, y, x) order, as the unsharded run.  Covers the REAL record (32 248 floats / person, SMPL-X 10 475 vertices), the int32 image-id
side gather, >= 3 pipelined asynchronous exchanges with different counts per step and a rank without persons, and
``distributed.forward_sharded`` end to end with a CPU stand-in for the model."""
import os

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from multi_hmr_amd import collate, distributed

SMALL = [("scores", ()), ("loc", (2,)), ("rotvec", (53, 3)), ("v3d", (97, 3))]


def _fake_persons(image_ids, fields, salt=0.0):
    """Deterministic per-person records keyed by (global image id, slot within the call)."""
    P = len(image_ids)
    base = torch.tensor(image_ids, dtype=torch.float32) * 1000 + torch.arange(P) + salt
    out = {}
    for j, (k, shp) in enumerate(fields):
        n = 1
        for s_ in shp:
            n *= s_
        out[k] = (base[:, None] + 0.25 * j + 1e-3 * torch.arange(n)[None, :]).reshape(P, *shp)
    return out


class _StandInModel:
    """Reference-signature callable on CPU: image b of the global batch carries its id in x[b, 0, 0, 0] and the number of persons
    to 'detect' in x[b, 0, 0, 1]; the persons it returns are a deterministic function of the GLOBAL image id."""
    supports_image_index = True
    num_betas = 10

    def __call__(self, x, K=None, det_thresh=0.3, nms_kernel_size=3, return_image_index=False, **kw):
        gids = [int(v) for v in x[:, 0, 0, 0].tolist()]
        counts = [int(v) for v in x[:, 0, 0, 1].tolist()]
        ids = [b for b, c in enumerate(counts) for _ in range(c)]
        glob = [gids[b] for b in ids]
        batched = _fake_persons(glob, collate.RECORD)
        batched["scores"] = batched["scores"] + K[ids, 0, 0] * 0          # K is sliced with x
        persons = collate.persons_from_batched(batched, collate.RECORD)
        return (persons, torch.tensor(ids, dtype=torch.int32)) if return_image_index else persons


def _worker(rank, world, port, counts, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    imgs = collate.shard_images(len(counts), rank, world)
    local_ids = [b - imgs.start for b in imgs for _ in range(counts[b])]
    glob_ids = [b for b in imgs for _ in range(counts[b])]
    # (1) small records, synchronous == asynchronous
    batched = _fake_persons(glob_ids, SMALL)
    out, img = collate.allgather_persons(batched, image_offset=imgs.start, image_index=torch.tensor(local_ids, dtype=torch.long), fields=SMALL)
    pend = collate.allgather_persons_async(batched, capacity=8, image_offset=imgs.start, image_index=torch.tensor(local_ids, dtype=torch.long),
                                           fields=SMALL)
    out2, img2 = pend.wait()
    assert torch.equal(img2, img) and all(torch.equal(out2[k], out[k]) for k in out)
    # (2) the real 32 248-float record, three exchanges in flight at once, counts differ per step, one rank empty in step 1
    steps = []
    for s_ in range(3):
        cs = [(c + s_ * (b + 1)) % 4 for b, c in enumerate(counts)]
        if s_ == 1:
            cs = [0 if b in imgs and rank == 1 else c for b, c in enumerate(cs)]        # rank 1 detects nobody
            cs_all = [0 if b in collate.shard_images(len(counts), 1, world) else c for b, c in enumerate(cs)]
        else:
            cs_all = cs
        lid = [b - imgs.start for b in imgs for _ in range(cs_all[b])]
        gid = [b for b in imgs for _ in range(cs_all[b])]
        bt = _fake_persons(gid, collate.RECORD, salt=0.5 * s_)
        steps.append((cs_all, collate.allgather_persons_async(bt, capacity=16, image_offset=imgs.start,      # the SAME capacity on every rank
                                                             image_index=torch.tensor(lid, dtype=torch.long))))
    results = []
    for cs_all, pend in steps:                    # waited only after all three were enqueued
        o, im = pend.wait()
        results.append((cs_all, {k: v.clone() for k, v in o.items()}, im.clone()))
    # (3) forward_sharded with the stand-in model: global batch on every rank
    B = len(counts)
    x = torch.zeros(B, 3, 4, 4)
    x[:, 0, 0, 0] = torch.arange(B).float()
    x[:, 0, 0, 1] = torch.tensor(counts).float()
    K = torch.eye(3).repeat(B, 1, 1)
    humans, himg = distributed.forward_sharded(_StandInModel(), x, K, device=torch.device("cpu"), return_image_index=True, fields=collate.RECORD)
    nobody = distributed.forward_sharded(_StandInModel(), x * 0, K, device=torch.device("cpu"), fields=collate.RECORD)
    if rank == 0:          # by value (numpy): torch tensors would travel as shared-memory handles that die with this process
        npy = lambda d: {k: v.numpy().copy() for k, v in d.items()}
        q.put((npy(out), img.numpy().copy(), [(c, npy(o), im.numpy().copy()) for c, o, im in results], [npy(h) for h in humans],
               himg.numpy().copy(), len(nobody)))
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
    out, img, results, humans, himg, n_nobody = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    t = torch.from_numpy
    out, img, himg = {k: t(v) for k, v in out.items()}, t(img), t(himg)
    results = [(c, {k: t(v) for k, v in o.items()}, t(im)) for c, o, im in results]
    humans = [{k: t(v) for k, v in h.items()} for h in humans]
    glob_ids = [b for b, c in enumerate(counts) for _ in range(c)]
    assert img.tolist() == glob_ids and img.dtype == torch.int64
    assert list(collate.shard_images(7, 0, 2)) == [0, 1, 2, 3] and list(collate.shard_images(7, 1, 2)) == [4, 5, 6]
    # records: the slot index restarts per rank, the image id is global
    exp = torch.cat([_fake_persons([b for b in rng for _ in range(counts[b])], SMALL)["scores"] for rng in (range(0, 4), range(4, 7))])
    assert torch.equal(out["scores"], exp)
    assert out["v3d"].shape == (len(glob_ids), 97, 3)
    # pipelined real-size exchanges: each step's collation equals the rank-major concatenation of what the ranks sent
    for s_, (cs_all, o, im) in enumerate(results):
        gid = [b for b, c in enumerate(cs_all) for _ in range(c)]
        assert im.tolist() == gid, s_
        parts = [_fake_persons([b for b in rng for _ in range(cs_all[b])], collate.RECORD, salt=0.5 * s_) for rng in (range(0, 4), range(4, 7))]
        for k, shp in collate.RECORD:
            assert torch.equal(o[k], torch.cat([p[k] for p in parts])), (s_, k)
        assert o["v3d"].shape[1:] == (10475, 3)
    assert sum(results[1][0][4:]) == 0 and sum(results[1][0][:4]) > 0          # step 1: rank 1 really was empty
    # forward_sharded == the unsharded run of the same stand-in model
    x = torch.zeros(7, 3, 4, 4)
    x[:, 0, 0, 0] = torch.arange(7).float()
    x[:, 0, 0, 1] = torch.tensor(counts).float()
    ref, rid = _StandInModel()(x, K=torch.eye(3).repeat(7, 1, 1), return_image_index=True)
    assert himg.tolist() == rid.tolist() == glob_ids and len(humans) == len(ref) == 11
    for h, r in zip(humans, ref):
        assert list(h.keys()) == list(r.keys()) == collate.PERSON_KEYS
        # the sharded run restarts the person slot per rank; the stand-in encodes (image id, slot) -> compare the image-id part
        assert all(h[k].shape == r[k].shape for k in h)
        assert int(h["scores"].item()) // 1000 == int(r["scores"].item()) // 1000
    assert n_nobody == 0


def test_single_process_passthrough():
    b = _fake_persons([0, 0, 1], SMALL)
    out, img = collate.allgather_persons(b, image_offset=5, image_index=torch.tensor([0, 0, 1]), fields=SMALL)
    assert img.tolist() == [5, 5, 6] and torch.equal(out["rotvec"], b["rotvec"])
    assert collate.record_width() == 1 + 2 + 3 + 3 + 159 + 10 + 10 + 381 + 254 + 31425 == 32248
    # the record layout follows the tensors: 11 betas, another vertex count
    b2 = {k: torch.zeros(2, *shp) for k, shp in collate.RECORD}
    b2["shape"], b2["v3d"] = torch.zeros(2, 11), torch.zeros(2, 500, 3)
    f = collate.fields_of(b2)
    assert dict(f)["shape"] == (11,) and dict(f)["v3d"] == (500, 3)
    o, _ = collate.allgather_persons(b2)
    assert o["shape"].shape == (2, 11) and o["v3d"].shape == (2, 500, 3)
    humans = distributed.forward_sharded(_StandInModel(), torch.tensor([[[[3.0, 2.0]]]]).expand(1, 3, 1, 2), torch.eye(3)[None],
                                         device=torch.device("cpu"), fields=collate.RECORD)
    assert len(humans) == 2 and humans[0]["v3d"].shape == (10475, 3)
