"""-m gpu, needs >= 2 GPUs (skipped on a 1-GPU box): the image-sharded path over RCCL.

Two ranks, one per GPU (``torch.multiprocessing.spawn``, backend "nccl" = RCCL over xGMI, rendezvous on 127.0.0.1):
``distributed.forward_sharded`` on the global batch must return, on every rank, the person list of the unsharded run -- same
persons, same (b, y, x) order (reference model.py:329-347), same values -- and the asynchronous fixed-capacity collation must deliver
the same records while the next forward is running.  The CPU twin of this test (gloo) is tests/test_collate_gloo.py."""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rank_main(rank, world, port, out_path):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "tests", "golden"))
    import make_golden
    import synthetic
    import torch.distributed as dist
    from multi_hmr_amd import Model, collate, distributed
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    try:
        # the group really has `world` ranks on `world` different GPUs: a silent single-rank fallback (every rank its own group of one)
        # would pass every comparison below against its own unsharded run
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)
        ids = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
        dist.all_gather(ids, torch.tensor([torch.cuda.current_device()], device=dev))
        assert dist.get_world_size() == world and float(ones.item()) == float(world), (dist.get_world_size(), float(ones.item()))
        assert sorted(int(i.item()) for i in ids) == list(range(world)), ids
        cfg = dict(make_golden.CASES["vits_448_infer"], batch=5)          # 5 images over 2 ranks: 3 + 2
        gold = np.load(os.path.join(GOLD, "vits_448_infer.npz"))
        sd = make_golden.case_state_dict(cfg)
        sd["mlp_classif.2.bias"] = torch.from_numpy(gold["classif_bias"])
        model = Model(backbone=cfg["backbone"], img_size=cfg["img_size"], smplx_data=synthetic.make_smplx_data(0),
                      mean_params=synthetic.make_mean_params(0), backbone_depth=cfg["depth_override"], precision="f16")
        model.load_state_dict(sd, strict=True)
        model = model.to(dev).eval()
        x, K, _ = make_golden.case_inputs(cfg)
        kw = dict(det_thresh=float(gold["det_thresh"]), nms_kernel_size=cfg["nms_kernel_size"])
        humans, img = distributed.forward_sharded(model, x, K, return_image_index=True, **kw)
        whole, wimg = model(x.to(dev), K=K.to(dev), return_image_index=True, **kw)        # the unsharded run, on this rank's GPU
        ok = len(humans) == len(whole) > 0 and img.tolist() == wimg.tolist()
        worst = 0.0
        for p, q in zip(humans, whole):
            for k in p:
                worst = max(worst, float((p[k] - q[k]).abs().max()))
        # asynchronous fixed-capacity collation with a forward enqueued behind it
        imgs = collate.shard_images(x.shape[0], rank, world)
        batched, ids = model(x[imgs.start:imgs.stop].to(dev), K=K[imgs.start:imgs.stop].to(dev), return_batched=True, **kw)
        pend = collate.allgather_persons_async(batched, capacity=64, image_offset=imgs.start, image_index=ids)
        model(x[imgs.start:imgs.stop].to(dev), K=K[imgs.start:imgs.stop].to(dev), **kw)     # runs while the exchange travels
        got, gimg = pend.wait()
        ok_async = gimg.tolist() == wimg.tolist() and all(
            float((got[k] - torch.stack([h[k] for h in whole])).abs().max()) == 0.0 for k in ("v3d", "scores", "rotvec"))
        if rank == 0:
            torch.save(dict(ok=bool(ok), worst=worst, ok_async=bool(ok_async), n=len(humans), ranks=int(ones.item())), out_path)
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (the round's box has one)")
def test_forward_sharded_over_rccl_two_ranks(tmp_path):
    import torch.multiprocessing as mp
    out = str(tmp_path / "res.pt")
    mp.spawn(_rank_main, args=(2, _free_port(), out), nprocs=2, join=True)
    res = torch.load(out)
    assert res["ok"] and res["n"] > 0 and res["ranks"] == 2, res
    # sharding changes nothing: every kernel is batch-invariant, the collation moves fp32 records
    assert res["worst"] == 0.0, res
    assert res["ok_async"], res
