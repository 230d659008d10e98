#!/usr/bin/env python
"""RCCL smoke for the collation path on however many GPUs the launcher gives (1 on a gpurun box):
python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port 29511 tools/nccl_smoke.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multi_hmr_amd import collate

rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29511")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(local)
dev = torch.device("cuda", local)
torch.distributed.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
P = 3 + rank
g = torch.Generator(device=dev).manual_seed(rank)
batched = {k: torch.randn(P, *shp, generator=g, device=dev) for k, shp in collate.RECORD}
out, img = collate.allgather_persons(batched, image_offset=rank * 4, image_index=torch.arange(P, device=dev) % 4)
torch.distributed.barrier(); torch.cuda.synchronize()
tot = sum(3 + r for r in range(world))
assert out["v3d"].shape == (tot, 10475, 3) and img.shape == (tot,), (out["v3d"].shape, img.shape)
lo = sum(3 + r for r in range(rank))
assert torch.equal(out["v3d"][lo: lo + P], batched["v3d"]) and torch.equal(out["scores"][lo: lo + P], batched["scores"])
pend = collate.allgather_persons_async(batched, capacity=16, image_offset=rank * 4, image_index=torch.arange(P, device=dev) % 4)
y = torch.randn(2048, 2048, device=dev) * 2.0          # other work enqueued while the collectives are in flight
out2, img2 = pend.wait()
assert torch.equal(img2, img) and all(torch.equal(out2[k], out[k]) for k in out)
if rank == 0:
    print(f"RCCL collation OK: world {world}, {tot} persons, record width {collate.record_width()} floats")
torch.distributed.destroy_process_group()
