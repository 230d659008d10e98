#!/usr/bin/env python
"""Headline benchmark: images/s of the whole Multi-HMR forward (ViT-L 896x896, batch 32 per GPU, 8 pinned
queries per image -> HPH -> SMPL-X LBS), BASELINE.json config #4 (`multiHMR_896_L`, image-sharded over N GPUs).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" = one full `Model.forward` over one batch of synthetic images already resident in HBM (seeded
N(0,1) pixels, seeded random weights of the named architecture, synthetic SMPL-X arrays), detections pinned to
8 per image through the reference's own `idx=` / `is_training=True` hook (model.py:141-151), plus -- for N > 1 --
the RCCL all-gather that collates every rank's persons.  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from multi_hmr_amd import Model, _lib, collate, synthetic  # noqa: E402

PEAK_MFMA_TFLOPS = 2500.0     # bf16/f16 dense, /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters"
PEAK_HBM_GBS = 8000.0         # HBM3E spec, same table (6.29 TB/s measured float4 copy)


def flops_per_image(S, C, L, heads_depth=2, inner=256):
    """algorithmic FLOPs of one image (T = N + 1 real tokens, no padding)."""
    G = S // 14
    N, T = G * G, G * G + 1
    gemm = 2 * N * 588 * C + L * 24 * T * C * C + 2 * N * C * C + heads_depth * 2 * N * (C + 99) * 2 * inner
    attn = L * 4 * T * T * C
    return gemm, attn


def lbs_bytes(P, nb=10):
    """Algorithmic HBM bytes of one LBS launch: constants once + per-person in/out."""
    const = 10475 * 3 * 4 * (1 + nb + 10 + 486) + 55 * 3 * (nb + 11) * 4 + 10475 * 4 * 8
    return const + P * (764 + 10475 * 3 * 4 + 10475 * 2 * 4 + 127 * 5 * 4)


def pmc_traffic():
    """HBM/fabric bytes per GEMM launch (average over the ViT-L 896 b32 launches) from the committed rocprofv3 PMC passes
    (FETCH_SIZE x2 + WRITE_SIZE, separate --pmc runs, tools/pmc_traffic.py -> profiles/r01_v6_pmc.json); counters
    cannot be read from inside the timed process, so this is null when the summary is absent."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_v6_pmc.json")) as f:
            return json.load(f).get("_gemm_avg_bytes_per_launch")
    except (OSError, ValueError):
        return None


def prof_window(kind):
    _lib.check(_lib.lib().mhmr_prof_enable(kind), "prof_enable")


def prof_collect():
    n, ms, work = C.c_int(0), C.c_double(0), C.c_double(0)
    _lib.check(_lib.lib().mhmr_prof_collect(C.byref(n), C.byref(ms), C.byref(work)), "prof_collect")
    _lib.lib().mhmr_prof_enable(-1)
    return n.value, ms.value, work.value


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16"])
    ap.add_argument("--backbone", default="dinov2_vitl14")
    ap.add_argument("--img-size", type=int, default=896)
    ap.add_argument("--batch", type=int, default=32, help="images per GPU")
    ap.add_argument("--persons", type=int, default=8, help="pinned detections per image")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the attention / LBS roofline side measurements")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local)
        torch.distributed.init_process_group("nccl", device_id=torch.device("cuda", local))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    S, B, q = args.img_size, args.batch, args.persons
    default_workload = (args.backbone, S, B, args.dtype) == ("dinov2_vitl14", 896, 32, "bf16")   # what the committed PMC pass measured
    cfg = synthetic.VIT_CFG[args.backbone]
    smplx_data, mean_params = synthetic.make_smplx_data(0), synthetic.make_mean_params(0)
    model = Model(backbone=args.backbone, img_size=S, smplx_data=smplx_data, mean_params=mean_params, precision=args.dtype)
    model.load_state_dict(synthetic.make_state_dict(args.backbone, S, seed=0, mean_params=mean_params), strict=True)
    model = model.to(dev).eval()

    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    x = torch.randn(B, 3, S, S, generator=g, device=dev)          # random, not zero-filled (DVFS, MICROARCH "DVFS give-back")
    K = synthetic.get_camera_K(S, B).to(dev)
    idx = tuple(t.to(dev) for t in synthetic.make_pinned_idx(B, S // 14, q, seed=rank))

    pending = []          # N > 1: the collation of step i travels over xGMI while step i + 1 computes (waited one step later)

    def step():
        out = model(x, idx=idx, K=K, is_training=True)
        if world > 1:
            out["scores"] = out["scores"][idx[0], idx[1], idx[2], 0]      # per-person score slot of the record
            pending.append(collate.allgather_persons_async(out, capacity=B * q, image_offset=rank * B, image_index=idx[0]))
            if len(pending) > 1:
                pending.pop(0).wait()
        return out

    def drain():
        while pending:
            pending.pop(0).wait()

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    drain()
    barrier()
    prof_window(0)                       # hipEvent brackets around every GEMM launch of the timed region
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    drain()               # every step's persons are collated on every rank inside the timed region
    barrier()
    dt = time.perf_counter() - t0
    n_gemm, ms_gemm, _ = prof_collect()
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())

    gemm_fl, attn_fl = flops_per_image(S, cfg["embed_dim"], cfg["depth"])
    ms_step = 1e3 * dt / args.steps
    value = world * B * args.steps / dt
    gemm_tf = gemm_fl * B * args.steps / (ms_gemm * 1e-3) / 1e12 if ms_gemm > 0 else 0.0
    result = {
        "metric": f"images/sec (whole node) ViT-{args.backbone[-3].upper()} {S}x{S} bs{B}", "value": round(value, 3), "unit": "images/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_step, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": args.dtype, "data": "synthetic (seeded N(0,1) images, random-init weights, synthetic SMPL-X)",
        "config": {"workload": f"multiHMR_{S}_{args.backbone[-3].upper()} full forward: {args.backbone} {S}x{S}, {B} images/GPU, "
                               f"{q} pinned persons/image -> HPH (depth 2) -> SMPL-X LBS; image-sharded x{world}",
                   "global_batch": world * B, "parallelism": f"dp{world} (images)"},
        "mfma_utilisation_whole_forward": round((gemm_fl + attn_fl) * B * args.steps / dt / 1e12 / PEAK_MFMA_TFLOPS, 4),
        "roofline": {"kernel": "gemm256_kernel (persistent 256x256x64 8-phase, v_mfma_f32_16x16x32; all ViT linears + heads)",
                     "bound": "mfma", "achieved": round(gemm_tf, 1), "peak": PEAK_MFMA_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(gemm_tf / PEAK_MFMA_TFLOPS, 4), "traffic": pmc_traffic() if default_workload else None,
                     "launches": n_gemm, "avg_launch_ms": round(ms_gemm / max(n_gemm, 1), 4),
                     "algorithmic_flops_per_launch": round(gemm_fl * B * args.steps / max(n_gemm, 1))},
    }

    if rank == 0 and not args.no_extras:
        # side measurements outside the timed region: attention kernel and the LBS vertex kernel (config #5: 160 persons)
        prof_window(1)
        model(x, idx=idx, K=K, is_training=True)
        n_a, ms_a, _ = prof_collect()
        att_tf = attn_fl * B / (ms_a * 1e-3) / 1e12 if ms_a > 0 else 0.0
        result["roofline_attention"] = {"kernel": "attn_kernel (flash, d=64)", "bound": "mfma", "achieved": round(att_tf, 1),
                                        "peak": PEAK_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(att_tf / PEAK_MFMA_TFLOPS, 4),
                                        "launches": n_a, "avg_launch_ms": round(ms_a / max(n_a, 1), 4)}
        result["lbs"] = lbs_bench(model, dev, P=160)
    if world > 1:
        torch.distributed.barrier()

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(args, smplx_data, mean_params)
    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        torch.distributed.destroy_process_group()


def lbs_bench(model, dev, P=160, iters=20):
    """ms/person of the SMPL-X layer alone (BASELINE.json second metric; config #5 = 8 images x 20 persons)."""
    L = _lib.lib()
    Pk = model._packed
    lb, cs = Pk["lbs"], Pk["lbs_struct"]
    g = torch.Generator(device=dev).manual_seed(5)
    f = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
    pose = 0.3 * torch.randn(P, 53, 3, generator=g, device=dev)
    shape, expr = torch.randn(P, 10, generator=g, device=dev), torch.randn(P, 10, generator=g, device=dev)
    loc, dist = 1288 * torch.rand(P, 2, generator=g, device=dev), 2 + 6 * torch.rand(P, 1, generator=g, device=dev)
    K = synthetic.get_camera_K(1288, 8).to(dev)
    det_b = torch.arange(P, device=dev, dtype=torch.int32) // 20
    V = lb["V"]
    bufs = [f((P + 15) // 16 * 16, lb["Kb"]), f(P, 55, 12), f(P, 24), f(P, V, 3), f(P, V, 2), f(P, 127, 3), f(P, 127, 2), f(P, 3)]
    stream = torch.cuda.current_stream(dev).cuda_stream

    def run():
        _lib.check(L.mhmr_lbs_forward(C.byref(cs), pose.data_ptr(), shape.data_ptr(), expr.data_ptr(), loc.data_ptr(), dist.data_ptr(),
                                      K.data_ptr(), det_b.data_ptr(), P, *[b.data_ptr() for b in bufs], stream), "lbs")
    for _ in range(3):
        run()
    torch.cuda.synchronize(dev)
    prof_window(2)
    t0 = time.perf_counter()
    for _ in range(iters):
        run()
    torch.cuda.synchronize(dev)
    wall = (time.perf_counter() - t0) / iters
    n, ms, _ = prof_collect()
    avg = ms / max(n, 1) * 1e-3
    gbs = lbs_bytes(P) / avg / 1e9 if avg > 0 else 0.0
    return {"persons": P, "ms_per_person": round(1e3 * wall / P, 6), "layer_ms": round(1e3 * wall, 4),
            "roofline": {"kernel": "lbs_vertex_kernel", "bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                         "frac": round(gbs / PEAK_HBM_GBS, 4), "traffic": None, "avg_launch_ms": round(avg * 1e3, 4)}}


def cpu_baseline(args, smplx_data, mean_params):
    """The oracle (reference algorithm restated, CPU fp32, all host cores) on a bounded sample of the same workload:
    ONE image of the same architecture / resolution with the same number of pinned persons."""
    from oracle.multihmr_ref import OracleModel
    S, q = args.img_size, args.persons
    sd = synthetic.make_state_dict(args.backbone, S, seed=0, mean_params=mean_params)
    ref = OracleModel(sd, smplx_data, backbone=args.backbone, img_size=S)
    x = torch.randn(1, 3, S, S, generator=torch.Generator().manual_seed(1234))
    K = synthetic.get_camera_K(S, 1)
    idx = synthetic.make_pinned_idx(1, S // 14, q, seed=0)
    t0 = time.perf_counter()
    ref.forward(x, idx=idx, K=K, is_training=True)
    dt = time.perf_counter() - t0
    return {"value": round(1.0 / dt, 4), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"1 image {S}x{S} {args.backbone}, {q} persons, full forward incl. HPH + LBS, fp32 torch CPU, single run ({dt:.1f} s)"}


if __name__ == "__main__":
    main()
