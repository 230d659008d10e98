#!/usr/bin/env python
"""How much of the max-norm error of `rotmat` is the 6D -> rotation decode (review item 8 of round 5; utils/humans.py:12-22)?

Runs the CPU fp32 oracle on a full-size golden case (default vitl_896_full, the benchmark's size), captures the 6D read-out that goes into
the Gram-Schmidt decode, and measures IN FP64, per (person, joint), the amplification of the decode: a perturbation delta of the 6D vector
with relative size eps (|delta|_inf = eps * |6D|_inf over the whole tensor -- the normalisation of tests/parity.py's max norm) changes
the rotation matrix by |dR|_inf = amp * eps.  Gram-Schmidt divides by |a1| and by |a2 - (b1.a2) b1|: a joint whose 6D columns are short or
nearly parallel amplifies.  Output: the distribution of those two lengths, the per-joint amplification (analytic bound and sampled), and
what rel-L2 / max-norm rotmat errors a given rel-L2 error of the 6D read-out turns into.
   python tools/rotmat_amplification.py [case] > profiles/r06_rotmat_amplification.txt            (CPU only; a few minutes for ViT-L 896^2)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import synthetic  # noqa: E402
from oracle import multihmr_ref, roma_ref  # noqa: E402


def gs(M):
    x, y = M[..., 0], M[..., 1]
    x = x / x.norm(dim=-1, keepdim=True)
    y = y - (x * y).sum(-1, keepdim=True) * x
    y = y / y.norm(dim=-1, keepdim=True)
    return torch.stack((x, y, torch.cross(x, y, dim=-1)), dim=-1)


def main():
    import make_golden
    name = sys.argv[1] if len(sys.argv) > 1 else "vitl_896_full"
    cfg = make_golden.CASES[name]
    gold = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    x, K, idx = make_golden.case_inputs(cfg)
    captured = {}
    orig = roma_ref.special_gramschmidt

    def spy(M, epsilon=0.0):
        captured["M"] = M.detach().clone()
        return orig(M, epsilon)
    roma_ref.special_gramschmidt = spy
    multihmr_ref.roma_ref.special_gramschmidt = spy
    model = multihmr_ref.OracleModel(make_golden.case_state_dict(cfg), synthetic.make_smplx_data(0), backbone=cfg["backbone"],
                                     img_size=cfg["img_size"], depth_override=cfg["depth_override"])
    out = model.forward(x, idx=idx, K=K, is_training=True)
    M = captured["M"].double()                      # [P * 53, 3, 2]
    P = out["rotmat"].shape[0]
    R0 = gs(M)
    print(f"# {name}: {P} persons x 53 joints; oracle rotmat vs golden: {float((out['rotmat'].double() - torch.from_numpy(gold['rotmat']).double()).abs().max()):.2e} (max abs)")
    a1 = M[..., 0].norm(dim=-1)
    b1 = M[..., 0] / a1[:, None]
    a2p = (M[..., 1] - (b1 * M[..., 1]).sum(-1, keepdim=True) * b1).norm(dim=-1)
    minf = float(M.abs().max())
    q = lambda t, p: float(torch.quantile(t, p))
    print(f"6D read-out: |.|_inf = {minf:.3f}, rms = {float(M.pow(2).mean().sqrt()):.3f}")
    print(f"|a1|                 : min {float(a1.min()):.3f}  1% {q(a1, .01):.3f}  median {q(a1, .5):.3f}  max {float(a1.max()):.3f}")
    print(f"|a2 - (b1.a2) b1|    : min {float(a2p.min()):.3f}  1% {q(a2p, .01):.3f}  median {q(a2p, .5):.3f}  max {float(a2p.max()):.3f}")
    # analytic first-order bound per joint: dR <= |d|/|a1| (column x), |d| (1/|a2p| + |a2|/(|a1| |a2p|)) (column y), sum for z
    bound = minf * (1.0 / a1 + 1.0 / a2p)
    # sampled amplification: 64 random directions per joint
    g = torch.Generator().manual_seed(0)
    eps = 1e-6
    amp = torch.zeros(M.shape[0], dtype=torch.float64)
    l2amp = []
    for _ in range(64):
        d = torch.randn(M.shape, generator=g, dtype=torch.float64)
        d = d / d.abs().max() * (eps * minf)                             # |delta|_inf = eps |6D|_inf, over the whole tensor
        dR = gs(M + d) - R0
        amp = torch.maximum(amp, dR.abs().amax(dim=(1, 2)) / eps)
        l2amp.append(float(dR.norm() / R0.norm()) / float(d.norm() / M.norm()))
    print(f"max-norm amplification per joint (|dR|_inf / (|d|_inf / |6D|_inf)), 64 random directions: median {q(amp, .5):.2f}  90% {q(amp, .9):.2f}  "
          f"99% {q(amp, .99):.2f}  max {float(amp.max()):.2f}   (first-order bound per joint: median {q(bound, .5):.2f}, max {float(bound.max()):.2f})")
    print(f"rel-L2 amplification (|dR|_2 / |R|_2) / (|d|_2 / |6D|_2): mean {np.mean(l2amp):.2f}")
    worst = torch.argsort(amp, descending=True)[:8]
    print("the eight most amplifying (person, joint): " + ", ".join(f"({int(i) // 53},{int(i) % 53}) amp {float(amp[i]):.1f} |a1| {float(a1[i]):.2f} |a2p| {float(a2p[i]):.2f}" for i in worst))
    # what a Gaussian read-out error of relative L2 size e turns into
    for e in (1.0e-4, 2.0e-4, 3.0e-4):
        mx, l2 = [], []
        for t in range(32):
            d = torch.randn(M.shape, generator=g, dtype=torch.float64)
            d = d / d.norm() * (e * float(M.norm()))
            dR = gs(M + d) - R0
            mx.append(float(dR.abs().max()))
            l2.append(float(dR.norm() / R0.norm()))
        print(f"a 6D read-out error of rel-L2 {e:.0e} (white) -> rotmat rel-L2 {np.mean(l2):.2e}, max norm {np.mean(mx):.2e} (worst of 32 draws {np.max(mx):.2e}); "
              f"6D max norm of the same draws ~ {e * float(M.norm()) / np.sqrt(M.numel()) * 4.0 / minf:.2e} (4 sigma)")


if __name__ == "__main__":
    main()
