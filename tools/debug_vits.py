"""Bisect a backbone mismatch: HIP ViT features vs the CPU oracle's, per depth / batch (debug aid)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
from multi_hmr_amd import Model
import synthetic
from oracle import dinov2_ref

name, S = sys.argv[1], int(sys.argv[2])
smplx_data, mean_params = synthetic.make_smplx_data(0), synthetic.make_mean_params(0)
for depth in [int(d) for d in sys.argv[3].split(",")]:
    for B in [int(b) for b in sys.argv[4].split(",")]:
        sd = synthetic.make_state_dict(name, S, seed=21, depth_override=depth, mean_params=mean_params)
        m = Model(backbone=name, img_size=S, smplx_data=smplx_data, mean_params=mean_params, precision="f16", backbone_depth=depth)
        m.load_state_dict(sd, strict=True)
        m = m.to("cuda:0").eval()
        x = torch.randn(B, 3, S, S, generator=torch.Generator().manual_seed(5))
        z = m.backbone_features(x.cuda()).cpu()
        enc = dinov2_ref.build(name, depth_override=depth)
        enc.load_state_dict({k[len("backbone.encoder."):]: v for k, v in sd.items() if k.startswith("backbone.encoder.")}, strict=True)
        with torch.no_grad():
            ref = enc.get_intermediate_layers(x, n=1, norm=True)[0] if hasattr(enc, "get_intermediate_layers") else enc(x)
        e = [float((z[b] - ref[b]).norm() / ref[b].norm()) for b in range(B)]
        print(f"{name} S={S} depth={depth} B={B} GEMM128={os.environ.get('MHMR_GEMM128')}: rel-L2 per image {['%.2e' % v for v in e]}", flush=True)
