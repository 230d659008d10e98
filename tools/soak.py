import torch, sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multi_hmr_amd import Model
import synthetic
sm, mp = synthetic.make_smplx_data(0), synthetic.make_mean_params(0)
m = Model(backbone="dinov2_vitl14", img_size=896, smplx_data=sm, mean_params=mp, precision="bf16")
m.load_state_dict(synthetic.make_state_dict("dinov2_vitl14", 896, seed=0, mean_params=mp), strict=True)
m = m.to("cuda:0").eval()
g = torch.Generator(device="cuda:0").manual_seed(1)
x = torch.randn(32, 3, 896, 896, generator=g, device="cuda:0")
K = synthetic.get_camera_K(896, 32).cuda()
idx = tuple(t.cuda() for t in synthetic.make_pinned_idx(32, 64, 8, seed=0))
ref = None
for i in range(6):
    o = m(x, idx=idx, K=K, is_training=True)
    cur = {k: o[k].clone() for k in ("scores", "v3d", "rotmat", "transl", "j2d")}
    if ref is None: ref = cur
    else:
        for k in ref: assert torch.equal(ref[k], cur[k]), (i, k)
print("B=32 full forward x6: bit-identical;", "finite:", all(torch.isfinite(v).all().item() for v in ref.values()), "persons", ref["v3d"].shape[0])
h = m(x[:4], K=K[:4], det_thresh=0.5, nms_kernel_size=3)
print("inference mode on 4 images:", len(h), "persons")
