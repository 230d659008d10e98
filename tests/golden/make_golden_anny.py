"""tests/golden/anny_hph.npz: output of the reference's OWN multi_hmr_anny/hph.py (imported by path; it only needs torch +
einops) on the seeded case of oracle.anny_hph_ref.make_case.  Run in the build container only."""
import importlib.util
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import anny_hph_ref  # noqa: E402

spec = importlib.util.spec_from_file_location("ref_anny_hph", "/root/reference/multi_hmr_anny/hph.py")
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)

sd, x, context, mask = anny_hph_ref.make_case()
m = ref.HPH(dim=512, depth=8, heads=16, dim_head=32, mlp_dim=2048, dropout=0.0).eval()
print(m.load_state_dict(sd, strict=True))
with torch.no_grad():
    y = m(x, context, mask)
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "anny_hph.npz"), y=y.numpy())
print(y.shape, float(y.abs().mean()))
