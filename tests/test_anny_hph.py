"""Anny-variant HPH: oracle vs the golden produced by the reference's own
multi_hmr_anny/hph.py, and (-m gpu) the HIP path vs both."""
import os

import numpy as np
import pytest
import torch

from oracle import anny_hph_ref

GOLD = os.path.join(os.path.dirname(__file__), "golden", "anny_hph.npz")


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


def test_oracle_matches_reference_golden():
    sd, x, context, mask = anny_hph_ref.make_case()
    y = anny_hph_ref.forward(sd, x, context, mask, depth=8, heads=16)
    gold = np.load(GOLD)["y"]
    real = mask.bool().numpy()
    assert rel(y.numpy()[real], gold[real]) < 1e-5          # padded rows are garbage in the reference and are dropped by its caller


def test_state_dict_keys_match_reference_names():
    from multi_hmr_amd.anny_hph import HPH
    sd, *_ = anny_hph_ref.make_case()
    m = HPH(dim=512, depth=8, heads=16, dim_head=32, mlp_dim=2048, dropout=0.0)
    assert set(m.state_dict().keys()) == set(sd.keys())
    assert all(tuple(m.state_dict()[k].shape) == tuple(v.shape) for k, v in sd.items())


@pytest.mark.gpu
@pytest.mark.parametrize("precision,tol", [("f16", 1e-3), ("bf16", 1e-2)])
def test_hip_path_matches_reference_golden(precision, tol):
    from multi_hmr_amd.anny_hph import HPH
    sd, x, context, mask = anny_hph_ref.make_case()
    m = HPH(dim=512, depth=8, heads=16, dim_head=32, mlp_dim=2048, dropout=0.0, precision=precision)
    m.load_state_dict(sd, strict=True)
    m = m.to("cuda:0").eval()
    y = m(x.cuda(), context.cuda(), mask.cuda()).cpu().numpy()
    gold = np.load(GOLD)["y"]
    real = mask.bool().numpy()
    e = rel(y[real], gold[real])
    print(f"\n[parity anny_hph {precision}] rel-L2 {e:.2e}")
    assert e < tol
    assert np.all(y[~real] == 0)
