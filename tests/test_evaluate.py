"""Evaluation harness.  CPU: matching / IoU / P-R-F1 against goldens produced by the reference's own
utils/training.py (tests/golden/make_golden_eval.py); known-answer tests pinning the restated roma registration.
GPU: mhmr_eval_mesh_errors against the fp64 oracle (oracle/eval_ref.py), incl. the 10475-vertex mesh size, reflections,
and the Evaluator loop end to end."""
import os

import numpy as np
import pytest
import torch

from multi_hmr_amd import evaluate as ev
from oracle import eval_ref, roma_ref

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "eval_match.npz"))


def test_matching_iou_prf1_match_reference_goldens():
    cnt = miss = fp = 0
    for i in range(int(G["n"])):
        pred, gt = G[f"pred{i}"], G[f"gt{i}"]
        best, fps, misses = ev.match_2d_greedy(pred, gt, np.ones_like(gt[..., 0]).astype(np.bool_))
        assert np.array_equal(np.asarray(best, dtype=np.int64).reshape(-1, 2), G[f"best{i}"]), i
        assert list(fps) == list(G[f"fp{i}"]) and list(misses) == list(G[f"miss{i}"]), i
        iou = np.asarray([[ev.get_bbx_overlap(p, g) for g in gt] for p in pred])
        assert np.array_equal(iou, G[f"iou{i}"])
        cnt, miss, fp = cnt + len(gt), miss + len(misses), fp + len(fps)
        assert np.array_equal(np.asarray(ev.compute_prf1(cnt, miss, fp), dtype=np.float64), G[f"prf{i}"])
    edge = [ev.compute_prf1(0, 0, 0), ev.compute_prf1(5, 5, 2), ev.compute_prf1(7, 2, 3)]
    assert np.array_equal(np.asarray(edge, dtype=np.float64), G["prf_edge"])


def _random_similarity(g, dtype=torch.float64):
    R = roma_ref.rotvec_to_rotmat(torch.randn(3, generator=g, dtype=dtype))
    return R, torch.randn(3, generator=g, dtype=dtype), float(torch.rand((), generator=g, dtype=dtype) * 1.5 + 0.3)


def test_registration_oracle_known_answers():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(500, 3, generator=g, dtype=torch.float64)
    R, t, s = _random_similarity(g)
    y = s * x @ R.T + t
    R2, t2, s2 = roma_ref.rigid_points_registration(x, y, compute_scaling=True)
    assert torch.allclose(R2, R, atol=1e-10) and torch.allclose(t2, t, atol=1e-10) and abs(float(s2) - s) < 1e-10
    # with noise: the optimum beats perturbed transforms, and the rotation stays proper even when the best orthogonal map is a reflection
    y = y + 0.05 * torch.randn(500, 3, generator=g, dtype=torch.float64)
    R2, t2, s2 = roma_ref.rigid_points_registration(x, y, compute_scaling=True)
    cost = lambda R_, t_, s_: float(((s_ * x @ R_.T + t_ - y) ** 2).sum())
    c0 = cost(R2, t2, s2)
    for _ in range(20):
        dR = roma_ref.rotvec_to_rotmat(0.01 * torch.randn(3, generator=g, dtype=torch.float64))
        assert cost(dR @ R2, t2 + 0.01 * torch.randn(3, generator=g, dtype=torch.float64), s2 * (1 + 0.01 * float(torch.randn((), generator=g)))) > c0
    ym = y * torch.tensor([1.0, 1.0, -1.0], dtype=torch.float64)             # mirrored target
    Rm, _, _ = roma_ref.rigid_points_registration(x, ym, compute_scaling=True)
    assert abs(float(torch.det(Rm)) - 1.0) < 1e-10


def test_mesh_errors_refuses_cpu():
    from multi_hmr_amd import _lib
    with pytest.raises(_lib.MhmrError):
        ev.mesh_errors(torch.zeros(1, 4, 3), torch.zeros(1, 4, 3))


@pytest.mark.gpu
@pytest.mark.parametrize("M,V", [(1, 14), (3, 127), (5, 10475), (2, 6890)])
def test_gpu_mesh_errors_match_fp64_oracle(M, V):
    g = torch.Generator().manual_seed(M * 1000 + V)
    gt = torch.randn(M, V, 3, generator=g) * 0.4 + torch.randn(M, 1, 3, generator=g) * 3
    pred = torch.empty_like(gt)
    for m in range(M):
        R, t, s = _random_similarity(g, torch.float32)
        pred[m] = (s * gt[m] @ R.T + t) + 0.02 * torch.randn(V, 3, generator=g)
    if M > 1:
        pred[1] = pred[1] * torch.tensor([1.0, -1.0, 1.0])       # a mirrored prediction: rotation must stay proper
    pc, gc = pred[:, 0].clone(), gt[:, 0].clone()
    pve, pa, rts = ev.mesh_errors(pred.cuda(), gt.cuda(), pc.cuda(), gc.cuda(), return_transform=True)
    for m in range(M):
        pve_r, pa_r, R, t, s = eval_ref.mesh_errors(pred[m], pc[m], gt[m], gc[m])
        assert abs(float(pve[m]) - float(pve_r)) <= 1e-5 * float(pve_r), (m, float(pve[m]), float(pve_r))
        assert abs(float(pa[m]) - float(pa_r)) <= 1e-4 * max(float(pa_r), 1.0), (m, float(pa[m]), float(pa_r))
        assert torch.allclose(rts[m, :9].cpu().double().view(3, 3), R, atol=1e-5)
        assert abs(float(rts[m, 12]) - float(s)) < 1e-5 and torch.allclose(rts[m, 9:12].cpu().double(), t, atol=1e-4)
    pve0, pa0 = ev.mesh_errors(gt.cuda(), gt.cuda())
    assert float(pve0.abs().max()) == 0.0 and float(pa0.abs().max()) < 1e-3


@pytest.mark.gpu
def test_gpu_evaluator_loop():
    g = torch.Generator().manual_seed(1)
    V, J = 10475, 127
    gt_v = torch.randn(3, V, 3, generator=g) * 0.3
    gt_j = torch.randn(3, J, 3, generator=g) * 0.3
    centers = torch.tensor([[200.0, 300.0], [500.0, 320.0], [760.0, 280.0]])
    gt_j2d = centers[:, None] + torch.randn(3, J, 2, generator=g) * 40
    gt = dict(j2d=gt_j2d, v3d=gt_v, j3d=gt_j, transl_pelvis=gt_j[:, [0]])
    humans = []
    for p in (1, 0):                                  # two detections (gt persons 1 and 0), gt person 2 is missed
        humans.append(dict(j2d=(gt_j2d[p] + torch.randn(J, 2, generator=g) * 3).cuda(), v3d=(gt_v[p] + 0.01).cuda(),
                           j3d=(gt_j[p] + 0.01).cuda(), transl_pelvis=(gt_j[p, [0]] + 0.01).cuda()))
    humans.append(dict(j2d=(torch.tensor([50.0, 800.0]) + torch.randn(J, 2, generator=g) * 20).cuda(), v3d=gt_v[2].cuda(), j3d=gt_j[2].cuda(),
                       transl_pelvis=gt_j[2, [0]].cuda()))                                   # a false positive far away
    e = ev.Evaluator()
    e.update(humans, gt)
    e.update([], gt)                                  # an image with no detection: 3 more misses
    s = e.summary()
    assert s["matched"] == 2 and s["count"] == 6
    assert s["pve"] < 1e-2 and s["pa_pve"] < 1e-2 and s["mpjpe"] < 1e-2      # a pure translation cancels with the pelvis centring
    assert (s["precision"], s["recall"]) == ev.compute_prf1(6, 4, 1)[:2]
