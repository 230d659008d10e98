"""Accuracy harness: the reference's evaluation loop ``Trainer.evaluate`` (``train.py:336-482``) around the
MI355X path -- greedy 2D matching of predictions to ground truth, detection precision / recall / F1, per-vertex error and
Procrustes-aligned per-vertex error (and the same on regressed joints).  Matching is a few-person combinatorial loop and
stays on the host exactly as in the reference (``utils/training.py:9-195``); the mesh metrics run in
``mhmr_eval_mesh_errors`` on the device where the vertices already are.  Datasets are not part of this repository: the
caller provides ground truth dicts with the keys the reference's ``prepare_gt`` produces (``j2d [G,J,2]``, ``v3d [G,V,3]``,
``transl_pelvis [G,1,3]``, ``K``)."""
from __future__ import annotations

from itertools import product

import numpy as np
import torch

from . import _lib


def compute_prf1(count, miss, fp):
    """utils/training.py:9-24 (precision, recall, F1 in percent, each rounded to 2 decimals before scaling)."""
    if count == 0:
        return 0, 0, 0
    tp, fn = count - miss, miss
    if tp == 0:
        return 0., 0., 0.
    f1 = round(tp / (tp + 0.5 * (fp + fn)), 2)
    recall = round(tp / (tp + fn), 2)
    precision = round(tp / (tp + fp), 2)
    return 100. * precision, 100. * recall, 100. * f1


def get_bbx_overlap(p1, p2):
    """utils/training.py:149-195: IoU of the two keypoint sets' axis-aligned boxes (inclusive +1 pixel convention)."""
    lo1, lo2, hi1, hi2 = np.min(p1, axis=0), np.min(p2, axis=0), np.max(p1, axis=0), np.max(p2, axis=0)
    assert lo1[0] < hi1[0] and lo1[1] < hi1[1] and lo2[0] < hi2[0] and lo2[1] < hi2[1]
    x_left, y_top = max(lo1[0], lo2[0]), max(lo1[1], lo2[1])
    x_right, y_bottom = min(hi1[0], hi2[0]), min(hi1[1], hi2[1])
    inter = max(0, x_right - x_left + 1) * max(0, y_bottom - y_top + 1)
    a1 = (hi1[0] - lo1[0] + 1) * (hi1[1] - lo1[1] + 1)
    a2 = (hi2[0] - lo2[0] + 1) * (hi2[1] - lo2[1] + 1)
    return inter / float(a1 + a2 - inter)


def match_2d_greedy(pred_kps, gtkp, valid_mask, iou_thresh=0.05, valid=None):
    """utils/training.py:26-147: visit (prediction, ground truth) pairs by ascending 2D keypoint distance; a pair whose boxes
    overlap by >= iou_thresh and whose members are both free is a match; the closest remaining pair failing the IoU test is
    counted as one false positive and ends that round.  Returns (matches [n,2] (pred, gt), false-positive pred ids,
    missed gt ids)."""
    n_pred, n_gt = len(pred_kps), len(gtkp)
    combs = list(product(range(n_pred), range(n_gt)))
    err = np.empty(len(combs))
    for c, (p, g) in enumerate(combs):
        vmask = valid_mask[g]
        assert vmask.sum() > 0, "no valid points"
        err[c] = np.linalg.norm(pred_kps[p][vmask, :2] - gtkp[g][vmask, :2], 2)
    gt_assigned = np.zeros(n_gt, dtype=bool)
    op_assigned = np.zeros(n_pred, dtype=bool)
    best, fp_counter = [], 0
    while gt_assigned.sum() < n_gt and op_assigned.sum() + fp_counter < n_pred:
        found = false_positive = False
        while not found:
            if np.all(np.isinf(err)):
                print("something went wrong here")     # the reference prints and would loop forever; stop instead
                return _finish(best, n_pred, n_gt, valid)
            i = int(np.argmin(err))
            p, g = combs[i]
            iou = get_bbx_overlap(pred_kps[p], gtkp[g])
            err[i] = np.inf
            if not op_assigned[p] and not gt_assigned[g] and iou >= iou_thresh:
                found = True
            elif iou < iou_thresh:
                found = false_positive = True
                fp_counter += 1
        if valid is not None:
            if valid[g]:
                if not false_positive:
                    best.append((p, g))
                    op_assigned[p] = gt_assigned[g] = True
            else:
                gt_assigned[g] = True
        elif not false_positive:
            best.append((p, g))
            op_assigned[p] = gt_assigned[g] = True
    return _finish(best, n_pred, n_gt, valid)


def _finish(best, n_pred, n_gt, valid):
    best = np.array(best)
    ops = sorted(int(b[0]) for b in best)
    gts = sorted(int(b[1]) for b in best)
    false_positives = [int(i) for i in np.setdiff1d(np.arange(n_pred), ops)]
    misses = [int(i) for i in np.setdiff1d(np.arange(n_gt), gts) if valid is None or valid[i]]
    return best, false_positives, misses


def mesh_errors(pred_pts, gt_pts, pred_center=None, gt_center=None, return_transform=False):
    """PVE and PA-PVE in millimetres for M pairs (train.py:372-389): ``pred_pts`` / ``gt_pts`` ``[M,V,3]`` fp32 device tensors,
    optional centres ``[M,3]``.  One launch of ``mhmr_eval_mesh_errors``; no host sync."""
    if pred_pts.device.type != "cuda":
        raise _lib.MhmrError("mesh_errors runs on the HIP device only (no CPU fallback)")
    M, V = int(pred_pts.shape[0]), int(pred_pts.shape[1])
    dev = pred_pts.device
    f = lambda t: None if t is None else t.to(device=dev, dtype=torch.float32).reshape(M, 3).contiguous()
    p, g = pred_pts.to(torch.float32).contiguous(), gt_pts.to(device=dev, dtype=torch.float32).contiguous()
    assert g.shape == p.shape and p.shape[2] == 3
    pc, gc = f(pred_center), f(gt_center)
    pve, pa = torch.empty(M, device=dev), torch.empty(M, device=dev)
    rts = torch.empty(M, 13, device=dev) if return_transform else None
    ptr = lambda t: None if t is None else t.data_ptr()
    _lib.check(_lib.lib().mhmr_eval_mesh_errors(p.data_ptr(), g.data_ptr(), ptr(pc), ptr(gc), M, V, pve.data_ptr(), pa.data_ptr(), ptr(rts),
                                                torch.cuda.current_stream(dev).cuda_stream), "mhmr_eval_mesh_errors")
    return (pve, pa, rts) if return_transform else (pve, pa)


class Evaluator:
    """Accumulates the reference's metrics over batches::

        ev = Evaluator()
        for x, gt in data:                       # gt: dict(j2d [G,J,2], v3d [G,V,3], transl_pelvis [G,1,3], K)
            humans = model(x, is_training=False, K=gt['K'], det_thresh=0.2, nms_kernel_size=3)
            ev.update(humans, gt)
        print(ev.summary())                      # pve, pa_pve (mm), precision, recall, f1_score (%)
    """

    def __init__(self):
        self.count = self.miss = self.fp = 0
        self._sums = {k: torch.zeros((), dtype=torch.float64) for k in ("pve", "pa_pve", "mpjpe", "pa_mpjpe")}
        self._n = self._nj = 0

    @torch.no_grad()
    def update(self, pred, gt):
        kp_gt = gt["j2d"].detach().cpu().numpy()
        if len(pred) == 0:
            self.count += len(kp_gt)
            self.miss += len(kp_gt)
            return
        kp_pred = np.asarray([h["j2d"].detach().cpu().numpy()[:kp_gt.shape[1]] for h in pred])
        best, fps, misses = match_2d_greedy(kp_pred, kp_gt, np.ones_like(kp_gt[..., 0]).astype(np.bool_))
        self.count += len(kp_gt)
        self.miss += len(misses)
        self.fp += len(fps)
        if len(best) == 0:
            return
        dev = pred[0]["v3d"].device
        pid, gid = [int(b[0]) for b in best], [int(b[1]) for b in best]
        v_hat = torch.stack([pred[i]["v3d"] for i in pid])
        c_hat = torch.stack([pred[i]["transl_pelvis"].reshape(3) for i in pid])
        v_gt = gt["v3d"][gid].to(dev)
        c_gt = gt["transl_pelvis"][gid].reshape(-1, 3).to(dev)
        pve, pa = mesh_errors(v_hat, v_gt, c_hat, c_gt)
        self._sums["pve"] += pve.double().sum().cpu()
        self._sums["pa_pve"] += pa.double().sum().cpu()
        if "j3d" in gt:        # joint errors on the body model's own joints, pelvis-centred (the reference regresses H36M joints
            j_gt = gt["j3d"][gid].to(dev)          # from SMPL vertices for 3DPW only, train.py:398-423: that needs assets we do not have)
            J = j_gt.shape[1]
            j_hat = torch.stack([pred[i]["j3d"][:J] for i in pid])
            mp, pamp = mesh_errors(j_hat, j_gt, j_hat[:, 0], j_gt[:, 0])
            self._sums["mpjpe"] += mp.double().sum().cpu()
            self._sums["pa_mpjpe"] += pamp.double().sum().cpu()
            self._nj += len(best)
        self._n += len(best)

    def summary(self):
        precision, recall, f1 = compute_prf1(self.count, self.miss, self.fp)
        out = {k: float(self._sums[k] / max(self._n, 1)) for k in ("pve", "pa_pve")}
        if self._nj:
            out.update({k: float(self._sums[k] / self._nj) for k in ("mpjpe", "pa_mpjpe")})
        out.update(precision=precision, recall=recall, f1_score=f1, matched=self._n, count=self.count)
        return out
