#!/usr/bin/env python
"""Golden vectors for the evaluation helpers: runs the REFERENCE's own /root/reference/utils/training.py (imports cleanly:
torch + numpy only) -- match_2d_greedy, compute_prf1, get_bbx_overlap -- on seeded synthetic keypoint sets and stores inputs
and outputs in tests/golden/eval_match.npz.  Run in the build container only (the GPU box has no /root/reference)."""
import importlib.util
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def cases():
    out = []
    rng = np.random.RandomState(0)
    for c in range(12):
        n_gt, n_pred, J = rng.randint(1, 6), rng.randint(1, 7), 17
        centers = rng.rand(n_gt, 1, 2) * 700 + 100
        gt = centers + rng.randn(n_gt, J, 2) * 40
        pred = []
        for p in range(n_pred):
            if p < n_gt and rng.rand() < 0.8:                      # a detection of gt person p, jittered
                pred.append(gt[p] + rng.randn(J, 2) * 6)
            else:                                                  # a spurious detection somewhere else
                pred.append(rng.rand(1, 2) * 800 + rng.randn(J, 2) * 40)
        pred = np.stack(pred)[rng.permutation(n_pred)]
        out.append((pred.astype(np.float32), gt.astype(np.float32)))
    return out


def main():
    spec = importlib.util.spec_from_file_location("ref_training", "/root/reference/utils/training.py")
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    store = {}
    cnt = miss = fp = 0
    for i, (pred, gt) in enumerate(cases()):
        best, fps, misses = ref.match_2d_greedy(pred, gt, np.ones_like(gt[..., 0]).astype(np.bool_))
        store[f"pred{i}"], store[f"gt{i}"] = pred, gt
        store[f"best{i}"] = np.asarray(best, dtype=np.int64).reshape(-1, 2)
        store[f"fp{i}"] = np.asarray(fps, dtype=np.int64)
        store[f"miss{i}"] = np.asarray(misses, dtype=np.int64)
        store[f"iou{i}"] = np.asarray([[ref.get_bbx_overlap(p, g) for g in gt] for p in pred])
        cnt, miss, fp = cnt + len(gt), miss + len(misses), fp + len(fps)
        store[f"prf{i}"] = np.asarray(ref.compute_prf1(cnt, miss, fp), dtype=np.float64)
    store["n"] = np.asarray(len(cases()))
    store["prf_edge"] = np.asarray([ref.compute_prf1(0, 0, 0), ref.compute_prf1(5, 5, 2), ref.compute_prf1(7, 2, 3)], dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, "eval_match.npz"), **store)
    print("wrote eval_match.npz:", {k: v.shape for k, v in list(store.items())[:6]})


if __name__ == "__main__":
    sys.exit(main())
