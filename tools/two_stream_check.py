#!/usr/bin/env python
"""Diagnostic: two models / host threads / streams; every ViT workspace buffer is compared with the serial run, for a given backbone depth (MHMR_LIBDIR = another build of the library, tools/run_with_lib.py)."""
import os, sys, threading
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_golden, synthetic
from multi_hmr_amd import Model, _lib
if os.environ.get('MHMR_LIBDIR'):
    _lib.LIB_PATH = os.path.join(ROOT, os.environ['MHMR_LIBDIR'], 'libmhmr.so')
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cfg = dict(make_golden.CASES["vitl_224_train"]); cfg["depth_override"] = depth
sm, mp = synthetic.make_smplx_data(0), synthetic.make_mean_params(0)
x, K, idx = make_golden.case_inputs(cfg)
# MHMR_SOAK_BATCH images (the case's two, repeated); with MHMR_SPLIT=n the backbone of every model runs n image blocks on side
# streams of its own (model.Model._run_backbone): 2 host threads x n streams each
nb = int(os.environ.get("MHMR_SOAK_BATCH", "2"))
xc = x.repeat((nb + 1) // 2, 1, 1, 1)[:nb].contiguous().cuda()
def build(seed):
    sd = synthetic.make_state_dict(cfg["backbone"], cfg["img_size"], seed=seed, depth_override=depth)
    m = Model(backbone=cfg["backbone"], img_size=cfg["img_size"], smplx_data=sm, mean_params=mp, backbone_depth=depth, precision="f16")
    m.load_state_dict(sd, strict=True)
    return m.to("cuda:0").eval()
models = [build(2), build(91)]
NAMES = ("resid", "xn", "qk", "vt", "att", "hid", "pstats", "rowstats", "feat32")
def snap(m):
    ws = m._workspace(m._packed, xc.shape[0])
    out = {"feat32": ws["feat32"].clone()}
    for i, part in enumerate(ws["parts"]):
        out.update({f"{k}[{i}]": part["bufs"][k].clone() for k in NAMES if k in part["bufs"]})
    return out
serial = []
NBAD = [0]
for m in models:
    m.backbone_features(xc); torch.cuda.synchronize(); serial.append(snap(m))
streams = [torch.cuda.Stream() for _ in models]
def worker(i):
    with torch.cuda.stream(streams[i]):
        for rep in range(int(os.environ.get('REPS','10'))):
            models[i].backbone_features(xc)
            streams[i].synchronize()
            s = snap(models[i]); streams[i].synchronize()
            bad = {k: int((s[k] != serial[i][k]).sum()) for k in s if not torch.equal(s[k], serial[i][k])}
            if bad:
                NBAD[0] += 1
                det = {}
                for k in bad:
                    d = (s[k] != serial[i][k])
                    rows = d.reshape(d.shape[0], -1).any(1).nonzero().flatten()
                    det[k] = (bad[k], rows[:6].tolist(), int(rows.numel()))
                    if bad[k] <= 64:
                        a2, b2 = s[k].reshape(d.shape[0], -1), serial[i][k].reshape(d.shape[0], -1)
                        ij = (a2 != b2).nonzero()
                        det[k + "_where"] = [(int(r), int(c), float(a2[r, c]), float(b2[r, c])) for r, c in ij.tolist()]
                print(f"depth {depth} thread {i} rep {rep}: {det}", flush=True)
for s in streams: s.wait_stream(torch.cuda.current_stream())
ts = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
[t.start() for t in ts]; [t.join() for t in ts]
print("done depth", depth, "batch", nb, "image blocks", models[0]._nsplit(nb), "reps", os.environ.get('REPS', '10'), "source", _lib.lib().mhmr_source_hash().decode(),
      "mismatching forwards", NBAD[0])
