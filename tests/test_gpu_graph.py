"""-m gpu: the hipGraph replay of the inference forward (multi_hmr_amd/graphed.py) against the eager forward it was recorded from.

The graph is the eager path's own launches under stream capture, so the bar is bit equality -- for several inputs in a row through ONE
recording (the graph reads its static input buffers, not whatever was there at capture time), for a threshold that overflows the recorded
capacity (eager re-run), for an image nobody is detected in, and for a batch that runs its backbone as two image blocks on two streams
(fork / join events inside the capture)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import make_golden  # noqa: E402
from multi_hmr_amd import GraphedForward, Model, forward_model  # noqa: E402
from multi_hmr_amd import _lib  # noqa: E402

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def build(cfg, smplx_data, mean_params, sd, **kw):
    m = Model(backbone=cfg["backbone"], img_size=cfg["img_size"], smplx_data=smplx_data, mean_params=mean_params,
              backbone_depth=cfg["depth_override"], precision="f16", **kw)
    m.load_state_dict(sd, strict=True)
    return m.to("cuda:0").eval()


def same_persons(got, ref, what):
    assert len(got) == len(ref), (what, len(got), len(ref))
    for i, (p, q) in enumerate(zip(got, ref)):
        assert list(p.keys()) == list(q.keys())
        for k in p:
            assert torch.equal(p[k], q[k]), (what, i, k, float((p[k] - q[k]).abs().max()))


def test_graph_replay_equals_the_eager_forward(smplx_data, mean_params):
    cfg = make_golden.CASES["vits_448_infer"]
    gold = np.load(os.path.join(GOLD, "vits_448_infer.npz"))
    sd = make_golden.case_state_dict(cfg)
    sd["mlp_classif.2.bias"] = torch.from_numpy(gold["classif_bias"])
    x, K, _ = make_golden.case_inputs(cfg)
    xc, Kc = x.cuda(), K.cuda()
    thr, k = float(gold["det_thresh"]), cfg["nms_kernel_size"]
    model = build(cfg, smplx_data, mean_params, sd)
    gf = GraphedForward(model, batch=xc.shape[0], det_thresh=thr, nms_kernel_size=k, capacity=16)

    def eager(xx, KK, t=thr):
        return build(cfg, smplx_data, mean_params, sd)(xx, K=KK, det_thresh=t, nms_kernel_size=k)      # fresh model: the exact path

    ref = eager(xc, Kc)
    assert len(ref) == int(gold["num_humans"]) > 0          # the golden's person count (the reference's own detections)
    same_persons(gf(xc, Kc), ref, "first replay")
    # other inputs through the same recording: images in another order, another camera
    perm = torch.tensor([2, 0, 1], device="cuda")
    x2, K2 = xc[perm].contiguous(), (Kc * torch.tensor([1.1, 1.1, 1.0], device="cuda").view(1, 3, 1)).contiguous()
    same_persons(gf(x2, K2), eager(x2, K2), "permuted images, other K")
    same_persons(gf(xc, Kc), ref, "back to the first input")
    # views of the graph's own buffers when copy=False, and the batched form
    o, ids = gf(xc, Kc, copy=False, return_batched=True)
    ob, idb = model(xc, K=Kc, det_thresh=thr, nms_kernel_size=k, return_batched=True)
    assert torch.equal(ids, idb) and all(torch.equal(o[n], ob[n]) for n in ob)
    assert o["v3d"].data_ptr() == gf.out["v3d"].data_ptr()
    same_persons(gf(torch.zeros_like(xc), Kc), eager(torch.zeros_like(xc), Kc), "blank images")
    assert gf.overflows == 0
    # nobody detected (scores are sigmoids: none reaches 2)
    assert GraphedForward(model, batch=xc.shape[0], det_thresh=2.0, nms_kernel_size=k)(xc, Kc) == []
    # a recording whose capacity is too small for what a lower threshold finds (every local maximum): the eager re-run, same persons
    low = 1e-9
    n_low = len(eager(xc, Kc, low))
    assert n_low > 16
    gl = GraphedForward(model, batch=xc.shape[0], det_thresh=low, nms_kernel_size=k, capacity=16)
    same_persons(gl(xc, Kc), eager(xc, Kc, low), "overflow")
    assert gl.overflows == 1
    # forward_model(use_graph=True): the cached recording of the model
    same_persons(forward_model(model, xc, Kc, det_thresh=thr, nms_kernel_size=k, use_graph=True), ref, "forward_model(use_graph=True)")
    same_persons(forward_model(model, xc, Kc, det_thresh=thr, nms_kernel_size=k, use_graph=True), ref, "forward_model(use_graph=True), cached")
    assert len(model._graphs) == 1
    # a re-pack invalidates the recording loudly
    model.repack()
    with pytest.raises(_lib.MhmrError):
        gf(xc, Kc)


def test_graph_with_image_blocks_on_side_streams(smplx_data, mean_params):
    """Eight images = two backbone blocks on two streams (Model._nsplit): the fork / join events are recorded into the graph."""
    cfg = dict(make_golden.CASES["vits_448_infer"], batch=8, img_size=224)
    sd = make_golden.case_state_dict(cfg)
    x, K, _ = make_golden.case_inputs(cfg)
    xc, Kc = x.cuda(), K.cuda()
    model = build(cfg, smplx_data, mean_params, sd)
    assert model._nsplit(8) == 2
    # a threshold a little under the third-highest score of every image, so that persons exist whatever the seeded classifier says
    model(xc, K=Kc, det_thresh=2.0, nms_kernel_size=1)
    P, ws, _ = model._prepare(xc)
    thr = float(ws["scores"].view(8, -1).topk(3, dim=1).values[:, 2].min()) * 0.999
    ref = build(cfg, smplx_data, mean_params, sd)(xc, K=Kc, det_thresh=thr, nms_kernel_size=1)
    assert len(ref) >= 24
    gf = GraphedForward(model, batch=8, det_thresh=thr, nms_kernel_size=1, capacity=len(ref) + 8)
    for rep in range(3):
        same_persons(gf(xc, Kc), ref, f"replay {rep}")
