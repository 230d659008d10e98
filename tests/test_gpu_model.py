"""-m gpu: end-to-end parity of the HIP path (multi_hmr_amd.Model, through libmhmr.so) against
(a) the golden vectors produced by the reference's own model.py (tests/golden/*.npz) and
(b) the portable CPU oracle on the same seeded inputs.

Tolerances: tests/parity.py (BASELINE.json north_star: 1e-3 relative for person scores, SMPL-X parameters and 3D vertices; met with
f16 MFMA operands, the precision the reference's own GPU path uses -- fp16 autocast, demo.py:117.  With bf16 operands (8 mantissa
bits) the measured deviation through the backbone is 2-8e-3 -- the arithmetic limit of the format, SURVEY.md Appendix E -- so the
bf16 mode is held to 2e-2 and its measured error is printed)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import make_golden  # noqa: E402
import parity  # noqa: E402
from multi_hmr_amd import Model  # noqa: E402
import synthetic  # noqa: E402
from oracle import roma_ref  # noqa: E402
from parity import CHECKED, TOL, rel  # noqa: E402

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def build(cfg, smplx_data, mean_params, precision, sd=None):
    sd = make_golden.case_state_dict(cfg) if sd is None else sd
    m = Model(backbone=cfg["backbone"], img_size=cfg["img_size"], smplx_data=smplx_data, mean_params=mean_params,
              backbone_depth=cfg["depth_override"], precision=precision, **cfg.get("model_kwargs", {}))
    missing, unexpected = m.load_state_dict(sd, strict=True)
    return m.to("cuda:0").eval()


@pytest.mark.parametrize("precision", ["f16", "bf16"])
@pytest.mark.parametrize("name", ["vits_224_train", "vitb_224_train", "vitl_224_train", "vits_224_bands8"])   # bands8: 51 camera channels
def test_training_mode_matches_reference_golden(name, precision, smplx_data, mean_params):
    cfg = make_golden.CASES[name]
    gold = np.load(os.path.join(GOLD, name + ".npz"))
    model = build(cfg, smplx_data, mean_params, precision)
    x, K, idx = make_golden.case_inputs(cfg)
    z = model.backbone_features(x.cuda()).cpu()
    e_bb = rel(z[:, :: max(1, z.shape[1] // 64)].numpy(), gold["backbone"])
    out = model(x.cuda(), idx=tuple(i.cuda() for i in idx), K=K.cuda(), is_training=True)
    assert set(out.keys()) == set(gold.files) - {"backbone"}
    errs = {k: rel(out[k].cpu().numpy(), gold[k]) for k in CHECKED}
    # rotvec is discontinuous at pi: compare the rotation it encodes
    errs["rotvec"] = rel(roma_ref.rotvec_to_rotmat(out["rotvec"].cpu()).numpy(), roma_ref.rotvec_to_rotmat(torch.from_numpy(gold["rotvec"])).numpy())
    vmax_mm = 1e3 * float(np.abs(out["v3d"].cpu().numpy() - gold["v3d"]).max())
    print(f"\n[parity {name} {precision}] backbone rel-L2 {e_bb:.2e}; max vertex error {vmax_mm:.3f} mm; " +
          " ".join(f"{k}={v:.1e}" for k, v in errs.items()))
    assert e_bb < 4 * TOL[precision], e_bb       # features are not a north-star output; informational bound
    parity.assert_within(errs, precision, name)


@pytest.mark.parametrize("precision", ["f16", "bf16"])
@pytest.mark.parametrize("name", ["vits_448_infer", "vitl_448_infer"])      # depth-2 ViT-S; the FULL-depth ViT-L
def test_inference_mode_person_list_matches_reference_golden(name, precision, smplx_data, mean_params):
    cfg = make_golden.CASES[name]
    gold = np.load(os.path.join(GOLD, name + ".npz"))
    sd = make_golden.case_state_dict(cfg)
    sd["mlp_classif.2.bias"] = torch.from_numpy(gold["classif_bias"])
    model = build(cfg, smplx_data, mean_params, precision, sd)
    x, K, _ = make_golden.case_inputs(cfg)
    humans = model(x.cuda(), K=K.cuda(), is_training=False, det_thresh=float(gold["det_thresh"]), nms_kernel_size=cfg["nms_kernel_size"])
    assert isinstance(humans, list)
    if precision == "bf16" and len(humans) != int(gold["num_humans"]):
        pytest.skip(f"bf16 flipped a detection within {float(gold['score_margin']):.1e} of the threshold ({len(humans)} vs {int(gold['num_humans'])})")
    assert len(humans) == int(gold["num_humans"])
    assert list(humans[0].keys()) == ["scores", "loc", "transl", "transl_pelvis", "rotvec", "expression", "shape", "v3d", "j3d", "j2d"]
    assert humans[0]["scores"].dim() == 0 and humans[0]["transl_pelvis"].shape == (1, 3) and humans[0]["v3d"].shape == (10475, 3)
    for k in humans[0].keys():
        got = torch.stack([h[k] for h in humans]).cpu()
        if k == "v3d":
            got = got[:, :: cfg.get("vstride", 1)]
        if k == "rotvec":
            e = rel(roma_ref.rotvec_to_rotmat(got).numpy(), roma_ref.rotvec_to_rotmat(torch.from_numpy(gold["h_rotvec"])).numpy())
        else:
            e = rel(got.numpy(), gold["h_" + k])
        assert e < parity.tolerance(k, precision), (k, e)
    # nobody above the threshold -> empty list (model.py:241-243)
    assert model(x.cuda(), K=K.cuda(), det_thresh=2.0) == []


def test_hip_path_matches_portable_oracle_fresh_seed(smplx_data, mean_params):
    """Same comparison against the oracle itself on a configuration that has no committed golden (different seed,
    ragged person counts incl. >8 queries in one image so the cross-attention chunking is exercised)."""
    from oracle.multihmr_ref import OracleModel
    cfg = dict(backbone="dinov2_vits14", img_size=336, depth_override=3, batch=3, persons=[11, 0, 4], seed=11)
    sd = make_golden.case_state_dict(cfg)
    x, K, idx = make_golden.case_inputs(cfg)
    ref = OracleModel(sd, smplx_data, backbone=cfg["backbone"], img_size=cfg["img_size"], depth_override=3).forward(x, idx=idx, K=K, is_training=True)
    model = build(cfg, smplx_data, mean_params, "f16", sd)
    out = model(x.cuda(), idx=tuple(i.cuda() for i in idx), K=K.cuda(), is_training=True)
    for k in CHECKED:
        assert rel(out[k].cpu().numpy(), ref[k].numpy()) < parity.tolerance(k, "f16"), k


def test_forward_model_wrapper_and_autocast(smplx_data, mean_params):
    """demo.forward_model wraps the call in fp16 autocast (demo.py:117): results must be unchanged."""
    from multi_hmr_amd import forward_model
    cfg = make_golden.CASES["vits_224_train"]
    model = build(cfg, smplx_data, mean_params, "f16")
    x, K, _ = make_golden.case_inputs(cfg)
    a = forward_model(model, x.cuda(), K.cuda(), det_thresh=0.5, nms_kernel_size=3)
    b = model(x.cuda(), K=K.cuda(), det_thresh=0.5, nms_kernel_size=3)
    assert len(a) == len(b) and len(a) > 0
    assert all(torch.equal(p["v3d"], q["v3d"]) for p, q in zip(a, b))


def test_two_host_threads_two_streams_are_independent(smplx_data, mean_params):
    """include/mhmr.h: every entry point is re-entrant across streams.  mhmr_vit_forward used to fork the V projection onto ONE
    process-global side stream with ONE event pair, so two host threads driving two streams could interleave their record / wait calls
    (ctypes releases the GIL) and a V GEMM could start before its own LayerNorm; now everything runs on the caller's stream.  Two
    models with different weights, two threads, two streams, many interleaved forwards: every result bit-equal to the serial one."""
    import threading
    cfg = make_golden.CASES["vitl_224_train"]          # ViT-L: the token-row map, the class-row kernels and the folded LayerNorms run
    x, K, idx = make_golden.case_inputs(cfg)
    xc, Kc, ic = x.cuda(), K.cuda(), tuple(i.cuda() for i in idx)
    sds = [make_golden.case_state_dict(cfg), synthetic.make_state_dict(cfg["backbone"], cfg["img_size"], seed=91, depth_override=cfg["depth_override"])]
    models = [build(cfg, smplx_data, mean_params, "f16", sd) for sd in sds]
    serial = [m(xc, idx=ic, K=Kc, is_training=True) for m in models]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in models]
    results, errors = [[] for _ in models], []

    def worker(i):
        try:
            with torch.cuda.stream(streams[i]):
                for _ in range(6):
                    results[i].append({k: v.clone() for k, v in models[i](xc, idx=ic, K=Kc, is_training=True).items()})
            streams[i].synchronize()
        except Exception as e:          # noqa: BLE001
            errors.append(e)
    for s in streams:
        s.wait_stream(torch.cuda.current_stream())
    threads = [threading.Thread(target=worker, args=(i,)) for i in range(len(models))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    torch.cuda.synchronize()
    assert not errors, errors
    for i in range(len(models)):
        for out in results[i]:
            for k in ("scores", "v3d", "rotmat", "shape", "expression", "transl"):
                assert torch.equal(out[k], serial[i][k]), (i, k)
    assert not torch.equal(serial[0]["v3d"], serial[1]["v3d"])


def test_load_state_dict_after_a_forward_repacks_everything(smplx_data, mean_params):
    """Regression (round-1 advisor finding): the cached workspace holds a descriptor with raw pointers into the packed weights; a
    forward, then load_state_dict(other weights), then a forward at the SAME batch size must run entirely on the new weights."""
    cfg = make_golden.CASES["vits_224_train"]
    x, K, idx = make_golden.case_inputs(cfg)
    xc, Kc, ic = x.cuda(), K.cuda(), tuple(i.cuda() for i in idx)
    sd_a = make_golden.case_state_dict(cfg)
    sd_b = synthetic.make_state_dict(cfg["backbone"], cfg["img_size"], seed=77, depth_override=cfg["depth_override"])
    m = build(cfg, smplx_data, mean_params, "f16", sd_a)
    out_a = m(xc, idx=ic, K=Kc, is_training=True)
    m.load_state_dict(sd_b, strict=True)
    out_b = m(xc, idx=ic, K=Kc, is_training=True)
    fresh = build(cfg, smplx_data, mean_params, "f16", sd_b)(xc, idx=ic, K=Kc, is_training=True)
    for k in ("scores", "v3d", "rotmat", "shape", "transl"):
        assert torch.equal(out_b[k], fresh[k]), k
    assert not torch.equal(out_a["v3d"], out_b["v3d"])
    # a different batch size afterwards gets its own workspace (the two most recent sizes are cached)
    out_1 = m(xc[:1], idx=tuple(i[ic[0] == 0] for i in ic), K=Kc[:1], is_training=True)
    assert float((out_1["v3d"] - fresh["v3d"][ic[0] == 0]).abs().max()) < 1e-5


@pytest.mark.parametrize("backbone", ["dinov2_vits14", "dinov2_vitb14"])      # ViT-B: token-row map + folded LayerNorms (raw residual rows are what gets rounded)
@pytest.mark.parametrize("scale", [100.0, 1500.0])
def test_f16_operands_with_massive_activation_channels(scale, backbone, smplx_data, mean_params):
    """Trained ViTs carry a few 'massive activation' channels (10^2-10^3 x the typical magnitude) that an MLP writes into the residual
    stream early and every later block has to live with; SURVEY.md Appendix E asked whether 16-bit storage of qk / hid / att survives
    them.  Synthetic stand-in: block 0's fc2 bias writes +scale into residual channel 7 and -scale/2 into channel 200, a hidden unit of
    block 1 sits at +scale (fc1 bias), and one q / k channel pair of block 2 is biased so that every attention score of head 0 shifts
    by ~scale^2/50 (the softmax reference level has to absorb it).  The fp32 residual stream and LayerNorm keep the 16-bit operands in
    range: nothing overflows, and the result stays within the f16 tolerance of the CPU oracle run on the same weights."""
    from oracle.multihmr_ref import OracleModel
    cfg = dict(backbone=backbone, img_size=224, depth_override=4, batch=2, persons=[2, 3], seed=9)
    sd = make_golden.case_state_dict(cfg)
    Cd = sd["backbone.encoder.blocks.0.attn.qkv.weight"].shape[1]
    p = "backbone.encoder.blocks."
    sd[p + "0.mlp.fc2.bias"] = sd[p + "0.mlp.fc2.bias"].clone()
    sd[p + "0.mlp.fc2.bias"][7] += scale
    sd[p + "0.mlp.fc2.bias"][200] -= 0.5 * scale
    sd[p + "1.mlp.fc1.bias"] = sd[p + "1.mlp.fc1.bias"].clone()
    sd[p + "1.mlp.fc1.bias"][11] += scale
    sd[p + "2.attn.qkv.bias"] = sd[p + "2.attn.qkv.bias"].clone()
    sd[p + "2.attn.qkv.bias"][5] += scale / 5            # q, head 0
    sd[p + "2.attn.qkv.bias"][Cd + 5] += scale / 10      # k, head 0
    x, K, idx = make_golden.case_inputs(cfg)
    ref = OracleModel(sd, smplx_data, backbone=cfg["backbone"], img_size=cfg["img_size"], depth_override=4).forward(x, idx=idx, K=K, is_training=True)
    model = build(cfg, smplx_data, mean_params, "f16", sd)
    out = model(x.cuda(), idx=tuple(i.cuda() for i in idx), K=K.cuda(), is_training=True)
    errs = {k: rel(out[k].cpu().numpy(), ref[k].numpy()) for k in CHECKED}
    print(f"\n[massive activations x{scale:g} {backbone}] " + " ".join(f"{k}={v:.1e}" for k, v in errs.items()))
    for k in CHECKED:
        assert torch.isfinite(out[k]).all(), k
    parity.assert_within(errs, "f16", f"massive x{scale:g}")


def test_forward_sharded_without_a_process_group_is_the_plain_forward(smplx_data, mean_params):
    from multi_hmr_amd import distributed
    cfg = make_golden.CASES["vits_448_infer"]
    gold = np.load(os.path.join(GOLD, "vits_448_infer.npz"))
    sd = make_golden.case_state_dict(cfg)
    sd["mlp_classif.2.bias"] = torch.from_numpy(gold["classif_bias"])
    model = build(cfg, smplx_data, mean_params, "f16", sd)
    x, K, _ = make_golden.case_inputs(cfg)
    kw = dict(det_thresh=float(gold["det_thresh"]), nms_kernel_size=cfg["nms_kernel_size"])
    a = model(x.cuda(), K=K.cuda(), **kw)
    b, img = distributed.forward_sharded(model, x, K, return_image_index=True, **kw)      # host tensors in, sliced and moved per rank
    assert len(a) == len(b) == int(gold["num_humans"]) and img.tolist() == sorted(img.tolist())
    assert all(torch.equal(p["v3d"], q["v3d"]) and torch.equal(p["scores"], q["scores"]) for p, q in zip(a, b))
    assert distributed.person_fields(model)[-1] == ("v3d", (10475, 3))


def test_person_groups_on_the_device_match_the_host_bookkeeping():
    """mhmr_person_groups (csrc/hph.hip) against the reference's host-side bookkeeping (torch.where order, model.py:146-151; rebatch /
    pad_to_max, utils/tensor_manip.py:7-45): counts, write offsets, self-attention groups, <= 8-query work items -- from the detection
    counts and from a training-hook idx, with empty images, > 8 persons in one image, a capacity below the total, and launch bounds
    above the real table sizes (the padding entries must be empty groups / count-0 items)."""
    import ctypes as C
    from multi_hmr_amd import _lib
    L = _lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    for counts, cap in (([3, 0, 11, 1, 0, 8, 9], 64), ([3, 0, 11, 1, 0, 8, 9], 20), ([0, 0, 0], 8), ([17], 17), ([1] * 40, 64)):
        B, total = len(counts), sum(counts)
        kept, left = [], cap
        for c in counts:
            kept.append(min(c, max(left, 0)))
            left -= kept[-1]
        gstart, chunks, start = [0], [], 0
        for b, c in enumerate(kept):
            if c == 0:
                continue
            for q0 in range(0, c, 8):
                chunks += [b, start + q0, min(8, c - q0)]
            start += c
            gstart.append(start)
        ngc, ncc = min(B, cap) + 2, cap // 8 + min(B, cap) + 3          # bounds above the sufficient ones
        for from_counts in (True, False):
            if not from_counts and cap < total:
                continue                                                    # (the hook's capacity is its person count)
            cnt = torch.tensor(counts, dtype=torch.int32).cuda()
            det_b = torch.tensor([b for b, c in enumerate(counts) for _ in range(c)], dtype=torch.int32).cuda()
            base, gs, ch, info = (torch.full((n,), -7, dtype=torch.int32).cuda() for n in (B, ngc + 1, 3 * ncc, 4))
            _lib.check(L.mhmr_person_groups(cnt.data_ptr() if from_counts else None, det_b.data_ptr() if total else None, total, B, cap,
                                            base.data_ptr(), gs.data_ptr(), ngc, ch.data_ptr(), ncc, info.data_ptr(), st), "groups")
            assert info.tolist() == [start, len(gstart) - 1, len(chunks) // 3, total], (counts, cap, from_counts, info.tolist())
            assert base.tolist() == [sum(counts[:b]) for b in range(B)]
            assert gs.tolist() == gstart + [start] * (ngc + 1 - len(gstart))
            assert ch.tolist() == chunks + [0] * (3 * ncc - len(chunks))


def test_training_hook_follows_every_new_idx(smplx_data, mean_params):
    """Round-3 advisor finding: the hook used to remember per-image counts for an idx tensor it recognised by ADDRESS; a new idx of
    the same shape that re-used a freed address was grouped with the old counts.  Nothing is remembered now (the groups are made on
    the device from the idx handed in): forwards with different idx tensors of one shape -- allocated and freed so that addresses
    recur -- each equal the forward of a fresh model on that idx."""
    cfg = dict(backbone="dinov2_vits14", img_size=224, depth_override=2, batch=4, persons=[3, 1, 0, 4], seed=5)
    sd = make_golden.case_state_dict(cfg)
    x, K, idx = make_golden.case_inputs(cfg)
    xc, Kc = x.cuda(), K.cuda()
    model = build(cfg, smplx_data, mean_params, "f16", sd)
    P = int(idx[0].shape[0])
    variants = [[3, 1, 0, 4], [0, 4, 4, 0], [8, 0, 0, 0], [2, 2, 2, 2]]       # same person count, different images
    for round_ in range(2):
        for cnt in variants:
            img = torch.tensor([b for b, c in enumerate(cnt) for _ in range(c)])
            assert img.numel() == P
            cur = (img, idx[1].clone(), idx[2].clone(), idx[3].clone())
            ic = tuple(t.cuda() for t in cur)              # new device tensors every time; the previous ones were freed
            got = model(xc, idx=ic, K=Kc, is_training=True)
            ref = build(cfg, smplx_data, mean_params, "f16", sd)(xc, idx=tuple(t.cuda() for t in cur), K=Kc, is_training=True)
            for k in ("v3d", "rotmat", "shape", "expression", "transl"):
                assert torch.equal(got[k], ref[k]), (cnt, k)
            del ic, got


def test_fixed_capacity_inference_equals_the_exact_path(smplx_data, mean_params):
    """Inference enqueues the heads for a person-row capacity and reads the person count back after the last launch.  First call
    (no capacity known: count first), second call (capacity above the count: padding rows), a threshold that detects MORE persons than
    the capacity (overflow -> re-run at the exact size), and one that detects nobody: always the persons of the exact path, bit for bit,
    and the padding never touches the context rows of cells nobody detected."""
    cfg = make_golden.CASES["vits_448_infer"]
    gold = np.load(os.path.join(GOLD, "vits_448_infer.npz"))
    sd = make_golden.case_state_dict(cfg)
    sd["mlp_classif.2.bias"] = torch.from_numpy(gold["classif_bias"])
    x, K, _ = make_golden.case_inputs(cfg)
    xc, Kc = x.cuda(), K.cuda()
    thr = float(gold["det_thresh"])

    def exact(t):
        m = build(cfg, smplx_data, mean_params, "f16", sd)         # a fresh model has no capacity: it takes the count first
        return m(xc, K=Kc, det_thresh=t, nms_kernel_size=cfg["nms_kernel_size"])

    model = build(cfg, smplx_data, mean_params, "f16", sd)
    seq = [thr, thr, thr * 0.5, thr, 2.0, thr, thr * 0.25, thr * 0.25]
    for i, t in enumerate(seq):
        cap_before = dict(model._person_cap)
        got, ids = model(xc, K=Kc, det_thresh=t, nms_kernel_size=cfg["nms_kernel_size"], return_image_index=True)
        ref = exact(t)
        assert len(got) == len(ref) == int(ids.numel()), (i, t, len(got), len(ref), cap_before)
        for p, q in zip(got, ref):
            for k in p:
                assert torch.equal(p[k], q[k]), (i, t, k, cap_before)
    assert len(exact(thr * 0.25)) > len(exact(thr)) > 0


@pytest.mark.parametrize("name", ["vitl_224_train", "vits_224_train"])
def test_backbone_image_blocks_on_side_streams_change_nothing(name, smplx_data, mean_params):
    """Model(split=n): the backbone of n image blocks on streams of their own (fork / join around mhmr_vit_forward).  Every kernel is
    batch-invariant, so the outputs are bit-equal to the single-stream run -- also when forwards follow each other without a
    synchronisation in between (the next forward's side streams must wait for the previous forward's consumers of the workspaces)."""
    cfg = dict(make_golden.CASES[name], batch=4)
    cfg["persons"] = [2, 0, 3, 1]
    sd = make_golden.case_state_dict(cfg)
    x, K, idx = make_golden.case_inputs(cfg)
    xc, Kc, ic = x.cuda(), K.cuda(), tuple(i.cuda() for i in idx)

    def mk(split):
        m = Model(backbone=cfg["backbone"], img_size=cfg["img_size"], smplx_data=smplx_data, mean_params=mean_params,
                  backbone_depth=cfg["depth_override"], precision="f16", split=split)
        m.load_state_dict(sd, strict=True)
        return m.to("cuda:0").eval()
    one = mk(1)(xc, idx=ic, K=Kc, is_training=True)
    # (ViT-S with ONE image per block takes the 128x128 kernel where B * Tp = 384 rows are no whole 256-row tiles: another accumulation
    # order in the last bit, not a batch dependence of any one kernel -- compared at the block sizes that keep the kernel choice)
    for split in ((2, 4) if "vitl" in name else (2,)):
        m = mk(split)
        assert m._nsplit(4) == split
        outs = [m(xc if i % 2 == 0 else xc.flip(0), idx=ic, K=Kc, is_training=True) for i in range(6)]       # back to back, no sync
        for k in ("scores", "v3d", "rotmat", "shape", "expression", "transl", "j2d"):
            assert torch.equal(outs[0][k], one[k]) and torch.equal(outs[4][k], one[k]), (split, k)
        assert not torch.equal(outs[1]["v3d"], one["v3d"])


def test_two_host_threads_with_split_backbones_soak(smplx_data, mean_params):
    """The two-thread / two-stream gate at full depth with 8-image batches whose backbone runs as two image blocks on side streams of
    each model (2 threads x 2 streams each, interleaving GEMMs, attention, class-row kernels and folded LayerNorms of two different
    weight sets on one GPU): 40 forwards per thread, every one bit-equal to the serial run.  This is the configuration in which the
    round-3 packed-fp32 failure (csrc/mhmr_common.h; root cause still open) showed within a few dozen forwards of an SLP build."""
    import threading
    cfg = dict(make_golden.CASES["vitl_224_train"], batch=8, persons=[1, 0, 2, 1, 0, 1, 3, 0])
    x, K, idx = make_golden.case_inputs(cfg)
    xc, Kc, ic = x.cuda(), K.cuda(), tuple(i.cuda() for i in idx)
    sds = [make_golden.case_state_dict(cfg), synthetic.make_state_dict(cfg["backbone"], cfg["img_size"], seed=91, depth_override=cfg["depth_override"])]

    def mk(sd):
        m = Model(backbone=cfg["backbone"], img_size=cfg["img_size"], smplx_data=smplx_data, mean_params=mean_params, precision="f16", split=2)
        m.load_state_dict(sd, strict=True)
        return m.to("cuda:0").eval()
    models = [mk(sd) for sd in sds]
    assert models[0]._nsplit(8) == 2
    serial = [m(xc, idx=ic, K=Kc, is_training=True) for m in models]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in models]
    bad, errors = [0, 0], []

    def worker(i):
        try:
            with torch.cuda.stream(streams[i]):
                for _ in range(40):
                    out = models[i](xc, idx=ic, K=Kc, is_training=True)
                    if not all(torch.equal(out[k], serial[i][k]) for k in ("scores", "v3d", "rotmat", "shape", "expression", "transl")):
                        bad[i] += 1
            streams[i].synchronize()
        except Exception as e:          # noqa: BLE001
            errors.append(e)
    for s in streams:
        s.wait_stream(torch.cuda.current_stream())
    threads = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    torch.cuda.synchronize()
    assert not errors, errors
    assert bad == [0, 0], bad
    assert not torch.equal(serial[0]["v3d"], serial[1]["v3d"])


@pytest.mark.parametrize("name", ["vitl_224_train", "vitb_224_train"])
def test_row_statistics_inside_the_class_row_launches(name, smplx_data, mean_params, monkeypatch):
    """Round 6 (csrc/vit_cls.hip): under the token-row map the class-row launches of proj / fc2 carry the patch rows' row statistics and the
    class rows get block sums of their own -- no ln_stats launch.  Against the separate launches (MHMR_CLS_STATS=0, read per call): the
    patch rows' statistics are the same arithmetic (bit-equal), the class rows' variance is E[x^2] - mean^2 from sixteen-column block sums
    instead of the centred form, so the features agree to the 16-bit noise level, not to the last bit; run to run it is bit-reproducible."""
    cfg = dict(make_golden.CASES[name], batch=48, persons=None)      # two image blocks of 24 at 224^2: not tiny -> token-row map + class-row kernel
    model = build(cfg, smplx_data, mean_params, "f16")
    x, _, _ = make_golden.case_inputs(cfg)
    x = x.cuda()
    from multi_hmr_amd import vit
    z1 = model.backbone_features(x).clone()
    P = model._packed
    assert P["fold"] and vit.row_map(P, 48 // model._nsplit(48)) and "cls_pstats" in model._workspace(P, 48)["parts"][0]["bufs"]
    z1b = model.backbone_features(x).clone()
    assert torch.equal(z1, z1b)
    monkeypatch.setenv("MHMR_CLS_STATS", "0")
    z0 = model.backbone_features(x).clone()
    monkeypatch.delenv("MHMR_CLS_STATS")
    r = float((z1.double() - z0.double()).norm() / z0.double().norm())
    assert 0.0 <= r < 5e-4, r          # (measured 3.3e-4 / 1.9e-4 on ViT-L / ViT-B at 224^2, where the class token is one of 257 keys: half the contract)
    assert torch.isfinite(z1).all()
