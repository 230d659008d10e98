"""Anny-variant Multi-HMR: drop-in for ``multi_hmr_anny.multi_hmr.Multi_HMR`` (``demo.py`` imports it as
``ModelAnny``) up to the body model.

Same constructor arguments, same ``state_dict`` keys (``encoder.backbone.*`` = hub DINOv2, ``encoder.mlp_det``,
``encoder.mlp_fov_unique``, ``dec_to_token``, ``decoder.transformer.layers.*``, ``mlp_offset / mlp_pose / mlp_shape /
mlp_dist``, buffers ``dec_pos_emb``, ``init_body_pose``, ``encoder.fov_max``, parameters ``eye``, ``useful_rotmat``), same
``forward(x, K=None, idx=None, is_training=False, det_thresh=0.3, nms_kernel_size=3)``.

Compute (all in ``libmhmr.so``): ``mhmr_vit_forward`` (backbone, patch features + class token), ``mhmr_gemm16`` +
``mhmr_anny_scores`` (detection head, encoder.py:57-58), ``mhmr_anny_camera`` (field of view -> K, encoder.py:47-56),
``mhmr_detect_count/write`` (NMS + threshold, multi_hmr.py:117-124), ``mhmr_gemm16`` patch-scatter epilogue (``dec_to_token`` + the
sin-cos position embedding, multi_hmr.py:127-128), ``mhmr_xattn_layers_forward`` (decoder, via ``anny_hph.HPH``),
``mhmr_linear_f32`` (read-out MLPs) and ``mhmr_anny_decode`` (multi_hmr.py:144-166).

The parametric body model is the third-party ``anny`` package (``requirements.txt:28``, unpinned, not installed here):
pass ``body_model=<callable(pose_parameters=[P,163,4,4], phenotype_kwargs=dict) -> dict(vertices, bone_poses, ...)>`` to get
``v3d / j3d / j2d / v2d`` exactly as ``multi_hmr.py:168-182`` derives them; without it the outputs stop at the parameters."""
from __future__ import annotations

import ctypes as C
import math

import numpy as np
import torch
from torch import nn

from . import _lib, constants, packing, vit
from .anny_hph import HPH
from .model import PATCH, _Encoder, _Holder
from .packing import roundup
from .constants import VIT_CFG

PHENOTYPE_KEYS = ["age", "gender", "weight", "height", "muscle", "proportions"]     # multi_hmr.py:162


class _AnnyEncoder(_Holder):
    """Key names of multi_hmr_anny/encoder.py:16-31."""

    def __init__(self, name, depth_override=None):
        super().__init__()
        cfg = dict(VIT_CFG[name])
        if depth_override is not None:
            cfg["depth"] = depth_override
        self.name = name
        self.backbone = _Encoder(**cfg)
        self.patch_size, self.embed_dim = PATCH, cfg["embed_dim"]
        D = self.embed_dim
        self.mlp_det = nn.Sequential(nn.Linear(D, D), nn.ReLU(), nn.Linear(D, 1))
        self.mlp_fov_unique = nn.Sequential(nn.Linear(D, D), nn.ReLU(), nn.Linear(D, 1))
        self.register_buffer("fov_max", torch.tensor([math.pi]))


class Multi_HMR(nn.Module):
    def __init__(self, img_size=896, backbone="dinov2_vits14", pretrained_backbone=False, xat_dim=512, xat_depth=8, xat_heads=16,
                 xat_dim_head=32, xat_mlp_dim=4 * 512, xat_dropout=0.0, person_center="head", num_betas=11,
                 default_pose_parameterization="root_relative_world", *args, **kwargs):
        super().__init__()
        if kwargs.get("simple_depth_encoding", 1) != 1:
            raise AssertionError("simple_depth_encoding must be 1 (multi_hmr_anny/multi_hmr.py:41)")
        if pretrained_backbone:
            raise RuntimeError("pretrained_backbone=True needs torch.hub (network); load a checkpoint state_dict instead")
        if isinstance(img_size, (list, tuple)):
            img_size = img_size[0]
        self.img_size = img_size
        self.precision = kwargs.get("precision", "f16")
        self.encoder = _AnnyEncoder(backbone, depth_override=kwargs.get("backbone_depth"))
        assert self.img_size % self.encoder.patch_size == 0, "Invalid img size"
        self.patch_size = self.encoder.patch_size
        G = img_size // self.patch_size
        self.register_buffer("dec_pos_emb", torch.from_numpy(constants.anny_sincos_pos_embed(xat_dim, G)).float())
        self.dec_to_token = nn.Linear(self.encoder.embed_dim, xat_dim)
        self.decoder = HPH(dim=xat_dim, depth=xat_depth, heads=xat_heads, dim_head=xat_dim_head, mlp_dim=xat_mlp_dim, dropout=xat_dropout,
                           precision=self.precision)
        D = xat_dim
        self.n_joints, self.num_betas = constants.ANNY_NUM_JOINTS, num_betas
        J = self.n_joints
        self.mlp_offset = nn.Sequential(nn.Linear(D, D), nn.ReLU(), nn.Linear(D, 2))
        self.mlp_pose = nn.Sequential(nn.Linear(D + J * 6, D), nn.ReLU(), nn.Linear(D, J * 6))
        self.mlp_shape = nn.Sequential(nn.Linear(D, D), nn.ReLU(), nn.Linear(D, num_betas))
        self.mlp_dist = nn.Sequential(nn.Linear(D, D), nn.ReLU(), nn.Linear(D, 1))
        self.person_center = person_center
        self.body_model = kwargs.get("body_model")          # third-party `anny` model, optional (see the module docstring)
        self.person_center_idx = None
        if self.body_model is not None:
            self.person_center_idx = list(self.body_model.bone_labels).index(person_center)
        self.eye = nn.Parameter(torch.eye(3).unsqueeze(0), requires_grad=False)
        self.useful_rotmat = nn.Parameter(torch.tensor(constants.ANNY_USEFUL_ROTMAT).unsqueeze(0), requires_grad=False)
        self.register_buffer("init_body_pose", constants.anny_init_body_pose())
        self._packed, self._ws = None, vit.WorkspaceCache()
        for p in self.parameters():
            p.requires_grad_(False)

    # ------------------------------------------------------------------------------------------------------ packing
    def load_state_dict(self, state_dict, strict=True, **kw):
        self.repack()
        return super().load_state_dict(state_dict, strict=strict, **kw)

    def _apply(self, fn, *a, **k):
        self.repack()
        return super()._apply(fn, *a, **k)

    def repack(self):
        """Drop the packed weights and the workspace whose descriptor points into them."""
        self._packed = None
        self._ws.clear()

    def _pack(self, device):
        self._ws.clear()
        enc = self.encoder.backbone
        P = vit.pack_encoder(enc, self.img_size, self.precision, device)
        tdt, Cd = P["tdt"], P["C"]
        P["Kc"] = roundup(Cd, 64)
        f32 = lambda t: t.detach().to(device=device, dtype=torch.float32).contiguous()
        op = lambda t: t.detach().to(device=device, dtype=torch.float32).to(tdt).contiguous()
        e = self.encoder
        D, J = self.dec_to_token.out_features, self.n_joints
        P.update(D=D, det0_w=op(e.mlp_det[0].weight), det0_b=f32(e.mlp_det[0].bias), det2_w=f32(e.mlp_det[2].weight.reshape(-1)),
                 det2_b=f32(e.mlp_det[2].bias), fov0_w=f32(e.mlp_fov_unique[0].weight), fov0_b=f32(e.mlp_fov_unique[0].bias),
                 fov2_w=f32(e.mlp_fov_unique[2].weight), fov2_b=f32(e.mlp_fov_unique[2].bias), norm_w=f32(enc.norm.weight),
                 norm_b=f32(enc.norm.bias), tok_w=op(self.dec_to_token.weight), tok_b=f32(self.dec_to_token.bias),
                 # the patch-scatter epilogue adds pos[1 + n]: pos row 0 is the (unused) class-token entry
                 dec_pos=torch.cat([torch.zeros(1, D, device=device), f32(self.dec_pos_emb)], 0).contiguous(),
                 useful=f32(self.useful_rotmat.reshape(-1)), init_pose=f32(self.init_body_pose.reshape(-1)))
        Kpose = roundup(D + 6 * J, 16)
        pose0 = torch.zeros(D, Kpose, device=device)
        pose0[:, : D + 6 * J] = f32(self.mlp_pose[0].weight)
        P.update(Kpose=Kpose, pose0_w=pose0.contiguous(), pose0_b=f32(self.mlp_pose[0].bias), pose2_w=f32(self.mlp_pose[2].weight),
                 pose2_b=(f32(self.mlp_pose[2].bias) + f32(self.init_body_pose.reshape(-1))).contiguous())
        for name in ("mlp_offset", "mlp_shape", "mlp_dist"):
            m = getattr(self, name)
            P[name] = (f32(m[0].weight), f32(m[0].bias), f32(m[2].weight), f32(m[2].bias))
        self._packed = P
        return P

    def _workspace(self, P, B):
        def extra(P, B, z):
            Mp = roundup(B * P["N"], 128)
            return dict(ctx16=z(Mp, P["Kc"]), hid_det=z(Mp, P["C"]), scores=z(B * P["N"], dtype=torch.float32),
                        logits=z(B * P["N"], dtype=torch.float32), counts=z(B, dtype=torch.int32),
                        dec_emb=z(1 + Mp, P["D"], dtype=torch.float32))
        return self._ws.get(P, B, extra)

    # ------------------------------------------------------------------------------------------------------ forward
    @torch.no_grad()
    def forward(self, x, K=None, idx=None, is_training=False, det_thresh=0.3, nms_kernel_size=3, conf_thresh=None,
                dist_thresh_nms=None, *args, **kwargs):
        if not x.is_cuda:
            raise _lib.MhmrError("multi_hmr_amd.anny_model.Multi_HMR runs only on an MI355X (HIP) tensor; there is no CPU fallback")
        with torch.autocast("cuda", enabled=False):
            return self._forward(x.float().contiguous(), K, idx, is_training, det_thresh, int(nms_kernel_size))

    def _linear(self, L, st, X, W, b, act, M, N, Kd, ldx=None):
        Y = torch.empty(M, N, dtype=torch.float32, device=X.device)
        _lib.check(L.mhmr_linear_f32(X.data_ptr(), ldx or X.shape[1], None, W.data_ptr(), W.shape[1], b.data_ptr(), None, 0, Y.data_ptr(), N,
                                     M, N, Kd, act, st), "mhmr_linear_f32")
        return Y

    def _forward(self, x, K, idx, is_training, det_thresh, nms_kernel_size):
        dev = x.device
        P = self._packed if self._packed is not None and self._packed["device"] == dev else self._pack(dev)
        B = x.shape[0]
        ws = self._workspace(P, B)
        L, st = _lib.lib(), torch.cuda.current_stream(dev).cuda_stream
        Cd, N, G, D, J, nb = P["C"], P["N"], P["G"], P["D"], self.n_joints, self.num_betas
        Mp, Tp = roundup(B * N, 128), ws["Tp"]
        f = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)

        # ---- encoder (encoder.py:33-67): backbone, class token -> field of view -> K, patch-level detection scores ----
        _lib.check(L.mhmr_vit_forward(C.byref(ws["vit_desc"]), x.data_ptr(), ws["feat32"].data_ptr(), ws["ctx16"].data_ptr(), P["Kc"], st),
                   "mhmr_vit_forward")
        cls_rows = ws["resid"].view(B, Tp, Cd)[:, N].contiguous()                      # un-normed class tokens (a [B, C] copy; the class token is the LAST token row of an image)
        cls = f(B, Cd)
        _lib.check(L.mhmr_layernorm_f32(cls_rows.data_ptr(), P["norm_w"].data_ptr(), P["norm_b"].data_ptr(), cls.data_ptr(), B, Cd, 1e-6, st),
                   "mhmr_layernorm_f32")
        h = self._linear(L, st, cls, P["fov0_w"], P["fov0_b"], _lib.ACT_RELU, B, Cd, Cd)
        fov_logit = self._linear(L, st, h, P["fov2_w"], P["fov2_b"], _lib.ACT_NONE, B, 1, Cd)
        fov, K_reg = f(B, 1), f(B, 3, 3)
        _lib.check(L.mhmr_anny_camera(fov_logit.data_ptr(), B, self.img_size, float(self.encoder.fov_max.item()), fov.data_ptr(),
                                      K_reg.data_ptr(), st), "mhmr_anny_camera")
        Kmat = K_reg if K is None else K.to(device=dev, dtype=torch.float32).contiguous()
        _lib.check(L.mhmr_gemm16(ws["ctx16"].data_ptr(), P["Kc"], P["det0_w"].data_ptr(), Cd, Mp, Cd, Cd, P["det0_b"].data_ptr(), None,
                                 ws["hid_det"].data_ptr(), Cd, None, 0, Tp, 1, Mp, _lib.EPI_OP16_RELU, P["dt_id"], st), "mlp_det.0")
        _lib.check(L.mhmr_anny_scores(ws["hid_det"].data_ptr(), Cd, P["det2_w"].data_ptr(), P["det2_b"].data_ptr(), ws["scores"].data_ptr(),
                                      ws["logits"].data_ptr(), B * N, Cd, P["dt_id"], st), "mhmr_anny_scores")
        scores = ws["scores"].view(B, G, G).clone()          # outputs never alias the per-batch workspace (the next forward rewrites it)
        scores_logits = ws["logits"].view(B, G, G).clone()

        # ---- detections (multi_hmr.py:117-124) ----
        if not is_training:
            if idx is None:
                _lib.check(L.mhmr_detect_count(ws["scores"].data_ptr(), B, G, nms_kernel_size, float(det_thresh), ws["counts"].data_ptr(), st),
                           "mhmr_detect_count")
                counts = ws["counts"].cpu()                                                   # the one host sync (torch.where in the reference)
                Pn = int(counts.sum())
                if Pn == 0:
                    return []
                base = (torch.cumsum(counts, 0) - counts).to(torch.int32).to(dev)
                det_b, det_y, det_x = (torch.empty(Pn, dtype=torch.int32, device=dev) for _ in range(3))
                det_s = f(Pn)
                _lib.check(L.mhmr_detect_write(ws["scores"].data_ptr(), B, G, nms_kernel_size, float(det_thresh), base.data_ptr(),
                                               det_b.data_ptr(), det_y.data_ptr(), det_x.data_ptr(), det_s.data_ptr(), st), "mhmr_detect_write")
                idx = (det_b.long(), det_y.long(), det_x.long())
        assert idx is not None, "is_training=True needs the ground-truth idx"
        idx = tuple(t.to(dev).long() for t in idx[:3])
        Pn = int(idx[0].shape[0])
        det_b, det_y, det_x = (t.to(torch.int32).contiguous() for t in idx)

        # ---- decoder tokens (127-128): dec_to_token(feat) + sin-cos position embedding, one GEMM with the row-scatter epilogue ----
        dec_emb = ws["dec_emb"]
        _lib.check(L.mhmr_gemm16(ws["ctx16"].data_ptr(), P["Kc"], P["tok_w"].data_ptr(), Cd, Mp, D, Cd, P["tok_b"].data_ptr(), None,
                                 dec_emb.data_ptr(), D, P["dec_pos"].data_ptr(), N, N, 1, B * N, _lib.EPI_PATCH, P["dt_id"], st),
                   "dec_to_token")
        tokens = dec_emb[: B * N].view(B, N, D)             # the epilogue writes row (m / N) * Tp + m % N with Tp = N here

        # ---- queries / context (130-138) and the decoder (141-142) ----
        values, counts = torch.unique(idx[0], sorted=True, return_counts=True)
        cl = counts.tolist()
        nmax = max(cl)
        q = tokens[idx[0], idx[1] * G + idx[2]]
        queries = torch.zeros(len(cl), nmax, D, device=dev)
        mask = torch.zeros(len(cl), nmax, device=dev)
        o = 0
        for i, c in enumerate(cl):
            queries[i, :c] = q[o: o + c]
            mask[i, :c] = 1
            o += c
        y = self.decoder(x=queries, context=tokens[values], mask=mask)
        y = torch.cat([y[i, :c] for i, c in enumerate(cl)], 0).contiguous()                  # [P, D]

        # ---- read-outs (144-166) ----
        def mlp(name, n_out):
            w0, b0, w2, b2 = P[name]
            return self._linear(L, st, self._linear(L, st, y, w0, b0, _lib.ACT_RELU, Pn, D, D), w2, b2, _lib.ACT_NONE, Pn, n_out, D)
        offset, dist_logit, shape_logit = mlp("mlp_offset", 2), mlp("mlp_dist", 1), mlp("mlp_shape", nb)
        pose_in = torch.zeros(Pn, P["Kpose"], device=dev)
        pose_in[:, :D] = y
        pose_in[:, D: D + 6 * J] = P["init_pose"]
        hpose = self._linear(L, st, pose_in, P["pose0_w"], P["pose0_b"], _lib.ACT_RELU, Pn, D, P["Kpose"])
        rot6d = self._linear(L, st, hpose, P["pose2_w"], P["pose2_b"], _lib.ACT_NONE, Pn, 6 * J, D)      # bias already holds + init_body_pose
        rotmat, rotvec, shape, loc, dist, transl = f(Pn, J, 3, 3), f(Pn, J, 3), f(Pn, nb), f(Pn, 2), f(Pn, 1), f(Pn, 3)
        _lib.check(L.mhmr_anny_decode(rot6d.data_ptr(), P["useful"].data_ptr(), J, shape_logit.data_ptr(), nb, dist_logit.data_ptr(),
                                      offset.data_ptr(), det_b.data_ptr(), det_y.data_ptr(), det_x.data_ptr(), Kmat.data_ptr(), self.patch_size,
                                      Pn, rotmat.data_ptr(), rotvec.data_ptr(), shape.data_ptr(), loc.data_ptr(), dist.data_ptr(),
                                      transl.data_ptr(), st), "mhmr_anny_decode")

        out = {"scores": scores, "scores_logits": scores_logits, "K": Kmat, "K_regressed": K_reg, "fov_regressed": fov, "loc": loc,
               "offset": offset, "dist": dist, "dist_postprocessed": dist_logit, "shape": shape, "rotvec": rotvec, "rotmat": rotmat,
               "transl": transl, "feat": ws["feat32"].view(B, G, G, Cd).clone()}
        if self.body_model is not None:          # multi_hmr.py:160-182, with the caller's anny model
            _shape = {k: shape[:, l] for l, k in enumerate(self.body_model.phenotype_labels) if k in PHENOTYPE_KEYS}
            homo = torch.zeros(Pn, J, 4, 4, device=dev)
            homo[:, :, :3, :3] = rotmat
            homo[:, :, 3, 3] = 1
            bm = self.body_model(pose_parameters=homo, phenotype_kwargs=_shape)
            v3d, j3d = bm["vertices"], bm["bone_poses"][:, :, :3, -1]
            center = j3d[:, [self.person_center_idx]]
            v3d, j3d = v3d - center + transl.unsqueeze(1), j3d - center + transl.unsqueeze(1)
            proj = lambda p: (p @ Kmat[idx[0]].transpose(1, 2))[..., :2] / (p @ Kmat[idx[0]].transpose(1, 2))[..., 2:]
            out.update(v3d=v3d, j3d=j3d, v2d=proj(v3d), j2d=proj(j3d), transl_pelvis=j3d[:, [0]], blendshape_coeffs=bm.get("blendshape_coeffs"))
        if is_training:
            return out
        persons = []
        for i in range(Pn):
            person = {"K": Kmat[idx[0]][i], "K_regressed": K_reg[idx[0]][i], "loc": loc[i], "transl": transl[i], "rotvec": rotvec[i],
                      "rotmat": rotmat[i], "shape": shape[i], "fov": fov}
            for k in ("transl_pelvis", "v3d", "j3d", "j2d"):
                if k in out:
                    person[k] = out[k][i]
            persons.append(person)
        return sorted(persons, key=lambda p: p["transl"][2].item())          # closest to the camera first (multi_hmr.py:235)
