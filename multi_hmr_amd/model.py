"""Drop-in for the reference ``model.Model`` (naver/multi-hmr model.py:30-349) on MI355X.

Same constructor signature (extra kwargs swallowed, model.py:33-50), same ``nn.Module`` protocol and
``state_dict`` key names, same ``forward`` signature and return values
(model.py:205-349) -- but ``forward`` is a sequence of calls into ``libmhmr.so`` (hand-written gfx950 HIP
kernels behind the C ABI of include/mhmr.h).  PyTorch only owns device memory and the stream.  There is no
CPU / eager fallback: without the library, or on a non-GPU tensor, ``forward`` raises.
"""
from __future__ import annotations

import ctypes as C
import math
import os
import threading

import numpy as np
import torch
from torch import nn

from . import _lib, packing, vit
from .packing import roundup
from .constants import SMPLX_JOINT_NAMES, VIT_CFG

PATCH = 14
SMPLX_DIR = "models"                           # reference utils/constants.py:7
MEAN_PARAMS = "models/smpl_mean_params.npz"    # reference utils/constants.py:8


# ------------------------------------------------------------------------------------------------------------
# Parameter holders: plain nn.Modules that only exist to give state_dict() the reference's key names.
# ------------------------------------------------------------------------------------------------------------
class _Holder(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("parameter holder; compute happens in libmhmr.so")


class _LayerScale(_Holder):
    def __init__(self, dim):
        super().__init__()
        self.gamma = nn.Parameter(torch.ones(dim))


class _VitAttn(_Holder):
    def __init__(self, dim):
        super().__init__()
        self.qkv = nn.Linear(dim, 3 * dim)
        self.proj = nn.Linear(dim, dim)


class _VitMlp(_Holder):
    def __init__(self, dim):
        super().__init__()
        self.fc1 = nn.Linear(dim, 4 * dim)
        self.fc2 = nn.Linear(4 * dim, dim)


class _VitBlock(_Holder):
    def __init__(self, dim):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = _VitAttn(dim)
        self.ls1 = _LayerScale(dim)
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = _VitMlp(dim)
        self.ls2 = _LayerScale(dim)


class _PatchEmbed(_Holder):
    def __init__(self, dim):
        super().__init__()
        self.proj = nn.Conv2d(3, dim, kernel_size=PATCH, stride=PATCH)


class _Encoder(_Holder):
    """Key names of torch.hub dinov2_vit{s,b,l}14 (reference blocks/dinov2.py:12)."""

    def __init__(self, embed_dim, depth, num_heads):
        super().__init__()
        self.embed_dim, self.num_heads, self.patch_size = embed_dim, num_heads, PATCH
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, 1 + 37 * 37, embed_dim))
        self.mask_token = nn.Parameter(torch.zeros(1, embed_dim))
        self.patch_embed = _PatchEmbed(embed_dim)
        self.blocks = nn.ModuleList([_VitBlock(embed_dim) for _ in range(depth)])
        self.norm = nn.LayerNorm(embed_dim, eps=1e-6)
        nn.init.trunc_normal_(self.pos_embed, std=0.02)
        nn.init.normal_(self.cls_token, std=1e-6)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=0.02)
                nn.init.zeros_(m.bias)


class Dinov2Backbone(_Holder):
    """Reference blocks/dinov2.py:8-26 (name, encoder, patch_size, embed_dim attributes)."""

    def __init__(self, name="dinov2_vitb14", pretrained=False, depth_override=None):
        super().__init__()
        if pretrained:
            raise RuntimeError("pretrained_backbone=True needs torch.hub (network); load a checkpoint state_dict instead")
        cfg = dict(VIT_CFG[name])
        if depth_override is not None:
            cfg["depth"] = depth_override
        self.name = name
        self.encoder = _Encoder(**cfg)
        self.patch_size, self.embed_dim = PATCH, cfg["embed_dim"]


class _Fn(_Holder):
    pass


class _PreNorm(_Holder):
    def __init__(self, dim, fn):
        super().__init__()
        self.norm = nn.LayerNorm(dim)
        self.fn = fn


def _sa(dim, inner):
    f = _Fn()
    f.to_qkv = nn.Linear(dim, inner * 3, bias=False)
    f.to_out = nn.Sequential(nn.Linear(inner, dim), nn.Dropout(0.0))
    return f


def _ca(dim, ctx, inner):
    f = _Fn()
    f.to_kv = nn.Linear(ctx, inner * 2, bias=False)
    f.to_q = nn.Linear(dim, inner, bias=False)
    f.to_out = nn.Sequential(nn.Linear(inner, dim), nn.Dropout(0.0))
    return f


def _ff(dim, hidden):
    f = _Fn()
    f.net = nn.Sequential(nn.Linear(dim, hidden), nn.GELU(), nn.Dropout(0.0), nn.Linear(hidden, dim), nn.Dropout(0.0))
    return f


class _CrossAttnStack(_Holder):
    def __init__(self, dim, depth, heads, dim_head, mlp_dim, context_dim):
        super().__init__()
        inner = heads * dim_head
        self.layers = nn.ModuleList([nn.ModuleList([_PreNorm(dim, _sa(dim, inner)), _PreNorm(dim, _ca(dim, context_dim, inner)),
                                                    _PreNorm(dim, _ff(dim, mlp_dim))]) for _ in range(depth)])


class _TransformerDecoder(_Holder):
    """Key names of blocks/cross_attn_transformer.py:302-359."""

    def __init__(self, token_dim, dim, depth, heads, mlp_dim, dim_head, context_dim):
        super().__init__()
        self.to_token_embedding = nn.Linear(token_dim, dim)
        self.pos_embedding = nn.Parameter(torch.randn(1, 1, dim))
        self.transformer = _CrossAttnStack(dim, depth, heads, dim_head, mlp_dim, context_dim)


class HPH(_Holder):
    """Parameters / buffers of the reference HPH (model.py:352-477)."""

    def __init__(self, context_dim, dim, depth, heads, mlp_dim, dim_head, at_token_res, num_betas, mean_params):
        super().__init__()
        assert num_betas in (10, 11) and dim_head == 32
        self.num_betas, self.res, self.depth, self.heads = num_betas, at_token_res, (depth,), (heads,)
        self.npose = 6 * 53
        self.transformer = _TransformerDecoder(self.npose + num_betas + 3 + context_dim, dim, depth, heads, mlp_dim, dim_head, context_dim)
        self.decpose, self.decshape = nn.Linear(dim, self.npose), nn.Linear(dim, num_betas)
        self.deccam, self.decexpression = nn.Linear(dim, 3), nn.Linear(dim, 10)
        # set_smpl_init (model.py:440-477)
        init_body_pose = torch.eye(3).reshape(1, 3, 3).repeat(53, 1, 1)[:, :, :2].flatten(1).reshape(1, -1)
        init_body_pose[:, : 24 * 6] = torch.from_numpy(np.asarray(mean_params["pose"][:], dtype=np.float32)).float()
        init_betas = torch.from_numpy(np.asarray(mean_params["shape"], dtype=np.float32)).unsqueeze(0)
        init_cam = torch.from_numpy(np.asarray(mean_params["cam"], dtype=np.float32)).unsqueeze(0)
        init_betas_kid = torch.cat([init_betas, torch.zeros_like(init_betas[:, [0]])], 1)
        init_expression = 0.0 * init_betas.clone()
        if num_betas == 11:
            init_betas = torch.cat([init_betas, torch.zeros_like(init_betas[:, :1])], 1)
        for n, v in (("init_body_pose", init_body_pose), ("init_betas", init_betas), ("init_betas_kid", init_betas_kid),
                     ("init_cam", init_cam), ("init_expression", init_expression)):
            self.register_buffer(n, v)
        # init_learned_queries (model.py:424-438)
        for n in ("cross_queries_x", "cross_queries_y", "cross_values_x", "cross_values_y"):
            p = nn.Parameter(torch.zeros(at_token_res, context_dim))
            nn.init.normal_(p, std=0.2)
            setattr(self, n, p)


class _BodyModel:
    """What callers read from ``smpl_layer[...].bm_x`` (demo.py:310: ``.faces``)."""

    def __init__(self, faces, num_vertices):
        self.faces, self.num_vertices = faces, num_vertices


class SMPL_Layer(_Holder):
    """Reference blocks/smpl_layer.py:18-45 attributes; the arithmetic lives in lbs.hip."""

    def __init__(self, data, type="smplx", gender="neutral", num_betas=10, kid=False, person_center=None):
        super().__init__()
        assert type == "smplx"
        self.type, self.kid, self.num_betas = type, kid, num_betas
        self.joint_names = list(SMPLX_JOINT_NAMES)
        self.person_center = person_center
        self.person_center_idx = self.joint_names.index(person_center) if person_center is not None else None
        self.bm_x = _BodyModel(np.asarray(data["f"], dtype=np.int64), int(np.asarray(data["v_template"]).shape[0]))


def _load_smplx_data(explicit):
    if explicit is not None:
        return explicit
    path = os.path.join(SMPLX_DIR, "smplx", "SMPLX_NEUTRAL.npz")
    if not os.path.isfile(path):
        raise FileNotFoundError(f"{path} not found (reference blocks/smpl_layer.py:38 loads it through smplx.create); "
                                "pass smplx_data=<dict with the SMPLX_NEUTRAL.npz arrays> to Model(...)")
    return dict(np.load(path, allow_pickle=True))


def _load_mean_params(explicit):
    if explicit is not None:
        return explicit
    if not os.path.isfile(MEAN_PARAMS):
        raise FileNotFoundError(f"{MEAN_PARAMS} not found (reference model.py:442); pass mean_params=<dict pose/shape/cam>")
    return dict(np.load(MEAN_PARAMS))


class Model(nn.Module):
    """A ViT backbone followed by the HPH head and the SMPL-X layer, on hand-written gfx950 kernels."""

    def __init__(self, backbone="dinov2_vitb14", pretrained_backbone=False, img_size=896, camera_embedding="geometric",
                 camera_embedding_num_bands=16, camera_embedding_max_resolution=64, nearness=True, xat_depth=2,
                 xat_num_heads=8, dict_smpl_layer=None, person_center="head", clip_dist=True, num_betas=10, *args, **kwargs):
        super().__init__()
        if isinstance(img_size, (list, tuple)):
            img_size = img_size[0]
        # clip_dist is stored as a 1-tuple exactly as the reference does (model.py:56): `if self.clip_dist:` (model.py:200) is then
        # always true, so the reference clamps the distance to [0, 50] whatever the argument says -- and so does hph_decode_kernel
        self.img_size, self.nearness, self.clip_dist = img_size, nearness, (clip_dist,)
        self.xat_depth, self.xat_num_heads, self.num_betas = xat_depth, xat_num_heads, num_betas
        # MFMA operand format of the backbone: "f16" | "bf16" | "f16x3" (operand pairs, three products per term, fp32 attention: ~5x the
        # time, for checkpoints a single 16-bit rounding per operand does not survive) | "auto" (the default: "f16" unless the
        # weights' attention-logit statistic says otherwise, vit.logit_gain -- decided once, at pack time; `packed_precision` tells)
        self.precision = kwargs.get("precision", os.environ.get("MHMR_PRECISION", "auto"))
        if self.precision not in vit.PRECISIONS:
            raise ValueError(f"precision must be one of {list(vit.PRECISIONS)}")
        # low-half weight passes of the V / output projections (vit.DEFAULT_WLO; "" = none): what puts every output within 1e-3
        self.wlo = kwargs.get("wlo", os.environ.get("MHMR_WLO"))
        vit.parse_wlo(vit.DEFAULT_WLO if self.wlo is None else self.wlo, 64)       # fail early on a malformed spec
        # LayerNorm folded into the neighbouring GEMMs (None = wherever the shapes allow it: vit.fold_eligible)
        self.lnfold = kwargs.get("lnfold")
        self.backbone = Dinov2Backbone(backbone, pretrained=pretrained_backbone, depth_override=kwargs.get("backbone_depth"))
        self.embed_dim, self.patch_size = self.backbone.embed_dim, self.backbone.patch_size
        assert self.img_size % self.patch_size == 0, "Invalid img size"
        self.fovn = 60
        # camera embedding (reference model.py:67-83).  None is accepted by the constructor exactly as the reference accepts it
        # (camera_embed_dim = 0, no `camera` module) -- and, as in the reference, such a model cannot run forward(): model.py:262 calls
        # embedd_camera() unconditionally, which needs self.camera (AttributeError there, AttributeError here).
        self.camera_embedding = camera_embedding
        self.camera_embed_dim, self._camera_num_bands = 0, 16
        self._camera_max_resolution = camera_embedding_max_resolution
        if camera_embedding is not None:
            if camera_embedding != "geometric":
                raise NotImplementedError("Only geometric camera embedding is implemented")
            if not 1 <= int(camera_embedding_num_bands) <= 20:
                raise NotImplementedError("camera_embedding_num_bands must be 1 ... 20 (the camera embedding kernel writes the 3 + 6 bands "
                                          "channels with one thread each, <= 128 columns behind the features); the released checkpoints use 16")
            self._camera_num_bands = int(camera_embedding_num_bands)
            self.camera_embed_dim = 3 + 2 * 3 * self._camera_num_bands       # 99 for 16 bands (FourierPositionEncoding.channels)
        self.mlp_classif = nn.Sequential(nn.Linear(self.embed_dim, self.embed_dim), nn.ReLU(), nn.Linear(self.embed_dim, 1))
        self.mlp_offset = nn.Sequential(nn.Linear(self.embed_dim, self.embed_dim), nn.ReLU(), nn.Linear(self.embed_dim, 2))
        self.nrot = 53
        self._smplx_data = _load_smplx_data(kwargs.get("smplx_data"))
        self.smpl_layer = nn.ModuleDict({f"neutral_{nb}": SMPL_Layer(self._smplx_data, num_betas=nb, person_center=person_center)
                                         for nb in (10, 11)})
        self.x_attention_head = HPH(context_dim=self.embed_dim + self.camera_embed_dim, dim=1024, depth=xat_depth,
                                    heads=xat_num_heads, mlp_dim=1024, dim_head=32, at_token_res=img_size // PATCH,
                                    num_betas=num_betas, mean_params=_load_mean_params(kwargs.get("mean_params")))
        self._packed = None                 # tensors + descriptors of the current (device, precision, parameters)
        self._ws = vit.WorkspaceCache()     # the workspaces of the most recent batch sizes
        self._person_cap = {}               # batch size -> person-row capacity of the next inference forward (fixed-capacity heads)
        self._streams = None                # side streams + events of the split backbone (_run_backbone)
        # workspaces, side streams / events and the capacity dict belong to the instance: concurrent forwards of ONE Model from two host
        # threads would share them, so a forward holds this lock from start to end (one Model per thread for concurrency)
        self._lock = threading.RLock()
        self.split = kwargs.get("split")    # image blocks of the backbone on streams of their own (None: MHMR_SPLIT; 0 / unset = the automatic rule of _nsplit)
        for p in self.parameters():
            p.requires_grad_(False)

    # -------------------------------------------------------------------------------------------------- packing
    def load_state_dict(self, state_dict, strict=True, **kw):
        self.repack()
        return super().load_state_dict(state_dict, strict=strict, **kw)

    def _apply(self, fn, *a, **k):
        self.repack()
        return super()._apply(fn, *a, **k)

    def repack(self):
        """Drop the packed weights AND the workspace whose descriptor points into them (call after mutating parameters in place;
        load_state_dict / .to() do it automatically)."""
        self._packed = None
        self._ws.clear()

    def _pack(self, device):
        self._ws.clear()
        P = vit.pack_encoder(self.backbone.encoder, self.img_size, self.precision, device, self.wlo, self.lnfold)
        dt_id, tdt = P["dt_id"], P["tdt"]
        C, G, N = P["C"], P["G"], P["N"]
        f32 = lambda t: t.detach().to(device=device, dtype=torch.float32).contiguous()
        op = lambda t: t.detach().to(device=device, dtype=torch.float32).to(tdt).contiguous()
        keep = P["keep"]      # tensors referenced by raw pointers in the descriptors

        def k(t):
            keep.append(t)
            return t.data_ptr()

        # heads
        E, nbands = self.camera_embed_dim, self._camera_num_bands
        Cc = C + E
        Kc = roundup(Cc, 64)
        P["E"], P["nbands"] = E, nbands
        P["Cc"], P["Kc"] = Cc, Kc
        P["cls0_w"], P["cls0_b"] = op(self.mlp_classif[0].weight), f32(self.mlp_classif[0].bias)
        P["cls2_w"], P["cls2_b"] = f32(self.mlp_classif[2].weight.reshape(-1)), f32(self.mlp_classif[2].bias)
        # blocks/camera_embed.py:46: linspace(1, max_resolution / 2, num_bands) per ray component
        P["freq"] = torch.stack([torch.linspace(1.0, self._camera_max_resolution / 2, nbands) for _ in range(3)]).to(device).contiguous()

        # HPH
        hp = self.x_attention_head
        heads, depth, dim, mlp, nb = self.xat_num_heads, self.xat_depth, 1024, 1024, self.num_betas
        inner = 32 * heads
        n_kv = roundup(2 * inner, 128)
        Ktok = roundup(Cc + 318 + nb + 3, 16)
        tok_w = torch.zeros(dim, Ktok, device=device)
        tok_w[:, : Cc + 321 + nb] = f32(hp.transformer.to_token_embedding.weight)
        tok_b = f32(hp.transformer.to_token_embedding.bias) + f32(hp.transformer.pos_embedding[0, 0])
        layers = (_lib.HphLayer * depth)()
        for l, (sa, ca, ff) in enumerate(hp.transformer.transformer.layers):
            y = layers[l]
            y.ln_sa_w, y.ln_sa_b, y.to_qkv = k(f32(sa.norm.weight)), k(f32(sa.norm.bias)), k(f32(sa.fn.to_qkv.weight))
            y.sa_out_w, y.sa_out_b = k(f32(sa.fn.to_out[0].weight)), k(f32(sa.fn.to_out[0].bias))
            y.ln_ca_w, y.ln_ca_b = k(f32(ca.norm.weight)), k(f32(ca.norm.bias))
            kvw = torch.zeros(n_kv, Kc, device=device)
            kvw[: 2 * inner, :Cc] = f32(ca.fn.to_kv.weight)
            y.to_kv16, y.to_q = k(kvw.to(tdt).contiguous()), k(f32(ca.fn.to_q.weight))
            y.ca_out_w, y.ca_out_b = k(f32(ca.fn.to_out[0].weight)), k(f32(ca.fn.to_out[0].bias))
            y.ln_ff_w, y.ln_ff_b = k(f32(ff.norm.weight)), k(f32(ff.norm.bias))
            y.ff1_w, y.ff1_b = k(f32(ff.fn.net[0].weight)), k(f32(ff.fn.net[0].bias))
            y.ff2_w, y.ff2_b = k(f32(ff.fn.net[3].weight)), k(f32(ff.fn.net[3].bias))
        dec_w = torch.cat([f32(m.weight) for m in (hp.decpose, hp.decshape, hp.deccam, hp.decexpression)], 0).contiguous()
        dec_b = (torch.cat([f32(m.bias) for m in (hp.decpose, hp.decshape, hp.deccam, hp.decexpression)], 0) +
                 torch.cat([f32(hp.init_body_pose[0]), f32(hp.init_betas[0]), f32(hp.init_cam[0]), f32(hp.init_expression[0])], 0))
        init_tail = torch.cat([f32(hp.init_body_pose[0]), f32(hp.init_betas[0]), f32(hp.init_cam[0])], 0).contiguous()
        P["hph"] = dict(layers=layers, heads=heads, depth=depth, dim=dim, mlp=mlp, nb=nb, inner=inner, n_kv=n_kv, Ktok=Ktok,
                        Ndec=318 + nb + 13,
                        off1_w=k(f32(self.mlp_offset[0].weight)), off1_b=k(f32(self.mlp_offset[0].bias)),
                        off2_w=k(f32(self.mlp_offset[2].weight)), off2_b=k(f32(self.mlp_offset[2].bias)),
                        cq_x=k(f32(hp.cross_queries_x)), cq_y=k(f32(hp.cross_queries_y)), cv_x=k(f32(hp.cross_values_x)),
                        cv_y=k(f32(hp.cross_values_y)), init_tail=k(init_tail), tok_w=k(tok_w.contiguous()), tok_b=k(tok_b.contiguous()),
                        dec_w=k(dec_w), dec_b=k(dec_b.contiguous()))
        assert n_kv == 2 * inner, "xat_num_heads must be even (to_kv rows are tiled by 128)"

        # SMPL-X
        layer = self.smpl_layer[f"neutral_{nb}"]
        P["lbs"] = packing.pack_smplx(self._smplx_data, nb, device, layer.person_center_idx if layer.person_center_idx is not None else -1)
        P["lbs_struct"] = packing.lbs_consts_struct(P["lbs"])
        P["lbs_sync"] = torch.zeros(1 + 160, dtype=torch.int32, device=device)      # mhmr_lbs_forward_fused: ticket + ready flags
        self._packed = P
        return P

    @property
    def packed_precision(self):
        """The operand format the current pack runs ('f16', 'bf16' or 'f16x3'); None before the first forward / after a repack."""
        return self._packed["precision"] if self._packed is not None else None

    def _nsplit(self, B):
        """Image blocks of the batch that run the backbone on streams of their own (``split=`` / MHMR_SPLIT; 0 or unset = automatic).
        Images never interact inside the backbone, so the blocks are independent launch sequences: the persistent GEMM / attention
        launches of one block start on the CUs that the tail of the other block's launch leaves idle.  That pays when a launch is only
        a few rounds of 256 tiles long -- measured (profiles/r04_split_colgroup_ab.txt, r04_session_c_*.txt): config 5 (1288^2, 8 images:
        4.2 rounds in the N = 1024 linears) +4.2 %, config 2 (ViT-S 672^2, 16 images) +5.2 %, config 3 +2 %; the headline (896^2, 32
        images: 8 exact rounds) +0.3 %, where it is left off so that every kernel has the chip to itself (per-kernel hipEvent / rocprofv3
        durations then mean what they say).  Automatic rule: two blocks when the batch is even, >= 8 images, and the narrowest block
        linear is under six rounds.  Results do not depend on it as long as the blocks select the same kernels as the whole batch would
        (every kernel is batch-invariant: tests/test_gpu_fullsize.py, test_backbone_image_blocks_on_side_streams_change_nothing); where the
        block size changes the kernel choice or the row padding (ViT-S with one image per block: the 128x128 kernel and Tp a multiple of
        128) outputs agree to rounding order, not bit for bit -- and since the automatic rule is the default for B >= 8, such a batch can
        differ in the last bits from a run with split=1."""
        n = self.split if self.split is not None else int(os.environ.get("MHMR_SPLIT", "0"))
        if n <= 0:
            T = (self.img_size // PATCH) ** 2 + 1
            rounds = (B * roundup(T, 64) / 256.0) * max(self.embed_dim // 256, 1) / 256.0
            n = 2 if (B >= 8 and B % 2 == 0 and rounds < 6.0) else 1
        while n > 1 and (B % n or B // n < 1):
            n -= 1
        return n

    def _workspace(self, P, B):
        def extra(P, B, z):
            Mp = roundup(B * P["N"], 128)
            return dict(ctx16=z(Mp, P["Kc"]), zK=z(B * P["N"], P["E"], dtype=torch.float32), scores=z(B * P["N"], dtype=torch.float32),
                        counts=z(B, dtype=torch.int32), kv=z(Mp, P["hph"]["n_kv"], dtype=torch.float32),
                        hid_cls=z(Mp, P["C"]))
        return self._ws.get(P, B, extra, self._nsplit(B))

    # -------------------------------------------------------------------------------------------------- forward
    def _side_streams(self, dev, n):
        """n - 1 side streams (+ one fork and n - 1 join events) of this model on `dev`, created once."""
        key = (dev.index, n)
        if self._streams is None or self._streams[0] != key:
            self._streams = (key, [torch.cuda.Stream(device=dev) for _ in range(n - 1)], torch.cuda.Event(),
                             [torch.cuda.Event() for _ in range(n - 1)])
        return self._streams[1:]

    def _run_backbone(self, P, ws, x):
        """mhmr_vit_forward over the image blocks of the workspace: block 0 on the caller's stream, the others on side streams that
        fork from it and join it again (so the caller's stream semantics hold: everything enqueued before is visible to every block,
        everything enqueued after sees every block's output)."""
        L = _lib.lib()
        dev, parts, Kc, N, Cd = x.device, ws["parts"], P["Kc"], P["N"], P["C"]
        main = torch.cuda.current_stream(dev)
        esz_ctx = ws["ctx16"].element_size()
        img_elems = x.shape[1] * x.shape[2] * x.shape[3]

        def launch(part, stream):
            i0 = part["img0"]
            _lib.check(L.mhmr_vit_forward(C.byref(part["desc"]), x.data_ptr() + i0 * img_elems * 4, ws["feat32"].data_ptr() + i0 * N * Cd * 4,
                                          ws["ctx16"].data_ptr() + i0 * N * Kc * esz_ctx, Kc, stream.cuda_stream), "mhmr_vit_forward")

        if len(parts) == 1 or os.environ.get("MHMR_SPLIT_SEQ"):        # (MHMR_SPLIT_SEQ: the blocks one after the other on one stream, A/B only)
            for part in parts:
                launch(part, main)
            return
        streams, fork, joins = self._side_streams(dev, len(parts))
        fork.record(main)
        for part, st, ev in zip(parts[1:], streams, joins):
            st.wait_event(fork)
            launch(part, st)
            ev.record(st)
        launch(parts[0], main)
        for ev in joins:
            main.wait_event(ev)

    def backbone_features(self, x):
        """[B,3,S,S] -> [B,N,C] fp32 patch features (reference blocks/dinov2.py:16-26), as a view of the workspace."""
        with self._lock:                      # (workspaces belong to the instance: same lock as forward(); round-5 advisor finding)
            P, ws, stream = self._prepare(x)
            self._run_backbone(P, ws, x)
            return ws["feat32"].view(x.shape[0], P["N"], P["C"])

    def _prepare(self, x):
        if self.camera_embedding is None:
            raise AttributeError("'Model' object has no attribute 'camera' (camera_embedding=None: the reference's forward fails the same "
                                 "way at model.py:262 -> :183)")
        if not x.is_cuda:
            raise _lib.MhmrError("multi_hmr_amd.Model.forward runs only on an MI355X (HIP) tensor; there is no CPU fallback")
        if x.dim() != 4 or x.shape[1] != 3 or x.shape[2] != self.img_size or x.shape[3] != self.img_size:
            raise ValueError(f"x must be [B,3,{self.img_size},{self.img_size}]")
        P = self._packed
        if P is None or P["device"] != x.device:
            P = self._pack(x.device)
        ws = self._workspace(P, x.shape[0])
        return P, ws, torch.cuda.current_stream(x.device).cuda_stream

    @torch.no_grad()
    def forward(self, x, idx=None, det_thresh=0.3, nms_kernel_size=3, K=None, is_training=False, *args, **kwargs):
        """Same contract as the reference ``Model.forward`` (model.py:205-349): inference -> list of per-person dicts
        (empty list if nobody is detected); ``is_training=True`` (needs ``idx``) -> dict of batched tensors.
        Extension used by ``distributed.forward_sharded``: ``return_image_index=True`` (inference only) -> (persons, image id [P]);
        ``return_batched=True`` (inference only) -> (dict of batched tensors [P, ...], image id [P]) instead of the per-person list."""
        with self._lock, torch.autocast("cuda", enabled=False):     # demo.forward_model wraps us in fp16 autocast (demo.py:117)
            return self._forward(x.float().contiguous(), idx, det_thresh, nms_kernel_size, K, is_training,
                                 bool(kwargs.get("return_image_index", False)), bool(kwargs.get("return_batched", False)))

    supports_image_index = True
    supports_batched = True
    #: keys of a person dict, in the reference's order (model.py:330-346)
    PERSON_KEYS = ("scores", "loc", "transl", "transl_pelvis", "rotvec", "expression", "shape", "v3d", "j3d", "j2d")

    def _forward(self, x, idx, det_thresh, nms_kernel_size, K, is_training, with_ids=False, batched=False):
        L = _lib.lib()
        P, ws, stream = self._prepare(x)
        dev, B, G = x.device, x.shape[0], P["G"]
        K = K.to(device=dev, dtype=torch.float32).contiguous()
        assert K.shape == (B, 3, 3)

        self._front(P, ws, x, K, stream)
        scores = ws["scores"].view(B, G, G, 1)

        if is_training:
            # the caller's idx (training hook, model.py:150-151); persons sorted by image as torch.where leaves them
            assert idx is not None
            idx = tuple(i.to(dev) for i in idx)
            Pn = int(idx[0].shape[0])
            out = {"scores": scores.clone()}
            if Pn == 0:
                return out
            det = torch.stack([idx[0], idx[1], idx[2]]).to(torch.int32).contiguous()
            gstart_t, chunks_t, info, ngc, ncc = self._tables(B, Pn, dev)
            _lib.check(L.mhmr_person_groups(None, det[0].data_ptr(), Pn, B, Pn, None, gstart_t.data_ptr(), ngc, chunks_t.data_ptr(), ncc,
                                            info.data_ptr(), stream), "mhmr_person_groups")
            heads = self._heads(P, ws, K, det, Pn, gstart_t, ngc, chunks_t, ncc, None, stream)
            out.update({n: t for n, t in heads.items() if n not in ("scores", "_flat")})      # (scores here = the [B, G, G, 1] map, model.py:349)
            return out

        # NMS + threshold + ordered compaction (model.py:141-149)
        thr = float(det_thresh[0] if isinstance(det_thresh, list) else det_thresh)
        k = int(nms_kernel_size)
        self._count(ws, B, G, k, thr, stream)

        # Fixed capacity: the heads are enqueued for `cap` person rows (a little above the previous batch's count; rows behind the real
        # persons are padding that the kernels compute and nobody reads), and the ONE host synchronisation -- the person count, which the
        # reference takes in torch.where in the MIDDLE of its forward (model.py:146) -- comes after the last launch, when it costs the GPU
        # nothing.  The first call, and a batch with more persons than the capacity, take the count first (exact sizes).
        cap = self._person_cap.get(B)
        o = persons = None
        if cap is not None:
            o, det, info, persons = self._detect_and_heads(P, ws, K, k, thr, cap, not batched, stream)
            Pn = int(info[3].item())                       # the host sync
            if Pn > cap:
                o = persons = None
        else:
            Pn = int(ws["counts"].sum().item())            # the host sync (first call)
        # next capacity: an eighth above this batch's count (a batch that detects more than that re-runs its heads at the exact size,
        # which costs what the reference's mid-forward count always costs); whole 16-person groups of the SMPL-X layer
        self._person_cap[B] = roundup(max(Pn + max(Pn // 8, 8), 16), 16)
        if Pn == 0:
            return (([] if not batched else {}), torch.zeros(0, dtype=torch.int32, device=dev)) if (with_ids or batched) else []
        if o is None:
            o, det, info, persons = self._detect_and_heads(P, ws, K, k, thr, Pn, not batched, stream)
        ids = det[0][:Pn]
        if batched:
            return {n: o[n][:Pn] for n in self.PERSON_KEYS}, ids
        persons = persons[:Pn]
        return (persons, ids) if with_ids else persons

    # ---- the pieces of the inference forward (shared with graphed.GraphedForward, which records them into a hipGraph once)
    def _front(self, P, ws, x, K, stream):
        """Steps 1-3 of the forward: backbone, camera embedding, detection scores -> ws["feat32"], ws["ctx16"], ws["zK"], ws["scores"]."""
        L = _lib.lib()
        B, G, N, Cdim, Kc, dt = x.shape[0], P["G"], P["N"], P["C"], P["Kc"], P["dt_id"]
        Mp = ws["ctx16"].shape[0]
        # 1. backbone (model.py:229) -> feat32 + 16-bit context operand
        self._run_backbone(P, ws, x)
        # 2. camera embedding (model.py:262) -> zK + context operand columns C..C+98
        _lib.check(L.mhmr_camera_embed(K.data_ptr(), P["freq"].data_ptr(), B, G, PATCH, ws["zK"].data_ptr(), ws["ctx16"].data_ptr(), Kc,
                                       Cdim, dt, P["nbands"], stream), "mhmr_camera_embed")
        # 3. detection scores (model.py:135): mlp_classif.0 + ReLU on MFMA, then the C->1 read-out + clamped sigmoid
        _lib.check(L.mhmr_gemm16(ws["ctx16"].data_ptr(), Kc, P["cls0_w"].data_ptr(), Cdim, Mp, Cdim, Cdim, P["cls0_b"].data_ptr(), None,
                                 ws["hid_cls"].data_ptr(), Cdim, None, 0, 128, 1, Mp, _lib.EPI_OP16_RELU, dt, stream), "mhmr_gemm16(classif)")
        _lib.check(L.mhmr_detect_scores(ws["hid_cls"].data_ptr(), Cdim, P["cls2_w"].data_ptr(), P["cls2_b"].data_ptr(), ws["scores"].data_ptr(),
                                        B * N, Cdim, dt, stream), "mhmr_detect_scores")

    def _count(self, ws, B, G, k, thr, stream):
        """Per-image person counts of the NMS + threshold (model.py:141-146) -> ws["counts"]."""
        _lib.check(_lib.lib().mhmr_detect_count(ws["scores"].data_ptr(), B, G, k, thr, ws["counts"].data_ptr(), stream), "mhmr_detect_count")

    @staticmethod
    def _tables(B, cap, dev, with_det=False):
        """The person set's bookkeeping tables -- per-image write offsets, the ragged query groups of the decoder (rebatch / pad_to_max
        semantics, utils/tensor_manip.py:7-45, without the padding) -- are filled ON THE DEVICE (mhmr_person_groups): the training hook
        needs no host round trip at all, inference reads the person count back AFTER the whole forward is enqueued.
        One zeroed int32 block (one fill launch, not one per table) -> gstart [ngc + 1], chunks [3 ncc], info [4], ngc, ncc and, with
        ``with_det``, the detection triple det [3, cap] and the per-image write offsets base [B] of the inference path."""
        ngc, ncc = min(B, cap), cap // 8 + min(B, cap)
        sizes = [ngc + 1, 3 * max(ncc, 1), 4] + ([3 * cap, B] if with_det else [])
        offs = [0]
        for n in sizes:
            offs.append(offs[-1] + -(-n // 16) * 16)              # 64-byte aligned pieces
        blk = torch.zeros(offs[-1], dtype=torch.int32, device=dev)
        pieces = [blk[a:a + n] for a, n in zip(offs, sizes)]
        out = (pieces[0], pieces[1], pieces[2], ngc, ncc)
        return out + (pieces[3].view(3, cap), pieces[4]) if with_det else out

    def _person_dicts(self, o, rows):
        """Per-person dicts (model.py:329-347); v2d / rotmat are computed but not exposed, as in the reference.  One unbind per key
        instead of rows x 10 indexing calls (3 ms -> 1 ms of host time at 256 persons)."""
        keys = self.PERSON_KEYS
        return [dict(zip(keys, vals)) for vals in zip(*(o[n][:rows].unbind(0) for n in keys))]

    def _detect_and_heads(self, P, ws, K, k, thr, cap, with_dicts, stream):
        """Ordered compaction of the detections into `cap` person rows + HPH + SMPL-X layer for those rows (ws["counts"] is in flight)."""
        L = _lib.lib()
        dev, B, G = K.device, K.shape[0], P["G"]
        gstart_t, chunks_t, info, ngc, ncc, det, base = self._tables(B, cap, dev, with_det=True)
        o = self._alloc_outputs(P, cap, dev)
        scores_det = o["scores"].zero_()
        # The dicts are VIEWS of the output buffers: they are made here, BEFORE the heads are enqueued -- the host is far ahead of the
        # GPU at this point (the backbone has just been enqueued and runs for >100 ms; its launches are what the later ones queue
        # behind), so this millisecond of host work is free, whereas after the last launch the GPU has only the ~1 ms tail left
        # (measured: profiles/r04_session_c_inference_host_cfg5_cfg2.txt).  Dicts of padding rows are dropped unread.
        persons = self._person_dicts(o, cap) if with_dicts else None
        _lib.check(L.mhmr_person_groups(ws["counts"].data_ptr(), None, 0, B, cap, base.data_ptr(), gstart_t.data_ptr(), ngc,
                                        chunks_t.data_ptr(), ncc, info.data_ptr(), stream), "mhmr_person_groups")
        _lib.check(L.mhmr_detect_write_cap(ws["scores"].data_ptr(), B, G, k, thr, base.data_ptr(), det[0].data_ptr(), det[1].data_ptr(),
                                           det[2].data_ptr(), scores_det.data_ptr(), cap, stream), "mhmr_detect_write_cap")
        self._heads(P, ws, K, det, cap, gstart_t, ngc, chunks_t, ncc, info, stream, o)
        return o, det, info, persons

    #: output tensors of the heads: name -> trailing shape (rows = persons)
    OUTPUT_SHAPES = (("offset", (2,)), ("loc", (2,)), ("rotmat", (53, 3, 3)), ("rotvec", (53, 3)), ("shape", None), ("expression", (10,)),
                     ("dist_postprocessed", (1,)), ("dist", (1,)), ("v3d", None), ("v2d", None), ("j3d", (127, 3)), ("j2d", (127, 2)),
                     ("transl", (3,)), ("scores", ()))

    def _alloc_outputs(self, P, Pn, dev, flat=None):
        """The output tensors of the heads for Pn person rows (views of them are what forward returns), carved out of ONE allocation
        (``o["_flat"]``, every tensor 256-byte aligned): one allocator call per forward instead of fourteen, and a consumer that has to
        keep the results past the next forward of a recorded graph copies one buffer (graphed.GraphedForward).  ``flat``: build the views
        on an existing buffer of the same layout (such a copy)."""
        nb, V = P["hph"]["nb"], P["lbs"]["V"]
        var = {"shape": (nb,), "v3d": (V, 3), "v2d": (V, 2)}
        off, lay = 0, []
        for name, shp in self.OUTPUT_SHAPES:
            shp = var[name] if shp is None else shp
            n = Pn * int(np.prod(shp, dtype=np.int64))
            lay.append((name, shp, off, n))
            off += -(-n // 64) * 64
        if flat is None:
            flat = torch.empty(off, dtype=torch.float32, device=dev)
        assert flat.numel() == off
        o = {name: flat[a:a + n].view(Pn, *shp) for name, shp, a, n in lay}
        o["transl_pelvis"] = o["j3d"][:, 0:1]             # [Pn, 1, 3] view of joint 0 (the reference: j3d[:, [0]])
        o["_flat"] = flat
        return o

    def _heads(self, P, ws, K, det, Pn, gstart_t, ngc, chunks_t, ncc, info, stream, o=None):
        """HPH (model.py:258-283, 287-298) + SMPL-X layer (model.py:319-321) for Pn person rows -> dict of batched tensors (written
        into ``o`` when the caller allocated them: _alloc_outputs)."""
        L = _lib.lib()
        dev, B, G, N, Cdim, Kc, dt = K.device, K.shape[0], P["G"], P["N"], P["C"], P["Kc"], P["dt_id"]
        h = P["hph"]
        f = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        wsp = dict(zc=f(Pn, Cdim), token=f(Pn, h["Ktok"]), x=f(Pn, h["dim"]), xn=f(Pn, h["dim"]),
                   t1=f(Pn, max(3 * h["inner"], h["mlp"], Cdim)), t2=torch.zeros(Pn, h["inner"], dtype=torch.float32, device=dev),
                   dec=f(Pn, h["Ndec"]), det_row=torch.empty(Pn, dtype=torch.int32, device=dev))
        d = _lib.HphDesc()
        d.dtype, d.C, d.G, d.N, d.Kc = dt, Cdim, G, N, Kc
        d.dim, d.heads, d.mlp, d.depth, d.nb, d.Ktok, d.Ndec = h["dim"], h["heads"], h["mlp"], h["depth"], h["nb"], h["Ktok"], h["Ndec"]
        d.patch, d.nearness = PATCH, int(bool(self.nearness))
        d.fn = float(self.img_size / (2 * np.tan(np.radians(self.fovn) / 2)))
        for n in ("off1_w", "off1_b", "off2_w", "off2_b", "cq_x", "cq_y", "cv_x", "cv_y", "init_tail", "tok_w", "tok_b", "dec_w", "dec_b"):
            setattr(d, n, h[n])
        d.layers = C.cast(h["layers"], C.POINTER(_lib.HphLayer))
        for n, t in wsp.items():
            setattr(d, n, t.data_ptr())
        d.kv = ws["kv"].data_ptr()
        d.nvalid = info.data_ptr() if info is not None else None
        d.cam_dim = P["E"]
        if o is None:
            o = self._alloc_outputs(P, Pn, dev)
        offset, loc, rotmat, rotvec, shape, expression = o["offset"], o["loc"], o["rotmat"], o["rotvec"], o["shape"], o["expression"]
        dist_pp, dist = o["dist_postprocessed"], o["dist"]
        # ngc / Pn / ncc are upper bounds of the group count, the largest group and the work-item count (include/mhmr.h)
        _lib.check(L.mhmr_hph_forward(C.byref(d), ws["feat32"].data_ptr(), ws["zK"].data_ptr(), ws["ctx16"].data_ptr(), det[0].data_ptr(),
                                      det[1].data_ptr(), det[2].data_ptr(), Pn, gstart_t.data_ptr(), ngc, Pn,
                                      chunks_t.data_ptr(), ncc, K.data_ptr(), B, offset.data_ptr(), loc.data_ptr(),
                                      rotmat.data_ptr(), rotvec.data_ptr(), shape.data_ptr(), expression.data_ptr(), dist_pp.data_ptr(),
                                      dist.data_ptr(), stream), "mhmr_hph_forward")
        lb = P["lbs"]
        V = lb["V"]
        v3d, v2d, j3d, j2d, transl = o["v3d"], o["v2d"], o["j3d"], o["j2d"], o["transl"]
        ws_F, ws_A, ws_xf = f(roundup(Pn, 16), lb["Kb"]), f(roundup(Pn, 16), 768), f(Pn, 24)
        # pose kernel + vertex kernel; MHMR_LBS_FUSED=1: the one-launch form (pose role = leading workgroups of the vertex grid), built and
        # bit-identical but measured slower on this chip (csrc/lbs.hip) -- its flag workspace is the pack's: zeroed once, left zero by every call
        args = [C.byref(P["lbs_struct"]), rotvec.data_ptr(), shape.data_ptr(), expression.data_ptr(), loc.data_ptr(), dist.data_ptr(), K.data_ptr(),
                det[0].data_ptr(), Pn, ws_F.data_ptr(), ws_A.data_ptr(), ws_xf.data_ptr(), v3d.data_ptr(), v2d.data_ptr(), j3d.data_ptr(), j2d.data_ptr(),
                transl.data_ptr()]
        if os.environ.get("MHMR_LBS_FUSED") == "1":
            _lib.check(L.mhmr_lbs_forward_fused(*args, P["lbs_sync"].data_ptr(), stream), "mhmr_lbs_forward_fused")
        else:
            _lib.check(L.mhmr_lbs_forward(*args, stream), "mhmr_lbs_forward")
        return o
