"""Drop-in for the Anny-variant Human Perception Head ``multi_hmr_anny.hph.HPH`` (reference multi_hmr_anny/hph.py:142-151):
the same pre-norm (self-attention, cross-attention, feed-forward) stack as the Multi-HMR HPH with other constants
(dim 512, 16 heads x 32, mlp 2048, depth 8, context_dim = dim, no mask multiplies) -- SURVEY.md section 8 row a10.
Same constructor, same ``state_dict`` keys, same ``forward(x, context, mask)``; compute = ``mhmr_xattn_layers_forward``."""
from __future__ import annotations

import ctypes as C

import torch
from torch import nn

from . import _lib, packing
from .model import _ca, _ff, _Holder, _PreNorm, _sa
from .packing import roundup


class _Stack(_Holder):
    def __init__(self, dim, depth, heads, dim_head, mlp_dim):
        super().__init__()
        inner = heads * dim_head
        self.layers = nn.ModuleList([nn.ModuleList([_PreNorm(dim, _sa(dim, inner)), _PreNorm(dim, _ca(dim, dim, inner)),
                                                    _PreNorm(dim, _ff(dim, mlp_dim))]) for _ in range(depth)])


class HPH(nn.Module):
    def __init__(self, dim, depth, heads, dim_head, mlp_dim, dropout=0.0, precision="f16"):
        super().__init__()
        if dim_head != 32:
            raise NotImplementedError("the attention kernels are built for dim_head = 32 (both released HPH variants)")
        if dropout:
            raise NotImplementedError("inference path: dropout must be 0")
        self.dim, self.depth, self.heads, self.mlp_dim, self.precision = dim, depth, heads, mlp_dim, precision
        self.transformer = _Stack(dim, depth, heads, dim_head, mlp_dim)
        self._packed = None
        for p in self.parameters():
            p.requires_grad_(False)

    def load_state_dict(self, *a, **k):
        self._packed = None
        return super().load_state_dict(*a, **k)

    def _apply(self, fn, *a, **k):
        self._packed = None
        return super()._apply(fn, *a, **k)

    def _pack(self, device):
        dt_id, tdt = packing.OP_DTYPES[self.precision]
        f32 = lambda t: t.detach().to(device=device, dtype=torch.float32).contiguous()
        keep, layers = [], (_lib.HphLayer * self.depth)()

        def k(t):
            keep.append(t)
            return t.data_ptr()
        Kc = roundup(self.dim, 64)
        for l, (sa, ca, ff) in enumerate(self.transformer.layers):
            y = layers[l]
            y.ln_sa_w, y.ln_sa_b, y.to_qkv = k(f32(sa.norm.weight)), k(f32(sa.norm.bias)), k(f32(sa.fn.to_qkv.weight))
            y.sa_out_w, y.sa_out_b = k(f32(sa.fn.to_out[0].weight)), k(f32(sa.fn.to_out[0].bias))
            y.ln_ca_w, y.ln_ca_b = k(f32(ca.norm.weight)), k(f32(ca.norm.bias))
            kvw = torch.zeros(64 * self.heads, Kc, device=device)
            kvw[:, : self.dim] = f32(ca.fn.to_kv.weight)
            y.to_kv16, y.to_q = k(kvw.to(tdt).contiguous()), k(f32(ca.fn.to_q.weight))
            y.ca_out_w, y.ca_out_b = k(f32(ca.fn.to_out[0].weight)), k(f32(ca.fn.to_out[0].bias))
            y.ln_ff_w, y.ln_ff_b = k(f32(ff.norm.weight)), k(f32(ff.norm.bias))
            y.ff1_w, y.ff1_b = k(f32(ff.fn.net[0].weight)), k(f32(ff.fn.net[0].bias))
            y.ff2_w, y.ff2_b = k(f32(ff.fn.net[3].weight)), k(f32(ff.fn.net[3].bias))
        self._packed = dict(device=device, dt_id=dt_id, tdt=tdt, layers=layers, keep=keep, Kc=Kc)
        return self._packed

    @torch.no_grad()
    def forward(self, x, context, mask=None):
        """x [B', nmax, dim] padded queries, context [B', N, dim], mask [B', nmax] (1 = real query) -> [B', nmax, dim].
        Real rows equal the reference's; padded rows (garbage in the reference, dropped by its caller at
        multi_hmr_anny/multi_hmr.py:141) are returned as zeros."""
        if not x.is_cuda:
            raise _lib.MhmrError("multi_hmr_amd.anny_hph.HPH runs only on an MI355X (HIP) tensor; there is no CPU fallback")
        with torch.autocast("cuda", enabled=False):
            P_ = self._packed if self._packed is not None and self._packed["device"] == x.device else self._pack(x.device)
            L = _lib.lib()
            dev, (Bp, nmax, dim), N = x.device, x.shape, context.shape[1]
            assert dim == self.dim and context.shape == (Bp, N, dim)
            if mask is None:
                mask = torch.ones(Bp, nmax, device=dev)
            counts = [int(c) for c in mask.sum(1).round().long().tolist()]
            keep_rows = mask.reshape(-1) > 0.5
            xr = x.reshape(Bp * nmax, dim)[keep_rows].float().contiguous()              # ragged [P, dim]
            Pn = xr.shape[0]
            out = torch.zeros(Bp, nmax, dim, device=dev)
            if Pn == 0:
                return out
            gstart, chunks, start = [0], [], 0
            for b, c in enumerate(counts):
                if c == 0:
                    continue
                for q0 in range(0, c, 8):
                    chunks += [b, start + q0, min(8, c - q0)]
                start += c
                gstart.append(start)
            meta = torch.tensor(gstart + chunks, dtype=torch.int32).to(dev)
            Kc, inner = P_["Kc"], 32 * self.heads
            Mctx = roundup(Bp * N, 128)
            ctx16 = torch.zeros(Mctx, Kc, dtype=P_["tdt"], device=dev)
            ctx16[: Bp * N, :dim] = context.reshape(Bp * N, dim).float().to(P_["tdt"])
            f = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
            xn, t1, t2, kv = f(Pn, dim), f(Pn, max(3 * inner, self.mlp_dim)), f(Pn, inner), f(Mctx, 2 * inner)
            _lib.check(L.mhmr_xattn_layers_forward(C.cast(P_["layers"], C.POINTER(_lib.HphLayer)), self.depth, dim, self.heads, self.mlp_dim,
                                                   Kc, N, Bp, P_["dt_id"], xr.data_ptr(), xn.data_ptr(), t1.data_ptr(), t2.data_ptr(),
                                                   kv.data_ptr(), ctx16.data_ptr(), meta[: len(gstart)].data_ptr(), len(gstart) - 1,
                                                   max(counts), meta[len(gstart):].data_ptr(), len(chunks) // 3, Pn,
                                                   torch.cuda.current_stream(dev).cuda_stream), "mhmr_xattn_layers_forward")
            out.reshape(Bp * nmax, dim)[keep_rows] = xr
            return out
