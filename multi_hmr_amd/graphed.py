"""The inference forward of a ``Model`` at a FIXED batch size, recorded once into a hipGraph and replayed.

Why: at batch 1 -- the case the reference's README publishes its runtimes for (README.md:87-91, timed at demo.py:333-338) -- the forward
is 150 (ViT-S) to 380 (ViT-L) kernel launches of a few microseconds each, and what the host does per launch (argument marshalling through
ctypes, ``torch.empty`` of the per-call buffers, the launch call itself) is of the same order as the kernels.  The eager forward
(``Model.forward``) already has no host synchronisation before its last launch (fixed person capacity, model.py ``_detect_and_heads``), which
is exactly what makes it recordable: the graph holds backbone -> camera embedding -> detection scores -> NMS counts -> ordered compaction into
``capacity`` person rows -> HPH -> SMPL-X layer; the ONE read-back (the person count) comes after the replay.

    gf = GraphedForward(model, batch=1, det_thresh=0.3, nms_kernel_size=3)
    humans = gf(x, K)            # same list of per-person dicts as model(x, K=K, det_thresh=0.3, nms_kernel_size=3)

* The kernels, their arguments and their order are the eager path's own (the graph is recorded by running ``Model``'s pieces under stream
  capture), so the outputs are the eager path's outputs at the same capacity, bit for bit (tests/test_gpu_graph.py).
* A batch that detects MORE persons than ``capacity`` is re-run through the eager path at its exact size (``overflows`` counts them) --
  the HIP path either way; nothing here computes on the CPU.
* The graph reads ``gf.x`` / ``gf.K`` and writes its own output buffers: ``__call__`` copies the caller's tensors in (device to device) and
  returns views of ONE copy of its output block unless ``copy=False`` (views of the block itself, valid until the next call).  A producer that writes the preprocessed image
  straight into ``gf.x`` (``preprocess.py``) saves the copy: ``gf.replay()``.
* One instance = one (model, device, batch, threshold, NMS window, capacity); the model's weights are read through the packed copies the
  model holds, so ``load_state_dict`` / a device move after recording needs a new instance (checked: the pack's identity).
"""
from __future__ import annotations

import torch

from . import _lib


class GraphedForward:
    def __init__(self, model, batch: int = 1, det_thresh=0.3, nms_kernel_size: int = 3, capacity: int = 32, device=None, warmup: int = 2):
        if not torch.cuda.is_available():
            raise _lib.MhmrError("GraphedForward records a hipGraph: it needs the MI355X; there is no CPU fallback")
        self.model = model
        self.batch = int(batch)
        self.det_thresh = float(det_thresh[0] if isinstance(det_thresh, list) else det_thresh)
        self.nms_kernel_size = int(nms_kernel_size)
        self.capacity = max(16, -(-int(capacity) // 16) * 16)            # whole 16-person groups of the SMPL-X layer
        self.overflows = 0
        dev = torch.device(device) if device is not None else next(model.parameters()).device
        if dev.type != "cuda":
            raise _lib.MhmrError("GraphedForward needs the model on the GPU (model.cuda())")
        S = model.img_size
        self.x = torch.zeros(self.batch, 3, S, S, dtype=torch.float32, device=dev)
        self.K = torch.tensor([[float(S), 0.0, S / 2.0], [0.0, float(S), S / 2.0], [0.0, 0.0, 1.0]], device=dev).repeat(self.batch, 1, 1).contiguous()
        with model._lock, torch.no_grad(), torch.autocast("cuda", enabled=False):
            # eager warm-up: packs the weights, allocates the workspace, sets the kernels' per-device attributes -- none of which may
            # happen under stream capture
            # (two calls at least: the first call of a model sizes the person capacity from the count and, on an empty image, never reaches the heads)
            # (the warm-up runs on a blank image: the person capacity it leaves for this batch size would be the minimum -- an eager
            # forward of a real batch right behind it would overflow and run its heads twice; round-5 advisor finding.  Put back what was there.)
            cap_before = model._person_cap.get(self.batch)
            for _ in range(max(int(warmup), 2)):
                model(self.x, K=self.K, det_thresh=self.det_thresh, nms_kernel_size=self.nms_kernel_size)
            if cap_before is None:
                model._person_cap.pop(self.batch, None)
            else:
                model._person_cap[self.batch] = cap_before
            P, ws, _ = model._prepare(self.x)
            self._pack, self._ws = P, ws           # the graph holds raw pointers into both: keep them alive past a repack()
            torch.cuda.synchronize(dev)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                stream = torch.cuda.current_stream(dev).cuda_stream
                model._front(P, ws, self.x, self.K, stream)
                model._count(ws, self.batch, P["G"], self.nms_kernel_size, self.det_thresh, stream)
                self.out, self.det, self.info, _ = model._detect_and_heads(P, ws, self.K, self.nms_kernel_size, self.det_thresh, self.capacity,
                                                                           False, stream)
            torch.cuda.synchronize(dev)

    def replay(self) -> int:
        """Replay on the current stream with whatever ``self.x`` / ``self.K`` hold; returns the person count (the one host sync)."""
        if self.model._packed is not self._pack:
            raise _lib.MhmrError("the model was re-packed (load_state_dict / device move) after this graph was recorded: make a new GraphedForward")
        with self.model._lock:
            self.graph.replay()
            return int(self.info[3].item())

    @torch.no_grad()
    def __call__(self, x, K, copy: bool = True, return_batched: bool = False):
        """``Model.forward(x, K=K, is_training=False)`` for x [batch, 3, S, S]: list of per-person dicts (``return_batched``: (dict of
        [P, ...] tensors, image id [P]) as ``Model.forward(return_batched=True)``)."""
        if tuple(x.shape) != tuple(self.x.shape) or tuple(K.shape) != tuple(self.K.shape):
            raise ValueError(f"this graph was recorded for x {tuple(self.x.shape)} and K {tuple(self.K.shape)}")
        keys = self.model.PERSON_KEYS
        with self.model._lock:
            if x.data_ptr() != self.x.data_ptr():
                self.x.copy_(x, non_blocking=True)
            if K.data_ptr() != self.K.data_ptr():
                self.K.copy_(K, non_blocking=True)
            Pn = self.replay()
            if Pn > self.capacity:          # more persons than rows were recorded for: the eager path at the exact size
                self.overflows += 1
                return self.model(x, K=K, det_thresh=self.det_thresh, nms_kernel_size=self.nms_kernel_size, return_batched=return_batched)
            if Pn == 0:
                return ({}, torch.zeros(0, dtype=torch.int32, device=self.x.device)) if return_batched else []
            # copy: ONE device copy of the output block (all capacity rows), the person rows are views of that copy
            src = self.model._alloc_outputs(self._pack, self.capacity, self.x.device, flat=self.out["_flat"].clone()) if copy else self.out
            o = {n: src[n][:Pn] for n in keys}
            if return_batched:
                ids = self.det[0][:Pn]
                return o, (ids.clone() if copy else ids)
        return self.model._person_dicts(o, Pn)


def graphed(model, batch: int = 1, det_thresh=0.3, nms_kernel_size: int = 3, capacity: int = 32) -> GraphedForward:
    """The model's cached GraphedForward for these settings (recorded on first use, re-recorded after a re-pack)."""
    thr = float(det_thresh[0] if isinstance(det_thresh, list) else det_thresh)
    key = (int(batch), thr, int(nms_kernel_size), int(capacity))
    cache = model.__dict__.setdefault("_graphs", {})
    gf = cache.get(key)
    if gf is None or gf._pack is not model._packed:
        gf = cache[key] = GraphedForward(model, batch, thr, nms_kernel_size, capacity)
    return gf
