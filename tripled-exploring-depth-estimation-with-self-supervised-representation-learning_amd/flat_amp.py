"""Flat mixed-precision parameter store for the data-parallel training step.

Under plain autocast every convolution weight is cast fp32 -> bf16 once per forward and every weight
gradient bf16 -> fp32 once per backward: ~650 tiny kernels per step for the tripleD model, plus ~320
per-parameter optimiser/clip launches.  Here all of that is a handful of large launches over
contiguous HBM buffers (sized for a 288 GB device, where duplicating the weights costs nothing):

  flat_w   fp32 master copy of EVERY parameter (module fp32 parameters are views into it)
  flat_lp  bf16 working copy of the convolution weights/biases (module parameters are views into it);
           refreshed from flat_w by ONE cast kernel after each optimiser step
  flat_g   fp32 gradients of every parameter, filled by ONE multi-tensor copy after backward;
           it is what the RCCL all-reduce runs on (few, large buckets) and what clip + Adam consume

Numerically this is the autocast path: convolutions see bf16(weight) and produce bf16 weight gradients
in both cases; the master update is the same fp32 Adam.

``lowp=False`` keeps every parameter fp32 (autocast casts per use, as without this class) and only provides
the flat fp32 gradient/parameter buffers: gradients are gathered by batched concatenations after backward,
all-reduced in a few large buckets, clipped and stepped by one single-tensor Adam.  That form costs the same as
the plain step on one GPU and is what ``bench.py --gpus N`` (N > 1) uses between its two HIP graphs.
"""
import collections
import contextlib
import os

import torch
import torch.distributed as dist
import torch.nn as nn

ALIGN = 8      # elements: 16 B in bf16, 32 B in fp32
# what a checkpoint's param_group carries across: the optimiser's hyper-parameters.  The execution flags (capturable,
# fused, foreach, differentiable) belong to the process that runs the optimiser, not to the file.
HYPER_KEYS = ("lr", "betas", "eps", "weight_decay", "amsgrad", "maximize", "initial_lr")


class FlatMixedPrecision:
    def __init__(self, model, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, max_norm=None,
                 lowp_dtype=torch.bfloat16, process_group=None, bucket_bytes=64 << 20, lowp=None):
        params = [p for p in model.parameters() if p.requires_grad]
        dev = params[0].device
        lowp_ids = set()
        for m in model.modules():
            if isinstance(m, nn.Conv2d) and (lowp is None or lowp):
                for p in m.parameters(recurse=False):
                    if p.requires_grad:
                        lowp_ids.add(id(p))
        self.lowp = [p for p in params if id(p) in lowp_ids]
        self.full = [p for p in params if id(p) not in lowp_ids]
        self.params = self.lowp + self.full
        # every parameter starts on an 8-element boundary: the convolution kernels use 16-byte vector loads
        # on their (bf16) weight pointer
        offsets, off, n_lp = [], 0, 0
        for p in self.params:
            offsets.append(off)
            off += (p.numel() + ALIGN - 1) // ALIGN * ALIGN
            if self.lowp and p is self.lowp[-1]:
                n_lp = off
        n_all = off
        self.flat_w = torch.zeros(n_all, device=dev, dtype=torch.float32)
        self.flat_g = torch.zeros(n_all, device=dev, dtype=torch.float32)
        self.flat_lp = torch.zeros(n_lp, device=dev, dtype=lowp_dtype)
        self.flat_glp = torch.zeros(n_lp, device=dev, dtype=lowp_dtype)
        self._pads = {}
        self.n_lp = n_lp
        self.offsets = offsets
        self._offset_of = {id(p): off for p, off in zip(self.params, offsets)}

        def view(buf, p, off):
            return torch.as_strided(buf, p.size(), p.stride(), off)    # keeps channels_last strides

        with torch.no_grad():
            for p, off in zip(self.params, offsets):
                w_view = view(self.flat_w, p, off)
                w_view.copy_(p)
                if id(p) in lowp_ids:
                    p.data = view(self.flat_lp, p, off)    # the module now computes with the bf16 working copy
                else:
                    p.data = w_view                        # fp32 parameters (BatchNorm, Linear) live in the master buffer
            self.flat_lp.copy_(self.flat_w[:n_lp])
        self.master = nn.Parameter(self.flat_w, requires_grad=True)
        self.master.grad = self.flat_g
        on_gpu = dev.type == "cuda"
        self.optimizer = torch.optim.Adam([self.master], lr=lr, betas=betas, eps=eps, weight_decay=weight_decay,
                                          capturable=on_gpu, fused=on_gpu)
        self.max_norm = max_norm
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        self.use_avg = dist.is_available() and dist.is_initialized() and dist.get_backend(process_group) == "nccl"
        per = max(1, bucket_bytes // 4)
        self.buckets = [(s, min(s + per, n_all)) for s in range(0, n_all, per)]

    # ------------------------------------------------------------------ checkpoint format
    # A checkpoint written with the flat store is indistinguishable from one written without it (mmcv 0.4.4 layout,
    # reference: mono/apis/trainer.py:163-189 through mmcv's CheckpointHook): 'state_dict' holds the fp32 MASTER value of
    # every parameter under the module's own key, 'optimizer' the per-parameter Adam state in model.parameters() order.

    def _master_view(self, p):
        return torch.as_strided(self.flat_w, p.size(), p.stride(), self._offset_of[id(p)])

    def module_state_dict(self, model):
        """model.state_dict() with the fp32 master weights in place of the bf16 working copies."""
        out = collections.OrderedDict()
        params = dict(model.named_parameters())
        for key, val in model.state_dict().items():
            p = params.get(key)
            out[key] = self._master_view(p).detach().clone() if p is not None and id(p) in self._offset_of else val
        return out

    def load_module_state_dict(self, model, state_dict, strict=False):
        """Load fp32 weights into the master buffer (and refresh the bf16 working copy); buffers go through the module."""
        params = dict(model.named_parameters())
        rest, missing, mismatched = {}, [], []
        with torch.no_grad():
            for key, val in state_dict.items():
                p = params.get(key)
                if p is not None and id(p) in self._offset_of:
                    if tuple(val.shape) != tuple(p.shape):      # as nn.Module.load_state_dict: report, never broadcast
                        mismatched.append("%s: checkpoint %s vs model %s" % (key, tuple(val.shape), tuple(p.shape)))
                        continue
                    self._master_view(p).copy_(val.to(self.flat_w.device, torch.float32))
                else:
                    rest[key] = val
            if self.n_lp:
                self.flat_lp.copy_(self.flat_w[:self.n_lp])
        for key, p in params.items():
            if id(p) in self._offset_of and key not in state_dict:
                missing.append(key)
        own = dict(model.named_buffers())
        own.update({k: p for k, p in params.items() if id(p) not in self._offset_of})
        unexpected = [k for k in rest if k not in own]
        with torch.no_grad():
            for key, val in rest.items():
                if key in own:
                    if tuple(val.shape) != tuple(own[key].shape):
                        mismatched.append("%s: checkpoint %s vs model %s" % (key, tuple(val.shape), tuple(own[key].shape)))
                        continue
                    own[key].copy_(val)
        missing += [k for k in own if k not in state_dict]
        if mismatched:
            raise RuntimeError("size mismatch for " + "; ".join(mismatched))
        if strict and (missing or unexpected):
            raise RuntimeError("missing keys %s, unexpected keys %s" % (missing, unexpected))
        return missing, unexpected

    def optimizer_state_dict(self, model):
        """torch.optim.Adam.state_dict() as a per-parameter optimiser over model.parameters() would write it."""
        flat_state = self.optimizer.state.get(self.master, {})
        group = self.optimizer.param_groups[0]
        order = [p for p in model.parameters() if p.requires_grad]
        state = {}
        if flat_state:
            for i, p in enumerate(order):
                off = self._offset_of[id(p)]
                view = lambda buf: torch.as_strided(buf, p.size(), p.stride(), off).detach().clone()   # noqa: E731
                state[i] = {"step": flat_state["step"].detach().clone(), "exp_avg": view(flat_state["exp_avg"]),
                            "exp_avg_sq": view(flat_state["exp_avg_sq"])}
        def plain(v):      # a device-side lr tensor goes to the file as the number the host wrote into it
            if not torch.is_tensor(v):
                return v
            host = getattr(v, "_host_value", None)
            return host if host is not None else float(v)
        pg = {k: plain(group[k]) for k in HYPER_KEYS if k in group}
        pg["params"] = list(range(len(order)))
        return {"state": state, "param_groups": [pg]}

    def load_optimizer_state_dict(self, model, osd):
        """Accepts the per-parameter format above (also what the reference's checkpoints hold)."""
        order = [p for p in model.parameters() if p.requires_grad]
        if len(osd["param_groups"]) != 1 or len(osd["param_groups"][0]["params"]) != len(order):
            raise ValueError("optimizer state does not match the model: %d parameters expected" % len(order))
        own = self.optimizer.param_groups[0]
        for k in HYPER_KEYS:
            if k in osd["param_groups"][0]:
                if torch.is_tensor(own.get(k)):      # a device-side lr (read by a captured step): keep the tensor
                    own[k].fill_(float(osd["param_groups"][0][k]))
                    own[k]._host_value = float(osd["param_groups"][0][k])
                else:
                    own[k] = osd["param_groups"][0][k]
        if not osd["state"]:
            return
        dev = self.flat_w.device
        st = self.optimizer.state[self.master]
        if not st:
            # torch keeps `step` on the device whenever the optimiser is fused OR capturable
            on_dev = bool(own.get("capturable") or own.get("fused"))
            st["step"] = torch.zeros((), dtype=torch.float32, device=dev if on_dev else "cpu")
            st["exp_avg"] = torch.zeros_like(self.flat_w)
            st["exp_avg_sq"] = torch.zeros_like(self.flat_w)
        with torch.no_grad():
            for i, p in enumerate(order):
                ps = osd["state"].get(i)
                if ps is None:
                    continue
                off = self._offset_of[id(p)]
                torch.as_strided(st["exp_avg"], p.size(), p.stride(), off).copy_(ps["exp_avg"].to(dev))
                torch.as_strided(st["exp_avg_sq"], p.size(), p.stride(), off).copy_(ps["exp_avg_sq"].to(dev))
                st["step"].copy_(torch.as_tensor(ps["step"], dtype=torch.float32))

    @contextlib.contextmanager
    def full_precision(self):
        """Inside: the module computes with the fp32 master weights (validation runs in the reference's precision,
        without autocast); outside: back on the bf16 working copy."""
        saved = [(p, p.data) for p in self.lowp]
        try:
            for p in self.lowp:
                p.data = self._master_view(p)
            yield
        finally:
            for p, d in saved:
                p.data = d

    def zero_grad(self):
        """Gradients are produced fresh by autograd each step (set_to_none) and gathered by collect()."""
        for p in self.params:
            p.grad = None

    def _flat_sources(self, params, offsets, end, dtype):
        """Per-parameter gradients as 1-D memory-order views, interleaved with the zero pads of the flat layout."""
        out = []
        for i, (p, off) in enumerate(zip(params, offsets)):
            n = p.numel()
            g = p.grad
            if g is None:
                g = self._zeros(n, dtype)                      # unused parameter: its slot stays zero
            elif g.stride() != p.stride() or g.dtype != dtype:
                g = self._relayout(g, p, dtype)                # rare: autograd normally follows the layout contract
            out.append(torch.as_strided(g, (n,), (1,), g.storage_offset()))
            nxt = offsets[i + 1] if i + 1 < len(offsets) else end
            if nxt - off - n:
                out.append(self._zeros(nxt - off - n, dtype))
        return out

    def _zeros(self, n, dtype):
        key = (n, dtype)
        if key not in self._pads:
            self._pads[key] = torch.zeros(n, device=self.flat_g.device, dtype=dtype)
        return self._pads[key]

    @staticmethod
    def _relayout(g, p, dtype):
        out = torch.empty_strided(p.size(), p.stride(), device=g.device, dtype=dtype)
        out.copy_(g)
        return out

    def _gather_native(self, params, offsets, dtype):
        """One hand-written gather (td_gather_flat) for the gradients of ``params``: pointers in the kernel arguments, the
        bf16 -> fp32 conversion on the way, pads untouched (they are zero from construction)."""
        import ctypes
        from . import native
        n = len(params)
        srcs = (ctypes.c_void_p * n)()
        dst = (ctypes.c_longlong * n)(*offsets)
        num = (ctypes.c_longlong * n)(*[p.numel() for p in params])
        keep = []
        for i, p in enumerate(params):
            g = p.grad
            if g is None:
                srcs[i] = None
                continue
            if g.stride() != p.stride() or g.dtype != dtype:
                g = self._relayout(g, p, dtype)
            keep.append(g)
            srcs[i] = g.data_ptr()
        native.check(native.load().td_gather_flat(srcs, dst, num, n, native.DTYPE_CODES[dtype], native.ptr(self.flat_g),
                                                  native.stream()), "td_gather_flat")

    def collect(self):
        """Per-parameter gradients (fresh autograd allocations, bf16 for convolutions / fp32 otherwise) -> the
        flat fp32 gradient buffer.  HIP device: two hand-written gathers (bf16 group with the conversion fused, fp32 group);
        elsewhere batched concatenations (128 tensors per launch) plus ONE bf16->fp32 cast."""
        k = len(self.lowp)
        if self.flat_g.is_cuda and not os.environ.get("TD_NO_NATIVE_GATHER") and (not k or self.flat_lp.dtype == torch.bfloat16):
            if k:
                self._gather_native(self.lowp, self.offsets[:k], self.flat_lp.dtype)
            if self.full:
                self._gather_native(self.full, self.offsets[k:], torch.float32)
            return
        if k:
            torch.cat(self._flat_sources(self.lowp, self.offsets[:k], self.n_lp, self.flat_lp.dtype), out=self.flat_glp)
            self.flat_g[:self.n_lp].copy_(self.flat_glp)
        if self.full:
            torch.cat(self._flat_sources(self.full, self.offsets[k:], self.flat_g.numel(), torch.float32),
                      out=self.flat_g[self.n_lp:])

    def allreduce(self, force=False):
        """Average flat_g over the ranks: a few large RCCL all-reduces, issued back to back.  A one-rank job skips
        the collectives unless ``force`` (the single-GPU RCCL test of this path)."""
        if self.world == 1 and not force:
            return
        works = []
        for s, e in self.buckets:
            chunk = self.flat_g[s:e]
            op = dist.ReduceOp.AVG if self.use_avg else dist.ReduceOp.SUM
            works.append((dist.all_reduce(chunk, op=op, group=self.group, async_op=True), chunk))
        for w, chunk in works:
            w.wait()
            if not self.use_avg:
                chunk.div_(self.world)

    def _fused_step_possible(self):
        g = self.optimizer.param_groups[0]
        return (self.flat_w.is_cuda and not os.environ.get("TD_NO_FUSED_ADAM") and g.get("weight_decay", 0) == 0
                and not g.get("amsgrad", False) and not g.get("maximize", False) and self.flat_w.numel() % 4 == 0 and self.n_lp % 4 == 0
                and (self.n_lp == 0 or self.flat_lp.dtype == torch.bfloat16))

    def _adam_state(self):
        """torch.optim.Adam's own state entry of the master parameter (created like torch's lazy init, so that checkpoints and a
        later plain optimizer.step() see what they expect)."""
        st = self.optimizer.state[self.master]
        if not st:
            st["step"] = torch.zeros((), dtype=torch.float32, device=self.flat_w.device)
            st["exp_avg"] = torch.zeros_like(self.flat_w)
            st["exp_avg_sq"] = torch.zeros_like(self.flat_w)
        return st

    def step(self):
        """clip_grad_norm_(max_norm, 2) + Adam on the flat buffers, then refresh the bf16 working copy.  On a HIP device the
        gradient scale, the update and the cast are ONE pass (td_adam_flat, csrc/td_optim.hip: 30 B per parameter instead of
        the ~50 B of ATen's scale / multi-tensor Adam / cast passes); elsewhere the same steps through torch."""
        total = None
        if self.max_norm is not None:
            total = torch.linalg.vector_norm(self.flat_g)
        if self._fused_step_possible():
            from . import native
            lib = native.load()
            st, grp = self._adam_state(), self.optimizer.param_groups[0]
            st["step"] += 1
            lr = grp["lr"]
            native.check(lib.td_adam_flat(native.ptr(self.flat_w), native.ptr(self.flat_g), native.ptr(st["exp_avg"]),
                                          native.ptr(st["exp_avg_sq"]), native.ptr(self.flat_lp) if self.n_lp else None,
                                          self.flat_w.numel(), self.n_lp, native.ptr(st["step"]),
                                          native.ptr(lr) if torch.is_tensor(lr) else None, 0.0 if torch.is_tensor(lr) else float(lr),
                                          float(grp["betas"][0]), float(grp["betas"][1]), float(grp["eps"]),
                                          native.ptr(total) if total is not None else None,
                                          float(self.max_norm) if self.max_norm is not None else 0.0, native.stream()), "td_adam_flat")
            return total
        if total is not None:
            self.flat_g.mul_(torch.clamp(self.max_norm / (total + 1e-6), max=1.0))
        self.optimizer.step()
        if self.n_lp:
            self.flat_lp.copy_(self.flat_w[:self.n_lp])
        return total
