"""Routing of the loss hot path to the hand-written HIP kernels.

``HipLossBackend`` is the product implementation: every call lands in libtripled_hip.so through
tripled_amd.ops and raises if the library is missing or the tensors are not on a HIP device --
there is no CPU fallback.  The backend is a small duck-typed interface so that the tests and
the benchmark's cpu_baseline leg can substitute the CPU oracle (oracle/backend.py) to run the
same model code on the host.

Per training step:
  ctx = backend.begin_step(opt, target, sources, K, inv_K)      # identity term + image pyramid, once
  loss, min_index, warped = backend.photometric(ctx, disp_s, Ts, noise, keep_warped)
  loss = backend.smooth(ctx, disp_s, scale, weight, normalize)
"""
import torch


class StepContext:
    __slots__ = ("opt", "target", "sources", "K", "inv_K", "idloss", "pyramid", "P_cache", "frames")


class HipLossBackend:
    name = "hip"

    def __init__(self):
        import tripled_amd  # noqa: F401  (the alias loader at the repo root)
        from tripled_amd import ops, native
        native.load()          # fail at construction, not in the middle of a step
        self.ops = ops

    def begin_step(self, opt, target, sources, K, inv_K):
        if not target.is_cuda:
            raise RuntimeError("HipLossBackend needs inputs on a HIP device (got %s); the loss hot path "
                               "has no CPU implementation in the product package" % target.device)
        ctx = StepContext()
        ctx.opt, ctx.target, ctx.sources, ctx.K, ctx.inv_K = opt, target, list(sources), K, inv_K
        # RGBX pixels for the per-scale kernels, once per step: a by-product of the identity-term kernel when it runs
        ctx.frames = self.ops.pack_frames(target, ctx.sources, pack=not opt.automask)
        ctx.idloss = self.ops.photo_identity(ctx.frames) if opt.automask else None
        ctx.pyramid = {}
        ctx.P_cache = None
        return ctx

    def image_at(self, ctx, h, w):
        key = (h, w)
        if key not in ctx.pyramid:
            ctx.pyramid[key] = self.ops.area_downsample(ctx.target, h, w)
        return ctx.pyramid[key]

    def photometric(self, ctx, disp, Ts, noise, keep_warped=False, P=None):
        """``P`` [n_src,B,3,4] = (K @ T)[:, :3, :] when the caller already holds it (ops.pose_transforms emits it
        with the transforms); otherwise it is formed from ``Ts`` once per step, not once per scale."""
        opt = ctx.opt
        if P is None:
            key = tuple(id(T) for T in Ts)
            if ctx.P_cache is None or ctx.P_cache[0] != key:
                ctx.P_cache = (key, torch.stack([torch.matmul(ctx.K, T)[:, :3, :] for T in Ts], 0))
            P = ctx.P_cache[1]
        loss, argmin, warped = self.ops.photometric_scale_loss(
            disp, P, ctx.frames, None, ctx.inv_K, ctx.idloss, noise,
            opt.min_depth, opt.max_depth, len(opt.scales), keep_warped)
        return loss, argmin, (list(warped.unbind(0)) if keep_warped else None)

    def smooth(self, ctx, disp, weight, normalize):
        img = self.image_at(ctx, disp.shape[2], disp.shape[3])
        return self.ops.smooth_loss(disp, img, normalize, weight)


_default = None


def default_backend():
    global _default
    if _default is None:
        _default = HipLossBackend()
    return _default
