"""Oracle implementation of the model's loss-backend interface (mono.model.hotpath), so the
same model code can run end to end on the CPU for parity tests and for the benchmark's
cpu_baseline leg.  Test infrastructure -- see oracle/__init__.py."""
import torch

from . import geometry, photometric, smooth


class _Ctx:
    pass


class OracleLossBackend:
    name = "oracle"

    def begin_step(self, opt, target, sources, K, inv_K):
        c = _Ctx()
        c.opt, c.target, c.sources, c.K, c.inv_K = opt, target, list(sources), K, inv_K
        return c

    def photometric(self, ctx, disp, Ts, noise, keep_warped=False, P=None):
        # P (the product's pre-multiplied K @ T) is ignored: the oracle follows the reference and forms it from Ts
        opt = ctx.opt
        draws = None
        if noise is not None:
            draws = [noise[i].unsqueeze(1) for i in range(noise.shape[0])]
        loss, idx, warped = photometric.photometric_scale_loss(
            ctx.target, ctx.sources, disp, ctx.K, ctx.inv_K, Ts, draws, opt.min_depth, opt.max_depth,
            automask=bool(opt.automask), n_scales=len(opt.scales))
        return loss, idx, (warped if keep_warped else None)

    def smooth(self, ctx, disp, weight, normalize):
        d = smooth.mean_normalize(disp) if normalize else disp
        return weight * smooth.smooth_loss(d, ctx.target)
