"""torch.autograd.Functions over the C ABI of libtripled_hip.so.

These are the fused replacements for the reference's unfused loss ops; each docstring cites
the reference lines it stands in for.  All launches go to torch's current HIP stream and do
not synchronise, so a training step that uses them can be captured in a HIP graph.
"""
import os

import torch

from . import native


def _f32c(t):
    if t.dtype != torch.float32:
        t = t.float()
    return t if t.is_contiguous() else t.contiguous()


class PackedFrames:
    """The target and source frames of a step as RGBX pixels ([B,H,W,4] fp32, x = 0), the format the per-scale photometric
    forward reads (csrc/td_common.h "packed frames": one 16-byte load per pixel / bilinear tap instead of three dword loads from
    the NCHW planes), next to the NCHW originals the identity term and the backward read.  The RGBX copies are written by the
    identity-term kernel from the pixels it reads anyway (``photo_identity``), or by td_pack_rgbx when no identity term runs."""

    def __init__(self, tgt, srcs, pack=True):
        self.tgt_planar = _f32c(tgt.detach())
        self.srcs_planar = tuple(_f32c(s.detach()) for s in srcs)
        if self.tgt_planar.dim() != 4 or self.tgt_planar.shape[1] != 3:
            raise ValueError("colour frames are [B,3,H,W], got %s" % (tuple(self.tgt_planar.shape),))
        B, _, H, W = self.tgt_planar.shape
        self.shape = (B, H, W)
        dev = self.tgt_planar.device
        self.tgt = torch.empty(B, H, W, 4, device=dev, dtype=torch.float32)
        self.srcs = tuple(torch.empty(B, H, W, 4, device=dev, dtype=torch.float32) for _ in self.srcs_planar)
        self.packed = False
        if pack:
            self.pack()

    def pack(self):
        if not self.packed:
            lib = native.load()
            B, H, W = self.shape
            for planar, out in zip((self.tgt_planar,) + self.srcs_planar, (self.tgt,) + self.srcs):
                native.check(lib.td_pack_rgbx(native.ptr(planar), B, H, W, native.ptr(out), native.stream()), "td_pack_rgbx")
            self.packed = True
        return self


def pack_frames(tgt, srcs, pack=True):
    """``pack=False``: the RGBX copies are left to ``photo_identity`` (which writes them as a by-product)."""
    return PackedFrames(tgt, srcs, pack)


def _frames(tgt, srcs):
    return tgt.pack() if isinstance(tgt, PackedFrames) else PackedFrames(tgt, srcs)


def photo_identity(tgt, srcs=None):
    """Auto-mask identity term for every source frame, once per step.
    compute_reprojection_loss(inputs[("color", f, 0)], target) at
    mono/model/mono_fm_joint_inpaint/net.py:101-104 -> [B, n_src, H, W] (a view: the memory is [B,H,W,n_src], the layout
    td_photo_fwd reads).  ``tgt``: the [B,3,H,W] target with ``srcs`` the list of sources, or a PackedFrames -- whose RGBX copies
    this call fills if they are not packed yet."""
    lib = native.load()
    fr = tgt if isinstance(tgt, PackedFrames) else PackedFrames(tgt, srcs, pack=False)
    B, H, W = fr.shape
    out = torch.empty(B, H, W, len(fr.srcs), device=fr.tgt.device, dtype=torch.float32)
    emit = not fr.packed
    native.check(lib.td_photo_identity(native.ptr(fr.tgt_planar), native.ptr_array(fr.srcs_planar), len(fr.srcs), B, H, W,
                                       native.ptr(out), native.ptr(fr.tgt) if emit else None,
                                       native.ptr_array(fr.srcs) if emit else None, native.stream()), "td_photo_identity")
    fr.packed = True
    return out.permute(0, 3, 1, 2)


def area_downsample(img, h, w):
    """F.interpolate(img, (h, w), mode='area') for integer factors
    (mono/model/mono_fm_joint/net.py:283, :311).  No gradient (the image is an input)."""
    lib = native.load()
    img = _f32c(img.detach())
    B, C, H, W = img.shape
    if (H, W) == (h, w):
        return img
    out = torch.empty(B, C, h, w, device=img.device, dtype=torch.float32)
    native.check(lib.td_area_downsample(native.ptr(img), B, C, H, W, h, w, native.ptr(out), native.stream()),
                 "td_area_downsample")
    return out


class _PhotometricScaleLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, disp, P, tgt, srcs, tgt_planar, srcs_planar, invK, idloss, noise, min_depth, max_depth, n_scales, keep_warped):
        lib = native.load()
        disp = _f32c(disp)
        P = _f32c(P)
        B, H, W, _ = tgt.shape                     # RGBX frames [B,H,W,4]
        hs, ws = disp.shape[2], disp.shape[3]
        n_src = len(srcs)
        dev = tgt.device
        argmin = torch.empty(B, H, W, device=dev, dtype=torch.uint8)
        nblk = lib.td_photo_num_blocks(B, H, W)
        partial = torch.empty(nblk, device=dev, dtype=torch.float32)
        warped = torch.empty(n_src, B, 3, H, W, device=dev, dtype=torch.float32) if keep_warped else None
        loss = torch.empty(1, device=dev, dtype=torch.float32)
        need_grad = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        coef = torch.empty(B, 9, H, W, device=dev, dtype=torch.float32) if need_grad else None
        st = native.stream()
        native.check(lib.td_photo_fwd(native.ptr(tgt), native.ptr_array(srcs), n_src, native.ptr(disp),
                                      native.ptr(P), native.ptr(invK), native.ptr(idloss), native.ptr(noise),
                                      B, H, W, hs, ws, float(min_depth), float(max_depth),
                                      native.ptr(argmin), native.ptr(warped), None, native.ptr(partial),
                                      native.ptr(coef), st),
                     "td_photo_fwd")
        inv_count = 1.0 / (float(B) * H * W * n_scales)
        native.check(lib.td_sum_scaled(native.ptr(partial), nblk, inv_count, native.ptr(loss), st), "td_sum_scaled")
        ctx.save_for_backward(disp, P, tgt_planar, invK, argmin, coef if coef is not None else argmin, tgt, *srcs_planar, *srcs)
        ctx.meta = (min_depth, max_depth, inv_count, idloss is not None, n_src)
        ctx.mark_non_differentiable(argmin)
        if keep_warped:
            ctx.mark_non_differentiable(warped)
            return loss.reshape(()), argmin, warped
        return loss.reshape(()), argmin, torch.empty(0, device=dev)

    @staticmethod
    def backward(ctx, g_loss, _g_argmin, _g_warped):
        lib = native.load()
        disp, P, tgt, invK, argmin, coef, tgt_x, *both = ctx.saved_tensors
        min_depth, max_depth, inv_count, automask, n_src = ctx.meta
        srcs, srcs_x = both[:n_src], both[n_src:]      # NCHW frames, and their RGBX copies (read at the finest scale)
        B, _, H, W = tgt.shape                     # NCHW frames
        hs, ws = disp.shape[2], disp.shape[3]
        dev = tgt.device
        g = _f32c(g_loss.reshape(1))
        d_up = torch.empty(n_src, B, H, W, device=dev, dtype=torch.float32)      # one plane per source frame
        nblk = lib.td_photo_bwd_num_blocks(B, H, W)
        dP_part = torch.empty(nblk, n_src * 12, device=dev, dtype=torch.float32)
        st = native.stream()
        native.check(lib.td_photo_bwd(native.ptr(tgt), native.ptr_array(srcs), native.ptr(tgt_x), native.ptr_array(srcs_x), n_src,
                                      native.ptr(disp),
                                      native.ptr(P), native.ptr(invK), native.ptr(argmin), native.ptr(coef),
                                      int(automask), native.ptr(g), inv_count, B, H, W, hs, ws, float(min_depth),
                                      float(max_depth), native.ptr(d_up), native.ptr(dP_part), st),
                     "td_photo_bwd")
        d_disp = torch.empty_like(disp)
        native.check(lib.td_upsample_adjoint_planes(native.ptr(d_up), n_src, B, H, W, hs, ws, native.ptr(d_disp), 0, st),
                     "td_upsample_adjoint_planes")
        dP = torch.empty_like(P)
        native.check(lib.td_reduce_dP(native.ptr(dP_part), n_src, B, H, W, native.ptr(dP), st), "td_reduce_dP")
        return d_disp, dP, None, None, None, None, None, None, None, None, None, None, None


def photometric_scale_loss(disp, P, tgt, srcs, invK, idloss=None, noise=None, min_depth=0.1,
                           max_depth=100.0, n_scales=4, keep_warped=False):
    """One scale of generate_images_pred + automask + min-reprojection
    (mono/model/mono_fm_joint/net.py:181-194; mono/model/mono_fm_joint_inpaint/net.py:101-117).

    disp [B,1,hs,ws]; P [n_src,B,3,4] = (K @ T)[:, :3, :] per source frame; ``tgt`` [B,3,H,W] with ``srcs``, or a PackedFrames
    (``srcs`` ignored); idloss [B,n_src,H,W] from photo_identity (None = no automask); noise [n_src,B,H,W] N(0,1) draws or None.
    Returns (loss = mean(min)/n_scales, argmin uint8 [B,H,W], warped [n_src,B,3,H,W] or empty).
    """
    fr = _frames(tgt, srcs)                        # RGBX frames (packed here unless the caller packed them once per step)
    invK = _f32c(invK)
    if idloss is not None:
        # the kernel reads [B,H,W,n_src]; photo_identity's result is a [B,n_src,H,W] VIEW of exactly that memory
        idloss = idloss.permute(0, 2, 3, 1)
        if idloss.dtype != torch.float32 or not idloss.is_contiguous():
            idloss = idloss.float().contiguous()
    if noise is not None:
        noise = _f32c(noise)
    return _PhotometricScaleLoss.apply(disp, P, fr.tgt, fr.srcs, fr.tgt_planar, fr.srcs_planar, invK, idloss, noise, min_depth,
                                       max_depth, n_scales, keep_warped)


class _SmoothLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, disp, img, normalize, weight):
        lib = native.load()
        disp = _f32c(disp)
        B, _, h, w = disp.shape
        dev = disp.device
        mean = torch.empty(B, device=dev, dtype=torch.float32)
        nblk = lib.td_smooth_num_blocks(B, h, w)
        partial = torch.empty(nblk, 6, device=dev, dtype=torch.float32)
        loss = torch.empty(1, device=dev, dtype=torch.float32)
        st = native.stream()
        native.check(lib.td_smooth_fwd(native.ptr(disp), native.ptr(img), B, h, w, int(normalize),
                                       native.ptr(mean), native.ptr(partial), st), "td_smooth_fwd")
        native.check(lib.td_smooth_finish(native.ptr(partial), B, h, w, float(weight), native.ptr(loss), st),
                     "td_smooth_finish")
        ctx.save_for_backward(disp, img, mean)
        ctx.meta = (bool(normalize), float(weight), nblk)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g_loss):
        lib = native.load()
        disp, img, mean = ctx.saved_tensors
        normalize, weight, nblk = ctx.meta
        B, _, h, w = disp.shape
        dev = disp.device
        g = _f32c(g_loss.reshape(1))
        g_hat = torch.empty(B, h, w, device=dev, dtype=torch.float32)
        dot = torch.empty(nblk, device=dev, dtype=torch.float32)
        d_disp = torch.empty_like(disp)
        native.check(lib.td_smooth_bwd(native.ptr(disp), native.ptr(img), native.ptr(mean), B, h, w,
                                       int(normalize), native.ptr(g), weight, native.ptr(g_hat),
                                       native.ptr(dot), native.ptr(d_disp), 0, native.stream()), "td_smooth_bwd")
        return d_disp, None, None, None


def smooth_loss(disp, img_at_scale, normalize=True, weight=1.0):
    """weight * get_smooth_loss(disp / (mean(disp) + 1e-7), img)
    (mono/model/mono_fm_joint_inpaint/net.py:122-131, mono/model/mono_fm_joint/net.py:279-307).
    ``img_at_scale`` is the target area-resized to disp's size (area_downsample)."""
    img = _f32c(img_at_scale)
    if img.shape[2:] != disp.shape[2:]:
        raise ValueError("img_at_scale must already have disp's spatial size")
    return _SmoothLoss.apply(disp, img, normalize, weight)


def _raw(t):
    import ctypes
    return ctypes.c_void_p(t.data_ptr())


class _MaxPool5(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        lib = native.load()
        N, C, H, W = x.shape
        out = torch.empty_like(x, memory_format=torch.channels_last)
        idx = torch.empty((N, H, W, C), device=x.device, dtype=torch.uint8)
        native.check(lib.td_maxpool5_fwd(_raw(x), native.DTYPE_CODES[x.dtype], N, H, W, C, _raw(out), _raw(idx),
                                         native.stream()), "td_maxpool5_fwd")
        ctx.save_for_backward(idx)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = native.load()
        (idx,) = ctx.saved_tensors
        N, H, W, C = idx.shape
        if not g.is_contiguous(memory_format=torch.channels_last):
            g = g.contiguous(memory_format=torch.channels_last)
        gin = torch.empty_like(g, memory_format=torch.channels_last)
        native.check(lib.td_maxpool5_bwd(_raw(g), _raw(idx), native.DTYPE_CODES[g.dtype], N, H, W, C, _raw(gin),
                                         native.stream()), "td_maxpool5_bwd")
        return gin


class _MaxPool3s2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        lib = native.load()
        N, C, H, W = x.shape
        Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        out = torch.empty((N, C, Ho, Wo), device=x.device, dtype=x.dtype, memory_format=torch.channels_last)
        idx = torch.empty((N, Ho, Wo, C), device=x.device, dtype=torch.uint8)
        native.check(lib.td_maxpool3s2_fwd(_raw(x), native.DTYPE_CODES[x.dtype], N, H, W, C, _raw(out), _raw(idx),
                                           native.stream()), "td_maxpool3s2_fwd")
        ctx.save_for_backward(idx)
        ctx.in_hw = (H, W)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = native.load()
        (idx,) = ctx.saved_tensors
        N, _, _, C = idx.shape
        H, W = ctx.in_hw
        if not g.is_contiguous(memory_format=torch.channels_last):
            g = g.contiguous(memory_format=torch.channels_last)
        gin = torch.empty((N, C, H, W), device=g.device, dtype=g.dtype, memory_format=torch.channels_last)
        native.check(lib.td_maxpool3s2_bwd(_raw(g), _raw(idx), native.DTYPE_CODES[g.dtype], N, H, W, C, _raw(gin),
                                           native.stream()), "td_maxpool3s2_bwd")
        return gin


def maxpool3s2(x):
    """nn.MaxPool2d(3, 2, 1) on a channels_last CUDA tensor (reference: resnet.py:101)."""
    if not maxpool5_supported(x):
        raise native.NativeLibraryError("maxpool3s2 needs a channels_last f32/bf16 HIP tensor with C % 8 == 0")
    return _MaxPool3s2.apply(x)


def maxpool5_supported(x):
    return (x.is_cuda and x.dim() == 4 and x.dtype in native.DTYPE_CODES and x.shape[1] % 8 == 0
            and x.is_contiguous(memory_format=torch.channels_last))


def maxpool5(x):
    """nn.MaxPool2d(5, 1, 2) on a channels_last CUDA tensor (reference: layers.py:208,213)."""
    if not maxpool5_supported(x):
        raise native.NativeLibraryError("maxpool5 needs a channels_last f32/bf16 HIP tensor with C % 8 == 0")
    return _MaxPool5.apply(x)


class _CRP(torch.autograd.Function):
    """Chained residual pooling (reference: CRPBlock, mono/model/mono_fm_joint/layers.py:200-215): top_0 = x,
    top_i = conv1x1_i(maxpool5(top_{i-1})), out = x + sum_i top_i -- as ONE autograd node over hand-written kernels:

      forward   per stage: the 5x5 pool (column march) and the pointwise convolution on the MFMA GEMM whose epilogue also
                advances the running sum (td_conv1x1_fwd_sum: two outputs, no separate add pass)
      backward  per stage, last to first: weight gradient (td_conv1x1_wgrad), data gradient (td_conv1x1_dgrad) and the pool's
                backward with the gradient the stage's input ALSO receives straight from the running sum added while it is
                written out (td_maxpool5_bwd_add): autograd's gradient-accumulation adds are gone

    Against the per-op nodes: 8 tensor adds per block and direction fewer, and the twelve 1x1 convolution calls of a block
    (forward / data / weight gradient through MIOpen / CK, the weight gradients with their zero-fill and cast passes) run on the
    deterministic hand-written GEMMs."""

    @staticmethod
    def forward(ctx, x, *ws):
        lib = native.load()
        strm = native.stream()
        Nb, C, H, W = x.shape
        M = Nb * H * W
        BF = native.DTYPE_CODES[x.dtype]
        top, run = x, x
        saved = []
        for w in ws:
            pooled = torch.empty_like(x, memory_format=torch.channels_last)
            idx = torch.empty((Nb, H, W, C), device=x.device, dtype=torch.uint8)
            native.check(lib.td_maxpool5_fwd(_raw(top), BF, Nb, H, W, C, _raw(pooled), _raw(idx), strm), "td_maxpool5_fwd")
            top = torch.empty_like(x, memory_format=torch.channels_last)
            run_out = torch.empty_like(x, memory_format=torch.channels_last)
            native.check(lib.td_conv1x1_fwd_sum(_raw(pooled), _raw(w), M, C, C, _raw(run), _raw(top), _raw(run_out), strm),
                         "td_conv1x1_fwd_sum")
            run = run_out
            saved += [pooled, idx]
        ctx.save_for_backward(*ws, *saved)
        ctx.n = len(ws)
        return run

    @staticmethod
    def backward(ctx, g):
        with wgrad_group():      # the node's 1x1 weight gradients leave as one grouped launch at the end
            return _CRP._backward(ctx, g)

    @staticmethod
    def _backward(ctx, g):
        lib = native.load()
        strm = native.stream()
        n = ctx.n
        ws, saved = ctx.saved_tensors[:n], ctx.saved_tensors[n:]
        Nb, C, H, W = g.shape
        M = Nb * H * W
        if g.dtype != torch.bfloat16 or not g.is_contiguous(memory_format=torch.channels_last):
            g = g.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        BF = native.DTYPE_CODES[g.dtype]
        g_top = g
        dws = [None] * n
        for i in reversed(range(n)):
            w, pooled, idx = ws[i], saved[2 * i], saved[2 * i + 1]
            dws[i] = conv1x1_wgrad(g_top, pooled, w, M, C, C, H, W, 1)
            d_pool = torch.empty_like(g, memory_format=torch.channels_last)
            native.check(lib.td_conv1x1_dgrad(_raw(g_top), _raw(w), M, 1, C, C, None, _raw(d_pool), strm), "td_conv1x1_dgrad")
            g_prev = torch.empty_like(g, memory_format=torch.channels_last)
            native.check(lib.td_maxpool5_bwd_add(_raw(d_pool), _raw(idx), _raw(g), BF, Nb, H, W, C, _raw(g_prev), strm),
                         "td_maxpool5_bwd_add")
            g_top = g_prev
        return (g_top,) + tuple(dws)


def crp_supported(x, ws):
    C = x.shape[1]
    return (x.is_cuda and x.dim() == 4 and x.dtype == torch.bfloat16 and x.is_contiguous(memory_format=torch.channels_last)
            and C % 64 == 0 and len(ws) >= 1
            and all(w.dtype == torch.bfloat16 and tuple(w.shape) == (C, C, 1, 1)
                    and (w.is_contiguous() or w.is_contiguous(memory_format=torch.channels_last)) for w in ws))


def crp_block(x, ws):
    """x + sum_i top_i, top_i = conv1x1(maxpool5(top_{i-1}), ws[i]) (reference: layers.py:200-215) on bf16 channels_last HIP
    tensors with C % 64 == 0 (one autograd node, see _CRP)."""
    if not crp_supported(x, ws):
        raise native.NativeLibraryError("crp_block needs a bf16 channels_last HIP tensor with C % 64 == 0 and [C, C, 1, 1] weights")
    return _CRP.apply(x, *ws)


class _JoinChannels(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, tail):
        lib = native.load()
        N, C0, H, W = a.shape
        C1, C2 = b.shape[1], tail.shape[1]
        out = torch.empty((N, C0 + C1 + 8, H, W), device=a.device, dtype=a.dtype, memory_format=torch.channels_last)
        native.check(lib.td_join_fwd(_raw(a), _raw(b), _raw(tail), native.DTYPE_CODES[a.dtype], N * H * W, C0, C1, C2,
                                     _raw(out), native.stream()), "td_join_fwd")
        ctx.dims = (N, C0, C1, C2, H, W)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = native.load()
        N, C0, C1, C2, H, W = ctx.dims
        if not g.is_contiguous(memory_format=torch.channels_last):
            g = g.contiguous(memory_format=torch.channels_last)
        mk = lambda c: torch.empty((N, c, H, W), device=g.device, dtype=g.dtype, memory_format=torch.channels_last)
        ga, gb, gt = mk(C0), mk(C1), mk(C2)
        native.check(lib.td_join_bwd(_raw(g), native.DTYPE_CODES[g.dtype], N * H * W, C0, C1, C2, _raw(ga), _raw(gb),
                                     _raw(gt), native.stream()), "td_join_bwd")
        return ga, gb, gt


class _JoinChannelsUp2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b_half, tail):
        lib = native.load()
        N, C0, H, W = a.shape
        C1, C2 = b_half.shape[1], tail.shape[1]
        out = torch.empty((N, C0 + C1 + 8, H, W), device=a.device, dtype=a.dtype, memory_format=torch.channels_last)
        native.check(lib.td_join_up2_fwd(_raw(a), _raw(b_half), _raw(tail), native.DTYPE_CODES[a.dtype], N, H, W, C0, C1, C2,
                                         _raw(out), native.stream()), "td_join_up2_fwd")
        ctx.dims = (N, C0, C1, C2, H, W)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = native.load()
        N, C0, C1, C2, H, W = ctx.dims
        if not g.is_contiguous(memory_format=torch.channels_last):
            g = g.contiguous(memory_format=torch.channels_last)
        mk = lambda c, h, w: torch.empty((N, c, h, w), device=g.device, dtype=g.dtype, memory_format=torch.channels_last)
        ga, gb, gt = mk(C0, H, W), mk(C1, H // 2, W // 2), mk(C2, H, W)
        native.check(lib.td_join_up2_bwd(_raw(g), native.DTYPE_CODES[g.dtype], N, H, W, C0, C1, C2, _raw(ga), _raw(gb), _raw(gt),
                                         native.stream()), "td_join_up2_bwd")
        return ga, gb, gt


def join_channels_up2_supported(a, b_half, tail):
    cl = lambda t: t.is_contiguous(memory_format=torch.channels_last)
    return (a.is_cuda and a.dim() == 4 and a.dtype in native.DTYPE_CODES and b_half.dtype == a.dtype and tail.dtype == a.dtype
            and a.shape[1] % 8 == 0 and b_half.shape[1] % 8 == 0 and 1 <= tail.shape[1] <= 8
            and a.shape[0] == b_half.shape[0] == tail.shape[0] and a.shape[2:] == tail.shape[2:]
            and a.shape[2] == 2 * b_half.shape[2] and a.shape[3] == 2 * b_half.shape[3]
            and cl(a) and cl(b_half) and cl(tail))


def join_channels_up2(a, b_half, tail):
    """torch.cat((a, upsample_x2_nearest(b_half), tail, zeros), 1) up to C0 + C1 + 8 channels, without materialising the
    up-sampled operand (reference: depth_decoder.py:89-103, where x was up-sampled at the end of the previous stage)."""
    if not join_channels_up2_supported(a, b_half, tail):
        raise native.NativeLibraryError("join_channels_up2 needs channels_last f32/bf16 HIP tensors, b at half resolution")
    return _JoinChannelsUp2.apply(a, b_half, tail)


def join_channels_supported(a, b, tail):
    cl = lambda t: t.is_contiguous(memory_format=torch.channels_last)
    return (a.is_cuda and a.dim() == 4 and a.dtype in native.DTYPE_CODES and b.dtype == a.dtype and tail.dtype == a.dtype
            and a.shape[1] % 8 == 0 and b.shape[1] % 8 == 0 and 1 <= tail.shape[1] <= 8
            and a.shape[0] == b.shape[0] == tail.shape[0] and a.shape[2:] == b.shape[2:] == tail.shape[2:]
            and cl(a) and cl(b) and cl(tail))


def join_channels(a, b, tail):
    """torch.cat((a, b, tail, zeros), 1) up to C0 + C1 + 8 channels on channels_last HIP tensors
    (reference: depth_decoder.py:89-103 plus the channel-alignment padding of Conv3x3)."""
    if not join_channels_supported(a, b, tail):
        raise native.NativeLibraryError("join_channels needs channels_last f32/bf16 HIP tensors, C0 % 8 == C1 % 8 == 0, C2 <= 8")
    return _JoinChannels.apply(a, b, tail)


class _BatchNormAct(torch.autograd.Function):
    """F.batch_norm(training=True) [+ residual] [-> relu] on a channels_last tensor, three launches each way."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, residual, momentum, eps, relu, groups):
        lib = native.load()
        N, C, H, W = x.shape
        M = N * H * W
        y = torch.empty_like(x, memory_format=torch.channels_last)
        mean = torch.empty(groups * C, device=x.device, dtype=torch.float32)
        invstd = torch.empty(groups * C, device=x.device, dtype=torch.float32)
        ws = torch.empty(lib.td_bn_workspace_floats(M, groups, C), device=x.device, dtype=torch.float32)
        native.check(lib.td_bn_fwd(_raw(x), _raw(residual) if residual is not None else None, native.DTYPE_CODES[x.dtype],
                                   native.ptr(weight), native.ptr(bias),
                                   native.ptr(running_mean) if running_mean is not None else None,
                                   native.ptr(running_var) if running_var is not None else None,
                                   float(momentum), float(eps), int(relu), M, groups, C, _raw(y), native.ptr(mean),
                                   native.ptr(invstd), native.ptr(ws), native.stream()), "td_bn_fwd")
        # the ReLU mask is re-derived from x in backward unless a residual went into the pre-activation
        ctx.save_for_backward(x, y if (relu and residual is not None) else None, weight, bias, mean, invstd)
        ctx.relu, ctx.has_res, ctx.groups = bool(relu), residual is not None, groups
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = native.load()
        x, y, weight, bias, mean, invstd = ctx.saved_tensors
        N, C, H, W = x.shape
        M = N * H * W
        if dy.dtype != x.dtype or not dy.is_contiguous(memory_format=torch.channels_last):
            dy = dy.to(x.dtype).contiguous(memory_format=torch.channels_last)
        dx = torch.empty_like(x, memory_format=torch.channels_last)
        # relu + residual: the masked gradient is a second output; without relu the residual gradient is dy itself
        dres = torch.empty_like(x, memory_format=torch.channels_last) if (ctx.has_res and ctx.relu) else None
        dgamma = torch.empty(C, device=x.device, dtype=torch.float32)
        dbeta = torch.empty(C, device=x.device, dtype=torch.float32)
        ws = torch.empty(lib.td_bn_workspace_floats(M, ctx.groups, C), device=x.device, dtype=torch.float32)
        native.check(lib.td_bn_bwd(_raw(dy), _raw(x), _raw(y) if y is not None else None, native.DTYPE_CODES[x.dtype],
                                   native.ptr(weight), native.ptr(bias), native.ptr(mean), native.ptr(invstd), int(ctx.relu), M, ctx.groups, C,
                                   _raw(dx), _raw(dres) if dres is not None else None, native.ptr(dgamma),
                                   native.ptr(dbeta), native.ptr(ws), native.stream()), "td_bn_bwd")
        if ctx.has_res and not ctx.relu:
            dres = dy
        return dx, dgamma.to(weight.dtype), dbeta.to(weight.dtype), None, None, dres, None, None, None, None


class _Conv1x1BatchNormAct(torch.autograd.Function):
    """relu?( batch_norm( conv1x1(x, w) ) [+ residual] ) with the convolution on the hand-written MFMA GEMM and the batch
    statistics taken from its epilogue (no statistics pass over the convolution output): GEMM, [shrink of the partial rows when
    there are more than 96], apply with the statistics finished in its prologue.  Backward: td_bn_bwd, the data gradient through
    ATen (MIOpen), the weight gradient on the hand-written MFMA kernel (td_conv1x1_wgrad).  (The bottlenecks of the training
    step take the block-level node _Bottleneck since round 4; this per-layer node serves the BasicBlock down-sample branch and
    TD_NO_FUSED_BLOCK.)"""

    @staticmethod
    def forward(ctx, x, w, weight, bias, running_mean, running_var, residual, momentum, eps, relu, groups, stride):
        lib = native.load()
        Nb, K, Hi, Wi = x.shape
        N = w.shape[0]
        Ho, Wo = (Hi - 1) // stride + 1, (Wi - 1) // stride + 1
        M = Nb * Ho * Wo
        yc = torch.empty((Nb, N, Ho, Wo), device=x.device, dtype=torch.bfloat16, memory_format=torch.channels_last)
        S = lib.td_conv1x1_stat_rows(M, groups, N)
        part = torch.empty(groups * S * N * 2, device=x.device, dtype=torch.float32)
        native.check(lib.td_conv1x1_fwd(_raw(x), _raw(w), M, groups, K, N, Hi, Wi, stride, _raw(yc), native.ptr(part),
                                        native.stream()), "td_conv1x1_fwd")
        y = torch.empty_like(yc, memory_format=torch.channels_last)
        mean = torch.empty(groups * N, device=x.device, dtype=torch.float32)
        invstd = torch.empty(groups * N, device=x.device, dtype=torch.float32)
        native.check(lib.td_bn_fwd_from_partials(_raw(yc), _raw(residual) if residual is not None else None,
                                                 native.DTYPE_CODES[yc.dtype], native.ptr(weight), native.ptr(bias),
                                                 native.ptr(running_mean) if running_mean is not None else None,
                                                 native.ptr(running_var) if running_var is not None else None,
                                                 float(momentum), float(eps), int(relu), M, groups, N, native.ptr(part), S, _raw(y),
                                                 native.ptr(mean), native.ptr(invstd), native.stream()), "td_bn_fwd_from_partials")
        ctx.save_for_backward(x, w, yc, y if (relu and residual is not None) else None, weight, bias, mean, invstd)
        ctx.relu, ctx.has_res, ctx.groups, ctx.stride = bool(relu), residual is not None, groups, stride
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = native.load()
        x, w, yc, y, weight, bias, mean, invstd = ctx.saved_tensors
        Nb, N, Ho, Wo = yc.shape
        M = Nb * Ho * Wo
        if dy.dtype != yc.dtype or not dy.is_contiguous(memory_format=torch.channels_last):
            dy = dy.to(yc.dtype).contiguous(memory_format=torch.channels_last)
        dyc = torch.empty_like(yc, memory_format=torch.channels_last)
        dres = torch.empty_like(yc, memory_format=torch.channels_last) if (ctx.has_res and ctx.relu) else None
        dgamma = torch.empty(N, device=x.device, dtype=torch.float32)
        dbeta = torch.empty(N, device=x.device, dtype=torch.float32)
        ws = torch.empty(lib.td_bn_workspace_floats(M, ctx.groups, N), device=x.device, dtype=torch.float32)
        native.check(lib.td_bn_bwd(_raw(dy), _raw(yc), _raw(y) if y is not None else None, native.DTYPE_CODES[yc.dtype],
                                   native.ptr(weight), native.ptr(bias), native.ptr(mean), native.ptr(invstd), int(ctx.relu), M,
                                   ctx.groups, N, _raw(dyc), _raw(dres) if dres is not None else None, native.ptr(dgamma),
                                   native.ptr(dbeta), native.ptr(ws), native.stream()), "td_bn_bwd")
        if ctx.has_res and not ctx.relu:
            dres = dy
        dx = dw = None
        if ctx.needs_input_grad[0]:      # data gradient: MIOpen / CK
            dx = torch.ops.aten.convolution_backward(dyc, x, w, None, [ctx.stride, ctx.stride], [0, 0], [1, 1], False, [0, 0], 1,
                                                     [True, False, False])[0]
        if ctx.needs_input_grad[1]:      # weight gradient: hand-written MFMA kernel (transposed LDS reads, ordered slab sum)
            K, Hi, Wi = x.shape[1], x.shape[2], x.shape[3]
            dw = conv1x1_wgrad(dyc, x, w, M, K, N, Hi, Wi, ctx.stride)
        return dx, dw, dgamma.to(weight.dtype), dbeta.to(weight.dtype), None, None, dres, None, None, None, None, None


def conv1x1_bn_act_supported(x, w, bn_weight, stride=1):
    return (x.is_cuda and x.dim() == 4 and x.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and w.dim() == 4
            and w.shape[2] == 1 and w.shape[3] == 1 and w.shape[1] == x.shape[1] and x.shape[1] % 64 == 0 and w.shape[0] % 64 == 0
            and stride in (1, 2) and bn_weight is not None and bn_weight.dtype == torch.float32
            and x.is_contiguous(memory_format=torch.channels_last)
            and (w.is_contiguous() or w.is_contiguous(memory_format=torch.channels_last)))


def conv1x1_bn_act(x, w, weight, bias, running_mean, running_var, momentum, eps, residual=None, relu=False, groups=1, stride=1):
    """conv1x1 -> training-mode BatchNorm2d -> [+ residual] -> [relu] (reference: resnet.py:66-86 conv1/bn1, conv3/bn3 and the
    down-sample branch :119-127) for bf16 channels_last HIP tensors, Cin % 64 == Cout % 64 == 0."""
    if x.shape[0] % groups:
        raise ValueError("batch %d is not %d stacked passes" % (x.shape[0], groups))
    if not conv1x1_bn_act_supported(x, w, weight, stride):
        raise native.NativeLibraryError("conv1x1_bn_act needs bf16 channels_last HIP tensors with Cin % 64 == Cout % 64 == 0")
    if residual is not None and (residual.dtype != x.dtype or not residual.is_contiguous(memory_format=torch.channels_last)):
        residual = residual.to(x.dtype).contiguous(memory_format=torch.channels_last)
    return _Conv1x1BatchNormAct.apply(x, w, weight, bias, running_mean, running_var, residual, momentum, eps, relu, groups, stride)


ACT_CODES = {None: 0, "elu": 1, "leaky_relu": 2}


class _ConvBiasAct(torch.autograd.Function):
    """act(conv2d(x, w) + b) for a stride-1 unpadded convolution (the decoders' Conv3x3 after its reflection pad): the
    convolution through MIOpen WITHOUT its bias, then bias + activation in one in-place pass (td_bias_act_fwd); backward: the
    activation's adjoint and the bias gradient in one pass (td_bias_act_bwd) in front of MIOpen's data / weight gradients.
    Reference: ConvBlock (layers.py:143-155), depth_decoder.py:89-103."""

    @staticmethod
    def forward(ctx, x, w, b, act):
        lib = native.load()
        y = torch.ops.aten.convolution(x, w, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1)
        if not y.is_contiguous(memory_format=torch.channels_last):
            y = y.contiguous(memory_format=torch.channels_last)
        B, C, H, W = y.shape
        native.check(lib.td_bias_act_fwd(_raw(y), _raw(b) if b is not None else None, native.DTYPE_CODES[b.dtype] if b is not None else 0,
                                         native.DTYPE_CODES[y.dtype], B * H * W, C, ACT_CODES[act], _raw(y), native.stream()),
                     "td_bias_act_fwd")
        ctx.save_for_backward(x, w, y if act is not None else None, b)
        ctx.act = act
        return y

    @staticmethod
    def backward(ctx, g):
        lib = native.load()
        x, w, a, b = ctx.saved_tensors
        B, C, H, W = g.shape
        if g.dtype != w.dtype or not g.is_contiguous(memory_format=torch.channels_last):
            g = g.to(w.dtype).contiguous(memory_format=torch.channels_last)
        M = B * H * W
        gy = torch.empty_like(g, memory_format=torch.channels_last) if ctx.act is not None else g
        db = torch.empty_like(b) if b is not None and ctx.needs_input_grad[2] else None
        ws = torch.empty(lib.td_bias_act_workspace_floats(M, C), device=g.device, dtype=torch.float32)
        if ctx.act is not None or db is not None:
            native.check(lib.td_bias_act_bwd(_raw(g), _raw(a) if a is not None else None, native.DTYPE_CODES[g.dtype], M, C,
                                             ACT_CODES[ctx.act], _raw(gy) if ctx.act is not None else None,
                                             _raw(db) if db is not None else None, native.DTYPE_CODES[db.dtype] if db is not None else 0,
                                             native.ptr(ws), native.stream()), "td_bias_act_bwd")
        dx, dw, _ = torch.ops.aten.convolution_backward(gy, x, w, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1,
                                                        [ctx.needs_input_grad[0], ctx.needs_input_grad[1], False])
        return dx, dw, db, None


def conv_bias_act_supported(x, w, b, act):
    cout = w.shape[0]
    return (act in ACT_CODES and x.is_cuda and x.dim() == 4 and x.dtype == torch.bfloat16 and w.dtype == torch.bfloat16
            and x.is_contiguous(memory_format=torch.channels_last) and (b is None or b.dtype in native.DTYPE_CODES)
            and 8 <= cout <= 256 and cout % 8 == 0 and 256 % (cout // 8) == 0 and x.shape[1] == w.shape[1])


def conv_bias_act(x, w, b, act=None):
    """act(F.conv2d(x, w, b)) (stride 1, no padding) with bias + activation and their adjoints fused (see _ConvBiasAct)."""
    if not conv_bias_act_supported(x, w, b, act):
        raise native.NativeLibraryError("conv_bias_act needs bf16 channels_last HIP tensors, Cout in {8, 16, 32, 64, 128, 256}")
    return _ConvBiasAct.apply(x, w, b, act)


# ---- weight gradients of the 1x1 convolutions: immediate, or enqueued and launched as one group ---------------------------
_WGRAD_QUEUE = None          # the innermost active wgrad_group, or None
WGRAD_GROUP = os.environ.get("TD_WGRAD_GROUP", "node")      # "node" | "step" | "off"


class wgrad_group:
    """A scope whose 1x1 weight gradients (conv1x1_wgrad) are ENQUEUED and launched together at its exit by
    td_conv1x1_wgrad_group -- 2 launches per <= 40 problems instead of 2 per problem, bit-equal results.

    ``assign=False`` (a backward node around its own three or four weight gradients: scope "node"): conv1x1_wgrad returns the
    output tensor at once and the exit, still inside the node's backward, writes it -- autograd never sees an unwritten buffer.
    ``assign=True`` (around a whole ``loss.backward()``: scope "step", TD_WGRAD_GROUP=step): a deferred node hands autograd NO
    gradient for its weight (None); the exit computes the group and then stores (first use) or adds (a weight used twice) each
    result into ``weight.grad`` itself.  Only leaf weights are deferred that way and per-parameter gradient hooks do not see
    them, so tripled_amd.step.TrainStep allows it only with the flat parameter store and without the overlapped bucket engine.
    Measured (profiles/r04/wgrad_group_v1.txt): "step" is SLOWER than no grouping -- at the end of backward every dz and x comes
    from HBM again, while the immediate launches find them in L2 / MALL -- so "node" is the default."""

    def __init__(self, assign=False, scope="node"):
        self.assign, self.scope, self.items = assign, scope, []

    def __enter__(self):
        global _WGRAD_QUEUE
        self.prev = _WGRAD_QUEUE
        # an inner "node" scope inside an active "step" scope leaves the queue of the step in place
        self.active = WGRAD_GROUP == self.scope and not (self.prev is not None and self.prev.assign)
        if self.active:
            _WGRAD_QUEUE = self
        return self

    def __exit__(self, *exc):
        global _WGRAD_QUEUE
        if self.active:
            _WGRAD_QUEUE = self.prev
            if self.items and exc[0] is None:
                flush_wgrads(self.items)
        return False


def deferred_wgrads():
    """``with deferred_wgrads(): loss.backward()`` -- the "step" scope of wgrad_group (active under TD_WGRAD_GROUP=step)."""
    return wgrad_group(assign=True, scope="step")


def flush_wgrads(queue):
    import ctypes
    lib = native.load()
    n = len(queue)
    P, LL, I = (ctypes.c_void_p * n), (ctypes.c_longlong * n), (ctypes.c_int * n)
    dy, x, dw, ws = P(), P(), P(), P()
    for i, q in enumerate(queue):
        dy[i], x[i], dw[i], ws[i] = q["dy"].data_ptr(), q["x"].data_ptr(), q["dw"].data_ptr(), q["ws"].data_ptr()
    native.check(lib.td_conv1x1_wgrad_group(n, dy, x, LL(*[q["M"] for q in queue]), I(*[q["K"] for q in queue]),
                                            I(*[q["N"] for q in queue]), I(*[q["Hi"] for q in queue]), I(*[q["Wi"] for q in queue]),
                                            I(*[q["stride"] for q in queue]), I(*[native.DTYPE_CODES[q["dw"].dtype] for q in queue]),
                                            dw, ws, native.stream()), "td_conv1x1_wgrad_group")
    for q in queue:
        w = q["w"]
        if w is None:
            continue
        if w.grad is None:
            w.grad = q["dw"]
        else:
            w.grad.add_(q["dw"])


def conv1x1_wgrad(dz, inp, w, M, K, N, Hi, Wi, stride):
    """dW of a 1x1 convolution (td_conv1x1_wgrad) for an autograd backward to return: launched now, or enqueued in the
    innermost wgrad_group -- which returns the tensor its exit will write ("node" scope), or None for a leaf weight whose
    .grad the exit fills in ("step" scope)."""
    lib = native.load()
    dw = torch.empty_like(w)
    wsw = torch.empty(lib.td_conv1x1_wgrad_workspace_floats(M, K, N), device=dz.device, dtype=torch.float32)
    q = _WGRAD_QUEUE
    if q is not None and (not q.assign or (w.is_leaf and w.requires_grad)):
        q.items.append(dict(dy=dz, x=inp, dw=dw, ws=wsw, w=w if q.assign else None, M=M, K=K, N=N, Hi=Hi, Wi=Wi, stride=stride))
        return None if q.assign else dw
    native.check(lib.td_conv1x1_wgrad(_raw(dz), _raw(inp), M, K, N, Hi, Wi, stride, native.DTYPE_CODES[w.dtype], _raw(dw),
                                      native.ptr(wsw), native.stream()), "td_conv1x1_wgrad")
    return dw


def conv3x3_wgrad_wins(B, Ho, Wo, C, N, stride):
    """Dispatch table of td_conv3x3_wgrad, from profiles/r04/conv3x3_wgrad_bench_v2.txt: the kernel beats MIOpen's weight
    gradient (its zero-fill and cast passes included) on the 64-channel maps of 48 x 160 and larger; everywhere else MIOpen is
    20-45 % faster and keeps the layer."""
    return stride == 1 and C == 64 and N == 64 and Wo >= 160 and Ho >= 48 and B * Ho * Wo < (1 << 31) - 128


def conv3x3_wgrad(dy, x, w, pad):
    """dW of a 3x3 stride-1 convolution (td_conv3x3_wgrad): dy [B,N,Ho,Wo], x [B,C,Hi,Wi] bf16 channels_last -> like w."""
    lib = native.load()
    B, N, Ho, Wo = dy.shape
    C = x.shape[1]
    dw = torch.empty_like(w, memory_format=torch.channels_last)
    ws = torch.empty(lib.td_conv3x3_wgrad_workspace_floats(B, Ho, Wo, C, N), device=x.device, dtype=torch.float32)
    native.check(lib.td_conv3x3_wgrad(_raw(dy), _raw(x), B, Ho, Wo, C, N, pad, native.DTYPE_CODES[w.dtype], _raw(dw), native.ptr(ws),
                                      native.stream()), "td_conv3x3_wgrad")
    return dw


class BnState:
    """What a BatchNorm layer contributes to a fused block: affine parameters are passed as tensors (autograd), the rest here."""
    __slots__ = ("running_mean", "running_var", "momentum", "eps")

    def __init__(self, bn):
        self.running_mean, self.running_var = bn.running_mean, bn.running_var
        self.momentum, self.eps = float(bn.momentum), float(bn.eps)


def _cl_empty(n, c, h, w, like):
    return torch.empty((n, c, h, w), device=like.device, dtype=torch.bfloat16, memory_format=torch.channels_last)


class _Bottleneck(torch.autograd.Function):
    """The whole ResNet bottleneck (reference: Bottleneck.forward, mono/model/mono_fm_joint/resnet.py:66-86) as ONE autograd
    node, so that BatchNorm work can ride on the neighbouring 1x1 GEMMs across layer boundaries (csrc/td_conv1x1.hip, "fused
    forms"):

      forward   conv1 GEMM (+ bn1 statistics) -> bn1 apply + relu -> conv2 (3x3: MIOpen) -> bn2 statistics pass
                -> conv3 GEMM with relu(bn2(.)) applied to its operand while it is staged (+ bn3 statistics)
                -> [down-sample GEMM + bn] -> bn3 apply + identity + relu
      backward  bn3 backward -> conv3 data gradient whose epilogue forms bn2's backward sums; conv3 weight gradient
                -> bn2 dx -> conv2 backward (MIOpen) -> bn1 statistics pass
                -> conv1 data gradient with bn1's backward applied to its operand while it is staged and the identity
                   branch's gradient added in its epilogue; conv1 weight gradient

    Against the per-layer nodes this drops the bn2 apply pass, the bn2 backward statistics pass, the bn1 dx pass and the
    residual-gradient add (4 launches and their passes over the activations per block), and both data gradients run on the
    hand-written MFMA GEMM."""

    @staticmethod
    def forward(ctx, x, w1, w2, w3, g1, b1, g2, b2, g3, b3, wd, gd, bd, st1, st2, st3, std, groups, stride):
        lib = native.load()
        strm = native.stream()
        Nb, Cin, H, W = x.shape
        c = w1.shape[0]
        Cout = w3.shape[0]
        M1 = Nb * H * W
        f32 = dict(device=x.device, dtype=torch.float32)

        def gemm_stats(inp, w, M, K, N, Hi, Wi, s_):
            y = _cl_empty(Nb, N, (Hi - 1) // s_ + 1, (Wi - 1) // s_ + 1, x)
            S = lib.td_conv1x1_stat_rows(M, groups, N)
            part = torch.empty(groups * S * N * 2, **f32)
            native.check(lib.td_conv1x1_fwd(_raw(inp), _raw(w), M, groups, K, N, Hi, Wi, s_, _raw(y), native.ptr(part), strm),
                         "td_conv1x1_fwd")
            return y, part, S

        def apply(z, part, S, gamma, beta, st, M, C, res, relu):
            y = torch.empty_like(z, memory_format=torch.channels_last)
            mean, invstd = torch.empty(groups * C, **f32), torch.empty(groups * C, **f32)
            native.check(lib.td_bn_fwd_from_partials(_raw(z), _raw(res) if res is not None else None, native.DTYPE_CODES[z.dtype],
                                                     native.ptr(gamma), native.ptr(beta), native.ptr(st.running_mean),
                                                     native.ptr(st.running_var), st.momentum, st.eps, int(relu), M, groups, C,
                                                     native.ptr(part), S, _raw(y), native.ptr(mean), native.ptr(invstd), strm),
                         "td_bn_fwd_from_partials")
            return y, mean, invstd

        # conv1 -> bn1 -> relu
        z1, p1, S1 = gemm_stats(x, w1, M1, Cin, c, H, W, 1)
        a1, mean1, invstd1 = apply(z1, p1, S1, g1, b1, st1, M1, c, None, True)
        # conv2 (3x3, stride s): MIOpen
        z2 = torch.ops.aten.convolution(a1, w2, None, [stride, stride], [1, 1], [1, 1], False, [0, 0], 1)
        if not z2.is_contiguous(memory_format=torch.channels_last):
            z2 = z2.contiguous(memory_format=torch.channels_last)
        Ho, Wo = z2.shape[2], z2.shape[3]
        M2 = Nb * Ho * Wo
        # bn2 statistics pass; its apply + relu ride on conv3's operand staging
        R2 = lib.td_bn_partial_rows(M2, groups, c)
        p2 = torch.empty(groups * R2 * c * 2, **f32)
        native.check(lib.td_bn_fwd_partials(_raw(z2), native.DTYPE_CODES[z2.dtype], M2, groups, c, native.ptr(p2), strm),
                     "td_bn_fwd_partials")
        z3 = _cl_empty(Nb, Cout, Ho, Wo, x)
        a2 = _cl_empty(Nb, c, Ho, Wo, x)
        S3 = lib.td_conv1x1_stat_rows(M2, groups, Cout)
        p3 = torch.empty(groups * S3 * Cout * 2, **f32)
        mean2, invstd2 = torch.empty(groups * c, **f32), torch.empty(groups * c, **f32)
        native.check(lib.td_conv1x1_fwd_bnrelu(_raw(z2), _raw(w3), M2, groups, c, Cout, native.ptr(p2), R2, native.ptr(g2),
                                               native.ptr(b2), native.ptr(st2.running_mean), native.ptr(st2.running_var),
                                               st2.momentum, st2.eps, native.ptr(mean2), native.ptr(invstd2), _raw(a2), _raw(z3),
                                               native.ptr(p3), strm), "td_conv1x1_fwd_bnrelu")
        # identity branch
        zd = meand = invstdd = None
        if wd is not None:
            zd, pd, Sd = gemm_stats(x, wd, M2, Cin, Cout, H, W, stride)
            shortcut, meand, invstdd = apply(zd, pd, Sd, gd, bd, std, M2, Cout, None, False)
        else:
            shortcut = x
        y, mean3, invstd3 = apply(z3, p3, S3, g3, b3, st3, M2, Cout, shortcut, True)
        ctx.save_for_backward(x, w1, w2, w3, g1, b1, g2, b2, g3, b3, wd, gd, bd, z1, a1, z2, a2, z3, y, zd, mean1, invstd1,
                              mean2, invstd2, mean3, invstd3, meand, invstdd)
        ctx.groups, ctx.stride = groups, stride
        return y

    @staticmethod
    def backward(ctx, dy):
        with wgrad_group():      # the node's 1x1 weight gradients leave as one grouped launch at the end
            return _Bottleneck._backward(ctx, dy)

    @staticmethod
    def _backward(ctx, dy):
        lib = native.load()
        strm = native.stream()
        (x, w1, w2, w3, g1, b1, g2, b2, g3, b3, wd, gd, bd, z1, a1, z2, a2, z3, y, zd, mean1, invstd1, mean2, invstd2, mean3,
         invstd3, meand, invstdd) = ctx.saved_tensors
        groups, stride = ctx.groups, ctx.stride
        Nb, Cin, H, W = x.shape
        c, Cout = w1.shape[0], w3.shape[0]
        Ho, Wo = z2.shape[2], z2.shape[3]
        M1, M2 = Nb * H * W, Nb * Ho * Wo
        f32 = dict(device=x.device, dtype=torch.float32)
        BF = native.DTYPE_CODES[torch.bfloat16]
        if dy.dtype != torch.bfloat16 or not dy.is_contiguous(memory_format=torch.channels_last):
            dy = dy.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)

        wgrad = conv1x1_wgrad          # immediate, or enqueued inside deferred_wgrads()

        def bn_bwd(dyy, zz, yy, gamma, beta, mean, invstd, relu, M, C, want_res):
            dz = torch.empty_like(zz, memory_format=torch.channels_last)
            dres = torch.empty_like(zz, memory_format=torch.channels_last) if want_res else None
            dg, db = torch.empty(C, **f32), torch.empty(C, **f32)
            ws = torch.empty(lib.td_bn_workspace_floats(M, groups, C), **f32)
            native.check(lib.td_bn_bwd(_raw(dyy), _raw(zz), _raw(yy) if yy is not None else None, BF, native.ptr(gamma),
                                       native.ptr(beta), native.ptr(mean), native.ptr(invstd), int(relu), M, groups, C, _raw(dz),
                                       _raw(dres) if dres is not None else None, native.ptr(dg), native.ptr(db), native.ptr(ws),
                                       strm), "td_bn_bwd")
            return dz, dres, dg, db

        # bn3 (+ relu, + identity): dz3 and the identity branch's gradient (dy masked by the block's relu)
        dz3, dres, dg3, db3 = bn_bwd(dy, z3, y, g3, b3, mean3, invstd3, True, M2, Cout, True)
        # conv3: data gradient with bn2's backward sums in the epilogue; weight gradient against relu(bn2(z2))
        da2 = torch.empty_like(z2, memory_format=torch.channels_last)
        S2 = lib.td_conv1x1_stat_rows(M2, groups, c)
        q2 = torch.empty(groups * S2 * c * 2, **f32)
        native.check(lib.td_conv1x1_dgrad_bnsums(_raw(dz3), _raw(w3), M2, groups, Cout, c, _raw(z2), native.ptr(g2), native.ptr(b2),
                                                 native.ptr(mean2), native.ptr(invstd2), _raw(da2), native.ptr(q2), strm),
                     "td_conv1x1_dgrad_bnsums")
        dw3 = wgrad(dz3, a2, w3, M2, c, Cout, Ho, Wo, 1)
        # bn2 dx from those sums
        dz2 = torch.empty_like(z2, memory_format=torch.channels_last)
        dg2, db2 = torch.empty(c, **f32), torch.empty(c, **f32)
        native.check(lib.td_bn_bwd_from_partials(_raw(da2), _raw(z2), None, BF, native.ptr(g2), native.ptr(b2), native.ptr(mean2),
                                                 native.ptr(invstd2), 1, M2, groups, c, native.ptr(q2), S2, _raw(dz2), None,
                                                 native.ptr(dg2), native.ptr(db2), strm), "td_bn_bwd_from_partials")
        # conv2: MIOpen -- except the weight gradient of the shapes where the hand-written 3x3 kernel is ahead of MIOpen's
        # split-K kernel + its zero-fill / cast passes (profiles/r04/conv3x3_wgrad_bench_v2.txt: 64 -> 64 channels at 48 x 160)
        own_wgrad = conv3x3_wgrad_wins(Nb, Ho, Wo, c, c, stride)
        da1, dw2, _ = torch.ops.aten.convolution_backward(dz2, a1, w2, None, [stride, stride], [1, 1], [1, 1], False, [0, 0], 1,
                                                          [True, not own_wgrad, False])
        if own_wgrad:
            dw2 = conv3x3_wgrad(dz2, a1, w2, pad=1)
        if not da1.is_contiguous(memory_format=torch.channels_last):
            da1 = da1.contiguous(memory_format=torch.channels_last)
        # bn1 statistics pass; its dx rides on conv1's data gradient
        R1 = lib.td_bn_partial_rows(M1, groups, c)
        q1 = torch.empty(groups * R1 * c * 2, **f32)
        native.check(lib.td_bn_bwd_partials(_raw(da1), _raw(z1), None, BF, native.ptr(g1), native.ptr(b1), native.ptr(mean1),
                                            native.ptr(invstd1), 1, M1, groups, c, native.ptr(q1), strm), "td_bn_bwd_partials")
        # identity branch: x itself, or the down-sample convolution + bn
        dwd = dgd = dbd = None
        if wd is not None:
            dzd, _, dgd, dbd = bn_bwd(dres, zd, None, gd, bd, meand, invstdd, False, M2, Cout, False)
            if stride == 1:
                res_in = torch.empty_like(x, memory_format=torch.channels_last)
                native.check(lib.td_conv1x1_dgrad(_raw(dzd), _raw(wd), M2, groups, Cout, Cin, None, _raw(res_in), strm),
                             "td_conv1x1_dgrad")
            else:      # strided 1x1: the gradient lands on every stride-th pixel (MIOpen writes the zeros in between)
                res_in = torch.ops.aten.convolution_backward(dzd, x, wd, None, [stride, stride], [0, 0], [1, 1], False, [0, 0], 1,
                                                             [True, False, False])[0]
                if res_in.dtype != torch.bfloat16 or not res_in.is_contiguous(memory_format=torch.channels_last):
                    res_in = res_in.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
            dwd = wgrad(dzd, x, wd, M2, Cin, Cout, H, W, stride)
        else:
            res_in = dres
        dx = torch.empty_like(x, memory_format=torch.channels_last)
        dz1 = torch.empty_like(z1, memory_format=torch.channels_last)
        dg1, db1 = torch.empty(c, **f32), torch.empty(c, **f32)
        native.check(lib.td_conv1x1_dgrad_bnbwd(_raw(da1), _raw(z1), _raw(w1), M1, groups, c, Cin, native.ptr(q1), R1, native.ptr(g1),
                                                native.ptr(b1), native.ptr(mean1), native.ptr(invstd1), native.ptr(dg1),
                                                native.ptr(db1), _raw(dz1), _raw(res_in), _raw(dx), strm), "td_conv1x1_dgrad_bnbwd")
        dw1 = wgrad(dz1, x, w1, M1, Cin, c, H, W, 1)
        cast = lambda t, like: None if t is None else t.to(like.dtype)
        return (dx, dw1, dw2, dw3, cast(dg1, g1), cast(db1, b1), cast(dg2, g2), cast(db2, b2), cast(dg3, g3), cast(db3, b3),
                dwd, cast(dgd, gd) if gd is not None else None, cast(dbd, bd) if bd is not None else None,
                None, None, None, None, None, None)


def bottleneck_supported(x, w1, w2, w3, g1, wd=None, stride=1):
    bf, cl = torch.bfloat16, torch.channels_last
    ok_w = lambda w: w.dtype == bf and (w.is_contiguous() or w.is_contiguous(memory_format=cl))
    c = w1.shape[0]
    return (x.is_cuda and x.dim() == 4 and x.dtype == bf and x.is_contiguous(memory_format=cl) and ok_w(w1) and ok_w(w3)
            and w2.dtype == bf and w2.is_contiguous(memory_format=cl) and g1.dtype == torch.float32
            and x.shape[1] % 64 == 0 and c % 64 == 0 and c <= 512 and w3.shape[0] % 64 == 0 and stride in (1, 2)
            and tuple(w1.shape[1:]) == (x.shape[1], 1, 1) and tuple(w3.shape[1:]) == (c, 1, 1) and tuple(w2.shape) == (c, c, 3, 3)
            and (wd is None or (ok_w(wd) and tuple(wd.shape) == (w3.shape[0], x.shape[1], 1, 1)))
            and (wd is not None or (stride == 1 and w3.shape[0] == x.shape[1])))


def bottleneck(x, w1, bn1, w2, bn2, w3, bn3, wd=None, bnd=None, groups=1, stride=1):
    """relu(bn3(conv3(relu(bn2(conv2(relu(bn1(conv1(x)))))))) + shortcut(x)) in training mode (reference: resnet.py:66-86), one
    autograd node over the fused kernels; bn* are the BatchNorm modules (affine parameters differentiable, running statistics
    updated in place)."""
    if x.shape[0] % groups:
        raise ValueError("batch %d is not %d stacked passes" % (x.shape[0], groups))
    if not bottleneck_supported(x, w1, w2, w3, bn1.weight, wd, stride):
        raise native.NativeLibraryError("bottleneck needs bf16 channels_last HIP tensors, channels % 64 == 0, planes <= 512")
    return _Bottleneck.apply(x, w1, w2, w3, bn1.weight, bn1.bias, bn2.weight, bn2.bias, bn3.weight, bn3.bias, wd,
                             bnd.weight if bnd is not None else None, bnd.bias if bnd is not None else None,
                             BnState(bn1), BnState(bn2), BnState(bn3), BnState(bnd) if bnd is not None else None, groups, stride)


def batchnorm_act_supported(x, weight):
    return (x.is_cuda and x.dim() == 4 and x.dtype in native.DTYPE_CODES and x.shape[1] % 64 == 0
            and weight is not None and weight.dtype == torch.float32
            and x.is_contiguous(memory_format=torch.channels_last))


def batchnorm_act(x, weight, bias, running_mean, running_var, momentum, eps, residual=None, relu=False, groups=1):
    """Training-mode BatchNorm2d + optional residual add + optional ReLU (reference: resnet.py:30-49, 66-86).
    ``groups`` > 1: the batch is that many stacked passes, each normalised with its own batch statistics."""
    if x.shape[0] % groups:
        raise ValueError("batch %d is not %d stacked passes" % (x.shape[0], groups))
    if not batchnorm_act_supported(x, weight):
        raise native.NativeLibraryError("batchnorm_act needs a channels_last f32/bf16 HIP tensor with C % 64 == 0")
    if residual is not None and (residual.dtype != x.dtype or residual.shape != x.shape
                                 or not residual.is_contiguous(memory_format=torch.channels_last)):
        residual = residual.to(x.dtype).contiguous(memory_format=torch.channels_last)
    return _BatchNormAct.apply(x, weight, bias, running_mean, running_var, residual, momentum, eps, relu, groups)


class _SyncBatchNormAct(torch.autograd.Function):
    """_BatchNormAct with the per-channel statistics summed over a process group between the two kernel stages."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, residual, momentum, eps, relu, groups, process_group):
        import torch.distributed as dist
        lib = native.load()
        N, C, H, W = x.shape
        M = N * H * W
        code = native.DTYPE_CODES[x.dtype]
        packed = torch.empty(groups * C * 2 + 1, device=x.device, dtype=torch.float32)      # sums + row count: ONE collective
        ws = torch.empty(lib.td_bn_workspace_floats(M, groups, C), device=x.device, dtype=torch.float32)
        st = native.stream()
        native.check(lib.td_bn_sync_fwd_sums(_raw(x), code, M, groups, C, native.ptr(packed), native.ptr(ws), st),
                     "td_bn_sync_fwd_sums")
        packed[-1:].fill_(float(M // groups))
        dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=process_group)
        y = torch.empty_like(x, memory_format=torch.channels_last)
        mean = torch.empty(groups * C, device=x.device, dtype=torch.float32)
        invstd = torch.empty(groups * C, device=x.device, dtype=torch.float32)
        native.check(lib.td_bn_sync_fwd_apply(_raw(x), _raw(residual) if residual is not None else None, code, native.ptr(packed),
                                              _raw(packed[-1:]), native.ptr(weight), native.ptr(bias),
                                              native.ptr(running_mean) if running_mean is not None else None,
                                              native.ptr(running_var) if running_var is not None else None,
                                              float(momentum), float(eps), int(relu), M, groups, C, _raw(y), native.ptr(mean),
                                              native.ptr(invstd), native.stream()), "td_bn_sync_fwd_apply")
        ctx.save_for_backward(x, y if (relu and residual is not None) else None, weight, bias, mean, invstd, packed[-1:].clone())
        ctx.relu, ctx.has_res, ctx.groups, ctx.group = bool(relu), residual is not None, groups, process_group
        return y

    @staticmethod
    def backward(ctx, dy):
        import torch.distributed as dist
        lib = native.load()
        x, y, weight, bias, mean, invstd, count = ctx.saved_tensors
        N, C, H, W = x.shape
        M = N * H * W
        G = ctx.groups
        code = native.DTYPE_CODES[x.dtype]
        if dy.dtype != x.dtype or not dy.is_contiguous(memory_format=torch.channels_last):
            dy = dy.to(x.dtype).contiguous(memory_format=torch.channels_last)
        local = torch.empty(G * C * 2, device=x.device, dtype=torch.float32)
        ws = torch.empty(lib.td_bn_workspace_floats(M, G, C), device=x.device, dtype=torch.float32)
        yp = _raw(y) if y is not None else None
        native.check(lib.td_bn_sync_bwd_sums(_raw(dy), _raw(x), yp, code, native.ptr(weight), native.ptr(bias), native.ptr(mean),
                                             native.ptr(invstd), int(ctx.relu), M, G, C, native.ptr(local), native.ptr(ws),
                                             native.stream()), "td_bn_sync_bwd_sums")
        total = local.clone()
        dist.all_reduce(total, op=dist.ReduceOp.SUM, group=ctx.group)
        dx = torch.empty_like(x, memory_format=torch.channels_last)
        dres = torch.empty_like(x, memory_format=torch.channels_last) if (ctx.has_res and ctx.relu) else None
        dgamma = torch.empty(C, device=x.device, dtype=torch.float32)
        dbeta = torch.empty(C, device=x.device, dtype=torch.float32)
        coef = torch.empty(G * C * 3, device=x.device, dtype=torch.float32)
        native.check(lib.td_bn_sync_bwd_dx(_raw(dy), _raw(x), yp, code, native.ptr(local), native.ptr(total), native.ptr(count),
                                           native.ptr(weight), native.ptr(bias), native.ptr(mean), native.ptr(invstd), int(ctx.relu),
                                           M, G, C, _raw(dx), _raw(dres) if dres is not None else None, native.ptr(dgamma),
                                           native.ptr(dbeta), native.ptr(coef), native.stream()), "td_bn_sync_bwd_dx")
        if ctx.has_res and not ctx.relu:
            dres = dy
        return dx, dgamma.to(weight.dtype), dbeta.to(weight.dtype), None, None, dres, None, None, None, None, None


def sync_batchnorm_act(x, weight, bias, running_mean, running_var, momentum, eps, process_group, residual=None, relu=False,
                       groups=1):
    """batchnorm_act with batch statistics over ALL ranks of ``process_group`` (the reference's syncbn=True,
    mono/apis/trainer.py:156-157): one all-reduce of 2*groups*C+1 floats each way, the passes over the activation
    in the same hand-written kernels as the local form."""
    if x.shape[0] % groups:
        raise ValueError("batch %d is not %d stacked passes" % (x.shape[0], groups))
    if not batchnorm_act_supported(x, weight):
        raise native.NativeLibraryError("sync_batchnorm_act needs a channels_last f32/bf16 HIP tensor with C % 64 == 0")
    if residual is not None and (residual.dtype != x.dtype or residual.shape != x.shape
                                 or not residual.is_contiguous(memory_format=torch.channels_last)):
        residual = residual.to(x.dtype).contiguous(memory_format=torch.channels_last)
    return _SyncBatchNormAct.apply(x, weight, bias, running_mean, running_var, residual, momentum, eps, relu, groups,
                                   process_group)


def edge_weights(img_at_scale, a, scale6):
    """Per-pixel, per-term image weights scale_k * exp(-a * mean_c |d_k I|) -> [B,6,h,w] (no gradient)."""
    import ctypes
    lib = native.load()
    img = _f32c(img_at_scale.detach())
    B, _, h, w = img.shape
    out = torch.empty(B, 6, h, w, device=img.device, dtype=torch.float32)
    arr = (ctypes.c_float * 6)(*[float(v) for v in scale6])
    native.check(lib.td_edge_weights(native.ptr(img), B, h, w, float(a), arr, native.ptr(out), native.stream()),
                 "td_edge_weights")
    return out


class _FeatReg(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, Wt):
        lib = native.load()
        B, C, h, w = feat.shape
        nblk = lib.td_featreg_num_blocks(B, h, w, C)
        partial = torch.empty(nblk, device=feat.device, dtype=torch.float32)
        loss = torch.empty(1, device=feat.device, dtype=torch.float32)
        st = native.stream()
        native.check(lib.td_featreg_fwd(_raw(feat), native.DTYPE_CODES[feat.dtype], native.ptr(Wt), B, h, w, C,
                                        native.ptr(partial), st), "td_featreg_fwd")
        native.check(lib.td_sum_scaled(native.ptr(partial), nblk, 1.0, native.ptr(loss), st), "td_sum_scaled")
        ctx.save_for_backward(feat, Wt)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        lib = native.load()
        feat, Wt = ctx.saved_tensors
        B, C, h, w = feat.shape
        grad = torch.empty_like(feat, memory_format=torch.channels_last)
        gs = _f32c(g.reshape(1))
        native.check(lib.td_featreg_bwd(_raw(feat), native.DTYPE_CODES[feat.dtype], native.ptr(Wt), native.ptr(gs),
                                        B, h, w, C, _raw(grad), native.stream()), "td_featreg_bwd")
        return grad, None


def featreg_supported(feat):
    return (feat.is_cuda and feat.dim() == 4 and feat.dtype in native.DTYPE_CODES and feat.shape[1] % 8 == 0
            and feat.shape[2] >= 3 and feat.shape[3] >= 3 and feat.is_contiguous(memory_format=torch.channels_last))


def feature_regularization(feat, img_at_scale, dis, cvt):
    """-dis * smooth1 + cvt * smooth2 of get_feature_regularization_loss
    (mono/model/mono_fm_joint/net.py:309-330) on a channels_last feature map, fused."""
    B, C, h, w = feat.shape
    n = float(B * C)
    counts = [n * h * (w - 1), n * (h - 1) * w, n * h * (w - 2), n * (h - 1) * (w - 1), n * (h - 1) * (w - 1),
              n * (h - 2) * w]
    coef = [-dis, -dis, cvt, cvt, cvt, cvt]
    Wt = edge_weights(img_at_scale, 1.0, [c / k for c, k in zip(coef, counts)])
    return _FeatReg.apply(feat, Wt)


class _ReconSum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y, hole):
        lib = native.load()
        B, _, h, w = x.shape
        n = lib.td_recon_num_tasks(B, h, w)
        partial = torch.empty(n, device=x.device, dtype=torch.float32)
        out = torch.empty(1, device=x.device, dtype=torch.float32)
        st = native.stream()
        native.check(lib.td_recon_fwd(native.ptr(x), native.ptr(y), native.ptr(hole), B, h, w, native.ptr(partial), st),
                     "td_recon_fwd")
        native.check(lib.td_sum_scaled(native.ptr(partial), n, 1.0, native.ptr(out), st), "td_sum_scaled")
        ctx.save_for_backward(x, y, hole)
        return out.reshape(())

    @staticmethod
    def backward(ctx, g):
        lib = native.load()
        x, y, hole = ctx.saved_tensors
        B, _, h, w = x.shape
        dx = torch.empty_like(x)
        gs = _f32c(g.reshape(1))
        native.check(lib.td_recon_bwd(native.ptr(x), native.ptr(y), native.ptr(hole), native.ptr(gs), B, h, w,
                                      native.ptr(dx), native.stream()), "td_recon_bwd")
        return dx, None, None


def masked_reconstruction_sum(pred, target, hole):
    """sum_p hole[p] * (0.85 * mean_c SSIM(pred, target) + 0.15 * mean_c robust_l1) with gradient to
    ``pred`` (mono/model/mono_fm_joint_inpaint/net.py:84-89).  pred/target [B,3,h,w], hole [B,h,w]."""
    if pred.shape[2] < 3 or pred.shape[3] < 3:
        raise native.NativeLibraryError("masked_reconstruction_sum needs h, w >= 3")
    pred = pred.float().contiguous()          # (channels-last bf16 decoder output -> planar f32)
    return _ReconSum.apply(pred, _f32c(target.detach()), _f32c(hole.detach()))


class _ReflPad1(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        lib = native.load()
        N, C, H, W = x.shape
        out = torch.empty((N, C, H + 2, W + 2), device=x.device, dtype=x.dtype, memory_format=torch.channels_last)
        native.check(lib.td_reflpad1_fwd(_raw(x), native.DTYPE_CODES[x.dtype], N, H, W, C, _raw(out), native.stream()),
                     "td_reflpad1_fwd")
        return out

    @staticmethod
    def backward(ctx, g):
        lib = native.load()
        N, C, Ho, Wo = g.shape
        if not g.is_contiguous(memory_format=torch.channels_last):
            g = g.contiguous(memory_format=torch.channels_last)
        gin = torch.empty((N, C, Ho - 2, Wo - 2), device=g.device, dtype=g.dtype, memory_format=torch.channels_last)
        native.check(lib.td_reflpad1_bwd(_raw(g), native.DTYPE_CODES[g.dtype], N, Ho - 2, Wo - 2, C, _raw(gin),
                                         native.stream()), "td_reflpad1_bwd")
        return gin


class _Up2ReflPad1(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        lib = native.load()
        N, C, H, W = x.shape
        out = torch.empty((N, C, 2 * H + 2, 2 * W + 2), device=x.device, dtype=x.dtype, memory_format=torch.channels_last)
        native.check(lib.td_up2_reflpad1_fwd(_raw(x), native.DTYPE_CODES[x.dtype], N, H, W, C, _raw(out),
                                             native.stream()), "td_up2_reflpad1_fwd")
        return out

    @staticmethod
    def backward(ctx, g):
        lib = native.load()
        N, C, Ho, Wo = g.shape
        H, W = (Ho - 2) // 2, (Wo - 2) // 2
        if not g.is_contiguous(memory_format=torch.channels_last):
            g = g.contiguous(memory_format=torch.channels_last)
        gin = torch.empty((N, C, H, W), device=g.device, dtype=g.dtype, memory_format=torch.channels_last)
        native.check(lib.td_up2_reflpad1_bwd(_raw(g), native.DTYPE_CODES[g.dtype], N, H, W, C, _raw(gin),
                                             native.stream()), "td_up2_reflpad1_bwd")
        return gin


def up2_reflpad1_supported(x):
    return (x.is_cuda and x.dim() == 4 and x.dtype in native.DTYPE_CODES and x.shape[1] % 8 == 0
            and x.is_contiguous(memory_format=torch.channels_last))


def up2_reflpad1(x):
    """ReflectionPad2d(1)(interpolate(x, scale_factor=2, mode="nearest")) on a channels_last HIP tensor
    (reference: decoder.py:40-57)."""
    if not up2_reflpad1_supported(x):
        raise native.NativeLibraryError("up2_reflpad1 needs a channels_last f32/bf16 HIP tensor with C % 8 == 0")
    return _Up2ReflPad1.apply(x)


def reflpad1_supported(x):
    return (x.is_cuda and x.dim() == 4 and x.dtype in native.DTYPE_CODES and x.shape[1] % 8 == 0
            and x.shape[2] >= 2 and x.shape[3] >= 2 and x.is_contiguous(memory_format=torch.channels_last))


def reflpad1(x):
    """nn.ReflectionPad2d(1) on a channels_last HIP tensor (reference: layers.py:171-184)."""
    if not reflpad1_supported(x):
        raise native.NativeLibraryError("reflpad1 needs a channels_last f32/bf16 HIP tensor with C % 8 == 0")
    return _ReflPad1.apply(x)


class _FeatureWarpLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, tgt_f, disp, P, invK, min_depth, max_depth, *src_f):
        lib = native.load()
        B, C, h, w = tgt_f.shape
        hs, ws = disp.shape[2], disp.shape[3]
        dev = tgt_f.device
        n_src = len(src_f)
        disp = _f32c(disp)
        P = _f32c(P)
        argmin = torch.empty(B, h, w, device=dev, dtype=torch.uint8)
        nblk = lib.td_featwarp_num_blocks(B, h, w)
        partial = torch.empty(nblk, device=dev, dtype=torch.float32)
        loss = torch.empty(1, device=dev, dtype=torch.float32)
        st = native.stream()
        native.check(lib.td_featwarp_fwd(_raw(tgt_f), native.ptr_array(src_f), n_src, native.DTYPE_CODES[tgt_f.dtype],
                                         native.ptr(disp), native.ptr(P), native.ptr(invK), B, h, w, C, hs, ws,
                                         float(min_depth), float(max_depth), native.ptr(argmin), native.ptr(partial), st),
                     "td_featwarp_fwd")
        inv_count = 1.0 / float(B * h * w)
        native.check(lib.td_sum_scaled(native.ptr(partial), nblk, inv_count, native.ptr(loss), st), "td_sum_scaled")
        ctx.save_for_backward(tgt_f, disp, P, invK, argmin, *src_f)
        ctx.meta = (min_depth, max_depth, inv_count, nblk)
        ctx.mark_non_differentiable(argmin)
        return loss.reshape(()), argmin

    @staticmethod
    def backward(ctx, g, _g_idx):
        lib = native.load()
        tgt_f, disp, P, invK, argmin, *src_f = ctx.saved_tensors
        min_depth, max_depth, inv_count, nblk = ctx.meta
        B, C, h, w = tgt_f.shape
        hs, ws = disp.shape[2], disp.shape[3]
        dev = tgt_f.device
        n_src = len(src_f)
        gs = _f32c(g.reshape(1))
        d_tgt = torch.empty_like(tgt_f, memory_format=torch.channels_last)
        # int32 fixed-point accumulators of the scatter (integer atomics: order-independent, bit-reproducible)
        d_src32 = [torch.zeros((B, h, w, C), device=dev, dtype=torch.int32) for _ in src_f]
        d_up = torch.empty(B, h, w, device=dev, dtype=torch.float32)
        dP_part = torch.empty(nblk, n_src * 12, device=dev, dtype=torch.float32)
        st = native.stream()
        native.check(lib.td_featwarp_bwd(_raw(tgt_f), native.ptr_array(src_f), n_src, native.DTYPE_CODES[tgt_f.dtype],
                                         native.ptr(disp), native.ptr(P), native.ptr(invK), native.ptr(argmin),
                                         native.ptr(gs), inv_count, B, h, w, C, hs, ws, float(min_depth),
                                         float(max_depth), _raw(d_tgt), native.ptr_array(d_src32), native.ptr(d_up),
                                         native.ptr(dP_part), st), "td_featwarp_bwd")
        d_disp = torch.empty_like(disp)
        native.check(lib.td_upsample_adjoint(native.ptr(d_up), B, h, w, hs, ws, native.ptr(d_disp), 0, st),
                     "td_upsample_adjoint")
        dP = torch.empty_like(P)
        native.check(lib.td_reduce_partials(native.ptr(dP_part), n_src, B, nblk // B, native.ptr(dP), st),
                     "td_reduce_partials")
        # accumulators -> the gradient in the feature dtype, logical NCHW view of channels-last memory
        d_src = []
        for acc in d_src32:
            out = torch.empty((B, h, w, C), device=dev, dtype=tgt_f.dtype)
            native.check(lib.td_featwarp_dsrc_finish(native.ptr(acc), native.ptr(gs), inv_count, C, acc.numel(),
                                                     native.DTYPE_CODES[tgt_f.dtype], native.ptr(out), st), "td_featwarp_dsrc_finish")
            d_src.append(out.permute(0, 3, 1, 2))
        d_src = tuple(d_src)
        return (d_tgt, d_disp, dP, None, None, None) + d_src


def featwarp_supported(tgt_f, src_f):
    ok = lambda t: (t.is_cuda and t.dim() == 4 and t.dtype in native.DTYPE_CODES and t.shape[1] % 64 == 0
                    and t.is_contiguous(memory_format=torch.channels_last))
    return ok(tgt_f) and all(ok(s) and s.dtype == tgt_f.dtype and s.shape == tgt_f.shape for s in src_f) \
        and 1 <= len(src_f) <= 2 and tgt_f.shape[2] >= 2 and tgt_f.shape[3] >= 2


def feature_warp_min_loss(tgt_f, src_f, disp, P, invK, min_depth, max_depth):
    """mean over pixels of min_f mean_c sqrt((tgt_f - warp_f(src_f))^2 + 1e-6): generate_features_pred +
    compute_perceptional_loss + min over frames (reference: mono_fm_joint/net.py:196-223,63-65;
    mono_fm_joint_inpaint/net.py:58-70).  P / invK are at the feature resolution.
    Returns (loss, argmin uint8 [B,h,w])."""
    return _FeatureWarpLoss.apply(tgt_f, disp, P, _f32c(invK), min_depth, max_depth, *src_f)


class _RobustL1Map(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, weight):
        lib = native.load()
        B, C, H, W = pred.shape
        out = torch.empty(B, 1, H, W, device=pred.device, dtype=torch.float32)
        native.check(lib.td_l1map_fwd(_raw(pred), native.DTYPE_CODES[pred.dtype], native.strides_array(pred), native.ptr(target),
                                      B, C, H, W, float(weight), native.ptr(out), native.stream()), "td_l1map_fwd")
        ctx.save_for_backward(pred, target)
        ctx.weight = float(weight)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = native.load()
        pred, target = ctx.saved_tensors
        B, C, H, W = pred.shape
        g = _f32c(g)                                         # (an expanded mean-gradient becomes a dense map here)
        dpred = torch.empty_strided(pred.size(), pred.stride(), device=pred.device, dtype=pred.dtype)
        native.check(lib.td_l1map_bwd(_raw(pred), native.DTYPE_CODES[pred.dtype], native.strides_array(pred), native.ptr(target),
                                      native.ptr(g), B, C, H, W, ctx.weight, _raw(dpred), native.stream()), "td_l1map_bwd")
        return dpred, None, None


def robust_l1_map_supported(pred, target):
    return (pred.is_cuda and pred.dim() == 4 and pred.dtype in native.DTYPE_CODES and target.shape == pred.shape
            and pred.shape[1] <= 64 and torch.ops.aten.is_non_overlapping_and_dense(pred))


def robust_l1_map(pred, target, weight=1.0):
    """weight * mean_c sqrt((pred - target)^2 + 1e-6) -> [B,1,H,W]: compute_perceptional_loss between a decoder
    output (any dense layout, f32/bf16, read in place) and an image (reference: compute_auto_res_loss,
    mono_fm_joint_inpaint/net.py:520-527; compute_colorization_loss, :310-323).  Gradient to ``pred`` only."""
    if not robust_l1_map_supported(pred, target):
        raise native.NativeLibraryError("robust_l1_map needs a dense f32/bf16 HIP prediction and a same-shape target")
    return _RobustL1Map.apply(pred, _f32c(target.detach()), weight)


def rgb2lab(rgb, l_cent=50.0, l_norm=50.0, ab_norm=110.0):
    """Normalised Lab of an sRGB image batch [B,3,H,W] (reference: color_conversions.py:106-114); no gradient."""
    lib = native.load()
    rgb = _f32c(rgb.detach())
    B, C, H, W = rgb.shape
    if C != 3:
        raise ValueError("rgb2lab expects 3 channels")
    lab = torch.empty_like(rgb)
    native.check(lib.td_rgb2lab(native.ptr(rgb), B, H, W, float(l_cent), float(l_norm), float(ab_norm), native.ptr(lab),
                                native.stream()), "td_rgb2lab")
    return lab


class _PoseTransforms(torch.autograd.Function):
    @staticmethod
    def forward(ctx, axisangle, translation, K, invert):
        lib = native.load()
        n = len(invert)
        B = axisangle.shape[0] // n
        T = torch.empty(n, B, 4, 4, device=axisangle.device, dtype=torch.float32)
        P = torch.empty(n, B, 3, 4, device=axisangle.device, dtype=torch.float32)
        native.check(lib.td_pose_fwd(native.ptr(axisangle), native.ptr(translation), native.int_array(invert), native.ptr(K), n, B,
                                     native.ptr(T), native.ptr(P), native.stream()), "td_pose_fwd")
        ctx.save_for_backward(axisangle, translation, K)
        ctx.invert = tuple(invert)
        return T, P

    @staticmethod
    def backward(ctx, gT, gP):
        lib = native.load()
        axisangle, translation, K = ctx.saved_tensors
        n = len(ctx.invert)
        B = axisangle.shape[0] // n
        ga, gt = torch.empty_like(axisangle), torch.empty_like(translation)
        gT_c = _f32c(gT) if gT is not None else None      # converted copies stay alive across the launch
        gP_c = _f32c(gP) if gP is not None else None
        native.check(lib.td_pose_bwd(native.ptr(axisangle), native.ptr(translation), native.int_array(ctx.invert), native.ptr(K), n, B,
                                     native.ptr(gT_c) if gT_c is not None else None,
                                     native.ptr(gP_c) if gP_c is not None else None,
                                     native.ptr(ga), native.ptr(gt), native.stream()), "td_pose_bwd")
        return ga, gt, None, None


def pose_transforms(axisangle, translation, K, invert):
    """All frame pairs of a step at once: (T [n,B,4,4], P [n,B,3,4] = (K @ T)[:, :3, :]) from the PoseDecoder's
    pair-major outputs axisangle / translation [n*B,3] (reference: transformation_from_parameters,
    mono_fm_joint/net.py:225-277; Project.forward's K @ T, layers.py:73-75).  ``invert[i]``: frame id < 0."""
    a = _f32c(axisangle.reshape(-1, 3))
    t = _f32c(translation.reshape(-1, 3))
    if a.shape[0] % len(invert):
        raise ValueError("pose rows %d are not %d pairs of equal batch" % (a.shape[0], len(invert)))
    return _PoseTransforms.apply(a, t, _f32c(K), [bool(v) for v in invert])


def color_jitter_expand(frames_u8, aug):
    """uint8 frames [N,3,H,W] + per-image jitter parameters [N,9] -> (color, color_aug) float32 [N,3,H,W]
    (reference: ToTensor + ColorJitter in MonoDataset.preprocess, mono/datasets/mono_dataset.py:83-101)."""
    lib = native.load()
    if frames_u8.dtype != torch.uint8 or frames_u8.dim() != 4 or frames_u8.shape[1] != 3:
        raise ValueError("frames_u8 must be uint8 [N,3,H,W]")
    frames_u8 = frames_u8.contiguous()
    aug = _f32c(aug)
    N, _, H, W = frames_u8.shape
    if aug.shape != (N, 9):
        raise ValueError("aug must be [N,9]")
    color = torch.empty(N, 3, H, W, device=frames_u8.device, dtype=torch.float32)
    color_aug = torch.empty_like(color)
    scratch = torch.empty(N, device=frames_u8.device, dtype=torch.float32)
    native.check(lib.td_color_jitter(native.ptr(frames_u8), native.ptr(aug), N, H, W, native.ptr(scratch), native.ptr(color),
                                     native.ptr(color_aug), native.stream()), "td_color_jitter")
    return color, color_aug


def quantize_fp8(t):
    """Per-tensor current-scaling fp8 quantisation of a dense f32/bf16 HIP tensor: (q float8_e4m3fn with t's shape,
    inv_scale [1] float32 with t ~= q * inv_scale).  Two launches, no host round trip (csrc/td_fp8.hip)."""
    lib = native.load()
    if not (t.is_cuda and t.dtype in native.DTYPE_CODES and t.numel() % 8 == 0 and torch.ops.aten.is_non_overlapping_and_dense(t)):
        raise native.NativeLibraryError("quantize_fp8 needs a dense f32/bf16 HIP tensor with numel % 8 == 0")
    n = t.numel()
    code = native.DTYPE_CODES[t.dtype]
    partials = torch.empty(lib.td_fp8_num_blocks(n), device=t.device, dtype=torch.float32)
    q = torch.empty_strided(t.size(), t.stride(), device=t.device, dtype=torch.uint8)
    inv_scale = torch.empty(1, device=t.device, dtype=torch.float32)
    st = native.stream()
    native.check(lib.td_fp8_amax_partials(_raw(t), code, n, native.ptr(partials), st), "td_fp8_amax_partials")
    native.check(lib.td_fp8_quantize(_raw(t), code, n, native.ptr(partials), _raw(q), native.ptr(inv_scale), st), "td_fp8_quantize")
    return q.view(torch.float8_e4m3fn), inv_scale


class _Conv1x1FP8(torch.autograd.Function):
    """y = conv2d(x, w[, b]) for a 1x1 stride-1 convolution on channels-last activations, forward GEMM in fp8 on the MFMA
    (hipBLASLt through torch._scaled_mm, fp32 accumulation, bf16 out); backward in the activation dtype from the
    saved full-precision operands."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        N, C, H, W = x.shape
        K = weight.shape[0]
        w = weight.to(x.dtype).contiguous(memory_format=torch.channels_last)
        xq, sx = quantize_fp8(x)
        wq, sw = quantize_fp8(w)
        a = xq.permute(0, 2, 3, 1).reshape(N * H * W, C)                 # [M, Cin] row-major view of the NHWC memory
        b = wq.reshape(K, C).t()                                         # [Cin, Cout] column-major
        y = torch._scaled_mm(a, b, scale_a=sx, scale_b=sw, bias=bias.to(x.dtype) if bias is not None else None,
                             out_dtype=x.dtype)
        ctx.save_for_backward(x, w)
        ctx.has_bias = bias is not None
        ctx.wdtype = weight.dtype
        return y.reshape(N, H, W, K).permute(0, 3, 1, 2)                 # logical NCHW, channels-last memory

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        gy = gy.to(x.dtype).contiguous(memory_format=torch.channels_last)
        gx, gw, gb = torch.ops.aten.convolution_backward(gy, x, w, [w.shape[0]] if ctx.has_bias else None, [1, 1], [0, 0], [1, 1],
                                                          False, [0, 0], 1, [True, True, ctx.has_bias])
        return gx, gw.to(ctx.wdtype), (gb.to(ctx.wdtype) if ctx.has_bias else None)


def conv1x1_fp8_supported(x, weight):
    return (x.is_cuda and x.dim() == 4 and x.dtype == torch.bfloat16 and x.is_contiguous(memory_format=torch.channels_last)
            and weight.dim() == 4 and weight.shape[2:] == (1, 1) and weight.shape[1] == x.shape[1]
            and x.shape[1] % 16 == 0 and weight.shape[0] % 16 == 0 and hasattr(torch, "float8_e4m3fn"))


def conv1x1_fp8(x, weight, bias=None):
    """1x1 stride-1 convolution with the forward GEMM on the fp8 MFMA path (BASELINE config 5); see _Conv1x1FP8."""
    if not conv1x1_fp8_supported(x, weight):
        raise native.NativeLibraryError("conv1x1_fp8 needs a channels_last bf16 HIP activation, Cin % 16 == Cout % 16 == 0")
    return _Conv1x1FP8.apply(x, weight, bias)
