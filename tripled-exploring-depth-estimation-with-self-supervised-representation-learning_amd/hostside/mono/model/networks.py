"""Convolutional networks of the TripleD / FeatDepth model families, written once here and
re-exported under the reference's module paths (mono/model/mono_fm_joint/{resnet,depth_encoder,
depth_decoder,pose_encoder,pose_decoder,encoder,decoder,layers}.py).

Parameter and buffer names are the reference's (they are the checkpoint keys, SURVEY.md
section 5): e.g. ``DepthEncoder.encoder.layer1.0.conv1.weight``, ``DepthDecoder.reduce4.conv.weight``,
``DepthDecoder.crp4.0.1_pointwise.conv.weight``, ``Decoder.upconv5.conv.conv.weight``.

The convolutions are the one dense contraction of the step; they run on MFMA through
PyTorch-ROCm (MIOpen / hipBLASLt) under bf16 autocast with channels_last activations.
"""
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F



FUSED_BN_OFF = bool(os.environ.get("TD_NO_FUSED_BN"))


def _ops():
    from tripled_amd import ops
    return ops


def _dense_cl(x):
    """A HIP activation whose channels are a multiple of 8 but which is a strided view (e.g. the channel halves
    the disentangled model cuts out of the scene embedding, mono_fm_joint_inpaint/net.py:488-492) is packed into
    dense channels-last memory so that the hand-written kernels take it; everything else is returned as is."""
    if x.is_cuda and x.dim() == 4 and x.shape[1] % 8 == 0 and x.dtype in (torch.float32, torch.bfloat16) \
            and not x.is_contiguous(memory_format=torch.channels_last) and x.stride(1) == 1:
        return x.contiguous(memory_format=torch.channels_last)
    return x


def _fell_back(site, x, why=""):
    """An activation that lives on a HIP device takes an ATen composition instead of the hand-written kernel
    (tripled_amd.dispatch: counted; raises in strict mode).  The accounting covers the training step: CPU tensors
    and no-grad passes (the fp32 evaluation forward of the validation hooks, which keeps the reference's
    precision and feeds 513-channel fp32 concatenations the 8-channel-vector kernels do not take) are not reported."""
    if x.is_cuda and torch.is_grad_enabled():
        from tripled_amd import dispatch
        dispatch.fallback(site, why or "dtype %s, shape %s, channels_last %s" % (
            x.dtype, tuple(x.shape), x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last)))


class BatchNorm(nn.BatchNorm2d):
    """nn.BatchNorm2d with the per-layer ``num_batches_tracked += 1`` taken out of forward().  The
    counter is not used by the computation when momentum is a number (0.1 here); bumping 252 of them
    one tiny kernel at a time costs ~1 ms per step, so ``bump_batch_counters(model)`` advances all of
    them with a single multi-tensor add once per training forward.  Same parameters/buffers/state_dict
    keys and the same outputs as nn.BatchNorm2d.

    On channels_last HIP activations (C % 64 == 0) training-mode normalisation runs in the hand-written
    kernels of csrc/td_bn.hip; ``fused()`` additionally folds the residual add and ReLU that follow the
    normalisation in every ResNet block into the same pass (TD_NO_FUSED_BN=1 restores ATen/MIOpen)."""

    _pending = 0
    _sync = None            # (process_group,) when the batch statistics span the data-parallel ranks (enable_sync_batchnorm)

    def _hip_ok_params(self):
        return bool(self.training and self.track_running_stats and self.momentum is not None and self.affine) and not FUSED_BN_OFF

    def _hip_ok(self, x):
        if not (x.is_cuda and self._hip_ok_params()):
            return False
        return _ops().batchnorm_act_supported(x, self.weight)

    def forward(self, x):
        if self.training and self.track_running_stats and self.momentum is not None:
            g = _BN_GROUPS[0]
            self._pending += g
            if self._hip_ok(x):
                if self._sync is not None:
                    return _ops().sync_batchnorm_act(x, self.weight, self.bias, self.running_mean, self.running_var,
                                                     self.momentum, self.eps, self._sync[0], groups=g)
                return _ops().batchnorm_act(x, self.weight, self.bias, self.running_mean, self.running_var,
                                            self.momentum, self.eps, groups=g)
            _fell_back("BatchNorm", x)
            if self._sync is not None:
                return _ATenSyncBatchNorm.apply(x, self.weight, self.bias, self.running_mean, self.running_var,
                                                self.momentum, self.eps, g, self._sync[0])
            if g > 1:      # stacked passes without the HIP kernels: one ATen call per pass, in order
                return torch.cat([F.batch_norm(c, self.running_mean, self.running_var, self.weight, self.bias, True,
                                               self.momentum, self.eps) for c in x.chunk(g, 0)], 0)
            return F.batch_norm(x, self.running_mean, self.running_var, self.weight, self.bias, True,
                                self.momentum, self.eps)
        return super().forward(x)

    def fused(self, x, residual=None, relu=True):
        """bn(x) [+ residual] [-> relu] -- one HIP apply pass on channels_last CUDA activations in training."""
        if self._hip_ok(x):
            g = _BN_GROUPS[0]
            self._pending += g
            if self._sync is not None:
                return _ops().sync_batchnorm_act(x, self.weight, self.bias, self.running_mean, self.running_var,
                                                 self.momentum, self.eps, self._sync[0], residual=residual, relu=relu, groups=g)
            return _ops().batchnorm_act(x, self.weight, self.bias, self.running_mean, self.running_var,
                                        self.momentum, self.eps, residual=residual, relu=relu, groups=g)
        return _plain_bn_act(self, x, residual, relu)      # (BatchNorm.forward reports the fallback)


_BN_GROUPS = [1]


class _ATenSyncBatchNorm(torch.autograd.Function):
    """Synchronised batch normalisation from torch ops, device-agnostic (CPU ranks over gloo in the tests; HIP tensors
    in a layout the kernels do not take).  Same staging as tripled_amd.ops.sync_batchnorm_act: local per-channel sums
    -> one all-reduce -> normalise; the parameter gradients stay local (the gradient all-reduce averages them)."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, momentum, eps, groups, process_group):
        import torch.distributed as dist
        n, c = x.shape[0], x.shape[1]
        xs = x.float().reshape(groups, n // groups, c, -1)
        packed = torch.cat([xs.sum((1, 3)).reshape(-1), (xs * xs).sum((1, 3)).reshape(-1),
                            xs.new_full((1,), float(xs.shape[1] * xs.shape[3]))])
        dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=process_group)
        count = packed[-1]
        mean = (packed[:groups * c] / count).reshape(groups, 1, c, 1)
        var = ((packed[groups * c:2 * groups * c] / count).reshape(groups, 1, c, 1) - mean * mean).clamp_min(0)
        invstd = torch.rsqrt(var + eps)
        if running_mean is not None:
            with torch.no_grad():
                for g in range(groups):     # one momentum update per stacked pass, in order
                    running_mean.mul_(1 - momentum).add_(momentum * mean[g].reshape(-1))
                    running_var.mul_(1 - momentum).add_(momentum * (var[g].reshape(-1) * (count / (count - 1))))
        xhat = (xs - mean) * invstd
        y = xhat * weight.float().reshape(1, 1, c, 1) + bias.float().reshape(1, 1, c, 1)
        ctx.save_for_backward(xhat, invstd, weight, count)
        ctx.shape, ctx.dtype, ctx.groups, ctx.group = x.shape, x.dtype, groups, process_group
        return y.reshape(x.shape).to(x.dtype)

    @staticmethod
    def backward(ctx, dy):
        import torch.distributed as dist
        xhat, invstd, weight, count = ctx.saved_tensors
        groups, c = ctx.groups, ctx.shape[1]
        g = dy.float().reshape(xhat.shape)
        sg, sgx = g.sum((1, 3)), (g * xhat).sum((1, 3))               # [groups, c]
        total = torch.cat([sg.reshape(-1), sgx.reshape(-1)])
        dist.all_reduce(total, op=dist.ReduceOp.SUM, group=ctx.group)
        tg = (total[:groups * c] / count).reshape(groups, 1, c, 1)
        tgx = (total[groups * c:] / count).reshape(groups, 1, c, 1)
        dx = (g - tg - xhat * tgx) * (invstd * weight.float().reshape(1, 1, c, 1))
        return (dx.reshape(ctx.shape).to(ctx.dtype), sgx.sum(0).to(weight.dtype), sg.sum(0).to(weight.dtype),
                None, None, None, None, None, None)


def enable_sync_batchnorm(model, process_group=None, force=False):
    """The reference's ``syncbn=True`` (torch.nn.SyncBatchNorm.convert_sync_batchnorm, mono/apis/trainer.py:156-157) for
    this build's BatchNorm modules: batch statistics are summed over the ranks of ``process_group`` (one all-reduce of
    2*C+1 floats per layer and direction) while the passes over the activations stay in the hand-written kernels.
    A one-rank group leaves the layers local unless ``force`` (tests).  Returns the number of layers switched."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        raise RuntimeError("enable_sync_batchnorm needs an initialised process group")
    if dist.get_world_size(process_group) == 1 and not force:
        return 0
    n = 0
    for m in model.modules():
        if isinstance(m, BatchNorm):
            m._sync = (process_group,)
            n += 1

    def foreign(parent):
        # normalisation layers that are not this build's BatchNorm (the optional 1x1 skip heads) take torch's SyncBatchNorm
        for name, child in parent.named_children():
            if isinstance(child, nn.modules.batchnorm._BatchNorm) and not isinstance(child, (BatchNorm, nn.SyncBatchNorm)):
                setattr(parent, name, nn.SyncBatchNorm.convert_sync_batchnorm(child, process_group))
            else:
                foreign(child)
    foreign(model)
    return n


class bn_groups:
    """``with bn_groups(G): net(torch.cat(passes, 0))`` -- every BatchNorm inside treats the batch as G stacked
    passes with separate batch statistics (and G running-statistics updates, G counter increments), so one
    launch sequence over the stacked batch equals the reference's G separate forward passes of that network
    (mono_fm_joint/net.py:172-178 pose pairs, :221 source-frame features)."""

    def __init__(self, groups):
        self.groups = int(groups)

    def __enter__(self):
        self.prev = _BN_GROUPS[0]
        _BN_GROUPS[0] = self.groups

    def __exit__(self, *exc):
        _BN_GROUPS[0] = self.prev
        return False


def bn_groups_supported(model):
    """Stacked passes need every normalisation layer to be this module's BatchNorm (not SyncBatchNorm etc.)."""
    return not any(isinstance(m, nn.modules.batchnorm._BatchNorm) and not isinstance(m, BatchNorm)
                   for m in model.modules())


def _plain_bn_act(bn, x, residual, relu):
    y = bn(x)
    if residual is not None:
        y = y + residual
    return F.relu(y, inplace=True) if relu else y


def bn_act(bn, x, residual=None, relu=True):
    """bn -> [+ residual] -> [relu] for any normalisation module (SyncBatchNorm after conversion included)."""
    if isinstance(bn, BatchNorm):
        return bn.fused(x, residual, relu)
    return _plain_bn_act(bn, x, residual, relu)


FUSED_1X1_OFF = bool(os.environ.get("TD_NO_MFMA_1X1"))
FUSED_BLOCK_OFF = bool(os.environ.get("TD_NO_FUSED_BLOCK"))      # bottlenecks as per-layer autograd nodes (round 3's path)


def conv_bn_act(conv, bn, x, residual=None, relu=True):
    """conv -> bn -> [+ residual] -> [relu].  A 1x1 convolution (stride 1 or the stride-2 down-sample branch) in front of this
    build's BatchNorm on bf16 channels_last HIP activations runs as the hand-written MFMA GEMM whose epilogue forms the batch
    statistics (tripled_amd.ops.conv1x1_bn_act: csrc/td_conv1x1.hip); everything else takes conv (MIOpen) + bn_act."""
    if (not FUSED_1X1_OFF and conv.kernel_size == (1, 1) and conv.padding == (0, 0) and conv.groups == 1 and conv.bias is None
            and conv.stride[0] == conv.stride[1] and isinstance(bn, BatchNorm) and bn._sync is None and not _FP8_1X1[0]
            and x.is_cuda and torch.is_autocast_enabled() and torch.get_autocast_gpu_dtype() == torch.bfloat16
            and bn._hip_ok_params()):
        xb = _dense_cl(x if x.dtype == torch.bfloat16 else x.to(torch.bfloat16))
        w = conv.weight if conv.weight.dtype == torch.bfloat16 else conv.weight.to(torch.bfloat16)
        if _ops().conv1x1_bn_act_supported(xb, w, bn.weight, conv.stride[0]):
            g = _BN_GROUPS[0]
            bn._pending += g
            return _ops().conv1x1_bn_act(xb, w, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.momentum, bn.eps,
                                         residual=residual, relu=relu, groups=g, stride=conv.stride[0])
    return bn_act(bn, conv(x), residual, relu)


def bump_batch_counters(model):
    """num_batches_tracked += (forward calls since the last bump) for every BatchNorm of ``model``,
    as one multi-tensor add."""
    counters, amounts = [], []
    for m in model.modules():
        if isinstance(m, BatchNorm) and m._pending:
            counters.append(m.num_batches_tracked)
            amounts.append(m._pending)
            m._pending = 0
    if counters:
        torch._foreach_add_(counters, amounts)


def install_counter_hook(model):
    """Advance the BatchNorm counters once per training forward of ``model``."""
    def hook(module, args, output):
        if module.training:
            bump_batch_counters(module)
    model.register_forward_hook(hook)


# (block kind, blocks per stage)  -- reference: mono/model/mono_fm_joint/resnet.py:147-187
RESNET_SPECS = {18: ("basic", (2, 2, 2, 2)), 34: ("basic", (3, 4, 6, 3)),
                50: ("bottleneck", (3, 4, 6, 3)), 101: ("bottleneck", (3, 4, 23, 3))}


_FP8_1X1 = [False]


def set_fp8_conv1x1(on):
    """cfg.fp8_conv1x1 / bench.py --fp8: the forward GEMM of every eligible 1x1 stride-1 convolution (bf16 channels-last
    activation, Cin % 16 == Cout % 16 == 0) runs on the fp8 MFMA path (tripled_amd.ops.conv1x1_fp8); weights, gradients
    and the backward stay as they are.  Returns the previous setting."""
    prev = _FP8_1X1[0]
    _FP8_1X1[0] = bool(on)
    return prev


class Conv2d(nn.Conv2d):
    """nn.Conv2d (same parameters and state_dict keys) whose 1x1 stride-1 case can take the fp8 path."""

    def forward(self, x):
        if _FP8_1X1[0] and self.kernel_size == (1, 1) and self.stride == (1, 1) and self.padding == (0, 0) and self.groups == 1 \
                and x.is_cuda and torch.is_autocast_enabled():
            xb = x if x.dtype == torch.bfloat16 else x.to(torch.bfloat16)
            xb = _dense_cl(xb)
            if _ops().conv1x1_fp8_supported(xb, self.weight):
                return _ops().conv1x1_fp8(xb, self.weight, self.bias)
            _fell_back("Conv2d.fp8_1x1", x)
        return super().forward(x)


def _conv(cin, cout, k, stride=1, pad=0, bias=False):
    return Conv2d(cin, cout, kernel_size=k, stride=stride, padding=pad, bias=bias)


class BasicBlock(nn.Module):
    """Two 3x3 convs + identity (reference: resnet.py:18-49)."""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, use_residual=True):
        super().__init__()
        self.conv1 = _conv(inplanes, planes, 3, stride, 1)
        self.bn1 = BatchNorm(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = _conv(planes, planes, 3, 1, 1)
        self.bn2 = BatchNorm(planes)
        self.downsample = downsample
        self.stride = stride
        self.use_residual = use_residual

    def forward(self, x):
        y = bn_act(self.bn1, self.conv1(x))
        shortcut = None
        if self.use_residual:
            ds = self.downsample
            if ds is None:
                shortcut = x
            elif isinstance(ds, nn.Sequential) and len(ds) == 2 and isinstance(ds[0], nn.Conv2d):
                shortcut = conv_bn_act(ds[0], ds[1], x, relu=False)      # 1x1 stride-2 branch: MFMA GEMM + epilogue statistics
            else:
                shortcut = ds(x)
        return bn_act(self.bn2, self.conv2(y), shortcut)


class Bottleneck(nn.Module):
    """1x1 -> 3x3 (strided) -> 1x1 x4 (reference: resnet.py:52-86)."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = _conv(inplanes, planes, 1)
        self.bn1 = BatchNorm(planes)
        self.conv2 = _conv(planes, planes, 3, stride, 1)
        self.bn2 = BatchNorm(planes)
        self.conv3 = _conv(planes, planes * 4, 1)
        self.bn3 = BatchNorm(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def _shortcut(self, x):
        ds = self.downsample
        if ds is None:
            return x
        if isinstance(ds, nn.Sequential) and len(ds) == 2 and isinstance(ds[0], nn.Conv2d):
            return conv_bn_act(ds[0], ds[1], x, relu=False)
        return ds(x)

    def _fused(self, x):
        """The block as one autograd node over the fused kernels (tripled_amd.ops.bottleneck: BatchNorm passes folded into the
        1x1 GEMMs' operand staging / epilogues, both data gradients hand-written), or None when the layout is not one it takes."""
        if FUSED_BLOCK_OFF or FUSED_1X1_OFF or _FP8_1X1[0] or not (self.training and x.is_cuda and torch.is_grad_enabled()):
            return None
        if not (torch.is_autocast_enabled() and torch.get_autocast_gpu_dtype() == torch.bfloat16):
            return None
        ds = self.downsample
        bns = [self.bn1, self.bn2, self.bn3]
        wd = bnd = None
        if ds is not None:
            if not (isinstance(ds, nn.Sequential) and len(ds) == 2 and isinstance(ds[0], nn.Conv2d) and ds[0].kernel_size == (1, 1)
                    and ds[0].bias is None and ds[0].stride == (self.stride, self.stride)):
                return None
            wd, bnd = ds[0].weight, ds[1]
            bns.append(bnd)
        if not all(isinstance(b, BatchNorm) and b._sync is None and b._hip_ok_params() for b in bns):
            return None
        if any(cv.bias is not None or cv.groups != 1 for cv in (self.conv1, self.conv2, self.conv3)):
            return None
        bf = lambda w: w if w.dtype == torch.bfloat16 else w.to(torch.bfloat16)
        xb = _dense_cl(x if x.dtype == torch.bfloat16 else x.to(torch.bfloat16))
        w1, w2, w3 = bf(self.conv1.weight), bf(self.conv2.weight), bf(self.conv3.weight)
        wd = bf(wd) if wd is not None else None
        if not _ops().bottleneck_supported(xb, w1, w2, w3, self.bn1.weight, wd, self.stride):
            return None
        g = _BN_GROUPS[0]
        for b in bns:
            b._pending += g
        return _ops().bottleneck(xb, w1, self.bn1, w2, self.bn2, w3, self.bn3, wd, bnd, groups=g, stride=self.stride)

    def forward(self, x):
        y = self._fused(x)
        if y is not None:
            return y
        y = conv_bn_act(self.conv1, self.bn1, x)
        y = bn_act(self.bn2, self.conv2(y))
        return conv_bn_act(self.conv3, self.bn3, y, self._shortcut(x))


class ResNet(nn.Module):
    """torchvision-layout ResNet trunk (reference: resnet.py:89-144).  ``avgpool``/``fc`` exist
    only so that ImageNet / reference checkpoints load with matching keys; the encoders never
    call them."""

    def __init__(self, block, layers, num_classes=1000, in_channels=3):
        super().__init__()
        self.inplanes = 64
        self.conv1 = _conv(in_channels, 64, 7, 2, 3)
        self.bn1 = BatchNorm(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2)
        self.layer4 = self._make_layer(block, 512, layers[3], stride=2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512 * block.expansion, num_classes)
        init_resnet_weights(self)

    def _make_layer(self, block, planes, blocks, stride=1):
        out_ch = planes * block.expansion
        shortcut = None
        if stride != 1 or self.inplanes != out_ch:
            shortcut = nn.Sequential(_conv(self.inplanes, out_ch, 1, stride), BatchNorm(out_ch))
        stage = [block(self.inplanes, planes, stride, shortcut)]
        self.inplanes = out_ch
        stage += [block(out_ch, planes) for _ in range(1, blocks)]
        return nn.Sequential(*stage)

    def stem(self, x):
        return bn_act(self.bn1, self.conv1(x))

    def pool(self, x):
        """self.maxpool (3x3, stride 2, padding 1): hand-written kernel on channels-last HIP activations."""
        if x.is_cuda and x.shape[1] % 8 == 0 and x.dtype in (torch.float32, torch.bfloat16) \
                and x.is_contiguous(memory_format=torch.channels_last) and not os.environ.get("TD_NO_HIP_STEM_POOL"):
            return _ops().maxpool3s2(x)
        _fell_back("ResNet.maxpool", x)
        return self.maxpool(x)

    def pyramid(self, x, extra=None):
        """The five feature maps every encoder here exposes (strides 2, 4, 8, 16, 32);
        ``extra`` optionally adds a conditioning tensor to each level (Encoder.forward)."""
        if extra is None:
            f0 = self.stem(x)
            f1 = self.layer1(self.pool(f0))
            f2 = self.layer2(f1)
            f3 = self.layer3(f2)
            return [f0, f1, f2, f3, self.layer4(f3)]
        f0 = self.stem(x) + extra[0]
        f1 = self.layer1(self.pool(f0)) + extra[1]
        f2 = self.layer2(f1) + extra[2]
        f3 = self.layer3(f2) + extra[3]
        f4 = self.layer4(f3) + extra[4]
        return [f0, f1, f2, f3, f4]

    def forward(self, x):
        return self.pyramid(x)[-1]


def init_resnet_weights(net):
    for m in net.modules():
        if isinstance(m, nn.Conv2d):
            nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
        elif isinstance(m, nn.BatchNorm2d):
            nn.init.constant_(m.weight, 1)
            nn.init.constant_(m.bias, 0)


def build_resnet(num_layers, in_channels=3):
    if num_layers not in RESNET_SPECS:
        raise ValueError("{} is not a valid number of resnet layers".format(num_layers))
    kind, layers = RESNET_SPECS[num_layers]
    return ResNet(BasicBlock if kind == "basic" else Bottleneck, list(layers), in_channels=in_channels)


def resnet18(pretrained_path=None):
    return _maybe_load(build_resnet(18), pretrained_path)


def resnet34(pretrained_path=None, **kw):
    return _maybe_load(build_resnet(34), pretrained_path, "resnet34.pth")


def resnet50(pretrained_path=None, **kw):
    return _maybe_load(build_resnet(50), pretrained_path, "resnet50.pth")


def resnet101(pretrained_path=None, **kw):
    return _maybe_load(build_resnet(101), pretrained_path, "resnet101.pth")


def _maybe_load(net, path, fname=None):
    if path is not None:
        import os
        full = os.path.join(path, fname) if fname else path
        net.load_state_dict(torch.load(full, map_location="cpu", weights_only=True))
    return net


def enc_channels(num_layers):
    ch = np.array([64, 64, 128, 256, 512])
    if num_layers > 34:
        ch[1:] *= 4
    return ch


def _load_encoder_weights(net, path, n_images=1):
    state = torch.load(path, map_location="cpu", weights_only=True)
    if n_images > 1:   # reference: pose_encoder.py:45-48
        state["conv1.weight"] = torch.cat([state["conv1.weight"]] * n_images, 1) / n_images
    net.load_state_dict(state, strict=False)


class DepthEncoder(nn.Module):
    """ImageNet-normalised ResNet pyramid (reference: depth_encoder.py:8-43)."""

    def __init__(self, num_layers, pretrained_path=None):
        super().__init__()
        self.num_ch_enc = enc_channels(num_layers)
        self.encoder = build_resnet(num_layers)
        if pretrained_path is not None:
            _load_encoder_weights(self.encoder, pretrained_path)

    def forward(self, input_image):
        self.features = self.encoder.pyramid((input_image - 0.45) / 0.225)
        return self.features


class PoseEncoder(nn.Module):
    """ResNet over the channel-concatenated frame pair (reference: pose_encoder.py:52-92)."""

    def __init__(self, num_layers, pretrained_path=None, num_input_images=2):
        super().__init__()
        self.num_ch_enc = enc_channels(num_layers)
        self.encoder = build_resnet(num_layers, in_channels=3 * num_input_images)
        if pretrained_path is not None:
            _load_encoder_weights(self.encoder, pretrained_path, num_input_images)

    def forward(self, input_image):
        self.features = self.encoder.pyramid((input_image - 0.45) / 0.225)
        return self.features


class Encoder(nn.Module):
    """Feature extractor of the auto-encoder branch: un-normalised input, optional additive
    conditioning per level (reference: encoder.py:8-46)."""

    def __init__(self, num_layers, pretrained_path=None):
        super().__init__()
        self.num_ch_enc = enc_channels(num_layers)
        self.encoder = build_resnet(num_layers)
        if pretrained_path is not None:
            _load_encoder_weights(self.encoder, pretrained_path)
        self.features = []

    def forward(self, input_image, input_features=None):
        self.features = self.encoder.pyramid(input_image, input_features)
        return self.features

    def stem_only(self, input_image):
        """features[0] alone (what generate_features_pred consumes)."""
        return self.encoder.stem(input_image)


class PoseDecoder(nn.Module):
    """1x1 squeeze, two 3x3, 1x1 -> 6; spatial mean * 0.01 (reference: pose_decoder.py:5-26)."""

    def __init__(self, num_ch_enc, stride=1):
        super().__init__()
        self.reduce = Conv2d(int(num_ch_enc[-1]), 256, 1)
        self.conv1 = nn.Conv2d(256, 256, 3, stride, 1)
        self.conv2 = nn.Conv2d(256, 256, 3, stride, 1)
        self.conv3 = nn.Conv2d(256, 6, 1)
        self.relu = nn.ReLU()

    def forward(self, input_features):
        x = self.relu(self.reduce(input_features[-1]))
        x = self.relu(self.conv1(x))
        x = self.relu(self.conv2(x))
        x = self.conv3(x).float()
        x = 0.01 * x.mean(3).mean(2).view(-1, 1, 1, 6)
        return x[..., :3], x[..., 3:]


# ---------------------------------------------------------------------------
# decoder building blocks (reference: layers.py:110-215)

def upsample(x):
    """x2 nearest-neighbour (reference: layers.py upsample).  CUDA autocast lists upsample_nearest2d as an
    fp32 op: a bf16 activation would come back as float32 and drag the following reflection pad and the
    cast in front of the next convolution through twice the bytes.  Nearest sampling is exact in any dtype,
    so it runs outside autocast in the activation's own dtype."""
    if x.is_cuda and torch.is_autocast_enabled():
        with torch.autocast("cuda", enabled=False):
            return F.interpolate(x, scale_factor=2, mode="nearest")
    return F.interpolate(x, scale_factor=2, mode="nearest")


class Conv1x1(nn.Module):
    def __init__(self, in_channels, out_channels, bias=False):
        super().__init__()
        self.conv = Conv2d(int(in_channels), int(out_channels), kernel_size=1, stride=1, bias=bias)

    def forward(self, x):
        return self.conv(x)


def _round8(n):
    return (n + 7) // 8 * 8


class _ConvImmediate(torch.autograd.Function):
    """conv2d whose forward AND backward run with MIOpen's find mode off (immediate-mode solver choice).

    ROCm 7.2: the tuning search of ConvAsmImplicitGemmGTCDynamicBwdXdlopsNHWC for the backward-data problem
    `convbfp16 -n 4 -c 16 -H 322 -W 1026 -k 16 -y 3 -x 3` (NHWC) -- iconv1 of the image decoders at the 320x1024
    configuration -- ends in a GPU memory access fault (MIOPEN_LOG_LEVEL=5 trace: the fault follows
    "Starting search: ConvAsmImplicitGemmGTCDynamicBwdXdlopsNHWC" for exactly this problem); with the search skipped the
    same layer runs.  The flag is read when the op executes, so the backward needs the guard too."""

    @staticmethod
    def forward(ctx, x, w, b):
        with torch.backends.cudnn.flags(enabled=True, benchmark=False):
            y = F.conv2d(x, w, b)
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        with torch.backends.cudnn.flags(enabled=True, benchmark=False):
            gx, gw, gb = torch.ops.aten.convolution_backward(gy, x, w, [w.shape[0]] if ctx.has_bias else None, [1, 1], [0, 0], [1, 1],
                                                              False, [0, 0], 1, [True, True, ctx.has_bias])
        return gx, gw, (gb if ctx.has_bias else None)


class _KeepChannels(torch.autograd.Function):
    """y[:, :c] of a channel-padded convolution output.  Forward: the view itself (nothing is copied).  Backward: ONE
    zero-padding pass; autograd's slice backward is a zero-fill of the full tensor plus a strided copy into it
    (two passes over up to 12 x 8 x 192 x 640 per head, twelve heads per step)."""

    @staticmethod
    def forward(ctx, y, c):
        ctx.pad = y.shape[1] - c
        return y.narrow(1, 0, c)

    @staticmethod
    def backward(ctx, g):
        return F.pad(g, (0, 0, 0, 0, 0, ctx.pad)), None


FUSED_BIAS_ACT_OFF = bool(os.environ.get("TD_NO_FUSED_BIAS_ACT"))
FUSED_CRP_OFF = bool(os.environ.get("TD_NO_FUSED_CRP"))        # CRP blocks as per-op autograd nodes (round 3's path)


def _activate(y, act):
    if act == "elu":
        return F.elu(y, inplace=True)
    if act == "leaky_relu":
        return F.leaky_relu(y)
    return y


def _conv2d_guarded(x, w, b, act=None):
    """act(F.conv2d(x, w, b)), act in (None, "elu", "leaky_relu").  Under bf16 autocast on channels-last HIP activations the bias,
    the activation and their adjoints (incl. the bias gradient) are fused around MIOpen's convolution
    (tripled_amd.ops.conv_bias_act).  The narrow (<= 16 channels) layers on inputs wider than 1024 columns keep _ConvImmediate."""
    if x.is_cuda and x.shape[1] <= 16 and w.shape[0] <= 16 and x.shape[3] > 1024 and torch.backends.cudnn.benchmark:
        if torch.is_autocast_enabled():
            dt = torch.get_autocast_dtype("cuda")
            x, w, b = x.to(dt), w.to(dt), (b.to(dt) if b is not None else None)
        return _activate(_ConvImmediate.apply(x, w, b), act)
    if (act is not None or b is not None) and x.is_cuda and not FUSED_BIAS_ACT_OFF and torch.is_grad_enabled() \
            and torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.bfloat16:
        xb = x if x.dtype == torch.bfloat16 else x.to(torch.bfloat16)
        wb = w if w.dtype == torch.bfloat16 else w.to(torch.bfloat16)
        if _ops().conv_bias_act_supported(xb, wb, b, act):
            return _ops().conv_bias_act(xb, wb, b, act)
    return _activate(F.conv2d(x, w, b), act)


class Conv3x3(nn.Module):
    """3x3 conv after a 1-pixel reflection (or zero) pad.

    On HIP devices channel counts that are not multiples of 8 are zero-padded on the fly (input channels:
    513 -> 520 for the CRP decoder's concat convs; output channels: 1 / 2 / 3 -> 8 for the disparity and
    image heads).  The parameters keep the reference's shapes (checkpoint keys); zero channels contribute
    nothing and the extra outputs are sliced away, but MIOpen's vectorised implicit-GEMM kernels then apply
    (513->256 @48x160: 976 us -> ~420 us forward, 2.2 ms -> ~0.9 ms backward on MI355X)."""

    def __init__(self, in_channels, out_channels, use_refl=True):
        super().__init__()
        self.use_refl = use_refl
        self.pad = nn.ReflectionPad2d(1) if use_refl else nn.ZeroPad2d(1)
        self.conv = nn.Conv2d(int(in_channels), int(out_channels), 3)

    def _pad_input(self, x):
        x = _dense_cl(x)
        if self.use_refl and x.shape[1] % 8 == 0 and x.dtype in (torch.float32, torch.bfloat16) \
                and x.is_contiguous(memory_format=torch.channels_last) and x.shape[2] >= 2 and x.shape[3] >= 2:
            from tripled_amd import ops          # gather-form pad / adjoint (ATen's backward uses atomics)
            return ops.reflpad1(x)
        if self.use_refl:
            _fell_back("Conv3x3.pad", x)
        return self.pad(x)

    def forward(self, x, act=None):
        """act(conv(pad(x))); ``act`` (None / "elu" / "leaky_relu") is the activation that follows this convolution in the
        decoders -- handed in so that it can be fused with the bias (``_conv2d_guarded``)."""
        # bf16 (autocast) only: MIOpen's f32 NHWC backward-data path crashes for 8-channel outputs
        # (conv -n 2 -c 16 -H 98 -W 162 -k 8 -y 3 -x 3, ROCm 7.2), so f32 models keep the plain convolution
        low_precision = x.is_cuda and (x.dtype == torch.bfloat16 or torch.is_autocast_enabled())
        if not low_precision or os.environ.get("TD_NO_CHANNEL_PAD"):
            return _activate(self.conv(self._pad_input(x) if x.is_cuda else self.pad(x)), act)
        w, b = self.conv.weight, self.conv.bias
        cout, cin = w.shape[0], w.shape[1]
        cin_p, cout_p = _round8(cin), _round8(cout)
        if cin_p == cin and cout_p == cout and x.shape[1] == cin:
            return _conv2d_guarded(self._pad_input(x), w, b, act)
        if x.shape[1] == cin and cin_p != cin:               # caller did not pre-pad the channels
            x = torch.cat((x, x.new_zeros(x.shape[0], cin_p - cin, x.shape[2], x.shape[3])), 1)
        if x.shape[1] != cin_p:
            raise ValueError("Conv3x3 expects %d (or pre-padded %d) input channels, got %d" % (cin, cin_p, x.shape[1]))
        if not x.is_contiguous(memory_format=torch.channels_last):
            x = x.contiguous(memory_format=torch.channels_last)
        w = F.pad(w, (0, 0, 0, 0, 0, cin_p - cin, 0, cout_p - cout))
        if b is not None and cout_p != cout:
            b = F.pad(b, (0, cout_p - cout))
        if cout_p != cout:
            return _activate(_KeepChannels.apply(_conv2d_guarded(self._pad_input(x), w, b), cout), act)
        return _conv2d_guarded(self._pad_input(x), w, b, act)

    def forward_up(self, x, act=None):
        """act(self(upsample(x))): x2 nearest + reflection pad in one HIP pass when the layout allows it (output channels
        padded to a multiple of 8 like forward(), e.g. the 1-channel disparity heads)."""
        w, b = self.conv.weight, self.conv.bias
        cout, cin = w.shape[0], w.shape[1]
        x = _dense_cl(x)
        low_precision = x.is_cuda and (x.dtype == torch.bfloat16 or torch.is_autocast_enabled())
        if (self.use_refl and x.is_cuda and x.shape[1] == cin and cin % 8 == 0 and (cout % 8 == 0 or low_precision)
                and x.dtype in (torch.float32, torch.bfloat16) and x.is_contiguous(memory_format=torch.channels_last)
                and not os.environ.get("TD_NO_FUSED_UPSAMPLE")):
            cout_p = _round8(cout)
            if cout_p != cout:
                w = F.pad(w, (0, 0, 0, 0, 0, 0, 0, cout_p - cout))
                b = F.pad(b, (0, cout_p - cout)) if b is not None else None
            if cout_p != cout:
                return _activate(_KeepChannels.apply(_conv2d_guarded(_ops().up2_reflpad1(x), w, b), cout), act)
            return _conv2d_guarded(_ops().up2_reflpad1(x), w, b, act)
        _fell_back("Conv3x3.forward_up", x)
        return self.forward(upsample(x), act)


class Conv5x5(nn.Module):
    def __init__(self, in_channels, out_channels, use_refl=True):
        super().__init__()
        self.pad = nn.ReflectionPad2d(2) if use_refl else nn.ZeroPad2d(2)
        self.conv = nn.Conv2d(int(in_channels), int(out_channels), 5)

    def forward(self, x):
        return self.conv(self.pad(x))


class ConvBlock(nn.Module):
    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.conv = Conv3x3(in_channels, out_channels)
        self.nonlin = nn.ELU(inplace=True)

    def forward(self, x):
        return self.conv(x, act="elu")          # self.nonlin = ELU(alpha 1), fused with the convolution's bias

    def forward_up(self, x):
        """self(upsample(x)) without materialising the up-sampled tensor (Conv3x3.forward_up)."""
        return self.conv.forward_up(x, act="elu")


class CRPBlock(nn.Module):
    """Chained residual pooling: n x (5x5 max-pool -> 1x1 conv), running sum
    (reference: layers.py:200-215).  Sub-module names '<i>_pointwise' are checkpoint keys."""

    def __init__(self, in_planes, out_planes, n_stages):
        super().__init__()
        for i in range(n_stages):
            setattr(self, "{}_{}".format(i + 1, "pointwise"),
                    Conv1x1(in_planes if i == 0 else out_planes, out_planes, False))
        self.stride = 1
        self.n_stages = n_stages
        self.maxpool = nn.MaxPool2d(kernel_size=5, stride=1, padding=2)

    def _pool(self, t):
        # channels-last HIP activations go to the hand-written 5x5 kernel (1-byte arg-max,
        # gather-form backward); anything else (CPU tests, NCHW) to ATen's max_pool2d
        if t.is_cuda and t.is_contiguous(memory_format=torch.channels_last) and t.shape[1] % 8 == 0 \
                and t.dtype in (torch.float32, torch.bfloat16):
            from tripled_amd import ops
            return ops.maxpool5(t)
        _fell_back("CRPBlock.maxpool", t)
        return self.maxpool(t)

    def _fused(self, x):
        """The whole block as one autograd node over the hand-written pool / GEMM kernels (tripled_amd.ops.crp_block), or None."""
        if FUSED_CRP_OFF or _FP8_1X1[0] or not (self.training and x.is_cuda and torch.is_grad_enabled()):
            return None
        if not (torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.bfloat16):
            return None
        convs = [getattr(self, "{}_{}".format(i + 1, "pointwise")).conv for i in range(self.n_stages)]
        if any(c.bias is not None for c in convs):
            return None
        xb = _dense_cl(x if x.dtype == torch.bfloat16 else x.to(torch.bfloat16))
        ws = [c.weight if c.weight.dtype == torch.bfloat16 else c.weight.to(torch.bfloat16) for c in convs]
        if not _ops().crp_supported(xb, ws):
            return None
        return _ops().crp_block(xb, ws)

    def forward(self, x):
        y = self._fused(x)
        if y is not None:
            return y
        top = x
        for i in range(self.n_stages):
            top = getattr(self, "{}_{}".format(i + 1, "pointwise"))(self._pool(top))
            x = top + x
        return x


class ReflPad1(nn.ReflectionPad2d):
    """nn.ReflectionPad2d(1) that takes the gather-form HIP kernels on channels-last activations."""

    def __init__(self):
        super().__init__(1)

    def forward(self, x):
        if x.is_cuda and x.shape[1] % 8 == 0 and x.dtype in (torch.float32, torch.bfloat16) \
                and x.is_contiguous(memory_format=torch.channels_last) and x.shape[2] >= 2 and x.shape[3] >= 2:
            return _ops().reflpad1(x)
        _fell_back("ReflPad1", x)
        return super().forward(x)


def upshuffle(in_planes, upscale_factor):
    """Sub-pixel x2 up-sampler (reference: layers.py:130-141), ICNR-style initialisation."""
    block = nn.Sequential(ReflPad1(),
                          nn.Conv2d(in_planes, in_planes * upscale_factor ** 2, 3, 1, 0),
                          nn.PixelShuffle(upscale_factor), nn.ELU(inplace=True))
    r2 = upscale_factor ** 2
    w = block[1].weight
    sub = torch.empty(w.shape[0] // r2, w.shape[1], w.shape[2], w.shape[3])
    nn.init.kaiming_normal_(sub)
    with torch.no_grad():
        w.copy_(sub.repeat_interleave(r2, dim=0))
    return block


class DepthDecoder(nn.Module):
    """CRP disparity decoder (reference: depth_decoder.py:8-115): four stages from 1/32 to 1/4
    resolution, each up-sampled x2 before its sigmoid disparity head, so ("disp", 0, s) has
    1/2^(s+1) of the input resolution."""

    def __init__(self, num_ch_enc, use_shuffle=False):
        super().__init__()
        width, stages = 256, 4
        self.do = nn.Dropout(p=0.5)
        self.use_shuffle = use_shuffle
        if use_shuffle:
            for i in (1, 2, 3, 4):
                setattr(self, "up%d" % i, upshuffle(width, 2))
        self.reduce4 = Conv1x1(num_ch_enc[4], 512, bias=False)
        self.reduce3 = Conv1x1(num_ch_enc[3], width, bias=False)
        self.reduce2 = Conv1x1(num_ch_enc[2], width, bias=False)
        self.reduce1 = Conv1x1(num_ch_enc[1], width, bias=False)
        self.iconv4 = Conv3x3(512, width)
        self.iconv3 = Conv3x3(width * 2 + 1, width)
        self.iconv2 = Conv3x3(width * 2 + 1, width)
        self.iconv1 = Conv3x3(width * 2 + 1, width)
        for i in (4, 3, 2, 1):
            setattr(self, "crp%d" % i, nn.Sequential(CRPBlock(width, width, stages)))
        for i in (4, 3, 2, 1):
            setattr(self, "merge%d" % i, Conv3x3(width, width))
        for i in (4, 3, 2, 1):
            setattr(self, "disp%d" % i, nn.Sequential(Conv3x3(width, 1), nn.Sigmoid()))

    def _stage(self, i, x):
        """Returns (stage output, disparity, deferred).  deferred = True: the stage output is still at HALF resolution --
        its x2 nearest up-sampling is folded into the consumers (the disparity head's pad here, the channel join of the
        next stage), so the up-sampled 256-channel map (189 MB at the last stage of C2) is never materialised."""
        x = getattr(self, "iconv%d" % i)(x, act="leaky_relu")
        x = getattr(self, "crp%d" % i)(x)
        x = getattr(self, "merge%d" % i)(x, act="leaky_relu")
        head = getattr(self, "disp%d" % i)
        if self.use_shuffle:
            # the reference up-samples stage 1 with up2 (depth_decoder.py:105), kept as is
            x = getattr(self, "up%d" % (2 if i == 1 else i))(x)
        elif x.is_cuda and x.dtype == torch.bfloat16 and x.is_contiguous(memory_format=torch.channels_last) \
                and x.shape[1] % 8 == 0 and not os.environ.get("TD_NO_FUSED_UPSAMPLE"):
            return x, head[1](head[0].forward_up(x)), True
        else:
            x = upsample(x)
        return x, head(x), False

    def forward(self, input_features, frame_id=0):
        _, l1, l2, l3, l4 = input_features
        l4 = self.do(l4)
        l3 = self.do(l3)
        def joined(r, x, d, deferred):
            if deferred:
                tail = d.to(x.dtype)
                if _ops().join_channels_up2_supported(r, x, tail) and not os.environ.get("TD_NO_CHANNEL_PAD"):
                    return _ops().join_channels_up2(r, x, tail)
                _fell_back("DepthDecoder.join_up2", x)
                x = upsample(x)
            parts = [r, x, d.to(x.dtype)]
            c = sum(p.shape[1] for p in parts)
            if x.is_cuda and c % 8 and x.dtype == torch.bfloat16 and not os.environ.get("TD_NO_CHANNEL_PAD"):
                if _round8(c) == r.shape[1] + x.shape[1] + 8 and _ops().join_channels_supported(*parts):
                    return _ops().join_channels(*parts)        # one HIP pass (ATen cat: 528 us at 48x160)
                _fell_back("DepthDecoder.join", x)
                # 513 -> 520 zero channels: see Conv3x3
                parts.append(x.new_zeros(x.shape[0], _round8(c) - c, x.shape[2], x.shape[3]))
            return torch.cat(parts, 1)

        x, d4, up = self._stage(4, self.reduce4(l4))
        x, d3, up = self._stage(3, joined(self.reduce3(l3), x, d4, up))
        x, d2, up = self._stage(2, joined(self.reduce2(l2), x, d3, up))
        x, d1, up = self._stage(1, joined(self.reduce1(l1), x, d2, up))
        self.outputs = {("disp", frame_id, 3): d4, ("disp", frame_id, 2): d3,
                        ("disp", frame_id, 1): d2, ("disp", frame_id, 0): d1}
        return self.outputs


class Decoder(nn.Module):
    """Auto-encoder image decoder: 5 x (conv, x2 nearest, conv), RGB heads at 4 scales
    (reference: decoder.py:7-57)."""
    out_key = "res_img"

    def __init__(self, num_ch_enc, num_output_channels=3, num_ch_dec=(16, 32, 64, 128, 256)):
        super().__init__()
        self.num_ch_dec = num_ch_dec
        d = num_ch_dec
        self.upconv5 = ConvBlock(num_ch_enc[4], d[4])
        self.upconv4 = ConvBlock(d[4], d[3])
        self.upconv3 = ConvBlock(d[3], d[2])
        self.upconv2 = ConvBlock(d[2], d[1])
        self.upconv1 = ConvBlock(d[1], d[0])
        self.iconv5 = ConvBlock(d[4], d[4])
        self.iconv4 = ConvBlock(d[3], d[3])
        self.iconv3 = ConvBlock(d[2], d[2])
        self.iconv2 = ConvBlock(d[1], d[1])
        self.iconv1 = ConvBlock(d[0], d[0])
        self.disp4 = Conv3x3(d[3], num_output_channels)
        self.disp3 = Conv3x3(d[2], num_output_channels)
        self.disp2 = Conv3x3(d[1], num_output_channels)
        self.disp1 = Conv3x3(d[0], num_output_channels)
        self.sigmoid = nn.Sigmoid()

    def _heads(self, outputs, key, frame_id, i4, i3, i2, i1):
        outputs[(key, frame_id, 3)] = self.sigmoid(self.disp4(i4))
        outputs[(key, frame_id, 2)] = self.sigmoid(self.disp3(i3))
        outputs[(key, frame_id, 1)] = self.sigmoid(self.disp2(i2))
        outputs[(key, frame_id, 0)] = self.sigmoid(self.disp1(i1))
        return outputs

    def forward(self, input_features, frame_id=0):
        x = self.iconv5.forward_up(self.upconv5(input_features[4]))      # iconv(upsample(upconv(.)))
        i4 = self.iconv4.forward_up(self.upconv4(x))
        i3 = self.iconv3.forward_up(self.upconv3(i4))
        i2 = self.iconv2.forward_up(self.upconv2(i3))
        i1 = self.iconv1.forward_up(self.upconv1(i2))
        self.outputs = self._heads({}, "res_img", frame_id, i4, i3, i2, i1)
        return self.outputs


class ColorDecoder(Decoder):
    """Colour (disentangled) decoder: the Decoder trunk with the predicted disparities added
    into its feature maps and optional encoder skips (reference: decoder.py:60-112)."""

    def __init__(self, num_ch_enc, num_output_channels=3, skip_connection_multiplier=1):
        super().__init__(num_ch_enc, num_output_channels, num_ch_dec=(16, 32, 64, 128, 256))
        self.skip_connection_multiplier = skip_connection_multiplier
        d = self.num_ch_dec
        self.upconv5_skip = ConvBlock(num_ch_enc[3], d[3])
        self.upconv4_skip = ConvBlock(num_ch_enc[2], d[2])
        self.upconv3_skip = ConvBlock(num_ch_enc[1], d[1])
        self.upconv2_skip = ConvBlock(num_ch_enc[0], d[0])

    def forward(self, input_features, outputs=None, frame_id=0, skip_layers=(None, None, None, None)):
        e1, e2, e3, e4, e5 = input_features
        m = self.skip_connection_multiplier

        def with_disp(low, scale):
            # disparity resized to the up-sampled resolution of ``low``
            dsp = F.interpolate(outputs[("disp", frame_id, scale)], [2 * low.shape[2], 2 * low.shape[3]], mode="bilinear",
                                align_corners=False)
            return (dsp * m).to(low.dtype)     # keep the decoder trunk in the activation dtype under autocast

        # nearest up-sampling commutes with the skip additions (upsample(a) + upsample(b) == upsample(a + b), also
        # after bf16 rounding), so they are taken at the low resolution and the up-sampling is fused into the pad
        l5 = self.upconv5(e5)
        i5 = self.iconv5.forward_up(l5) + with_disp(l5, 3)
        l4 = self.upconv4(i5)
        if skip_layers[0]:
            l4 = l4 + self.upconv5_skip(e4)
        i4 = self.iconv4.forward_up(l4) + with_disp(l4, 2)
        l3 = self.upconv3(i4)
        if skip_layers[1]:
            l3 = l3 + self.upconv4_skip(e3)
        i3 = self.iconv3.forward_up(l3) + with_disp(l3, 1)
        l2 = self.upconv2(i3)
        if skip_layers[2]:
            l2 = l2 + self.upconv3_skip(e2)
        i2 = self.iconv2.forward_up(l2) + with_disp(l2, 0)
        l1 = self.upconv1(i2)
        if skip_layers[3]:
            l1 = l1 + self.upconv2_skip(e1)
        i1 = self.iconv1.forward_up(l1)
        return self._heads(outputs, "auto_res_img", frame_id, i4, i3, i2, i1)


class IdentityPartial(nn.Module):
    """Keeps the left (or right) 1/part_ratio of the channels (reference: layers.py:392-406)."""

    def __init__(self, part_ratio=2, use_right=True):
        super().__init__()
        self.part_ratio = part_ratio
        self.use_right = use_right

    def forward(self, embedding):
        cut = embedding.size(1) // self.part_ratio
        return embedding[:, cut:] if self.use_right else embedding[:, :cut]
