"""FeatDepth-joint base model (reference: mono/model/mono_fm_joint/net.py).  It carries the
geometry helpers and loss methods every TripleD variant inherits.  The photometric and
smoothness terms run in the fused HIP kernels (mono.model.hotpath); the auxiliary terms
(feature regularisation, feature-metric warp, reconstruction) are torch ops for now
(SURVEY.md section 8f rank 1)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..registry import MONO
from ..networks import (DepthEncoder, DepthDecoder, PoseEncoder, PoseDecoder, Encoder, Decoder, install_counter_hook,
                        bn_groups, bn_groups_supported)
from .layers import SSIM, Backproject, Project
from .. import hotpath


def resize_bilinear(img, size):
    """F.interpolate(img, size, mode="bilinear", align_corners=False).  When ``img`` already has that size the
    sampling positions are the pixel centres (scale 1, lambda 0), i.e. the op is a bit-exact identity, and
    the copy is skipped (the KITTI configs feed 192x640 images to a 192x640 pose network)."""
    if list(img.shape[-2:]) == list(size):
        return img
    return F.interpolate(img, size, mode="bilinear", align_corners=False)


@MONO.register_module
class mono_fm_joint(nn.Module):
    def __init__(self, options):
        super().__init__()
        self.opt = options
        if self.opt.get("use_diffnet", False) or self.opt.get("use_hr_depth", False):
            raise NotImplementedError("use_diffnet / use_hr_depth are off in every supported config "
                                      "(the reference's diffnet branch needs a network download)")
        self.DepthEncoder = DepthEncoder(self.opt.depth_num_layers, self.opt.depth_pretrained_path)
        self.DepthDecoder = DepthDecoder(self.DepthEncoder.num_ch_enc, self.opt.get("depth_use_shuffle", False))
        self.PoseEncoder = PoseEncoder(self.opt.pose_num_layers, self.opt.pose_pretrained_path)
        self.PoseDecoder = PoseDecoder(self.PoseEncoder.num_ch_enc)
        self.Encoder = Encoder(self.opt.get("extractor_num_layers", 50), self.opt.extractor_pretrained_path)
        self.Decoder = Decoder(self.Encoder.num_ch_enc)
        self.ssim = SSIM()
        self.backproject = Backproject(self.opt.imgs_per_gpu, self.opt.height, self.opt.width)
        self.project = Project(self.opt.imgs_per_gpu, self.opt.height, self.opt.width)
        self._loss_backend = None
        self._noise_fn = None
        install_counter_hook(self)

    # ------------------------------------------------------------------ plumbing
    def set_loss_backend(self, backend):
        """Substitute the loss hot-path implementation (tests / cpu_baseline use the oracle)."""
        self._loss_backend = backend

    def set_noise_source(self, fn):
        """fn(shape, device) -> N(0,1) tensor; used to replay recorded auto-mask noise in tests."""
        self._noise_fn = fn

    @property
    def loss_backend(self):
        if self._loss_backend is None:
            self._loss_backend = hotpath.default_backend()
        return self._loss_backend

    def _automask_noise(self, n, shape, device):
        """The reference draws torch.randn(shape) on the CPU generator per source frame and
        scale and uploads it (mono_fm_joint_inpaint/net.py:105).  'device' (default) draws on the
        GPU generator instead; 'cpu' reproduces the reference's generator stream."""
        if self._noise_fn is not None:
            return torch.stack([self._noise_fn(shape, device).reshape(shape) for _ in range(n)], 0)
        if self.opt.get("automask_noise", "device") == "cpu":
            return torch.stack([torch.randn(shape) for _ in range(n)], 0).to(device, non_blocking=True)
        return torch.randn((n,) + tuple(shape), device=device)

    # ------------------------------------------------------------------ forward
    def forward(self, inputs):
        outputs = self.DepthDecoder(self.DepthEncoder(inputs["color_aug", 0, 0]))
        if self.training:
            outputs.update(self.predict_poses(inputs))
            features = self.Encoder(inputs[("color", 0, 0)])
            outputs.update(self.Decoder(features, 0))
            loss_dict = self.compute_losses(inputs, outputs, features)
            return outputs, loss_dict
        return outputs

    # ------------------------------------------------------------------ small loss pieces
    def robust_l1(self, pred, target):
        eps = 1e-3
        return torch.sqrt(torch.pow(target - pred, 2) + eps ** 2)

    def compute_perceptional_loss(self, tgt_f, src_f):
        return self.robust_l1(tgt_f, src_f).mean(1, True)

    def compute_reprojection_loss(self, pred, target):
        l1 = self.robust_l1(pred, target).mean(1, True)
        ssim = self.ssim(pred, target).mean(1, True)
        return 0.85 * ssim + 0.15 * l1

    def disp_to_depth(self, disp, min_depth, max_depth):
        lo, hi = 1 / max_depth, 1 / min_depth
        scaled = lo + (hi - lo) * disp
        return scaled, 1 / scaled

    def gradient(self, D):
        return D[:, :, :, 1:] - D[:, :, :, :-1], D[:, :, 1:] - D[:, :, :-1]

    # ------------------------------------------------------------------ hot path
    def _begin_step(self, inputs):
        srcs = [inputs[("color", f, 0)] for f in self.opt.frame_ids[1:]]
        return self.loss_backend.begin_step(self.opt, inputs[("color", 0, 0)].float(), [s.float() for s in srcs],
                                            inputs["K"].float(), inputs["inv_K"].float())

    def _frame_transforms(self, inputs, outputs):
        return [inputs["stereo_T"] if f == "s" else outputs[("cam_T_cam", 0, f)].float()
                for f in self.opt.frame_ids[1:]]

    def _photometric_scale(self, ctx, inputs, outputs, scale, loss_dict):
        """generate_images_pred + automask + min-reprojection for one scale (reference:
        mono_fm_joint/net.py:181-194, mono_fm_joint_inpaint/net.py:93-117), fused."""
        opt = self.opt
        target = ctx.target
        noise = None
        if opt.automask:
            b, _, h, w = target.shape
            noise = self._automask_noise(len(ctx.sources), (b, h, w), target.device)
        keep = bool(opt.get("keep_warped_images", False))
        loss, min_index, warped = self.loss_backend.photometric(
            ctx, outputs[("disp", 0, scale)].float(), self._frame_transforms(inputs, outputs), noise, keep,
            P=outputs.get(("cam_P", 0)))
        if warped is not None:
            for f, wimg in zip(opt.frame_ids[1:], warped):
                outputs[("color", f, scale)] = wimg
        outputs[("min_index", scale)] = min_index
        loss_dict[("min_reconstruct_loss", scale)] = loss

    def _smooth_scale(self, ctx, outputs, scale, loss_dict):
        """disp mean-normalisation + get_smooth_loss (reference: mono_fm_joint_inpaint/net.py:119-131)."""
        opt = self.opt
        weight = opt.smoothness_weight / (2 ** scale) / len(opt.scales)
        loss_dict[("smooth_loss", scale)] = self.loss_backend.smooth(
            ctx, outputs[("disp", 0, scale)].float(), weight, bool(opt.disp_norm))

    def generate_images_pred(self, inputs, outputs, scale):
        """Public counterpart of the reference method: fills outputs[("color", f, scale)]."""
        ctx = self._begin_step(inputs)
        _, _, warped = self.loss_backend.photometric(ctx, outputs[("disp", 0, scale)].float(),
                                                     self._frame_transforms(inputs, outputs), None, True)
        for f, wimg in zip(self.opt.frame_ids[1:], warped):
            outputs[("color", f, scale)] = wimg
        return outputs

    def get_smooth_loss(self, disp, img):
        """Un-weighted smooth1 + smooth2 of the reference (net.py:279-302), via the HIP kernel."""
        ctx = hotpath.StepContext()
        ctx.target, ctx.pyramid = img.float(), {}
        return self.loss_backend.smooth(ctx, disp.float(), 1.0, False)

    # ------------------------------------------------------------------ auxiliary terms (torch ops)
    def get_feature_regularization_loss(self, feature, img):
        """net.py:309-330: -dis * first-order + cvt * second-order edge-aware terms (a = 1)."""
        if feature.is_cuda:
            from tripled_amd import ops
            h, w = feature.shape[2:]
            if ops.featreg_supported(feature) and img.shape[2] % h == 0 and img.shape[3] % w == 0:
                # fused HIP path: reads the (bf16/f32, channels-last) feature map directly
                return ops.feature_regularization(feature, ops.area_downsample(img, h, w), self.opt.dis, self.opt.cvt)
            from tripled_amd import dispatch
            dispatch.fallback("feature_regularization", "dtype %s, shape %s" % (feature.dtype, tuple(feature.shape)))
        feature = feature.float()
        img = F.adaptive_avg_pool2d(img, feature.shape[2:])
        f_dx, f_dy = self.gradient(feature)
        i_dx, i_dy = self.gradient(img)
        f_dxx, f_dxy = self.gradient(f_dx)
        f_dyx, f_dyy = self.gradient(f_dy)
        i_dxx, i_dxy = self.gradient(i_dx)
        i_dyx, i_dyy = self.gradient(i_dy)

        def term(fd, idf):
            return torch.mean(fd.abs() * torch.exp(-idf.abs().mean(1, True)))

        smooth1 = term(f_dx, i_dx) + term(f_dy, i_dy)
        smooth2 = term(f_dxx, i_dxx) + term(f_dxy, i_dxy) + term(f_dyx, i_dyx) + term(f_dyy, i_dyy)
        return -self.opt.dis * smooth1 + self.opt.cvt * smooth2

    def _source_features(self, img):
        """features[0] of the extractor for a source frame.  The reference runs the whole
        ResNet and discards everything but the stem output (net.py:221); that is kept unless
        ``prune_extractor_tail`` is set (identical losses/gradients, Encoder.layer1-4 BN running
        statistics then no longer see the source frames)."""
        if self.opt.get("prune_extractor_tail", False):
            return self.Encoder.stem_only(img)
        return self.Encoder(img)[0]

    def _fused_features_possible(self, inputs, extractor):
        """Decided BEFORE any extractor pass runs (a declined fused path must not leave extra BatchNorm
        running-statistics updates behind): HIP inputs, nobody asked for the warped feature maps, no stereo
        frame, and the extractor produces channels-last activations."""
        opt = self.opt
        if not inputs[("color", 0, 0)].is_cuda or opt.get("keep_warped_images", False) or "s" in opt.frame_ids:
            return False
        if not extractor.encoder.conv1.weight.is_contiguous(memory_format=torch.channels_last):
            from tripled_amd import dispatch
            dispatch.fallback("feature_metric_warp", "the extractor is not in channels_last memory format")
            return False
        return True

    def _fused_feature_metric(self, inputs, outputs, tgt_f):
        """Feature-metric term through the fused HIP kernel (no warped feature maps are materialised).
        ``tgt_f``: the target features, or a callable that produces them AFTER the source-frame passes (the
        reference's order in mono_fm, which fixes the order of the BatchNorm running-statistics updates).
        None when the feature layout is not one the kernel takes (reported as a fallback)."""
        opt = self.opt
        from tripled_amd import ops
        imgs = [inputs[("color", f, 0)] for f in opt.frame_ids[1:]]
        if self._batch_frames(imgs):
            # both source frames through the extractor in one stacked pass (per-pass BatchNorm statistics)
            with bn_groups(len(imgs)):
                stacked = self._source_features(torch.cat(imgs, 0))
            src_f = list(stacked.split(imgs[0].shape[0], 0))      # split: its backward is one cat, not zero-filled slices
        else:
            src_f = [self._source_features(img) for img in imgs]
        if callable(tgt_f):
            tgt_f = tgt_f()
        if not ops.featwarp_supported(tgt_f, src_f):
            from tripled_amd import dispatch
            dispatch.fallback("feature_metric_warp", "dtype %s, shape %s" % (tgt_f.dtype, tuple(tgt_f.shape)))
            return None
        inv_K = inputs["inv_K"].float().clone()
        inv_K[:, :, 0:2] = inv_K[:, :, 0:2] * 2          # see generate_features_pred
        P_full = outputs.get(("cam_P", 0))
        if P_full is not None:
            # K' = diag(.5, .5, 1, 1) K  =>  (K' T)[:3] is (K T)[:3] with rows 0 and 1 halved (exact: powers of two)
            P = torch.cat([P_full[:, :, :2] * 0.5, P_full[:, :, 2:]], 2)
        else:
            K = inputs["K"].float().clone()
            K[:, 0:2, :] = K[:, 0:2, :] / 2
            P = torch.stack([torch.matmul(K, outputs[("cam_T_cam", 0, f)].float())[:, :3, :] for f in opt.frame_ids[1:]], 0)
        return ops.feature_warp_min_loss(tgt_f, src_f, outputs[("disp", 0, 0)].float(), P, inv_K,
                                         opt.min_depth, opt.max_depth)

    def generate_features_pred(self, inputs, outputs):
        """net.py:196-223: warp the extractor's stem features of each source frame to the target
        view at half resolution (K rows 0,1 halved, inv_K re-derived)."""
        opt = self.opt
        h2, w2 = int(opt.height / 2), int(opt.width / 2)
        disp = F.interpolate(outputs[("disp", 0, 0)].float(), [h2, w2], mode="bilinear", align_corners=False)
        _, depth = self.disp_to_depth(disp, opt.min_depth, opt.max_depth)
        K = inputs["K"].float().clone()
        K[:, 0:2, :] = K[:, 0:2, :] / 2
        if opt.get("feature_invK", "scaled") == "pinv":
            inv_K = torch.linalg.pinv(K)      # the reference's per-sample torch.pinverse (SVD, slow)
        else:
            # K' = diag(.5,.5,1,1) K  =>  K'^-1 = K^-1 diag(2,2,1,1): reuse the batch's inv_K
            # (the dataset ships inv_K = pinv(K) of an invertible K, mono_dataset.py:174-180)
            inv_K = inputs["inv_K"].float().clone()
            inv_K[:, :, 0:2] = inv_K[:, :, 0:2] * 2
        b = depth.shape[0]
        ys, xs = torch.meshgrid(torch.arange(h2, device=depth.device, dtype=torch.float32),
                                torch.arange(w2, device=depth.device, dtype=torch.float32), indexing="ij")
        pix = torch.stack([xs.reshape(-1), ys.reshape(-1), torch.ones(h2 * w2, device=depth.device)], 0)
        pts = depth.reshape(b, 1, -1) * torch.matmul(inv_K[:, :3, :3], pix.unsqueeze(0))
        pts = torch.cat([pts, torch.ones(b, 1, h2 * w2, device=depth.device)], 1)
        for frame_id in opt.frame_ids[1:]:
            T = inputs["stereo_T"] if frame_id == "s" else outputs[("cam_T_cam", 0, frame_id)].float()
            cam = torch.matmul(torch.matmul(K, T)[:, :3, :], pts)
            uv = cam[:, :2, :] / (cam[:, 2:3, :] + 1e-7)
            gx = (uv[:, 0].reshape(b, h2, w2) / (w2 - 1) - 0.5) * 2
            gy = (uv[:, 1].reshape(b, h2, w2) / (h2 - 1) - 0.5) * 2
            src_f = self._source_features(inputs[("color", frame_id, 0)]).float()
            outputs[("feature", frame_id, 0)] = F.grid_sample(src_f, torch.stack([gx, gy], -1), mode="bilinear",
                                                              padding_mode="border", align_corners=False)
        return outputs

    # ------------------------------------------------------------------ poses
    def predict_poses(self, inputs, pose_feats=None):
        """net.py:164-179: pose net on [previous, current] pairs resized to 192x640."""
        outputs = {}
        if pose_feats is None:
            pose_feats = {f: resize_bilinear(inputs["color_aug", f, 0], [192, 640]) for f in self.opt.frame_ids}
        frames = [f for f in self.opt.frame_ids[1:] if f != "s"]
        pairs = [torch.cat([pose_feats[f], pose_feats[0]] if f < 0 else [pose_feats[0], pose_feats[f]], 1)
                 for f in frames]
        if self._batch_frames(pairs):
            # one pass over the stacked pairs; BatchNorm keeps one set of batch statistics per pair
            # (networks.bn_groups), so this equals the reference's separate passes (net.py:172-178)
            with bn_groups(len(pairs)):
                axisangle, translation = self.PoseDecoder(self.PoseEncoder(torch.cat(pairs, 0)))
            n = pairs[0].shape[0]
            per_pair = list(zip(axisangle.split(n, 0), translation.split(n, 0)))
            stacked = (axisangle, translation)
        else:
            per_pair = [self.PoseDecoder(self.PoseEncoder(x)) for x in pairs]
            stacked = None
        if pairs and pairs[0].is_cuda:
            # one HIP launch for every pair: Rodrigues + translation + K @ T (tripled_amd.ops.pose_transforms)
            from tripled_amd import ops
            if stacked is None:
                stacked = (torch.cat([a for a, _ in per_pair], 0), torch.cat([t for _, t in per_pair], 0))
            T, P = ops.pose_transforms(stacked[0], stacked[1], inputs["K"].float(), [f < 0 for f in frames])
            for i, f in enumerate(frames):
                outputs[("cam_T_cam", 0, f)] = T[i]
            if len(frames) == len(self.opt.frame_ids) - 1:        # no stereo frame: P covers every source frame
                outputs[("cam_P", 0)] = P
            return outputs
        for f, (axisangle, translation) in zip(frames, per_pair):
            outputs[("cam_T_cam", 0, f)] = self.transformation_from_parameters(
                axisangle[:, 0], translation[:, 0], invert=(f < 0))
        return outputs

    def _batch_frames(self, tensors):
        """Stack same-network passes over different frames into one batch: only where the grouped BatchNorm
        kernels run (HIP, training) so that batch statistics stay per pass."""
        if not (len(tensors) > 1 and self.training and tensors[0].is_cuda and self.opt.get("batch_frame_passes", True)):
            return False
        if getattr(self, "_bn_groups_ok", None) is None:
            self._bn_groups_ok = bn_groups_supported(self)
        return self._bn_groups_ok

    def transformation_from_parameters(self, axisangle, translation, invert=False):
        """net.py:225-236: M = R^T Trans(-t) when inverting, else Trans(t) R."""
        R = self.rot_from_axisangle(axisangle)
        t = translation
        if invert:
            R = R.transpose(1, 2)
            t = -t
        T = self.get_translation_matrix(t)
        return torch.matmul(R, T) if invert else torch.matmul(T, R)

    def get_translation_matrix(self, translation_vector):
        t = translation_vector.contiguous().view(-1, 3, 1)
        b = t.shape[0]
        eye = torch.eye(4, device=t.device, dtype=t.dtype).unsqueeze(0).expand(b, 4, 4)
        top = torch.cat([eye[:, :3, :3], t], 2)
        return torch.cat([top, eye[:, 3:, :]], 1)

    def rot_from_axisangle(self, vec):
        """net.py:248-277 (Rodrigues, axis = v / (|v| + 1e-7)), built without in-place writes."""
        angle = torch.norm(vec, 2, 2, True)
        axis = vec / (angle + 1e-7)
        ca, sa = torch.cos(angle).reshape(-1), torch.sin(angle).reshape(-1)
        C = 1 - ca
        x, y, z = axis[:, 0, 0], axis[:, 0, 1], axis[:, 0, 2]
        o, l = torch.zeros_like(x), torch.ones_like(x)
        rot = torch.stack([x * (x * C) + ca, x * (y * C) - z * sa, z * (x * C) + y * sa, o,
                           x * (y * C) + z * sa, y * (y * C) + ca, y * (z * C) - x * sa, o,
                           z * (x * C) - y * sa, y * (z * C) + x * sa, z * (z * C) + ca, o,
                           o, o, o, l], 1)
        return rot.reshape(-1, 4, 4)

    # ------------------------------------------------------------------ orchestrator
    def compute_losses(self, inputs, outputs, features):
        """net.py:73-155: feature regularisation, then per scale the auto-encoder reconstruction,
        photometric min-reprojection, perceptual min-loss and smoothness terms."""
        opt = self.opt
        loss_dict = {}
        target = inputs[("color", 0, 0)]
        n_scales = len(opt.scales)
        for i in range(5):
            loss_dict[("feature_regularization_loss", i)] = \
                self.get_feature_regularization_loss(features[i], target) * (1.0 / ((2 ** i) * 5))     # (/ 2^i / 5: exact powers of two times 0.2, one launch)
        ctx = self._begin_step(inputs)
        for scale in opt.scales:
            res_img = outputs[("res_img", 0, scale)].float()
            t_rs = F.interpolate(target, list(res_img.shape[2:]), mode="bilinear", align_corners=False)
            loss_dict[("img_reconstruct_loss", scale)] = self.compute_reprojection_loss(res_img, t_rs).mean() / n_scales
            pm = {}
            self._photometric_scale(ctx, inputs, outputs, scale, pm)
            outputs = self.generate_features_pred(inputs, outputs)
            loss_dict[("min_reconstruct_loss", scale)] = pm[("min_reconstruct_loss", scale)]
            cands = [self.compute_perceptional_loss(features[0].float(), outputs[("feature", f, 0)])
                     for f in opt.frame_ids[1:]]
            vals, outputs[("min_index", scale)] = torch.min(torch.cat(cands, 1), dim=1)
            loss_dict[("min_perceptional_loss", scale)] = opt.perception_weight * vals.mean() / n_scales
            self._smooth_scale(ctx, outputs, scale, loss_dict)
        return loss_dict
