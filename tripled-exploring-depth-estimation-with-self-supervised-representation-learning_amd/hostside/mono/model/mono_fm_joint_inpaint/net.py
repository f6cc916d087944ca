"""TripleD model family (reference: mono/model/mono_fm_joint_inpaint/net.py).  Registered here:
the classes the BASELINE configs name -- ``mono_fm_joint_inpaint`` (base),
``mono_fm_joint_inpaint_disentangle`` (cfg_kitti_tripleD) and
``mono_fm_joint_inpaint_disentangle_distill_sep_colorize`` (the all-aux-heads config).  The
reference's six other ablation classes are out of scope."""
import argparse
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..registry import MONO
from ..mono_fm_joint.net import mono_fm_joint, resize_bilinear
from ..attention import skip_attention
from ..networks import ColorDecoder, Conv1x1, DepthDecoder, Encoder, IdentityPartial, bn_groups
from .color_conversions import rgb2lab


def _streams():
    """tripled_amd.streams where the HIP package is importable (the model files also run stand-alone on the CPU)."""
    try:
        from tripled_amd import streams
    except ImportError:
        return None
    return streams


@MONO.register_module
class mono_fm_joint_inpaint(mono_fm_joint):
    """In-painting auto-encoder variant (reference :20-133)."""

    def __init__(self, options):
        super().__init__(options)
        self.use_perceptual = True
        if self.opt.get("freeze_extractor", False):
            for p in self.Encoder.parameters():
                p.requires_grad = False
        if self.opt.perception_weight == 0.0:
            del self.Encoder
            del self.Decoder
            self.use_perceptual = False
        if self.opt.get("img_reconstruct_weight", 1) == 0 and hasattr(self, "Decoder"):
            del self.Decoder

    def _autoencode(self, inputs, outputs, masked):
        if not self.use_perceptual:
            return None
        img = inputs[("color", 0, 0)]
        features = self.Encoder(img * inputs[("mask", 0, 0)] if masked else img)
        if self.opt.get("img_reconstruct_weight", 1) != 0:
            outputs.update(self.Decoder(features, 0))
        return features

    def forward(self, inputs):
        outputs = self.DepthDecoder(self.DepthEncoder(inputs["color_aug", 0, 0]))
        if self.training:
            outputs.update(self.predict_poses(inputs))
            features = self._autoencode(inputs, outputs, masked=True)
            return outputs, self.compute_losses(inputs, outputs, features)
        return outputs

    def _image_l1_map(self, pred, target, weight):
        """weight * compute_perceptional_loss(target, pred) as a [B,1,H,W] map: one HIP pass over the decoder output
        in its own dtype/layout on the GPU (tripled_amd.ops.robust_l1_map), the reference's ops on the CPU."""
        if pred.is_cuda:
            from tripled_amd import dispatch, ops
            if ops.robust_l1_map_supported(pred, target):
                return ops.robust_l1_map(pred, target, weight)
            dispatch.fallback("image_l1_map", "dtype %s, shape %s, strides %s" % (pred.dtype, tuple(pred.shape), pred.stride()))
        return self.compute_perceptional_loss(target, pred.float()) * weight

    def _masked_reconstruction(self, inputs, outputs, scale):
        """reference :80-91: photometric loss of the auto-encoder output against the resized target,
        averaged over erased pixels (mask == 0)."""
        opt = self.opt
        res_img = outputs[("res_img", 0, scale)].float()
        size = list(res_img.shape[2:])
        # (at scale 0 the size is the input's own: bilinear resampling at the pixel centres is a bit-exact identity, skipped)
        t_rs = resize_bilinear(inputs[("color", 0, 0)], size)
        hole = 1 - resize_bilinear(inputs[("mask", 0, 0)], size)
        if res_img.is_cuda and size[0] >= 3 and size[1] >= 3:
            # fused HIP path: SSIM + L1 + masked sum in one streaming kernel (and one for the gradient);
            # the 3 mask channels weight the same per-pixel loss, so they collapse to one plane
            from tripled_amd import ops
            loss = ops.masked_reconstruction_sum(res_img, t_rs, hole.sum(1)) / torch.sum(hole)
        else:
            if res_img.is_cuda:
                from tripled_amd import dispatch
                dispatch.fallback("masked_reconstruction", "size %s" % (size,))
            loss = self.compute_reprojection_loss(res_img, t_rs)
            loss = torch.sum(loss * hole) / torch.sum(hole)
        return loss / len(opt.scales) * opt.get("img_reconstruct_weight", 1)

    def _feature_regularization(self, features, target, i):
        # (/ 2^i / 5: exact powers of two times 0.2, one launch)
        return self.get_feature_regularization_loss(features[i], target) * (1.0 / ((2 ** i) * 5))

    def _autoencoder_losses(self, inputs, auto_out, features):
        """The loss terms that read nothing but the auto-encoder's own outputs (feature regularisation, reference :52-55, and
        the masked reconstruction of every scale, :80-91) can be computed where the auto-encoder runs -- on its side stream,
        next to the depth chain -- and handed to compute_losses, which puts them into the loss dictionary at their usual places.
        Opt-in (TD_EARLY_AUX_LOSSES=1): measured 3.4 ms SLOWER per step than computing them after the join (30.6 against 27.3 ms,
        profiles/r04/branch_streams_v1.txt) -- the graph's side branch then ends in 9 small loss nodes whose backward heads the
        auto-encoder's backward chain, and the replay serialises more of the two chains."""
        if features is None or os.environ.get("TD_EARLY_AUX_LOSSES", "0") != "1":
            return {}
        early = {}
        target = inputs[("color", 0, 0)]
        for i in range(5):
            early[("feature_regularization_loss", i)] = self._feature_regularization(features, target, i)
        if self.opt.get("img_reconstruct_weight", 1) != 0:
            for scale in self.opt.scales:
                early[("img_reconstruct_loss", scale)] = self._masked_reconstruction(inputs, auto_out, scale)
        return early

    def compute_losses(self, inputs, outputs, features):
        """reference :47-133."""
        opt = self.opt
        loss_dict = {}
        target = inputs[("color", 0, 0)]
        early = self.__dict__.pop("_branch_losses", None) or {}
        if features is not None:
            for i in range(5):
                key = ("feature_regularization_loss", i)
                loss_dict[key] = early[key] if key in early else self._feature_regularization(features, target, i)
            fused = self._fused_feature_metric(inputs, outputs, features[0]) \
                if self._fused_features_possible(inputs, self.Encoder) else None
            if fused is not None:
                loss, outputs["min_index"] = fused
                loss_dict["min_perceptional_loss"] = opt.perception_weight * loss
            else:
                outputs = self.generate_features_pred(inputs, outputs)
                tgt_f = features[0].float()
                cands = [self.compute_perceptional_loss(tgt_f, outputs[("feature", f, 0)]) for f in opt.frame_ids[1:]]
                vals, outputs["min_index"] = torch.min(torch.cat(cands, 1), dim=1)
                loss_dict["min_perceptional_loss"] = opt.perception_weight * vals.mean()
        ctx = self._begin_step(inputs)
        for scale in opt.scales:
            if features is not None and opt.get("img_reconstruct_weight", 1) != 0:
                key = ("img_reconstruct_loss", scale)
                loss_dict[key] = early[key] if key in early else self._masked_reconstruction(inputs, outputs, scale)
            self._photometric_scale(ctx, inputs, outputs, scale, loss_dict)
            self._smooth_scale(ctx, outputs, scale, loss_dict)
        return loss_dict


def _skip_layer(kind, channels, last):
    """depth_skip_layer_i for a non-disentangled level (reference :427-439): an attention gate ('ca', 'pa', 'asca':
    mono/model/attention.py), a 1x1 conv + BatchNorm + ELU on the last level ('1x1'), the identity otherwise."""
    gate = skip_attention(kind, channels)
    if gate is not None:
        return gate
    if kind == "1x1" and last:
        return nn.Sequential(Conv1x1(channels, channels), nn.BatchNorm2d(channels), nn.ELU())
    return nn.Identity()


@MONO.register_module
class mono_fm_joint_inpaint_disentangle(mono_fm_joint_inpaint):
    """cfg_kitti_tripleD's model (reference :398-532): the scene embedding of the depth encoder is
    split channel-wise at the flagged levels; one half feeds the depth decoder, the other the
    colour decoder, which also receives the predicted disparities."""

    def __init__(self, options):
        super().__init__(options)
        opt = self.opt
        self.depth_skip_type = opt.get("depth_skip_type", "use_half")
        self.depth_disentangle_type = opt.get("depth_disentangle_type", "use_half")
        self.color_skip_type = opt.get("color_skip_type", "use_half")
        self.use_pfp = opt.get("use_pfp", False)
        enc_ch = self.DepthEncoder.num_ch_enc
        n_levels = len(opt.disentangle_layers)

        depth_ch = []
        for ind, split in enumerate(opt.disentangle_layers):
            c = int(enc_ch[ind])
            if split:
                # (reference :410-426: the gate, if any, in front of the half that goes to the depth decoder)
                gate = skip_attention(self.depth_skip_type, c)
                head = [gate] if gate is not None else []
                if self.depth_disentangle_type == "use_half":
                    layer = nn.Sequential(*head, IdentityPartial(part_ratio=2, use_right=False))
                else:
                    layer = nn.Sequential(*head, Conv1x1(c, c // 2), nn.BatchNorm2d(c // 2), nn.ELU())
                depth_ch.append(c // 2)
            else:
                layer = _skip_layer(self.depth_skip_type, c, ind == n_levels - 1)
                depth_ch.append(c)
            setattr(self, "depth_skip_layer_{}".format(ind), layer)
        self.DepthDecoder = DepthDecoder(depth_ch, opt.get("depth_use_shuffle", False))

        color_ch = []
        opt["color_skip_layers"] = opt.get("color_skip_layers", (False, False, False, False))
        if self.color_skip_type == "1x1":
            ind = 0
            for ind, on in enumerate(opt.color_skip_layers):
                c = int(enc_ch[ind])
                if on:
                    layer = nn.Sequential(Conv1x1(c, c // 2), nn.BatchNorm2d(c // 2), nn.ELU())
                    color_ch.append(c // 2)
                else:
                    layer = nn.Identity()
                    color_ch.append(c)
                setattr(self, "color_skip_layer_{}".format(ind), layer)
            setattr(self, "color_skip_layer_{}".format(ind + 1), nn.Identity())
            color_ch.append(int(enc_ch[-1]))
        else:
            color_ch = [int(enc_ch[i]) // 2 if split else int(enc_ch[i])
                        for i, split in enumerate(opt.disentangle_layers)]
        self.ColorDecoder = ColorDecoder(color_ch, num_output_channels=3,
                                         skip_connection_multiplier=options.get("skip_connection_multiplier", 1))

    def forward(self, inputs):
        opt = self.opt
        fork = (self.training and getattr(self, "branch_streams", True) and _streams() is not None
                and _streams().enabled(inputs["color_aug", 0, 0]))
        if fork:
            # the auto-encoder and (without use_pfp) the pose network need only the input images: queued on two side streams,
            # they overlap the depth chain below, forward and backward (tripled_amd.streams).  Queueing them after the depth
            # chain instead (from an event taken here) measured the same within noise (profiles/r04/branch_streams_v1.txt).
            dev = inputs["color_aug", 0, 0].device
            auto, auto_out = _streams().Branch(dev, 0), {}
            with auto:
                features = self._autoencode(inputs, auto_out, masked=False)
                early = self._autoencoder_losses(inputs, auto_out, features)
            pose = None
            if not opt.get("use_pfp", False):
                pose = _streams().Branch(dev, 1)
                with pose:
                    pose_out = self.predict_poses(inputs)
        scene = self.DepthEncoder(inputs["color_aug", 0, 0])
        depth_emb = [getattr(self, "depth_skip_layer_{}".format(i))(scene[i])
                     for i in range(len(opt.disentangle_layers))]
        outputs = self.DepthDecoder(depth_emb)
        if not self.training:
            return outputs
        if self.color_skip_type == "1x1":
            n = len(opt.color_skip_layers)
            color_emb = [getattr(self, "color_skip_layer_{}".format(i))(scene[i]) for i in range(n)]
            color_emb.append(getattr(self, "color_skip_layer_{}".format(n))(scene[-1]))
        else:
            color_emb = [scene[i][:, scene[i].size(1) // 2:] if split else scene[i]
                         for i, split in enumerate(opt.disentangle_layers)]
        outputs = self.ColorDecoder(color_emb, outputs, skip_layers=opt.color_skip_layers)
        if fork:
            if pose is None:
                feats = {f: resize_bilinear(inputs["color_aug", f, 0], [192, 640]) for f in opt.frame_ids[1:]}
                feats[0] = resize_bilinear(outputs[("auto_res_img", 0, 0)].float(), [192, 640])
                pose_out = self.predict_poses(inputs, feats)
            else:
                pose.join(pose_out)
            outputs.update(pose_out)
            auto.join(features, auto_out, early)
            outputs.update(auto_out)
            self._branch_losses = early
            return outputs, self.compute_losses(inputs, outputs, features)
        if opt.get("use_pfp", False):
            feats = {f: resize_bilinear(inputs["color_aug", f, 0], [192, 640]) for f in opt.frame_ids[1:]}
            feats[0] = resize_bilinear(outputs[("auto_res_img", 0, 0)].float(), [192, 640])
            outputs.update(self.predict_poses(inputs, feats))
        else:
            outputs.update(self.predict_poses(inputs))
        features = self._autoencode(inputs, outputs, masked=False)
        return outputs, self.compute_losses(inputs, outputs, features)

    def compute_auto_res_loss(self, inputs, outputs):
        """reference :520-527 -- note: a per-pixel MAP, reduced later by batch_processor's mean."""
        if not self.opt.auto_res_weight > 0.0:
            return {}
        return {"auto_res_loss": self._image_l1_map(outputs[("auto_res_img", 0, 0)], inputs[("color", 0, 0)],
                                                    self.opt.auto_res_weight)}

    def compute_losses(self, inputs, outputs, features):
        loss_dict = super().compute_losses(inputs, outputs, features)
        loss_dict.update(self.compute_auto_res_loss(inputs, outputs))
        return loss_dict


@MONO.register_module
class mono_fm_joint_inpaint_disentangle_distill_sep_colorize(mono_fm_joint_inpaint):
    """All-aux-heads config (reference :261-329): depth decoder on the left half of the flagged
    levels plus a separate colourisation network (Lab: L in, ab out) as distillation target."""

    def __init__(self, options):
        super().__init__(options)
        opt = self.opt
        for ind, split in enumerate(opt.disentangle_layers):
            if split:
                self.DepthEncoder.num_ch_enc[ind] = self.DepthEncoder.num_ch_enc[ind] // 2
        self.DepthDecoder = DepthDecoder(self.DepthEncoder.num_ch_enc, opt.get("depth_use_shuffle", False))
        self.ColorizeEncoder = Encoder(opt.get("colorize_num_layers", 50), opt.colorize_pretrained_path)
        self.ColorizeDecoder = ColorDecoder(self.ColorizeEncoder.num_ch_enc, num_output_channels=2,
                                            skip_connection_multiplier=options.get("skip_connection_multiplier", 1))
        self.to_lab = rgb2lab

    def forward(self, inputs):
        opt = self.opt
        img = inputs[("color", 0, 0)]
        fork = (self.training and getattr(self, "branch_streams", True) and _streams() is not None
                and _streams().enabled(img))
        if fork:
            # auto-encoder, pose network and (unconditioned) colourisation encoder need only the input images: on side
            # streams, beside the depth chain (tripled_amd.streams; see mono_fm_joint_inpaint_disentangle.forward)
            auto, auto_out = _streams().Branch(img.device, 0), {}
            with auto:
                features = self._autoencode(inputs, auto_out, masked=False)
                early = self._autoencoder_losses(inputs, auto_out, features)
            pose = _streams().Branch(img.device, 1)
            with pose:
                pose_out = self.predict_poses(inputs)
            grey_emb = None
            if not opt.get("cond_encoder", False):
                colorize = _streams().Branch(img.device, 2)
                with colorize:
                    lab = self._lab(img)
                    grey_emb = self.ColorizeEncoder(lab[:, 0:1].expand(-1, 3, -1, -1), None)
        scene = self.DepthEncoder(inputs["color_aug", 0, 0])
        depth_emb = [scene[i][:, :scene[i].size(1) // 2] if split else scene[i]
                     for i, split in enumerate(opt.disentangle_layers)]
        outputs = self.DepthDecoder(depth_emb)
        if not self.training:
            return outputs
        if fork:
            pose.join(pose_out)
            outputs.update(pose_out)
            if grey_emb is None:
                lab = self._lab(img)
                grey_emb = self.ColorizeEncoder(lab[:, 0:1].expand(-1, 3, -1, -1), depth_emb)
            else:
                colorize.join(lab, grey_emb)
            outputs = self.ColorizeDecoder(grey_emb, outputs)
            inputs["gt_ab"] = lab[:, 1:]
            auto.join(features, auto_out, early)
            outputs.update(auto_out)
            self._branch_losses = early
            return outputs, self.compute_losses(inputs, outputs, features)
        outputs.update(self.predict_poses(inputs))
        lab = self._lab(img)
        grey = lab[:, 0:1].expand(-1, 3, -1, -1)
        grey_emb = self.ColorizeEncoder(grey, depth_emb if opt.get("cond_encoder", False) else None)
        outputs = self.ColorizeDecoder(grey_emb, outputs)
        inputs["gt_ab"] = lab[:, 1:]
        features = self._autoencode(inputs, outputs, masked=False)
        return outputs, self.compute_losses(inputs, outputs, features)

    def _lab(self, img):
        if img.is_cuda:
            from tripled_amd import ops
            return ops.rgb2lab(img, 50.0, 50.0, 110.0)      # one HIP pass (color_conversions.py: ~25 element-wise launches)
        return self.to_lab(img, argparse.Namespace(l_cent=50.0, l_norm=50.0, ab_norm=110.0))

    def compute_colorization_loss(self, inputs, outputs):
        """reference :310-323."""
        opt = self.opt
        if not opt.colorize_weight > 0.0:
            return {}
        loss = self._image_l1_map(outputs[("auto_res_img", 0, 0)], inputs["gt_ab"], 1.0)
        if opt.get("use_distill_mask", False):
            hole = 1 - inputs[("mask", 0, 0)][:, 0:1]
            loss = torch.sum(loss * hole) / torch.sum(hole)
        return {"distill_colorize_loss": loss * opt.colorize_weight}

    def compute_losses(self, inputs, outputs, features):
        loss_dict = super().compute_losses(inputs, outputs, features)
        loss_dict.update(self.compute_colorization_loss(inputs, outputs))
        return loss_dict
