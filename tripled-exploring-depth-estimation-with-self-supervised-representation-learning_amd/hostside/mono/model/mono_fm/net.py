"""FeatDepth model with a separate feature extractor -- cfg_kitti_fm, BASELINE config #1
(reference: mono/model/mono_fm/net.py).  Same hot path as the joint family; the perceptual
term is evaluated per scale with weight perception_weight / n_scales."""
import torch

from ..registry import MONO
from ..networks import DepthDecoder, DepthEncoder, Encoder, PoseDecoder, PoseEncoder, install_counter_hook
from ..mono_fm_joint.layers import SSIM, Backproject, Project
from ..mono_fm_joint.net import mono_fm_joint


def build_extractor(num_layers, pretrained_path, use_extractor_directly=False):
    """reference :15-26: a frozen copy of a pre-trained auto-encoder's Encoder when a checkpoint
    is given, otherwise a trainable randomly initialised one."""
    if use_extractor_directly:
        return Encoder(num_layers, pretrained_path)
    extractor = Encoder(num_layers, None)
    if pretrained_path is not None:
        ckpt = torch.load(pretrained_path, map_location="cpu", weights_only=True)
        own = extractor.state_dict()
        for name in own:
            own[name].copy_(ckpt["state_dict"]["Encoder." + name])
        for p in extractor.parameters():
            p.requires_grad = False
    return extractor


@MONO.register_module
class mono_fm(mono_fm_joint):
    def __init__(self, options):
        torch.nn.Module.__init__(self)
        self.opt = options
        self.DepthEncoder = DepthEncoder(self.opt.depth_num_layers, self.opt.depth_pretrained_path)
        self.DepthDecoder = DepthDecoder(self.DepthEncoder.num_ch_enc)
        self.PoseEncoder = PoseEncoder(self.opt.pose_num_layers, self.opt.pose_pretrained_path)
        self.PoseDecoder = PoseDecoder(self.PoseEncoder.num_ch_enc)
        self.extractor = build_extractor(self.opt.get("extractor_num_layers", 50),
                                         self.opt.extractor_pretrained_path,
                                         self.opt.get("use_extractor_directly", False))
        self.ssim = SSIM()
        self.backproject = Backproject(self.opt.imgs_per_gpu, self.opt.height, self.opt.width)
        self.project = Project(self.opt.imgs_per_gpu, self.opt.height, self.opt.width)
        self._loss_backend = None
        self._noise_fn = None
        install_counter_hook(self)

    def forward(self, inputs):
        outputs = self.DepthDecoder(self.DepthEncoder(inputs["color_aug", 0, 0]))
        if self.training:
            outputs.update(self.predict_poses(inputs))
            return outputs, self.compute_losses(inputs, outputs)
        return outputs

    def _source_features(self, img):
        if self.opt.get("prune_extractor_tail", False):
            return self.extractor.stem_only(img)
        return self.extractor(img)[0]

    def _target_features(self, img):
        if self.opt.get("prune_extractor_tail", False):
            return self.extractor.stem_only(img)
        return self.extractor(img)[0]

    def compute_losses(self, inputs, outputs):
        """reference :69-133.  Per scale: photometric min-reprojection, perceptual min-loss over the
        warped extractor features (the reference re-runs the extractor on the target once per
        source frame, :113 -- kept), smoothness."""
        opt = self.opt
        loss_dict = {}
        n_scales = len(opt.scales)
        ctx = self._begin_step(inputs)
        for scale in opt.scales:
            self._photometric_scale(ctx, inputs, outputs, scale, loss_dict)
            fused = None
            if self._fused_features_possible(inputs, self.extractor):
                def target_features():
                    # the reference evaluates the extractor on the target once per source frame (:113), after the
                    # source frames (:198 inside generate_features_pred); the passes are kept (they advance the
                    # BatchNorm running statistics) and the last one feeds the fused warp + min kernel
                    for _ in opt.frame_ids[1:]:
                        f = self._target_features(inputs[("color", 0, 0)])
                    return f
                fused = self._fused_feature_metric(inputs, outputs, target_features)
            if fused is not None:
                loss, outputs[("min_index", scale)] = fused
                loss_dict[("min_perceptional_loss", scale)] = opt.perception_weight * loss / n_scales
            else:
                outputs = self.generate_features_pred(inputs, outputs)
                cands = []
                for f in opt.frame_ids[1:]:
                    tgt_f = self._target_features(inputs[("color", 0, 0)]).float()
                    cands.append(self.compute_perceptional_loss(tgt_f, outputs[("feature", f, 0)]))
                vals, outputs[("min_index", scale)] = torch.min(torch.cat(cands, 1), dim=1)
                loss_dict[("min_perceptional_loss", scale)] = opt.perception_weight * vals.mean() / n_scales
            self._smooth_scale(ctx, outputs, scale, loss_dict)
        return loss_dict
