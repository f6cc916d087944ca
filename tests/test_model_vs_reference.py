"""In-container end-to-end cross-check against the REAL reference model classes (imported from
/root/reference with the stub recipe of tools/gen_golden.py).  Skipped where the reference
checkout is absent (the GPU box).  Both models get the same weights (strict state_dict load --
the checkpoint-key contract), the same inputs and the same recorded auto-mask noise; the
build's model runs its loss hot path through the oracle backend here (CPU)."""
import os
import sys

import pytest
import torch

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "mono")), reason="reference checkout not present")


class Opt(dict):
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


def _reference_classes():
    """Import the reference's model modules, return the classes, and restore sys.modules so the
    build's own ``mono`` package can be imported afterwards."""
    sys.dont_write_bytecode = True
    saved = {k: v for k, v in sys.modules.items() if k == "mono" or k.startswith("mono.")}
    for k in saved:
        del sys.modules[k]
    cuda_orig = torch.Tensor.cuda
    try:
        from tools import gen_golden
        gen_golden.install_reference(REF)
        import importlib
        inpaint = importlib.import_module("mono.model.mono_fm_joint_inpaint.net")
        fm = importlib.import_module("mono.model.mono_fm.net")
        joint = importlib.import_module("mono.model.mono_fm_joint.net")
        classes = {"mono_fm_joint_inpaint_disentangle": inpaint.mono_fm_joint_inpaint_disentangle,
                   "mono_fm_joint_inpaint": inpaint.mono_fm_joint_inpaint,
                   "mono_fm_joint_inpaint_disentangle_distill_sep_colorize":
                       inpaint.mono_fm_joint_inpaint_disentangle_distill_sep_colorize,
                   "mono_fm": fm.mono_fm, "mono_fm_joint": joint.mono_fm_joint}
        tap = gen_golden.NoiseTap
    finally:
        for k in [k for k in sys.modules if k == "mono" or k.startswith("mono.")]:
            del sys.modules[k]
        sys.modules.update(saved)
    return classes, tap, cuda_orig


def _options(name, B, H, W):
    o = Opt(name=name, depth_num_layers=18, pose_num_layers=18, extractor_num_layers=18, frame_ids=[0, -1, 1],
            imgs_per_gpu=B, height=H, width=W, scales=[0, 1, 2, 3], min_depth=0.1, max_depth=100.0,
            depth_pretrained_path=None, pose_pretrained_path=None, extractor_pretrained_path=None,
            automask=True, disp_norm=True, dis=1e-3, cvt=1e-3, perception_weight=1e-3, smoothness_weight=1e-3,
            auto_res_weight=5e-3, disentangle_layers=[False, False, False, False, True],
            skip_connection_multiplier=1, depth_skip_type=None, color_skip_type=None,
            color_skip_layers=[False] * 4, depth_use_shuffle=False, depth_disentangle_type="use_half",
            freeze_extractor=False, colorize_num_layers=18, colorize_pretrained_path=None, colorize_weight=1e-2)
    return o


def _inputs(B, H, W, stereo=False):
    from tests.util import kitti_K, make_triplet
    g = torch.Generator().manual_seed(0)
    fr = make_triplet(g, B, H, W)
    K, iK = kitti_K(B, H, W)
    inputs = {}
    if stereo:      # the other camera of the rig: the target shifted by a few pixels, and the rig's extrinsics (mono_dataset.py:194-199)
        fr["s"] = torch.roll(fr[0], 3, dims=3).contiguous()
        stereo_T = torch.eye(4).repeat(B, 1, 1)
        stereo_T[:, 0, 3] = -0.015
        inputs["stereo_T"] = stereo_T
    for f in fr:
        inputs[("color", f, 0)] = fr[f]
        inputs[("color_aug", f, 0)] = (fr[f] * 0.9 + 0.03).contiguous()
    mask = torch.ones(B, 3, H, W)
    mask[:, :, 10:26, 30:46] = 0
    mask[:, :, 50:66, 80:96] = 0
    inputs[("mask", 0, 0)] = mask
    inputs["K"], inputs["inv_K"] = K, iK
    return inputs


@pytest.mark.parametrize("name", ["mono_fm_joint_inpaint_disentangle", "mono_fm",
                                  "mono_fm_joint_inpaint_disentangle_distill_sep_colorize"])
def test_same_weights_same_losses(name):
    _compare(name, {})


# the option sets of the BASELINE configs that the 18-layer default above does not cover
CONFIG_OPTION_SETS = {
    # config/cfg_kitti_tripleD.py (C2-C4): ResNet50 depth encoder and feature auto-encoder (Bottleneck blocks, 2048-channel heads)
    "C2_resnet50": ("mono_fm_joint_inpaint_disentangle", dict(depth_num_layers=50, extractor_num_layers=50)),
    # config/cfg_kitti_fm_joint_inpaint_disentangle_distill_full_colorize.py (C5): no disentangled layer, the colourisation term
    # masked by the erased regions, its weights
    "C5_options": ("mono_fm_joint_inpaint_disentangle_distill_sep_colorize",
                   dict(disentangle_layers=[False] * 5, use_distill_mask=True, img_reconstruct_weight=1, colorize_weight=5e-3)),
}


@pytest.mark.parametrize("case", sorted(CONFIG_OPTION_SETS))
def test_config_option_sets_match_the_reference(case):
    name, overrides = CONFIG_OPTION_SETS[case]
    _compare(name, overrides)


# the other switches of the reference's model options that this build's classes implement (reference: mono_fm_joint_inpaint/net.py),
# each against the real class with the same weights: losses, disparities, arg-min selection and every parameter gradient
TRIPLED, COLORIZE = "mono_fm_joint_inpaint_disentangle", "mono_fm_joint_inpaint_disentangle_distill_sep_colorize"
REFERENCE_OPTION_SETS = {
    "joint_class": ("mono_fm_joint", {}),                                                            # mono_fm_joint/net.py (cfg_kitti_fm_joint.py)
    "base_inpaint_class": ("mono_fm_joint_inpaint", {}),                                             # :20-133
    "use_pfp": (TRIPLED, dict(use_pfp=True)),                                                        # :405, :502-506 pose from the restored image
    "depth_use_shuffle": (TRIPLED, dict(depth_use_shuffle=True)),                                    # :444-447, depth_decoder.py:65-104
    "freeze_extractor": (TRIPLED, dict(freeze_extractor=True)),                                      # :24-27
    "skip_1x1": (TRIPLED, dict(depth_skip_type="1x1", color_skip_type="1x1", color_skip_layers=[True] * 4)),   # :434, :451, :485
    "all_layers_disentangled": (TRIPLED, dict(disentangle_layers=[True] * 5, depth_skip_type="use_half", color_skip_type="use_half",
                                              color_skip_layers=[True] * 4)),                        # :419-426
    "no_automask_no_disp_norm": (TRIPLED, dict(automask=False, disp_norm=False)),                    # the stereo configs' switches, :101-131
    "no_image_reconstruction": (TRIPLED, dict(img_reconstruct_weight=0)),                            # :80-91 skipped
    # attention gates on the depth skips (:410-436; this build: mono/model/attention.py): channel, pixel, both-moments, and in front
    # of a disentangled half / a learned (1x1 conv + BatchNorm + ELU) half
    "skip_channel_attention": (TRIPLED, dict(depth_skip_type="ca")),
    "skip_pixel_attention": (TRIPLED, dict(depth_skip_type="pa")),
    "skip_moment_attention_all_levels_split": (TRIPLED, dict(depth_skip_type="asca", disentangle_layers=[True] * 5,
                                                             color_skip_type="use_half", color_skip_layers=[True] * 4)),
    "skip_attention_learned_halves": (TRIPLED, dict(depth_skip_type="ca", depth_disentangle_type="conv",
                                                    disentangle_layers=[False, True, False, True, True])),
    # the stereo pair as a fourth frame with the rig's fixed transform instead of a predicted pose (mono_fm_joint/net.py:164-179, :186-189;
    # cfg_kitti_fm_refine.py: frame_ids [0, -1, 1, 's'] switches auto-mask and disparity normalisation off)
    "stereo_frame_fm": ("mono_fm", dict(frame_ids=[0, -1, 1, "s"], automask=False, disp_norm=False)),
    "stereo_frame_tripled": (TRIPLED, dict(frame_ids=[0, -1, 1, "s"], automask=False, disp_norm=False)),
    # (cond_encoder with a disentangled last layer fails inside the REFERENCE itself: encoder.py:46 adds 512 and 256 channels)
    "cond_encoder": (COLORIZE, dict(cond_encoder=True, disentangle_layers=[False] * 5)),             # :296-299, :364-367
}


@pytest.mark.parametrize("case", sorted(REFERENCE_OPTION_SETS))
def test_model_switches_match_the_reference(case):
    name, overrides = REFERENCE_OPTION_SETS[case]
    _compare(name, overrides)


@pytest.mark.parametrize("shape", [(1, 128, 192), (3, 96, 320)])
def test_other_batch_and_image_shapes_match_the_reference(shape):
    """One image per step / a wide 1:3.3 frame like KITTI's (the default cases run 2 x 96 x 128)."""
    _compare(TRIPLED, {}, shape)


def _compare(name, overrides, shape=(2, 96, 128)):
    import tripled_amd  # noqa: F401
    classes, NoiseTap, cuda_orig = _reference_classes()
    try:
        from mono.model import MONO
        from oracle.backend import OracleLossBackend
        B, H, W = shape
        torch.manual_seed(3)
        ref = classes[name](Opt(_options(name, B, H, W), **overrides))
        mine = MONO.module_dict[name](Opt(_options(name, B, H, W), **overrides))
        missing = mine.load_state_dict(ref.state_dict(), strict=True)   # identical checkpoint keys
        assert not missing.missing_keys and not missing.unexpected_keys
        # ... and the same parameter ORDER: a checkpoint's optimiser state is indexed by position in model.parameters()
        # (torch.optim state_dict), so resuming a reference run needs the registration order, not just the names
        assert [n for n, _ in mine.named_parameters()] == [n for n, _ in ref.named_parameters()]
        assert list(mine.state_dict()) == list(ref.state_dict())          # (this build's extra buffers are non-persistent)
        mine.set_loss_backend(OracleLossBackend())
        for m in (ref, mine):
            m.train()
            m.DepthDecoder.do.eval()          # dropout is the one RNG consumer we do not replay
        stereo = "s" in overrides.get("frame_ids", ())
        ref_in = _inputs(B, H, W, stereo)
        with NoiseTap() as tap:
            ref_out, ref_loss = ref(ref_in)
        draws = list(tap.draws)
        mine.set_noise_source(lambda shape, device: draws.pop(0))
        out, loss = mine(_inputs(B, H, W, stereo))
        assert [str(k) for k in loss] == [str(k) for k in ref_loss]
        for k in ref_loss:
            a, b = loss[k].mean(), ref_loss[k].mean()
            assert abs(float(a) - float(b)) < 1e-6 + 1e-4 * abs(float(b)), (k, float(a), float(b))
        for s in range(4):
            assert float((out[("disp", 0, s)] - ref_out[("disp", 0, s)]).abs().max()) < 1e-5
            assert (out[("min_index", s)] == ref_out[("min_index", s)]).float().mean() > 0.995
        tot = sum(v.mean() for v in loss.values())
        ref_tot = sum(v.mean() for v in ref_loss.values())
        tot.backward()
        ref_tot.backward()
        ref_params = dict(ref.named_parameters())
        checked = 0
        for n, p in mine.named_parameters():
            rg = ref_params[n].grad
            if rg is None:
                assert p.grad is None or float(p.grad.abs().max()) == 0.0, n
                continue
            scale = float(rg.abs().max())
            assert float((p.grad - rg).abs().max()) <= 2e-3 * scale + 1e-9, n
            checked += 1
        assert checked > 100
        # inference (model.eval(): running BatchNorm statistics, which the training pass above has just updated on both sides;
        # the forward returns the outputs only -- reference mono_fm_joint_inpaint/net.py:477-499, mono_fm/net.py:53-66)
        ref.eval()
        mine.eval()
        with torch.no_grad():
            ref_eval, my_eval = ref(_inputs(B, H, W, stereo)), mine(_inputs(B, H, W, stereo))
        assert isinstance(my_eval, dict) and isinstance(ref_eval, dict)
        for s in range(4):
            assert float((my_eval[("disp", 0, s)] - ref_eval[("disp", 0, s)]).abs().max()) < 1e-5, s
    finally:
        torch.Tensor.cuda = cuda_orig


@pytest.mark.parametrize("name", ["mono_fm", COLORIZE])
def test_initialisation_follows_the_reference_schemes(name):
    """Training from scratch starts from the same distributions: every tensor the reference initialises to a constant (BatchNorm
    scales and shifts, running statistics, zeroed biases) is that constant here, and every random tensor has the reference's
    standard deviation (kaiming / PyTorch-default fans; sampling noise of two independent draws allowed: 8 % on >= 2048 elements).
    Same seed -> same DRAWS holds only up to the first sub-network built twice by the reference's class hierarchy (the parent
    constructor's DepthDecoder / the 3-channel stem the pose encoder replaces consume generator state): checked for the depth
    encoder, which both build first."""
    import tripled_amd  # noqa: F401
    classes, _, cuda_orig = _reference_classes()
    try:
        from mono.model import MONO
        ov = dict(depth_num_layers=50, extractor_num_layers=50)
        torch.manual_seed(7)
        ref = classes[name](Opt(_options(name, 2, 96, 128), **ov))
        torch.manual_seed(7)
        mine = MONO.module_dict[name](Opt(_options(name, 2, 96, 128), **ov))
        sa, sb = ref.state_dict(), mine.state_dict()
        assert list(sa) == list(sb)
        random_tensors = 0
        for k in sa:
            x, y = sa[k].float(), sb[k].float()
            assert x.shape == y.shape, k
            if k.startswith("DepthEncoder."):
                assert torch.equal(x, y), k                       # same seed, same draws
                continue
            if x.numel() == 1:
                if not sa[k].is_floating_point():
                    assert torch.equal(x, y), k                   # num_batches_tracked
                continue                                          # (a one-element random bias: nothing to compare)
            if float(x.std()) == 0.0:
                assert torch.equal(x, y), k                       # constants
                continue
            if x.numel() >= 2048:
                ratio = float(y.std()) / float(x.std())
                assert abs(ratio - 1) < 0.08, (k, float(x.std()), float(y.std()))
                assert abs(float(y.mean()) - float(x.mean())) < 0.1 * float(x.std()), k
                random_tensors += 1
        assert random_tensors > 100
    finally:
        torch.Tensor.cuda = cuda_orig
