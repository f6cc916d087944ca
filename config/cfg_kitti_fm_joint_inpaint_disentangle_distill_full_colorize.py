# All auxiliary heads (in-painting auto-encoder + separate Lab colourisation network) --
# BASELINE config #5 at 8 images per GPU.  Keys follow the reference's config of the same name.
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _common import runtime, schedule

HEIGHT, WIDTH, IMGS_PER_GPU = 192, 640, 8
FRAME_IDS = [0, -1, 1]
DEPTH_LAYERS, POSE_LAYERS, FEAT_LAYERS, COLORIZE_LAYERS = 50, 18, 50, 50
STEREO = "s" in FRAME_IDS

data = dict(
    name="kitti_inpaint", split="exp", height=HEIGHT, width=WIDTH, frame_ids=FRAME_IDS,
    in_path=os.environ.get("KITTI_RAW", "/data/kitti_raw"),
    gt_depth_path=os.environ.get("KITTI_GT_DEPTHS", "/data/kitti_raw/gt_depths.npz"),
    png=True, stereo_scale=STEREO, erase_shape=[16, 16], erase_count=16,
    map_cfg=dict(alphas=[0.1, 0.4, 0.7, 1.0], blur_kernel_size=(9, 9), map_n=2),
    allow_synthetic=os.environ.get("TD_ALLOW_SYNTHETIC", "0") == "1", synthetic_length=8 * 64,
)

model = dict(
    name="mono_fm_joint_inpaint_disentangle_distill_sep_colorize",
    depth_num_layers=DEPTH_LAYERS, pose_num_layers=POSE_LAYERS, extractor_num_layers=FEAT_LAYERS,
    colorize_num_layers=COLORIZE_LAYERS,
    frame_ids=FRAME_IDS, imgs_per_gpu=IMGS_PER_GPU, height=HEIGHT, width=WIDTH, scales=[0, 1, 2, 3],
    min_depth=0.1, max_depth=100.0,
    depth_pretrained_path=None, pose_pretrained_path=None, extractor_pretrained_path=None,
    colorize_pretrained_path=None,
    automask=not STEREO, disp_norm=not STEREO,
    dis=1e-3, cvt=1e-3, perception_weight=1e-3, smoothness_weight=1e-3, auto_res_weight=5e-3,
    disentangle_layers=[False, False, False, False, False], skip_connection_multiplier=1,
    colorize_weight=5e-3, use_distill_mask=True, img_reconstruct_weight=1,
)

imgs_per_gpu = IMGS_PER_GPU
workers_per_gpu = 4
globals().update(schedule(lr=1e-4, steps=[20, 30], total_epochs=40))
globals().update(runtime())
