# FeatDepth (mono_fm) on KITTI -- BASELINE config #1 uses it with ResNet18 everywhere, two
# 192x640 triplets, on the CPU.  Keys follow the reference's cfg_kitti_fm.py.
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _common import runtime, schedule

HEIGHT, WIDTH, IMGS_PER_GPU = 192, 640, 2
FRAME_IDS = [0, -1, 1]
DEPTH_LAYERS, POSE_LAYERS, FEAT_LAYERS = 18, 18, 18
STEREO = "s" in FRAME_IDS

data = dict(
    name="kitti", split="exp", height=HEIGHT, width=WIDTH, frame_ids=FRAME_IDS,
    in_path=os.environ.get("KITTI_RAW", "/data/kitti_raw"),
    gt_depth_path=os.environ.get("KITTI_GT_DEPTHS", "/data/kitti_raw/gt_depths.npz"),
    png=True, stereo_scale=STEREO, allow_synthetic=os.environ.get("TD_ALLOW_SYNTHETIC", "0") == "1", synthetic_length=2 * 32,
)

model = dict(
    name="mono_fm",
    depth_num_layers=DEPTH_LAYERS, pose_num_layers=POSE_LAYERS, extractor_num_layers=FEAT_LAYERS,
    frame_ids=FRAME_IDS, imgs_per_gpu=IMGS_PER_GPU, height=HEIGHT, width=WIDTH, scales=[0, 1, 2, 3],
    min_depth=0.1, max_depth=100.0,
    depth_pretrained_path=None, pose_pretrained_path=None, extractor_pretrained_path=None,
    automask=not STEREO, disp_norm=not STEREO, perception_weight=1e-3, smoothness_weight=1e-3,
)

imgs_per_gpu = IMGS_PER_GPU
workers_per_gpu = 4
globals().update(schedule(lr=1e-4, steps=[20, 30], total_epochs=40))
globals().update(runtime())
