# TripleD on KITTI at the BASELINE resolution: 192x640, 12 images per GPU, ResNet50 depth and
# feature networks, ResNet18 pose network.  Same keys as the reference's cfg_kitti_tripleD.py
# (which ships 320x1024 / 3 per GPU -- change the three constants below to reproduce it).
# No pre-trained weights or KITTI data are available offline: the *_pretrained_path entries are
# None and the data section falls back to synthetic triplets when in_path does not exist AND TD_ALLOW_SYNTHETIC=1
# is set (offline smoke runs); otherwise a missing KITTI path is an error, as in the reference.
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _common import runtime, schedule

HEIGHT, WIDTH, IMGS_PER_GPU = 192, 640, 12
FRAME_IDS = [0, -1, 1]
DEPTH_LAYERS, POSE_LAYERS, FEAT_LAYERS = 50, 18, 50
STEREO = "s" in FRAME_IDS

data = dict(
    name="kitti_inpaint", split="exp", height=HEIGHT, width=WIDTH, frame_ids=FRAME_IDS,
    in_path=os.environ.get("KITTI_RAW", "/data/kitti_raw"),
    gt_depth_path=os.environ.get("KITTI_GT_DEPTHS", "/data/kitti_raw/gt_depths.npz"),
    png=True, stereo_scale=STEREO, erase_shape=[16, 16], erase_count=16,
    allow_synthetic=os.environ.get("TD_ALLOW_SYNTHETIC", "0") == "1", synthetic_length=12 * 64,
)

model = dict(
    name="mono_fm_joint_inpaint_disentangle",
    depth_num_layers=DEPTH_LAYERS, pose_num_layers=POSE_LAYERS, extractor_num_layers=FEAT_LAYERS,
    frame_ids=FRAME_IDS, imgs_per_gpu=IMGS_PER_GPU, height=HEIGHT, width=WIDTH, scales=[0, 1, 2, 3],
    min_depth=0.1, max_depth=100.0,
    depth_pretrained_path=None, pose_pretrained_path=None, extractor_pretrained_path=None,
    automask=not STEREO, disp_norm=not STEREO,
    dis=1e-3, cvt=1e-3, perception_weight=1e-3, smoothness_weight=1e-3, auto_res_weight=5e-3,
    disentangle_layers=[False, False, False, False, True], skip_connection_multiplier=1,
    depth_skip_type=None, color_skip_type=None, color_skip_layers=[False, False, False, False],
    depth_use_shuffle=False, depth_disentangle_type="use_half", freeze_extractor=False,
)

imgs_per_gpu = IMGS_PER_GPU
workers_per_gpu = 4
globals().update(schedule(lr=1e-4, steps=[10, 20], total_epochs=20))
globals().update(runtime(tensorboard=True))
