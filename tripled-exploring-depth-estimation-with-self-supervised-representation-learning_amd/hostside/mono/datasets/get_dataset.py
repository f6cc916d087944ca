"""Dataset factory (reference: mono/datasets/get_dataset.py:73-103).  KITTI raw data is not
available offline; 'synthetic' (and any name when cfg.in_path does not exist and
cfg.allow_synthetic is set) yields SyntheticTripletDataset with the same sample contract."""
import os

from .synthetic import SyntheticTripletDataset


def get_dataset(cfg, training=True):
    name = cfg["name"]
    in_path = cfg.get("in_path", None)
    have_data = in_path is not None and os.path.isdir(str(in_path))
    if name == "synthetic" or (not have_data and cfg.get("allow_synthetic", False)):
        return SyntheticTripletDataset(
            length=cfg.get("synthetic_length", 256) if training else cfg.get("synthetic_val_length", 8),
            height=cfg["height"], width=cfg["width"],
            frame_ids=cfg["frame_ids"] if training else [0],
            erase_shape=cfg.get("erase_shape", (16, 16)), erase_count=cfg.get("erase_count", 16),
            with_mask=True, with_gt=not training)
    if name in ("kitti", "kitti_inpaint"):
        from .kitti_dataset import KITTIInpaintDataset, KITTIRAWDataset, read_split
        cls = KITTIInpaintDataset if name == "kitti_inpaint" else KITTIRAWDataset
        filenames = read_split(cfg["split"], "train" if training else "val", cfg.get("split_dir", None))
        return cls(in_path, filenames, cfg["height"], cfg["width"], cfg["frame_ids"] if training else [0],
                   is_train=training, img_ext=".png" if cfg.get("png", True) else ".jpg",
                   gt_depth_path=cfg.get("gt_depth_path", None), cfg=cfg)
    raise NotImplementedError("dataset '%s' is outside the KITTI depth training path" % name)
