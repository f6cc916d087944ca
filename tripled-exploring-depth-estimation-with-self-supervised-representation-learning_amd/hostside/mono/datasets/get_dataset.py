"""Dataset factory (reference: mono/datasets/get_dataset.py:73-103).  KITTI raw data is not
available offline; name 'synthetic' yields SyntheticTripletDataset with the same sample contract.  A KITTI
config whose in_path does not exist is an error (as in the reference) unless cfg.allow_synthetic opts in
(the shipped configs read it from TD_ALLOW_SYNTHETIC=1); the substitution is then logged as a warning and
recorded on the dataset object (``substituted_for``), which the checkpoint hook copies into the meta."""
import logging
import os

from .synthetic import SyntheticTripletDataset


def get_dataset(cfg, training=True):
    name = cfg["name"]
    in_path = cfg.get("in_path", None)
    have_data = in_path is not None and os.path.isdir(str(in_path))
    substitute = name != "synthetic" and not have_data and cfg.get("allow_synthetic", False)
    if name in ("kitti", "kitti_inpaint") and not have_data and not substitute:
        raise FileNotFoundError("dataset '%s': in_path %r is not a directory (set KITTI_RAW, or TD_ALLOW_SYNTHETIC=1 for an "
                                "offline smoke run on synthetic triplets)" % (name, in_path))
    if substitute:
        logging.getLogger().warning("in_path %r missing: using SYNTHETIC %s data instead of '%s' -- losses and "
                                    "validation metrics of this run say nothing about KITTI", in_path,
                                    "training" if training else "validation", name)
    if name == "synthetic" or substitute:
        ds = SyntheticTripletDataset(
            length=cfg.get("synthetic_length", 256) if training else cfg.get("synthetic_val_length", 8),
            height=cfg["height"], width=cfg["width"],
            frame_ids=cfg["frame_ids"] if training else [0],
            erase_shape=cfg.get("erase_shape", (16, 16)), erase_count=cfg.get("erase_count", 16),
            with_mask=True, with_gt=not training, wire=cfg.get("wire", "float32"), augment=training)
        ds.substituted_for = name if substitute else None
        return ds
    if name in ("kitti", "kitti_inpaint"):
        from .kitti_dataset import KITTIInpaintDataset, KITTIRAWDataset, read_split
        cls = KITTIInpaintDataset if name == "kitti_inpaint" else KITTIRAWDataset
        filenames = read_split(cfg["split"], "train" if training else "val", cfg.get("split_dir", None))
        return cls(in_path, filenames, cfg["height"], cfg["width"], cfg["frame_ids"] if training else [0],
                   is_train=training, img_ext=".png" if cfg.get("png", True) else ".jpg",
                   gt_depth_path=cfg.get("gt_depth_path", None), cfg=cfg)
    raise NotImplementedError("dataset '%s' is outside the KITTI depth training path" % name)
