"""KITTI raw frame-triplet datasets (reference: mono/datasets/mono_dataset.py:40-201,
mono/datasets/kitti_dataset.py:121-202), written without torchvision: PIL does the decoding, the
LANCZOS resize (the reference's Image.ANTIALIAS) and the colour jitter (the same PIL ImageEnhance /
HSV-shift primitives torchvision's ColorJitter applies to PIL images, in a random order).

Sample contract (what the models consume): ("color", f, 0) and ("color_aug", f, 0) float [3,H,W] in
[0,1] for every f in frame_idxs, "K" / "inv_K" [4,4] (normalised intrinsics scaled by W, H; inv_K =
pinv(K)), ("mask", 0, 0) uint8 [3,H,W] for the in-painting variant, "stereo_T" when 's' is a frame id,
"gt_depth" in validation.  Split lists are plain text: "<folder> <frame_index> <l|r>" per line.
"""
import os
import random

import numpy as np
import torch
from PIL import Image, ImageEnhance
from torch.utils.data import Dataset

LANCZOS = getattr(Image, "Resampling", Image).LANCZOS
FLIP = getattr(Image, "Transpose", Image).FLIP_LEFT_RIGHT


def read_split(split, which, split_dir=None):
    """Lines of <split_dir>/<split>/<which>_files.txt (default: mono/datasets/splits next to this file)."""
    root = split_dir or os.path.join(os.path.dirname(__file__), "splits")
    path = os.path.join(root, split, "{}_files.txt".format(which))
    with open(path) as f:
        return f.read().splitlines()


def pil_loader(path):
    with open(path, "rb") as f:
        with Image.open(f) as img:
            return img.convert("RGB")


def to_tensor(img):
    """PIL RGB -> float [3,H,W] in [0,1] (torchvision ToTensor semantics)."""
    return to_uint8(img).float().div_(255.0)


def to_uint8(img):
    """PIL RGB -> uint8 [3,H,W] (the 'uint8' wire format: 3 bytes per pixel instead of 2 x 12)."""
    arr = np.array(img, dtype=np.uint8)
    return torch.from_numpy(arr).permute(2, 0, 1).contiguous()


def _shift_hue(img, hue_factor):
    h, s, v = img.convert("HSV").split()
    arr = np.asarray(h, dtype=np.uint8).astype(np.int16)
    arr = ((arr + int(hue_factor * 255)) % 256).astype(np.uint8)
    return Image.merge("HSV", (Image.fromarray(arr, "L"), s, v)).convert("RGB")


class ColorJitter:
    """One draw of (order, brightness, contrast, saturation, hue), applied identically to every image it is
    called on -- the reference's stated intent for the frames of one sample (mono_dataset.py:89-95)."""

    def __init__(self, brightness, contrast, saturation, hue):
        self.order = torch.randperm(4).tolist()
        self.factors = [float(torch.empty(1).uniform_(*brightness)), float(torch.empty(1).uniform_(*contrast)),
                        float(torch.empty(1).uniform_(*saturation)), float(torch.empty(1).uniform_(*hue))]

    def as_row(self):
        """(enabled, op0..op3, brightness, contrast, saturation, hue): the parameter row td_color_jitter consumes."""
        return torch.tensor([1.0] + [float(o) for o in self.order] + self.factors, dtype=torch.float32)

    def __call__(self, img):
        for op in self.order:
            if op == 0:
                img = ImageEnhance.Brightness(img).enhance(self.factors[0])
            elif op == 1:
                img = ImageEnhance.Contrast(img).enhance(self.factors[1])
            elif op == 2:
                img = ImageEnhance.Color(img).enhance(self.factors[2])
            else:
                img = _shift_hue(img, self.factors[3])
        return img


class MonoDataset(Dataset):
    brightness, contrast, saturation, hue = (0.8, 1.2), (0.8, 1.2), (0.8, 1.2), (-0.1, 0.1)

    def __init__(self, data_path, filenames, height, width, frame_idxs, cfg=None, is_train=False, img_ext=".jpg",
                 gt_depth_path=None):
        super().__init__()
        self.data_path, self.filenames = data_path, filenames
        self.height, self.width = height, width
        self.frame_idxs, self.is_train, self.img_ext = list(frame_idxs), is_train, img_ext
        self.cfg = cfg if cfg is not None else {}
        self.loader = pil_loader
        self.gt_depth_path = gt_depth_path
        self.flag = np.zeros(len(self), dtype=np.int64)        # single aspect-ratio group for the samplers
        self.gt_depths = None
        if not is_train and gt_depth_path is not None and os.path.exists(str(gt_depth_path)):
            # the reference loads this archive with allow_pickle=True; object arrays are refused here
            self.gt_depths = np.load(gt_depth_path, allow_pickle=False)["data"]

    def __len__(self):
        return len(self.filenames)

    def resize(self, img):
        return img.resize((self.width, self.height), LANCZOS)

    def get_color(self, folder, frame_index, side, do_flip):
        raise NotImplementedError

    def postprocess(self, inputs):
        """Hook for subclasses (in-painting masks)."""

    def __getitem__(self, index):
        inputs = {}
        do_color_aug = self.is_train and random.random() > 0.5
        do_flip = self.is_train and random.random() > 0.5
        parts = self.filenames[index].split()
        folder = parts[0]
        frame_index = int(parts[1]) if len(parts) == 3 else 0
        side = parts[2] if len(parts) == 3 else None
        if self.gt_depths is not None:
            inputs["gt_depth"] = self.gt_depths[index]
        jitter = ColorJitter(self.brightness, self.contrast, self.saturation, self.hue) if do_color_aug else None
        # wire = "uint8": frames travel as bytes, ToTensor and the colour jitter run on the device
        # (mono.datasets.device_expand / csrc/td_augment.hip); "float32" is the reference's format
        u8_wire = self.cfg.get("wire", "float32") == "uint8"
        if u8_wire:
            inputs["aug"] = jitter.as_row() if jitter is not None else torch.zeros(9)
        for i in self.frame_idxs:
            if i == "s":
                img = self.get_color(folder, frame_index, {"r": "l", "l": "r"}[side], do_flip)
            else:
                try:
                    img = self.get_color(folder, frame_index + i, side, do_flip)
                except (FileNotFoundError, OSError):            # sequence boundary: repeat the centre frame
                    img = self.get_color(folder, frame_index, side, do_flip)
            img = self.resize(img)
            if u8_wire:
                inputs[("color_u8", i)] = to_uint8(img)
                continue
            inputs[("color", i, 0)] = to_tensor(img)
            inputs[("color_aug", i, 0)] = to_tensor(jitter(img)) if jitter is not None else inputs[("color", i, 0)].clone()
        K = self.K.copy()
        K[0, :] *= self.width
        K[1, :] *= self.height
        inputs["K"] = torch.from_numpy(K)
        inputs["inv_K"] = torch.from_numpy(np.linalg.pinv(K))
        self.postprocess(inputs)
        if "s" in self.frame_idxs:
            stereo_T = np.eye(4, dtype=np.float32)
            stereo_T[0, 3] = (-1 if side == "l" else 1) * (-1 if do_flip else 1) * 0.015
            inputs["stereo_T"] = torch.from_numpy(stereo_T)
        return inputs


class KITTIDataset(MonoDataset):
    K = np.array([[0.58, 0, 0.5, 0], [0, 1.92, 0.5, 0], [0, 0, 1, 0], [0, 0, 0, 1]], dtype=np.float32)
    full_res_shape = (1242, 375)
    side_map = {"2": 2, "3": 3, "l": 2, "r": 3}

    def get_image_path(self, folder, frame_index, side):
        name = "{:010d}{}".format(frame_index, self.img_ext)
        return os.path.join(self.data_path, folder, "image_0{}/data".format(self.side_map[side]), name)

    def get_color(self, folder, frame_index, side, do_flip):
        img = self.loader(self.get_image_path(folder, frame_index, side))
        return img.transpose(FLIP) if do_flip else img

    def check_depth(self):
        parts = self.filenames[0].split()
        velo = os.path.join(self.data_path, parts[0], "velodyne_points/data/{:010d}.bin".format(int(parts[1])))
        return os.path.isfile(velo)


class KITTIRAWDataset(KITTIDataset):
    pass


class KITTIInpaintDataset(KITTIDataset):
    """Adds ("mask", 0, 0): ones with `erase_count` zeroed `erase_shape` rectangles (one centred square when
    erase_count == 1), reference kitti_dataset.py:167-182."""

    def postprocess(self, inputs):
        image = inputs[("color", 0, 0)] if ("color", 0, 0) in inputs else inputs[("color_u8", 0)]
        eh, ew = self.cfg["erase_shape"]
        count = self.cfg["erase_count"]
        mask = torch.ones(image.shape, dtype=torch.uint8)
        if count == 1:
            off = int((image.shape[1] - eh) / 2)
            mask[:, off:off + eh, off:off + eh] = 0
        else:
            for _ in range(count):
                row = int(torch.randint(0, image.shape[1] - eh - 1, (1,)))
                col = int(torch.randint(0, image.shape[2] - ew - 1, (1,)))
                mask[:, row:row + eh, col:col + ew] = 0
        inputs[("mask", 0, 0)] = mask
