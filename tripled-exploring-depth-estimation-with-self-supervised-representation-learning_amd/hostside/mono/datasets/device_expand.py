"""Device-side expansion of the 'uint8' wire format (cfg.data.wire = "uint8").

The loader ships ("color_u8", f) uint8 [B,3,H,W] per frame and "aug" [B,9] (one colour-jitter draw per sample, shared
by its frames as in the reference, mono/datasets/mono_dataset.py:89-95); one HIP launch pair turns them into the
("color", f, 0) / ("color_aug", f, 0) float tensors the models consume (csrc/td_augment.hip).  The host->device copy
is 3 bytes per pixel and frame instead of 24, and ToTensor + ColorJitter leave the loader workers."""
import torch


def has_uint8_frames(data):
    return isinstance(data, dict) and any(isinstance(k, tuple) and k and k[0] == "color_u8" for k in data)


def expand_device_batch(data):
    """In place: replaces the ("color_u8", f) / "aug" entries of a device-resident batch dict."""
    if not has_uint8_frames(data):
        return data
    from tripled_amd import native, ops
    frames = sorted((k for k in data if isinstance(k, tuple) and k[0] == "color_u8"), key=lambda k: str(k[1]))
    first = data[frames[0]]
    if not first.is_cuda:
        raise native.NativeLibraryError("the uint8 wire format is expanded by a HIP kernel: move the batch to the device first "
                                        "(or load with wire='float32')")
    B = first.shape[0]
    stacked = torch.cat([data[k] for k in frames], 0)                       # [F*B,3,H,W] uint8
    aug = data["aug"].float().repeat(len(frames), 1)                         # the frames of a sample share its draw
    color, color_aug = ops.color_jitter_expand(stacked, aug)
    for i, k in enumerate(frames):
        data[("color", k[1], 0)] = color[i * B:(i + 1) * B]
        data[("color_aug", k[1], 0)] = color_aug[i * B:(i + 1) * B]
        del data[k]
    del data["aug"]
    return data
