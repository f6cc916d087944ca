"""Host-side mirror of the reference's ``mono`` package for the training hot path: same module
paths, class names, constructor signatures, config keys and state_dict keys, so configs such as
cfg_kitti_tripleD.py and reference checkpoints drop in.  The loss hot path is routed to the
hand-written HIP kernels (tripled_amd.ops); the convolutional networks are PyTorch-ROCm modules."""
