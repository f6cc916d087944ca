import torch.nn as nn


class MMDataParallel(nn.Module):
    """``--launcher none`` wrapper.  The reference used single-process nn.DataParallel over
    cfg.gpus; this stack is one process per GPU, so the wrapper only keeps the ``.module``
    indirection and refuses more than one device."""

    def __init__(self, module, device_ids=None, output_device=None, dim=0):
        super().__init__()
        if device_ids is not None and len(device_ids) > 1:
            raise NotImplementedError("MMDataParallel over several GPUs in one process is not supported; "
                                      "use the pytorch launcher (one process per GPU, RCCL)")
        self.module = module
        self.device_ids = list(device_ids) if device_ids else []

    def forward(self, *inputs, **kwargs):
        return self.module(*inputs, **kwargs)
