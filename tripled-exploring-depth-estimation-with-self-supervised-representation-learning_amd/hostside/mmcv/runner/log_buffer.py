from collections import OrderedDict

import numpy as np
import torch


class LogBuffer:
    """Windowed averages of the logged scalars.  Values may arrive as 0-d device tensors: they
    are kept as-is and only read back (one stacked transfer) when an average is requested, so a
    training iteration does not synchronise the device for logging."""

    def __init__(self):
        self.val_history = OrderedDict()
        self.n_history = OrderedDict()
        self.output = OrderedDict()
        self.ready = False

    def clear(self):
        self.val_history.clear()
        self.n_history.clear()
        self.clear_output()

    def clear_output(self):
        self.output.clear()
        self.ready = False

    def update(self, vars, count=1):
        assert isinstance(vars, dict)
        for key, var in vars.items():
            self.val_history.setdefault(key, []).append(var)
            self.n_history.setdefault(key, []).append(count)

    def _materialise(self, n):
        pending = []
        for key, hist in self.val_history.items():
            lo = 0 if n <= 0 else max(0, len(hist) - n)
            for i in range(lo, len(hist)):
                if isinstance(hist[i], torch.Tensor):
                    pending.append((key, i))
        if pending:
            flat = torch.stack([self.val_history[k][i].detach().float().reshape(()) for k, i in pending]).cpu().tolist()
            for (k, i), v in zip(pending, flat):
                self.val_history[k][i] = v

    def average(self, n=0):
        """Average the latest n values (all when n == 0), weighted by sample counts."""
        assert n >= 0
        self._materialise(n)
        for key in self.val_history:
            values = np.array(self.val_history[key][-n:], dtype=np.float64)
            nums = np.array(self.n_history[key][-n:], dtype=np.float64)
            self.output[key] = float(np.sum(values * nums) / np.sum(nums))
        self.ready = True
