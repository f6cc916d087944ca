"""Independent sub-networks of one training step on concurrent HIP streams.

The TripleD step holds three chains that meet only in the losses: the depth encoder / decoder (+ colour decoder), the pose
network on the frame pairs, and the feature auto-encoder on the target image (reference: mono_fm_joint_inpaint/net.py:488-514
runs them one after the other on the one CUDA stream).  Most of their kernels are latency-bound launches of <= 1 block per CU
(the 14-20 us floor of the small GEMMs, the 3-6 us element-wise passes), so issuing the chains on separate
streams lets the hardware fill one chain's tails and gaps with the other's work.  ``Branch`` is the fork / join:

    with Branch(device, 0) as b:          # side stream 0 waits for the work already queued on the current stream
        feats = encoder(img)              # ... queued on the side stream
    ...                                   # the current stream goes on meanwhile
    b.join(feats)                         # the current stream waits for the branch; its tensors are marked as used here

The autograd engine replays every backward node on the stream its forward ran on and orders the streams itself, so the backward
pass forks the same way.  Inside a HIP-graph capture the side streams join the capture at the fork and the chains become
parallel branches of the graph.  ``join`` calls ``record_stream`` on the tensors that cross, which keeps the caching allocator
from handing their blocks back to the side stream while the consumer may still read them.
TD_BRANCH_STREAMS=0 turns the forks off (every chain on the current stream, in the reference's order)."""
import os

import torch

ENABLED = os.environ.get("TD_BRANCH_STREAMS", "1") != "0"
_side = {}


def enabled(t):
    return ENABLED and t.is_cuda


def side_stream(device, idx, avoid=()):
    """The cached side stream ``idx`` of ``device`` -- never one of ``avoid`` and never another index's stream.
    torch.cuda.Stream() hands out a pool of 32 HIP streams per device round-robin, so a stream created here can BE the stream
    the caller is on (a capture stream created 32 creations later): forking onto the current stream would record an event on
    a stream and make the same stream wait for it, and the "branch" would run in line with its caller.  (Found while looking
    for the cause of a hipGraphLaunch segfault ~300 tests into the GPU suite of round 4; a minimal self-wait capture replays
    fine -- tools/diag_self_wait_capture.py --, so this defect is fixed but not shown to be that cause: DESIGN.md section 13.)"""
    index = device.index if device.index is not None else torch.cuda.current_device()
    key = (index, idx)
    taken = {int(a.cuda_stream) for a in avoid} | {int(v.cuda_stream) for k, v in _side.items() if k[0] == index and k != key}
    cached = _side.get(key)
    tries = 0
    while cached is None or int(cached.cuda_stream) in taken or int(cached.cuda_stream) == 0:
        cached = torch.cuda.Stream(device=index)
        tries += 1
        if tries > 64:
            raise RuntimeError("no free HIP stream for branch %d on device %d" % (idx, index))
    _side[key] = cached
    return cached


def _tensors(tree):
    if torch.is_tensor(tree):
        yield tree
    elif isinstance(tree, dict):
        for v in tree.values():
            yield from _tensors(v)
    elif isinstance(tree, (list, tuple)):
        for v in tree:
            yield from _tensors(v)


class Branch:
    def __init__(self, device, idx):
        self.stream = side_stream(device, idx, avoid=(torch.cuda.current_stream(),))
        self.scope = None

    def __enter__(self):
        self.stream.wait_stream(torch.cuda.current_stream())
        self.scope = torch.cuda.stream(self.stream)
        self.scope.__enter__()
        return self

    def __exit__(self, *exc):
        self.scope.__exit__(*exc)
        return False

    def join(self, *trees):
        main = torch.cuda.current_stream()
        if int(main.cuda_stream) != int(self.stream.cuda_stream):      # (a join from another stream than the fork's)
            main.wait_stream(self.stream)
        for t in _tensors(trees):
            if t.is_cuda:
                t.record_stream(main)
