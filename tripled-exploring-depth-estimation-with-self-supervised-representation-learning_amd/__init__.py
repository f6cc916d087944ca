"""MI355X-native self-supervised depth training path (TripleD hot path).

The directory name is fixed by the build contract and is not a Python identifier; import it
through the alias module at the repo root (``import tripled_amd``), which loads this package
and puts the host-side mirror of the reference interface (``mono``, and the minimal ``mmcv``
shim when the real mmcv is absent) on ``sys.path``.

Layout
  csrc/     hand-written HIP kernels for gfx950 + the C ABI (include/tripled_hip.h)
  lib/      libtripled_hip.so (built in-tree by ``__graft_entry__.build()`` / ``make -C csrc``)
  native.py ctypes binding of the C ABI (fails loudly when the library is missing)
  ops.py    torch.autograd.Functions over the C ABI (the fused loss ops)
  hostside/ ``mono`` (models, apis, core, datasets) and the ``mmcv`` shim
"""
import os
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
HOSTSIDE = os.path.join(PKG_DIR, "hostside")


# ROCm 7.2's hipGraph AQL-packet-capture fast path replays a single-stream capture of the ~4 000-node training
# step wrongly (garbage loss terms / NaN gradients from the second replay on, tools/diag_capture.py, DESIGN.md
# section 6); the general replay path is correct.  The flag is read when the HIP runtime initialises, so it is
# set on import of this package (before any HIP call); an explicit setting in the environment wins.
def _packet_capture_off_at_hip_init():
    preset = os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE")
    if preset is not None:
        return preset == "0"
    torch_mod = sys.modules.get("torch")
    hip_up = bool(torch_mod is not None and torch_mod.cuda.is_initialized())
    os.environ["DEBUG_CLR_GRAPH_PACKET_CAPTURE"] = "0"
    return not hip_up       # set now; it only takes effect if the runtime has not started yet


# step.capture_step refuses a single-stream capture when this is False
PACKET_CAPTURE_OFF_AT_HIP_INIT = _packet_capture_off_at_hip_init()
# dmabuf IPC (what the host driver supports): RCCL's intra-node transport needs it before the first HIP call
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def _activate():
    if HOSTSIDE not in sys.path:
        sys.path.insert(0, HOSTSIDE)


_activate()
