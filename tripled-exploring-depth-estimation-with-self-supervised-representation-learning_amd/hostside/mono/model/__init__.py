"""Model zoo of the training hot path.  (The reference's mono/model/__init__.py:9-10 imports a
``segmentation_base`` package that does not exist in its tree; nothing like that here.)"""
from .registry import MONO, SEGMENTATION  # noqa: F401
from .mono_fm.net import mono_fm  # noqa: F401
from .mono_fm_joint.net import mono_fm_joint  # noqa: F401
from .mono_fm_joint_inpaint.net import (mono_fm_joint_inpaint, mono_fm_joint_inpaint_disentangle,  # noqa: F401
                                        mono_fm_joint_inpaint_disentangle_distill_sep_colorize)
