"""MI355X-native 2D->3D mask projection, multi-view fusion and refinement.

Drop-in for the hot path of Beyond-Fixed-Forms (reference tools/projection_2d_to_3d.py and
tools/refinement.py): same config.yaml keys, same mask_2d / stage-1 / output dict contracts,
the inner loops run as hand-written HIP kernels for gfx950 behind a C ABI (include/bff_hip.h).

There is no CPU fallback: every compute entry point raises if libbff_hip.so is missing.
"""

import os as _os

# The scene pipeline keeps PIPELINE_DEPTH scenes in flight, one HIP stream each (pipeline.PIPELINE_DEPTH).  The HIP
# runtime multiplexes a process's streams onto 4 hardware queues unless told otherwise, and two streams on one queue
# run strictly one after the other; it reads this when it initialises (the first HIP call of the process), so the
# package has to be imported before anything touches the GPU for the setting to take effect (measured on config 2:
# 4 scenes in flight, 1.31 -> 1.02 ms per scene with 8 queues; back to 1.31 with 4).
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from .config import Config, load_config  # noqa: F401,E402

__all__ = ["Config", "load_config"]
