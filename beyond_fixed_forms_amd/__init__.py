"""MI355X-native 2D->3D mask projection, multi-view fusion and refinement.

Drop-in for the hot path of Beyond-Fixed-Forms (reference tools/projection_2d_to_3d.py and
tools/refinement.py): same config.yaml keys, same mask_2d / stage-1 / output dict contracts,
the inner loops run as hand-written HIP kernels for gfx950 behind a C ABI (include/bff_hip.h).

There is no CPU fallback: every compute entry point raises if libbff_hip.so is missing.
"""

from .config import Config, load_config  # noqa: F401

__all__ = ["Config", "load_config"]
