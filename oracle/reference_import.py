"""Import the real reference helpers from /root/reference.  TEST INFRASTRUCTURE ONLY.

Works only in the build container (the reference does not travel to the GPU box).  The hot-path
modules import packages that are not installed here (cv2, munch, configs, open3d, clip); none
of them is touched by the helper functions we call, so empty stub modules stand in for them.
Nothing is written under /root/reference (bytecode writing is disabled).
"""
from __future__ import annotations

import os
import sys
import types

REFERENCE_ROOT = "/root/reference"


def available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "tools", "projection_2d_to_3d.py"))


def load():
    """Returns (projection_2d_to_3d, refinement, rle_encode_decode) reference modules."""
    if not available():
        raise RuntimeError("reference tree not present (expected in the build container only)")
    sys.dont_write_bytecode = True
    for name in ("cv2", "munch", "configs", "open3d", "clip"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["configs"].config = types.SimpleNamespace(min_aggragated_masks=2)
    sys.modules["munch"].Munch = dict
    cwd = os.getcwd()
    for p in (REFERENCE_ROOT, os.path.join(REFERENCE_ROOT, "tools")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.chdir(REFERENCE_ROOT)           # utils/rle_encode_decode.py does sys.path.append("./")
    try:
        import projection_2d_to_3d as proj
        import refinement as refi
        from utils import rle_encode_decode as rle
    finally:
        os.chdir(cwd)
    return proj, refi, rle
