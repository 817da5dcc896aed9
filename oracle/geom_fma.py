"""ctypes binding of oracle/geom_fma.c.  TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libgeom_fma.so")
_lib = None


def build():
    subprocess.run(["make", "-C", _HERE], check=True, stdout=subprocess.DEVNULL)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
        _lib.bff_ref_view.restype = None
    return _lib


def view(xyz, inv_pose, k33, depth, thresh=0.08):
    """-> (pts_cam (N,3) f64, pix (N,2) i64, vis (N,) bool) with the explicit fma chain."""
    xyz = np.ascontiguousarray(xyz[:, :3], dtype=np.float64)
    inv_pose = np.ascontiguousarray(inv_pose, dtype=np.float64)
    k33 = np.ascontiguousarray(k33, dtype=np.float64)
    depth = np.ascontiguousarray(depth, dtype=np.float32)
    n = xyz.shape[0]
    pts = np.empty((n, 3), np.float64)
    pix = np.empty((n, 2), np.int64)
    vis = np.empty(n, np.uint8)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    lib().bff_ref_view(p(xyz), ctypes.c_int64(n), p(inv_pose), p(k33), p(depth),
                       ctypes.c_int(depth.shape[0]), ctypes.c_int(depth.shape[1]), ctypes.c_double(thresh),
                       p(pts), p(pix), p(vis))
    return pts, pix, vis.astype(bool)
