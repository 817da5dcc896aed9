"""Disk formats of the reference <-> in-memory scene inputs (the file-level drop-in boundary).

Reads exactly the files tools/projection_2d_to_3d.py reads (P:370-400, 422-436, 526-535) and
tools/refinement.py reads (R:172-193), writes what they write (P:630-634, R:422-428 and the scene
checkpoints P:320-334, R:41-55).  Depth decoding: the reference uses cv2.imread(IMREAD_UNCHANGED) /
1000 and cv2.resize (bilinear); cv2 is used when importable, otherwise PIL decodes the 16-bit PNG and
`resize_bilinear_f32` restates cv2's INTER_LINEAR (parity unpinned: no cv2 in the build container).
"""
from __future__ import annotations

import os

import numpy as np
import torch
import yaml

from .synthetic import SceneInputs

DEPTH_SCALE = 1000            # hard-coded at P:346


def _axis_taps(n_dst, n_src, axis):
    """2-tap tables of one axis with the coefficient arithmetic of OpenCV's resize (INTER_LINEAR, float32
    images; restated from memory of imgproc/resize.cpp -- cv2 is not in the build container: parity unpinned):
    scale = 1 / (n_dst / n_src) in double; the source coordinate is CAST TO FLOAT32 before its floor is
    subtracted (fx = (float)((dx + 0.5) * scale - 0.5); sx = floor(fx); fx -= sx);
    x axis: sx < 0 -> (sx, fx) = (0, 0); sx >= n_src - 1 -> (n_src - 1, 0), so the border columns are copied;
    y axis: the fraction is kept and only the two row indices are clamped into the image."""
    scale = 1.0 / (float(n_dst) / float(n_src))
    f = ((np.arange(n_dst, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
    i0 = np.floor(f).astype(np.int64)
    a = (f - i0.astype(np.float32)).astype(np.float32)
    if axis == "x":
        lo, hi = i0 < 0, i0 >= n_src - 1
        a[lo | hi] = 0.0
        i0 = np.where(lo, 0, np.where(hi, n_src - 1, i0))
        i1 = np.minimum(i0 + 1, n_src - 1)
    else:
        i1 = np.clip(i0 + 1, 0, n_src - 1)
        i0 = np.clip(i0, 0, n_src - 1)
    return i0, i1, a


def bilinear_taps(h_src, w_src, height, width):
    """Tap tables of the bilinear resize (x0, x1, ax, y0, y1, ay) as int32 / float32 arrays."""
    x0, x1, ax = _axis_taps(width, w_src, "x")
    y0, y1, ay = _axis_taps(height, h_src, "y")
    return (x0.astype(np.int32), x1.astype(np.int32), ax, y0.astype(np.int32), y1.astype(np.int32), ay)


def resize_bilinear_f32(img: np.ndarray, width: int, height: int) -> np.ndarray:
    """cv2.resize(img, (width, height)) with INTER_LINEAR for float32 input: half-pixel centres,
    edge clamp, horizontal then vertical 2-tap passes in float32."""
    h, w = img.shape
    if (h, w) == (height, width):
        return img.copy()
    img = img.astype(np.float32, copy=False)
    x0, x1, ax, y0, y1, ay = bilinear_taps(h, w, height, width)
    rows = img[:, x0] * (np.float32(1) - ax) + img[:, x1] * ax
    return rows[y0] * (np.float32(1) - ay)[:, None] + rows[y1] * ay[:, None]


def read_matrix_txt(path: str) -> np.ndarray:
    """np.loadtxt for the small whitespace-separated float matrices of a ScanNet scene (pose/<n>.txt, intrinsic_*.txt):
    same values (both parse every token with a correctly rounded decimal -> float64 conversion), ~10x less time --
    P:422 reads one such file per frame."""
    with open(path) as f:
        rows = [ln.split() for ln in f if ln.strip() and not ln.lstrip().startswith("#")]
    return np.array(rows, dtype=np.float64)


def load_depth(path: str, width: int, height: int) -> np.ndarray:
    """P:432-436: 16-bit PNG -> float32 metres -> (height, width)."""
    try:
        import cv2  # noqa: F401
        d = cv2.imread(path, cv2.IMREAD_UNCHANGED).astype(np.float32) / DEPTH_SCALE
        return cv2.resize(d, (width, height))
    except ImportError:
        from PIL import Image
        d = np.asarray(Image.open(path)).astype(np.float32) / DEPTH_SCALE
        return resize_bilinear_f32(d, width, height)


def load_depth_raw(path: str) -> np.ndarray:
    """The 16-bit PNG as stored (uint16 millimetres); scaling and resizing happen on the device."""
    from PIL import Image
    d = np.asarray(Image.open(path))
    if d.dtype != np.uint16:
        raise ValueError(f"{path}: expected a 16-bit depth PNG, got {d.dtype}")
    return d


def decode_depth_pngs(paths, out: np.ndarray = None, n_threads: int = 4) -> np.ndarray:
    """16-bit depth PNGs (P:431-433) -> uint16 [F][h][w]: the whole batch through the native decoder of libbff_host.so
    (inflate + the five PNG row filters on native threads, GIL released, written straight into `out` -- e.g. pinned
    staging -- when given); files it declines (another bit depth or colour type, interlaced, damaged, a different size)
    are decoded by PIL, which also decides whether such a file is an error."""
    import ctypes
    from .ingest import host_lib
    lib = host_lib()
    paths = [os.fspath(p) for p in paths]
    if not paths:
        return np.zeros((0, 0, 0), np.uint16) if out is None else out[:0]
    hw = (ctypes.c_int32 * 2)()
    if lib.bff_host_png_size(paths[0].encode(), ctypes.cast(hw, ctypes.c_void_p)) != 0:
        first = load_depth_raw(paths[0])                                   # PIL decides what the first file is
        hw[0], hw[1] = first.shape
    h, w = int(hw[0]), int(hw[1])
    if out is None:
        out = np.empty((len(paths), h, w), dtype=np.uint16)
    frames = out.reshape(-1)[:len(paths) * h * w].reshape(len(paths), h, w)
    status = np.zeros(len(paths), dtype=np.int32)
    got = lib.bff_host_decode_depth_pngs(paths, frames.ctypes.data, h, w, status.ctypes.data, int(n_threads))
    if got < 0:
        raise ValueError("bff_host_decode_depth_pngs: bad arguments")
    for i in np.flatnonzero(status):                                       # declined: the general decoder
        d = load_depth_raw(paths[i])
        if d.shape != (h, w):
            raise ValueError(f"{paths[i]}: depth frame of size {d.shape}, the scene's frames are {(h, w)}")
        frames[i] = d
    return frames


def load_scene(cfg, cls: str, scene_id: str, depth_on_device: bool = False, staging=None) -> SceneInputs:
    """Everything P:370-400 + the per-frame files of P:422-436 and P:526-563 for one scene.
    depth_on_device: keep the depth frames as raw uint16 (`SceneInputs.depths_raw`), decoded as one batch by the native
    PNG decoder; prepare_scene uploads them as they are and the sweep does /1000 + resize per point.  staging (an
    ingest.Staging, given by the loader thread that will upload the scene): the frames are decoded straight into its
    pinned "depth" buffer in upload order, so the upload needs no further copy."""
    scene_dir = os.path.join(cfg.scene_2d_dir, scene_id)
    cam_intr = read_matrix_txt(os.path.join(scene_dir, "intrinsic", "intrinsic_color.txt"))     # P:376
    points = np.load(os.path.join(cfg.scene_npy_dir, f"{scene_id}.npy"))                         # P:387
    # the mask_2d file is the user's own upstream output (segmentation_2d.py:500-504): a pickled list of
    # dicts holding numpy count arrays, loaded exactly as the reference does (P:396)
    mask_2d = torch.load(os.path.join(cfg.mask_2d_dir, cls, f"{scene_id}.pth"), weights_only=False)
    color_dir = os.path.join(scene_dir, "color")
    color_files = [f for f in os.listdir(color_dir) if f.endswith(".jpg")] if os.path.isdir(color_dir) else []
    from .scene import viewed_frame_ids
    # frames in upload order: mask frames in list order, then the frames only the detection-ratio sweep looks at
    need = list(dict.fromkeys(fr["frame_id"][:-4] for fr in mask_2d))
    if (not cfg.if_occurance_threshold) and cfg.if_detected_ratio_threshold:
        need = list(dict.fromkeys(need + viewed_frame_ids(color_files, cfg.downsample_ratio)))
    w, h = int(cfg.width_2d), int(cfg.height_2d)
    poses = {f: read_matrix_txt(os.path.join(scene_dir, "pose", f"{f}.txt")) for f in need}      # P:422
    if depth_on_device:
        paths = [os.path.join(scene_dir, "depth", f"{f}.png") for f in need]
        out = None
        if staging is not None and need:
            import ctypes
            from .ingest import host_lib
            hw = (ctypes.c_int32 * 2)()
            if host_lib().bff_host_png_size(paths[0].encode(), ctypes.cast(hw, ctypes.c_void_p)) == 0:
                staging.wait()                       # the previous scene's copy out of the pinned buffer is done
                nbytes = 2 * len(need) * int(hw[0]) * int(hw[1])
                out = staging.get("depth", nbytes).numpy()[:nbytes].view(np.uint16)
        frames = decode_depth_pngs(paths, out=out)
        scene = SceneInputs(scene_id=scene_id, points=points, cam_intr=cam_intr, poses=poses, depths={},
                            depths_raw={f: frames[i] for i, f in enumerate(need)},
                            mask_2d=mask_2d, color_files=color_files, height=h, width=w)
        if out is not None:
            scene.depth_staged = (staging, list(need))
        return scene
    depths = {f: load_depth(os.path.join(scene_dir, "depth", f"{f}.png"), w, h) for f in need}   # P:431-436
    return SceneInputs(scene_id=scene_id, points=points, cam_intr=cam_intr, poses=poses, depths=depths,
                       mask_2d=mask_2d, color_files=color_files, height=h, width=w)


def scene_checkpoint_file(stage: str, cls: str) -> str:
    """P:320-322 / R:41-43."""
    return f"checkpoints/{stage}_checkpoint_{cls}.yaml"


def read_scene_checkpoint(stage, cls):
    p = scene_checkpoint_file(stage, cls)
    if os.path.exists(p):
        with open(p) as f:
            return yaml.safe_load(f) or {}
    return {}


def write_scene_checkpoint(stage, cls, ckpt):
    p = scene_checkpoint_file(stage, cls)
    os.makedirs(os.path.dirname(p), exist_ok=True)       # the reference never creates it
    with open(p, "w") as f:
        yaml.safe_dump(ckpt, f)


def save_result(result: dict, out_dir: str, cls: str, scene_id: str):
    os.makedirs(os.path.join(out_dir, cls), exist_ok=True)
    torch.save(result, os.path.join(out_dir, cls, f"{scene_id}.pth"))
