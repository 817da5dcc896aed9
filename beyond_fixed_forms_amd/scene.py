"""Host-side preparation of one scene: reference disk formats -> HBM-resident kernel inputs.

Mirrors the loading part of the reference scene loop (tools/projection_2d_to_3d.py:376-400,
:422-436, :526-535): intrinsics `[:3,:3]`, cloud `[:, :3]` with a homogeneous 1, per-frame
`np.linalg.inv(pose)` (kept on the host in float64, exactly as the reference computes it), depth
images, and the RLE `mask_2d` list -- which is turned into flat run tables instead of being decoded
to dense (M,1,H,W) tensors.
"""
from __future__ import annotations

import dataclasses
from typing import List, Optional

import numpy as np
import torch

DEPTH_THRESH = 0.08      # hard-coded at the reference call sites projection_2d_to_3d.py:438,565


def runs_from_rles(rles, what="mask"):
    """list of {"length","counts"} -> (start int32[R], end int32[R], offs int32[n+1]) with the decode
    semantics of rle_decode_batch (rle_encode_decode.py:45-57): 1-based (start, len) pairs cast to
    int32, `mask[lo:hi] = 1` per run (python slicing clips hi at `length`).  Runs are returned
    0-based, clipped, non-empty, sorted and disjoint per mask (overlapping or unsorted inputs are
    merged, which leaves the decoded mask unchanged)."""
    n = len(rles)
    sizes = np.fromiter((np.asarray(r["counts"]).size for r in rles), dtype=np.int64, count=n)
    if np.any(sizes % 2):
        raise ValueError(f"{what} RLE with an odd number of counts")
    flat = (np.concatenate([np.asarray(r["counts"]).reshape(-1) for r in rles]) if n and sizes.sum()
            else np.zeros(0, np.int64)).astype(np.int32)
    length = np.fromiter((int(r["length"]) for r in rles), dtype=np.int64, count=n)
    start = flat[0::2].astype(np.int64) - 1
    end = start + flat[1::2].astype(np.int64)
    owner = np.repeat(np.arange(n), sizes // 2)
    if np.any(start < 0):
        raise ValueError(f"{what} RLE with start < 1 (negative python slice in the reference decoder)")
    end = np.minimum(end, length[owner])
    ok = end > start
    start, end, owner = start[ok], end[ok], owner[ok]
    same = owner[1:] == owner[:-1]
    if np.any(same & (start[1:] < end[:-1])):          # rare: normalise per mask
        s2, e2, o2 = [], [], []
        for g in np.unique(owner):
            sel = owner == g
            order = np.argsort(start[sel], kind="stable")
            s, e = start[sel][order], end[sel][order]
            cs, ce = [s[0]], [e[0]]
            for a, b in zip(s[1:], e[1:]):
                if a <= ce[-1]:
                    ce[-1] = max(ce[-1], b)
                else:
                    cs.append(a); ce.append(b)
            s2 += cs; e2 += ce; o2 += [g] * len(cs)
        start, end, owner = np.array(s2, np.int64), np.array(e2, np.int64), np.array(o2, np.int64)
    offs = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(np.bincount(owner, minlength=n), out=offs[1:])
    return start.astype(np.int32), end.astype(np.int32), offs.astype(np.int32)


def morton_order(xyz: np.ndarray) -> np.ndarray:
    """Permutation that sorts points along a 3-D Morton (Z-order) curve, 10 bits per axis over the
    bounding box.  Purely a layout choice: every result of the path is invariant to the point order
    (per-point tests and set cardinalities), and outputs are returned in the original order.  Sorted
    points make instance bit rows block-sparse (the Gram skips empty chunks) and the per-frame depth /
    mask gathers of neighbouring lanes land on neighbouring pixels."""
    p = np.nan_to_num(xyz.astype(np.float64), nan=0.0, posinf=0.0, neginf=0.0)
    lo, hi = p.min(axis=0), p.max(axis=0)
    q = ((p - lo) / np.maximum(hi - lo, 1e-300) * 1023.0).astype(np.uint64)
    q = np.minimum(q, 1023)

    def spread(v):
        v = (v | (v << 16)) & np.uint64(0x030000FF)
        v = (v | (v << 8)) & np.uint64(0x0300F00F)
        v = (v | (v << 4)) & np.uint64(0x030C30C3)
        v = (v | (v << 2)) & np.uint64(0x09249249)
        return v
    code = spread(q[:, 0]) | (spread(q[:, 1]) << np.uint64(1)) | (spread(q[:, 2]) << np.uint64(2))
    return np.argsort(code, kind="stable")


@dataclasses.dataclass
class DeviceScene:
    """Everything one scene needs, resident in HBM (see DESIGN.md 'Data layout')."""
    scene_id: str
    n_points: int
    nw: int
    height: int
    width: int
    cam_intr: np.ndarray                 # (3,3) float64, host (becomes kernel arguments)
    xyz: torch.Tensor                    # f64 [3][n_pad]
    depth: Optional[torch.Tensor]        # f32 [n_depth][H*W] metres, or None when depth_raw is resident instead
    # per kernel frame (mask frames in mask_2d order, then viewed-only frames)
    inv_pose: torch.Tensor               # f64 [F][16]
    depth_index: torch.Tensor            # i32 [F]
    frame_mask: torch.Tensor             # i32 [F]  index into maskbits or -1
    frame_rowbase: torch.Tensor          # i32 [F]
    frame_nmask: torch.Tensor            # i32 [F]
    frame_flags: torch.Tensor            # i32 [F]  bit0: counts towards viewed_count
    n_frames: int
    n_mask_frames: int                   # the first n_mask_frames entries carry masks
    n_viewed: int                        # number of frames of the detection-ratio sweep
    word_bits: int
    n_rows: int                          # Ins = total number of 2-D masks
    run_start: torch.Tensor
    run_end: torch.Tensor
    mask_run_offs: torch.Tensor
    view_mask_offs: torch.Tensor
    conf: torch.Tensor                   # (Ins,) float16/float32 device
    labels: List[str]                    # Ins label strings (host)
    label_id: torch.Tensor               # i32 [Ins]
    n_label_ids: int = 1                 # number of distinct label strings
    stage1: Optional[dict] = None
    unsort: Optional[torch.Tensor] = None   # i32 [N]: position of original point o in the sorted cloud (None = unsorted)
    tile_bounds: Optional[torch.Tensor] = None   # f64 [tiles][6]: boxes of the sweep's point tiles (frustum culling)
    perm: Optional[torch.Tensor] = None     # i32 [N]: original index of sorted position s (inverse of `unsort`)
    depth_raw: Optional[torch.Tensor] = None   # depth at the sensor's resolution: int16 [n_depth][hs][ws] (the uint16
                                               # millimetres of the PNGs) or -- depth_size given -- [n_depth][tiled texels]
                                               # in 8 x 8 tiles, int16 or float32 metres; the sweep evaluates the bilinear
                                               # resize (and / 1000 for int16) per point (P:432-436)
    depth_size: Optional[tuple] = None         # (hs, ws) of the tiled frames

    @property
    def sweep_depth(self):
        """What the projection sweep gathers from: the raw frames when resident, else the float32 (H, W) images."""
        return self.depth_raw if self.depth_raw is not None else self.depth


def tile_raw_depth():
    """Layout of resident sensor-resolution depth: "f32" (default) 8 x 8-texel tiles of float32 metres (`/ 1000` done once
    per texel on the way in); BFF_DEPTH_TILES=u16: tiles of the uint16 millimetres; BFF_DEPTH_TILES=0: None, the frames
    row-major as stored.  All three give bit-identical results."""
    import os
    v = os.environ.get("BFF_DEPTH_TILES", "f32")
    return None if v == "0" else ("u16" if v == "u16" else "f32")


RAW_DEPTH_MAX_POINTS = 500_000


def keep_raw_depth(n_points: int = 0, height: int = 0, width: int = 0) -> bool:
    """Raw 16-bit depth stays resident at the sensor's resolution and is resized per point inside the sweep -- for
    clouds up to RAW_DEPTH_MAX_POINTS points; larger ones take the separate scale + resize pass into float32 (H, W)
    images (bit-identical; the pass costs 8 x the bytes, but the sweep of a 10^6-point cloud is bound by its geometry and
    runs at a higher occupancy without the resize: config 4, 1.54 vs 1.84 ms; config 2: 0.31 vs 0.28 ms).
    BFF_DEPTH_RESIZE_PASS=1 / 0 forces the pass / the in-sweep resize.  Images whose tap table (12 B per row and column)
    would not fit 48 KB of LDS always take the pass."""
    import os
    if height + width > 4096:
        return False
    v = os.environ.get("BFF_DEPTH_RESIZE_PASS")
    if v in ("0", "1"):
        return v == "0"
    return n_points <= RAW_DEPTH_MAX_POINTS


def viewed_frame_ids(color_files, downsample_ratio):
    """Reference projection_2d_to_3d.py:528-535,545."""
    files = [f for f in color_files if f.endswith(".jpg")]
    files.sort(key=lambda x: int(x.split(".")[0]))
    return [f[:-4] for f in files[::downsample_ratio]]


def prepare_scene(scene, cfg, device="cuda", with_viewed=True, sort_points=True, raw_depth_resident=None) -> DeviceScene:
    """Upload one scene.  `scene` is duck-typed like beyond_fixed_forms_amd.synthetic.SceneInputs
    (the reference's on-disk objects held in memory)."""
    dev = torch.device(device)
    h, w = int(cfg.height_2d), int(cfg.width_2d)
    pts = np.asarray(scene.points)[:, :3].astype(np.float64, copy=False)          # :387
    n = pts.shape[0]
    nw = (n + 63) // 64
    n_pad = max(1024, ((n + 1023) // 1024) * 1024)
    soa = np.zeros((3, n_pad), dtype=np.float64)
    unsort = perm = None
    if sort_points and n > 1:
        perm = morton_order(pts)                    # sorted position s holds original point perm[s]
        soa[:, :n] = pts[perm].T
        unsort = np.empty(n, dtype=np.int32)
        unsort[perm] = np.arange(n, dtype=np.int32)
    else:
        soa[:, :n] = pts.T
    cam_intr = np.asarray(scene.cam_intr, dtype=np.float64)[:3, :3].copy()          # :376

    # ---- frame table: every 2-D mask frame in list order (chunks of <= word_bits masks), then the
    # frames of the detection-ratio sweep that carry no masks
    max_m = max((len(fr["segmented_frame_masks"]) for fr in scene.mask_2d), default=0)
    word_bits = 32 if max_m <= 32 else 64
    viewed = viewed_frame_ids(scene.color_files, cfg.downsample_ratio) if with_viewed else []
    viewed_left = dict.fromkeys(viewed)              # ordered set of frames still to be counted
    depth_slot, depth_list = {}, []
    raw_depth = getattr(scene, "depths_raw", None)

    def slot(fid):
        if fid not in depth_slot:
            if raw_depth is not None:                 # uploaded as uint16, scaled + resized on the device
                d = np.asarray(raw_depth[fid])
                if d.dtype != np.uint16 or d.ndim != 2:
                    raise ValueError(f"raw depth {fid}: expected a 2-D uint16 array")
                depth_list.append(d)
            else:
                d = np.asarray(scene.depths[fid], dtype=np.float32)
                if d.shape != (h, w):
                    raise ValueError(f"depth {fid}: shape {d.shape} != ({h},{w})")
                depth_list.append(d.reshape(-1))
            depth_slot[fid] = len(depth_list) - 1
        return depth_slot[fid]

    inv, d_idx, f_mask, f_rowbase, f_nmask, f_flags = [], [], [], [], [], []
    all_rles, view_mask_offs, conf_list, labels = [], [0], [], []
    row = 0
    for fr in scene.mask_2d:                                                        # :413-421
        fid = fr["frame_id"][:-4]
        rles = fr["segmented_frame_masks"]
        m = len(rles)
        if not (len(fr["confidences"]) == m and len(fr["labels"]) == m):
            raise ValueError(f"frame {fid}: masks / confidences / labels differ in length")
        ipose = np.linalg.inv(np.asarray(scene.poses[fid], dtype=np.float64))       # :425
        first = True
        for c0 in range(0, m, word_bits):
            mc = min(word_bits, m - c0)
            inv.append(ipose); d_idx.append(slot(fid))
            f_mask.append(len(view_mask_offs) - 1); f_rowbase.append(row); f_nmask.append(mc)
            counted = first and fid in viewed_left
            if counted:
                del viewed_left[fid]
            f_flags.append(1 if counted else 0)
            first = False
            all_rles += list(rles[c0:c0 + mc])
            view_mask_offs.append(view_mask_offs[-1] + mc)
            row += mc
        conf_list.append(fr["confidences"])
        labels += list(fr["labels"])
    n_mask_frames = len(inv)
    for fid in viewed_left:                                                         # :538-567
        inv.append(np.linalg.inv(np.asarray(scene.poses[fid], dtype=np.float64)))
        d_idx.append(slot(fid)); f_mask.append(-1); f_rowbase.append(0); f_nmask.append(0); f_flags.append(1)

    for r in all_rles:
        if int(r["length"]) != h * w:
            raise ValueError(f"mask RLE length {r['length']} != H*W = {h * w}")
    rs, re, roffs = runs_from_rles(all_rles, "2-D mask")
    if conf_list:
        dts = {c.dtype for c in conf_list}
        if len(dts) != 1:
            raise TypeError(f"mixed confidence dtypes {dts}")
        conf = torch.cat([c.reshape(-1).cpu() for c in conf_list])
    else:
        conf = torch.zeros(0, dtype=torch.float16)
    ids = {}
    label_id = np.array([ids.setdefault(s, len(ids)) for s in labels], dtype=np.int32)

    def t(a, dtype):
        return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype).to(dev)

    nf = len(inv)
    raw_keep = raw_size = None

    def frames_to_device(frames, np_dtype, torch_dtype):
        """list of equally shaped host arrays -> one device tensor [F][...], frame by frame (no 1.5 GB np.stack:
        the copies go straight from the callers' arrays)."""
        out = torch.empty((len(frames),) + tuple(frames[0].shape), dtype=torch_dtype, device=dev)
        for i, f in enumerate(frames):
            out[i].copy_(torch.from_numpy(np.ascontiguousarray(f, dtype=np_dtype)))
        return out

    if raw_depth is not None and depth_list:
        from . import _lib
        from .io import bilinear_taps
        hs, ws = depth_list[0].shape
        if any(d.shape != (hs, ws) for d in depth_list):
            raise ValueError("raw depth frames of different sizes")
        taps = None
        if (hs, ws) != (h, w):
            taps = tuple(torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in bilinear_taps(hs, ws, h, w))
        raw_dev = frames_to_device([d.view(np.int16) for d in depth_list], np.int16, torch.int16)
        if keep_raw_depth(n, h, w) if raw_depth_resident is None else raw_depth_resident:
            depth_dev, raw_keep = None, raw_dev
            if tile_raw_depth():
                raw_keep, raw_size = _lib.tile_depth(raw_dev, metres=tile_raw_depth() == "f32"), (hs, ws)
        else:
            depth_dev = _lib.depth_from_u16(raw_dev, h, w, taps)
    elif depth_list:
        depth_dev = frames_to_device(depth_list, np.float32, torch.float32)
    else:
        depth_dev = torch.zeros((0, h * w), dtype=torch.float32, device=dev)
    xyz_dev = t(soa, torch.float64)
    bounds = None
    if dev.type == "cuda" and n:
        from . import _lib
        bounds = _lib.point_tile_bounds(xyz_dev, n)      # built once per scene, next to the spatial sort it relies on
    return DeviceScene(
        scene_id=scene.scene_id, n_points=n, nw=nw, height=h, width=w, cam_intr=cam_intr,
        xyz=xyz_dev, tile_bounds=bounds,
        depth=depth_dev,
        inv_pose=t(np.stack(inv).reshape(nf, 16) if nf else np.zeros((0, 16)), torch.float64),
        depth_index=t(np.array(d_idx, np.int32), torch.int32), frame_mask=t(np.array(f_mask, np.int32), torch.int32),
        frame_rowbase=t(np.array(f_rowbase, np.int32), torch.int32),
        frame_nmask=t(np.array(f_nmask, np.int32), torch.int32), frame_flags=t(np.array(f_flags, np.int32), torch.int32),
        n_frames=nf, n_mask_frames=n_mask_frames, n_viewed=len(viewed), word_bits=word_bits, n_rows=row,
        run_start=t(rs, torch.int32), run_end=t(re, torch.int32), mask_run_offs=t(roffs, torch.int32),
        view_mask_offs=t(np.array(view_mask_offs, np.int32), torch.int32),
        conf=conf.to(dev), labels=labels, label_id=t(label_id, torch.int32), n_label_ids=max(1, len(ids)),
        stage1=getattr(scene, "stage1", None), unsort=None if unsort is None else t(unsort, torch.int32),
        perm=None if perm is None else t(perm.astype(np.int32), torch.int32), depth_raw=raw_keep, depth_size=raw_size)
