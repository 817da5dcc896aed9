"""Deterministic synthetic RGB-D scenes of the shapes BASELINE.json names.

A scene is produced in exactly the *disk formats of the reference*, held in memory
(`SceneInputs`): an (N,6) float64 cloud like `<scene>.npy`, a 4x4 intrinsic like
`intrinsic_color.txt`, 4x4 camera-to-world poses, float32 (H,W) depth in metres (what
`cv2.imread(png)/1000` + `cv2.resize` yields, reference projection_2d_to_3d.py:431-436), the
`mask_2d` list-of-dicts with RLE masks that `segmentation_2d.py:297-305,500` writes, and an
Open3DIS-style stage-1 dict (`refinement.py:182-193`).

Geometry: a 6x4x3 m room with 10 axis-aligned cuboids; depth is an exact ray cast of that
geometry + hash noise (~5 mm), 5 % dropped pixels, quantised to 1 mm like a 16-bit PNG.
Heavy steps run with torch on `device` (GPU when present) but use only integer hashes and
IEEE +,-,*,/ so the result does not depend on the device.
"""
from __future__ import annotations

import dataclasses
import math
from typing import Dict, List, Optional

import numpy as np
import torch

ROOM = np.array([6.0, 4.0, 3.0])

SHAPES = {
    # name: (N, V_all, H, W, M)
    "tiny": (4000, 6, 120, 160, 4),
    "c1": (20_000, 10, 480, 640, 5),
    "c2": (200_000, 300, 968, 1296, 30),
    "c4": (1_000_000, 600, 968, 1296, 64),
}


@dataclasses.dataclass
class SceneInputs:
    scene_id: str
    points: np.ndarray                 # (N,6) float64 xyz+rgb            <scene>.npy
    cam_intr: np.ndarray               # (4,4) float64                    intrinsic_color.txt
    poses: Dict[str, np.ndarray]       # frame id -> (4,4) float64        pose/<id>.txt
    depths: Dict[str, np.ndarray]      # frame id -> (H,W) float32 metres depth/<id>.png decoded
    mask_2d: List[dict]                # segmentation_2d.py output (RLE)
    color_files: List[str]             # names in color/
    stage1: Optional[dict] = None      # Open3DIS stage-1 result
    height: int = 0
    width: int = 0
    point_object: Optional[np.ndarray] = None   # generator ground truth (not an input)
    depths_raw: Optional[Dict[str, np.ndarray]] = None   # frame id -> uint16 (h,w) millimetres as stored in the PNG;
                                                          # when given, /1000 + resize run on the device
    depth_staged: Optional[tuple] = None                 # (ingest.Staging, frame ids): depths_raw already lies in that
                                                          # staging's pinned "depth" buffer in this order (io.load_scene)


def _hash32(x: torch.Tensor) -> torch.Tensor:
    """Integer avalanche hash on int64 tensors holding 32-bit values (device independent)."""
    m = 0xFFFFFFFF
    x = x & m
    x = ((x ^ (x >> 16)) * 0x7FEB352D) & m
    x = ((x ^ (x >> 15)) * 0x846CA68B) & m
    return x ^ (x >> 16)


def _cuboids(rng: np.random.Generator, n_obj: int = 10):
    """n_obj axis-aligned cuboids standing on the floor: 10 on a 5 x 2 grid (the default scenes), more on a
    denser grid with proportionally smaller footprints."""
    lo, hi = [], []
    if n_obj <= 10:
        gx, gy, scale = np.linspace(0.9, 5.1, 5), np.array([1.0, 3.0]), 1.0
    else:
        nx = int(math.ceil(math.sqrt(n_obj * 1.5)))
        ny = int(math.ceil(n_obj / nx))
        gx, gy = np.linspace(0.6, 5.4, nx), np.linspace(0.5, 3.5, ny)
        scale = min(4.8 / nx, 3.0 / ny) / 1.3
    k = 0
    for y in gy:
        for x in gx:
            if k >= n_obj:
                break
            sx, sy = rng.uniform(0.4, 0.9, 2) * scale
            h = rng.uniform(0.4, 1.2)
            cx, cy = x + rng.uniform(-0.1, 0.1) * scale, y + rng.uniform(-0.3, 0.3) * scale
            lo.append([cx - sx / 2, cy - sy / 2, 0.0])
            hi.append([cx + sx / 2, cy + sy / 2, h])
            k += 1
    return np.array(lo), np.array(hi)


def _sample_box_surface(rng, lo, hi, n, skip_bottom):
    """n points uniform (area weighted) on the faces of an axis-aligned box."""
    d = hi - lo
    faces = []  # (axis, side)
    for ax in range(3):
        for side in (0, 1):
            if skip_bottom and ax == 2 and side == 0:
                continue
            faces.append((ax, side))
    area = np.array([d[(ax + 1) % 3] * d[(ax + 2) % 3] for ax, _ in faces])
    which = rng.choice(len(faces), size=n, p=area / area.sum())
    pts = lo + rng.random((n, 3)) * d
    for f, (ax, side) in enumerate(faces):
        sel = which == f
        pts[sel, ax] = hi[ax] if side else lo[ax]
    return pts


def _look_at(pos, target):
    f = target - pos
    f /= np.linalg.norm(f)
    r = np.cross(f, np.array([0.0, 0.0, 1.0]))
    r /= np.linalg.norm(r)
    d = np.cross(f, r)
    pose = np.eye(4)
    pose[:3, 0], pose[:3, 1], pose[:3, 2], pose[:3, 3] = r, d, f, pos
    return pose


def _raycast(pose, fx, cx, cy, h, w, lo, hi, device):
    """Exact depth (camera z) and object id per pixel for the room + cuboids."""
    dt = torch.float64
    v, u = torch.meshgrid(torch.arange(h, device=device, dtype=dt),
                          torch.arange(w, device=device, dtype=dt), indexing="ij")
    dcam = torch.stack([(u - cx) / fx, (v - cy) / fx, torch.ones_like(u)], dim=-1)   # (H,W,3)
    rot = torch.as_tensor(pose[:3, :3], device=device, dtype=dt)
    org = torch.as_tensor(pose[:3, 3], device=device, dtype=dt)
    dw = dcam @ rot.T
    dw = torch.where(dw.abs() < 1e-12, torch.full_like(dw, 1e-12), dw)
    room_hi = torch.as_tensor(ROOM, device=device, dtype=dt)
    t_room = torch.maximum((0.0 - org) / dw, (room_hi - org) / dw).min(dim=-1).values  # exit distance
    best_t = t_room
    best_id = torch.full((h, w), -1, device=device, dtype=torch.int64)
    for k in range(lo.shape[0]):
        l = torch.as_tensor(lo[k], device=device, dtype=dt)
        hh = torch.as_tensor(hi[k], device=device, dtype=dt)
        t1, t2 = (l - org) / dw, (hh - org) / dw
        tmin = torch.minimum(t1, t2).max(dim=-1).values
        tmax = torch.maximum(t1, t2).min(dim=-1).values
        hit = (tmax >= tmin) & (tmin > 0) & (tmin < best_t)
        best_t = torch.where(hit, tmin, best_t)
        best_id = torch.where(hit, torch.full_like(best_id, k), best_id)
    return best_t, best_id      # dcam z == 1 so the ray parameter is the camera-space depth


def _rle_from_dense(dense: torch.Tensor):
    """bool (M, L) -> list of {"length", "counts"} in the reference format
    (rle_encode_decode.py:10-32: 1-based start, length pairs)."""
    m, length = dense.shape
    z = torch.zeros((m, 1), dtype=torch.bool, device=dense.device)
    padded = torch.cat([z, dense, z], dim=1)
    out = []
    for i in range(m):
        edges = torch.nonzero(padded[i, 1:] != padded[i, :-1]).view(-1) + 1
        edges[1::2] -= edges[::2]
        out.append(dict(length=length, counts=edges.cpu().numpy()))
    return out


def make_scene(shape="c1", seed: int = 0, device="cpu", query: str = "table", n_labels: int = 1,
               n_points: int = None, n_views: int = None, height: int = None, width: int = None,
               n_masks: int = None, downsample_ratio: int = 10, n_stage1: int = 100,
               shuffle_points: bool = True, conf_dtype=torch.float16, cut_masks: bool = True,
               n_objects: int = 10, distinct_masks: bool = False, dilate: bool = True) -> SceneInputs:
    """Build one synthetic scene.  `shape` names a BASELINE config (see SHAPES); explicit
    keyword sizes override it.  `n_labels` > 1 mixes several label strings among the masks.
    cut_masks=False: every 2-D mask is the whole (dilated) silhouette of its object instead of a random half of
    it -- the views of one object then merge into one instance and most instances survive the filters (a scene
    with many stage-2 instances for the benchmark; the default keeps the golden fixtures' scenes).
    n_objects: number of cuboids; distinct_masks=True: a view's masks show different objects as far as there are
    visible ones (then repeats), instead of independent draws; dilate=False: exact silhouettes (no 1-3 px bleed)."""
    n0, v0, h0, w0, m0 = SHAPES[shape] if isinstance(shape, str) else shape
    n, v_all = n_points or n0, n_views or v0
    h, w, m_per = height or h0, width or w0, n_masks or m0
    rng = np.random.default_rng(seed)
    device = torch.device(device)
    lo, hi = _cuboids(rng, n_objects)
    n_obj = lo.shape[0]

    # ---- cloud: half on the room faces, half on the cuboids (area weighted)
    n_room = n // 2
    pts = [_sample_box_surface(rng, np.zeros(3), ROOM, n_room, skip_bottom=False)]
    obj = [np.full(n_room, -1)]
    area = np.array([2 * (d[0] * d[2] + d[1] * d[2]) + d[0] * d[1] for d in (hi - lo)])
    per = np.floor((n - n_room) * area / area.sum()).astype(int)
    per[0] += (n - n_room) - per.sum()
    for k in range(n_obj):
        pts.append(_sample_box_surface(rng, lo[k], hi[k], per[k], skip_bottom=True))
        obj.append(np.full(per[k], k))
    xyz, obj = np.concatenate(pts), np.concatenate(obj)
    if shuffle_points:
        perm = rng.permutation(n)
        xyz, obj = xyz[perm], obj[perm]
    points = np.concatenate([xyz, rng.integers(0, 256, (n, 3)).astype(np.float64)], axis=1)

    # ---- cameras on a circle of radius 1.5 m at height 1.5 m looking at the centre, jittered
    fx = 0.9 * w
    cx, cy = w / 2 - 0.5, h / 2 - 0.5
    cam_intr = np.eye(4)
    cam_intr[0, 0] = cam_intr[1, 1] = fx
    cam_intr[0, 2], cam_intr[1, 2] = cx, cy
    centre = np.array([ROOM[0] / 2, ROOM[1] / 2, 0.0])
    n_color = (v_all - 1) * downsample_ratio + 1
    color_files = [f"{i}.jpg" for i in range(n_color)]
    frame_ids = [str(i * downsample_ratio) for i in range(v_all)]
    poses, depths, mask_2d = {}, {}, []
    labels_pool = [query] + [f"{query} variant {i}" for i in range(1, n_labels)]
    for vi, fid in enumerate(frame_ids):
        ang = 2 * math.pi * vi / v_all + rng.uniform(-0.05, 0.05)
        pos = centre + np.array([1.5 * math.cos(ang), 1.5 * math.sin(ang), 1.5 + rng.uniform(-0.1, 0.1)])
        target = centre + np.array([rng.uniform(-0.4, 0.4), rng.uniform(-0.4, 0.4), 0.6 + rng.uniform(-0.3, 0.3)])
        pose = _look_at(pos, target)
        poses[fid] = pose
        zc, oid = _raycast(pose, fx, cx, cy, h, w, lo, hi, device)
        pix = torch.arange(h * w, device=device, dtype=torch.int64).view(h, w)
        hsh = _hash32(pix * 0x9E3779B1 + (vi + 1) * 0x85EBCA77 + seed * 0xC2B2AE3D)
        noise = ((hsh & 0xFF) + ((hsh >> 8) & 0xFF) + ((hsh >> 16) & 0xFF) - 382).to(torch.float64) * (0.005 / 128)
        mm = torch.round((zc + noise) * 1000.0).clamp(0, 65535)
        drop = (_hash32(hsh + 0x27D4EB2F) % 100) < 5
        mm = torch.where(drop, torch.zeros_like(mm), mm)
        depths[fid] = (mm.to(torch.float32) / 1000.0).cpu().numpy()      # f32(u16) / 1000, as :432-435

        visible = [k for k in range(n_obj) if bool((oid == k).any())]
        if not visible:
            continue                               # segmentation_2d.py:271-274: frame skipped
        if distinct_masks:
            order = rng.permutation(visible)
            chosen = np.resize(order, m_per)
        else:
            chosen = rng.choice(visible, size=m_per, replace=True)
        vv, uu = torch.meshgrid(torch.arange(h, device=device, dtype=torch.float32),
                                torch.arange(w, device=device, dtype=torch.float32), indexing="ij")
        dense = torch.zeros((m_per, h, w), dtype=torch.bool, device=device)
        for j, k in enumerate(chosen):
            base = oid == int(k)
            ys, xs = torch.nonzero(base, as_tuple=True)
            x0, x1, y0, y1 = xs.min().item(), xs.max().item(), ys.min().item(), ys.max().item()
            # keep a random half-plane through the box that retains 50-100 % of its extent
            if not cut_masks:
                part = base
                rng.random(); rng.uniform(0.5, 1.0); rng.random()       # same random stream as the cut variant
            elif rng.random() < 0.5:
                cut = x0 + (x1 - x0 + 1) * rng.uniform(0.5, 1.0)
                part = base & ((uu <= cut) if rng.random() < 0.5 else (uu >= x0 + x1 - cut))
            else:
                cut = y0 + (y1 - y0 + 1) * rng.uniform(0.5, 1.0)
                part = base & ((vv <= cut) if rng.random() < 0.5 else (vv >= y0 + y1 - cut))
            r = int(rng.integers(1, 4))
            dense[j] = (torch.nn.functional.max_pool2d(part[None, None].float(), 2 * r + 1, 1, r)[0, 0] > 0) if dilate else part
        conf = torch.from_numpy(rng.uniform(0.2, 0.5, m_per)).to(conf_dtype)
        mask_2d.append({
            "frame_id": f"{fid}.jpg",
            "segmented_frame_masks": _rle_from_dense(dense.view(m_per, -1)),
            "confidences": conf,
            "labels": [labels_pool[int(i)] for i in rng.integers(0, len(labels_pool), m_per)],
        })

    stage1 = _make_stage1(rng, xyz, obj, n_obj, query, n_stage1)
    return SceneInputs(scene_id=f"scene{seed:04d}_00", points=points, cam_intr=cam_intr, poses=poses,
                       depths=depths, mask_2d=mask_2d, color_files=color_files, stage1=stage1,
                       height=h, width=w, point_object=obj)


def with_sensor_depth(scene: SceneInputs, factor: int = 2) -> SceneInputs:
    """The same scene with its depth frames as the 16-bit PNGs of a sensor with 1/factor of the working resolution
    hold them (ScanNet: 480x640 depth for 968x1296 colour): `depths_raw` = every factor-th pixel in millimetres.  The
    float32 (H, W) images are dropped: the device path resizes per point inside the sweep (P:431-436)."""
    import copy
    out = copy.copy(scene)
    out.depths_raw = {f: np.ascontiguousarray(np.round(d[::factor, ::factor].astype(np.float64) * 1000.0).astype(np.uint16))
                      for f, d in scene.depths.items()}
    out.depths = {}
    return out


def _make_stage1(rng, xyz, obj, n_obj, query, n_stage1):
    """Open3DIS-like stage-1 dict: the cuboids' point sets + random box crops, 1-D RLE."""
    from .labels import SCANNET200_LABELS
    n = xyz.shape[0]
    rows = [obj == k for k in range(min(n_obj, n_stage1))]
    while len(rows) < n_stage1:
        c = rng.uniform([0, 0, 0], ROOM)
        s = rng.uniform(0.2, 1.0, 3)
        rows.append(np.all((xyz >= c - s) & (xyz <= c + s), axis=1))
    dense = torch.from_numpy(np.stack(rows))
    qidx = SCANNET200_LABELS.index(query.replace(" ", "_")) if query.replace(" ", "_") in SCANNET200_LABELS else 1
    cls = rng.integers(0, len(SCANNET200_LABELS), n_stage1)
    cls[rng.choice(n_stage1, size=min(10, n_stage1), replace=False)] = qidx
    return {
        "ins": _rle_from_dense(dense),
        "conf": torch.from_numpy(rng.uniform(0.1, 1.0, n_stage1).astype(np.float32)),
        "final_class": [int(c) for c in cls],
    }


def make_text_bank(dim: int = 768, seed: int = 0, dtype=torch.float16):
    """Synthetic stand-in for CLIP text embeddings of the 200 ScanNet200 labels (config 5):
    unit-normalised N(0,1) rows.  Returns (bank (200,D), dict label -> row)."""
    from .labels import SCANNET200_LABELS
    g = torch.Generator().manual_seed(seed)
    bank = torch.randn(len(SCANNET200_LABELS), dim, generator=g)
    bank = (bank / bank.norm(dim=1, keepdim=True)).to(dtype)
    return bank, {lab: i for i, lab in enumerate(SCANNET200_LABELS)}
