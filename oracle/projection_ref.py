"""CPU restatement of the reference's 2D->3D projection stage.  TEST INFRASTRUCTURE ONLY.

Follows /root/reference/tools/projection_2d_to_3d.py.  Every function cites the
reference lines it restates.  The arithmetic deliberately uses the *same library
calls* as the reference (NumPy float64 matmul / round / astype, torch float32
matmul / unique / comparisons) so that dtype promotion and rounding are inherited,
not re-derived.  Pinned by tests/golden/proj_helpers_*.npz (see make_golden.py).

Scene inputs are duck-typed (``scene`` needs the attributes below) so the disk
format of the reference maps 1:1 to memory:

  scene.scene_id      str
  scene.points        (N, >=3) float64    -- np.load(<scene>.npy)            (:387)
  scene.cam_intr      (>=3, >=3) float64  -- np.loadtxt(intrinsic_color.txt) (:376)
  scene.poses         {frame_id: (4,4) float64}   -- np.loadtxt(pose/<id>.txt)  (:422,:546)
  scene.depths        {frame_id: (H,W) float32}   -- cv2.imread/1000 + cv2.resize (:432-436);
                                                     enters already decoded+resized (cv2 absent:
                                                     parity unpinned for that step)
  scene.mask_2d       list of per-frame dicts as torch.load(mask_2d .pth) gives (:396)
  scene.color_files   list of "<n>.jpg" names in the scene's color/ dir       (:528)
"""
from __future__ import annotations

import math
import warnings

import numpy as np
import torch

from . import rle_ref

DEPTH_THRESH = 0.08  # hard-coded at the call sites :438 and :565


# --------------------------------------------------------------------------- geometry
def homogeneous_cloud(points: np.ndarray) -> np.ndarray:
    """(N,>=3) -> (4,N) float64 with a row of ones.  Reference :387-390."""
    xyz = points[:, :3]
    return np.concatenate([xyz, torch.ones([xyz.shape[0], 1])], axis=1).T


def world_to_camera(cloud_h: np.ndarray, cam_pose: np.ndarray) -> np.ndarray:
    """pts_cam (N,3) = (inv(pose) @ cloud_h).T[:, :3].  Reference :424-425 / :548-549."""
    return (np.linalg.inv(cam_pose) @ cloud_h).T[:, :3]


def world_to_camera_inv(cloud_h: np.ndarray, inv_pose: np.ndarray) -> np.ndarray:
    """Same as world_to_camera but with the inverse already taken (fixtures store it so
    that LAPACK differences between hosts cannot move a borderline point)."""
    return (inv_pose @ cloud_h).T[:, :3]


def project_to_pixels(pts_cam: np.ndarray, cam_intr: np.ndarray) -> np.ndarray:
    """round_half_even((K @ pts^T) / z)[:2] -> int64 (N,2) as (x, y).  Reference :37-48.

    NaN/inf (z == 0) become INT64_MIN through astype on x86, i.e. out of bounds."""
    p = pts_cam.T
    with np.errstate(all="ignore"), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        uvw = cam_intr @ p / p[2]
        return np.round(uvw[:2].T).astype(np.int64)


def visibility(pts_cam: np.ndarray, pix: np.ndarray, depth_im: np.ndarray,
               depth_thresh: float = DEPTH_THRESH) -> np.ndarray:
    """in-bounds & depth != 0 & |z - depth| < thresh -> bool (N,).  Reference :51-70.

    z is float64, depth float32: the subtraction promotes to float64; there is no z > 0 test."""
    h, w = depth_im.shape
    x, y = pix[:, 0], pix[:, 1]
    inb = (x >= 0) & (x < w) & (y >= 0) & (y < h)
    d = depth_im[y[inb], x[inb]]
    ok = (d != 0) & (np.abs(pts_cam[inb][:, 2] - d) < depth_thresh)
    vis = np.zeros(pix.shape[0], dtype=np.bool_)
    vis[inb] = ok
    return vis


def masked_points(pix: np.ndarray, vis: np.ndarray, pred_masks: np.ndarray) -> np.ndarray:
    """bool (M,N): mask value at the pixel of every visible point.  Reference :73-92."""
    m = pred_masks.shape[0]
    out = np.zeros((m, pix.shape[0]), dtype=np.bool_)
    x, y = pix[vis].T
    for k in range(m):
        out[k, vis] = pred_masks[k, y, x]
    return out


# --------------------------------------------------------------------------- aggregation
def label_equality(labels) -> torch.Tensor:
    """S[i,j] = labels[i] == labels[j], bool (Ins,Ins).  Reference :169-187 (vectorised:
    string equality is an equivalence, so comparing integer ids of the strings is identical)."""
    ids = {}
    idx = torch.tensor([ids.setdefault(s, len(ids)) for s in labels], dtype=torch.int64)
    return idx[:, None] == idx[None, :]


def pairwise_iou(ins_masks: torch.Tensor) -> torch.Tensor:
    """f32 IoU matrix, 0/0 -> NaN.  Reference :149-166."""
    f = ins_masks.float()
    inter = torch.matmul(f, f.T)
    area = torch.sum(f, dim=1)
    union = area.unsqueeze(1) + area.unsqueeze(0) - inter
    return inter / union


def connected_groups(adj: torch.Tensor):
    """Components of the closure of ``adj`` listed like reference :250-274.

    The reference iterates R <- clamp(R@A + A, 0, 1) num_nodes times (full transitive
    closure), then scans rows in index order: an unvisited row i yields the ascending list of
    j with R[i,j] > 0 (possibly [] when row i is empty, e.g. an empty mask whose IoU with
    itself is NaN) and marks those j visited.  Restated with a frontier search over the same
    (possibly asymmetric) matrix, which gives exactly row i of the closure."""
    a = (adj > 0).cpu().numpy()
    n = a.shape[0]
    visited = np.zeros(n, dtype=bool)
    comps = []
    for i in range(n):
        if visited[i]:
            continue
        reach = a[i].copy()
        frontier = np.flatnonzero(reach)
        while frontier.size:
            nxt = a[frontier].any(axis=0) & ~reach
            reach |= nxt
            frontier = np.flatnonzero(nxt)
        comps.append(np.flatnonzero(reach).tolist())
        visited[reach] = True
    return comps


def merge_groups(ins_masks, confidences, labels, merge_matrix, min_members):
    """Reference merge_masks :190-247.  Returns (masks, conf, labels, groups)."""
    groups = connected_groups(merge_matrix.float())
    groups = [g for g in groups if len(g) >= min_members]
    agg_m, agg_c, agg_l = [], [], []
    for g in groups:
        if g == []:
            continue
        row = torch.zeros(ins_masks.shape[1], dtype=torch.bool)
        conf = []
        for k in g:
            row |= ins_masks[k]
            conf.append(confidences[k])
        agg_m.append(row)
        agg_c.append(sum(conf) / len(conf))   # sequential, in the dtype of the inputs
        agg_l.append(labels[g[0]])
    if not agg_m:
        return torch.tensor([[]]), torch.tensor([]), [], []
    return torch.stack(agg_m), torch.tensor(agg_c), agg_l, groups


def aggregate(raw: dict, iou_threshold: float, min_members: int):
    """Reference aggregate :100-146 (feature_similarity_threshold is unused there)."""
    labels = raw["final_class"]
    same = label_equality(labels)
    iou = pairwise_iou(raw["ins"])
    merge = same & (iou > iou_threshold)
    masks, conf, lab, groups = merge_groups(raw["ins"], raw["conf"], labels, merge, min_members)
    if groups == []:
        return {"ins": torch.tensor([[]]), "conf": torch.tensor([]), "final_class": []}, []
    return {"ins": masks, "conf": conf, "final_class": lab}, groups


def resolve_overlaps(masks: torch.Tensor, groups) -> torch.Tensor:
    """Reference solve_overlapping :277-301 (in place, order dependent)."""
    size = [len(g) for g in groups]
    k = len(masks)
    pairs = [(i, j) for i in range(k) for j in range(i + 1, k) if torch.any(masks[i] & masks[j])]
    for i, j in pairs:
        if size[i] > size[j]:
            masks[j] &= ~masks[i]
        else:
            masks[i] &= ~masks[j]
    return masks


# --------------------------------------------------------------------------- scene loop
def viewed_frame_ids(color_files, downsample_ratio):
    """Reference :528-535,:545."""
    files = [f for f in color_files if f.endswith(".jpg")]
    files.sort(key=lambda x: int(x.split(".")[0]))
    return [f[:-4] for f in files[::downsample_ratio]]


def empty_result():
    """Reference :468-470 / :499-501."""
    return {"ins": torch.tensor([[]]), "conf": torch.tensor([]), "final_class": []}


def project_scene_ref(scene, cfg, return_debug: bool = False, stage_times: dict = None):
    """One iteration of the scene loop, reference :365-634, on in-memory inputs.

    Returns the dict the reference saves to mask_3d_dir/<cls>/<scene>.pth.  stage_times (optional dict):
    receives the wall time of the stages SURVEY.md 8(d) lists -- (i) projection+votes, (ii) Gram+merge,
    (iii) ratio-filter sweep, (iv) overlap+filters."""
    import time
    _t = [time.perf_counter()]

    def lap(name):
        if stage_times is not None:
            now = time.perf_counter()
            stage_times[name] = stage_times.get(name, 0.0) + now - _t[0]
            _t[0] = now

    dbg = {}
    cam_intr = np.asarray(scene.cam_intr)[:3, :3]                               # :376
    cloud_h = homogeneous_cloud(np.asarray(scene.points))                      # :387-390
    n = cloud_h.shape[1]
    frames = rle_ref.decode_2d_masks_ref([dict(f) for f in scene.mask_2d],
                                         (cfg.height_2d, cfg.width_2d))         # :400
    masked_counts = torch.zeros(n)                                             # :402
    raw = {"ins": [], "conf": [], "final_class": []}
    for fr in frames:                                                          # :413
        frame_id = fr["frame_id"][:-4]
        pred = fr["segmented_frame_masks"].to(torch.float32).squeeze(dim=1).numpy()
        pts = world_to_camera(cloud_h, np.asarray(scene.poses[frame_id]))      # :424-425
        pix = project_to_pixels(pts, cam_intr)                                 # :426
        depth = scene.depths[frame_id]                                         # :431-436
        vis = visibility(pts, pix, depth, DEPTH_THRESH)                        # :437-439
        mp = torch.from_numpy(masked_points(pix, vis, pred))                   # :441-445
        for k in range(mp.shape[0]):                                           # :454-457
            raw["ins"].append(mp[k])
            raw["conf"].append(fr["confidences"][k])
            raw["final_class"].append(fr["labels"][k])
        for row in mp:                                                         # :459-461
            masked_counts[row] += 1
    dbg["masked_counts_raw"] = masked_counts.clone()
    lap("i_projection_votes")
    if len(raw["conf"]) == 0:                                                  # :465-478
        return (empty_result(), dbg) if return_debug else empty_result()
    raw["ins"] = torch.stack(raw["ins"], dim=0)                                # :481
    raw["conf"] = torch.tensor(raw["conf"])                                    # :484
    dbg["raw_ins"] = raw["ins"]
    agg, groups = aggregate(raw, cfg.iou_thres, cfg.min_aggragated_masks)      # :489
    dbg["groups"] = groups
    lap("ii_gram_merge")
    if len(agg["conf"]) == 0:                                                  # :496-509
        return (empty_result(), dbg) if return_debug else empty_result()

    if cfg.if_occurance_threshold:                                             # :512-522
        uniq = masked_counts.unique()
        thr = uniq[math.floor(cfg.occurance_threshold * uniq.shape[0])]
        masked_counts[masked_counts < thr] = 0
        dbg["thr"] = float(thr)
    elif cfg.if_detected_ratio_threshold:                                      # :524-578
        viewed = torch.zeros(n)
        for frame_id in viewed_frame_ids(scene.color_files, cfg.downsample_ratio):
            pts = world_to_camera(cloud_h, np.asarray(scene.poses[frame_id]))
            pix = project_to_pixels(pts, cam_intr)
            vis = visibility(pts, pix, scene.depths[frame_id], DEPTH_THRESH)
            viewed += torch.tensor(vis)
        ratio = masked_counts / (viewed + 1)
        uniq = ratio.unique()
        thr = uniq[math.floor(cfg.detected_ratio_threshold * uniq.shape[0])]
        masked_counts[ratio < thr] = 0
        dbg["viewed_counts"] = viewed
        dbg["thr"] = float(thr)
    keep_pts = masked_counts > 0                                               # :583
    dbg["keep_pts"] = keep_pts
    lap("iii_ratio_filter_sweep")

    before = agg["ins"].sum(dim=1)                                             # :592
    agg["ins"] = resolve_overlaps(agg["ins"], groups)                          # :594
    agg["ins"] &= keep_pts.unsqueeze(0)                                        # :595
    after = agg["ins"].sum(dim=1)                                              # :596
    keep = (after > cfg.remove_small_masks) & (after > cfg.remove_filtered_masks * before)
    dbg["before"], dbg["after"], dbg["keep"] = before, after, keep
    out = {
        "ins": agg["ins"][keep],                                               # :601-607
        "conf": agg["conf"][keep],                                             # :610-616
        "final_class": [c for c, k in zip(agg["final_class"], keep.tolist()) if k],  # :617-623
    }
    lap("iv_overlap_filters")
    return (out, dbg) if return_debug else out
