"""Stage 2 of the pipeline on MI355X: 2-D masks -> aggregated, filtered 3-D instance masks.

Host-side mirror of the per-scene loop of the reference (tools/projection_2d_to_3d.py:365-634).
All per-point / per-pixel / per-pair arithmetic runs in libbff_hip.so (include/bff_hip.h); the host
keeps only what the reference keeps sequential and tiny: grouping component labels into the
reference's list order, the order-dependent overlap decisions (P:295-299), thresholds picked from
sets of distinct values, and dict assembly.  `P:` = tools/projection_2d_to_3d.py.
"""
from __future__ import annotations

import dataclasses
from typing import List

import numpy as np
import torch

from . import _lib
from .scene import DEPTH_THRESH, DeviceScene, prepare_scene
from .timing import span


@dataclasses.dataclass
class Stage2Result:
    """Output of the projection stage, kept bit-packed on the device.

    `to_dict()` gives exactly what the reference saves at P:630-634:
    {"ins": bool (K,N) device tensor, "conf": (K,) tensor, "final_class": list[str]} or, when
    nothing survives, the reference's empty form (P:468-470, 499-501)."""
    scene_id: str
    n_points: int
    rows: torch.Tensor            # int64 [K][nw] bit rows (device)
    conf: torch.Tensor            # (K,) in the dtype of the input confidences (device)
    final_class: List[str]
    groups: list                  # indices of the raw 2-D masks merged into each *pre-filter* instance
    debug: dict

    @property
    def empty(self):
        return self.rows.shape[0] == 0 and self.debug.get("empty_form", False)

    def to_dict(self):
        if self.debug.get("empty_form", False):
            dev = self.rows.device
            return {"ins": torch.tensor([[]]).to(dev), "conf": torch.tensor([]).to(dev), "final_class": []}
        return {"ins": _lib.unpack_rows(self.rows, self.n_points), "conf": self.conf,
                "final_class": list(self.final_class)}


def _empty(ds: DeviceScene, debug) -> Stage2Result:
    debug["empty_form"] = True
    return Stage2Result(ds.scene_id, ds.n_points, torch.zeros((0, ds.nw), dtype=torch.int64, device=ds.xyz.device),
                        torch.zeros(0, device=ds.xyz.device), [], [], debug)


def groups_from_labels(comp: np.ndarray, has_self_loop: np.ndarray, min_members: int = 0):
    """Component ids (any integer naming the connected component of node i) -> the list
    find_unconnected_subgraphs_tensor returns (P:262-274), already filtered by len >= min_members
    (P:203): components in order of their smallest member, members ascending.  A node with no edge at
    all -- a singleton component whose self-IoU is NaN (empty mask) or not above the threshold --
    yields an empty list; a node without a self loop that has neighbours (possible only for
    iou_thres < 0) is a normal member, since the closure reaches it back through a neighbour."""
    n = comp.shape[0]
    if n == 0:
        return []
    order = np.argsort(comp, kind="stable")                    # grouped by component, members ascending
    cs = comp[order]
    cut = np.flatnonzero(np.diff(cs)) + 1
    starts = np.concatenate([[0], cut])
    ends = np.concatenate([cut, [n]])
    size = ends - starts
    first = order[starts]
    void = (size == 1) & ~has_self_loop[first]                  # isolated node without a self loop -> []
    eff = np.where(void, 0, size)
    keep = np.flatnonzero(eff >= min_members)
    out = [(int(first[g]), [] if void[g] else order[starts[g]:ends[g]].tolist()) for g in keep]
    out.sort(key=lambda kv: kv[0])
    return [m for _, m in out]


def run_projection(ds: DeviceScene, cfg, debug_out: bool = False, timers=None, phases=None) -> Stage2Result:
    """P:402-634 for one uploaded scene.  `phases` (dict, diagnostic): wall time per phase with a device
    synchronize at every phase boundary."""
    import time
    _t = [time.perf_counter()]

    def mark(name):
        if phases is not None:
            torch.cuda.synchronize()
            now = time.perf_counter()
            phases[name] = phases.get(name, 0.0) + (now - _t[0])
            _t[0] = now

    dev = ds.xyz.device
    dbg = {}
    n, nw = ds.n_points, ds.nw
    do_ratio = (not cfg.if_occurance_threshold) and bool(cfg.if_detected_ratio_threshold)

    # a1: RLE -> per-pixel mask words (never the dense (M,1,H,W) tensors of P:400)
    n_mviews = ds.view_mask_offs.shape[0] - 1
    maskbits = torch.empty((n_mviews, ds.height * ds.width), device=dev,
                           dtype=torch.int32 if ds.word_bits == 32 else torch.int64)
    with span(timers, "rle_to_maskbits"):
        _lib.rle_to_maskbits(ds.run_start, ds.run_end, ds.mask_run_offs, ds.view_mask_offs, n_mviews,
                             ds.height * ds.width, ds.word_bits, maskbits)

    # a2-a8 (+a15): one fused sweep over the frames (P:413-461 and P:538-567)
    rows = torch.empty((ds.n_rows, nw), dtype=torch.int64, device=dev)
    masked = torch.zeros(n, dtype=torch.int32, device=dev)                          # P:402
    viewed = torch.zeros(n, dtype=torch.int32, device=dev) if do_ratio else None    # P:537
    n_frames = ds.n_frames if do_ratio else ds.n_mask_frames
    with span(timers, "project_views"):
        _lib.project_views(ds.xyz, n, ds.inv_pose[:n_frames], ds.cam_intr, ds.depth, ds.depth_index, ds.height,
                           ds.width, DEPTH_THRESH, maskbits if n_mviews else None, ds.word_bits, ds.frame_mask,
                           ds.frame_rowbase, ds.frame_nmask, ds.frame_flags, rows if ds.n_rows else None,
                           masked, viewed)
    del maskbits
    # a14/a15: point filter (P:512-583), entirely on the device: the threshold never visits the host
    if cfg.if_occurance_threshold or do_ratio:
        frac = cfg.detected_ratio_threshold if do_ratio else cfg.occurance_threshold
        thr_dev, lat_info = _lib.point_threshold(masked, viewed if do_ratio else None, frac)
        keep = _lib.ratio_keep(masked, viewed if do_ratio else None, thr_dev, True)
    else:
        thr_dev = lat_info = None
        keep = _lib.ratio_keep(masked, None, 0.0, False)
    mark("decode+sweep")
    # per-point arrays and bit rows are in the (spatially sorted) device point order; `unsorted` maps
    # bit rows back to the caller's point order
    unsorted = (lambda r: _lib.permute_bits(r, ds.unsort, n)) if ds.unsort is not None else (lambda r: r)
    if debug_out:
        back = (lambda v: v[ds.unsort.long()]) if ds.unsort is not None else (lambda v: v.clone())
        dbg["raw_rows"], dbg["masked_counts_raw"] = unsorted(rows), back(masked)
    if ds.n_rows == 0:                                                              # P:465-478
        return _empty(ds, dbg)

    # a9-a12: components of the IoU / label merge graph (P:100-146, 250-274) in one pass.  Rows are tiled
    # in the order (label, heavy-bin signature) so that a tile's rows occupy few chunks.
    with span(timers, "row_stats"):
        area, _mean_word, cmask, hist, sig = _lib.row_stats(rows)
        order = torch.argsort(sig, stable=True)
        if len(set(ds.labels)) > 1:               # several label strings: cluster by label first
            order = order[torch.argsort(ds.label_id[order], stable=True)]
        order = order.to(torch.int32)
    with span(timers, "merge_components"):
        comp = _lib.merge_components(rows, area, ds.label_id, cfg.iou_thres, order, cmask, hist)
        got = _lib.fetch(comp, area, *([lat_info, thr_dev] if lat_info is not None else []))   # one sync
        comp_h, area_h = got[0], got[1]
    if lat_info is not None:
        if got[2][0] == 0 or np.isnan(got[3][0]):
            raise IndexError("index out of range: unique()[floor(t * n)] (P:516 / P:574)")
        dbg["thr"] = float(got[3][0])
        if debug_out and do_ratio:
            dbg["viewed_counts"] = back(viewed)
    mark("stats+components")
    self_loop = (area_h > 0) & bool(np.float32(1.0) > np.float32(cfg.iou_thres))
    groups = groups_from_labels(comp_h, self_loop, cfg.min_aggragated_masks)        # P:203
    dbg["groups"] = groups
    merged = [g for g in groups if g != []]                                         # P:216-217
    if not merged:                                                                  # P:230-236, 496-509
        return _empty(ds, dbg)

    # a13: OR of member rows, sequential mean of confidences, label of the first member (P:214-226)
    offs = np.zeros(len(merged) + 1, dtype=np.int32)
    np.cumsum([len(g) for g in merged], out=offs[1:])
    members = np.concatenate([np.asarray(g, dtype=np.int32) for g in merged])
    offs_d, members_d = torch.from_numpy(offs).to(dev), torch.from_numpy(members).to(dev)
    agg = _lib.or_reduce_groups(rows, offs_d, members_d, max(len(g) for g in merged))
    conf = _lib.group_conf_mean(ds.conf, offs_d, members_d)
    agg_labels = [ds.labels[g[0]] for g in merged]
    mark("grouping+or_reduce")
    if not debug_out:
        del rows

    mark("point_filter")
    # a16: overlap resolution (P:592-596).  `groups` (not `merged`) indexes the sizes, as in the
    # reference where num_masks comes from mask_indeces_to_be_merged (P:285) -- identical unless
    # min_aggragated_masks == 0.
    before = _lib.popcount_rows(agg)                                                # P:592
    k = agg.shape[0]
    inter = _lib.cross_popcount(agg, agg).cpu().numpy()
    size = [len(g) for g in groups]
    ops = []
    for i in range(k):
        for j in range(i + 1, k):
            if inter[i, j] > 0:                                                     # P:291 (state before any edit)
                ops.append((0, j, i) if size[i] > size[j] else (0, i, j))           # P:296-299
    if ops:
        _lib.apply_row_ops(agg, torch.tensor(ops, dtype=torch.int32).to(dev))
    _lib.and_rows(agg, keep)                                                        # P:595
    after = _lib.popcount_rows(agg)                                                 # P:596
    mark("overlap")

    # a17: size filters with the reference's dtype promotion (int64 vs python scalars, P:601-606)
    before_t, after_t = before.cpu().to(torch.int64), after.cpu().to(torch.int64)
    keep_rows = (after_t > cfg.remove_small_masks) & (after_t > cfg.remove_filtered_masks * before_t)
    idx = torch.nonzero(keep_rows).view(-1).to(torch.int32)
    dbg.update(before=before_t, after=after_t, keep=keep_rows)
    out_rows = unsorted(_lib.gather_rows(agg, idx.to(dev))) if idx.numel() else agg[:0]
    out_conf = conf[keep_rows.to(dev)]
    out_labels = [c for c, kk in zip(agg_labels, keep_rows.tolist()) if kk]
    mark("size_filter+output")
    return Stage2Result(ds.scene_id, n, out_rows, out_conf, out_labels, groups, dbg)


def project_scene(scene, cfg, device="cuda", return_result: bool = False, debug_out: bool = False):
    """Reference-shaped entry point: in-memory scene inputs -> the dict saved at P:630-634."""
    _lib.load()                                     # fail loudly before any work if the library is missing
    ds = prepare_scene(scene, cfg, device=device,
                       with_viewed=(not cfg.if_occurance_threshold) and bool(cfg.if_detected_ratio_threshold))
    res = run_projection(ds, cfg, debug_out=debug_out)
    return res if return_result else res.to_dict()
