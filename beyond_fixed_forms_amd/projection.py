"""Stage 2 of the pipeline on MI355X: 2-D masks -> aggregated, filtered 3-D instance masks.

Host-side mirror of the per-scene loop of the reference (tools/projection_2d_to_3d.py:365-634).
All per-point / per-pixel / per-pair arithmetic runs in libbff_hip.so (include/bff_hip.h); the host
keeps only what the reference keeps sequential and tiny: grouping component labels into the
reference's list order, the order-dependent overlap decisions (P:295-299), thresholds picked from
sets of distinct values, and dict assembly.  `P:` = tools/projection_2d_to_3d.py.
"""
from __future__ import annotations

import dataclasses
from typing import List, Optional

import numpy as np
import torch

import os

from . import _lib, pipeline
from .scene import DEPTH_THRESH, DeviceScene, prepare_scene
from .timing import merge_span, span, sweep_span


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
    conf_host: Optional[torch.Tensor] = None      # `conf` again, on the host (read back with the last fetch)
    prefetch: Optional[dict] = None               # refinement inputs read back with the same fetch (see projection_back)

    @property
    def empty(self):
        return self.rows.shape[0] == 0 and self.debug.get("empty_form", False)

    def to_dict(self):
        if self.debug.get("empty_form", False):
            dev = self.rows.device
            return {"ins": torch.tensor([[]]).to(dev), "conf": torch.tensor([]).to(dev), "final_class": []}
        return {"ins": _lib.unpack_rows(self.rows, self.n_points), "conf": self.conf,
                "final_class": list(self.final_class)}

    def to_rle_dict(self):
        """Same result with "ins" as a list of RLE dicts (the Open3DIS stage-1 storage format,
        rle_encode_batch RLE:10-32), encoded on the device."""
        return {"ins": _lib.rows_to_rle(self.rows, self.n_points), "conf": self.conf,
                "final_class": list(self.final_class)}


class _LazyGroups(list):
    """The reference's mask_indeces_to_be_merged (list of lists), materialised from CSR on first use --
    building ~10^4 Python ints per scene is not needed on the hot path."""

    def __init__(self, offs, members, fetch=None):
        super().__init__()
        self._csr = (offs, members)
        self._dev = fetch                     # (offs, members) device tensors, read back on first use

    def _fill(self):
        if self._dev is not None:
            offs, members = self._dev
            self._dev = None
            self._csr = (offs.cpu().numpy(), members.cpu().numpy())
        if self._csr is not None:
            offs, members = self._csr
            self._csr = None
            super().extend(members[offs[g]:offs[g + 1]].tolist() for g in range(len(offs) - 1))

    def __len__(self):
        if self._dev is not None:
            return self._dev[0].shape[0] - 1
        return len(self._csr[0]) - 1 if self._csr is not None else super().__len__()

    def __iter__(self):
        self._fill(); return super().__iter__()

    def __getitem__(self, i):
        self._fill(); return super().__getitem__(i)

    def __eq__(self, other):
        self._fill()
        if isinstance(other, _LazyGroups):
            other._fill()
        return list.__eq__(self, other)

    __hash__ = None

    def __repr__(self):
        self._fill(); return super().__repr__()


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


def component_csr(comp: np.ndarray, has_self_loop: np.ndarray, min_members: int):
    """Component ids -> the groups merge_masks keeps (P:203-226) in CSR form, without Python lists:
    (offs int32 [K+1], members int32, sizes int32 [K], n_void).  Same order and content as
    groups_from_labels(...): components by smallest member, members ascending; a singleton without a
    self loop is the reference's empty list `[]` -- it has length 0, so it survives the filter only for
    min_members <= 0, where it is skipped when merging (P:216-217) but still occupies a slot of
    mask_indeces_to_be_merged; n_void reports how many there are (callers treat that rare config on the
    slow path)."""
    n = comp.shape[0]
    order = np.argsort(comp, kind="stable")
    cs = comp[order]
    cut = np.flatnonzero(cs[1:] != cs[:-1]) + 1
    starts = np.concatenate([np.zeros(1, np.int64), cut])
    ends = np.concatenate([cut, np.array([n], np.int64)])
    size = ends - starts
    first = order[starts]
    void = (size == 1) & ~has_self_loop[first]
    keep = ~void & (size >= max(min_members, 1))
    rank = np.argsort(first[keep], kind="stable")            # groups ordered by their smallest member
    st, en = starts[keep][rank], ends[keep][rank]
    sizes = (en - st).astype(np.int32)
    offs = np.zeros(sizes.size + 1, dtype=np.int32)
    np.cumsum(sizes, out=offs[1:])
    # members = concatenation of order[st[g]:en[g]]
    idx = np.repeat(st - offs[:-1], sizes) + np.arange(offs[-1])
    members = order[idx].astype(np.int32)
    return offs, members, sizes, int(void.sum()) if min_members <= 0 else 0


@dataclasses.dataclass
class _Front:
    """Everything the GPU-only first half of a scene leaves behind (all device tensors live on the stream the
    front was issued on)."""
    ds: DeviceScene
    cfg: object
    dbg: dict
    debug_out: bool
    rows: Optional[torch.Tensor] = None
    masked: Optional[torch.Tensor] = None
    viewed: Optional[torch.Tensor] = None
    keep: Optional[torch.Tensor] = None
    thr_dev: Optional[torch.Tensor] = None
    lat_info: Optional[torch.Tensor] = None
    area: Optional[torch.Tensor] = None
    comp: Optional[torch.Tensor] = None
    cmask: Optional[torch.Tensor] = None       # chunk occupancy flags of `rows` (what the sweep stored into)
    arena: Optional[object] = None             # _lib.RowArena that lent `rows` (None: a tensor of its own)
    do_ratio: bool = False
    fast: Optional[dict] = None                # handle of pipeline.issue: the whole device side is already in flight
    stage1: Optional[object] = None            # refinement.DeviceStage1 whose first device pass rides along


def fast_path_ok(ds: DeviceScene) -> bool:
    """bff_scene_project handles every scene with masks and frames; BFF_NO_FAST=1 forces the step-by-step path."""
    return (ds.n_rows > 0 and ds.n_points > 0 and ds.view_mask_offs.shape[0] > 1 and ds.n_mask_frames > 0
            and ds.xyz.is_cuda and os.environ.get("BFF_NO_FAST") != "1")


def projection_front(ds: DeviceScene, cfg, debug_out: bool = False, timers=None, stage1=None, fast=None) -> _Front:
    """Issue the device work of P:402-634 for one uploaded scene.  Nothing here waits for the GPU, so a caller can
    issue the next scene on another stream while the host finishes this one (`projection_back`).

    Fast form (default unless debug_out): ONE native call (bff_scene_project) enqueues everything -- RLE decode,
    fused sweep, point filter, row statistics, components, the groups of P:203-226 formed on the device, OR of the
    members, overlap resolution, size-filter counts and, when `stage1` (a refinement.DeviceStage1 of the same
    scene) is given, the refinement's first device pass (R:186-217) -- and ends with an asynchronous copy of a
    small header to the host.  Step-by-step form (debug_out, or fast=False): decode, sweep, threshold, statistics
    and components are issued one by one and `projection_back` continues on the host."""
    if fast is None:
        fast = not debug_out
    if fast and fast_path_ok(ds) and (stage1 is None or stage1.n_points == ds.n_points):
        do_ratio = (not cfg.if_occurance_threshold) and bool(cfg.if_detected_ratio_threshold)
        fr = _Front(ds, cfg, {}, False, do_ratio=do_ratio, stage1=stage1)
        with sweep_span(timers, "project_views"), merge_span(timers, "merge_components"):
            fr.fast = pipeline.issue(ds, cfg, DEPTH_THRESH, stage1, ds.n_frames if do_ratio else ds.n_mask_frames)
        return fr
    with _lib.launch_stream():
        fr = _projection_front(ds, cfg, debug_out, timers)
        fr.stage1 = stage1
        return fr


def _projection_front(ds, cfg, debug_out, timers) -> _Front:
    dev = ds.xyz.device
    fr = _Front(ds, cfg, {}, debug_out)
    n, nw = ds.n_points, ds.nw
    do_ratio = fr.do_ratio = (not cfg.if_occurance_threshold) and bool(cfg.if_detected_ratio_threshold)

    # a1: RLE -> per-pixel mask words (never the dense (M,1,H,W) tensors of P:400)
    n_mviews = ds.view_mask_offs.shape[0] - 1
    maskbits = torch.empty((n_mviews, ds.height * ds.width), device=dev,
                           dtype=torch.int32 if ds.word_bits == 32 else torch.int64)
    segmap = torch.empty((n_mviews, _lib.segmap_words(ds.height * ds.width)), dtype=torch.int32, device=dev)
    with span(timers, "rle_to_maskbits"):
        _lib.rle_to_maskbits(ds.run_start, ds.run_end, ds.mask_run_offs, ds.view_mask_offs, n_mviews,
                             ds.height * ds.width, ds.word_bits, maskbits, segmap)

    # a2-a8 (+a15): one fused sweep over the frames (P:413-461 and P:538-567)
    # the sweep stores only the sectors that receive a point, so the rows start out zero: a view of the stream's
    # recycled zero arena (the back half clears what this scene stored), fresh zeros when the rows are handed out
    with span(timers, "zero_rows"):
        if ds.n_rows and n_mviews and not debug_out:
            fr.arena = _lib.RowArena.for_current_stream(dev)
            rows = fr.rows = fr.arena.take(ds.n_rows, nw, dev)
        else:
            rows = fr.rows = torch.zeros((ds.n_rows, nw), dtype=torch.int64, device=dev)
    masked = fr.masked = torch.zeros(n, dtype=torch.int32, device=dev)                          # P:402
    viewed = fr.viewed = torch.zeros(n, dtype=torch.int32, device=dev) if do_ratio else None    # P:537
    n_frames = ds.n_frames if do_ratio else ds.n_mask_frames
    # the sweep flags, per row, the 512-point chunks it stores into: the later passes read nothing else
    cmask_in = fr.cmask = _lib.chunk_mask_buffer(ds.n_rows, nw, dev).zero_() if (ds.n_rows and n_mviews) else None
    with sweep_span(timers, "project_views"):
        _lib.project_views(ds.xyz, n, ds.inv_pose[:n_frames], ds.cam_intr, ds.sweep_depth, ds.depth_index, ds.height,
                           ds.width, DEPTH_THRESH, maskbits if n_mviews else None, ds.word_bits, ds.frame_mask,
                           ds.frame_rowbase, ds.frame_nmask, ds.frame_flags, rows if ds.n_rows else None,
                           masked, viewed, segmap if n_mviews else None, cmask_in, ds.tile_bounds, depth_size=ds.depth_size)
    del maskbits
    # a14/a15: point filter (P:512-583), entirely on the device: the threshold never visits the host
    if cfg.if_occurance_threshold or do_ratio:
        frac = cfg.detected_ratio_threshold if do_ratio else cfg.occurance_threshold
        fr.thr_dev, fr.lat_info = _lib.point_threshold(masked, viewed if do_ratio else None, frac)
        fr.keep = _lib.ratio_keep(masked, viewed if do_ratio else None, fr.thr_dev, True)
    else:
        fr.keep = _lib.ratio_keep(masked, None, 0.0, False)
    if ds.n_rows == 0:                                                              # P:465-478
        return fr

    # a9-a12: components of the IoU / label merge graph (P:100-146, 250-274) in one pass.  Rows are tiled
    # in the order (label, heavy-bin signature) so that a tile's rows occupy few chunks.
    with span(timers, "row_stats"):
        area, _mean_word, cmask, hist, sig = _lib.row_stats(rows, cmask_in)
        order = _lib.argsort_i64(sig, _lib.SIGNATURE_BITS)
        if ds.n_label_ids > 1:                    # several label strings: cluster by label first (stable on top)
            order = order[_lib.argsort_i64(ds.label_id[order.long()].to(torch.int64), 32).long()].contiguous()
    with merge_span(timers, "merge_components"):
        fr.comp = _lib.merge_components(rows, area, ds.label_id, cfg.iou_thres, order, cmask, hist)
    fr.area = area
    return fr


def projection_back(fr: _Front, timers=None, phases=None, stage1=None, want_groups: bool = True) -> Stage2Result:
    """Host half: wait for the device, apply the size filters (P:601-606), select, assemble (P:608-634).  Must run
    on the stream the front was issued on.

    stage1 (optional, a refinement.DeviceStage1 of the same scene; normally given to projection_front already):
    when the refinement of this class follows in the same process, its first device pass (stage-1 decode, areas,
    stage-1 x stage-2 and stage-1 x stage-1 intersections, R:186-217) is read back with this stage's fetch, which
    saves the refinement a synchronisation; `refine_class` picks it up from the result when given the same object.
    want_groups=False: skip keeping the reference's mask_indeces_to_be_merged lists (diagnostic output)."""
    if stage1 is None:
        stage1 = fr.stage1
    with _lib.launch_stream():
        if fr.fast is not None:
            return _fast_back(fr, stage1, want_groups)
        return _projection_back(fr, timers, phases, stage1)


class _WsRows:
    """RowArena look-alike for the general path running on a bff_scene_project workspace: release() clears the
    chunks the sweep stored, which is what the fast path would have done on the device."""

    def __init__(self, ws):
        self.ws = ws

    def release(self, rows, cmask):
        _lib.call("bff_clear_flagged_chunks", _lib._ptr(rows, torch.int64), rows.shape[0], rows.shape[1],
                  _lib._ptr(cmask, torch.int64))
        self.ws.rows_dirty = False


def _fast_back(fr: _Front, stage1, want_groups) -> Stage2Result:
    from .pipeline import HDR_K, HDR_NUNIQUE, HDR_THR, hdr_offsets
    ds, cfg, dbg = fr.ds, fr.cfg, fr.dbg
    h = fr.fast
    hdr = pipeline.collect(h)                                        # the one synchronisation of the scene
    ws, both, s1_rows = h["ws"], h["both"], h["s1_rows"]             # (after it: a re-issued scene has a new `both`)
    GROUP_CAP = h["cap"]
    HDR_SIZES, HDR_FIRST, HDR_BEFORE, HDR_AFTER, HDR_CONF, HDR_CROSS = hdr_offsets(GROUP_CAP)
    dev = ds.xyz.device
    n, nw = ds.n_points, ds.nw
    k_all, flags = int(hdr[HDR_K]), int(hdr[HDR_K + 1])
    filtered = h["params"].filter_mode != 0
    if filtered:
        thr = float(hdr[HDR_THR:HDR_THR + 1].view(np.float32)[0])
        if hdr[HDR_NUNIQUE] == 0 or np.isnan(thr):
            ws.rows_dirty = True
            raise IndexError("index out of range: unique()[floor(t * n)] (P:516 / P:574)")
        dbg["thr"] = thr
    if flags:
        # more groups than the device forms by itself (or min_aggragated_masks <= 0 with empty components): the
        # group tables are incomplete -- continue from the components on the host, exactly as the step-by-step path
        mw = max(_lib.load().bff_chunk_mask_words(nw), 1)
        hd = ws.t["hdr"]
        slow = _Front(ds, cfg, dbg, False, rows=ws.view("rows", ds.n_rows, nw), masked=ws.view("masked", n),
                      viewed=ws.view("viewed", n) if fr.do_ratio else None, keep=ws.view("keep", nw),
                      thr_dev=hd[HDR_THR:HDR_THR + 1].view(torch.float32) if filtered else None,
                      lat_info=hd[HDR_NUNIQUE:HDR_NUNIQUE + 1] if filtered else None,
                      area=ws.view("area", ds.n_rows), comp=ws.view("comp", ds.n_rows),
                      cmask=ws.view("chunk_mask", ds.n_rows, mw), arena=_WsRows(ws), do_ratio=fr.do_ratio)
        dbg["path"] = "general (device tables incomplete: %d groups, flags %d)" % (k_all, flags)
        return _projection_back(slow, None, None, stage1 if (stage1 is not None and stage1.n_points == n) else None)
    dbg["path"] = "fast"
    k = k_all
    if k == 0:                                                                      # P:230-236, 496-509
        dbg["groups"] = []
        return _empty(ds, dbg)
    sizes = hdr[HDR_SIZES:HDR_SIZES + k]
    first = hdr[HDR_FIRST:HDR_FIRST + k]
    before, after = hdr[HDR_BEFORE:HDR_BEFORE + k], hdr[HDR_AFTER:HDR_AFTER + k]
    conf_np = hdr[HDR_CONF:HDR_CONF + GROUP_CAP].view(np.float16 if ds.conf.dtype == torch.float16 else np.float32)[:k]
    # a17: size filters with the reference's dtype promotion (P:601-606): `after > 5` on integers; `after > 0.4 *
    # before` is an int64 tensor against python-float * int64 tensor = float32 arithmetic on both sides
    keep_rows = (after > cfg.remove_small_masks) & \
        (after.astype(np.float32) > np.float32(cfg.remove_filtered_masks) * before.astype(np.float32))
    sel = np.flatnonzero(keep_rows).astype(np.int32)
    dbg.update(before=torch.from_numpy(before.astype(np.int64)), after=torch.from_numpy(after.astype(np.int64)),
               keep=torch.from_numpy(keep_rows))
    agg_u = both[:GROUP_CAP]
    if sel.size:
        # one upload for the row selection and the kept confidences (raw bytes behind the int32 indices)
        conf_sel = np.ascontiguousarray(conf_np[sel])
        n_sel, cw = sel.size, (conf_sel.nbytes + 3) // 4
        packed = np.zeros(n_sel + cw, dtype=np.int32)
        packed[:n_sel] = sel
        packed[n_sel:].view(np.uint8)[:conf_sel.nbytes] = conf_sel.view(np.uint8)
        packed_d = _lib.upload(packed, torch.int32, dev)
        out_rows = _lib.gather_rows(agg_u, packed_d[:n_sel])
        conf_host = torch.from_numpy(conf_sel.copy())
        out_conf = packed_d[n_sel:].view(ds.conf.dtype)[:n_sel]
    else:
        out_rows = agg_u[:0]
        conf_host = torch.from_numpy(conf_np[:0].copy())
        out_conf = torch.zeros(0, dtype=ds.conf.dtype, device=dev)
    labels = ds.labels
    out_labels = [labels[first[g]] for g in sel]
    pre = None
    if s1_rows and stage1 is not None:
        cross = hdr[HDR_CROSS:HDR_CROSS + s1_rows * (GROUP_CAP + s1_rows)].reshape(s1_rows, GROUP_CAP + s1_rows)
        inter11 = np.ascontiguousarray(cross[:, GROUP_CAP:])
        pre = dict(stage1=stage1, s1=both[GROUP_CAP:], area1=np.ascontiguousarray(np.diagonal(inter11)),
                   area2=after[sel], inter=np.ascontiguousarray(cross[:, sel]), inter11=inter11)
    if want_groups:
        # mask_indeces_to_be_merged (P:203-247) stays on the device until someone looks at it
        total = int(sizes.sum())
        groups = _LazyGroups(None, None, fetch=(ws.view("goffs", k + 1).clone(), ws.view("gmembers", total).clone()))
    else:
        groups = _LazyGroups(np.zeros(1, np.int32), np.zeros(0, np.int32))
    dbg["groups"] = groups
    return Stage2Result(ds.scene_id, n, out_rows, out_conf, out_labels, groups, dbg, conf_host, pre)


def run_projection(ds: DeviceScene, cfg, debug_out: bool = False, timers=None, phases=None, stage1=None) -> Stage2Result:
    """P:402-634 for one uploaded scene.  `phases` (dict, diagnostic): wall time per phase with a device
    synchronize at every phase boundary."""
    if phases is not None:
        import time
        t0 = time.perf_counter()
    fr = projection_front(ds, cfg, debug_out, timers, stage1=stage1, fast=False if phases is not None else None)
    if phases is not None:
        torch.cuda.synchronize()
        phases["front (decode, sweep, threshold, stats, components)"] = \
            phases.get("front (decode, sweep, threshold, stats, components)", 0.0) + time.perf_counter() - t0
    return projection_back(fr, timers, phases, stage1)


def _recycle_rows(fr: _Front):
    """Hand the raw rows back: a borrowed arena view is cleared where the sweep stored (stream-ordered)."""
    if fr.arena is not None and fr.rows is not None:
        fr.arena.release(fr.rows, fr.cmask)
    fr.rows = fr.arena = None


def _projection_back(fr: _Front, timers, phases, stage1=None) -> Stage2Result:
    import time
    _t = [time.perf_counter()]

    def mark(name):
        if phases is not None:
            torch.cuda.synchronize()
            now = time.perf_counter()
            phases[name] = phases.get(name, 0.0) + (now - _t[0])
            _t[0] = now

    ds, cfg, dbg, debug_out = fr.ds, fr.cfg, fr.dbg, fr.debug_out
    dbg.setdefault("path", "step")
    dev = ds.xyz.device
    n = ds.n_points
    rows, masked, viewed, keep, do_ratio = fr.rows, fr.masked, fr.viewed, fr.keep, fr.do_ratio
    # per-point arrays and bit rows are in the (spatially sorted) device point order; `unsorted` maps
    # bit rows back to the caller's point order
    if ds.unsort is None:
        unsorted = lambda r: r
    elif ds.perm is not None:           # scatter of the set bits: aggregated rows hold a few percent of the points
        unsorted = lambda r: _lib.scatter_bits(r, ds.perm, n)
    else:
        unsorted = lambda r: _lib.permute_bits(r, ds.unsort, n)
    back = (lambda v: v[ds.unsort.long()]) if ds.unsort is not None else (lambda v: v.clone())
    if debug_out:
        dbg["raw_rows"], dbg["masked_counts_raw"] = unsorted(rows), back(masked)
    if ds.n_rows == 0:                                                              # P:465-478
        return _empty(ds, dbg)
    got = _lib.fetch(fr.comp, fr.area, *([fr.lat_info, fr.thr_dev] if fr.lat_info is not None else []))   # one sync
    comp_h, area_h = got[0], got[1]
    if fr.lat_info is not None:
        if got[2][0] == 0 or np.isnan(got[3][0]):
            raise IndexError("index out of range: unique()[floor(t * n)] (P:516 / P:574)")
        dbg["thr"] = float(got[3][0])
        if debug_out and do_ratio:
            dbg["viewed_counts"] = back(viewed)
    mark("read-back")
    self_loop = (area_h > 0) & bool(np.float32(1.0) > np.float32(cfg.iou_thres))
    csr = _lib.host_component_csr(comp_h, self_loop, cfg.min_aggragated_masks)                    # P:203
    offs, members, sizes, n_void = csr if csr is not None else component_csr(comp_h, self_loop, cfg.min_aggragated_masks)
    if n_void:      # min_aggragated_masks <= 0 keeps empty components in the size list (P:285): exact, slower path
        groups = groups_from_labels(comp_h, self_loop, cfg.min_aggragated_masks)
        size_list = [len(g) for g in groups]
    else:
        groups, size_list = None, None
    k_groups = sizes.shape[0]
    if k_groups == 0:                                                               # P:230-236, 496-509
        dbg["groups"] = [] if groups is None else groups
        _recycle_rows(fr)
        return _empty(ds, dbg)

    # a13: OR of member rows, sequential mean of confidences, label of the first member (P:214-226)
    # one upload for the three group tables (offsets, members, sizes)
    n_offs, n_mem = offs.shape[0], members.shape[0]
    packed = _lib.upload(np.concatenate([offs, members, sizes.astype(np.int32, copy=False)]), torch.int32, dev)
    offs_d, members_d, sizes_d = packed[:n_offs], packed[n_offs:n_offs + n_mem], packed[n_offs + n_mem:]
    agg, conf = _lib.or_reduce_groups(rows, offs_d, members_d, int(sizes.max()), ds.conf, chunk_mask=fr.cmask)
    first_member = members[offs[:-1]]
    agg_labels = [ds.labels[i] for i in first_member]
    if not debug_out:
        del rows
        _recycle_rows(fr)                         # last reader done: the arena gets its zeros back
    mark("grouping+or_reduce")

    # a16: overlap resolution (P:592-596), decided and applied on the device
    if size_list is not None:   # sizes indexed like mask_indeces_to_be_merged, which still holds the empty components (P:285)
        sizes_d = _lib.upload(np.asarray(size_list[:agg.shape[0]], dtype=np.int32), torch.int32, dev)
    before, after = _lib.resolve_overlaps_filtered(agg, sizes_d, keep)               # P:592-596
    mark("overlap")

    # a17: size filters with the reference's dtype promotion (int64 vs python scalars, P:601-606)
    agg_u = pre = None
    if stage1 is not None and stage1.n_points == n:
        # the refinement's first device pass rides on this fetch: rows in the caller's point order for all K
        # groups (the selection below only picks among them);
        # one buffer [K groups | S1 stage-1 rows] so that one intersection launch serves both products
        k_all, s1_n = agg.shape[0], stage1.row_run_offs.shape[0] - 1
        both = torch.empty((k_all + s1_n, ds.nw), dtype=torch.int64, device=dev)
        agg_u, s1 = both[:k_all], both[k_all:]
        if ds.unsort is None:
            agg_u.copy_(agg)
        elif ds.perm is not None:
            agg_u.zero_()
            _lib.scatter_bits(agg, ds.perm, n, out=agg_u)
        else:
            _lib.permute_bits(agg, ds.unsort, n, out=agg_u)
        _lib.rle_to_rows(stage1.run_start, stage1.run_end, stage1.row_run_offs, n, out=s1)
        cross = _lib.cross_popcount(s1, both)                        # [S1][K + S1]
        before_h, after_h, conf_h, cross_h = _lib.fetch(before, after, conf, cross)
        inter_h, inter11_h = cross_h[:, :k_all], cross_h[:, k_all:]
        area1_h = np.ascontiguousarray(np.diagonal(inter11_h))       # |row & row| = its area
    else:
        before_h, after_h, conf_h = _lib.fetch(before, after, conf)                 # the second (last) sync
    before_t, after_t = torch.from_numpy(before_h).to(torch.int64), torch.from_numpy(after_h).to(torch.int64)
    keep_rows = (after_t > cfg.remove_small_masks) & (after_t > cfg.remove_filtered_masks * before_t)
    idx = torch.nonzero(keep_rows).view(-1).to(torch.int32)
    dbg.update(before=before_t, after=after_t, keep=keep_rows)
    idx_d = _lib.upload(idx, torch.int32, dev)
    if agg_u is not None:
        out_rows = _lib.gather_rows(agg_u, idx_d) if idx.numel() else agg_u[:0]
        sel = idx.long().numpy()
        pre = dict(stage1=stage1, s1=s1, area1=area1_h, area2=after_h[sel], inter=np.ascontiguousarray(inter_h[:, sel]),
                   inter11=inter11_h)
    else:
        out_rows = unsorted(_lib.gather_rows(agg, idx_d)) if idx.numel() else agg[:0]
    out_conf = conf[idx_d.long()]
    conf_host = torch.from_numpy(conf_h)[idx.long()]                # the same values, already on the host
    out_labels = [c for c, kk in zip(agg_labels, keep_rows.tolist()) if kk]
    mark("size_filter+output")
    lazy = _LazyGroups(offs, members) if groups is None else groups
    dbg["groups"] = lazy
    return Stage2Result(ds.scene_id, n, out_rows, out_conf, out_labels, lazy, dbg, conf_host, pre)


def project_scene(scene, cfg, device="cuda", return_result: bool = False, debug_out: bool = False):
    """Reference-shaped entry point: in-memory scene inputs -> the dict saved at P:630-634."""
    _lib.load()                                     # fail loudly before any work if the library is missing
    with_viewed = (not cfg.if_occurance_threshold) and bool(cfg.if_detected_ratio_threshold)
    if torch.device(device).type == "cuda" and not debug_out:
        from .ingest import prepare_scene_fast      # native run tables, packed pinned uploads, cloud laid out on the device
        ds = prepare_scene_fast(scene, cfg, device=device, with_viewed=with_viewed)
    else:
        ds = prepare_scene(scene, cfg, device=device, with_viewed=with_viewed)
    res = run_projection(ds, cfg, debug_out=debug_out)
    return res if return_result else res.to_dict()
