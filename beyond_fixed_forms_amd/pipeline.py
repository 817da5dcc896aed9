"""One native call per scene: ctypes mirrors of bff_scene / bff_scene_params / bff_scene_workspace
(include/bff_hip.h) and the scratch they point to.

`bff_scene_project` issues every device step of the projection stage (P:402-634) and the refinement's first device
pass (R:186-217) on one stream and ends with an asynchronous copy of a small header into pinned host memory.  The
host thread's share of a scene is then: fill nothing (the structs are cached per scene), one call, and -- after the
stream has caught up -- a few NumPy lines over the header (`projection._fast_back`)."""
from __future__ import annotations

import ctypes
import os
import time
from ctypes import c_double, c_float, c_int32, c_int64, c_size_t, c_void_p

import numpy as np
import torch

from . import _lib

GROUP_CAP = 256            # kept groups the device forms by itself, by default ...
GROUP_CAP_MAX = 512        # ... and for scenes that were seen to keep more (re-issued once with the larger tables)
HDR_K, HDR_NUNIQUE, HDR_THR, HDR_OVERFLOW = 0, 4, 5, 6


def hdr_offsets(cap):
    """(sizes, first, before, after, conf, cross) word offsets of the header for a workspace with `cap` group slots
    (BFF_HDR_* of include/bff_hip.h)."""
    return tuple(16 + k * cap for k in range(6))


class SceneStruct(ctypes.Structure):
    _fields_ = [("n_points", c_int64), ("n_pad", c_int64), ("nw", c_int64),
                ("xyz", c_void_p), ("tile_bounds", c_void_p), ("inv_pose", c_void_p), ("cam_intr", c_double * 9),
                ("depth", c_void_p), ("depth_raw", c_void_p),
                ("depth_index", c_void_p), ("frame_mask", c_void_p), ("frame_rowbase", c_void_p),
                ("frame_nmask", c_void_p), ("frame_flags", c_void_p),
                ("run_start", c_void_p), ("run_end", c_void_p), ("mask_run_offs", c_void_p), ("view_mask_offs", c_void_p),
                ("conf", c_void_p), ("label_id", c_void_p), ("unsort", c_void_p), ("perm", c_void_p),
                ("s1_run_start", c_void_p), ("s1_run_end", c_void_p), ("s1_row_run_offs", c_void_p),
                ("height", c_int32), ("width", c_int32), ("n_frames", c_int32), ("n_mviews", c_int32),
                ("word_bits", c_int32), ("n_rows", c_int32), ("conf_f16", c_int32), ("n_label_ids", c_int32),
                ("s1_rows", c_int32), ("depth_h", c_int32), ("depth_w", c_int32), ("depth_tiled", c_int32)]


class ParamsStruct(ctypes.Structure):
    _fields_ = [("depth_thresh", c_double), ("filter_fraction", c_double), ("iou_thres", c_float),
                ("min_members", c_int32), ("filter_mode", c_int32), ("filter_sort", c_int32)]


_WS_PTRS = ["maskbits", "segmap", "labels", "rows", "chunk_mask", "keep", "tile_mask", "agg", "both",
            "masked", "viewed", "sel_scratch", "area", "mean_word", "order", "parent", "comp", "count",
            "gmembers", "goffs", "slices", "pair_scratch", "vals", "vals_sorted", "hist", "merge_scratch", "chunk_pop",
            "sig", "sig_keys", "sig_sorted", "sort_temp"]


class WorkspaceStruct(ctypes.Structure):
    _fields_ = [(n, c_void_p) for n in _WS_PTRS] + [("sort_temp_bytes", c_size_t), ("zero_bytes", c_size_t),
                                                      ("hdr", c_void_p), ("hdr_host", c_void_p),
                                                      ("group_cap", c_int32), ("pad_", c_int32),
                                                      ("heavy_stream", c_void_p), ("events", c_void_p * 4),
                                                      ("aux_stream", c_void_p), ("aux_events", c_void_p * 2)]


# Scenes in flight on the device, one HIP stream (and one SceneWorkspace) each.  A scene's device work is a chain of
# ~50 launches of which three fill the chip; the rest of the chain only overlaps with OTHER scenes' kernels.  Measured
# on config 2 (bench.py --depth): 2 -> 1.31 ms per scene, 3 -> 1.09, 4 -> 1.02, 6 -> 1.18 (the host thread, which
# issues and collects every scene, and the chip are then both busy ~70 % of the time).
PIPELINE_DEPTH = 4

# BFF_HEAVY_STREAMS=k > 0: the three chip-filling kernels of a scene (decode, sweep, tile pass) go to one of k shared
# streams, so that at most k of them run side by side while the scenes' chains of small kernels overlap freely (with four
# scenes in flight on four streams they run up to four at a time, each at a fraction of its speed: config 2, the sweep
# 0.30 ms alone, 0.6-0.8 ms in the loop).  Measured at config 2 (scenes/s): off 1047, k = 3: 875, 2: 822, 1: 760 -- every
# one of those kernels leaves the chip half idle on its own (the sweep's waves sit on gathers 2/3 of their time) and the
# others fill it: OFF by default (0: everything on the scene's own stream).
HEAVY_STREAMS = int(os.environ.get("BFF_HEAVY_STREAMS", "0"))
# BFF_AUX_STREAM=1: a second stream per workspace for the point filter's threshold chain (fork after the sweep, join
# before the overlap resolution).  It shortens a scene's device span by 0.04 ms and costs a sixth of the throughput
# (config 2: 961-984 vs 1149-1155 scenes/s: eight streams contend for the hardware queues): OFF by default.
AUX_STREAM = os.environ.get("BFF_AUX_STREAM", "0") == "1"

_TRACE_ISSUE = bool(os.environ.get("BFF_TRACE_ISSUE"))       # report scene calls that take more than 2 ms to enqueue
_scene_streams = {}
_heavy_streams = {}


def heavy_stream(device, k):
    """The k-th (mod HEAVY_STREAMS) heavy stream of a device, created once."""
    if HEAVY_STREAMS <= 0:
        return None
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    have = _heavy_streams.setdefault(idx, [])
    while len(have) < HEAVY_STREAMS:
        have.append(torch.cuda.Stream(device=dev))
    return have[k % HEAVY_STREAMS]


def scene_streams(device, depth=None):
    """The PIPELINE_DEPTH HIP streams scenes are issued on, created once per device: a SceneWorkspace (gigabytes) belongs
    to a stream and lives as long as the process, so callers that run one class after the other must come back to the
    same streams instead of creating new ones."""
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    depth = PIPELINE_DEPTH if depth is None else depth
    have = _scene_streams.setdefault(idx, [])
    while len(have) < depth:
        have.append(torch.cuda.Stream(device=dev))
    return have[:depth]


_checked = False


def _check_layout():
    global _checked
    if _checked:
        return
    lib = _lib.load()
    for which, cls in enumerate((SceneStruct, ParamsStruct, WorkspaceStruct)):
        if lib.bff_scene_struct_bytes(which) != ctypes.sizeof(cls):
            raise _lib.BffLibraryError(f"{cls.__name__}: {ctypes.sizeof(cls)} bytes here, "
                                       f"{lib.bff_scene_struct_bytes(which)} in libbff_hip.so (rebuild)")
    _checked = True


def _p(t):
    return None if t is None else c_void_p(t.data_ptr())


def scene_struct(ds, stage1=None, n_frames=None):
    """bff_scene of an uploaded scene (+ optionally its resident stage-1 run tables).  The struct only holds
    pointers: `ds` / `stage1` must stay alive while it is used (callers keep it next to them)."""
    _check_layout()
    s = SceneStruct()
    s.n_points, s.n_pad, s.nw = ds.n_points, ds.xyz.shape[1], ds.nw
    s.xyz, s.tile_bounds, s.inv_pose = _p(ds.xyz), _p(ds.tile_bounds), _p(ds.inv_pose)
    s.cam_intr = (c_double * 9)(*[float(v) for v in np.asarray(ds.cam_intr).reshape(-1)])
    if ds.depth_raw is not None:
        s.depth, s.depth_raw = None, _p(ds.depth_raw)
        hs, ws = ds.depth_size if ds.depth_size is not None else ds.depth_raw.shape[1:3]
        s.depth_h, s.depth_w = int(hs), int(ws)
        s.depth_tiled = 0 if ds.depth_size is None else (2 if ds.depth_raw.dtype == torch.float32 else 1)
    else:
        s.depth, s.depth_raw = _p(ds.depth), None
    for k in ("depth_index", "frame_mask", "frame_rowbase", "frame_nmask", "frame_flags", "run_start", "run_end",
              "mask_run_offs", "view_mask_offs", "label_id"):
        setattr(s, k, _p(getattr(ds, k)))
    s.conf, s.unsort, s.perm = _p(ds.conf), _p(ds.unsort), _p(ds.perm)
    if ds.unsort is not None and ds.perm is None:
        raise ValueError("a spatially sorted scene needs its inverse permutation (DeviceScene.perm)")
    s.height, s.width = ds.height, ds.width
    s.n_frames = ds.n_frames if n_frames is None else n_frames
    s.n_mviews = ds.view_mask_offs.shape[0] - 1
    s.word_bits, s.n_rows = ds.word_bits, ds.n_rows
    s.conf_f16 = 1 if ds.conf.dtype == torch.float16 else 0
    s.n_label_ids = ds.n_label_ids
    if stage1 is not None:
        s.s1_run_start, s.s1_run_end, s.s1_row_run_offs = _p(stage1.run_start), _p(stage1.run_end), _p(stage1.row_run_offs)
        s.s1_rows = stage1.row_run_offs.shape[0] - 1
    return s


def params_struct(cfg, depth_thresh, filter_sort=False):
    _check_layout()
    p = ParamsStruct()
    p.filter_sort = 1 if filter_sort else 0
    ratio = (not cfg.if_occurance_threshold) and bool(cfg.if_detected_ratio_threshold)
    p.depth_thresh = float(depth_thresh)
    p.filter_mode = 1 if cfg.if_occurance_threshold else (2 if ratio else 0)
    p.filter_fraction = float(cfg.detected_ratio_threshold if ratio else cfg.occurance_threshold)
    p.iou_thres = float(cfg.iou_thres)
    p.min_members = int(cfg.min_aggragated_masks)
    return p


class SceneWorkspace:
    """Scratch of bff_scene_project for one stream, grown on demand and reused scene after scene.  `rows` is the
    zero arena of the instance rows: all zero whenever no call is in flight (the call clears what its sweep
    stored; a call whose results were never collected leaves it marked dirty and it is zeroed again)."""
    _per_stream = {}

    def __init__(self, device):
        self.device = torch.device(device)
        self.t = {}                      # name -> tensor
        self.cap = {}                    # name -> elements
        self.struct = WorkspaceStruct()
        self.hdr_host = None
        self.in_flight = False
        self.rows_dirty = False
        self.both_primed = 0
        self._fit_key = None

    @classmethod
    def for_current_stream(cls, device):
        dev = torch.device(device)
        key = (dev.index if dev.index is not None else torch.cuda.current_device(), _lib.raw_stream())
        ws = cls._per_stream.get(key)
        if ws is None:
            st = torch.cuda.current_stream(device)
            ws = cls._per_stream[key] = cls(st.device)
            ws.stream = st                    # the stream this workspace belongs to (the key above is its raw handle)
            ws.heavy = heavy_stream(st.device, len(cls._per_stream) - 1)      # workspaces take the heavy streams in turn
            if ws.heavy is not None:
                lib = _lib.load()
                ws.struct.heavy_stream = c_void_p(ws.heavy.cuda_stream)
                evs = [lib.bff_event_create() for _ in range(4)]
                if not all(evs):
                    raise _lib.BffLibraryError("bff_event_create failed")
                ws.struct.events = (c_void_p * 4)(*evs)
            if AUX_STREAM:
                # the point filter's threshold chain runs beside the components chain (bff_scene_workspace.aux_stream)
                lib = _lib.load()
                ws.aux = torch.cuda.Stream(device=st.device)
                ws.struct.aux_stream = c_void_p(ws.aux.cuda_stream)
                evs = [lib.bff_event_create() for _ in range(2)]
                if not all(evs):
                    raise _lib.BffLibraryError("bff_event_create failed")
                ws.struct.aux_events = (c_void_p * 2)(*evs)
        return ws

    def _need(self, name, numel, dtype, zero=False):
        t = self.t.get(name)
        if t is None or t.numel() < numel or t.dtype != dtype:
            t = (torch.zeros if zero else torch.empty)(max(int(numel), 1), dtype=dtype, device=self.device)
            self.t[name] = t
            setattr(self.struct, name, c_void_p(t.data_ptr()))
            return True
        return False

    def fit(self, ds, s1_rows, cap=GROUP_CAP):
        """Make every buffer large enough for scene `ds` (+ s1_rows stage-1 masks) with `cap` group slots."""
        n, nw, n_rows = ds.n_points, ds.nw, ds.n_rows
        hw = ds.height * ds.width
        n_mviews = ds.view_mask_offs.shape[0] - 1
        key = (n, nw, n_rows, hw, n_mviews, ds.word_bits, s1_rows, cap)
        if key == self._fit_key:                      # same sizes as the last scene on this stream: nothing to check
            return self
        self._fit_key = key
        self.struct.group_cap = cap
        lib = _lib.load()
        mw = max(lib.bff_chunk_mask_words(nw), 1)
        nt = (n_rows + 63) // 64
        i32, i64, f32 = torch.int32, torch.int64, torch.float32
        self._need("maskbits", n_mviews * hw * (1 if ds.word_bits == 32 else 2), i32)
        self._need("labels", n_mviews * int(lib.bff_label_plane_stride(hw)), torch.uint8)
        if self._need("rows", n_rows * nw, i64, zero=True):
            self.rows_dirty = False
        # everything the call's steps expect zeroed lives in ONE allocation, cleared by one fill per scene
        # (bff_scene_workspace): name -> (bytes, dtype of the view)
        seg_words = 2 * n_mviews * _lib.segmap_words(hw)
        hdr_words = int(lib.bff_scene_header_words(s1_rows, cap))
        use_cpop = bool(lib.bff_merge_uses_chunk_bound(nw))
        parts = [("masked", 4 * n, i32), ("viewed", 4 * n, i32), ("count", 4 * n_rows, i32),
                 ("chunk_mask", 8 * n_rows * mw, i64), ("segmap", 4 * seg_words, i32), ("hdr", 4 * hdr_words, i32),
                 ("agg", 8 * cap * nw, i64), ("merge_scratch", 4 * int(lib.bff_merge_scratch_words(n_rows)), i32)]
        if use_cpop:
            parts.append(("chunk_pop", 2 * n_rows * mw * 64, torch.int16))
        offs, at = {}, 0
        for name, nbytes, _dt in parts:
            offs[name] = at
            at += (nbytes + 63) // 64 * 64
        zbytes = max(at, 64)
        if self._need("zero_block", zbytes, torch.uint8):
            pass
        zb = self.t["zero_block"]
        for name, nbytes, dt in parts:
            o = offs[name]
            setattr(self.struct, name, c_void_p(zb.data_ptr() + o))
            self.t[name] = zb[o:o + nbytes].view(dt)
        self.struct.zero_bytes = zbytes
        if not use_cpop:
            self._need("chunk_pop", 1, torch.int16)
        self._need("keep", nw, i64)
        self._need("tile_mask", nt * mw, i64)
        self._need("sel_scratch", (n + 1023) // 1024, i32)
        self._need("pair_scratch", int(lib.bff_point_threshold_scratch_words(n)), i32)
        for k in ("area", "mean_word", "order", "parent", "comp", "gmembers"):
            self._need(k, n_rows, i32)
        self._need("goffs", cap + 1, i32)
        self._need("slices", 3 * lib.bff_group_slice_cap(n_rows, cap), i32)
        self._need("vals", n, f32)
        self._need("vals_sorted", n, f32)
        self._need("hist", n_rows * 64, i32)
        for k in ("sig", "sig_keys", "sig_sorted"):
            self._need(k, n_rows, i64)
        # both library sorts share one temp buffer
        need = ctypes.c_size_t(0)
        _lib.call("bff_sort_f32", None, None, n, None, ctypes.byref(need))
        nb = int(need.value)
        _lib.call("bff_argsort_i64", None, None, None, n_rows, 62, None, ctypes.byref(need))
        nb = max(nb, int(need.value))
        self._need("sort_temp", nb, torch.uint8)
        self.struct.sort_temp_bytes = self.t["sort_temp"].numel()
        if self.hdr_host is None or self.hdr_host.numel() < hdr_words:
            self.hdr_host = torch.empty(hdr_words, dtype=i32, pin_memory=True)
            self.struct.hdr_host = c_void_p(self.hdr_host.data_ptr())
        return self

    def view(self, name, *shape):
        n = int(np.prod(shape))
        return self.t[name][:n].view(*shape)


def issue(ds, cfg, depth_thresh, stage1=None, n_frames=None):
    """Enqueue the whole device side of one scene on the current stream.  Returns a handle for `collect`."""
    t0 = time.perf_counter() if _TRACE_ISSUE else 0.0
    dev = ds.xyz.device
    key = (id(stage1), n_frames)
    cache = ds.__dict__.setdefault("_scene_structs", {})
    ent = cache.get(key)
    if ent is None:
        ent = cache[key] = (scene_struct(ds, stage1, n_frames), stage1)      # keeps stage1's tensors alive too
    sc = ent[0]
    s1_rows = int(sc.s1_rows)
    cap = int(ds.__dict__.get("_group_cap", GROUP_CAP))           # GROUP_CAP_MAX once the scene was seen to keep more groups
    ws = SceneWorkspace.for_current_stream(dev).fit(ds, s1_rows, cap)
    if ws.in_flight or ws.rows_dirty:             # a call whose results were never collected: the arena may be dirty
        ws.t["rows"].zero_()
        ws.rows_dirty = False
    n_both = (cap + s1_rows) * ds.nw
    if ws.both_primed < n_both:
        # `both` outlives the workspace's reuse (results are views of it), so it comes from torch's allocator, whose
        # pools are per stream: the first few scenes of a stream would each pay a hipMalloc (~6 ms) until the pool
        # holds the 3-4 blocks that are alive at a time.  Take and return them once, when the workspace is sized.
        prime = [torch.empty(n_both, dtype=torch.int64, device=dev) for _ in range(4)]
        del prime
        ws.both_primed = n_both
    both = torch.empty((cap + s1_rows, ds.nw), dtype=torch.int64, device=dev)     # outlives the workspace's reuse
    ws.struct.both = c_void_p(both.data_ptr())
    # the threshold of the point filter: from the set of distinct values (two launches; default) or by a radix sort of all
    # values (12 launches; BFF_FILTER_SORT=1, and automatically for a scene with more distinct values than the set holds)
    use_sort = os.environ.get("BFF_FILTER_SORT") == "1" or bool(ds.__dict__.get("_filter_sort", False))
    pr = params_struct(cfg, depth_thresh, filter_sort=use_sort)
    ws.in_flight = True
    t1 = time.perf_counter() if _TRACE_ISSUE else 0.0
    _lib.call("bff_scene_project", ctypes.byref(sc), ctypes.byref(pr), ctypes.byref(ws.struct))
    if _TRACE_ISSUE and time.perf_counter() - t0 > 2e-3:      # the one-time stalls of a process's first scene calls
        import sys
        now = time.perf_counter()
        print(f"slow issue: {1e3 * (now - t0):.2f} ms, of which the native call {1e3 * (now - t1):.2f} ms", file=sys.stderr)
    return dict(ws=ws, both=both, s1_rows=s1_rows, params=pr, stream=ws.stream, cap=cap,
                args=(ds, cfg, depth_thresh, stage1, n_frames))


def collect(h):
    """Wait for the header of `issue` and return it as an int32 array (a copy: the pinned buffer is reused)."""
    t0 = time.perf_counter()
    h["stream"].synchronize()
    _lib.sync_wait_s += time.perf_counter() - t0
    ws = h["ws"]
    cap = h["cap"]
    words = hdr_offsets(cap)[5] + h["s1_rows"] * (cap + h["s1_rows"])
    hdr = ws.hdr_host.numpy()[:words].copy()
    ws.in_flight = False
    if (hdr[HDR_K + 1] & 1) and cap < GROUP_CAP_MAX and hdr[HDR_K] <= GROUP_CAP_MAX and hdr[HDR_OVERFLOW] == 0 \
            and os.environ.get("BFF_GROUP_CAP_FIXED") != "1":
        # more kept groups than the default tables hold, but within the large ones: run the scene again with those (and
        # remember it for this scene) -- the general host path costs more than a second call
        ds = h["args"][0]
        ds.__dict__["_group_cap"] = GROUP_CAP_MAX
        ws.rows_dirty = True                      # flags != 0: the call left the raw rows for the host
        with torch.cuda.stream(h["stream"]):
            h2 = issue(*h["args"])
        h.update(h2)
        return collect(h)
    if hdr[HDR_OVERFLOW] != 0 and not h["params"].filter_sort:
        # more distinct filter values than the pair formulation holds: everything after the sweep is void.  Run the
        # scene again with the sorting formulation (and remember it for this scene).
        ds = h["args"][0]
        ds.__dict__["_filter_sort"] = True
        ws.rows_dirty = True
        with torch.cuda.stream(h["stream"]):
            h2 = issue(*h["args"])
        h.update(h2)
        return collect(h)
    if hdr[HDR_K + 1] != 0:
        ws.rows_dirty = True                      # the general path reads the rows and clears them itself
    return hdr


def pipelined(n, front, back, depth=None):
    """The software pipeline every caller uses: `front(i)` issues the device work of item i without waiting for
    anything, `back(i, handle)` finishes it on the host; `depth` items are in flight, i.e. while the host works on
    the back half of item i the device already runs items i+1 .. i+depth-1.  Generator of back()'s results."""
    depth = PIPELINE_DEPTH if depth is None else max(1, int(depth))
    inflight, issued = [], 0
    for i in range(n):
        while issued < n and issued - i < depth:
            inflight.append(front(issued))
            issued += 1
        yield back(i, inflight.pop(0))


def project_stream(scenes, cfg, device, consume, n_loaders=2, with_stage1=True, depth=None, want_groups=False):
    """Projection stage (P:365-634) of a sequence of scenes on one GPU, overlapped end to end: loader threads read
    / ingest scenes ahead on their own streams (ingest.Ingestor: native run tables, pinned uploads, cloud laid out on
    the device), PIPELINE_DEPTH scenes are in flight on the compute streams, the host half of scene i runs under the
    kernels of the next ones.  `scenes`: SceneInputs-like objects or zero-argument callables that load one (called on
    a loader thread).  `consume(k, DeviceStage1 | None, Stage2Result)` is called in order, ON THE STREAM the scene was
    projected on (whatever it enqueues is ordered after the scene's kernels without any synchronisation); a load error
    surfaces at that scene's turn."""
    from .ingest import Ingestor
    from .projection import projection_back, projection_front
    dev = torch.device(device)
    depth = PIPELINE_DEPTH if depth is None else depth
    n = len(scenes)
    if n == 0:
        return
    with_viewed = (not cfg.if_occurance_threshold) and bool(cfg.if_detected_ratio_threshold)
    if dev.type != "cuda":                      # host tensors: no streams, no loaders (the kernels themselves need the GPU)
        from .refinement import prepare_stage1
        from .scene import prepare_scene
        for k, sc in enumerate(scenes):
            sc = sc() if callable(sc) else sc
            ds = prepare_scene(sc, cfg, device=device, with_viewed=with_viewed)
            st1 = prepare_stage1(sc.stage1, device) if (with_stage1 and getattr(sc, "stage1", None) is not None) else None
            consume(k, st1, projection_back(projection_front(ds, cfg, stage1=st1), want_groups=want_groups))
        return
    streams = scene_streams(dev, depth)
    ing = Ingestor(cfg, dev, n_loaders=n_loaders, with_viewed=with_viewed, with_stage1=with_stage1)
    lookahead = depth + n_loaders
    futs = {i: ing.submit(scenes[i]) for i in range(min(lookahead, n))}

    def front(i):
        ds, st1, ev = futs.pop(i).result()
        if i + lookahead < n:
            futs[i + lookahead] = ing.submit(scenes[i + lookahead])
        st = streams[i % depth]
        st.wait_event(ev)                       # the uploads ran on the loader's stream
        with _lib.on_stream(st):
            return projection_front(ds, cfg, stage1=st1), st1

    def back(i, h):
        fr, st1 = h
        with _lib.on_stream(streams[i % depth]):
            consume(i, st1, projection_back(fr, want_groups=want_groups))

    try:
        for _ in pipelined(n, front, back, depth):
            pass
    finally:
        ing.close()
