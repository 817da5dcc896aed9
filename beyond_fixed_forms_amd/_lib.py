"""ctypes binding of libbff_hip.so (C ABI: include/bff_hip.h).

There is deliberately no fallback: if the library is missing or a call is rejected this raises.
Tensors are passed as raw device pointers; the launch stream is torch's current stream, so torch
events and torch ops order correctly around the kernels.
"""
from __future__ import annotations

import ctypes
import os
import time
from ctypes import c_double, c_float, c_int32, c_int64, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BFF_HIP_LIB") or os.path.join(_HERE, "lib", "libbff_hip.so")   # BFF_HIP_LIB: A/B builds
CSRC_DIR = os.path.join(_HERE, "csrc")

_P, _I, _L, _D, _F = c_void_p, c_int32, c_int64, c_double, c_float

# name -> argument ctypes (every entry point returns int); mirrors include/bff_hip.h one to one
SIGNATURES = {
    "bff_rle_to_maskbits": [_P, _P, _P, _P, _I, _L, _I, _P, _P, _P],
    "bff_rle_to_labels": [_P, _P, _P, _P, _I, _L, _I, _P, _P, _P, _P],
    "bff_project_views": [_P, _L, _L, _P, _P, _I, _P, _P, _I, _I, _D, _P, _P, _P, _I, _P, _P, _P, _P, _P, _L, _L, _P, _P, _P, _P, _P],
    "bff_depth_tile_u16": [_P, _I, _I, _I, _P, _I, _P],
    "bff_project_views_u16": [_P, _L, _L, _P, _P, _I, _P, _I, _I, _I, _P, _I, _I, _D, _P, _P, _P, _I, _P, _P, _P, _P, _P, _L, _L, _P, _P, _P, _P, _P],
    "bff_point_tile_bounds": [_P, _L, _L, _P, _P],
    "bff_popcount_rows": [_P, _P, _I, _L, _P, _P],
    "bff_cross_popcount": [_P, _P, _I, _P, _P, _I, _L, _P, _P],
    "bff_row_stats": [_P, _I, _L, _P, _P, _P, _I, _P, _P, _P, _P],
    "bff_clear_flagged_chunks": [_P, _I, _L, _P, _P],
    "bff_merge_components": [_P, _I, _L, _P, _I, _P, _P, _P, _P, _P, _P, _F, _P, _I, _P, _P, _P, _P],
    "bff_merge_adjacency": [_P, _I, _L, _P, _P, _P, _P, _P, _P, _F, _P, _P, _P],
    "bff_permute_bits": [_P, _I, _L, _P, _L, _L, _P, _P],
    "bff_components_round": [_P, _I, _P, _P, _P, _P],
    "bff_or_reduce_groups": [_P, _L, _P, _P, _I, _I, _P, _P, _I, _P, _P, _P],
    "bff_resolve_overlaps": [_P, _I, _L, _P, _P, _P, _P, _P],
    "bff_group_conf_mean": [_P, _I, _P, _P, _I, _P, _P],
    "bff_apply_row_ops": [_P, _L, _P, _I, _P],
    "bff_overlap_ops": [_P, _P, _I, _P, _P],
    "bff_and_rows": [_P, _I, _L, _P, _P],
    "bff_gather_rows": [_P, _P, _I, _L, _P, _P],
    "bff_unpack_rows": [_P, _I, _L, _L, _P, _P],
    "bff_pack_rows": [_P, _I, _L, _L, _P, _P],
    "bff_rle_to_rows": [_P, _P, _P, _I, _L, _L, _P, _P],
    "bff_ids_to_rows": [_P, _L, _P, _I, _L, _P, _P],
    "bff_rle_count_runs": [_P, _I, _L, _P, _P],
    "bff_rle_encode_rows": [_P, _I, _L, _P, _L, _P, _P],
    "bff_ratio_keep": [_P, _P, _L, _F, _P, _I, _L, _P, _P],
    "bff_point_values": [_P, _P, _L, _P, _P],
    "bff_point_threshold_pairs": [_P, _P, _L, _D, _P, _P, _P, _P, _P],
    "bff_select_unique_rank": [_P, _L, _D, _P, _P, _P, _P],
    "bff_cosine_gemm_f16": [_P, _I, _P, _I, _I, _P, _P],
    "bff_cosine_rows": [_P, _I, _P, _I, _I, _I, _P, _P],
    "bff_normalized_gemm_f16": [_P, _I, _P, _I, _I, _P, _P],
    "bff_description_means": [_P, _P, _I, _I, _I, _P, _P],
    "bff_group_components": [_P, _P, _P, _I, _F, _I, _I, _P, _I, _P, _P, _P, _P, _P, _P, _P],
    "bff_or_reduce_grouped": [_P, _L, _I, _P, _I, _P, _P, _P, _P, _P, _I, _P, _P, _P],
    "bff_resolve_overlaps_dev": [_P, _I, _L, _P, _P, _P, _P, _P, _P],
    "bff_clear_flagged_chunks_unless": [_P, _I, _L, _P, _P, _P],
    "bff_scene_project": [_P, _P, _P, _P],
    "bff_diag_gather": [_P, _L, _L, _P, _P],
    "bff_diag_sweep_lines": [_P, _L, _L, _P, _P, _I, _P, _P, _I, _I, _D, _P, _I, _P, _P, _P, _L, _P, _P],
    "bff_diag_sweep_lines_u16": [_P, _L, _L, _P, _P, _I, _P, _I, _I, _I, _P, _I, _I, _D, _P, _I, _P, _P, _P, _L, _P, _P],
    "bff_scatter_bits": [_P, _I, _L, _P, _L, _L, _P, _P, _P],
    "bff_cross_popcount_dev": [_P, _I, _P, _I, _L, _P, _P, _I, _I, _P],
    "bff_cloud_layout": [_P, _L, _L, _L, _I, _P, _P, _P, _P, _P, _P, _P, _P],
    "bff_sort_f32": [_P, _P, _L, _P, _P, _P],
    "bff_argsort_i64": [_P, _P, _P, _I, _I, _P, _P, _P],
    "bff_depth_from_u16": [_P, _I, _I, _I, _P, _P, _P, _P, _P, _P, _I, _I, _F, _P, _P],
}
PLAIN = {"bff_abi_version": (c_int32, []), "bff_last_error": (ctypes.c_char_p, []), "bff_arch": (ctypes.c_char_p, []),
         "bff_chunk_mask_words": (c_int32, [c_int64]), "bff_label_plane_stride": (c_int64, [c_int64]), "bff_resolve_overlaps_max_rows": (c_int32, []),
         "bff_point_tile_size": (c_int32, []), "bff_depth_tiled_texels": (c_int64, [c_int32, c_int32]), "bff_merge_scratch_words": (c_int64, [c_int32]), "bff_merge_uses_chunk_bound": (c_int32, [c_int64]),
         "bff_profile_next_merge": (c_int32, [_P, _P]), "bff_group_slice_cap": (c_int32, [c_int32, c_int32]),
         "bff_point_threshold_scratch_words": (c_int64, [c_int64]), "bff_point_threshold_capacity": (c_int32, []), "bff_point_threshold_capacity_set": (c_int32, [c_int32]), "bff_scene_header_words": (c_int32, [c_int32, c_int32]), "bff_scene_struct_bytes": (c_int32, [c_int32]),
         "bff_host_component_csr": (c_int32, [_P, _P, _I, _I, _P, _P, _P, _P]),
         "bff_profile_next_sweep": (c_int32, [_P, _P]), "bff_event_create": (c_void_p, []),
         "bff_event_destroy": (c_int32, [_P]), "bff_event_elapsed_ms": (c_int32, [_P, _P, _P]),
         "bff_event_record": (c_int32, [_P, _P]), "bff_event_synchronize": (c_int32, [_P])}
ABI_VERSION = 8


class BffLibraryError(RuntimeError):
    pass


_lib = None


def load():
    """Load libbff_hip.so once; raise BffLibraryError (never fall back) if that is impossible."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise BffLibraryError(
            f"{LIB_PATH} not found: build it with `make -C {CSRC_DIR}` (or __graft_entry__.build()). "
            "This package has no CPU fallback.")
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:
        raise BffLibraryError(f"cannot load {LIB_PATH}: {e}") from e
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes, fn.restype = args, c_int32
    for name, (res, args) in PLAIN.items():
        fn = getattr(lib, name)
        fn.argtypes, fn.restype = args, res
    if lib.bff_abi_version() != ABI_VERSION:
        raise BffLibraryError(f"{LIB_PATH}: ABI {lib.bff_abi_version()} != expected {ABI_VERSION}; rebuild")
    _lib = lib
    return lib


import threading

_tls = threading.local()  # .cache = (c_void_p, torch stream) pinned for the duration of a `with launch_stream():` block
                          # (per thread: loader threads of the ingestion pipeline launch on their own streams)


def _cached_stream():
    return getattr(_tls, "cache", None)


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)      # the handle without building a Stream object
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


def raw_stream():
    """hipStream_t of torch's current stream on the current device as an int (torch.cuda.current_stream() costs ~8 us
    of Python per call -- more than a kernel launch; a scene makes a dozen of them)."""
    if _raw_stream is not None and _raw_device is not None:
        return _raw_stream(_raw_device())
    return torch.cuda.current_stream().cuda_stream


def _stream():
    c = _cached_stream()
    if c is not None:
        return c[0]
    return c_void_p(raw_stream())


_get_cur = getattr(torch._C, "_cuda_getCurrentStream", None)
_set_cur = getattr(torch._C, "_cuda_setStream", None)


class on_stream:
    """`with torch.cuda.stream(st)` without its Python: the current stream's ids are read and set through the two C calls
    torch.cuda.current_stream / set_stream wrap (a torch.cuda.stream block costs ~20 us of Python, and a scene enters two)."""
    __slots__ = ("st", "prev", "ctx")

    def __init__(self, st):
        self.st, self.prev, self.ctx = st, None, None

    def __enter__(self):
        if _get_cur is None or _set_cur is None or self.st is None:
            self.ctx = torch.cuda.stream(self.st)
            return self.ctx.__enter__()
        st = self.st
        if _raw_device is not None and _raw_device() != st.device_index:      # another device: the full context
            self.ctx = torch.cuda.stream(st)
            return self.ctx.__enter__()
        self.prev = _get_cur(st.device_index)                 # (stream_id, device_index, device_type)
        _set_cur(stream_id=st.stream_id, device_index=st.device_index, device_type=st.device_type)
        return st

    def __exit__(self, *exc):
        if self.ctx is not None:
            return self.ctx.__exit__(*exc)
        p = self.prev
        _set_cur(stream_id=p[0], device_index=p[1], device_type=p[2])
        return False


class _PinnedStream:
    """(raw handle, torch stream) of the current stream, the handle taken the cheap way at once, the torch.cuda.Stream
    object (8 us of Python to build) only when somebody asks for it -- most blocks only launch."""
    __slots__ = ("raw", "_obj")

    def __init__(self):
        self.raw = c_void_p(raw_stream())
        self._obj = None

    def __getitem__(self, k):
        if k == 0:
            return self.raw
        if self._obj is None:
            self._obj = torch.cuda.current_stream()
        return self._obj


class launch_stream:
    """Look torch's current stream up once for a whole sequence of kernel launches (the lookup costs more
    than a launch).  Entry points such as run_projection / refine_class wrap their bodies in it; the stream
    must not be switched inside the block."""

    def __enter__(self):
        self._outer = _cached_stream()
        _tls.cache = _PinnedStream()
        return self

    def __exit__(self, *exc):
        _tls.cache = self._outer
        return False


def _ptr(t, dtype=None):
    if t is None:
        return None
    if not t.is_cuda:
        raise ValueError("bff kernels take device tensors (no CPU path)")
    if not t.is_contiguous():
        raise ValueError("bff kernels take contiguous tensors")
    if dtype is not None and t.dtype != dtype:
        raise TypeError(f"expected {dtype}, got {t.dtype}")
    return c_void_p(t.data_ptr())


def call(name, *args):
    lib = load()
    rc = getattr(lib, name)(*args, _stream())
    if rc != 0:
        raise RuntimeError(f"{name} failed ({rc}): {lib.bff_last_error().decode()}")


i32, i64, f32, f64, u8 = torch.int32, torch.int64, torch.float32, torch.float64, torch.uint8


# ------------------------------------------------------------------ typed wrappers
def segmap_words(n_pixels):
    return ((n_pixels + 127) // 128 + 31) // 32


def rle_to_maskbits(run_start, run_end, mask_run_offs, view_mask_offs, n_views, n_pixels, word_bits, maskbits,
                    segmap=None):
    call("bff_rle_to_maskbits", _ptr(run_start, i32), _ptr(run_end, i32), _ptr(mask_run_offs, i32),
         _ptr(view_mask_offs, i32), n_views, n_pixels, word_bits,
         _ptr(maskbits, torch.int32 if word_bits == 32 else torch.int64), _ptr(segmap, i32))


def label_plane_stride(n_pixels):
    return (n_pixels + 127) // 128 * 128


def rle_to_labels(run_start, run_end, mask_run_offs, view_mask_offs, n_views, n_pixels, word_bits, labels, words, segmap):
    """Segment-wise palette blocks (uint8 [n_views][label_plane_stride(n_pixels)]: 64 B of 4-bit piece numbers + the
    pieces' words per 128-pixel segment) or mask words; segmap: int32 [n_views][2 * segmap_words(n_pixels)] (occupied |
    word form)."""
    call("bff_rle_to_labels", _ptr(run_start, i32), _ptr(run_end, i32), _ptr(mask_run_offs, i32),
         _ptr(view_mask_offs, i32), n_views, n_pixels, word_bits, _ptr(labels, u8),
         _ptr(words, torch.int32 if word_bits == 32 else torch.int64), _ptr(segmap, i32))


def project_views(xyz_soa, n_points, inv_pose, cam_intr, depth, depth_index, height, width, depth_thresh,
                  maskbits, word_bits, frame_mask, frame_rowbase, frame_nmask, frame_flags,
                  rows, masked_count, viewed_count, segmap=None, chunk_mask=None, tile_bounds=None, labels=None,
                  depth_size=None):
    """depth: float32 [n_depth][H*W] metres, or int16 [n_depth][hs][ws] (the uint16 millimetres of the PNGs): then
    /1000 + the bilinear resize to (height, width) are evaluated per point inside the sweep (bff_project_views_u16).
    depth_size = (hs, ws): `depth` is [n_depth][depth_tiled_texels(hs, ws)], frames in 8 x 8 tiles (tile_depth): int16
    (the millimetres) or float32 (metres, already divided by 1000)."""
    k = (c_double * 9)(*[float(v) for v in cam_intr.reshape(-1)])
    n_frames = inv_pose.shape[0]
    nw = (n_points + 63) // 64
    if depth.dtype == torch.int16 or depth_size is not None:
        hs, ws = (depth_size if depth_size is not None else depth.shape[1:3])
        layout = 0 if depth_size is None else (2 if depth.dtype == f32 else 1)
        head = ("bff_project_views_u16", _ptr(xyz_soa, f64), n_points, xyz_soa.shape[1], _ptr(inv_pose, f64),
                ctypes.cast(k, c_void_p), n_frames, _ptr(depth), int(hs), int(ws), layout)
    else:
        head = ("bff_project_views", _ptr(xyz_soa, f64), n_points, xyz_soa.shape[1], _ptr(inv_pose, f64),
                ctypes.cast(k, c_void_p), n_frames, _ptr(depth, f32))
    call(*head, _ptr(depth_index, i32), height, width,
         float(depth_thresh), _ptr(maskbits), _ptr(labels, u8), _ptr(segmap, i32), word_bits, _ptr(frame_mask, i32), _ptr(frame_rowbase, i32),
         _ptr(frame_nmask, i32), _ptr(frame_flags, i32), _ptr(rows, i64),
         0 if rows is None else rows.shape[0], nw, _ptr(chunk_mask, i64), _ptr(masked_count, i32),
         _ptr(viewed_count, i32), _ptr(tile_bounds, f64))


def tile_depth(raw, metres=True):
    """int16 [F][hs][ws] (uint16 millimetres as stored) -> [F][tiled texels]: every frame in 8 x 8-texel tiles
    (bff_depth_tile_u16), the layout the sweep gathers its taps from; metres: float32 `value / 1000` (P:432-435) instead
    of the uint16 values, so that the sweep neither converts nor divides."""
    f, hs, ws = raw.shape
    out = torch.empty((f, int(load().bff_depth_tiled_texels(hs, ws))), dtype=f32 if metres else torch.int16, device=raw.device)
    call("bff_depth_tile_u16", _ptr(raw, torch.int16), f, hs, ws, _ptr(out), 1 if metres else 0)
    return out


def point_tile_bounds(xyz_soa, n_points):
    """float64 [tiles][6] bounding boxes of the sweep's point tiles (frustum culling table of project_views)."""
    tile = load().bff_point_tile_size()
    out = torch.empty((max(1, (n_points + tile - 1) // tile), 6), dtype=f64, device=xyz_soa.device)
    call("bff_point_tile_bounds", _ptr(xyz_soa, f64), n_points, xyz_soa.shape[1], _ptr(out))
    return out


def popcount_rows(rows, idx=None):
    n = rows.shape[0] if idx is None else idx.shape[0]
    area = torch.empty(n, dtype=i32, device=rows.device)
    call("bff_popcount_rows", _ptr(rows, i64), _ptr(idx, i32), n, rows.shape[1], _ptr(area))
    return area


def cross_popcount(a, b, ia=None, ib=None):
    na = a.shape[0] if ia is None else ia.shape[0]
    nb = b.shape[0] if ib is None else ib.shape[0]
    out = torch.empty((na, nb), dtype=i32, device=a.device)
    call("bff_cross_popcount", _ptr(a, i64), _ptr(ia, i32), na, _ptr(b, i64), _ptr(ib, i32), nb, a.shape[1], _ptr(out))
    return out


def chunk_mask_buffer(n_rows, nw, device):
    """Uninitialised chunk occupancy masks i64 [n_rows][mw] (zero them before handing them to project_views)."""
    return torch.empty((n_rows, max(load().bff_chunk_mask_words(nw), 1)), dtype=i64, device=device)


class RowArena:
    """Zero-filled instance rows without a zero-fill per scene.  One flat int64 buffer per stream that is all zero
    whenever it is free: `take` hands out a [n_rows][nw] view of it, `release` zeroes exactly the chunks the sweep
    flagged (bff_clear_flagged_chunks, ~1 % of the buffer) on the same stream.  A buffer that was taken and never
    released (an exception, a caller that only runs the front half) is simply dropped: the next `take` starts from
    fresh zeros, so a dirty buffer can never be handed out."""
    _arenas = {}

    def __init__(self):
        self.buf = None
        self.busy = False

    @classmethod
    def for_current_stream(cls, device):
        st = _cached_stream()[1] if _cached_stream() is not None else torch.cuda.current_stream(device)
        key = (st.device.index, st.cuda_stream)
        a = cls._arenas.get(key)
        if a is None:
            a = cls._arenas[key] = cls()
        return a

    def take(self, n_rows, nw, device):
        need = n_rows * nw
        if self.busy or self.buf is None or self.buf.numel() < need:
            self.buf = torch.zeros(max(need, 1), dtype=i64, device=device)      # fresh zeros (first use, growth, or dirty)
        self.busy = True
        return self.buf[:need].view(n_rows, nw)

    def release(self, rows, cmask):
        """rows: the view handed out by take(); cmask: the chunk flags the sweep wrote for it."""
        if not self.busy or self.buf is None or rows.data_ptr() != self.buf.data_ptr():
            return                                                  # not ours (any more): nothing to recycle
        if rows.numel():
            call("bff_clear_flagged_chunks", _ptr(rows, i64), rows.shape[0], rows.shape[1], _ptr(cmask, i64))
        self.busy = False


last_chunk_pop = None     # per-chunk point counts of the latest row_stats call (merge_components' second-level bound)


def row_stats(rows, cmask=None):
    """-> (area i32 [R], mean_word i32 [R], chunk_mask i64 [R][mw], hist i32 [R][64], signature i64 [R]).
    cmask given (from project_views): only the flagged chunks are read.  The per-chunk point counts of the rows
    (uint16 [R][64 * mw], the second-level bound of merge_components) are kept in `hist.chunk_pop`."""
    global last_chunk_pop
    n = rows.shape[0]
    given = cmask is not None
    area = torch.empty(n, dtype=i32, device=rows.device)
    mean_word = torch.empty(n, dtype=i32, device=rows.device)
    if not given:
        cmask = chunk_mask_buffer(n, rows.shape[1], rows.device)
    hist = torch.empty((n, 64), dtype=i32, device=rows.device)
    sig = torch.empty(n, dtype=i64, device=rows.device)
    cpop = torch.empty((n, cmask.shape[1] * 64), dtype=torch.int16, device=rows.device)
    call("bff_row_stats", _ptr(rows, i64), n, rows.shape[1], _ptr(area), _ptr(mean_word), _ptr(cmask, i64),
         1 if given else 0, _ptr(hist), _ptr(sig), _ptr(cpop))
    hist.chunk_pop = cpop                        # travels with the histogram it refines
    return area, mean_word, cmask, hist, sig


def merge_components(rows, area, label_id, iou_thres, order, chunk_mask, hist, diag=None, parent=None,
                     coarse_stride=0, use_chunk_bound=True):
    """comp[i] = smallest row index of the component of row i in the merge graph (no adjacency matrix).
    coarse_stride > 1 (optional, off by default: measured slower at config 2): the tile pass first runs on
    every coarse_stride-th row of `order`, then the full pass starts from that forest."""
    n = rows.shape[0]
    nt = (n + 63) // 64
    tmask = torch.empty((nt, chunk_mask.shape[1]), dtype=i64, device=rows.device)
    init = parent is None
    if init:
        parent = torch.empty(n, dtype=i32, device=rows.device)
    comp = torch.empty(n, dtype=i32, device=rows.device)
    hist_sorted = torch.empty(int(load().bff_merge_scratch_words(n)), dtype=i32, device=rows.device)  # scratch, see bff_hip.h

    cpop = getattr(hist, "chunk_pop", None) if use_chunk_bound else None

    def run(ordr, init_parent, out):
        call("bff_merge_components", _ptr(rows, i64), n, rows.shape[1], _ptr(ordr, i32), ordr.shape[0],
             _ptr(chunk_mask, i64), _ptr(tmask), _ptr(hist, i32), _ptr(hist_sorted), _ptr(area, i32),
             _ptr(label_id, i32), float(iou_thres), _ptr(parent), int(init_parent), _ptr(out), _ptr(diag, i32),
             _ptr(cpop, torch.int16))

    if init and coarse_stride > 1 and n >= 64 * coarse_stride:
        run(order[::coarse_stride].contiguous(), True, None)
        run(order, False, comp)
    else:
        run(order, init, comp)
    return comp


def merge_adjacency(rows, area, label_id, iou_thres, order=None, chunk_mask=None, hist=None, want_inter=False):
    """Adjacency bit matrix indexed by position in `order` (identity when None).  Cross-check path."""
    n = rows.shape[0]
    aw = (n + 63) // 64
    adj = torch.empty((n, aw), dtype=i64, device=rows.device)
    inter = torch.empty((n, n), dtype=i32, device=rows.device) if want_inter else None
    tmask = None
    if chunk_mask is not None:
        tmask = torch.empty((aw, chunk_mask.shape[1]), dtype=i64, device=rows.device)
    call("bff_merge_adjacency", _ptr(rows, i64), n, rows.shape[1], _ptr(order, i32), _ptr(chunk_mask, i64),
         _ptr(tmask), _ptr(hist, i32), _ptr(area, i32), _ptr(label_id, i32), float(iou_thres), _ptr(adj), _ptr(inter))
    return (adj, inter) if want_inter else adj


def permute_bits(rows, idx, n_out, out=None):
    """out[r] bit o = rows[r] bit idx[o].  out: optional destination (contiguous int64 [rows][ceil(n_out/64)])."""
    nw_out = (n_out + 63) // 64
    if out is None:
        out = torch.empty((rows.shape[0], nw_out), dtype=i64, device=rows.device)
    call("bff_permute_bits", _ptr(rows, i64), rows.shape[0], rows.shape[1], _ptr(idx, i32), n_out, nw_out, _ptr(out))
    return out


def scatter_bits(rows, perm, n_out, out=None):
    """out[r] bit perm[s] = rows[r] bit s for the set bits only (undoes the spatial point sort; `out` is zeroed here
    unless given, in which case it must be zero).  perm: int32 [n], original index of sorted position s."""
    nw_out = (n_out + 63) // 64
    if out is None:
        out = torch.zeros((rows.shape[0], nw_out), dtype=i64, device=rows.device)
    call("bff_scatter_bits", _ptr(rows, i64), rows.shape[0], rows.shape[1], _ptr(perm, i32), n_out, nw_out, _ptr(out), None)
    return out


def components(adj, max_rounds=10_000):
    """label[i] = smallest index of i's connected component (iterates bff_components_round)."""
    n = adj.shape[0]
    a = torch.arange(n, dtype=i32, device=adj.device)
    b = torch.empty_like(a)
    changed = torch.zeros(1, dtype=i32, device=adj.device)
    for _ in range(max_rounds):
        changed.zero_()
        call("bff_components_round", _ptr(adj, i64), n, _ptr(a), _ptr(b), _ptr(changed))
        a, b = b, a
        if int(changed.item()) == 0:
            return a
    raise RuntimeError("bff_components_round did not converge")


def or_reduce_groups(rows, group_offs, members, max_group_size, conf=None, chunk_mask=None):
    """out[g] = OR of the member rows.  conf given (float16/float32 per row): also returns the groups' sequential
    confidence means (group_conf_mean), computed by extra blocks of the same launch."""
    k = group_offs.shape[0] - 1
    out = torch.empty((k, rows.shape[1]), dtype=i64, device=rows.device)
    mean = None
    if conf is not None:
        if conf.dtype not in (torch.float16, torch.float32):
            raise TypeError(f"confidences must be float16 or float32, got {conf.dtype}")
        mean = torch.empty(k, dtype=conf.dtype, device=conf.device)
    call("bff_or_reduce_groups", _ptr(rows, i64), rows.shape[1], _ptr(group_offs, i32), _ptr(members, i32), k,
         int(max_group_size), _ptr(out), _ptr(conf), 1 if (conf is not None and conf.dtype == torch.float16) else 0,
         _ptr(mean), _ptr(chunk_mask, i64))
    return out if conf is None else (out, mean)


def group_conf_mean(conf, group_offs, members):
    if conf.dtype not in (torch.float16, torch.float32):
        raise TypeError(f"confidences must be float16 or float32, got {conf.dtype}")
    k = group_offs.shape[0] - 1
    out = torch.empty(k, dtype=conf.dtype, device=conf.device)
    call("bff_group_conf_mean", _ptr(conf), 1 if conf.dtype == torch.float16 else 0, _ptr(group_offs, i32),
         _ptr(members, i32), k, _ptr(out))
    return out


def apply_row_ops(rows, ops):
    call("bff_apply_row_ops", _ptr(rows, i64), rows.shape[1], _ptr(ops, i32), ops.shape[0])


def resolve_overlaps(rows, sizes):
    """solve_overlapping P:277-301 on bit rows, in place, without leaving the device.
    sizes: int32 device tensor, number of raw masks merged into each row."""
    if rows.shape[0] >= 2:
        resolve_overlaps_filtered(rows, sizes, None)


def resolve_overlaps_replay(rows, sizes):
    """The same as the reference spells it: the ordered list of overlapping pairs (intersections before any edit), then
    one and-not per pair in that order.  Three launches; the cross-check of the closed form and the path for more rows
    than bff_resolve_overlaps_max_rows()."""
    k = rows.shape[0]
    if k < 2:
        return
    inter = cross_popcount(rows, rows)
    ops = torch.empty(1 + 3 * (k * (k - 1) // 2), dtype=i32, device=rows.device)
    call("bff_overlap_ops", _ptr(inter, i32), _ptr(sizes, i32), k, _ptr(ops))
    call("bff_apply_row_ops", _ptr(rows, i64), rows.shape[1], _ptr(ops), -1)


def resolve_overlaps_filtered(rows, sizes, keep):
    """solve_overlapping (P:277-301), `&= keep` (P:595; None: no filter) and the popcounts before / after (P:592, 596) in
    one launch -> (before, after) int32 device tensors.  More than bff_resolve_overlaps_max_rows() rows: the ordered
    replay and the separate steps."""
    k = rows.shape[0]
    if k <= load().bff_resolve_overlaps_max_rows():
        before = torch.empty(k, dtype=i32, device=rows.device)
        after = torch.empty(k, dtype=i32, device=rows.device)
        call("bff_resolve_overlaps", _ptr(rows, i64), k, rows.shape[1], _ptr(sizes, i32),
             _ptr(keep, i64) if keep is not None else None, _ptr(before), _ptr(after))
        return before, after
    before = popcount_rows(rows)
    resolve_overlaps_replay(rows, sizes)
    if keep is not None:
        and_rows(rows, keep)
    return before, popcount_rows(rows)


def and_rows(rows, keep):
    call("bff_and_rows", _ptr(rows, i64), rows.shape[0], rows.shape[1], _ptr(keep, i64))


def gather_rows(rows, idx):
    out = torch.empty((idx.shape[0], rows.shape[1]), dtype=i64, device=rows.device)
    call("bff_gather_rows", _ptr(rows, i64), _ptr(idx, i32), idx.shape[0], rows.shape[1], _ptr(out))
    return out


def unpack_rows(rows, n_points):
    dense = torch.empty((rows.shape[0], n_points), dtype=torch.bool, device=rows.device)
    call("bff_unpack_rows", _ptr(rows, i64), rows.shape[0], rows.shape[1], n_points, _ptr(dense))
    return dense


def pack_rows(dense):
    if dense.dtype not in (torch.bool, torch.uint8):
        raise TypeError("pack_rows takes bool/uint8 rows")
    n = dense.shape[1]
    nw = (n + 63) // 64
    rows = torch.empty((dense.shape[0], nw), dtype=i64, device=dense.device)
    call("bff_pack_rows", _ptr(dense), dense.shape[0], n, nw, _ptr(rows))
    return rows


def rle_to_rows(run_start, run_end, row_run_offs, n_points, out=None):
    k = row_run_offs.shape[0] - 1
    nw = (n_points + 63) // 64
    rows = out if out is not None else torch.empty((k, nw), dtype=i64, device=run_start.device)
    call("bff_rle_to_rows", _ptr(run_start, i32), _ptr(run_end, i32), _ptr(row_run_offs, i32), k, n_points, nw, _ptr(rows))
    return rows


def ids_to_rows(ids, values):
    """int64 id per point (device) + int64 values (device) -> bit rows [len(values)][nw]: rows[v] = (ids == values[v])."""
    n = ids.shape[0]
    nw = (n + 63) // 64
    rows = torch.empty((values.shape[0], nw), dtype=i64, device=ids.device)
    call("bff_ids_to_rows", _ptr(ids, i64), n, _ptr(values, i64), values.shape[0], nw, _ptr(rows))
    return rows


def rows_to_rle(rows, n_points):
    """Bit rows -> list of {"length", "counts"} in the reference's RLE format (rle_encode_batch RLE:10-32)."""
    import numpy as np
    k = rows.shape[0]
    if k == 0:
        return []
    n_runs = torch.empty(k, dtype=i32, device=rows.device)
    call("bff_rle_count_runs", _ptr(rows, i64), k, rows.shape[1], _ptr(n_runs))
    nr = n_runs.cpu().numpy().astype(np.int64)
    offs = np.zeros(k + 1, dtype=np.int64)
    np.cumsum(nr, out=offs[1:])
    total = int(offs[-1])
    counts = torch.empty(2 * total, dtype=i64, device=rows.device)
    call("bff_rle_encode_rows", _ptr(rows, i64), k, rows.shape[1], _ptr(torch.from_numpy(offs[:-1].copy()).to(rows.device)),
         total, _ptr(counts))
    flat = counts.cpu().numpy()
    return [dict(length=int(n_points), counts=flat[2 * offs[r]:2 * offs[r + 1]].copy()) for r in range(k)]


_temp_bytes = {}


def _sort_temp(kind, n, device, query):
    """Device scratch for a library sort of n keys (size from the ABI's query call, cached per (kind, n))."""
    key = (kind, n)
    if key not in _temp_bytes:
        need = ctypes.c_size_t(0)
        query(ctypes.byref(need))
        _temp_bytes[key] = int(need.value)
    nbytes = _temp_bytes[key]
    return torch.empty(max(nbytes, 1), dtype=torch.uint8, device=device), ctypes.c_size_t(nbytes)


def sort_f32(vals):
    """Ascending sort of a float32 device vector (rocPRIM radix sort behind bff_sort_f32)."""
    n = vals.shape[0]
    out = torch.empty_like(vals)
    temp, nb = _sort_temp("f32", n, vals.device,
                          lambda need: call("bff_sort_f32", None, None, n, None, need))
    call("bff_sort_f32", _ptr(vals, f32), _ptr(out), n, _ptr(temp), ctypes.byref(nb))
    return out


SIGNATURE_BITS = 30       # bff_row_stats signatures are 30-bit keys


def argsort_i64(keys, key_bits=64):
    """Stable ascending argsort of an int64 device vector -> int32 order (bff_argsort_i64).  key_bits < 64: the
    keys are known to be non-negative and below 2^key_bits (fewer radix passes)."""
    n = keys.shape[0]
    order = torch.empty(n, dtype=i32, device=keys.device)
    scratch = torch.empty_like(keys)
    temp, nb = _sort_temp(("i64", key_bits), n, keys.device,
                          lambda need: call("bff_argsort_i64", None, None, None, n, key_bits, None, need))
    call("bff_argsort_i64", _ptr(keys, i64), _ptr(scratch), _ptr(order), n, key_bits, _ptr(temp), ctypes.byref(nb))
    return order


def point_threshold(masked, viewed, fraction):
    """Device-resident threshold unique()[floor(fraction * n_unique)] of masked/(viewed+1) (or of masked when
    viewed is None), P:513-518 / 571-576.  Returns (thr f32[1], n_unique i32[1]) on the device."""
    n = masked.shape[0]
    dev = masked.device
    vals = torch.empty(n, dtype=f32, device=dev)
    call("bff_point_values", _ptr(masked, i32), _ptr(viewed, i32), n, _ptr(vals))
    vals = sort_f32(vals)
    scratch = torch.empty(max(1, (n + 1023) // 1024), dtype=i32, device=dev)
    thr = torch.empty(1, dtype=f32, device=dev)
    n_unique = torch.empty(1, dtype=i32, device=dev)
    call("bff_select_unique_rank", _ptr(vals), n, float(fraction), _ptr(scratch), _ptr(thr), _ptr(n_unique))
    return thr, n_unique


def point_threshold_pairs(masked, viewed, fraction):
    """point_threshold without sorting: the statistic is a function of the integer pair (masked, viewed); every block
    lists the distinct float32 values of its points, one block merges the lists in an LDS set and radix-selects the
    rank.  Returns (thr f32[1], n_unique i32[1], overflow i32[1]) on the device; overflow != 0: more distinct values
    than the set holds (bff_point_threshold_capacity()), use point_threshold."""
    n = masked.shape[0]
    dev = masked.device
    scratch = torch.empty(int(load().bff_point_threshold_scratch_words(n)), dtype=i32, device=dev)
    thr = torch.empty(1, dtype=f32, device=dev)
    n_unique = torch.empty(1, dtype=i32, device=dev)
    overflow = torch.zeros(1, dtype=i32, device=dev)
    call("bff_point_threshold_pairs", _ptr(masked, i32), _ptr(viewed, i32), n, float(fraction), _ptr(scratch), _ptr(thr),
         _ptr(n_unique), _ptr(overflow))
    return thr, n_unique, overflow


_pinned = {}


def host_component_csr(comp, has_self_loop, min_members):
    """numpy in, numpy out: (offs int32 [K+1], members int32, sizes int32 [K], n_void) or None if an id is out
    of range (caller falls back to the NumPy implementation)."""
    import numpy as np
    n = comp.shape[0]
    comp = np.ascontiguousarray(comp, dtype=np.int32)
    sl = np.ascontiguousarray(has_self_loop, dtype=np.uint8)
    offs = np.empty(n + 1, dtype=np.int32)
    members = np.empty(max(n, 1), dtype=np.int32)
    sizes = np.empty(max(n, 1), dtype=np.int32)
    n_void = c_int32(0)
    p = lambda a: a.ctypes.data_as(c_void_p)
    k = load().bff_host_component_csr(p(comp), p(sl), n, int(min_members), p(offs), p(members), p(sizes), ctypes.byref(n_void))
    if k < 0:
        return None
    return offs[:k + 1], members[:offs[k]], sizes[:k], int(n_void.value)


_up_ring = {}             # (dtype, capacity) -> [pinned buffers, their numpy views, events, next]
_NP_OF = {torch.int32: "int32", torch.int64: "int64", torch.float32: "float32", torch.float16: "float16"}


def upload(data, dtype, device):
    """Small host array / list -> device tensor without blocking the host: the values go through a ring of
    pinned staging buffers and an asynchronous copy on the current stream (a pageable `.to(device)` would stall
    the host until everything queued on the stream before it has finished)."""
    import numpy as np
    a = data.numpy() if torch.is_tensor(data) else np.asarray(data)
    shape, n = a.shape, a.size
    if torch.device(device).type != "cuda" or n == 0:
        return torch.as_tensor(a).to(dtype).to(device)
    cap = max(64, 1 << (n - 1).bit_length())
    ring = _up_ring.get((dtype, cap))
    if ring is None:
        bufs = [torch.empty(cap, dtype=dtype, pin_memory=True) for _ in range(8)]
        ring = _up_ring[(dtype, cap)] = [bufs, [b.numpy() for b in bufs], [None] * 8, 0]
    k = ring[3]
    ring[3] = (k + 1) & 7
    lib = load()
    ev = ring[2][k]
    if ev is not None:
        lib.bff_event_synchronize(ev)          # the copy that last used this buffer (8 uploads ago) is long done
    else:
        ev = ring[2][k] = c_void_p(lib.bff_event_create())     # a hipEvent_t: recorded on the raw stream handle below
        if not ev:
            raise BffLibraryError("bff_event_create failed")
    np.copyto(ring[1][k][:n], a.reshape(-1), casting="same_kind")
    out = ring[0][k][:n].to(device, non_blocking=True)
    if lib.bff_event_record(ev, _stream()) != 0:               # on the current stream (no torch stream lookup)
        raise RuntimeError(f"bff_event_record failed: {lib.bff_last_error().decode()}")
    return out.reshape(shape)


sync_wait_s = 0.0          # wall time spent blocked in fetch()'s synchronisations (a host-side profile counter)


def fetch(*tensors):
    """Device tensors -> numpy arrays with ONE stream synchronisation (async copies into reused pinned
    staging buffers; the returned arrays are copies, so the staging can be reused by the next call)."""
    host = []
    for k, t in enumerate(tensors):
        key = (k, t.dtype, t.numel())
        buf = _pinned.get(key)
        if buf is None:
            if len(_pinned) > 256:
                _pinned.clear()
            buf = _pinned[key] = torch.empty(t.numel(), dtype=t.dtype, pin_memory=True)
        buf.copy_(t.reshape(-1), non_blocking=True)
        host.append((buf, t.shape))
    global sync_wait_s
    t0 = time.perf_counter()
    (_cached_stream()[1] if _cached_stream() is not None else torch.cuda.current_stream()).synchronize()
    sync_wait_s += time.perf_counter() - t0
    return [b.numpy().reshape(shape).copy() for b, shape in host]


def ratio_keep(masked, viewed, thr, use_thr):
    """thr: python float, or a float32 device tensor of one element (threshold stays on the device)."""
    n = masked.shape[0]
    nw = (n + 63) // 64
    keep = torch.empty(nw, dtype=i64, device=masked.device)
    thr_dev = thr if torch.is_tensor(thr) else None
    call("bff_ratio_keep", _ptr(masked, i32), _ptr(viewed, i32), n, 0.0 if thr_dev is not None else float(thr),
         _ptr(thr_dev, f32), int(bool(use_thr)), nw, _ptr(keep))
    return keep


def depth_from_u16(raw, height, width, taps=None, depth_scale=1000.0):
    """uint16-as-int16 device tensor [F][h][w] (raw millimetres) -> float32 [F][height*width] metres, resized on
    the device.  taps: (x0, x1, ax, y0, y1, ay) device tensors from io.bilinear_taps (None if sizes are equal)."""
    f, hs, ws = raw.shape
    out = torch.empty((f, height * width), dtype=f32, device=raw.device)
    t = taps if taps is not None else (None,) * 6
    call("bff_depth_from_u16", _ptr(raw, torch.int16), f, hs, ws, _ptr(t[0], i32), _ptr(t[1], i32), _ptr(t[2], f32),
         _ptr(t[3], i32), _ptr(t[4], i32), _ptr(t[5], f32), height, width, float(depth_scale), _ptr(out))
    return out


def cosine_gemm_f16(a, b):
    if a.dtype != torch.float16 or b.dtype != torch.float16:
        raise TypeError("cosine_gemm_f16 takes float16 operands")
    out = torch.empty((a.shape[0], b.shape[0]), dtype=f32, device=a.device)
    call("bff_cosine_gemm_f16", _ptr(a), a.shape[0], _ptr(b), b.shape[0], a.shape[1], _ptr(out))
    return out


def cosine_rows(a, b):
    """cos[i][j] of float16 or float32 embedding rows, rounded op by op in that dtype like the reference's
    tensor expression (R:109-114) -> float32 [na][nb] (float16 values when the inputs are float16)."""
    if a.dtype != b.dtype or a.dtype not in (torch.float16, torch.float32):
        raise TypeError("cosine_rows takes two float16 or two float32 tensors")
    if a.shape[1] != b.shape[1]:
        raise ValueError("cosine_rows: embedding widths differ")
    out = torch.empty((a.shape[0], b.shape[0]), dtype=f32, device=a.device)
    call("bff_cosine_rows", _ptr(a), a.shape[0], _ptr(b), b.shape[0], a.shape[1],
         1 if a.dtype == torch.float16 else 0, _ptr(out))
    return out


def normalized_gemm_f16(a, b):
    """F.normalize(a) @ b.T for float16 rows (SEG:388-393: b = the already normalised text means): float32 [na][nb]."""
    if a.dtype != torch.float16 or b.dtype != torch.float16:
        raise TypeError("normalized_gemm_f16 takes float16 operands")
    out = torch.empty((a.shape[0], b.shape[0]), dtype=f32, device=a.device)
    call("bff_normalized_gemm_f16", _ptr(a), a.shape[0], _ptr(b), b.shape[0], a.shape[1], _ptr(out))
    return out


def description_means(desc, offs):
    """compute_avg_description_encodings (SEG:324-337) on stacked description encodings: desc float16/float32 [n][dim],
    offs int32 [n_classes + 1] -> normalised class means [n_classes][dim] in desc's dtype."""
    if desc.dtype not in (torch.float16, torch.float32):
        raise TypeError("description_means takes float16 or float32 encodings")
    out = torch.empty((offs.shape[0] - 1, desc.shape[1]), dtype=desc.dtype, device=desc.device)
    call("bff_description_means", _ptr(desc), _ptr(offs, i32), offs.shape[0] - 1, desc.shape[1],
         1 if desc.dtype == torch.float16 else 0, _ptr(out))
    return out
