"""Where the projection sweep's time goes (config 2 by default): the kernel with parts of its work removed.
  geometry only      intrinsics shifted so that no pixel is in bounds: transform + divide + round + bounds test, no gather
  + depth            no mask frames: geometry + depth gathers + visibility (+ viewed counter)
  full               production form (palette / word segments, rows, both counters)
each with and without frustum culling, for float32 (H, W) depth and for the raw uint16 frames resized per point."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from beyond_fixed_forms_amd import _lib
from beyond_fixed_forms_amd.config import Config
from beyond_fixed_forms_amd.projection import DEPTH_THRESH
from beyond_fixed_forms_amd.scene import prepare_scene
from beyond_fixed_forms_amd.synthetic import make_scene, with_sensor_depth

_lib.load()
shape = sys.argv[1] if len(sys.argv) > 1 else "c2"
dev = "cuda"
scene = make_scene(shape, seed=0, device=dev, query="table")
cfg = Config.with_defaults(width_2d=scene.width, height_2d=scene.height)
ds_f32 = prepare_scene(scene, cfg, device=dev)
os.environ["BFF_DEPTH_TILES"] = "0"
ds_u16 = prepare_scene(with_sensor_depth(scene), cfg, device=dev)
os.environ["BFF_DEPTH_TILES"] = "u16"
ds_u16t = prepare_scene(with_sensor_depth(scene), cfg, device=dev)
os.environ["BFF_DEPTH_TILES"] = "f32"
ds_f32t = prepare_scene(with_sensor_depth(scene), cfg, device=dev)
n, nw, hw = ds_f32.n_points, ds_f32.nw, ds_f32.height * ds_f32.width
n_mviews = ds_f32.view_mask_offs.shape[0] - 1
wdt = torch.int32 if ds_f32.word_bits == 32 else torch.int64
maskbits = torch.empty((n_mviews, hw), device=dev, dtype=wdt)
labels = torch.empty((n_mviews, _lib.label_plane_stride(hw)), device=dev, dtype=torch.uint8)
segmap = torch.empty((n_mviews, 2 * _lib.segmap_words(hw)), dtype=torch.int32, device=dev)
_lib.rle_to_labels(ds_f32.run_start, ds_f32.run_end, ds_f32.mask_run_offs, ds_f32.view_mask_offs, n_mviews, hw, ds_f32.word_bits,
                   labels, maskbits, segmap)
rows = torch.zeros((ds_f32.n_rows, nw), dtype=torch.int64, device=dev)
masked = torch.zeros(n, dtype=torch.int32, device=dev)
viewed = torch.zeros(n, dtype=torch.int32, device=dev)
cm = _lib.chunk_mask_buffer(ds_f32.n_rows, nw, dev).zero_()


def timeit(f, reps=10):
    for _ in range(2):
        f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def variants(ds, tag):
    k_off = np.array(ds.cam_intr, dtype=np.float64).copy()
    k_off[0, 2] = -1e9                      # every u far below 0: nothing in bounds
    for cull in (True, False):
        tb = ds.tile_bounds if cull else None
        geo = lambda: _lib.project_views(ds.xyz, n, ds.inv_pose, k_off, ds.sweep_depth, ds.depth_index, ds.height, ds.width,
                                         DEPTH_THRESH, None, ds.word_bits, ds.frame_mask, ds.frame_rowbase, ds.frame_nmask,
                                         ds.frame_flags, None, masked, viewed, None, None, None, depth_size=ds.depth_size)
        dep = lambda: _lib.project_views(ds.xyz, n, ds.inv_pose, ds.cam_intr, ds.sweep_depth, ds.depth_index, ds.height, ds.width,
                                         DEPTH_THRESH, None, ds.word_bits, ds.frame_mask, ds.frame_rowbase, ds.frame_nmask,
                                         ds.frame_flags, None, masked, viewed, None, None, tb, depth_size=ds.depth_size)
        full = lambda: _lib.project_views(ds.xyz, n, ds.inv_pose, ds.cam_intr, ds.sweep_depth, ds.depth_index, ds.height, ds.width,
                                          DEPTH_THRESH, maskbits, ds.word_bits, ds.frame_mask, ds.frame_rowbase, ds.frame_nmask,
                                          ds.frame_flags, rows, masked, viewed, segmap, cm, tb, labels=labels, depth_size=ds.depth_size)
        k_same = np.array([[0.0, 0.0, ds.width / 2.0], [0.0, 0.0, ds.height / 2.0], [0.0, 0.0, 1.0]])     # every point -> one pixel
        dep_same = lambda: _lib.project_views(ds.xyz, n, ds.inv_pose, k_same, ds.sweep_depth, ds.depth_index, ds.height, ds.width,
                                              DEPTH_THRESH, None, ds.word_bits, ds.frame_mask, ds.frame_rowbase, ds.frame_nmask,
                                              ds.frame_flags, None, masked, viewed, None, None, None, depth_size=ds.depth_size)
        line = f"{tag:9s} culling {'on ' if cull else 'off'}:"
        if not cull:
            line += f"  depth, all points on ONE pixel {timeit(dep_same):7.1f} us"
        if not cull:
            line += f"  geometry only {timeit(geo):7.1f} us"
        line += f"  + depth {timeit(dep):7.1f} us   full {timeit(full):7.1f} us"
        print(line, flush=True)


variants(ds_f32, "f32")
variants(ds_f32t, "f32 tiles")
viewed.zero_()
_lib.project_views(ds_f32.xyz, n, ds_f32.inv_pose, ds_f32.cam_intr, ds_f32.depth, ds_f32.depth_index, ds_f32.height, ds_f32.width,
                   DEPTH_THRESH, None, ds_f32.word_bits, ds_f32.frame_mask, ds_f32.frame_rowbase, ds_f32.frame_nmask,
                   torch.ones_like(ds_f32.frame_flags), None, None, viewed, None, None, None)
torch.cuda.synchronize()
print("visible (point, frame) pairs: %.1f M of %.1f M" % (viewed.sum().item() / 1e6, n * ds_f32.n_frames / 1e6))
