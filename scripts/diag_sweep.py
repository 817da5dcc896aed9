"""Sweep at config 2 with and without the mask gathers; run / row-segment statistics of the 2-D RLE masks."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from beyond_fixed_forms_amd import _lib
from beyond_fixed_forms_amd.config import Config
from beyond_fixed_forms_amd.projection import DEPTH_THRESH
from beyond_fixed_forms_amd.scene import prepare_scene
from beyond_fixed_forms_amd.synthetic import make_scene
_lib.load()

scene = make_scene("c2", seed=0, device="cuda", query="chair")
cfg = Config.with_defaults(width_2d=scene.width, height_2d=scene.height)
ds = prepare_scene(scene, cfg, device="cuda")
dev = "cuda"
n, nw = ds.n_points, ds.nw
hw = ds.height * ds.width
n_mviews = ds.view_mask_offs.shape[0] - 1
maskbits = torch.empty((n_mviews, hw), device=dev, dtype=torch.int32)
segmap = torch.empty((n_mviews, _lib.segmap_words(hw)), dtype=torch.int32, device=dev)
_lib.rle_to_maskbits(ds.run_start, ds.run_end, ds.mask_run_offs, ds.view_mask_offs, n_mviews, hw, ds.word_bits, maskbits, segmap)
rows = torch.zeros((ds.n_rows, nw), dtype=torch.int64, device=dev)
masked = torch.zeros(n, dtype=torch.int32, device=dev)
viewed = torch.zeros(n, dtype=torch.int32, device=dev)
cm = _lib.chunk_mask_buffer(ds.n_rows, nw, dev).zero_()
def timeit(f, reps=20):
    for _ in range(3): f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
full = lambda: _lib.project_views(ds.xyz, n, ds.inv_pose, ds.cam_intr, ds.depth, ds.depth_index, ds.height, ds.width, DEPTH_THRESH,
                                  maskbits, ds.word_bits, ds.frame_mask, ds.frame_rowbase, ds.frame_nmask, ds.frame_flags, rows, masked, viewed, segmap, cm)
nomask = lambda: _lib.project_views(ds.xyz, n, ds.inv_pose, ds.cam_intr, ds.depth, ds.depth_index, ds.height, ds.width, DEPTH_THRESH,
                                    None, ds.word_bits, ds.frame_mask, ds.frame_rowbase, ds.frame_nmask, ds.frame_flags, None, masked, viewed, None)
print("sweep full      %.1f us" % timeit(full))
print("row_stats dense  %.1f us" % timeit(lambda: _lib.row_stats(rows)))
print("row_stats sparse %.1f us" % timeit(lambda: _lib.row_stats(rows, cm)))
print("sweep no masks  %.1f us" % timeit(nomask))
print("decode          %.1f us" % timeit(lambda: _lib.rle_to_maskbits(ds.run_start, ds.run_end, ds.mask_run_offs, ds.view_mask_offs, n_mviews, hw, ds.word_bits, maskbits, segmap)))
viewed.zero_(); nomask(); torch.cuda.synchronize()
print("visible (point, frame) pairs: %.1f M of %.1f M" % (viewed.sum().item() / 1e6, n * ds.n_frames / 1e6))
rs, re = ds.run_start.cpu().long(), ds.run_end.cpu().long()
print("runs: %d (%.0f per view); mean run length %.1f px" % (rs.numel(), rs.numel() / n_mviews, (re - rs).float().mean().item()))
W = ds.width
span_rows = ((re - 1) // W - rs // W + 1)
print("row segments: %d (runs crossing a row boundary: %d); per (view,row): %.2f" % (span_rows.sum().item(), (span_rows > 1).sum().item(), span_rows.sum().item() / (n_mviews * ds.height)))
print("nonzero segmap fraction: %.3f" % (sum(bin(int(x) & 0xffffffff).count("1") for x in segmap.cpu().flatten()[:20000].tolist()) / (20000 * 32)))
