"""Time the 2-D RLE decoder alone on one synthetic scene (config 2 by default), in its three output forms."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from beyond_fixed_forms_amd import _lib
from beyond_fixed_forms_amd.config import Config
from beyond_fixed_forms_amd.scene import prepare_scene
from beyond_fixed_forms_amd.synthetic import make_scene

_lib.load()
shape = sys.argv[1] if len(sys.argv) > 1 else "c2"
dev = "cuda"
scene = make_scene(shape, seed=0, device=dev, query="table", cut_masks=False) if "many" in sys.argv else make_scene(shape, seed=0, device=dev, query="table")
cfg = Config.with_defaults(width_2d=scene.width, height_2d=scene.height)
ds = prepare_scene(scene, cfg, device=dev)
hw = ds.height * ds.width
n_mviews = ds.view_mask_offs.shape[0] - 1
wdt = torch.int32 if ds.word_bits == 32 else torch.int64
maskbits = torch.empty((n_mviews, hw), device=dev, dtype=wdt)
labels = torch.empty((n_mviews, _lib.label_plane_stride(hw)), device=dev, dtype=torch.uint8)
segmap = torch.empty((n_mviews, 2 * _lib.segmap_words(hw)), dtype=torch.int32, device=dev)
print("mask views", n_mviews, "pixels", hw, "masks", ds.mask_run_offs.shape[0] - 1, "runs", ds.run_start.shape[0], "word bits", ds.word_bits)


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


lab = lambda: _lib.rle_to_labels(ds.run_start, ds.run_end, ds.mask_run_offs, ds.view_mask_offs, n_mviews, hw, ds.word_bits,
                                 labels, maskbits, segmap)
print("palette form  %7.1f us" % timeit(lab))
seg = (segmap.view(n_mviews, -1, 2)[:, :, 0].contiguous().view(torch.int32))
nz = sum(bin(x & 0xFFFFFFFF).count("1") for x in seg.flatten().tolist())
print("segments with a mask pixel: %d of %d" % (nz, n_mviews * ((hw + 127) // 128)))
