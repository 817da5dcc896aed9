"""PCIe-inclusive cost of one config-2 scene: prepare_scene (host arrays -> HBM) for the f32 depth boundary and for
the raw 16-bit depth path, next to the on-device step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from beyond_fixed_forms_amd import _lib
from beyond_fixed_forms_amd.config import Config
from beyond_fixed_forms_amd.projection import run_projection
from beyond_fixed_forms_amd.scene import prepare_scene
from beyond_fixed_forms_amd.synthetic import make_scene
_lib.load()
dev = "cuda"
shape = sys.argv[1] if len(sys.argv) > 1 else "c2"
scene = make_scene(shape, seed=0, device=dev, query="table")
cfg = Config.with_defaults(width_2d=scene.width, height_2d=scene.height)
def timed(f, reps=3):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): out = f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps, out
gb_f32 = sum(d.nbytes for d in scene.depths.values()) / 1e9
t_f32, ds = timed(lambda: prepare_scene(scene, cfg, device=dev))
print(f"prepare_scene, f32 depth at HxW ({gb_f32:.2f} GB of depth): {t_f32 * 1e3:8.1f} ms  -> {1 / t_f32:6.1f} scenes/s upload-bound")
# the same scene with its depth as 16-bit frames at ScanNet's native 480x640 (values resampled; timing only)
import copy
import numpy as np
hs, ws = 480, 640
yy = (np.arange(hs) * scene.height // hs)[:, None]
xx = (np.arange(ws) * scene.width // ws)[None, :]
sc_raw = copy.copy(scene)
sc_raw.depths_raw = {k: np.ascontiguousarray((d[yy, xx] * 1000.0).round().astype(np.uint16)) for k, d in scene.depths.items()}
sc_raw.depths = {}
gb_raw = sum(d.nbytes for d in sc_raw.depths_raw.values()) / 1e9
t_raw, _ = timed(lambda: prepare_scene(sc_raw, cfg, device=dev))
print(f"prepare_scene, raw uint16 depth at {hs}x{ws} ({gb_raw:.2f} GB), /1000 + resize on the device: {t_raw * 1e3:8.1f} ms  -> {1 / t_raw:6.1f} scenes/s")
t_run, _ = timed(lambda: run_projection(ds, cfg), reps=10)
print(f"run_projection on resident inputs: {t_run * 1e3:8.2f} ms")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); prepare_scene(scene, cfg, device=dev); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
