"""Where does merge_components spend its time at config 2?  (thr variants: 2.0 -> prologue only)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from beyond_fixed_forms_amd import _lib
from beyond_fixed_forms_amd.config import Config
from beyond_fixed_forms_amd.projection import run_projection
from beyond_fixed_forms_amd.scene import prepare_scene
from beyond_fixed_forms_amd.synthetic import make_scene
dev = "cuda:0"
scene = make_scene("c2", seed=0, device=dev)
cfg = Config.with_defaults(width_2d=scene.width, height_2d=scene.height)
ds = prepare_scene(scene, cfg, device=dev)
res = run_projection(ds, cfg, debug_out=True)
rows = _lib.permute_bits(res.debug["raw_rows"], torch.argsort(ds.unsort).to(torch.int32), ds.n_points)
area, mean_word, cmask, hist, sig = _lib.row_stats(rows)
def timeit(name, fn, reps=10):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); print(f"{name:46s} {1e6*(time.perf_counter()-t)/reps:8.1f} us")
o_sig = torch.argsort(sig, stable=True).to(torch.int32)
o_mean = torch.argsort(mean_word.long()).to(torch.int32)
o_id = torch.arange(rows.shape[0], dtype=torch.int32, device=dev)
for name, o in (("signature", o_sig),):
    for thr in (0.2, 0.9, 2.0):
        timeit(f"merge_components order={name} thr={thr}", lambda: _lib.merge_components(rows, area, ds.label_id, thr, o, cmask, hist))
timeit("row_stats", lambda: _lib.row_stats(rows))
timeit("argsort(sig)", lambda: torch.argsort(sig, stable=True))

for cs in (0,):
    timeit(f"merge_components signature thr=0.2 coarse_stride={cs}", lambda: _lib.merge_components(rows, area, ds.label_id, 0.2, o_sig, cmask, hist, coarse_stride=cs))
    d = torch.zeros(16, dtype=torch.int32, device=dev)      # counters + phase clocks; d[15] = 0: no block timeline
    _lib.merge_components(rows, area, ds.label_id, 0.2, o_sig, cmask, hist, diag=d, coarse_stride=cs)
    print("   coarse", cs, "tiles evaluated, chunk visits, candidate pairs, unions:", d.tolist()[:4])
for name, o in (("signature", o_sig),):
    for thr in (0.2, 0.9):
        d = torch.zeros(16, dtype=torch.int32, device=dev)      # counters + phase clocks; d[15] = 0: no block timeline
        _lib.merge_components(rows, area, ds.label_id, thr, o, cmask, hist, diag=d)
        print(name, thr, "tiles evaluated, chunk visits, candidate pairs, unions:", d.tolist()[:4])

# floor: start from the converged forest
comp = _lib.merge_components(rows, area, ds.label_id, 0.2, o_sig, cmask, hist)
d = torch.zeros(16, dtype=torch.int32, device=dev)      # counters + phase clocks; d[15] = 0: no block timeline
_lib.merge_components(rows, area, ds.label_id, 0.2, o_sig, cmask, hist, diag=d, parent=comp.clone())
print("converged start: tiles evaluated, chunk visits, candidate pairs, unions:", d.tolist()[:4])
pc = comp.clone()
timeit("merge_components from converged forest", lambda: _lib.merge_components(rows, area, ds.label_id, 0.2, o_sig, cmask, hist, parent=pc))
