"""merge_components at config 2: one launch vs near-diagonal launch first (BFF_MERGE_SPLIT=0/1, read per call)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from beyond_fixed_forms_amd import _lib
from beyond_fixed_forms_amd.config import Config
from beyond_fixed_forms_amd.projection import projection_front
from beyond_fixed_forms_amd.scene import prepare_scene
from beyond_fixed_forms_amd.synthetic import make_scene
_lib.load()
dev = "cuda"
shape = sys.argv[1] if len(sys.argv) > 1 else "c2"
scene = make_scene(shape, seed=0, device=dev, query="table")
cfg = Config.with_defaults(width_2d=scene.width, height_2d=scene.height)
ds = prepare_scene(scene, cfg, device=dev)
fr = projection_front(ds, cfg)
rows = fr.rows
area, mw_, cmask, hist, sig = _lib.row_stats(rows)
order = _lib.argsort_i64(sig, 30)
def timeit(name, f, reps=20):
    for _ in range(3): f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    print(f"{name:40s} {a.elapsed_time(b) / reps * 1e3:8.1f} us")
ref = None
for split in ("0", "0"):
    os.environ["BFF_MERGE_SPLIT"] = split
    comp = _lib.merge_components(rows, area, ds.label_id, cfg.iou_thres, order, cmask, hist)
    ref = comp if ref is None else ref
    assert torch.equal(comp, ref)
    d = torch.zeros(4, dtype=torch.int32, device=dev)
    _lib.merge_components(rows, area, ds.label_id, cfg.iou_thres, order, cmask, hist, diag=d)
    timeit(f"merge_components split={split}", lambda: _lib.merge_components(rows, area, ds.label_id, cfg.iou_thres, order, cmask, hist))
    print("   tiles evaluated, chunk visits, candidate pairs, unions:", d.tolist())
