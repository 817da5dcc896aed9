"""Phase clocks of the components' tile pass (diagnostic build of the kernel, BFF_MERGE_DIAG=1: thread 0 of every block
adds the cycles of each phase; 2 blocks per CU instead of 3).  usage: python scripts/diag_merge_phases.py [c2|c4] [default|many]"""
import os, sys
os.environ["BFF_MERGE_DIAG"] = "1"
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
kind = sys.argv[2] if len(sys.argv) > 2 else "default"
var = dict(cut_masks=False, n_objects=40, distinct_masks=True, dilate=False) if kind == "many" else {}
scene = make_scene(shape, seed=0, device=dev, query="table", **var)
cfg = Config.with_defaults(width_2d=scene.width, height_2d=scene.height)
ds = prepare_scene(scene, cfg, device=dev)
fr = projection_front(ds, cfg, fast=False)
area, _mw, cmask, hist, sig = _lib.row_stats(fr.rows, fr.cmask)
order = _lib.argsort_i64(sig, _lib.SIGNATURE_BITS)
for rep in range(2):
    d = torch.zeros(16, dtype=torch.int32, device=dev)
    _lib.merge_components(fr.rows, area, ds.label_id, cfg.iou_thres, order, cmask, hist, diag=d)
v = d.cpu().tolist()
names = {4: "rows, roots, histogram staging", 5: "per-pair histogram bound", 11: "chunk-level bound", 6: "pair / chunk lists",
         7: "pair-list pass (incl. its unions)", 8: "4x4-block pass (incl. its unions)"}
tot = sum(v[k] for k in names)
print(f"{shape} {kind}: tile pairs at the exact stage {v[0]}, chunk visits {v[1]}, candidate pairs {v[2]}, unions {v[3]}, "
      f"blocks on the pair-list path {v[9]}, on the 4x4 path {v[10]}")
for k, n in names.items():
    print(f"  {n:38s} {v[k] * 64 / 1e6:9.1f} Mcycles  {100.0 * v[k] / max(tot, 1):5.1f} %")
