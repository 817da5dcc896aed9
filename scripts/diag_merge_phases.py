"""Where merge_components spends its block time at config 2 / 4: cycle counters per phase (diag mode)."""
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
fr = projection_front(ds, cfg, debug_out=True)
rows = fr.rows
area, mw_, cmask, hist, sig = _lib.row_stats(rows)
order = _lib.argsort_i64(sig, 30)
names = ["tiles", "chunk visits", "candidate pairs", "unions", "roots+hist staging", "pair bounds", "lists", "pair-list pass",
         "dense pass", "tiles via pair list", "tiles via dense pass"]
for label, parent in (("from scratch", None), ("from the converged forest", "conv")):
    if parent == "conv":
        parent = _lib.merge_components(rows, area, ds.label_id, cfg.iou_thres, order, cmask, hist).clone()
    cap = 40000
    d = torch.zeros(16 + 2 * cap, dtype=torch.int32, device=dev)
    d[15] = cap
    _lib.merge_components(rows, area, ds.label_id, cfg.iou_thres, order, cmask, hist, diag=d, parent=parent)
    v = d[:16].tolist()
    tl = d[16:16 + 2 * cap].view(-1, 2).cpu().numpy().astype("int64") & 0xffffffff
    tl = tl[(tl[:, 1] != 0)]
    t0 = tl[:, 0].min()
    st, en = (tl[:, 0] - t0) * 0.01, (tl[:, 1] - t0) * 0.01                        # us (100 MHz ticks)
    import numpy as np
    span = en.max()
    grid = np.linspace(0, span, 41)
    conc = [(int(((st <= g) & (en > g)).sum())) for g in grid]
    dur = en - st
    print(f"  blocks with work {len(st)}, kernel span {span:.0f} us, block duration mean {dur.mean():.1f} / p50 {np.median(dur):.1f} / p90 {np.percentile(dur, 90):.1f} / max {dur.max():.1f} us")
    print("  running blocks over time (40 samples):", conc)
    print(f"\n{shape} {label}:")
    for k in range(4):
        print(f"  {names[k]:24s} {v[k]}")
    for k in (9, 10):
        print(f"  {names[k]:24s} {v[k]}")
    tot = sum(v[4:9])
    for k in range(4, 9):
        cyc = v[k] * 64
        print(f"  {names[k]:24s} {cyc / 2.4e3 / 1e3:9.1f} ms of block time  ({100.0 * v[k] / max(1, tot):5.1f} %)   -> / 768 resident blocks = {cyc / 2.4e3 / 768:7.1f} us")
