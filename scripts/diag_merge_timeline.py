"""Block timeline of the components' tile pass at PRODUCTION occupancy (BFF_MERGE_DIAG=2: start / end of every
block, nothing else): how many blocks run over time, how long they take by position in the list, where the tail is.
usage: BFF_MERGE_DIAG=2 python scripts/diag_merge_timeline.py [c2|c4] [default|many]"""
import os, sys
os.environ.setdefault("BFF_MERGE_DIAG", "2")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
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
rows = fr.rows
area, mw_, cmask, hist, sig = _lib.row_stats(rows, fr.cmask)
order = _lib.argsort_i64(sig, 30)
cap = 60000
for rep in range(2):
    d = torch.zeros(16 + 4 * cap, dtype=torch.int32, device=dev)
    d[15] = cap
    d[14] = 1
    _lib.merge_components(rows, area, ds.label_id, cfg.iou_thres, order, cmask, hist, diag=d)
tl = d[16:16 + 2 * cap].view(-1, 2).cpu().numpy().astype("int64") & 0xffffffff
info = d[16 + 2 * cap:16 + 4 * cap].view(-1, 2).cpu().numpy().astype("int64")
idx = np.flatnonzero(tl[:, 1] != 0)
info = info[idx]
n_pairs, shared, sparse_stages, dense_stages = info[:, 0] & 0xffff, info[:, 0] >> 16, info[:, 1] & 0xffff, info[:, 1] >> 16
tl = tl[idx]
t0 = tl[:, 0].min()
st, en = (tl[:, 0] - t0) * 0.01, (tl[:, 1] - t0) * 0.01
dur = en - st
span = en.max()
print(f"{shape} {kind}: {len(st)} blocks with work, kernel span {span:.0f} us, sum of block time {dur.sum() / 1e3:.1f} ms "
      f"(/ 768 slots = {dur.sum() / 768:.0f} us), mean {dur.mean():.1f} / p50 {np.median(dur):.1f} / p90 {np.percentile(dur, 90):.1f} / max {dur.max():.1f} us")
grid = np.linspace(0, span, 41)
print("running blocks over time:", [int(((st <= g) & (en > g)).sum()) for g in grid])
# by position in the list: when do blocks start, how long do they run
for lo in range(0, len(idx), max(1, len(idx) // 10)):
    sel = slice(lo, lo + max(1, len(idx) // 10))
    print(f"  list positions {idx[sel][0]:6d}..{idx[sel][-1]:6d}: start {st[sel].min():7.1f}..{st[sel].max():7.1f} us, "
          f"duration mean {dur[sel].mean():6.1f} max {dur[sel].max():6.1f} us")
late = np.argsort(-en)[:8]
print("last blocks to finish (list position, start, duration):", [(int(idx[i]), round(float(st[i]), 1), round(float(dur[i]), 1)) for i in late])

order_ = np.argsort(-dur)[:25]
print("longest blocks: (duration us, candidate pairs, shared chunks, sparse stages, dense stages)")
for i in order_:
    print(f"   {dur[i]:7.1f}  pairs {n_pairs[i]:5d}  chunks {shared[i]:4d}  sparse {sparse_stages[i]:3d}  dense {dense_stages[i]:3d}  start {st[i]:6.1f}")
reached = n_pairs > 0
print(f"blocks that reached the exact stage: {int(reached.sum())}; of all block time {dur[reached].sum() / dur.sum() * 100:.0f} % is theirs; "
      f"time per stage (their time / their stages): {dur[reached].sum() / max(1, (sparse_stages + dense_stages)[reached].sum()):.2f} us")
for name, sel in (("sparse", reached & (dense_stages == 0)), ("dense", dense_stages > 0)):
    if sel.any():
        print(f"  {name}: {int(sel.sum())} blocks, {dur[sel].sum() / 1e3:.1f} ms, stages {int((sparse_stages + dense_stages)[sel].sum())}, "
              f"{dur[sel].sum() / max(1, (sparse_stages + dense_stages)[sel].sum()):.2f} us per stage, mean shared chunks {shared[sel].mean():.0f}")
print(f"  blocks that stopped at the bounds: {int((~reached).sum())}, {dur[~reached].sum() / 1e3:.1f} ms, mean {dur[~reached].mean():.1f} us")
