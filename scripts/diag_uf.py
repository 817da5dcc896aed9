import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from beyond_fixed_forms_amd import _lib
from beyond_fixed_forms_amd.config import Config
from beyond_fixed_forms_amd.projection import run_projection, groups_from_labels
from beyond_fixed_forms_amd.scene import prepare_scene
from beyond_fixed_forms_amd.synthetic import make_scene
dev = "cuda:0"
scene = make_scene("c2", seed=0, device=dev)
cfg = Config.with_defaults(width_2d=scene.width, height_2d=scene.height)
ds = prepare_scene(scene, cfg, device=dev)
res = run_projection(ds, cfg, debug_out=True)
raw = res.debug["raw_rows"]                      # original point order
srt = _lib.permute_bits(raw, torch.argsort(ds.unsort).to(torch.int32), ds.n_points)
def parts(c):
    c = c.cpu().numpy(); 
    return groups_from_labels(c, np.ones(len(c), bool), 2)
for name, rows in (("sorted", srt), ("unsorted", raw)):
    area, mw, cmask, hist, sig = _lib.row_stats(rows)
    adj = _lib.merge_adjacency(rows, area, ds.label_id, 0.2)
    ref = parts(_lib.components(adj))
    print(name, "reference groups:", len(ref), [len(g) for g in ref][:12])
    for oname, order in (("sig", torch.argsort(sig, stable=True).to(torch.int32)), ("identity", torch.arange(rows.shape[0], dtype=torch.int32, device=dev))):
        for rep in range(3):
            got = parts(_lib.merge_components(rows, area, ds.label_id, 0.2, order, cmask, hist))
            print("   ", oname, rep, "equal" if got == ref else f"DIFF groups {len(got)} {[len(g) for g in got][:12]}")
