"""Back-half kernels of config 2 in isolation (after one front): or_reduce dense vs through chunk masks, etc."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from beyond_fixed_forms_amd import _lib
from beyond_fixed_forms_amd.config import Config
from beyond_fixed_forms_amd.projection import projection_front
from beyond_fixed_forms_amd.scene import prepare_scene
from beyond_fixed_forms_amd.synthetic import make_scene
_lib.load()
dev = "cuda"
scene = make_scene(sys.argv[1] if len(sys.argv) > 1 else "c2", seed=0, device=dev, query="table")
cfg = Config.with_defaults(width_2d=scene.width, height_2d=scene.height)
ds = prepare_scene(scene, cfg, device=dev)
fr = projection_front(ds, cfg)
comp_h, area_h = _lib.fetch(fr.comp, fr.area)
offs, members, sizes, n_void = _lib.host_component_csr(comp_h, area_h > 0, cfg.min_aggragated_masks)
print("groups", len(sizes), "members", len(members), "largest", int(sizes.max()))
offs_d, members_d = torch.from_numpy(offs).to(dev), torch.from_numpy(members).to(dev)
def timeit(name, f, reps=30):
    for _ in range(3): f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    print(f"{name:34s} {a.elapsed_time(b) / reps * 1e3:8.1f} us")
d = _lib.or_reduce_groups(fr.rows, offs_d, members_d, int(sizes.max()))
timeit("or_reduce", lambda: _lib.or_reduce_groups(fr.rows, offs_d, members_d, int(sizes.max())))
timeit("group_conf_mean", lambda: _lib.group_conf_mean(ds.conf, offs_d, members_d))
timeit("or_reduce + conf means, one launch", lambda: _lib.or_reduce_groups(fr.rows, offs_d, members_d, int(sizes.max()), ds.conf))
agg = d.clone()
sizes_d = torch.from_numpy(sizes).to(dev)
timeit("popcount_rows(agg)", lambda: _lib.popcount_rows(agg))
timeit("cross_popcount(agg, agg)", lambda: _lib.cross_popcount(agg, agg))
timeit("resolve_overlaps, ordered replay (3 launches)", lambda: _lib.resolve_overlaps_replay(agg.clone(), sizes_d))
timeit("resolve_overlaps_filtered (1 launch)", lambda: _lib.resolve_overlaps_filtered(agg.clone(), sizes_d, fr.keep))
timeit("  (clone alone)", lambda: agg.clone())
timeit("and_rows", lambda: _lib.and_rows(agg, fr.keep))
