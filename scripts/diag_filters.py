"""How selective are the tile-level and pair-level bounds of merge_components at config 2 / 4?"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from beyond_fixed_forms_amd import _lib
from beyond_fixed_forms_amd.config import Config
from beyond_fixed_forms_amd.projection import run_projection
from beyond_fixed_forms_amd.scene import prepare_scene
from beyond_fixed_forms_amd.synthetic import make_scene
dev = "cuda:0"
shape = sys.argv[1] if len(sys.argv) > 1 else "c2"
scene = make_scene(shape, seed=0, device=dev)
cfg = Config.with_defaults(width_2d=scene.width, height_2d=scene.height)
ds = prepare_scene(scene, cfg, device=dev)
res = run_projection(ds, cfg, debug_out=True)
rows = _lib.permute_bits(res.debug["raw_rows"], torch.argsort(ds.unsort).to(torch.int32), ds.n_points)
area, mean_word, cmask, hist, sig = _lib.row_stats(rows)
n = rows.shape[0]; nt = (n + 63) // 64
order = torch.argsort(sig, stable=True)
h = hist[order].float(); a = area[order].float()
pad = nt * 64 - n
hp = torch.cat([h, torch.zeros(pad, 64, device=dev)]); ap = torch.cat([a, torch.zeros(pad, device=dev)])
hmax = hp.view(nt, 64, 64).max(1).values                       # (nt, 64 bins)
apos = torch.where(ap > 0, ap, torch.full_like(ap, 1e9)).view(nt, 64).min(1).values
u = torch.minimum(hmax[:, None, :], hmax[None, :, :]).sum(-1)  # (nt, nt)
den = apos[:, None] + apos[None, :] - u
passq = (~(den > 0)) | (u / den > 0.2)
iu = torch.triu_indices(nt, nt)
print(shape, "tile pairs", iu.shape[1], "pass tile bound", int(passq[iu[0], iu[1]].sum()))
# signature-disjoint + light-mass bound
sg = sig[order]
heavy = torch.stack([((sg >> (62 - b)) & 1) for b in range(63)], 1).bool()      # (n, 63)
heavy = torch.cat([heavy, torch.zeros(n, 1, dtype=torch.bool, device=dev)], 1)
light = (h * (~heavy).float()).sum(1) / a.clamp(min=1)                            # light fraction per row
print("light fraction: mean %.3f  p90 %.3f  max %.3f" % (light.mean().item(), light.quantile(0.9).item(), light.max().item()))
hv = torch.cat([heavy, torch.zeros(pad, 64, dtype=torch.bool, device=dev)]).view(nt, 64, 64).any(1)   # tile heavy-bin union
lmax = torch.cat([light, torch.zeros(pad, device=dev)]).view(nt, 64).max(1).values
disjoint = ~(hv[:, None, :] & hv[None, :, :]).any(-1)
ok_light = (lmax[:, None] + lmax[None, :]) < (1.0 / 6.0) * 2 * 0.999           # crude: L_i + L_j <= (a_i + a_j)/6 if both fractions < 1/6
rej = disjoint & (lmax[:, None] < 1 / 6) & (lmax[None, :] < 1 / 6)
print("tile pairs rejected by signature-disjoint+light bound:", int(rej[iu[0], iu[1]].sum()))
print("tile pairs surviving both:", int((passq & ~rej)[iu[0], iu[1]].sum()))
# finer histogram: 256 bins -> how many candidate tile pairs?
