"""How early could the tile pass reject a near miss?  For the candidate pairs of one scene (same label, histogram bound
passes the IoU test) this replays the exact count bin by bin (64 bins of the sorted cloud, the order the LDS stages run
in) and reports where `IoU(counted so far + histogram bound of the bins still to come)` falls to the threshold -- a
test the kernel does not make today: it counts a non-edge to its last shared chunk.
usage: python scripts/diag_merge_bounds.py [c2] [many]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from beyond_fixed_forms_amd import _lib
from beyond_fixed_forms_amd.config import Config
from beyond_fixed_forms_amd.projection import DEPTH_THRESH
from beyond_fixed_forms_amd.scene import prepare_scene
from beyond_fixed_forms_amd.synthetic import make_scene

_lib.load()
shape = sys.argv[1] if len(sys.argv) > 1 else "c2"
many = "many" in sys.argv
dev = "cuda"
scene = make_scene(shape, seed=0, device=dev, query="table", cut_masks=not many)
cfg = Config.with_defaults(width_2d=scene.width, height_2d=scene.height)
ds = prepare_scene(scene, cfg, device=dev)
n, nw, hw = ds.n_points, ds.nw, ds.height * ds.width
n_mviews = ds.view_mask_offs.shape[0] - 1
wdt = torch.int32 if ds.word_bits == 32 else torch.int64
maskbits = torch.empty((n_mviews, hw), device=dev, dtype=wdt)
labels = torch.empty((n_mviews, _lib.label_plane_stride(hw)), device=dev, dtype=torch.uint8)
segmap = torch.empty((n_mviews, 2 * _lib.segmap_words(hw)), dtype=torch.int32, device=dev)
_lib.rle_to_labels(ds.run_start, ds.run_end, ds.mask_run_offs, ds.view_mask_offs, n_mviews, hw, ds.word_bits, labels, maskbits, segmap)
rows = torch.zeros((ds.n_rows, nw), dtype=torch.int64, device=dev)
masked = torch.zeros(n, dtype=torch.int32, device=dev)
viewed = torch.zeros(n, dtype=torch.int32, device=dev)
cm = _lib.chunk_mask_buffer(ds.n_rows, nw, dev).zero_()
_lib.project_views(ds.xyz, n, ds.inv_pose, ds.cam_intr, ds.sweep_depth, ds.depth_index, ds.height, ds.width, DEPTH_THRESH, maskbits,
                   ds.word_bits, ds.frame_mask, ds.frame_rowbase, ds.frame_nmask, ds.frame_flags, rows, masked, viewed, segmap, cm,
                   ds.tile_bounds, labels=labels, depth_size=ds.depth_size)
torch.cuda.synchronize()
R = ds.n_rows
lut = torch.tensor([bin(i).count("1") for i in range(256)], dtype=torch.int16, device=dev)


def popc_words(x):                       # int64 [..., W] -> int32 [..., W]
    return lut[x.contiguous().view(torch.uint8).long()].view(*x.shape, 8).sum(-1, dtype=torch.int32)


bw = (nw + 63) // 64                     # words per bin
pad = 64 * bw - nw
pc = torch.cat([popc_words(rows[i:i + 512]) for i in range(0, R, 512)])
hist = torch.nn.functional.pad(pc, (0, pad)).view(R, 64, bw).sum(-1).float()        # [R, 64]
area = hist.sum(1)
thr = float(cfg.iou_thres)
print(f"{R} rows, {int((area > 0).sum())} non-empty; IoU threshold {thr}")
# candidate pairs (i < j): histogram bound passes the test
cand = []
for i0 in range(0, R, 256):
    ub = torch.minimum(hist[i0:i0 + 256, None, :], hist[None, :, :]).sum(-1)          # [256, R]
    ub = torch.minimum(ub, torch.minimum(area[i0:i0 + 256, None], area[None, :]))
    iou = ub / (area[i0:i0 + 256, None] + area[None, :] - ub)
    ii, jj = torch.nonzero(iou > thr, as_tuple=True)
    keep = ii + i0 < jj
    cand.append(torch.stack([ii[keep] + i0, jj[keep]], 1))
cand = torch.cat(cand)
print(f"candidate pairs after the histogram bound: {cand.shape[0]}")
g = torch.Generator(device="cpu").manual_seed(0)
sel = cand[torch.randperm(cand.shape[0], generator=g)[:60000].to(dev)]
rows_p = torch.nn.functional.pad(rows, (0, pad))
per_bin_I, per_bin_ub = [], []
for k in range(0, sel.shape[0], 2000):
    a, b = sel[k:k + 2000, 0], sel[k:k + 2000, 1]
    per_bin_I.append(popc_words(rows_p[a] & rows_p[b]).view(-1, 64, bw).sum(-1).float())
    per_bin_ub.append(torch.minimum(hist[a], hist[b]))
I = torch.cat(per_bin_I)
UB = torch.cat(per_bin_ub)
ai, aj = area[sel[:, 0]], area[sel[:, 1]]
tot = I.sum(1)
edge = tot / (ai + aj - tot) > thr
print(f"sample {sel.shape[0]}: edges {int(edge.sum())}, non-edges {int((~edge).sum())}")
# work of a pair ~ bins in which both rows have points (chunks both occupy); replay in bin order
both = (hist[sel[:, 0]] > 0) & (hist[sel[:, 1]] > 0)
work_total = both.sum(1).float()
cumI = I.cumsum(1)
suffix = UB.flip(1).cumsum(1).flip(1) - UB                   # bound of the bins after b
best = cumI + suffix
iou_best = best / (ai[:, None] + aj[:, None] - best)
rejected_after = (iou_best <= thr)                            # [pairs, 64]: could stop after bin b
first = torch.where(rejected_after.any(1), rejected_after.float().argmax(1), torch.full_like(tot, 63).long())
cumwork = both.float().cumsum(1)
work_done = cumwork.gather(1, first[:, None]).squeeze(1)
ne = ~edge
print("non-edges: share of their shared bins still counted with the suffix-bound test: %.2f (mean), %.2f (median)" %
      ((work_done[ne] / work_total[ne].clamp(min=1)).mean().item(), (work_done[ne] / work_total[ne].clamp(min=1)).median().item()))
# edges settle early today when IoU(partial) > thr
iou_part = cumI / (ai[:, None] + aj[:, None] - cumI)
e_first = torch.where((iou_part > thr).any(1), (iou_part > thr).float().argmax(1), torch.full_like(tot, 63).long())
e_done = cumwork.gather(1, e_first[:, None]).squeeze(1)
print("edges: share counted before IoU(partial) passes: %.2f (mean)" % (e_done[edge] / work_total[edge].clamp(min=1)).mean().item())
w_ne, w_e = work_total[ne].sum().item(), work_total[edge].sum().item()
print("pair-bin work: non-edges %.0f (%.0f%%), edges %.0f; with the test: non-edges %.0f" % (w_ne, 100 * w_ne / (w_ne + w_e), w_e, work_done[ne].sum().item()))
