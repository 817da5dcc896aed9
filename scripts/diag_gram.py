"""Diagnostic: how sparse is the Gram at config 2?  (tile pairs surviving the bound, chunks per tile pair)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from beyond_fixed_forms_amd import _lib
from beyond_fixed_forms_amd.config import Config
from beyond_fixed_forms_amd.projection import run_projection
from beyond_fixed_forms_amd.scene import prepare_scene
from beyond_fixed_forms_amd.synthetic import make_scene
dev = "cuda:0"
scene = make_scene(sys.argv[1] if len(sys.argv) > 1 else "c2", seed=0, device=dev)
cfg = Config.with_defaults(width_2d=scene.width, height_2d=scene.height)
ds = prepare_scene(scene, cfg, device=dev)
res = run_projection(ds, cfg, debug_out=True)
# rows in sorted point order are needed: redo the sweep pieces by hand
rows_sorted = _lib.permute_bits(res.debug["raw_rows"], torch.argsort(ds.unsort).to(torch.int32), ds.n_points)
area, mean_word, cmask, hist = _lib.row_stats(rows_sorted)
n = rows_sorted.shape[0]
order = torch.argsort((ds.label_id.long() << 32) | mean_word.long())
nt = (n + 63) // 64
bits = torch.from_numpy(np.unpackbits(cmask.cpu().numpy().view(np.uint8), axis=-1, bitorder="little")).to(dev).float()
bits = bits[order]
pad = nt * 64 - n
bits = torch.cat([bits, torch.zeros(pad, bits.shape[1], device=dev)])
tile = (bits.view(nt, 64, -1).sum(1) > 0).float()
print("chunks per row: mean %.1f  per tile-union: mean %.1f of %d" % (bits[:n].sum(1).mean().item(), tile.sum(1).mean().item(), (ds.nw + 7) // 8))
tp = tile @ tile.T
iu = torch.triu_indices(nt, nt)
print("tile pairs %d, mean shared chunks %.1f, total chunk visits %.0f" % (iu.shape[1], tp[iu[0], iu[1]].mean().item(), tp[iu[0], iu[1]].sum().item()))
h = hist[order].float()
a = area[order].float()
cand_tile = torch.zeros(nt, nt, dtype=torch.bool, device=dev)
ncand = 0
for bi in range(nt):
    hi_ = h[bi * 64:(bi + 1) * 64]
    ub = torch.minimum(hi_[:, None, :], h[None, :, :]).sum(-1)          # (64, n)
    ai = a[bi * 64:(bi + 1) * 64]
    fi = torch.minimum(ub, torch.minimum(ai[:, None], a[None, :]))
    iou = fi / (ai[:, None] + a[None, :] - fi)
    c = iou > 0.2
    ncand += int(c.sum())
    ct = torch.nn.functional.pad(c, (0, nt * 64 - n)).view(c.shape[0], nt, 64).any(2).any(0)
    cand_tile[bi] = ct
ct = cand_tile[iu[0], iu[1]]
print("candidate pairs %d of %d; candidate tile pairs %d of %d; chunk visits in candidates %.0f" % (ncand, n * n, int(ct.sum()), iu.shape[1], tp[iu[0], iu[1]][ct].sum().item()))
rows_u = res.debug["raw_rows"]
print("areas: mean %.0f max %d; groups %s" % (area.float().mean().item(), int(area.max()), [len(g) for g in res.groups]))

# ---- alternative row orderings: effect on tile unions and candidate chunk visits
def evaluate(name, key):
    order = torch.argsort(key)
    b = torch.from_numpy(np.unpackbits(cmask.cpu().numpy().view(np.uint8), axis=-1, bitorder="little")).to(dev).float()[order]
    b = torch.cat([b, torch.zeros(pad, b.shape[1], device=dev)])
    tile = (b.view(nt, 64, -1).sum(1) > 0).float()
    tp = tile @ tile.T
    h = hist[order].float(); a = area[order].float()
    cand_tile = torch.zeros(nt, nt, dtype=torch.bool, device=dev)
    for bi in range(nt):
        hi_ = h[bi * 64:(bi + 1) * 64]
        ub = torch.minimum(hi_[:, None, :], h[None, :, :]).sum(-1)
        ai = a[bi * 64:(bi + 1) * 64]
        fi = torch.minimum(ub, torch.minimum(ai[:, None], a[None, :]))
        c = (fi / (ai[:, None] + a[None, :] - fi)) > 0.2
        cand_tile[bi] = torch.nn.functional.pad(c, (0, nt * 64 - n)).view(c.shape[0], nt, 64).any(2).any(0)
    ct = cand_tile[iu[0], iu[1]]
    print("%-28s tile-union %.1f  cand tile pairs %d  cand chunk visits %.0f" % (name, tile.sum(1).mean().item(), int(ct.sum()), tp[iu[0], iu[1]][ct].sum().item()))

hf = hist.float()
cum = hf.cumsum(1)
median_bin = (cum < (area.float()[:, None] / 2)).sum(1).clamp(max=63)
argmax_bin = hf.argmax(1)
mw = mean_word.long().clamp(max=1 << 20)
evaluate("mean_word (current)", mw)
evaluate("median_bin, mean_word", median_bin.long() * (1 << 21) + mw)
evaluate("argmax_bin, mean_word", argmax_bin.long() * (1 << 21) + mw)
# mean position restricted to the median bin +-1 (robust centre)
w = torch.arange(64, device=dev).float()[None, :]
near = ((w - median_bin[:, None].float()).abs() <= 1).float()
centre = (hf * near * w).sum(1) / (hf * near).sum(1).clamp(min=1)
evaluate("robust centre (bin units)", (centre * 1000).long())
# two-level: argmax bin, then second-largest bin
top2 = hf.topk(2, dim=1).indices
evaluate("argmax_bin, 2nd bin", top2[:, 0] * 64 * (1 << 10) + top2[:, 1] * (1 << 10) + (mw >> 4))
# signature of heavy bins (bins holding >= 15% of the area) as a bitmask key
heavy = (hf >= 0.15 * area.float()[:, None])
sig = (heavy.long() << torch.arange(63, -1, -1, device=dev)[None, :]).sum(1)
evaluate("heavy-bin signature", sig)
