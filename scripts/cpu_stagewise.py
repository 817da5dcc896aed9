"""Stage-wise CPU timings of the reference path beside the GPU path, on this machine's host cores.

The oracle (CPU restatement, oracle/) gives stages (i) projection+votes, (ii) Gram+merge as the oracle does it
(integer label ids, frontier search), (iii) ratio-filter sweep, (iv) overlap+filters, (v) refinement.  The
reference's own formulation of (ii) -- a Python double loop over label strings (P:169-187) and the transitive
closure by Ins rounds of clamp(R@A + A) (P:250-274) -- is restated here, timed at small instance counts and
extrapolated (O(Ins^2) and O(Ins^4)), clearly labelled as such.

usage: python scripts/cpu_stagewise.py [c1|c2] [n_sample_views]
"""
import copy, functools, os, sys, time, warnings
print = functools.partial(print, flush=True)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from beyond_fixed_forms_amd.config import Config
from beyond_fixed_forms_amd.synthetic import make_scene, make_text_bank
from oracle.make_golden_shared import bank_encoder
from oracle.projection_ref import project_scene_ref
from oracle.refinement_ref import refine_class_ref

shape = sys.argv[1] if len(sys.argv) > 1 else "c2"
n_sample = int(sys.argv[2]) if len(sys.argv) > 2 else 100
THREADS = int(os.environ.get("BFF_CPU_THREADS", "16"))      # the GPU box gives one GPU's share of the host: 16 cores
torch.set_num_threads(THREADS)
print(f"host: {os.cpu_count()} logical cores; torch threads {torch.get_num_threads()}; model:",
      next((l.split(':')[1].strip() for l in open('/proc/cpuinfo') if l.startswith('model name')), '?'))
scene = make_scene(shape, seed=0, query="table", device="cuda" if torch.cuda.is_available() else "cpu")   # generator on the GPU: seconds instead of minutes
cfg = Config.with_defaults(width_2d=scene.width, height_2d=scene.height)
n_views = len(scene.mask_2d)
sub = copy.copy(scene)
sub.mask_2d = [dict(f) for f in scene.mask_2d[:n_sample]]
sub.color_files = [f for f in scene.color_files if int(f[:-4]) < n_sample * cfg.downsample_ratio]
bank, index = make_text_bank(768, seed=0)
enc = bank_encoder(bank.float(), index)
st = {}
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    res = project_scene_ref(sub, cfg, stage_times=st)
    t0 = time.perf_counter()
    try:
        refine_class_ref([(sub.scene_id, sub.stage1, res)], cfg, "table", enc)
    except IndexError:
        pass
    st["v_refinement"] = time.perf_counter() - t0
ins = sum(len(f["segmented_frame_masks"]) for f in sub.mask_2d)
ins_full = sum(len(f["segmented_frame_masks"]) for f in scene.mask_2d)
scale = n_views / max(1, len(sub.mask_2d))
print(f"\n{shape}: oracle on {len(sub.mask_2d)} of {n_views} mask views (Ins = {ins} of {ins_full}), N = {scene.points.shape[0]}")
print(f"{'stage':34s} {'sample s':>10s} {'scaled to the scene s':>24s}")
rows = [("i_projection_votes", scale, "linear in views"), ("ii_gram_merge", scale ** 2, "oracle formulation, ~Ins^2"),
        ("iii_ratio_filter_sweep", scale, "linear in frames"), ("iv_overlap_filters", 1.0, "K^2 N, K small"),
        ("v_refinement", 1.0, "S1 K N")]
tot = 0.0
for k, f, note in rows:
    v = st.get(k, 0.0)
    tot += v * f
    print(f"{k:34s} {v:10.3f} {v * f:24.3f}   ({note})")
print(f"{'sum (oracle formulation)':34s} {sum(st.values()):10.3f} {tot:24.3f}")

# the reference's own formulation of stage (ii), small sizes, extrapolated
def label_loop(labels):                       # P:169-187: Python double loop over strings
    n = len(labels)
    m = torch.zeros((n, n))
    for i in range(n):
        for j in range(n):
            if labels[i] == labels[j]:
                m[i, j] = 1
    return m

def closure(adj):                             # P:250-274: n rounds of clamp(R @ A + A, 0, 1)
    n = adj.shape[0]
    r = adj.clone()
    for _ in range(n):
        r = torch.clamp(r @ adj + adj, 0, 1)
    return r

print("\nreference formulation of stage (ii), measured small and extrapolated to Ins =", ins_full)
g = torch.Generator().manual_seed(0)
for n in (256, 512, 1024):
    labels = ["table"] * n
    t0 = time.perf_counter(); label_loop(labels); t1 = time.perf_counter()
    a = (torch.rand((n, n), generator=g) < 4.0 / n).float(); a = ((a + a.T) > 0).float()
    t2 = time.perf_counter(); closure(a); t3 = time.perf_counter()
    print(f"  Ins={n:5d}: label loop {t1 - t0:8.3f} s -> x(Ins/n)^2 = {(t1 - t0) * (ins_full / n) ** 2:12.1f} s;   "
          f"closure {t3 - t2:8.3f} s -> x(Ins/n)^4 = {(t3 - t2) * (ins_full / n) ** 4:14.1f} s")
