"""cProfile of the host side of one bench step (run on the GPU box): where the non-kernel time goes."""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from beyond_fixed_forms_amd.config import Config  # noqa: E402
from beyond_fixed_forms_amd.projection import run_projection  # noqa: E402
from beyond_fixed_forms_amd.refinement import TextSimilarity, prepare_stage1, refine_class  # noqa: E402
from beyond_fixed_forms_amd.scene import prepare_scene  # noqa: E402
from beyond_fixed_forms_amd.synthetic import make_scene, make_text_bank  # noqa: E402

shape = sys.argv[1] if len(sys.argv) > 1 else "c2"
dev = "cuda:0"
scene = make_scene(shape, seed=0, device=dev)
cfg = Config.with_defaults(width_2d=scene.width, height_2d=scene.height)
ds = prepare_scene(scene, cfg, device=dev)
bank, index = make_text_bank(768, seed=0)
sim = TextSimilarity(bench.bank_encoder(bank.float(), index), dev)
stage1 = prepare_stage1(scene.stage1, dev)


def step():
    res = run_projection(ds, cfg)
    return refine_class([(scene.scene_id, stage1, res)], cfg, "table", sim, dev)


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    r = run_projection(ds, cfg)
torch.cuda.synchronize()
t1 = time.perf_counter()
for _ in range(10):
    refine_class([(scene.scene_id, stage1, r)], cfg, "table", sim, dev)
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"projection {1e2 * (t1 - t0):.2f} ms/step, refinement {1e2 * (t2 - t1):.2f} ms/step")
ph = {}
for _ in range(10):
    run_projection(ds, cfg, phases=ph)
print('phases (ms/step, serialised):', {k: round(100 * v, 3) for k, v in ph.items()})
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    step()
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
