"""Host-inclusive ingestion at config 2: time of ingest.prepare_scene_fast per scene (one thread, with a cProfile) and
the pipeline rate for 1..4 loader threads."""
import cProfile, copy, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from beyond_fixed_forms_amd import _lib, ingest
from beyond_fixed_forms_amd.config import Config
from beyond_fixed_forms_amd.refinement import TextSimilarity
from beyond_fixed_forms_amd.synthetic import make_scene, make_text_bank
_lib.load()
dev = "cuda:0"
scenes = []
for k in range(2):
    sc = make_scene("c2", seed=k, device=dev, query="table")
    sc.depths_raw = {f: np.ascontiguousarray(np.round(d[::2, ::2].astype(np.float64) * 1000.0).astype(np.uint16)) for f, d in sc.depths.items()}
    scenes.append(sc)
cfg = Config.with_defaults(width_2d=scenes[0].width, height_2d=scenes[0].height)
st = ingest.Staging()
for _ in range(3):
    ds = ingest.prepare_scene_fast(scenes[0], cfg, dev, staging=st)
torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(6):
    ds = ingest.prepare_scene_fast(scenes[k % 2], cfg, dev, staging=st)
torch.cuda.synchronize()
print(f"prepare_scene_fast, one thread, raw 16-bit depth: {(time.perf_counter() - t0) / 6 * 1e3:.2f} ms per scene")
pr = cProfile.Profile(); pr.enable()
for k in range(4):
    ds = ingest.prepare_scene_fast(scenes[k % 2], cfg, dev, staging=st)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
bank, index = make_text_bank(768, seed=0)
sim = TextSimilarity(lambda t: bank[index[t.replace(" ", "_")]][None, :], dev)
for nl in (1, 2, 3, 4):
    for nt in (2, 4):
        r = ingest.bench_host_inclusive(scenes, cfg, dev, "table", sim, steps=30, n_loaders=nl, native_threads=nt)
        print(f"loaders {nl} native threads {nt}: {r['value']:.1f} scenes/s ({r['ms_per_scene']:.2f} ms per scene)")
