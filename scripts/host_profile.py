"""cProfile of the host thread over N pipelined config-2 steps (where does the host's time per scene go?)."""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from beyond_fixed_forms_amd import _lib
from beyond_fixed_forms_amd import distributed as bdist
from beyond_fixed_forms_amd.config import Config
from beyond_fixed_forms_amd.projection import projection_back, projection_front
from beyond_fixed_forms_amd.refinement import TextSimilarity, prepare_stage1, refine_class
from beyond_fixed_forms_amd.scene import prepare_scene
from beyond_fixed_forms_amd.synthetic import make_scene, make_text_bank
import bench
_lib.load()
dev = "cuda:0"
shape = sys.argv[1] if len(sys.argv) > 1 else "c2"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
scene = make_scene(shape, seed=0, device=dev, query="chair")
cfg = Config.with_defaults(width_2d=scene.width, height_2d=scene.height)
ds = prepare_scene(scene, cfg, device=dev)
stage1 = prepare_stage1(scene.stage1, dev)
bank, index = make_text_bank(768, seed=0)
sim = TextSimilarity(bench.bank_encoder(bank.float(), index), dev)
streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
def front(i):
    with torch.cuda.stream(streams[i % 2]):
        return projection_front(ds, cfg)
def back(i, fr):
    with torch.cuda.stream(streams[i % 2]):
        res = projection_back(fr)
        fin = refine_class([(scene.scene_id, stage1, res)], cfg, "chair", sim, dev)
        rows = fin[scene.scene_id].rows
        return bdist.gather_final_rows(rows if rows is not None else torch.zeros((0, ds.nw), dtype=torch.int64, device=dev))
def run(k):
    nxt = front(0)
    for i in range(k):
        cur = nxt
        if i + 1 < k:
            nxt = front(i + 1)
        back(i, cur)
    torch.cuda.synchronize()
run(5)
pr = cProfile.Profile()
pr.enable(); run(steps); pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(45)
st.sort_stats("cumtime").print_stats(45)
