"""GPU parity, whole path: project_scene / refine_class against the golden scene fixtures (results
of the reference's helpers) and against the oracle on fresh seeds and edge configurations."""
import os
import warnings

import numpy as np
import pytest
import torch

import golden_io as gio
from oracle import projection_ref as pref, refinement_ref as rref
from oracle.make_golden_shared import bank_encoder

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
Z = lambda name: np.load(os.path.join(gio.GOLDEN_DIR, name))


@pytest.fixture(scope="module")
def api():
    from beyond_fixed_forms_amd import _lib, projection, refinement
    _lib.load()
    return projection, refinement


def cfg_for(scene, **over):
    from beyond_fixed_forms_amd.config import Config
    return Config.with_defaults(width_2d=scene.width, height_2d=scene.height, **over)


def same(got: dict, exp: dict):
    """Bit-identical masks, identical conf values+dtype, identical labels, identical empty forms."""
    if isinstance(exp["ins"], list):
        assert isinstance(got["ins"], list) and got["ins"] == [] and got["conf"] == [] and got["final_class"] == []
        return
    assert got["ins"].dtype == exp["ins"].dtype and tuple(got["ins"].shape) == tuple(exp["ins"].shape)
    assert torch.equal(got["ins"].cpu(), exp["ins"])
    assert got["conf"].dtype == exp["conf"].dtype and torch.equal(got["conf"].cpu(), exp["conf"])
    assert list(got["final_class"]) == list(exp["final_class"])


def check_golden(got: dict, z, prefix, n):
    g = gio.result_to_arrays({"ins": got["ins"] if isinstance(got["ins"], list) else got["ins"].cpu(),
                              "conf": got["conf"] if isinstance(got["conf"], list) else got["conf"].cpu(),
                              "final_class": got["final_class"]}, n)
    for k in ("kind", "conf_dtype"):
        assert str(g[k]) == str(z[f"{prefix}.{k}"]), (prefix, k)
    assert np.array_equal(g["ins_packed"], z[f"{prefix}.ins_packed"])
    assert np.array_equal(g["conf"], z[f"{prefix}.conf"])
    assert list(g["final_class"]) == list(z[f"{prefix}.final_class"])


@pytest.mark.parametrize("name", ["scene_tiny_seed0", "scene_tiny_seed1", "scene_tiny_seed2", "scene_c1_seed0"])
def test_golden_scene(api, name):
    """BASELINE config 1 (20k points, 10 views @640x480, 5 masks/view) and three small scenes: the HIP
    path reproduces what the reference's helpers produced, stage 2 and final."""
    projection, refinement = api
    from beyond_fixed_forms_amd.synthetic import make_text_bank
    z = Z(name + ".npz")
    scene = gio.scene_from_arrays(z)
    cfg = cfg_for(scene)
    n = scene.points.shape[0]
    res = projection.project_scene(scene, cfg, DEV, return_result=True, debug_out=True)
    assert res.groups == gio.loads_groups(z["dbg.groups"])
    assert np.array_equal(np.array(np.float32(res.debug["thr"])).view(np.uint32), z["dbg.thr_bits"])
    assert np.array_equal(res.debug["masked_counts_raw"].cpu().numpy(), z["dbg.masked_counts_raw"])
    assert np.array_equal(res.debug["viewed_counts"].cpu().numpy(), z["dbg.viewed_counts"])
    check_golden(res.to_dict(), z, "stage2", n)
    bank, index = make_text_bank(int(z["bank_dim"]), seed=int(z["bank_seed"]))
    sim = refinement.TextSimilarity(bank_encoder(bank.float(), index), DEV)
    for stage2 in (res, {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in res.to_dict().items()}):
        fin = refinement.refine_class([(scene.scene_id, scene.stage1, stage2)], cfg, "table", sim, DEV)
        check_golden(fin[scene.scene_id].to_dict(), z, "final", n)
    # the PRODUCTION path against the same fixtures, directly: native ingestion (prepare_scene_fast) + the one-call
    # device path (bff_scene_project) with the refinement's first pass riding along, as the CLI and bench.py run it
    prod = projection.project_scene(scene, cfg, DEV, return_result=True, debug_out=False)
    assert prod.debug["path"] == "fast"
    assert prod.groups == gio.loads_groups(z["dbg.groups"])
    assert np.array_equal(np.array(np.float32(prod.debug["thr"])).view(np.uint32), z["dbg.thr_bits"])
    check_golden(prod.to_dict(), z, "stage2", n)
    from beyond_fixed_forms_amd.ingest import prepare_scene_fast
    st1 = refinement.prepare_stage1(scene.stage1, DEV)
    ds = prepare_scene_fast(scene, cfg, DEV)
    prod2 = projection.projection_back(projection.projection_front(ds, cfg, stage1=st1))
    assert prod2.debug["path"] == "fast" and prod2.prefetch is not None
    check_golden(prod2.to_dict(), z, "stage2", n)
    fin = refinement.refine_class([(scene.scene_id, st1, prod2)], cfg, "table", sim, DEV)
    check_golden(fin[scene.scene_id].to_dict(), z, "final", n)


def test_golden_class_three_scenes(api):
    """Cross-scene similarity threshold (R:316-324) with one empty stage-2 scene."""
    _, refinement = api
    from beyond_fixed_forms_amd.synthetic import make_text_bank
    z = Z("class_tiny_3scenes.npz")
    trip, ns = [], []
    for i in range(3):
        n = int(z[f"s{i}.n"])
        s1 = {"ins": gio.unpack_rles(z[f"s{i}.s1_len"], z[f"s{i}.s1_counts"], z[f"s{i}.s1_offs"]),
              "conf": torch.from_numpy(z[f"s{i}.s1_conf"].copy()), "final_class": [int(c) for c in z[f"s{i}.s1_class"]]}
        if str(z[f"s{i}.stage2.kind"]) == "rows":
            conf = torch.from_numpy(z[f"s{i}.stage2.conf"].copy())
            conf = conf.half() if str(z[f"s{i}.stage2.conf_dtype"]) == "torch.float16" else conf
            s2 = {"ins": torch.from_numpy(gio.unpack_bool_rows(z[f"s{i}.stage2.ins_packed"], n)), "conf": conf,
                  "final_class": [str(c) for c in z[f"s{i}.stage2.final_class"]]}
        else:
            s2 = pref.empty_result()
        trip.append((str(z[f"s{i}.scene_id"]), s1, s2)); ns.append(n)
    bank, index = make_text_bank(64, seed=9)
    sim = refinement.TextSimilarity(bank_encoder(bank.float(), index), DEV)
    cfg = cfg_for(type("S", (), {"width": 160, "height": 120}))
    fin, dbg = refinement.refine_class(trip, cfg, "table", sim, DEV, return_debug=True)
    assert abs(dbg["sim_thres"] - float(z["sim_thres"])) <= 1e-4
    for i, (sid, _, _) in enumerate(trip):
        check_golden(fin[sid].to_dict(), z, f"s{i}.final", ns[i])


CASES = {
    "seed20": dict(shape="tiny", seed=20),
    "two_labels_f32conf": dict(shape="tiny", seed=21, n_labels=2, conf_dtype=torch.float32),
    "m40_u64_words": dict(shape="tiny", seed=22, n_masks=40),
    "m70_chunked": dict(shape="tiny", seed=23, n_masks=70, n_views=4),
    "ragged_n": dict(shape="tiny", seed=24, n_points=4001),
    "small_n": dict(shape="tiny", seed=25, n_points=130, n_stage1=12),
    "odd_image": dict(shape="tiny", seed=26, height=97, width=131),
    "c1_seed5": dict(shape="c1", seed=5),
}


@pytest.mark.parametrize("case", list(CASES))
@pytest.mark.parametrize("mode", ["ratio", "occurrence", "nofilter"])
def test_scene_vs_oracle(api, case, mode):
    """Fresh seeds and edge shapes (M > 32 -> 64-bit mask words, M > 64 -> chunked frames, N not a
    multiple of 64, tiny N, odd image sizes, both filter branches): stage-2 and final results
    bit-identical to the oracle."""
    projection, refinement = api
    from beyond_fixed_forms_amd.synthetic import make_scene, make_text_bank
    if mode != "ratio" and case not in ("seed20", "m40_u64_words"):
        pytest.skip("filter branches are covered on two scenes")
    scene = make_scene(**CASES[case])
    over = {"ratio": {}, "occurrence": dict(if_occurance_threshold=True),
            "nofilter": dict(if_occurance_threshold=False, if_detected_ratio_threshold=False)}[mode]
    cfg = cfg_for(scene, **over)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        exp, dbg = pref.project_scene_ref(scene, cfg, return_debug=True)
    res = projection.project_scene(scene, cfg, DEV, return_result=True, debug_out=True)
    n = scene.points.shape[0]
    raw = np.unpackbits(res.debug["raw_rows"].cpu().numpy().view(np.uint8), axis=-1, bitorder="little")[:, :n].astype(bool)
    assert np.array_equal(raw, dbg["raw_ins"].numpy())
    assert np.array_equal(res.debug["masked_counts_raw"].cpu().numpy(), dbg["masked_counts_raw"].numpy().astype(np.int32))
    assert res.groups == dbg.get("groups", [])
    same(res.to_dict(), exp)
    bank, index = make_text_bank(64, seed=CASES[case]["seed"])
    enc = bank_encoder(bank.float(), index)
    sim = refinement.TextSimilarity(enc, DEV)
    if len(exp["conf"]) == 0:
        # no stage-2 instance survives -> no similarities -> the reference indexes an empty list (R:324)
        with pytest.raises(IndexError):
            rref.refine_class_ref([(scene.scene_id, scene.stage1, exp)], cfg, "table", enc)
        with pytest.raises(IndexError):
            refinement.refine_class([(scene.scene_id, scene.stage1, res)], cfg, "table", sim, DEV)
        return
    fin = refinement.refine_class([(scene.scene_id, scene.stage1, res)], cfg, "table", sim, DEV)
    fexp = rref.refine_class_ref([(scene.scene_id, scene.stage1, exp)], cfg, "table", enc)
    same(fin[scene.scene_id].to_dict(), fexp[scene.scene_id])
    # the pipelined form: the refinement's first device pass rides on the projection's last fetch
    from beyond_fixed_forms_amd.scene import prepare_scene
    st1 = refinement.prepare_stage1(scene.stage1, DEV)
    res2 = projection.run_projection(prepare_scene(scene, cfg, device=DEV), cfg, stage1=st1)
    assert res2.prefetch is not None and torch.equal(res2.rows, res.rows) and torch.equal(res2.conf, res.conf)
    fin2 = refinement.refine_class([(scene.scene_id, st1, res2)], cfg, "table", sim, DEV)
    assert res2.prefetch is None                                        # consumed
    same(fin2[scene.scene_id].to_dict(), fexp[scene.scene_id])


@pytest.mark.parametrize("variant", ["dup_stage1_and_shared_match", "nothing_merges", "only_shared_match"])
def test_refinement_merge_branches(api, variant):
    """Crafted stage-1 / stage-2 masks drive both order-dependent branches of R:211-281: matched stage-1 masks
    that overlap (IoU > 0.1) are OR-ed in place, stage-2 masks matched to the same stage-1 mask are merged
    (any / mean).  Final results identical to the oracle, also when nothing merges (no recomputation)."""
    _, refinement = api
    from beyond_fixed_forms_amd.labels import SCANNET200_LABELS
    from beyond_fixed_forms_amd.synthetic import make_text_bank
    from oracle.rle_ref import rle_encode_batch_ref
    n = 1000
    def m(*ranges):
        r = np.zeros(n, bool)
        for a, b in ranges:
            r[a:b] = True
        return r
    s1 = [m((0, 100)), m((10, 110)), m((200, 300)), m((400, 450)), m((700, 720)), m((440, 500))]
    if variant == "dup_stage1_and_shared_match":
        s2 = [m((0, 90)), m((15, 110)), m((200, 290)), m((210, 300)), m((600, 650)), m((405, 452))]
    elif variant == "only_shared_match":
        s2 = [m((200, 290)), m((210, 300)), m((405, 440))]
    else:
        s2 = [m((0, 90)), m((205, 300)), m((700, 715))]
    q = SCANNET200_LABELS.index("table")
    stage1 = {"ins": rle_encode_batch_ref(torch.from_numpy(np.stack(s1))), "conf": torch.ones(len(s1)),
              "final_class": [q, 3, q, 7, q, q]}
    conf = torch.tensor([0.31, 0.22, 0.43, 0.27, 0.39, 0.25][:len(s2)], dtype=torch.float16)
    stage2 = {"ins": torch.from_numpy(np.stack(s2)), "conf": conf, "final_class": ["table"] * len(s2)}
    bank, index = make_text_bank(64, seed=2)
    enc = bank_encoder(bank.float(), index)
    cfg = cfg_for(type("S", (), {"width": 8, "height": 8}))
    exp = rref.refine_class_ref([("s_00", stage1, {k: (v.clone() if torch.is_tensor(v) else list(v)) for k, v in stage2.items()})],
                                cfg, "table", enc)
    got = refinement.refine_class([("s_00", stage1, stage2)], cfg, "table", refinement.TextSimilarity(enc, DEV), DEV)
    same(got["s_00"].to_dict(), exp["s_00"])


def test_row_arena_recycling(api):
    """The instance rows are a zero arena that is cleared sparsely once their last reader is done: the workspace of
    bff_scene_project (one per stream; the clear is part of the call) and _lib.RowArena for the step-by-step path.
    Several scenes of different shapes through the same arenas, a front whose results are never collected (arena
    possibly dirty: it must be zeroed again, never handed out dirty) and the plain-tensor debug path all give the same
    results, and the arenas are all zero whenever nothing is in flight."""
    projection, _ = api
    from beyond_fixed_forms_amd import _lib, pipeline
    from beyond_fixed_forms_amd.scene import prepare_scene
    from beyond_fixed_forms_amd.synthetic import make_scene
    scenes = [make_scene("tiny", seed=31), make_scene("tiny", seed=32, n_points=4001, n_masks=40), make_scene("c1", seed=33)]
    ref = []
    for sc in scenes:
        cfg = cfg_for(sc)
        ref.append(projection.run_projection(prepare_scene(sc, cfg, device=DEV), cfg, debug_out=True))    # own tensors
    arena = _lib.RowArena.for_current_stream(torch.device(DEV))
    for fast in (True, False):
        for rnd in range(2):
            for sc, exp in zip(scenes, ref):
                cfg = cfg_for(sc)
                ds = prepare_scene(sc, cfg, device=DEV)
                if rnd == 1 and sc is scenes[1]:
                    projection.projection_front(ds, cfg, fast=fast)          # abandoned: its rows stay dirty
                    if not fast:
                        assert arena.busy
                got = projection.projection_back(projection.projection_front(ds, cfg, fast=fast))
                assert got.debug.get("path", "step") == ("fast" if fast else "step")
                if fast:
                    ws = pipeline.SceneWorkspace.for_current_stream(torch.device(DEV))
                    torch.cuda.synchronize()
                    assert not ws.in_flight and not ws.rows_dirty and int(ws.t["rows"].count_nonzero()) == 0
                else:
                    assert not arena.busy and (arena.buf is None or int(arena.buf.count_nonzero()) == 0)
                assert torch.equal(got.rows, exp.rows) and torch.equal(got.conf, exp.conf) and got.groups == exp.groups


def test_raw_depth_path_equals_float_depth_path(api):
    """A scene given as raw uint16 depth at sensor resolution (scaled + resized on the device) gives the same
    result as the same scene with the depth resized on the host (io.resize_bilinear_f32)."""
    projection, _ = api
    from beyond_fixed_forms_amd import io
    from beyond_fixed_forms_amd.synthetic import make_scene
    scene = make_scene("tiny", seed=31)
    small = {f: np.round(d[::2, ::2].astype(np.float64) * 1000).astype(np.uint16) for f, d in scene.depths.items()}
    host = make_scene("tiny", seed=31)
    host.depths = {f: io.resize_bilinear_f32(r.astype(np.float32) / np.float32(1000), scene.width, scene.height)
                   for f, r in small.items()}
    dev = make_scene("tiny", seed=31)
    dev.depths, dev.depths_raw = {}, small
    cfg = cfg_for(scene)
    a = projection.project_scene(host, cfg, DEV, return_result=True, debug_out=True)
    b = projection.project_scene(dev, cfg, DEV, return_result=True, debug_out=True)
    assert torch.equal(a.debug["raw_rows"], b.debug["raw_rows"]) and torch.equal(a.rows, b.rows)
    assert torch.equal(a.debug["viewed_counts"], b.debug["viewed_counts"])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        exp = pref.project_scene_ref(host, cfg)
    same(b.to_dict(), exp)


def test_text_bank_round_trip(api, tmp_path):
    """The similarity service saved to disk and reloaded without any text encoder gives identical similarities."""
    _, refinement = api
    from beyond_fixed_forms_amd.synthetic import make_text_bank
    bank, index = make_text_bank(96, seed=3)                 # 96 -> padded to a multiple of 32 internally
    sim = refinement.TextSimilarity(bank_encoder(bank.float(), index), DEV)
    a = sim.similarities("table", ["chair", "office_chair", "table"])
    sim.save(str(tmp_path / "bank.pt"))
    sim2 = refinement.TextSimilarity.from_file(str(tmp_path / "bank.pt"), DEV)
    assert sim2.similarities("table", ["chair", "office_chair", "table"]) == a
    with pytest.raises(KeyError):
        sim2.similarities("never seen query", ["chair"])
    exp = torch.nn.functional.cosine_similarity(bank[index["table"]].double()[None], bank[[index[k] for k in ("chair", "office_chair", "table")]].double())
    assert np.abs(np.array(a) - exp.numpy()).max() <= 1e-4


def test_empty_inputs(api):
    """No 2-D masks at all (P:465-478) and nothing merged (P:496-509): the reference's empty form."""
    projection, refinement = api
    from beyond_fixed_forms_amd.synthetic import make_scene, make_text_bank
    scene = make_scene("tiny", seed=30)
    cfg = cfg_for(scene)
    scene.mask_2d = []
    got = projection.project_scene(scene, cfg, DEV)
    assert tuple(got["ins"].shape) == (1, 0) and got["ins"].dtype == torch.float32 and len(got["conf"]) == 0
    scene = make_scene("tiny", seed=30)
    cfg_hi = cfg_for(scene, iou_thres=1.0)          # nothing can exceed IoU 1 -> no merges
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        exp = pref.project_scene_ref(scene, cfg_hi)
    got = projection.project_scene(scene, cfg_hi, DEV, return_result=True)
    assert exp["final_class"] == [] and got.to_dict()["final_class"] == [] and tuple(got.to_dict()["ins"].shape) == (1, 0)
    bank, index = make_text_bank(64, seed=1)
    enc = bank_encoder(bank.float(), index)
    # a class whose only scene has an empty stage 2 has no similarities: the reference raises IndexError (R:324)
    with pytest.raises(IndexError):
        rref.refine_class_ref([(scene.scene_id, scene.stage1, exp)], cfg, "table", enc)
    with pytest.raises(IndexError):
        refinement.refine_class([(scene.scene_id, scene.stage1, got)], cfg, "table", refinement.TextSimilarity(enc, DEV), DEV)


def test_missing_library_fails_loudly(monkeypatch):
    from beyond_fixed_forms_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libbff_hip.so")
    with pytest.raises(_lib.BffLibraryError):
        _lib.load()


def _run_class_worker(rank, world, port, q, over=None):
    import os
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)      # both ranks share the box's one GPU
    from beyond_fixed_forms_amd import distributed as bd
    from beyond_fixed_forms_amd.config import Config
    from beyond_fixed_forms_amd.refinement import TextSimilarity
    from beyond_fixed_forms_amd.synthetic import make_scene, make_text_bank
    scenes = [make_scene("tiny", seed=50 + i) for i in range(5)]
    for i, sc in enumerate(scenes):
        sc.scene_id = f"scene{50 + i:04d}_00"
    cfg = Config.with_defaults(width_2d=scenes[0].width, height_2d=scenes[0].height, **(over or {}))
    if over:                      # occurrence mode loads no viewed-only frames from disk: drop them here too
        for sc in scenes:
            keep = {f["frame_id"][:-4] for f in sc.mask_2d}
            sc.poses = {k: v for k, v in sc.poses.items() if k in keep}
            sc.depths = {k: v for k, v in sc.depths.items() if k in keep}
    bank, index = make_text_bank(64, seed=7)
    out = bd.run_class(scenes, cfg, "table", TextSimilarity(bank_encoder(bank.float(), index), DEV), DEV,
                       weights=[s.points.shape[0] * len(s.mask_2d) for s in scenes])
    if rank == 0:
        q.put({sid: (None if f.rows is None else f.rows.cpu(), f.conf if isinstance(f.conf, list) else f.conf.cpu(), list(f.final_class),
                     f.n_points) for sid, f in out.items()})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("over", [None, dict(if_occurance_threshold=True)], ids=["ratio", "occurrence"])
def test_run_class_two_ranks_equals_single_process(api, over):
    """distributed.run_class with 2 ranks (sharded scenes, ONE all-gather of similarity sets + row bounds, ONE
    gather of padded result matrices) gives, on rank 0, exactly the results of the single-process class loop --
    and of the oracle.  Occurrence mode: scenes as io.load_scene delivers them, without the viewed-only frames."""
    import socket
    import torch.multiprocessing as mp
    projection, refinement = api
    from beyond_fixed_forms_amd.synthetic import make_scene, make_text_bank
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_run_class_worker, args=(r, 2, port, q, over)) for r in range(2)]
    [p.start() for p in procs]
    got = q.get(timeout=300)
    [p.join(timeout=120) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    scenes = [make_scene("tiny", seed=50 + i) for i in range(5)]
    for i, sc in enumerate(scenes):
        sc.scene_id = f"scene{50 + i:04d}_00"
    cfg = cfg_for(scenes[0], **(over or {}))
    bank, index = make_text_bank(64, seed=7)
    enc = bank_encoder(bank.float(), index)
    trip = []
    for sc in scenes:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            trip.append((sc.scene_id, sc.stage1, pref.project_scene_ref(sc, cfg)))
    exp = rref.refine_class_ref(trip, cfg, "table", enc)
    assert sorted(got) == sorted(exp)
    for sid, (rows, conf, cls, n_points) in got.items():
        e = exp[sid]
        if isinstance(e["ins"], list):
            assert rows is None and cls == []
            continue
        n = e["ins"].shape[1]
        assert n_points == n
        dense = np.unpackbits(rows.numpy().view(np.uint8), axis=-1, bitorder="little")[:, :n].astype(bool)
        assert np.array_equal(dense, e["ins"].numpy()) and torch.equal(conf, e["conf"]) and cls == e["final_class"]


def _rccl_one_rank_worker(port, q):
    import os
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", BFF_FORCE_COLLECTIVES="1")
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    from beyond_fixed_forms_amd import distributed as bd
    from beyond_fixed_forms_amd.config import Config
    from beyond_fixed_forms_amd.refinement import TextSimilarity
    from beyond_fixed_forms_amd.synthetic import make_scene, make_text_bank
    scenes = [make_scene("tiny", seed=50 + i) for i in range(3)]
    for i, sc in enumerate(scenes):
        sc.scene_id = f"scene{50 + i:04d}_00"
    cfg = Config.with_defaults(width_2d=scenes[0].width, height_2d=scenes[0].height)
    bank, index = make_text_bank(64, seed=7)
    sim = TextSimilarity(bank_encoder(bank.float(), index), "cuda:0")
    ex = bd.ClassExchange("cuda:0")
    ids = [s.scene_id for s in scenes]
    from beyond_fixed_forms_amd.pipeline import project_stream
    batch = bd.ClassBatch(cfg, "table", sim, "cuda:0", ids, len(ids), exchange=ex)
    project_stream(scenes, cfg, "cuda:0", lambda k, st1, res: batch.add(ids[k], st1, res), n_loaders=1)
    out = batch.finish().results()                 # decoded from the buffers the RCCL gather delivered
    assert dist.get_backend() == "nccl" and ex.calls == 1 and ex._comm is not None and batch.gathered is not None
    # the plain helpers over RCCL as well
    pooled = bd.exchange_similarities([[0.25, 0.5], [0.5]], device="cuda:0")
    rows = bd.gather_final_rows(torch.arange(12, dtype=torch.int64, device="cuda:0").view(3, 4))
    q.put(({sid: (None if f.rows is None else f.rows.cpu(), f.conf if isinstance(f.conf, list) else f.conf.cpu(), list(f.final_class),
                  f.n_points) for sid, f in out.items()}, pooled, rows[0].cpu()))
    dist.barrier()
    dist.destroy_process_group()


def test_rccl_branch_with_one_rank(api):
    """The `nccl` (= RCCL) branch of the class's two collectives -- ClassExchange on a stream of its own with pinned
    staging + all_gather_into_tensor, the gather of the packed result matrices, the plain helpers -- executed on this
    box's one GPU as a process group of ONE rank (BFF_FORCE_COLLECTIVES=1: a class then exchanges and gathers although
    nobody else is there).  Results decoded from the gathered buffers == the oracle's."""
    import socket
    import torch.multiprocessing as mp
    from beyond_fixed_forms_amd.synthetic import make_scene, make_text_bank
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_one_rank_worker, args=(port, q))
    p.start()
    got, pooled, rows = q.get(timeout=300)
    p.join(timeout=120)
    assert p.exitcode == 0
    assert pooled == [[0.25, 0.5]] and torch.equal(rows, torch.arange(12, dtype=torch.int64).view(3, 4))
    scenes = [make_scene("tiny", seed=50 + i) for i in range(3)]
    for i, sc in enumerate(scenes):
        sc.scene_id = f"scene{50 + i:04d}_00"
    cfg = cfg_for(scenes[0])
    bank, index = make_text_bank(64, seed=7)
    enc = bank_encoder(bank.float(), index)
    trip = []
    for sc in scenes:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            trip.append((sc.scene_id, sc.stage1, pref.project_scene_ref(sc, cfg)))
    exp = rref.refine_class_ref(trip, cfg, "table", enc)
    assert sorted(got) == sorted(exp)
    for sid, (rows, conf, cls, n_points) in got.items():
        e = exp[sid]
        if isinstance(e["ins"], list):
            assert rows is None and cls == []
            continue
        n = e["ins"].shape[1]
        dense = np.unpackbits(rows.numpy().view(np.uint8), axis=-1, bitorder="little")[:, :n].astype(bool)
        assert n_points == n and np.array_equal(dense, e["ins"].numpy()) and torch.equal(conf, e["conf"]) and cls == e["final_class"]


@pytest.mark.parametrize("enc_dtype", [torch.float32, torch.float16])
def test_similarity_set_near_ties(api, enc_dtype):
    """The class threshold is sorted(set(similarities))[int(n * 0.2)] (R:321-324): it depends on which values TIE.
    Labels are crafted whose cosines with the query differ by 2e-6 .. 3e-4: distinct in float32, partly collapsing
    in float16 (CLIP on a GPU emits fp16; cosines are then multiples of 2^-11 .. 2^-13).  The product computes the
    cosines in the encoder's dtype with the reference's op-by-op rounding (bff_cosine_rows), so the SET -- its
    size, its order, hence the index picked and the masks dropped -- equals the oracle's on the same embeddings."""
    _, refinement = api
    from beyond_fixed_forms_amd.labels import SCANNET200_LABELS
    from oracle.rle_ref import rle_encode_batch_ref
    n, k, d = 4000, 12, 64
    gen = torch.Generator().manual_seed(41)
    q = torch.randn(d, generator=gen, dtype=torch.float64)
    q /= q.norm()
    target = [0.30, 0.30 + 2e-6, 0.30 + 6e-6, 0.3001, 0.3003, 0.31, 0.31 + 3e-6, 0.35, 0.3502, 0.40, 0.41, 0.41 + 5e-5]
    bank = torch.randn(len(SCANNET200_LABELS), d, generator=gen, dtype=torch.float64)
    labs = [SCANNET200_LABELS[5 + 3 * i] for i in range(k)]
    for c, lab in zip(target, labs):
        r = torch.randn(d, generator=gen, dtype=torch.float64)
        r -= (r @ q) * q
        r /= r.norm()
        bank[SCANNET200_LABELS.index(lab)] = (c * q + (1 - c * c) ** 0.5 * r) * 3.7
    bank[SCANNET200_LABELS.index("table")] = q * 2.5
    bank = bank.to(enc_dtype)
    index = {lab: i for i, lab in enumerate(SCANNET200_LABELS)}
    enc = bank_encoder(bank, index)
    # stage 1: k disjoint masks with the crafted labels (+ one with the query label); stage 2: one mask per
    # stage-1 mask with IoU 0.9 > refiment_iou_thres, so every matched mask is kept or dropped by its similarity
    s1 = np.zeros((k + 1, n), bool)
    s2 = np.zeros((k, n), bool)
    for i in range(k):
        s1[i, 300 * i: 300 * i + 200] = True
        s2[i, 300 * i + 10: 300 * i + 200] = True
    s1[k, 3800:3900] = True
    stage1 = {"ins": rle_encode_batch_ref(torch.from_numpy(s1)), "conf": torch.ones(k + 1),
              "final_class": [SCANNET200_LABELS.index(l) for l in labs] + [SCANNET200_LABELS.index("table")]}
    conf = torch.linspace(0.2, 0.5, k).to(torch.float16)
    cfg = cfg_for(type("S", (), {"width": 8, "height": 8}))
    # three scenes of the class see different subsets of the labels; the set is pooled over all of them
    scenes_p, scenes_o = [], []
    for s, sel in enumerate(([0, 1, 2, 3, 9], [4, 5, 6, 7, 10], [1, 6, 8, 11, 2])):
        st2 = {"ins": torch.from_numpy(s2[sel]), "conf": conf[sel].clone(), "final_class": ["table"] * len(sel)}
        scenes_p.append((f"s{s}_00", stage1, st2))
        scenes_o.append((f"s{s}_00", stage1, {kk: (v.clone() if torch.is_tensor(v) else list(v)) for kk, v in st2.items()}))
    fexp, odbg = rref.refine_class_ref(scenes_o, cfg, "table", enc, return_debug=True)
    sim = refinement.TextSimilarity(enc, DEV)
    assert sim.dtype == enc_dtype
    fin, dbg = refinement.refine_class(scenes_p, cfg, "table", sim, DEV, return_debug=True)
    got_set = sorted(set(v for st in dbg["states"] for v in st.sims))
    exp_set = odbg["sim_unique"]
    assert len(got_set) == len(exp_set)                                   # the same values tie
    if enc_dtype == torch.float16:
        assert len(exp_set) < k                                           # ... and in fp16 some do
        assert got_set == exp_set and dbg["sim_thres"] == odbg["sim_thres"]
    else:
        assert len(exp_set) == k                                          # 2e-6 apart: all distinct in float32
        assert np.abs(np.array(got_set) - np.array(exp_set)).max() <= 5e-7      # a few float32 ulps; the values are >= 2e-6 apart
        assert abs(dbg["sim_thres"] - odbg["sim_thres"]) <= 5e-7
    # same order of the labels along the sorted set -> same index -> same masks dropped
    for (sid, _, _) in scenes_p:
        same(fin[sid].to_dict(), fexp[sid])
    dropped = sum(5 + 1 - fexp[sid]["ins"].shape[0] for sid, _, _ in scenes_p)     # 5 matched + 1 other per scene
    assert dropped > 0                                                    # the threshold did bite


@pytest.mark.parametrize("case", ["more_groups_than_the_device_forms", "groups_beyond_the_default_tables", "hundred_groups",
                                  "min_members_0", "many_stage2"])
def test_fast_path_and_its_fallbacks(api, case):
    """bff_scene_project (one native call per scene, groups formed on the device) against the oracle and against the
    step-by-step path: (a) a scene whose merge graph has more kept groups (552) than BFF_GROUP_CAP_MAX = 512 -- the device
    tables are incomplete, the host continues from the components (general path); (a') 325 groups: more than the
    default tables (BFF_GROUP_CAP = 256) hold, so the scene is issued again with the large ones and stays on the one-call
    path (fused overlap pass on half-word columns); (a'') ~90 groups, handled on the device (pair masks of several
    words in the fused overlap pass); (b) min_aggragated_masks = 0, where the reference's empty
    components survive the filter (general path too); (c) a scene with many surviving stage-2 instances on the fast
    path.  Stage-2 and final results are bit-identical in all cases."""
    projection, refinement = api
    from beyond_fixed_forms_amd.scene import prepare_scene
    from beyond_fixed_forms_amd.synthetic import make_scene, make_text_bank
    over = {}
    if case == "more_groups_than_the_device_forms":
        scene = make_scene("tiny", seed=30, n_labels=90, n_masks=64, n_views=40, cut_masks=False)
    elif case == "groups_beyond_the_default_tables":
        scene = make_scene("tiny", seed=30, n_labels=50, n_masks=64, n_views=24, cut_masks=False)
    elif case == "hundred_groups":
        scene = make_scene("tiny", seed=30, n_labels=12, n_masks=64, n_views=6, cut_masks=False)
    elif case == "min_members_0":
        scene = make_scene("tiny", seed=31)
        over = dict(min_aggragated_masks=0)
    else:
        scene = make_scene("c1", seed=32, n_views=40, n_masks=30, cut_masks=False, n_objects=30, distinct_masks=True,
                           dilate=False)
    cfg = cfg_for(scene, **over)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        exp, dbg = pref.project_scene_ref(scene, cfg, return_debug=True)
    st1 = refinement.prepare_stage1(scene.stage1, DEV)
    ds = prepare_scene(scene, cfg, device=DEV)
    res = projection.run_projection(ds, cfg, stage1=st1)                       # fast front
    slow = projection.run_projection(ds, cfg, debug_out=True)                  # step by step
    if case == "many_stage2":
        assert res.debug["path"] == "fast" and exp["ins"].shape[0] >= 5
    elif case == "hundred_groups":
        assert res.debug["path"] == "fast" and 64 < len(dbg["groups"]) <= 256
    elif case == "groups_beyond_the_default_tables":
        assert res.debug["path"] == "fast" and 256 < len(dbg["groups"]) <= 512 and ds.__dict__.get("_group_cap") == 512
    else:
        assert res.debug["path"].startswith("general")
        if case == "more_groups_than_the_device_forms":
            assert len(dbg["groups"]) > 512
    assert list(res.groups) == list(slow.groups) == dbg["groups"]
    same(res.to_dict(), exp)
    same(slow.to_dict(), exp)
    assert torch.equal(res.rows, slow.rows) and torch.equal(res.conf, slow.conf)
    assert abs(res.debug["thr"] - slow.debug["thr"]) == 0
    bank, index = make_text_bank(64, seed=3)
    enc = bank_encoder(bank, index)                                            # float16 embeddings, as CLIP on a GPU
    sim = refinement.TextSimilarity(enc, DEV)
    if len(exp["conf"]) == 0:
        return
    fexp = rref.refine_class_ref([(scene.scene_id, scene.stage1, exp)], cfg, "table", enc)
    assert res.prefetch is not None
    fin = refinement.refine_class([(scene.scene_id, st1, res)], cfg, "table", sim, DEV)
    same(fin[scene.scene_id].to_dict(), fexp[scene.scene_id])
    # the workspace is reused: a second scene through the same stream, then the first one again
    other = make_scene("tiny", seed=33)
    ocfg = cfg_for(other)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        oexp = pref.project_scene_ref(other, ocfg)
    same(projection.run_projection(prepare_scene(other, ocfg, device=DEV), ocfg).to_dict(), oexp)
    again = projection.run_projection(ds, cfg)
    same(again.to_dict(), exp)
    assert list(again.groups) == dbg["groups"]


def test_overlapped_ingestion_equals_prepare_scene(api):
    """ingest.prepare_scene_fast (native run tables, batched pose inverses, packed pinned uploads, cloud sorted and
    laid out on the device) builds the same DeviceScene as scene.prepare_scene -- same sorted cloud, same inverse
    permutation, same inverse poses bit for bit, same depth -- and the Ingestor's loader threads (own streams, events)
    feed the device path with identical results, for float32 depth and for raw 16-bit depth resized on the device."""
    projection, refinement = api
    from beyond_fixed_forms_amd import ingest
    from beyond_fixed_forms_amd.scene import prepare_scene
    from beyond_fixed_forms_amd.synthetic import make_scene
    scenes = [make_scene("tiny", seed=60 + i, n_masks=(5, 40, 9)[i]) for i in range(3)]
    cfg = cfg_for(scenes[0])
    for sc in scenes:
        a = prepare_scene(sc, cfg, device=DEV)
        b = ingest.prepare_scene_fast(sc, cfg, device=DEV)
        torch.cuda.synchronize()
        for k in ("xyz", "unsort", "inv_pose", "depth", "depth_index", "frame_mask", "frame_rowbase", "frame_nmask",
                  "frame_flags", "view_mask_offs", "conf", "label_id", "tile_bounds", "mask_run_offs", "run_start", "run_end"):
            assert torch.equal(getattr(a, k), getattr(b, k)), k
        assert (a.n_frames, a.n_mask_frames, a.n_viewed, a.word_bits, a.n_rows, a.labels) == \
               (b.n_frames, b.n_mask_frames, b.n_viewed, b.word_bits, b.n_rows, b.labels)
    # raw 16-bit frames at half resolution, scaled + resized on the device: same as the step-by-step upload of them
    import copy
    raw = copy.copy(scenes[0])
    raw.depths_raw = {f: np.ascontiguousarray(np.round(d[::2, ::2].astype(np.float64) * 1000).astype(np.uint16))
                      for f, d in raw.depths.items()}
    a, b = prepare_scene(raw, cfg, device=DEV), ingest.prepare_scene_fast(raw, cfg, device=DEV)
    assert a.depth is None and b.depth is None and torch.equal(a.depth_raw, b.depth_raw)      # resident as stored
    assert a.depth_size == b.depth_size == (cfg.height_2d // 2, cfg.width_2d // 2)           # frames in 8 x 8 tiles
    # the same frames already in page-locked memory in upload order (what a decoder with a pinned output leaves): read in
    # place, no packing; a block in another order is not trusted and takes the packing path -- same scene either way
    from beyond_fixed_forms_amd.scene import viewed_frame_ids
    ids = list(dict.fromkeys([fr["frame_id"][:-4] for fr in raw.mask_2d] + viewed_frame_ids(raw.color_files, cfg.downsample_ratio)))
    f0 = raw.depths_raw[ids[0]]
    real_lib, packs = ingest.host_lib(), []

    class Counting:                                  # the native library with a counter on the packing entry point
        def __getattr__(self, name):
            if name == "bff_host_pack_frames":
                packs.append(1)
            return getattr(real_lib, name)

    for id_list, packed in ((ids, 0), (ids[::-1], 1)):
        block = torch.empty((len(id_list),) + tuple(f0.shape), dtype=torch.int16).pin_memory()
        view = block.numpy().view(np.uint16)
        for k, f in enumerate(id_list):
            view[k] = raw.depths_raw[f]
        st = copy.copy(raw)
        st.depths_raw = {f: view[k] for k, f in enumerate(id_list)}
        st.depth_staged = (block, id_list)
        del packs[:]
        ingest._host = Counting()
        try:
            c = ingest.prepare_scene_fast(st, cfg, device=DEV)
        finally:
            ingest._host = real_lib
        torch.cuda.synchronize()
        assert len(packs) == packed
        assert torch.equal(c.depth_raw, a.depth_raw) and torch.equal(c.depth_index, a.depth_index)
    monkey_env = dict(os.environ)
    os.environ["BFF_DEPTH_RESIZE_PASS"] = "1"                # the separate scale + resize pass: float32 (H, W) images
    try:
        a, b = prepare_scene(raw, cfg, device=DEV), ingest.prepare_scene_fast(raw, cfg, device=DEV)
    finally:
        os.environ.clear(); os.environ.update(monkey_env)
    assert torch.equal(a.depth, b.depth) and a.depth.shape[1] == cfg.height_2d * cfg.width_2d
    # loader threads: results through the pipeline == results of the plain path == oracle
    ing = ingest.Ingestor(cfg, DEV, n_loaders=2, native_threads=2)
    futs = [ing.submit(sc) for sc in scenes * 2]
    st = torch.cuda.Stream(device=DEV)
    for i, f in enumerate(futs):
        ds, st1, ev = f.result()
        st.wait_event(ev)
        with torch.cuda.stream(st):
            res = projection.run_projection(ds, cfg, stage1=st1)
        sc = scenes[i % 3]
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            exp = pref.project_scene_ref(sc, cfg)
        same(res.to_dict(), exp)
    ing.close()


def test_value_set_threshold_in_the_scene_call(api, monkeypatch):
    """bff_scene_project takes the point-filter threshold from the set of distinct values (bff_point_threshold_pairs:
    two launches) -- same threshold bits and results as sorting all values (BFF_FILTER_SORT=1, the 12-launch path it
    falls back to when a scene has more distinct values than the set holds)."""
    projection, _ = api
    from beyond_fixed_forms_amd.scene import prepare_scene
    from beyond_fixed_forms_amd.synthetic import make_scene
    for mode_over in ({}, dict(if_occurance_threshold=True)):
        scene = make_scene("tiny", seed=71)
        cfg = cfg_for(scene, **mode_over)
        ds = prepare_scene(scene, cfg, device=DEV)
        monkeypatch.delenv("BFF_FILTER_SORT", raising=False)
        a = projection.run_projection(ds, cfg)
        monkeypatch.setenv("BFF_FILTER_SORT", "1")
        b = projection.run_projection(ds, cfg)
        assert a.debug["path"] == b.debug["path"] == "fast"
        assert np.float32(a.debug["thr"]).tobytes() == np.float32(b.debug["thr"]).tobytes()
        assert torch.equal(a.rows, b.rows) and torch.equal(a.conf, b.conf) and list(a.groups) == list(b.groups)


def test_more_distinct_filter_values_than_the_set_holds(api):
    """A scene whose filter statistic takes more distinct values than the merging set accepts (forced here by shrinking
    every hash partition's set to 1 value): the header's overflow word makes the host re-issue the scene with the sorting formulation --
    same results as the oracle."""
    projection, _ = api
    from beyond_fixed_forms_amd import _lib
    from beyond_fixed_forms_amd.scene import prepare_scene
    from beyond_fixed_forms_amd.synthetic import make_scene
    scene = make_scene("tiny", seed=72)
    cfg = cfg_for(scene)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        exp = pref.project_scene_ref(scene, cfg)
    ds = prepare_scene(scene, cfg, device=DEV)
    lib = _lib.load()
    assert lib.bff_point_threshold_capacity_set(1) == 1
    try:
        res = projection.run_projection(ds, cfg)
    finally:
        lib.bff_point_threshold_capacity_set(0)
    assert res.debug["path"] == "fast" and ds.__dict__.get("_filter_sort") is True
    same(res.to_dict(), exp)


RAW_DEPTH_CASES = {
    # scene keywords, size of the stored 16-bit depth frames (None: same size as the working image)
    "half": (dict(shape="tiny", seed=80), lambda h, w: (h // 2, w // 2)),
    "same_size": (dict(shape="tiny", seed=81), None),
    "scannet_ratio": (dict(shape="c1", seed=82, n_views=4), lambda h, w: (240, 320)),
    "odd_ratio_up": (dict(shape="tiny", seed=83, height=97, width=131), lambda h, w: (41, 67)),
    "down_and_up": (dict(shape="tiny", seed=84), lambda h, w: (h * 2 + 3, w - 29)),     # rows shrink, columns stretch
    "one_row_source": (dict(shape="tiny", seed=85), lambda h, w: (1, 2)),
}


@pytest.mark.parametrize("case", list(RAW_DEPTH_CASES))
def test_sweep_resizes_raw_depth_per_point(api, case):
    """Depth resident as the PNGs store it (uint16, sensor resolution): the sweep evaluates /1000 + the bilinear resize
    at the pixel each point projects to (bff_project_views_u16).  Raw rows, both counters and the final masks are
    bit-identical (a) to the two-step path (bff_depth_from_u16 into float32 (H, W) images, then the sweep) and (b) to
    the oracle fed with the host restatement of the resize (io.resize_bilinear_f32; cv2 itself is unpinned) -- on
    the step-by-step path and on the one-call production path."""
    import copy
    projection, refinement = api
    from beyond_fixed_forms_amd.ingest import prepare_scene_fast
    from beyond_fixed_forms_amd.io import resize_bilinear_f32
    from beyond_fixed_forms_amd.scene import prepare_scene
    from beyond_fixed_forms_amd.synthetic import make_scene
    kw, size = RAW_DEPTH_CASES[case]
    scene = make_scene(**kw)
    h, w = scene.height, scene.width
    rng = np.random.default_rng(7)
    raw = copy.copy(scene)
    raw.depths_raw = {}
    for f, d in scene.depths.items():
        mm = np.round(d.astype(np.float64) * 1000).astype(np.uint16)
        if size is not None:
            hs, ws = size(h, w)
            yy = np.minimum((np.arange(hs) * h) // hs, h - 1)
            xx = np.minimum((np.arange(ws) * w) // ws, w - 1)
            mm = np.ascontiguousarray(mm[yy][:, xx])
            mm[rng.random(mm.shape) < 0.02] = 0                                   # sensor holes
        raw.depths_raw[f] = mm
    host = copy.copy(scene)                       # what the reference would hold after cv2.imread / 1000 + cv2.resize
    host.depths = {f: resize_bilinear_f32(m.astype(np.float32) / np.float32(1000), w, h) for f, m in raw.depths_raw.items()}
    cfg = cfg_for(scene)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        exp, dbg = pref.project_scene_ref(host, cfg, return_debug=True)
    ds_raw = prepare_scene(raw, cfg, device=DEV, raw_depth_resident=True)            # 8 x 8-texel tiles of float32 metres (default)
    try:
        os.environ["BFF_DEPTH_TILES"] = "0"
        ds_lin = prepare_scene(raw, cfg, device=DEV, raw_depth_resident=True)        # uint16 frames row-major, as stored
        os.environ["BFF_DEPTH_TILES"] = "u16"
        ds_u16 = prepare_scene(raw, cfg, device=DEV, raw_depth_resident=True)        # uint16 in tiles
    finally:
        del os.environ["BFF_DEPTH_TILES"]
    ds_two = prepare_scene(raw, cfg, device=DEV, raw_depth_resident=False)
    assert ds_raw.depth is None and ds_raw.depth_size is not None and ds_raw.depth_raw.dtype == torch.float32
    assert ds_lin.depth_size is None and ds_u16.depth_raw.dtype == torch.int16 and ds_u16.depth_size is not None and ds_two.depth_raw is None
    c = projection.run_projection(ds_lin, cfg, debug_out=True)
    c2 = projection.run_projection(ds_u16, cfg, debug_out=True)
    from beyond_fixed_forms_amd.scene import viewed_frame_ids
    slots = list(dict.fromkeys([fr["frame_id"][:-4] for fr in raw.mask_2d] + viewed_frame_ids(raw.color_files, cfg.downsample_ratio)))
    host_depth = torch.from_numpy(np.stack([host.depths[f].reshape(-1) for f in slots]))
    assert torch.equal(ds_two.depth.cpu(), host_depth)                            # device resize pass == host restatement
    a = projection.run_projection(ds_raw, cfg, debug_out=True)
    b = projection.run_projection(ds_two, cfg, debug_out=True)
    for k in ("raw_rows", "masked_counts_raw", "viewed_counts"):
        assert torch.equal(a.debug[k], b.debug[k]) and torch.equal(c.debug[k], b.debug[k]) and torch.equal(c2.debug[k], b.debug[k]), k
    n = scene.points.shape[0]
    rawbits = np.unpackbits(a.debug["raw_rows"].cpu().numpy().view(np.uint8), axis=-1, bitorder="little")[:, :n].astype(bool)
    assert np.array_equal(rawbits, dbg["raw_ins"].numpy())
    assert np.array_equal(a.debug["viewed_counts"].cpu().numpy(), dbg["viewed_counts"].numpy().astype(np.int32))
    same(a.to_dict(), exp)
    same(b.to_dict(), exp)
    prod = projection.projection_back(projection.projection_front(prepare_scene_fast(raw, cfg, DEV), cfg))   # one native call
    assert prod.debug["path"] == "fast"
    same(prod.to_dict(), exp)
