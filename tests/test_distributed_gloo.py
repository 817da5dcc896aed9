"""The N > 1 path on CPU: world_size 2, gloo.  Scenes shard across ranks, the similarity sets are
all-gathered (the one exchange step of the path), final bit rows are gathered on rank 0."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from beyond_fixed_forms_amd import distributed as bd
    from beyond_fixed_forms_amd.refinement import sim_threshold
    scenes = [f"scene{i:04d}_00" for i in range(5)]
    mine = bd.shard_scenes(scenes)
    all_sims = {0: [0.31, 0.12], 1: [], 2: [0.12, 0.77, 0.5], 3: [0.05], 4: [0.31]}       # per scene
    local = [all_sims[i] for i in mine]
    pooled = bd.exchange_similarities(local)
    thr = sim_threshold(pooled, 0.2)
    rows = torch.full((rank + 1, 3 + rank), rank + 7, dtype=torch.int64)                 # ragged shapes per rank
    gathered = bd.gather_final_rows(rows)
    # the same with the sizes agreed in the similarity exchange: one all-gather + one gather, no size exchange
    ex = bd.ClassExchange()
    pooled2 = ex(local, bounds=(rows.shape[0], rows.shape[1]))
    assert sorted(map(sorted, pooled2)) == sorted(map(sorted, pooled)) and ex.bounds == (2, 4)
    g2 = bd.gather_final_rows(rows, bounds=ex.bounds)
    assert (g2 is None) == (gathered is None)
    if g2 is not None:
        assert [g.tolist() for g in g2] == [g.tolist() for g in gathered]
    # the result transport of a class (ClassBatch): metadata rows + bit rows of all scenes in ONE padded gather
    from beyond_fixed_forms_amd.refinement import FinalResult
    ids = [f"scene{i:04d}_00" for i in range(5)]
    gen = torch.Generator().manual_seed(rank)
    local = {}
    for i in mine:
        r = (i * 3 + 1) % 4                                   # 1, 0 (-> empty tensor form), 3, 2, ... rows
        if i == 3:
            local[ids[i]] = FinalResult(ids[i], 300 + i, None, [], [])      # the reference's list-valued empty form
        else:
            local[ids[i]] = FinalResult(ids[i], 300 + i, torch.randint(-2 ** 62, 2 ** 62, (r, 5 + i), generator=gen),
                                        torch.rand(r, generator=gen).to(torch.float16 if i % 2 else torch.float32), ["q"] * r)
    s_max = 3
    ex2 = bd.ClassExchange()
    ex2([[0.1]], bounds=(sum(0 if v.rows is None else v.rows.shape[0] for v in local.values()),
                         max([v.rows.shape[1] for v in local.values() if v.rows is not None] + [1])))
    r_max, width = ex2.bounds[0], max(ex2.bounds[1], 8)
    mat = bd.pack_class_results(local, {sid: k for k, sid in enumerate(ids)}, s_max, "cpu", bounds=(r_max, width))
    g3 = bd.gather_final_rows(mat, bounds=(bd.packed_header_rows(s_max, r_max, width) + r_max, width))
    plain = lambda d: {k: (v.n_points, None if v.rows is None else v.rows.tolist(),
                           v.conf if isinstance(v.conf, list) else (str(v.conf.dtype), v.conf.tolist()), v.final_class)
                       for k, v in d.items()}
    merged = None
    if g3 is not None:
        merged = {}
        for g in g3:
            merged.update(bd.unpack_class_results(g.rows(), ids, s_max, "q"))
        merged = plain(merged)
    mine_plain = plain(local)
    q.put((rank, mine, thr, None if gathered is None else [g.tolist() for g in gathered], mine_plain, merged))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in procs]
    out = sorted(q.get(timeout=120) for _ in range(world))
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    (r0, mine0, thr0, g0, loc0, merged0), (r1, mine1, thr1, g1, loc1, merged1) = out
    assert merged1 is None and merged0 == {**loc0, **loc1} and len(merged0) == 5      # rank 0 holds every scene's result
    assert mine0 == [0, 2, 4] and mine1 == [1, 3]
    from beyond_fixed_forms_amd.refinement import sim_threshold
    single = sim_threshold([[0.31, 0.12], [], [0.12, 0.77, 0.5], [0.05], [0.31]], 0.2)
    assert thr0 == thr1 == single                                  # same threshold as the single-process class loop
    assert g1 is None and g0 == [[[7] * 3], [[8] * 4, [8] * 4]]


def test_shard_scenes_weighted():
    from beyond_fixed_forms_amd.distributed import shard_scenes
    ids = list("abcdef")
    w = [5, 1, 9, 3, 7, 2]
    parts = [shard_scenes(ids, r, 3, w) for r in range(3)]
    assert sorted(i for p in parts for i in p) == list(range(6))
    assert parts[0] == [2, 3] and parts[1] == [4, 5] and parts[2] == [0, 1]      # 9,3 | 7,2 | 5,1
    assert shard_scenes(ids, 0, 1) == list(range(6))
