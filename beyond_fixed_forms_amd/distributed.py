"""One process per GPU: scenes shard across ranks, RCCL carries only what the path really exchanges.

The reference is a single process; scenes are independent through the whole
projection stage, and refinement has exactly one cross-scene dependency: the similarity threshold is
a percentile of the *set* of similarities over all scenes of the class (tools/refinement.py:316-324).
So: scenes round-robin over ranks, one all-gather of the (tiny) per-rank similarity lists, and one
gather of the final bit-packed masks to rank 0.  Over xGMI these messages are latency-bound (KBs to
a few MB), so flat all-gather / gather calls are used, never a ring reduction.

Works with backend "nccl" (= RCCL on ROCm) on GPUs and "gloo" on CPU tensors (tests).
"""
from __future__ import annotations

from typing import List, Sequence

import torch
import torch.distributed as dist

MAX_SIMS = 256          # a class meets at most one similarity per ScanNet200 label (198) per query


def _coll_device(t: torch.Tensor):
    """RCCL (backend "nccl") moves device tensors; gloo (CPU tests, single-GPU rehearsals) needs host copies."""
    return t if dist.get_backend() == "nccl" else t.cpu()


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_scenes(scene_ids: Sequence, rank: int = None, world_size: int = None, weights: Sequence[float] = None):
    """Static partition: scenes sorted by descending weight (N*V; default: listing order) are dealt
    round-robin.  Returns the indices owned by `rank`, in listing order."""
    r, w = world()
    rank = r if rank is None else rank
    world_size = w if world_size is None else world_size
    order = list(range(len(scene_ids)))
    if weights is not None:
        order.sort(key=lambda i: (-weights[i], i))
    mine = [order[k] for k in range(rank, len(order), world_size)]
    return sorted(mine)


def exchange_similarities(local_sims: List[List[float]], device="cpu") -> List[List[float]]:
    """All ranks' per-scene similarity lists -> one pooled list of lists (only the *set* of values
    matters to the threshold, refinement.py:321-324).  One flat all-gather of MAX_SIMS+1 doubles."""
    rank, ws = world()
    if ws == 1:
        return local_sims
    uniq = sorted(set(s for sims in local_sims for s in sims))
    if len(uniq) > MAX_SIMS:
        raise ValueError(f"{len(uniq)} distinct similarities on one rank (> {MAX_SIMS})")
    buf = torch.zeros(MAX_SIMS + 1, dtype=torch.float64, device=device)
    buf[0] = len(uniq)
    if uniq:
        buf[1:1 + len(uniq)] = torch.tensor(uniq, dtype=torch.float64)
    buf = _coll_device(buf)
    out = [torch.empty_like(buf) for _ in range(ws)]
    dist.all_gather(out, buf)
    return [o[1:1 + int(o[0].item())].tolist() for o in out]


class ClassExchange:
    """The one exchange of a class as a callable for `refine_class(exchange_sims=...)`: a single all-gather
    carries every rank's similarity set AND its bound on the final rows it will produce (known after pass 1:
    matched stage-2 rows + other stage-1 masks of the query label, R:293, 340-392), so that the later gather of
    the results (`gather_final_rows(rows, bounds=exchange.bounds)`) needs no size exchange and no host
    synchronisation of its own."""
    takes_bounds = True

    def __init__(self, device="cpu"):
        self.device = device
        self.bounds = None                      # (max rows, max words) over all ranks after a call

    def __call__(self, local_sims, bounds=(0, 1)):
        rank, ws = world()
        if ws == 1:
            self.bounds = (int(bounds[0]), int(bounds[1]))
            return local_sims
        uniq = sorted(set(s for sims in local_sims for s in sims))
        if len(uniq) > MAX_SIMS:
            raise ValueError(f"{len(uniq)} distinct similarities on one rank (> {MAX_SIMS})")
        host = torch.zeros(MAX_SIMS + 3, dtype=torch.float64)
        host[0], host[1], host[2] = len(uniq), float(bounds[0]), float(bounds[1])
        if uniq:
            host[3:3 + len(uniq)] = torch.tensor(uniq, dtype=torch.float64)
        buf = host if dist.get_backend() != "nccl" else host.to(self.device)
        out = [torch.empty_like(buf) for _ in range(ws)]
        dist.all_gather(out, buf)
        out = [o.cpu() for o in out]
        self.bounds = (max(int(o[1]) for o in out), max(1, max(int(o[2]) for o in out)))
        return [o[3:3 + int(o[0])].tolist() for o in out]


def gather_final_rows(rows: torch.Tensor, dst: int = 0, bounds=None):
    """Gather every rank's final bit rows (int64 [R][nw], R and nw may differ per rank) on `dst`.
    Returns, on dst, a list with one int64 [R_r][nw_r] tensor per rank; elsewhere None.

    bounds = (max rows, max words) agreed beforehand (ClassExchange.bounds): ONE collective and no host
    synchronisation -- every rank sends a [1 + max rows][max words] buffer whose first row holds its real
    (R, nw); dst gets the padded buffers back as `PaddedRows` (decoded on demand)."""
    rank, ws = world()
    if ws == 1:
        return [rows]
    dev = rows.device
    if bounds is not None:
        r_max, w_max = int(bounds[0]), max(2, int(bounds[1]))
        if rows.shape[0] > r_max or rows.shape[1] > w_max:
            raise ValueError(f"rows {tuple(rows.shape)} exceed the agreed bounds {(r_max, w_max)}")
        pad = torch.zeros((r_max + 1, w_max), dtype=torch.int64, device=dev)
        pad[0, 0], pad[0, 1] = rows.shape[0], rows.shape[1]
        if rows.numel():
            pad[1:1 + rows.shape[0], :rows.shape[1]] = rows
        pad = _coll_device(pad)
        bufs = [torch.empty_like(pad) for _ in range(ws)] if rank == dst else None
        dist.gather(pad, bufs, dst=dst)
        return None if rank != dst else [PaddedRows(b, dev) for b in bufs]
    shape = _coll_device(torch.tensor([rows.shape[0], rows.shape[1]], dtype=torch.int64, device=dev))
    shapes = [torch.empty_like(shape) for _ in range(ws)]
    dist.all_gather(shapes, shape)
    shapes = [s.cpu() for s in shapes]
    r_max = max(int(s[0]) for s in shapes)
    w_max = max(int(s[1]) for s in shapes)
    pad = torch.zeros((max(r_max, 1), max(w_max, 1)), dtype=torch.int64, device=dev)
    pad[:rows.shape[0], :rows.shape[1]] = rows
    pad = _coll_device(pad)
    bufs = [torch.empty_like(pad) for _ in range(ws)] if rank == dst else None
    dist.gather(pad, bufs, dst=dst)
    if rank != dst:
        return None
    return [b[:int(s[0]), :int(s[1])].to(dev) for b, s in zip(bufs, shapes)]


class PaddedRows:
    """One rank's gathered buffer: row 0 = (R, nw), rows 1..R the bit rows.  `.rows()` decodes (reads the header
    on the host, i.e. waits for the gather)."""

    def __init__(self, buf, device):
        self.buf, self.device = buf, device

    def rows(self):
        r, w = (int(v) for v in self.buf[0, :2].cpu())
        return self.buf[1:1 + r, :w].to(self.device)

    def tolist(self):
        return self.rows().tolist()


_CONF_CODES = {torch.float32: 0, torch.float16: 1, torch.float64: 2}


def pack_class_results(local, scene_index, s_max, device):
    """{scene_id: (rows int64 [R][nw] | None, conf tensor | [], final_class)} of this rank -> ONE int64 matrix
    [s_max + sum R][nw_max + 1] for a single gather: first s_max descriptor rows (global scene index, R, nw,
    saved-as-lists flag, confidence dtype code; unused ones hold -1), then the bit rows of all scenes, each with its
    confidence (float64 bits) in the extra last column.  final_class needs no transport: every row carries the
    query (R:343, 390)."""
    nw_max = max([r.shape[1] for r, _, _ in local.values() if r is not None] + [4])     # >= 5 columns for the descriptors
    total = sum(0 if r is None else r.shape[0] for r, _, _ in local.values())
    out = torch.zeros((s_max + total, nw_max + 1), dtype=torch.int64, device=device)
    desc = torch.full((s_max, 5), -1, dtype=torch.int64)
    at = s_max
    for k, (sid, (rows, conf, _cls)) in enumerate(local.items()):
        r = 0 if rows is None else rows.shape[0]
        code = 0
        if rows is not None and torch.is_tensor(conf):
            code = _CONF_CODES[conf.dtype]
        if r:
            out[at:at + r, :rows.shape[1]] = rows
            out[at:at + r, nw_max] = torch.as_tensor(conf).to(torch.float64).view(torch.int64).to(device)
        desc[k] = torch.tensor([scene_index[sid], r, 0 if rows is None else rows.shape[1], 1 if rows is None else 0, code])
        at += r
    out[:s_max, :5] = desc.to(device)
    return out


def unpack_class_results(mat, scene_ids, s_max, text_prompt):
    """Inverse of pack_class_results for one rank's matrix (host or device tensor)."""
    mat = mat.cpu()
    nw_max = mat.shape[1] - 1
    out, at = {}, s_max
    dtypes = {v: k for k, v in _CONF_CODES.items()}
    for k in range(s_max):
        idx, r, nw, is_list, code = (int(v) for v in mat[k, :5])
        if idx < 0:
            continue
        if is_list:
            out[scene_ids[idx]] = (None, [], [])
            continue
        rows = mat[at:at + r, :nw].contiguous()
        conf = mat[at:at + r, nw_max].contiguous().view(torch.float64).to(dtypes[code])
        out[scene_ids[idx]] = (rows, conf, [text_prompt] * r)
        at += r
    return out


def run_class(scenes, cfg, text_prompt: str, sim, device, weights: Sequence[float] = None):
    """One query class over many scenes on all ranks (the multi-GPU form of running
    tools/projection_2d_to_3d.py + tools/refinement.py for that class).

    scenes: list of SceneInputs-like objects (same list on every rank; only the rank's shard is touched).
    Rank r projects and refines the scenes `shard_scenes` gives it.  Two collectives, the same two `bench.py
    --gpus N` times: ONE all-gather (ClassExchange: every rank's similarity set -- the threshold is a percentile
    over the set of all scenes' similarities, refinement.py:316-324 -- and its bound on the rows it will deliver)
    and ONE gather of equally padded result matrices (bit rows + confidences + a few descriptor rows;
    pack_class_results).  No object collectives, no size exchange; the gathered header is read after the loop.
    Returns, on rank 0, {scene_id: (rows int64 [R][nw] or None, conf, final_class)} for ALL scenes; on the other
    ranks the dict of their own shard."""
    from .projection import projection_back, projection_front
    from .refinement import prepare_stage1, refine_class
    from .scene import prepare_scene
    rank, ws = world()
    ids = [s.scene_id for s in scenes]
    mine = shard_scenes(ids, weights=weights)
    s_max = max(1, -(-len(scenes) // ws))                  # most scenes any rank owns
    with_viewed = (not cfg.if_occurance_threshold) and bool(cfg.if_detected_ratio_threshold)
    # software pipeline over PIPELINE_DEPTH HIP streams: the device work of the next scenes is issued before the host
    # finishes scene k
    from .pipeline import PIPELINE_DEPTH as depth, scene_streams
    on_gpu = torch.device(device).type == "cuda"
    streams = scene_streams(device) if on_gpu else None      # the same streams (and workspaces) for every class

    def front(k):
        sc = scenes[mine[k]]
        ds = prepare_scene(sc, cfg, device=device, with_viewed=with_viewed)
        st1 = prepare_stage1(sc.stage1, device)
        if streams is None:
            return projection_front(ds, cfg, stage1=st1), st1
        streams[k % depth].wait_stream(torch.cuda.current_stream())   # the uploads ran on the current stream
        with torch.cuda.stream(streams[k % depth]):
            return projection_front(ds, cfg, stage1=st1), st1

    def back(k, fr):
        if streams is None:
            return projection_back(fr)
        with torch.cuda.stream(streams[k % depth]):
            res = projection_back(fr)
        torch.cuda.current_stream().wait_stream(streams[k % depth])   # results are used on the current stream
        return res

    trip, inflight, issued = [], [], 0
    for k, i in enumerate(mine):
        while issued < len(mine) and issued - k < depth:       # scenes k .. k + depth - 1 are on the device
            inflight.append(front(issued))
            issued += 1
        cur, st1 = inflight.pop(0)
        trip.append((scenes[i].scene_id, st1, back(k, cur)))
    exchange = ClassExchange(device) if ws > 1 else None
    final = refine_class(trip, cfg, text_prompt, sim, device, exchange_sims=exchange) if trip or ws > 1 else {}
    local = {sid: (r.rows, r.conf, list(r.final_class)) for sid, r in final.items()}
    if ws == 1:
        return local
    mat = pack_class_results(local, {sid: i for i, sid in enumerate(ids)}, s_max, device)
    r_max, w_max = exchange.bounds
    gathered = gather_final_rows(mat, bounds=(s_max + r_max, max(w_max, 4) + 1))
    if rank != 0:
        return local
    out = {}
    for g in gathered:                                      # the header is read here, after the loop
        out.update(unpack_class_results(g.rows(), ids, s_max, text_prompt))
    return out
