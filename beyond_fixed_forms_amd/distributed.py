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


def run_class(scenes, cfg, text_prompt: str, sim, device, weights: Sequence[float] = None):
    """One query class over many scenes on all ranks (the multi-GPU form of running
    tools/projection_2d_to_3d.py + tools/refinement.py for that class).

    scenes: list of SceneInputs-like objects (same list on every rank; only the rank's shard is touched).
    Rank r projects and refines the scenes `shard_scenes` gives it; the similarity sets are pooled across
    ranks before the threshold is taken (refinement.py:316-324), so every rank applies the threshold the
    single-process class loop would.  Returns, on rank 0, {scene_id: (rows int64 [R][nw] or None, conf,
    final_class)} for ALL scenes (bit rows gathered over RCCL, small metadata over the object channel);
    on the other ranks the dict of their own shard."""
    from .projection import projection_back, projection_front
    from .refinement import prepare_stage1, refine_class
    from .scene import prepare_scene
    rank, ws = world()
    mine = shard_scenes([s.scene_id for s in scenes], weights=weights)
    # two-stage software pipeline over two HIP streams: the GPU-only front half of scene k+1 (decode, sweep,
    # components) is issued before the host finishes the back half of scene k
    streams = [torch.cuda.Stream(device=device), torch.cuda.Stream(device=device)] if torch.device(device).type == "cuda" else None

    def front(k):
        ds = prepare_scene(scenes[mine[k]], cfg, device=device,
                           with_viewed=(not cfg.if_occurance_threshold) and bool(cfg.if_detected_ratio_threshold))
        if streams is None:
            return projection_front(ds, cfg)
        streams[k % 2].wait_stream(torch.cuda.current_stream())       # the upload ran on the current stream
        with torch.cuda.stream(streams[k % 2]):
            return projection_front(ds, cfg)

    def back(k, fr, st1):
        if streams is None:
            return projection_back(fr, stage1=st1)
        with torch.cuda.stream(streams[k % 2]):
            res = projection_back(fr, stage1=st1)
        torch.cuda.current_stream().wait_stream(streams[k % 2])       # results are used on the current stream
        return res

    trip = []
    nxt = front(0) if mine else None
    for k, i in enumerate(mine):
        cur = nxt
        if k + 1 < len(mine):
            nxt = front(k + 1)
        st1 = prepare_stage1(scenes[i].stage1, device)
        trip.append((scenes[i].scene_id, st1, back(k, cur, st1)))
    exchange = (lambda sims: exchange_similarities(sims, device=device)) if ws > 1 else None
    final = refine_class(trip, cfg, text_prompt, sim, device, exchange_sims=exchange) if trip or ws > 1 else {}
    local = {sid: (r.rows, r.conf, list(r.final_class)) for sid, r in final.items()}
    if ws == 1:
        return local
    # rows of all local scenes stacked into one [sum R][nw_max] buffer -> one gather; metadata as objects
    nw_max = max([r.shape[1] for r, _, _ in local.values() if r is not None] + [1])
    parts, meta = [], []
    for sid, (rows, conf, cls) in local.items():
        k = 0 if rows is None else rows.shape[0]
        if k:
            pad = torch.zeros((k, nw_max), dtype=torch.int64, device=device)
            pad[:, :rows.shape[1]] = rows
            parts.append(pad)
        meta.append((sid, k, None if rows is None else rows.shape[1],
                     conf if isinstance(conf, list) else conf.cpu(), cls, rows is None))
    stacked = torch.cat(parts) if parts else torch.zeros((0, nw_max), dtype=torch.int64, device=device)
    gathered = gather_final_rows(stacked)
    metas = [None] * ws
    dist.all_gather_object(metas, meta)
    if rank != 0:
        return local
    out = {}
    for r in range(ws):
        at = 0
        for sid, k, nw, conf, cls, is_list in metas[r]:
            rows = None if is_list else gathered[r][at:at + k, :nw].contiguous()
            at += k
            out[sid] = (rows, conf, cls)
    return out
