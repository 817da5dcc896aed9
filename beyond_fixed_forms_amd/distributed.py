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

import os

MAX_SIMS = 256          # a class meets at most one similarity per ScanNet200 label (198) per query


def _collectives(ws):
    """Does a class exchange / gather anything?  With several ranks: yes.  BFF_FORCE_COLLECTIVES=1 also with ONE rank in an
    initialised process group -- the way a single-GPU box executes the RCCL branch (stream of its own, pinned staging,
    all_gather_into_tensor, gather) that otherwise first runs on the 8-GPU node (tests/test_gpu_scene.py)."""
    return ws > 1 or (os.environ.get("BFF_FORCE_COLLECTIVES") == "1" and dist.is_available() and dist.is_initialized())


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
    if not _collectives(ws):
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
    """The one exchange of a class as a callable for `ClassRefiner(exchange_sims=...)`: a single all-gather
    carries every rank's similarity set AND its bound on the final rows it will produce (known after pass 1:
    matched stage-2 rows + other stage-1 masks of the query label, R:293, 340-392), so that the later gather of
    the results (`gather_final_rows(rows, bounds=exchange.bounds)`) needs no size exchange and no host
    synchronisation of its own.  It is called ONCE per class (the reference picks one threshold per class,
    R:316-324) and is the only point of a class at which the host waits for other ranks.  Over RCCL the 2 KB
    message travels on a stream of its own (pinned staging both ways), so the wait never includes the scene
    kernels queued on the compute streams."""
    takes_bounds = True

    def __init__(self, device="cpu"):
        self.device = device
        self.bounds = None                      # (max rows, max words) over all ranks after a call
        self.calls = 0
        self._comm = None                       # (stream, pinned send, pinned receive, device send, device receive)

    def _rccl_buffers(self, ws):
        if self._comm is None or self._comm[2].shape[0] != ws:
            dev = torch.device(self.device)
            self._comm = (torch.cuda.Stream(device=dev),
                          torch.zeros(MAX_SIMS + 3, dtype=torch.float64).pin_memory(),
                          torch.zeros((ws, MAX_SIMS + 3), dtype=torch.float64).pin_memory(),
                          torch.zeros(MAX_SIMS + 3, dtype=torch.float64, device=dev),
                          torch.zeros((ws, MAX_SIMS + 3), dtype=torch.float64, device=dev))
        return self._comm

    def __call__(self, local_sims, bounds=(0, 1)):
        rank, ws = world()
        self.calls += 1
        if not _collectives(ws):
            self.bounds = (int(bounds[0]), int(bounds[1]))
            return local_sims
        uniq = sorted(set(s for sims in local_sims for s in sims))
        if len(uniq) > MAX_SIMS:
            raise ValueError(f"{len(uniq)} distinct similarities on one rank (> {MAX_SIMS})")
        if dist.get_backend() == "nccl":
            st, send_h, recv_h, send_d, recv_d = self._rccl_buffers(ws)
            host = send_h
        else:
            host = torch.zeros(MAX_SIMS + 3, dtype=torch.float64)
        host.zero_()
        host[0], host[1], host[2] = len(uniq), float(bounds[0]), float(bounds[1])
        if uniq:
            host[3:3 + len(uniq)] = torch.tensor(uniq, dtype=torch.float64)
        if dist.get_backend() == "nccl":
            with torch.cuda.stream(st):
                send_d.copy_(send_h, non_blocking=True)
                dist.all_gather_into_tensor(recv_d, send_d)
                recv_h.copy_(recv_d, non_blocking=True)
            st.synchronize()                    # this stream only: the scene streams keep running
            out = recv_h
        else:
            got = [torch.empty_like(host) for _ in range(ws)]
            dist.all_gather(got, host)
            out = torch.stack(got)
        self.bounds = (max(int(o[1]) for o in out), max(1, max(int(o[2]) for o in out)))
        return [o[3:3 + int(o[0])].tolist() for o in out]


def gather_final_rows(rows: torch.Tensor, dst: int = 0, bounds=None):
    """Gather every rank's final bit rows (int64 [R][nw], R and nw may differ per rank) on `dst`.
    Returns, on dst, a list with one int64 [R_r][nw_r] tensor per rank; elsewhere None.

    bounds = (max rows, max words) agreed beforehand (ClassExchange.bounds): ONE collective and no host
    synchronisation -- every rank sends a [1 + max rows][max words] buffer whose first row holds its real
    (R, nw); dst gets the padded buffers back as `PaddedRows` (decoded on demand)."""
    rank, ws = world()
    if not _collectives(ws):
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
_META_FIXED = 4            # meta words before the scene descriptors: header rows, s_max, total rows, reserved
_DESC = 6                  # words per scene slot


def packed_header_rows(s_max, r_cap, width):
    """Leading rows of a packed class result: 4 fixed words + 6 per scene slot + one confidence per result row."""
    return -(-(_META_FIXED + _DESC * s_max + r_cap) // width)


def pack_class_results(final, scene_index, s_max, device, bounds=None):
    """{scene_id: refinement.FinalResult} of this rank -> ONE int64 matrix [H + sum R][W] for a single gather.
    The H leading rows are host-built metadata, uploaded in one copy: H, s_max, sum R; then per scene slot (global
    scene index, R, nw, saved-as-lists flag, confidence dtype code, point count N; unused slots hold -1); then the
    confidence (float64 bits) of every result row.  The bit rows of all scenes follow, each padded to W words.
    final_class needs no transport: every row carries the query (R:343, 390).
    bounds = (row bound, word bound) agreed by all ranks (ClassExchange.bounds) fixes H and W for everybody."""
    import numpy as np
    nw_max = max([f.rows.shape[1] for f in final.values() if f.rows is not None] + [1])
    total = sum(0 if f.rows is None else f.rows.shape[0] for f in final.values())
    r_cap, width = (total, nw_max) if bounds is None else (int(bounds[0]), int(bounds[1]))
    width = max(width, 8)
    if total > r_cap or nw_max > width or len(final) > s_max:
        raise ValueError(f"results ({len(final)} scenes, {total} rows, {nw_max} words) exceed the agreed bounds")
    h = packed_header_rows(s_max, r_cap, width)
    meta = np.full(h * width, -1, dtype=np.int64)
    meta[:_META_FIXED] = (h, s_max, total, 0)
    at, pieces = 0, []
    for k, (sid, f) in enumerate(final.items()):
        rows, conf = f.rows, f.conf
        r = 0 if rows is None else rows.shape[0]
        code = _CONF_CODES[conf.dtype] if (rows is not None and torch.is_tensor(conf)) else 0
        meta[_META_FIXED + _DESC * k:_META_FIXED + _DESC * (k + 1)] = (
            scene_index[sid], r, 0 if rows is None else rows.shape[1], 1 if rows is None else 0, code, f.n_points)
        if r:
            c0 = _META_FIXED + _DESC * s_max + at
            meta[c0:c0 + r] = torch.as_tensor(conf).detach().cpu().to(torch.float64).view(torch.int64).numpy()
            pieces.append((at, rows))
        at += r
    out = torch.zeros((h + total, width), dtype=torch.int64, device=device)
    head = torch.from_numpy(meta)
    if torch.device(device).type == "cuda":
        head = head.pin_memory()
    out[:h].view(-1).copy_(head, non_blocking=True)
    for at, rows in pieces:
        out[h + at:h + at + rows.shape[0], :rows.shape[1]] = rows
    return out


def unpack_class_results(mat, scene_ids, s_max, text_prompt, device="cpu"):
    """Inverse of pack_class_results for one rank's matrix: {scene_id: FinalResult} with the bit rows on `device`."""
    from .refinement import FinalResult
    host = mat.cpu()
    flat = host.reshape(-1)
    h, s_got, total = (int(v) for v in flat[:3])
    if s_got != s_max:
        raise ValueError(f"packed result with {s_got} scene slots, expected {s_max}")
    out, at = {}, 0
    dtypes = {v: k for k, v in _CONF_CODES.items()}
    conf0 = _META_FIXED + _DESC * s_max
    for k in range(s_max):
        idx, r, nw, is_list, code, n_points = (int(v) for v in flat[_META_FIXED + _DESC * k:_META_FIXED + _DESC * (k + 1)])
        if idx < 0:
            continue
        sid = scene_ids[idx]
        if is_list:
            out[sid] = FinalResult(sid, n_points, None, [], [])
            continue
        rows = mat[h + at:h + at + r, :nw].to(device).contiguous()
        conf = flat[conf0 + at:conf0 + at + r].contiguous().view(torch.float64).to(dtypes[code])
        out[sid] = FinalResult(sid, n_points, rows, conf, [text_prompt] * r)
        at += r
    return out


class ClassBatch:
    """One query class over this rank's scenes: pass 1 of the refinement scene by scene as the projection results
    arrive (`add`), then `finish()`: ONE all-gather (ClassExchange: every rank's similarity set -- the threshold is a
    percentile over the set of all scenes' similarities, refinement.py:316-324 -- and its bound on the rows it will
    deliver), pass 2, ONE gather of equally padded result matrices to rank 0 (bit rows + confidences + a few
    descriptor words; pack_class_results).  No object collectives, no size exchange, no per-scene host wait; the gathered
    buffers are decoded by `results()` whenever the caller wants them (bench.py: after the timed loop).

    all_ids: the scene ids of the WHOLE class in listing order, the same list on every rank; s_max: most scenes any
    rank owns."""

    def __init__(self, cfg, text_prompt, sim, device, all_ids, s_max, exchange: "ClassExchange" = None):
        from .refinement import ClassRefiner
        rank, ws = world()
        self.rank, self.ws = rank, ws
        self.device, self.text_prompt = device, text_prompt
        self.ids, self.s_max = list(all_ids), int(s_max)
        self.index = {sid: i for i, sid in enumerate(self.ids)}
        self.multi = _collectives(ws)
        self.exchange = exchange if exchange is not None else (ClassExchange(device) if self.multi else None)
        self.refiner = ClassRefiner(cfg, text_prompt, sim, device, exchange_sims=self.exchange)
        self.final = None
        self.gathered = None

    def add(self, scene_id, stage1, stage2):
        return self.refiner.add(scene_id, stage1, stage2)

    def finish(self):
        final = self.refiner.finish() if (self.refiner.order or self.multi) else {}
        self.final = final
        if self.multi:
            r_max, w_max = self.exchange.bounds
            width = max(w_max, 8)
            mat = pack_class_results(final, self.index, self.s_max, self.device, bounds=(r_max, width))
            # every rank sends the same shape: header rows + the largest row count any rank delivers
            self.gathered = gather_final_rows(mat, bounds=(packed_header_rows(self.s_max, r_max, width) + r_max, width))
        return self

    def results(self):
        """rank 0: {scene_id: FinalResult} for ALL scenes of the class; other ranks: their own shard."""
        if not self.multi or self.rank != 0:
            return self.final
        out = {}
        for g in self.gathered:                                 # the gathered headers are read here, not in the loop
            out.update(unpack_class_results(g.rows(), self.ids, self.s_max, self.text_prompt, self.device))
        return out


def run_class(scenes, cfg, text_prompt: str, sim, device, weights: Sequence[float] = None, n_loaders: int = 2,
              ids: Sequence[str] = None):
    """One query class over many scenes on all ranks (the multi-GPU form of running
    tools/projection_2d_to_3d.py + tools/refinement.py for that class).

    scenes: list of SceneInputs-like objects, or of zero-argument callables that load one (same list on every rank;
    only the rank's shard is touched).  Rank r ingests (loader threads with their own streams, ingest.Ingestor),
    projects (PIPELINE_DEPTH scenes in flight) and refines the scenes `shard_scenes` gives it; two collectives in
    all (ClassBatch).  Returns, on rank 0, {scene_id: refinement.FinalResult} for ALL scenes; on the other ranks the
    dict of their own shard."""
    from .pipeline import project_stream
    rank, ws = world()
    ids = list(ids) if ids is not None else [s.scene_id for s in scenes]
    mine = shard_scenes(ids, weights=weights)
    s_max = max(1, -(-len(scenes) // ws))                  # most scenes any rank owns
    batch = ClassBatch(cfg, text_prompt, sim, device, ids, s_max)
    project_stream([scenes[i] for i in mine], cfg, device, lambda k, st1, res: batch.add(ids[mine[k]], st1, res),
                   n_loaders=n_loaders)
    return batch.finish().results()
