"""Stage 3 on MI355X: refine Open3DIS stage-1 masks with the stage-2 masks of one query class.

Host-side mirror of the reference's two-pass ``__main__`` block (tools/refinement.py:135-428,
`R:` below).  Bit rows stay on the device; the {0,1} matmuls (R:84) are popcount kernels, stage-1
RLE is decoded on the device (R:26-39), text similarities come from one MFMA cosine GEMM against an
embedding bank instead of two CLIP encoder calls per matched mask (R:93-115).  The short, order
dependent bookkeeping (R:230-281) stays on the host, as in the reference.
"""
from __future__ import annotations

import dataclasses
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np
import torch

from . import _lib
from .labels import SCANNET200_LABELS, idx_to_label
from .projection import Stage2Result
from .scene import runs_from_rles


# --------------------------------------------------------------------------- text similarity
class TextSimilarity:
    """cos(query, label) for every label the class loop can meet, computed once per query on the device
    (replaces compute_clip_similarity R:93-115, called twice per matched mask).

    `encode_text(str) -> (1, D) tensor` is the CLIP text encoder (or any stand-in).  The embeddings keep the
    ENCODER'S dtype -- float16 (CLIP on a GPU, what the reference's recorded run used) or float32 (CLIP on the
    CPU; anything else is widened to float32) -- and bff_cosine_rows rounds every tensor op of R:109-114 to that
    dtype, because the class threshold is an order statistic of the *set* of these values (R:321-324): with
    fp16 embeddings the cosines are multiples of 2^-11 and ties collapse, which moves the index.  Against the
    reference on the same embeddings the values agree to the last bit except where its float32 BLAS dot sits
    on a rounding boundary (<= 1 ulp of the dtype; tolerance of north_star: 1e-4)."""

    def __init__(self, encode_text: Callable[[str], torch.Tensor], device="cuda", labels: Sequence[str] = SCANNET200_LABELS):
        self.device = torch.device(device)
        self.labels = list(labels)
        with torch.no_grad():
            rows = [encode_text(lab).reshape(1, -1) for lab in self.labels]
        self.dtype = self._dtype_of(rows[0])
        self.bank = torch.cat(rows).to(self.dtype).to(self.device).contiguous()
        self.encode_text = encode_text
        self.index = {lab: i for i, lab in enumerate(self.labels)}
        self._cache: Dict[str, np.ndarray] = {}
        self._query_emb: Dict[str, torch.Tensor] = {}

    @staticmethod
    def _dtype_of(t):
        return torch.float16 if t.dtype == torch.float16 else torch.float32

    def save(self, path: str):
        """Persist the label bank and every query embedding seen so far (plain tensors + strings), so later
        class runs need no text encoder at all."""
        torch.save({"labels": self.labels, "bank": self.bank.cpu(), "queries": dict(self._query_emb)}, path)

    @classmethod
    def from_file(cls, path: str, device="cuda", encode_text: Optional[Callable[[str], torch.Tensor]] = None):
        """Rebuild the service from `save()` output (loaded with weights_only=True: tensors and strings only).
        Queries missing from the file need `encode_text`."""
        blob = torch.load(path, map_location="cpu", weights_only=True)
        self = cls.__new__(cls)
        self.device = torch.device(device)
        self.labels = list(blob["labels"])
        self.dtype = cls._dtype_of(blob["bank"])
        self.bank = blob["bank"].to(self.dtype).to(self.device).contiguous()
        self.index = {lab: i for i, lab in enumerate(self.labels)}
        self._cache = {}
        self._query_emb = dict(blob.get("queries", {}))

        def enc(text):
            if text in self._query_emb:
                return self._query_emb[text]
            if encode_text is None:
                raise KeyError(f"no embedding for {text!r} in {path} and no text encoder given")
            return encode_text(text)
        self.encode_text = enc
        return self

    def query(self, text: str) -> np.ndarray:
        """float32 [n_labels]: cosine of `text` against every bank label (values of the bank's dtype)."""
        if text not in self._cache:
            with torch.no_grad():
                emb = self.encode_text(text).reshape(1, -1).to(self.dtype)
                self._query_emb[text] = emb.cpu()
            self._cache[text] = _lib.cosine_rows(emb.to(self.device).contiguous(), self.bank)[0].cpu().numpy()
        return self._cache[text]

    def similarities(self, text: str, labels: Sequence[str]) -> List[float]:
        row = self.query(text)
        return [float(row[self.index[lab]]) for lab in labels]


# --------------------------------------------------------------------------- inputs
def _stage2_rows(stage2, n_points, device):
    """Stage-2 result (Stage2Result or the reference's dict) -> (bit rows [K][nw], conf (K,) on CPU)."""
    if isinstance(stage2, Stage2Result):
        return stage2.rows, stage2.conf_host if stage2.conf_host is not None else stage2.conf.cpu()
    conf = stage2["conf"]
    if len(conf) == 0:                                                              # R:196
        return torch.zeros((0, (n_points + 63) // 64), dtype=torch.int64, device=device), torch.as_tensor(conf).cpu()
    ins = stage2["ins"]
    if isinstance(ins, list):                       # stored as RLE dicts (Stage2Result.to_rle_dict / BFF_SAVE_RLE=1)
        rs, re, offs = runs_from_rles(ins, "stage-2")
        t = lambda a: torch.from_numpy(a).to(device)
        return _lib.rle_to_rows(t(rs), t(re), t(offs), n_points), torch.as_tensor(conf).cpu()
    ins = ins.to(device)
    if ins.dtype not in (torch.bool, torch.uint8):
        ins = ins != 0
    return _lib.pack_rows(ins.contiguous()), conf.cpu()


@dataclasses.dataclass
class DeviceStage1:
    """Open3DIS stage-1 result of one scene with its RLE run tables resident in HBM.  Stage-1 masks do
    not depend on the query class, so a scene's tables are uploaded once and decoded on the device
    (bff_rle_to_rows) for every class."""
    n_points: int
    run_start: torch.Tensor
    run_end: torch.Tensor
    row_run_offs: torch.Tensor
    labels: List[str]


def prepare_stage1(stage1: dict, device="cuda") -> DeviceStage1:
    """R:186-193: RLE "ins" -> run tables, class indices -> label strings."""
    rles = stage1["ins"]
    n_points = int(rles[0]["length"])
    if any(int(r["length"]) != n_points for r in rles):
        raise ValueError("stage-1 masks of different lengths")
    rs, re, offs = runs_from_rles(rles, "stage-1")
    t = lambda a: torch.from_numpy(a).to(device)
    return DeviceStage1(n_points, t(rs), t(re), t(offs), [idx_to_label(int(i)) for i in stage1["final_class"]])


def _stage1_rows(stage1, device):
    """stage-1 dict or DeviceStage1 -> (fresh bit rows [S1][nw], n_points, label strings)."""
    st = stage1 if isinstance(stage1, DeviceStage1) else prepare_stage1(stage1, device)
    rows = _lib.rle_to_rows(st.run_start, st.run_end, st.row_run_offs, st.n_points)
    return rows, st.n_points, st.labels


def _iou(inter: np.ndarray, area_a: np.ndarray, area_b: np.ndarray) -> np.ndarray:
    """calculate_iou_between_stages R:69-90 from integer intersections: (m, n) float32.  NumPy float32 arithmetic is
    the same IEEE arithmetic as the reference's float32 tensors (exact integers, one subtraction, one division)."""
    inter = inter.astype(np.float32)
    union = area_a.astype(np.float32)[:, None] + area_b.astype(np.float32)[None, :] - inter
    with np.errstate(invalid="ignore", divide="ignore"):
        return (inter / union).T


@dataclasses.dataclass
class _SceneState:
    scene_id: str
    n_points: int
    ious: object                 # float32 array (m,) or []
    sims: list
    matched1: Optional[torch.Tensor]   # bit rows (m, nw)
    stage2_rows: Optional[torch.Tensor]
    stage2_conf: object
    other1: torch.Tensor         # bit rows (o, nw)
    ready: object = None         # event on the stream pass 1 was issued on (None on the CPU)


def _pass1_scene(scene_id, stage1, stage2, cfg, text_prompt, query_us, sim: TextSimilarity, device) -> _SceneState:
    pre = stage2.prefetch if isinstance(stage2, Stage2Result) else None
    if pre is not None and pre["stage1"] is stage1:
        s1, n, s1_labels = pre["s1"], stage1.n_points, stage1.labels     # decoded by the projection stage, still untouched
        stage2.prefetch = None                                            # pass 1 edits s1 in place: single use
    else:
        pre = None
        s1, n, s1_labels = _stage1_rows(stage1, device)
    s2, conf2 = _stage2_rows(stage2, n, device)
    i32 = lambda lst: _lib.upload(np.asarray(lst, dtype=np.int32), torch.int32, device)
    if len(conf2) == 0:                                                             # R:196-205
        other = [i for i, lab in enumerate(s1_labels) if lab == query_us]
        return _SceneState(scene_id, n, [], [], None, None, [], _lib.gather_rows(s1, i32(other)) if other else s1[:0])

    # one read-back for everything pass 1 needs from the device: areas, stage-1 x stage-2 intersections, and
    # the stage-1 x stage-1 intersections (so the duplicate test R:217 needs no second round trip) -- or none at
    # all when the projection stage already fetched them (Stage2Result.prefetch).  The bookkeeping below runs on
    # NumPy arrays and python lists: K is tens of rows, a tensor op per element would cost more than the kernels.
    if pre is not None:
        area1, area2, inter, inter11 = (pre[k] for k in ("area1", "area2", "inter", "inter11"))
    else:
        area1, area2, inter, inter11 = _lib.fetch(_lib.popcount_rows(s1), _lib.popcount_rows(s2),
                                                  _lib.cross_popcount(s1, s2), _lib.cross_popcount(s1, s1))
    iou = _iou(inter, area1, area2)                                                 # R:208  (K, S1)
    best = np.argmax(iou, axis=1)                                                   # R:211 (first maximum, NaN counts as one)
    m_iou = _iou(inter11[best][:, best], area1[best], area1[best])                  # R:217
    k = len(best)
    np.fill_diagonal(m_iou, 0)                                                      # R:221
    m_adj = (m_iou > cfg.stage1_iou_thres).tolist()                                 # R:224 (float32 against the python scalar)
    best_l = best.tolist()

    chosen, ops = [], []                                                            # R:230-249
    absorbed_by = [-1] * k
    for i in range(k):
        if absorbed_by[i] != -1:
            chosen.append(best_l[absorbed_by[i]])
            continue
        chosen.append(best_l[i])
        row = m_adj[i]
        if any(row):
            for j in range(k):
                if row[j]:
                    absorbed_by[j] = i
                    ops.append((1, best_l[i], best_l[j]))                           # R:248 in-place OR
    if ops:
        _lib.apply_row_ops(s1, _lib.upload(np.asarray(ops, dtype=np.int32), torch.int32, device))

    # R:258-281: stage-2 masks matched to the same stage-1 mask are merged; each current row is
    # tracked as the list of original stage-2 rows it is the OR of
    parts = [[i] for i in range(k)]
    regrouped = False
    if len(set(chosen)) != k:                                                       # some stage-1 mask was chosen twice
        chosen_t = torch.tensor(chosen)
        uniq, cnt = torch.unique(chosen_t, return_counts=True)
        for u, c in zip(uniq, cnt):
            if c > 1:
                sel = chosen_t == u
                mconf = conf2[sel].mean()                                           # in the confidence dtype, as R:270
                merged = [p for keep, part in zip(sel.tolist(), parts) if keep for p in part]
                parts = [part for keep, part in zip(sel.tolist(), parts) if not keep] + [merged]
                conf2 = torch.cat([conf2[~sel], mconf.unsqueeze(0)])
                chosen_t = torch.cat([chosen_t[~sel], u.unsqueeze(0)])
        regrouped = True
        offs = np.zeros(len(parts) + 1, dtype=np.int32)
        np.cumsum([len(p) for p in parts], out=offs[1:])
        s2 = _lib.or_reduce_groups(s2, _lib.upload(offs, torch.int32, device),
                                   i32([p for part in parts for p in part]), max(len(p) for p in parts))

    if ops or regrouped:            # rows changed: recompute R:285-288; otherwise iou / best are what they were
        area1, area2, inter = _lib.fetch(_lib.popcount_rows(s1), _lib.popcount_rows(s2), _lib.cross_popcount(s1, s2))
        iou = _iou(inter, area1, area2)                                             # R:285
        best = np.argmax(iou, axis=1)                                               # R:288
        best_l = best.tolist()
    taken = set(best_l)
    other = [i for i, lab in enumerate(s1_labels) if lab == query_us and i not in taken]    # R:293
    labels = [s1_labels[i] for i in best_l]                                         # R:297
    sims = sim.similarities(text_prompt, labels)                                    # R:299-302
    picks = i32(best_l + other)                                                      # one upload for both selections
    return _SceneState(scene_id, n, iou[np.arange(len(best_l)), best], sims,
                       _lib.gather_rows(s1, picks[:len(best_l)]), s2, conf2,
                       _lib.gather_rows(s1, picks[len(best_l):]) if other else s1[:0])


@dataclasses.dataclass
class FinalResult:
    """Final output of one scene, bit-packed on the device; to_dict() = what R:354-357 / R:405-408 /
    R:423-426 save (including the reference's list-valued empty forms)."""
    scene_id: str
    n_points: int
    rows: Optional[torch.Tensor]      # int64 [R][nw] or None when the reference saves python lists
    conf: object
    final_class: List[str]

    def to_dict(self):
        if self.rows is None:
            return {"ins": [], "conf": [], "final_class": list(self.final_class)}
        return {"ins": _lib.unpack_rows(self.rows, self.n_points), "conf": self.conf,
                "final_class": list(self.final_class)}

    def to_rle_dict(self):
        """Same result with "ins" as RLE dicts (what eval_scannet200.py:123-124 also accepts), encoded on the device."""
        rows = self.rows if self.rows is not None else None
        return {"ins": [] if rows is None else _lib.rows_to_rle(rows, self.n_points), "conf": self.conf,
                "final_class": list(self.final_class)}


def sim_threshold(all_sims: Sequence[Sequence[float]], percentile: float) -> float:
    """R:321-324: sorted(set(all similarities of all scenes))[int(n * percentile)]."""
    uniq = sorted(set(s for sims in all_sims for s in sims))
    return uniq[int(len(uniq) * percentile)]


def _pass2_scene(st: _SceneState, cfg, text_prompt, sim_thres) -> FinalResult:
    """R:330-428 for one scene."""
    dev = st.other1.device
    pieces, cls = [], []
    n_other = st.other1.shape[0]
    if n_other:                                                                     # R:340-343
        pieces.append(st.other1)
        cls += [text_prompt] * n_other
    half = [torch.tensor(0.5)] * n_other
    if len(st.ious) == 0:                                                           # R:348-358
        if n_other == 0:
            return FinalResult(st.scene_id, st.n_points, None, [], [])
        return FinalResult(st.scene_id, st.n_points, st.other1, torch.stack(half), cls)
    order, kept = [], []                        # (source, index) in output order; stage-2 rows whose confidence is kept
    thr = np.float32(cfg.refiment_iou_thres)    # float32 tensor element against the python scalar (R:362)
    sims = st.sims
    for m, v in enumerate(st.ious.tolist() if hasattr(st.ious, "tolist") else st.ious):    # R:360-392
        if np.float32(v) > thr:
            if sims[m] < sim_thres:
                continue
            order.append((1, m))
        else:
            order.append((2, m))
        kept.append(m)
        cls.append(text_prompt)
    if n_other == 0 and not order:                                                  # R:402-409
        return FinalResult(st.scene_id, st.n_points, None, [], [])
    if order:
        both = torch.cat([st.matched1, st.stage2_rows])
        k = st.matched1.shape[0]
        idx = _lib.upload(np.asarray([m if src == 1 else k + m for src, m in order], dtype=np.int32), torch.int32, dev)
        pieces.append(_lib.gather_rows(both, idx))
    rows = torch.cat(pieces) if len(pieces) > 1 else pieces[0]
    # R:412 torch.stack of 0-d tensors: the 0.5 of the "other" masks is float32, the rest keeps the stage-2 dtype;
    # stack of mixed dtypes promotes, exactly what cat of the two typed vectors does
    conf_kept = st.stage2_conf[torch.as_tensor(kept, dtype=torch.long)] if kept else st.stage2_conf[:0]
    conf = torch.cat([torch.stack(half), conf_kept]) if n_other else conf_kept
    return FinalResult(st.scene_id, st.n_points, rows, conf, cls)                   # R:411-412


class ClassRefiner:
    """The two passes of R:135-428 for one query class as an object, so that a caller can run pass 1 scene by scene as
    the projection results arrive (`add`, R:166-312) and the class-wide part -- ONE similarity threshold over all scenes
    of the class (R:316-324), then pass 2 (R:330-428) -- when the class batch is complete (`finish`).
    `exchange_sims` widens the similarity set to all ranks (the one cross-scene dependency of the path)."""

    def __init__(self, cfg, text_prompt: str, sim: TextSimilarity, device="cuda", exchange_sims=None):
        self.cfg, self.text_prompt, self.sim, self.device = cfg, text_prompt, sim, device
        self.query_us = text_prompt.replace(" ", "_")                               # R:142
        self.exchange_sims = exchange_sims
        self.order = []                     # scene ids in the order given (R:154), including the skipped ones
        self.states = []                    # pass-1 states of the scenes that have both files (R:175-178)
        self.sim_thres = None

    def add(self, scene_id, stage1, stage2):
        """Pass 1 of one scene (device work is enqueued on the current stream)."""
        self.order.append(scene_id)
        if stage1 is None or stage2 is None:
            return None
        with _lib.launch_stream():
            st = _pass1_scene(scene_id, stage1, stage2, self.cfg, self.text_prompt, self.query_us, self.sim, self.device)
        if st.other1.is_cuda:               # pass 2 may run on another stream (scenes are issued round-robin on several)
            st.ready = torch.cuda.Event()
            st.ready.record(torch.cuda.current_stream(st.other1.device))
        self.states.append(st)
        return st

    def bounds(self):
        """What this rank can at most deliver: matched stage-2 rows + other stage-1 masks, widest row."""
        rows = sum(len(st.ious) + st.other1.shape[0] for st in self.states)
        words = max([st.other1.shape[1] for st in self.states] + [1])
        return rows, words

    def finish(self):
        with _lib.launch_stream():
            return self._finish()

    def _finish(self):
        all_sims = [st.sims for st in self.states]
        ex = self.exchange_sims
        if ex is not None and getattr(ex, "takes_bounds", False):
            pool = ex(all_sims, bounds=self.bounds())
        else:
            pool = ex(all_sims) if ex is not None else all_sims
        self.sim_thres = thres = sim_threshold(pool, self.cfg.refinment_sim_percentile)      # R:321-324
        out = {}
        for s, scene_id in enumerate(self.order):                                   # R:330 (index s, as the reference)
            st = self.states[s]
            if st.ready is not None:        # rows of pass 1 were produced on that scene's stream
                cur = torch.cuda.current_stream(st.other1.device)
                cur.wait_event(st.ready)
                for t in (st.matched1, st.stage2_rows, st.other1):
                    if torch.is_tensor(t) and t.is_cuda:
                        t.record_stream(cur)
            res = _pass2_scene(st, self.cfg, self.text_prompt, thres)
            res.scene_id = scene_id
            out[scene_id] = res
        return out


def refine_class(scenes, cfg, text_prompt: str, sim: TextSimilarity, device="cuda",
                 exchange_sims: Optional[Callable[[List[List[float]]], List[List[float]]]] = None,
                 return_debug: bool = False):
    """Reference R:135-428 on in-memory inputs.

    scenes: list of (scene_id, stage1 dict / DeviceStage1 / None, stage2) in the sorted order of the stage-2
    directory listing (R:154); stage2 is a Stage2Result, the reference's saved dict, or None for a
    missing file (R:175-178: such scenes are skipped in pass 1).  `exchange_sims` lets the
    multi-GPU driver widen the similarity set of pass 1 to all ranks (the one cross-scene
    dependency of the path, R:316-324).  Returns {scene_id: FinalResult}."""
    _lib.load()
    with _lib.launch_stream():
        return _refine_class(scenes, cfg, text_prompt, sim, device, exchange_sims, return_debug)


def _refine_class(scenes, cfg, text_prompt, sim, device, exchange_sims, return_debug):
    ref = ClassRefiner(cfg, text_prompt, sim, device, exchange_sims)
    for scene_id, stage1, stage2 in scenes:                                         # R:166
        ref.add(scene_id, stage1, stage2)
    out = ref.finish()
    if return_debug:
        return out, {"sim_thres": ref.sim_thres, "states": ref.states}
    return out
