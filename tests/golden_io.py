"""(De)serialisation of golden fixtures: reference-format objects <-> flat npz arrays.

Fixtures are *data* (inputs and expected outputs) produced by oracle/make_golden.py from the
real reference helpers; this module only packs them.  No reference source is stored.
"""
from __future__ import annotations

import json
import os

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def pack_rles(rles):
    """list of {"length","counts"} -> (lengths int64[n], counts int64[total], offsets int64[n+1])."""
    lengths = np.array([r["length"] for r in rles], dtype=np.int64)
    cs = [np.asarray(r["counts"], dtype=np.int64).reshape(-1) for r in rles]
    offs = np.zeros(len(rles) + 1, dtype=np.int64)
    if cs:
        offs[1:] = np.cumsum([c.size for c in cs])
    counts = np.concatenate(cs) if cs else np.zeros(0, dtype=np.int64)
    return lengths, counts, offs


def unpack_rles(lengths, counts, offs):
    return [dict(length=int(lengths[i]), counts=np.asarray(counts[offs[i]:offs[i + 1]], dtype=np.int64))
            for i in range(len(lengths))]


def pack_bool_rows(rows) -> np.ndarray:
    """bool (R,N) -> uint8 (R, ceil(N/8)) little-endian bit order (bit k of byte j = column 8j+k)."""
    rows = np.asarray(rows, dtype=bool)
    return np.packbits(rows, axis=-1, bitorder="little")


def unpack_bool_rows(packed, n) -> np.ndarray:
    return np.unpackbits(packed, axis=-1, count=n, bitorder="little").astype(bool)


def scene_to_arrays(scene) -> dict:
    """SceneInputs -> dict of arrays (depth stored as uint16 millimetres: the synthetic depth
    is exactly float32(mm)/1000, like a decoded 16-bit PNG, reference :432-435)."""
    fids = sorted(scene.poses.keys(), key=int)
    depth_mm = np.stack([np.round(scene.depths[f].astype(np.float64) * 1000.0).astype(np.uint16) for f in fids])
    for k, f in enumerate(fids):   # the mm representation must be lossless
        assert np.array_equal(depth_mm[k].astype(np.float32) / np.float32(1000), scene.depths[f])
    out = {
        "scene_id": np.array(scene.scene_id),
        "xyz": np.ascontiguousarray(scene.points[:, :3]),
        "cam_intr": scene.cam_intr,
        "frame_ids": np.array(fids),
        "poses": np.stack([scene.poses[f] for f in fids]),
        "depth_mm": depth_mm,
        "color_files": np.array(scene.color_files),
        "mask_frame_ids": np.array([fr["frame_id"] for fr in scene.mask_2d]),
        "mask_m": np.array([len(fr["segmented_frame_masks"]) for fr in scene.mask_2d], dtype=np.int64),
        "mask_conf": np.concatenate([fr["confidences"].to(torch.float32).numpy() for fr in scene.mask_2d])
        if scene.mask_2d else np.zeros(0, np.float32),   # f16 -> f32 is lossless; dtype kept below
        "mask_conf_dtype": np.array(str(scene.mask_2d[0]["confidences"].dtype) if scene.mask_2d else "torch.float16"),
        "mask_labels": np.array([l for fr in scene.mask_2d for l in fr["labels"]]),
    }
    allr = [r for fr in scene.mask_2d for r in fr["segmented_frame_masks"]]
    out["mask_len"], out["mask_counts"], out["mask_offs"] = pack_rles(allr)
    if scene.stage1 is not None:
        out["s1_len"], out["s1_counts"], out["s1_offs"] = pack_rles(scene.stage1["ins"])
        out["s1_conf"] = scene.stage1["conf"].numpy()
        out["s1_class"] = np.array(scene.stage1["final_class"], dtype=np.int64)
    return out


def scene_from_arrays(z, prefix=""):
    from beyond_fixed_forms_amd.synthetic import SceneInputs
    g = lambda k: z[prefix + k]
    fids = [str(f) for f in g("frame_ids")]
    xyz = g("xyz")
    points = np.concatenate([xyz, np.zeros_like(xyz)], axis=1)
    depth_mm = g("depth_mm")
    depths = {f: depth_mm[k].astype(np.float32) / np.float32(1000) for k, f in enumerate(fids)}
    poses = {f: g("poses")[k] for k, f in enumerate(fids)}
    rles = unpack_rles(g("mask_len"), g("mask_counts"), g("mask_offs"))
    conf = torch.from_numpy(g("mask_conf").copy())
    if str(g("mask_conf_dtype")) == "torch.float16":
        conf = conf.half()
    labels = [str(s) for s in g("mask_labels")]
    mask_2d, k = [], 0
    for fid, m in zip(g("mask_frame_ids"), g("mask_m")):
        m = int(m)
        mask_2d.append({"frame_id": str(fid), "segmented_frame_masks": rles[k:k + m],
                        "confidences": conf[k:k + m].clone(), "labels": labels[k:k + m]})
        k += m
    stage1 = None
    if prefix + "s1_len" in z:
        stage1 = {"ins": unpack_rles(g("s1_len"), g("s1_counts"), g("s1_offs")),
                  "conf": torch.from_numpy(g("s1_conf").copy()),
                  "final_class": [int(c) for c in g("s1_class")]}
    h, w = depth_mm.shape[1:]
    return SceneInputs(scene_id=str(g("scene_id")), points=points, cam_intr=g("cam_intr"), poses=poses,
                       depths=depths, mask_2d=mask_2d, color_files=[str(c) for c in g("color_files")],
                       stage1=stage1, height=int(h), width=int(w))


def result_to_arrays(res: dict, n: int) -> dict:
    """{"ins","conf","final_class"} -> arrays; handles the reference's empty forms
    (ins = tensor([[]]) f32 (1,0), or python lists)."""
    ins = res["ins"]
    if isinstance(ins, list):
        kind, packed, k = "list", np.zeros((0, (n + 7) // 8), np.uint8), 0
    elif ins.numel() == 0:
        kind, packed, k = "empty_tensor", np.zeros((0, (n + 7) // 8), np.uint8), 0
    else:
        kind, packed, k = "rows", pack_bool_rows(ins.cpu().numpy().astype(bool)), ins.shape[0]
    conf = res["conf"]
    conf_dtype = "list" if isinstance(conf, list) else str(conf.dtype)
    conf_arr = np.zeros(0, np.float32) if isinstance(conf, list) else conf.detach().cpu().float().numpy()
    return {"kind": np.array(kind), "ins_packed": packed, "k": np.array(k), "n": np.array(n),
            "conf": conf_arr, "conf_dtype": np.array(conf_dtype),
            "final_class": np.array([str(c) for c in res["final_class"]], dtype=str)}


def dumps_groups(groups) -> np.ndarray:
    return np.array(json.dumps(groups))


def loads_groups(arr):
    return json.loads(str(arr))


def eval_case(z, name):
    """One case of tests/golden/eval_assign.npz -> (preds, gts_sem, gts_ins, use_label, expected flat arrays)."""
    sem, ins = z[f"{name}.sem"], z[f"{name}.ins"]
    n = sem.shape[0]
    masks = np.unpackbits(z[f"{name}.pred_masks"], axis=-1, count=n, bitorder="little")
    preds = [{"scan_id": str(s), "label_id": float(l), "conf": float(c), "pred_mask": (m * v).astype(np.uint8)}
             for s, l, c, m, v in zip(z[f"{name}.pred_scan"], z[f"{name}.pred_label"], z[f"{name}.pred_conf"], masks,
                                      z[f"{name}.pred_values"])]
    exp = {k.split(".out.")[1]: z[k] for k in z.files if k.startswith(f"{name}.out.")}
    return preds, sem, ins, bool(z[f"{name}.use_label"]), exp


def same_assignment(got: dict, exp: dict):
    for k, v in exp.items():
        if v.dtype.kind == "f":
            assert np.array_equal(got[k], v), k                 # ious are float(int) / int: identical everywhere
        elif v.dtype.kind == "U":
            assert list(got[k]) == list(v), k
        else:
            assert np.array_equal(got[k], v), k
