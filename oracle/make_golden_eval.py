"""Generate tests/golden/eval_assign.npz from the REAL reference evaluator.  TEST INFRASTRUCTURE ONLY.

Run in the build container (needs /root/reference):   python -m oracle.make_golden_eval

Imports evaluation/eval/scannetv2_inst_eval.py of the reference (stub modules stand in for the uninstalled cv2,
open3d and plyfile, none of which the method touches) and calls ScanNetEval.assign_instances_for_scan on seeded
synthetic scenes; only inputs and the flattened outputs are stored (oracle/eval_ref.flatten_assignment).
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.eval_ref import flatten_assignment  # noqa: E402

REFERENCE_ROOT = "/root/reference"


def load_reference_evaluator():
    sys.dont_write_bytecode = True
    for name in ("cv2", "open3d", "plyfile"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["plyfile"].PlyData = object
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    from evaluation.dataset.scannet200 import INSTANCE_CAT_SCANNET_200
    from evaluation.eval.scannetv2_inst_eval import ScanNetEval
    return ScanNetEval, list(INSTANCE_CAT_SCANNET_200)


def make_case(seed, n, n_pred, use_label):
    """A scene: blocky ground-truth instances over 12 classes (incl. ids the evaluator shifts below zero, ignored
    instance ids, void classes), predictions that overlap them partly / not at all / are too small / carry an
    unknown label."""
    rng = np.random.default_rng(seed)
    sem = np.zeros(n, np.int32)
    ins = np.full(n, -1, np.int32)
    at, k = 0, 0
    while at < n:
        ln = int(rng.integers(40, 900))
        sem[at:at + ln] = rng.choice([0, 1, 2, 3, 4, 7, 9, 30, 150, 199, 250])     # 0/1 -> clamped, 250 -> void class
        ins[at:at + ln] = k if rng.random() > 0.1 else -2                          # -2 -> ignored
        k += 1
        at += ln
    perm = rng.permutation(n)
    sem, ins = sem[perm], ins[perm]
    preds = []
    for p in range(n_pred):
        kind = p % 6
        mask = np.zeros(n, np.uint8)
        if kind == 0:                                   # most of one instance + noise
            target = int(rng.integers(0, k))
            mask[(ins == target) & (rng.random(n) < 0.8)] = 1
            mask[rng.random(n) < 0.01] = 1
        elif kind == 1:                                 # random points
            mask[rng.random(n) < 0.05] = 3              # any non-zero value counts
        elif kind == 2:                                 # too small (< 100 points)
            mask[rng.choice(n, 50, replace=False)] = 1
        elif kind == 3:                                 # two instances
            for target in rng.integers(0, k, 2):
                mask[ins == target] = 1
        elif kind == 4:                                 # empty
            pass
        else:                                           # exactly one instance
            mask[ins == int(rng.integers(0, k))] = 1
        label = float(rng.choice([1, 2, 3, 6, 8, 29, 149, 198, 400, 0]))           # 400 / 0: not a valid id -> skipped
        preds.append({"scan_id": f"scene{seed:04d}_00", "label_id": label, "conf": float(np.round(rng.random(), 3)),
                      "pred_mask": mask})
    return sem, ins, preds


def main():
    ScanNetEval, labels = load_reference_evaluator()
    out = {"class_labels": np.asarray(labels, dtype=str)}
    names = []
    for name, (seed, n, n_pred, use_label) in {"labelled_a": (1, 20_000, 24, True), "labelled_b": (2, 5_001, 12, True),
                                               "agnostic": (3, 12_345, 18, False), "no_preds": (4, 3_000, 0, True)}.items():
        sem, ins, preds = make_case(seed, n, n_pred, use_label)
        ev = ScanNetEval(labels, use_label=use_label, dataset_name="scannet200")
        gt2pred, pred2gt = ev.assign_instances_for_scan(preds, sem.copy(), ins.copy())
        flat = flatten_assignment(gt2pred, pred2gt, ev.eval_class_labels)
        names.append(name)
        out[f"{name}.sem"], out[f"{name}.ins"] = sem, ins
        out[f"{name}.use_label"] = np.array(use_label)
        out[f"{name}.pred_label"] = np.asarray([p["label_id"] for p in preds], dtype=np.float64)
        out[f"{name}.pred_conf"] = np.asarray([p["conf"] for p in preds], dtype=np.float64)
        out[f"{name}.pred_scan"] = np.asarray([p["scan_id"] for p in preds], dtype=str)
        out[f"{name}.pred_masks"] = np.packbits(np.stack([p["pred_mask"] != 0 for p in preds]), axis=-1, bitorder="little") \
            if preds else np.zeros((0, (n + 7) // 8), np.uint8)
        out[f"{name}.pred_values"] = np.asarray([int(p["pred_mask"].max()) if p["pred_mask"].any() else 0 for p in preds])
        for k, v in flat.items():
            out[f"{name}.out.{k}"] = v
        print(f"  {name}: {len(preds)} predictions -> {flat['pred'].shape[0]} kept, {flat['pred_matches'].shape[0]} matches, "
              f"{flat['gt'].shape[0]} gt instances")
    out["cases"] = np.asarray(names, dtype=str)
    path = os.path.join(ROOT, "tests", "golden", "eval_assign.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
