"""CPU restatement of the reference's prediction <-> ground-truth assignment.  TEST INFRASTRUCTURE ONLY.

Follows /root/reference/evaluation/eval/scannetv2_inst_eval.py:265-365 (`ScanNetEval.assign_instances_for_scan`) and
evaluation/eval/instance_eval_util.py:70-174 (`Instance`, `get_instances`), NumPy only.  Pinned by
tests/golden/eval_assign.npz, which oracle/make_golden_eval.py wrote by calling the reference's own method.
"""
from __future__ import annotations

from copy import deepcopy

import numpy as np


def get_instances_ref(ids, class_ids, class_labels, id2label, encode=1000):
    """instance_eval_util.py:158-174."""
    instances = {label: [] for label in class_labels}
    for iid in np.unique(ids):
        if iid == 0:
            continue
        label_id = int(iid // encode)                                    # Instance.get_label_id :96
        vert_count = int((ids == iid).sum())                             # :99
        if label_id in class_ids:
            instances[id2label[label_id]].append({"instance_id": int(iid), "label_id": label_id, "vert_count": vert_count,
                                                  "med_dist": -1, "dist_conf": 0.0, "box": np.zeros((6))})
    return instances


def assign_instances_ref(preds, gts_sem, gts_ins, class_labels, use_label=True, dataset_name="scannet200"):
    """scannetv2_inst_eval.py:265-365."""
    encode = 10000 if dataset_name == "scannetpp" else 1000              # :23-27
    valid_class_ids = np.arange(len(class_labels)) + 1                   # :30
    id2label = {valid_class_ids[i]: class_labels[i] for i in range(len(class_labels))}
    eval_class_labels = list(class_labels) if use_label else ["class_agnostic"]
    min_region = 10 if dataset_name == "stpls3d" else 100                # :42-45
    offs = {"scannetv2": 2, "scannet200": 2, "replica": 1, "stpls3d": 1, "scannetpp": 105}.get(dataset_name, 0)
    gts_sem = np.array(gts_sem) - offs + 1                               # :270-281
    gts_sem[gts_sem < 0] = 0
    gts_ins = np.array(gts_ins) + 1
    ignore = gts_ins < 0
    gts = gts_sem * encode + gts_ins                                     # :288
    gts[ignore] = 0
    gt_instances = get_instances_ref(gts, valid_class_ids, class_labels, id2label, encode)
    if use_label:
        gt2pred = deepcopy(gt_instances)
        for label in gt2pred:
            for gt in gt2pred[label]:
                gt["matched_pred"] = []
    else:
        agn = []
        for _, instances in gt_instances.items():
            agn += deepcopy(instances)
        for gt in agn:
            gt["matched_pred"] = []
        gt2pred = {eval_class_labels[0]: agn}
    pred2gt = {label: [] for label in eval_class_labels}
    num_pred_instances = 0
    bool_void = np.logical_not(np.in1d(gts // encode, valid_class_ids))  # :306
    for pred in preds:
        if use_label:
            label_id = pred["label_id"]
            if label_id not in id2label:
                continue
            label_name = id2label[label_id]
        else:
            label_name = eval_class_labels[0]
        pred_mask = np.not_equal(pred["pred_mask"], 0)
        num = np.count_nonzero(pred_mask)
        if num < min_region:
            continue
        inst = {"filename": "{}_{}".format(pred["scan_id"], num_pred_instances), "pred_id": num_pred_instances,
                "label_id": pred["label_id"] if use_label else None, "vert_count": num, "confidence": pred["conf"],
                "void_intersection": np.count_nonzero(np.logical_and(bool_void, pred_mask))}
        matched_gt = []
        for gt_num, gt_inst in enumerate(gt2pred[label_name]):
            intersection = np.count_nonzero(np.logical_and(gts == gt_inst["instance_id"], pred_mask))
            if intersection > 0:
                gt_copy, pred_copy = gt_inst.copy(), inst.copy()
                gt_copy["intersection"] = pred_copy["intersection"] = intersection
                iou = float(intersection) / (gt_copy["vert_count"] + pred_copy["vert_count"] - intersection)
                gt_copy["iou"] = pred_copy["iou"] = iou
                matched_gt.append(gt_copy)
                gt2pred[label_name][gt_num]["matched_pred"].append(pred_copy)
        inst["matched_gt"] = matched_gt
        num_pred_instances += 1
        pred2gt[label_name].append(inst)
    return gt2pred, pred2gt


def flatten_assignment(gt2pred, pred2gt, labels):
    """The nested dicts -> flat arrays (what the golden file stores and every implementation is compared on):
    per label in `labels` order, every prediction with its matched ground-truth instances and every ground-truth
    instance with its matched predictions, in list order."""
    pred_rows, pm_rows, gt_rows, gm_rows, filenames = [], [], [], [], []
    for li, lab in enumerate(labels):
        for p in pred2gt.get(lab, []):
            pred_rows.append([li, p["pred_id"], -1 if p["label_id"] is None else int(p["label_id"]), p["vert_count"],
                              p["void_intersection"], len(p["matched_gt"])])
            filenames.append(p["filename"])
            for g in p["matched_gt"]:
                pm_rows.append([li, p["pred_id"], g["instance_id"], g["label_id"], g["vert_count"], g["intersection"]])
        for gi, g in enumerate(gt2pred.get(lab, [])):
            gt_rows.append([li, gi, g["instance_id"], g["label_id"], g["vert_count"], len(g["matched_pred"])])
            for p in g["matched_pred"]:
                gm_rows.append([li, gi, p["pred_id"], p["vert_count"], p["void_intersection"], p["intersection"]])
    i64 = lambda rows, w: np.asarray(rows, dtype=np.int64).reshape(-1, w)
    ious_p = np.asarray([g["iou"] for lab in labels for p in pred2gt.get(lab, []) for g in p["matched_gt"]], dtype=np.float64)
    ious_g = np.asarray([p["iou"] for lab in labels for g in gt2pred.get(lab, []) for p in g["matched_pred"]], dtype=np.float64)
    confs = np.asarray([float(p["confidence"]) for lab in labels for p in pred2gt.get(lab, [])], dtype=np.float64)
    return {"pred": i64(pred_rows, 6), "pred_matches": i64(pm_rows, 6), "gt": i64(gt_rows, 6), "gt_matches": i64(gm_rows, 6),
            "iou_pred_side": ious_p, "iou_gt_side": ious_g, "conf": confs, "filenames": np.asarray(filenames, dtype=str)}
