"""The consumer right after the hot path: ScanNet instance evaluation's prediction <->
ground-truth assignment, reference evaluation/eval/scannetv2_inst_eval.py:265-365 (`assign_instances_for_scan`) with
instance_eval_util.py:158-174 (`get_instances`), on the bit-row primitives.

The reference counts, pair by pair, `np.count_nonzero(np.logical_and(gts == instance_id, pred_mask))` over N points
(O(P x G x N) byte operations per scene); here the predicted masks and the ground-truth instances are bit rows on
the device and ONE popcount Gram (bff_cross_popcount) delivers every intersection, the void intersections and the
vertex counts; the host then assembles the reference's nested dicts in the reference's order.  The AP curves built
from these dicts (evaluate_matches, :62-214) are unchanged host logic and stay in the reference's evaluator.
"""
from __future__ import annotations

from copy import deepcopy

import numpy as np
import torch

from . import _lib

MIN_REGION_SIZE = {"stpls3d": 10}        # scannetv2_inst_eval.py:42-45 (everything else: 100)
SEM_OFFSET = {"scannetv2": 2, "scannet200": 2, "replica": 1, "stpls3d": 1, "scannetpp": 105}     # :270-281


def pred_gt_overlaps(pred_rows: torch.Tensor, gts: torch.Tensor, instance_ids, void_mask=None):
    """pred_rows: int64 bit rows [P][nw] (device) of the predicted masks (FinalResult.rows / Stage2Result.rows, or
    _lib.pack_rows of a dense bool matrix); gts: int64 [N] encoded ground-truth ids per point
    (`gts_sem * encode_value + gts_ins`, :286-291); instance_ids: the `instance_id` of the GT instances to match.

    Returns numpy arrays
      intersection [P][G]   = count_nonzero(logical_and(gts == instance_id, pred_mask))          (:334)
      pred_count   [P]      = count_nonzero(pred_mask)                                           (:318)
      gt_count     [G]      = number of points of each GT instance (its `vert_count`)
      void_inter   [P]      = count_nonzero(logical_and(bool_void, pred_mask)) if void_mask is given (:328)
    """
    dev = pred_rows.device
    gts = gts.to(dev).to(torch.int64).contiguous()
    vals = torch.as_tensor(np.asarray(instance_ids, dtype=np.int64)).to(dev)
    gt_rows = _lib.ids_to_rows(gts, vals)
    out = [_lib.cross_popcount(pred_rows, gt_rows), _lib.popcount_rows(pred_rows), _lib.popcount_rows(gt_rows)]
    if void_mask is not None:
        void_rows = _lib.pack_rows(torch.as_tensor(void_mask).to(dev).reshape(1, -1).to(torch.bool).contiguous())
        out.append(_lib.cross_popcount(pred_rows, void_rows))
    res = _lib.fetch(*out)
    return res[0], res[1], res[2], (res[3][:, 0] if void_mask is not None else None)


def encode_gt(gts_sem, gts_ins, dataset_name="scannet200"):
    """:270-291: shift the semantic ids of the dataset, clamp negatives, `sem * encode_value + (ins + 1)`, 0 where the
    instance id is ignored.  Returns (gts int array [N], encode_value)."""
    encode = 10000 if dataset_name == "scannetpp" else 1000                       # :23-27
    sem = np.array(gts_sem) - SEM_OFFSET.get(dataset_name, 0) + 1
    sem[sem < 0] = 0
    ins = np.array(gts_ins) + 1
    gts = sem * encode + ins
    gts[ins < 0] = 0
    return gts, encode


def assign_instances_for_scan(preds, gts_sem, gts_ins, class_labels, use_label=True, dataset_name="scannet200",
                              device="cuda", pred_rows=None):
    """ScanNetEval.assign_instances_for_scan (:265-365) -> (gt2pred, pred2gt), the same nested dicts.

    preds: list of {"scan_id", "label_id", "conf", "pred_mask"}; pred_mask is an (N,) array (anything != 0 is set),
    or ignored when `pred_rows` (int64 bit rows [len(preds)][nw] on the device, e.g. FinalResult.rows) is given.
    class_labels: the evaluator's valid_class_labels (ids 1..len)."""
    _lib.load()
    dev = torch.device(device)
    labels = list(class_labels)
    valid_ids = np.arange(len(labels)) + 1                                        # :30
    id2label = {int(i): lab for i, lab in zip(valid_ids, labels)}
    eval_labels = labels if use_label else ["class_agnostic"]                     # :55-58
    gts, encode = encode_gt(gts_sem, gts_ins, dataset_name)
    n = gts.shape[0]
    min_region = MIN_REGION_SIZE.get(dataset_name, 100)

    # ---- ground-truth instances (get_instances): ascending id, id 0 skipped, only valid classes
    inst_ids = np.unique(gts)
    inst_ids = inst_ids[inst_ids != 0]
    gts_dev = torch.from_numpy(np.ascontiguousarray(gts, dtype=np.int64)).to(dev)
    # ---- predictions that reach the counting stage (label known)
    keep = []
    for k, pred in enumerate(preds):
        if use_label and pred["label_id"] not in id2label:                        # :311-312
            continue
        keep.append(k)
    if pred_rows is None:
        if keep:
            dense = torch.from_numpy(np.stack([np.not_equal(np.asarray(preds[k]["pred_mask"]), 0) for k in keep]))
            for k in keep:
                assert np.asarray(preds[k]["pred_mask"]).shape[0] == n            # :320
            rows = _lib.pack_rows(dense.to(dev).contiguous())
        else:
            rows = torch.zeros((0, (n + 63) // 64), dtype=torch.int64, device=dev)
    else:
        rows = _lib.gather_rows(pred_rows, torch.tensor(keep, dtype=torch.int32, device=dev)) if keep else pred_rows[:0]
    bool_void = np.logical_not(np.in1d(gts // encode, valid_ids))                 # :306
    if len(keep) and len(inst_ids):
        inter, pred_count, gt_count, void_inter = pred_gt_overlaps(rows, gts_dev, inst_ids, bool_void)
    else:
        gt_rows = _lib.ids_to_rows(gts_dev, torch.from_numpy(inst_ids.astype(np.int64)).to(dev)) if len(inst_ids) else None
        gt_count = _lib.popcount_rows(gt_rows).cpu().numpy() if gt_rows is not None else np.zeros(0, np.int32)
        if len(keep):
            void_rows = _lib.pack_rows(torch.from_numpy(bool_void).to(dev).reshape(1, -1).contiguous())
            pred_count = _lib.popcount_rows(rows).cpu().numpy()
            void_inter = _lib.cross_popcount(rows, void_rows).cpu().numpy()[:, 0]
        else:
            pred_count = void_inter = np.zeros(0, np.int32)
        inter = np.zeros((len(keep), len(inst_ids)), np.int32)

    gt_instances = {lab: [] for lab in labels}
    col_of = {}                                                                    # (label, position) -> Gram column
    for c, iid in enumerate(inst_ids):
        label_id = int(iid // encode)
        if label_id in id2label:
            lab = id2label[label_id]
            col_of[(lab, len(gt_instances[lab]))] = c
            gt_instances[lab].append({"instance_id": int(iid), "label_id": label_id, "vert_count": int(gt_count[c]),
                                      "med_dist": -1, "dist_conf": 0.0, "box": np.zeros((6))})
    if use_label:                                                                  # :294-298
        gt2pred = deepcopy(gt_instances)
        for lab in gt2pred:
            for gt in gt2pred[lab]:
                gt["matched_pred"] = []
        cols = {lab: [col_of[(lab, k)] for k in range(len(v))] for lab, v in gt2pred.items()}
    else:                                                                          # :300-308
        agnostic, acols = [], []
        for lab, instances in gt_instances.items():
            agnostic += deepcopy(instances)
            acols += [col_of[(lab, k)] for k in range(len(instances))]
        for gt in agnostic:
            gt["matched_pred"] = []
        gt2pred = {eval_labels[0]: agnostic}
        cols = {eval_labels[0]: acols}

    # ---- predictions that are large enough to count (:323-324), numbered in input order (:358-360)
    pred2gt = {lab: [] for lab in eval_labels}
    pred_count = np.asarray(pred_count, dtype=np.int64)
    big = np.flatnonzero(pred_count >= min_region)
    records, by_label = [], {lab: [] for lab in eval_labels}
    for number, r in enumerate(big.tolist()):
        pred = preds[keep[r]]
        lab = id2label[pred["label_id"]] if use_label else eval_labels[0]
        rec = {"filename": "{}_{}".format(pred["scan_id"], number), "pred_id": number,
               "label_id": pred["label_id"] if use_label else None, "vert_count": int(pred_count[r]),
               "confidence": pred["conf"], "void_intersection": int(void_inter[r])}
        records.append(rec)
        by_label[lab].append((r, len(records) - 1))
    # ---- matches = the non-zero entries of the Gram block (predictions of a label) x (ground truth of that label);
    # np.nonzero walks it row-major, i.e. prediction by prediction and within one by ground-truth position, the order
    # in which the reference's nested loops meet them (:331-357).  IoU = I / (|gt| + |pred| - I) in float64 (:340)
    matched = [[] for _ in records]
    for lab, members in by_label.items():
        gt_list = gt2pred[lab]
        if not members or not gt_list:
            continue
        r_idx = np.array([m[0] for m in members])
        block = np.asarray(inter, dtype=np.int64)[np.ix_(r_idx, np.asarray(cols[lab], dtype=np.int64))]
        pi, gi = np.nonzero(block > 0)
        hit = block[pi, gi]
        gt_verts = np.array([g["vert_count"] for g in gt_list], dtype=np.int64)
        iou = hit.astype(np.float64) / (gt_verts[gi] + pred_count[r_idx][pi] - hit).astype(np.float64)
        for p_, g_, i_, u_ in zip(pi.tolist(), gi.tolist(), hit.tolist(), iou.tolist()):
            rec_no = members[p_][1]
            gt = gt_list[g_]
            # the ground-truth side keeps a snapshot of the prediction as it is before its own matches are attached,
            # the prediction side a shallow snapshot of the ground-truth entry (its match list stays the shared one)
            gt["matched_pred"].append(dict(records[rec_no], intersection=i_, iou=u_))
            matched[rec_no].append(dict(gt, intersection=i_, iou=u_))
    for rec, m in zip(records, matched):
        rec["matched_gt"] = m
    for lab, members in by_label.items():
        pred2gt[lab] = [records[k] for _, k in members]
    return gt2pred, pred2gt
