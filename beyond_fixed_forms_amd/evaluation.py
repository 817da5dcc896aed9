"""The consumer right after the hot path: prediction <-> ground-truth overlap counting of the ScanNet instance
evaluation (reference evaluation/eval/scannetv2_inst_eval.py:265-349, `assign_instances_for_scan`), on the
bit-row primitives.  Only the O(P x G x N) counting is done here; the AP bookkeeping around it (matching,
confidence sorting, precision/recall curves) is unchanged host logic and stays in the reference's evaluator.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib


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
