"""The CLIP box filter in front of the hot path: tools/segmentation_2d.py:324-337
(`compute_avg_description_encodings`) and :340-402 (`bbox_filter`), the only image-feature x text-embedding product in
the reference -- its result becomes the `confidences` of the mask_2d input.  The encoders themselves (CLIP image /
text towers) are upstream neural inference and stay where they are; this module does the arithmetic around them on
the device: the per-class mean of normalised description encodings, and  F.normalize(box_emb) @ text_mean.T  >= thr
as ONE MFMA GEMM over all the boxes handed in -- of one frame, as the reference calls it, or batched over every frame
and query of a scene (BASELINE config 5: thousands of 768-d features x a 200-class bank), where the matrix cores
have something to do.
"""
from __future__ import annotations

from typing import List, Sequence

import numpy as np
import torch

from . import _lib


def average_description_embeddings(encodings: Sequence[torch.Tensor], device="cuda") -> torch.Tensor:
    """SEG:324-337: encodings[c] = the CLIP text encodings (n_c, D) of the descriptions of class c (as
    `clip_model.encode_text` returns them).  Each is L2-normalised (F.normalize), averaged over the descriptions, the
    means are stacked and normalised again -> (n_classes, D) in the encodings' dtype (float16 / float32)."""
    _lib.load()
    dt = torch.float16 if encodings[0].dtype == torch.float16 else torch.float32
    desc = torch.cat([e.reshape(-1, e.shape[-1]).to(dt) for e in encodings]).to(device).contiguous()
    offs = np.zeros(len(encodings) + 1, dtype=np.int32)
    np.cumsum([e.reshape(-1, e.shape[-1]).shape[0] for e in encodings], out=offs[1:])
    return _lib.description_means(desc, torch.from_numpy(offs).to(device))


def _pad32(x):
    pad = (-x.shape[1]) % 32
    return torch.nn.functional.pad(x, (0, pad)) if pad else x


def box_similarities(box_embeddings: torch.Tensor, text_means: torch.Tensor) -> torch.Tensor:
    """SEG:388-393: F.normalize(box_embeddings) @ text_means.T -> float32 (n_boxes, n_classes) on the matrix cores
    (float16 operands, float32 accumulate; the box rows are normalised in the epilogue, the text means are taken as
    given -- compute_avg_description_encodings normalised them).  Any number of boxes: batch them."""
    a = _pad32(box_embeddings.to(torch.float16)).contiguous()
    b = _pad32(text_means.to(a.device).to(torch.float16)).contiguous()
    return _lib.normalized_gemm_f16(a, b)


def bbox_filter(boxes: torch.Tensor, phrases: List[str], box_embeddings: torch.Tensor, capt_feature_ensembled: torch.Tensor,
                clip_threshold: float = 0.5):
    """SEG:340-402 after the encoder: keep the boxes whose similarity with the (single) caption feature reaches the
    threshold.  Returns (boxes_filtered, logits_filtered (n_kept, n_classes), phrases_filtered) like the reference."""
    if boxes is None or len(boxes) == 0:                                            # SEG:354-355
        return boxes, [], []
    sims = box_similarities(box_embeddings.to(capt_feature_ensembled.device if capt_feature_ensembled.is_cuda else "cuda"),
                            capt_feature_ensembled)
    mask = (sims >= clip_threshold).squeeze(1).cpu()                                # SEG:396
    keep = torch.nonzero(mask).view(-1)
    return boxes[keep.to(boxes.device)], sims[keep.to(sims.device)], [phrases[i] for i in keep.tolist()]
