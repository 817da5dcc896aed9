"""CPU restatement of the reference RLE codec.  TEST INFRASTRUCTURE ONLY.

Follows /root/reference/tools/utils/rle_encode_decode.py (:10-99) and
/root/reference/tools/refinement.py rle_decode (:26-39).  Format: ``counts`` holds
(start, length) pairs over the row-major flattened mask, start is 1-based.
Pinned by tests/golden/rle_*.npz.
"""
from __future__ import annotations

import numpy as np
import torch


def rle_decode_ref(rle) -> np.ndarray:
    """{"length", "counts"} -> uint8 (length,).  Reference refinement.py:26-39 and
    rle_encode_decode.py:45-57 (identical bodies)."""
    counts = rle["counts"]
    starts = np.asarray(counts[0::2], dtype=np.int32) - 1
    lens = np.asarray(counts[1::2], dtype=np.int32)
    out = np.zeros(rle["length"], dtype=np.uint8)
    for lo, n in zip(starts, lens):
        out[lo:lo + n] = 1      # later runs overwrite with the same value; slices clip at the end
    return out


def rle_decode_batch_ref(rles) -> torch.Tensor:
    """list of RLE dicts -> uint8 (n, length).  Reference rle_encode_decode.py:35-61."""
    return torch.from_numpy(np.stack([rle_decode_ref(r) for r in rles]))


def rle_encode_batch_ref(masks: torch.Tensor):
    """bool (n, length) -> list of RLE dicts.  Reference rle_encode_decode.py:10-32."""
    n, length = masks.shape[:2]
    pad = torch.zeros((n, 1), dtype=torch.bool)
    m = torch.cat([pad, masks.to(torch.bool), pad], dim=1)
    out = []
    for i in range(n):
        edges = torch.nonzero(m[i, 1:] != m[i, :-1]).view(-1) + 1
        edges[1::2] -= edges[::2]
        out.append(dict(length=length, counts=edges.numpy()))
    return out


def decode_2d_masks_ref(frames, image_shape):
    """Per frame: RLE list -> uint8 (M,1,H,W).  Reference rle_encode_decode.py:82-99."""
    for fr in frames:
        dense = rle_decode_batch_ref(fr["segmented_frame_masks"])
        fr["segmented_frame_masks"] = dense.view(dense.shape[0], 1, *image_shape)
    return frames


def encode_2d_masks_ref(frames):
    """Per frame: bool (M,1,H,W) -> RLE list.  Reference rle_encode_decode.py:63-80."""
    for fr in frames:
        m = fr["segmented_frame_masks"]
        fr["segmented_frame_masks"] = rle_encode_batch_ref(m.view(m.shape[0], -1))
    return frames
