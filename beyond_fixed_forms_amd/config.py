"""configs/config.yaml -> attribute-style config (the reference uses Munch.fromDict(yaml.safe_load(..)),
projection_2d_to_3d.py:339 / refinement.py:138; `munch` is not a dependency here).

The keys, including their spelling (`min_aggragated_masks`, `if_occurance_threshold`,
`refinment_sim_percentile`, `refiment_iou_thres`), are the reference's (configs/config.yaml:9-67).
"""
from __future__ import annotations

import yaml

# Defaults = the values shipped in the reference's configs/config.yaml.
DEFAULTS = dict(
    width_2d=1296, height_2d=968, downsample_ratio=10,
    iou_thres=0.2, similarity_thres=0.75, min_aggragated_masks=2,
    if_occurance_threshold=False, occurance_threshold=0.3,
    if_detected_ratio_threshold=True, detected_ratio_threshold=0.38,
    remove_filtered_masks=0.4, remove_small_masks=5,
    stage1_iou_thres=0.1, refinment_sim_percentile=0.2, refiment_iou_thres=0.45,
)


class Config(dict):
    """dict with attribute access (the subset of Munch behaviour the hot path relies on)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    @classmethod
    def with_defaults(cls, **over):
        c = cls(DEFAULTS)
        c.update(over)
        return c


def load_config(path: str) -> Config:
    with open(path, "r") as f:
        return Config(yaml.safe_load(f.read()))
