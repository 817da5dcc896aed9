"""Generate tests/golden/box_filter.npz from the REAL reference functions.  TEST INFRASTRUCTURE ONLY.

Run in the build container (needs /root/reference):   python -m oracle.make_golden_boxfilter

Imports tools/segmentation_2d.py of the reference and calls compute_avg_description_encodings (:324-337) and
bbox_filter (:340-402) with injected embeddings: the CLIP encoders, GroundingDINO, SAM, torchvision and the prompt
generator are not installed here (and are upstream neural inference, out of scope), so stub modules stand in for them
-- a stub CLIP model returns the injected text / image embeddings, the image transform is a pass-through -- and only
the arithmetic around the similarity product is exercised: F.normalize, the per-class mean of description encodings,
the re-normalisation, `box_embeddings @ capt_feature_ensembled.T`, the `>= clip_threshold` filter.  Stored: the
injected embeddings (inputs) and the functions' outputs."""
from __future__ import annotations

import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE_ROOT = "/root/reference"


class _Mod(types.ModuleType):
    def __getattr__(self, name):               # any attribute: a harmless callable / class placeholder
        if name.startswith("__"):
            raise AttributeError(name)
        return lambda *a, **k: None


def load_reference_segmentation():
    sys.dont_write_bytecode = True
    names = ["groundingdino", "groundingdino.datasets", "groundingdino.datasets.transforms", "groundingdino.models",
             "groundingdino.util", "groundingdino.util.box_ops", "groundingdino.util.slconfig", "groundingdino.util.utils",
             "groundingdino.util.inference", "segment_anything", "descriptor_generator", "clip", "termcolor", "munch",
             "torchvision", "torchvision.transforms", "cv2", "open3d", "configs"]
    for n in names:
        m = sys.modules.setdefault(n, _Mod(n))
        if isinstance(m, _Mod):
            m.__path__ = []                        # a package, so that `import a.b.c` walks through it
        if "." in n:                               # parent.child is the child MODULE, not a placeholder
            parent, child = n.rsplit(".", 1)
            setattr(sys.modules[parent], child, m)
    box_ops = sys.modules["groundingdino.util.box_ops"]

    def box_cxcywh_to_xyxy(x):                 # groundingdino.util.box_ops (public formula: centre/size -> corners)
        cx, cy, w, h = x.unbind(-1)
        return torch.stack([cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h], dim=-1)
    box_ops.box_cxcywh_to_xyxy = box_cxcywh_to_xyxy
    sys.modules["groundingdino.util"].box_ops = box_ops
    tv = sys.modules["torchvision.transforms"]
    tv.Compose = lambda ts: (lambda img: torch.zeros(3, 4, 4))        # the stub encoder ignores pixels
    for n in ("Resize", "CenterCrop", "ToTensor", "Normalize"):
        setattr(tv, n, lambda *a, **k: None)
    tv.InterpolationMode = types.SimpleNamespace(BICUBIC=3)
    sys.modules["munch"].Munch = dict
    sys.modules["termcolor"].colored = lambda s, *a, **k: s
    for p in (REFERENCE_ROOT, os.path.join(REFERENCE_ROOT, "tools")):
        if p not in sys.path:
            sys.path.insert(0, p)
    cwd = os.getcwd()
    os.chdir(REFERENCE_ROOT)
    try:
        import segmentation_2d as seg
    finally:
        os.chdir(cwd)
    seg.device = torch.device("cpu")
    return seg


class StubClip:
    """encode_text(tokens) / encode_image(region): injected embeddings in call order."""

    def __init__(self, text_table=None, image_rows=None):
        self.text_table, self.image_rows, self.calls = text_table, image_rows, 0

    def encode_text(self, tok):
        return self.text_table[int(tok)]

    def encode_image(self, region):
        row = self.image_rows[self.calls:self.calls + 1]
        self.calls += 1
        return row


def main():
    seg = load_reference_segmentation()
    out = {}
    gen = torch.Generator().manual_seed(17)
    names = []
    for tag, dt, d in (("f32", torch.float32, 64), ("f16", torch.float16, 768)):
        # ---- compute_avg_description_encodings: 3 classes with 5 / 1 / 8 descriptions
        counts = [5, 1, 8]
        table = [torch.randn(c, d, generator=gen).mul(3.0).to(dt) for c in counts]
        sys.modules["descriptor_generator"].descr_generator_selector = \
            lambda base, method=None: {f"class{k}": k for k in range(len(counts))}
        sys.modules["clip"].tokenize = lambda v: torch.tensor(v)
        text_model = StubClip(text_table=table)
        with torch.no_grad():
            means = seg.compute_avg_description_encodings("prompt", text_model, mode="waffle")
        out[f"{tag}.desc"] = torch.cat(table).float().numpy()
        out[f"{tag}.desc_counts"] = np.asarray(counts)
        out[f"{tag}.desc_means"] = means.float().numpy()
        # ---- bbox_filter: 40 boxes; similarities spread around the threshold 0.2, some exactly equal to it
        n_box = 40
        q = means[0:1].float()                                        # the single query class (n_classes = 1 in the pipeline)
        emb = torch.randn(n_box, d, generator=gen)
        tgt = torch.linspace(-0.1, 0.6, n_box)
        qn = (q / q.norm()).reshape(-1)
        for i in range(n_box):                                        # cos(emb_i, q) ~ tgt_i
            r = emb[i] - (emb[i] @ qn) * qn
            emb[i] = (tgt[i] * qn + (1 - tgt[i] ** 2).sqrt() * r / r.norm()) * (1.0 + i % 7)
        emb = emb.to(dt)
        image = torch.rand(3, 50, 70, generator=gen)
        boxes = torch.rand(n_box, 4, generator=gen) * 0.5 + 0.25
        phrases = [f"p{i}" for i in range(n_box)]
        img_model = StubClip(image_rows=emb)
        with torch.no_grad():
            b_f, logits_f, phr_f = seg.bbox_filter(image, boxes.clone(), phrases, means[0:1], clip_threshold=0.2,
                                                   clip_model=img_model)
        out[f"{tag}.box_emb"] = emb.float().numpy()
        out[f"{tag}.boxes"] = boxes.numpy()
        out[f"{tag}.kept"] = np.asarray([int(p[1:]) for p in phr_f])
        out[f"{tag}.logits"] = logits_f.float().numpy().reshape(-1)
        out[f"{tag}.boxes_kept"] = b_f.numpy()
        names.append(tag)
        print(f"  {tag}: means {tuple(means.shape)} {means.dtype}, {len(phr_f)} of {n_box} boxes kept")
    out["cases"] = np.asarray(names)
    path = os.path.join(ROOT, "tests", "golden", "box_filter.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
