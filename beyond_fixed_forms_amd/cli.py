"""Command-line drop-ins with the reference's argv (`--config <yaml> --cls "<query>"`), file layout,
checkpoints and exit-code contract: exit code 0 on success, any exception
propagates to a non-zero exit, empty results are saved, not errors."""
from __future__ import annotations

import argparse
import os
from concurrent.futures import ThreadPoolExecutor

import torch

from . import _lib
from .config import load_config
from .io import (load_scene, read_scene_checkpoint, save_result, write_scene_checkpoint)
from .projection import project_scene
from .refinement import TextSimilarity, refine_class


def _parser(desc):
    p = argparse.ArgumentParser(description=desc)                 # P:314-318 / R:128-132
    p.add_argument("--config", type=str, required=True, help="Config")
    p.add_argument("--cls", type=str, required=True, help="Class")
    return p


def projection_main(argv=None):
    """python tools/projection_2d_to_3d.py --config configs/config.yaml --cls "<class>"   (P:336-634)"""
    args = _parser("Beyond-Fixed-Forms 2D->3D projection (MI355X)").parse_args(argv)
    cfg = load_config(args.config)
    _lib.load()
    cls = args.cls
    ckpt = read_scene_checkpoint("projection_2d_to_3d", cls)
    seg_dir = os.path.join(cfg.mask_2d_dir, cls)
    scene_ids = [s[:-4] for s in sorted(s for s in os.listdir(seg_dir) if s.endswith("_00.pth"))]   # P:363
    # BFF_DEPTH_ON_DEVICE=1: upload the 16-bit depth PNGs as they are, scale + resize them on the GPU
    on_dev = os.environ.get("BFF_DEPTH_ON_DEVICE") == "1"
    # the files of scene k+1 (point cloud, ~300 depth PNGs, poses, the mask dict) are read by one background
    # thread while the GPU works on scene k; a load error surfaces at that scene's turn, after scene k is saved
    with ThreadPoolExecutor(max_workers=1) as pool:
        pending = pool.submit(load_scene, cfg, cls, scene_ids[0], depth_on_device=on_dev) if scene_ids else None
        for k, scene_id in enumerate(scene_ids):
            print("Working on", scene_id, "class", cls)
            scene = pending.result()
            pending = (pool.submit(load_scene, cfg, cls, scene_ids[k + 1], depth_on_device=on_dev)
                       if k + 1 < len(scene_ids) else None)
            _project_and_save(scene, scene_id, cfg, cls, ckpt)
    return 0


def _project_and_save(scene, scene_id, cfg, cls, ckpt):
    res = project_scene(scene, cfg, device="cuda", return_result=True)
    if not res.debug.get("empty_form", False):
        ckpt[scene_id] = True                                                             # P:580-581
        write_scene_checkpoint("projection_2d_to_3d", cls, ckpt)
    # BFF_SAVE_RLE=1 stores "ins" as RLE dicts (Open3DIS format; refinement.py and eval_scannet200.py:123-124
    # read both forms) instead of the reference's dense bool matrix
    out = res.to_rle_dict() if os.environ.get("BFF_SAVE_RLE") == "1" and not res.debug.get("empty_form") else res.to_dict()
    save_result(out, cfg.mask_3d_dir, cls, scene_id)                                        # P:630-634


def _clip_available():
    try:
        import clip  # noqa: F401
        return True
    except ImportError:
        return False


def _text_encoder(path):
    """CLIP ViT-L/14 text encoder (R:147) when the `clip` package is importable; otherwise a file of
    precomputed text embeddings {text: (D,) tensor} given by BFF_TEXT_EMBEDDINGS."""
    if path:
        table = torch.load(path, weights_only=True)
        return lambda text: table[text].reshape(1, -1)
    import clip                                                                         # noqa: F401
    dev = "cuda" if torch.cuda.is_available() else "cpu"
    model, _ = clip.load("ViT-L/14", device=dev)

    def enc(text):
        with torch.no_grad():
            return model.encode_text(clip.tokenize([text]).to(dev))
    return enc


def refinement_main(argv=None):
    """python tools/refinement.py --config configs/config.yaml --cls "<class>"            (R:135-428)"""
    args = _parser("Beyond-Fixed-Forms refinement (MI355X)").parse_args(argv)
    cfg = load_config(args.config)
    _lib.load()
    cls = args.cls
    ckpt = read_scene_checkpoint("refinement", cls)
    # BFF_TEXT_BANK=<file>: similarity service persisted by an earlier run (bank of the 198 labels + queries);
    # created on first use, so the CLIP text encoder runs once per label ever, not twice per matched mask
    bank_file = os.environ.get("BFF_TEXT_BANK")
    if bank_file and os.path.exists(bank_file):
        need_enc = os.environ.get("BFF_TEXT_EMBEDDINGS") or _clip_available()
        sim = TextSimilarity.from_file(bank_file, "cuda", _text_encoder(os.environ.get("BFF_TEXT_EMBEDDINGS")) if need_enc else None)
    else:
        sim = TextSimilarity(_text_encoder(os.environ.get("BFF_TEXT_EMBEDDINGS")), "cuda")
    stage2_dir = os.path.join(cfg.mask_3d_dir, cls)
    scenes = []
    for name in sorted(s for s in os.listdir(stage2_dir) if s.endswith("_00.pth")):        # R:154
        scene_id = name.replace(".pth", "")
        p1 = os.path.join(cfg.stage_1_results_dir, f"{scene_id}.pth")
        p2 = os.path.join(stage2_dir, f"{scene_id}.pth")
        if os.path.exists(p1) and os.path.exists(p2):                                       # R:175-178
            # user data files written by Open3DIS / by the projection stage (R:182-183)
            scenes.append((scene_id, torch.load(p1, map_location="cpu", weights_only=False),
                           torch.load(p2, map_location="cpu", weights_only=False)))
        else:
            scenes.append((scene_id, None, None))
    out = refine_class(scenes, cfg, cls, sim, "cuda")
    if bank_file:
        sim.save(bank_file)
    for scene_id, res in out.items():
        d = res.to_rle_dict() if os.environ.get("BFF_SAVE_RLE") == "1" else res.to_dict()
        d = {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in d.items()}               # reference saves CPU tensors
        save_result(d, cfg.final_output_dir, cls, scene_id)                                 # R:422-426
        if res.rows is not None and len(res.final_class):
            ckpt[scene_id] = True                                                           # R:427-428
            write_scene_checkpoint("refinement", cls, ckpt)
    return 0
