"""The file-level drop-in boundary on a GPU box: the reference's directory layout on disk, our stage scripts
run as subprocesses with the reference's argv, outputs compared with the oracle (bit-identical masks)."""
import os
import subprocess
import sys
import warnings

import numpy as np
import pytest
import torch
import yaml

from oracle import projection_ref as pref, refinement_ref as rref
from oracle.make_golden_shared import bank_encoder

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def write_scene(root, scene, cls):
    from PIL import Image
    sd = root / "2d" / scene.scene_id
    for sub in ("intrinsic", "pose", "depth", "color"):
        (sd / sub).mkdir(parents=True, exist_ok=True)
    for d in ("npy", f"m2d/{cls}", "stage1"):
        (root / d).mkdir(parents=True, exist_ok=True)
    np.savetxt(sd / "intrinsic" / "intrinsic_color.txt", scene.cam_intr)
    np.save(root / "npy" / f"{scene.scene_id}.npy", scene.points)
    for f in scene.color_files:
        (sd / "color" / f).write_bytes(b"")
    for fid, pose in scene.poses.items():
        np.savetxt(sd / "pose" / f"{fid}.txt", pose)
        Image.fromarray(np.round(scene.depths[fid].astype(np.float64) * 1000).astype(np.uint16)).save(sd / "depth" / f"{fid}.png")
    torch.save(scene.mask_2d, root / "m2d" / cls / f"{scene.scene_id}.pth")
    torch.save(scene.stage1, root / "stage1" / f"{scene.scene_id}.pth")


def test_stage_scripts_end_to_end(tmp_path):
    from beyond_fixed_forms_amd.config import Config
    from beyond_fixed_forms_amd.labels import SCANNET200_LABELS
    from beyond_fixed_forms_amd.synthetic import make_scene, make_text_bank
    cls = "table"
    scenes = [make_scene("tiny", seed=s) for s in (40, 41)]
    scenes[1].scene_id = "scene0041_00"
    for sc in scenes:
        write_scene(tmp_path, sc, cls)
    cfg = Config.with_defaults(width_2d=scenes[0].width, height_2d=scenes[0].height,
                               scene_2d_dir=str(tmp_path / "2d"), scene_npy_dir=str(tmp_path / "npy"),
                               mask_2d_dir=str(tmp_path / "m2d"), mask_3d_dir=str(tmp_path / "m3d"),
                               stage_1_results_dir=str(tmp_path / "stage1"), final_output_dir=str(tmp_path / "final"))
    (tmp_path / "config.yaml").write_text(yaml.safe_dump(dict(cfg)))
    bank, index = make_text_bank(64, seed=5)
    torch.save({lab: bank[i].float() for i, lab in enumerate(SCANNET200_LABELS)} | {cls: bank[index[cls]].float()},
               tmp_path / "text.pt")
    env = dict(os.environ, BFF_TEXT_EMBEDDINGS=str(tmp_path / "text.pt"))
    for script in ("projection_2d_to_3d.py", "refinement.py"):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", script), "--config", str(tmp_path / "config.yaml"),
                            "--cls", cls], cwd=tmp_path, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
    enc = bank_encoder(bank.float(), index)
    trip = []
    for sc in scenes:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            exp = pref.project_scene_ref(sc, cfg)
        got = torch.load(tmp_path / "m3d" / cls / f"{sc.scene_id}.pth", map_location="cpu", weights_only=False)
        assert torch.equal(got["ins"], exp["ins"]) and torch.equal(got["conf"], exp["conf"])
        assert got["final_class"] == exp["final_class"]
        trip.append((sc.scene_id, sc.stage1, exp))
    fexp = rref.refine_class_ref(trip, cfg, cls, enc)
    for sc in scenes:
        got = torch.load(tmp_path / "final" / cls / f"{sc.scene_id}.pth", map_location="cpu", weights_only=False)
        assert torch.equal(got["ins"], fexp[sc.scene_id]["ins"]) and torch.equal(got["conf"], fexp[sc.scene_id]["conf"])
        assert got["final_class"] == fexp[sc.scene_id]["final_class"]
    ck = yaml.safe_load((tmp_path / "checkpoints" / f"projection_2d_to_3d_checkpoint_{cls}.yaml").read_text())
    assert ck == {sc.scene_id: True for sc in scenes}
    assert (tmp_path / "checkpoints" / f"refinement_checkpoint_{cls}.yaml").exists()
    # same run storing RLE instead of dense masks: identical content
    env_rle = dict(env, BFF_SAVE_RLE="1", BFF_DEPTH_ON_DEVICE="1")      # and depth decoded PNG -> GPU directly
    for script in ("projection_2d_to_3d.py", "refinement.py"):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", script), "--config", str(tmp_path / "config.yaml"),
                            "--cls", cls], cwd=tmp_path, env=env_rle, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
    from oracle.rle_ref import rle_decode_ref
    for sc in scenes:
        got = torch.load(tmp_path / "final" / cls / f"{sc.scene_id}.pth", map_location="cpu", weights_only=False)
        dec = np.stack([rle_decode_ref(r) for r in got["ins"]]).astype(bool)
        assert np.array_equal(dec, fexp[sc.scene_id]["ins"].numpy())
    # a failing stage must exit non-zero (run_evl.py relies on subprocess.run(check=True))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "projection_2d_to_3d.py"), "--config",
                        str(tmp_path / "config.yaml"), "--cls", "no such class"], cwd=tmp_path, capture_output=True)
    assert r.returncode != 0


def test_stage_scripts_two_ranks(tmp_path):
    """The drop-in scripts as the multi-GPU job they become with BFF_GPUS=2: each script starts two ranks of itself
    under torch.distributed.run (here both on the box's one GPU, collectives over gloo: BFF_REHEARSE_ON_ONE_GPU=1),
    scenes are sharded, every rank ingests + projects its shard through the stream pipeline, the refinement makes ONE
    similarity exchange and ONE gather, rank 0 writes the final files and the checkpoints.  Outputs = the oracle's."""
    from beyond_fixed_forms_amd.config import Config
    from beyond_fixed_forms_amd.labels import SCANNET200_LABELS
    from beyond_fixed_forms_amd.synthetic import make_scene, make_text_bank
    cls = "table"
    scenes = [make_scene("tiny", seed=60 + k) for k in range(5)]
    for k, sc in enumerate(scenes):
        sc.scene_id = f"scene{60 + k:04d}_00"
        write_scene(tmp_path, sc, cls)
    cfg = Config.with_defaults(width_2d=scenes[0].width, height_2d=scenes[0].height,
                               scene_2d_dir=str(tmp_path / "2d"), scene_npy_dir=str(tmp_path / "npy"),
                               mask_2d_dir=str(tmp_path / "m2d"), mask_3d_dir=str(tmp_path / "m3d"),
                               stage_1_results_dir=str(tmp_path / "stage1"), final_output_dir=str(tmp_path / "final"))
    (tmp_path / "config.yaml").write_text(yaml.safe_dump(dict(cfg)))
    bank, index = make_text_bank(64, seed=6)
    torch.save({lab: bank[i].float() for i, lab in enumerate(SCANNET200_LABELS)} | {cls: bank[index[cls]].float()},
               tmp_path / "text.pt")
    env = dict(os.environ, BFF_TEXT_EMBEDDINGS=str(tmp_path / "text.pt"), BFF_GPUS="2", BFF_REHEARSE_ON_ONE_GPU="1",
               BFF_DEPTH_ON_DEVICE="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    for script in ("projection_2d_to_3d.py", "refinement.py"):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", script), "--config", str(tmp_path / "config.yaml"),
                            "--cls", cls], cwd=tmp_path, env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
    enc = bank_encoder(bank.float(), index)
    trip = []
    for sc in scenes:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            exp = pref.project_scene_ref(sc, cfg)
        got = torch.load(tmp_path / "m3d" / cls / f"{sc.scene_id}.pth", map_location="cpu", weights_only=False)
        assert torch.equal(got["ins"], exp["ins"]) and torch.equal(got["conf"], exp["conf"])
        assert got["final_class"] == exp["final_class"]
        trip.append((sc.scene_id, sc.stage1, exp))
    fexp = rref.refine_class_ref(trip, cfg, cls, enc)
    for sc in scenes:
        got = torch.load(tmp_path / "final" / cls / f"{sc.scene_id}.pth", map_location="cpu", weights_only=False)
        e = fexp[sc.scene_id]
        if isinstance(e["ins"], list):
            assert got["ins"] == [] and got["conf"] == []
            continue
        assert torch.equal(got["ins"], e["ins"]) and torch.equal(got["conf"], e["conf"]) and got["final_class"] == e["final_class"]
    ck = yaml.safe_load((tmp_path / "checkpoints" / f"projection_2d_to_3d_checkpoint_{cls}.yaml").read_text())
    assert ck == {sc.scene_id: True for sc in scenes}
    assert (tmp_path / "checkpoints" / f"refinement_checkpoint_{cls}.yaml").exists()
    # a failing rank fails the job (run_evl.py relies on subprocess.run(check=True))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "projection_2d_to_3d.py"), "--config",
                        str(tmp_path / "config.yaml"), "--cls", "no such class"], cwd=tmp_path, env=env, capture_output=True,
                       timeout=600)
    assert r.returncode != 0
