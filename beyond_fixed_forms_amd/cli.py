"""Command-line drop-ins with the reference's argv (`--config <yaml> --cls "<query>"`), file layout,
checkpoints and exit-code contract: exit code 0 on success, any exception
propagates to a non-zero exit, empty results are saved, not errors.

One process per GPU.  `BFF_GPUS=<n>|all` makes either script start n ranks of itself under
`python -m torch.distributed.run` (decided before anything touches the GPU; the child's exit code is
returned, so `run_evl.py`'s `subprocess.run(check=True)` sees one process as before); a script that finds RANK /
WORLD_SIZE in its environment (torchrun) IS a rank.  Scenes are dealt to the ranks by `distributed.shard_scenes`;
every rank reads, ingests (loader threads, `ingest.Ingestor`) and projects its shard with PIPELINE_DEPTH scenes in
flight (`pipeline.project_stream`).  The refinement makes the class's one exchange and one gather
(`distributed.ClassBatch`) and rank 0 writes the final files; the projection stage's files are the hand-over between
the two processes and are written by the rank that produced them (their `final_class` strings come from that rank's
mask_2d file), the class checkpoint by rank 0.
"""
from __future__ import annotations

import argparse
import functools
import os
import socket
import subprocess
import sys

import torch
import torch.distributed as dist

from . import _lib
from .config import load_config
from .io import (load_scene, read_scene_checkpoint, save_result, write_scene_checkpoint)
from .refinement import TextSimilarity


def _parser(desc):
    p = argparse.ArgumentParser(description=desc)                 # P:314-318 / R:128-132
    p.add_argument("--config", type=str, required=True, help="Config")
    p.add_argument("--cls", type=str, required=True, help="Class")
    return p


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(argv=None):
    """BFF_GPUS=<n>|all and not yet a rank: run n ranks of this very script under torch.distributed.run as a child
    process and return its exit code (None: stay a single process).  Nothing here initialises the GPU
    (`torch.cuda.device_count()` does not on this platform), so the launcher is chosen before any HIP call."""
    want = os.environ.get("BFF_GPUS", "").strip().lower()
    if not want or "RANK" in os.environ or "WORLD_SIZE" in os.environ:
        return None
    n = torch.cuda.device_count() if want == "all" else int(want)
    if n <= 1:
        return None
    args = list(sys.argv[1:] if argv is None else argv)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(sys.argv[0])] + args
    return subprocess.run(cmd).returncode


def _host_threads():
    """The host side makes only tiny torch CPU calls; by default each opens an OpenMP region as wide as the machine
    (times the ranks of a node).  BFF_TORCH_THREADS overrides (0: leave torch's setting alone)."""
    n = int(os.environ.get("BFF_TORCH_THREADS", "4"))
    if n > 0:
        torch.set_num_threads(n)


def init_ranks():
    """-> (rank, world size, device).  Under torchrun: one GPU per rank over RCCL (backend "nccl");
    BFF_REHEARSE_ON_ONE_GPU=1 puts every rank on cuda:0 with gloo collectives -- only to rehearse the N > 1 code
    path on a single-GPU box."""
    ws = int(os.environ.get("WORLD_SIZE", "1"))
    if ws <= 1:
        return 0, 1, "cuda"
    rehearse = os.environ.get("BFF_REHEARSE_ON_ONE_GPU") == "1"
    local = 0 if rehearse else int(os.environ.get("LOCAL_RANK", "0"))
    if rehearse:
        dist.init_process_group("gloo")
    else:
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))
    torch.cuda.set_device(local)
    return dist.get_rank(), ws, f"cuda:{local}"


def _finish_ranks():
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def _coll_device(device):
    return device if dist.get_backend() == "nccl" else "cpu"


def projection_main(argv=None):
    """python tools/projection_2d_to_3d.py --config configs/config.yaml --cls "<class>"   (P:336-634)"""
    args = _parser("Beyond-Fixed-Forms 2D->3D projection (MI355X)").parse_args(argv)
    rc = launch_ranks(argv)
    if rc is not None:
        return rc
    cfg = load_config(args.config)
    _lib.load()
    _host_threads()
    cls = args.cls
    from .distributed import shard_scenes
    from .pipeline import project_stream
    rank, ws, device = init_ranks()
    ckpt = read_scene_checkpoint("projection_2d_to_3d", cls)
    seg_dir = os.path.join(cfg.mask_2d_dir, cls)
    scene_ids = [s[:-4] for s in sorted(s for s in os.listdir(seg_dir) if s.endswith("_00.pth"))]   # P:363
    # heavier scenes first when dealing them to the ranks: the cloud file's size stands for N
    weights = None
    if ws > 1:
        weights = [os.path.getsize(p) if os.path.exists(p) else 0
                   for p in (os.path.join(cfg.scene_npy_dir, f"{s}.npy") for s in scene_ids)]
    mine = shard_scenes(scene_ids, rank, ws, weights)
    # BFF_DEPTH_ON_DEVICE=1: upload the 16-bit depth PNGs as they are, scale + resize them on the GPU
    on_dev = os.environ.get("BFF_DEPTH_ON_DEVICE") == "1"
    # the files of the next scenes (point cloud, ~300 depth PNGs, poses, the mask dict) are read and uploaded by loader
    # threads while the GPU works on the current ones; a load error surfaces at that scene's turn, after the earlier
    # scenes are saved
    sources = [functools.partial(load_scene, cfg, cls, scene_ids[i], depth_on_device=on_dev) for i in mine]
    done = []

    def consume(k, _st1, res):
        scene_id = scene_ids[mine[k]]
        print("Working on", scene_id, "class", cls)
        if not res.debug.get("empty_form", False):
            done.append(mine[k])
            if ws == 1:
                ckpt[scene_id] = True                                                     # P:580-581
                write_scene_checkpoint("projection_2d_to_3d", cls, ckpt)
        # BFF_SAVE_RLE=1 stores "ins" as RLE dicts (Open3DIS format; refinement.py and eval_scannet200.py:123-124
        # read both forms) instead of the reference's dense bool matrix
        out = res.to_rle_dict() if os.environ.get("BFF_SAVE_RLE") == "1" and not res.debug.get("empty_form") else res.to_dict()
        save_result(out, cfg.mask_3d_dir, cls, scene_id)                                    # P:630-634

    project_stream(sources, cfg, device, consume, n_loaders=int(os.environ.get("BFF_LOADERS", "2")), with_stage1=False)
    if ws > 1:
        # one small reduction tells rank 0 which scenes were completed anywhere; it alone writes the class checkpoint
        flags = torch.zeros(max(len(scene_ids), 1), dtype=torch.int32, device=_coll_device(device))
        if done:
            flags[torch.tensor(done, dtype=torch.long)] = 1
        dist.reduce(flags, dst=0, op=dist.ReduceOp.SUM)
        if rank == 0:
            for i in torch.nonzero(flags.cpu()).view(-1).tolist():
                ckpt[scene_ids[i]] = True
            write_scene_checkpoint("projection_2d_to_3d", cls, ckpt)
        _finish_ranks()
    return 0


def _clip_available():
    try:
        import clip  # noqa: F401
        return True
    except ImportError:
        return False


def _text_encoder(path, device="cuda"):
    """CLIP ViT-L/14 text encoder (R:147) when the `clip` package is importable; otherwise a file of
    precomputed text embeddings {text: (D,) tensor} given by BFF_TEXT_EMBEDDINGS."""
    if path:
        table = torch.load(path, weights_only=True)
        return lambda text: table[text].reshape(1, -1)
    import clip                                                                         # noqa: F401
    dev = device if torch.cuda.is_available() else "cpu"
    model, _ = clip.load("ViT-L/14", device=dev)

    def enc(text):
        with torch.no_grad():
            return model.encode_text(clip.tokenize([text]).to(dev))
    return enc


def refinement_main(argv=None):
    """python tools/refinement.py --config configs/config.yaml --cls "<class>"            (R:135-428)"""
    args = _parser("Beyond-Fixed-Forms refinement (MI355X)").parse_args(argv)
    rc = launch_ranks(argv)
    if rc is not None:
        return rc
    cfg = load_config(args.config)
    _lib.load()
    _host_threads()
    cls = args.cls
    from .distributed import ClassBatch, shard_scenes
    rank, ws, device = init_ranks()
    ckpt = read_scene_checkpoint("refinement", cls)
    # BFF_TEXT_BANK=<file>: similarity service persisted by an earlier run (bank of the 198 labels + queries);
    # created on first use, so the CLIP text encoder runs once per label ever, not twice per matched mask
    bank_file = os.environ.get("BFF_TEXT_BANK")
    emb_file = os.environ.get("BFF_TEXT_EMBEDDINGS")
    if bank_file and os.path.exists(bank_file):
        need_enc = emb_file or _clip_available()
        sim = TextSimilarity.from_file(bank_file, device, _text_encoder(emb_file, device) if need_enc else None)
    else:
        sim = TextSimilarity(_text_encoder(emb_file, device), device)
    stage2_dir = os.path.join(cfg.mask_3d_dir, cls)
    ids = [s.replace(".pth", "") for s in sorted(s for s in os.listdir(stage2_dir) if s.endswith("_00.pth"))]   # R:154
    p1 = lambda sid: os.path.join(cfg.stage_1_results_dir, f"{sid}.pth")
    p2 = lambda sid: os.path.join(stage2_dir, f"{sid}.pth")
    have = [os.path.exists(p1(sid)) and os.path.exists(p2(sid)) for sid in ids]             # R:175-178
    # A scene without its stage-1 file is skipped by pass 1 only (R:178), which shifts the reference's per-scene lists
    # against its scene loop in pass 2 (R:330): that order dependence is reproduced by keeping such a class on one rank.
    owners = ws if all(have) else 1
    mine = shard_scenes(ids, rank, owners) if rank < owners else []
    s_max = max(1, -(-len(ids) // owners))
    batch = ClassBatch(cfg, cls, sim, device, ids, s_max)
    for i in mine:
        if have[i]:
            # user data files written by Open3DIS / by the projection stage (R:182-183)
            batch.add(ids[i], torch.load(p1(ids[i]), map_location="cpu", weights_only=False),
                      torch.load(p2(ids[i]), map_location="cpu", weights_only=False))
        else:
            batch.add(ids[i], None, None)
    batch.finish()
    if bank_file and rank == 0:
        sim.save(bank_file)
    if rank == 0:
        res = batch.results()                                   # all scenes of the class (gathered over RCCL when ws > 1)
        for scene_id in ids:
            if scene_id not in res:
                continue
            fr = res[scene_id]
            d = fr.to_rle_dict() if os.environ.get("BFF_SAVE_RLE") == "1" else fr.to_dict()
            d = {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in d.items()}           # reference saves CPU tensors
            save_result(d, cfg.final_output_dir, cls, scene_id)                             # R:422-426
            if fr.rows is not None and len(fr.final_class):
                ckpt[scene_id] = True                                                       # R:427-428
                write_scene_checkpoint("refinement", cls, ckpt)
    if ws > 1:
        _finish_ranks()
    return 0
