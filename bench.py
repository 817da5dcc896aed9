"""Headline benchmark: scenes/sec of 2D->3D projection + refinement (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W          (N = 1)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (N > 1)

A "step" is one pass of the hot path over one batch = ONE scene per GPU of BASELINE config 2
(200k points x 300 views @968x1296, 30 masks/view, stage-1 S1=100, 768-d text bank): RLE decode ->
fused projection sweep -> IoU/label adjacency -> components -> merge -> ratio filter -> overlap/size
filters -> refinement pass 1 (stage-1 RLE decode, cross IoU, text cosines).  As in the reference, the
refinement is per query CLASS (R:316-324: one similarity threshold over all scenes of the class): every
`--class-batch` steps (default 8 scenes per GPU) the class is closed with ONE all-gather of the ranks'
similarity sets, pass 2, and ONE gather of the final bit-packed masks on rank 0 -- inside the timed region.
Inputs are resident in HBM (uploaded before the timed region) in the reference's formats: float64 cloud,
float32 depth, RLE runs.  Scenes shard one per GPU (weak scaling); N = 1 runs the same loop without collectives.

The JSON line also carries `roofline` for the HBM-bound projection sweep (HIP-event timed on the
launch stream, live) and `cpu_baseline`: the oracle (CPU restatement of the reference) timed on a
bounded sample of the same scene on this box's host cores.
"""
from __future__ import annotations

import os

try:
    _NCPU = min(len(os.sched_getaffinity(0)), 32)
except AttributeError:
    _NCPU = min(os.cpu_count() or 1, 32)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")     # one hardware queue per scene in flight (see beyond_fixed_forms_amd/__init__.py)
os.environ.setdefault("OPENBLAS_NUM_THREADS", str(_NCPU))
os.environ.setdefault("OMP_NUM_THREADS", str(_NCPU))

import argparse
import copy
import json

import numpy as np
import sys
import time
import warnings

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from beyond_fixed_forms_amd import _lib, distributed as bdist  # noqa: E402
from beyond_fixed_forms_amd.config import Config  # noqa: E402
from beyond_fixed_forms_amd.pipeline import PIPELINE_DEPTH, pipelined, scene_streams  # noqa: E402
from beyond_fixed_forms_amd.projection import projection_back, projection_front  # noqa: E402
from beyond_fixed_forms_amd.refinement import TextSimilarity, prepare_stage1  # noqa: E402
from beyond_fixed_forms_amd.scene import prepare_scene  # noqa: E402
from beyond_fixed_forms_amd.synthetic import SHAPES, make_scene, make_text_bank, with_sensor_depth  # noqa: E402
from beyond_fixed_forms_amd.timing import KernelTimers  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s; ~6.3 TB/s achievable)
QUERY = "table"


def algorithmic_bytes(ds, n_frames_swept):
    """SURVEY.md section 8(d) per-unit figures for one fused sweep: 24 B xyz + 4 B depth per (frame, point);
    ceil(M/8) B mask-word gather per (mask-frame, point); the two int32 counters read+written once (16 N) --
    that is what project_views_kernel moves.  SURVEY's fourth term, every instance bit written once (V*M*N/8),
    is not moved at all any more: the rows are a recycled zero arena, the kernel stores only the 32-B sectors
    that receive a point (~1 %) and the back half clears exactly those; it is returned separately and NOT
    credited to the kernel.  Each (frame, point) is counted once although the launch serves both the mask
    sweep and the viewed sweep of the reference.  Returns (bytes of the kernel, SURVEY's row-write term)."""
    n = ds.n_points
    mask_word = ds.word_bits // 8
    return (n_frames_swept * n * 28 + ds.n_mask_frames * n * mask_word + 16 * n, ds.n_rows * n // 8)


def bank_encoder(bank, index):
    def enc(text):
        return bank[index[text.replace(" ", "_")]][None, :]
    return enc


def cpu_baseline(scene, cfg, enc, n_sample_views=100, threads=None):
    """Oracle (CPU restatement of the reference) on a bounded sample (~10-30 s of CPU work): the first
    `n_sample_views` frames of the same scene at full N / HxW / M, whole path (projection, aggregation of
    the sample's instances, ratio sweep, filters, refinement); the projection time is scaled linearly to the
    scene's views, which favours the CPU (its aggregation grows quadratically with the instance count)."""
    from oracle.projection_ref import project_scene_ref
    from oracle.refinement_ref import refine_class_ref
    threads = threads or min(host_cores(), 32)
    torch.set_num_threads(threads)
    sub = copy.copy(scene)
    sub.mask_2d = [dict(f) for f in scene.mask_2d[:n_sample_views]]
    ratio = cfg.downsample_ratio
    keep = {f["frame_id"] for f in sub.mask_2d}
    sub.color_files = [f for f in scene.color_files if int(f[:-4]) < n_sample_views * ratio]
    if getattr(scene, "depths_raw", None) is not None and not scene.depths:
        # the reference holds float32 (H, W) depth after cv2.imread / 1000 + cv2.resize (P:431-436); that step is done
        # here BEFORE the clock starts (NumPy restatement of the resize, slower than cv2: timing it would flatter the GPU)
        from beyond_fixed_forms_amd.io import resize_bilinear_f32
        from beyond_fixed_forms_amd.scene import viewed_frame_ids
        need = {f["frame_id"][:-4] for f in sub.mask_2d} | set(viewed_frame_ids(sub.color_files, ratio))
        sub.depths = {f: resize_bilinear_f32(scene.depths_raw[f].astype(np.float32) / np.float32(1000), scene.width, scene.height)
                      for f in need}
        sub.depths_raw = None
    stages = {}
    t0 = time.perf_counter()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        res = project_scene_ref(sub, cfg, stage_times=stages)
        t1 = time.perf_counter()
        try:
            refine_class_ref([(sub.scene_id, sub.stage1, res)], cfg, QUERY, enc)
        except IndexError:
            pass
    t2 = time.perf_counter()
    stages["v_refinement"] = t2 - t1
    n_views = len(scene.mask_2d)
    est_scene_s = (t1 - t0) * n_views / max(1, len(sub.mask_2d)) + (t2 - t1)
    return {"stages_sample_s": {k: round(v, 4) for k, v in stages.items()},     # SURVEY 8(d) stages (i)-(v), on the sample
            "value": 1.0 / est_scene_s, "unit": "scenes/s", "cores": threads, "kind": "port",
            "sample": f"oracle projection+refinement on {len(sub.mask_2d)} of {n_views} mask views and "
                      f"{(len(sub.color_files) + ratio - 1) // ratio} viewed frames at full N/HxW/M ({t2 - t0:.1f} s CPU), "
                      f"projection scaled linearly to {n_views} views; starts from the reference's host formats and so "
                      f"INCLUDES the RLE decode of the 2-D masks (the GPU `value` starts from HBM-resident run tables; "
                      f"see host_inclusive for the GPU rate from host arrays); the reference's O(Ins^2) label loop and "
                      f"O(Ins^4) closure are NOT included (oracle uses integer ids and a frontier search)",
            "sample_seconds": t2 - t0}


def cpu_stagewise(args):
    """`--cpu-stagewise`: stage-wise CPU timings of the reference path beside the GPU path on this
    machine's host cores -- part of the cpu_baseline leg.  The oracle gives stages (i) projection+votes, (ii)
    Gram+merge as the oracle does it (integer label ids, frontier search), (iii) ratio-filter sweep, (iv)
    overlap+filters, (v) refinement on a sample of the views, scaled as noted.  The reference's own formulation of
    (ii) -- a Python double loop over label strings (P:169-187) and the transitive closure by Ins rounds of
    clamp(R@A + A) (P:250-274) -- is restated here, timed at small instance counts and extrapolated (O(Ins^2),
    O(Ins^4)), clearly labelled as such."""
    import functools
    say = functools.partial(print, flush=True)
    threads = int(os.environ.get("BFF_CPU_THREADS", str(min(host_cores(), 32))))      # the GPU box gives one GPU's share of the host
    torch.set_num_threads(threads)
    say(f"host: {os.cpu_count()} logical cores; torch threads {threads}; model:",
        next((ln.split(":")[1].strip() for ln in open("/proc/cpuinfo") if ln.startswith("model name")), "?"))
    n_sample = args.cpu_sample_views
    scene = make_scene(args.shape, seed=0, query=QUERY, device="cuda" if torch.cuda.is_available() else "cpu")
    cfg = Config.with_defaults(width_2d=scene.width, height_2d=scene.height)
    bank, index = make_text_bank(768, seed=0)
    base = cpu_baseline(scene, cfg, bank_encoder(bank.float(), index), n_sample_views=n_sample, threads=threads)
    st = base["stages_sample_s"]
    n_views = len(scene.mask_2d)
    n_used = min(n_sample, n_views)
    ins_full = sum(len(f["segmented_frame_masks"]) for f in scene.mask_2d)
    scale = n_views / max(1, n_used)
    say(f"\n{args.shape}: oracle on {n_used} of {n_views} mask views, Ins = {ins_full} in the whole scene, N = {scene.points.shape[0]}")
    say(f"{'stage':34s} {'sample s':>10s} {'scaled to the scene s':>24s}")
    rows = [("i_projection_votes", scale, "linear in views"), ("ii_gram_merge", scale ** 2, "oracle formulation, ~Ins^2"),
            ("iii_ratio_filter_sweep", scale, "linear in frames"), ("iv_overlap_filters", 1.0, "K^2 N, K small"),
            ("v_refinement", 1.0, "S1 K N")]
    tot = 0.0
    for k, f, note in rows:
        v = st.get(k, 0.0)
        tot += v * f
        say(f"{k:34s} {v:10.3f} {v * f:24.3f}   ({note})")
    say(f"{'sum (oracle formulation)':34s} {sum(st.values()):10.3f} {tot:24.3f}")

    def label_loop(labels):                       # P:169-187: Python double loop over strings
        n = len(labels)
        m = torch.zeros((n, n))
        for i in range(n):
            for j in range(n):
                if labels[i] == labels[j]:
                    m[i, j] = 1
        return m

    def closure(adj):                             # P:250-274: n rounds of clamp(R @ A + A, 0, 1)
        r = adj.clone()
        for _ in range(adj.shape[0]):
            r = torch.clamp(r @ adj + adj, 0, 1)
        return r

    say("\nreference formulation of stage (ii), measured small and extrapolated to Ins =", ins_full)
    g = torch.Generator().manual_seed(0)
    for n in (256, 512, 1024):
        t0 = time.perf_counter(); label_loop(["table"] * n); t1 = time.perf_counter()
        a = (torch.rand((n, n), generator=g) < 4.0 / n).float()
        a = ((a + a.T) > 0).float()
        t2 = time.perf_counter(); closure(a); t3 = time.perf_counter()
        say(f"  Ins={n:5d}: label loop {t1 - t0:8.3f} s -> x(Ins/n)^2 = {(t1 - t0) * (ins_full / n) ** 2:12.1f} s;   "
            f"closure {t3 - t2:8.3f} s -> x(Ins/n)^4 = {(t3 - t2) * (ins_full / n) ** 4:14.1f} s")


MFMA_F16_PEAK_TFLOPS = 2500.0     # dense f16/bf16 MFMA peak of MI355X (MI355X_MICROARCH.md; not the 2:1 sparse figure)


def bench_cosine(args):
    """BASELINE config 5: CLIP ViT-L/14 768-d features x 200-class text bank on the MFMA path.
    One step = cos(A x 768, 200 x 768) for A = 9000 features (the instance count of config 2)."""
    dev = "cuda:0"
    torch.cuda.set_device(0)
    _lib.load()
    g = torch.Generator().manual_seed(0)
    a = torch.randn(9000, 768, generator=g).half().to(dev)
    b = torch.randn(200, 768, generator=g).half().to(dev)
    for _ in range(args.warmup):
        _lib.cosine_gemm_f16(a, b)
    torch.cuda.synchronize()
    # the kernel is shorter than a Python-level launch (allocate the output, look the stream up, ctypes): time it with
    # everything hoisted out of the loop, so that the events bracket kernels queued back to back
    out = torch.empty((a.shape[0], b.shape[0]), dtype=torch.float32, device=dev)
    fn = getattr(_lib.load(), "bff_cosine_gemm_f16")
    argv = (_lib._ptr(a), a.shape[0], _lib._ptr(b), b.shape[0], a.shape[1], _lib._ptr(out), _lib._stream())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(args.steps):
        fn(*argv)
    e1.record()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    ms = e0.elapsed_time(e1) / args.steps
    flop = 2.0 * a.shape[0] * b.shape[0] * a.shape[1]
    ref = (a.double() @ b.double().T) / (a.double().norm(dim=1, keepdim=True) * b.double().norm(dim=1, keepdim=True).T)
    err = float((out.double() - ref).abs().max())
    tf = flop / (ms * 1e-3) / 1e12
    print(json.dumps({
        "metric": "cosine GEMMs/sec (9000x768 f16 features x 200x768 f16 text bank)", "value": args.steps / elapsed,
        "unit": "gemms/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
        "config": {"workload": "c5: 9000 x 768 f16 features against a 200 x 768 f16 bank, f32 accumulate + normalise"},
        "roofline": {"bound": "mfma", "kernel": "cosine_gemm_f16_bank_kernel", "achieved": tf, "peak": MFMA_F16_PEAK_TFLOPS,
                     "unit": "TFLOP/s", "frac": tf / MFMA_F16_PEAK_TFLOPS, "traffic": None,
                     "note": "2.8 GFLOP per launch (SURVEY section 8d).  Bank-stationary kernel: one block per CU, wave t keeps the 16-column "
                             "strip t of the bank in registers for the whole k range (96 VGPRs), the block's 16-row tiles of A go through "
                             "LDS once; A is read from HBM once.  Measured split of the launch (parts switched off): empty shell 6.4 us, A "
                             "0.5 us, bank strips 7 us (16 half lines per load instruction), LDS fragment reads + MFMAs 6.4 us, 7.2 MB of "
                             "results 3.7 us.  Earlier forms: LDS-staged bank slices 26.4 us, one wave per 16x16 tile 52-60 us"},
        "max_abs_err_vs_f64": err}))


def host_cores():
    """Logical cores this process may run on (the GPU box gives a 1-GPU job a share of the host)."""
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


SCENE_VARIANTS = [
    # the scenes rotated through the timed loop (seed offsets); "many": 40 smaller objects, one exact full-silhouette
    # mask per visible object and view -> many stage-2 instances survive the filters, so the back half, the
    # refinement and the gather are timed on K >> 1; "default": the generator SURVEY 8(d) describes
    dict(kind="default"), dict(kind="many", cut_masks=False, n_objects=40, distinct_masks=True, dilate=False),
    dict(kind="default"), dict(kind="many", cut_masks=False, n_objects=40, distinct_masks=True, dilate=False),
]


def merge_traffic(ds, cfg, dev):
    """Counters of one tile pass (a diagnostic launch outside the timed region): tile pairs that reached the exact
    stage, 512-point chunks staged through LDS, candidate row pairs, unions pushed into the global forest."""
    fr = projection_front(ds, cfg, fast=False)
    area, _mw, cmask, hist, sig = _lib.row_stats(fr.rows, fr.cmask)
    order = _lib.argsort_i64(sig, _lib.SIGNATURE_BITS)
    d = torch.zeros(16, dtype=torch.int32, device=dev)
    _lib.merge_components(fr.rows, area, ds.label_id, cfg.iou_thres, order, cmask, hist, diag=d)
    v = d.cpu().tolist()
    if fr.arena is not None:
        fr.arena.release(fr.rows, fr.cmask)
    return dict(tile_pairs=v[0], chunk_visits=v[1], candidate_pairs=v[2], unions=v[3])


def compulsory_traffic(ds, dev):
    """Bytes the sweep HAS to move from HBM (measured on the device, outside the timed region): every 128-byte line
    of the depth images and of the mask images (label bytes or words, segment by segment) that some point's gather touches, once (bff_diag_sweep_lines);
    the cloud once per tile of 8 frames (xyz stays in registers across a tile); the two counters read+written once.
    Re-fetches of a line by other waves (served by L2 / Infinity Cache or not) are what `traffic` has on top."""
    import ctypes
    n, hw = ds.n_points, ds.height * ds.width
    n_mviews = ds.view_mask_offs.shape[0] - 1
    maskbits = torch.empty((n_mviews, hw), device=dev, dtype=torch.int32 if ds.word_bits == 32 else torch.int64)
    labels = torch.empty((n_mviews, _lib.label_plane_stride(hw)), device=dev, dtype=torch.uint8)
    segmap = torch.empty((n_mviews, 2 * _lib.segmap_words(hw)), dtype=torch.int32, device=dev)
    _lib.rle_to_labels(ds.run_start, ds.run_end, ds.mask_run_offs, ds.view_mask_offs, n_mviews, hw, ds.word_bits,
                       labels, maskbits, segmap)
    del maskbits, labels
    line_words = (((hw + 15) // 16 + 31) // 32 + 1) // 2 * 2
    if ds.depth_raw is not None:
        per_frame = ds.depth_raw.shape[1] * (ds.depth_raw.shape[2] if ds.depth_raw.dim() == 3 else 1)
        per_line = 128 // ds.depth_raw.element_size()
        line_words = max(line_words, (((per_frame + per_line - 1) // per_line + 31) // 32 + 1) // 2 * 2)
    dl = torch.zeros((ds.n_frames, line_words), dtype=torch.int32, device=dev)
    ml = torch.zeros((ds.n_frames, line_words), dtype=torch.int32, device=dev)
    ll = torch.zeros((ds.n_frames, line_words), dtype=torch.int32, device=dev)
    k = (ctypes.c_double * 9)(*[float(v) for v in ds.cam_intr.reshape(-1)])
    if ds.depth_raw is not None:
        _lib.call("bff_diag_sweep_lines_u16", _lib._ptr(ds.xyz), n, ds.xyz.shape[1], _lib._ptr(ds.inv_pose), ctypes.cast(k, ctypes.c_void_p),
                  ds.n_frames, _lib._ptr(ds.depth_raw), *(ds.depth_size if ds.depth_size is not None else ds.depth_raw.shape[1:3]),
                  0 if ds.depth_size is None else (2 if ds.depth_raw.dtype == torch.float32 else 1), _lib._ptr(ds.depth_index),
                  ds.height, ds.width, 0.08, _lib._ptr(segmap), ds.word_bits, _lib._ptr(ds.frame_mask), _lib._ptr(dl), _lib._ptr(ml),
                  line_words, _lib._ptr(ll))
    else:
        _lib.call("bff_diag_sweep_lines", _lib._ptr(ds.xyz), n, ds.xyz.shape[1], _lib._ptr(ds.inv_pose), ctypes.cast(k, ctypes.c_void_p),
                  ds.n_frames, _lib._ptr(ds.depth), _lib._ptr(ds.depth_index), ds.height, ds.width, 0.08, _lib._ptr(segmap),
                  ds.word_bits, _lib._ptr(ds.frame_mask), _lib._ptr(dl), _lib._ptr(ml), line_words, _lib._ptr(ll))
    count = lambda t: int(_lib.popcount_rows(t.view(torch.int64)).sum().item())
    depth_lines, mask_lines, label_lines = count(dl), count(ml), count(ll)
    tiles = (ds.n_frames + 7) // 8
    xyz_bytes, counter_bytes = 24 * n * tiles, 16 * n
    return {"bytes": 128 * (depth_lines + mask_lines + label_lines) + xyz_bytes + counter_bytes,
            "depth_lines_128B": depth_lines, "depth_format": ("sensor-resolution frames, " + str(ds.depth_raw.dtype).replace("torch.", "")) if ds.depth_raw is not None else "f32 (H, W)", "mask_word_lines_128B": mask_lines, "mask_label_lines_128B": label_lines, "xyz_bytes (once per 8-frame tile)": xyz_bytes,
            "counter_bytes": counter_bytes}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=25,
                    help="timed steps; a step is ONE QUERY CLASS per GPU: --class-batch (8) scenes through projection + the class's "
                         "refinement (one similarity exchange + one gather at N > 1)")
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--shape", default="c2", choices=list(SHAPES) + ["c5"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-stagewise", action="store_true",
                    help="print the stage-wise CPU (oracle) timings of SURVEY 8(d) for --shape and exit")
    ap.add_argument("--cpu-sample-views", type=int, default=100)
    ap.add_argument("--no-pipeline", action="store_true",
                    help="finish every scene before starting the next (default: the device work of the next "
                         "scene is issued on a second HIP stream while the host finishes the current one)")
    ap.add_argument("--host-profile", action="store_true", help="cProfile of the timed loop to stderr")
    ap.add_argument("--depth", type=int, default=PIPELINE_DEPTH, help="scenes in flight on the device (one HIP stream each)")
    ap.add_argument("--scenes", type=int, default=4, help="resident scenes rotated through the timed loop")
    ap.add_argument("--depth-format", choices=["u16", "f32"], default="u16",
                    help="u16 (default): depth resident as the PNGs store it (uint16 mm at half the working resolution, "
                         "ScanNet's sensor ratio), /1000 + bilinear resize evaluated per point inside the sweep (P:431-436); "
                         "f32: float32 (H, W) images as after cv2.resize")
    ap.add_argument("--class-batch", type=int, default=8,
                    help="scenes per GPU that form one query class: one similarity exchange + one gather per batch")
    ap.add_argument("--include-upload", action="store_true", help="(kept for old command lines: the host-inclusive leg always runs at N = 1)")
    ap.add_argument("--no-host-inclusive", action="store_true",
                    help="skip the host-inclusive leg (every step takes host arrays -- reference formats, raw 16-bit depth -- "
                         "through the overlapped ingestion pipeline before the device path; reported as `host_inclusive`)")
    args = ap.parse_args()
    # the host side of the GPU path makes only tiny torch CPU calls; left at the default every one of them opens an OpenMP
    # region as wide as the host (256 logical cores on the GPU box).  cpu_baseline sets its own thread count afterwards.
    torch.set_num_threads(int(os.environ.get("BFF_TORCH_THREADS", "4")))
    if args.shape == "c5":
        return bench_cosine(args)
    if args.cpu_stagewise:
        return cpu_stagewise(args)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # BFF_REHEARSE_ON_ONE_GPU=1: all ranks share cuda:0 and the collectives run over gloo -- only to rehearse
    # the N > 1 code path on a single-GPU box; the driver's runs use one GPU per rank over RCCL.
    rehearse = os.environ.get("BFF_REHEARSE_ON_ONE_GPU") == "1"
    if rehearse:
        local_rank = 0
    # BFF_FORCE_COLLECTIVES=1 at N = 1: a process group of ONE rank over RCCL, and every class makes its exchange and its
    # gather although nobody else is there -- what a single-GPU box can measure of the RCCL branch (config.collectives)
    forced = world == 1 and os.environ.get("BFF_FORCE_COLLECTIVES") == "1"
    if forced:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(f"cuda:{local_rank}"))
    if world > 1:
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local_rank)
    dev = f"cuda:{local_rank}"
    _lib.load()

    # ---- resident scenes of this rank (different seeds and two generator variants), uploaded once: inputs in HBM
    n, v, h, w, m = SHAPES[args.shape]
    t0 = time.perf_counter()
    n_scenes = max(1, args.scenes)
    scenes, dss, stage1s = [], [], []
    cfg = None
    for k in range(n_scenes):
        var = dict(SCENE_VARIANTS[k % len(SCENE_VARIANTS)])
        var.pop("kind")
        sc = make_scene(args.shape, seed=rank * n_scenes + k, device=dev, query=QUERY, **var)
        cfg = Config.with_defaults(width_2d=sc.width, height_2d=sc.height)
        if args.depth_format == "u16":
            sc = with_sensor_depth(sc)
        scenes.append(sc)
        dss.append(prepare_scene(sc, cfg, device=dev))
        stage1s.append(prepare_stage1(sc.stage1, dev))          # class-independent: uploaded with the scene
    bank, index = make_text_bank(768, seed=0)                  # float16, like CLIP text features on a GPU
    enc = bank_encoder(bank, index)
    sim = TextSimilarity(enc, dev)
    t_setup = time.perf_counter() - t0
    exchange = bdist.ClassExchange("cpu" if rehearse else dev) if (world > 1 or forced) else None     # sims + result-size bounds, one all-gather per class
    if world > 1 or forced:
        dist.barrier()

    cbatch = max(1, args.class_batch)
    n_timed = args.steps * cbatch                      # scenes per rank inside the timed region: a step is one class batch
    timers = KernelTimers(reserve=4 * n_timed)         # events are created before the clock starts
    depth = max(2, args.depth)
    streams = scene_streams(dev, depth)
    host = {"front_issue_s": 0.0, "back_s": 0.0, "class_finish_s": 0.0}     # host wall time per part (back includes its sync wait)
    results, last_class = {}, {}

    trace_front = [] if os.environ.get("BFF_BENCH_TRACE_FRONT") else None
    def front(i, tm=None):
        t = time.perf_counter()
        k = i % n_scenes
        with _lib.on_stream(streams[i % depth]):
            fr = projection_front(dss[k], cfg, timers=tm, stage1=stage1s[k])
        host["front_issue_s"] += time.perf_counter() - t
        if trace_front is not None:
            trace_front.append(round((time.perf_counter() - t) * 1e3, 3))
        return fr

    def run_steps(k, tm=None):
        """k scenes per rank, one after the other through the whole path, in query classes of `cbatch` scenes per rank.
        Pipelined form: while the host works on the back half of scene i (header, size filter, refinement pass 1 --
        and, at the end of a class, the exchange, pass 2 and the gather), the device work of scenes i+1 .. i+depth-1
        already runs on the other streams; class boundaries do not drain the pipeline."""
        cur = {"batch": None}

        def back(i, fr):
            t = time.perf_counter()
            ks = i % n_scenes
            j = i % cbatch
            with _lib.on_stream(streams[i % depth]):
                res = projection_back(fr, want_groups=False)
                if j == 0:
                    nb = min(cbatch, k - i)             # scenes of this class on every rank
                    ids = [f"r{r}s{q}" for r in range(world) for q in range(nb)]
                    cur["batch"] = bdist.ClassBatch(cfg, QUERY, sim, dev, ids, nb, exchange=exchange)
                    cur["n"] = nb
                batch = cur["batch"]
                batch.add(f"r{rank}s{j}", stage1s[ks], res)
                results[ks] = [int(res.rows.shape[0]), None]
                if j + 1 == cur["n"]:
                    t1 = time.perf_counter()
                    batch.finish()                      # ONE exchange, pass 2, ONE gather (rank 0 keeps the buffers)
                    host["class_finish_s"] += time.perf_counter() - t1
                    for q in range(cur["n"]):
                        f = batch.final[f"r{rank}s{q}"]
                        results[(i - cur["n"] + 1 + q) % n_scenes][1] = 0 if f.rows is None else int(f.rows.shape[0])
                    last_class["batch"] = batch
            host["back_s"] += time.perf_counter() - t

        for _ in pipelined(k, lambda i: front(i, tm), back, 1 if args.no_pipeline else depth):
            pass

    # Priming (setup, not part of the W warm-up steps the contract asks for): the first ~14 scene calls of a process
    # include one-time costs -- a workspace per stream (~6 ms each) and three more calls that block ~6 ms inside the
    # HIP runtime while its per-queue pools grow (traced with BFF_TRACE_ISSUE=1; none afterwards).  A driver run with
    # --warmup 5 --steps 20 would otherwise time those instead of the steady state.
    priming = max(0, 16 - args.warmup * cbatch)
    run_steps(priming)
    run_steps(args.warmup * cbatch)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    fence()
    host.update(front_issue_s=0.0, back_s=0.0, class_finish_s=0.0)
    _lib.sync_wait_s = 0.0
    if exchange is not None:
        exchange.calls = 0
    prof = None
    if args.host_profile:                              # where the host thread's time goes (stderr; slows the loop)
        import cProfile
        prof = cProfile.Profile()
        prof.enable()
    t0 = time.perf_counter()
    run_steps(n_timed, timers)
    fence()
    elapsed = time.perf_counter() - t0
    if prof is not None:
        import pstats
        prof.disable()
        pstats.Stats(prof, stream=sys.stderr).sort_stats("tottime").print_stats(35)
    n_exchanges = exchange.calls if exchange is not None else 0
    host_ms = {"front_issue": round(host["front_issue_s"] / n_timed * 1e3, 4),          # per SCENE
               "back": round(host["back_s"] / n_timed * 1e3, 4),
               "of_back_class_finish": round(host["class_finish_s"] / n_timed * 1e3, 4),
               "of_which_waiting_for_gpu": round(_lib.sync_wait_s / n_timed * 1e3, 4)}
    host_ms["host_work"] = round(host_ms["front_issue"] + host_ms["back"] - host_ms["of_which_waiting_for_gpu"], 4)
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearse else dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # the two big kernels alone on the chip: a short strictly sequential loop after the timed region (in the
    # pipelined loop they share the GPU with the other scene's kernels)
    seq_timers = KernelTimers(reserve=8 * n_scenes)
    if not args.no_pipeline:
        was = args.no_pipeline
        args.no_pipeline = True
        run_steps(min(2 * n_scenes, n_timed), seq_timers)
        args.no_pipeline = was
        fence()

    # device span of the one native call per scene (first kernel start -> last kernel end on an otherwise idle GPU,
    # no host involvement in between): what the chain of ~50 launches costs end to end, per scene variant
    spans = []
    for k in range(n_scenes):
        st = streams[k % depth]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record(st)
        fr = front(k)
        e1.record(st)
        with torch.cuda.stream(st):
            projection_back(fr, want_groups=False)
        torch.cuda.synchronize()
        spans.append(round(e0.elapsed_time(e1), 4))

    upload_leg = None
    if world == 1 and not args.no_host_inclusive:
        from beyond_fixed_forms_amd.ingest import bench_host_inclusive
        upload_leg = bench_host_inclusive(scenes, cfg, dev, QUERY, sim, steps=min(max(n_timed, 24), 48),
                                          n_loaders=int(os.environ.get("BFF_BENCH_LOADERS", "4")),
                                          native_threads=int(os.environ.get("BFF_BENCH_NATIVE_THREADS", "4")))

    # what rank 0 holds after the last class: every rank's final masks, decoded from the gathered buffers only now
    gathered_check = None
    if last_class.get("batch") is not None:
        lb = last_class["batch"]
        got = lb.results()
        if rank == 0:
            own = {sid: f for sid, f in lb.final.items()}
            same_own = all((got[sid].rows is None and f.rows is None) or
                           (got[sid].rows is not None and f.rows is not None and torch.equal(got[sid].rows.to(dev), f.rows))
                           for sid, f in own.items())
            gathered_check = {"scenes_on_rank0": len(got), "expected": len(lb.ids), "rank0_rows_round_trip": bool(same_own),
                              "final_masks_total": int(sum(0 if f.rows is None else f.rows.shape[0] for f in got.values()))}
            assert len(got) == len(lb.ids) and same_own, gathered_check

    if trace_front is not None and rank == 0:
        print("front() ms per call:", trace_front, file=sys.stderr)
    if rank == 0:
        ks = timers.summary()
        sq = seq_timers.summary() if not args.no_pipeline else ks
        ds = dss[0]
        n_swept = ds.n_frames
        pv = ks["project_views"]
        abytes, zbytes = algorithmic_bytes(ds, n_swept)
        achieved = abytes / (pv[2] * 1e-3) / 1e9
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tfile):
            with open(tfile) as f:
                traffic = json.load(f).get(f"project_views_{args.shape}" + ("_u16" if ds.depth_raw is not None else ""))
        mt = merge_traffic(ds, cfg, dev)
        comp = compulsory_traffic(ds, dev)
        t_alone = sq["project_views"][2] * 1e-3
        comp.update(achieved=comp["bytes"] / t_alone / 1e9, frac=comp["bytes"] / t_alone / 1e9 / HBM_PEAK_GBS,
                    note="HBM bytes the sweep cannot avoid (each touched 128-B line once, cloud once per frame tile) over its "
                         "duration alone on the chip; `achieved`/`frac` above use SURVEY 8(d)'s algorithmic bytes, which "
                         "charge 28 B per (frame, point) even where caches serve them")
        mc_ms, mc_alone = ks["merge_components"][2], sq["merge_components"][2]
        l2_bytes = mt["chunk_visits"] * 128 * 64            # every visited chunk: 128 rows x 8 words staged through LDS
        out = {
            "metric": "scenes/sec (2D->3D projection+refinement), 200k pts x 300 views",
            "value": world * n_timed / elapsed, "unit": "scenes/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{args.shape}: one query class of {cbatch} scenes per GPU and step (every scene through projection "
                                   f"and refinement pass 1, then the class's threshold + pass 2), each scene {n} pts x {len(scenes[0].mask_2d)} mask views + "
                                   f"{ds.n_viewed} viewed frames @{h}x{w}, {m} masks/view (Ins={ds.n_rows}), "
                                   f"stage-1 S1={len(scenes[0].stage1['ins'])}, 198x768 f16 text bank; inputs RESIDENT in "
                                   f"HBM (uploaded before the timed region); depth " +
                                   (f"at the sensor's resolution ({'x'.join(str(int(v)) for v in (ds.depth_size or ds.depth_raw.shape[1:3]))}, "
                                    f"{'float32 metres = uint16 / 1000' if ds.depth_raw.dtype == torch.float32 else 'uint16 mm'}"
                                    f"{', 8x8-texel tiles' if ds.depth_size else ''}), bilinear resize "
                                    f"per point inside the sweep" if ds.depth_raw is not None else "float32 (H, W)") +
                                   f"; {n_scenes} different scenes rotate through the loop",
                       "scenes_per_step": world * cbatch, "scenes_in_timed_region": world * n_timed,
                       "priming_scenes_in_setup": priming,
                       "sharding": "one query class of class_batch scenes per GPU and step (the reference thresholds once per class, "
                                   "R:316-324): ONE all-gather of similarity sets + ONE RCCL gather of final masks per step (none at "
                                   "N = 1); `--class-batch 1` is round 2's one-scene step",
                       "class_batch": cbatch, "similarity_exchanges_in_timed_region": n_exchanges,
                       "collectives": ("RCCL, forced on a process group of one rank (BFF_FORCE_COLLECTIVES=1)" if forced else
                                       "none (N = 1)" if world == 1 else "gloo rehearsal on one GPU" if rehearse else "RCCL"),
                       "pipelining": "none" if args.no_pipeline else
                       f"{depth} HIP streams: the device work of the next {depth - 1} scene(s) overlaps the host half of scene i",
                       "scene_variants": [SCENE_VARIANTS[k % len(SCENE_VARIANTS)]["kind"] for k in range(n_scenes)]},
            "roofline": {"bound": "hbm", "kernel": "project_views_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic["bytes"] if traffic else None,      # HBM bytes per launch from the PMC counters
                         "traffic_detail": traffic,
                         "algorithmic_bytes_per_launch": abytes, "avg_launch_ms": pv[2], "launches": pv[0],
                         "survey_row_write_bytes_not_credited": zbytes,
                         "note": "avg_launch_ms is the kernel's duration INSIDE the timed loop, where it shares the chip with the "
                                 "kernels of the other scenes in flight (HIP events on the dispatch); `alone_on_chip` is the same "
                                 "kernel in a strictly sequential loop right after it and is what a rocprofv3 kernel trace "
                                 "shows, because the profiler serialises the streams (profiles/README.md)",
                         "compulsory": comp,
                         "alone_on_chip": {"avg_launch_ms": sq["project_views"][2],
                                           "achieved": abytes / (sq["project_views"][2] * 1e-3) / 1e9,
                                           "frac": abytes / (sq["project_views"][2] * 1e-3) / 1e9 / HBM_PEAK_GBS}},
            # second bound: the components' tile pass is AND + popcount over LDS-staged chunks -- not HBM, not MFMA
            "roofline_merge": {"bound": "l2->lds staging + VALU popcount", "kernel": "merge_components_kernel",
                               "avg_launch_ms": mc_ms, "alone_on_chip_ms": mc_alone, "launches": ks["merge_components"][0],
                               "l2_bytes_staged": l2_bytes, "achieved": l2_bytes / (mc_alone * 1e-3) / 1e9,
                               "peak": 34500.0, "unit": "GB/s (L2, MI355X_MICROARCH.md)",
                               "frac": l2_bytes / (mc_alone * 1e-3) / 1e9 / 34500.0, **mt},
            "host_ms": host_ms, "scene_call_device_span_ms": spans,      # wall time of the host thread per SCENE: issuing, the host half, and the part of it spent blocked on the GPU
            "kernels_ms": {k: round(vv[2], 4) for k, vv in ks.items()},   # the kernels' own durations (events on the dispatch)
            "result": {"per_scene (stage2_instances, final_masks)": [results.get(k) for k in range(n_scenes)],
                       "last_class_on_rank0": gathered_check},
            "setup_s": round(t_setup, 1),
        }
        if upload_leg is not None:
            out["host_inclusive"] = upload_leg
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(scenes[0], cfg, bank_encoder(bank.float(), index))
        print(json.dumps(out))
    if world > 1 or forced:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
