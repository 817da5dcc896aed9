"""CPU tests of the host-side logic (no GPU, no compute calls into the HIP library)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import golden_io as gio
from oracle import projection_ref as pref, rle_ref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cabi_library_loads_and_exports_every_declared_symbol():
    from beyond_fixed_forms_amd import _lib
    header = open(os.path.join(ROOT, "include", "bff_hip.h")).read()
    declared = set(re.findall(r"\b(bff_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    assert os.path.exists(_lib.LIB_PATH), "run __graft_entry__.build() first"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    missing = [s for s in sorted(declared) if not hasattr(lib, s)]
    assert not missing, missing
    assert declared == set(_lib.SIGNATURES) | set(_lib.PLAIN), "python binding table out of sync with the header"
    lib.bff_abi_version.restype = ctypes.c_int32
    lib.bff_arch.restype = ctypes.c_char_p
    assert lib.bff_abi_version() == _lib.ABI_VERSION and lib.bff_arch() == b"gfx950"


def test_entry_points_reject_bad_arguments_without_a_gpu():
    """Argument validation happens on the host before any launch."""
    from beyond_fixed_forms_amd import _lib
    lib = _lib.load()
    assert lib.bff_popcount_rows(None, None, -1, 0, None, None) == -1
    assert b"bad sizes" in lib.bff_last_error()
    assert lib.bff_cosine_gemm_f16(None, 1, None, 1, 33, None, None) == -1
    assert lib.bff_rle_to_maskbits(None, None, None, None, 1, 10, 48, None, None, None) == -1
    assert lib.bff_popcount_rows(None, None, 0, 0, None, None) == 0          # empty work is fine


def test_size_limits_are_rejected_on_the_host():
    """Maximum sizes (DESIGN: 2^31 pixels per image, 4096 row chunks = 2.1 M points per call, 4096 rows for the
    one-pass overlap resolution): beyond them every entry point answers BFF_E_LIMIT before touching the GPU."""
    import ctypes
    from beyond_fixed_forms_amd import _lib
    lib = _lib.load()
    buf = ctypes.create_string_buffer(64)
    p = ctypes.cast(buf, ctypes.c_void_p)
    too_many_words = 8 * 4096 + 8
    assert lib.bff_row_stats(p, 1, too_many_words, p, p, p, 0, p, p, None, None) == -2 and b"4096 chunks" in lib.bff_last_error()
    assert lib.bff_merge_components(p, 64, too_many_words, None, 64, p, p, p, p, p, p, 0.2, p, 1, p, None, None, None) == -2
    assert lib.bff_resolve_overlaps(p, 4097, 10, p, None, p, p, None) == -2 and b"4096 rows" in lib.bff_last_error()
    assert lib.bff_resolve_overlaps_max_rows() == 4096
    assert lib.bff_rle_to_maskbits(p, p, p, p, 1, 1 << 31, 32, p, None, None) == -2
    assert lib.bff_rle_to_labels(p, p, p, p, 1, 1 << 31, 32, p, p, None, None) == -2
    assert lib.bff_rle_to_labels(p, p, p, p, 1, 100, 32, None, p, None, None) == -1
    assert lib.bff_project_views(p, 10, 1024, p, p, 1, p, p, 65536, 65536, 0.08, None, None, None, 32, None, None, None, p,
                                 None, 0, 1, None, None, None, None, None) == -2 and b"2^31 pixels" in lib.bff_last_error()


def test_no_cpu_fallback():
    from beyond_fixed_forms_amd import _lib
    with pytest.raises(ValueError):
        _lib.popcount_rows(torch.zeros((2, 2), dtype=torch.int64))            # CPU tensor is refused


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "beyond_fixed_forms_amd")
    for name in os.listdir(pkg):
        if name.endswith(".py"):
            src = open(os.path.join(pkg, name)).read()
            assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M), name


def test_runs_from_rles_matches_reference_decoder():
    from beyond_fixed_forms_amd.scene import runs_from_rles
    z = np.load(os.path.join(gio.GOLDEN_DIR, "refine_helpers.npz"))
    rles = gio.unpack_rles(z["rle1d.len"], z["rle1d.counts"], z["rle1d.offs"])
    rs, re_, offs = runs_from_rles(rles)
    for i, r in enumerate(rles):
        got = np.zeros(int(r["length"]), np.uint8)
        for a, b in zip(rs[offs[i]:offs[i + 1]], re_[offs[i]:offs[i + 1]]):
            got[a:b] = 1
        assert np.array_equal(got, z[f"rle1d.dec{i}"]), i
        seg_s, seg_e = rs[offs[i]:offs[i + 1]], re_[offs[i]:offs[i + 1]]
        assert np.all(seg_e > seg_s) and np.all(seg_s[1:] >= seg_e[:-1])
    assert runs_from_rles([])[2].tolist() == [0]


def test_groups_from_labels_matches_closure_components():
    from beyond_fixed_forms_amd.projection import groups_from_labels
    rng = np.random.default_rng(0)
    for trial in range(20):
        n = int(rng.integers(1, 60))
        a = rng.random((n, n)) < 0.05
        a = a | a.T
        alive = rng.random(n) < 0.85
        a[np.arange(n), np.arange(n)] = True
        a[~alive] = False
        a[:, ~alive] = False
        exp = pref.connected_groups(torch.from_numpy(a).float())
        # component label = smallest member index (what the device propagation converges to)
        label = np.arange(n)
        for comp in exp:
            for i in comp:
                label[i] = comp[0]
        assert groups_from_labels(label.astype(np.int32), alive) == exp
        assert groups_from_labels(label.astype(np.int32), alive, 2) == [g for g in exp if len(g) >= 2]
        from beyond_fixed_forms_amd.projection import component_csr
        for mm in (0, 1, 2, 3):
            offs, members, sizes, n_void = component_csr(label.astype(np.int32), alive, mm)
            want = [g for g in exp if len(g) >= max(mm, 1)]
            assert [members[offs[g]:offs[g + 1]].tolist() for g in range(len(sizes))] == want
            assert sizes.tolist() == [len(g) for g in want]
            assert n_void == (sum(1 for g in exp if g == []) if mm <= 0 else 0)
            from beyond_fixed_forms_amd import _lib
            o2, m2, s2, v2 = _lib.host_component_csr(label.astype(np.int32), alive, mm)      # native twin
            assert np.array_equal(o2, offs) and np.array_equal(m2, members) and np.array_equal(s2, sizes) and v2 == n_void
    from beyond_fixed_forms_amd import _lib
    assert _lib.host_component_csr(np.array([0, 5], np.int32), np.ones(2, bool), 1) is None     # id out of range
    # a node without a self loop that has a neighbour is a normal member (iou_thres < 0 with an empty mask)
    a = np.array([[1, 1, 0], [1, 0, 0], [0, 0, 0]], bool)
    assert pref.connected_groups(torch.from_numpy(a).float()) == [[0, 1], []]
    assert groups_from_labels(np.array([0, 0, 2]), np.array([True, False, False])) == [[0, 1], []]


def test_prepare_scene_frame_table_cpu():
    """Frame table / run tables are built on the host; check them without touching a GPU."""
    from beyond_fixed_forms_amd.config import Config
    from beyond_fixed_forms_amd.scene import prepare_scene
    from beyond_fixed_forms_amd.synthetic import make_scene
    sc = make_scene("tiny", seed=3, n_masks=70, n_views=3)
    cfg = Config.with_defaults(width_2d=sc.width, height_2d=sc.height)
    ds = prepare_scene(sc, cfg, device="cpu")
    assert ds.word_bits == 64 and ds.n_rows == 70 * len(sc.mask_2d)
    assert ds.frame_nmask.tolist()[:2] == [64, 6] and ds.frame_rowbase.tolist()[:3] == [0, 64, 70]
    assert ds.frame_flags.tolist()[:2] == [1, 0]                     # a chunked frame is counted once
    assert int(ds.frame_flags.sum()) == ds.n_viewed == 3
    assert ds.xyz.shape[1] % 1024 == 0 and ds.nw == (sc.points.shape[0] + 63) // 64
    assert np.allclose(ds.inv_pose[0].reshape(4, 4).numpy() @ sc.poses["0"], np.eye(4), atol=1e-12)
    sc2 = make_scene("tiny", seed=3)
    sc2.mask_2d[0]["segmented_frame_masks"][0]["length"] = 5
    with pytest.raises(ValueError):
        prepare_scene(sc2, Config.with_defaults(width_2d=sc2.width, height_2d=sc2.height), device="cpu")


def test_disk_round_trip_and_cli_paths(tmp_path):
    """The reference's directory layout -> load_scene gives back the same inputs (depth via 16-bit PNG)."""
    from PIL import Image
    from beyond_fixed_forms_amd import io
    from beyond_fixed_forms_amd.config import Config
    from beyond_fixed_forms_amd.synthetic import make_scene
    sc = make_scene("tiny", seed=4)
    cfg = Config.with_defaults(width_2d=sc.width, height_2d=sc.height, scene_2d_dir=str(tmp_path / "2d"),
                               scene_npy_dir=str(tmp_path / "npy"), mask_2d_dir=str(tmp_path / "m2d"))
    sd = tmp_path / "2d" / sc.scene_id
    for sub in ("intrinsic", "pose", "depth", "color"):
        (sd / sub).mkdir(parents=True)
    (tmp_path / "npy").mkdir(); (tmp_path / "m2d" / "table").mkdir(parents=True)
    np.savetxt(sd / "intrinsic" / "intrinsic_color.txt", sc.cam_intr)
    np.save(tmp_path / "npy" / f"{sc.scene_id}.npy", sc.points)
    for f in sc.color_files:
        (sd / "color" / f).write_bytes(b"")
    for fid, pose in sc.poses.items():
        np.savetxt(sd / "pose" / f"{fid}.txt", pose)
        mm = np.round(sc.depths[fid].astype(np.float64) * 1000).astype(np.uint16)
        Image.fromarray(mm).save(sd / "depth" / f"{fid}.png")
    torch.save(sc.mask_2d, tmp_path / "m2d" / "table" / f"{sc.scene_id}.pth")
    got = io.load_scene(cfg, "table", sc.scene_id)
    assert np.array_equal(got.points, sc.points) and np.allclose(got.cam_intr, sc.cam_intr)
    for fid in sc.poses:
        assert np.array_equal(got.depths[fid], sc.depths[fid])           # same size -> resize is the identity
        assert np.allclose(got.poses[fid], sc.poses[fid])
    assert sorted(got.color_files) == sorted(sc.color_files) and len(got.mask_2d) == len(sc.mask_2d)
    up = io.resize_bilinear_f32(sc.depths["0"], 2 * sc.width, 2 * sc.height)
    assert up.shape == (2 * sc.height, 2 * sc.width) and up.dtype == np.float32
    assert abs(float(up.mean()) - float(sc.depths["0"].mean())) < 0.02
    assert io.scene_checkpoint_file("refinement", "table") == "checkpoints/refinement_checkpoint_table.yaml"
    # the pose / intrinsic reader == np.loadtxt on the files np.savetxt wrote (and on hand-written ScanNet-style text)
    for fid in list(sc.poses)[:3]:
        pth = sd / "pose" / f"{fid}.txt"
        assert np.array_equal(io.read_matrix_txt(str(pth)), np.loadtxt(pth))
    hand = tmp_path / "hand.txt"
    hand.write_text("1170.187988 0.000000 647.750000 0.000000\n0.000000 1170.187988 483.750000 0.000000\n"
                    "0 0 1 0\n-1.5e-3 2E+1 0.1 1\n")
    assert np.array_equal(io.read_matrix_txt(str(hand)), np.loadtxt(hand))


def test_sim_threshold_semantics():
    from beyond_fixed_forms_amd.refinement import sim_threshold
    assert sim_threshold([[0.5, 0.1], [], [0.1, 0.9, 0.3]], 0.2) == 0.1        # sorted set: .1 .3 .5 .9 -> [0]
    assert sim_threshold([[0.5, 0.1], [0.9, 0.3, 0.7]], 0.2) == 0.3            # 5 values -> index 1
    with pytest.raises(IndexError):
        sim_threshold([[]], 0.2)


def test_bilinear_resize_hand_derived_fixture():
    """cv2.resize(INTER_LINEAR) restatement against a fixture derived by hand from the definition (not from the
    code): 2x3 -> 5x7.  Source x of destination column dx is (dx + 0.5) * 3/7 - 0.5 = -2/7, 1/7, 4/7, 1, 10/7,
    13/7, 16/7: column 0 lies left of the image (copies column 0), column 6 has floor 2 = last column (copies it),
    the others blend columns (0,1) or (1,2) with fractions 1/7, 4/7, 0, 3/7, 6/7.  Source y of row dy is
    (dy + 0.5) * 2/5 - 0.5 = -0.3, 0.1, 0.5, 0.9, 1.3: rows 0 and 4 clamp both taps onto one row, so they equal
    that row; rows 1..3 blend rows (0,1) with 0.1, 0.5, 0.9.  cv2 itself is absent here: parity with it stays
    unpinned, this pins the restatement to the published definition."""
    from beyond_fixed_forms_amd import io
    src = np.array([[10.0, 20.0, 40.0], [110.0, 220.0, 440.0]], np.float32)
    got = io.resize_bilinear_f32(src, 7, 5)
    assert got.shape == (5, 7) and got.dtype == np.float32
    fx = [None, 1 / 7, 4 / 7, 0.0, 3 / 7, 6 / 7, None]
    left = [0, 0, 0, 1, 1, 1, 2]
    hrow = lambda r: np.array([r[left[d]] if fx[d] is None else r[left[d]] * (1 - fx[d]) + r[left[d] + 1] * fx[d]
                               for d in range(7)])
    h0, h1 = hrow(src[0].astype(np.float64)), hrow(src[1].astype(np.float64))
    fy = [None, 0.1, 0.5, 0.9, None]
    exp = np.stack([h0, h0 * 0.9 + h1 * 0.1, h0 * 0.5 + h1 * 0.5, h0 * 0.1 + h1 * 0.9, h1])
    assert np.abs(got.astype(np.float64) - exp).max() <= 440 * 2.0 ** -21          # float32 arithmetic, few ulps
    assert np.array_equal(got[:, 0], got[:, 0]) and got[0, 0] == 10.0 and got[4, 6] == 440.0     # corners are copies
    assert np.array_equal(got[0, [0, 3, 6]], src[0]) and np.array_equal(got[4, [0, 3, 6]], src[1])
    # tap tables: the x border taps have fraction exactly 0, the y fraction is kept when the rows clamp
    x0, x1, ax, y0, y1, ay = io.bilinear_taps(2, 3, 5, 7)
    assert x0.tolist() == [0, 0, 0, 1, 1, 1, 2] and ax[0] == 0 and ax[6] == 0 and ax[3] == 0
    assert y0.tolist() == [0, 0, 0, 0, 1] and y1.tolist() == [0, 1, 1, 1, 1]
    assert abs(float(ay[0]) - 0.7) < 1e-6 and abs(float(ay[4]) - 0.3) < 1e-6
    # the fraction is float32(source coordinate) - floor, as cv2 computes it (not the float64 fraction cast down)
    big = io.bilinear_taps(480, 640, 968, 1296)
    f = np.float32((np.float64(1295) + 0.5) * (1.0 / (1296 / 640)) - 0.5)
    assert big[2][1295] == 0.0 and big[0][1295] == 639
    f = np.float32((np.float64(1000) + 0.5) * (1.0 / (1296 / 640)) - 0.5)
    assert big[2][1000] == np.float32(f - np.float32(np.floor(f)))
    # identity
    assert np.array_equal(io.resize_bilinear_f32(src, 3, 2), src)


def test_native_run_tables_match_the_numpy_builder():
    """libbff_host.so (ingestion): RLE dicts -> run tables with the decode semantics of rle_decode_batch (RLE:45-57)
    == scene.runs_from_rles up to the empty runs the native builder keeps in place (start == end: they decode to
    nothing); inputs it declines (overlapping / unsorted runs) return None, malformed ones raise like the NumPy path."""
    from beyond_fixed_forms_amd import ingest
    from beyond_fixed_forms_amd.scene import runs_from_rles
    from beyond_fixed_forms_amd.synthetic import make_scene

    class HostStaging(ingest.Staging):           # pageable stand-in: pinned memory needs a GPU runtime
        def get(self, name, nbytes):
            t = self.buf.get(name)
            if t is None or t.numel() < nbytes:
                t = self.buf[name] = torch.empty(max(int(nbytes), 64), dtype=torch.uint8)
            return t

    st = HostStaging()
    sc = make_scene("tiny", seed=6, n_masks=7)
    rles = [r for fr in sc.mask_2d for r in fr["segmented_frame_masks"]]
    hw = sc.height * sc.width
    for threads in (1, 3):
        got = ingest.pack_rles(rles, hw, st, "m", threads)
        exp = runs_from_rles(rles)
        assert all(np.array_equal(g.numpy(), e) for g, e in zip(got, exp))
    # int32 counts, an empty mask, a full mask, a run clipped by the end, a run entirely past the end, zero-length runs
    odd = [dict(length=50, counts=np.array([], dtype=np.int64)), dict(length=50, counts=np.array([1, 50])),
           dict(length=50, counts=np.array([3, 4, 20, 1, 45, 10], dtype=np.int32)), dict(length=50, counts=np.array([60, 5])),
           dict(length=50, counts=np.array([5, 0, 5, 3, 8, 0, 9, 2]))]
    rs, re, offs = (t.numpy() for t in ingest.pack_rles(odd, 50, st, "o", 2))
    dense = lambda s, e, o: [sorted(set(p for k in range(o[g], o[g + 1]) for p in range(s[k], e[k]))) for g in range(len(o) - 1)]
    assert dense(rs, re, offs) == dense(*runs_from_rles(odd))
    assert all(np.all(np.diff(re[offs[g]:offs[g + 1]]) >= 0) for g in range(len(odd)))       # ends stay monotone
    assert ingest.pack_rles([dict(length=50, counts=np.array([10, 10, 15, 10]))], 50, st, "x") is None     # overlap: slow path
    assert ingest.pack_rles([dict(length=50, counts=np.array([20, 2, 5, 2]))], 50, st, "x") is None        # unsorted
    assert ingest.pack_rles([dict(length=50, counts=np.array([1.0, 2.0]))], 50, st, "x") is None           # not integers
    with pytest.raises(ValueError):
        ingest.pack_rles([dict(length=50, counts=np.array([0, 3]))], 50, st, "x")                          # start < 1
    with pytest.raises(ValueError):
        ingest.pack_rles([dict(length=50, counts=np.array([1, 3, 7]))], 50, st, "x")                       # odd
    with pytest.raises(ValueError):
        ingest.pack_rles([dict(length=49, counts=np.array([1, 3]))], 50, st, "x")                          # length != H*W
    # frames
    frames = [np.full((5, 7), k, np.uint16) for k in range(9)]
    dst = torch.empty(9 * 70, dtype=torch.uint8)
    assert ingest.host_lib().bff_host_pack_frames(frames, dst.data_ptr(), 70, 3) == 9
    assert np.array_equal(dst.numpy().view(np.uint16).reshape(9, 5, 7), np.stack(frames))
    assert ingest.host_lib().bff_host_pack_frames(frames + [np.zeros((5, 8), np.uint16)], dst.data_ptr(), 70, 2) == -1


def _write_png16(path, img, filters, interlace=0, bitdepth=16, n_idat=2, truncate=0):
    """A 16-bit grayscale PNG written by hand: the row filter of every row is chosen by the caller."""
    import struct
    import zlib
    h, w = img.shape
    bpp = 2 if bitdepth == 16 else 1
    be = img.astype(">u2").tobytes() if bitdepth == 16 else img.astype(np.uint8).tobytes()
    stride = w * bpp
    raw, prev = bytearray(), bytes(stride)
    for y in range(h):
        cur = be[y * stride:(y + 1) * stride]
        ft = filters[y]
        out = bytearray(stride)
        for i in range(stride):
            a = cur[i - bpp] if i >= bpp else 0
            b = prev[i]
            c = prev[i - bpp] if i >= bpp else 0
            if ft == 0:
                p = 0
            elif ft == 1:
                p = a
            elif ft == 2:
                p = b
            elif ft == 3:
                p = (a + b) >> 1
            else:
                pp = a + b - c
                pa, pb, pc = abs(pp - a), abs(pp - b), abs(pp - c)
                p = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
            out[i] = (cur[i] - p) & 255
        raw.append(ft)
        raw += out
        prev = cur
    chunk = lambda t, d: struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xFFFFFFFF)
    comp = zlib.compress(bytes(raw), 6)
    cuts = [len(comp) * k // n_idat for k in range(n_idat + 1)]
    data = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, bitdepth, 0, 0, 0, interlace)) + \
        chunk(b"tEXt", b"k\0v") + b"".join(chunk(b"IDAT", comp[a:b]) for a, b in zip(cuts[:-1], cuts[1:])) + chunk(b"IEND", b"")
    with open(path, "wb") as f:
        f.write(data[:len(data) - truncate] if truncate else data)


def test_native_png_decoder_equals_pil_on_all_filter_types(tmp_path):
    """io.decode_depth_pngs (libbff_host.so: chunk walk, inflate, the five PNG row filters at 2 bytes per pixel,
    byte swap; native threads) against PIL and against the source arrays: every filter type alone, mixed per row,
    several IDAT chunks, an ancillary chunk, a PIL-written file; into a caller-provided buffer (pinned staging)."""
    from PIL import Image
    from beyond_fixed_forms_amd import io as bio
    rng = np.random.default_rng(3)
    h, w = 37, 53
    paths, imgs = [], []
    for i in range(8):
        img = rng.integers(0, 65536, (h, w)).astype(np.uint16)
        if i == 5:
            img[:] = (np.arange(w)[None, :] * 1000 + np.arange(h)[:, None]) & 0xFFFF        # gradients: filters matter
        filters = [i % 5] * h if i < 5 else [int(v) for v in rng.integers(0, 5, h)]
        p = tmp_path / f"{i}.png"
        _write_png16(p, img, filters, n_idat=1 + i % 3)
        paths.append(str(p)); imgs.append(img)
    Image.fromarray(imgs[0]).save(tmp_path / "pil.png")
    paths.append(str(tmp_path / "pil.png")); imgs.append(imgs[0])
    got = bio.decode_depth_pngs(paths, n_threads=3)
    assert got.dtype == np.uint16 and got.shape == (len(paths), h, w)
    for g, e, p in zip(got, imgs, paths):
        assert np.array_equal(g, e) and np.array_equal(np.asarray(Image.open(p)), e), p
    buf = np.zeros(len(paths) * h * w + 7, dtype=np.uint16)
    again = bio.decode_depth_pngs(paths, out=buf, n_threads=1)
    assert np.shares_memory(again, buf) and np.array_equal(again, got)
    assert bio.decode_depth_pngs([]).shape[0] == 0


def test_native_png_decoder_declines_what_it_does_not_cover(tmp_path):
    """Files outside the decoder's scope fall to PIL, which then decides: an 8-bit image is an error for a depth
    frame, an interlaced 16-bit image is decoded by PIL (same values), a truncated file raises; a file of another size
    than the batch's is an error."""
    from PIL import Image
    from beyond_fixed_forms_amd import io as bio
    from beyond_fixed_forms_amd.ingest import host_lib
    import ctypes
    rng = np.random.default_rng(4)
    img = rng.integers(0, 65536, (20, 31)).astype(np.uint16)
    ok = tmp_path / "ok.png"
    _write_png16(ok, img, [4] * 20)
    g8 = tmp_path / "g8.png"
    Image.fromarray((img >> 8).astype(np.uint8)).save(g8)
    hw = (ctypes.c_int32 * 2)()
    assert host_lib().bff_host_png_size(str(g8).encode(), ctypes.cast(hw, ctypes.c_void_p)) == 3
    assert host_lib().bff_host_png_size(str(ok).encode(), ctypes.cast(hw, ctypes.c_void_p)) == 0 and (hw[0], hw[1]) == (20, 31)
    with pytest.raises(ValueError):
        bio.decode_depth_pngs([str(ok), str(g8)])
    other = tmp_path / "other.png"
    _write_png16(other, img[:10], [0] * 10)
    with pytest.raises(ValueError):
        bio.decode_depth_pngs([str(ok), str(other)])
    cut = tmp_path / "cut.png"
    _write_png16(cut, img, [1] * 20, truncate=40)
    with pytest.raises(Exception):
        bio.decode_depth_pngs([str(ok), str(cut)])
    status = np.zeros(2, np.int32)
    out = np.zeros((2, 20, 31), np.uint16)
    n = host_lib().bff_host_decode_depth_pngs([str(ok), str(cut)], out.ctypes.data, 20, 31, status.ctypes.data, 2)
    assert n == 1 and status[0] == 0 and status[1] != 0 and np.array_equal(out[0], img) and not out[1].any()
