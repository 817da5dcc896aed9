"""GPU parity, kernel by kernel: every C-ABI entry point against the oracle / golden vectors.

Bar: bit-exact for masks, counters, popcounts, adjacency and components; float32-exact for the
confidence mean; 1e-4 absolute for the cosine GEMM (tolerance stated by BASELINE.json north_star).
"""
import os

import numpy as np
import pytest
import torch

import golden_io as gio
from oracle import projection_ref as pref, rle_ref

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
Z = lambda name: np.load(os.path.join(gio.GOLDEN_DIR, name))


@pytest.fixture(scope="module")
def lib():
    from beyond_fixed_forms_amd import _lib
    _lib.load()
    assert torch.cuda.is_available()
    return _lib


def unpack(rows, n):
    """int64 bit rows (device) -> bool numpy (R,n), independent of bff_unpack_rows."""
    b = rows.cpu().numpy().view(np.uint8)
    return np.unpackbits(b, axis=-1, bitorder="little")[:, :n].astype(bool)


def pack_np(dense):
    n = dense.shape[1]
    nw = (n + 63) // 64
    pad = np.zeros((dense.shape[0], nw * 64), bool)
    pad[:, :n] = dense
    return torch.from_numpy(np.packbits(pad, axis=-1, bitorder="little").view(np.int64).copy()).to(DEV)


def maskbits_from_dense(lib, masks_list, hw):
    """list over views of dense (M,H*W) bool -> device maskbits via the RLE decode kernel."""
    from beyond_fixed_forms_amd.scene import runs_from_rles
    rles, voffs = [], [0]
    for m in masks_list:
        rles += rle_ref.rle_encode_batch_ref(torch.from_numpy(m))
        voffs.append(voffs[-1] + m.shape[0])
    wb = 32 if max(m.shape[0] for m in masks_list) <= 32 else 64
    rs, re, offs = runs_from_rles(rles)
    t = lambda a: torch.from_numpy(np.asarray(a, dtype=np.int32)).to(DEV)
    out = torch.empty((len(masks_list), hw), dtype=torch.int32 if wb == 32 else torch.int64, device=DEV)
    lib.rle_to_maskbits(t(rs), t(re), t(offs), t(voffs), len(masks_list), hw, wb, out)
    # with a segment bitmap the all-zero 128-pixel segments are skipped; what IS flagged must equal the full decode
    out2 = torch.full_like(out, -1)
    seg = torch.empty((len(masks_list), lib.segmap_words(hw)), dtype=torch.int32, device=DEV)
    lib.rle_to_maskbits(t(rs), t(re), t(offs), t(voffs), len(masks_list), hw, wb, out2, seg)
    bits = np.unpackbits(seg.cpu().numpy().view(np.uint8), axis=-1, bitorder="little")[:, :(hw + 127) // 128].astype(bool)
    full = out.cpu().numpy()
    nz = np.stack([np.add.reduceat((full[v] != 0).astype(np.int64), np.arange(0, hw, 128)) > 0 for v in range(len(masks_list))])
    assert np.array_equal(bits, nz)
    per_px = np.repeat(bits, 128, axis=1)[:, :hw]
    assert np.array_equal(out2.cpu().numpy()[per_px], full[per_px]) and (out2.cpu().numpy()[~per_px] == -1).all()
    maskbits_from_dense.last_segmap = seg
    maskbits_from_dense.last_sparse = out2
    # palette / word segments: a 128-pixel segment with at most 16 (32-bit words) / 8 (64-bit) PIECES -- maximal runs of
    # pixels with the same word -- is ONE 128-byte block of the label plane: 64 bytes of 4-bit piece numbers, then the
    # pieces' words in order; any other occupied segment is written as words into the word plane; nothing else is written
    ls = lib.label_plane_stride(hw)
    assert ls == lib.load().bff_label_plane_stride(hw) and ls % 128 == 0 and 0 <= ls - hw < 128
    lab = torch.full((len(masks_list), ls), 0xEE, dtype=torch.uint8, device=DEV)
    ovf = torch.full_like(out, -1)
    seg3 = torch.empty((len(masks_list), 2 * lib.segmap_words(hw)), dtype=torch.int32, device=DEV)
    lib.rle_to_labels(t(rs), t(re), t(offs), t(voffs), len(masks_list), hw, wb, lab, ovf, seg3)
    assert torch.equal(seg3[:, 0::2], seg)
    n_seg = (hw + 127) // 128
    pal_max, wdt = (16, np.uint32) if wb == 32 else (8, np.uint64)
    fmt = np.unpackbits(seg3[:, 1::2].contiguous().cpu().numpy().view(np.uint8), axis=-1, bitorder="little")[:, :n_seg].astype(bool)
    got_lab, got_ovf = lab.cpu().numpy(), ovf.cpu().numpy()
    fu = full.astype(np.uint64 if wb == 64 else np.uint32)
    for v in range(len(masks_list)):
        for sg in range(n_seg):
            real = fu[v, sg * 128:min(hw, sg * 128 + 128)]               # index bytes past the image edge are unspecified
            words = real
            starts = np.flatnonzero(np.concatenate([[True], words[1:] != words[:-1]]))
            piece_of = np.cumsum(np.concatenate([[0], (words[1:] != words[:-1]).astype(np.int64)]))
            block = got_lab[v, sg * 128:sg * 128 + 128]
            occupied = bool((real != 0).any())
            assert bool(bits[v, sg]) == occupied
            assert bool(fmt[v, sg]) == (occupied and starts.size > pal_max), (v, sg, starts.size)
            if not occupied or fmt[v, sg]:
                assert (block == 0xEE).all(), (v, sg)                    # no palette block for empty / word segments
                if fmt[v, sg]:
                    assert np.array_equal(got_ovf[v, sg * 128:sg * 128 + real.size].view(wdt), real)
                else:
                    assert (got_ovf[v, sg * 128:sg * 128 + real.size] == -1).all()
                continue
            assert (got_ovf[v, sg * 128:sg * 128 + real.size] == -1).all()
            idx = np.stack([block[:64] & 15, block[:64] >> 4], axis=1).reshape(-1)
            pal = block[64:].view(wdt)
            assert np.array_equal(idx[:real.size], piece_of), (v, sg)
            assert np.array_equal(pal[:starts.size], words[starts]), (v, sg)
            assert (block[64 + starts.size * pal.itemsize:] == 0xEE).all()
    assert (got_lab[:, n_seg * 128:] == 0xEE).all()
    maskbits_from_dense.last_labels_segmap = seg3
    maskbits_from_dense.last_labels = (lab, ovf)
    return out, wb


# ------------------------------------------------------------------ RLE decode (a1)
@pytest.mark.parametrize("m,hw", [(1, 100), (5, 5000), (32, 2048 * 16 + 77), (33, 4099), (64, 2048 * 3), (7, 2048 * 17)])
def test_rle_to_maskbits(lib, m, hw):
    rng = np.random.default_rng(m * 1000 + hw)
    dense = np.zeros((m, hw), bool)
    for k in range(m):
        x = rng.random(hw) < rng.choice([0.0, 0.02, 0.5, 1.0], p=[0.1, 0.4, 0.4, 0.1])
        # long runs, some crossing the 2048-pixel chunk and the 16-chunk band boundaries
        for _ in range(4):
            a = int(rng.integers(0, hw)); b = min(hw, a + int(rng.integers(1, 5000)))
            x[a:b] = True
        dense[k] = x if rng.random() < 0.9 else False
    dense[0, 0] = dense[0, -1] = True
    bits, wb = maskbits_from_dense(lib, [dense], hw)
    got = bits.cpu().numpy()[0].astype(np.uint64 if wb == 64 else np.uint32)
    exp = np.zeros(hw, dtype=np.uint64)
    for k in range(m):
        exp |= dense[k].astype(np.uint64) << np.uint64(k)
    assert np.array_equal(got.astype(np.uint64), exp)
    # the oracle decoder agrees with the dense masks we encoded
    dec = rle_ref.rle_decode_batch_ref(rle_ref.rle_encode_batch_ref(torch.from_numpy(dense))).numpy().astype(bool)
    assert np.array_equal(dec, dense)


def test_rle_unsorted_overlapping_runs_are_normalised(lib):
    from beyond_fixed_forms_amd.scene import runs_from_rles
    rle = dict(length=300, counts=np.array([50, 20, 10, 30, 60, 5, 290, 50]))   # unsorted, overlapping, clipped
    rs, re, offs = runs_from_rles([rle])
    exp = rle_ref.rle_decode_ref(rle).astype(bool)
    got = np.zeros(300, bool)
    for a, b in zip(rs, re):
        got[a:b] = True
    assert np.array_equal(got, exp) and np.all(rs[1:] >= re[:-1])
    with pytest.raises(ValueError):
        runs_from_rles([dict(length=10, counts=np.array([0, 3]))])


# ------------------------------------------------------------------ projection sweep (a2-a7, a15)
def run_view(lib, xyz, inv_pose, k33, depth, masks):
    n = xyz.shape[0]
    nw = (n + 63) // 64
    n_pad = ((n + 1023) // 1024) * 1024
    soa = np.zeros((3, n_pad)); soa[:, :n] = xyz.T
    h, w = depth.shape
    m = masks.shape[0]
    bits, wb = maskbits_from_dense(lib, [masks.reshape(m, -1).astype(bool)], h * w)
    i32 = lambda a: torch.tensor(a, dtype=torch.int32, device=DEV)
    rows = torch.zeros((m, nw), dtype=torch.int64, device=DEV)       # the sweep stores only sectors with a point
    mc = torch.zeros(n, dtype=torch.int32, device=DEV)
    vc = torch.zeros(n, dtype=torch.int32, device=DEV)
    args = (torch.from_numpy(soa).to(DEV), n, torch.from_numpy(inv_pose.reshape(1, 16).copy()).to(DEV), k33,
            torch.from_numpy(depth.reshape(1, -1).copy()).to(DEV), i32([0]), h, w, 0.08)
    cm = lib.chunk_mask_buffer(m, nw, DEV).zero_()
    lib.project_views(*args, bits, wb, i32([0]), i32([0]), i32([m]), i32([1]), rows, mc, vc, chunk_mask=cm)
    # the chunk flags the sweep leaves are exactly the occupancy masks a full pass over the rows computes, and
    # the statistics read through them equal the dense ones
    dense_stats = lib.row_stats(rows)
    assert torch.equal(cm, dense_stats[2])
    for a, b in zip(lib.row_stats(rows, cm), dense_stats):
        assert torch.equal(a, b)
    # same sweep over the sparsely written image + segment bitmap
    rows2, mc2, vc2 = torch.zeros_like(rows), torch.zeros_like(mc), torch.zeros_like(vc)
    lib.project_views(*args, maskbits_from_dense.last_sparse, wb, i32([0]), i32([0]), i32([m]), i32([1]), rows2, mc2, vc2,
                      segmap=maskbits_from_dense.last_segmap)
    assert torch.equal(rows2, rows) and torch.equal(mc2, mc) and torch.equal(vc2, vc)
    # same sweep through the label / word segments
    lab, ovf = maskbits_from_dense.last_labels
    rows4, mc4, vc4 = torch.zeros_like(rows), torch.zeros_like(mc), torch.zeros_like(vc)
    lib.project_views(*args, ovf, wb, i32([0]), i32([0]), i32([m]), i32([1]), rows4, mc4, vc4,
                      segmap=maskbits_from_dense.last_labels_segmap, labels=lab)
    assert torch.equal(rows4, rows) and torch.equal(mc4, mc) and torch.equal(vc4, vc)
    # and with the frustum-culling table of the point tiles (boxes may hold inf / 1e300 / denormals here)
    rows3, mc3, vc3 = torch.zeros_like(rows), torch.zeros_like(mc), torch.zeros_like(vc)
    lib.project_views(*args, bits, wb, i32([0]), i32([0]), i32([m]), i32([1]), rows3, mc3, vc3,
                      tile_bounds=lib.point_tile_bounds(args[0], n))
    assert torch.equal(rows3, rows) and torch.equal(mc3, mc) and torch.equal(vc3, vc)
    return unpack(rows, n), mc.cpu().numpy(), vc.cpu().numpy()


@pytest.mark.parametrize("case", [str(c) for c in Z("proj_helpers.npz")["cases"]])
def test_project_views_golden(lib, case):
    """Against the reference's own outputs, incl. exact half pixels, +-ulp around the depth
    threshold, z <= 0, z == 0 (NaN/inf -> INT64_MIN), out-of-bounds points and denormals."""
    z = Z("proj_helpers.npz")
    g = lambda k: z[f"{case}.{k}"]
    n = g("xyz").shape[0]
    masked, mc, vc = run_view(lib, g("xyz"), g("inv_pose"), g("K"), g("depth"), g("masks"))
    exp = gio.unpack_bool_rows(g("masked"), n)
    assert np.array_equal(vc.astype(bool), g("vis"))
    assert np.array_equal(masked, exp)
    assert np.array_equal(mc, exp.sum(0))


def test_project_views_random_vs_c_oracle(lib):
    """200k points, 6 frames, ScanNet-like intrinsics: pixel-exact against the fma-chain C oracle."""
    from oracle import geom_fma
    rng = np.random.default_rng(3)
    n, h, w, m = 200_000, 240, 320, 9
    xyz = rng.uniform(-4, 4, (n, 3))
    k33 = np.array([[288.0123, 0.0, 159.5], [0.0, 288.0123, 119.5], [0.0, 0.0, 1.0]])
    for f in range(6):
        a = rng.uniform(-3, 3)
        pose = np.eye(4)
        pose[:3, :3] = [[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]]
        pose[:3, 3] = rng.uniform(-1, 1, 3)
        inv = np.linalg.inv(pose)
        pts, pix, _ = geom_fma.view(xyz, inv, k33, np.zeros((h, w), np.float32))
        depth = rng.uniform(0.5, 6, (h, w)).astype(np.float32)
        ok = (pix[:, 0] >= 0) & (pix[:, 0] < w) & (pix[:, 1] >= 0) & (pix[:, 1] < h)
        sel = ok & (rng.random(n) < 0.7)
        depth[pix[sel, 1], pix[sel, 0]] = (pts[sel, 2] + rng.uniform(-0.1, 0.1, sel.sum())).astype(np.float32)
        masks = rng.random((m, h, w)) < 0.3
        _, _, vis = geom_fma.view(xyz, inv, k33, depth)
        exp = pref.masked_points(pix, vis, masks.astype(np.float32))
        masked, mc, vc = run_view(lib, xyz, inv, k33, depth, masks)
        assert np.array_equal(vc.astype(bool), vis)
        assert np.array_equal(masked, exp)
        assert np.array_equal(mc, exp.sum(0))
        assert vis.sum() > 1000


def test_clear_flagged_chunks(lib):
    rng = np.random.default_rng(11)
    n = 5 * 512 + 77                                            # ragged last chunk
    d = random_rows(rng, 9, n, 0.01)
    d[:, 1024:1536] = False                                     # a chunk nobody occupies
    rows = pack_np(d)
    cm = lib.row_stats(rows)[2]
    untouched = rows.clone()
    lib.call("bff_clear_flagged_chunks", lib._ptr(rows, torch.int64), rows.shape[0], rows.shape[1], lib._ptr(cm, torch.int64))
    assert int(rows.count_nonzero()) == 0 and int(untouched.count_nonzero()) > 0


# ------------------------------------------------------------------ bit-row primitives
def random_rows(rng, r, n, p=0.1):
    d = rng.random((r, n)) < p
    return d


@pytest.mark.parametrize("r,n", [(1, 1), (3, 64), (70, 1000), (130, 20_000)])
def test_pack_unpack_popcount(lib, r, n):
    rng = np.random.default_rng(r * n)
    d = random_rows(rng, r, n, 0.3)
    rows = lib.pack_rows(torch.from_numpy(d).to(DEV))
    assert np.array_equal(unpack(rows, n), d)
    assert torch.equal(rows, pack_np(d))                       # padding bits are zero
    assert np.array_equal(lib.unpack_rows(rows, n).cpu().numpy(), d)
    assert np.array_equal(lib.popcount_rows(rows).cpu().numpy(), d.sum(1))
    idx = torch.tensor(rng.integers(0, r, 5), dtype=torch.int32, device=DEV)
    assert np.array_equal(lib.popcount_rows(rows, idx).cpu().numpy(), d[idx.cpu().numpy()].sum(1))
    assert np.array_equal(unpack(lib.gather_rows(rows, idx), n), d[idx.cpu().numpy()])


@pytest.mark.parametrize("na,nb,n", [(1, 1, 10), (5, 70, 3000), (129, 65, 4097), (64, 64, 64 * 32)])
def test_cross_popcount(lib, na, nb, n):
    rng = np.random.default_rng(na + nb + n)
    a, b = random_rows(rng, na, n, 0.2), random_rows(rng, nb, n, 0.3)
    exp = a.astype(np.int32) @ b.astype(np.int32).T
    assert np.array_equal(lib.cross_popcount(pack_np(a), pack_np(b)).cpu().numpy(), exp)
    ia = torch.tensor(rng.integers(0, na, 7), dtype=torch.int32, device=DEV)
    got = lib.cross_popcount(pack_np(a), pack_np(a), ia, ia).cpu().numpy()
    sel = a[ia.cpu().numpy()].astype(np.int32)
    assert np.array_equal(got, sel @ sel.T)


@pytest.mark.parametrize("r,n", [(12, 300), (200, 5000), (130, 64)])
def test_merge_adjacency_and_components(lib, r, n):
    """Adjacency == same_label & (iou > f32(0.2)) of the oracle (P:100-146) incl. empty rows (NaN),
    components == the oracle's closure components."""
    rng = np.random.default_rng(r + n)
    centres = rng.integers(0, n, 8)
    d = np.zeros((r, n), bool)
    for i in range(r):
        c = centres[rng.integers(0, 8)]
        wdt = int(rng.integers(1, max(2, n // 10)))
        d[i, max(0, c - wdt): c + wdt] = True
        d[i] &= rng.random(n) < 0.8
    d[rng.integers(0, r, 3)] = False                              # empty masks -> NaN IoU
    labels = [("a", "b")[int(x)] for x in rng.random(r) < 0.3]
    t = torch.from_numpy(d)
    iou = pref.pairwise_iou(t)
    merge = pref.label_equality(labels) & (iou > 0.2)
    rows = pack_np(d)
    ids = {}
    lid = torch.tensor([ids.setdefault(s, len(ids)) for s in labels], dtype=torch.int32, device=DEV)
    area = lib.popcount_rows(rows)
    adj, inter = lib.merge_adjacency(rows, area, lid, 0.2, want_inter=True)
    assert np.array_equal(inter.cpu().numpy(), d.astype(np.int32) @ d.astype(np.int32).T)
    assert np.array_equal(unpack(adj, r), merge.numpy())
    from beyond_fixed_forms_amd.projection import groups_from_labels
    lab = lib.components(adj).cpu().numpy()
    got = groups_from_labels(lab, np.diag(merge.numpy()))
    assert got == pref.connected_groups(merge.float())


@pytest.mark.parametrize("thr", [0.2, 0.0, -1.0])
def test_merge_adjacency_block_sparse_equals_dense(lib, thr):
    """Row order + chunk skipping change nothing: adjacency (indexed by position in `order`) and the
    Gram equal the dense all-words result, incl. thr < 0 where empty intersections DO count."""
    rng = np.random.default_rng(17)
    r, n = 300, 40_000                     # 625 words = 79 chunks of 8 words
    d = np.zeros((r, n), bool)
    for i in range(r):
        c = int(rng.integers(0, 6)) * 6000 + int(rng.integers(0, 500))
        d[i, c: c + int(rng.integers(50, 3000))] = True
        d[i] &= rng.random(n) < 0.7
    d[5] = False; d[77] = False
    labels = rng.integers(0, 2, r)
    rows = pack_np(d)
    lid = torch.tensor(labels, dtype=torch.int32, device=DEV)
    area, mean_word, cmask, hist, sig = lib.row_stats(rows)
    bw = -(-((n + 63) // 64) // 64)
    exp_hist = np.add.reduceat(np.pad(d, ((0, 0), (0, 64 * 64 * bw - n))), np.arange(0, 64 * 64 * bw, 64 * bw), axis=1)
    assert np.array_equal(hist.cpu().numpy(), exp_hist)
    assert np.array_equal(area.cpu().numpy(), d.sum(1))
    occ = np.add.reduceat(np.pad(d, ((0, 0), (0, (-n) % 512))), np.arange(0, n, 512), axis=1) > 0
    assert np.array_equal(unpack(cmask, occ.shape[1]), occ)
    assert mean_word[5].item() == 0x7fffffff
    order = torch.argsort((lid.long() << 32) | mean_word.long()).to(torch.int32)
    dense_adj, dense_inter = lib.merge_adjacency(rows, area, lid, thr, want_inter=True)
    sp_adj, sp_inter = lib.merge_adjacency(rows, area, lid, thr, order=order, chunk_mask=cmask, want_inter=True)
    o = order.cpu().numpy()
    exp_inter = d.astype(np.int32) @ d.astype(np.int32).T
    assert np.array_equal(dense_inter.cpu().numpy(), exp_inter)
    assert np.array_equal(sp_inter.cpu().numpy(), exp_inter)
    iou = pref.pairwise_iou(torch.from_numpy(d))
    merge = ((torch.from_numpy(labels)[:, None] == torch.from_numpy(labels)[None, :]) & (iou > thr)).numpy()
    assert np.array_equal(unpack(dense_adj, r), merge)
    assert np.array_equal(unpack(sp_adj, r), merge[o][:, o])
    # with the histogram bound (production path: no Gram output) the adjacency is still identical
    hb_adj = lib.merge_adjacency(rows, area, lid, thr, order=order, chunk_mask=cmask, hist=hist)
    assert torch.equal(hb_adj, sp_adj)
    # production path: components by on-device union-find, no adjacency matrix
    from beyond_fixed_forms_amd.projection import groups_from_labels
    exp_groups = pref.connected_groups(torch.from_numpy(merge).float())
    self_loop = np.diag(merge)
    for ordr in (order, torch.argsort(sig, stable=True).to(torch.int32),
                 torch.from_numpy(rng.permutation(r).astype(np.int32)).to(DEV)):
        comp = lib.merge_components(rows, area, lid, thr, ordr, cmask, hist).cpu().numpy()
        assert groups_from_labels(comp, self_loop) == exp_groups


def test_permute_bits(lib):
    rng = np.random.default_rng(4)
    n = 10_007
    d = random_rows(rng, 5, n, 0.3)
    perm = rng.permutation(n)                       # sorted position s holds original point perm[s]
    unsort = np.empty(n, np.int32); unsort[perm] = np.arange(n, dtype=np.int32)
    sorted_rows = pack_np(d[:, perm])
    back = lib.permute_bits(sorted_rows, torch.from_numpy(unsort).to(DEV), n)
    assert np.array_equal(unpack(back, n), d) and torch.equal(back, pack_np(d))


def test_morton_order_is_a_permutation():
    from beyond_fixed_forms_amd.scene import morton_order
    rng = np.random.default_rng(0)
    p = rng.uniform(-3, 3, (5000, 3)); p[7] = np.nan
    o = morton_order(p)
    assert sorted(o.tolist()) == list(range(5000))
    q = p[o]
    # neighbours along the curve are close in space (median step far below the box size)
    assert np.nanmedian(np.linalg.norm(np.diff(q, axis=0), axis=1)) < 0.6


def test_components_long_chain(lib):
    """A path graph of 3000 nodes (worst case for label propagation) + isolated nodes."""
    n = 3000
    adj = np.zeros((n, n), bool)
    perm = np.random.default_rng(0).permutation(n)
    for a, b in zip(perm[:-1], perm[1:]):
        if a % 50 and b % 50:
            adj[a, b] = adj[b, a] = True
    adj[np.arange(n), np.arange(n)] = True
    lab = lib.components(pack_np(adj)).cpu().numpy()
    from beyond_fixed_forms_amd.projection import groups_from_labels
    assert groups_from_labels(lab, np.ones(n, bool)) == pref.connected_groups(torch.from_numpy(adj).float())


@pytest.mark.parametrize("dtype", [torch.float16, torch.float32])
def test_or_reduce_and_conf_mean(lib, dtype):
    rng = np.random.default_rng(5)
    r, n = 40, 777
    d = random_rows(rng, r, n, 0.05)
    r, n = 1400, 777
    d = random_rows(rng, r, n, 0.002)
    groups = [[0, 5, 7], [1], [2, 3, 4, 6, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23], [39, 38],
              list(range(40, 1400))[::-1]]          # > 1024 members: several LDS stages, several z slices
    offs = torch.tensor(np.cumsum([0] + [len(g) for g in groups]), dtype=torch.int32, device=DEV)
    mem = torch.tensor([i for g in groups for i in g], dtype=torch.int32, device=DEV)
    out = lib.or_reduce_groups(pack_np(d), offs, mem, max(len(g) for g in groups))
    assert np.array_equal(unpack(out, n), np.stack([d[g].any(0) for g in groups]))
    conf = torch.from_numpy(rng.uniform(0.2, 0.5, r)).to(dtype)
    got = lib.group_conf_mean(conf.to(DEV), offs, mem).cpu()
    exp = torch.tensor([sum([conf[i] for i in g]) / len(g) for g in groups])     # P:225, sequential in dtype
    assert got.dtype == dtype and torch.equal(got, exp)
    # both in one launch (the means on extra blocks beside the OR)
    out2, got2 = lib.or_reduce_groups(pack_np(d), offs, mem, max(len(g) for g in groups), conf.to(DEV))
    assert torch.equal(out2, out) and torch.equal(got2.cpu(), exp)


def test_or_reduce_long_rows_through_chunk_flags(lib):
    """Rows of >= 8192 words (config 4 is 15 625) are OR-ed through their chunk flags: same result as the dense pass,
    members' words outside their flagged chunks are never read (poisoned here)."""
    rng = np.random.default_rng(11)
    r, nw = 90, 8192 + 77
    n = nw * 64 - 13
    rows = torch.zeros((r, nw), dtype=torch.int64, device=DEV)
    dense = np.zeros((r, nw), np.uint64)
    for i in range(r):                                   # a few occupied 8-word chunks per row, some shared
        for c in rng.choice(nw // 8, size=int(rng.integers(0, 12)), replace=False):
            dense[i, 8 * c:8 * c + 8] = rng.integers(0, 2 ** 63, 8, dtype=np.uint64) * (rng.random(8) < 0.6)
    dense[3, -5:] = np.uint64(7)                          # the ragged last chunk
    dense[:, -1] &= np.uint64((1 << (64 - 13)) - 1)
    rows.copy_(torch.from_numpy(dense.view(np.int64)))
    stats = lib.row_stats(rows)
    cm = stats[2]
    # poison every unflagged chunk: the flagged pass must not see it
    flags = np.unpackbits(cm.cpu().numpy().view(np.uint8), axis=-1, bitorder="little")[:, :(nw + 7) // 8].astype(bool)
    poisoned = dense.copy()
    per_word = np.repeat(flags, 8, axis=1)[:, :nw]
    poisoned[~per_word] = np.uint64(0xDEADBEEF)
    rows_p = torch.from_numpy(poisoned.view(np.int64)).to(DEV)
    groups = [[0, 1, 2], [3], list(range(4, 80)), [89, 85]]
    offs = torch.tensor(np.cumsum([0] + [len(g) for g in groups]), dtype=torch.int32, device=DEV)
    mem = torch.tensor([i for g in groups for i in g], dtype=torch.int32, device=DEV)
    exp = np.stack([np.bitwise_or.reduce(dense[g], axis=0) for g in groups])
    got = lib.or_reduce_groups(rows_p, offs, mem, max(len(g) for g in groups), chunk_mask=cm)
    assert np.array_equal(got.cpu().numpy().view(np.uint64), exp)
    assert np.array_equal(lib.or_reduce_groups(rows, offs, mem, max(len(g) for g in groups)).cpu().numpy().view(np.uint64), exp)


def test_row_ops_and_rows(lib):
    rng = np.random.default_rng(8)
    r, n = 6, 500
    d = random_rows(rng, r, n, 0.4)
    ops = [(0, 1, 0), (0, 2, 1), (1, 3, 2), (2, 4, 3), (0, 0, 5), (1, 5, 0)]
    exp = d.copy()
    for op, dst, src in ops:
        exp[dst] = (exp[dst] & ~exp[src]) if op == 0 else (exp[dst] | exp[src]) if op == 1 else exp[src]
    rows = pack_np(d)
    lib.apply_row_ops(rows, torch.tensor(ops, dtype=torch.int32, device=DEV))
    assert np.array_equal(unpack(rows, n), exp)
    keep = rng.random(n) < 0.5
    lib.and_rows(rows, pack_np(keep[None])[0])
    assert np.array_equal(unpack(rows, n), exp & keep)


def test_resolve_overlaps_golden(lib):
    """solve_overlapping (P:277-301) decided and applied on the device == the reference's results."""
    z = Z("agg_helpers.npz")
    for case in ("ovl_equal", "ovl_mixed", "ovl_none"):
        n = int(z[f"{case}.n"])
        rows = pack_np(gio.unpack_bool_rows(z[f"{case}.ins"], n))
        lib.resolve_overlaps(rows, torch.tensor(z[f"{case}.sizes"], dtype=torch.int32, device=DEV))
        assert np.array_equal(unpack(rows, n), gio.unpack_bool_rows(z[f"{case}.resolved"], n)), case
    rng = np.random.default_rng(9)
    for k, p in ((23, 0.15), (110, 0.01), (140, 0.2)):         # segments of 2, 7, 9 rows per wave
        d = rng.random((k, 3000)) < p
        sizes = rng.integers(2, 6, k)
        exp = pref.resolve_overlaps(torch.from_numpy(d.copy()), [list(range(s)) for s in sizes]).numpy()
        rows = pack_np(d)
        lib.resolve_overlaps(rows, torch.tensor(sizes, dtype=torch.int32, device=DEV))
        assert np.array_equal(unpack(rows, 3000), exp)


@pytest.mark.parametrize("k,n,p", [(1, 100, 0.1), (2, 64, 0.5), (17, 5000, 0.08), (33, 777, 0.6), (64, 777, 0.08), (129, 3000, 0.3),
                                   (200, 4100, 0.08), (512, 500, 0.05), (513, 900, 0.08), (1500, 700, 0.02), (4200, 300, 0.01)])
def test_resolve_overlaps_closed_form_equals_the_ordered_replay(lib, k, n, p):
    """One pass (every point stays in the row with the largest size, ties: the largest index -- a prefix OR in that order)
    == the reference's loop spelled out: popcount, the ordered pair list of P:289-292 replayed pair by pair (P:295-299),
    and_rows, popcount.  Segment lengths 1 .. 32 rows per wave, the two-pass form (k > 512) and the replay fallback
    (k > bff_resolve_overlaps_max_rows()); many size ties."""
    rng = np.random.default_rng(k)
    d = random_rows(rng, k, n, p)
    keep = pack_np(rng.random((1, n)) < 0.7)[0]
    sizes = torch.from_numpy(rng.integers(1, 6, k).astype(np.int32)).to(DEV)
    ref = pack_np(d)
    before_ref = lib.popcount_rows(ref)
    lib.resolve_overlaps_replay(ref, sizes)
    lib.and_rows(ref, keep)
    after_ref = lib.popcount_rows(ref)
    rows = pack_np(d)
    before, after = lib.resolve_overlaps_filtered(rows, sizes, keep)
    assert torch.equal(rows, ref) and torch.equal(before, before_ref) and torch.equal(after, after_ref)
    if k <= 300:        # and without a filter, against the oracle's loop
        exp = pref.resolve_overlaps(torch.from_numpy(d.copy()), [list(range(s)) for s in sizes.tolist()]).numpy()
        rows = pack_np(d)
        lib.resolve_overlaps(rows, sizes)
        assert np.array_equal(unpack(rows, n), exp)


@pytest.mark.parametrize("n", [1, 63, 64, 65, 128, 10_001, 16_384 + 64, 200_000])
def test_rows_to_rle_matches_reference_encoder(lib, n):
    """Device RLE encoder == rle_encode_batch (RLE:10-32), incl. runs touching either end, word-aligned
    lengths, empty and full rows; decode(encode(x)) == x through the device decoder too."""
    from beyond_fixed_forms_amd.scene import runs_from_rles
    rng = np.random.default_rng(n)
    d = np.stack([rng.random(n) < p for p in (0.0, 1.0, 0.5, 0.02, 0.98)] +
                 [np.repeat(rng.random(n // 37 + 1) < 0.5, 37)[:n], np.arange(n) % 2 == 0, np.arange(n) % 2 == 1])
    d[3, 0] = d[3, -1] = True
    rows = pack_np(d)
    got = lib.rows_to_rle(rows, n)
    exp = rle_ref.rle_encode_batch_ref(torch.from_numpy(d))
    for g, e in zip(got, exp):
        assert g["length"] == e["length"] == n and np.array_equal(g["counts"], e["counts"])
    rs, re, offs = runs_from_rles(got)
    t = lambda a: torch.from_numpy(a).to(DEV)
    assert torch.equal(lib.rle_to_rows(t(rs), t(re), t(offs), n), rows)


def test_rle_to_rows_golden(lib):
    from beyond_fixed_forms_amd.scene import runs_from_rles
    z = Z("refine_helpers.npz")
    rles = gio.unpack_rles(z["rle1d.len"], z["rle1d.counts"], z["rle1d.offs"])
    t = lambda a: torch.from_numpy(a).to(DEV)
    for i, r in enumerate(rles):
        rs, re, offs = runs_from_rles([r])
        rows = lib.rle_to_rows(t(rs), t(re), t(offs), int(r["length"]))
        assert np.array_equal(unpack(rows, int(r["length"]))[0], z[f"rle1d.dec{i}"].astype(bool)), i
    rng = np.random.default_rng(1)
    d = random_rows(rng, 9, 10_001, 0.3)
    d[3] = False; d[4] = True
    enc = rle_ref.rle_encode_batch_ref(torch.from_numpy(d))
    rs, re, offs = runs_from_rles(enc)
    assert np.array_equal(unpack(lib.rle_to_rows(t(rs), t(re), t(offs), 10_001), 10_001), d)


@pytest.mark.parametrize("mode", ["ratio", "occurrence"])
@pytest.mark.parametrize("n", [1, 1000, 50_000, 200_001])
def test_point_filters(lib, mode, n):
    """Threshold = unique()[floor(t * n_unique)] exactly as torch computes it (P:513-518, 571-576), selected
    and applied on the device."""
    import math
    rng = np.random.default_rng(n)
    masked = rng.integers(0, 700, n) * (rng.random(n) < 0.4)
    viewed = rng.integers(0, 300, n)
    mt, vt = torch.tensor(masked, dtype=torch.float32), torch.tensor(viewed, dtype=torch.float32)
    md = torch.tensor(masked, dtype=torch.int32, device=DEV)
    vd = torch.tensor(viewed, dtype=torch.int32, device=DEV) if mode == "ratio" else None
    frac = 0.38 if mode == "ratio" else 0.3
    stat = mt / (vt + 1) if mode == "ratio" else mt
    uniq = stat.unique()
    thr = uniq[math.floor(frac * uniq.shape[0])]
    mt[stat < thr] = 0
    thr_dev, n_unique = lib.point_threshold(md, vd, frac)
    assert int(n_unique.item()) == uniq.shape[0]
    assert thr_dev.cpu().numpy()[0] == thr.numpy()
    keep = lib.ratio_keep(md, vd, thr_dev, True)
    assert np.array_equal(unpack(keep[None], n)[0], (mt > 0).numpy())
    assert torch.equal(lib.ratio_keep(md, vd, float(thr), True), keep)
    keep0 = lib.ratio_keep(md, None, 0.0, False)
    assert np.array_equal(unpack(keep0[None], n)[0], masked > 0)
    fits = True                              # 64 partitions x 6144 values: every case here is far inside the capacity
    for f in (0.0, 0.999999, 1.0):          # first, last, out of range (python: IndexError)
        t2, _ = lib.point_threshold(md, vd, f)
        t3, nu3, ovf = lib.point_threshold_pairs(md, vd, f)
        k = math.floor(f * uniq.shape[0])
        assert int(ovf.item()) == (0 if fits else 1) and (not fits or int(nu3.item()) == uniq.shape[0])
        if k < uniq.shape[0]:
            assert t2.cpu().numpy()[0] == uniq[k].numpy() and (not fits or t3.cpu().numpy()[0] == uniq[k].numpy())
        else:
            assert np.isnan(t2.cpu().numpy()[0]) and np.isnan(t3.cpu().numpy()[0])
    # the set formulation (no sort): same threshold, same count, bit for bit
    t3, nu3, ovf = lib.point_threshold_pairs(md, vd, frac)
    assert int(ovf.item()) == (0 if fits else 1)
    assert not fits or (int(nu3.item()) == uniq.shape[0] and t3.cpu().numpy()[0] == thr.numpy())



def test_point_threshold_pairs_out_of_range_and_overflow(lib):
    """Large counts (thousands of masks / frames per point) and many distinct values; more distinct values than the
    merging set holds (bff_point_threshold_capacity()) set the overflow flag (the caller then sorts)."""
    import math
    rng = np.random.default_rng(77)
    n = 50_000
    masked = rng.integers(0, 9000, n)
    viewed = rng.integers(0, 3000, n)
    cap = int(lib.load().bff_point_threshold_capacity())          # per hash partition (64 of them)
    md = torch.tensor(masked, dtype=torch.int32, device=DEV)
    vd = torch.tensor(viewed, dtype=torch.int32, device=DEV)
    stat = torch.tensor(masked, dtype=torch.float32) / (torch.tensor(viewed, dtype=torch.float32) + 1)
    uniq = stat.unique()
    for f in (0.0, 0.38, 0.77):
        thr, nu, ovf = lib.point_threshold_pairs(md, vd, f)
        assert int(ovf.item()) == 0 and int(nu.item()) == uniq.shape[0]
        assert thr.cpu().numpy()[0] == uniq[math.floor(f * uniq.shape[0])].numpy()
    assert uniq.shape[0] > 30_000
    big = torch.arange(1_000_000, dtype=torch.int32, device=DEV)               # 10^6 distinct values: ~15 600 per partition
    _, _, ovf = lib.point_threshold_pairs(big, None, 0.3)
    assert int(ovf.item()) == 1
    ok = torch.arange(200_000, dtype=torch.int32, device=DEV) + 5000           # 200 k distinct values (~3 100 per partition) fit
    thr, nu, ovf = lib.point_threshold_pairs(ok, None, 0.3)
    assert int(ovf.item()) == 0 and int(nu.item()) == 200_000 and thr.cpu().numpy()[0] == np.float32(5000 + 60_000)
    assert lib.load().bff_point_threshold_capacity_set(2) == 2                 # test hook: tiny partitions overflow
    try:
        _, _, ovf = lib.point_threshold_pairs(ok, None, 0.3)
        assert int(ovf.item()) == 1
    finally:
        assert lib.load().bff_point_threshold_capacity_set(0) == cap


@pytest.mark.parametrize("hs,ws,h,w", [(480, 640, 968, 1296), (48, 64, 97, 131), (120, 160, 120, 160), (100, 90, 37, 41)])
def test_depth_ingestion_matches_host_restatement(lib, hs, ws, h, w):
    """uint16 millimetres -> float32 metres at (h, w) on the device == io.resize_bilinear_f32(png / 1000), bit for
    bit (up- and down-scaling, identity).  Parity against cv2 itself is unpinned (no cv2 offline)."""
    from beyond_fixed_forms_amd import io
    rng = np.random.default_rng(hs * w)
    raw = rng.integers(0, 6000, (3, hs, ws)).astype(np.uint16)
    raw[0, :5] = 0
    exp = np.stack([io.resize_bilinear_f32(r.astype(np.float32) / np.float32(1000), w, h) for r in raw])
    taps = None if (hs, ws) == (h, w) else tuple(torch.from_numpy(a).to(DEV) for a in io.bilinear_taps(hs, ws, h, w))
    got = lib.depth_from_u16(torch.from_numpy(raw.view(np.int16)).to(DEV), h, w, taps).cpu().numpy().reshape(3, h, w)
    assert np.array_equal(got.view(np.uint32), exp.view(np.uint32))


# banks wider than 64 columns take the row-tile kernel (k split over 4 waves, 13 column tiles per pass): cover fewer
# k-steps than waves (dim 64, 96), a ragged last column tile, a second pass (nb > 208) and ragged rows
@pytest.mark.parametrize("na,nb,dim", [(1, 198, 768), (37, 200, 64), (300, 198, 512), (16, 16, 32), (50, 65, 96),
                                       (33, 250, 160), (100, 431, 768),
                                       # >= 2048 rows against a wide bank: 64-row blocks, the bank's k-slices through LDS
                                       (2100, 198, 768), (2048, 250, 96), (2500, 70, 32)])
def test_cosine_gemm(lib, na, nb, dim):
    """Config 5: CLIP-sized embeddings against a 200-label bank; |cos - f64 reference| <= 1e-4."""
    g = torch.Generator().manual_seed(na * dim)
    a = torch.randn(na, dim, generator=g).half()
    b = torch.randn(nb, dim, generator=g).half()
    got = lib.cosine_gemm_f16(a.to(DEV), b.to(DEV)).cpu().double()
    ad, bd = a.double(), b.double()
    exp = (ad @ bd.T) / (ad.norm(dim=1, keepdim=True) * bd.norm(dim=1, keepdim=True).T)
    assert (got - exp).abs().max().item() <= 1e-4
    # asymmetric exact-integer check of the MFMA fragment maps (would catch a transposed C write)
    ai = torch.arange(na * dim).reshape(na, dim).remainder(7).half()
    bi = torch.arange(nb * dim).reshape(nb, dim).remainder(5).add(1).half()
    got = lib.cosine_gemm_f16(ai.to(DEV), bi.to(DEV)).cpu().double()
    exp = (ai.double() @ bi.double().T) / (ai.double().norm(dim=1, keepdim=True) * bi.double().norm(dim=1, keepdim=True).T)
    ok = torch.isfinite(exp)
    assert (got[ok] - exp[ok]).abs().max().item() <= 1e-4


def test_evaluation_overlaps_match_reference_formulas(lib):
    """pred <-> GT overlap counting of assign_instances_for_scan (scannetv2_inst_eval.py:318-349): the numbers the
    reference computes pair by pair with count_nonzero(logical_and(gts == id, pred_mask))."""
    from beyond_fixed_forms_amd.evaluation import pred_gt_overlaps
    rng = np.random.default_rng(12)
    n, p_, g_ = 50_001, 17, 23
    sem = rng.integers(0, 40, n)
    ins = rng.integers(0, g_, n)
    gts = sem * 1000 + ins + 1
    gts[rng.random(n) < 0.1] = 0
    ids = np.unique(gts)[1:][:g_]
    pred = rng.random((p_, n)) < 0.05
    pred[3] = False
    bool_void = rng.random(n) < 0.2
    inter, pc, gc, vi = pred_gt_overlaps(pack_np(pred), torch.from_numpy(gts), ids, bool_void)
    for a in range(p_):
        assert pc[a] == np.count_nonzero(pred[a]) and vi[a] == np.count_nonzero(np.logical_and(bool_void, pred[a]))
        for b, gid in enumerate(ids):
            assert inter[a, b] == np.count_nonzero(np.logical_and(gts == gid, pred[a]))
    assert np.array_equal(gc, [(gts == gid).sum() for gid in ids])


@pytest.mark.parametrize("n", [1, 2, 1000, 200_003])
def test_library_sorts(lib, n):
    g = torch.Generator().manual_seed(n)
    v = torch.rand(n, generator=g).mul(5).floor().div(3)            # many ties, non-negative like the ratios
    v[0] = 0.0
    got = lib.sort_f32(v.to(DEV)).cpu()
    assert torch.equal(got, torch.sort(v).values)
    keys = torch.randint(-5, 5, (n,), generator=g, dtype=torch.int64) * (1 << 40)
    order = lib.argsort_i64(keys.to(DEV)).cpu().long()
    assert torch.equal(order, torch.argsort(keys, stable=True))     # stable: ties keep index order
    small = torch.randint(0, 1 << 30, (n,), generator=g, dtype=torch.int64) >> torch.randint(0, 30, (n,), generator=g)
    order = lib.argsort_i64(small.to(DEV), 30).cpu().long()           # 30-bit keys: fewer radix passes
    assert torch.equal(order, torch.argsort(small, stable=True))


# ------------------------------------------------------------------ golden aggregation cases on the device
AGG_CASES = ["random_f16", "random_f32", "chain_empty_single", "borderline", "nothing_merges", "min_members_1"]


@pytest.mark.parametrize("case", AGG_CASES)
def test_aggregation_golden_on_device(lib, case):
    """Every aggregate() fixture the reference's helpers produced (tests/golden/agg_helpers.npz: IoU exactly 1/5
    which is NOT > f32(0.2), a chain a~b~c, empty rows -> NaN IoU -> `[]`, singletons, a second label,
    min_aggragated_masks = 1, nothing merging) through the device path: merge matrix, components, kept groups,
    OR of the members, sequential confidence mean in the confidence dtype, label of the first member."""
    from beyond_fixed_forms_amd.projection import component_csr, groups_from_labels
    z = Z("agg_helpers.npz")
    g = lambda k: z[f"{case}.{k}"]
    n = int(g("n"))
    d = gio.unpack_bool_rows(g("ins"), n)
    labels = [str(s) for s in g("labels")]
    conf = torch.from_numpy(g("conf").copy())
    if str(g("conf_dtype")) == "torch.float16":
        conf = conf.half()
    min_members = int(g("min_members"))
    rows = pack_np(d)
    ids = {}
    lid = torch.tensor([ids.setdefault(s, len(ids)) for s in labels], dtype=torch.int32, device=DEV)
    area, _mw, cmask, hist, sig = lib.row_stats(rows)
    assert np.array_equal(area.cpu().numpy(), d.sum(1))
    # a10/a11: same_label & (iou > f32(0.2)), bit for bit the reference's merge matrix
    adj, inter = lib.merge_adjacency(rows, area, lid, 0.2, want_inter=True)
    assert np.array_equal(unpack(adj, len(labels)), g("merge"))
    a = area.cpu().numpy().astype(np.float32)
    fi = inter.cpu().numpy().astype(np.float32)
    with np.errstate(invalid="ignore", divide="ignore"):
        iou = fi / (a[:, None] + a[None, :] - fi)                        # float32, the expression of P:149-166
    assert np.array_equal(iou.view(np.uint32), g("iou_bits"))
    # a12: components (production union-find, three row orders) == find_unconnected_subgraphs_tensor
    self_loop = np.diag(g("merge"))
    r = len(labels)
    rng = np.random.default_rng(3)
    for order in (torch.arange(r, dtype=torch.int32, device=DEV), lib.argsort_i64(sig, lib.SIGNATURE_BITS),
                  torch.from_numpy(rng.permutation(r).astype(np.int32)).to(DEV)):
        comp = lib.merge_components(rows, area, lid, 0.2, order, cmask, hist).cpu().numpy()
        assert groups_from_labels(comp, self_loop) == gio.loads_groups(g("components"))
        assert groups_from_labels(comp, self_loop, min_members) == gio.loads_groups(g("groups"))
    assert groups_from_labels(lib.components(adj).cpu().numpy(), self_loop) == gio.loads_groups(g("components"))
    # a13: merge_masks on the kept groups
    offs, members, sizes, _ = component_csr(comp, self_loop, min_members)
    exp_kind = str(g("agg.kind"))
    if len(sizes) == 0:
        assert exp_kind != "rows"
        return
    t32 = lambda x: torch.from_numpy(np.ascontiguousarray(x, dtype=np.int32)).to(DEV)
    out, mean = lib.or_reduce_groups(rows, t32(offs), t32(members), int(sizes.max()), conf.to(DEV))
    assert exp_kind == "rows"
    assert np.array_equal(np.packbits(unpack(out, n), axis=-1, bitorder="little"), g("agg.ins_packed"))
    assert str(mean.dtype) == str(g("agg.conf_dtype"))
    assert np.array_equal(mean.cpu().float().numpy(), g("agg.conf"))
    assert [labels[i] for i in members[offs[:-1]]] == [str(s) for s in g("agg.final_class")]


def test_cosine_golden_on_device(lib):
    """compute_clip_similarity (R:93-115) fixtures: the embeddings of tests/golden/refine_helpers.npz through
    (a) bff_cosine_rows, which rounds every tensor op to the embedding dtype like the reference (float16: the
    values are multiples of 2^-11, identical texts give 0.99951172, equal rows tie exactly), and (b) the MFMA
    GEMM bff_cosine_gemm_f16 (float32 normalisation; tolerance of north_star: 1e-4)."""
    z = Z("refine_helpers.npz")
    e32 = torch.from_numpy(z["clip.emb_f32"][:, 0, :].copy())
    e16 = torch.from_numpy(z["clip.emb_f16"][:, 0, :].copy()).half()
    got32 = lib.cosine_rows(e32.to(DEV), e32.to(DEV)).cpu().numpy().astype(np.float64)
    d32 = np.abs(got32 - z["clip.sims_f32"])
    assert d32.max() <= 1e-4                                             # north_star's tolerance
    # float32: bff_cosine_rows rounds the exact dot / norms once each (float64 accumulation), the reference's BLAS
    # sdot and its vectorised norm accumulate in float32 in machine-dependent orders.  Everything agrees to one ulp
    # of 1.0 (2^-23) except self-similarities: ours are exactly 1.0 (dot == norm^2 before rounding), the reference's
    # own dot and norm disagree in the last bits (golden [3, 3] = 1 + 2^-22)
    off = d32 > 2.0 ** -23
    assert d32.max() <= 2.0 ** -22 and np.all(got32[off] == 1.0), (d32.max(), np.argwhere(off))
    got16 = lib.cosine_rows(e16.to(DEV), e16.to(DEV)).cpu().numpy().astype(np.float64)
    assert np.array_equal(got16, got16.astype(np.float16).astype(np.float64))       # float16 values
    assert got16[0, 0] == z["clip.sims_f16"][0, 0] == 0.99951171875      # the reference's self-similarity in fp16
    assert got16[0, 5] == got16[0, 0] and np.array_equal(got16[5], got16[0])        # equal embeddings tie
    # float16: products of float16 values and their 64-term sums are exact in float64, so the kernel's value of every
    # tensor op is the correctly rounded one -- bit-equal to the golden file on all 36 entries (checked against a NumPy
    # restatement of the kernel's arithmetic when this assertion was tightened from ">= 34")
    assert np.array_equal(got16, z["clip.sims_f16"]), np.argwhere(got16 != z["clip.sims_f16"])
    # MFMA GEMM on the same float16 embeddings (dim 64 = 2 k-steps): float32 normalisation
    pad = lambda x: x.to(DEV).contiguous()
    gemm = lib.cosine_gemm_f16(pad(e16), pad(e16)).cpu().numpy().astype(np.float64)
    e = e16.double()
    exact = ((e @ e.T) / (e.norm(dim=1, keepdim=True) * e.norm(dim=1, keepdim=True).T)).numpy()
    assert np.abs(gemm - exact).max() <= 1e-4
    # a-priori bound, not an observed one: the golden fp16 values carry four float16 roundings (dot, norms, their
    # product, the quotient) of half an ulp each (2^-12 relative below 1.0) = 2^-10 from the exact cosine, which the
    # GEMM approximates within north_star's 1e-4
    assert np.abs(gemm - z["clip.sims_f16"]).max() <= 2.0 ** -10 + 1e-4
    # 9000 x 768 against a 200-label bank (BASELINE config 5 at full size), vs float64
    gen = torch.Generator().manual_seed(5)
    a = torch.randn(9000, 768, generator=gen).half()
    b = torch.randn(200, 768, generator=gen).half()
    big = lib.cosine_gemm_f16(a.to(DEV), b.to(DEV)).cpu().double()
    ad, bd = a.double(), b.double()
    ref = (ad @ bd.T) / (ad.norm(dim=1, keepdim=True) * bd.norm(dim=1, keepdim=True).T)
    assert (big - ref).abs().max().item() <= 1e-4
    # and bff_cosine_rows against the same float64 values, on both dtypes
    sub = lib.cosine_rows(a[:50].to(DEV), b.to(DEV)).cpu().double()
    assert (sub - ref[:50]).abs().max().item() <= 2.0 ** -10
    sub32 = lib.cosine_rows(a[:50].float().to(DEV), b.float().to(DEV)).cpu().double()
    assert (sub32 - ref[:50]).abs().max().item() <= 5e-7


@pytest.mark.parametrize("na,nb,dim", [(9000, 200, 768), (20_001, 198, 768), (3000, 70, 512), (2049, 256, 32), (2048, 65, 736)])
def test_cosine_gemm_bank_stationary_shapes(lib, na, nb, dim, monkeypatch):
    """The bank-stationary MFMA kernel (bank strips in registers, A through LDS once; config 5's shape) on ragged row /
    column counts, several rounds of row tiles (na > 4 x CUs x 16), short k ranges: within north_star's 1e-4 of the
    float64 cosine, and equal to the LDS-staged kernel it replaces up to float32 summation order."""
    gen = torch.Generator().manual_seed(na + nb)
    a = torch.randn(na, dim, generator=gen).half().to(DEV)
    b = torch.randn(nb, dim, generator=gen).half().to(DEV)
    got = lib.cosine_gemm_f16(a, b).cpu().double()
    ad, bd = a.cpu().double(), b.cpu().double()
    ref = (ad @ bd.T) / (ad.norm(dim=1, keepdim=True) * bd.norm(dim=1, keepdim=True).T)
    assert got.shape == ref.shape and (got - ref).abs().max().item() <= 1e-4
    raw = lib.normalized_gemm_f16(a, b).cpu().double() if hasattr(lib, "normalized_gemm_f16") else None
    if raw is not None:                                       # bank rows taken as already normalised: a / |a| . b
        ref2 = (ad / ad.norm(dim=1, keepdim=True)) @ bd.T
        assert (raw - ref2).abs().max().item() <= 1e-4 * bd.norm(dim=1).max().item()


@pytest.mark.parametrize("thr", [-1.0, -0.5, 0.0])
def test_components_negative_threshold_with_empty_tiles(lib, thr):
    """iou_thres < 0: an empty row (IoU 0 with any non-empty row of its label, NaN with another empty row) joins
    the component of its label.  150 of 300 rows are empty, so whole 64-row tiles are empty in the sorted order --
    tiles the tile-pair filter may only drop when 0 > thr is false."""
    from beyond_fixed_forms_amd.projection import groups_from_labels
    rng = np.random.default_rng(23)
    r, n = 300, 20_000
    d = np.zeros((r, n), bool)
    for i in range(150):
        c = int(rng.integers(0, 5)) * 4000 + int(rng.integers(0, 300))
        d[i, c: c + int(rng.integers(50, 2000))] = True
    d = d[rng.permutation(r)]
    labels = rng.integers(0, 2, r)
    t = torch.from_numpy(d)
    iou = pref.pairwise_iou(t)
    merge = ((torch.from_numpy(labels)[:, None] == torch.from_numpy(labels)[None, :]) & (iou > thr)).numpy()
    exp = pref.connected_groups(torch.from_numpy(merge).float())
    rows = pack_np(d)
    lid = torch.tensor(labels, dtype=torch.int32, device=DEV)
    area, _mw, cmask, hist, sig = lib.row_stats(rows)
    self_loop = np.diag(merge)
    for order in (lib.argsort_i64(sig, lib.SIGNATURE_BITS), torch.arange(r, dtype=torch.int32, device=DEV)):
        comp = lib.merge_components(rows, area, lid, thr, order, cmask, hist).cpu().numpy()
        assert groups_from_labels(comp, self_loop) == exp
    if thr < 0:
        assert any(len(g) > 100 for g in exp)                    # the empty rows did join


def test_frustum_culling_is_exact(lib):
    """Sweep with and without the tile boxes on a spatially sorted cloud, cameras INSIDE the cloud, looking in all
    directions (most tiles are culled for most frames; boxes straddle the camera plane, lie behind it, or touch an
    image border): rows and both counters are bit-identical.  Points behind the camera that project in bounds and
    pass the depth test (no z > 0 test in the reference, P:57-67) are present on purpose."""
    from beyond_fixed_forms_amd.scene import morton_order
    rng = np.random.default_rng(31)
    n, h, w, m, nf = 60_000, 120, 160, 6, 24
    xyz = rng.uniform(-3, 3, (n, 3))
    xyz[:2000] = rng.normal(0, 0.02, (2000, 3))                       # a dense blob around the first camera
    xyz = xyz[morton_order(xyz)]
    nw, n_pad = (n + 63) // 64, ((n + 1023) // 1024) * 1024
    soa = np.zeros((3, n_pad)); soa[:, :n] = xyz.T
    k33 = np.array([[140.3, 0.0, 79.5], [0.0, 140.3, 59.5], [0.0, 0.0, 1.0]])
    inv = []
    for f in range(nf):
        a, b = rng.uniform(-np.pi, np.pi, 2)
        rz = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])
        ry = np.array([[np.cos(b), 0, np.sin(b)], [0, 1, 0], [-np.sin(b), 0, np.cos(b)]])
        pose = np.eye(4)
        pose[:3, :3] = rz @ ry
        pose[:3, 3] = 0 if f == 0 else rng.uniform(-2.5, 2.5, 3)
        inv.append(np.linalg.inv(pose))
    depth = rng.uniform(0.02, 5.0, (nf, h * w)).astype(np.float32)
    depth[0] = 0.03                                                   # tiny depth: points just behind camera 0 pass |z - d| < 0.08
    masks = [rng.random((m, h * w)) < 0.5 for _ in range(nf)]
    bits, wb = maskbits_from_dense(lib, masks, h * w)
    i32 = lambda a: torch.tensor(a, dtype=torch.int32, device=DEV)
    xs = torch.from_numpy(soa).to(DEV)
    args = (xs, n, torch.from_numpy(np.stack(inv).reshape(nf, 16)).to(DEV), k33, torch.from_numpy(depth).to(DEV),
            i32(list(range(nf))), h, w, 0.08, bits, wb, i32(list(range(nf))), i32([m * f for f in range(nf)]),
            i32([m] * nf), i32([1] * nf))
    out = []
    for tb in (None, lib.point_tile_bounds(xs, n)):
        rows = torch.zeros((m * nf, nw), dtype=torch.int64, device=DEV)
        mc = torch.zeros(n, dtype=torch.int32, device=DEV)
        vc = torch.zeros(n, dtype=torch.int32, device=DEV)
        lib.project_views(*args, rows, mc, vc, tile_bounds=tb)
        out.append((rows, mc, vc))
    for a, b in zip(*out):
        assert torch.equal(a, b)
    assert int(out[0][2].sum().item()) > 1000
    # the boxes themselves
    tb = lib.point_tile_bounds(xs, n).cpu().numpy()
    for t in (0, 7, tb.shape[0] - 1):
        pts = xyz[256 * t: 256 * (t + 1)]
        assert np.array_equal(tb[t, :3], pts.min(0)) and np.array_equal(tb[t, 3:], pts.max(0))
    # behind-the-camera visibility exists in this data (otherwise the back cone would be untested)
    cam0 = (inv[0] @ np.concatenate([xyz, np.ones((n, 1))], 1).T).T
    assert ((cam0[:, 2] < 0) & (np.abs(cam0[:, 2] - 0.03) < 0.08)).sum() > 10


@pytest.mark.parametrize("name", ["labelled_a", "labelled_b", "agnostic", "no_preds"])
def test_evaluation_assignment_golden(lib, name):
    """evaluation.assign_instances_for_scan (bit rows + one popcount Gram on the device) reproduces the nested dicts
    of the reference's ScanNetEval.assign_instances_for_scan (scannetv2_inst_eval.py:265-365) on the golden cases:
    labels the evaluator does not know, predictions below the minimum region size, empty masks, void classes,
    ignored instance ids, class-agnostic mode, no predictions at all; also from pre-packed device rows."""
    from beyond_fixed_forms_amd.evaluation import assign_instances_for_scan
    from oracle.eval_ref import flatten_assignment
    z = Z("eval_assign.npz")
    labels = [str(s) for s in z["class_labels"]]
    preds, sem, ins, use_label, exp = gio.eval_case(z, name)
    ev_labels = labels if use_label else ["class_agnostic"]
    gt2pred, pred2gt = assign_instances_for_scan(preds, sem, ins, labels, use_label=use_label, device=DEV)
    gio.same_assignment(flatten_assignment(gt2pred, pred2gt, ev_labels), exp)
    if preds:
        rows = pack_np(np.stack([p["pred_mask"] != 0 for p in preds]))
        slim = [{k: v for k, v in p.items() if k != "pred_mask"} for p in preds]
        gt2pred, pred2gt = assign_instances_for_scan(slim, sem, ins, labels, use_label=use_label, device=DEV, pred_rows=rows)
        gio.same_assignment(flatten_assignment(gt2pred, pred2gt, ev_labels), exp)


@pytest.mark.parametrize("tag", ["f32", "f16"])
def test_box_filter_golden(lib, tag):
    """The arithmetic around the CLIP box filter (segmentation_2d.py:324-337, 388-396) against what the reference's own
    functions produced with injected embeddings (tests/golden/box_filter.npz): per-class means of normalised
    description encodings, F.normalize(box_emb) @ text_mean.T on the matrix cores, the `>= 0.2` decisions.
    Tolerance: 1e-4 (north_star) for float32 encodings; the reference's fp16 path rounds every tensor op to fp16,
    so its values sit up to a few fp16 ulps (2^-11 each) from the exact ones the device approximates."""
    from beyond_fixed_forms_amd import boxfilter
    z = Z("box_filter.npz")
    dt = torch.float32 if tag == "f32" else torch.float16
    counts = z[f"{tag}.desc_counts"]
    desc = torch.from_numpy(z[f"{tag}.desc"]).to(dt)
    enc = [desc[a:b] for a, b in zip(np.cumsum([0] + list(counts[:-1])), np.cumsum(counts))]
    means = boxfilter.average_description_embeddings(enc, DEV)
    assert means.dtype == dt and tuple(means.shape) == tuple(z[f"{tag}.desc_means"].shape)
    tol = 1e-4 if tag == "f32" else 3 * 2.0 ** -11
    assert np.abs(means.float().cpu().numpy() - z[f"{tag}.desc_means"]).max() <= tol
    # the filter, fed with the reference's own text mean (so the comparison isolates the product)
    emb = torch.from_numpy(z[f"{tag}.box_emb"]).to(dt)
    text = torch.from_numpy(z[f"{tag}.desc_means"][0:1]).to(dt)
    boxes = torch.from_numpy(z[f"{tag}.boxes"])
    phrases = [f"p{i}" for i in range(emb.shape[0])]
    b_f, logits, phr = boxfilter.bbox_filter(boxes, phrases, emb.to(DEV), text.to(DEV), clip_threshold=0.2)
    sims = boxfilter.box_similarities(emb.to(DEV), text.to(DEV)).cpu().numpy()[:, 0]
    exact = (torch.nn.functional.normalize(emb.double()) @ text.double().T).numpy()[:, 0]
    assert np.abs(sims - exact).max() <= 1e-4                                     # vs float64 on the same inputs
    kept_ref = z[f"{tag}.kept"]
    assert np.abs(sims[kept_ref] - z[f"{tag}.logits"]).max() <= tol
    clear = np.abs(exact - 0.2) > tol                                             # decisions away from the threshold
    got_kept = np.array([int(p[1:]) for p in phr])
    assert np.array_equal(got_kept[clear[got_kept]], kept_ref[clear[kept_ref]])
    assert torch.equal(b_f, boxes[torch.from_numpy(got_kept)]) and logits.shape == (len(got_kept), 1)
    # batched form: every box of many frames against a 200-label bank in one launch (BASELINE config 5 shape)
    gen = torch.Generator().manual_seed(9)
    many = torch.randn(5000, 768, generator=gen).half()
    bank = torch.nn.functional.normalize(torch.randn(200, 768, generator=gen)).half()
    big = boxfilter.box_similarities(many.to(DEV), bank.to(DEV)).cpu().double()
    ref = torch.nn.functional.normalize(many.double()) @ bank.double().T
    assert (big - ref).abs().max().item() <= 1e-4
    assert boxfilter.bbox_filter(boxes[:0], [], emb[:0], text, 0.2)[1:] == ([], [])                # SEG:354-355
