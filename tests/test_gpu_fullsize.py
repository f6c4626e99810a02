"""BASELINE.json configurations at their full sizes on one MI355X (-m gpu).

cfg 2 / cfg 3 live in ``test_gpu_parity.py`` (``test_full_size_*``); here: landmark selection at 1 M / 1 k and
16 M / 4 k against the float64 oracle, cfg 5 (16 M swiss cheese, 4 k landmarks) and cfg 4 (2 M points in 6D, 2 k
landmarks, ``max_dimension=2``, ``points_per_edge=8``) end to end.  The reference's own large-scale checks are
``tests/test_flooder.py:119-157`` (cross-path agreement within 1e-4) and ``:207-211`` (monotone filtration)."""
import os

import numpy as np
import pytest
import torch

import flooder_amd as fa
from flooder_amd import _native, core
from oracle import flood_oracle as fo
from helpers import assert_close_filtration, assert_tree_matches_kdtree, kdtree_face_values

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "the -m gpu tests need a GPU"
    _native.load()  # fails loudly when the HIP library is missing
    return torch.device("cuda:0")


def _assert_fps(points_np, got_idx):
    """Float64 replay of the selection: every pick is a farthest point (near-ties within float32 rounding
    allowed and counted); picks that attain the float64 maximum exactly must also follow ``exact_fps``' tie break
    (first index), so with no near-tie the selection IS ``exact_fps``' selection."""
    rep = fo.check_fps(points_np, got_idx)
    assert rep["exact"] + rep["near_ties"] == len(got_idx) - 1
    assert rep["exact_first_index"] == rep["exact"], rep
    return rep


def test_fps_1m_1k_matches_exact_fps(dev):
    """cfg 2's landmark selection: 1 M Gaussian points, 1000 landmarks, every pick replayed in float64."""
    torch.manual_seed(42)
    pts = torch.randn(1_000_000, 3)
    got = core.fps_indices(pts.to(dev), 1000, 0).cpu().numpy()
    rep = _assert_fps(pts.numpy(), got)
    assert rep["near_ties"] <= 3, rep
    assert np.array_equal(got[:48], fo.exact_fps(pts.numpy(), 48, 0))


@pytest.fixture(scope="module")
def cheese16m(dev):
    pts = fa.generate_swiss_cheese_points(16_000_000, k=6, seed=42)[0]
    tp = pts.to(dev)
    idx = core.fps_indices(tp, 4000, 0)
    return pts, tp, idx


def test_cfg5_fps_16m_4k(cheese16m):
    """cfg 5's landmark selection: the first 128 picks replayed in float64 over all 16 M points (a uniform
    cloud of 16 M points does produce float64 near-ties below float32 resolution; they are counted, and the
    picks are index-equal to ``exact_fps`` when there is none), all 4000 picks distinct and well separated."""
    pts, tp, idx = cheese16m
    got = idx.cpu().numpy()
    assert len(set(got.tolist())) == 4000
    _assert_fps(pts.numpy(), got[:128])
    # FPS is a greedy 2-approximation of the k-centre problem: the picks are pairwise at least as far apart as the
    # covering radius after the last pick (checked on the landmark set with a kd-tree, float64)
    from scipy.spatial import cKDTree
    L = pts.numpy()[got].astype(np.float64)
    dd, _ = cKDTree(L).query(L, k=2)
    sep = dd[:, 1].min()
    rng = np.random.default_rng(0)
    probe = pts.numpy()[rng.choice(pts.shape[0], 200_000, replace=False)].astype(np.float64)
    cover = cKDTree(L).query(probe)[0].max()
    assert sep >= cover * (1 - 1e-5), (sep, cover)


def test_cfg5_full_size_16m_cheese(cheese16m, dev):
    """BASELINE cfg 5 at full size: 16 M swiss-cheese points (five-level box tree, 256 MB cloud), 4000
    landmarks, points_per_edge 30.  EVERY tetrahedron (25 k x 4960 samples), triangle and edge is compared with
    scipy's kd-tree over all 16 M points (all host cores: seconds on the GPU box)."""
    pts, tp, idx = cheese16m
    lms = tp[idx]
    st = fa.flood_complex(tp, lms, return_simplex_tree=True)
    tets = st.simplices_of_dimension(3)
    vals = st.filtrations_of_dimension(3)
    assert len(tets) > 20_000 and np.isfinite(vals).all()
    assert all(st.filtration([i]) == 0.0 for i in range(0, 4000, 97))
    P, L = pts.numpy(), lms.cpu().numpy()
    if (os.cpu_count() or 1) >= 32:
        n = assert_tree_matches_kdtree(st, P, L, 30, 3, "cfg5", lower=True, strict=True)
        assert n == len(tets) + len(st.simplices_of_dimension(2)) + len(st.simplices_of_dimension(1))
    else:   # (125 M queries over 16 M points: on a few cores the 200 deepest + 2000 random tetrahedra)
        rng = np.random.default_rng(1)
        pick = np.unique(np.concatenate([np.argsort(-vals)[:200], rng.choice(len(tets), size=2000, replace=False)]))
        assert assert_tree_matches_kdtree(st, P, L, 30, 3, "cfg5 tetrahedra", pick_top=pick, lower=False, strict=True) >= 2000
    big = np.argsort(-vals)[:200]                     # the 200 tetrahedra reaching deepest into the voids
    assert float(vals[big].min()) > 5 * float(np.median(vals))
    for d in (1, 2, 3):  # monotone: faces never above cofaces (tests/test_flooder.py:207-211)
        rows = st.simplices_of_dimension(d)
        own = st.filtrations_of_dimension(d)
        for j in range(d + 1):
            loc = st._locate(d - 1, np.delete(rows, j, axis=1))
            assert (loc >= 0).all() and (st.filtrations_of_dimension(d - 1)[loc] <= own).all()


def test_cfg4_full_size_2m_6d(dev):
    """BASELINE cfg 4 at full size: 2 M Gaussian points in 6D, 2000 landmarks, max_dimension 2,
    points_per_edge 8 (the tractable setting).  EVERY triangle (1.05 M x 36 samples) and every edge is
    compared with a 6-D kd-tree over all points (all host cores); all triangles through the monotone rule."""
    from scipy.spatial import cKDTree
    torch.manual_seed(42)
    pts = torch.randn(2_000_000, 6)
    tp = pts.to(dev)
    lms = fa.generate_landmarks(tp, 2000, start_idx=0)
    P = pts.numpy()
    got_idx = core.fps_indices(tp, 2000, 0).cpu().numpy()
    _assert_fps(P, got_idx[:64])
    st = fa.flood_complex(tp, lms, max_dimension=2, points_per_edge=8, return_simplex_tree=True)
    tris = st.simplices_of_dimension(2)
    vals = st.filtrations_of_dimension(2)
    assert len(tris) > 500_000 and np.isfinite(vals).all()
    L = lms.cpu().numpy()
    tree = cKDTree(P)
    if (os.cpu_count() or 1) >= 32:
        ref = kdtree_face_values(tree, L, tris, 8, 2)          # (n, 36) samples each, 6-D kd-tree, all host cores
        assert_close_filtration(vals, ref, P, "cfg4: every triangle", strict=True)
    else:   # (38 M six-dimensional queries take minutes on a few cores: the 100 largest + 2000 random triangles there)
        rng = np.random.default_rng(2)
        pick = np.unique(np.concatenate([np.argsort(-vals)[:100], rng.choice(len(tris), size=2000, replace=False)]))
        assert_close_filtration(vals[pick], kdtree_face_values(tree, L, tris[pick], 8, 2), P, "cfg4 triangle sample", strict=True)
    e = st.simplices_of_dimension(1)
    ev = st.filtrations_of_dimension(1)
    assert np.isfinite(ev).all() and (st.filtrations_of_dimension(0) == 0.0).all()
    for j in range(3):  # monotone over all triangles
        loc = st._locate(1, np.delete(tris, j, axis=1))
        assert (loc >= 0).all() and (ev[loc] <= vals).all()
    # edges: their own samples (the 8 lattice points of the edge) against the kd-tree
    assert_close_filtration(ev, kdtree_face_values(tree, L, e, 8, 1), P, "cfg4: every edge", strict=True)


@pytest.mark.parametrize("case", ["gauss1m", "torus300k", "eight2d", "line1d", "dups"])
def test_bucketed_fps_equals_brute_force(dev, case):
    """flooder_fps_indexed_f32 (buckets over the Hilbert-sorted cloud, as the reference's bucket FPS) selects exactly
    what flooder_fps_f32 (one full sweep per landmark) selects - every index, including ties (duplicated points)."""
    g = torch.Generator().manual_seed(5)
    if case == "gauss1m":
        pts, k, start = torch.randn(1_000_000, 3, generator=g), 1000, 0
    elif case == "torus300k":
        pts, k, start = torch.as_tensor(fo.noisy_torus(300_000, seed=5)), 700, 1234
    elif case == "eight2d":
        pts, k, start = fa.generate_figure_eight_points_2d(400_000, noise_std=0.02, seed=5).float(), 500, 7
    elif case == "line1d":
        pts, k, start = torch.rand(250_000, 1, generator=g), 300, 0
    else:
        base = torch.rand(150_000, 3, generator=g)
        pts, k, start = torch.cat([base, base[:100_000]]), 400, 3       # 100 k exact duplicates
    tp = pts.to(dev)
    a = core.fps_indices(tp, k, start, method="brute").cpu().numpy()
    b = core.fps_indices(tp, k, start, method="bucket").cpu().numpy()
    assert np.array_equal(a, b)
    lib = _native.load()
    for rpl, sw in ((4, 8), (1, 1), (4, 1)):
        try:
            assert lib.flooder_set_option(b"fps_rpl", rpl) == 0 and lib.flooder_set_option(b"fps_switch", sw) == 0
            c = core.fps_indices(tp, min(k, 200), start, method="bucket").cpu().numpy()
        finally:
            lib.flooder_set_option(b"fps_rpl", 0)
            lib.flooder_set_option(b"fps_switch", 0)
        assert np.array_equal(a[:len(c)], c), (rpl, sw)


def test_bucketed_fps_16m_equals_brute_force(cheese16m):
    pts, tp, idx = cheese16m            # (the fixture's selection runs the default = bucketed path)
    brute = core.fps_indices(tp, 600, 0, method="brute").cpu().numpy()
    assert np.array_equal(idx.cpu().numpy()[:600], brute)


@pytest.mark.parametrize("case", ["lattice_ties", "two_clusters", "dim5", "dim8", "all_points", "dups_of_start"])
def test_batched_fps_hard_cases_equal_brute_force(dev, case, monkeypatch):
    """flooder_fps_batched_f32 selects several landmarks per launch when the runners-up of the arg-max provably stay
    the next landmarks (csrc/flood_fps2.hip).  Inputs built to break a wrong rule: a regular lattice (thousands of
    exactly equal running minima: only the index decides), two tight clusters (the runners-up are neighbours of the
    winner: the batch must stop), dimensions 5 and 8, as many landmarks as points (the minima reach zero), copies of
    the start point.  Against the brute-force kernel (one full sweep per landmark) and, for the first picks, numpy."""
    rng = np.random.default_rng(9)
    if case == "lattice_ties":
        g = np.arange(48, dtype=np.float32)
        P = np.stack(np.meshgrid(g, g, g, indexing="ij"), axis=-1).reshape(-1, 3)
        P = P[rng.permutation(len(P))]
        k, start = 700, 5
    elif case == "two_clusters":
        P = np.concatenate([rng.normal(0, 1e-3, (150_000, 3)), rng.normal(0, 1e-3, (150_000, 3)) + 1.0]).astype(np.float32)
        k, start = 300, 0
    elif case == "dim5":
        P, k, start = rng.normal(size=(250_000, 5)).astype(np.float32), 500, 17
    elif case == "dim8":
        P, k, start = rng.random((220_000, 8)).astype(np.float32), 400, 0
    elif case == "all_points":
        P, k, start = rng.normal(size=(3000, 3)).astype(np.float32), 3000, 1
    else:
        base = rng.normal(size=(210_000, 3)).astype(np.float32)
        base[1000:1400] = base[7]
        P, k, start = base, 450, 7
    tp = torch.as_tensor(P, device=dev)
    monkeypatch.setattr(core, "FPS_BUCKET_MIN_POINTS", 0)
    a = core.fps_indices(tp, k, start, method="brute").cpu().numpy()
    monkeypatch.setattr(core, "FPS_BATCHED", True)
    b = core.fps_indices(tp, k, start, method="bucket").cpu().numpy()
    assert np.array_equal(a, b), int(np.argmax(a != b))
    assert core.LAST_FPS_LAUNCHES < k or case in ("two_clusters", "all_points")
    assert np.array_equal(a[:40], fo.exact_fps(P, 40, start))
    lib = _native.load()
    for sw in (2, 7):   # switch to the batched steps almost at once
        try:
            assert lib.flooder_set_option(b"fps_switch", sw) == 0
            c = core.fps_indices(tp, min(k, 250), start, method="bucket").cpu().numpy()
        finally:
            lib.flooder_set_option(b"fps_switch", 0)
        assert np.array_equal(a[:len(c)], c), sw
    # the other paths of the same entry point: one candidate per lane (what > 64 candidates fall back to) and rounds
    # of launches with a counter read-back (what a host without pinned memory falls back to)
    for opt in (b"fps_lane_best", b"fps_rounds"):
        try:
            assert lib.flooder_set_option(opt, 1) == 0
            c = core.fps_indices(tp, k, start, method="bucket").cpu().numpy()
        finally:
            lib.flooder_set_option(opt, 0)
        assert np.array_equal(a, c), opt
    if P.shape[1] <= 3:  # the one-landmark-per-launch kernels stay selectable
        monkeypatch.setattr(core, "FPS_BATCHED", False)
        assert np.array_equal(a, core.fps_indices(tp, k, start, method="bucket").cpu().numpy())


def test_method_switch_and_byte_offset_bound_at_2_28_points(dev):
    """The cell sweep addresses the cloud with 32-bit BYTE offsets (16-byte rows: below 2^28 - 16 points) and
    ``method="auto"`` switches to the tree sweep at N >= 2^28 - 64 (``core.py`` flood_complex).  A 3.2 GB cloud on
    both sides of the switch: the last size "auto" gives to the cell sweep, the first it gives to the tree sweep (the
    cell sweep still accepts it when asked: byte offsets up to 4 GiB - 1 KiB), and 2^28 points, which the cell sweep
    must refuse.  The landmarks sit in the corner of the cube with the largest coordinates, so the rows their
    neighbourhoods need are among the last of the curve order.  cell == tree sweep bit for bit; the longest edge
    against a brute-force minimum over all points in float64."""
    n_all = 1 << 28
    n_cell, n_tree = n_all - 80, n_all - 64
    g = torch.Generator(device=dev).manual_seed(7)
    pts = torch.rand((n_all, 3), generator=g, device=dev, dtype=torch.float32)
    lms = torch.tensor([[0.9990 + 0.0008 * ((i >> 0) & 1), 0.9990 + 0.0008 * ((i >> 1) & 1), 0.9990 + 0.0008 * ((i >> 2) & 1)]
                        for i in range(8)], dtype=torch.float32, device=dev)
    pts[:8] = lms
    lib = _native.load()
    for n in (n_cell, n_tree):
        cloud = pts[:n]
        index = core.PointIndex(cloud)
        auto = fa.flood_complex(cloud, lms, points_per_edge=6, return_simplex_tree=True, index=index)
        other = fa.flood_complex(cloud, lms, points_per_edge=6, method="bvh" if n == n_cell else "cell",
                                 return_simplex_tree=True, index=index)
        for d in (1, 2, 3):
            assert np.array_equal(auto.filtrations_of_dimension(d), other.filtrations_of_dimension(d)), (n, d)
        e = auto.simplices_of_dimension(1)
        ev = auto.filtrations_of_dimension(1)
        j = int(np.argmax(ev))
        w = torch.linspace(0, 1, 6, device=dev, dtype=torch.float64)
        a, b = lms[int(e[j, 0])].double(), lms[int(e[j, 1])].double()
        samples = ((1 - w)[:, None] * a[None] + w[:, None] * b[None]).float().double()
        best = torch.full((6,), float("inf"), device=dev, dtype=torch.float64)
        for c0 in range(0, n, 1 << 23):
            blk = cloud[c0:c0 + (1 << 23)].to(torch.float64)
            d2 = ((blk[None, :, :] - samples[:, None, :]) ** 2).sum(-1)
            best = torch.minimum(best, d2.min(dim=1).values)
        ref = float(best.max().sqrt())
        assert abs(ev[j] - ref) <= 2e-5 * ref + 5e-7, (n, ev[j], ref)   # (the sample coordinates are rebuilt in float32 on the device)
        del index
    with pytest.raises(RuntimeError, match="too large"):
        fa.flood_complex(pts, lms, points_per_edge=6, method="cell")
    del pts
    torch.cuda.empty_cache()
