"""Parity of the HIP path (through the C ABI in libflooder_hip.so) against the reference's golden
outputs and against the oracle.  Runs on a real MI355X only (-m gpu)."""
import os

import numpy as np
import pytest
import torch

import flooder_amd as fa
from flooder_amd import _native, core
from oracle import flood_oracle as fo
from helpers import assert_tree_matches_kdtree, GOLDEN, assert_close_filtration, e2e_cases, load_e2e, dict_values

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "the -m gpu tests need a GPU"
    lib = _native.load()  # fails loudly when the HIP library is missing
    buf = (b" " * 64)
    import ctypes
    cbuf = ctypes.create_string_buffer(64)
    assert lib.flooder_device_arch(0, cbuf, 64) == 0
    assert cbuf.value.decode().startswith("gfx950"), cbuf.value
    return torch.device("cuda:0")


def native_min_dist(points, centers, radii, samples, dev):
    """compute_mask + compute_filtration replacement for explicit sample points: one pass per
    simplex with verts = identity so that p = sum_j w_j e_j = the given sample exactly."""
    lib = _native.load()
    pts = torch.as_tensor(points, device=dev)
    dim = pts.shape[1]
    axis = 0
    order = torch.argsort(pts[:, axis])
    pad = core._pad_rows(pts[order], lib.flooder_padded_dim(dim))
    search = pad[:, axis].contiguous()
    out, cnts = [], []
    for s in range(samples.shape[0]):
        verts = torch.eye(dim, device=dev).unsqueeze(0)
        w = torch.as_tensor(samples[s], device=dev)
        faces = core._FaceTable(None, w.shape[0], dev)
        core.LAST_STATS.reset()
        _, dist = core._sweep_dimension_hip(pad, search, axis, dim, verts,
                                            torch.as_tensor(centers[s:s + 1], device=dev),
                                            torch.as_tensor(radii[s:s + 1], device=dev), w, faces, None,
                                            want_dist=True)
        out.append(dist.cpu().numpy()[0])
        cnts.append(core.LAST_STATS.candidate_pairs)
    return np.stack(out), np.array(cnts)


@pytest.mark.parametrize("name", ["k2d", "k3d", "k5d"])
def test_kernels_match_reference_triton_vectors(name, dev):
    """Kernel-level golden vectors produced by the reference's own Triton kernels."""
    z = np.load(os.path.join(GOLDEN, f"kernel_{name}.npz"))
    d, cnts = native_min_dist(z["points"], z["centers"], z["radii"], z["samples"], dev)
    assert np.array_equal(cnts, z["mask_rowsum"])  # integer work: exact
    ref = z["min_dist"]
    assert np.array_equal(np.isinf(d), np.isinf(ref))
    fin = np.isfinite(ref)
    assert_close_filtration(d[fin], ref[fin], z["points"], name)


def test_dpp_wave_reductions(dev):
    lib = _native.load()
    g = torch.Generator().manual_seed(0)
    for trial in range(8):
        x = torch.rand(64, generator=g)
        if trial == 1:
            x[17] = float("inf")
        if trial == 2:
            x[:] = float("inf")
        xd = x.to(dev)
        out = torch.empty(128, device=dev)
        assert lib.flooder_selftest(_native.ptr(xd), _native.ptr(out), 0) == 0
        torch.cuda.synchronize()
        assert torch.equal(out[:64].cpu(), torch.full((64,), float(x.min())))
        assert torch.equal(out[64:].cpu(), torch.full((64,), float(x.max())))


@pytest.mark.parametrize("method", ["auto", "bvh", "ball"])
@pytest.mark.parametrize("name", e2e_cases())
def test_e2e_matches_reference_goldens(name, method, dev):
    z, kw, keys = load_e2e(name)
    torch.manual_seed(int(z["weight_seed"]))
    fc = fa.flood_complex(torch.as_tensor(z["points"], device=dev), torch.as_tensor(z["landmarks"], device=dev),
                          method=method, **kw)
    assert set(keys) == set(fc)
    assert_close_filtration(dict_values(fc, keys), z["filtration_f32"], z["points"], name)


def test_plain_and_packed_variants_are_bit_identical(dev):
    lib = _native.load()
    z, kw, keys = load_e2e("torus3d_grid")
    pts, lms = torch.as_tensor(z["points"], device=dev), torch.as_tensor(z["landmarks"], device=dev)
    try:
        assert lib.flooder_set_option(b"sweep_variant", 1) == 0
        a = fa.flood_complex(pts, lms, method="ball", **kw)
    finally:
        assert lib.flooder_set_option(b"sweep_variant", 0) == 0
    b = fa.flood_complex(pts, lms, method="ball", **kw)
    assert a == b


@pytest.mark.parametrize("name", ["torus3d_grid30", "cheese3d_grid", "eight2d_rand", "gauss6d_maxdim2"])
def test_culled_sweep_is_bit_identical_to_ball_sweep(name, dev):
    """Culling must not change a single bit: exact-NN (bvh) == exhaustive in-ball sweep (ball)."""
    z, kw, keys = load_e2e(name)
    pts, lms = torch.as_tensor(z["points"], device=dev), torch.as_tensor(z["landmarks"], device=dev)
    torch.manual_seed(1)
    a = fa.flood_complex(pts, lms, method="bvh", **kw)
    torch.manual_seed(1)
    b = fa.flood_complex(pts, lms, method="ball", **kw)
    assert a == b
    if pts.shape[1] in (2, 3):
        torch.manual_seed(1)
        c = fa.flood_complex(pts, lms, method="cell", **kw)
        assert a == c


@pytest.mark.parametrize("name", ["torus3d_grid30", "eight2d_rand", "gauss6d_maxdim2"])
def test_sorted_sample_sweep_is_bit_identical(name, dev, monkeypatch):
    """The tree sweep over spatially sorted samples (csrc/flood_sorted.hip: tiles of 64 consecutive samples of a
    Z-order of ALL (simplex, sample) pairs, default above 3D) against the per-simplex tree sweep and, on the golden
    inputs, against the reference's values: not a bit may differ."""
    z, kw, keys = load_e2e(name)
    pts, lms = torch.as_tensor(z["points"], device=dev), torch.as_tensor(z["landmarks"], device=dev)
    monkeypatch.setattr(core, "BVH_SORTED_MIN_SAMPLES", 0)
    res = {}
    for mode in (True, False):
        monkeypatch.setattr(core, "BVH_SORTED_SAMPLES", mode)
        torch.manual_seed(int(z["weight_seed"]))
        res[mode] = fa.flood_complex(pts, lms, method="bvh", **kw)
    assert res[True] == res[False]
    assert_close_filtration(dict_values(res[True], keys), z["filtration_f32"], z["points"], name)


@pytest.mark.parametrize("n,dim", [(17, 6), (1000, 8), (1025, 4), (5000, 4), (70_001, 5), (300_000, 6)])
def test_kd_order_is_a_permutation_whose_aligned_groups_are_tree_cells(n, dim, dev, monkeypatch):
    """Point index above 3D (csrc/flood_index.hip, kd_order): the rows are a permutation of the cloud, and every
    aligned group of 16 * 2^j rows splits into its two halves along ONE axis - exactly where kd_local_kernel sorts
    exact coordinates (groups of up to 1024 rows), within 2^-8 of the group's extent where a radix sort of quantised
    coordinates does (above)."""
    g = torch.Generator().manual_seed(n + dim)
    pts = torch.randn(n, dim, generator=g).to(dev)
    monkeypatch.setattr(core, "KD_ORDER_ABOVE_DIM", 3)
    idx = core.PointIndex(pts)
    assert idx.kd
    o = idx.order32.long().cpu().numpy()
    assert np.array_equal(np.sort(o), np.arange(n))
    rows = idx.pts[:n, :dim].cpu().numpy()
    assert np.array_equal(rows, pts.cpu().numpy()[o])
    assert bool(torch.isinf(idx.pts[n:]).all())
    size = 32
    while size // 2 < n:
        slack = 0.0 if size <= 1024 else 2.0 ** -8
        for a in range(0, n, size):
            left, right = rows[a:a + size // 2], rows[a + size // 2:a + size]
            if len(right) == 0:
                continue
            grp = rows[a:a + size]
            ext = grp.max(0) - grp.min(0)
            gap = right.min(0) - left.max(0) + slack * ext * 1.01   # >= 0 along the split axis
            assert (gap >= 0).any(), (size, a, gap)
        size *= 2
        if size > 8192 and n > 100_000:   # (the upper levels of the large cloud: a few groups each)
            size *= 4


@pytest.mark.parametrize("case", ["duplicates", "line", "one_point", "two_clusters"])
def test_kd_order_on_degenerate_clouds(case, dev, monkeypatch):
    """k-d order where a split has nothing to split: every point repeated, a cloud on a line in 5D, ONE point repeated,
    two far clusters of different size.  Still a permutation, and the sweep over it equals the sweep over the curve order."""
    rng = np.random.default_rng(11)
    dim = 5
    if case == "duplicates":
        base = rng.normal(size=(3000, dim)).astype(np.float32)
        P = np.concatenate([base, base, base[:1500]])
    elif case == "line":
        t = rng.uniform(-1, 1, size=(9000, 1)).astype(np.float32)
        P = (t * np.array([[1.0, -2.0, 0.5, 0.0, 3.0]], dtype=np.float32)).astype(np.float32)
    elif case == "one_point":
        P = np.tile(np.array([[0.25, -1.0, 2.0, 0.0, 7.0]], dtype=np.float32), (2100, 1))
    else:
        P = np.concatenate([rng.normal(size=(7000, dim)), 50.0 + 0.01 * rng.normal(size=(333, dim))]).astype(np.float32)
    tp = torch.as_tensor(P, device=dev)
    monkeypatch.setattr(core, "KD_ORDER_ABOVE_DIM", 3)
    idx = core.PointIndex(tp)
    assert idx.kd
    o = idx.order32.long().cpu().numpy()
    assert np.array_equal(np.sort(o), np.arange(len(P)))
    assert np.array_equal(idx.pts[:len(P), :dim].cpu().numpy(), P[o])
    if case == "one_point":
        return
    L = tp[torch.as_tensor(fo.exact_fps(P, 12, 0), device=dev)]
    monkeypatch.setattr(core, "BVH_SORTED_MIN_SAMPLES", 0)
    out = {}
    for above in (3, 8):
        monkeypatch.setattr(core, "KD_ORDER_ABOVE_DIM", above)
        try:
            out[above] = fa.flood_complex(tp, L, max_dimension=2, points_per_edge=5, method="bvh")
        except Exception as e:   # (a degenerate landmark set: Qhull refuses it on either order alike)
            out[above] = repr(type(e))
    assert out[3] == out[8]


@pytest.mark.parametrize("dim,kw", [(6, dict(max_dimension=2, points_per_edge=6)), (4, dict(max_dimension=3, points_per_edge=4)),
                                    (5, dict(max_dimension=2, num_rand=40))])
def test_kd_order_and_curve_order_give_the_same_filtration(dim, kw, dev, monkeypatch):
    """Any order of the rows is a valid index (the boxes are taken from the rows): k-d order and Hilbert order, sorted
    sweep and per-simplex tree sweep, bit for bit - and the landmark selection picks the same indices on either."""
    rng = np.random.default_rng(dim)
    P = rng.normal(size=(60_000, dim)).astype(np.float32)
    tp = torch.as_tensor(P, device=dev)
    monkeypatch.setattr(core, "BVH_SORTED_MIN_SAMPLES", 0)
    out, lms = {}, {}
    for above in (3, 8):
        monkeypatch.setattr(core, "KD_ORDER_ABOVE_DIM", above)
        lm = fa.generate_landmarks(tp, 28, start_idx=0)
        lms[above] = lm.cpu().numpy()
        for srt in (True, False):
            monkeypatch.setattr(core, "BVH_SORTED_SAMPLES", srt)
            torch.manual_seed(3)
            out[above, srt] = fa.flood_complex(tp, lm, method="bvh", **kw)
    assert np.array_equal(lms[3], lms[8])
    assert np.array_equal(lms[3], P[fo.exact_fps(P, 28, 0)])
    assert out[3, True] == out[8, True] == out[3, False] == out[8, False]


@pytest.mark.parametrize("refresh", [1, 4, 1000])
def test_fused_sorted_sweep_equals_unfused(refresh, dev, monkeypatch):
    """The sorted sweep that delivers the face maxima itself and drops every sample that cannot raise one (default
    above 3D) against the sorted sweep that stores all per-sample minima and takes the maxima afterwards: grid and
    random weights, 4-D and 6-D, however often the running maxima are re-read."""
    lib = _native.load()
    monkeypatch.setattr(core, "BVH_SORTED_MIN_SAMPLES", 0)
    rng = np.random.default_rng(5)
    assert lib.flooder_set_option(b"sorted_refresh", refresh) == 0
    try:
        for dim, n, k, kw in ((6, 40_000, 24, dict(max_dimension=2, points_per_edge=6)),
                              (4, 30_000, 30, dict(max_dimension=3, points_per_edge=4)),
                              (5, 20_000, 20, dict(max_dimension=2, num_rand=40))):
            P = rng.normal(size=(n, dim)).astype(np.float32)
            L = P[fo.exact_fps(P, k, 0)]
            tp, tl = torch.as_tensor(P, device=dev), torch.as_tensor(L, device=dev)
            out = {}
            for fused in (True, False):
                monkeypatch.setattr(core, "SORTED_FUSED_FACES", fused)
                torch.manual_seed(2)
                out[fused] = fa.flood_complex(tp, tl, method="bvh", **kw)
            assert out[True] == out[False], (dim, kw)
    finally:
        assert lib.flooder_set_option(b"sorted_refresh", 4) == 0


def test_sorted_sample_sweep_random_dimensions(dev, monkeypatch):
    """Dimensions 4, 5, 7 and 8 (one key width each), ragged sample counts (R not a multiple of 64, last tile partly
    dead), duplicated points: sorted-sample sweep == per-simplex tree sweep == kd-tree."""
    from scipy.spatial import cKDTree
    monkeypatch.setattr(core, "BVH_SORTED_MIN_SAMPLES", 0)
    rng = np.random.default_rng(11)
    for dim, n, k, ppe in ((4, 30_000, 40, 5), (5, 20_000, 30, 4), (7, 8_000, 14, 3), (8, 6_000, 12, 3)):
        P = rng.normal(size=(n, dim)).astype(np.float32)
        P[n // 2: n // 2 + 500] = P[:500]                 # exact duplicates
        L = P[fo.exact_fps(P, k, 0)]
        tp, tl = torch.as_tensor(P, device=dev), torch.as_tensor(L, device=dev)
        out = {}
        for mode in (True, False):
            monkeypatch.setattr(core, "BVH_SORTED_SAMPLES", mode)
            out[mode] = fa.flood_complex(tp, tl, max_dimension=2, points_per_edge=ppe, return_simplex_tree=True)
        a, b = out[True], out[False]
        for d in (1, 2):
            assert np.array_equal(a.filtrations_of_dimension(d), b.filtrations_of_dimension(d)), (dim, d)
        assert_tree_matches_kdtree(a, P, L, ppe, 2, f"sorted sweep dim {dim}")


def test_index_from_host_streams_the_cloud_and_index_kwarg_reuses_it(dev):
    """BASELINE cfg 5's "chunked point streaming from host pinned memory" (``core.index_from_host``): the index built
    from chunked pinned-memory copies equals the index of the resident cloud bit for bit, and
    ``flood_complex(..., index=...)`` returns what a call without it returns (3D: cell sweep; 6D: tree sweep)."""
    rng = np.random.default_rng(3)
    for dim, n, k, kw in ((3, 300_000, 120, dict(points_per_edge=12)), (6, 50_000, 20, dict(max_dimension=2, points_per_edge=5))):
        P = rng.normal(size=(n, dim)).astype(np.float32)
        L = P[fo.exact_fps(P, k, 0)]
        tp, tl = torch.as_tensor(P, device=dev), torch.as_tensor(L, device=dev)
        idx_s, raw = core.index_from_host(torch.as_tensor(P), dev, chunk_rows=70_001)   # ragged last chunk
        idx_r = core.PointIndex(tp)
        assert torch.equal(raw, tp)
        for off in (0, 8):   # (the other words of the 16-float box record are never written)
            assert torch.equal(idx_s.box[off:off + dim], idx_r.box[off:off + dim])
        assert torch.equal(idx_s.pts, idx_r.pts) and torch.equal(idx_s.order32, idx_r.order32)
        assert torch.equal(idx_s.nodes, idx_r.nodes)
        assert core.h2d_ms_of(idx_s) > 0.0
        a = fa.flood_complex(raw, tl, index=idx_s, **kw)
        b = fa.flood_complex(tp, tl, **kw)
        assert a == b
        with pytest.raises(ValueError):
            fa.flood_complex(tp[:1000], tl, index=idx_s, **kw)
    got = fa.flood_complex(raw, 20, index=idx_s, max_dimension=1, points_per_edge=4)     # landmarks by count + index
    assert got == fa.flood_complex(tp, 20, max_dimension=1, points_per_edge=4)


def test_dense_chunk_tile_launch_changes_nothing(dev):
    """Option "cell_tiles": chunks whose kept points overflow the LDS stage hand their tiles of 64 samples to a third
    launch (one sample per lane) instead of the exhaustive loop.  Off by default (slower); bit-identical when on: a
    surface cloud (most chunks near the sheet overflow) and a dense Gaussian core."""
    lib = _native.load()
    for pts, n_l, ppe in ((fo.noisy_torus(400_000, seed=3), 150, 30),
                          (np.random.default_rng(4).normal(size=(500_000, 3)).astype(np.float32) * 0.05, 40, 30)):
        lms = pts[fo.exact_fps(pts, n_l, 0)]
        tp, tl = torch.as_tensor(pts, device=dev), torch.as_tensor(lms, device=dev)
        off = fa.flood_complex(tp, tl, points_per_edge=ppe)
        try:
            assert lib.flooder_set_option(b"cell_tiles", 1) == 0
            on = fa.flood_complex(tp, tl, points_per_edge=ppe)
        finally:
            lib.flooder_set_option(b"cell_tiles", 0)
        assert on == off


def test_cell_sweep_queue_and_pass_options_change_nothing(dev):
    """Round-3 scheduling switches of the cell sweep: the single pass over the candidates of a dense chunk
    ("cell_one_pass": off / default / always, i.e. even where the chunk is given up afterwards), the weight-class
    order of the simplex lists, the deferred chunks ahead of the heavy ones, launch sizing of a short queue.  They
    change the order and the amount of work, never a value: a surface cloud (overflowing chunks), a dense Gaussian
    core (exhaustive chunks) and a cloud large enough for runs of four."""
    lib = _native.load()
    cases = ((fo.noisy_torus(400_000, seed=3), 150, 30),
             (np.random.default_rng(4).normal(size=(500_000, 3)).astype(np.float32) * 0.05, 40, 30),
             (np.random.default_rng(6).normal(size=(600_000, 3)).astype(np.float32), 700, 30))
    for pts, n_l, ppe in cases:
        tp = torch.as_tensor(pts, device=dev)
        tl = fa.generate_landmarks(tp, n_l, start_idx=0)
        ref = fa.flood_complex(tp, tl, points_per_edge=ppe)
        for opts in ({b"cell_one_pass": 0}, {b"cell_one_pass": 100000}, {b"cell_weight_classes": 0},
                     {b"cell_listed_first": 0}, {b"cell_chunks_per_block": 48, b"cell_min_grid": 64},
                     {b"cell_super_min_chunks": 0}, {b"cell_drop": 0}, {b"cell_chunk_major": 0},
                     {b"cell_drop": 0, b"cell_chunk_major": 0}):
            try:
                for k, v in opts.items():
                    assert lib.flooder_set_option(k, v) == 0
                got = fa.flood_complex(tp, tl, points_per_edge=ppe)
            finally:
                for k, v in ((b"cell_one_pass", 125), (b"cell_weight_classes", 1), (b"cell_listed_first", 1),
                             (b"cell_chunks_per_block", 12), (b"cell_min_grid", 384), (b"cell_super_min_chunks", 49152),
                             (b"cell_drop", 1), (b"cell_chunk_major", 1)):
                    lib.flooder_set_option(k, v)
            assert got == ref, opts


def test_many_simplices_short_queue_cell_equals_tree(dev):
    """A short queue (few chunks: one per simplex here) of MANY simplices: the weight-class order of the queue is
    built from global memory instead of the LDS stage (more than 7680 simplices), and every simplex must still be
    swept exactly once: cell sweep == tree sweep bit for bit, and a sample of the triangles against the kd-tree."""
    rng = np.random.default_rng(12)
    pts = rng.normal(size=(200_000, 2)).astype(np.float32)
    tp = torch.as_tensor(pts, device=dev)
    tl = fa.generate_landmarks(tp, 6000, start_idx=0)
    a = fa.flood_complex(tp, tl, points_per_edge=4, method="cell", return_simplex_tree=True)
    b = fa.flood_complex(tp, tl, points_per_edge=4, method="bvh", return_simplex_tree=True)
    n_tri = len(a.simplices_of_dimension(2))
    assert n_tri > 7680
    for d in range(3):
        assert np.array_equal(a.filtrations_of_dimension(d), b.filtrations_of_dimension(d)), d
    pick = np.sort(rng.choice(n_tri, size=600, replace=False))
    assert_tree_matches_kdtree(a, pts, tl.cpu().numpy(), 4, 2, "many simplices, short queue", pick_top=pick, lower=False)


def test_landmarks_outside_cloud_match_cpu_path(dev):
    """Landmarks that are NOT points of the cloud: the culled sweep still returns the exact value of
    the reference CPU path (the reference's own GPU path is only a bound there)."""
    rng = np.random.default_rng(5)
    pts = rng.normal(size=(20000, 3)).astype(np.float32)
    lms = rng.normal(size=(60, 3)).astype(np.float32) * 0.8
    ref = fo.flood_complex_oracle(pts, lms, points_per_edge=9)
    fc = fa.flood_complex(torch.as_tensor(pts, device=dev), torch.as_tensor(lms, device=dev), points_per_edge=9)
    keys = sorted(ref)
    assert_close_filtration(dict_values(fc, keys), dict_values(ref, keys), pts, "off-cloud landmarks")


def test_medium_cloud_against_kdtree_oracle(dev):
    """100 k torus / 300 landmarks: HIP path vs the oracle (reference CPU algorithm) on the same input."""
    pts = fo.noisy_torus(100_000, seed=42)
    idx = fo.exact_fps(pts, 300, 0)
    lms = pts[idx]
    ref = fo.flood_complex_oracle(pts, lms, points_per_edge=12)
    fc = fa.flood_complex(torch.as_tensor(pts, device=dev), torch.as_tensor(lms, device=dev), points_per_edge=12)
    keys = sorted(ref)
    assert set(keys) == set(fc)
    assert_close_filtration(dict_values(fc, keys), dict_values(ref, keys), pts, "torus100k")
    # vertices are witnesses of themselves: exactly zero
    assert all(fc[(i,)] == 0.0 for i in range(300))


def test_random_mode_and_landmark_count_gpu(dev):
    pts = fo.noisy_torus(10_000, seed=1)
    tp = torch.as_tensor(pts, device=dev)
    torch.manual_seed(5)
    fc = fa.flood_complex(tp, 200, points_per_edge=None, num_rand=512)
    lms = pts[fo.exact_fps(pts, 200, 0)]
    torch.manual_seed(5)
    ref = fo.flood_complex_oracle(pts, lms, points_per_edge=None, num_rand=512)
    keys = sorted(ref)
    assert set(keys) == set(fc)
    assert_close_filtration(dict_values(fc, keys), dict_values(ref, keys), pts, "rand512")


def test_fps_matches_exact_fps(dev):
    pts = fo.noisy_torus(50_000, seed=3)
    ref = fo.exact_fps(pts, 256, 17)
    got = core.fps_indices(torch.as_tensor(pts, device=dev), 256, 17).cpu().numpy()
    assert np.array_equal(got, ref)
    L = fa.generate_landmarks(torch.as_tensor(pts, device=dev), 64, start_idx=0)
    assert L.shape == (64, 3) and L.dtype == torch.float32 and L.is_cuda


def test_landmarks_clamped_to_point_count(dev):
    pts = fo.noisy_torus(500, seed=9)
    tp = torch.as_tensor(pts, device=dev)
    fc = fa.flood_complex(tp, 2000, points_per_edge=6)  # more landmarks than points (reference test_triton)
    ref = fo.flood_complex_oracle(pts, pts[fo.exact_fps(pts, 500, 0)], points_per_edge=6)
    keys = sorted(ref)
    assert set(keys) == set(fc)
    assert_close_filtration(dict_values(fc, keys), dict_values(ref, keys), pts, "clamp")


def test_workspace_grouping_is_invisible(dev, monkeypatch):
    z, kw, keys = load_e2e("cheese3d_grid")
    pts, lms = torch.as_tensor(z["points"], device=dev), torch.as_tensor(z["landmarks"], device=dev)
    a = fa.flood_complex(pts, lms, method="ball", **kw)
    monkeypatch.setattr(core, "CAND_WORKSPACE_BYTES", 1)  # floor of 2^20 rows per group
    b = fa.flood_complex(pts, lms, batch_size=7, method="ball", **kw)
    assert a == b


@pytest.mark.parametrize("name", ["torus3d_grid", "eight2d_rand", "gauss6d_maxdim2", "cheese3d_grid"])
def test_float64_input_gpu(dev, name):
    """float64 tensors run the float64 device kernels (the reference instantiates its Triton kernels with
    DTYPE = fp64): values match the reference's float64 CPU result to double precision, not just to float32."""
    z, kw, keys = load_e2e(name)
    torch.manual_seed(int(z["weight_seed"]))
    with pytest.warns(RuntimeWarning):
        fc = fa.flood_complex(torch.as_tensor(z["points"], device=dev).double(),
                              torch.as_tensor(z["landmarks"], device=dev).double(), **kw)
    got, ref = dict_values(fc, keys), z["filtration_f64"]
    if kw.get("num_rand"):
        # random weights go through a float32 log on the HOST (core.py:425): its last bit depends on the host's
        # vector unit, so the float64 golden of another machine is only float32-accurate here - the reference CPU
        # algorithm (kd-tree, float64) is re-run on this host with the same seed instead
        assert np.abs(got - ref).max() < 1e-6
        torch.manual_seed(int(z["weight_seed"]))
        with pytest.warns(RuntimeWarning):
            cpu = fa.flood_complex(torch.as_tensor(z["points"]).double(), torch.as_tensor(z["landmarks"]).double(), **kw)
        ref = dict_values(cpu, keys)
    assert set(keys) == set(fc)
    scale = float(np.abs(z["points"]).max())
    assert np.abs(got - ref).max() <= 1e-12 * scale + 1e-11 * np.abs(ref).max(), np.abs(got - ref).max()


def test_float32_and_float64_device_results_agree(dev):
    """The reference's precision test (tests/test_flooder.py:214-246): float32 and float64 runs within 3e-6."""
    pts = fo.noisy_torus(20_000, seed=42)
    lms = pts[fo.exact_fps(pts, 300, 0)]
    for kw in (dict(points_per_edge=20), dict(points_per_edge=None, num_rand=512)):
        torch.manual_seed(42)
        a = fa.flood_complex(torch.as_tensor(pts, device=dev), torch.as_tensor(lms, device=dev), **kw)
        torch.manual_seed(42)
        with pytest.warns(RuntimeWarning):
            b = fa.flood_complex(torch.as_tensor(pts, device=dev).double(), torch.as_tensor(lms, device=dev).double(), **kw)
        assert set(a) == set(b)
        assert max(abs(a[k] - b[k]) for k in a) < 3e-6


def test_shard_min_reduce_equals_unsharded(dev):
    """Multi-GPU rule on one GPU: min over shards of the per-sample buffer == unsharded, bit for bit."""
    pts = fo.noisy_torus(60_000, seed=11)
    lms = pts[fo.exact_fps(pts, 150, 0)]
    tp, tl = torch.as_tensor(pts, device=dev), torch.as_tensor(lms, device=dev)
    full = fa.flood_complex(tp, tl, points_per_edge=10)
    axis = int(np.argmax(pts.max(0) - pts.min(0)))
    captured = []
    for r in range(3):
        fa.flood_complex(tp[r::3].contiguous(), tl, points_per_edge=10, sort_axis=axis,
                         reduce_hook=lambda buf, c=captured: c.append(buf.clone()))
    merged = torch.minimum(torch.minimum(captured[0], captured[1]), captured[2])
    out = fa.flood_complex(tp[0::3].contiguous(), tl, points_per_edge=10, sort_axis=axis,
                           reduce_hook=lambda buf: buf.copy_(merged))
    assert out == full


def test_simplex_shards_min_reduce_equals_unsharded(dev):
    """Simplex sharding on one GPU: ranks 0..2 of 3 each sweep every third simplex; the element-wise MIN
    of their (S, F) face buffers is the unsharded result, bit for bit."""
    pts = fo.noisy_torus(60_000, seed=11)
    lms = pts[fo.exact_fps(pts, 150, 0)]
    tp, tl = torch.as_tensor(pts, device=dev), torch.as_tensor(lms, device=dev)
    full = fa.flood_complex(tp, tl, points_per_edge=10)
    bufs = []
    for r in range(3):
        fa.flood_complex(tp, tl, points_per_edge=10, simplex_shard=(r, 3),
                         face_reduce_hook=lambda buf, c=bufs: c.append(buf.clone()))
    merged = torch.minimum(torch.minimum(bufs[0], bufs[1]), bufs[2])
    assert torch.isfinite(merged).all()
    out = fa.flood_complex(tp, tl, points_per_edge=10, simplex_shard=(0, 3),
                           face_reduce_hook=lambda buf: buf.copy_(merged))
    assert out == full


def test_persistence_intervals_match_oracle(dev):
    """Persistence intervals of the GPU result vs of the oracle's filtration values (same Z/2 reduction):
    every bar longer than 1e-4 agrees within the filtration tolerance."""
    from flooder_amd.simplex_tree import SimplexTree
    pts = fo.noisy_torus(100_000, seed=21)
    lms = pts[fo.exact_fps(pts, 400, 0)]
    ref = fo.flood_complex_oracle(pts, lms, points_per_edge=8)
    st = fa.flood_complex(torch.as_tensor(pts, device=dev), torch.as_tensor(lms, device=dev), points_per_edge=8,
                          return_simplex_tree=True)
    by_len = {}
    for k, v in ref.items():
        by_len.setdefault(len(k), []).append((k, v))
    rt = SimplexTree.from_arrays([np.array([k for k, _ in by_len[L]]) for L in sorted(by_len)])
    for L in sorted(by_len):
        rt.assign_filtration_bulk(np.array([k for k, _ in by_len[L]]), np.array([v for _, v in by_len[L]]))
    st.compute_persistence()
    rt.compute_persistence()
    for d in range(3):
        a = st.persistence_intervals_in_dimension(d)
        b = rt.persistence_intervals_in_dimension(d)
        a = a[(a[:, 1] - a[:, 0]) > 1e-4]
        b = b[(b[:, 1] - b[:, 0]) > 1e-4]
        assert a.shape == b.shape, (d, a.shape, b.shape)
        assert np.allclose(a, b, rtol=1e-5, atol=2e-6)
    h1 = st.persistence_intervals_in_dimension(1)
    assert ((h1[:, 1] - h1[:, 0]) > 0.4).sum() == 2  # the torus' two loops


def test_edge_cases_gpu(dev):
    torch.manual_seed(0)
    # 1-D ambient space
    x = torch.rand(5000, 1)
    lms = x[fo.exact_fps(x.numpy(), 20, 0)]
    fc = fa.flood_complex(x.to(dev), lms.to(dev), points_per_edge=9)
    ref = fo.flood_complex_oracle(x.numpy(), lms.numpy(), points_per_edge=9)
    keys = sorted(ref)
    assert set(fc) == set(keys)
    assert_close_filtration(dict_values(fc, keys), dict_values(ref, keys), x.numpy(), "1d")
    # tiny cloud, fewer points than one leaf; single simplex
    pts = torch.rand(7, 3)
    fc = fa.flood_complex(pts.to(dev), pts[:4].clone().to(dev), points_per_edge=4)
    ref = fo.flood_complex_oracle(pts.numpy(), pts[:4].numpy(), points_per_edge=4)
    keys = sorted(ref)
    assert_close_filtration(dict_values(fc, keys), dict_values(ref, keys), pts.numpy(), "tiny")
    # duplicated points and a large coordinate offset
    p = torch.rand(4000, 3) * 0.01 + 500.0
    p[2000:] = p[:2000]
    lm = p[fo.exact_fps(p.numpy(), 30, 0)]
    fc = fa.flood_complex(p.to(dev), lm.to(dev), points_per_edge=5)
    ref = fo.flood_complex_oracle(p.numpy(), lm.numpy(), points_per_edge=5)
    keys = sorted(ref)
    assert_close_filtration(dict_values(fc, keys), dict_values(ref, keys), p.numpy(), "offset+duplicates")
    with pytest.raises(RuntimeError):
        fa.flood_complex(torch.zeros((0, 3), device=dev), 4)


def test_fuzz_methods_agree(dev):
    """Random small configurations: cell == bvh == ball bit for bit, and all within tolerance of the oracle."""
    rng = np.random.default_rng(123)
    for case in range(12):
        dim = int(rng.choice([2, 3, 3]))
        n = int(rng.choice([80, 700, 6000]))
        pts = rng.normal(size=(n, dim)).astype(np.float32)
        if case % 3 == 0:
            pts = (rng.normal(size=(n, dim)) * 0.03 + rng.normal(size=(6, dim))[rng.integers(0, 6, n)]).astype(np.float32)
        n_l = int(min(n, rng.choice([dim + 2, 25, 90])))
        lms = pts[fo.exact_fps(pts, n_l, 0)]
        kw = dict(points_per_edge=int(rng.choice([2, 4, 9]))) if case % 4 else dict(points_per_edge=None, num_rand=150)
        tp, tl = torch.as_tensor(pts, device=dev), torch.as_tensor(lms, device=dev)
        res = {}
        for m in ("cell", "bvh", "ball"):
            torch.manual_seed(case)
            res[m] = fa.flood_complex(tp, tl, method=m, **kw)
        torch.manual_seed(case)
        ref = fo.flood_complex_oracle(pts, lms, **kw)
        assert res["cell"] == res["ball"] and res["bvh"] == res["ball"], f"case {case}"
        keys = sorted(ref)
        assert_close_filtration(dict_values(res["cell"], keys), dict_values(ref, keys), pts, f"case {case}")


def test_full_size_properties_1m_gaussian(dev):
    """BASELINE cfg 2 (1 M Gaussian 3D, 1 k landmarks, ppe 30): monotone filtration, exact-zero vertices, and EVERY
    tetrahedron, triangle and edge against the kd-tree oracle over all points (the reference's own full-size check
    is the cross-path agreement of ``tests/test_flooder.py:119-157``; 30 M queries take seconds on the host's cores)."""
    torch.manual_seed(42)
    pts = torch.randn(1_000_000, 3)
    tp = pts.to(dev)
    lms = fa.generate_landmarks(tp, 1000, start_idx=0)
    st = fa.flood_complex(tp, lms, return_simplex_tree=True)
    assert all(st.filtration([i]) == 0.0 for i in range(1000))
    vals = st.filtrations_of_dimension(3)
    assert np.isfinite(vals).all()
    for d in (1, 2, 3):  # faces never above cofaces
        rows = st.simplices_of_dimension(d)
        own = st.filtrations_of_dimension(d)
        for j in range(d + 1):
            idx = st._locate(d - 1, np.delete(rows, j, axis=1))
            assert (st.filtrations_of_dimension(d - 1)[idx] <= own).all()
    n = assert_tree_matches_kdtree(st, pts.numpy(), lms.cpu().numpy(), 30, 3, "cfg2 1M gaussian", strict=True)
    assert n == sum(len(st.simplices_of_dimension(d)) for d in (1, 2, 3)) and n > 25_000


@pytest.mark.parametrize("cloud", ["torus", "cheese"])
def test_full_size_far_field_clouds_match_oracle_sample(dev, cloud):
    """BASELINE cfg 3 (1 M noisy torus) and a 2 M-point slice of cfg 5 (swiss cheese): samples inside the tube /
    the voids lie far from every point, so most of their tiles take the exact tree finish.  EVERY tetrahedron,
    triangle and edge is checked against the kd-tree oracle, the ones crossing the empty regions included."""
    if cloud == "torus":
        pts = fa.generate_noisy_torus_points_3d(1_000_000, seed=42)
        n_lms = 1000
    else:
        pts = fa.generate_swiss_cheese_points(2_000_000, k=6, seed=42)[0]
        n_lms = 1500
    tp = pts.to(dev)
    lms = fa.generate_landmarks(tp, n_lms, start_idx=0)
    st = fa.flood_complex(tp, lms, return_simplex_tree=True)
    vals = st.filtrations_of_dimension(3)
    assert np.isfinite(vals).all()
    assert_tree_matches_kdtree(st, pts.numpy(), lms.cpu().numpy(), 30, 3, f"{cloud} full check", strict=True)
    big = np.argsort(-vals)[:15]                      # the farthest-reaching tetrahedra: tube interior / voids
    assert float(vals[big].min()) > 5 * float(np.median(vals))    # the complex really contains far-field tetrahedra


def test_degenerate_density_clouds_gpu(dev):
    """Clouds that stress the cell sweep's capacity paths: a planar cloud in 3D (zero-volume chunk boxes), and two
    tight clusters joined by a sparse bridge (chunks that overflow the LDS stage next to chunks with an empty box)."""
    from scipy.spatial import KDTree
    rng = np.random.default_rng(5)
    # planar cloud embedded in 3D, landmarks lifted slightly off the plane so that the triangulation is 3D
    xy = rng.random((60_000, 2)).astype(np.float32)
    plane = np.concatenate([xy, np.zeros((xy.shape[0], 1), np.float32)], axis=1)
    lift = plane[fo.exact_fps(plane, 60, 0)].copy()
    lift[:, 2] = (rng.random(60) * 0.05).astype(np.float32)
    # clusters + bridge
    a = rng.normal(0.0, 0.01, (150_000, 3))
    b = rng.normal(0.0, 0.01, (150_000, 3)) + np.array([1.0, 0.2, -0.3])
    bridge = np.linspace(0, 1, 400)[:, None] * np.array([1.0, 0.2, -0.3]) + rng.normal(0, 0.002, (400, 3))
    clusters = np.concatenate([a, b, bridge]).astype(np.float32)
    for name, P, L in (("planar", plane, lift), ("clusters", clusters, clusters[fo.exact_fps(clusters, 120, 0)])):
        st = fa.flood_complex(torch.from_numpy(P).to(dev), torch.from_numpy(L).to(dev), return_simplex_tree=True)
        tets = st.simplices_of_dimension(3)
        vals = st.filtrations_of_dimension(3)
        assert np.isfinite(vals).all()
        w, _, _ = fo.generate_grid(30, 3)
        pick = rng.choice(len(tets), size=min(40, len(tets)), replace=False)
        dist, _ = KDTree(P).query(np.matmul(w[None], L[tets[pick]].astype(np.float64)).astype(np.float32))
        # tree values are monotonised: a tetrahedron's value is at least its own sweep value
        own = dist.max(axis=1)
        assert (vals[pick] >= own * (1 - 1e-5) - 1e-6).all(), name
        # and equals it whenever no face exceeds it (always true here: faces are subsets of the samples)
        assert_close_filtration(vals[pick], own, P, f"{name} tetrahedra sample")


def test_needle_and_sliver_simplices_cell_equals_tree(dev, monkeypatch):
    """Ill-conditioned simplices (near-collinear landmark triples, near-coplanar quadruples), also far from the
    coordinate origin: the cell sweep's face-plane filter must never drop a true nearest neighbour, i.e. the cell
    sweep equals the tree sweep (which has no such filter) bit for bit, and both match the kd-tree oracle.
    (The landmarks are near-degenerate ON PURPOSE - twelve collinear, twenty-five coplanar points - so their Delaunay
    triangulation hangs on round-off: the oracle triangulates with Qhull, and so does the product here; the exact
    native routine decides the ties differently, tests/test_delaunay.py.)"""
    from flooder_amd import simplex_tree as stm

    monkeypatch.setattr(stm, "NATIVE_DELAUNAY", False)
    rng = np.random.default_rng(77)
    for case, (offset, eps) in enumerate([(0.0, 1e-4), (0.0, 1e-6), (300.0, 1e-5), (0.0, 0.0)]):
        cloud = rng.normal(size=(30_000, 3)).astype(np.float64)
        t = np.linspace(-1.5, 1.5, 12)
        line = np.stack([t, 0.3 * t, -0.2 * t], axis=1) + rng.normal(size=(12, 3)) * eps      # needles
        u, v = np.meshgrid(np.linspace(-1, 1, 5), np.linspace(-1, 1, 5))
        plane = np.stack([u.ravel(), v.ravel(), 0.5 + 0.1 * u.ravel()], axis=1) + rng.normal(size=(25, 3)) * eps  # slivers
        extra = cloud[fo.exact_fps(cloud.astype(np.float32), 20, 0)]
        lms = np.concatenate([line, plane, extra]) + offset
        pts = np.concatenate([cloud + offset, lms]).astype(np.float32)   # landmarks are points of the cloud
        lms = pts[-lms.shape[0]:]
        tp, tl = torch.as_tensor(pts, device=dev), torch.as_tensor(lms, device=dev)
        a = fa.flood_complex(tp, tl, points_per_edge=12, method="cell")
        b = fa.flood_complex(tp, tl, points_per_edge=12, method="bvh")
        assert a == b, f"case {case}"
        ref = fo.flood_complex_oracle(pts, lms, points_per_edge=12)
        keys = sorted(ref)
        assert set(keys) == set(a)
        assert_close_filtration(dict_values(a, keys), dict_values(ref, keys), pts, f"needles case {case}")


@pytest.mark.parametrize("cloud", ["torus", "gauss", "cheese", "eight2d"])
def test_fused_face_maxima_equal_unfused(dev, cloud, monkeypatch):
    """The cell sweep with the per-face maxima folded in (settled samples delivered by atomic max, the finish dropping
    every sample that cannot raise a face maximum) against sweep -> finish -> face-max over the full (S, R) buffer:
    identical dictionaries, grid and random mode."""
    if cloud == "torus":
        pts = fo.noisy_torus(300_000, seed=7)
        n_l, dim = 250, 3
    elif cloud == "gauss":
        pts = np.random.default_rng(7).normal(size=(300_000, 3)).astype(np.float32)
        n_l, dim = 250, 3
    elif cloud == "cheese":
        pts = fa.generate_swiss_cheese_points(400_000, k=6, seed=7)[0].numpy()
        n_l, dim = 300, 3
    else:
        pts = fa.generate_figure_eight_points_2d(200_000, noise_std=0.01, seed=7).numpy().astype(np.float32)
        n_l, dim = 200, 2
    tp = torch.as_tensor(pts, device=dev)
    tl = fa.generate_landmarks(tp, n_l, start_idx=0)
    for kw in (dict(points_per_edge=20), dict(points_per_edge=None, num_rand=700)):
        res = {}
        for fused in (True, False):
            monkeypatch.setattr(core, "FUSED_FACES", fused)
            torch.manual_seed(11)
            res[fused] = fa.flood_complex(tp, tl, method="cell", **kw)
        assert res[True] == res[False], (cloud, kw)


@pytest.mark.parametrize("cloud", ["gauss", "torus"])
def test_long_queue_lists_from_one_launch_equal_the_two_launches(dev, cloud):
    """Light / heavy simplex lists of a LONG queue (>= 49152 chunks): one class_order_kernel launch against the split +
    reorder pair it replaces - same list lengths, same face values bit for bit (a simplex lost or listed twice would
    show), with and without the witness sweep in front (which marks the simplices it handled)."""
    lib = _native.load()
    if cloud == "gauss":
        pts = np.random.default_rng(5).normal(size=(400_000, 3)).astype(np.float32)
    else:
        pts = fo.noisy_torus(400_000, seed=5)
    tp = torch.as_tensor(pts, device=dev)
    lms = fa.generate_landmarks(tp, 600, start_idx=0)
    stree, simplices = core._build_complex(lms, 3)
    verts = lms[torch.as_tensor(simplices[3], device=dev)].contiguous()
    weights, vertex_idxs, face_idxs = core.generate_grid(30, 3, dev, torch.float32)
    faces = core._FaceTable(face_idxs, weights.shape[0], dev)
    assert verts.shape[0] * ((weights.shape[0] + 255) // 256) >= 49152, "not a long queue"
    index = core.PointIndex(tp)
    keep = core.WIT_MIN_SIMPLICES, core.CELL_WITNESS
    try:
        core.WIT_MIN_SIMPLICES = 0
        for witness in (True, False):
            core.CELL_WITNESS = witness
            got = {}
            for launches in (1, 2):
                assert lib.flooder_set_option(b"cell_split_launches", launches) == 0
                st = torch.zeros(40, dtype=torch.int64, device=dev)
                out, _ = core._sweep_dimension_cell(index, verts, weights, faces, None, stats=st)
                torch.cuda.synchronize()
                got[launches] = (out.clone(), core.LAST_STATS.light_heavy)
            assert got[1][1] == got[2][1], (cloud, witness, got[1][1], got[2][1])
            assert sum(got[1][1]) > 0
            assert torch.equal(got[1][0].view(torch.int32), got[2][0].view(torch.int32)), (cloud, witness)
    finally:
        core.WIT_MIN_SIMPLICES, core.CELL_WITNESS = keep
        assert lib.flooder_set_option(b"cell_split_launches", 1) == 0


@pytest.mark.parametrize("cloud", ["gauss3d", "torus3d", "eight2d"])
def test_launch_lean_paths_change_nothing(dev, cloud, monkeypatch):
    """Round 6's launch-lean forms against the ones they replace: the index with the radix sort on caller-zeroed state
    (``core.SORT_STATE_BY_CALLER``: same permutation, rows, tree, density grid) and the sweep prepared by ONE launch
    (``core.SWEEP_PREPARE_FUSED``: zero fill + simplex weights + plane rows; same weights, same face values bit for bit)."""
    lib = _native.load()
    if cloud == "gauss3d":
        pts = np.random.default_rng(11).normal(size=(300_000, 3)).astype(np.float32)
        n_l, dim = 400, 3
    elif cloud == "torus3d":
        pts = fo.noisy_torus(250_000, seed=11)
        n_l, dim = 300, 3
    else:
        pts = fa.generate_figure_eight_points_2d(150_000, noise_std=0.01, seed=11).numpy().astype(np.float32)
        n_l, dim = 200, 2
    tp = torch.as_tensor(pts, device=dev)
    idx = {}
    for own in (True, False):
        monkeypatch.setattr(core, "SORT_STATE_BY_CALLER", own)
        idx[own] = core.PointIndex(tp)
    torch.cuda.synchronize()
    for name in ("order32", "pts", "nodes", "dens"):
        a, b = getattr(idx[True], name), getattr(idx[False], name)
        assert torch.equal(a.view(torch.int32), b.view(torch.int32)), (cloud, name)
    lms = fa.generate_landmarks(tp, n_l, start_idx=0)
    stree, simplices = core._build_complex(lms, dim)
    verts = lms[torch.as_tensor(simplices[dim], device=dev)].contiguous()
    S = verts.shape[0]
    # the weights of the prepare launch = flooder_simplex_weight_f32's, its zero fill clears what it is given
    w_a = torch.empty(S, dtype=torch.float32, device=dev)
    w_b = torch.empty(S, dtype=torch.float32, device=dev)
    planes = torch.full((24 * S,), float("nan"), dtype=torch.float32, device=dev)
    junk = torch.full((5000,), 7, dtype=torch.int32, device=dev)
    index = idx[True]
    _native.check(lib.flooder_simplex_weight_f32(_native.ptr(index.nodes), index.n, dim, _native.ptr(verts), dim + 1, S,
                                                 _native.ptr(w_a), 0), "flooder_simplex_weight_f32")
    _native.check(lib.flooder_simplex_prepare_f32(_native.ptr(index.nodes), index.n, dim, _native.ptr(verts), dim + 1, S,
                                                  _native.ptr(w_b), _native.ptr(planes), _native.ptr(junk), junk.numel(), 0),
                  "flooder_simplex_prepare_f32")
    lib.flooder_simplex_planes_forget()   # (no sweep follows here)
    torch.cuda.synchronize()
    assert torch.equal(w_a.view(torch.int32), w_b.view(torch.int32))
    assert int(junk.abs().sum()) == 0 and not bool(torch.isnan(planes).any())
    res = {}
    for fused in (True, False):
        monkeypatch.setattr(core, "SWEEP_PREPARE_FUSED", fused)
        res[fused] = fa.flood_complex(tp, lms, method="cell", points_per_edge=20)
    assert res[True] == res[False], cloud


def test_simplex_weight_orders_dense_simplices_first(dev):
    """flooder_simplex_weight_f32: the estimate tracks the true number of points in each simplex's bounding box."""
    pts = np.random.default_rng(3).normal(size=(400_000, 3)).astype(np.float32)
    tp = torch.as_tensor(pts, device=dev)
    tl = fa.generate_landmarks(tp, 200, start_idx=0)
    _, simplices = core._build_complex(tl, 3)
    verts = tl[torch.as_tensor(simplices[3], device=dev)]
    index = core.PointIndex(tp)
    order = core.simplex_order(index, verts).cpu().numpy()
    assert sorted(order.tolist()) == list(range(verts.shape[0]))
    V = verts.cpu().numpy()
    lo, hi = V.min(axis=1), V.max(axis=1)
    true = np.array([((pts >= lo[i]) & (pts <= hi[i])).all(axis=1).sum() for i in range(V.shape[0])])
    top = set(np.argsort(-true)[: len(true) // 10].tolist())
    assert len(top & set(order[: len(true) // 5].tolist())) >= 0.8 * len(top)


@pytest.mark.parametrize("cloud", ["gauss3d", "eight2d", "torus"])
def test_runs_of_four_chunks_equal_chunk_by_chunk(dev, cloud, monkeypatch):
    """The two-launch cell sweep (runs of four chunks against one shared stage, then the deferred chunks) forced on
    for small inputs - every simplex 'sparse', no minimum queue length, several stage-fit limits - against the
    chunk-by-chunk sweep: identical dictionaries."""
    lib = _native.load()
    if cloud == "gauss3d":
        pts = np.random.default_rng(11).normal(size=(150_000, 3)).astype(np.float32)
        n_l = 120
    elif cloud == "eight2d":
        pts = fa.generate_figure_eight_points_2d(120_000, noise_std=0.02, seed=3).numpy().astype(np.float32)
        n_l = 150
    else:
        pts = fo.noisy_torus(200_000, seed=13)
        n_l = 150
    tp = torch.as_tensor(pts, device=dev)
    tl = fa.generate_landmarks(tp, n_l, start_idx=0)
    for kw in (dict(points_per_edge=30), dict(points_per_edge=None, num_rand=3000)):
        monkeypatch.setattr(core, "CELL_SUPER", False)
        torch.manual_seed(5)
        ref = fa.flood_complex(tp, tl, method="cell", **kw)
        monkeypatch.setattr(core, "CELL_SUPER", True)
        for weight, n0 in ((10 ** 9, 480), (2000, 480), (10 ** 9, 100)):
            try:
                for name, val in ((b"cell_super_min_chunks", 0), (b"cell_super_sparse", 10 ** 9),
                                  (b"cell_super_weight", weight), (b"cell_super_n0", n0)):
                    assert lib.flooder_set_option(name, val) == 0
                torch.manual_seed(5)
                got = fa.flood_complex(tp, tl, method="cell", **kw)
            finally:
                for name, val in ((b"cell_super_min_chunks", 49152), (b"cell_super_sparse", 600),
                                  (b"cell_super_weight", 2000), (b"cell_super_n0", 480)):
                    lib.flooder_set_option(name, val)
            assert got == ref, (cloud, kw, weight, n0)


@pytest.mark.parametrize("cloud", ["torus", "eight2d"])
def test_finish_hard_tiles_and_ordering_change_nothing(dev, cloud, monkeypatch):
    """The finish's schedule - flagged tiles by descending probe bound, tiles over the leaf budget handed to a
    workgroup of 16 waves that search interleaved shares of the tree and combine their minima in LDS, with or without
    the top pass (one sample per simplex first) - must not change a single bit of the face values: plain order / one wave per tile against several budgets and a hard list
    too short for its entries (those are finished by their producer)."""
    lib = _native.load()
    if cloud == "torus":
        pts = torch.as_tensor(fo.noisy_torus(300_000, seed=13), device=dev)
        d = 3
    else:
        pts = fa.generate_figure_eight_points_2d(200_000, noise_std=0.02, seed=3).to(dev)
        d = 2
    lms = fa.generate_landmarks(pts, 200, start_idx=0)
    _, simplices = core._build_complex(lms, d)
    verts = lms[torch.as_tensor(simplices[d], device=dev)].contiguous()
    weights, _, fi = core.generate_grid(30, d, dev, torch.float32)
    faces = core._FaceTable(fi, weights.shape[0], dev)
    index = core.PointIndex(pts)
    stats = torch.zeros(16, dtype=torch.int64, device=dev)

    def run(order, budget, cap, top=0):
        monkeypatch.setattr(core, "FINISH_HARD_CAP", cap)
        try:
            # (finish_budget_min: the floor under a short list's budget - 1 here, so that budget 1 means one leaf)
            for name, val in ((b"finish_order", order), (b"finish_budget", budget), (b"finish_top", top), (b"finish_budget_min", 1)):
                assert lib.flooder_set_option(name, val) == 0
            stats.zero_()
            out, _ = core._sweep_dimension_cell(index, verts, weights, faces, None, stats=stats)
            torch.cuda.synchronize()
            return out.cpu().numpy(), core.LAST_STATS.hard_entries
        finally:
            for name, val in ((b"finish_order", 1), (b"finish_budget", 14), (b"finish_top", 0), (b"finish_budget_min", 64)):
                lib.flooder_set_option(name, val)

    ref, hard = run(0, 0, 32768)
    assert hard == (0, 0)
    unfused_before = core.FUSED_FACES
    monkeypatch.setattr(core, "FUSED_FACES", False)
    plain, _ = core._sweep_dimension_cell(index, verts, weights, faces, None)
    monkeypatch.setattr(core, "FUSED_FACES", unfused_before)
    np.testing.assert_array_equal(ref, plain.cpu().numpy())
    seen = 0
    for order, budget, cap, top in ((1, 0, 32768, 0), (1, 1, 32768, 0), (0, 1, 32768, 0), (1, 4, 32768, 1), (1, 1, 8, 0),
                                    (1, 14, 32768, 0), (0, 0, 32768, 1), (1, 1, 32768, 1)):
        got, hard = run(order, budget, cap, top)
        np.testing.assert_array_equal(got, ref, err_msg=str((cloud, order, budget, cap, top)))
        seen += hard[1]
    assert seen > 0, "no round ever exceeded the budget: the hard-entry launches were not exercised"


@pytest.mark.parametrize("n,dim", [(1, 3), (17, 3), (1000, 2), (100_003, 3), (65_536, 3), (20_001, 6)])
def test_fused_index_rows_equal_gather_then_build(dev, n, dim):
    """flooder_index_rows_f32 (rows gathered into curve order, leaf boxes reduced from the rows in registers, inner
    levels) against the two separate entry points: identical rows and identical boxes on every level, padding included."""
    lib = _native.load()
    g = torch.Generator().manual_seed(n + dim)
    pts = torch.randn((n, dim), generator=g).to(dev)
    index = core.PointIndex(pts)
    st = _native.current_stream_ptr(dev)
    rows = torch.empty_like(index.pts)
    nodes = torch.full_like(index.nodes, float("nan"))
    _native.check(lib.flooder_gather_rows_f32(_native.ptr(pts), n, dim, dim, _native.ptr(index.order32),
                                              _native.ptr(rows), rows.shape[0], st), "flooder_gather_rows_f32")
    _native.check(lib.flooder_bvh_build_f32(_native.ptr(rows), n, dim, _native.ptr(nodes), st), "flooder_bvh_build_f32")
    torch.cuda.synchronize()
    assert torch.equal(rows.view(torch.int32), index.pts.view(torch.int32))
    assert torch.equal(nodes.view(torch.int32), index.nodes.view(torch.int32))


def test_index_sort_is_a_stable_sort(dev):
    """flooder_index_sort (radix sort over the used key bits; uint32 words for keys of at most 32 bits) against torch's
    stable sort: identical permutations, duplicates included."""
    lib = _native.load()
    g = torch.Generator().manual_seed(9)
    # (12288 = 3 x 512 x 8 and 1638400 = 200 x 1024 x 8: whole blocks only, in the small and the large block shape)
    for n, bits in ((1, 30), (63, 30), (5000, 8), (12_288, 24), (100_000, 30), (1_000_003, 30), (300_000, 17),
                    (200_000, 36), (1_638_400, 24), (4_000_000, 30)):
        keys = torch.randint(0, 1 << bits, (n,), generator=g, dtype=torch.int64)
        if n > 1000:
            keys[: n // 3] = keys[n // 3: 2 * (n // 3)]                     # plenty of duplicates
        want = torch.sort(keys, stable=True).indices.numpy()
        buf = torch.zeros(n, dtype=torch.int64, device=dev)
        if bits <= 32:
            buf.view(torch.int32)[:n] = keys.to(torch.int32).to(dev)       # narrow keys: n uint32 words
        else:
            buf.copy_(keys)
        order = torch.empty(n, dtype=torch.int32, device=dev)
        out = torch.empty(n, dtype=torch.int64, device=dev)
        nb = int(lib.flooder_index_sort_bytes(n))
        tmp = torch.empty(nb, dtype=torch.uint8, device=dev)
        _native.check(lib.flooder_index_sort(_native.ptr(buf), n, bits, _native.ptr(out), _native.ptr(order),
                                             _native.ptr(tmp), nb, 0), "flooder_index_sort")
        torch.cuda.synchronize()
        assert np.array_equal(order.cpu().numpy().astype(np.int64), want), (n, bits)
        # the same library kernels on state the caller zeroes (no fill launches): same permutation, same sorted codes,
        # three times over (stale state of an earlier call must not matter once it is zeroed again)
        words = int(lib.flooder_index_sort_state_words(n, bits))
        assert (words > 0) == (bits <= 32)
        for rep in range(4 if words else 0):   # (block shape by size, then each of the three forced)
            assert lib.flooder_set_option(b"sort_shape", rep) == 0
            state = torch.full((words,), -1 if rep else 7, dtype=torch.int32, device=dev)
            state.zero_()
            order2 = torch.full((n,), -1, dtype=torch.int32, device=dev)
            out2 = torch.full((n,), -1, dtype=torch.int64, device=dev)
            _native.check(lib.flooder_index_sort_zeroed(_native.ptr(buf), n, bits, _native.ptr(out2), _native.ptr(order2),
                                                        _native.ptr(tmp), nb, _native.ptr(state), 0),
                          "flooder_index_sort_zeroed")
            torch.cuda.synchronize()
            assert torch.equal(order2, order), (n, bits, rep)
            assert torch.equal(out2.view(torch.int32)[:n], out.view(torch.int32)[:n]), (n, bits, rep)
        assert lib.flooder_set_option(b"sort_shape", 0) == 0


def test_index_of_a_modified_cloud_is_refused(dev):
    """``flood_complex(index=...)``: an index built before the cloud tensor was written to in place is refused (same
    storage, other version counter); an untouched tensor passes."""
    g = torch.Generator().manual_seed(1)
    pts = torch.randn(50_000, 3, generator=g).to(dev)
    lms = fa.generate_landmarks(pts, 60, start_idx=0)
    idx = core.PointIndex(pts)
    a = fa.flood_complex(pts, lms, points_per_edge=6, index=idx)
    assert a == fa.flood_complex(pts, lms, points_per_edge=6)
    pts.mul_(1.5)
    with pytest.raises(ValueError, match="modified in place"):
        fa.flood_complex(pts, lms, points_per_edge=6, index=idx)


def test_the_index_of_generate_landmarks_is_reused_by_flood_complex(dev, monkeypatch):
    """``lms = generate_landmarks(points, n); flood_complex(points, lms)``: the second call takes the PointIndex the
    first one built from the SAME tensor object (identity + version counter; one entry) - on every rank of a sharded
    run the replicated index build is the part that does not divide.  Another tensor, an in-place write, an inference
    tensor or ``forget_index`` rebuild it; the values are the same either way."""
    g = torch.Generator().manual_seed(21)
    pts = torch.randn(200_000, 3, generator=g).to(dev)
    built = []
    orig = core.PointIndex.__init__

    def spy(self, *a, **k):
        built.append(1)
        return orig(self, *a, **k)

    monkeypatch.setattr(core.PointIndex, "__init__", spy)
    core.forget_index()
    # default: no state between calls (the reference keeps none) - the explicit handle is the way to share the index
    assert core.INDEX_CACHE is False
    lms, idx = fa.generate_landmarks(pts, 150, start_idx=0, return_index=True)
    assert len(built) == 1 and isinstance(idx, core.PointIndex)
    a0 = fa.flood_complex(pts, lms, points_per_edge=8, index=idx)
    assert len(built) == 1
    assert fa.flood_complex(pts, lms, points_per_edge=8) == a0 and len(built) == 2   # plain call: its own index
    assert fa.flood_complex(pts, 150, points_per_edge=8) == a0 and len(built) == 3   # integer landmarks: ONE index for both steps
    built.clear()
    monkeypatch.setattr(core, "INDEX_CACHE", True)                     # opt-in: remembered by tensor identity + version
    lms = fa.generate_landmarks(pts, 150, start_idx=0)
    assert len(built) == 1
    a = fa.flood_complex(pts, lms, points_per_edge=8)
    assert a == a0
    assert len(built) == 1, "flood_complex rebuilt the index generate_landmarks had just built"
    b = fa.flood_complex(pts.clone(), lms, points_per_edge=8)          # another tensor: its own index
    assert len(built) == 2 and a == b
    c = fa.flood_complex(pts, lms, points_per_edge=8)                  # (one entry: the clone's index replaced it)
    assert len(built) == 3 and a == c
    pts.add_(0.0)                                                      # in-place write: version counter moves
    fa.flood_complex(pts, lms, points_per_edge=8)
    assert len(built) == 4
    fa.flood_complex(pts, lms, points_per_edge=8)
    assert len(built) == 4
    core.forget_index()
    fa.flood_complex(pts, lms, points_per_edge=8)
    assert len(built) == 5
    with torch.inference_mode():                                       # no version counter: never remembered, never crashes
        pin = torch.randn(100_000, 3, generator=g).to(dev)
        lin = fa.generate_landmarks(pin, 80, start_idx=0)
        n0 = len(built)
        fa.flood_complex(pin, lin, points_per_edge=6)
        assert len(built) == n0 + 1
    import gc
    tmp = torch.randn(250_000, 3, generator=g).to(dev)                  # the entry dies with its tensor
    fa.generate_landmarks(tmp, 80, start_idx=0)
    assert core._LAST_INDEX[2] is not None and core._LAST_INDEX[0]() is tmp
    del tmp
    gc.collect()
    assert core._LAST_INDEX[2] is None
    monkeypatch.setattr(core, "INDEX_CACHE", False)
    core.forget_index()
    fa.flood_complex(pts, lms, points_per_edge=8)
    n1 = len(built)
    fa.flood_complex(pts, lms, points_per_edge=8)
    assert len(built) == n1 + 1


def test_cloud_kind_words_tell_a_surface_from_a_volume(dev):
    """The statistic behind the density grid (flooder_cloud_kind / the index build): share of the points in interior
    cells of the coarse grid - high for clouds that fill a volume, low for the torus surface; the cell sweep tries one
    cell size per chunk on the latter (option cell_surface_pct) and the values are the same bits either way."""
    lib = _native.load()
    torch.manual_seed(3)
    clouds = {"gauss": torch.randn(300_000, 3), "torus": fa.generate_noisy_torus_points_3d(300_000, seed=2),
              "cheese": fa.generate_swiss_cheese_points(300_000, k=6, seed=5)[0]}
    pct = {}
    for name, p in clouds.items():
        idx = core.PointIndex(p.to(dev))
        k = idx.dens[64 ** 3:64 ** 3 + 4].cpu().tolist()
        assert k[3] == 300_000
        pct[name] = 100.0 * k[2] / k[3]
        # the stand-alone entry point on a grid of its own gives the same words
        grid = torch.zeros(int(lib.flooder_density_grid_words(3)), dtype=torch.int32, device=dev)
        st = _native.current_stream_ptr(dev)
        _native.check(lib.flooder_density_grid_f32(_native.ptr(idx.nodes), idx.n, 3, _native.ptr(idx.box), _native.ptr(grid), st), "grid")
        _native.check(lib.flooder_cloud_kind(_native.ptr(grid), 3, st), "kind")
        assert grid[64 ** 3 + 2:64 ** 3 + 4].cpu().tolist() == k[2:]
    assert pct["gauss"] > 85 and pct["cheese"] > 85 and pct["torus"] < 35, pct
    tp = clouds["torus"].to(dev)
    lms = fa.generate_landmarks(tp, 300, start_idx=0)
    a = fa.flood_complex(tp, lms, points_per_edge=20)
    assert lib.flooder_set_option(b"cell_surface_pct", 0) == 0
    try:
        b = fa.flood_complex(tp, lms, points_per_edge=20)
    finally:
        assert lib.flooder_set_option(b"cell_surface_pct", 60) == 0
    assert a == b
