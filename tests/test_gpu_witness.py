"""Witness sweep (csrc/flood_wit.hip): face values must equal the cell sweep's and the tree sweep's bit for bit,
whatever it decides to handle, drop, settle itself or hand on.  Runs on a real MI355X only (-m gpu)."""
import numpy as np
import pytest
import torch

import flooder_amd as fa
from flooder_amd import _native, core
from oracle import flood_oracle as fo

pytestmark = pytest.mark.gpu

WIT_DEFAULTS = {"wit_max_eval": 768, "wit_max_leaves": 400, "wit_adaptive": 0, "wit_weight": 800, "wit_cmax_pct": 250, "wit_cmax_ext_pct": 60, "wit_min_bins": 48, "wit_flags": 0,
                "wit_max_open": 48, "wit_max_live_pct": 12, "wit_max_in_pct": 8}
WIT_SURFACE_PCT = 60   # the product's default: the witness sweep stands back on clouds that lie on a surface


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "the -m gpu tests need a GPU"
    _native.load()
    return torch.device("cuda:0")


@pytest.fixture(autouse=True)
def restore_options():
    keep = core.WIT_MIN_SIMPLICES, core.WIT_MAX_POINTS_PER_SIMPLEX
    core.WIT_MIN_SIMPLICES = 0    # (the product skips the witness sweep on short queues: the tests here want it run)
    core.WIT_MAX_POINTS_PER_SIMPLEX = 1 << 40   # (... and on clouds with many points per simplex)
    assert _native.load().flooder_set_option(b"wit_surface_pct", 0) == 0   # (... and on clouds that lie on a surface)
    yield
    core.WIT_MIN_SIMPLICES, core.WIT_MAX_POINTS_PER_SIMPLEX = keep
    lib = _native.load()
    assert lib.flooder_set_option(b"wit_surface_pct", WIT_SURFACE_PCT) == 0
    for k, v in WIT_DEFAULTS.items():
        assert lib.flooder_set_option(k.encode(), v) == 0
    core.CELL_WITNESS = True


def set_options(**kw):
    lib = _native.load()
    for k, v in kw.items():
        assert lib.flooder_set_option(k.encode(), int(v)) == 0, k


def clouds(name, n):
    g = torch.Generator().manual_seed(7)
    if name == "gauss":
        return torch.randn(n, 3, generator=g)
    if name == "torus":
        return torch.as_tensor(fo.noisy_torus(n, seed=3))
    if name == "cube":
        return torch.rand(n, 3, generator=g)
    if name == "two_blobs":   # a dense and a sparse cluster far apart: long empty simplices between them
        a = torch.randn(n // 2, 3, generator=g) * 0.05
        b = torch.randn(n - n // 2, 3, generator=g) + torch.tensor([6.0, 0.0, 0.0])
        return torch.cat([a, b])
    raise ValueError(name)


def run(points, lms, witness, **kw):
    core.CELL_WITNESS = witness
    return fa.flood_complex(points, lms, method="cell", **kw)


def assert_same(a, b, what):
    assert set(a) == set(b), what
    keys = sorted(a)
    va = np.array([a[k] for k in keys], dtype=np.float32)
    vb = np.array([b[k] for k in keys], dtype=np.float32)
    bad = va.view(np.uint32) != vb.view(np.uint32)
    assert not bad.any(), f"{what}: {int(bad.sum())} of {bad.size} values differ, worst {np.abs(va - vb).max():.3e}"


def sweep_stats(points, lms, ppe=30):
    """Counters of the witness sweep for the top simplices of a complex (through the sweep entry the product uses)."""
    dev = points.device
    d = points.shape[1]
    stree, simplices = core._build_complex(lms, d)
    verts = lms[torch.as_tensor(simplices[d], device=dev)].contiguous()
    weights, vertex_idxs, face_idxs = core.generate_grid(ppe, d, dev, torch.float32)
    faces = core._FaceTable(face_idxs, weights.shape[0], dev)
    st = torch.zeros(40, dtype=torch.int64, device=dev)
    core._sweep_dimension_cell(core.PointIndex(points), verts, weights, faces, None, stats=st)
    torch.cuda.synchronize()
    return st[16:40].cpu().numpy()


@pytest.mark.parametrize("name,n,n_lms", [("gauss", 200_000, 400), ("torus", 150_000, 300), ("cube", 100_000, 250),
                                          ("two_blobs", 120_000, 300)])
def test_witness_sweep_changes_nothing(name, n, n_lms, dev):
    pts = clouds(name, n).to(dev)
    lms = fa.generate_landmarks(pts, n_lms, start_idx=0)
    on = run(pts, lms, True)
    off = run(pts, lms, False)
    assert_same(on, off, f"{name}: witness sweep on / off")
    tree = fa.flood_complex(pts, lms, method="bvh")
    assert_same(on, tree, f"{name}: witness sweep / tree sweep")


def test_witness_sweep_takes_the_sparse_simplices_and_leaves_the_dense_ones(dev):
    pts = clouds("gauss", 300_000).to(dev)
    lms = fa.generate_landmarks(pts, 500, start_idx=0)
    st = sweep_stats(pts, lms)
    handled = int(st[0])
    n_top = core._build_complex(lms, 3)[1][3].shape[0]
    assert 0 < handled < n_top, (handled, n_top)   # the dense core of the cloud is the cell sweep's
    assert st[5] > 0.9 * 242 * handled, "coarse samples of a sparse simplex are (almost) all certified by the stage"
    assert st[6] < 0.1 * 4960 * handled, "more than a tenth of the samples survive the bound"
    dense = clouds("cube", 2_000_000).to(dev)
    lms_d = fa.generate_landmarks(dense, 300, start_idx=0)
    st_d = sweep_stats(dense, lms_d)
    assert int(st_d[0]) == 0, "a dense uniform cloud has nothing for the witness sweep"


def test_witness_sweep_stands_back_on_a_surface_cloud(dev):
    """Decided on the device from the density grid's cloud-kind words (flood_common.hpp): on the noisy torus no simplex
    is tried at all with the default option, every simplex is looked at with the gate off - and the values are the same."""
    tor = clouds("torus", 400_000).to(dev)
    lms = fa.generate_landmarks(tor, 400, start_idx=0)
    st_open = sweep_stats(tor, lms)
    assert int(st_open[:4].sum()) > 0, "gate off: the sweep looks at the simplices (handled / heavy / over / dense)"
    off = run(tor, lms, True)
    set_options(wit_surface_pct=WIT_SURFACE_PCT)
    st_gate = sweep_stats(tor, lms)
    assert int(st_gate[:12].sum()) == 0, st_gate[:12]
    assert_same(run(tor, lms, True), off, "torus: surface gate on / off")
    gau = clouds("gauss", 300_000).to(dev)   # a volume cloud is not touched by the gate
    lms_g = fa.generate_landmarks(gau, 500, start_idx=0)
    assert int(sweep_stats(gau, lms_g)[0]) > 0


def test_witness_sweep_random_weights_and_off_cloud_landmarks(dev):
    pts = clouds("gauss", 100_000).to(dev)
    lms = fa.generate_landmarks(pts, 200, start_idx=0)
    for seed in (0, 1):
        torch.manual_seed(seed)
        on = run(pts, lms, True, num_rand=3000, max_dimension=3)
        torch.manual_seed(seed)
        off = run(pts, lms, False, num_rand=3000, max_dimension=3)
        assert_same(on, off, "random weights")
    g = torch.Generator().manual_seed(5)
    off_cloud = (torch.randn(150, 3, generator=g) * 1.5).to(dev)   # landmarks that are no cloud points
    assert_same(run(pts, off_cloud, True), run(pts, off_cloud, False), "landmarks off the cloud")


@pytest.mark.parametrize("opts", [
    dict(wit_weight=1_000_000, wit_max_in_pct=100000),            # every simplex is tried, however dense
    dict(wit_cmax_pct=60), dict(wit_cmax_pct=1200, wit_cmax_ext_pct=400),   # tiny / huge staged region
    dict(wit_flags=1), dict(wit_flags=2), dict(wit_flags=4), dict(wit_flags=8), dict(wit_flags=15),
    dict(wit_max_open=0), dict(wit_max_live_pct=0), dict(wit_min_bins=64), dict(wit_min_bins=1, wit_max_eval=0),
    dict(wit_min_bins=1, wit_max_eval=64, wit_weight=5000), dict(wit_min_bins=1, wit_max_leaves=1024, wit_weight=3000),
    dict(wit_max_open=100000, wit_max_live_pct=100, wit_flags=8),  # nothing is abandoned: queue / list overflow paths
])
def test_witness_sweep_options_change_nothing(opts, dev):
    pts = clouds("two_blobs", 150_000).to(dev)
    lms = fa.generate_landmarks(pts, 300, start_idx=0)
    ref = run(pts, lms, False)
    set_options(**opts)
    assert_same(run(pts, lms, True), ref, str(opts))
    tor = clouds("torus", 100_000).to(dev)
    lms_t = fa.generate_landmarks(tor, 200, start_idx=0)
    set_options(**WIT_DEFAULTS)
    ref_t = run(tor, lms_t, False)
    set_options(**opts)
    assert_same(run(tor, lms_t, True), ref_t, f"torus {opts}")


@pytest.mark.parametrize("case", ["triangles_in_3d", "plane_cloud", "large_lattice", "ragged_rows"])
def test_witness_sweep_other_shapes(case, dev):
    """Simplices of lower dimension than the cloud (no face planes: the region is a box), a 2-D cloud, the largest
    lattice the kernel takes (R = 7770 of 8192), a row count that is no multiple of 64 or 256."""
    g = torch.Generator().manual_seed(21)
    if case == "triangles_in_3d":
        pts, n_l, kw = torch.randn(120_000, 3, generator=g), 120, dict(max_dimension=2, points_per_edge=40)
    elif case == "plane_cloud":
        pts, n_l, kw = torch.randn(150_000, 2, generator=g), 2500, dict(points_per_edge=40)
    elif case == "large_lattice":
        pts, n_l, kw = torch.randn(60_000, 3, generator=g), 60, dict(points_per_edge=35)
    else:
        pts, n_l, kw = torch.randn(100_000, 3, generator=g), 250, dict(points_per_edge=17)   # R = 969
    pts = pts.to(dev)
    lms = fa.generate_landmarks(pts, n_l, start_idx=0)
    d = kw.get("max_dimension", pts.shape[1])
    w, _, fi = core.generate_grid(kw["points_per_edge"], d, dev, torch.float32)
    assert core.SamplePlan(w, core._FaceTable(fi, w.shape[0], dev)).wit is not None, "no witness plan for this lattice"
    on = run(pts, lms, True, **kw)
    off = run(pts, lms, False, **kw)
    assert_same(on, off, case)
    assert_same(on, fa.flood_complex(pts, lms, method="bvh", **kw), case + " / tree sweep")
