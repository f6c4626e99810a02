"""The oracle (oracle/flood_oracle.py) pinned against outputs of the reference itself
(tests/golden/*.npz, made by oracle/make_goldens.py) and the reference's committed known-answer data
(docs/animation/*.csv of the reference, copied as data)."""
import os

import numpy as np
import pytest
import torch

from oracle import flood_oracle as fo
from helpers import GOLDEN, e2e_cases, load_e2e, dict_values

GRID_CASES = [(2, 1), (5, 2), (8, 3), (30, 3), (4, 4), (3, 6), (20, 2)]


@pytest.mark.parametrize("n,dim", GRID_CASES)
def test_generate_grid_matches_reference(n, dim):
    g = np.load(os.path.join(GOLDEN, "grid_vectors.npz"))
    w, v, f = fo.generate_grid(n, dim)
    assert np.array_equal(w, g[f"w_{n}_{dim}"])
    for k in range(dim + 1):
        assert np.array_equal(v[k], g[f"v_{n}_{dim}_{k}"])
        assert np.array_equal(f[k], g[f"f_{n}_{dim}_{k}"])


def test_uniform_weights_match_reference():
    g = np.load(os.path.join(GOLDEN, "grid_vectors.npz"))
    torch.manual_seed(42)
    assert np.array_equal(fo.generate_uniform_weights(64, 3), g["u_64_3_seed42"])
    assert np.array_equal(fo.generate_uniform_weights(7, 0), g["u_7_0"])


@pytest.mark.parametrize("name", e2e_cases())
def test_kdtree_oracle_matches_reference_cpu_path(name):
    z, kw, keys = load_e2e(name)
    torch.manual_seed(int(z["weight_seed"]))
    fc = fo.flood_complex_oracle(z["points"], z["landmarks"], **kw)
    assert set(keys) == set(fc)
    # same arithmetic as the reference CPU branch: agreement to float32 matmul rounding
    assert np.abs(dict_values(fc, keys) - z["filtration_f32"]).max() < 5e-7


@pytest.mark.parametrize("name", ["torus3d_grid", "eight2d_rand", "gauss4d_grid", "lms_eq_pts2d"])
def test_masked_oracle_matches_reference_cpu_path(name):
    """GPU formulation (ball mask + direct differences) == kd-tree path when landmarks are points."""
    z, kw, keys = load_e2e(name)
    torch.manual_seed(int(z["weight_seed"]))
    fc = fo.flood_complex_oracle(z["points"], z["landmarks"], mode="masked", **kw)
    ref = z["filtration_f32"]
    got = dict_values(fc, keys)
    assert (np.abs(got - ref) <= 1e-5 * np.abs(ref) + 5e-7).all()


@pytest.mark.parametrize("name", ["k2d", "k3d", "k5d"])
def test_masked_min_dist_matches_reference_triton_kernels(name):
    z = np.load(os.path.join(GOLDEN, f"kernel_{name}.npz"))
    d = fo.masked_min_dist(z["samples"], z["points"], z["centers"], z["radii"])
    ref = z["min_dist"]
    assert np.array_equal(np.isfinite(d), np.isfinite(ref))
    fin = np.isfinite(ref)
    assert np.abs(d[fin] - ref[fin]).max() <= 1e-6


def _docs_kat():
    pts = np.loadtxt(os.path.join(GOLDEN, "docs_animation_points.csv"), delimiter=",").astype(np.float32)
    lms = np.loadtxt(os.path.join(GOLDEN, "docs_animation_landmarks.csv"), delimiter=",").astype(np.float32)
    edges = np.loadtxt(os.path.join(GOLDEN, "docs_animation_edges.csv"), delimiter=",")
    tris = np.loadtxt(os.path.join(GOLDEN, "docs_animation_triangles.csv"), delimiter=",")
    return pts, lms, edges, tris


def test_docs_known_answer_values():
    """94 filtration values committed in the reference's docs (written by its own flood_complex with
    the real gudhi + fpsample stack); reproduced at points_per_edge=31."""
    pts, lms, edges, tris = _docs_kat()
    fc = fo.flood_complex_oracle(pts, lms, points_per_edge=31)
    for a, b, f in edges:
        assert abs(fc[(int(a), int(b))] - f) < 2e-8
    for a, b, c, f in tris:
        assert abs(fc[(int(a), int(b), int(c))] - f) < 2e-8


def test_docs_landmarks_are_exact_fps_order():
    pts, lms, _, _ = _docs_kat()
    start = int(np.argmin(np.abs(pts - lms[0]).sum(axis=1)))
    idx = fo.exact_fps(pts, len(lms), start)
    assert np.array_equal(pts[idx], lms)


def test_make_filtration_non_decreasing():
    f = {(0,): 0.0, (1,): 0.0, (2,): 0.0, (0, 1): 0.5, (0, 2): 0.2, (1, 2): 0.1, (0, 1, 2): 0.3}
    m = fo.make_filtration_non_decreasing(f)
    assert m[(0, 1, 2)] == 0.5 and m[(0, 1)] == 0.5 and m[(1, 2)] == 0.1
