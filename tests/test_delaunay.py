"""Native 3-D Delaunay triangulation of the landmarks (csrc/delaunay3d.cpp; replaces what the reference
takes from gudhi.DelaunayComplex, flooder/core.py:130-138) against Qhull and against the defining property."""
import itertools
from fractions import Fraction

import numpy as np
import pytest
from scipy.spatial import ConvexHull, Delaunay

from flooder_amd import simplex_tree as stm
from oracle import flood_oracle as fo


def native(P):
    cells = stm._delaunay3d_native(np.asarray(P, dtype=np.float64))
    return None if cells is None else stm._unique_rows(np.sort(cells, axis=1))


def qhull(P):
    return stm._unique_rows(np.sort(Delaunay(np.asarray(P, dtype=np.float64)).simplices.astype(np.int64), axis=1))


def clouds():
    rng = np.random.default_rng(7)
    yield "gauss", rng.normal(size=(3000, 3)).astype(np.float32)
    yield "cube", rng.random(size=(2000, 3)).astype(np.float32)
    yield "torus", fo.noisy_torus(2500, seed=4)
    yield "far_from_origin", (rng.normal(size=(1500, 3)) * 0.01 + np.array([100.0, -250.0, 40.0])).astype(np.float32)
    yield "anisotropic", (rng.normal(size=(1500, 3)) * np.array([1.0, 1e-2, 1e-4])).astype(np.float32)
    yield "sphere_shell", (lambda v: (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32))(rng.normal(size=(800, 3)))
    yield "five", rng.normal(size=(5, 3)).astype(np.float32)


@pytest.mark.parametrize("name,P", list(clouds()), ids=[c[0] for c in clouds()])
def test_native_delaunay_equals_qhull(name, P):
    """Points in general position have ONE Delaunay triangulation: the same set of tetrahedra as Qhull's (which equals
    gudhi's / CGAL's on the reference's committed clouds)."""
    got = native(P)
    assert got is not None and stm.LAST_DELAUNAY["native"], stm.LAST_DELAUNAY
    ref = qhull(P)
    if np.array_equal(got, ref):
        return
    # Qhull works in floating point: on a cloud far from the origin (coordinates quantised to 1.5e-5, near-ties
    # everywhere) a handful of its tetrahedra have a point strictly INSIDE their circumsphere.  The native routine
    # decides every predicate exactly - as CGAL behind gudhi does - so where the two differ, every tetrahedron only the
    # native one has must be empty, and every one only Qhull has must be violated (or be an exact tie).
    assert name == "far_from_origin", "the two triangulations should agree on this cloud"
    sa, sb = set(map(tuple, got.tolist())), set(map(tuple, ref.tolist()))
    assert len(got) == len(ref) and len(sa - sb) <= len(sa) // 100

    def worst(tet):
        o = _exact_orient(P, tet)
        assert o != 0
        return max((1 if o > 0 else -1) * _exact_insphere(P, tet, e) for e in range(len(P)) if e not in tet)

    assert all(worst(t) < 0 for t in sa - sb)
    assert all(worst(t) >= 0 for t in sb - sa)


def test_edge_links_local_table_and_hash_table_agree():
    """The new tetrahedra of an insertion are linked through a table over the locally numbered boundary vertices, or
    through a hash table when the boundary is large (include/flooder_host.h: flooder_delaunay3d_local_edges): the
    same tetrahedra either way - on random clouds and on a lattice, whose cospherical points make large cavities."""
    import ctypes

    from flooder_amd import build

    rng = np.random.default_rng(3)
    g = np.arange(9, dtype=np.float32)
    lattice = np.stack(np.meshgrid(g, g, g, indexing="ij"), axis=-1).reshape(-1, 3)
    lattice = lattice[rng.permutation(len(lattice))]
    cases = [rng.normal(size=(1200, 3)).astype(np.float32), lattice, fo.noisy_torus(900, seed=2)]
    host = ctypes.CDLL(build.HOST_LIB)
    default = host.flooder_delaunay3d_local_edges(-1)
    assert default == 160
    try:
        for P in cases:
            out = {}
            for limit in (160, 6, 0):
                host.flooder_delaunay3d_local_edges(limit)
                out[limit] = native(P)
                assert out[limit] is not None
            assert np.array_equal(out[160], out[0]) and np.array_equal(out[160], out[6])
    finally:
        host.flooder_delaunay3d_local_edges(default)


def test_landmark_sets_of_the_benchmark_clouds():
    """What the path really triangulates: farthest-point landmarks (well spread, nothing like random points)."""
    for P in (fo.noisy_torus(40_000, seed=1), np.random.default_rng(2).normal(size=(40_000, 3)).astype(np.float32)):
        L = P[fo.exact_fps(P, 600, 0)]
        got = native(L)
        assert got is not None
        assert np.array_equal(got, qhull(L))
        assert np.array_equal(stm.delaunay_cells(L), got)          # (the product's entry takes the native path)


def _exact_insphere(P, tet, e):
    rows = []
    for v in tet:
        d = [Fraction(float(P[v][k])) - Fraction(float(P[e][k])) for k in range(3)]
        rows.append(d + [sum(x * x for x in d)])
    def det(m):
        if len(m) == 1:
            return m[0][0]
        return sum((-1) ** j * m[0][j] * det([r[:j] + r[j + 1:] for r in m[1:]]) for j in range(len(m)))
    return det(rows)


def _exact_orient(P, tet):
    a, b, c, d = ([Fraction(float(x)) for x in P[v]] for v in tet)
    m = [[a[k] - d[k] for k in range(3)], [b[k] - d[k] for k in range(3)], [c[k] - d[k] for k in range(3)]]
    return (m[0][0] * (m[1][1] * m[2][2] - m[1][2] * m[2][1]) - m[0][1] * (m[1][0] * m[2][2] - m[1][2] * m[2][0])
            + m[0][2] * (m[1][0] * m[2][1] - m[1][1] * m[2][0]))


@pytest.mark.parametrize("case", ["lattice", "cospherical", "coplanar_on_hull"])
def test_degenerate_inputs_give_a_valid_delaunay_triangulation(case):
    """Cospherical points have several Delaunay triangulations; whichever comes out must BE one: non-flat tetrahedra,
    volumes adding up to the hull's, no point strictly inside any circumsphere (checked in rational arithmetic)."""
    rng = np.random.default_rng(3)
    if case == "lattice":
        g = np.arange(4, dtype=np.float32)
        P = np.stack(np.meshgrid(g, g, g, indexing="ij"), axis=-1).reshape(-1, 3)
    elif case == "cospherical":
        v = np.array(list(itertools.product((-1.0, 1.0), repeat=3)), dtype=np.float32)       # cube corners
        P = np.concatenate([v, rng.normal(size=(12, 3)).astype(np.float32) * 0.3])
    else:
        sq = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [1, 1, 0], [0.5, 0.5, 0], [0.25, 0.75, 0]], dtype=np.float32)
        P = np.concatenate([sq, np.array([[0.3, 0.4, 1.0], [0.6, 0.2, 0.7], [0.5, 0.5, 0.2]], dtype=np.float32)])
    T = native(P)
    if T is None:   # declined (the cavity of a degenerate insertion may come out inconsistent): Qhull takes over
        assert np.array_equal(stm.delaunay_cells(P), qhull(P))
        return
    vol = 0
    for tet in T:
        o = _exact_orient(P, tet)
        assert o != 0
        vol += abs(o)
        sgn = 1 if o > 0 else -1
        for e in range(len(P)):
            if e not in tet:
                assert sgn * _exact_insphere(P, tet, e) <= 0, (tet, e)
    assert abs(float(vol) / 6.0 - ConvexHull(P.astype(np.float64)).volume) < 1e-9 * max(1.0, float(vol))


def test_inputs_the_routine_declines_fall_back_to_qhull():
    rng = np.random.default_rng(5)
    P = rng.normal(size=(200, 3))                       # float64 with full mantissas: does not scale to 58-bit integers
    import ctypes

    out = np.empty((2000, 4), dtype=np.int32)
    assert int(stm._load_host().flooder_delaunay3d(P.ctypes.data, 200, out.ctypes.data, 2000)) == stm.E_RANGE
    # ... round 6: the d-dimensional routine (a grid of up to 121 bits) takes what the incremental one declines for RANGE
    got = native(P)
    assert got is not None and stm.LAST_DELAUNAY["routine"] == "nd" and np.array_equal(got, qhull(P))
    assert np.array_equal(stm.delaunay_cells(P), qhull(P))
    huge = P.copy()
    huge[0, 0], huge[1, 0] = 2.0 ** 90, 2.0 ** -40        # beyond that grid too: Qhull
    assert native(huge) is None and stm.LAST_DELAUNAY["code"] < -(1 << 40)
    Q = rng.normal(size=(50, 3)).astype(np.float32)
    Q[7] = Q[3]                                          # a duplicate point
    assert native(Q) is None
    flat = np.concatenate([rng.normal(size=(30, 2)), np.zeros((30, 1))], axis=1).astype(np.float32)   # all coplanar
    assert native(flat) is None
    assert native(np.full((10, 3), np.nan)) is None


# ------------------------------------------------------------------------------------------------ two dimensions
def clouds_2d():
    rng = np.random.default_rng(11)
    yield "gauss", rng.normal(size=(3000, 2)).astype(np.float32)
    yield "square", rng.random(size=(2000, 2)).astype(np.float32)
    t = rng.random(2500) * 2 * np.pi
    r = 1.0 + 0.1 * rng.normal(size=2500)
    yield "annulus", np.stack([r * np.cos(t), r * np.sin(t)], axis=1).astype(np.float32)
    yield "figure_eight", np.stack([np.sin(t), np.sin(t) * np.cos(t)], axis=1).astype(np.float32) + (0.02 * rng.normal(size=(2500, 2))).astype(np.float32)
    yield "far_from_origin", (rng.normal(size=(1500, 2)) * 0.01 + np.array([100.0, -250.0])).astype(np.float32)
    yield "anisotropic", (rng.normal(size=(1500, 2)) * np.array([1.0, 1e-4])).astype(np.float32)
    yield "four", rng.normal(size=(4, 2)).astype(np.float32)


def _exact_orient2(P, tri):
    a, b, c = ([Fraction(float(x)) for x in P[v]] for v in tri)
    return (a[0] - c[0]) * (b[1] - c[1]) - (a[1] - c[1]) * (b[0] - c[0])


def _exact_incircle(P, tri, e):
    rows = []
    for v in tri:
        d = [Fraction(float(P[v][k])) - Fraction(float(P[e][k])) for k in range(2)]
        rows.append(d + [d[0] * d[0] + d[1] * d[1]])
    (a, b, c), (d, e_, f), (g, h, i) = rows
    return a * (e_ * i - f * h) - b * (d * i - f * g) + c * (d * h - e_ * g)


@pytest.mark.parametrize("name,P", list(clouds_2d()), ids=[c[0] for c in clouds_2d()])
def test_native_delaunay_2d_equals_qhull(name, P):
    """csrc/delaunay2d.cpp: the same triangles as Qhull on points in general position; where the two differ (near-ties
    that Qhull's floating point decides the other way) every triangle only the native routine has is empty and every
    one only Qhull has holds a point inside or on its circumcircle, in rational arithmetic."""
    got = native(P)
    assert got is not None and stm.LAST_DELAUNAY["native"], stm.LAST_DELAUNAY
    assert np.array_equal(stm.delaunay_cells(P), got)              # (the product's entry takes the native path)
    ref = qhull(P)
    if np.array_equal(got, ref):
        return
    sa, sb = set(map(tuple, got.tolist())), set(map(tuple, ref.tolist()))
    assert len(sa - sb) <= max(4, len(sa) // 100), (len(sa - sb), len(sa))

    def worst(tri):
        o = _exact_orient2(P, tri)
        if o == 0:
            return 1          # (a flat triangle: Qhull's, never ours)
        return max((1 if o > 0 else -1) * _exact_incircle(P, tri, e) for e in range(len(P)) if e not in tri)

    assert all(worst(t) < 0 for t in sa - sb)
    assert all(worst(t) >= 0 for t in sb - sa)


def test_native_delaunay_2d_counter_clockwise_and_fps_landmarks():
    """Raw output: counter-clockwise triangles (include/flooder_host.h); and the landmark sets the path really
    triangulates (farthest points of a 2-D cloud)."""
    rng = np.random.default_rng(12)
    P = rng.normal(size=(500, 2)).astype(np.float32)
    raw = stm._delaunay_native(P.astype(np.float64))
    a, b, c = (P[raw[:, k]].astype(np.float64) for k in range(3))
    assert np.all((a[:, 0] - c[:, 0]) * (b[:, 1] - c[:, 1]) - (a[:, 1] - c[:, 1]) * (b[:, 0] - c[:, 0]) > 0)
    cloud = rng.normal(size=(30_000, 2)).astype(np.float32)
    L = cloud[fo.exact_fps(cloud, 500, 0)]
    assert np.array_equal(native(L), qhull(L))


@pytest.mark.parametrize("case", ["lattice", "cocircular", "collinear_on_hull"])
def test_degenerate_2d_inputs_give_a_valid_delaunay_triangulation(case):
    """Cocircular points have several Delaunay triangulations; whichever comes out must BE one: no flat triangle, areas
    adding up to the hull's, no point strictly inside any circumcircle (rational arithmetic)."""
    rng = np.random.default_rng(13)
    if case == "lattice":
        g = np.arange(7, dtype=np.float32)
        P = np.stack(np.meshgrid(g, g, indexing="ij"), axis=-1).reshape(-1, 2)
        P = P[rng.permutation(len(P))]
    elif case == "cocircular":
        v = np.array([[3, 4], [4, 3], [-3, 4], [-4, 3], [3, -4], [4, -3], [-3, -4], [-4, -3], [5, 0], [0, 5], [-5, 0], [0, -5]], dtype=np.float32)
        P = np.concatenate([v, rng.normal(size=(10, 2)).astype(np.float32)])
    else:
        P = np.array([[0, 0], [1, 0], [2, 0], [3, 0], [4, 0], [0, 3], [4, 3], [2, 1], [1.5, 0], [2, 3]], dtype=np.float32)
    T = native(P)
    if T is None:   # declined: Qhull takes over
        assert np.array_equal(stm.delaunay_cells(P), qhull(P))
        return
    area = 0
    for tri in T:
        o = _exact_orient2(P, tri)
        assert o != 0
        area += abs(o)
        sgn = 1 if o > 0 else -1
        for e in range(len(P)):
            if e not in tri:
                assert sgn * _exact_incircle(P, tri, e) <= 0, (tri, e)
    assert abs(float(area) / 2.0 - ConvexHull(P.astype(np.float64)).volume) < 1e-9 * max(1.0, float(area))


def test_2d_inputs_the_routine_declines_fall_back_to_qhull():
    rng = np.random.default_rng(14)
    P = rng.normal(size=(200, 2))                       # float64 with full mantissas
    out = np.empty((800, 3), dtype=np.int32)
    assert int(stm._load_host().flooder_delaunay2d(P.ctypes.data, 200, out.ctypes.data, 800)) == stm.E_RANGE
    got = native(P)                                      # (round 6: taken by the d-dimensional routine instead)
    assert got is not None and stm.LAST_DELAUNAY["routine"] == "nd" and np.array_equal(got, qhull(P))
    assert np.array_equal(stm.delaunay_cells(P), qhull(P))
    Q = rng.normal(size=(50, 2)).astype(np.float32)
    Q[7] = Q[3]                                          # a duplicate point
    assert native(Q) is None
    line = np.stack([np.arange(20.0), 2.0 * np.arange(20.0)], axis=1).astype(np.float32)   # all collinear
    assert native(line) is None
    assert native(np.full((10, 2), np.inf)) is None
