"""Delaunay triangulation above three dimensions on all host cores (csrc/delaunay_nd.cpp; replaces what
the reference takes from gudhi.DelaunayComplex, flooder/core.py:130-138), the parallel face tables (csrc/cell_faces.cpp,
core.py:135-138) and the reference's own gudhi output (docs/visualization/*) as the pin of all native routines."""
import ctypes
import itertools
import os
from fractions import Fraction

import numpy as np
import pytest
from scipy.spatial import Delaunay

from flooder_amd import build, simplex_tree as stm
from helpers import GOLDEN

E_BASE = -(1 << 40)


def host():
    lib = stm._load_host()
    assert lib is not None
    return lib


def nd(P, threads=0):
    """Rows of flooder_delaunay_nd (already ascending ids, lexicographic order), or the decline code."""
    lib = host()
    P = np.ascontiguousarray(P, dtype=np.float64)
    out = ctypes.POINTER(ctypes.c_int32)()
    rc = int(lib.flooder_delaunay_nd(P.ctypes.data, P.shape[0], P.shape[1], threads, ctypes.byref(out)))
    if rc < 0:
        return rc
    try:
        return np.ctypeslib.as_array(out, shape=(rc, P.shape[1] + 1)).astype(np.int64)
    finally:
        lib.flooder_host_free(out)


def qhull(P):
    return stm._unique_rows(np.sort(Delaunay(np.asarray(P, dtype=np.float64)).simplices.astype(np.int64), axis=1))


@pytest.mark.parametrize("dim,n", [(2, 700), (3, 600), (4, 500), (5, 300), (6, 200), (6, 400), (7, 100), (8, 60)])
def test_nd_equals_qhull_on_general_position_clouds(dim, n):
    """Points in general position have ONE Delaunay triangulation: the same simplices as Qhull (and as the 2-D / 3-D
    incremental routines), rows ascending, table in lexicographic order without duplicates."""
    rng = np.random.default_rng(100 * dim + n)
    P = rng.normal(size=(n, dim)).astype(np.float32)
    got = nd(P)
    assert isinstance(got, np.ndarray), got
    assert (np.diff(got, axis=1) > 0).all()
    assert np.array_equal(got, stm._unique_rows(got)), "rows not in lexicographic order or not distinct"
    assert np.array_equal(got, qhull(P))
    if dim in (2, 3):
        inc = stm._delaunay_native(P.astype(np.float64))
        assert stm.LAST_DELAUNAY["routine"] == "incremental"
        assert np.array_equal(got, stm._unique_rows(np.sort(inc, axis=1)))


def test_threads_and_instruction_sets_give_the_same_table():
    """1 thread, all threads, generic / AVX2 / AVX-512 scan (where the CPU has them): the same rows, bit for bit - the
    vector code only proposes, every decision is covered by its error bound or made exactly."""
    lib = host()
    rng = np.random.default_rng(5)
    P = rng.normal(size=(300, 5)).astype(np.float32)
    want = nd(P, threads=1)
    assert np.array_equal(want, nd(P, threads=0)) and np.array_equal(want, nd(P, threads=3))
    assert lib.flooder_delaunay_nd_stat(2) == 3
    flags = open("/proc/cpuinfo").read()
    old = lib.flooder_delaunay_nd_isa(-1)
    try:
        for isa, need in ((0, ""), (1, "avx2"), (2, "avx512dq")):
            if need and need not in flags:
                continue
            lib.flooder_delaunay_nd_isa(isa)
            assert np.array_equal(want, nd(P)), f"isa {isa}"
    finally:
        lib.flooder_delaunay_nd_isa(old)


def _det(rows):
    """Exact determinant of a list of rows of Fractions (Laplace; tiny matrices only)."""
    n = len(rows)
    if n == 1:
        return rows[0][0]
    return sum((-1) ** j * rows[0][j] * _det([r[:j] + r[j + 1:] for r in rows[1:]]) for j in range(n) if rows[0][j] != 0)


def _insphere_sign(P, simplex, e):
    """> 0: e strictly inside the circumsphere of the simplex (exact rational arithmetic)."""
    F = [[Fraction(float(x)) for x in P[i]] for i in simplex]
    E = [Fraction(float(x)) for x in P[e]]
    rows = [[a - b for a, b in zip(r, E)] for r in F]
    lifted = [r + [sum(x * x for x in r)] for r in rows]
    orient = _det([[a - b for a, b in zip(r, F[0])] for r in F[1:]])
    assert orient != 0
    d = len(E)
    s = _det(lifted) * (1 if orient > 0 else -1) * (-1) ** d       # (sign convention checked on a regular simplex below)
    return (s > 0) - (s < 0)


def test_insphere_convention_of_the_checker():
    d = 4
    S = np.vstack([np.zeros(d), np.eye(d)])
    P = np.vstack([S, np.full(d, 0.25), np.full(d, 5.0)])
    assert _insphere_sign(P, range(d + 1), d + 1) > 0 and _insphere_sign(P, range(d + 1), d + 2) < 0


@pytest.mark.parametrize("dim,n,scale", [(4, 40, 1.0), (5, 30, 1.0), (4, 45, 2.0 ** -7)])
def test_empty_circumsphere_property_exactly(dim, n, scale):
    """The defining property, in exact rational arithmetic: no point strictly inside the circumsphere of any simplex,
    the simplices cover the hull (same count as Qhull).  The third case snaps the coordinates to a coarse grid far from
    the origin - near-ties everywhere, which is where the exact stage of the routine decides."""
    rng = np.random.default_rng(dim * 1000 + n)
    P = rng.normal(size=(n, dim))
    if scale != 1.0:
        P = np.round(P / scale * 0.05) * scale + 100.0
        P = np.unique(P, axis=0)
    P = P.astype(np.float32)
    got = nd(P)
    if not isinstance(got, np.ndarray):
        assert got < E_BASE and scale != 1.0      # an exact tie on the grid: declined, Qhull then
        return
    assert len(got) == len(qhull(P))
    for s in got[:: max(1, len(got) // 60)]:
        assert all(_insphere_sign(P, s, e) <= 0 for e in range(len(P)) if e not in s)


def test_exact_stage_is_reached_and_decides_like_qhull():
    """In 6 - 8 dimensions a few comparisons per cloud fall inside the error bounds of the floating-point forms
    (slivers whose inverse is badly conditioned) and are decided over the multi-word integers: the table is still
    Qhull's."""
    lib = host()
    rng = np.random.default_rng(11)
    total = 0
    for dim, n in ((8, 70), (6, 300), (8, 70), (7, 100)):
        P = rng.normal(size=(n, dim)).astype(np.float32)
        got = nd(P)
        assert isinstance(got, np.ndarray)
        total += lib.flooder_delaunay_nd_stat(0)
        assert np.array_equal(got, qhull(P))
    assert total > 0, "no case reached the exact predicates"


def test_large_low_dimensional_inputs_go_to_qhull(monkeypatch):
    """The scan per pivot is O(n): above ``ND_MAX_POINTS[dim]`` points (2 - 5 dimensions) Qhull's n log n wins and is
    asked instead."""
    rng = np.random.default_rng(3)
    P = rng.normal(size=(300, 4))
    monkeypatch.setitem(stm.ND_MAX_POINTS, 4, 100)
    cells = stm.delaunay_cells(P)
    assert not stm.LAST_DELAUNAY["native"] and np.array_equal(cells, qhull(P))
    monkeypatch.setitem(stm.ND_MAX_POINTS, 4, 1000)
    assert np.array_equal(stm.delaunay_cells(P), cells) and stm.LAST_DELAUNAY["native"]


def test_declines_what_it_cannot_triangulate_and_delaunay_cells_falls_back():
    rng = np.random.default_rng(2)
    P = rng.normal(size=(100, 4))
    dup = np.vstack([P, P[:1]])
    assert nd(dup) == E_BASE - 4                                   # duplicate points
    g = np.arange(4, dtype=np.float64)
    lattice = np.stack(np.meshgrid(g, g, g, g, indexing="ij"), axis=-1).reshape(-1, 4)
    assert nd(lattice) < E_BASE                                    # cospherical points everywhere: an exact tie
    flat = np.hstack([P[:, :3], np.zeros((100, 1))])
    assert nd(flat) < E_BASE                                       # all points on a hyperplane
    assert nd(np.full((20, 4), np.nan)) < E_BASE
    assert nd(P[:, :1].repeat(9, axis=1)) < E_BASE                 # dimension 9
    wide = P.copy()
    wide[0, 0] = 2.0 ** 90
    wide[1, 0] = 2.0 ** -40
    assert nd(wide) == E_BASE - 2                                  # exponent spread beyond the 121-bit grid
    cells = stm.delaunay_cells(lattice)                            # Qhull takes over
    assert not stm.LAST_DELAUNAY["native"] and cells.shape[1] == 5 and len(cells) > 0
    assert np.array_equal(stm.delaunay_cells(P), qhull(P)) and stm.LAST_DELAUNAY["native"]


@pytest.mark.parametrize("name", ["gauss4d_grid", "gauss6d_maxdim2"])
def test_landmarks_of_the_reference_goldens(name):
    """The landmark sets of the reference-generated end-to-end fixtures above 3-D (oracle/make_goldens.py): the native
    cells are the cells the fixture's simplices were enumerated from."""
    z = np.load(os.path.join(GOLDEN, f"e2e_{name}.npz"))
    L = z["landmarks"]
    cells = stm.delaunay_cells(L)
    assert stm.LAST_DELAUNAY["native"] and stm.LAST_DELAUNAY["routine"] == "nd"
    assert np.array_equal(cells, qhull(L))
    md = int(z["max_dimension"])
    top = L.shape[1] if md < 0 else min(md, L.shape[1])
    keys = {tuple(int(v) for v in row if v >= 0) for row in z["simplices"]}
    for d in range(top + 1):
        rows, _ = stm.faces_of_cells(cells, d, len(L))
        assert {tuple(r) for r in rows.tolist()} == {k for k in keys if len(k) == d + 1}


@pytest.mark.parametrize("dim,n,k", [(4, 400, 3), (6, 150, 3), (6, 150, 2), (5, 200, 6), (3, 500, 2)])
def test_parallel_face_table_equals_numpy(dim, n, k, monkeypatch):
    """flooder_cell_faces: the distinct k-vertex faces of the cells, sorted - what numpy's unique over the packed keys
    gives (simplex_tree.faces_of_cells), and what core.py:135-138 buckets out of stree.get_simplices()."""
    rng = np.random.default_rng(dim + n + k)
    cells = qhull(rng.normal(size=(n, dim)))
    monkeypatch.setattr(stm, "NATIVE_FACES_MIN", 10 ** 15)
    want, _ = stm.faces_of_cells(cells, k - 1, n, want_index=False)
    got = stm._faces_native(cells, k - 1, n)
    assert got is not None and got.dtype == np.int64 and np.array_equal(got, want)
    monkeypatch.setattr(stm, "NATIVE_FACES_MIN", 1)
    assert np.array_equal(stm.faces_of_cells(cells, k - 1, n, want_index=False)[0], want)
    brute = sorted({tuple(c[list(cb)]) for c in cells.tolist() for c in [np.array(c)] for cb in itertools.combinations(range(dim + 1), k)})
    assert got.tolist() == [list(map(int, r)) for r in brute]


def test_wide_keys_give_the_same_tables():
    """Where n_points^k >= 2^62 (the 6- and 7-vertex faces of a 6-D complex over 2000 landmarks) the packed keys are 128
    bits wide: written out, sorted, made distinct; lookups and the monotone pass search them the same way.  Forced here
    by passing a large base for a small complex: the tables must not depend on it."""
    lib = host()
    rng = np.random.default_rng(12)
    cells = qhull(rng.normal(size=(90, 5)))
    for k in (3, 5, 6):
        narrow = stm._faces_native(cells, k - 1, 90)
        wide = stm._faces_native(cells, k - 1, 1 << 20)                # 60 / 100 / 120 bits: wide keys from k = 5
        assert narrow is not None and wide is not None and np.array_equal(narrow, wide)
    table = stm._faces_native(cells, 3, 90)
    lower = stm._faces_native(cells, 2, 90)
    q = np.concatenate([table[::3], np.sort(rng.integers(0, 90, size=(200, 4)), axis=1)])
    out = {}
    for base in (90, 1 << 24):
        o = np.empty(len(q), dtype=np.int64)
        assert lib.flooder_locate_rows(q.ctypes.data, len(q), 4, table.ctypes.data, len(table), base, o.ctypes.data, 2) == 0
        out[base] = o
    assert np.array_equal(out[90], out[1 << 24]) and (out[90][: len(table[::3])] >= 0).all()
    lv = rng.random(len(lower))
    lv[::7] = np.nan
    res = {}
    for base in (90, 1 << 24):
        v = rng.random(len(table)) * 0.5
        v[::5] = np.nan
        v0 = np.random.default_rng(1).random(len(table)) * 0.5
        v0[::5] = np.nan
        rc = lib.flooder_raise_dimension(table.ctypes.data, len(table), 4, lower.ctypes.data, len(lower), lv.ctypes.data,
                                         v0.ctypes.data, base, 2)
        assert rc > 0
        res[base] = v0
    assert np.array_equal(res[90], res[1 << 24], equal_nan=True)


def test_face_table_declines_keys_that_do_not_fit():
    lib = host()
    cells = np.array([[0, 1, 2, 3, 4, 5, 6]], dtype=np.int32)
    out = ctypes.POINTER(ctypes.c_int32)()
    assert lib.flooder_cell_faces(cells.ctypes.data, 1, 7, 7, 1 << 20, 1, ctypes.byref(out)) < E_BASE   # (2^20)^7: 140 bits
    assert lib.flooder_cell_faces(cells.ctypes.data, 1, 7, 7, 1 << 17, 1, ctypes.byref(out)) == 1      # 119 bits: wide keys
    lib.flooder_host_free(out)
    rc = lib.flooder_cell_faces(cells.ctypes.data, 1, 7, 3, 7, 1, ctypes.byref(out))
    assert rc == 35
    lib.flooder_host_free(out)


def test_parallel_row_lookup_equals_numpy(monkeypatch):
    """flooder_locate_rows against the searchsorted path of SimplexTree._locate: present rows, absent rows, ids outside
    the table's range."""
    rng = np.random.default_rng(4)
    cells = qhull(rng.normal(size=(150, 4)))
    st = stm.SimplexTree.from_cells(cells, 150)
    for d in (1, 2, 4):
        rows = st.simplices_of_dimension(d)
        q = np.concatenate([rows[rng.permutation(len(rows))[:400]],
                            np.sort(rng.integers(0, 150, size=(300, d + 1)), axis=1),
                            np.sort(rng.integers(140, 170, size=(20, d + 1)), axis=1)])
        monkeypatch.setattr(stm, "NATIVE_LOCATE_MIN", 10 ** 15)
        want = st._locate(d, q)
        monkeypatch.setattr(stm, "NATIVE_LOCATE_MIN", 1)
        got = st._locate(d, q)
        assert np.array_equal(got, want) and (got[:400] >= 0).all() and (got < 0).any()


def test_parallel_monotone_pass_equals_numpy(monkeypatch):
    """flooder_raise_dimension (one step of make_filtration_non_decreasing on all cores) against the numpy pass: NaN own
    values take the facets' maximum, NaN facet values do not take part, values already above their facets stay."""
    rng = np.random.default_rng(8)
    cells = qhull(rng.normal(size=(120, 5)))

    def tree():
        st = stm.SimplexTree.from_cells(cells, 120)
        r = np.random.default_rng(9)
        for d in range(6):
            v = r.random(len(st.simplices_of_dimension(d)))
            v[r.random(len(v)) < 0.3] = np.nan
            st._vals[d] = v
        st._cell_faces = {d: None for d in range(6)}          # (no cell -> face rows: the facets are located by key)
        return st

    a, b = tree(), tree()
    monkeypatch.setattr(stm, "NATIVE_RAISE_MIN", 10 ** 15)
    ca = a.make_filtration_non_decreasing()
    monkeypatch.setattr(stm, "NATIVE_RAISE_MIN", 1)
    cb = b.make_filtration_non_decreasing()
    assert ca == cb is True
    for d in range(6):
        assert np.array_equal(a._vals[d], b._vals[d], equal_nan=True)
    assert not b.make_filtration_non_decreasing()              # idempotent


def test_host_abi_from_plain_c(tmp_path):
    """examples/host_abi_example.c: the host entry points called from C, linked against libflooder_host.so - the same
    cell count as the Python binding gives for the same cloud."""
    import shutil
    import subprocess

    cc = shutil.which("gcc") or shutil.which("cc")
    if cc is None:
        pytest.skip("no C compiler")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "host_abi_example"
    lib = build.build_host()
    subprocess.run([cc, "-O2", f"-I{os.path.join(root, 'include')}", os.path.join(root, "examples", "host_abi_example.c"),
                    "-o", str(exe), lib, f"-Wl,-rpath,{os.path.dirname(lib)}"], check=True)
    out = subprocess.run([str(exe), "5", "160"], check=True, capture_output=True, text=True).stdout
    assert "dim 5, 160 points" in out and "faces of cell 0 found 20" in out
    n_cells = int(out.split(":")[1].split()[0])
    # the same xorshift cloud in numpy
    s, vals = 88172645463325252, []
    for _ in range(160 * 5):
        s ^= (s << 13) & 0xFFFFFFFFFFFFFFFF
        s ^= s >> 7
        s ^= (s << 17) & 0xFFFFFFFFFFFFFFFF
        vals.append(np.float32((s >> 11) / 9007199254740992.0 - 0.5))
    P = np.array(vals, dtype=np.float32).reshape(160, 5)
    assert len(qhull(P)) == n_cells


# ------------------------------------------------------------------------------------------------ gudhi's own output
VIS = ["virus", "coral", "lockwasher"]


def _vis(name):
    z = np.load(os.path.join(GOLDEN, f"docs_visualization_{name}.npz"))
    tets = stm._unique_rows(np.sort(z["tetrahedra"].astype(np.int64), axis=1))
    return z, z["landmarks"], tets


@pytest.mark.parametrize("name", VIS)
def test_native_3d_reproduces_gudhi_tetrahedra(name):
    """The reference commits three 1000-landmark clouds WITH the tetrahedra gudhi.DelaunayComplex (CGAL) gave for them
    (docs/visualization/*/{landmarks,tetrahedra}.csv, the data of docs/visualizations.md; fixtures:
    oracle/make_goldens.py visualization): 5601 / 6432 / 5813 tetrahedra.  Every native routine and the Qhull path
    reproduce them - as float32 landmarks (what flood_complex passes on the default path) through the incremental 3-D
    routine, and as the float64 values of the files."""
    z, L, tets = _vis(name)
    assert len(tets) == {"virus": 5601, "coral": 6432, "lockwasher": 5813}[name] and L.shape == (1000, 3)
    assert np.array_equal(qhull(L), tets)
    # float32 landmarks: the incremental routine takes them
    L32 = L.astype(np.float32)
    inc = stm._delaunay_native(L32.astype(np.float64))
    assert inc is not None and stm.LAST_DELAUNAY["routine"] == "incremental"
    assert np.array_equal(stm._unique_rows(np.sort(inc, axis=1)), tets)
    # the gift-wrapping routine on the same input, and on the float64 values
    assert np.array_equal(nd(L32), tets) and np.array_equal(nd(L), tets)
    # edges and triangles of the files are the faces of the tetrahedra
    for kind, d in (("edges", 1), ("triangles", 2)):
        want = stm._unique_rows(np.sort(z[kind].astype(np.int64), axis=1))
        assert np.array_equal(stm.faces_of_cells(tets, d, 1000)[0], want)


@pytest.mark.parametrize("name", VIS)
def test_float64_values_of_the_gudhi_clouds_decline_or_run_natively(name):
    """As float64 the decimal values of the files carry full 53-bit mantissas: `virus` scales to the incremental
    routine's 58-bit integer grid, `coral` and `lockwasher` do not - flooder_delaunay3d declines them with its RANGE
    code, and delaunay_cells hands them to the d-dimensional routine (a grid of up to 121 bits) instead of Qhull: native and exact
    either way, and gudhi's tetrahedra again."""
    _, L, tets = _vis(name)
    lib = host()
    out = np.empty((9000, 4), dtype=np.int32)
    rc = int(lib.flooder_delaunay3d(np.ascontiguousarray(L).ctypes.data, 1000, out.ctypes.data, 9000))
    if name == "virus":
        assert rc == len(tets)
    else:
        assert rc == stm.E_RANGE == E_BASE - 2
    cells = stm.delaunay_cells(L)
    assert stm.LAST_DELAUNAY["native"]
    assert stm.LAST_DELAUNAY["routine"] == ("incremental" if name == "virus" else "nd")
    assert np.array_equal(cells, tets)
