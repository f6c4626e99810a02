"""Host logic of flooder_amd on CPU: API surface, error behaviour, the CPU branch against the
reference's golden outputs, the simplex tree, and the C-ABI library's symbols."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import flooder_amd as fa
from flooder_amd import _native, core
from flooder_amd.simplex_tree import SimplexTree, delaunay_simplices
from helpers import GOLDEN, e2e_cases, load_e2e, dict_values

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("name", e2e_cases())
def test_cpu_branch_matches_reference(name):
    z, kw, keys = load_e2e(name)
    torch.manual_seed(int(z["weight_seed"]))
    fc = fa.flood_complex(torch.as_tensor(z["points"]), torch.as_tensor(z["landmarks"]), **kw)
    assert set(keys) == set(fc)
    assert np.abs(dict_values(fc, keys) - z["filtration_f32"]).max() < 5e-7


def test_cpu_branch_float64_matches_reference():
    z, kw, keys = load_e2e("torus3d_grid")
    with pytest.warns(RuntimeWarning):
        fc = fa.flood_complex(torch.as_tensor(z["points"]).double(), torch.as_tensor(z["landmarks"]).double(), **kw)
    assert np.abs(dict_values(fc, keys) - z["filtration_f64"]).max() < 1e-12


@pytest.mark.parametrize("n,dim", [(2, 1), (5, 2), (8, 3), (30, 3), (4, 4), (3, 6), (20, 2)])
def test_generate_grid_matches_reference(n, dim):
    g = np.load(os.path.join(GOLDEN, "grid_vectors.npz"))
    w, v, f = fa.generate_grid(n, dim, torch.device("cpu"), torch.float32)
    assert np.array_equal(w.numpy(), g[f"w_{n}_{dim}"])
    for k in range(dim + 1):
        assert np.array_equal(v[k].numpy(), g[f"v_{n}_{dim}_{k}"])
        assert np.array_equal(f[k].numpy(), g[f"f_{n}_{dim}_{k}"])


def test_uniform_weights_match_reference():
    g = np.load(os.path.join(GOLDEN, "grid_vectors.npz"))
    torch.manual_seed(42)
    w = fa.generate_uniform_weights(64, 3, torch.device("cpu"), torch.float32)
    assert np.array_equal(w.numpy(), g["u_64_3_seed42"])


def test_landmarks_int_and_return_tree():
    z, kw, keys = load_e2e("torus3d_grid")
    pts = torch.as_tensor(z["points"])
    st = fa.flood_complex(pts, 60, return_simplex_tree=True, **kw)  # FPS from index 0, as the golden
    vals = {tuple(s): f for s, f in st.get_simplices()}
    assert set(vals) == set(keys)
    assert np.abs(dict_values(vals, keys) - z["filtration_f32"]).max() < 5e-7
    for simplex, filtration in st.get_simplices():  # reference test_filtration_condition
        faces = list(st.get_boundaries(simplex))
        assert len(faces) == (len(simplex) if len(simplex) > 1 else 0)
        for _, ff in faces:
            assert ff <= filtration


def test_generate_landmarks_contract():
    torch.manual_seed(42)
    X = torch.rand(2000, 2)
    for n in (64, 256):
        L = fa.generate_landmarks(X, n)
        assert L.shape == (n, 2) and L.dtype == torch.float32 and L.device == X.device
    L = fa.generate_landmarks(X, 5000)
    assert L.shape == (2000, 2)  # clamped to the number of points (core.py:328)
    with pytest.raises(RuntimeError):
        fa.generate_landmarks(X, 0)
    L0 = fa.generate_landmarks(X, 10, start_idx=7)
    assert torch.equal(L0[0], X[7])


def test_error_behaviour():
    pts = torch.rand(100, 3)
    with pytest.raises(RuntimeError):
        fa.flood_complex(pts, torch.rand(10, 3).double())
    with pytest.raises(TypeError):
        fa.flood_complex(pts.half(), pts[:10].half())
    if torch.cuda.is_available():
        with pytest.raises(RuntimeError):
            fa.flood_complex(pts.cuda(), pts[:10])


def test_simplex_tree_basics():
    st = SimplexTree()
    assert st.insert([0, 1, 2], 1.0)
    assert st.num_simplices() == 7 and st.num_vertices() == 3 and st.dimension() == 2
    assert not st.insert([0, 1], 2.0)  # present: keeps min(old, new)
    assert st.filtration([0, 1]) == 1.0
    st.assign_filtration([0, 1], 3.0)
    assert st.filtration([0, 1]) == 3.0
    assert st.make_filtration_non_decreasing()
    assert st.filtration([0, 1, 2]) == 3.0
    order = [tuple(s) for s, _ in st.get_simplices()]
    assert order == [(0,), (0, 1), (0, 1, 2), (0, 2), (1,), (1, 2), (2,)]
    assert [tuple(s) for s, _ in st.get_boundaries([0, 1, 2])] == [(1, 2), (0, 2), (0, 1)]
    assert st.find([1, 2]) and not st.find([1, 3])


def test_delaunay_buckets_are_closed_complex():
    rng = np.random.default_rng(0)
    pts = rng.normal(size=(40, 3))
    dims = delaunay_simplices(pts)
    assert [a.shape[1] for a in dims] == [1, 2, 3, 4]
    st = SimplexTree.from_arrays(dims)
    for tet in dims[3][:20]:
        assert len(list(st.get_boundaries(tet))) == 4
    # Euler characteristic of a triangulated ball
    assert dims[0].shape[0] - dims[1].shape[0] + dims[2].shape[0] - dims[3].shape[0] == 1


def test_native_library_loads_and_exports_header_symbols():
    """The C-ABI shared library loads and exports every function include/flooder_hip.h declares."""
    import ctypes

    header = open(os.path.join(ROOT, "include", "flooder_hip.h")).read()
    declared = set(re.findall(r"\b(flooder_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    assert os.path.exists(_native.LIB_PATH), "libflooder_hip.so not built (python -m flooder_amd.build)"
    lib = ctypes.CDLL(_native.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
    assert declared == set(_native.SIGNATURES) | set(_native.POSITIONAL_SIGNATURES), "ctypes binding and header disagree"
    # what flooder_amd itself calls takes at most a dozen arguments on the default path (2-D / 3-D fused sweep, the
    # sorted-sample sweep above 3-D, the batched landmark selection: parameter blocks bound by field name); the longer
    # positional signatures left in SIGNATURES belong to method="ball", mode="blocks", the unfused sweeps and the
    # one-per-launch selection
    long_ones = {k for k, (_, a) in _native.SIGNATURES.items() if len(a) > 12}
    assert long_ones == {"flooder_ball_fill_f32", "flooder_sweep_f32", "flooder_box_select_f32", "flooder_sweep_bvh_f32",
                         "flooder_sweep_bvh_items_f32", "flooder_sweep_cell_f32", "flooder_fps_indexed_f32"}
    assert lib.flooder_abi_version() == 1
    assert [lib.flooder_padded_dim(d) for d in (1, 2, 3, 4, 5, 8)] == [2, 2, 4, 4, 8, 8]


def test_parameter_blocks_have_the_layout_of_the_header(tmp_path):
    """The ctypes Structures of ``_native`` against the C structs of include/flooder_hip.h: same size, every field at
    the same offset (compiled with the host C compiler; a maintainer's cgo / JNA binding is checked the same way)."""
    import shutil
    import subprocess

    cc = shutil.which("gcc") or shutil.which("cc")
    if cc is None:
        pytest.skip("no C compiler")
    blocks = {"flooder_fused_sweep_t": _native.FusedSweep, "flooder_sorted_sweep_t": _native.SortedSweep,
              "flooder_fps_batched_t": _native.FpsBatched}
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{os.path.join(ROOT, "include", "flooder_hip.h")}"',
             'int main(void) {']
    for cname, cls in blocks.items():
        lines.append(f'printf("{cname} sizeof %zu\\n", sizeof({cname}));')
        for fname, _ in cls._fields_:
            lines.append(f'printf("{cname} {fname} %zu\\n", offsetof({cname}, {fname}));')
    lines += ['return 0; }']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run([cc, "-o", str(exe), str(src)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split("\n")
    seen = 0
    for line in out:
        if not line:
            continue
        cname, what, val = line.split()
        cls = blocks[cname]
        assert int(val) == (ctypes.sizeof(cls) if what == "sizeof" else getattr(cls, what).offset), line
        seen += 1
    assert seen == sum(len(c._fields_) + 1 for c in blocks.values())
    blk = _native.FusedSweep(n_pts=5, alpha=0.5)
    assert blk.size == ctypes.sizeof(_native.FusedSweep) and blk.abi == 1 and blk.n_pts == 5 and not blk.top
    with pytest.raises(TypeError):
        _native.FusedSweep(no_such_field=1)


def test_host_libraries_export_header_symbols():
    """include/flooder_host.h: the host-side entry points (Delaunay of the landmarks, persistence, dict hand-off) are
    exported by libflooder_host.so / libflooder_py.so."""
    import ctypes

    from flooder_amd import build

    header = open(os.path.join(ROOT, "include", "flooder_host.h")).read()
    declared = set(re.findall(r"\b(flooder_[a-z0-9_]+)\s*\(", header))
    assert declared == {"flooder_delaunay3d", "flooder_delaunay2d", "flooder_delaunay3d_local_edges", "flooder_persistence_z2",
                        "flooder_filtration_order", "flooder_dict_update", "flooder_delaunay_nd", "flooder_host_free",
                        "flooder_delaunay_nd_stat", "flooder_delaunay_nd_isa", "flooder_cell_faces", "flooder_widen_i32", "flooder_raise_dimension", "flooder_locate_rows"}
    build.build_host()
    host = ctypes.CDLL(build.HOST_LIB)
    assert all(hasattr(host, f) for f in declared - {"flooder_dict_update"})
    if build.build_py():
        assert hasattr(ctypes.PyDLL(build.PY_LIB), "flooder_dict_update")


def test_rocm_tensor_without_library_fails_loudly(monkeypatch):
    """No silent fallback: a missing library is an ImportError for GPU inputs."""
    monkeypatch.setattr(_native, "_lib", None)
    monkeypatch.setattr(_native, "_load_error", None)
    monkeypatch.setattr(_native, "LIB_PATH", "/nonexistent/libflooder_hip.so")
    with pytest.raises(ImportError):
        _native.load()
    with pytest.raises(ImportError):
        fa.flood_complex(torch.rand(10, 2), torch.rand(4, 2), use_triton=True)


def test_edge_cases_cpu():
    # empty cloud: a clean error
    with pytest.raises(RuntimeError):
        fa.flood_complex(torch.zeros((0, 3)), 5)
    # 1-D ambient space: the "Delaunay complex" is the chain of sorted landmarks
    torch.manual_seed(0)
    x = torch.rand(200, 1)
    fc = fa.flood_complex(x, 12, points_per_edge=9)
    ref = fo_complex(x.numpy(), 12, points_per_edge=9)
    assert set(fc) == set(ref)
    assert max(abs(fc[k] - ref[k]) for k in ref) < 1e-6
    assert sum(len(k) == 2 for k in fc) == 11
    # a single simplex (landmarks == dim + 1 points)
    pts = torch.rand(50, 2)
    fc = fa.flood_complex(pts, pts[:3].clone(), points_per_edge=5)
    assert set(fc) == {(0,), (1,), (2,), (0, 1), (0, 2), (1, 2), (0, 1, 2)}
    assert fc[(0,)] == 0.0 and fc[(0, 1, 2)] >= max(fc[(0, 1)], fc[(0, 2)], fc[(1, 2)])


def fo_complex(pts, n_lms, **kw):
    from oracle import flood_oracle as fo
    lms = pts[fo.exact_fps(pts, n_lms, 0)]
    return fo.flood_complex_oracle(pts, lms, **kw)


def test_sample_order_gives_compact_chunks():
    """core.sample_order: a permutation whose aligned runs of 256 / 64 rows are compact patches of the simplex
    (the cell sweep's chunks and the tree sweep's tiles): far tighter than the grid's own order or a Morton curve."""
    import torch
    from flooder_amd import core

    w, _, _ = core.generate_grid(30, 3, torch.device("cpu"), torch.float32)
    order = core.sample_order(w)
    R = w.shape[0]
    assert sorted(order.tolist()) == list(range(R))
    corners = np.eye(4) - 0.25
    X = w.numpy().astype(np.float64) @ corners          # regular-simplex coordinates (hyperplane in R^4)

    def box_volumes(perm, chunk):
        out = []
        for a in range(0, R, chunk):
            P = X[perm[a:a + chunk]]
            e = np.sort(P.max(0) - P.min(0))[1:]        # the three in-plane extents dominate; drop the smallest
            out.append(float(np.prod(e + 0.03)))
        return np.array(out)

    ident = np.arange(R)
    for chunk in (256, 64):
        v_new, v_id = box_volumes(order, chunk), box_volumes(ident, chunk)
        assert v_new.sum() < 0.8 * v_id.sum()
        assert v_new.max() <= v_id.max()
    # small tables are left alone; edges and triangles get valid permutations too
    for ppe, d in ((30, 1), (5, 2), (30, 2), (8, 3)):
        w, _, _ = core.generate_grid(ppe, d, torch.device("cpu"), torch.float32)
        o = core.sample_order(w)
        assert sorted(o.tolist()) == list(range(w.shape[0]))
        if w.shape[0] <= 64:
            assert o.tolist() == list(range(w.shape[0]))


def test_lazy_tree_from_cells_equals_eager_tree():
    """SimplexTree.from_cells (face tables enumerated on first use, cell -> face row index) against from_arrays:
    same complex, same values after bulk assignment + monotone pass, whatever is enumerated late."""
    import itertools
    from flooder_amd.simplex_tree import SimplexTree, delaunay_cells, delaunay_simplices, faces_of_cells
    rng = np.random.default_rng(3)
    for dim, n in ((2, 60), (3, 80), (4, 40)):
        P = rng.normal(size=(n, dim))
        cells = delaunay_cells(P)
        tables = delaunay_simplices(P)
        eager = SimplexTree.from_arrays(tables)
        lazy = SimplexTree.from_cells(cells, n, eager=1)
        assert lazy.dimension() == dim
        # index: face j of cell c
        for d in range(dim + 1):
            table, index = faces_of_cells(cells, d, n)
            assert np.array_equal(table, tables[d])
            combos = list(itertools.combinations(range(dim + 1), d + 1))
            for j, cmb in enumerate(combos):
                assert np.array_equal(table[index[:, j]], cells[:, list(cmb)])
        # assign values to edges and vertices only, then monotone: higher tables inherit (some enumerated late)
        e = tables[1]
        ev = rng.random(e.shape[0])
        for st in (eager, lazy):
            st.assign_filtration_bulk(tables[0], np.zeros(n))
            st.assign_filtration_bulk(e, ev)
            st.make_filtration_non_decreasing()
        assert lazy._lazy                      # dimensions >= 2 still not enumerated
        for d in range(dim + 1):
            assert np.array_equal(lazy.simplices_of_dimension(d), eager.simplices_of_dimension(d))
            assert np.allclose(lazy.filtrations_of_dimension(d), eager.filtrations_of_dimension(d), equal_nan=True)
        assert lazy.to_dict() == eager.to_dict()
        assert lazy.num_simplices() == eager.num_simplices()
        # indexed assignment == located assignment
        a = SimplexTree.from_cells(cells, n)
        b = SimplexTree.from_cells(cells, n)
        vals = rng.random((cells.shape[0], dim + 1))
        combos = list(itertools.combinations(range(dim + 1), dim))      # facets
        assert a.assign_cell_faces(dim - 1, np.arange(cells.shape[0]), list(range(dim + 1)), vals)
        for j, cmb in enumerate(combos):
            b.assign_filtration_bulk(cells[:, list(cmb)], vals[:, j])
        assert np.array_equal(np.isnan(a.filtrations_of_dimension(dim - 1)), np.isnan(b.filtrations_of_dimension(dim - 1)))


def test_witness_plan_tables():
    """Coarse level of the witness sweep: valid slots, a coarse row first among its own parents, the lattice rule's
    count at the reference's default lattice, farthest-point samples for random weights, nothing for small tables."""
    from flooder_amd import core

    w, _, _ = core.generate_grid(30, 3, "cpu", torch.float32)
    perm = core.sample_order(w)
    rows, par, n_c = core.witness_plan(w, perm)
    assert n_c == 242 and rows.shape == (core.WIT_MAX_COARSE,) and par.shape == (w.shape[0],)
    assert (rows[:n_c] >= 0).all() and (rows[n_c:] == -1).all() and len(set(rows[:n_c].tolist())) == n_c
    slots = np.stack([(par >> (8 * j)) & 0xFF for j in range(4)], axis=1)
    assert (slots < n_c).all()
    assert (slots[rows[:n_c], 0] == np.arange(n_c)).all()
    # every face of the simplex holds coarse samples (its running maximum needs exact values of its own)
    wp = w.numpy()[perm]
    for face in [(0,), (1, 2), (0, 1, 3), (0, 1, 2, 3)]:
        others = [j for j in range(4) if j not in face]
        on_face = (wp[rows[:n_c]][:, others] == 0).all(axis=1) & (wp[rows[:n_c]][:, list(face)] > 0).all(axis=1)
        assert on_face.any(), face
    torch.manual_seed(0)
    wr = core.generate_uniform_weights(3000, 3, "cpu", torch.float32)
    rows_r, par_r, n_r = core.witness_plan(wr, core.sample_order(wr))
    assert 16 <= n_r <= core.WIT_MAX_COARSE and (((par_r >> 24) & 0xFF) < n_r).all()
    small, _, _ = core.generate_grid(8, 3, "cpu", torch.float32)
    assert core.witness_plan(small, core.sample_order(small)) is None


def test_gudhi_branches_with_a_stub_module(monkeypatch):
    """``_build_complex`` and the hand-off of ``flood_complex`` take the reference's own route when gudhi is importable
    (``gudhi.DelaunayComplex(...).create_simplex_tree()``, ``assign_filtration`` per simplex,
    ``make_filtration_non_decreasing``, ``get_simplices``: core.py:130-138, 278-288).  gudhi is absent from the build
    image, so a stub with that call surface (Qhull underneath) stands in: the result must be the one of the
    array-backed tree."""
    import itertools
    import sys
    import types

    from scipy.spatial import Delaunay

    import flooder_amd as fa
    from flooder_amd import core

    class StubTree:
        def __init__(self, simplices):
            self.f = {s: float("nan") for s in simplices}
            self.assigned = 0

        def get_simplices(self):
            for s in sorted(self.f, key=lambda t: (len(t), t)):
                yield list(s), self.f[s]

        def assign_filtration(self, simplex, value):
            self.f[tuple(simplex)] = float(value)
            self.assigned += 1

        def make_filtration_non_decreasing(self):
            for s in sorted(self.f, key=len):
                for j in range(len(s)):
                    if len(s) > 1:
                        self.f[s] = max(self.f[s], self.f[s[:j] + s[j + 1:]])

    class StubDelaunay:
        def __init__(self, points):
            cells = Delaunay(np.asarray(points, dtype=np.float64)).simplices
            self.simplices = set()
            for c in cells:
                c = tuple(sorted(int(v) for v in c))
                for k in range(1, len(c) + 1):
                    self.simplices.update(itertools.combinations(c, k))

        def create_simplex_tree(self):
            return StubTree(self.simplices)

    torch.manual_seed(3)
    pts = torch.randn(3000, 3)
    lms = fa.generate_landmarks(pts, 40, start_idx=0)
    want = fa.flood_complex(pts, lms, points_per_edge=6)
    stub = types.ModuleType("gudhi")
    stub.DelaunayComplex = StubDelaunay
    monkeypatch.setitem(sys.modules, "gudhi", stub)
    monkeypatch.setattr(core, "HAS_GUDHI", True)
    got = fa.flood_complex(pts, lms, points_per_edge=6)
    assert set(got) == set(want)
    assert all(abs(got[k] - want[k]) <= 1e-12 for k in want)
    tree = fa.flood_complex(pts, lms, points_per_edge=6, return_simplex_tree=True)
    assert isinstance(tree, StubTree) and tree.assigned >= len(want)   # (a face is assigned from every simplex that holds it, as in the reference)
    torch.manual_seed(1)
    got_r = fa.flood_complex(pts, lms, num_rand=50)
    monkeypatch.setattr(core, "HAS_GUDHI", False)
    torch.manual_seed(1)
    want_r = fa.flood_complex(pts, lms, num_rand=50)
    assert set(got_r) == set(want_r) and all(abs(got_r[k] - want_r[k]) <= 1e-12 for k in want_r)


def test_index_source_version_of_inference_tensors():
    """Tensors created under ``torch.inference_mode()`` keep no version counter (``_version`` raises): the staleness
    check of ``index=`` is skipped for them instead of crashing every GPU call."""
    from flooder_amd import core

    with torch.inference_mode():
        t = torch.randn(8, 3)
    assert core._tensor_version(t) is None
    u = torch.randn(8, 3)
    v0 = core._tensor_version(u)
    u.add_(1.0)
    assert core._tensor_version(u) == v0 + 1


def test_rows_are_subset_is_bit_exact():
    from flooder_amd import core

    g = torch.Generator().manual_seed(0)
    p = torch.randn(5000, 3, generator=g)
    assert core._rows_are_subset(p[::97], p)
    assert core._rows_are_subset(p[:0], p)
    q = p[::97].clone()
    q[3, 1] = torch.nextafter(q[3, 1], torch.tensor(10.0))
    assert not core._rows_are_subset(q, p)
    z = torch.zeros(1, 3)
    assert core._rows_are_subset(-z, torch.cat([p, z]))   # (-0.0 is the row 0.0)


def test_sample_plan_builds_its_host_side_tables_on_demand(monkeypatch):
    """The witness plan and the pilot rows cost host work (cKDTree, a device sync) and one sweep each reads them:
    nothing is built until asked for (a ``num_rand`` call never asks)."""
    from flooder_amd import core

    calls = []
    orig = core.witness_plan
    monkeypatch.setattr(core, "witness_plan", lambda w, perm: calls.append(1) or orig(w, perm))
    w, _, fi = core.generate_grid(14, 3, torch.device("cpu"), torch.float32)
    plan = core.SamplePlan(w, core._FaceTable(fi, w.shape[0], torch.device("cpu")))
    assert calls == [] and plan._late_rows is None
    assert plan.wit is not None and calls == [1]
    assert plan.wit is not None and calls == [1]          # (cached)
    late = plan.late_rows
    assert late is not None and int((late == 0).sum()) <= plan.faces.n_faces


def test_to_dict_in_c_equals_the_python_loop():
    """``SimplexTree.to_dict`` builds the tuples in one C pass (csrc/pyhandoff.c through ctypes.PyDLL): same keys, same
    Python types, same values as the .tolist() + zip loop it replaces."""
    from flooder_amd import simplex_tree as stm

    rng = np.random.default_rng(5)
    st = stm.SimplexTree.from_cells(stm.delaunay_cells(rng.normal(size=(300, 3))), 300)
    st._materialise_all()
    for d in st._rows:
        st._vals[d][:] = rng.random(st._rows[d].shape[0])
    if stm._py_lib() is None:
        pytest.skip("no C compiler / Python headers here: the Python loop is the only path")
    a = st.to_dict()
    keep = stm._PY_LIB
    stm._PY_LIB = None
    try:
        b = st.to_dict()
    finally:
        stm._PY_LIB = keep
    assert a == b and list(a) == list(b)
    k = next(iter(a))
    assert type(k) is tuple and type(k[0]) is int and type(a[k]) is float


def test_diagnostic_builds_still_compile():
    """The timer / trace variants the tools build on the GPU box (-DFLOODER_PHASE_TIMERS, _WAVE_END, _WAVE_END_FIN,
    _SORTED_TIMERS, _WIT_TIMERS) are not part of the product build: a syntax-only device compile of each keeps them
    from rotting (a renamed parameter had broken the sorted sweep's timer build unnoticed)."""
    import shutil
    import subprocess
    from concurrent.futures import ThreadPoolExecutor

    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc here")
    csrc = os.path.join(ROOT, "flooder_amd", "csrc")
    jobs = [("flood_cell.hip", "-DFLOODER_PHASE_TIMERS"), ("flood_cell.hip", "-DFLOODER_WAVE_END"),
            ("flood_finish.hip", "-DFLOODER_PHASE_TIMERS"), ("flood_finish.hip", "-DFLOODER_WAVE_END_FIN"),
            ("flood_sorted.hip", "-DFLOODER_SORTED_TIMERS"), ("flood_wit.hip", "-DFLOODER_WIT_TIMERS")]

    def compile_one(job):
        src, flag = job
        p = subprocess.run([hipcc, "-std=c++17", "--offload-arch=gfx950", "-DFLOODER_BUILD", flag, "--cuda-device-only",
                            "-fsyntax-only", os.path.join(csrc, src)], capture_output=True, text=True)
        return job, p.returncode, p.stderr[-600:]

    with ThreadPoolExecutor(max_workers=6) as pool:
        for job, rc, err in pool.map(compile_one, jobs):
            assert rc == 0, f"{job}: {err}"


def test_shard_slot_fill_marks_the_faces_a_rank_does_not_have():
    """Simplex shards of the cell sweep keep one word per distinct face of the complex (core.shared_face_slots):
    what a rank contributes to all_reduce(MIN) is +inf in the words none of its simplices touches
    (core.shard_slot_fill) - the minimum over the ranks is then the value of whichever rank has the face."""
    import torch

    from flooder_amd import core

    slot_all = torch.tensor([[0, 1, 2], [1, 3, 4], [2, 4, 5], [6, 7, 8]], dtype=torch.int32)
    values_true = torch.arange(1, 10, dtype=torch.float32)
    combined = torch.full((9,), float("inf"))
    for rank in range(2):
        mine = slot_all[rank::2]
        fill = core.shard_slot_fill(mine, 9)
        touched = torch.zeros(9, dtype=torch.bool)
        touched[mine.reshape(-1).long()] = True
        assert torch.equal(torch.isinf(fill), ~touched) and torch.all(fill[touched] == 0)
        part = torch.where(touched, values_true, torch.zeros(9))      # what the sweep leaves: zeros where untouched
        combined = torch.minimum(combined, torch.maximum(part, fill))
    assert torch.equal(combined, values_true)
    assert torch.all(torch.isinf(core.shard_slot_fill(slot_all[:0], 9)))   # a rank without simplices


def test_built_kernels_hold_no_vector_exec_writes_before_dpp_asm():
    """csrc/flood_common.hpp writes the wave reductions as DPP instructions inside asm blocks and places the wait states
    a DPP read needs itself.  One hazard it does not cover: five wait states after a VECTOR instruction that writes
    EXEC (v_cmpx).  hipcc's wave64 control flow uses scalar EXEC writes only; if a compiler ever emits v_cmpx into
    these kernels this test says so.  Also: the asm blocks are really there (v_min_f32_dpp)."""
    import shutil
    import subprocess
    import tempfile

    from flooder_amd import build

    lib = build.HIP_LIB if hasattr(build, "HIP_LIB") else os.path.join(ROOT, "flooder_amd", "libflooder_hip.so")
    bundler = "/opt/rocm/lib/llvm/bin/clang-offload-bundler"
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not (os.path.exists(lib) and os.path.exists(bundler) and os.path.exists(objdump)):
        pytest.skip("no built library or no LLVM tools here")
    with tempfile.TemporaryDirectory() as tmp:
        # the shared object embeds the gfx950 code object in its .hip_fatbin section
        fat = os.path.join(tmp, "fat.bin")
        objcopy = "/opt/rocm/lib/llvm/bin/llvm-objcopy"
        if not os.path.exists(objcopy):
            pytest.skip("no llvm-objcopy")
        subprocess.run([objcopy, "--dump-section", f".hip_fatbin={fat}", lib, os.path.join(tmp, "unused.so")], check=True)
        # (one offload bundle per translation unit, back to back)
        blob = open(fat, "rb").read()
        magic = b"__CLANG_OFFLOAD_BUNDLE__"
        starts = [i for i in range(len(blob)) if blob.startswith(magic, i)]
        assert len(starts) >= 10, "one bundle per .hip source expected"
        asm = ""
        for n, a in enumerate(starts):
            part = os.path.join(tmp, f"b{n}.bin")
            open(part, "wb").write(blob[a:starts[n + 1] if n + 1 < len(starts) else len(blob)])
            co = os.path.join(tmp, f"dev{n}.co")
            p = subprocess.run([bundler, "--unbundle", "--type=o", f"--input={part}",
                                "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], capture_output=True, text=True)
            if p.returncode != 0 or not os.path.exists(co) or os.path.getsize(co) == 0:
                pytest.skip("could not unbundle the device code: " + p.stderr[-200:])
            asm += subprocess.run([objdump, "-d", co], capture_output=True, text=True, check=True).stdout
    assert "v_min_f32_dpp" in asm and "v_max_u32_dpp" in asm and "v_add_u32_dpp" in asm
    assert "v_cmpx" not in asm
