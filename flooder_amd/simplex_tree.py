"""Array-backed simplex tree and Delaunay hand-off for the Flood complex.

The reference hands its result to ``gudhi.SimplexTree`` (reference
``flooder/core.py:130-138`` builds it from ``gudhi.DelaunayComplex`` and
``core.py:278-288`` assigns filtration values, monotonises and returns either the
tree or ``dict(stree.get_simplices())``).  gudhi is a third-party C++/CGAL wheel
that is not part of the reference repository; where it is importable the product
uses it, otherwise this module provides the subset of its interface the path
needs, with the same names and argument meaning:

``SimplexTree``: ``insert``, ``assign_filtration``, ``filtration``, ``find``,
``get_simplices``, ``get_skeleton``, ``get_boundaries``, ``num_simplices``,
``num_vertices``, ``dimension``, ``make_filtration_non_decreasing``,
``compute_persistence``, ``persistence``, ``persistence_intervals_in_dimension``.

``DelaunayComplex(points).create_simplex_tree()``: Delaunay triangulation through
Qhull (``scipy.spatial.Delaunay``); SURVEY.md section 4 records that Qhull and gudhi/CGAL
give identical simplices on the reference's three committed 1000-landmark clouds.

Storage is one sorted ``(n_d, d+1)`` int64 array plus one float64 value array per
dimension, so the bulk operations the hot path uses (assign all values of one
dimension, monotonise) are vectorised instead of one Python call per simplex.
"""

from __future__ import annotations

import itertools
import os
import math
from typing import Dict, Iterable, Iterator, List, Optional, Sequence, Tuple

import numpy as np

__all__ = ["SimplexTree", "DelaunayComplex", "delaunay_simplices", "delaunay_cells", "faces_of_cells", "HAS_GUDHI"]

try:  # pragma: no cover - gudhi is absent from the build image
    import gudhi as _gudhi  # type: ignore

    HAS_GUDHI = True
except Exception:  # pragma: no cover
    _gudhi = None
    HAS_GUDHI = False


INDEX_MAX_FACES = 8_000_000  # cell -> face row indices are kept for tables enumerated from at most this many faces


_PY_LIB = False   # False: not looked for yet; None: not available


def _py_lib():
    """``libflooder_py.so`` (CPython-API helpers of the hand-off, ``csrc/pyhandoff.c``) through ``ctypes.PyDLL``, or
    None where it cannot be built (no compiler / no Python headers: the Python loop does the same)."""
    global _PY_LIB
    if _PY_LIB is False:
        _PY_LIB = None
        try:
            import ctypes

            from . import build

            path = build.build_py()
            if path:
                lib = ctypes.PyDLL(path)
                lib.flooder_dict_update.restype = ctypes.c_int
                lib.flooder_dict_update.argtypes = [ctypes.py_object, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int,
                                                    ctypes.c_void_p, ctypes.py_object]
                _PY_LIB = lib
        except Exception:
            _PY_LIB = None
    return _PY_LIB


def _lex_sort_rows(rows: np.ndarray) -> np.ndarray:
    """Order that sorts integer rows lexicographically (first column most significant)."""
    if rows.shape[0] == 0:
        return np.zeros((0,), dtype=np.int64)
    return np.lexsort(rows.T[::-1])


def _unique_rows(rows: np.ndarray) -> np.ndarray:
    if rows.shape[0] == 0:
        return rows
    base = int(rows.max()) + 1
    k = rows.shape[1]
    if base ** k < 2 ** 62:  # pack each row into one int64 key: a 1-D sort instead of a lexsort
        mult = base ** np.arange(k - 1, -1, -1, dtype=np.int64)
        keys = np.unique(rows @ mult)
        out = np.empty((keys.shape[0], k), dtype=np.int64)
        for j in range(k - 1, -1, -1):
            out[:, j] = keys % base
            keys = keys // base
        return out
    rows = rows[_lex_sort_rows(rows)]
    keep = np.ones(rows.shape[0], dtype=bool)
    keep[1:] = np.any(rows[1:] != rows[:-1], axis=1)
    return rows[keep]


def delaunay_cells(points: np.ndarray) -> np.ndarray:
    """Top-dimensional cells of the Delaunay triangulation of ``points``: unique rows of ascending vertex
    ids, in lexicographic order (Qhull; what ``gudhi.DelaunayComplex`` triangulates with CGAL)."""
    from scipy.spatial import Delaunay

    points = np.ascontiguousarray(points, dtype=np.float64)
    n, dim = points.shape
    if dim == 1:
        order = np.argsort(points[:, 0], kind="stable")
        cells = np.stack([order[:-1], order[1:]], axis=1) if n > 1 else np.zeros((0, 2), np.int64)
    elif n <= dim:
        # fewer points than a full simplex: a single (n-1)-simplex on all points
        cells = np.arange(n, dtype=np.int64)[None, :]
    else:
        cells = _delaunay_native(points) if (2 <= dim <= 8 and NATIVE_DELAUNAY and n >= dim + 2) else None
        if cells is not None and dim > 3:
            return cells      # (flooder_delaunay_nd: distinct rows of ascending ids, already in lexicographic order)
        if cells is None:
            cells = Delaunay(points).simplices
    return _unique_rows(np.sort(np.asarray(cells, dtype=np.int64), axis=1))


NATIVE_DELAUNAY = True   # csrc/delaunay2d.cpp, delaunay3d.cpp, delaunay_nd.cpp (4 .. 8 dimensions, all host cores) in libflooder_host.so instead of Qhull; falls back to Qhull where they decline
HOST_THREADS_MAX = 24    # measured on the 2 x 64-core host of the MI355X box: 16 - 32 threads are the optimum (345 - 390 ms for cfg 4's Delaunay; 8: 660, 48: 480, 64: 580, 128: 890 - the barriers of the sub-rounds and the tables' cache lines cost more than the extra cores give)
DELAUNAY_THREADS = 0     # threads of the 4 .. 8-dimensional routine: 0 = the CPUs of this process's share (see _host_threads)
LAST_DELAUNAY = {"native": False, "code": 0}
_HOST_DT = False


def _host_threads() -> int:
    """Threads for the host-parallel Delaunay routine: ``DELAUNAY_THREADS`` / FLOODER_HOST_THREADS if set, else the
    CPUs this process may run on divided by the ranks of this node (LOCAL_WORLD_SIZE: one process per GPU all
    triangulating the same landmarks at once), at most ``HOST_THREADS_MAX``."""
    import os

    if DELAUNAY_THREADS > 0:
        return int(DELAUNAY_THREADS)
    env = os.environ.get("FLOODER_HOST_THREADS", "")
    if env.isdigit() and int(env) > 0:
        return int(env)
    try:
        cpus = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        cpus = os.cpu_count() or 1
    local = os.environ.get("LOCAL_WORLD_SIZE", "1")
    local = int(local) if local.isdigit() and int(local) > 0 else 1
    return max(1, min(HOST_THREADS_MAX, cpus // local))


def _take_rows(lib, out, count: int, width: int, threads: int) -> np.ndarray:
    """The library's malloc'ed (count, width) int32 table as an int64 array (widened on all cores), buffer released."""
    import ctypes

    try:
        rows = np.empty((count, width), dtype=np.int64)
        lib.flooder_widen_i32(ctypes.cast(out, ctypes.c_void_p), count * width, rows.ctypes.data, threads)
        return rows
    finally:
        lib.flooder_host_free(out)


def _load_host():
    """``libflooder_host.so`` with the Delaunay / face-table entry points bound (built on first use), or None."""
    global _HOST_DT
    if _HOST_DT is False:
        _HOST_DT = None
        try:
            import ctypes

            from . import build

            lib = ctypes.CDLL(build.build_host())
            for f in (lib.flooder_delaunay2d, lib.flooder_delaunay3d):
                f.restype = ctypes.c_int64
                f.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64]
            lib.flooder_delaunay_nd.restype = ctypes.c_int64
            lib.flooder_delaunay_nd.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int,
                                                ctypes.POINTER(ctypes.POINTER(ctypes.c_int32))]
            lib.flooder_host_free.restype = None
            lib.flooder_host_free.argtypes = [ctypes.c_void_p]
            lib.flooder_delaunay_nd_stat.restype = ctypes.c_long
            lib.flooder_delaunay_nd_stat.argtypes = [ctypes.c_int]
            lib.flooder_widen_i32.restype = None
            lib.flooder_widen_i32.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int]
            lib.flooder_locate_rows.restype = ctypes.c_int64
            lib.flooder_locate_rows.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_int64,
                                                ctypes.c_int64, ctypes.c_void_p, ctypes.c_int]
            lib.flooder_raise_dimension.restype = ctypes.c_int64
            lib.flooder_raise_dimension.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p,
                                                    ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                                    ctypes.c_int]
            lib.flooder_cell_faces.restype = ctypes.c_int64
            lib.flooder_cell_faces.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int64,
                                               ctypes.c_int, ctypes.POINTER(ctypes.POINTER(ctypes.c_int32))]
            _HOST_DT = lib
        except Exception:
            _HOST_DT = None
    return _HOST_DT


def _delaunay_native(points: np.ndarray) -> Optional[np.ndarray]:
    """Cells of the Delaunay triangulation from ``flooder_delaunay2d`` / ``flooder_delaunay3d`` (incremental insertion
    with exact predicates, host C++) or, in 4 .. 8 dimensions, ``flooder_delaunay_nd`` (gift wrapping over the facets on
    all host cores, exact where floating point cannot decide), or None where the routine declines the input
    (duplicate points, all points collinear / coplanar, coordinates that do not scale to 58-bit (121-bit: nd) integers
    - float64 clouds with a wide exponent range -, above 3-D also an exact tie: cospherical points) or the library is
    not there: the caller then asks Qhull as before."""
    _load_host()
    LAST_DELAUNAY["native"] = False
    if _HOST_DT is None:
        return None
    pts = np.ascontiguousarray(points, dtype=np.float64)
    n, dim = pts.shape
    if dim > 3:
        return _delaunay_nd(pts)
    fn = _HOST_DT.flooder_delaunay3d if dim == 3 else _HOST_DT.flooder_delaunay2d
    # (n points in general position: ~6.8 n tetrahedra, at most 2 n triangles)
    cap = 8 * n + 64 if dim == 3 else 2 * n + 16
    for _ in range(2):
        out = np.empty((cap, dim + 1), dtype=np.int32)
        rc = int(fn(pts.ctypes.data, n, out.ctypes.data, cap))
        if rc >= 0:
            LAST_DELAUNAY.update(native=True, code=rc, routine="incremental")
            return out[:rc].astype(np.int64)
        if rc < -(1 << 40):   # declined
            LAST_DELAUNAY["code"] = rc
            if rc == E_RANGE:
                # coordinates that do not scale to 58-bit integers (float64 values with full mantissas): the
                # d-dimensional routine works on a grid of up to 121 bits with multi-word integers - slower, exact all the same
                cells = _delaunay_nd(pts)
                return None if cells is None else _unique_rows(np.sort(cells, axis=1))
            return None
        cap = -rc
    return None


E_RANGE = -(1 << 40) - 2     # csrc/exact_int.hpp: the coordinates do not fit the routine's integer grid


# Gift wrapping scans all n points per pivot: its work grows with n x (number of simplices) ~ n^2 in low dimensions, where
# Qhull's incremental hull grows with n log n.  Measured in the build container (8 threads): 4-D, 30 000 points 5.6 s
# against Qhull's 6.4 s - beyond these sizes Qhull is asked instead (6-D and up: Qhull's own cost explodes first).
ND_MAX_POINTS = {2: 20_000, 3: 20_000, 4: 50_000, 5: 50_000}


def _delaunay_nd(pts: np.ndarray) -> Optional[np.ndarray]:
    """``flooder_delaunay_nd``: (count, dim + 1) int64 rows of ascending ids in lexicographic order, or None."""
    import ctypes

    n, dim = pts.shape
    if n > ND_MAX_POINTS.get(dim, 1 << 30):
        LAST_DELAUNAY["code"] = E_RANGE
        return None
    out = ctypes.POINTER(ctypes.c_int32)()
    threads = _host_threads()
    rc = int(_HOST_DT.flooder_delaunay_nd(pts.ctypes.data, n, dim, threads, ctypes.byref(out)))
    if rc < 0:
        LAST_DELAUNAY["code"] = rc
        return None
    cells = _take_rows(_HOST_DT, out, rc, dim + 1, threads)
    LAST_DELAUNAY.update(native=True, code=rc, threads=threads, routine="nd",
                         exact_calls=int(_HOST_DT.flooder_delaunay_nd_stat(0)))
    return cells


_delaunay3d_native = _delaunay_native   # (name of the round-5 3-D entry point, kept for tools and tests)

_C32_CACHE: list = [None, None]
NATIVE_LOCATE_MIN = 500_000    # query entries from which SimplexTree._locate runs in flooder_locate_rows
NATIVE_RAISE_MIN = 1_000_000   # facets (rows x (d + 1)) from which the monotone pass runs in flooder_raise_dimension
NATIVE_FACES_MIN = 2_000_000   # faces (cells x combinations) from which the table is enumerated by flooder_cell_faces


def _faces_native(cells: np.ndarray, d: int, n_points: int) -> Optional[np.ndarray]:
    """Sorted table of the distinct d-faces of ``cells`` from ``flooder_cell_faces`` (all host cores), or None where
    the packed keys would not fit / the library is not there."""
    import ctypes

    lib = _load_host() if NATIVE_DELAUNAY else None
    if lib is None:
        return None
    import weakref

    if _C32_CACHE[0] is not None and _C32_CACHE[0]() is cells:   # (several dimensions' tables from the same cells)
        c32 = _C32_CACHE[1]
    else:
        c32 = np.ascontiguousarray(cells, dtype=np.int32)
        try:
            _C32_CACHE[:] = [weakref.ref(cells, lambda _r: _C32_CACHE.__setitem__(slice(None), [None, None])), c32]
        except TypeError:
            _C32_CACHE[:] = [None, None]
    out = ctypes.POINTER(ctypes.c_int32)()
    rc = int(lib.flooder_cell_faces(c32.ctypes.data, c32.shape[0], c32.shape[1], d + 1,
                                    max(int(n_points), int(c32.max()) + 1), _host_threads(), ctypes.byref(out)))
    if rc < 0:
        return None
    return _take_rows(lib, out, rc, d + 1, _host_threads())


def faces_of_cells(cells: np.ndarray, d: int, n_points: int = 0,
                   want_index: bool = True) -> Tuple[np.ndarray, Optional[np.ndarray]]:
    """The d-dimensional faces of ``cells`` (rows of ascending vertex ids): ``(table, index)`` with ``table`` the
    sorted unique ``(n_d, d+1)`` rows and ``index[c, j]`` the row of ``table`` holding face j of cell c, faces
    numbered like ``itertools.combinations(range(width), d + 1)`` (``index`` is None when the packed keys would not
    fit 62 bits; vertices: every one of the ``n_points`` input points is a vertex of the complex, as in gudhi)."""
    width = cells.shape[1]
    if d + 1 > width or cells.shape[0] == 0:
        return np.zeros((0, d + 1), dtype=np.int64), None
    if d + 1 == width:
        return cells, np.arange(cells.shape[0], dtype=np.int64)[:, None]
    combos = list(itertools.combinations(range(width), d + 1))
    if d == 0 and n_points > 0:
        table = np.arange(n_points, dtype=np.int64)[:, None]
        return table, (cells.copy() if want_index else None)            # (face j of a cell = its j-th vertex = row id)
    base = int(cells.max()) + 1
    k = d + 1
    if want_index and base ** k < 2 ** 62:
        mult = base ** np.arange(k - 1, -1, -1, dtype=np.int64)
        keys = np.stack([cells[:, c] @ mult for c in combos], axis=1)          # (n_cells, n_combos)
        uk, inv = np.unique(keys.reshape(-1), return_inverse=True)
        table = np.empty((uk.shape[0], k), dtype=np.int64)
        rest = uk
        for j in range(k - 1, -1, -1):
            table[:, j] = rest % base
            rest = rest // base
        return table, inv.reshape(keys.shape).astype(np.int64)
    if not want_index and cells.shape[0] * len(combos) >= NATIVE_FACES_MIN:
        table = _faces_native(cells, d, n_points)
        if table is not None:
            return table, None
    if base ** k < 2 ** 62:  # packed keys, one combination at a time (no (n_cells x n_combos, k) intermediate)
        mult = base ** np.arange(k - 1, -1, -1, dtype=np.int64)
        rest = np.unique(np.concatenate([cells[:, c] @ mult for c in combos]))
        table = np.empty((rest.shape[0], k), dtype=np.int64)
        for j in range(k - 1, -1, -1):
            table[:, j] = rest % base
            rest = rest // base
        return table, None
    faces = np.concatenate([cells[:, c] for c in combos], axis=0)
    return _unique_rows(faces), None


def delaunay_simplices(points: np.ndarray, max_dimension: Optional[int] = None) -> List[np.ndarray]:
    """All simplices of the Delaunay triangulation of ``points``, bucketed by dimension.

    Returns ``out[d]`` = sorted unique ``(n_d, d+1)`` int64 array of ascending vertex
    ids, for d = 0 .. min(max_dimension, ambient dim).  Mirrors what the reference
    collects from ``stree.get_simplices()`` at ``core.py:135-138``.
    """
    points = np.ascontiguousarray(points, dtype=np.float64)
    n, dim = points.shape
    top = dim if max_dimension is None else min(max_dimension, dim)
    cells = delaunay_cells(points)
    return [faces_of_cells(cells, d, n)[0] for d in range(top + 1)]


class SimplexTree:
    """Minimal filtered simplicial complex with the gudhi ``SimplexTree`` call surface."""

    def __init__(self) -> None:
        self._rows: Dict[int, np.ndarray] = {}
        self._vals: Dict[int, np.ndarray] = {}
        self._pending: Dict[int, Dict[Tuple[int, ...], float]] = {}
        self._persistence = None
        # complexes built from top-dimensional cells (``from_cells``): face tables are enumerated on first use
        self._cells: Optional[np.ndarray] = None
        self._n_points = 0
        self._lazy: set = set()                       # dimensions not enumerated yet
        self._cell_faces: Dict[int, Optional[np.ndarray]] = {}   # d -> (n_cells, n_combos) rows of table d
        self._monotone = False                        # make_filtration_non_decreasing has run: late tables inherit

    @classmethod
    def from_cells(cls, cells: np.ndarray, n_points: int, eager: Optional[int] = None,
                   trusted: bool = False) -> "SimplexTree":
        """Complex spanned by top-dimensional ``cells`` (unique rows of ascending vertex ids) over ``n_points``
        vertices, filtration values NaN (what ``gudhi.DelaunayComplex(...).create_simplex_tree()`` returns).  Face
        tables of dimension <= ``eager`` (default: all) are enumerated now, the others when first touched - the
        6-D Delaunay complex of 2000 points has 15 million simplices, of which a ``max_dimension=2`` run needs
        1.4 million."""
        st = cls()
        cells = np.asarray(cells, dtype=np.int64)
        # Ascending vertex ids are relied upon: a face shared by several cells then has the same vertex ORDER in each
        # of them, so its samples (weights x vertices, same fma order) and its value are bit-identical from every
        # cell - which is why "last writer wins" (assign_cell_faces) and "max per distinct face" (the fused
        # shared-slot sweep) agree bit for bit, and why sharded and unsharded runs are bit-equal.
        # (``trusted``: rows that come straight from delaunay_cells, ascending by construction - the check is a pass over
        # ten million entries for the 6-D complex of cfg 4)
        if not trusted and cells.shape[1] > 1 and not (np.diff(cells, axis=1) > 0).all():
            raise ValueError("SimplexTree.from_cells: every cell must list its vertex ids in ascending order")
        st._cells = cells
        st._n_points = int(n_points)
        top = cells.shape[1] - 1
        st._lazy = set(range(top + 1))
        for d in range(top + 1 if eager is None else min(eager, top) + 1):
            st._materialise(d)
        return st

    def _materialise(self, d: int) -> None:
        if d not in self._lazy:
            return
        self._lazy.discard(d)
        # (the cell -> face rows cost an argsort of all faces: kept only for the tables enumerated up front, which
        # are the ones the sweep assigns values to)
        # (... and only while the argsort is small: the 6-D complex of 2000 points has 35 triangles in each of
        # 1.5 million cells - there the monotone pass and the hand-off locate rows by key instead)
        n_faces_all = self._cells.shape[0] * math.comb(self._cells.shape[1], d + 1)
        table, index = faces_of_cells(self._cells, d, self._n_points,
                                      want_index=not self._monotone and n_faces_all <= INDEX_MAX_FACES)
        self._cell_faces[d] = index
        if table.shape[0]:
            self._rows[d] = table
            self._vals[d] = np.full(table.shape[0], np.nan, dtype=np.float64)
            if self._monotone and d > 0:     # the monotone pass has run already: a late table takes its faces' maxima
                self._materialise(d - 1)
                self._raise_dimension(d)

    def _materialise_all(self) -> None:
        for d in sorted(self._lazy):
            self._materialise(d)

    def cell_face_index(self, d: int) -> Optional[np.ndarray]:
        """``index[c, j]`` = row of the dimension-d table holding face j (``itertools.combinations`` order of the
        kept vertex positions) of top cell c; None for complexes not built from cells."""
        if self._cells is None:
            return None
        self._materialise(d)
        return self._cell_faces.get(d)

    # ------------------------------------------------------------------ bulk API
    @classmethod
    def from_arrays(cls, simplices: Sequence[np.ndarray], filtration: float = float("nan")) -> "SimplexTree":
        st = cls()
        for rows in simplices:
            rows = np.asarray(rows, dtype=np.int64)
            if rows.ndim != 2 or rows.shape[0] == 0:
                continue
            d = rows.shape[1] - 1
            rows = _unique_rows(np.sort(rows, axis=1))
            st._rows[d] = rows
            st._vals[d] = np.full(rows.shape[0], filtration, dtype=np.float64)
        return st

    def simplices_of_dimension(self, d: int) -> np.ndarray:
        self._flush()
        self._materialise(d)
        return self._rows.get(d, np.zeros((0, d + 1), dtype=np.int64))

    def filtrations_of_dimension(self, d: int) -> np.ndarray:
        self._flush()
        self._materialise(d)
        return self._vals.get(d, np.zeros((0,), dtype=np.float64))

    def _locate(self, d: int, query: np.ndarray) -> np.ndarray:
        """Index of each (sorted-ascending) query row in the dimension-d table, -1 if absent."""
        self._materialise(d)
        table = self._rows.get(d)
        query = np.asarray(query, dtype=np.int64).reshape(-1, d + 1)
        if table is None or table.shape[0] == 0 or query.shape[0] == 0:
            return np.full(query.shape[0], -1, dtype=np.int64)
        if query.shape[0] * (d + 1) >= NATIVE_LOCATE_MIN and NATIVE_DELAUNAY:
            lib = _load_host()        # binary search on packed keys, all host cores (csrc/cell_faces.cpp)
            if lib is not None:
                q64 = np.ascontiguousarray(query, dtype=np.int64)
                t64 = np.ascontiguousarray(table, dtype=np.int64)
                out = np.empty(q64.shape[0], dtype=np.int64)
                n_pts = max(self._n_points, int(t64[:, -1].max()) + 1)   # (ascending rows: the last column holds the maxima)
                if int(lib.flooder_locate_rows(q64.ctypes.data, q64.shape[0], d + 1, t64.ctypes.data, t64.shape[0], n_pts,
                                               out.ctypes.data, _host_threads())) == 0:
                    return out
        base = int(max(table.max(), query.max())) + 1
        if base ** (d + 1) < 2 ** 62:
            mult = base ** np.arange(d, -1, -1, dtype=np.int64)
            tkey = table @ mult
            qkey = query @ mult
            pos = np.searchsorted(tkey, qkey)
            pos_c = np.minimum(pos, tkey.shape[0] - 1)
            hit = tkey[pos_c] == qkey
            return np.where(hit, pos_c, -1)
        # keys too wide for int64: sort-based matching on the rows themselves
        both = np.concatenate([table, query], axis=0)
        order = _lex_sort_rows(both)
        srt = both[order]
        new_group = np.ones(srt.shape[0], dtype=bool)
        new_group[1:] = np.any(srt[1:] != srt[:-1], axis=1)
        gid_sorted = np.cumsum(new_group) - 1
        gid = np.empty_like(gid_sorted)
        gid[order] = gid_sorted
        nt = table.shape[0]
        table_of_gid = np.full(gid_sorted[-1] + 1, -1, dtype=np.int64)
        table_of_gid[gid[:nt]] = np.arange(nt)
        return table_of_gid[gid[nt:]]

    def assign_filtration_bulk(self, simplices: np.ndarray, values: np.ndarray) -> None:
        """Vectorised ``assign_filtration`` for many simplices of one dimension.

        Later rows win on duplicates, as successive ``dict.update`` calls do in the
        reference (``core.py:258-263``).  Rows that are not in the complex are ignored
        (gudhi's ``assign_filtration`` leaves the tree unchanged for them).
        """
        self._flush()
        simplices = np.sort(np.asarray(simplices, dtype=np.int64), axis=1)
        values = np.asarray(values, dtype=np.float64).reshape(-1)
        if simplices.shape[0] == 0:
            return
        d = simplices.shape[1] - 1
        idx = self._locate(d, simplices)
        ok = idx >= 0
        self._vals[d][idx[ok]] = values[ok]
        self._persistence = None

    # ------------------------------------------------------------- gudhi surface
    def _flush(self) -> None:
        if not self._pending:
            return
        self._materialise_all()   # (explicit inserts: the complex is no longer the span of its cells)
        for d, items in self._pending.items():
            if not items:
                continue
            new_rows = np.array(list(items.keys()), dtype=np.int64).reshape(-1, d + 1)
            new_vals = np.array(list(items.values()), dtype=np.float64)
            if d in self._rows and self._rows[d].shape[0]:
                rows = np.concatenate([self._rows[d], new_rows], axis=0)
                vals = np.concatenate([self._vals[d], new_vals], axis=0)
            else:
                rows, vals = new_rows, new_vals
            order = _lex_sort_rows(rows)
            self._rows[d] = rows[order]
            self._vals[d] = vals[order]
        self._pending = {}

    def find(self, simplex: Iterable[int]) -> bool:
        key = tuple(sorted(int(v) for v in simplex))
        d = len(key) - 1
        if d in self._pending and key in self._pending[d]:
            return True
        return bool(self._locate(d, np.array([key]))[0] >= 0) if d >= 0 else False

    def assign_cell_faces(self, d: int, cell_rows: np.ndarray, face_cols: Sequence[int], values: np.ndarray) -> bool:
        """``values[i, j]`` -> face ``face_cols[j]`` (numbering of ``cell_face_index``) of top cell ``cell_rows[i]``,
        through the index kept from the enumeration of the faces instead of a search per row.  A face shared by
        several cells receives the same value from each of them (``from_cells`` insists on ascending vertex ids, see
        there), so which duplicate numpy's indexed assignment keeps does not matter.  False (nothing done) when no
        such index exists."""
        index = self.cell_face_index(d)
        if index is None or d not in self._vals:
            return False
        rows = index[np.asarray(cell_rows)][:, list(face_cols)]
        self._vals[d][rows.reshape(-1)] = np.asarray(values, dtype=np.float64).reshape(-1)
        self._persistence = None
        return True

    def insert(self, simplex: Iterable[int], filtration: float = 0.0) -> bool:
        """Insert a simplex and all its faces (gudhi semantics: existing simplices keep
        the smaller of their old value and ``filtration``; new ones get ``filtration``).
        Returns True if the simplex itself was not present before."""
        key = tuple(sorted(int(v) for v in simplex))
        if len(key) == 0:
            return False
        self._flush()
        was_new = not self.find(key)
        for k in range(1, len(key) + 1):
            d = k - 1
            faces = np.array(list(itertools.combinations(key, k)), dtype=np.int64)
            idx = self._locate(d, faces)
            present = idx >= 0
            if present.any():
                cur = self._vals[d][idx[present]]
                # NaN (unset) values are replaced, otherwise keep the smaller value
                newv = np.where(np.isnan(cur), filtration, np.minimum(cur, filtration))
                self._vals[d][idx[present]] = newv
            missing = faces[~present]
            if missing.shape[0]:
                pend = self._pending.setdefault(d, {})
                for row in missing:
                    pend[tuple(int(v) for v in row)] = float(filtration)
        self._flush()
        self._persistence = None
        return was_new

    def assign_filtration(self, simplex: Iterable[int], filtration: float) -> None:
        key = tuple(sorted(int(v) for v in simplex))
        self._flush()
        d = len(key) - 1
        idx = self._locate(d, np.array([key]))[0]
        if idx >= 0:
            self._vals[d][idx] = float(filtration)
            self._persistence = None

    def filtration(self, simplex: Iterable[int]) -> float:
        key = tuple(sorted(int(v) for v in simplex))
        self._flush()
        d = len(key) - 1
        idx = self._locate(d, np.array([key]))[0]
        return float(self._vals[d][idx]) if idx >= 0 else float("inf")

    def num_simplices(self) -> int:
        self._flush()
        self._materialise_all()
        return int(sum(r.shape[0] for r in self._rows.values()))

    def num_vertices(self) -> int:
        self._flush()
        return int(self._rows[0].shape[0]) if 0 in self._rows else 0

    def dimension(self) -> int:
        self._flush()
        if self._lazy:
            return max(max(self._lazy), max([d for d, r in self._rows.items() if r.shape[0]], default=-1))
        dims = [d for d, r in self._rows.items() if r.shape[0]]
        return max(dims) if dims else -1

    def get_simplices(self) -> Iterator[Tuple[List[int], float]]:
        """Yield ``(vertex list, filtration)`` in the depth-first order of a simplex tree:
        lexicographic, a simplex directly before the simplices it prefixes."""
        self._flush()
        self._materialise_all()
        dims = sorted(d for d, r in self._rows.items() if r.shape[0])
        if not dims:
            return
        width = dims[-1] + 1
        padded, vals, lens = [], [], []
        for d in dims:
            rows = self._rows[d]
            pad = np.full((rows.shape[0], width), -1, dtype=np.int64)
            pad[:, : d + 1] = rows
            padded.append(pad)
            vals.append(self._vals[d])
            lens.append(np.full(rows.shape[0], d + 1, dtype=np.int64))
        padded = np.concatenate(padded, axis=0)
        vals = np.concatenate(vals)
        lens = np.concatenate(lens)
        order = _lex_sort_rows(padded)
        rows_l = padded[order].tolist()
        vals_l = vals[order].tolist()
        lens_l = lens[order].tolist()
        for row, v, k in zip(rows_l, vals_l, lens_l):
            yield row[:k], v

    def to_dict(self) -> Dict[Tuple[int, ...], float]:
        """``{simplex tuple: filtration}`` of the whole complex (what ``dict(stree.get_simplices())`` gives,
        reference core.py:285-288) without a Python-level generator per simplex."""
        self._flush()
        self._materialise_all()
        out: Dict[Tuple[int, ...], float] = {}
        lib = _py_lib()
        cache: list = []
        for d in sorted(self._rows):
            rows = self._rows[d]
            if rows.shape[0]:
                if lib is not None:   # one pass in C over the integer table (csrc/pyhandoff.c)
                    r64 = np.ascontiguousarray(rows, dtype=np.int64)
                    v64 = np.ascontiguousarray(self._vals[d], dtype=np.float64)
                    if lib.flooder_dict_update(out, r64.ctypes.data, r64.shape[0], r64.shape[1], v64.ctypes.data, cache) != 0:
                        raise RuntimeError("flooder_dict_update failed")
                    continue
                # (tuples straight from the columns: a third faster than tuple() over rows.tolist())
                cols = [rows[:, j].tolist() for j in range(rows.shape[1])]
                out.update(zip(zip(*cols), self._vals[d].tolist()))
        return out

    def get_filtration(self) -> Iterator[Tuple[List[int], float]]:
        """Simplices sorted by (filtration, dimension, lexicographic), gudhi's filtration order."""
        self._flush()
        self._materialise_all()
        items = []
        for d, rows in self._rows.items():
            for row, v in zip(rows.tolist(), self._vals[d].tolist()):
                items.append((v, d, row))
        items.sort(key=lambda t: (t[0], t[1], t[2]))
        for v, _, row in items:
            yield row, v

    def get_skeleton(self, dimension: int) -> Iterator[Tuple[List[int], float]]:
        for simplex, v in self.get_simplices():
            if len(simplex) <= dimension + 1:
                yield simplex, v

    def get_boundaries(self, simplex: Iterable[int]) -> Iterator[Tuple[List[int], float]]:
        key = tuple(sorted(int(v) for v in simplex))
        if len(key) <= 1:
            return
        self._flush()
        d = len(key) - 2
        faces = np.array([key[:j] + key[j + 1:] for j in range(len(key))], dtype=np.int64)
        idx = self._locate(d, faces)
        for row, i in zip(faces.tolist(), idx.tolist()):
            if i >= 0:
                yield row, float(self._vals[d][i])

    def _facet_rows(self, d: int) -> Optional[np.ndarray]:
        """``out[i, j]`` = row of table d-1 holding the facet of simplex i of table d that omits its j-th vertex,
        read off the cell -> face indices of the two tables (every face lies in some cell); None without them."""
        if self._cells is None or self._cell_faces.get(d) is None or self._cell_faces.get(d - 1) is None or d == 0:
            return None
        width = self._cells.shape[1]
        hi_c = list(itertools.combinations(range(width), d + 1))
        lo_c = {c: i for i, c in enumerate(itertools.combinations(range(width), d))}
        idx_hi, idx_lo = self._cell_faces[d], self._cell_faces[d - 1]
        out = np.empty((self._rows[d].shape[0], d + 1), dtype=np.int64)
        for j, cmb in enumerate(hi_c):
            for m in range(d + 1):
                out[idx_hi[:, j], m] = idx_lo[:, lo_c[cmb[:m] + cmb[m + 1:]]]
        return out

    def _raise_dimension(self, d: int) -> bool:
        """One step of the monotone pass: dimension-d values raised to the maxima of their facets."""
        rows = self._rows.get(d)
        if d == 0 or rows is None or rows.shape[0] == 0 or (d - 1) not in self._rows:
            return False
        vals = self._vals[d]
        lower = self._vals[d - 1]
        facets = self._facet_rows(d)
        if facets is None and rows.shape[0] * (d + 1) >= NATIVE_RAISE_MIN and NATIVE_DELAUNAY:
            lib = _load_host()     # the facets located and compared on all host cores (csrc/cell_faces.cpp)
            if lib is not None:
                lo_rows = np.ascontiguousarray(self._rows[d - 1], dtype=np.int64)
                r64 = np.ascontiguousarray(rows, dtype=np.int64)
                v64 = np.ascontiguousarray(vals, dtype=np.float64).copy()
                rc = int(lib.flooder_raise_dimension(r64.ctypes.data, r64.shape[0], d + 1, lo_rows.ctypes.data,
                                                     lo_rows.shape[0], np.ascontiguousarray(lower, dtype=np.float64).ctypes.data,
                                                     v64.ctypes.data, max(self._n_points, int(r64.max()) + 1, int(lo_rows.max()) + 1),
                                                     _host_threads()))
                if rc >= 0:
                    if rc > 0:
                        self._vals[d] = v64
                    return rc > 0
        face_max = np.full(rows.shape[0], -np.inf)
        for j in range(d + 1):
            if facets is not None:
                idx = facets[:, j]
            else:
                face = np.delete(rows, j, axis=1)
                idx = self._locate(d - 1, face)
            fv = np.where(idx >= 0, lower[np.maximum(idx, 0)], -np.inf)
            fv = np.where(np.isnan(fv), -np.inf, fv)
            face_max = np.maximum(face_max, fv)
        own_nan = np.isnan(vals)
        raised = np.where(own_nan, face_max, np.maximum(vals, face_max))
        raised = np.where(np.isneginf(raised), vals, raised)
        diff = ~((raised == vals) | (np.isnan(raised) & np.isnan(vals)))
        if diff.any():
            self._vals[d] = raised
            return True
        return False

    def make_filtration_non_decreasing(self) -> bool:
        """Raise every simplex to at least the value of each of its faces, dimension by
        dimension (gudhi ``Simplex_tree::make_filtration_non_decreasing``).  A NaN face value
        does not propagate; a NaN own value is replaced by the maximum over its faces.
        Face tables that have not been enumerated yet take part when they are (``from_cells``)."""
        self._flush()
        changed = False
        for d in sorted(self._rows):
            changed = self._raise_dimension(d) or changed
        self._monotone = True
        if changed:
            self._persistence = None
        return changed

    # --------------------------------------------------------------- persistence
    def compute_persistence(self, homology_coeff_field: int = 2, min_persistence: float = 0.0,
                            persistence_dim_max: bool = False) -> None:
        from .persistence import persistence_pairs

        self._flush()
        self._materialise_all()
        self._persistence = persistence_pairs(self, min_persistence=min_persistence,
                                              persistence_dim_max=persistence_dim_max)

    def persistence(self, homology_coeff_field: int = 2, min_persistence: float = 0.0,
                    persistence_dim_max: bool = False) -> List[Tuple[int, Tuple[float, float]]]:
        self.compute_persistence(homology_coeff_field, min_persistence, persistence_dim_max)
        out = []
        for dim in sorted(self._persistence, reverse=True):
            arr = self._persistence[dim]
            order = np.argsort(-(arr[:, 1] - arr[:, 0]), kind="stable")
            out.extend((dim, (float(b), float(d))) for b, d in arr[order])
        return out

    def persistence_intervals_in_dimension(self, dimension: int) -> np.ndarray:
        if self._persistence is None:
            raise RuntimeError("compute_persistence() must be called before "
                               "persistence_intervals_in_dimension()")
        arr = self._persistence.get(dimension)
        if arr is None or arr.shape[0] == 0:
            return np.zeros((0, 2), dtype=np.float64)
        order = np.lexsort((arr[:, 1], arr[:, 0]))
        return arr[order]


class DelaunayComplex:
    """Stand-in for ``gudhi.DelaunayComplex`` (reference ``core.py:130-132``)."""

    def __init__(self, points) -> None:
        try:
            import torch

            if isinstance(points, torch.Tensor):
                points = points.detach().cpu().numpy()
        except Exception:  # pragma: no cover
            pass
        self._points = np.asarray(points, dtype=np.float64)

    def create_simplex_tree(self, *_, **__) -> SimplexTree:
        return SimplexTree.from_cells(delaunay_cells(self._points), self._points.shape[0])
