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
from typing import Dict, Iterable, Iterator, List, Optional, Sequence, Tuple

import numpy as np

__all__ = ["SimplexTree", "DelaunayComplex", "delaunay_simplices", "HAS_GUDHI"]

try:  # pragma: no cover - gudhi is absent from the build image
    import gudhi as _gudhi  # type: ignore

    HAS_GUDHI = True
except Exception:  # pragma: no cover
    _gudhi = None
    HAS_GUDHI = False


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


def delaunay_simplices(points: np.ndarray, max_dimension: Optional[int] = None) -> List[np.ndarray]:
    """All simplices of the Delaunay triangulation of ``points``, bucketed by dimension.

    Returns ``out[d]`` = sorted unique ``(n_d, d+1)`` int64 array of ascending vertex
    ids, for d = 0 .. min(max_dimension, ambient dim).  Mirrors what the reference
    collects from ``stree.get_simplices()`` at ``core.py:135-138``.
    """
    from scipy.spatial import Delaunay

    points = np.ascontiguousarray(points, dtype=np.float64)
    n, dim = points.shape
    top = dim if max_dimension is None else min(max_dimension, dim)
    if dim == 1:
        order = np.argsort(points[:, 0], kind="stable")
        cells = np.stack([order[:-1], order[1:]], axis=1) if n > 1 else np.zeros((0, 2), np.int64)
    elif n <= dim:
        # fewer points than a full simplex: a single (n-1)-simplex on all points
        cells = np.arange(n, dtype=np.int64)[None, :]
    else:
        cells = Delaunay(points).simplices
    cells = np.sort(np.asarray(cells, dtype=np.int64), axis=1)
    out: List[np.ndarray] = []
    width = cells.shape[1]
    for d in range(top + 1):
        if d + 1 > width:
            out.append(np.zeros((0, d + 1), dtype=np.int64))
            continue
        combos = list(itertools.combinations(range(width), d + 1))
        faces = np.concatenate([cells[:, c] for c in combos], axis=0)
        out.append(_unique_rows(faces))
    if out and out[0].shape[0] < n:
        # every input point is a vertex of the complex (gudhi inserts all of them)
        out[0] = np.arange(n, dtype=np.int64)[:, None]
    return out


class SimplexTree:
    """Minimal filtered simplicial complex with the gudhi ``SimplexTree`` call surface."""

    def __init__(self) -> None:
        self._rows: Dict[int, np.ndarray] = {}
        self._vals: Dict[int, np.ndarray] = {}
        self._pending: Dict[int, Dict[Tuple[int, ...], float]] = {}
        self._persistence = None

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
        return self._rows.get(d, np.zeros((0, d + 1), dtype=np.int64))

    def filtrations_of_dimension(self, d: int) -> np.ndarray:
        self._flush()
        return self._vals.get(d, np.zeros((0,), dtype=np.float64))

    def _locate(self, d: int, query: np.ndarray) -> np.ndarray:
        """Index of each (sorted-ascending) query row in the dimension-d table, -1 if absent."""
        table = self._rows.get(d)
        query = np.asarray(query, dtype=np.int64).reshape(-1, d + 1)
        if table is None or table.shape[0] == 0 or query.shape[0] == 0:
            return np.full(query.shape[0], -1, dtype=np.int64)
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
        return int(sum(r.shape[0] for r in self._rows.values()))

    def num_vertices(self) -> int:
        self._flush()
        return int(self._rows[0].shape[0]) if 0 in self._rows else 0

    def dimension(self) -> int:
        self._flush()
        dims = [d for d, r in self._rows.items() if r.shape[0]]
        return max(dims) if dims else -1

    def get_simplices(self) -> Iterator[Tuple[List[int], float]]:
        """Yield ``(vertex list, filtration)`` in the depth-first order of a simplex tree:
        lexicographic, a simplex directly before the simplices it prefixes."""
        self._flush()
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
        out: Dict[Tuple[int, ...], float] = {}
        for d in sorted(self._rows):
            rows = self._rows[d]
            if rows.shape[0]:
                out.update(zip(map(tuple, rows.tolist()), self._vals[d].tolist()))
        return out

    def get_filtration(self) -> Iterator[Tuple[List[int], float]]:
        """Simplices sorted by (filtration, dimension, lexicographic), gudhi's filtration order."""
        self._flush()
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

    def make_filtration_non_decreasing(self) -> bool:
        """Raise every simplex to at least the value of each of its faces, dimension by
        dimension (gudhi ``Simplex_tree::make_filtration_non_decreasing``).  A NaN face value
        does not propagate; a NaN own value is replaced by the maximum over its faces."""
        self._flush()
        changed = False
        dims = sorted(self._rows)
        for d in dims:
            if d == 0 or (d - 1) not in self._rows:
                continue
            rows = self._rows[d]
            if rows.shape[0] == 0:
                continue
            vals = self._vals[d]
            lower = self._vals[d - 1]
            face_max = np.full(rows.shape[0], -np.inf)
            for j in range(d + 1):
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
                changed = True
                self._vals[d] = raised
        if changed:
            self._persistence = None
        return changed

    # --------------------------------------------------------------- persistence
    def compute_persistence(self, homology_coeff_field: int = 2, min_persistence: float = 0.0,
                            persistence_dim_max: bool = False) -> None:
        from .persistence import persistence_pairs

        self._flush()
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
        return SimplexTree.from_arrays(delaunay_simplices(self._points))
