"""Persistent homology (Z/2) of a ``flooder_amd.SimplexTree``.

Stand-in for ``gudhi.SimplexTree.compute_persistence`` / ``persistence_intervals_in_dimension`` (gudhi is a
third-party C++ library; the reference calls it at ``flooder/cli.py:473-476`` and
``tests/test_flooder.py:55-71``).  The boundary-matrix reduction runs in host C++
(``csrc/persistence.cpp`` -> ``libflooder_host.so``); a pure-Python reduction of the same algorithm is kept
for cross-checking (``persistence_pairs_python``).

Conventions follow gudhi: simplices are ordered by (filtration value, dimension, vertices); intervals of
length <= ``min_persistence`` are dropped (``min_persistence=-1`` keeps everything); homology in the top
dimension of the complex is only reported with ``persistence_dim_max=True``; unpaired simplices give
intervals ``(birth, inf)``.
"""

from __future__ import annotations

import ctypes
import os
from typing import Dict, List, Tuple

import numpy as np

_HOST_LIB = None


def _host_lib():
    global _HOST_LIB
    if _HOST_LIB is None:
        from . import build

        path = build.HOST_LIB
        if not os.path.exists(path):
            build.build_host()
        lib = ctypes.CDLL(path)
        lib.flooder_persistence_z2.restype = ctypes.c_int
        lib.flooder_persistence_z2.argtypes = [ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                               ctypes.c_void_p]
        lib.flooder_filtration_order.restype = ctypes.c_int
        lib.flooder_filtration_order.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 8
        _HOST_LIB = lib
    return _HOST_LIB


def filtration_order(st):
    """Simplices of the tree in filtration order with their boundaries.

    Returns (dims int32 (n,), filt float64 (n,), bptr int64 (n+1,), bidx int64, rows) where ``rows[d]`` is the
    dimension-d simplex table and the boundary of simplex j (position in filtration order) is
    ``bidx[bptr[j]:bptr[j+1]]`` (positions in filtration order)."""
    st._flush()
    dims_present = sorted(d for d, r in st._rows.items() if r.shape[0])
    if not dims_present:
        z = np.zeros(0, dtype=np.int64)
        return np.zeros(0, np.int32), np.zeros(0), np.zeros(1, np.int64), z, {}
    top = dims_present[-1]
    counts = [st._rows[d].shape[0] if d in st._rows else 0 for d in range(top + 1)]
    if NATIVE_ORDER:
        out = _filtration_order_native(st, top, counts)
        if out is not None:
            return out
    offs = np.concatenate([[0], np.cumsum(counts)])  # global id = offs[d] + row index
    n = int(offs[-1])
    dims = np.concatenate([np.full(c, d, dtype=np.int32) for d, c in enumerate(counts)])
    filt = np.concatenate([st._vals[d] if d in st._vals else np.zeros(0) for d in range(top + 1)])
    filt_key = np.where(np.isnan(filt), np.inf, filt)
    # (filtration, dimension, lexicographic row order == table order)
    order = np.lexsort((np.arange(n), dims, filt_key))
    pos = np.empty(n, dtype=np.int64)
    pos[order] = np.arange(n)
    # boundaries in global ids
    bcount = np.where(dims > 0, dims + 1, 0).astype(np.int64)
    bptr_g = np.concatenate([[0], np.cumsum(bcount)])
    bidx_g = np.empty(int(bptr_g[-1]), dtype=np.int64)
    for d in range(1, top + 1):
        rows = st._rows.get(d)
        if rows is None or rows.shape[0] == 0:
            continue
        base = bptr_g[offs[d]]
        faces = np.empty((rows.shape[0], d + 1), dtype=np.int64)
        for j in range(d + 1):
            idx = st._locate(d - 1, np.delete(rows, j, axis=1))
            if (idx < 0).any():
                raise ValueError("complex is not closed under taking faces")
            faces[:, j] = offs[d - 1] + idx
        bidx_g[base:base + faces.size] = faces.reshape(-1)
    # permute to filtration order
    bcount_o = bcount[order]
    bptr = np.concatenate([[0], np.cumsum(bcount_o)]).astype(np.int64)
    bidx = np.empty_like(bidx_g)
    # gather boundaries of the simplices in filtration order
    starts = bptr_g[order]
    total = int(bptr[-1])
    if total:
        # entry e of the permuted boundary list belongs to simplex j = owner[e] and is its (e - bptr[j])-th face
        owner = np.repeat(np.arange(n, dtype=np.int64), bcount_o)
        take = starts[owner] + (np.arange(total, dtype=np.int64) - bptr[:-1][owner])
        bidx = pos[bidx_g[take]]
    else:
        bidx = np.zeros(0, np.int64)
    return dims[order].astype(np.int32), filt[order], bptr, bidx.astype(np.int64), order


NATIVE_ORDER = True   # filtration order + boundaries in host C++ (flooder_filtration_order); False: the numpy version


def _filtration_order_native(st, top: int, counts):
    """``filtration_order`` through ``flooder_filtration_order`` (csrc/persistence.cpp): one sort and a binary search
    per facet instead of packed-key ``searchsorted`` passes and a ``lexsort``.  None where it does not apply (a facet
    missing from its table: the numpy version then raises the error; a missing library)."""
    try:
        lib = _host_lib()
    except Exception:
        return None
    n = int(sum(counts))
    rows = np.concatenate([np.ascontiguousarray(st._rows[d], dtype=np.int64).reshape(-1) if c else np.zeros(0, np.int64)
                           for d, c in enumerate(counts)])
    vals = np.concatenate([np.asarray(st._vals[d], dtype=np.float64) if c else np.zeros(0)
                           for d, c in enumerate(counts)])
    cnt = np.asarray(counts, dtype=np.int64)
    nb = int(sum(c * (d + 1) for d, c in enumerate(counts) if d > 0))
    dims = np.empty(n, dtype=np.int32)
    filt = np.empty(n, dtype=np.float64)
    bptr = np.empty(n + 1, dtype=np.int64)
    bidx = np.empty(max(nb, 1), dtype=np.int64)
    order = np.empty(n, dtype=np.int64)
    rc = lib.flooder_filtration_order(top, cnt.ctypes.data, rows.ctypes.data, vals.ctypes.data, dims.ctypes.data,
                                      filt.ctypes.data, bptr.ctypes.data, bidx.ctypes.data, order.ctypes.data)
    if rc != 0:
        return None
    return dims, filt, bptr, bidx[:nb], order


def reduce_pairs(dims: np.ndarray, bptr: np.ndarray, bidx: np.ndarray) -> np.ndarray:
    lib = _host_lib()
    n = dims.shape[0]
    pair = np.empty(n, dtype=np.int64)
    dims = np.ascontiguousarray(dims, dtype=np.int32)
    bptr = np.ascontiguousarray(bptr, dtype=np.int64)
    bidx = np.ascontiguousarray(bidx, dtype=np.int64)
    rc = lib.flooder_persistence_z2(n, dims.ctypes.data, bptr.ctypes.data, bidx.ctypes.data, pair.ctypes.data)
    if rc != 0:
        raise RuntimeError("flooder_persistence_z2 failed")
    return pair


def reduce_pairs_python(dims: np.ndarray, bptr: np.ndarray, bidx: np.ndarray) -> np.ndarray:
    """Plain left-to-right column reduction with Python sets (reference for the C++ reduction)."""
    n = dims.shape[0]
    pair = np.full(n, -1, dtype=np.int64)
    low_to_col: Dict[int, int] = {}
    cols: Dict[int, set] = {}
    for j in range(n):
        c = set(bidx[bptr[j]:bptr[j + 1]].tolist())
        while c:
            low = max(c)
            k = low_to_col.get(low)
            if k is None:
                break
            c ^= cols[k]
        if c:
            low = max(c)
            low_to_col[low] = j
            cols[j] = c
            pair[low] = j
            pair[j] = low
    return pair


def intervals_from_pairs(dims, filt, pair, min_persistence=0.0, persistence_dim_max=False) -> Dict[int, np.ndarray]:
    """(birth, death) rows per dimension, in filtration order of the birth simplex: unpaired simplices give
    essential classes (death = inf), a pair (j, p > j) gives [filt[j], filt[p]); the top dimension is left out
    unless ``persistence_dim_max``; only intervals longer than ``min_persistence`` are kept."""
    dims = np.asarray(dims)
    if dims.size == 0:
        return {}
    filt = np.asarray(filt, dtype=np.float64)
    pair = np.asarray(pair)
    idx = np.arange(dims.shape[0])
    births = (pair == -1) | (pair > idx)
    death = np.where(pair == -1, np.inf, filt[np.where(pair >= 0, pair, 0)])
    keep = births & (death - filt > min_persistence)
    top = int(dims.max())
    if not persistence_dim_max:
        keep &= dims != top
    out: Dict[int, np.ndarray] = {}
    for d in np.unique(dims[keep]).tolist():
        sel = keep & (dims == d)
        out[int(d)] = np.stack((filt[sel], death[sel]), axis=1)
    return out


def persistence_pairs(st, min_persistence: float = 0.0, persistence_dim_max: bool = False) -> Dict[int, np.ndarray]:
    dims, filt, bptr, bidx, _ = filtration_order(st)
    pair = reduce_pairs(dims, bptr, bidx)
    return intervals_from_pairs(dims, filt, pair, min_persistence, persistence_dim_max)
