"""2-D alpha filtration and bottleneck distance.  TEST INFRASTRUCTURE ONLY (see ``oracle/flood_oracle.py``).

Restates, for the reference's homotopy-equivalence test ``tests/test_flooder.py:24-75`` (``test_vs_alpha``:
Flood complex with L = X against ``gudhi.AlphaComplex(X).create_simplex_tree(output_squared_values=False)``,
``gudhi.bottleneck_distance`` < 5e-4 in dimensions 0 and 1), the two gudhi (3.11.0, third-party, absent from
the reference tree and from this image) pieces that test needs:

* ``alpha_filtration_2d``  - the alpha-complex filtration as gudhi's ``Alpha_complex::create_complex``
  documents it: simplices of the Delaunay triangulation by decreasing dimension; a simplex without a value
  gets its circumradius; each facet tau of sigma gets ``min(value(tau), value(sigma))`` if it has a value, else
  ``value(sigma)`` if tau is not Gabriel for sigma (sigma's remaining vertex lies strictly inside tau's
  diametral ball), else stays open (and later gets its own circumradius).  Values are radii, not squared.
* ``bottleneck_distance``  - exact bottleneck distance between two persistence diagrams (L-infinity ground
  metric, diagonal projections; essential classes matched among themselves): binary search over the candidate
  costs with a perfect-matching test (``scipy.sparse.csgraph.maximum_bipartite_matching``, Hopcroft-Karp).

"Parity unpinned" against gudhi itself (it cannot be imported here); pinned by construction tests in
``tests/test_alpha.py`` (hand-checked triangles, bottleneck distances of small diagrams with known answers,
agreement of the matching-based value with a brute-force permutation search).
"""

from __future__ import annotations

import numpy as np


def _circumradius2_tri(a, b, c):
    """Squared circumradius of triangles (n,2) x3, float64."""
    ab = b - a
    ac = c - a
    d = 2.0 * (ab[:, 0] * ac[:, 1] - ab[:, 1] * ac[:, 0])
    ab2 = (ab ** 2).sum(1)
    ac2 = (ac ** 2).sum(1)
    ux = (ac[:, 1] * ab2 - ab[:, 1] * ac2) / d
    uy = (ab[:, 0] * ac2 - ac[:, 0] * ab2) / d
    return ux * ux + uy * uy


def alpha_filtration_2d(points: np.ndarray):
    """Alpha filtration of a planar point set.  Returns ``(vertices (n,1), edges (m,2), triangles (t,3))`` int64
    arrays with ascending vertex ids and their filtration values (radii) ``(fv, fe, ft)``."""
    from scipy.spatial import Delaunay

    P = np.asarray(points, dtype=np.float64)
    tri = np.sort(Delaunay(P).simplices.astype(np.int64), axis=1)
    ft2 = _circumradius2_tri(P[tri[:, 0]], P[tri[:, 1]], P[tri[:, 2]])
    # edges with their (<= 2) incident triangles and the vertex opposite to the edge in each
    e_all = np.concatenate([tri[:, [0, 1]], tri[:, [0, 2]], tri[:, [1, 2]]])
    opp = np.concatenate([tri[:, 2], tri[:, 1], tri[:, 0]])
    t_id = np.tile(np.arange(tri.shape[0]), 3)
    key = e_all[:, 0] * (P.shape[0] + 1) + e_all[:, 1]
    uk, inv = np.unique(key, return_inverse=True)
    edges = np.stack([uk // (P.shape[0] + 1), uk % (P.shape[0] + 1)], axis=1)
    mid = 0.5 * (P[edges[:, 0]] + P[edges[:, 1]])
    half2 = 0.25 * ((P[edges[:, 0]] - P[edges[:, 1]]) ** 2).sum(1)
    fe2 = np.full(edges.shape[0], np.nan)
    # propagate from the triangles (decreasing dimension): not-Gabriel edges take the smallest incident value
    inside = ((P[opp] - mid[inv]) ** 2).sum(1) < half2[inv]          # opposite vertex strictly inside the diametral disk
    order = np.argsort(ft2[t_id], kind="stable")[::-1]               # so that the minimum is written last
    for j in order:
        e = inv[j]
        if not np.isnan(fe2[e]):
            fe2[e] = min(fe2[e], ft2[t_id[j]])
        elif inside[j]:
            fe2[e] = ft2[t_id[j]]
    # an edge that got a value from one triangle must still take the min with its other triangle
    for j in range(e_all.shape[0]):
        e = inv[j]
        if not np.isnan(fe2[e]):
            fe2[e] = min(fe2[e], ft2[t_id[j]])
    gabriel = np.isnan(fe2)
    fe2[gabriel] = half2[gabriel]
    verts = np.arange(P.shape[0], dtype=np.int64)[:, None]
    return (verts, edges, tri), (np.zeros(P.shape[0]), np.sqrt(fe2), np.sqrt(ft2))


def bottleneck_distance(A: np.ndarray, B: np.ndarray) -> float:
    """Bottleneck distance between persistence diagrams ``A`` (n,2) and ``B`` (m,2) of (birth, death) pairs,
    death possibly +inf.  L-infinity ground metric; a point may be matched to its projection on the diagonal at
    cost (death - birth) / 2; essential classes (death = inf) are matched among themselves by sorted birth (the
    optimal matching on a line), and give +inf when their numbers differ."""
    from scipy.sparse import csr_matrix
    from scipy.sparse.csgraph import maximum_bipartite_matching

    A = np.asarray(A, dtype=np.float64).reshape(-1, 2)
    B = np.asarray(B, dtype=np.float64).reshape(-1, 2)
    ea, eb = A[np.isinf(A[:, 1])], B[np.isinf(B[:, 1])]
    if ea.shape[0] != eb.shape[0]:
        return float("inf")
    ess = float(np.abs(np.sort(ea[:, 0]) - np.sort(eb[:, 0])).max()) if ea.shape[0] else 0.0
    A, B = A[np.isfinite(A[:, 1])], B[np.isfinite(B[:, 1])]
    A, B = A[A[:, 1] > A[:, 0]], B[B[:, 1] > B[:, 0]]
    n, m = A.shape[0], B.shape[0]
    if n + m == 0:
        return ess
    da = 0.5 * (A[:, 1] - A[:, 0])      # cost of sending a point of A to the diagonal
    db = 0.5 * (B[:, 1] - B[:, 0])
    # left nodes: A (n) then diagonal copies of B (m); right nodes: B (m) then diagonal copies of A (n)
    C = np.zeros((n + m, m + n))
    if n and m:
        C[:n, :m] = np.maximum(np.abs(A[:, None, 0] - B[None, :, 0]), np.abs(A[:, None, 1] - B[None, :, 1]))
    C[:n, m:] = np.inf
    C[n:, :m] = np.inf
    C[np.arange(n), m + np.arange(n)] = da           # a_i <-> its own projection
    C[n + np.arange(m), np.arange(m)] = db           # b_j <-> its own projection
    C[n:, m:] = 0.0                                  # projection <-> projection: free
    cand = np.unique(C[np.isfinite(C)])

    def feasible(delta: float) -> bool:
        g = csr_matrix(C <= delta)
        match = maximum_bipartite_matching(g, perm_type="column")
        return bool((match >= 0).all())

    lo, hi = 0, cand.shape[0] - 1
    while lo < hi:
        mid = (lo + hi) // 2
        if feasible(float(cand[mid])):
            hi = mid
        else:
            lo = mid + 1
    return max(float(cand[lo]), ess)
