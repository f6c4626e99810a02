"""CPU oracle for the Flood-complex coverage sweep.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module; the product (``flooder_amd``) never does.

It restates, in numpy / scipy, the reference algorithm of plus-rkwitt/flooder for the hot
path named in BASELINE.json (citations are into the reference repository):

* ``generate_grid``             -> ``flooder/core.py:346-402``
* ``generate_uniform_weights``  -> ``flooder/core.py:405-427`` (CPU ``torch.rand`` stream)
* ``ball_prep``                 -> ``flooder/core.py:156-179``
* ``flood_complex_oracle``      -> ``flooder/core.py:32-288`` CPU branch (kd-tree, float64
                                   distances: ``core.py:127-128, 197-199``)
* ``masked_min_dist``           -> the GPU formulation: ``compute_mask_kernel``
                                   (``flooder/triton_kernels.py:99-158``: squared distance
                                   ``<=`` squared radius) followed by
                                   ``compute_filtration_kernel`` (``triton_kernels.py:12-45``:
                                   direct-difference squared distance, row min, sqrt)
* ``exact_fps``                 -> the selection order of ``fpsample.bucket_fps_kdline_sampling``
                                   (fpsample 0.3.3, pinned in the reference's ``environment.yml:38``;
                                   a third-party Rust wheel absent from the reference tree).  Bucket
                                   FPS is an acceleration of exact farthest-point sampling and returns
                                   the exact-FPS order; the reference's committed
                                   ``docs/animation/landmarks.csv`` pins that (tests/test_golden.py).
* ``make_filtration_non_decreasing`` -> gudhi 3.11.0 ``Simplex_tree`` semantics (third-party;
                                   call site ``core.py:280``): each simplex is raised to the max of
                                   its own value and its facets' values, by increasing dimension.

Pinning: ``tests/golden/*.npz`` are outputs of the *imported reference itself* run in the build
container (generator: ``oracle/make_goldens.py``), and ``tests/golden/docs_animation_*.csv`` are
the reference's own committed known-answer values; ``tests/test_oracle.py`` checks this module
against all of them.
"""

from __future__ import annotations

import itertools
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

# --------------------------------------------------------------------------- sampling


def generate_grid(n: int, dim: int, dtype=np.float32):
    """Barycentric lattice on the unit ``dim``-simplex with ``n`` points per edge.

    Returns (weights (C, dim+1), vertex_idxs, face_idxs) with the layout of ``core.py:346-402``:
    rows are the compositions of n-1 into dim+1 parts in ``itertools.combinations`` order, divided
    by n-1; ``face_idxs[k][f]`` lists the grid rows lying on the f-th face of codimension k and
    ``vertex_idxs[k][f]`` that face's vertex columns.
    """
    combs = np.array(list(itertools.combinations(range(n + dim - 1), dim)), dtype=np.int64)
    combs = combs.reshape(-1, dim)
    padded = np.concatenate(
        [np.full((combs.shape[0], 1), -1), combs, np.full((combs.shape[0], 1), n + dim - 1)], axis=1
    )
    grid = np.diff(padded, axis=1) - 1
    face_idxs, vertex_idxs = [], []
    axes = np.arange(dim + 1)
    for k in range(dim + 1):
        f_k, v_k = [], []
        for comb in itertools.combinations(range(dim + 1), k):
            if len(comb) == 0:
                mask = np.ones(len(grid), dtype=bool)
            else:
                mask = (grid[:, list(comb)] == 0).all(axis=1)
            f_k.append(np.nonzero(mask)[0])
            v_k.append(axes[~np.isin(axes, comb)])
        face_idxs.append(np.stack(f_k))
        vertex_idxs.append(np.stack(v_k))
    # the reference divides an int64 tensor by (n-1) into a tensor of the target dtype
    # (core.py:400-401); torch computes that quotient in the output dtype.
    weights = (grid.astype(dtype) / dtype(n - 1)).astype(dtype)
    return weights, vertex_idxs, face_idxs


def generate_uniform_weights(num_rand: int, dim: int, dtype=np.float32):
    """Dirichlet(1,..,1) weights drawn exactly as ``core.py:405-427`` does: from the global CPU
    ``torch.rand`` stream (so seeding torch reproduces the reference's samples)."""
    import torch

    if dim == 0:
        return np.ones((num_rand, 1), dtype=dtype)
    tdtype = torch.float32 if dtype == np.float32 else torch.float64
    w = -torch.log(1 - torch.rand(num_rand, dim + 1)).to(dtype=tdtype)
    w = w / w.sum(dim=1, keepdim=True)
    return w.numpy()


# --------------------------------------------------------------------------- complex


def delaunay_buckets(landmarks: np.ndarray, max_dimension: int) -> List[List[Tuple[int, ...]]]:
    """Simplices of the Delaunay complex of ``landmarks`` bucketed by dimension, ascending ids
    (what ``core.py:130-138`` reads out of ``gudhi.DelaunayComplex``)."""
    from scipy.spatial import Delaunay

    lm = np.asarray(landmarks, dtype=np.float64)
    n, dim = lm.shape
    if dim == 1:
        order = np.argsort(lm[:, 0], kind="stable")
        cells = [tuple(sorted((int(a), int(b)))) for a, b in zip(order[:-1], order[1:])]
    elif n <= dim:
        cells = [tuple(range(n))]
    else:
        cells = [tuple(sorted(int(v) for v in c)) for c in Delaunay(lm).simplices]
    all_simplices = set((i,) for i in range(n))
    for c in cells:
        for k in range(1, len(c) + 1):
            all_simplices.update(itertools.combinations(c, k))
    buckets: List[List[Tuple[int, ...]]] = [[] for _ in range(max_dimension + 1)]
    for s in sorted(all_simplices):
        if len(s) <= max_dimension + 1:
            buckets[len(s) - 1].append(s)
    return buckets, sorted(all_simplices)


def ball_prep(simplex_vertices: np.ndarray, d: int):
    """Bounding ball of each simplex as ``core.py:156-172``: centre = midpoint of the longest
    edge, radius = max vertex distance * (1.42 if d > 1 else 1.01) + 1e-3.  float32 arithmetic
    when the input is float32."""
    dt = simplex_vertices.dtype
    diff = simplex_vertices[:, :, None, :] - simplex_vertices[:, None, :, :]
    dist = np.sqrt((diff.astype(np.float64) ** 2).sum(-1)).astype(dt)
    flat = dist.reshape(dist.shape[0], -1).argmax(axis=1)
    i0, i1 = np.unravel_index(flat, (d + 1, d + 1))
    ar = np.arange(simplex_vertices.shape[0])
    centers = ((simplex_vertices[ar, i0] + simplex_vertices[ar, i1]) / dt.type(2.0)).astype(dt)
    vr = np.sqrt(((simplex_vertices - centers[:, None, :]).astype(np.float64) ** 2).sum(-1)).astype(dt)
    radii = (vr.max(axis=1) * dt.type(1.42 if d > 1 else 1.01) + dt.type(1e-3)).astype(dt)
    return centers, radii


def masked_min_dist(samples: np.ndarray, points: np.ndarray, centers: np.ndarray,
                    radii: np.ndarray) -> np.ndarray:
    """GPU formulation of the sweep for a batch of simplices (small inputs only).

    ``samples`` (S,R,d), ``points`` (m,d), ``centers`` (S,d), ``radii`` (S,) -> (S,R) distances in the
    dtype of ``samples``: candidates of simplex s are the points with
    ``sum_k (x_k - c_k)^2 <= r^2`` (``triton_kernels.py:137-148``); the value is
    ``sqrt(min_w sum_k (p_k - x_k)^2)`` accumulated in direct-difference form
    (``triton_kernels.py:36-44``); simplices with no candidate keep ``inf`` (``:70``).
    """
    dt = samples.dtype
    S, R, dim = samples.shape
    out = np.full((S, R), np.inf, dtype=dt)
    for s in range(S):
        d2c = np.zeros(points.shape[0], dtype=dt)
        for k in range(dim):
            diff = points[:, k] - centers[s, k]
            d2c += diff * diff
        cand = points[d2c <= radii[s] * radii[s]]
        if cand.shape[0] == 0:
            continue
        best = np.full(R, np.inf, dtype=dt)
        for lo in range(0, cand.shape[0], 4096):
            blk = cand[lo:lo + 4096]
            acc = np.zeros((R, blk.shape[0]), dtype=dt)
            for k in range(dim):
                diff = samples[s, :, k][:, None] - blk[:, k][None, :]
                acc += diff * diff
            best = np.minimum(best, acc.min(axis=1))
        out[s] = np.sqrt(best)
    return out


def make_filtration_non_decreasing(filt: Dict[Tuple[int, ...], float]) -> Dict[Tuple[int, ...], float]:
    """gudhi ``make_filtration_non_decreasing``: by increasing dimension, raise each simplex to the
    maximum of its own value and its facets' (already raised) values."""
    out = dict(filt)
    for s in sorted(out, key=len):
        if len(s) == 1:
            continue
        m = out[s]
        for j in range(len(s)):
            f = s[:j] + s[j + 1:]
            if f in out and not np.isnan(out[f]):
                m = out[f] if np.isnan(m) else max(m, out[f])
        out[s] = m
    return out


def flood_complex_oracle(points: np.ndarray, landmarks: np.ndarray, max_dimension: Optional[int] = None,
                         points_per_edge: Optional[int] = 30, num_rand: Optional[int] = None,
                         mode: str = "kdtree", workers: int = 1,
                         return_raw: bool = False):
    """Restatement of ``flood_complex`` (``core.py:32-288``).

    mode "kdtree": the reference CPU branch - ``scipy.spatial.KDTree(points).query(samples)``
    (``core.py:127-128,197-199``), float64 distances.  mode "masked": the GPU formulation
    (ball mask + direct-difference minimum in the input dtype, ``core.py:200-226``).
    Returns the simplex -> filtration dict after monotonisation (``core.py:278-288``); with
    ``return_raw`` also the dict before monotonisation.
    """
    from scipy.spatial import KDTree

    points = np.ascontiguousarray(points)
    landmarks = np.ascontiguousarray(landmarks)
    dt = points.dtype
    if max_dimension is None:
        max_dimension = points.shape[1]
    kdtree = KDTree(points) if mode == "kdtree" else None
    buckets, all_simplices = delaunay_buckets(landmarks, max_dimension)
    out: Dict[Tuple[int, ...], float] = {}
    axis = int(np.argmax(points.max(axis=0) - points.min(axis=0)))
    for d in range(max_dimension + 1):
        if num_rand is None and d < max_dimension:
            continue
        if len(buckets[d]) == 0:
            continue
        simp = np.array(buckets[d], dtype=np.int64)
        verts = landmarks[simp]  # (S, d+1, dim)
        centers, radii = ball_prep(verts, d)
        order = np.argsort(centers[:, axis], kind="stable")
        verts, centers, radii, simp = verts[order], centers[order], radii[order], simp[order]
        if num_rand is None:
            weights, vertex_idxs, face_idxs = generate_grid(points_per_edge, max_dimension, dt.type)
        else:
            weights = generate_uniform_weights(num_rand, d, dt.type)
        samples = np.matmul(weights[None], verts).astype(dt)  # (S,R,dim), core.py:188
        if mode == "kdtree":
            dist, _ = kdtree.query(samples, workers=workers)
        else:
            dist = masked_min_dist(samples, points, centers, radii)
        if num_rand is None:
            for f_idx, v_idx in zip(face_idxs, vertex_idxs):
                faces = simp[:, v_idx].reshape(-1, v_idx.shape[1])
                vals = dist[:, f_idx].max(axis=2).reshape(-1)
                out.update(zip(map(tuple, faces.tolist()), vals.tolist()))
        else:
            vals = dist.max(axis=1)
            out.update(zip(map(tuple, simp.tolist()), vals.tolist()))
    raw = dict(out)
    full = {s: out.get(s, float("nan")) for s in all_simplices}
    mono = make_filtration_non_decreasing(full)
    if return_raw:
        return mono, raw
    return mono


# --------------------------------------------------------------------------- landmarks


def exact_fps(points: np.ndarray, n_lms: int, start_idx: int = 0) -> np.ndarray:
    """Exact farthest-point sampling order (float64 distances, first-index tie break)."""
    pts = np.asarray(points, dtype=np.float64)
    n = pts.shape[0]
    n_lms = min(n_lms, n)
    idx = np.empty(n_lms, dtype=np.int64)
    idx[0] = start_idx
    d2 = ((pts - pts[start_idx]) ** 2).sum(axis=1)
    for i in range(1, n_lms):
        j = int(np.argmax(d2))
        idx[i] = j
        d2 = np.minimum(d2, ((pts - pts[j]) ** 2).sum(axis=1))
    return idx


def check_fps(points: np.ndarray, idx: np.ndarray, rtol: float = 2e-6) -> Dict[str, float]:
    """Replay a farthest-point selection ``idx`` in float64 and check the defining property of every pick:
    it is a farthest point from the picks before it.  Pick i must attain the maximum of the running minimum
    squared distance exactly (then it is what ``exact_fps`` would pick, up to the order of exact ties) or within
    ``rtol`` relative - a float32 implementation cannot separate candidates whose float64 distances differ by
    less than its rounding (about 1e-7 relative per distance), and on clouds of 10^7 points such near-ties do
    occur.  Raises AssertionError otherwise.  Returns the number of exact arg-max picks (and how many of them are
    also the FIRST index holding the maximum, ``exact_fps``' tie break: when every pick is, ``idx`` equals
    ``exact_fps(points, len(idx), idx[0])``), the number of near-tie picks and the worst relative shortfall.  Cost: one pass over the cloud per pick (fine for a few hundred picks at
    16 M points)."""
    pts = np.asarray(points)
    cols = [np.ascontiguousarray(pts[:, k], dtype=np.float64) for k in range(pts.shape[1])]
    idx = np.asarray(idx, dtype=np.int64)
    assert len(set(idx.tolist())) == len(idx) or pts.shape[0] < len(idx), "a point was picked twice"
    d2 = np.full(pts.shape[0], np.inf)
    tmp = np.empty_like(d2)
    acc = np.empty_like(d2)
    n_exact = n_near = n_first = 0
    worst = 0.0
    for i, j in enumerate(idx):
        if i > 0:
            m = float(d2.max())
            got = float(d2[j])
            if got == m:
                n_exact += 1
                n_first += int(j == int(np.argmax(d2)))   # exact_fps' tie break: the first index holding the max
            else:
                short = (m - got) / m if m > 0 else 0.0
                assert short <= rtol, f"pick {i} (index {j}) is not a farthest point: d2 {got:.9g} vs max {m:.9g}"
                n_near += 1
                worst = max(worst, short)
        acc.fill(0.0)
        for c in cols:
            np.subtract(c, c[j], out=tmp)
            np.multiply(tmp, tmp, out=tmp)
            acc += tmp
        np.minimum(d2, acc, out=d2)
    return dict(exact=n_exact, exact_first_index=n_first, near_ties=n_near, worst_rel_shortfall=worst)


# --------------------------------------------------------------------------- inputs


def noisy_torus(n: int, seed: int, R: float = 3.0, r: float = 1.0, noise_std: float = 0.02) -> np.ndarray:
    """Benchmark input of the reference's tests (``synthetic_data_generators.py:258-269``):
    two successive ``torch.rand(n)`` draws for the angles, then ``randn_like`` noise."""
    import torch

    torch.manual_seed(seed)
    theta = torch.rand(n) * 2 * torch.pi
    phi = torch.rand(n) * 2 * torch.pi
    x = (R + r * torch.cos(phi)) * torch.cos(theta)
    y = (R + r * torch.cos(phi)) * torch.sin(theta)
    z = r * torch.sin(phi)
    p = torch.stack((x, y, z), dim=1)
    return (p + torch.randn_like(p) * noise_std).numpy()


# --------------------------------------------------------------------------- timed CPU baseline
def kdtree_sweep_sample(points: np.ndarray, landmarks: np.ndarray, simplices: np.ndarray,
                        points_per_edge: int, d: int, n_sample: int, seed: int = 0, workers: int = 1):
    """The reference CPU path (``core.py:127-128, 188, 197-199, 251-257``) on a random subset of the
    top-dimensional ``simplices`` (rows of landmark ids, in the caller's order): kd-tree build,
    ``weights @ vertices`` samples, ``KDTree.query`` (float64), per-face maxima with the faces of all
    codimensions concatenated (codimension-major, the order of ``generate_grid``'s ``face_idxs``).

    Returns ``picked`` (row indices, ascending), ``face_max`` (n_sample, F), and wall-clock seconds of
    the tree build and of sample generation + query.  Used by ``bench.py`` as the CPU baseline and as
    a parity check of the GPU result at full size.
    """
    import time
    from scipy.spatial import KDTree

    points = np.ascontiguousarray(points)
    dt = points.dtype
    rng = np.random.default_rng(seed)
    picked = np.sort(rng.choice(simplices.shape[0], size=n_sample, replace=False))
    t0 = time.perf_counter()
    tree = KDTree(points)
    build_s = time.perf_counter() - t0
    weights, vertex_idxs, face_idxs = generate_grid(points_per_edge, d, dt.type)
    t0 = time.perf_counter()
    verts = landmarks[simplices[picked]]
    samples = np.matmul(weights[None], verts).astype(dt)
    dist, _ = tree.query(samples, workers=workers)
    cols = []
    for f_idx in face_idxs:
        cols.append(dist[:, f_idx].max(axis=2))
    face_max = np.concatenate(cols, axis=1)
    query_s = time.perf_counter() - t0
    return dict(picked=picked, face_max=face_max, build_s=build_s, query_s=query_s, n_sample=int(n_sample))
