"""Shared helpers of the parity tests."""
import glob
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# Parity bar (BASELINE.json north_star): filtration values within 1e-5 relative in fp32.
# The absolute floor covers the exact zeros (vertices: a landmark is its own nearest point) and the
# input quantisation: a float32 sample coordinate of magnitude c carries ulp(c) ~ 1.2e-7*c, which no
# float32 implementation can undercut; COORD_ULPS of it are allowed (observed errors stay below 1 ulp).
RTOL = 1e-5
COORD_ULPS = 2.0


def tolerances(points):
    scale = float(np.abs(np.asarray(points)).max())
    return RTOL, COORD_ULPS * np.finfo(np.float32).eps * max(scale, 1e-30)


def assert_baseline_gate(got, ref, points, what=""):
    """BASELINE.md section 3's parity gate, as written there (and as ``bench.py`` applies it): relative error
    ``|got - ref| / max(|ref|, 1e-6 * diameter) <= 1e-5`` - NO absolute ulp term.  The diameter is taken as the
    longest side of the cloud's bounding box (never more than the true diameter: the stricter reading).  Used on the
    BASELINE configurations at full size, where the HIP path has 60x margin; the small goldens keep the ulp term of
    ``assert_close_filtration`` (two of them differ from the reference's CPU branch by float32 matmul order)."""
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    P = np.asarray(points)
    floor = 1e-6 * float((P.max(axis=0) - P.min(axis=0)).max())
    fin = np.isfinite(ref)
    assert (np.isfinite(got) == fin).all(), f"{what}: infinities differ"
    rel = np.abs(got[fin] - ref[fin]) / np.maximum(np.abs(ref[fin]), floor)
    assert rel.size == 0 or rel.max() <= RTOL, (
        f"{what}: BASELINE gate (rel 1e-5, floor 1e-6 x diameter) failed: worst {rel.max():.3e} "
        f"on {int((rel > RTOL).sum())}/{rel.size} values")
    return float(rel.max()) if rel.size else 0.0


def assert_close_filtration(got, ref, points, what="", strict=False):
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    if strict:
        assert_baseline_gate(got, ref, points, what)
    rtol, atol = tolerances(points)
    both_inf = np.isinf(got) & np.isinf(ref)
    both_nan = np.isnan(got) & np.isnan(ref)
    err = np.where(both_inf | both_nan, 0.0, np.abs(got - ref))
    bound = atol + rtol * np.abs(np.where(np.isfinite(ref), ref, 0.0))
    bad = ~(err <= bound)
    assert not bad.any(), (
        f"{what}: {int(bad.sum())}/{bad.size} values off; worst abs err {np.nanmax(err):.3e} "
        f"(rtol {rtol:g}, atol {atol:.3e})")
    return float(np.nanmax(err)) if err.size else 0.0


def e2e_cases():
    return sorted(os.path.basename(p)[4:-4] for p in glob.glob(os.path.join(GOLDEN, "e2e_*.npz")))


def load_e2e(name):
    z = np.load(os.path.join(GOLDEN, f"e2e_{name}.npz"))
    ppe, nr, md = int(z["points_per_edge"]), int(z["num_rand"]), int(z["max_dimension"])
    kwargs = dict(points_per_edge=None if ppe < 0 else ppe, num_rand=None if nr < 0 else nr,
                  max_dimension=None if md < 0 else md)
    keys = [tuple(int(v) for v in row if v >= 0) for row in z["simplices"]]
    return z, kwargs, keys


def dict_values(fc, keys):
    return np.array([fc[k] for k in keys], dtype=np.float64)


def kdtree_face_values(tree, landmarks, simplices, points_per_edge, d, chunk_queries=4_000_000, dtype=np.float32):
    """Reference values of all dimension-``d`` simplices (rows of landmark ids): the reference's CPU computation
    (``core.py:188, 197-199, 251-257``) - barycentric lattice of ``points_per_edge`` per edge on the simplex itself,
    nearest neighbour of every sample by ``tree.query`` on all host cores, maximum over the samples.  Queried in
    chunks so that the (queries, dim) float64 copy scipy makes stays below a few hundred MB."""
    from oracle import flood_oracle as fo

    w, _, _ = fo.generate_grid(points_per_edge, d, dtype)
    R = w.shape[0]
    per = max(1, chunk_queries // R)
    out = np.empty(len(simplices), dtype=np.float64)
    for b in range(0, len(simplices), per):
        verts = landmarks[simplices[b:b + per]]
        samples = np.matmul(w[None], verts).astype(dtype)
        dist, _ = tree.query(samples, workers=-1)
        out[b:b + per] = dist.max(axis=1)
    return out


def assert_tree_matches_kdtree(st, points, landmarks, points_per_edge, top, what, pick_top=None, lower=True,
                               strict=False):
    """Every simplex of dimension ``top`` of the simplex tree ``st`` (or the rows ``pick_top`` of that table), and
    - ``lower`` - every simplex of the dimensions below, against the kd-tree over ALL ``points``: a face of a swept
    simplex carries the maximum over ITS OWN lattice samples (the zero weights of the parent's lattice rows contribute
    exact zeros), so each table is checked with the lattice of its own dimension.  ``strict``: BASELINE.md's gate
    (``assert_baseline_gate``) on top of the usual one.  Returns the number of values checked."""
    from scipy.spatial import cKDTree

    tree = cKDTree(points, balanced_tree=False, compact_nodes=False)
    n = 0
    for d in range(top, 0, -1):
        rows = st.simplices_of_dimension(d)
        vals = st.filtrations_of_dimension(d)
        if d == top and pick_top is not None:
            rows, vals = rows[pick_top], vals[pick_top]
        ref = kdtree_face_values(tree, landmarks, rows, points_per_edge, d, dtype=points.dtype.type)
        assert_close_filtration(vals, ref, points, f"{what}: dimension {d}", strict=strict)
        n += len(rows)
        if not lower:
            break
    return n
