"""Generate ``tests/golden/*.npz`` by running the REFERENCE ITSELF in the build container.

Run from the repo root, in the container that has ``/root/reference`` mounted::

    TRITON_INTERPRET=1 python oracle/make_goldens.py

The reference package is imported from ``/root/reference`` (read-only, never copied).  Two of
its third-party wheels are absent from the image, so two import shims are injected into
``sys.modules`` before the import:

* ``gudhi``    -> ``DelaunayComplex`` over Qhull (``scipy.spatial.Delaunay``) and a dict-backed simplex
                 tree (``get_simplices / assign_filtration / make_filtration_non_decreasing``).  Qhull and
                 gudhi/CGAL give identical simplices on the three committed 1000-landmark clouds of the
                 reference (``docs/visualization/*/tetrahedra.csv``).
* ``fpsample`` -> exact farthest-point sampling (what bucket-FPS computes).

What is recorded per case: the inputs (points, landmarks, kwargs, torch seed) and the outputs of

* the reference CPU branch of ``flood_complex`` (``flooder/core.py:197-199``), run unmodified;
* the reference Triton kernels ``compute_mask`` / ``compute_filtration`` run unmodified on CPU tensors
  under ``TRITON_INTERPRET=1`` for one batch (kernel-level vectors);
* ``generate_grid`` / ``generate_uniform_weights`` outputs.

Only data (arrays of numbers) is written; no reference source text is stored.
"""

from __future__ import annotations

import itertools
import os
import sys
import types

import numpy as np

os.environ.setdefault("TRITON_INTERPRET", "1")
REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


# ------------------------------------------------------------------ import shims
class _ShimTree:
    def __init__(self, simplices):
        self._f = {s: float("nan") for s in simplices}

    def get_simplices(self):
        for s in sorted(self._f):
            yield list(s), self._f[s]

    def assign_filtration(self, simplex, value):
        s = tuple(simplex)
        if s in self._f:
            self._f[s] = float(value)

    def make_filtration_non_decreasing(self):
        for s in sorted(self._f, key=len):
            if len(s) == 1:
                continue
            m = self._f[s]
            for j in range(len(s)):
                fv = self._f.get(s[:j] + s[j + 1:], float("nan"))
                if fv == fv:
                    m = fv if m != m else max(m, fv)
            self._f[s] = m


class _ShimDelaunay:
    def __init__(self, points):
        import torch
        from scipy.spatial import Delaunay

        if isinstance(points, torch.Tensor):
            points = points.detach().cpu().numpy()
        pts = np.asarray(points, dtype=np.float64)
        cells = Delaunay(pts).simplices
        simplices = set((i,) for i in range(len(pts)))
        for c in cells:
            c = tuple(sorted(int(v) for v in c))
            for k in range(1, len(c) + 1):
                simplices.update(itertools.combinations(c, k))
        self._simplices = simplices

    def create_simplex_tree(self, *a, **k):
        return _ShimTree(self._simplices)


def _shim_fps(points, n, h=None, start_idx=None):
    pts = np.asarray(points, dtype=np.float64)
    if start_idx is None:
        start_idx = 0
    idx = np.empty(n, dtype=np.int64)
    idx[0] = start_idx
    d2 = ((pts - pts[start_idx]) ** 2).sum(1)
    for i in range(1, n):
        j = int(np.argmax(d2))
        idx[i] = j
        d2 = np.minimum(d2, ((pts - pts[j]) ** 2).sum(1))
    return idx


def _install_shims():
    g = types.ModuleType("gudhi")
    g.DelaunayComplex = _ShimDelaunay
    g.SimplexTree = _ShimTree
    f = types.ModuleType("fpsample")
    f.bucket_fps_kdline_sampling = _shim_fps
    sys.modules["gudhi"] = g
    sys.modules["fpsample"] = f
    sys.path.insert(0, REF)


def _dict_to_arrays(fc):
    keys = sorted(fc)
    width = max(len(k) for k in keys)
    arr = np.full((len(keys), width), -1, dtype=np.int64)
    for i, k in enumerate(keys):
        arr[i, : len(k)] = k
    return arr, np.array([fc[k] for k in keys], dtype=np.float64)


def main():
    _install_shims()
    import torch
    import flooder  # the reference package
    from flooder import core as rcore
    from flooder.synthetic_data_generators import (
        generate_noisy_torus_points_3d,
        generate_figure_eight_points_2d,
        generate_swiss_cheese_points,
    )

    os.makedirs(OUT, exist_ok=True)

    # ---- 1. generate_grid / generate_uniform_weights vectors
    grids = {}
    for n, dim in [(2, 1), (5, 2), (8, 3), (30, 3), (4, 4), (3, 6), (20, 2)]:
        w, v_idx, f_idx = rcore.generate_grid(n, dim, torch.device("cpu"), torch.float32)
        grids[f"w_{n}_{dim}"] = w.numpy()
        for k in range(dim + 1):
            grids[f"v_{n}_{dim}_{k}"] = v_idx[k].numpy()
            grids[f"f_{n}_{dim}_{k}"] = f_idx[k].numpy()
    torch.manual_seed(42)
    grids["u_64_3_seed42"] = rcore.generate_uniform_weights(64, 3, torch.device("cpu"), torch.float32).numpy()
    grids["u_7_0"] = rcore.generate_uniform_weights(7, 0, torch.device("cpu"), torch.float32).numpy()
    np.savez_compressed(os.path.join(OUT, "grid_vectors.npz"), **grids)
    print("grid vectors:", len(grids))

    # ---- 2. end-to-end flood_complex, reference CPU branch
    cases = [
        # name, generator, n_pts, n_lms, kwargs
        ("torus3d_grid", "torus", 3000, 60, dict(points_per_edge=8)),
        ("torus3d_grid30", "torus", 2000, 40, dict(points_per_edge=30)),
        ("torus3d_rand", "torus", 3000, 60, dict(points_per_edge=None, num_rand=64)),
        ("torus3d_maxdim2", "torus", 2000, 50, dict(points_per_edge=10, max_dimension=2)),
        ("eight2d_grid", "eight", 1500, 80, dict(points_per_edge=20)),
        ("eight2d_rand", "eight", 1500, 80, dict(points_per_edge=None, num_rand=100)),
        ("cheese3d_grid", "cheese", 4000, 100, dict(points_per_edge=6)),
        ("gauss4d_grid", "gauss4", 1500, 30, dict(points_per_edge=4)),
        ("gauss6d_maxdim2", "gauss6", 1200, 24, dict(points_per_edge=5, max_dimension=2)),
        ("lms_eq_pts2d", "eight", 120, 120, dict(points_per_edge=12)),
    ]
    for name, gen, n_pts, n_lms, kw in cases:
        torch.manual_seed(42)
        np.random.seed(42)
        if gen == "torus":
            pts = generate_noisy_torus_points_3d(n_pts)
        elif gen == "eight":
            pts = generate_figure_eight_points_2d(n_pts)
        elif gen == "cheese":
            pts = generate_swiss_cheese_points(n_pts)[0]
        elif gen == "gauss4":
            pts = torch.randn(n_pts, 4)
        elif gen == "gauss6":
            pts = torch.randn(n_pts, 6)
        pts = pts.to(torch.float32).contiguous()
        lms = flooder.generate_landmarks(pts, n_lms, start_idx=0)
        torch.manual_seed(7)
        fc32 = flooder.flood_complex(pts, lms, use_triton=False, **kw)
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            torch.manual_seed(7)
            fc64 = flooder.flood_complex(pts.double(), lms.double(), use_triton=False, **kw)
        k32, v32 = _dict_to_arrays(fc32)
        k64, v64 = _dict_to_arrays(fc64)
        assert (k32 == k64).all()
        meta = dict(kw)
        np.savez_compressed(
            os.path.join(OUT, f"e2e_{name}.npz"),
            points=pts.numpy(), landmarks=lms.numpy(), simplices=k32, filtration_f32=v32,
            filtration_f64=v64, weight_seed=np.int64(7),
            points_per_edge=np.int64(meta.get("points_per_edge") or -1),
            num_rand=np.int64(meta.get("num_rand") or -1),
            max_dimension=np.int64(meta.get("max_dimension", -1)),
        )
        print(f"e2e {name}: {len(k32)} simplices, max|f32-f64| = {np.abs(v32 - v64).max():.3g}")

    # ---- 3. kernel-level vectors: reference Triton kernels under the interpreter, one batch
    from flooder.triton_kernels import compute_mask, compute_filtration

    torch.manual_seed(3)
    for name, dim, m, B, R in [("k3d", 3, 1500, 5, 40), ("k2d", 2, 900, 3, 33), ("k5d", 5, 700, 4, 17)]:
        pts = torch.rand(m, dim)
        centers = 0.25 + 0.5 * torch.rand(B, dim)
        radii = 0.15 + 0.2 * torch.rand(B)
        radii[0] = 1e-4  # an empty ball: row stays +inf
        x = centers[:, None, :] + 0.1 * torch.randn(B, R, dim)
        mask = compute_mask(pts, centers, radii, 16, 512, 512)
        row_idx, col_idx = torch.nonzero(mask, as_tuple=True)
        dist = compute_filtration(x, pts, row_idx, col_idx, 512, 16)
        np.savez_compressed(
            os.path.join(OUT, f"kernel_{name}.npz"),
            points=pts.numpy(), centers=centers.numpy(), radii=radii.numpy(), samples=x.numpy(),
            mask_rowsum=mask[:, :m].sum(1).numpy().astype(np.int64),
            mask=np.packbits(mask[:, :m].numpy(), axis=1), min_dist=dist.numpy(),
        )
        print(f"kernel {name}: candidates per ball {mask[:, :m].sum(1).tolist()}, "
              f"inf rows {int(torch.isinf(dist).all(1).sum())}")
    make_generator_vectors()


def make_generator_vectors():
    """Section 4 on its own: ``python oracle/make_goldens.py generators``."""
    _install_shims()
    import torch

    os.makedirs(OUT, exist_ok=True)
    # ---- 4. synthetic generators: the reference's clouds for fixed seeds (CPU draws)
    from flooder import synthetic_data_generators as sg

    gen = {}
    gen["fig8_plain"] = sg.generate_figure_eight_points_2d(300, seed=5).numpy()
    gen["fig8_gauss"] = sg.generate_figure_eight_points_2d(200, r_bounds=(0.1, 0.25), noise_std=0.01, seed=6).numpy()
    gen["fig8_uniform"] = sg.generate_figure_eight_points_2d(200, noise_std=0.02, noise_kind="uniform", seed=7).numpy()
    p, c, r = sg.generate_swiss_cheese_points(500, (0.0, 0.0, 0.0), (1.0, 1.0, 1.0), 6, (0.1, 0.2), seed=11)
    gen["cheese3_points"], gen["cheese3_centres"], gen["cheese3_radii"] = p.numpy(), c.numpy(), r.numpy()
    p, c, r = sg.generate_swiss_cheese_points(400, (0.0, -1.0), (2.0, 1.0), 3, (0.15, 0.3), seed=12)
    gen["cheese2_points"], gen["cheese2_centres"], gen["cheese2_radii"] = p.numpy(), c.numpy(), r.numpy()
    gen["annulus"] = sg.generate_annulus_points_2d(300, torch.tensor([0.5, -0.25]), 1.5, 0.4, seed=13).numpy()
    gen["torus"] = sg.generate_noisy_torus_points_3d(400, R=3.0, r=1.0, noise_std=0.02, seed=14).numpy()
    np.savez_compressed(os.path.join(OUT, "generators.npz"), **gen)
    print("generators:", {k: v.shape for k, v in gen.items()})


def make_visualization_vectors():
    """Section 5 on its own: ``python oracle/make_goldens.py visualization``.

    The reference commits three 1000-landmark clouds together with the simplices gudhi (CGAL) produced for them and
    their filtration values (``docs/visualization/{virus,coral,lockwasher}/{landmarks,edges,triangles,tetrahedra}.csv``,
    the data behind ``docs/visualizations.md``).  They are the only output of the REAL ``gudhi.DelaunayComplex`` in the
    repository: the fixtures keep the numbers - landmark coordinates as float64, vertex ids as int32, values as float64 -
    so that the native Delaunay routines are pinned to gudhi itself, not only to Qhull."""
    os.makedirs(OUT, exist_ok=True)
    for name in ("virus", "coral", "lockwasher"):
        d = os.path.join(REF, "docs", "visualization", name)
        lms = np.loadtxt(os.path.join(d, "landmarks.csv"), delimiter=",", dtype=np.float64)
        out = {"landmarks": lms}
        for kind in ("edges", "triangles", "tetrahedra"):
            t = np.loadtxt(os.path.join(d, f"{kind}.csv"), delimiter=",", dtype=np.float64)
            ids = t[:, :-1]
            assert (ids == np.round(ids)).all()
            out[kind] = ids.astype(np.int32)
            out[f"{kind}_filtration"] = t[:, -1]
        np.savez_compressed(os.path.join(OUT, f"docs_visualization_{name}.npz"), **out)
        print(f"visualization {name}: {lms.shape[0]} landmarks, "
              f"{ {k: v.shape[0] for k, v in out.items() if k in ('edges', 'triangles', 'tetrahedra')} }")


if __name__ == "__main__":
    if sys.argv[1:] == ["generators"]:
        make_generator_vectors()
    elif sys.argv[1:] == ["visualization"]:
        make_visualization_vectors()
    else:
        main()
        make_visualization_vectors()
