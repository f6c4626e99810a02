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


def assert_close_filtration(got, ref, points, what=""):
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
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
