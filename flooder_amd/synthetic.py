"""Synthetic point clouds of the reference's examples and benchmarks, with an optional ``device``.

Same names, arguments and distributions as ``flooder/synthetic_data_generators.py`` (figure eight :13-70,
swiss cheese :73-172, annulus :175-217, noisy torus :220-269).  On the CPU the random draws are made in the
reference's order and with its generators (numpy for the figure eight, torch for the others), so a seed gives
the same cloud as the reference (``tests/golden/generators.npz``); with ``device="cuda"`` the torch-based
generators draw on the device (16 M - 100 M-point clouds without a host round trip; a different random stream).
"""
from __future__ import annotations

from typing import Optional, Sequence, Tuple

import numpy as np
import torch

__all__ = [
    "generate_figure_eight_points_2d",
    "generate_swiss_cheese_points",
    "generate_annulus_points_2d",
    "generate_noisy_torus_points_3d",
]


def _seed_torch(seed: Optional[int], truthy: bool = False) -> None:
    # the reference seeds on ``if seed:`` for the swiss cheese (seed 0 = unseeded) and ``is not None`` elsewhere
    if (seed if truthy else seed is not None):
        torch.manual_seed(int(seed))


@torch.no_grad()
def generate_figure_eight_points_2d(n: int = 1000, r_bounds: Tuple[float, float] = (0.2, 0.3),
                                    centers=((0.3, 0.5), (0.7, 0.5)), noise_std: float = 0.0,
                                    noise_kind: str = "gaussian", seed: Optional[int] = None) -> torch.Tensor:
    """Two annular lobes, area-uniform in radius (``synthetic_data_generators.py:13-70``; numpy generator)."""
    if noise_kind not in ("gaussian", "uniform"):
        raise ValueError("noise_kind must be 'gaussian' or 'uniform'")
    if seed is not None:
        np.random.seed(seed)
    lobe = np.random.randint(0, 2, size=n)
    c = np.asarray(centers, dtype=np.float64)
    rad = np.sqrt(np.random.uniform(r_bounds[0] ** 2, r_bounds[1] ** 2, size=n))
    ang = np.random.uniform(0.0, 2 * np.pi, size=n)
    xy = np.stack((c[lobe, 0] + rad * np.cos(ang), c[lobe, 1] + rad * np.sin(ang)), axis=1)
    if noise_std > 0:
        if noise_kind == "gaussian":
            xy[:, 0] += np.random.normal(0.0, noise_std, size=n)
            xy[:, 1] += np.random.normal(0.0, noise_std, size=n)
        else:
            xy[:, 0] += np.random.uniform(-noise_std, noise_std, size=n)
            xy[:, 1] += np.random.uniform(-noise_std, noise_std, size=n)
    return torch.tensor(xy, dtype=torch.float32)


@torch.no_grad()
def generate_swiss_cheese_points(n: int = 1000, rect_min: Sequence[float] = (0.0, 0.0, 0.0),
                                 rect_max: Sequence[float] = (1.0, 1.0, 1.0), k: int = 6,
                                 void_radius_range: Tuple[float, float] = (0.1, 0.2), seed: Optional[int] = None,
                                 *, device="cpu", batch_factor: int = 4):
    """Uniform points in a box minus ``k`` balls (``synthetic_data_generators.py:73-172``).

    Returns ``(points (n, d), void centres (k, d), void radii (k,))``.  Voids are placed first (candidates a full
    maximal radius inside the box, accepted while they do not touch a void of an EARLIER round - as in the
    reference, voids accepted in the same round are not tested against each other and may overlap), then points
    are rejection sampled in rounds of ``batch_factor`` times the number still missing."""
    if len(rect_min) != len(rect_max):
        raise AssertionError("rect_min and rect_max must have the same dimension.")
    _seed_torch(seed, truthy=True)
    d = len(rect_min)
    r_lo, r_hi = void_radius_range
    lo = torch.tensor(rect_min, dtype=torch.float32, device=device)
    hi = torch.tensor(rect_max, dtype=torch.float32, device=device)
    centres = torch.empty((0, d), device=device)
    radii = torch.empty((0,), device=device)
    while centres.shape[0] < k:
        missing = k - centres.shape[0]
        batch = max(8, 2 * missing)
        c_new = (lo + r_hi) + (hi - lo - 2 * r_hi) * torch.rand(batch, d, device=device)
        r_new = r_lo + (r_hi - r_lo) * torch.rand(batch, device=device)
        if centres.numel() == 0:
            free = torch.ones(batch, dtype=torch.bool, device=device)
        else:
            free = (torch.cdist(c_new, centres) >= r_new[:, None] + radii[None, :]).all(dim=1)
        pick = free.nonzero(as_tuple=False).squeeze()[:missing]
        centres = torch.cat([centres, c_new[pick]], dim=0)
        radii = torch.cat([radii, r_new[pick]], dim=0)
    pts = torch.empty((0, d), dtype=lo.dtype, device=device)
    missing = n
    while missing:
        cand = lo + (hi - lo) * torch.rand(batch_factor * missing, d, device=device)
        if k:
            outside = (torch.cdist(cand, centres) >= radii[None, :]).all(dim=1)
        else:
            outside = torch.ones(cand.shape[0], dtype=torch.bool, device=device)
        pts = torch.cat([pts, cand[outside][:missing]], dim=0)
        missing = n - pts.shape[0]
    return pts, centres, radii


@torch.no_grad()
def generate_annulus_points_2d(n: int = 1000, center: torch.Tensor = torch.tensor([0.0, 0.0]), radius: float = 1.0,
                               width: float = 0.2, seed: Optional[int] = None, *, device="cpu") -> torch.Tensor:
    """Ring between ``radius - width`` and ``radius`` (``synthetic_data_generators.py:175-217``; the radius is
    ``radius - width + width * sqrt(u)`` exactly as there)."""
    if tuple(center.shape) != (2,):
        raise AssertionError("Center must be a 2D point.")
    if not (radius > 0 and width > 0):
        raise AssertionError("Radius and width must be positive.")
    _seed_torch(seed)
    ang = torch.rand(n, device=device) * 2 * torch.pi
    rad = radius - width + width * torch.sqrt(torch.rand(n, device=device))
    c = center.to(device)
    return torch.stack((c[0] + rad * torch.cos(ang), c[1] + rad * torch.sin(ang)), dim=1)


@torch.no_grad()
def generate_noisy_torus_points_3d(n: int = 1000, R: float = 3.0, r: float = 1.0, noise_std: float = 0.02,
                                   seed: Optional[int] = None, *, device="cpu") -> torch.Tensor:
    """Angles uniform on the torus (major ``R``, minor ``r``) plus isotropic Gaussian noise
    (``synthetic_data_generators.py:220-269``: theta, then phi, then the noise are drawn in this order)."""
    _seed_torch(seed)
    theta = torch.rand(n, device=device) * 2 * torch.pi
    phi = torch.rand(n, device=device) * 2 * torch.pi
    ring = R + r * torch.cos(phi)
    pts = torch.stack((ring * torch.cos(theta), ring * torch.sin(theta), r * torch.sin(phi)), dim=1)
    return pts + torch.randn_like(pts) * noise_std
