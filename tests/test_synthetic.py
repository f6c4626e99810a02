"""Synthetic generators against the reference's own outputs (tests/golden/generators.npz, written by
oracle/make_goldens.py from flooder/synthetic_data_generators.py) and against their defining properties."""
import os

import numpy as np
import pytest
import torch

from flooder_amd import synthetic as sg

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "generators.npz"))


def test_figure_eight_matches_reference_draws():
    np.testing.assert_array_equal(sg.generate_figure_eight_points_2d(300, seed=5).numpy(), GOLD["fig8_plain"])
    got = sg.generate_figure_eight_points_2d(200, r_bounds=(0.1, 0.25), noise_std=0.01, seed=6).numpy()
    np.testing.assert_array_equal(got, GOLD["fig8_gauss"])
    got = sg.generate_figure_eight_points_2d(200, noise_std=0.02, noise_kind="uniform", seed=7).numpy()
    np.testing.assert_array_equal(got, GOLD["fig8_uniform"])
    with pytest.raises(ValueError):
        sg.generate_figure_eight_points_2d(10, noise_std=0.1, noise_kind="laplace")


def test_swiss_cheese_matches_reference_draws():
    p, c, r = sg.generate_swiss_cheese_points(500, (0.0, 0.0, 0.0), (1.0, 1.0, 1.0), 6, (0.1, 0.2), seed=11)
    np.testing.assert_array_equal(p.numpy(), GOLD["cheese3_points"])
    np.testing.assert_array_equal(c.numpy(), GOLD["cheese3_centres"])
    np.testing.assert_array_equal(r.numpy(), GOLD["cheese3_radii"])
    p, c, r = sg.generate_swiss_cheese_points(400, (0.0, -1.0), (2.0, 1.0), 3, (0.15, 0.3), seed=12)
    np.testing.assert_array_equal(p.numpy(), GOLD["cheese2_points"])
    np.testing.assert_array_equal(c.numpy(), GOLD["cheese2_centres"])
    np.testing.assert_array_equal(r.numpy(), GOLD["cheese2_radii"])


def test_swiss_cheese_properties():
    p, c, r = sg.generate_swiss_cheese_points(2000, k=5, seed=3)
    assert p.shape == (2000, 3) and c.shape == (5, 3) and r.shape == (5,)
    assert float(p.min()) >= 0.0 and float(p.max()) <= 1.0
    dist = (p[:, None, :] - c[None, :, :]).norm(dim=2)
    assert bool((dist >= r[None, :] - 1e-6).all())                 # no point inside a void
    assert bool(((r >= 0.1) & (r <= 0.2)).all())
    p0, c0, r0 = sg.generate_swiss_cheese_points(100, k=0, seed=4)
    assert p0.shape == (100, 3) and c0.shape == (0, 3) and r0.shape == (0,)
    with pytest.raises(AssertionError):
        sg.generate_swiss_cheese_points(10, (0.0, 0.0), (1.0, 1.0, 1.0))


def test_annulus_and_torus_match_reference_draws():
    got = sg.generate_annulus_points_2d(300, torch.tensor([0.5, -0.25]), 1.5, 0.4, seed=13).numpy()
    np.testing.assert_array_equal(got, GOLD["annulus"])
    rad = np.linalg.norm(got - np.array([0.5, -0.25], dtype=np.float32), axis=1)
    assert rad.min() >= 1.1 - 1e-5 and rad.max() <= 1.5 + 1e-5
    got = sg.generate_noisy_torus_points_3d(400, R=3.0, r=1.0, noise_std=0.02, seed=14).numpy()
    np.testing.assert_array_equal(got, GOLD["torus"])
    with pytest.raises(AssertionError):
        sg.generate_annulus_points_2d(10, torch.zeros(3))
    with pytest.raises(AssertionError):
        sg.generate_annulus_points_2d(10, radius=-1.0)


@pytest.mark.gpu
def test_generators_on_device():
    dev = torch.device("cuda:0")
    t = sg.generate_noisy_torus_points_3d(100_000, seed=1, device=dev)
    assert t.device.type == "cuda" and t.shape == (100_000, 3)
    rho = torch.sqrt(t[:, 0] ** 2 + t[:, 1] ** 2)
    tube = torch.sqrt((rho - 3.0) ** 2 + t[:, 2] ** 2)
    assert float((tube - 1.0).abs().max()) < 0.2                    # noise_std 0.02: within 10 sigma of the tube
    p, c, r = sg.generate_swiss_cheese_points(200_000, k=6, seed=2, device=dev)
    assert p.device.type == "cuda" and p.shape == (200_000, 3)
    assert bool(((p[:, None, :] - c[None, :, :]).norm(dim=2) >= r[None, :] - 1e-6).all())
    a = sg.generate_annulus_points_2d(50_000, radius=2.0, width=0.5, seed=3, device=dev)
    rad = a.norm(dim=1)
    assert float(rad.min()) >= 1.5 - 1e-4 and float(rad.max()) <= 2.0 + 1e-4
