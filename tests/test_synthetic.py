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


def _ks(a, b):
    """Two-sample Kolmogorov-Smirnov statistic of two 1-D samples."""
    from scipy.stats import ks_2samp

    return float(ks_2samp(np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)).statistic)


@pytest.mark.gpu
def test_device_generators_draw_the_reference_distributions():
    """The device path is another random stream, so it is compared with the CPU path - which equals
    the reference's draws bit for bit (tests above) - as a DISTRIBUTION: Kolmogorov-Smirnov on every coordinate and on
    the radial / tube / void-distance marginals (200 k points each: a statistic below 0.006 is what two samples of one
    law give, a wrong radius law or a missing sqrt gives 0.05 and more), same seed -> same cloud, and the flood complex
    of a device-drawn cloud against the kd-tree oracle."""
    import flooder_amd as fa
    from oracle import flood_oracle as fo

    dev = torch.device("cuda:0")
    n = 200_000
    # torus
    a = sg.generate_noisy_torus_points_3d(n, seed=5).numpy()
    b = sg.generate_noisy_torus_points_3d(n, seed=5, device=dev)
    assert torch.equal(b, sg.generate_noisy_torus_points_3d(n, seed=5, device=dev))          # deterministic per seed
    assert not torch.equal(b, sg.generate_noisy_torus_points_3d(n, seed=6, device=dev))
    b = b.cpu().numpy()
    tube = lambda t: np.sqrt((np.sqrt(t[:, 0] ** 2 + t[:, 1] ** 2) - 3.0) ** 2 + t[:, 2] ** 2)
    for k in range(3):
        assert _ks(a[:, k], b[:, k]) < 0.006, k
    assert _ks(tube(a), tube(b)) < 0.006
    assert _ks(np.arctan2(a[:, 1], a[:, 0]), np.arctan2(b[:, 1], b[:, 0])) < 0.006
    # annulus (area law: radius - width + width * sqrt(u))
    a = sg.generate_annulus_points_2d(n, radius=2.0, width=0.5, seed=7).numpy()
    b = sg.generate_annulus_points_2d(n, radius=2.0, width=0.5, seed=7, device=dev).cpu().numpy()
    assert _ks(np.linalg.norm(a, axis=1), np.linalg.norm(b, axis=1)) < 0.006
    assert _ks(np.arctan2(a[:, 1], a[:, 0]), np.arctan2(b[:, 1], b[:, 0])) < 0.006
    # swiss cheese: voids by the same rule (inside the box by a full radius, radii in range), points uniform outside them
    pa, ca, ra = sg.generate_swiss_cheese_points(n, k=6, seed=9)
    pb, cb, rb = sg.generate_swiss_cheese_points(n, k=6, seed=9, device=dev)
    assert pb.device.type == "cuda" and cb.shape == (6, 3) and rb.shape == (6,)
    assert bool(((rb >= 0.1) & (rb <= 0.2)).all()) and bool((cb >= 0.2 - 1e-6).all()) and bool((cb <= 0.8 + 1e-6).all())
    assert bool((torch.cdist(pb, cb) >= rb[None, :] - 1e-6).all())
    # the same voids given, the point laws must agree: redraw the CPU points around the DEVICE voids by rejection
    g = torch.Generator().manual_seed(1)
    cand = torch.rand(3 * n, 3, generator=g)
    keep = (torch.cdist(cand, cb.cpu()) >= rb.cpu()[None, :]).all(dim=1)
    ref = cand[keep][:n].numpy()
    got = pb.cpu().numpy()
    for k in range(3):
        assert _ks(ref[:, k], got[:, k]) < 0.006, k
    dmin = lambda t: (np.linalg.norm(t[:, None, :] - cb.cpu().numpy()[None], axis=2) - rb.cpu().numpy()[None]).min(axis=1)
    assert _ks(dmin(ref[:50_000]), dmin(got[:50_000])) < 0.012
    # ... and the path takes a device-drawn cloud as it is: flood complex against the oracle
    pts = sg.generate_swiss_cheese_points(60_000, k=4, seed=3, device=dev)[0]
    lms = fa.generate_landmarks(pts, 60, start_idx=0)
    fc = fa.flood_complex(pts, lms, points_per_edge=8)
    ref = fo.flood_complex_oracle(pts.cpu().numpy(), lms.cpu().numpy(), points_per_edge=8)
    assert set(fc) == set(ref)
    err = max(abs(fc[k] - ref[k]) for k in ref)
    assert err < 1e-5 * max(ref.values()) + 2.5e-7


@pytest.mark.gpu
def test_cfg5_cloud_drawn_on_the_device_at_full_size():
    """16 M swiss-cheese points drawn in HBM (no host round trip): shape, box, no point inside a void - checked on the
    device in blocks - and the generator's rounds terminate (rejection sampling in rounds of 4 x the missing count)."""
    dev = torch.device("cuda:0")
    p, c, r = sg.generate_swiss_cheese_points(16_000_000, k=6, seed=42, device=dev)
    assert p.shape == (16_000_000, 3) and p.dtype == torch.float32 and p.device.type == "cuda"
    assert float(p.min()) >= 0.0 and float(p.max()) <= 1.0
    for i in range(0, p.shape[0], 2_000_000):
        blk = p[i:i + 2_000_000]
        assert bool((torch.cdist(blk, c) >= r[None, :] - 1e-6).all())
    occ = torch.histc(p[:, 0], bins=16, min=0.0, max=1.0)
    assert float(occ.min()) > 0.5 * float(occ.max())   # (uniform up to the voids' share of a slab)
