"""BASELINE.json configs[0] at its own size, on the CPU: 10 k-point noisy torus, 200 landmarks, points_per_edge 30,
persistent homology up to dimension 2 ("reference plumbing, no GPU").  The product's CPU branch (the reference's CPU
path, ``flooder/core.py:127-128, 197-199``) against the oracle on every simplex, then the hand-off the reference
makes to gudhi (``flooder/cli.py:473-476``): intervals in dimensions 0, 1, 2 with the torus' Betti numbers 1, 2, 1
as the long bars."""
import numpy as np
import torch

import flooder_amd as fa
from oracle import flood_oracle as fo
from helpers import assert_close_filtration, dict_values


def test_cfg1_torus_10k_200_cpu_equals_oracle_and_has_torus_homology():
    pts = fo.noisy_torus(10_000, seed=42)            # tests/test_flooder.py:32 seeds with 42
    tp = torch.as_tensor(pts)
    lms = fa.generate_landmarks(tp, 200, start_idx=0)
    L = lms.numpy()
    assert np.array_equal(L, pts[fo.exact_fps(pts, 200, 0)])
    st = fa.flood_complex(tp, lms, points_per_edge=30, return_simplex_tree=True)
    ref = fo.flood_complex_oracle(pts, L, points_per_edge=30)
    got = st.to_dict()
    assert set(got) == set(ref) and len(ref) > 4000
    keys = sorted(ref)
    assert_close_filtration(dict_values(got, keys), dict_values(ref, keys), pts, "cfg1")
    assert all(got[(i,)] == 0.0 for i in range(200))
    # persistent homology (cli.py:473-476: compute_persistence, then the intervals of every dimension below max_dim)
    st.compute_persistence()
    h0, h1, h2 = (st.persistence_intervals_in_dimension(d) for d in (0, 1, 2))
    assert np.isinf(h0[:, 1]).sum() == 1             # one component
    life1 = np.sort(h1[:, 1] - h1[:, 0])[::-1]
    life2 = np.sort(h2[:, 1] - h2[:, 0])[::-1]
    # torus of radii 3 and 1: the two 1-cycles die at about the tube radius (1) and the hole radius (2), the void at
    # about the tube radius; everything else is sampling noise an order of magnitude shorter
    assert life1[1] > 0.5 and life1[2] < 0.5 * life1[1], life1[:4]
    assert life2[0] > 0.5 and (len(life2) == 1 or life2[1] < 0.5 * life2[0]), life2[:3]
