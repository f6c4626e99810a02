"""Z/2 persistence of the filtered complex (host C++ reduction) - the hand-off the reference gets from gudhi
(flooder/cli.py:473-476, tests/test_flooder.py:55-71)."""
import numpy as np
import pytest
import torch

import flooder_amd as fa
from flooder_amd import persistence as ph
from flooder_amd.simplex_tree import SimplexTree, delaunay_simplices
from oracle import flood_oracle as fo
from helpers import load_e2e


def test_circle_and_hollow_tetrahedron():
    st = SimplexTree()
    for i, e in enumerate([(0, 1), (1, 2), (2, 3), (0, 3)]):
        st.insert(e, 1.0 + i)
    for v in range(4):
        st.assign_filtration([v], 0.0)
    st.compute_persistence(persistence_dim_max=True)
    h0 = st.persistence_intervals_in_dimension(0)
    h1 = st.persistence_intervals_in_dimension(1)
    assert h0.shape == (4, 2) and np.isinf(h0[:, 1]).sum() == 1
    assert h1.tolist() == [[4.0, float("inf")]]
    # gudhi's default does not report homology in the top dimension of the complex
    st.compute_persistence()
    assert st.persistence_intervals_in_dimension(1).shape == (0, 2)

    full = SimplexTree()
    full.insert([0, 1, 2, 3], 0.0)
    hollow = SimplexTree.from_arrays([full._rows[d] for d in (0, 1, 2)])
    for d in (0, 1, 2):
        hollow._vals[d][:] = float(d)
    hollow.compute_persistence(persistence_dim_max=True)
    assert hollow.persistence_intervals_in_dimension(2).tolist() == [[2.0, float("inf")]]
    with pytest.raises(RuntimeError):
        SimplexTree().persistence_intervals_in_dimension(0)


@pytest.mark.parametrize("seed,dim", [(0, 2), (1, 3), (2, 3), (3, 4)])
def test_cpp_reduction_equals_python_reduction(seed, dim):
    rng = np.random.default_rng(seed)
    pts = rng.normal(size=(50, dim))
    st = SimplexTree.from_arrays(delaunay_simplices(pts))
    for d, rows in st._rows.items():
        st._vals[d] = rng.random(rows.shape[0])
    st.make_filtration_non_decreasing()
    dims, filt, bptr, bidx, _ = ph.filtration_order(st)
    assert (np.diff(filt) >= 0).all()
    a = ph.reduce_pairs(dims, bptr, bidx)
    b = ph.reduce_pairs_python(dims, bptr, bidx)
    assert np.array_equal(a, b)
    # a triangulated ball: one essential class (a component), every other simplex is paired
    assert int((a == -1).sum()) == 1
    # Euler characteristic from the pairing
    betti = {}
    for j in np.nonzero(a == -1)[0]:
        betti[int(dims[j])] = betti.get(int(dims[j]), 0) + 1
    assert betti == {0: 1}


def _long_bars(st, dim, tau):
    iv = st.persistence_intervals_in_dimension(dim)
    return iv[(iv[:, 1] - iv[:, 0]) > tau]


def test_flood_complex_of_a_torus_has_torus_homology():
    """Reference claim (README / docs): Flood PH recovers the topology of the sampled shape."""
    pts = fo.noisy_torus(20_000, seed=4)
    lms = pts[fo.exact_fps(pts, 300, 0)]
    st = fa.flood_complex(torch.as_tensor(pts), torch.as_tensor(lms), points_per_edge=8, return_simplex_tree=True)
    st.compute_persistence()
    h0 = st.persistence_intervals_in_dimension(0)
    assert np.isinf(h0[:, 1]).sum() == 1
    assert len(_long_bars(st, 1, 0.4)) == 2      # two independent loops
    assert len(_long_bars(st, 2, 0.4)) == 1      # one void


@pytest.mark.parametrize("name", ["torus3d_grid", "eight2d_grid", "cheese3d_grid"])
def test_intervals_from_reference_values_and_from_product_agree(name):
    """Same reduction on the reference's golden filtration values and on the product's (CPU branch)."""
    z, kw, keys = load_e2e(name)
    fc = fa.flood_complex(torch.as_tensor(z["points"]), torch.as_tensor(z["landmarks"]), return_simplex_tree=True, **kw)
    ref = SimplexTree()
    by_len = {}
    for k, v in zip(keys, z["filtration_f32"]):
        by_len.setdefault(len(k), []).append((k, v))
    ref = SimplexTree.from_arrays([np.array([k for k, _ in by_len[L]]) for L in sorted(by_len)])
    for L in sorted(by_len):
        ref.assign_filtration_bulk(np.array([k for k, _ in by_len[L]]), np.array([v for _, v in by_len[L]]))
    fc.compute_persistence()
    ref.compute_persistence()
    for d in range(fc.dimension()):
        a = fc.persistence_intervals_in_dimension(d)
        b = ref.persistence_intervals_in_dimension(d)
        assert a.shape == b.shape
        assert np.allclose(a, b, atol=1e-6, equal_nan=True)
