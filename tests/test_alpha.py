"""The reference's homotopy-equivalence test (tests/test_flooder.py:24-75, ``test_vs_alpha``) restated: Flood
complex with L = X on the 1000-point figure eight against the alpha complex, bottleneck distance < 5e-4 in
dimensions 0 and 1 - the reference's only independent pin of filtration values AND persistence together.
gudhi's AlphaComplex and bottleneck_distance are restated in ``oracle/alpha.py`` (test infrastructure) and
checked here on cases with known answers first."""
import itertools

import numpy as np
import pytest
import torch

import flooder_amd as fa
from flooder_amd.simplex_tree import SimplexTree
from oracle import alpha as al


def test_alpha_filtration_hand_checked():
    # obtuse triangle: the long edge is not Gabriel and takes the circumradius; the short edges keep half lengths
    P = np.array([[0.0, 0.0], [4.0, 0.0], [2.0, 0.5]])
    (v, e, t), (fv, fe, ft) = al.alpha_filtration_2d(P)
    R = ft[0]
    assert t.tolist() == [[0, 1, 2]] and R == pytest.approx(4.25, rel=1e-12)
    val = {tuple(r): x for r, x in zip(e.tolist(), fe)}
    assert val[(0, 1)] == pytest.approx(R)                      # not Gabriel: vertex 2 inside the diametral disk
    assert val[(0, 2)] == pytest.approx(0.5 * np.hypot(2, 0.5))
    assert val[(1, 2)] == pytest.approx(0.5 * np.hypot(2, 0.5))
    assert (fv == 0).all()
    # acute (equilateral) triangle: all edges Gabriel
    P = np.array([[0.0, 0.0], [1.0, 0.0], [0.5, np.sqrt(3) / 2]])
    _, (fv, fe, ft) = al.alpha_filtration_2d(P)
    assert np.allclose(fe, 0.5) and ft[0] == pytest.approx(1 / np.sqrt(3))
    # two triangles sharing a non-Gabriel edge: it takes the smaller circumradius
    P = np.array([[0.0, 0.0], [4.0, 0.0], [2.0, 0.5], [2.0, -3.0]])
    (v, e, t), (fv, fe, ft) = al.alpha_filtration_2d(P)
    val = {tuple(r): x for r, x in zip(e.tolist(), fe)}
    if (0, 1) in val:
        assert val[(0, 1)] == pytest.approx(ft.min())
    # filtration is monotone
    for (a, b, c), x in zip(t.tolist(), ft):
        assert max(val[(a, b)], val[(a, c)], val[(b, c)]) <= x + 1e-12


def _brute_bottleneck(A, B):
    """Permutation search over the augmented diagrams (tiny inputs)."""
    n, m = len(A), len(B)
    best = np.inf
    X = [("p", a) for a in A] + [("d", None)] * m
    Y = [("p", b) for b in B] + [("d", None)] * n
    for perm in itertools.permutations(range(n + m)):
        cost = 0.0
        for i, j in enumerate(perm):
            (kx, x), (ky, y) = X[i], Y[j]
            if kx == "p" and ky == "p":
                c = max(abs(x[0] - y[0]), abs(x[1] - y[1]))
            elif kx == "p":
                c = (x[1] - x[0]) / 2
            elif ky == "p":
                c = (y[1] - y[0]) / 2
            else:
                c = 0.0
            cost = max(cost, c)
        best = min(best, cost)
    return best


def test_bottleneck_known_answers():
    A = np.array([[0.0, 1.0]])
    assert al.bottleneck_distance(A, np.zeros((0, 2))) == pytest.approx(0.5)
    assert al.bottleneck_distance(A, np.array([[0.0, 1.2]])) == pytest.approx(0.2)
    assert al.bottleneck_distance(A, np.array([[0.0, 5.0]])) == pytest.approx(2.5)   # both to the diagonal
    assert al.bottleneck_distance(np.array([[0.0, np.inf]]), np.array([[0.3, np.inf]])) == pytest.approx(0.3)
    assert al.bottleneck_distance(np.array([[0.0, np.inf]]), np.zeros((0, 2))) == float("inf")
    rng = np.random.default_rng(0)
    for _ in range(25):
        n, m = rng.integers(0, 4), rng.integers(0, 4)
        A = np.sort(rng.random((n, 2)), axis=1)
        B = np.sort(rng.random((m, 2)), axis=1)
        assert al.bottleneck_distance(A, B) == pytest.approx(_brute_bottleneck(A.tolist(), B.tolist()), abs=1e-12)


def _diagrams(stree):
    stree.compute_persistence()
    return [stree.persistence_intervals_in_dimension(i) for i in range(2)]


def _alpha_diagrams(X):
    (v, e, t), (fv, fe, ft) = al.alpha_filtration_2d(X)
    st = SimplexTree.from_arrays([v, e, t])
    st.assign_filtration_bulk(v, fv)
    st.assign_filtration_bulk(e, fe)
    st.assign_filtration_bulk(t, ft)
    return _diagrams(st)


def _vs_alpha(device, use_rand, batch_size):
    torch.manual_seed(42)
    np.random.seed(42)
    X = fa.generate_figure_eight_points_2d(1000)
    L = X
    kwargs = ({"num_rand": 20_000, "points_per_edge": None} if use_rand
              else {"num_rand": None, "points_per_edge": 130})
    stree = fa.flood_complex(X.to(device), L.to(device), return_simplex_tree=True, batch_size=batch_size, **kwargs)
    flood = _diagrams(stree)
    alpha = _alpha_diagrams(X.numpy())
    for dim in range(2):
        dist = al.bottleneck_distance(flood[dim], alpha[dim])
        assert dist < 5e-4, f"bottleneck distance too high in dimension {dim} (use_rand={use_rand}): {dist}"
    # the figure eight: one component, two loops that outlive everything else
    h1 = flood[1]
    life = np.sort(h1[:, 1] - h1[:, 0])[::-1]
    assert np.isinf(flood[0][:, 1]).sum() == 1 and life[1] > 5 * life[2]


def test_vs_alpha_cpu_grid():
    """points_per_edge = 130 as the reference; CPU (kd-tree) path."""
    _vs_alpha(torch.device("cpu"), use_rand=False, batch_size=8)


@pytest.mark.gpu
@pytest.mark.parametrize("batch_size", [8, 23])
@pytest.mark.parametrize("use_rand", [True, False])
def test_vs_alpha_gpu(use_rand, batch_size):
    """The reference's parametrisation (use_triton has no counterpart: ROCm tensors always run the HIP kernels)."""
    assert torch.cuda.is_available()
    _vs_alpha(torch.device("cuda:0"), use_rand, batch_size)
