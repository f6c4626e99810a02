"""Randomised check of flooder_delaunay_nd against Qhull outside the test suite: clouds of 2 - 8 dimensions, Gaussian /
uniform / anisotropic / clustered / far from the origin / snapped to a grid, float32 and float64 values.  On every cloud
the routine either returns Qhull's simplices (general position: the triangulation is unique) or declines (exact tie).
Where Qhull and the native table differ, the native simplices are checked for the empty-circumsphere property in exact
rational arithmetic on a sample (Qhull works in floating point).  usage: stress_delaunay_nd.py [n_clouds] [seed]"""
import sys, time
from fractions import Fraction
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from scipy.spatial import Delaunay
from flooder_amd import simplex_tree as stm
from test_delaunay_nd import nd, qhull, _insphere_sign



def _bareiss(m):
    """Exact determinant of a square matrix of Python ints (fraction-free elimination)."""
    m = [r[:] for r in m]
    n, sign, prev = len(m), 1, 1
    for k in range(n - 1):
        if m[k][k] == 0:
            sw = next((i for i in range(k + 1, n) if m[i][k] != 0), None)
            if sw is None:
                return 0
            m[k], m[sw] = m[sw], m[k]
            sign = -sign
        for i in range(k + 1, n):
            for j in range(k + 1, n):
                m[i][j] = (m[i][j] * m[k][k] - m[i][k] * m[k][j]) // prev
        prev = m[k][k]
    return sign * m[-1][-1]


def exact_violations(P, simplices):
    """For each simplex: is any point strictly inside its circumsphere?  Integer arithmetic on the dyadic grid."""
    Pf = [[Fraction(float(x)) for x in row] for row in P]
    den = 1
    for row in Pf:
        for x in row:
            den = max(den, x.denominator)
    Q = [[int(x * den) for x in row] for row in Pf]
    d = len(Q[0])
    bad = 0
    for s in simplices:
        b = Q[s[0]]
        rows = [[a - c for a, c in zip(Q[i], b)] for i in s[1:]]
        orient = _bareiss(rows)
        assert orient != 0
        lifted = [r + [sum(x * x for x in r)] for r in rows]
        for e in range(len(Q)):
            if e in s:
                continue
            w = [a - c for a, c in zip(Q[e], b)]
            h = _bareiss(lifted + [w + [sum(x * x for x in w)]])
            if h * orient < 0:        # power = H / orient < 0: strictly inside
                bad += 1
                break
    return bad


n_clouds = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
kinds = ["gauss", "uniform", "aniso", "clusters", "offset", "grid", "f64"]
same = declined = differ = wrong_native = 0
t_nat = t_qh = 0.0
for case in range(n_clouds):
    dim = int(rng.integers(2, 9))
    n = int(rng.integers(dim + 2, {2: 3000, 3: 2000, 4: 1200, 5: 500, 6: 250, 7: 120, 8: 70}[dim]))
    kind = kinds[case % len(kinds)]
    P = rng.normal(size=(n, dim))
    if kind == "uniform": P = rng.random(size=(n, dim))
    if kind == "aniso": P *= 10.0 ** rng.uniform(-3, 0, size=dim)
    if kind == "clusters": P = P * 0.05 + rng.normal(size=(8, dim))[rng.integers(0, 8, size=n)]
    if kind == "offset": P = P * 0.01 + rng.uniform(-300, 300, size=dim)
    if kind == "grid": P = np.round(P * 256) / 256
    P = P.astype(np.float64 if kind == "f64" else np.float32)
    P = np.unique(P, axis=0)
    if len(P) < dim + 2: continue
    t0 = time.perf_counter(); got = nd(P); t_nat += time.perf_counter() - t0
    if not isinstance(got, np.ndarray):
        declined += 1
        print(f"case {case}: {kind} dim {dim} n {len(P)}: declined ({got + (1 << 40)})", flush=True)
        continue
    t0 = time.perf_counter(); ref = qhull(P); t_qh += time.perf_counter() - t0
    if np.array_equal(got, ref):
        same += 1
        continue
    differ += 1
    sa, sb = set(map(tuple, got.tolist())), set(map(tuple, ref.tolist()))
    bad_nat = exact_violations(P, sorted(sa - sb)[:12])
    bad_qh = exact_violations(P, sorted(sb - sa)[:12])
    wrong_native += bad_nat
    print(f"case {case}: {kind} dim {dim} n {len(P)}: differs from Qhull ({len(sa - sb)} / {len(sb - sa)} simplices): of the first 12 "
          f"native-only simplices {bad_nat} hold a point strictly inside their circumsphere (exact integers), of Qhull's {bad_qh}", flush=True)
print(f"{n_clouds} clouds: {same} equal to Qhull, {declined} declined, {differ} differ (native simplices with a point inside their "
      f"circumsphere: {wrong_native}); native {t_nat:.1f} s, Qhull {t_qh:.1f} s")
