"""Randomised cross-check on the GPU: cell / bvh sweeps vs the exhaustive ball sweep and the kd-tree oracle."""
import sys, numpy as np, torch, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import flooder_amd as fa
from oracle import flood_oracle as fo
from helpers import assert_close_filtration, dict_values
dev = torch.device('cuda:0')
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 40
fails = 0
for case in range(n_cases):
    dim = int(rng.choice([2, 3, 3, 3, 4]))
    n = int(rng.choice([60, 500, 3000, 20000, 60000]))
    kind = rng.choice(["gauss", "uniform", "clusters", "plane", "dupes", "scaled"])
    pts = rng.normal(size=(n, dim))
    if kind == "uniform": pts = rng.random((n, dim))
    if kind == "clusters": pts = rng.normal(size=(n, dim)) * 0.05 + rng.normal(size=(8, dim))[rng.integers(0, 8, n)]
    if kind == "plane" and dim >= 3: pts[:, -1] = 1e-3 * rng.normal(size=n)
    if kind == "dupes": pts[n // 2:] = pts[: n - n // 2]
    if kind == "scaled": pts = pts * float(rng.choice([1e-3, 1e3])) + float(rng.choice([0, 50.0]))
    pts = pts.astype(np.float32)
    n_l = int(min(n, rng.choice([dim + 2, 20, 80, 250])))
    lms = pts[fo.exact_fps(pts, n_l, 0)]
    if len(np.unique(lms, axis=0)) < n_l:  # degenerate landmarks: Qhull would fail, skip
        continue
    mode_rand = rng.random() < 0.3
    kw = dict(points_per_edge=None, num_rand=int(rng.choice([16, 100, 700]))) if mode_rand else dict(points_per_edge=int(rng.choice([2, 3, 6, 11])))
    if dim == 4: kw["max_dimension"] = 2
    tp, tl = torch.as_tensor(pts, device=dev), torch.as_tensor(lms, device=dev)
    try:
        res = {}
        for m in (["cell", "bvh", "ball"] if dim in (2, 3) else ["bvh", "ball"]):
            torch.manual_seed(case); res[m] = fa.flood_complex(tp, tl, method=m, **kw)
        torch.manual_seed(case); ref = fo.flood_complex_oracle(pts, lms, **kw)
    except Exception as e:  # Qhull degeneracies etc.
        print("case", case, "skipped:", type(e).__name__, str(e)[:80]); continue
    keys = sorted(ref)
    ok = True
    for m, r in res.items():
        if r != res["ball"]:
            bad = [k for k in keys if r[k] != res["ball"][k]]
            print(f"case {case} [{kind} dim{dim} n{n} lm{n_l} {kw}] {m} != ball on {len(bad)} simplices e.g. {bad[:3]} {[ (r[k], res['ball'][k]) for k in bad[:3]]}"); ok = False
    try:
        assert_close_filtration(dict_values(res["ball"], keys), dict_values(ref, keys), pts, f"case {case}")
    except AssertionError as e:
        print(f"case {case} [{kind} dim{dim} n{n} lm{n_l} {kw}] ball vs oracle: {e}"); ok = False
    fails += (not ok)
print("cases", n_cases, "failures", fails)
