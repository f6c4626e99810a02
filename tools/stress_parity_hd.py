"""Randomised parity stress above 3D on the GPU box (not part of the test suite): clouds of different shape, anisotropy
and density contrast in 4 - 7 dimensions; the sorted-sample sweep over the k-d ordered index (the default there) against
the per-simplex tree sweep over the curve-ordered index (both exact: bit for bit), against a kd-tree on the host for
EVERY simplex, and sharded into tiles (3 ranks one after the other, combined as the all-reduce would).
usage: python tools/stress_parity_hd.py [n_cases]"""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
import flooder_amd as fa
from flooder_amd import core
from helpers import assert_tree_matches_kdtree
from oracle import flood_oracle as fo

dev = torch.device("cuda:0")
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(777)
core.BVH_SORTED_MIN_SAMPLES = 0
bad = 0
t0 = time.time()
for case in range(n_cases):
    dim = int(rng.integers(4, 8))
    kind = ("gauss", "aniso", "clusters", "shell", "subspace")[case % 5]
    n = int(rng.integers(20_000, 250_000))
    if kind == "gauss":
        P = rng.normal(size=(n, dim))
    elif kind == "aniso":
        P = rng.normal(size=(n, dim)) * 10.0 ** rng.uniform(-2, 1, size=dim)
    elif kind == "clusters":
        k = int(rng.integers(3, 9))
        c = rng.uniform(-1, 1, size=(k, dim)); sc = 10.0 ** rng.uniform(-3, -0.5, size=k); lab = rng.integers(0, k, size=n)
        P = c[lab] + rng.normal(size=(n, dim)) * sc[lab, None]
    elif kind == "shell":
        v = rng.normal(size=(n, dim)); P = v / np.linalg.norm(v, axis=1, keepdims=True) * (1.0 + 0.01 * rng.normal(size=(n, 1)))
    else:   # a 3-dimensional sheet in the ambient space, plus a little noise
        A = rng.normal(size=(3, dim)); P = rng.normal(size=(n, 3)) @ A + 1e-3 * rng.normal(size=(n, dim))
    P = np.ascontiguousarray(P, dtype=np.float32)
    n_l = int(rng.integers(dim + 6, 46 if dim <= 5 else 30))
    kw = dict(max_dimension=2, points_per_edge=int(rng.integers(4, 9))) if case % 3 else dict(max_dimension=2, num_rand=int(rng.integers(20, 60)))
    tp = torch.as_tensor(P, device=dev)
    lms = tp[torch.as_tensor(fo.exact_fps(P, n_l, 0), device=dev)]
    try:
        core.KD_ORDER_ABOVE_DIM, core.BVH_SORTED_SAMPLES = 3, True
        torch.manual_seed(case); a = fa.flood_complex(tp, lms, method="bvh", return_simplex_tree=True, **kw)
        core.KD_ORDER_ABOVE_DIM, core.BVH_SORTED_SAMPLES = 8, False
        torch.manual_seed(case); b = fa.flood_complex(tp, lms, method="bvh", **kw)
        core.KD_ORDER_ABOVE_DIM, core.BVH_SORTED_SAMPLES = 3, True
        da = a.to_dict()
        same = da == b
        parts = {}
        def collect():
            c = [0]
            def hook(full):
                parts.setdefault(c[0], []).append(full.clone()); c[0] += 1
            return hook
        def reduce():
            c = [0]
            def hook(full):
                full.copy_(torch.stack(parts.get(c[0], []) + [full]).amin(dim=0)); c[0] += 1
            return hook
        for r in (1, 2):
            torch.manual_seed(case); fa.flood_complex(tp, lms, method="bvh", simplex_shard=(r, 3), face_reduce_hook=collect(), **kw)
        torch.manual_seed(case); sh = fa.flood_complex(tp, lms, method="bvh", simplex_shard=(0, 3), face_reduce_hook=reduce(), **kw)
        shard_ok = sh == b
        ok_kd = True
        if "points_per_edge" in kw:
            try:
                assert_tree_matches_kdtree(a, P, lms.cpu().numpy(), kw["points_per_edge"], 2, f"case {case}", lower=True)
            except AssertionError as e:
                ok_kd = False; print("  kd-tree mismatch:", str(e)[:300])
        S = len(a.simplices_of_dimension(2))
        flag = "ok" if (same and shard_ok and ok_kd) else "FAILED"
    except Exception as e:   # (e.g. Qhull refusing a degenerate landmark set)
        S, same, shard_ok, ok_kd, flag = 0, None, None, None, f"skipped ({type(e).__name__})"
    bad += flag == "FAILED"
    print(f"case {case:2d} {kind:9s} dim={dim} n={n:7d} lms={n_l:3d} {str(kw):48s} triangles={S:6d}  sorted+kd==tree+curve={same} 3 tile shards=={shard_ok} kd-tree={ok_kd}  {flag}", flush=True)
print(f"{n_cases} cases, {bad} failed, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
