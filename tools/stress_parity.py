"""Randomised parity stress on the GPU box (not part of the test suite: ~3 minutes): clouds of different shape and
density contrast, the cell sweep against the tree sweep (both exact: bit for bit) and against a kd-tree on the host
(the oracle's computation); exact FPS batched against brute force.   usage: python tools/stress_parity.py [n_cases]"""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
import flooder_amd as fa
from flooder_amd import core
from helpers import assert_tree_matches_kdtree

dev = torch.device("cuda:0")
core.WIT_MIN_SIMPLICES = 0   # (the witness sweep on every queue, however short: this is a parity stress)
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(2024)
bad = 0
t0 = time.time()
for case in range(n_cases):
    kind = ("gauss", "torus", "cheese", "clusters", "plane2d", "shell")[case % 6]
    n = int(rng.integers(30_000, 500_000))
    if kind == "gauss":
        P = rng.normal(size=(n, 3)) * rng.uniform(0.2, 3.0, size=3)
    elif kind == "torus":
        P = fa.generate_noisy_torus_points_3d(n, seed=int(rng.integers(1 << 30))).numpy()
    elif kind == "cheese":
        P = fa.generate_swiss_cheese_points(n, k=int(rng.integers(2, 9)), seed=int(rng.integers(1 << 30)))[0].numpy()
    elif kind == "clusters":   # density contrast of three orders of magnitude
        k = int(rng.integers(3, 9))
        c = rng.uniform(-1, 1, size=(k, 3))
        sc = 10.0 ** rng.uniform(-3, -0.5, size=k)
        lab = rng.integers(0, k, size=n)
        P = c[lab] + rng.normal(size=(n, 3)) * sc[lab, None]
    elif kind == "plane2d":
        P = rng.normal(size=(n, 2)) * rng.uniform(0.3, 2.0, size=2)
    else:                       # a thin spherical shell: every tetrahedron inside is empty
        v = rng.normal(size=(n, 3))
        P = v / np.linalg.norm(v, axis=1, keepdims=True) * (1.0 + 0.01 * rng.normal(size=(n, 1)))
    P = np.ascontiguousarray(P, dtype=np.float32)
    n_l = int(rng.integers(60, 700))
    ppe = int(rng.choice([8, 12, 20, 30]))
    pts = torch.as_tensor(P, device=dev)
    start = int(rng.integers(0, n))
    # landmarks: batched bucketed FPS against brute force
    a = core.fps_indices(pts, n_l, start, method="brute").cpu().numpy()
    b = core.fps_indices(pts, n_l, start, method="bucket").cpu().numpy()
    ok_fps = np.array_equal(a, b)
    lms = pts[torch.as_tensor(a, device=dev)]
    d = P.shape[1]
    st_cell = fa.flood_complex(pts, lms, points_per_edge=ppe, method="cell", return_simplex_tree=True)
    st_bvh = fa.flood_complex(pts, lms, points_per_edge=ppe, method="bvh", return_simplex_tree=True)
    same = all(np.array_equal(st_cell.filtrations_of_dimension(k), st_bvh.filtrations_of_dimension(k)) for k in range(d + 1))
    S = len(st_cell.simplices_of_dimension(d))
    pick = np.sort(rng.choice(S, size=min(S, 400), replace=False))
    try:
        assert_tree_matches_kdtree(st_cell, P, lms.cpu().numpy(), ppe, d, f"case {case}", pick_top=pick, lower=False)
        ok_kd = True
    except AssertionError as e:
        ok_kd = False
        print("  kd-tree mismatch:", str(e)[:300])
    flag = "ok" if (ok_fps and same and ok_kd) else "FAILED"
    bad += flag != "ok"
    print(f"case {case:2d} {kind:9s} n={n:7d} lms={n_l:4d} ppe={ppe:2d} simplices={S:6d}  fps={ok_fps} cell==bvh={same} kd={ok_kd}  {flag}", flush=True)
print(f"{n_cases} cases, {bad} failed, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
