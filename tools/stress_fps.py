"""Randomised check of the batched landmark selection (flooder_fps_batched_f32) against the brute-force selection on one
MI355X: MANY landmarks per cloud (a fiftieth to a quarter of the points), where accepted landmarks lower other
candidates all the time - lattices with exact ties, clusters, duplicated points, 2 to 6 dimensions.
usage: python tools/stress_fps.py [cases]"""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from flooder_amd import core, _native
lib = _native.load()

dev = torch.device('cuda:0')
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(20260)
bad = 0
t0 = time.time()
for case in range(n_cases):
    dim = int(rng.integers(2, 7))
    n = int(rng.integers(2_000, 120_000))
    kind = ["gauss", "lattice", "clusters", "dups", "shell"][case % 5]
    if kind == "gauss":
        P = rng.normal(size=(n, dim))
    elif kind == "lattice":
        m = max(2, int(round(n ** (1.0 / dim))))
        g = np.stack(np.meshgrid(*[np.arange(m)] * dim, indexing="ij"), axis=-1).reshape(-1, dim).astype(np.float64)
        P = g[rng.permutation(len(g))]
    elif kind == "clusters":
        c = rng.normal(size=(8, dim)) * 5
        P = c[rng.integers(0, 8, size=n)] + rng.normal(size=(n, dim)) * rng.choice([1e-3, 0.1, 1.0], size=(n, 1))
    elif kind == "dups":
        base = rng.normal(size=(n // 3 + 1, dim))
        P = base[rng.integers(0, len(base), size=n)]
    else:
        v = rng.normal(size=(n, dim))
        P = v / np.linalg.norm(v, axis=1, keepdims=True) * (1 + 0.01 * rng.normal(size=(n, 1)))
    pts = torch.as_tensor(P.astype(np.float32), device=dev).contiguous()
    n = pts.shape[0]
    k = int(rng.integers(max(2, n // 50), max(3, n // 4)))
    start = int(rng.integers(0, n))
    index = core.PointIndex(pts)
    core.FPS_BATCHED = True
    # every other case through the form for large clouds (buckets of 256 rows, cap of 32 landmarks per launch, loops over
    # the lanes that hold the accepted candidates), which clouds of this size would not take by themselves
    rpl = 4 if case % 2 else 0
    assert lib.flooder_set_option(b"fps_rpl", rpl) == 0
    try:
        got = core.fps_indices(pts, k, start, method="bucket", index=index).cpu()
    finally:
        lib.flooder_set_option(b"fps_rpl", 0)
    ref = core.fps_indices(pts, k, start, method="brute").cpu()
    same = bool(torch.equal(got, ref))
    bad += 0 if same else 1
    first = int((got != ref).nonzero()[0]) if not same else -1
    print(f"case {case:3d} {kind:9s} dim={dim} n={n:7d} landmarks={k:6d} start={start:7d} rpl={rpl or 1} batched==brute={same}" + ("" if same else f"  FIRST DIFFERENCE at {first}"), flush=True)
print(f"{n_cases} cases, {bad} failed, {time.time() - t0:.0f} s")
