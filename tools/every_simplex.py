"""EVERY top simplex of a full-size workload against scipy's kd-tree over all points (the reference's CPU computation,
core.py:197-199, on all host cores) - the full-size tests compare samples of cfg 4 / cfg 5 only.
python tools/every_simplex.py cfg5|cfg4|cfg3|cfg2   (CPU time: the kd-tree queries; minutes on the GPU box's cores)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import flooder_amd as fa
from helpers import kdtree_face_values, tolerances
from scipy.spatial import cKDTree

which = sys.argv[1] if len(sys.argv) > 1 else "cfg5"
dev = torch.device("cuda:0")
torch.manual_seed(42)
if which == "cfg5":
    pts = fa.generate_swiss_cheese_points(16_000_000, k=6, seed=42)[0].float(); n_l, kw, top, ppe = 4000, {}, 3, 30
elif which == "cfg4":
    pts = torch.randn(2_000_000, 6); n_l, kw, top, ppe = 2000, dict(max_dimension=2, points_per_edge=8), 2, 8
elif which == "cfg3":
    pts = fa.generate_noisy_torus_points_3d(1_000_000, seed=42).float(); n_l, kw, top, ppe = 1000, {}, 3, 30
else:
    pts = torch.randn(1_000_000, 3); n_l, kw, top, ppe = 1000, {}, 3, 30
tp = pts.to(dev)
lms = fa.generate_landmarks(tp, n_l, start_idx=0)
t0 = time.time()
st = fa.flood_complex(tp, lms, return_simplex_tree=True, **kw)
torch.cuda.synchronize()
print(f"{which}: flood_complex {time.time() - t0:.2f} s (first call, incl. host triangulation)", flush=True)
P, L = pts.numpy(), lms.cpu().numpy()
t0 = time.time()
tree = cKDTree(P, balanced_tree=False, compact_nodes=False)
print(f"kd-tree over {len(P)} points: {time.time() - t0:.1f} s", flush=True)
rtol, atol = tolerances(P)
worst_all = 0.0
for d in range(top, 0, -1):
    rows, vals = st.simplices_of_dimension(d), np.asarray(st.filtrations_of_dimension(d), dtype=np.float64)
    t0 = time.time()
    ref = kdtree_face_values(tree, L, rows, ppe, d)
    err = np.abs(vals - ref)
    bad = ~(err <= atol + rtol * np.abs(ref))
    rel = float((err / np.maximum(np.abs(ref), 1e-30)).max())
    worst_all = max(worst_all, rel)
    print(f"dimension {d}: {len(rows)} simplices, all lattice samples of each, kd-tree {time.time() - t0:.1f} s: {int(bad.sum())} off (rtol {rtol:g}, atol {atol:.2e}), "
          f"worst |err| {err.max():.3e}, worst relative {rel:.3e}", flush=True)
    assert not bad.any()
print(f"{which}: every simplex of every dimension matches the kd-tree over all points; worst relative error {worst_all:.3e}")
