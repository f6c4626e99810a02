"""Diagnostic (library built with -DFLOODER_QUERY_DIAG): rows of cells a queried sample of the cell sweep visits
(0 - 9 in 3D) and how many samples are dropped against the simplex's running maximum.  usage: query_rows.py [cfg5|cfg2|cfg3] [chunk_major]"""
import sys, torch, numpy as np
sys.path.insert(0, '.')
import flooder_amd as fa
from flooder_amd import _native, core
lib = _native.load()
which = sys.argv[1] if len(sys.argv) > 1 else "cfg5"
if len(sys.argv) > 2:
    _native.check(lib.flooder_set_option(b"cell_chunk_major", int(sys.argv[2])), "opt")
core.CELL_WITNESS = False
torch.manual_seed(42)
dev = torch.device('cuda:0')
n_l = 1000
if which == "cfg3":
    pts = fa.generate_noisy_torus_points_3d(1_000_000, seed=42).to(dev)
elif which == "cfg5":
    pts = fa.generate_swiss_cheese_points(16_000_000, k=6, seed=42)[0].to(dev); n_l = 4000
else:
    pts = torch.randn(1_000_000, 3).to(dev)
lms = fa.generate_landmarks(pts, n_l, start_idx=0)
stree, simplices = core._build_complex(lms, 3)
verts = lms[torch.as_tensor(simplices[3], device=dev)]
weights, vi, fi = core.generate_grid(30, 3, dev, torch.float32)
faces = core._FaceTable(fi, weights.shape[0], dev)
index = core.PointIndex(pts)
stats = torch.zeros(128, dtype=torch.int64, device=dev)
core._sweep_dimension_cell(index, verts, weights, faces, None, stats=stats)
torch.cuda.synchronize()
h = stats[100:111].cpu().numpy().astype(float)
tot = h[:10].sum()
print(which, "queried samples", int(tot), "of", verts.shape[0] * weights.shape[0], "| rows visited 0..9 (%):", np.round(h[:10] / tot * 100, 1).tolist(),
      "| mean rows", round(float((h[:10] * np.arange(10)).sum() / tot), 2), "| dropped %", round(h[10] / tot * 100, 1))
