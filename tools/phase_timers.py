"""Diagnostic: per-phase cycle shares of the cell sweep (library built with -DFLOODER_PHASE_TIMERS).
usage: python tools/phase_timers.py [cfg2|cfg3]"""
import sys, torch, numpy as np
sys.path.insert(0, '.')
import flooder_amd as fa
from flooder_amd import _native, core
lib = _native.load()
core.CELL_WITNESS = False   # (its counters share stats[16:40] with the phase sums read below)
core.CELL_SUPER = len(sys.argv) > 2 and sys.argv[2] == 'super'  # (per-chunk records assume one work item per chunk; the phase sums do not)
torch.manual_seed(42)
dev = torch.device('cuda:0')
which = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
n_l = 1000
if which == "cfg3":
    pts = fa.generate_noisy_torus_points_3d(1_000_000, seed=42).to(dev)
elif which == "cfg5":
    pts = fa.generate_swiss_cheese_points(16_000_000, k=6, seed=42)[0].to(dev)
    n_l = 4000
else:
    pts = torch.randn(1_000_000, 3).to(dev)
lms = fa.generate_landmarks(pts, n_l, start_idx=0)
stree, simplices = core._build_complex(lms, 3)
simp = torch.as_tensor(simplices[3], device=dev)
verts = lms[simp]
weights, vi, fi = core.generate_grid(30, 3, dev, torch.float32)
faces = core._FaceTable(fi, weights.shape[0], dev)
index = core.PointIndex(pts)
stats = torch.zeros(64 + 16 * verts.shape[0] * ((weights.shape[0] + 255) // 256), dtype=torch.int64, device=dev)  # [64:] per-chunk cycles
for _ in range(2):
    stats.zero_()
    core._sweep_dimension_cell(index, verts, weights, faces, None, stats=stats)
torch.cuda.synchronize()
t = stats[16:28].cpu().numpy().astype(float)
n_chunks = verts.shape[0] * ((weights.shape[0] + 255) // 256)
names = ["pop", "samples+box+planes", "gather0", "density", "gather1", "count", "prefix", "scatter", "query", "brute", "output", "-"]
tot = t.sum()
for n, v in zip(names, t):
    print(f"{n:22s} {v/tot*100:6.2f} %   {v/n_chunks/2400:8.2f} us/chunk at 2.4 GHz  (raw cycles {v:.3e})")
print("stats", stats[:12].tolist())
