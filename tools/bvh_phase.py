"""Diagnostic: cycle shares inside the exact finish (tree sweep over the flagged tiles); timer build.
usage: python tools/bvh_phase.py [cfg3|cfg2]"""
import sys, torch, numpy as np
sys.path.insert(0, '.')
import flooder_amd as fa
from flooder_amd import _native, core
core.CELL_SUPER = False  # (the per-chunk records below assume one work item per chunk)
which = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
torch.manual_seed(42)
dev = torch.device('cuda:0')
if which == "cfg3":
    pts = fa.generate_noisy_torus_points_3d(1_000_000, seed=42).to(dev)
else:
    pts = torch.randn(1_000_000, 3).to(dev)
lms = fa.generate_landmarks(pts, 1000, start_idx=0)
stree, simplices = core._build_complex(lms, 3)
verts = lms[torch.as_tensor(simplices[3], device=dev)]
weights, vi, fi = core.generate_grid(30, 3, dev, torch.float32)
faces = core._FaceTable(fi, weights.shape[0], dev)
index = core.PointIndex(pts)
stats = torch.zeros(64 + 16 * verts.shape[0] * ((weights.shape[0] + 255) // 256), dtype=torch.int64, device=dev)
for _ in range(2):
    stats.zero_()
    core._sweep_dimension_cell(index, verts, weights, faces, None, stats=stats)
torch.cuda.synchronize()
t = stats[49:55].cpu().numpy().astype(float)
ev, te, nd = stats[9:12].cpu().tolist()
rounds = int(stats[15])
print(which, "rest pass of the finish: leaf evals", ev, "leaf tests", te, "node expansions", nd, "focus rounds", rounds, "(counts: all passes)")
for n, v, cnt in zip(["item / round setup", "node expansion (+refine)", "leaf select + test", "leaf evaluation", "bounds / live upkeep + delivery", "queue pop"],
                     t, [rounds, nd, te, ev, ev, 1]):
    print(f"{n:34s} {v / t.sum() * 100:6.2f} %   {v / max(cnt, 1):9.0f} cycles per event")
