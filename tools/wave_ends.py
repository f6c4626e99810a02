"""Diagnostic: start and end time of every wave of the cell sweep (library built with -DFLOODER_WAVE_END): how long
is the tail of the persistent kernel really?   usage: python tools/wave_ends.py [cfg2|cfg3] [W]"""
import sys, torch, numpy as np
sys.path.insert(0, '.')
import flooder_amd as fa
from flooder_amd import _native, core
core.CELL_SUPER = False  # (the per-chunk records below assume one work item per chunk)
which = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
W = int(sys.argv[2]) if len(sys.argv) > 2 else 1
torch.manual_seed(42)
dev = torch.device('cuda:0')
pts = (fa.generate_noisy_torus_points_3d(1_000_000, seed=42) if which == "cfg3" else torch.randn(1_000_000, 3)).to(dev)
lms = fa.generate_landmarks(pts, 1000, start_idx=0)
stree, simplices = core._build_complex(lms, 3)
verts = lms[torch.as_tensor(simplices[3], device=dev)]
verts = verts[torch.argsort(verts.mean(1)[:, 0])][0::W].contiguous()
weights, vi, fi = core.generate_grid(30, 3, dev, torch.float32)
faces = core._FaceTable(fi, weights.shape[0], dev)
index = core.PointIndex(pts)
stats = torch.zeros(64 + 2 * 8192, dtype=torch.int64, device=dev)
for _ in range(3):
    stats.zero_()
    core._sweep_dimension_cell(index, verts, weights, faces, None, stats=stats)
torch.cuda.synchronize()
t = stats[64:].cpu().numpy().reshape(-1, 2)
t = t[t[:, 1] > 0]
t0, t1 = t[:, 0].min(), t[:, 1].max()
end = np.sort(t[:, 1] - t0)
print(f"{which} W={W}: {len(t)} waves, span {t1 - t0} ticks (10 ns); wave end percentiles: " +
      " ".join(f"p{p}={np.percentile(end, p):.0f}" for p in (1, 10, 25, 50, 75, 90, 99, 100)))
print(f"mean end {end.mean():.0f} = {end.mean() / (t1 - t0) * 100:.1f}% of the span: a perfectly balanced queue would take ~{end.mean():.0f} ticks "
      f"({(1 - end.mean() / (t1 - t0)) * 100:.1f}% of the kernel is tail)")
