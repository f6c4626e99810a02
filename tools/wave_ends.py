"""Diagnostic: start and end time of every wave of the cell sweep (library built with -DFLOODER_WAVE_END): how long
is the tail of the persistent kernel really?   usage: python tools/wave_ends.py [cfg2|cfg3] [W]"""
import sys, torch, numpy as np
sys.path.insert(0, '.')
import flooder_amd as fa
from flooder_amd import _native, core
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
import os
if os.environ.get('NO_SUPER'):
    core.CELL_SUPER = False
for kv in filter(None, os.environ.get('OPTS', '').split(',')):
    _native.check(_native.load().flooder_set_option(kv.split('=')[0].encode(), int(kv.split('=')[1])), 'set_option')
stats = torch.zeros(40000 + 9 + 2 * 200000 + 64, dtype=torch.int64, device=dev)  # (also room for the per-item records)
for _ in range(3):
    stats.zero_()
    core._sweep_dimension_cell(index, verts, weights, faces, None, stats=stats)
torch.cuda.synchronize()
tall = stats[64:64 + 2 * 8192].cpu().numpy().reshape(-1, 2)
g0 = tall[tall[:, 1] > 0][:, 0].min() if (tall[:, 1] > 0).any() else 0
for name, t in (("runs of four", tall[:4096]), ("chunk by chunk", tall[4096:])):
    t = t[t[:, 1] > 0]
    if len(t) == 0:
        continue
    t0, t1 = t[:, 0].min(), t[:, 1].max()
    end = np.sort(t[:, 1] - t0)
    print(f"{which} W={W} {name}: {len(t)} waves, first start at {t0 - g0} ticks, span {t1 - t0} ticks (10 ns); wave end percentiles: " +
          " ".join(f"p{p}={np.percentile(end, p):.0f}" for p in (1, 10, 25, 50, 75, 90, 99, 100)))
    print(f"  mean end {end.mean():.0f} = {end.mean() / (t1 - t0) * 100:.1f}% of the span: a perfectly balanced queue would take ~{end.mean():.0f} ticks "
          f"({(1 - end.mean() / (t1 - t0)) * 100:.1f}% of the launch is tail)")

# the finish's last pass (its stats slice starts at word 9 of the buffer: 9 + 64 + 16384)
f = stats[9 + 64 + 16384: 9 + 64 + 16384 + 3 * 4096].cpu().numpy().reshape(-1, 3)
f = f[f[:, 1] > 0]
if len(f):
    f0, f1 = f[:, 0].min(), f[:, 1].max()
    fend = np.sort(f[:, 1] - f0)
    print(f"finish, last pass: {len(f)} waves, span {f1 - f0} ticks; wave end percentiles: " +
          " ".join(f"p{p}={np.percentile(fend, p):.0f}" for p in (1, 10, 25, 50, 75, 90, 99, 100)))
    print(f"  mean end {fend.mean():.0f} = {fend.mean() / (f1 - f0) * 100:.1f}% of the span; longest single item per wave: " +
          " ".join(f"p{p}={np.percentile(f[:, 2], p):.0f}" for p in (50, 90, 99, 100)) + " ticks")
    c = stats[9 + 64 + 16384 + 3 * 4096: 9 + 64 + 16384 + 4 * 4096].cpu().numpy()[:len(f)]
    top = np.argsort(-f[:, 2])[:12]
    print("  longest items (ticks, rounds, leaf evaluations, node expansions, leaf tests mod 65536):")
    for i in top:
        v = int(c[i])
        print(f"    {f[i, 2]:8d}  {v >> 48:5d} {(v >> 32) & 0xffff:6d} {(v >> 16) & 0xffff:6d} {v & 0xffff:6d}")
