"""Per-simplex records of the witness sweep (library built with -DFLOODER_WIT_TIMERS): what do the simplices it handles
well / abandons look like?  usage: python tools/wit_items.py cfg2|cfg3 [option=value ...]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import flooder_amd as fa
from flooder_amd import core, _native
wl = sys.argv[1]
dev = torch.device("cuda:0")
lib = _native.load()
for o in sys.argv[2:]:
    k, v = o.split("=")
    assert lib.flooder_set_option(k.encode(), int(v)) == 0
torch.manual_seed(42)
if wl == "cfg2":
    pts, n_l = torch.randn(1_000_000, 3), 1000
elif wl == "cfg3":
    from oracle import flood_oracle as fo
    pts, n_l = torch.as_tensor(fo.noisy_torus(1_000_000, seed=42)), 1000
else:
    from flooder_amd import synthetic
    pts, n_l = synthetic.generate_swiss_cheese_points(4_000_000, torch.tensor([0., 0, 0]), torch.tensor([1., 1, 1]), 6, (0.1, 0.2), seed=42)[0], 2000
pts = pts.to(dev).float().contiguous()
lms = fa.generate_landmarks(pts, n_l, start_idx=0)
stree, simplices = core._build_complex(lms, 3)
verts = lms[torch.as_tensor(simplices[3], device=dev)].contiguous()
weights, vi, fi = core.generate_grid(30, 3, dev, torch.float32)
faces = core._FaceTable(fi, weights.shape[0], dev)
S = verts.shape[0]
st = torch.zeros(96 + 12 * S, dtype=torch.int64, device=dev)
core._sweep_dimension_cell(core.PointIndex(pts), verts, weights, faces, None, stats=st)
torch.cuda.synchronize()
rec = st[80:80 + 12 * S].cpu().numpy().reshape(S, 12)   # (the sweep hands stats[16:] to the witness kernel)
w = rec[:, 0].astype(np.uint32).view(np.float32)
tried = rec[:, 10] != 0
print(wl, "S", S, "tried", tried.sum(), "handled", (rec[:, 10] == 1).sum())
names = ["weight", "leaves", "n_in", "n_stage", "bins", "open_coarse", "live", "tiles", "-", "-", "status", "ticks(10ns)"]
for status, lab in [(1, "handled"), (102, "abandon: over/live"), (103, "abandon: dense/open/focus")]:
    sel = rec[:, 10] == status
    if sel.sum() == 0:
        continue
    print(f"== {lab}: {sel.sum()} items, total {rec[sel, 11].sum() / 100:.0f} us of workgroup time, mean {rec[sel, 11].mean() / 100:.1f} us, max {rec[sel, 11].max() / 100:.1f} us")
    for j, nm in [(1, "leaves"), (2, "n_in"), (3, "n_stage"), (4, "bins"), (5, "open_coarse"), (6, "live")]:
        v = rec[sel, j]
        print(f"   {nm:12s} mean {v.mean():8.1f}  p10 {np.percentile(v, 10):7.0f} p50 {np.percentile(v, 50):7.0f} p90 {np.percentile(v, 90):7.0f} max {v.max():7.0f}")
    print(f"   weight       mean {w[sel].mean():8.1f}  p10 {np.percentile(w[sel], 10):7.0f} p50 {np.percentile(w[sel], 50):7.0f} p90 {np.percentile(w[sel], 90):7.0f}")
h = rec[:, 10] == 1
if h.sum() > 10:
    t = rec[h, 11] / 100.0
    for j, nm in [(1, "leaves"), (3, "n_stage"), (6, "live"), (5, "open_coarse")]:
        print(f"corr(time, {nm}) = {np.corrcoef(t, rec[h, j])[0, 1]:.2f}")
    order = np.argsort(-t)[:8]
    print("slowest handled:", [(round(t[i], 1), int(rec[h][i, 1]), int(rec[h][i, 3]), int(rec[h][i, 5]), int(rec[h][i, 6])) for i in order], "(us, leaves, n_stage, open, live)")
