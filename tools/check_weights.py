"""Diagnostic: flooder_simplex_weight_f32 against true point counts in the simplex boxes.  usage: check_weights.py [cfg2|cfg3|cfg5]"""
import sys, torch, numpy as np
sys.path.insert(0, '.')
import flooder_amd as fa
from flooder_amd import _native, core
which = sys.argv[1] if len(sys.argv) > 1 else "cfg5"
dev = torch.device('cuda:0')
torch.manual_seed(42)
if which == "cfg5":
    pts, k = fa.generate_swiss_cheese_points(16_000_000, k=6, seed=42)[0], 4000
elif which == "cfg3":
    pts, k = fa.generate_noisy_torus_points_3d(1_000_000, seed=42), 1000
else:
    pts, k = torch.randn(1_000_000, 3), 1000
tp = pts.to(dev)
index = core.PointIndex(tp)
lms = fa.generate_landmarks(tp, k, start_idx=0, index=index)
_, simplices = core._build_complex(lms, 3)
verts = lms[torch.as_tensor(simplices[3], device=dev)].contiguous()
lib = _native.load()
S = verts.shape[0]
w = torch.empty(S, device=dev)
_native.check(lib.flooder_simplex_weight_f32(_native.ptr(index.nodes), index.n, 3, _native.ptr(verts), 4, S, _native.ptr(w), 0), "w")
torch.cuda.synchronize()
wh = w.cpu().numpy()
print(which, "S", S, "weight percentiles", np.percentile(wh, [0, 10, 25, 50, 75, 90, 99, 100]).round(1), 'share below 250/500/1000/2000:', [(wh < t).mean().round(3) for t in (250, 500, 1000, 2000)])
lo, hi = verts.min(1).values, verts.max(1).values
for i in np.random.default_rng(0).choice(S, 8, replace=False):
    true = int(((tp >= lo[i]) & (tp <= hi[i])).all(1).sum())
    print(f"  simplex {i}: estimate {wh[i]:.0f} true {true}")
