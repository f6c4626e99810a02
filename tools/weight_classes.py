import sys, torch, numpy as np
sys.path.insert(0, '.')
import flooder_amd as fa
from flooder_amd import _native, core
torch.manual_seed(42)
dev = torch.device('cuda:0')
for which in ("cfg2", "cfg3"):
    pts = (fa.generate_noisy_torus_points_3d(1_000_000, seed=42) if which == "cfg3" else torch.randn(1_000_000, 3)).to(dev)
    lms = fa.generate_landmarks(pts, 1000, start_idx=0)
    stree, simplices = core._build_complex(lms, 3)
    verts = lms[torch.as_tensor(simplices[3], device=dev)]
    verts = verts[torch.argsort(verts.mean(1)[:, 0])].contiguous()
    index = core.PointIndex(pts)
    lib = _native.load()
    S = verts.shape[0]
    wgt = torch.empty(S, dtype=torch.float32, device=dev)
    _native.check(lib.flooder_simplex_weight_f32(_native.ptr(index.nodes), index.n, index.dim, _native.ptr(verts), 4, S, _native.ptr(wgt), _native.current_stream_ptr(dev)), "w")
    w = wgt.cpu().numpy()
    print(which, "S", S, "weight quantiles", {q: float(np.percentile(w, q)) for q in (10, 25, 50, 75, 90, 95, 99, 100)})
    print("  class counts (x limit 2000: >32,16,8,4,2,1,.5,.25,rest)", [int(((w > 2000 * t) & (w <= 2000 * t2)).sum()) for t, t2 in zip((32, 16, 8, 4, 2, 1, .5, .25, 0), (1e9, 32, 16, 8, 4, 2, 1, .5, .25))])
    if which == "cfg2":
        w8 = w[0::8]
        print("  long items' simplices (W=8):", {i: float(w8[i]) for i in (743, 331, 297, 183, 356, 519, 721, 736, 595, 133)})
