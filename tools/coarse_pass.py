"""How long does the exact nearest neighbour of a coarse sub-lattice of every simplex take through the sorted-sample
tree sweep?  usage: python tools/coarse_pass.py cfg3|cfg5|cfg2"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import flooder_amd as fa
from flooder_amd import core, _native
wl = sys.argv[1]
dev = torch.device("cuda:0"); lib = _native.load()
torch.manual_seed(42)
if wl == "cfg3":
    pts, n_l = fa.generate_noisy_torus_points_3d(1_000_000, seed=42), 1000
elif wl == "cfg5":
    pts, n_l = fa.generate_swiss_cheese_points(16_000_000, k=6, seed=42)[0], 4000
else:
    pts, n_l = torch.randn(1_000_000, 3), 1000
pts = pts.to(dev).float().contiguous()
lms = fa.generate_landmarks(pts, n_l, start_idx=0)
stree, simplices = core._build_complex(lms, 3)
verts = lms[torch.as_tensor(simplices[3], device=dev)].contiguous()
weights, vi, fi = core.generate_grid(30, 3, dev, torch.float32)
faces = core._FaceTable(fi, weights.shape[0], dev)
plan = core.SamplePlan(weights, faces)
index = core.PointIndex(pts)
S = verts.shape[0]
rows, par, nco = plan.wit
wc = plan.w_perm[rows[:nco].long()].contiguous()
n_s = S * nco
st = _native.current_stream_ptr(dev)
def run():
    keys = torch.empty(n_s, dtype=torch.int32, device=dev); ks = torch.empty_like(keys); order = torch.empty_like(keys)
    tb = int(lib.flooder_index_sort_bytes(n_s)); tmp = torch.empty(tb, dtype=torch.uint8, device=dev)
    d2 = torch.empty((S, nco), dtype=torch.int32, device=dev)
    q = torch.zeros(core.QUEUE_WORDS, dtype=torch.int32, device=dev)
    _native.check(lib.flooder_sample_keys_f32(_native.ptr(verts), _native.ptr(wc), 4, nco, S, 3, _native.ptr(index.box), _native.ptr(keys), st), "keys")
    _native.check(lib.flooder_index_sort(_native.ptr(keys), n_s, int(lib.flooder_sample_key_bits(3)), _native.ptr(ks), _native.ptr(order), _native.ptr(tmp), tb, st), "sort")
    _native.check(lib.flooder_sweep_bvh_sorted_f32(_native.ptr(index.pts), index.n, 3, _native.ptr(index.nodes), _native.ptr(verts), _native.ptr(wc), 4, nco, S, _native.ptr(order), _native.ptr(q), _native.ptr(d2), None, st), "sweep")
    return d2
for _ in range(2): run()
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(11)]
ev[0].record()
for i in range(10):
    d2 = run(); ev[i + 1].record()
torch.cuda.synchronize()
print(wl, "S", S, "coarse samples", n_s, "coarse pass ms", np.mean([ev[i].elapsed_time(ev[i + 1]) for i in range(10)]))
d = d2.view(torch.float32).sqrt()
print("max d per simplex: mean", float(d.amax(1).mean()), "lattice step ~", float((verts.amax(1) - verts.amin(1)).amax(1).mean() / 29))
