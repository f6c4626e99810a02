"""Diagnostic: how much of an end-to-end flood_complex call is device sweep (any method)?
usage: python tools/sweep_share.py cfg4s|cfg4|cfg2|cfg3"""
import sys, time, torch
sys.path.insert(0, '.')
import flooder_amd as fa
from flooder_amd import core
dev = torch.device('cuda:0')
which = sys.argv[1] if len(sys.argv) > 1 else "cfg4s"
torch.manual_seed(42)
cfg = {"cfg4s": (torch.randn(200_000, 6), 300, dict(max_dimension=2, points_per_edge=8)),
       "cfg4": (torch.randn(2_000_000, 6), 2000, dict(max_dimension=2, points_per_edge=8)),
       "cfg2": (torch.randn(1_000_000, 3), 1000, {}),
       "cfg3": (fa.generate_noisy_torus_points_3d(1_000_000, seed=42), 1000, {})}[which]
pts, n_lms, kw = cfg
acc = {}
def wrap(name):
    fn = getattr(core, name)
    def inner(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = fn(*a, **k)
        torch.cuda.synchronize(); acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
        return r
    setattr(core, name, inner)
for n in ("_sweep_dimension_cell", "_sweep_dimension_bvh", "_sweep_dimension_hip", "_build_complex"):
    wrap(n)
tp = pts.to(dev)
lms = fa.generate_landmarks(tp, n_lms, start_idx=0)
fa.flood_complex(tp[:10000], lms, **kw); acc.clear()
torch.cuda.synchronize(); t0 = time.perf_counter()
out = fa.flood_complex(tp, lms, **kw)
torch.cuda.synchronize(); total = time.perf_counter() - t0
print(which, "simplices", len(out), "total s", round(total, 4), {k: round(v, 4) for k, v in acc.items()})
