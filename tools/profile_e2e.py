"""End-to-end flood_complex on one MI355X: wall time and the host-side profile.  usage: profile_e2e.py [cfg2|cfg3|cfg4|cfg5]"""
import cProfile, pstats, sys, time, io
import numpy as np, torch
sys.path.insert(0, '.')
import flooder_amd as fa
which = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
dev = torch.device('cuda:0')
torch.manual_seed(42)
kw = {}
if which == "cfg2":
    pts, k = torch.randn(1_000_000, 3), 1000
elif which == "cfg3":
    pts, k = fa.generate_noisy_torus_points_3d(1_000_000, seed=42), 1000
elif which == "cfg5":
    pts, k = fa.generate_swiss_cheese_points(16_000_000, k=6, seed=42)[0], 4000
else:
    pts, k, kw = torch.randn(2_000_000, 6), 2000, dict(max_dimension=2, points_per_edge=8)
tp = pts.to(dev)
lms = fa.generate_landmarks(tp, k, start_idx=0)
fa.flood_complex(tp[:10000], lms, **kw); torch.cuda.synchronize()           # warm-up as examples/example_01
for mode, extra in (("dict", {}), ("simplex tree", dict(return_simplex_tree=True))):
    ts = []
    for _ in range(3 if which != "cfg4" else 1):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = fa.flood_complex(tp, lms, **kw, **extra); torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    print(f"{which} flood_complex -> {mode}: {min(ts) * 1e3:.2f} ms (best of {len(ts)})", flush=True)
pr = cProfile.Profile(); pr.enable()
out = fa.flood_complex(tp, lms, **kw, **(dict(return_simplex_tree=True) if 'tree' in sys.argv else {})); torch.cuda.synchronize()
pr.disable()
buf = io.StringIO(); pstats.Stats(pr, stream=buf).sort_stats('cumulative').print_stats(28); print(buf.getvalue()[:6000])
