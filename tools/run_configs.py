"""End-to-end flood_complex timings on the BASELINE.json configurations (GPU)."""
import sys, time, json, numpy as np, torch
sys.path.insert(0, '.')
import flooder_amd as fa
from flooder_amd import core
dev = torch.device('cuda:0')

def torus(n):
    theta = torch.rand(n) * 2 * torch.pi; phi = torch.rand(n) * 2 * torch.pi
    p = torch.stack(((3 + torch.cos(phi)) * torch.cos(theta), (3 + torch.cos(phi)) * torch.sin(theta), torch.sin(phi)), 1)
    return p + torch.randn_like(p) * 0.02

def cheese(n, k=6, seed=0):
    g = torch.Generator().manual_seed(seed)
    centres = torch.empty((0, 3)); radii = torch.empty((0,))
    while centres.shape[0] < k:
        c = 0.2 + 0.6 * torch.rand(8, 3, generator=g); r = 0.1 + 0.1 * torch.rand(8, generator=g)
        for ci, ri in zip(c, r):
            if centres.shape[0] < k and (centres.shape[0] == 0 or ((centres - ci).norm(dim=1) >= radii + ri).all()):
                centres = torch.cat([centres, ci[None]]); radii = torch.cat([radii, ri[None]])
    pts = torch.empty((0, 3))
    while pts.shape[0] < n:
        cand = torch.rand(4 * (n - pts.shape[0]) + 1000, 3, generator=g)
        good = (torch.cdist(cand, centres) >= radii[None]).all(1)
        pts = torch.cat([pts, cand[good]])[:n]
    return pts

def run(name, pts, n_lms, **kw):
    tp = pts.to(dev)
    t0 = time.perf_counter(); lms = fa.generate_landmarks(tp, n_lms, start_idx=0); torch.cuda.synchronize(); t_fps = time.perf_counter() - t0
    fa.flood_complex(tp[:10000], lms, **kw); torch.cuda.synchronize()  # warm-up as examples/example_01
    t0 = time.perf_counter(); st = fa.flood_complex(tp, lms, return_simplex_tree=True, **kw); torch.cuda.synchronize(); t_fc = time.perf_counter() - t0
    vals = np.concatenate([st.filtrations_of_dimension(d) for d in range(st.dimension() + 1)])
    print(json.dumps(dict(config=name, n=len(pts), dim=pts.shape[1], landmarks=n_lms, simplices=st.num_simplices(),
                          fps_s=round(t_fps, 4), flood_complex_s=round(t_fc, 4), finite=bool(np.isfinite(vals).all()),
                          max_filtration=float(np.nanmax(vals)), **{k: v for k, v in kw.items()})), flush=True)

torch.manual_seed(42)
which = sys.argv[1:] or ["cfg2", "cfg3"]
if "cfg2" in which: run("cfg2 1M gaussian 3D", torch.randn(1_000_000, 3), 1000)
if "cfg3" in which: run("cfg3 1M torus 3D", torus(1_000_000), 1000)
if "cfg5" in which: run("cfg5 16M cheese 3D", cheese(16_000_000), 4000)
if "cfg4s" in which: run("cfg4-small 200k gaussian 6D", torch.randn(200_000, 6), 300, max_dimension=2, points_per_edge=8)
if "fig8" in which:
    t = torch.rand(2_000_000) * 2 * torch.pi
    run("2M figure-eight 2D", torch.stack((torch.sin(t), torch.sin(t) * torch.cos(t)), 1) + 0.01 * torch.randn(2_000_000, 2), 1000)
if "cfg4" in which:  # BASELINE cfg 4: 2M-point 6D Gaussian, 2k landmarks; max_dimension=2, ppe 8
    run("cfg4 2M gaussian 6D", torch.randn(2_000_000, 6), 2000, max_dimension=2, points_per_edge=8)
