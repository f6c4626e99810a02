"""Wall time of whole flood_complex calls at cfg 2 (1 M Gaussian points, 1000 landmarks: with the landmark selection, with given
landmarks -> simplex tree / -> dict) and a cProfile of the host side of five calls.  Run on the GPU box from the repo root."""
import sys, time, cProfile, pstats, io
import torch
sys.path.insert(0, ".")
import flooder_amd as fa
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
pts = torch.randn(1_000_000, 3, generator=g).to(dev)
for _ in range(3):
    fa.flood_complex(pts, 1000, return_simplex_tree=True)
torch.cuda.synchronize()
ts = []
for _ in range(5):
    t0 = time.perf_counter(); st = fa.flood_complex(pts, 1000, return_simplex_tree=True); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
print("flood_complex(points, 1000) -> tree, ms:", [round(t * 1e3, 2) for t in ts])
lm = fa.generate_landmarks(pts, 1000, start_idx=0)
ts = []
for _ in range(5):
    t0 = time.perf_counter(); st = fa.flood_complex(pts, lm, return_simplex_tree=True); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
print("flood_complex(points, landmarks) -> tree, ms:", [round(t * 1e3, 2) for t in ts])
ts = []
for _ in range(3):
    t0 = time.perf_counter(); d = fa.flood_complex(pts, lm); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
print("flood_complex(points, landmarks) -> dict, ms:", [round(t * 1e3, 2) for t in ts])
pr = cProfile.Profile(); pr.enable()
for _ in range(5):
    st = fa.flood_complex(pts, lm, return_simplex_tree=True)
torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28); print(s.getvalue()[:6000])
